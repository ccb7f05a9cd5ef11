// xcd_barrier_probe.hip -- cost of a GROUP barrier among workgroups that sit on ONE XCD (measurement tooling).
// The decode step's wq|wk|wv -> attention seam is per head: head h's scores need only head h's q / k / v rows, its
// soft_max . V only its scores.  If the workgroups of a head share an XCD they share an L2, and a hand-off can stay
// inside it: atomics WITHOUT the device-scope bit execute in that L2 (device-scope ones go to the memory side, ~2 us per
// dependent round trip across the 8 XCDs), polls and data reads bypass the per-CU L1 (sc1) and hit the L2.
// Grid: 8 XCDs x NG groups x GS workgroups, workgroup b -> XCD b % 8 (round-robin dispatch; checked with XCC_ID).
// Each round: every workgroup publishes one value per lane, barrier, reads the values of the next member of its group,
// counts stale reads.  Variants:
//   0  L2-local: atomic add without scope bits, poll = sc1 load, data = sc1 loads
//   1  device scope: agent-scope atomic add + agent acquire/release fences (what any cross-XCD hand-off needs)
//   2  L2-local arrive, poll by atomic RMW (fetch_add 0) instead of a load
// build: hipcc --offload-arch=gfx950 -O3 tools/xcd_barrier_probe.hip -o tools/xcd_barrier_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ inline uint32_t load_sc1(const uint32_t *p) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int VAR>
__global__ void __launch_bounds__(256) k_rounds(uint32_t *counters, uint32_t *data, uint32_t *xcc, uint32_t *stale, uint32_t *timeout,
                                                int ng, int gs, int rounds, uint32_t epoch0) {
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3, grp = slot / gs, mem = slot % gs;
    const int gid = xcd * ng + grp;
    uint32_t *cnt = counters + gid * 32;                   // one 128-byte line per group
    uint32_t *mine = data + (size_t) b * 256, *next = data + (size_t) (((slot - mem + (mem + 1) % gs) << 3) | xcd) * 256;
    if (threadIdx.x == 0) {
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[b] = id & 0xf;
    }
    uint32_t bad = 0;
    __shared__ int s_abort;
    if (threadIdx.x == 0) s_abort = 0;
    __syncthreads();
    uint32_t *mine0 = mine, *next0 = next;
    for (int r = 1; r <= rounds; r++) {
        const uint32_t tag = epoch0 + r;
        mine = mine0 + (r & 1) * (2048 * 256); next = next0 + (r & 1) * (2048 * 256);     // (double-buffered: a fast member's round r + 1 must not overwrite what a slow one still reads)
        if (VAR == 1) {
            mine[threadIdx.x] = tag;
            __syncthreads();
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // L2 write-back of this XCD
                __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t) (r * gs)) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 18)) { timeout[0] = 1; s_abort = 1; break; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            if (s_abort) return;
            if (next[threadIdx.x] != tag) bad++;
        } else {
            __builtin_nontemporal_store(tag, mine + threadIdx.x);       // (any store reaches the L2: the L1 is write-through)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);     // no scope bits: executes in this XCD's L2
                int spins = 0;
                for (;;) {
                    const uint32_t v = VAR == 2 ? __hip_atomic_fetch_add(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : load_sc1(cnt);
                    if (v >= (uint32_t) (r * gs)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 18)) { timeout[0] = 1; s_abort = 1; break; }
                }
            }
            __syncthreads();
            if (s_abort) return;
            if (load_sc1(next + threadIdx.x) != tag) bad++;
        }
    }
    if (bad) atomicAdd(stale, bad);
}

// data-tagged hand-off: every lane publishes one 8-byte granule {value, tag}; the reader polls the granule itself.
//   CROSS = false: the reader is the next member of the group (same XCD): plain 8-byte store, sc1 load
//   CROSS = true : the reader is workgroup b + 1 (the NEXT XCD): write-through (agent-scope) store, sc1 load
// NREAD granules per lane are read (1: a neighbour's 2 KB; 4: 8 KB = what a mat-vec workgroup pulls of a quantized activation row)
template <bool CROSS, int NREAD>
__global__ void __launch_bounds__(256) k_tagged(unsigned long long *data, uint32_t *stale, uint32_t *timeout, int gs, int rounds, uint32_t epoch0) {
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3, mem = slot % gs, nb = gridDim.x;
    __shared__ int s_abort;
    if (threadIdx.x == 0) s_abort = 0;
    __syncthreads();
    uint32_t bad = 0;
    for (int r = 1; r <= rounds; r++) {
        const uint32_t tag = epoch0 + r;
        unsigned long long *mine = data + ((size_t) (r & 1) * 2048 + b) * 256;
        const unsigned long long v = ((unsigned long long) tag << 32) | (uint32_t) (b * 256 + threadIdx.x);
        if (CROSS) __hip_atomic_store(mine + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(mine + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int k = 0; k < NREAD; k++) {
            const int src = CROSS ? (b + 1 + k * 8 + k) % nb : (((slot - mem + (mem + 1 + k) % gs) << 3) | xcd);
            const unsigned long long *theirs = data + ((size_t) (r & 1) * 2048 + src) * 256;
            int spins = 0;
            for (;;) {
                const unsigned long long w = __hip_atomic_load(theirs + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t) (w >> 32) == tag) { if ((uint32_t) w != (uint32_t) (src * 256 + threadIdx.x)) bad++; break; }
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 18)) { timeout[0] = 1; s_abort = 1; break; }
            }
        }
        __syncthreads();          // (the round's reads are done before this workgroup overwrites the other buffer's slot two rounds later)
        if (s_abort) return;
    }
    if (bad) atomicAdd(stale, bad);
}

__global__ void __launch_bounds__(256) k_one(uint32_t *data, uint32_t *stale, int gs, uint32_t tag, int phase) {
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3, mem = slot % gs;
    uint32_t *mine = data + (size_t) b * 256, *next = data + (size_t) (((slot - mem + (mem + 1) % gs) << 3) | xcd) * 256;
    if (phase == 0) mine[threadIdx.x] = tag;
    else if (next[threadIdx.x] != tag) atomicAdd(stale, 1u);
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    uint32_t *d_cnt, *d_data, *d_xcc, *d_stale, *d_to;
    CHECK(hipMalloc((void **) &d_cnt, 8 * 64 * 128)); CHECK(hipMalloc((void **) &d_data, 2 * 2048 * 256 * 4));
    CHECK(hipMalloc((void **) &d_xcc, 2048 * 4)); CHECK(hipMalloc((void **) &d_stale, 4)); CHECK(hipMalloc((void **) &d_to, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int shapes[][2] = { { 4, 16 }, { 4, 8 }, { 2, 32 }, { 4, 12 }, { 8, 8 }, { 1, 32 }, { 4, 4 } };     // groups per XCD, workgroups per group
    for (auto &sh : shapes) {
        const int ng = sh[0], gs = sh[1], grid = 8 * ng * gs;
        printf("-- %d groups of %d workgroups per XCD (%d workgroups of 256 threads, %d per CU)\n", ng, gs, grid, (grid + 255) / 256);
        for (int var = 0; var < 3; var++) {
            CHECK(hipMemset(d_cnt, 0, 8 * 64 * 128)); CHECK(hipMemset(d_stale, 0, 4)); CHECK(hipMemset(d_to, 0, 4));
            auto launch = [&](int r, uint32_t ep) {
                if (var == 0) hipLaunchKernelGGL(k_rounds<0>, dim3(grid), dim3(256), 0, 0, d_cnt, d_data, d_xcc, d_stale, d_to, ng, gs, r, ep);
                if (var == 1) hipLaunchKernelGGL(k_rounds<1>, dim3(grid), dim3(256), 0, 0, d_cnt, d_data, d_xcc, d_stale, d_to, ng, gs, r, ep);
                if (var == 2) hipLaunchKernelGGL(k_rounds<2>, dim3(grid), dim3(256), 0, 0, d_cnt, d_data, d_xcc, d_stale, d_to, ng, gs, r, ep);
            };
            launch(10, 1000); CHECK(hipDeviceSynchronize());
            CHECK(hipMemset(d_cnt, 0, 8 * 64 * 128));
            CHECK(hipEventRecord(e0, 0)); launch(rounds, 5000); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            uint32_t stale, to; std::vector<uint32_t> xcc(grid);
            CHECK(hipMemcpy(&stale, d_stale, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&to, d_to, 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(xcc.data(), d_xcc, grid * 4, hipMemcpyDeviceToHost));
            int mism = 0; for (int b = 0; b < grid; b++) if (xcc[b] != xcc[b & 7]) mism++;
            const char *names[] = { "L2-local atomics, sc1 poll + sc1 data", "device-scope atomics + fences", "L2-local atomics, RMW poll + sc1 data" };
            printf("   %-42s %7.3f us per round   stale reads %u   timeouts %u   workgroups off their group's XCD %d (XCC ids of blocks 0-7: %u %u %u %u %u %u %u %u)\n",
                   names[var], ms * 1e3 / rounds, stale, to, mism, xcc[0], xcc[1], xcc[2], xcc[3], xcc[4], xcc[5], xcc[6], xcc[7]);
        }
        {
            unsigned long long *d_tag; CHECK(hipMalloc((void **) &d_tag, (size_t) 2 * 2048 * 256 * 8)); CHECK(hipMemset(d_tag, 0, (size_t) 2 * 2048 * 256 * 8));
            for (int var = 0; var < 4; var++) {
                CHECK(hipMemset(d_stale, 0, 4)); CHECK(hipMemset(d_to, 0, 4));
                auto launch = [&](int r, uint32_t ep) {
                    if (var == 0) hipLaunchKernelGGL((k_tagged<false, 1>), dim3(grid), dim3(256), 0, 0, d_tag, d_stale, d_to, gs, r, ep);
                    if (var == 1) hipLaunchKernelGGL((k_tagged<false, 4>), dim3(grid), dim3(256), 0, 0, d_tag, d_stale, d_to, gs, r, ep);
                    if (var == 2) hipLaunchKernelGGL((k_tagged<true, 1>), dim3(grid), dim3(256), 0, 0, d_tag, d_stale, d_to, gs, r, ep);
                    if (var == 3) hipLaunchKernelGGL((k_tagged<true, 4>), dim3(grid), dim3(256), 0, 0, d_tag, d_stale, d_to, gs, r, ep);
                };
                launch(10, 1000 + var * 100000); CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0, 0)); launch(rounds, 5000 + var * 100000); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                uint32_t stale, to; CHECK(hipMemcpy(&stale, d_stale, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&to, d_to, 4, hipMemcpyDeviceToHost));
                const char *names[] = { "tagged granules, same XCD, 2 KB read", "tagged granules, same XCD, 8 KB read", "tagged granules, OTHER XCDs, 2 KB read", "tagged granules, OTHER XCDs, 8 KB read" };
                printf("   %-42s %7.3f us per round   wrong values %u   timeouts %u\n", names[var], ms * 1e3 / rounds, stale, to);
            }
            CHECK(hipFree(d_tag));
        }
        // the same dependency as two kernel launches per round
        CHECK(hipMemset(d_stale, 0, 4));
        const int lr = 200;
        for (int r = 0; r < 3; r++) { hipLaunchKernelGGL(k_one, dim3(grid), dim3(256), 0, 0, d_data, d_stale, gs, 7u + r, 0); hipLaunchKernelGGL(k_one, dim3(grid), dim3(256), 0, 0, d_data, d_stale, gs, 7u + r, 1); }
        CHECK(hipDeviceSynchronize()); CHECK(hipMemset(d_stale, 0, 4));
        CHECK(hipEventRecord(e0, 0));
        for (int r = 0; r < lr; r++) { hipLaunchKernelGGL(k_one, dim3(grid), dim3(256), 0, 0, d_data, d_stale, gs, 100u + r, 0); hipLaunchKernelGGL(k_one, dim3(grid), dim3(256), 0, 0, d_data, d_stale, gs, 100u + r, 1); }
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        uint32_t stale; CHECK(hipMemcpy(&stale, d_stale, 4, hipMemcpyDeviceToHost));
        printf("   %-42s %7.3f us per launch (two launches carry one dependency)   stale reads %u\n", "kernel boundary (eager launches)", ms * 1e3 / (2 * lr), stale);
    }
    return 0;
}
