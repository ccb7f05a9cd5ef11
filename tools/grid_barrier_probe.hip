// Round-2 groundwork: what does a grid-wide hand-off cost on MI355X when every
// workgroup is co-resident?  N workgroups x 256 threads run R rounds of: a little work -> release
// (atomic add on a device-scope counter) -> bounded acquire spin until all N arrived.  Prints microseconds
// per round for several N, next to the time of R empty kernel launches for comparison.
//   build: hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o gpurun_out/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_rounds(unsigned *counter, int rounds, unsigned *fail, float *sink) {
    const unsigned n = gridDim.x;
    float acc = threadIdx.x;
    for (int r = 0; r < rounds; r++) {
        for (int i = 0; i < 64; i++) acc = acc * 1.0001f + 0.5f;              // stand-in for the phase's tail
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = n * (unsigned) (r + 1);
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1 << 22)) { *fail = 1; break; }               // give up instead of hanging the box
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    if (acc == 12345.678f) sink[0] = acc;
}
// variant: the last arriver publishes the epoch in a separate flag word; everybody else polls that word
// (read-only line) instead of the contended counter.  SLEEP = 0: busy poll.
template <int SLEEP>
__global__ void __launch_bounds__(256) k_rounds_flag(unsigned *counter, unsigned *flag, int rounds, unsigned *fail, float *sink) {
    const unsigned n = gridDim.x;
    float acc = threadIdx.x;
    for (int r = 0; r < rounds; r++) {
        for (int i = 0; i < 64; i++) acc = acc * 1.0001f + 0.5f;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned epoch = (unsigned) (r + 1);
            const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == n * epoch) {
                __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                int spins = 0;
                while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
                    if (++spins > (1 << 22)) { *fail = 1; break; }
                    if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
                }
            }
        }
        __syncthreads();
    }
    if (acc == 12345.678f) sink[0] = acc;
}
// variant: two levels -- 8 group counters (workgroup id & 7 ~ the XCD), the last arriver of a group bumps the top
// counter, the last of those publishes the flag
__global__ void __launch_bounds__(256) k_rounds_2lvl(unsigned *counters /*[16 * 32]*/, unsigned *flag, int rounds, unsigned *fail, float *sink) {
    const unsigned n = gridDim.x, grp = blockIdx.x & 7, ngrp = n < 8 ? n : 8, in_grp = (n + 7 - grp) / 8;
    float acc = threadIdx.x;
    for (int r = 0; r < rounds; r++) {
        for (int i = 0; i < 64; i++) acc = acc * 1.0001f + 0.5f;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned epoch = (unsigned) (r + 1);
            bool last = false;
            const unsigned old = __hip_atomic_fetch_add(counters + 32 * grp, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == in_grp * epoch) {
                const unsigned o2 = __hip_atomic_fetch_add(counters + 32 * 8, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                last = o2 + 1 == ngrp * epoch;
            }
            if (last) {
                __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                int spins = 0;
                while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
                    if (++spins > (1 << 22)) { *fail = 1; break; }
                }
            }
        }
        __syncthreads();
    }
    if (acc == 12345.678f) sink[0] = acc;
}
// variant: no read-modify-write at all -- workgroup i release-stores the epoch into its own slot, and thread t
// of every workgroup polls slot t (n <= 512: at most two slots per thread); the arrivals do not serialise
__global__ void __launch_bounds__(256) k_rounds_slots(unsigned *slots /*[512 * 16]: one 64-byte line each*/, int rounds, unsigned *fail, float *sink) {
    const unsigned n = gridDim.x;
    float acc = threadIdx.x;
    for (int r = 0; r < rounds; r++) {
        for (int i = 0; i < 64; i++) acc = acc * 1.0001f + 0.5f;
        __syncthreads();
        const unsigned epoch = (unsigned) (r + 1);
        if (threadIdx.x == 0) __hip_atomic_store(slots + 16 * blockIdx.x, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        for (unsigned t = threadIdx.x; t < n; t += 256) {
            int spins = 0;
            while (__hip_atomic_load(slots + 16 * t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch)
                if (++spins > (1 << 22)) { *fail = 1; break; }
        }
        __syncthreads();
    }
    if (acc == 12345.678f) sink[0] = acc;
}
// variant: LOCAL hand-offs -- groups of G workgroups synchronise among themselves only (a head's 12 producer
// workgroups and its consumers in a fused wq|wk|wv + attention kernel): one counter per group, 64 bytes apart
template <int G>
__global__ void __launch_bounds__(256) k_rounds_groups(unsigned *counters /*[64 * 16]*/, int rounds, unsigned *fail, float *sink) {
    const unsigned grp = blockIdx.x / G, members = min((unsigned) G, gridDim.x - grp * G);
    float acc = threadIdx.x;
    for (int r = 0; r < rounds; r++) {
        for (int i = 0; i < 64; i++) acc = acc * 1.0001f + 0.5f;
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counters + 16 * grp, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = members * (unsigned) (r + 1);
            int spins = 0;
            while (__hip_atomic_load(counters + 16 * grp, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target)
                if (++spins > (1 << 22)) { *fail = 1; break; }
        }
        __syncthreads();
    }
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ void k_empty(float *sink) { if (threadIdx.x == 9999) sink[0] = 1.0f; }

int main() {
    unsigned *counter, *fail;
    float *sink;
    CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&fail, 4)); CHECK(hipMalloc(&sink, 4));
    const int rounds = 2000;
    for (int n : { 64, 128, 256, 512 }) {
        CHECK(hipMemset(counter, 0, 4)); CHECK(hipMemset(fail, 0, 4));
        hipLaunchKernelGGL(k_rounds, dim3(n), dim3(256), 0, 0, counter, 10, fail, sink);     // warm-up
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemset(counter, 0, 4));
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_rounds, dim3(n), dim3(256), 0, 0, counter, rounds, fail, sink);
        CHECK(hipDeviceSynchronize());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        unsigned f = 0;
        CHECK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        printf("grid barrier, %3d workgroups x 256 threads: %.2f us per round%s\n", n, us / rounds, f ? "  (SPIN GAVE UP)" : "");
    }
    unsigned *cs, *flag;
    CHECK(hipMalloc(&cs, 16 * 32 * 4)); CHECK(hipMalloc(&flag, 4));
    for (int variant = 0; variant < 3; variant++)
        for (int n : { 128, 256, 512 }) {
            for (int pass = 0; pass < 2; pass++) {
                CHECK(hipMemset(cs, 0, 16 * 32 * 4)); CHECK(hipMemset(flag, 0, 4)); CHECK(hipMemset(fail, 0, 4));
                const int rr = pass ? rounds : 10;
                auto t0 = std::chrono::steady_clock::now();
                if (variant == 0) hipLaunchKernelGGL(k_rounds_flag<1>, dim3(n), dim3(256), 0, 0, cs, flag, rr, fail, sink);
                else if (variant == 1) hipLaunchKernelGGL(k_rounds_flag<0>, dim3(n), dim3(256), 0, 0, cs, flag, rr, fail, sink);
                else hipLaunchKernelGGL(k_rounds_2lvl, dim3(n), dim3(256), 0, 0, cs, flag, rr, fail, sink);
                CHECK(hipDeviceSynchronize());
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                unsigned f = 0;
                CHECK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
                if (pass) printf("%s, %3d workgroups: %.2f us per round%s\n", variant == 0 ? "counter + flag, s_sleep 1" : variant == 1 ? "counter + flag, busy poll " : "two-level counters + flag ", n, us / rr, f ? "  (SPIN GAVE UP)" : "");
            }
        }
    unsigned *slots;
    CHECK(hipMalloc(&slots, 512 * 16 * 4));
    for (int n : { 128, 256, 512 })
        for (int pass = 0; pass < 2; pass++) {
            CHECK(hipMemset(slots, 0, 512 * 16 * 4)); CHECK(hipMemset(fail, 0, 4));
            const int rr = pass ? rounds : 10;
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_rounds_slots, dim3(n), dim3(256), 0, 0, slots, rr, fail, sink);
            CHECK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            unsigned f = 0;
            CHECK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
            if (pass) printf("per-workgroup slots (stores + polls, no RMW), %3d workgroups: %.2f us per round%s\n", n, us / rr, f ? "  (SPIN GAVE UP)" : "");
        }
    unsigned *gc;
    CHECK(hipMalloc(&gc, 64 * 16 * 4));
    for (int G : { 4, 12, 32 })
        for (int pass = 0; pass < 2; pass++) {
            CHECK(hipMemset(gc, 0, 64 * 16 * 4)); CHECK(hipMemset(fail, 0, 4));
            const int rr = pass ? rounds : 10;
            auto t0 = std::chrono::steady_clock::now();
            if (G == 4) hipLaunchKernelGGL(k_rounds_groups<4>, dim3(256), dim3(256), 0, 0, gc, rr, fail, sink);
            else if (G == 12) hipLaunchKernelGGL(k_rounds_groups<12>, dim3(256), dim3(256), 0, 0, gc, rr, fail, sink);
            else hipLaunchKernelGGL(k_rounds_groups<32>, dim3(256), dim3(256), 0, 0, gc, rr, fail, sink);
            CHECK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            unsigned f = 0;
            CHECK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
            if (pass) printf("local hand-offs, 256 workgroups in groups of %2d: %.2f us per round%s\n", G, us / rr, f ? "  (SPIN GAVE UP)" : "");
        }
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    for (int i = 0; i < 100; i++) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, sink);
    CHECK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < rounds; i++) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, sink);
    CHECK(hipStreamSynchronize(st));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("empty kernel, 256 workgroups, back to back on one stream: %.2f us per launch\n", us / rounds);
    return 0;
}
