// pair_probe.hip -- is an IN-LAUNCH hand-off between two dependent decode mat-vecs cheaper than the kernel
// boundary it replaces?  (measurement tooling for DESIGN.md "decode launch structure")
//
// A = a mat-vec-shaped stream of nA row-groups that produces a small vector y; B = a stream of nB row-groups that
// cannot CONSUME before it has read all of y (the all-to-all seam of the decode step), but whose weight stream does
// not depend on y.  Three schedules of the pair, each captured 8x into a hipGraph and replayed:
//   serial : two launches, A then B                                (what the decode step does today)
//   fused  : ONE launch; blocks [0, gA) run A and finish with {write-through stores of y, vmcnt(0), one agent-scope
//            atomic add on one of 8 counters}; blocks [gA, gA + gB) put their first D chunks in flight, poll the 8
//            counters (one wave, relaxed agent-scope loads + s_sleep), read y with sc1 loads, then stream
//   fusedF : the same with one agent-scope acquire fence + plain loads of y instead of sc1 loads
// Every block is resident at once (checked from the occupancy API before running) -- the only way a waiting block
// cannot starve the block it waits for.  The fold of the stream into one dword per lane keeps the loads alive; B's
// result also folds y, and the run checks that every B block saw the y of ITS iteration (stale reads are counted).
// build: hipcc --offload-arch=gfx950 -O3 tools/pair_probe.hip -o tools/pair_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int TILE = 1280, NCNT = 8, YN = 4096;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Ring { u32x4 q; f32x2 s; };

template <int D>
__device__ __forceinline__ uint32_t stream_rows(const uint8_t *wbase, int nchunks, int lane, Ring (&r)[D], bool prefetched) {
#define LOADW(S, C) { const uint8_t *tp = wbase + (size_t) min((C), nchunks) * TILE; \
        r[S].q = __builtin_nontemporal_load((const u32x4 *) (tp + lane * 16)); \
        r[S].s = __builtin_nontemporal_load((const f32x2 *) (tp + 1024 + ((lane >> 3) * 8 + (lane & 3) * 2) * 4)); }
    if (!prefetched) {
#pragma unroll
        for (int i = 0; i < D; i++) LOADW(i, i)
    }
    uint32_t acc = 0;
    int c0 = 0;
    do {
#pragma unroll
        for (int i = 0; i < D; i++) {
            acc ^= r[i].q.x ^ r[i].q.y ^ r[i].q.z ^ r[i].q.w ^ __builtin_bit_cast(uint32_t, r[i].s.x) ^ __builtin_bit_cast(uint32_t, r[i].s.y);
            LOADW(i, c0 + D + i)
            __builtin_amdgcn_sched_barrier(0);
        }
        c0 += D;
    } while (c0 < nchunks);
    return acc;
}

struct Args {
    const uint8_t *wA, *wB; int ngA, ncA, nwA, ngB, ncB, nwB;
    uint32_t *y;            // YN dwords written by A (block b writes y[b * per .. ) with the iteration tag
    uint32_t *cnt;          // NCNT counters, 64 B apart
    uint32_t *out;          // B's per-block verdict: number of stale y words seen
    uint32_t tag;           // iteration tag (value every y word must carry)
    uint32_t target;        // counter sum that means "A of this iteration is complete"
};

// ---- role A: stream, then publish this block's slice of y
template <int D, bool SC1>
__device__ void role_a(const Args &a, int blk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blk * a.nwA + wave;
    Ring r[D];
    uint32_t acc = 0;
    if (wave < a.nwA && g < a.ngA) acc = stream_rows<D>(a.wA + (size_t) g * (a.ncA + 1) * TILE, a.ncA, lane, r, false);
    const int gridA = (a.ngA + a.nwA - 1) / a.nwA, per = (YN + gridA - 1) / gridA;
    const uint32_t v = a.tag + (acc == 0x12345678u);                      // (keeps the stream alive; never true)
    for (int i = threadIdx.x; i < per; i += blockDim.x) {
        const int idx = blk * per + i;
        if (idx < YN) {
            if (SC1) __hip_atomic_store(a.y + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // write-through
            else a.y[idx] = v;
        }
    }
    if (SC1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // every storing wave drains
        __syncthreads();
        if (threadIdx.x == 0) {
            // hierarchical fan-in (MI355X_MICROARCH "barrier-xcd" shape): shard counter -> top counter -> 8 go words.
            // Counters are monotonic over the ITER launches of a replay; `target` = blocks of A so far.
            const int sh = blk % NCNT, in_shard = (gridA - sh + NCNT - 1) / NCNT;                 // A blocks with blk % 8 == sh
            const uint32_t it1 = a.target / (uint32_t) gridA;                                       // iteration index + 1
            const uint32_t old = __hip_atomic_fetch_add(a.cnt + sh * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == (uint32_t) in_shard * it1) {
                const uint32_t t = __hip_atomic_fetch_add(a.cnt + NCNT * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t + 1 == (uint32_t) NCNT * it1)
                    for (int i = 0; i < NCNT; i++) __hip_atomic_store(a.cnt + (NCNT + 1 + i) * 16, it1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ---- role B: (FUSED: prefetch, wait for A) read y, stream
template <int D, int MODE /*0 serial, 1 fused + sc1 loads, 2 fused + acquire fence*/>
__device__ void role_b(const Args &a, int blk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blk * a.nwB + wave;
    const bool valid = wave < a.nwB && g < a.ngB;
    const uint8_t *wbase = a.wB + (size_t) (valid ? g : 0) * (a.ncB + 1) * TILE;
    Ring r[D];
    if (MODE != 0) {
#pragma unroll
        for (int i = 0; i < D; i++) {
            const uint8_t *tp = wbase + (size_t) min(i, a.ncB) * TILE;
            r[i].q = __builtin_nontemporal_load((const u32x4 *) (tp + lane * 16));
            r[i].s = __builtin_nontemporal_load((const f32x2 *) (tp + 1024 + ((lane >> 3) * 8 + (lane & 3) * 2) * 4));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (wave == 0) {
            const int gridA_ = (a.ngA + a.nwA - 1) / a.nwA;
            const uint32_t it1 = a.target / (uint32_t) gridA_;
            for (unsigned spins = 0;; spins++) {                           // ONE word per block, one lane, relaxed, with s_sleep
                const uint32_t c = __hip_atomic_load(a.cnt + (NCNT + 1 + blk % NCNT) * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_readfirstlane(c) >= it1 || spins > (1u << 20)) break;
                __builtin_amdgcn_s_sleep(8);
            }
            if (MODE == 2 && lane == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    // the "prologue": every block reads all of y (YN dwords, 16 per thread at 256 threads)
    uint32_t stale = 0;
    for (int i = threadIdx.x; i < YN; i += blockDim.x) {
        const uint32_t v = (MODE == 1) ? __hip_atomic_load(a.y + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.y[i];
        stale += v != a.tag;
    }
    uint32_t acc = valid ? stream_rows<D>(wbase, a.ncB, lane, r, MODE != 0) : 0u;
    stale += acc == 0x12345678u;
    for (int o = 32; o; o >>= 1) stale += __shfl_xor(stale, o);
    if (lane == 0 && stale) atomicAdd(a.out, stale);
}

template <int DA, int DB> __global__ void __launch_bounds__(512) k_a(Args a) { role_a<DA, false>(a, blockIdx.x); }
template <int DA, int DB> __global__ void __launch_bounds__(512) k_b(Args a) { role_b<DB, 0>(a, blockIdx.x); }
template <int DA, int DB, int MODE> __global__ void __launch_bounds__(512, 2) k_fused(Args a) {
    const int gridA = (a.ngA + a.nwA - 1) / a.nwA;
    if ((int) blockIdx.x < gridA) role_a<DA, true>(a, blockIdx.x);
    else role_b<DB, MODE>(a, blockIdx.x - gridA);
}

struct Pair { const char *name; int ngA, ncA, nwA, ngB, ncB, nwB; };

template <int DA, int DB>
void run_pair(const Pair &p, int reps, hipStream_t st) {
    const size_t bA = (size_t) p.ngA * (p.ncA + 1) * TILE, bB = (size_t) p.ngB * (p.ncB + 1) * TILE;
    const int NB = (int) ((400u << 20) / (bA + bB)) + 2, ITER = 8;
    uint8_t *wA, *wB; uint32_t *y, *cnt, *out;
    CHECK(hipMalloc((void **) &wA, bA * NB)); CHECK(hipMalloc((void **) &wB, bB * NB));
    CHECK(hipMemset(wA, 0x5a, bA * NB)); CHECK(hipMemset(wB, 0xa5, bB * NB));
    CHECK(hipMalloc((void **) &y, YN * 4)); CHECK(hipMalloc((void **) &cnt, (2 * NCNT + 1) * 64)); CHECK(hipMalloc((void **) &out, 4));
    const int gA = (p.ngA + p.nwA - 1) / p.nwA, gB = (p.ngB + p.nwB - 1) / p.nwB;
    const int bdim = 64 * (p.nwA > p.nwB ? p.nwA : p.nwB);
    int occ = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *) k_fused<DA, DB, 1>, bdim, 0));
    printf("-- %s: A %d x %d chunks (%d WGs of %d waves, %.1f MB), B %d x %d chunks (%d WGs of %d waves, %.1f MB); fused grid %d blocks of %d threads, occupancy API %d blocks/CU%s\n",
           p.name, p.ngA, p.ncA, gA, p.nwA, p.ngA * (double) p.ncA * TILE / 1e6, p.ngB, p.ncB, gB, p.nwB, p.ngB * (double) p.ncB * TILE / 1e6, gA + gB, bdim, occ,
           (gA + gB) <= 256 * (occ > 1 ? occ - 1 : occ) ? "" : "  ** NOT provably co-resident: fused variants skipped **");
    const bool fused_ok = (gA + gB) <= 256 * (occ > 1 ? occ - 1 : occ);      // one block per CU of margin (MI355X_MICROARCH: the API over-reports)
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; mode++) {
        if (mode && !fused_ok) continue;
        // capture ITER iterations (each on fresh weights) into one graph
        CHECK(hipMemsetAsync(cnt, 0, (2 * NCNT + 1) * 64, st)); CHECK(hipMemsetAsync(out, 0, 4, st)); CHECK(hipStreamSynchronize(st));
        hipGraph_t graph; hipGraphExec_t exec;
        CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        if (mode) CHECK(hipMemsetAsync(cnt, 0, (2 * NCNT + 1) * 64, st));              // counters re-armed once per replay
        for (int it = 0; it < ITER; it++) {
            Args a = { wA + (size_t) (it % NB) * bA, wB + (size_t) (it % NB) * bB, p.ngA, p.ncA, p.nwA, p.ngB, p.ncB, p.nwB, y, cnt, out, 1000u + it, (uint32_t) gA * (it + 1) };
            if (mode == 0) {
                hipLaunchKernelGGL((k_a<DA, DB>), dim3(gA), dim3(p.nwA * 64), 0, st, a);
                hipLaunchKernelGGL((k_b<DA, DB>), dim3(gB), dim3(p.nwB * 64), 0, st, a);
            } else if (mode == 1) hipLaunchKernelGGL((k_fused<DA, DB, 1>), dim3(gA + gB), dim3(bdim), 0, st, a);
            else hipLaunchKernelGGL((k_fused<DA, DB, 2>), dim3(gA + gB), dim3(bdim), 0, st, a);
        }
        CHECK(hipStreamEndCapture(st, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int i = 0; i < 3; i++) CHECK(hipGraphLaunch(exec, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemsetAsync(out, 0, 4, st));
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; i++) CHECK(hipGraphLaunch(exec, st));
        CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        uint32_t stale = 0; CHECK(hipMemcpy(&stale, out, 4, hipMemcpyDeviceToHost));
        const double us = ms * 1e3 / (reps * ITER);
        printf("   %-28s %7.2f us per A+B pair   %6.2f TB/s   stale y words seen: %u\n",
               mode == 0 ? "serial (2 launches)" : mode == 1 ? "fused, sc1 loads of y" : "fused, acquire fence", us,
               (p.ngA * (double) p.ncA + p.ngB * (double) p.ncB) * TILE / us * 1e-6, stale);
        CHECK(hipGraphExecDestroy(exec)); CHECK(hipGraphDestroy(graph));
    }
    CHECK(hipFree(wA)); CHECK(hipFree(wB)); CHECK(hipFree(y)); CHECK(hipFree(cnt)); CHECK(hipFree(out));
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 40;
    hipStream_t st; CHECK(hipStreamCreate(&st));
    const Pair wo_w13 = { "wo -> w1|w3", 512, 16, 1, 2752, 16, 8 };
    const Pair w13_w2 = { "w1|w3 -> w2", 2752, 16, 8, 512, 43, 4 };
    const Pair w2_qkv = { "w2 -> wq|wk|wv", 512, 43, 4, 1536, 16, 4 };
    const Pair wo_w13b = { "wo(4 waves) -> w1|w3", 512, 16, 4, 2752, 16, 8 };
    (void) wo_w13;
    run_pair<8, 4>(wo_w13b, reps, st);
    run_pair<4, 8>(w13_w2, reps, st);
    run_pair<10, 8>(w2_qkv, reps, st);
    return 0;
}
