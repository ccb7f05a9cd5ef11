// stream_probe.hip -- what the chip delivers for the decode mat-vec's ACCESS PATTERN, without the arithmetic
// (measurement tooling).  Every variant streams `ngroups` row-groups of `nchunks` 1280-byte tiles
// ([row-group][chunk] tiles, as QMat) from HBM and folds them into one dword per lane (so nothing is dead code):
//   reg<D>      : k_gemv's structure -- one wave per row-group, D-deep register ring of non-temporal loads
//   dma<R>      : the same stream through a wave-private LDS ring filled by LDS-DMA (global_load_lds), R slots
//   lin         : chip-wide linear read of the same bytes (float4 per lane, grid-stride): the copy-like ceiling
// Launches cycle over NB buffers (> 256 MB in total) so every launch reads cold HBM.  Reports us per launch and TB/s
// for back-to-back launches on one stream.
// build: hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o tools/stream_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int TILE = 1280;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int D>
__global__ void __launch_bounds__(512) k_reg(const uint8_t *__restrict__ wt, int ngroups, int nchunks, uint32_t *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int g = blockIdx.x * nw + wave;
    if (g >= ngroups) return;
    const uint8_t *wbase = wt + (size_t) g * (nchunks + 1) * TILE;
    u32x4 wq[D]; f32x2 ws[D];
#define LOADW(S, C) { const uint8_t *tp = wbase + (size_t) min((C), nchunks) * TILE; \
        wq[S] = __builtin_nontemporal_load((const u32x4 *) (tp + lane * 16)); \
        ws[S] = __builtin_nontemporal_load((const f32x2 *) (tp + 1024 + ((lane >> 3) * 8 + (lane & 3) * 2) * 4)); }
#pragma unroll
    for (int i = 0; i < D; i++) LOADW(i, i)
    uint32_t acc = 0;
    int c0 = 0;
    do {
#pragma unroll
        for (int i = 0; i < D; i++) {
            acc ^= wq[i].x ^ wq[i].y ^ wq[i].z ^ wq[i].w ^ __builtin_bit_cast(uint32_t, ws[i].x) ^ __builtin_bit_cast(uint32_t, ws[i].y);
            LOADW(i, c0 + D + i)
            __builtin_amdgcn_sched_barrier(0);
        }
        c0 += D;
    } while (c0 < nchunks);
    if (acc == 0x12345678u) out[g * 64 + lane] = acc;
}

// wave-private LDS ring of R slots; the whole ring is put in flight, then slot by slot: wait, read, refill
template <int R>
__global__ void __launch_bounds__(512) k_dma(const uint8_t *__restrict__ wt, int ngroups, int nchunks, uint32_t *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int g = blockIdx.x * nw + wave;
    if (g >= ngroups) return;
    uint8_t *ring = lds + wave * R * TILE;
    const uint32_t ring_lds = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint8_t *) ring;
    const uint64_t gb0 = (uint64_t) (wt + (size_t) g * (nchunks + 1) * TILE);
    const uint32_t voff_n = lane * 16, voff_s = lane * 4;
    auto issue = [&](int c, int slot) {
        const uint64_t gb = gb0 + (uint64_t) c * TILE;
        const uint32_t dst = ring_lds + slot * TILE;
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %4 nt\n\tglobal_load_lds_dword %2, %4 offset:1024 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff_n), "v"(voff_s), "s"(dst), "s"(gb) : "memory");
    };
    const int nfill = min(R, nchunks);
    for (int c = 0; c < nfill; c++) issue(c, c);
    uint32_t acc = 0;
    int slot = 0;
    const int nmain = nchunks - R;       // chunks consumed while refilling (<= 0: whole row in flight)
    for (int c = 0; c < nmain; c++) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"(2 * (R - 1)) : "memory");
        const u32x4 w = *(const u32x4 *) (ring + slot * TILE + lane * 16);
        const f32x2 s = *(const f32x2 *) (ring + slot * TILE + 1024 + ((lane >> 3) * 8 + (lane & 3) * 2) * 4);
        acc ^= w.x ^ w.y ^ w.z ^ w.w ^ __builtin_bit_cast(uint32_t, s.x) ^ __builtin_bit_cast(uint32_t, s.y);
        asm volatile("" : "+v"(acc));    // the reads have completed before the slot is refilled
        issue(c + R, slot);
        slot = slot + 1 == R ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (tail kept simple: the probe measures the stream, not the last R chunks)
    for (int c = max(nmain, 0); c < nchunks; c++) {
        const u32x4 w = *(const u32x4 *) (ring + slot * TILE + lane * 16);
        const f32x2 s = *(const f32x2 *) (ring + slot * TILE + 1024 + ((lane >> 3) * 8 + (lane & 3) * 2) * 4);
        acc ^= w.x ^ w.y ^ w.z ^ w.w ^ __builtin_bit_cast(uint32_t, s.x) ^ __builtin_bit_cast(uint32_t, s.y);
        slot = slot + 1 == R ? 0 : slot + 1;
    }
    if (acc == 0x12345678u) out[g * 64 + lane] = acc;
}

__global__ void __launch_bounds__(256) k_lin(const u32x4 *__restrict__ p, size_t n16, uint32_t *__restrict__ out) {
    uint32_t acc = 0;
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride);
        const u32x4 c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) { const u32x4 a = __builtin_nontemporal_load(p + i); acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}
__global__ void k_empty(uint32_t *out) { if (out == (uint32_t *) 1) out[0] = 1; }

struct Shape { const char *name; int ngroups, nchunks, nw; };

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 64;
    const Shape shapes[] = { { "w1|w3", 2752, 16, 8 }, { "wq|wk|wv", 1536, 16, 4 }, { "w2", 512, 43, 4 }, { "wo", 512, 16, 1 } };
    uint32_t *d_out; CHECK(hipMalloc((void **) &d_out, 4 << 20));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    for (const Shape &s : shapes) {
        const size_t bytes = (size_t) s.ngroups * (s.nchunks + 1) * TILE, algo = (size_t) s.ngroups * s.nchunks * TILE;
        const int NB = (int) ((600u << 20) / bytes) + 2;
        uint8_t *buf; CHECK(hipMalloc((void **) &buf, bytes * NB)); CHECK(hipMemset(buf, 0x5a, bytes * NB));
        auto timeit = [&](const char *label, auto launch) {
            for (int i = 0; i < 4; i++) launch(buf + (size_t) (i % NB) * bytes);
            CHECK(hipStreamSynchronize(st));
            CHECK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; i++) launch(buf + (size_t) (i % NB) * bytes);
            CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters;
            printf("%-9s %-14s %7.2f us  %6.2f TB/s\n", s.name, label, us, algo / us * 1e-6);
            CHECK(hipGetLastError());
        };
        const int grid = (s.ngroups + s.nw - 1) / s.nw;
        printf("-- %s: %d row-groups x %d chunks = %.1f MB, %d waves per workgroup, %d workgroups, %d buffers\n", s.name, s.ngroups, s.nchunks, algo / 1e6, s.nw, grid, NB);
        timeit("empty", [&](uint8_t *) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(s.nw * 64), 0, st, d_out); });
        timeit("lin", [&](uint8_t *b) { hipLaunchKernelGGL(k_lin, dim3(2048), dim3(256), 0, st, (const u32x4 *) b, bytes / 16, d_out); });
#define REG(D) timeit("reg" #D, [&](uint8_t *b) { hipLaunchKernelGGL((k_reg<D>), dim3(grid), dim3(s.nw * 64), 0, st, b, s.ngroups, s.nchunks, d_out); });
        REG(4) REG(8) REG(16)
        if (s.nchunks > 16) { REG(22) }
#define DMA(R, NW) { const int g2 = (s.ngroups + (NW) - 1) / (NW); const size_t l = (size_t) (NW) * (R) * TILE; \
            if (l <= 160 * 1024) { CHECK(hipFuncSetAttribute((const void *) k_dma<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) l)); \
            timeit("dma" #R "x" #NW, [&](uint8_t *b) { hipLaunchKernelGGL((k_dma<R>), dim3(g2), dim3((NW) * 64), l, st, b, s.ngroups, s.nchunks, d_out); }); } }
        DMA(4, 8) DMA(7, 8) DMA(8, 4) DMA(16, 4) DMA(16, 6) DMA(16, 2) DMA(16, 1) DMA(30, 2) DMA(30, 4) DMA(30, 1)
        CHECK(hipFree(buf));
    }
    return 0;
}
