// l2_prefetch_probe.hip -- does data fetched into an XCD's L2 by one launch survive for the next one? (measurement tooling)
// The decode step's L2 run-ahead prefetcher (k_prefetch) only pays if (a) lines touched by another kernel are still in the L2 when the
// consumer launch reads them, (b) the toucher knows which XCD the consumer workgroup runs on.  256 consumer workgroups read 64 KiB
// each (16 MiB: half of the 8 x 4 MiB of L2) with the decode kernels' non-temporal dwordx4 loads; per scenario the average time a
// workgroup needs for its region (s_memtime cycles -> ns at 100 MHz wall clock) and the launch duration:
//   cold          : after streaming 1 GiB through the caches
//   repeat        : the same consume launch again (its own lines, placed by itself)
//   touch same    : a touch launch (one dword per line; workgroup b touches region b) first, same stream, kernel boundary between
//   touch shifted : workgroup b touched region b + 1 (a neighbour XCD's region)
//   touch beside  : the touch launch runs concurrently on a second stream, started 20 us earlier (no kernel boundary in between on
//                   the consumer's stream except its own start)
// and the XCC_ID of workgroups 0..15 of a 256-workgroup launch (is it b % 8, numerically?).
// build: hipcc --offload-arch=gfx950 -O3 tools/l2_prefetch_probe.hip -o tools/l2_prefetch_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int REGION = 64 * 1024;

__global__ void __launch_bounds__(256) k_consume(const uint8_t *__restrict__ w, uint32_t *__restrict__ out, unsigned long long *__restrict__ cyc, uint32_t *__restrict__ xcc) {
    const uint8_t *p = w + (size_t) blockIdx.x * REGION;
    const unsigned long long t0 = wall_clock64();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < REGION / (256 * 16); i++) {
        const u32x4 v = __builtin_nontemporal_load((const u32x4 *) (p + ((size_t) i * 256 + threadIdx.x) * 16));
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        cyc[blockIdx.x] = wall_clock64() - t0;
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = id & 0xf;
    }
}
__global__ void __launch_bounds__(64) k_touch(const uint8_t *__restrict__ w, int shift, int nregions, uint32_t *__restrict__ out) {
    const uint8_t *p = w + (size_t) ((blockIdx.x + shift) % nregions) * REGION;
    uint32_t acc = 0;
    for (int off = threadIdx.x * 128; off < REGION; off += 64 * 128 * 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) { const int o = off + u * 64 * 128; if (o < REGION) acc ^= *(const uint32_t *) (p + o); }
    }
    if (acc == 0x12345678u) out[blockIdx.x * 64 + threadIdx.x] = acc;
}
__global__ void k_flush(const u32x4 *__restrict__ big, size_t n, uint32_t *__restrict__ out) {
    uint32_t acc = 0;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) { const u32x4 v = big[i]; acc ^= v.x ^ v.y; }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}
__global__ void k_spin(unsigned long long ticks) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4); }

int main() {
    const int NB = 256;
    uint8_t *w; uint32_t *out, *xcc; unsigned long long *cyc; u32x4 *big;
    const size_t big_n = (size_t) 1 << 26;         // 1 GiB of 16-byte elements
    CHECK(hipMalloc(&w, (size_t) NB * REGION)); CHECK(hipMemset(w, 1, (size_t) NB * REGION));
    CHECK(hipMalloc(&out, 1 << 20)); CHECK(hipMalloc(&xcc, NB * 4)); CHECK(hipMalloc(&cyc, NB * 8));
    CHECK(hipMalloc(&big, big_n * 16)); CHECK(hipMemset(big, 2, big_n * 16));
    hipStream_t s1, s2; CHECK(hipStreamCreate(&s1)); CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<unsigned long long> hc(NB); std::vector<uint32_t> hx(NB);
    auto flush = [&]() { hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, s1, big, big_n, out); };
    auto consume_timed = [&](const char *name) {
        CHECK(hipEventRecord(e0, s1));
        hipLaunchKernelGGL(k_consume, dim3(NB), dim3(256), 0, s1, w, out, cyc, xcc);
        CHECK(hipEventRecord(e1, s1));
        CHECK(hipStreamSynchronize(s1)); CHECK(hipStreamSynchronize(s2));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(hc.data(), cyc, NB * 8, hipMemcpyDeviceToHost));
        double sum = 0, mx = 0; for (auto c : hc) { sum += (double) c; mx = mx > (double) c ? mx : (double) c; }
        printf("%-14s launch %7.2f us   per-workgroup region read: avg %7.0f ns  max %7.0f ns\n", name, ms * 1e3, sum / NB * 10.0, mx * 10.0);
    };
    for (int rep = 0; rep < 3; rep++) {
        printf("-- repetition %d\n", rep);
        flush(); consume_timed("cold");
        consume_timed("repeat");
        flush(); hipLaunchKernelGGL(k_touch, dim3(NB), dim3(64), 0, s1, w, 0, NB, out); consume_timed("touch same");
        flush(); hipLaunchKernelGGL(k_touch, dim3(NB), dim3(64), 0, s1, w, 1, NB, out); consume_timed("touch shifted");
        flush(); hipLaunchKernelGGL(k_touch, dim3(NB), dim3(64), 0, s1, w, 8, NB, out); consume_timed("touch shift 8");
        flush(); CHECK(hipStreamSynchronize(s1));
        hipLaunchKernelGGL(k_touch, dim3(NB), dim3(64), 0, s2, w, 0, NB, out);          // beside: on the second stream ...
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, s1, 2000ull);                    // ... while the consumer's stream waits 20 us behind a small kernel
        consume_timed("touch beside");
    }
    CHECK(hipMemcpy(hx.data(), xcc, NB * 4, hipMemcpyDeviceToHost));
    printf("XCC_ID of workgroups 0..15:"); for (int i = 0; i < 16; i++) printf(" %u", hx[i]); printf("\n");
    int same = 0; for (int i = 0; i < NB; i++) same += hx[i] == hx[i % 8];
    printf("workgroups whose XCC_ID equals that of workgroup b %% 8: %d / %d\n", same, NB);
    return 0;
}
