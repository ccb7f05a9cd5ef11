// valu_rate_probe.hip -- VALU issue rate on gfx950 as a function of waves per SIMD (measurement tooling).
// Every wave runs ITERS x 32 independent instructions (32 accumulators, volatile asm, rolled loop) and times itself
// with s_memtime (shader clock); wave 0 of workgroup 0 reports cycles per instruction PER SIMD (its own cycles per
// instruction divided by the waves sharing the SIMD).  All 256 CUs are loaded the same way.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o tools/valu_rate_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int WHICH>
__global__ void __launch_bounds__(1024) k_probe(float *sink, uint64_t *cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f32x2 a[32], b = { 1.0001f, 0.9999f }, c = { 0.5f, 0.25f };
#pragma unroll
    for (int i = 0; i < 32; i++) a[i] = f32x2{ (float) lane + i, (float) i };
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int r = 0; r < iters; r++)
#pragma unroll
        for (int i = 0; i < 32; i++) {
            if (WHICH == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
            if (WHICH == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (WHICH == 2) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
            if (WHICH == 3) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
            if (WHICH == 4) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
            if (WHICH == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (WHICH == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
            if (WHICH == 7) asm volatile("v_dot8_i32_i4 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
            if (WHICH == 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
            if (WHICH == 9) asm volatile("v_lshrrev_b32 %0, 4, %0" : "+v"(a[i].x));
        }
    const uint64_t t1 = __builtin_readcyclecounter();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; i++) s += a[i].x + a[i].y;
    if (s == 123.456f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float *d_sink; uint64_t *d_cyc;
    CHECK(hipMalloc((void **) &d_sink, 4)); CHECK(hipMalloc((void **) &d_cyc, 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char *names[] = { "v_fma_f32", "v_pk_fma_f32", "v_fmac_f32", "v_and_or_b32", "v_pk_add_f16", "v_pk_mul_f32", "v_mul_f32", "v_dot8_i32_i4", "v_add_f32", "v_lshrrev_b32" };
    const int iters = 20000;
    printf("%-16s", "waves per SIMD:");
    for (int wps = 1; wps <= 4; wps++) printf("   %d: cyc/inst/SIMD  ns/inst/SIMD", wps);
    printf("\n");
    for (int which = 0; which < 10; which++) {
        printf("%-16s", names[which]);
        for (int wps = 1; wps <= 4; wps++) {
            auto launch = [&]() {
                const dim3 g(256), b(256 * wps);
                switch (which) {
                case 0: hipLaunchKernelGGL(k_probe<0>, g, b, 0, 0, d_sink, d_cyc, iters); break;
                case 1: hipLaunchKernelGGL(k_probe<1>, g, b, 0, 0, d_sink, d_cyc, iters); break;
                case 2: hipLaunchKernelGGL(k_probe<2>, g, b, 0, 0, d_sink, d_cyc, iters); break;
                case 3: hipLaunchKernelGGL(k_probe<3>, g, b, 0, 0, d_sink, d_cyc, iters); break;
                case 4: hipLaunchKernelGGL(k_probe<4>, g, b, 0, 0, d_sink, d_cyc, iters); break;
                case 5: hipLaunchKernelGGL(k_probe<5>, g, b, 0, 0, d_sink, d_cyc, iters); break;
                case 6: hipLaunchKernelGGL(k_probe<6>, g, b, 0, 0, d_sink, d_cyc, iters); break;
                case 7: hipLaunchKernelGGL(k_probe<7>, g, b, 0, 0, d_sink, d_cyc, iters); break;
                case 8: hipLaunchKernelGGL(k_probe<8>, g, b, 0, 0, d_sink, d_cyc, iters); break;
                default: hipLaunchKernelGGL(k_probe<9>, g, b, 0, 0, d_sink, d_cyc, iters); break;
                }
            };
            launch(); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0)); launch(); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            uint64_t cyc; CHECK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
            printf("   %16.2f  %12.3f", (double) cyc / ((double) iters * 32 * wps), (double) ms * 1e6 / ((double) iters * 32 * wps));
        }
        printf("\n");
    }
    return 0;
}
