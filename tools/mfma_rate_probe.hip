// mfma_rate_probe.hip -- issue cost of the instructions the exact prompt GEMM is built from (measurement tooling).
// Each variant runs ITERS x 16 independent instructions per wave in a rolled loop, one or two waves per SIMD on every
// CU, timed with HIP events; reported as nanoseconds per instruction per SIMD (the ratios between rows are the point).
//   v_mfma_i32_32x32x32_i8 (today: one per chain, operand masked)   v_mfma_f32_32x32x4_2b_f16 (K = 4, two chains per issue)
//   v_mfma_f32_16x16x4_4b_f16   v_pk_fma_f32   v_pk_add_f32   v_fma_f32   v_cvt_f32_i32   v_pk_add_f16   v_pk_mul_f32
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_rate_probe.hip -o tools/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef int i32x16v __attribute__((ext_vector_type(16)));
typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef float f32x32v __attribute__((ext_vector_type(32)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int WHICH>
__global__ void __launch_bounds__(512) k_probe(float *sink, int iters) {
    const int lane = threadIdx.x & 63;
    float s = 0.0f;
    if constexpr (WHICH == 0) {
        i32x4v a = { lane, lane + 1, lane + 2, lane + 3 }, b = { 1, 2, 3, 4 };
        i32x16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
            }
            asm volatile("" : "+v"(a));
        }
        s = (float) (c0[0] + c1[1] + c2[2] + c3[3]);
    } else if constexpr (WHICH == 1) {
        h4 a = { (_Float16) lane, 1, 2, 3 }, b = { 1, 2, 3, 4 };
        f32x32v c0 = {}, c1 = {};
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x4f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x4f16(a, b, c1, 0, 0, 0);
            }
            asm volatile("" : "+v"(a));
        }
        s = c0[0] + c1[17];
    } else if constexpr (WHICH == 2) {
        h4 a = { (_Float16) lane, 1, 2, 3 }, b = { 1, 2, 3, 4 };
        f32x16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f16(a, b, c3, 0, 0, 0);
            }
            asm volatile("" : "+v"(a));
        }
        s = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        f32x2 a[16], b = { 1.0001f, 0.9999f }, c = { 0.5f, 0.25f };
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = f32x2{ (float) lane + i, (float) i };
        for (int r = 0; r < iters; r++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (WHICH == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (WHICH == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (WHICH == 5) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (WHICH == 6) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i].x));
                if (WHICH == 7) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                if (WHICH == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (WHICH == 9) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
            }
#pragma unroll
        for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;
    }
    if (s == 123.456f) sink[0] = s;
}

int main() {
    float *d_sink;
    CHECK(hipMalloc((void **) &d_sink, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char *names[] = { "v_mfma_i32_32x32x32_i8", "v_mfma_f32_32x32x4_2b_f16", "v_mfma_f32_16x16x4_4b_f16", "v_pk_fma_f32", "v_pk_add_f32", "v_fma_f32",
                            "v_cvt_f32_i32", "v_pk_add_f16", "v_pk_mul_f32", "v_and_or_b32" };
    for (int wps = 1; wps <= 2; wps++) {
        printf("-- %d wave(s) per SIMD on every CU; ns per instruction per SIMD (and cycles at 2.4 GHz)\n", wps);
        for (int which = 0; which < 10; which++) {
            const int iters = which < 3 ? 2000 : 20000;
            auto launch = [&]() {
                const dim3 g(256), b(256 * wps);
                switch (which) {
                case 0: hipLaunchKernelGGL(k_probe<0>, g, b, 0, 0, d_sink, iters); break;
                case 1: hipLaunchKernelGGL(k_probe<1>, g, b, 0, 0, d_sink, iters); break;
                case 2: hipLaunchKernelGGL(k_probe<2>, g, b, 0, 0, d_sink, iters); break;
                case 3: hipLaunchKernelGGL(k_probe<3>, g, b, 0, 0, d_sink, iters); break;
                case 4: hipLaunchKernelGGL(k_probe<4>, g, b, 0, 0, d_sink, iters); break;
                case 5: hipLaunchKernelGGL(k_probe<5>, g, b, 0, 0, d_sink, iters); break;
                case 6: hipLaunchKernelGGL(k_probe<6>, g, b, 0, 0, d_sink, iters); break;
                case 7: hipLaunchKernelGGL(k_probe<7>, g, b, 0, 0, d_sink, iters); break;
                case 8: hipLaunchKernelGGL(k_probe<8>, g, b, 0, 0, d_sink, iters); break;
                default: hipLaunchKernelGGL(k_probe<9>, g, b, 0, 0, d_sink, iters); break;
                }
            };
            launch(); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0)); launch(); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double per = (double) ms * 1e6 / ((double) iters * 16 * wps);      // ns per instruction issued on one SIMD
            printf("   %-28s %7.2f ns  = %6.1f cycles\n", names[which], per, per * 2.4);
        }
    }
    return 0;
}
