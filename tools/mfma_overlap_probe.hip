// mfma_overlap_probe.hip -- do the matrix pipe and the VALU of one SIMD overlap? (measurement tooling for k_gemm_mfma4)
// A unit = one v_mfma_f32_16x16x4_4b_f16 + the 16 v_fma_f32 that consume its 16 result registers (what the exact prompt GEMM does per
// sub-tile and block).  Every wave runs ITERS x 16 units; 1, 2 or 4 waves per SIMD on every CU; HIP-event time / units issued per SIMD.
//   mode 0  MFMA only          mode 1  the 16 FMAs only (independent accumulators)
//   mode 2  MFMA, then its 16 dependent FMAs (the kernel today)
//   mode 3  software pipelined in the wave: MFMA of unit i + 1 is issued before the FMAs of unit i (two result sets)
//   mode 4  mode 3 with the FMAs of unit i split around the MFMA issue (8 before, 8 after)
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap_probe.hip -o tools/mfma_overlap_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16v __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define MF() (__extension__({ asm volatile("" : "+v"(a)); __builtin_amdgcn_mfma_f32_16x16x4f16(a, b, zero, 0, 0, 0); }))
#define FMA16(ACC, D, S) do { _Pragma("unroll") for (int e = 0; e < 16; e++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ACC[e]) : "v"(S), "v"(D[e])); } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k_probe(float *sink, int iters) {
    const int lane = threadIdx.x & 63;
    h4 a = { (_Float16) (float) (lane & 7), 1, 2, 3 }, b = { 1, 2, 3, 4 };
    f32x16v zero = {};
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; i++) acc[i] = 0.0f;
    float s = 1.0001f + lane * 1e-6f;
    f32x16v D0 = zero, D1 = zero;
    if constexpr (MODE == 3 || MODE == 4) D0 = MF();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            float *A = acc + 16 * (u & 1);
            if constexpr (MODE == 0) {
                D0 = MF();
                asm volatile("" :: "v"(D0));
            } else if constexpr (MODE == 1) {
                FMA16(A, D0, s);
            } else if constexpr (MODE == 2) {
                D0 = MF();
                A[0] = __builtin_fmaf(s, D0[0], A[0]);
                FMA16(A, D0, s);
            } else if constexpr (MODE == 3) {
                if (u & 1) { D0 = MF(); asm volatile("" : "+v"(D0)); A[0] = __builtin_fmaf(s, D1[0], A[0]); FMA16(A, D1, s); }
                else       { D1 = MF(); asm volatile("" : "+v"(D1)); A[0] = __builtin_fmaf(s, D0[0], A[0]); FMA16(A, D0, s); }
            } else {
                if (u & 1) {
                    A[0] = __builtin_fmaf(s, D1[0], A[0]);
#pragma unroll
                    for (int e = 0; e < 8; e++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(A[e]) : "v"(s), "v"(D1[e]));
                    D0 = MF(); asm volatile("" : "+v"(D0));
#pragma unroll
                    for (int e = 8; e < 16; e++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(A[e]) : "v"(s), "v"(D1[e]));
                } else {
                    A[0] = __builtin_fmaf(s, D0[0], A[0]);
#pragma unroll
                    for (int e = 0; e < 8; e++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(A[e]) : "v"(s), "v"(D0[e]));
                    D1 = MF(); asm volatile("" : "+v"(D1));
#pragma unroll
                    for (int e = 8; e < 16; e++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(A[e]) : "v"(s), "v"(D0[e]));
                }
            }
        }
        asm volatile("" : "+v"(a));
    }
    float r = D0[0] + D1[3];
#pragma unroll
    for (int i = 0; i < 32; i++) r += acc[i];
    if (r == 12345.678f) sink[threadIdx.x] = r;
}

template <int MODE>
static int run(const char *name, float *sink, int ncu) {
    const int iters = 4000;
    printf("%-68s", name);
    for (int w : { 1, 2, 4 }) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_probe<MODE>, dim3(ncu * w), dim3(256), 0, 0, sink, 100);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_probe<MODE>, dim3(ncu * w), dim3(256), 0, 0, sink, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %dw: %6.2f ns/unit/SIMD", w, ms * 1e6 / ((double) iters * 16 * w));
    }
    printf("\n");
    return 0;
}

// ---- part 2: one wave per SIMD issues only MFMAs (four independent accumulators), two more per SIMD only FMAs (a fixed 96 per iteration
// and wave).  what: 1 = MFMA waves work, 2 = FMA waves work, 3 = both.  workgroup = 12 waves, one per CU.  Per instruction type T.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x32v __attribute__((ext_vector_type(32)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef int i32x16v __attribute__((ext_vector_type(16)));
template <int T> struct Acc { typedef f32x16v type; };
template <> struct Acc<1> { typedef f32x32v type; };
template <> struct Acc<3> { typedef f32x4v type; };
template <> struct Acc<5> { typedef f32x4v type; };
template <> struct Acc<6> { typedef f32x4v type; };
template <> struct Acc<7> { typedef i32x16v type; };
template <> struct Acc<8> { typedef i32x16v type; };
template <> struct Acc<10> { typedef f32x4v type; };
template <int T, typename C> __device__ __forceinline__ C mf(h4 a, h4 b, h8 a8, h8 b8, C c) {
    if constexpr (T == 0) return __builtin_amdgcn_mfma_f32_16x16x4f16(a, b, c, 0, 0, 0);
    else if constexpr (T == 1) return __builtin_amdgcn_mfma_f32_32x32x4f16(a, b, c, 0, 0, 0);
    else if constexpr (T == 2) return __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c, 0, 0, 0);
    else if constexpr (T == 3) return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
    else if constexpr (T == 4) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c, 0, 0, 0);
    else if constexpr (T == 5) return __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c, 0, 0, 0);
    else if constexpr (T == 6) return __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0);
    else if constexpr (T == 7) return __builtin_amdgcn_mfma_i32_32x32x16_i8(__builtin_bit_cast(long, a), __builtin_bit_cast(long, b), c, 0, 0, 0);
    else if constexpr (T == 9) return __builtin_amdgcn_mfma_f32_32x32x2f32((float) a[0], (float) b[0], c, 0, 0, 0);
    else if constexpr (T == 10) return __builtin_amdgcn_mfma_f32_16x16x4f32((float) a[0], (float) b[0], c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4v, a8), __builtin_bit_cast(i32x4v, b8), c, 0, 0, 0);
}
template <int T>
__global__ void __launch_bounds__(768) k_split(float *sink, int iters, int what) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float r = 0.0f;
    if (wave < 4) {
        if (!(what & 1)) return;
        h4 a = { (_Float16) (float) (lane & 7), 1, 2, 3 }, b = { 1, 2, 3, 4 };
        h8 a8 = { (_Float16) (float) (lane & 7), 1, 2, 3, 1, 2, 3, 4 }, b8 = { 1, 2, 3, 4, 1, 0, 1, 0 };
        typename Acc<T>::type c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                c0 = mf<T>(a, b, a8, b8, c0); c1 = mf<T>(a, b, a8, b8, c1); c2 = mf<T>(a, b, a8, b8, c2); c3 = mf<T>(a, b, a8, b8, c3);
            }
            asm volatile("" : "+v"(a), "+v"(a8));
        }
        r = (float) (c0[0] + c1[1] + c2[2] + c3[3]);
    } else {
        if (!(what & 2)) return;
        float acc[32];
#pragma unroll
        for (int i = 0; i < 32; i++) acc[i] = 0.0f;
        const float s = 1.0001f + lane * 1e-6f, d = 0.5f;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 96; u++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[u & 31]) : "v"(s), "v"(d));
        }
#pragma unroll
        for (int i = 0; i < 32; i++) r += acc[i];
    }
    if (r == 12345.678f) sink[threadIdx.x] = r;
}
template <int T>
static int run_split(const char *name, float *sink, int ncu) {
    const int iters = 4000;
    float t[4] = { 0, 0, 0, 0 };
    for (int what = 1; what <= 3; what++) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_split<T>, dim3(ncu), dim3(768), 0, 0, sink, 100, what);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_split<T>, dim3(ncu), dim3(768), 0, 0, sink, iters, what);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        t[what] = ms * 1e6f / iters;
    }
    printf("split %-34s MFMA waves alone %7.2f ns (%5.2f per MFMA) | FMA waves alone %7.2f | together %7.2f = %4.0f %% of the sum, %4.0f %% of the max\n",
           name, t[1], t[1] / 16, t[2], t[3], 100.0 * t[3] / (t[1] + t[2]), 100.0 * t[3] / (t[1] > t[2] ? t[1] : t[2]));
    return 0;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    float *sink; CHECK(hipMalloc((void **) &sink, 4096));
    printf("# unit = v_mfma_f32_16x16x4_4b_f16 + 16 v_fma_f32; %d CUs, one workgroup of 4 waves per CU and wave-per-SIMD count\n", p.multiProcessorCount);
    run<0>("0 MFMA only", sink, p.multiProcessorCount);
    run<1>("1 16 FMAs only", sink, p.multiProcessorCount);
    run<2>("2 MFMA then its 16 dependent FMAs (k_gemm_mfma4)", sink, p.multiProcessorCount);
    run<3>("3 MFMA of unit i+1 issued before the FMAs of unit i", sink, p.multiProcessorCount);
    run<4>("4 MFMA of unit i+1 issued between the two halves of unit i's FMAs", sink, p.multiProcessorCount);
    printf("# split: one MFMA wave + two FMA waves (96 v_fma_f32 each per 16 MFMAs) per SIMD\n");
    run_split<0>("v_mfma_f32_16x16x4_4b_f16", sink, p.multiProcessorCount);
    run_split<1>("v_mfma_f32_32x32x4_2b_f16", sink, p.multiProcessorCount);
    run_split<2>("v_mfma_f32_32x32x8_f16", sink, p.multiProcessorCount);
    run_split<3>("v_mfma_f32_16x16x16_f16", sink, p.multiProcessorCount);
    run_split<4>("v_mfma_f32_32x32x16_f16", sink, p.multiProcessorCount);
    run_split<5>("v_mfma_f32_16x16x32_f16", sink, p.multiProcessorCount);
    run_split<6>("v_mfma_f32_4x4x4_16b_f16", sink, p.multiProcessorCount);
    run_split<7>("v_mfma_i32_32x32x16_i8", sink, p.multiProcessorCount);
    run_split<8>("v_mfma_i32_32x32x32_i8", sink, p.multiProcessorCount);
    run_split<9>("v_mfma_f32_32x32x2_f32", sink, p.multiProcessorCount);
    run_split<10>("v_mfma_f32_16x16x4_f32", sink, p.multiProcessorCount);
    return 0;
}
