// dense.hip -- f16 / f32 model files (ggml-model-f16.bin, f16 = 1; f32 = 0): SURVEY.md section 8f N3.
//
// Only the weight mat-muls and the embedding gather differ from a Q4_0 model; norm, RoPE, KV cache and
// attention are the same kernels (kernels.hip).  Reference semantics (x86 AVX2+F16C build):
//   f16 weights  ggml_compute_forward_mul_mat_f16_f32 (ggml.c:5681-5985): the activations are rounded to
//                fp16 (_cvtss_sh, RNE) once per mat-mul, then every output is ggml_vec_dot_f16
//                (ggml.c:1260-1297): 4 x 8 = 32 fp32 FMA chains over the widened halves (chain l =
//                elements l, l+32, ...), folded by GGML_F32x8_REDUCE (ggml.c:872-887).
//   f32 weights  ggml_compute_forward_mul_mat_f32 (ggml.c:5430-5680), ggml_vec_dot_f32 (:1223-1258): the
//                same 32 chains without any rounding of the inputs.
//   embedding    ggml_compute_forward_get_rows_f16 / _f32 (ggml.c:6787-6850): widen / copy.
// One half-wave (32 lanes = the 32 chains) owns RG weight rows x NC activation rows; the fold is the
// xor butterfly 8, 16, 4, 1, 2 (the reference's tree; float add commutes).  First version: correct and
// bandwidth-lean for decode (weights read once), not tuned.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "llamahip_internal.h"

namespace lh {

namespace {

__device__ __forceinline__ uint16_t f2h_rne(float f) { return __builtin_bit_cast(uint16_t, (_Float16) f); }      // v_cvt_f16_f32: RNE
__device__ __forceinline__ float h2f(uint16_t h) { return (float) __builtin_bit_cast(_Float16, h); }

template <int WT> struct WElem;
template <> struct WElem<0> { typedef float T; static __device__ __forceinline__ float widen(float v) { return v; } static __device__ __forceinline__ float act(float x) { return x; } };
template <> struct WElem<1> { typedef uint16_t T; static __device__ __forceinline__ float widen(uint16_t v) { return h2f(v); } static __device__ __forceinline__ float act(float x) { return h2f(f2h_rne(x)); } };

constexpr int RG = 4;      // weight rows per half-wave

// y[n][m] (+ resid[n][m]) = dot(W[m][:], act(x[n][:])).  grid (ceil(M / (8 * RG)), ceil(N / NC)), 256 threads.
template <int WT, int NC, int EPI>
__global__ void __launch_bounds__(256)
k_dense_mm(const void *__restrict__ wv, int M, int K, const float *__restrict__ x, long x_stride, int N,
           float *__restrict__ y, long y_stride, const float *__restrict__ resid, long resid_stride) {
    typedef typename WElem<WT>::T T;
    const T *w = (const T *) wv;
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const int m0 = (blockIdx.x * 8 + hw) * RG, n0 = blockIdx.y * NC;
    const T *wr[RG];
#pragma unroll
    for (int r = 0; r < RG; r++) wr[r] = w + (size_t) min(m0 + r, M - 1) * K + l;
    const float *xr[NC];
#pragma unroll
    for (int n = 0; n < NC; n++) xr[n] = x + (size_t) min(n0 + n, N - 1) * x_stride + l;
    float acc[RG][NC];
#pragma unroll
    for (int r = 0; r < RG; r++)
#pragma unroll
        for (int n = 0; n < NC; n++) acc[r][n] = 0.0f;
    constexpr int U = NC == 1 ? 8 : 2;                    // steps of 32 elements in flight
    for (int j0 = 0; j0 < K; j0 += 32 * U) {
        T wq[U][RG];
        float xq[U][NC];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int j = min(j0 + 32 * u, K - 32);        // K % 32 == 0; past the end: clamped re-read, not accumulated
#pragma unroll
            for (int r = 0; r < RG; r++) wq[u][r] = wr[r][j];
#pragma unroll
            for (int n = 0; n < NC; n++) xq[u][n] = xr[n][j];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (j0 + 32 * u < K) {
#pragma unroll
                for (int n = 0; n < NC; n++) {
                    const float xa = WElem<WT>::act(xq[u][n]);
#pragma unroll
                    for (int r = 0; r < RG; r++) acc[r][n] = fmaf(WElem<WT>::widen(wq[u][r]), xa, acc[r][n]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RG; r++)
#pragma unroll
        for (int n = 0; n < NC; n++) {
            float s = acc[r][n];
            s += __shfl_xor(s, 8);
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 4);
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            if (l == 0 && m0 + r < M && n0 + n < N) {
                if (EPI == EPI_RESID) s = s + resid[(size_t) (n0 + n) * resid_stride + m0 + r];
                y[(size_t) (n0 + n) * y_stride + m0 + r] = s;
            }
        }
}

template <int WT>
__global__ void k_embed_dense(const int32_t *__restrict__ tokens, const void *__restrict__ emb, float *__restrict__ x, int d) {
    typedef typename WElem<WT>::T T;
    const int n = blockIdx.x;
    const T *row = (const T *) emb + (size_t) tokens[n] * d;
    for (int i = threadIdx.x; i < d; i += blockDim.x) x[(size_t) n * d + i] = WElem<WT>::widen(row[i]);
}

template <int WT, int NC>
hipError_t go(const DMat &w, int epi, const float *x, long x_stride, int N, float *y, long y_stride,
              const float *resid, long resid_stride, hipStream_t st) {
    const dim3 grid((w.M + 8 * RG - 1) / (8 * RG), (N + NC - 1) / NC);
    if (epi == EPI_RESID)
        hipLaunchKernelGGL((k_dense_mm<WT, NC, EPI_RESID>), grid, dim3(256), 0, st, w.w, w.M, w.K, x, x_stride, N, y, y_stride, resid, resid_stride);
    else
        hipLaunchKernelGGL((k_dense_mm<WT, NC, EPI_STORE>), grid, dim3(256), 0, st, w.w, w.M, w.K, x, x_stride, N, y, y_stride, resid, resid_stride);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_dense_mm(const DMat &w, int epi, const float *x, long x_stride, int N, float *y, long y_stride,
                           const float *resid, long resid_stride, hipStream_t st) {
    if (w.K % 32 != 0 || (w.wtype != 0 && w.wtype != 1)) return hipErrorInvalidValue;
    if (w.wtype == 1) return N == 1 ? go<1, 1>(w, epi, x, x_stride, N, y, y_stride, resid, resid_stride, st)
                                    : go<1, 8>(w, epi, x, x_stride, N, y, y_stride, resid, resid_stride, st);
    return N == 1 ? go<0, 1>(w, epi, x, x_stride, N, y, y_stride, resid, resid_stride, st)
                  : go<0, 8>(w, epi, x, x_stride, N, y, y_stride, resid, resid_stride, st);
}

hipError_t launch_embed_dense(const int32_t *tokens, const void *emb, int wtype, float *x, int d, int N, hipStream_t st) {
    if (wtype == 1) hipLaunchKernelGGL(k_embed_dense<1>, dim3(N), dim3(256), 0, st, tokens, emb, x, d);
    else            hipLaunchKernelGGL(k_embed_dense<0>, dim3(N), dim3(256), 0, st, tokens, emb, x, d);
    return hipGetLastError();
}

}  // namespace lh
