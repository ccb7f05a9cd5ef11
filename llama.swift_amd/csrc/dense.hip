// dense.hip -- the other model file types: f16 / f32 (f16 = 1 / 0) and Q4_1 (f16 = 3; N2).
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
// xor butterfly 8, 16, 4, 1, 2 (the reference's tree; float add commutes).  Weights and activations are kept
// in a chain-major order (perm_index) so that a lane's loads are 16 / 32 bytes wide.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#include "llamahip_internal.h"

namespace lh {

namespace {

__device__ __forceinline__ uint16_t f2h_rne(float f) { return __builtin_bit_cast(uint16_t, (_Float16) f); }      // v_cvt_f16_f32: RNE
__device__ __forceinline__ float h2f(uint16_t h) { return (float) __builtin_bit_cast(_Float16, h); }

template <int WT> struct WElem;
template <> struct WElem<0> { typedef float T; static __device__ __forceinline__ float widen(float v) { return v; } static __device__ __forceinline__ float act(float x) { return x; } };
template <> struct WElem<1> { typedef uint16_t T; static __device__ __forceinline__ float widen(uint16_t v) { return h2f(v); } static __device__ __forceinline__ float act(float x) { return h2f(f2h_rne(x)); } };

// a lane's 8 consecutive operands of one group, kept as loaded (16 / 32 bytes) and widened at use: unpacked
// fp16 values would take one VGPR each (k_dense_mv: 170 VGPRs instead of 110)
typedef uint32_t du4 __attribute__((ext_vector_type(4)));
typedef float df4 __attribute__((ext_vector_type(4)));
template <int WT> struct WVec8;
// fp16 operands stay packed (two per VGPR) and are widened INSIDE the FMA: v_fma_mix_f32 reads its first
// source as the low / high half of a register (op_sel) and multiplies-adds in fp32 with one rounding --
// exactly fma((float) h, x, acc).  hipcc emits v_cvt_f32_f16 + v_fma_f32 for the C++ form and unpacks a
// whole batch first (246 VGPRs, or spills under a register cap), hence inline assembly.
template <> struct WVec8<1> {
    du4 v;
    __device__ __forceinline__ void load(const uint16_t *p) { v = __builtin_nontemporal_load((const du4 *) p); }
    __device__ __forceinline__ void fma_into(float &acc, int u, float x) const {
        const uint32_t word = u < 2 ? v.x : u < 4 ? v.y : u < 6 ? v.z : v.w;
        if (u & 1) asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(word), "v"(x));
        else       asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(word), "v"(x));
    }
};
template <> struct WVec8<0> {
    df4 a, b;
    __device__ __forceinline__ void load(const float *p) { a = __builtin_nontemporal_load((const df4 *) p); b = __builtin_nontemporal_load((const df4 *) p + 1); }
    __device__ __forceinline__ void fma_into(float &acc, int u, float x) const {
        const float wv = u == 0 ? a.x : u == 1 ? a.y : u == 2 ? a.z : u == 3 ? a.w : u == 4 ? b.x : u == 5 ? b.y : u == 6 ? b.z : b.w;
        acc = fmaf(wv, x, acc);
    }
};

constexpr int RG = 4;      // weight rows per half-wave

// Chain-major order inside groups of 256 elements (8 steps of the 32 chains): element g*256 + 32*s + l is stored
// at g*256 + l*st + s (st = steps in the group: 8, fewer in a tail group), so that lane l -- chain l -- finds
// its next 8 operands in ONE 16-byte (fp16) or 32-byte (fp32) load and a half-wave reads 512 / 1024 contiguous
// bytes.  The first version read 2 bytes per lane and step and reached 1.3 TB/s.
__device__ __forceinline__ long perm_index(long e, int K) {
    const long g = e >> 8;
    const int within = (int) (e & 255), l = within & 31, sidx = within >> 5;
    const int gs = (int) min((long) 256, (long) K - g * 256), st = gs >> 5;
    return g * 256 + (long) l * st + sidx;
}

// weights: raw rows -> permuted rows (load time)
template <int WT>
__global__ void k_dense_perm_rows(const void *__restrict__ src, void *__restrict__ dst, long M, int K) {
    typedef typename WElem<WT>::T T;
    const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= M * K) return;
    const long m = gid / K, e = gid % K;
    ((T *) dst)[m * K + perm_index(e, K)] = ((const T *) src)[gid];
}

// activations: round through fp16 where the weights are fp16 (ggml.c:5681-5985 INIT phase) and permute
template <int WT>
__global__ void k_dense_perm_act(const float *__restrict__ x, long x_stride, int K, float *__restrict__ xp) {
    const int n = blockIdx.x;
    for (int e = threadIdx.x; e < K; e += blockDim.x) xp[(size_t) n * K + perm_index(e, K)] = WElem<WT>::act(x[(size_t) n * x_stride + e]);
}

// Activation preparation fused with the rounding / permutation above (one launch instead of k_prep_qa with an
// fp32 side output + k_dense_perm_act: 16 instead of 61 us per layer at 7B shapes).  One thread per 16
// contiguous elements; NORM keeps a row in one workgroup (two block-wide double sums), the other modes are
// sliced over gridDim.y.  Arithmetic = make_y in kernels.hip:
//   MODE 1 plain    y = in0
//   MODE 2 norm     y = w * ((float)(x - mean) * scale)      ggml_norm + ggml_mul (ggml.c:5327-5385, :4555)
//   MODE 3 SiLU*up  y = silu_lut(in0) * in1                  (ggml.c:1956-1963, .mm:678-680)
__device__ __forceinline__ double dense_block_sum(double v, double *red) {
    for (int msk = 32; msk > 0; msk >>= 1) {
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __shfl_xor(lo, msk); hi = __shfl_xor(hi, msk);
        v += __hiloint2double(hi, lo);
    }
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();                                       // (red may still be read by the previous call)
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < nw; i++) s += red[i];
    return s;
}
template <int MODE, int WT>
__global__ void __launch_bounds__(1024)
k_dense_prep(const float *__restrict__ in0, const float *__restrict__ in1, long in_stride, long in1_stride, int K,
             float *__restrict__ xp, const uint16_t *__restrict__ T_silu) {
    __shared__ double red[16];
    const int n = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int nh = K >> 4;
    const int hi = blockIdx.y * nt + tid;
    const bool live = hi < nh;
    const int hc = min(hi, nh - 1);
    const df4 *a4 = (const df4 *) (in0 + (size_t) n * in_stride) + hc * 4;
    df4 xa[4], xb[4];
#pragma unroll
    for (int v = 0; v < 4; v++) xa[v] = a4[v];
    if (MODE == PREP_NORM) {
#pragma unroll
        for (int v = 0; v < 4; v++) xb[v] = ((const df4 *) in1)[hc * 4 + v];
    } else if (MODE == PREP_SILU_MUL) {
        const df4 *b4 = (const df4 *) (in1 + (size_t) n * in1_stride) + hc * 4;
#pragma unroll
        for (int v = 0; v < 4; v++) xb[v] = b4[v];
    }
    if (MODE == PREP_NORM) {
        double s1 = 0.0;
        if (live) {
#pragma unroll
            for (int v = 0; v < 4; v++) { s1 += (double) xa[v].x; s1 += (double) xa[v].y; s1 += (double) xa[v].z; s1 += (double) xa[v].w; }
        }
        const double mean = dense_block_sum(s1, red) / (double) K;
        double s2 = 0.0;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const double v0 = (double) xa[v].x - mean, v1 = (double) xa[v].y - mean;
            const double v2 = (double) xa[v].z - mean, v3 = (double) xa[v].w - mean;
            xa[v].x = (float) v0; xa[v].y = (float) v1; xa[v].z = (float) v2; xa[v].w = (float) v3;
            if (live) { s2 += v0 * v0; s2 += v1 * v1; s2 += v2 * v2; s2 += v3 * v3; }
        }
        const double sum2 = dense_block_sum(s2, red);
        const float scale = (float) (1.0 / sqrt(sum2 / (double) K + (double) 1e-5f));
#pragma unroll
        for (int v = 0; v < 4; v++) {
            xa[v].x = xb[v].x * (xa[v].x * scale); xa[v].y = xb[v].y * (xa[v].y * scale);
            xa[v].z = xb[v].z * (xa[v].z * scale); xa[v].w = xb[v].w * (xa[v].w * scale);
        }
    } else if (MODE == PREP_SILU_MUL) {
#pragma unroll
        for (int v = 0; v < 4; v++) {
            xa[v].x = h2f(T_silu[f2h_rne(xa[v].x)]) * xb[v].x; xa[v].y = h2f(T_silu[f2h_rne(xa[v].y)]) * xb[v].y;
            xa[v].z = h2f(T_silu[f2h_rne(xa[v].z)]) * xb[v].z; xa[v].w = h2f(T_silu[f2h_rne(xa[v].w)]) * xb[v].w;
        }
    }
    if (!live) return;
    // The fp32 products above must exist as fp32 values before they are rounded to fp16: left alone, hipcc
    // folds `(_Float16) (a * b)` into v_fma_mixlo_f16 -- ONE rounding of the exact product -- while the
    // reference rounds the product to fp32 (ggml_mul) and that to fp16 (the mat-mul's INIT phase).  Rare
    // (a few elements per thousand rows), and enough to flip logits in the 4th digit.
#pragma unroll
    for (int v = 0; v < 4; v++) asm volatile("" : "+v"(xa[v].x), "+v"(xa[v].y), "+v"(xa[v].z), "+v"(xa[v].w));
    // element e = hi*16 + i  ->  g*256 + l*st + sidx  (perm_index): the 16 elements share g and sidx
    const long e0 = (long) hi * 16;
    const long g = e0 >> 8;
    const int within = (int) (e0 & 255), l0 = within & 31, sidx = within >> 5;
    const int gs = (int) min((long) 256, (long) K - g * 256), stp = gs >> 5;
    float *o = xp + (size_t) n * K + g * 256 + sidx;
#pragma unroll
    for (int v = 0; v < 4; v++) {
        o[(size_t) (l0 + 4 * v + 0) * stp] = WElem<WT>::act(xa[v].x);
        o[(size_t) (l0 + 4 * v + 1) * stp] = WElem<WT>::act(xa[v].y);
        o[(size_t) (l0 + 4 * v + 2) * stp] = WElem<WT>::act(xa[v].z);
        o[(size_t) (l0 + 4 * v + 3) * stp] = WElem<WT>::act(xa[v].w);
    }
}

// y[n][m] (+ resid[n][m]) = dot(W[m][:], xp[n][:]) on permuted operands.  grid (ceil(M / (8 * RG)), ceil(N / NC)), 256 threads.
template <int WT, int NC, int EPI>
__global__ void __launch_bounds__(256)
k_dense_mm(const void *__restrict__ wv, int M, int K, const float *__restrict__ xp, int N,
           float *__restrict__ y, long y_stride, const float *__restrict__ resid, long resid_stride) {
    typedef typename WElem<WT>::T T;
    const T *w = (const T *) wv;
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const int m0 = (blockIdx.x * 8 + hw) * RG, n0 = blockIdx.y * NC;
    const T *wr[RG];
#pragma unroll
    for (int r = 0; r < RG; r++) wr[r] = w + (size_t) min(m0 + r, M - 1) * K;
    const float *xr[NC];
#pragma unroll
    for (int n = 0; n < NC; n++) xr[n] = xp + (size_t) min(n0 + n, N - 1) * K;
    float acc[RG][NC];
#pragma unroll
    for (int r = 0; r < RG; r++)
#pragma unroll
        for (int n = 0; n < NC; n++) acc[r][n] = 0.0f;
    const int ng = K >> 8;
    // full groups: 8 steps per lane and group.  A decode launch is a few hundred waves each streaming its rows
    // once, so the bytes in flight per lane decide the rate: UG groups (UG x RG vector loads) are issued before
    // the first is consumed.  Groups past the end are clamped re-reads that are not accumulated.
    constexpr int UG = NC == 1 ? 4 : 1;
    for (int g0 = 0; g0 < ng; g0 += UG) {
        T wq[UG][RG][8];
        float xq[UG][NC][8];
#pragma unroll
        for (int v = 0; v < UG; v++) {
            const size_t at = (size_t) min(g0 + v, ng - 1) * 256 + l * 8;
#pragma unroll
            for (int r = 0; r < RG; r++)
#pragma unroll
                for (int u = 0; u < 8; u++) wq[v][r][u] = wr[r][at + u];
#pragma unroll
            for (int n = 0; n < NC; n++)
#pragma unroll
                for (int u = 0; u < 8; u++) xq[v][n][u] = xr[n][at + u];
        }
#pragma unroll
        for (int v = 0; v < UG; v++) {
            if (g0 + v < ng) {
#pragma unroll
                for (int u = 0; u < 8; u++)
#pragma unroll
                    for (int n = 0; n < NC; n++)
#pragma unroll
                        for (int r = 0; r < RG; r++) acc[r][n] = fmaf(WElem<WT>::widen(wq[v][r][u]), xq[v][n][u], acc[r][n]);
            }
        }
    }
    const int st = (K & 255) >> 5;                         // tail group: st < 8 steps
    for (int u = 0; u < st; u++) {
        const size_t at = (size_t) ng * 256 + l * st + u;
#pragma unroll
        for (int n = 0; n < NC; n++) {
            const float xa = xr[n][at];
#pragma unroll
            for (int r = 0; r < RG; r++) acc[r][n] = fmaf(WElem<WT>::widen(wr[r][at]), xa, acc[r][n]);
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

// Decode (one activation row): the same 32 chains per row, software-pipelined.  k_dense_mm issues a batch
// of loads, waits for all of it, consumes it, and only then issues the next: the memory pipe idles half
// the time (2.5 TB/s).  Here two register buffers of UG groups alternate -- the loads of batch b + 1 are in
// flight while batch b is consumed -- in a straight-line loop body (no branch around a load, so the
// compiler's vmcnt waits stay counted; the prefetch past the end is a clamped re-read that is never
// consumed); groups that do not fill a double batch and the tail group run un-pipelined afterwards.
//   grid ceil(M / (HW * RG)), block HW * 32 threads (HW half-waves of RG rows each)
template <int WT, int EPI>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4)))   // <= 128 VGPRs: keeps the operands packed until use
k_dense_mv(const void *__restrict__ wv, int M, int K, const float *__restrict__ xp,
           float *__restrict__ y, const float *__restrict__ resid) {
    typedef typename WElem<WT>::T T;
    const T *w = (const T *) wv;
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5, nhw = blockDim.x >> 5;
    const int m0 = (blockIdx.x * nhw + hw) * RG;
    const T *wr[RG];
#pragma unroll
    for (int r = 0; r < RG; r++) wr[r] = w + (size_t) min(m0 + r, M - 1) * K;
    float acc[RG];
#pragma unroll
    for (int r = 0; r < RG; r++) acc[r] = 0.0f;
    const int ng = K >> 8;
    constexpr int UG = WT == 1 ? 2 : 1;                    // groups per buffer (fp32 weights are twice the registers)
    WVec8<WT> wq[2][UG][RG];
    df4 xq[2][UG][2];
#define LD_LOADB(B, G0)                                                                            \
    _Pragma("unroll")                                                                              \
    for (int v = 0; v < UG; v++) {                                                                 \
        const size_t at = (size_t) min((G0) + v, ng - 1) * 256 + l * 8;                            \
        _Pragma("unroll")                                                                          \
        for (int r = 0; r < RG; r++) wq[B][v][r].load(wr[r] + at);                                 \
        xq[B][v][0] = *(const df4 *) (xp + at); xq[B][v][1] = *(const df4 *) (xp + at + 4);        \
    }
#define LD_CONSUMEB(B)                                                                             \
    _Pragma("unroll")                                                                              \
    for (int v = 0; v < UG; v++)                                                                   \
        _Pragma("unroll")                                                                          \
        for (int u = 0; u < 8; u++)                                                                \
            _Pragma("unroll")                                                                      \
            for (int r = 0; r < RG; r++) wq[B][v][r].fma_into(acc[r], u, xq[B][v][u >> 2][u & 3]);
    const int ngm = ng / (2 * UG) * (2 * UG);              // groups covered by whole double batches
    if (ngm) {
        LD_LOADB(0, 0)
        for (int g0 = 0; g0 < ngm; g0 += 2 * UG) {
            LD_LOADB(1, g0 + UG)
            __builtin_amdgcn_sched_barrier(0);             // the next batch goes out BEFORE the wait for this one
            LD_CONSUMEB(0)
            __builtin_amdgcn_sched_barrier(0);
            LD_LOADB(0, g0 + 2 * UG)
            __builtin_amdgcn_sched_barrier(0);
            LD_CONSUMEB(1)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef LD_LOADB
#undef LD_CONSUMEB
    for (int g = ngm; g < ng; g++) {                       // leftover full groups
        const size_t at = (size_t) g * 256 + l * 8;
        T wl[RG][8];
        float xl[8];
#pragma unroll
        for (int r = 0; r < RG; r++)
#pragma unroll
            for (int u = 0; u < 8; u++) wl[r][u] = wr[r][at + u];
#pragma unroll
        for (int u = 0; u < 8; u++) xl[u] = xp[at + u];
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int r = 0; r < RG; r++) acc[r] = fmaf(WElem<WT>::widen(wl[r][u]), xl[u], acc[r]);
    }
    const int st = (K & 255) >> 5;                         // tail group: st < 8 steps
    for (int u = 0; u < st; u++) {
        const size_t at = (size_t) ng * 256 + l * st + u;
        const float xa = xp[at];
#pragma unroll
        for (int r = 0; r < RG; r++) acc[r] = fmaf(WElem<WT>::widen(wr[r][at]), xa, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < RG; r++) {
        float s = acc[r];
        s += __shfl_xor(s, 8);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (l == 0 && m0 + r < M) {
            if (EPI == EPI_RESID) s = s + resid[m0 + r];
            y[m0 + r] = s;
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
              const float *resid, long resid_stride, hipStream_t st, float *scratch) {
    if (x) hipLaunchKernelGGL(k_dense_perm_act<WT>, dim3(N), dim3(256), 0, st, x, x_stride, w.K, scratch);     // x == nullptr: scratch already holds the prepared rows
    if (N == 1) {
        // small matrices: 4 half-waves per workgroup so that every CU gets one (a 4096-row matrix is 256 workgroups)
        const int nhw = (w.M + 8 * RG - 1) / (8 * RG) >= 512 ? 8 : 4;
        const dim3 g1((w.M + nhw * RG - 1) / (nhw * RG));
        if (epi == EPI_RESID)
            hipLaunchKernelGGL((k_dense_mv<WT, EPI_RESID>), g1, dim3(nhw * 32), 0, st, w.w, w.M, w.K, scratch, y, resid);
        else
            hipLaunchKernelGGL((k_dense_mv<WT, EPI_STORE>), g1, dim3(nhw * 32), 0, st, w.w, w.M, w.K, scratch, y, resid);
        return hipGetLastError();
    }
    const dim3 grid((w.M + 8 * RG - 1) / (8 * RG), (N + NC - 1) / NC);
    if (epi == EPI_RESID)
        hipLaunchKernelGGL((k_dense_mm<WT, NC, EPI_RESID>), grid, dim3(256), 0, st, w.w, w.M, w.K, scratch, N, y, y_stride, resid, resid_stride);
    else
        hipLaunchKernelGGL((k_dense_mm<WT, NC, EPI_STORE>), grid, dim3(256), 0, st, w.w, w.M, w.K, scratch, N, y, y_stride, resid, resid_stride);
    return hipGetLastError();
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// Q4_1 model files.  The reference's Q4_1 path is scalar C end to end (no SIMD branch exists):
//   row layout    [nb floats min][nb floats d][nb * 16 nibble bytes]   (struct of arrays PER ROW, ggml.c:606-615)
//   activations   quantize_row_q4_1 (ggml.c:606-648): min / max of 32, d = (max - min) / 15, id = d ? 1/d : 0,
//                 code = (uint8) round((x - min) * id)
//   dot product   ggml_vec_dot_q4_1 (ggml.c:1584-1626): ONE float accumulator per output, walked over all
//                 blocks and byte pairs in order:  sumf += (d0*q0 + m0) * (d1*p0 + m1) + (d0*q1 + m0) * (d1*p1 + m1)
//                 with every product and sum rounded separately (-std=c11: no contraction).
// There is no parallelism inside an output, so a lane owns a whole weight row; the activation side
// (d1*p + m1, the same for every row) is expanded once per mat-mul by k_q41_act and reaches the lanes
// as SGPR operands.  Weights are regrouped at load into [row-block of 64][block][lane] so that a wave's
// loads are coalesced.  Correct first, not fast: K/2 dependent additions per output.
// ------------------------------------------------------------------------------------------------
typedef uint32_t du32x4 __attribute__((ext_vector_type(4)));
typedef float df32x2 __attribute__((ext_vector_type(2)));

namespace {

// raw rows -> MD[rb][block][lane] = {min, d}, NB[rb][block][lane] = 16 nibble bytes
__global__ void k_q41_repack(const uint8_t *__restrict__ raw, df32x2 *__restrict__ md, du32x4 *__restrict__ nbv, int M, int nb) {
    const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
    const int nrb = (M + 63) / 64;
    if (gid >= (long) nrb * nb * 64) return;
    const int lane = (int) (gid & 63), i = (int) ((gid >> 6) % nb), rb = (int) ((gid >> 6) / nb);
    const int row = rb * 64 + lane;
    df32x2 o = { 0.0f, 0.0f };
    du32x4 q = { 0u, 0u, 0u, 0u };
    if (row < M) {
        const uint8_t *base = raw + (size_t) row * nb * 24;
        uint32_t b0, b1;
        memcpy(&b0, base + 4 * i, 4);
        memcpy(&b1, base + 4 * (nb + i), 4);
        o = df32x2{ __builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1) };
        uint32_t t[4];
        memcpy(t, base + 8 * nb + 16 * i, 16);
        q = du32x4{ t[0], t[1], t[2], t[3] };
    }
    md[gid] = o;
    nbv[gid] = q;
}

// one activation row: quantize to Q4_1 and expand again (the operand ggml_vec_dot_q4_1 builds from src1)
__global__ void k_q41_act(const float *__restrict__ x, long x_stride, int K, float *__restrict__ a) {
    const int n = blockIdx.x;
    const float *xr = x + (size_t) n * x_stride;
    for (int i = threadIdx.x; i < K / 32; i += blockDim.x) {
        float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
        float v[32];
#pragma unroll
        for (int l = 0; l < 32; l++) { v[l] = xr[i * 32 + l]; if (v[l] < mn) mn = v[l]; if (v[l] > mx) mx = v[l]; }
        const float d = (mx - mn) / 15.0f;
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
#pragma unroll
        for (int l = 0; l < 32; l++) {
            const uint32_t q = (uint32_t) (uint8_t) round((double) ((v[l] - mn) * id)) & 0xF;     // pp = vi0 | vi1 << 4: 4 bits survive
            a[(size_t) n * K + i * 32 + l] = d * (float) q + mn;
        }
    }
}

template <int EPI>
__global__ void __launch_bounds__(64)
k_q41_mm(const df32x2 *__restrict__ md, const du32x4 *__restrict__ nbv, int M, int nb, const float *__restrict__ a, int K, int N,
         float *__restrict__ y, long y_stride, const float *__restrict__ resid, long resid_stride) {
    const int lane = threadIdx.x, rb = blockIdx.x, n = blockIdx.y;
    const float *ar = a + (size_t) n * K;                  // wave-uniform: scalar loads
    const size_t base = (size_t) rb * nb * 64 + lane;
    float sum = 0.0f;
    for (int i = 0; i < nb; i++) {
        const df32x2 w = md[base + (size_t) i * 64];
        const du32x4 q = nbv[base + (size_t) i * 64];
        const uint32_t qq[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t b = (qq[j >> 2] >> (8 * (j & 3))) & 0xFF;
            const float f0 = w.y * (float) (b & 0xF) + w.x;
            const float f1 = w.y * (float) (b >> 4) + w.x;
            const float t = f0 * ar[i * 32 + 2 * j], u = f1 * ar[i * 32 + 2 * j + 1];
            sum = sum + (t + u);
        }
    }
    const int m = rb * 64 + lane;
    if (m < M) {
        if (EPI == EPI_RESID) sum = sum + resid[(size_t) n * resid_stride + m];
        y[(size_t) n * y_stride + m] = sum;
    }
}

__global__ void k_embed_q41(const int32_t *__restrict__ tokens, const uint8_t *__restrict__ emb, float *__restrict__ x, int d) {
    const int n = blockIdx.x, nb = d / 32;
    const uint8_t *base = emb + (size_t) tokens[n] * nb * 24;             // dequantize_row_q4_1 (ggml.c:686-717)
    for (int e = threadIdx.x; e < d; e += blockDim.x) {
        const int i = e >> 5, l = e & 31;
        uint32_t b0, b1;
        memcpy(&b0, base + 4 * i, 4);
        memcpy(&b1, base + 4 * (nb + i), 4);
        const float mn = __builtin_bit_cast(float, b0), dd = __builtin_bit_cast(float, b1);
        const uint32_t byte = base[8 * nb + 16 * i + (l >> 1)];
        const int vi = (l & 1) ? (int) (byte >> 4) : (int) (byte & 0xF);
        x[(size_t) n * d + e] = (float) vi * dd + mn;
    }
}

// ggml_quantize_q4_1 (utils.cpp:487-544), one thread per block.  NOT quantize_row_q4_1: here `max` starts
// at std::numeric_limits<float>::min() (the smallest POSITIVE float), as the reference has it.
__global__ void k_quantize_q41_offline(const void *__restrict__ src, int f16, uint8_t *__restrict__ dst, long nrows, int nb) {
    const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nrows * nb) return;
    const long row = gid / nb;
    const int i = (int) (gid % nb);
    float v[32];
    if (f16) { const uint16_t *p = (const uint16_t *) src + gid * 32; for (int l = 0; l < 32; l++) v[l] = h2f(p[l]); }
    else     { const float *p = (const float *) src + gid * 32; for (int l = 0; l < 32; l++) v[l] = p[l]; }
    float mn = 3.402823466e+38f, mx = 1.175494351e-38f;
    for (int l = 0; l < 32; l++) { if (v[l] < mn) mn = v[l]; if (v[l] > mx) mx = v[l]; }
    const float d = (mx - mn) / 15.0f;
    const float id = d != 0.0f ? 1.0f / d : 0.0f;
    uint8_t *base = dst + (size_t) row * nb * 24;
    const uint32_t bm = __builtin_bit_cast(uint32_t, mn), bd = __builtin_bit_cast(uint32_t, d);
    for (int k = 0; k < 4; k++) { base[4 * i + k] = (uint8_t) (bm >> (8 * k)); base[4 * (nb + i) + k] = (uint8_t) (bd >> (8 * k)); }
    for (int l = 0; l < 32; l += 2) {
        const uint8_t q0 = (uint8_t) round((double) ((v[l] - mn) * id)), q1 = (uint8_t) round((double) ((v[l + 1] - mn) * id));
        base[8 * nb + 16 * i + l / 2] = (uint8_t) (q0 | (q1 << 4));
    }
}

}  // namespace

// f16 / f32 rows [row0, row0 + rows) of `w` from raw file-layout rows (load time)
hipError_t launch_dense_perm_rows(const void *raw, DMat &w, int row0, int rows, hipStream_t st) {
    const long total = (long) rows * w.K;
    const size_t esz = w.wtype == 1 ? 2 : 4;
    void *dst = (uint8_t *) w.w + (size_t) row0 * w.K * esz;
    if (w.wtype == 1) hipLaunchKernelGGL(k_dense_perm_rows<1>, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, raw, dst, (long) rows, w.K);
    else              hipLaunchKernelGGL(k_dense_perm_rows<0>, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, raw, dst, (long) rows, w.K);
    return hipGetLastError();
}

hipError_t launch_q41_repack(const uint8_t *raw, DMat &w, hipStream_t st) {
    const int nb = w.K / 32;
    const long total = (long) ((w.M + 63) / 64) * nb * 64;
    hipLaunchKernelGGL(k_q41_repack, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, raw, (df32x2 *) w.w, (du32x4 *) w.w2, w.M, nb);
    return hipGetLastError();
}

hipError_t launch_quantize_q41_offline(const void *src, int f16, uint8_t *dst, long nrows, int nb, hipStream_t st) {
    const long total = nrows * nb;
    hipLaunchKernelGGL(k_quantize_q41_offline, dim3((unsigned) ((total + 127) / 128)), dim3(128), 0, st, src, f16, dst, nrows, nb);
    return hipGetLastError();
}

hipError_t launch_dense_mm(const DMat &w, int epi, const float *x, long x_stride, int N, float *y, long y_stride,
                           const float *resid, long resid_stride, hipStream_t st, float *scratch) {
    if (w.wtype == 3) {
        if (w.K % 32 != 0 || !scratch) return hipErrorInvalidValue;
        hipLaunchKernelGGL(k_q41_act, dim3(N), dim3(64), 0, st, x, x_stride, w.K, scratch);
        const dim3 grid((w.M + 63) / 64, N);
        if (epi == EPI_RESID)
            hipLaunchKernelGGL((k_q41_mm<EPI_RESID>), grid, dim3(64), 0, st, (const df32x2 *) w.w, (const du32x4 *) w.w2, w.M, w.K / 32, scratch, w.K, N, y, y_stride, resid, resid_stride);
        else
            hipLaunchKernelGGL((k_q41_mm<EPI_STORE>), grid, dim3(64), 0, st, (const df32x2 *) w.w, (const du32x4 *) w.w2, w.M, w.K / 32, scratch, w.K, N, y, y_stride, resid, resid_stride);
        return hipGetLastError();
    }
    if (w.K % 32 != 0 || (w.wtype != 0 && w.wtype != 1) || !scratch) return hipErrorInvalidValue;
    if (w.wtype == 1) return N == 1 ? go<1, 1>(w, epi, x, x_stride, N, y, y_stride, resid, resid_stride, st, scratch)
                                    : go<1, 8>(w, epi, x, x_stride, N, y, y_stride, resid, resid_stride, st, scratch);
    return N == 1 ? go<0, 1>(w, epi, x, x_stride, N, y, y_stride, resid, resid_stride, st, scratch)
                  : go<0, 8>(w, epi, x, x_stride, N, y, y_stride, resid, resid_stride, st, scratch);
}

// norm / plain / SiLU*up -> rounded, permuted activation rows in `scratch` (then launch_dense_mm with x = nullptr).
// false = not available for this weight type / row width: use launch_prep + launch_dense_mm.
bool dense_prep_applies(int wtype, int mode, int K) {
    return mode >= 1 && mode <= 3 && (wtype == 0 || wtype == 1) && K % 32 == 0 && (mode != PREP_NORM || K / 16 <= 1024);
}
hipError_t launch_dense_prep(int mode, int wtype, const float *in0, const float *in1, long in_stride, long in1_stride,
                             int K, int N, float *scratch, const uint16_t *T_silu, hipStream_t st) {
    const int nh = K / 16;
    const int nt = mode == PREP_NORM ? (nh + 63) / 64 * 64 : 256;
    const dim3 grid(N, mode == PREP_NORM ? 1 : (nh + nt - 1) / nt);
#define LD_PREP(MODE, WT) hipLaunchKernelGGL((k_dense_prep<MODE, WT>), grid, dim3(nt), 0, st, in0, in1, in_stride, in1_stride, K, scratch, T_silu)
    if (wtype == 1) {
        if (mode == PREP_NORM) LD_PREP(PREP_NORM, 1); else if (mode == PREP_SILU_MUL) LD_PREP(PREP_SILU_MUL, 1); else LD_PREP(PREP_PLAIN, 1);
    } else {
        if (mode == PREP_NORM) LD_PREP(PREP_NORM, 0); else if (mode == PREP_SILU_MUL) LD_PREP(PREP_SILU_MUL, 0); else LD_PREP(PREP_PLAIN, 0);
    }
#undef LD_PREP
    return hipGetLastError();
}

hipError_t launch_embed_dense(const int32_t *tokens, const void *emb, int wtype, float *x, int d, int N, hipStream_t st) {
    if (wtype == 3) hipLaunchKernelGGL(k_embed_q41, dim3(N), dim3(256), 0, st, tokens, (const uint8_t *) emb, x, d);
    else if (wtype == 1) hipLaunchKernelGGL(k_embed_dense<1>, dim3(N), dim3(256), 0, st, tokens, emb, x, d);
    else            hipLaunchKernelGGL(k_embed_dense<0>, dim3(N), dim3(256), 0, st, tokens, emb, x, d);
    return hipGetLastError();
}

}  // namespace lh
