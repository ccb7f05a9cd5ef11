// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the quantized-LLaMA hot path.
//
// Every kernel reproduces the ARITHMETIC ORDER of the reference's x86 AVX2+FMA+F16C build of
// Sources/cpp/ggml.c (file:line cited per kernel), so results are bit-identical to it, not merely
// close: the reference quantizes activations to Q4_0 before every mat-mul (ggml.c:6134-6152), so a
// 1-ulp difference upstream can flip a 4-bit activation code downstream and move a logit by 1e-3.
// This file is compiled with -ffp-contract=off; FMAs appear only where the reference issues
// _mm256_fmadd_ps, and they are written explicitly (fmaf).
//
// HBM layouts (DESIGN.md "Data layout"):
//   Weight matrix W[M][K] Q4_0  ->  "chain-major" tiles of 1280 B = 8 rows x 8 blocks:
//       [row-group g = m/8][chunk c = b/8] { 64 lanes x 16 B nibbles | 64 x 4 B scales }
//     lane = r*8 + k (r = row in group, k = AVX2 lane / "chain" 0..7).  The reference's
//     _mm256_madd_epi16 gives lane k of its 8-float accumulator the elements
//     {2k, 2k+1, 16+2k, 17+2k} of every block (ggml.c:1443-1452); a GPU lane owns exactly that
//     chain, so its fp32 FMA sequence over the blocks is the reference's.
//     dword i of a lane covers blocks (2i, 2i+1) of the chunk: byte p = e_p(block 2i) | e_p(block 2i+1) << 4.
//   Quantized activations ("QA") for one row x[K]:
//       A  : uint32 [chunk c][chain k][block j]  4 signed nibbles (q-8) of chain k, in the LOW nibble
//            of each byte for even j, HIGH nibble for odd j  -> one v_dot8_i32_i4 per block
//       da : float  [block b]
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "llamahip_internal.h"

namespace lh {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float    f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint16_t f2h_bits(float f) {          // _cvtss_sh(x, 0): RNE (ggml.c:162)
    return __half_as_ushort(__float2half_rn(f));
}
__device__ __forceinline__ float h2f_bits(uint16_t h) {          // _cvtsh_ss / table_f32_f16 (ggml.c:161,263-267)
    return __half2float(__ushort_as_half(h));
}

template <int J>
__device__ __forceinline__ float bcast8(float v) {               // lane J of every aligned group of 8 lanes
    // ds_swizzle bit-mask mode: src = ((lane & and_mask) | or_mask) ^ xor_mask, and=0x18, or=J, xor=0
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x18 | (J << 5)));
}

__device__ __forceinline__ double shfl_xor_d(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return __hiloint2double(hi, lo);
}

// block-wide sums / max; `red` is LDS scratch of >= 32 doubles.  All threads get the result.
__device__ double block_sum_d(double v, double *red) {
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_d(v, m);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < nw; i++) s += red[i];
    return s;
}
__device__ float block_max_f(float v, double *red) {
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    float *r = (float *) red;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) r[w] = v;
    __syncthreads();
    float s = r[0];
    for (int i = 1; i < nw; i++) s = fmaxf(s, r[i]);
    return s;
}

// ------------------------------------------------------------------------------------------------
// repack: file-layout Q4_0 rows -> chain-major tiles (load time only)
// ------------------------------------------------------------------------------------------------
// one thread per (row-group, chunk, lane); src = M rows of nb blocks of 20 bytes (unaligned floats)
__global__ void k_repack_q4(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                            int M, int nb, int ngroups, int nchunks) {
    const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long) ngroups * nchunks * 64;
    if (gid >= total) return;
    const int lane = (int) (gid & 63);
    const long tile = gid >> 6;
    const int c = (int) (tile % nchunks);
    const int g = (int) (tile / nchunks);
    const int r = lane >> 3, k = lane & 7;
    const int m = g * 8 + r;
    uint32_t out[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t dw = 0;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int b = c * 8 + 2 * i + half;
            uint32_t e0 = 8, e1 = 8, e2 = 8, e3 = 8;      // q = 8 -> value 0 (padding)
            if (m < M && b < nb) {
                const uint8_t *blk = src + ((size_t) m * nb + b) * 20 + 4;
                const uint32_t lo = blk[k], hi = blk[8 + k];
                e0 = lo & 0xF; e1 = lo >> 4; e2 = hi & 0xF; e3 = hi >> 4;
            }
            const uint32_t packed = e0 | (e1 << 8) | (e2 << 16) | (e3 << 24);
            dw |= packed << (4 * half);
        }
        out[i] = dw;
    }
    uint8_t *t = dst + ((size_t) g * (nchunks + 1) + c) * TILE_BYTES;
    u32x4 v = { out[0], out[1], out[2], out[3] };
    *(u32x4 *) (t + lane * 16) = v;
    // scale of block c*8 + k of row m
    const int bs = c * 8 + k;
    float d = 0.0f;
    if (m < M && bs < nb) {
        const uint8_t *p = src + ((size_t) m * nb + bs) * 20;
        uint32_t bits = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t) p[3] << 24);
        d = __builtin_bit_cast(float, bits);
    }
    *(float *) (t + 1024 + lane * 4) = d;
    if (c == 0) {   // the zero tile closing this row-group (nibbles 8 = value 0, scales 0)
        uint8_t *z = dst + ((size_t) g * (nchunks + 1) + nchunks) * TILE_BYTES;
        u32x4 zv = { 0x88888888u, 0x88888888u, 0x88888888u, 0x88888888u };
        *(u32x4 *) (z + lane * 16) = zv;
        *(float *) (z + 1024 + lane * 4) = 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// embedding gather: ggml_get_rows on a Q4_0 matrix (ggml.c:6760-6785 -> dequantize_row_q4_0 :651-684)
// ------------------------------------------------------------------------------------------------
__global__ void k_embed(const int32_t *__restrict__ tokens, const uint8_t *__restrict__ emb,
                        float *__restrict__ x, int d) {
    const int n = blockIdx.x;
    const int tok = tokens[n];
    const uint8_t *row = emb + (size_t) tok * (d / 32) * 20;
    for (int i = threadIdx.x; i < d / 2; i += blockDim.x) {       // one byte = two elements
        const int b = i >> 4, j = i & 15;
        const uint8_t *blk = row + b * 20;
        const uint32_t bits = blk[0] | (blk[1] << 8) | (blk[2] << 16) | ((uint32_t) blk[3] << 24);
        const float dd = __builtin_bit_cast(float, bits);
        const uint32_t q = blk[4 + j];
        x[(size_t) n * d + 2 * i + 0] = (float) ((int) (q & 0xF) - 8) * dd;
        x[(size_t) n * d + 2 * i + 1] = (float) ((int) (q >> 4) - 8) * dd;
    }
}

// ------------------------------------------------------------------------------------------------
// activation preparation: [norm * weight | silu(gate) * up | plain]  ->  Q4_0 activation operands
// ------------------------------------------------------------------------------------------------
// Quantize 32 floats held in v[] exactly as quantize_row_q4_0's AVX2 branch (ggml.c:456-523):
//   d = amax/7.0f, id = amax != 0 ? 7.0f/amax : 0, q = RNE(x*id) + 8.
// Emits the 8 chain dwords (signed nibbles, low/high by block parity) and returns d.
__device__ __forceinline__ float quant_block(const float *v, int parity, uint32_t *chain /*[8]*/, uint8_t *raw /*[16] or null*/) {
    float amax = 0.0f;
#pragma unroll
    for (int l = 0; l < 32; l++) amax = fmaxf(amax, fabsf(v[l]));
    const float d = amax / 7.0f;
    const float id = (amax != 0.0f) ? 7.0f / amax : 0.0f;
    uint32_t q[32];
#pragma unroll
    for (int l = 0; l < 32; l++) q[l] = (uint32_t) ((int) __builtin_rintf(v[l] * id) + 8);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t e0 = (q[2 * k] - 8) & 0xF, e1 = (q[2 * k + 1] - 8) & 0xF;
        const uint32_t e2 = (q[16 + 2 * k] - 8) & 0xF, e3 = (q[17 + 2 * k] - 8) & 0xF;
        chain[k] = (e0 | (e1 << 8) | (e2 << 16) | (e3 << 24)) << (4 * parity);
    }
    if (raw) {
#pragma unroll
        for (int j = 0; j < 16; j++) raw[j] = (uint8_t) (q[2 * j] | (q[2 * j + 1] << 4));
    }
    return d;
}

// LDS index with one pad float per 32 so "one thread = one block" reads are conflict-free
__device__ __forceinline__ int pidx(int i) { return i + (i >> 5); }

// Produce y[K] in LDS (padded index) according to MODE, all threads of the block cooperating.
//   PREP_PLAIN    y = in0
//   PREP_NORM     y = w * ((float)(x - mean) * scale)          ggml_norm + ggml_mul, ggml.c:5327-5385, :4555
//   PREP_SILU_MUL y = silu_lut(in0) * in1                      ggml.c:1956-1963 + ggml_mul (.mm:678-680)
//   PREP_SUM      y = in0[0] + in0[1] + ... (in order)         attention partial buffers, ggml.c:5553-5577
template <int MODE>
__device__ void make_y(float *ybuf, double *red, const float *__restrict__ in0, const float *__restrict__ in1,
                       int K, const uint16_t *__restrict__ T_silu, int nsum, long sum_stride) {
    const int tid = threadIdx.x, nt = blockDim.x;
    if (MODE == PREP_PLAIN) {
        for (int i = tid; i < K; i += nt) ybuf[pidx(i)] = in0[i];
    } else if (MODE == PREP_SILU_MUL) {
        for (int i = tid; i < K; i += nt) {
            const float s = h2f_bits(T_silu[f2h_bits(in0[i])]);
            ybuf[pidx(i)] = s * in1[i];
        }
    } else if (MODE == PREP_SUM) {
        for (int i = tid; i < K; i += nt) {
            float s = in0[i];
            for (int j = 1; j < nsum; j++) s += in0[(size_t) j * sum_stride + i];
            ybuf[pidx(i)] = s;
        }
    } else {  // PREP_NORM
        double s = 0.0;
        for (int i = tid; i < K; i += nt) s += (double) in0[i];
        const double mean = block_sum_d(s, red) / (double) K;
        double s2 = 0.0;
        for (int i = tid; i < K; i += nt) {
            const double v = (double) in0[i] - mean;
            ybuf[pidx(i)] = (float) v;
            s2 += v * v;
        }
        const double sum2 = block_sum_d(s2, red);
        const float scale = (float) (1.0 / sqrt(sum2 / (double) K + (double) 1e-5f));
        for (int i = tid; i < K; i += nt) {
            const float yv = ybuf[pidx(i)] * scale;
            ybuf[pidx(i)] = in1[i] * yv;
        }
    }
    __syncthreads();
}

// Quantize ybuf[K] into QA operands at (A, da) -- generic pointers (global or LDS).
__device__ void quantize_y(const float *ybuf, int K, int Kp, uint32_t *A, float *da, uint8_t *raw_out) {
    const int nb = K / 32, nbp = Kp / 32;
    for (int b = threadIdx.x; b < nbp; b += blockDim.x) {
        const int c = b >> 3, j = b & 7;
        uint32_t chain[8];
        float d = 0.0f;
        if (b < nb) {
            float v[32];
#pragma unroll
            for (int l = 0; l < 32; l++) v[l] = ybuf[b * 33 + l];
            uint8_t raw[16];
            d = quant_block(v, j & 1, chain, raw_out ? raw : nullptr);
            if (raw_out) {
                uint8_t *o = raw_out + (size_t) b * 20;
                const uint32_t bits = __builtin_bit_cast(uint32_t, d);
                o[0] = bits & 0xFF; o[1] = (bits >> 8) & 0xFF; o[2] = (bits >> 16) & 0xFF; o[3] = bits >> 24;
                for (int t = 0; t < 16; t++) o[4 + t] = raw[t];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) chain[k] = 0;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) A[(c * 8 + k) * 8 + j] = chain[k];
        da[b] = d;
    }
}

// grid.x = rows; dynamic LDS = (K + K/32 + 64) floats + 32 doubles
template <int MODE>
__global__ void k_prep_qa(const float *__restrict__ in0, const float *__restrict__ in1, long in_stride, long in1_stride,
                          int K, int Kp, uint32_t *__restrict__ qa_A, float *__restrict__ qa_d,
                          float *__restrict__ y_out, uint8_t *__restrict__ raw_out,
                          const uint16_t *__restrict__ T_silu, int nsum, long sum_stride) {
    extern __shared__ double smem_d[];
    double *red = smem_d;
    float *ybuf = (float *) (smem_d + 32);
    const int n = blockIdx.x;
    make_y<MODE>(ybuf, red, in0 + (size_t) n * in_stride, in1 ? in1 + (size_t) n * in1_stride : nullptr,
                 K, T_silu, nsum, sum_stride);
    if (y_out)
        for (int i = threadIdx.x; i < K; i += blockDim.x) y_out[(size_t) n * K + i] = ybuf[pidx(i)];
    quantize_y(ybuf, K, Kp, qa_A + (size_t) n * Kp / 4, qa_d + (size_t) n * (Kp / 32),
               raw_out ? raw_out + (size_t) n * (K / 32) * 20 : nullptr);
}

// ------------------------------------------------------------------------------------------------
// Q4_0 x Q4_0 mat-vec / mat-mat:  ggml_compute_forward_mul_mat_q4_0_f32 (ggml.c:5987-6285) with
// ggml_vec_dot_q4_0's AVX2 arithmetic (ggml.c:1415-1466):
//     acc_k = fma(d_w*d_a, (float) isum_k, acc_k)   block after block, k = 0..7
//     y     = ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7))
// One wave = one row-group (8 rows x 8 chains).  Weights stream HBM -> VGPR (non-temporal dwordx4,
// register ring of DEPTH chunks), activations come from LDS (decode) or L1/L2 (multi-column).
// ------------------------------------------------------------------------------------------------
#define LH_STEP(J, WD, AD, DA)                                                                     \
    {                                                                                              \
        const float sc_ = bcast8<J>(sw) * (DA);                                                    \
        const int p_ = __builtin_amdgcn_sdot8((int) (WD), (int) (AD), 0, false);                   \
        acc = fmaf(sc_, (float) p_, acc);                                                          \
    }

__device__ __forceinline__ float fold8(float acc) {
    // ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)) -- float add commutes, so an xor butterfly is exact
    acc += __shfl_xor(acc, 4);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 1);
    return acc;
}

// Decode (one activation row).  QA is staged (PRE_QA) or computed (fused prologue) into LDS.
//   PRE : PRE_QA copy from global | PREP_* compute from fp32 inputs (see make_y)
//   EPI : EPI_STORE y = acc | EPI_RESID y = acc + resid
//   D   : register-ring depth in chunks (1280 B per wave each); RING = false when nchunks <= D (the
//         whole row-group is put in flight before the prologue, no refill), true otherwise
//         (host guarantees nchunks > D).
// The loop bodies are straight-line: loads past the end of the row are redirected to the zero tile
// that closes every row-group (scale 0 -> fma(0*da, p, acc) == acc), never branched around, so the compiler's waitcnt
// pass sees no control-flow merges and emits counted vmcnt waits (2*(D-1) loads stay in flight).
// dynamic LDS: [A: Kp bytes][da: Kp/32 floats] (+ prologue scratch for fused modes)
template <int PRE, int EPI, int D, bool RING>
__global__ void __launch_bounds__(256)
k_gemv(const uint8_t *__restrict__ wt, int ngroups, int nchunks, int M,
       const uint32_t *__restrict__ qa_A, const float *__restrict__ qa_d,
       const float *__restrict__ in0, const float *__restrict__ in1, int K,
       float *__restrict__ y, const float *__restrict__ resid,
       const uint16_t *__restrict__ T_silu, int nsum, long sum_stride) {
    extern __shared__ double smem_d[];
    uint32_t *ldsA = (uint32_t *) smem_d;
    float *ldsD = (float *) (ldsA + nchunks * 64);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int g = blockIdx.x * nw + wave;
    const bool valid = g < ngroups;
    const uint8_t *wbase = wt + (size_t) (valid ? g : 0) * (nchunks + 1) * TILE_BYTES;
    const int last = nchunks - 1;

    u32x4 wq[D];
    float ws[D];
#define LH_LOADW(SLOT, CH)                                                                                   \
    {                                                                                                        \
        const int ch_ = min((CH), nchunks);   /* tile `nchunks` of every row-group is the zero tile */      \
        const uint8_t *tp_ = wbase + (size_t) ch_ * TILE_BYTES;                                              \
        wq[SLOT] = __builtin_nontemporal_load((const u32x4 *) (tp_ + lane * 16));                            \
        ws[SLOT] = __builtin_nontemporal_load((const float *) (tp_ + 1024 + lane * 4));                      \
    }
    // weights do not depend on the activations: put the first D chunks in flight before the
    // prologue so the HBM latency overlaps it
#pragma unroll
    for (int i = 0; i < D; i++) LH_LOADW(i, i)

    if (PRE == PRE_QA) {
        for (int i = tid; i < nchunks * 64; i += blockDim.x) ldsA[i] = qa_A[i];
        for (int i = tid; i < nchunks * 8; i += blockDim.x) ldsD[i] = qa_d[i];
        __syncthreads();
    } else {
        double *red = (double *) (ldsD + nchunks * 8);
        float *ybuf = (float *) (red + 32);
        make_y<PRE>(ybuf, red, in0, in1, K, T_silu, nsum, sum_stride);
        quantize_y(ybuf, K, nchunks * 256, ldsA, ldsD, nullptr);
        __syncthreads();
    }

    const int k = lane & 7;
    float acc = 0.0f;
#define LH_CONSUME(SLOT, CH)                                                                       \
    {                                                                                              \
        const int cl_ = min((CH), last);                                                           \
        const u32x4 w = wq[SLOT];                                                                  \
        const float sw = ws[SLOT];                                                                 \
        const u32x4 *pa = (const u32x4 *) (ldsA + (cl_ * 8 + k) * 8);                              \
        const u32x4 a0 = pa[0], a1 = pa[1];                                                        \
        const f32x4 *pd = (const f32x4 *) (ldsD + cl_ * 8);                                        \
        const f32x4 d0 = pd[0], d1 = pd[1];                                                        \
        const uint32_t w0 = w.x ^ 0x88888888u, w1 = w.y ^ 0x88888888u;                             \
        const uint32_t w2 = w.z ^ 0x88888888u, w3 = w.w ^ 0x88888888u;                             \
        LH_STEP(0, w0, a0.x, d0.x) LH_STEP(1, w0, a0.y, d0.y)                                      \
        LH_STEP(2, w1, a0.z, d0.z) LH_STEP(3, w1, a0.w, d0.w)                                      \
        LH_STEP(4, w2, a1.x, d1.x) LH_STEP(5, w2, a1.y, d1.y)                                      \
        LH_STEP(6, w3, a1.z, d1.z) LH_STEP(7, w3, a1.w, d1.w)                                      \
    }
    int c0 = 0;
    if (RING) {
        do {
#pragma unroll
            for (int i = 0; i < D; i++) {
                LH_CONSUME(i, c0 + i)
                LH_LOADW(i, c0 + D + i)
                __builtin_amdgcn_sched_barrier(0);   // keep slot i+1's first use (and its vmcnt) below this refill
            }
            c0 += D;
        } while (c0 + D < nchunks);
    }
#pragma unroll
    for (int i = 0; i < D; i++) {
        LH_CONSUME(i, c0 + i)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef LH_CONSUME
#undef LH_LOADW

    acc = fold8(acc);
    const int m = g * 8 + (lane >> 3);
    if (valid && k == 0 && m < M) {
        if (EPI == EPI_RESID) acc = acc + resid[m];
        y[m] = acc;
    }
}

// Multi-column (prompt) variant: NC activation rows share every weight tile.  QA is read straight
// from global memory (L1/L2 resident: NC * 288 B per chunk), no LDS, no barriers.
template <int NC, int EPI>
__global__ void __launch_bounds__(256)
k_gemm_nc(const uint8_t *__restrict__ wt, int ngroups, int nchunks, int M,
          const uint32_t *__restrict__ qa_A, const float *__restrict__ qa_d,
          float *__restrict__ y, long y_stride, const float *__restrict__ resid, long resid_stride) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int g = blockIdx.x * nw + wave;
    if (g >= ngroups) return;
    const uint8_t *wbase = wt + (size_t) g * (nchunks + 1) * TILE_BYTES;
    const int k = lane & 7;
    const long strideA = (long) nchunks * 64, strideD = (long) nchunks * 8;
    float accs[NC];
#pragma unroll
    for (int n = 0; n < NC; n++) accs[n] = 0.0f;

    u32x4 w_next = __builtin_nontemporal_load((const u32x4 *) (wbase + lane * 16));
    float s_next = __builtin_nontemporal_load((const float *) (wbase + 1024 + lane * 4));
    for (int c = 0; c < nchunks; c++) {
        const u32x4 w = w_next;
        const float sw = s_next;
        if (c + 1 < nchunks) {
            w_next = __builtin_nontemporal_load((const u32x4 *) (wbase + (size_t) (c + 1) * TILE_BYTES + lane * 16));
            s_next = __builtin_nontemporal_load((const float *) (wbase + (size_t) (c + 1) * TILE_BYTES + 1024 + lane * 4));
        }
        const uint32_t w0 = w.x ^ 0x88888888u, w1 = w.y ^ 0x88888888u;
        const uint32_t w2 = w.z ^ 0x88888888u, w3 = w.w ^ 0x88888888u;
        const float s0 = bcast8<0>(sw), s1 = bcast8<1>(sw), s2 = bcast8<2>(sw), s3 = bcast8<3>(sw);
        const float s4 = bcast8<4>(sw), s5 = bcast8<5>(sw), s6 = bcast8<6>(sw), s7 = bcast8<7>(sw);
#pragma unroll
        for (int n = 0; n < NC; n++) {
            const u32x4 *pa = (const u32x4 *) (qa_A + n * strideA + (c * 8 + k) * 8);
            const u32x4 a0 = pa[0], a1 = pa[1];
            const f32x4 *pd = (const f32x4 *) (qa_d + n * strideD + c * 8);
            const f32x4 d0 = pd[0], d1 = pd[1];
            float acc = accs[n];
#define LH_STEPN(SW, WD, AD, DA) { const float sc_ = (SW) * (DA); const int p_ = __builtin_amdgcn_sdot8((int) (WD), (int) (AD), 0, false); acc = fmaf(sc_, (float) p_, acc); }
            LH_STEPN(s0, w0, a0.x, d0.x) LH_STEPN(s1, w0, a0.y, d0.y)
            LH_STEPN(s2, w1, a0.z, d0.z) LH_STEPN(s3, w1, a0.w, d0.w)
            LH_STEPN(s4, w2, a1.x, d1.x) LH_STEPN(s5, w2, a1.y, d1.y)
            LH_STEPN(s6, w3, a1.z, d1.z) LH_STEPN(s7, w3, a1.w, d1.w)
#undef LH_STEPN
            accs[n] = acc;
        }
    }
    const int m = g * 8 + (lane >> 3);
#pragma unroll
    for (int n = 0; n < NC; n++) {
        float acc = fold8(accs[n]);
        if (k == 0 && m < M) {
            if (EPI == EPI_RESID) acc = acc + resid[(size_t) n * resid_stride + m];
            y[(size_t) n * y_stride + m] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// RoPE + KV append (ggml.c:7076-7131, .mm:586-611).  The reference copies K un-rotated into the
// cache and rotates it there (mode 1); writing the rotated value directly is the same arithmetic.
// cos/sin come from a host table built with the host libm exactly as the reference computes them
// (theta = pow(10000, -i0/n_dims); sincos(p*theta)), kept in double.
// ------------------------------------------------------------------------------------------------
__global__ void k_rope_kv(const float *__restrict__ qkv, long qkv_stride, int d, int dh,
                          const double *__restrict__ sincos_tab /*[n_ctx][dh/2][2] = cos, sin*/,
                          float *__restrict__ qr, float *__restrict__ Kc, float *__restrict__ Vc, int n_past) {
    const int n = blockIdx.x;
    const int pos = n_past + n;
    const float *q = qkv + (size_t) n * qkv_stride, *k = q + d, *v = q + 2 * d;
    const double *tab = sincos_tab + (size_t) pos * dh;
    for (int i = threadIdx.x; i < d / 2; i += blockDim.x) {
        const int e = 2 * i;
        const int pr = (e % dh) >> 1;
        const double cs = tab[2 * pr], sn = tab[2 * pr + 1];
        {
            const double x0 = (double) q[e], x1 = (double) q[e + 1];
            qr[(size_t) n * d + e] = (float) (x0 * cs - x1 * sn);
            qr[(size_t) n * d + e + 1] = (float) (x0 * sn + x1 * cs);
        }
        {
            const double x0 = (double) k[e], x1 = (double) k[e + 1];
            Kc[(size_t) pos * d + e] = (float) (x0 * cs - x1 * sn);
            Kc[(size_t) pos * d + e + 1] = (float) (x0 * sn + x1 * cs);
        }
        Vc[(size_t) pos * d + e] = v[e];
        Vc[(size_t) pos * d + e + 1] = v[e + 1];
    }
}

// ------------------------------------------------------------------------------------------------
// attention for one (head, query row): KQ -> scale -> mask -> soft_max -> V*P
//   KQ      ggml_vec_dot_f32, AVX macro layer (ggml.c:1223-1258, reduce :872-887): 4 vectors x 8 lanes
//           = 32 FMA chains striding 32 elements; a half-wave (32 lanes) owns one key row.
//   scale   ggml.c:6649-6682 ; mask ggml.c:6921-6955 ; soft_max ggml.c:6982-7050 (fp16 exp LUT,
//           double sum -- exact in any order because every term is a multiple of 2^-24 <= 1)
//   V*P     "transposed src0" branch of mul_mat_f32 (ggml.c:5619-5665): the key range is split into
//           nth contiguous chunks, each accumulated by FMA into its own zeroed buffer, buffers added
//           in thread order (ggml.c:5553-5577).
// grid (H, N), block 256, dynamic LDS: [T floats scores][nth*dh floats partials][32 doubles]
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_attn(const float *__restrict__ qr, const float *__restrict__ Kc, const float *__restrict__ Vc,
       float *__restrict__ merged, float *__restrict__ dbg_p, float *__restrict__ dbg_kqv,
       int n_past, int N, int d, int dh, int nth, float kq_scale, const uint16_t *__restrict__ T_exp) {
    extern __shared__ double smem_d[];
    const int h = blockIdx.x, n = blockIdx.y;
    const int T = n_past + N;
    const int tmax = n_past + n;                      // keys 0..tmax are visible
    double *red = smem_d;
    float *sc = (float *) (smem_d + 32);
    float *part = sc + T;
    float *qs = part + nth * dh;
    const int tid = threadIdx.x;

    for (int i = tid; i < dh; i += blockDim.x) qs[i] = qr[(size_t) n * d + h * dh + i];
    __syncthreads();

    // ---- scores
    {
        const int hw = tid >> 5, l = tid & 31, nhw = blockDim.x >> 5;
        for (int t = hw; t <= tmax; t += nhw) {
            const float *kr = Kc + (size_t) t * d + h * dh;
            float s = 0.0f;
            for (int i = 0; i < dh; i += 32) s = fmaf(kr[i + l], qs[i + l], s);
            s += __shfl_xor(s, 8);
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 4);
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            if (l == 0) sc[t] = s * kq_scale;
        }
    }
    __syncthreads();

    // ---- soft_max over keys 0..tmax (masked keys are -inf -> 0)
    float mx = -INFINITY;
    for (int t = tid; t <= tmax; t += blockDim.x) mx = fmaxf(mx, sc[t]);
    mx = block_max_f(mx, red);
    double sum = 0.0;
    for (int t = tid; t <= tmax; t += blockDim.x) {
        const float e = h2f_bits(T_exp[f2h_bits(sc[t] - mx)]);
        sc[t] = e;
        sum += (double) e;
    }
    sum = block_sum_d(sum, red);
    const float inv = (float) (1.0 / sum);
    for (int t = tid; t <= tmax; t += blockDim.x) sc[t] *= inv;
    __syncthreads();
    if (dbg_p) {
        float *o = dbg_p + ((size_t) h * N + n) * T;
        for (int t = tid; t < T; t += blockDim.x) o[t] = t <= tmax ? sc[t] : 0.0f;
    }

    // ---- V*P with the reference's per-thread split of the key range
    {
        const int c = tid % dh, sub = tid / dh, nsub = blockDim.x / dh;
        const int dc = (T + nth - 1) / nth;
        for (int th = sub; th < nth; th += nsub) {
            const int t0 = dc * th;
            int t1 = t0 + dc < T ? t0 + dc : T;
            if (t1 > tmax + 1) t1 = tmax + 1;         // P = 0 beyond tmax: fma(v, 0, acc) == acc
            float acc = 0.0f;
            for (int t = t0; t < t1; t++) acc = fmaf(Vc[(size_t) t * d + h * dh + c], sc[t], acc);
            part[th * dh + c] = acc;
        }
    }
    __syncthreads();
    if (tid < dh) {
        float s = part[tid];
        for (int th = 1; th < nth; th++) s += part[th * dh + tid];
        merged[(size_t) n * d + h * dh + tid] = s;
        if (dbg_kqv) dbg_kqv[((size_t) h * N + n) * dh + tid] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// greedy argmax, lowest index on ties (harness definition of temperature 0)
// ------------------------------------------------------------------------------------------------
__global__ void k_argmax(const float *__restrict__ logits, int V, int32_t *__restrict__ out, int out_idx,
                         int32_t *__restrict__ next_token) {
    __shared__ float bv[1024];
    __shared__ int bi[1024];
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
        const float v = logits[i];
        if (v > best || (v == best && i < idx)) { best = v; idx = i; }
    }
    bv[threadIdx.x] = best; bi[threadIdx.x] = idx;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float v = bv[threadIdx.x + s]; const int i = bi[threadIdx.x + s];
            if (v > bv[threadIdx.x] || (v == bv[threadIdx.x] && i < bi[threadIdx.x])) { bv[threadIdx.x] = v; bi[threadIdx.x] = i; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int r = bi[0] == 0x7fffffff ? 0 : bi[0];
        out[out_idx] = r;
        if (next_token) *next_token = r;
    }
}

// elementwise add (ggml_add, ggml.c:4425-4476) -- only the debug/dump path uses it; the production
// path fuses the residual add into the GEMV epilogue (same single fp32 add)
__global__ void k_add(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ c, long n) {
    const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c[i] = a[i] + b[i];
}

// ================================================================================================
// host-side launchers
// ================================================================================================
#define LH_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

hipError_t init_kernel_attrs() {
    // fused prologues of wide models (K = 22016) need more than the default 64 KB of dynamic LDS
    const int cap = 160 * 1024;
#define LH_ATTR(KERNEL) do { hipError_t e_ = hipFuncSetAttribute((const void *) KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, cap); if (e_ != hipSuccess) return e_; } while (0)
    LH_ATTR(k_prep_qa<PREP_PLAIN>); LH_ATTR(k_prep_qa<PREP_NORM>); LH_ATTR(k_prep_qa<PREP_SILU_MUL>); LH_ATTR(k_prep_qa<PREP_SUM>);
#define LH_ATTR_G(PRE, EPI) LH_ATTR((k_gemv<PRE, EPI, 16, false>)); LH_ATTR((k_gemv<PRE, EPI, 16, true>)); LH_ATTR((k_gemv<PRE, EPI, 13, true>)); \
    LH_ATTR((k_gemv<PRE, EPI, 11, true>)); LH_ATTR((k_gemv<PRE, EPI, 10, true>)); LH_ATTR((k_gemv<PRE, EPI, 8, true>))
    LH_ATTR_G(PRE_QA, EPI_STORE); LH_ATTR_G(PRE_QA, EPI_RESID); LH_ATTR_G(PREP_NORM, EPI_STORE);
    LH_ATTR_G(PREP_PLAIN, EPI_RESID); LH_ATTR_G(PREP_SILU_MUL, EPI_RESID); LH_ATTR_G(PREP_SUM, EPI_RESID);
#undef LH_ATTR_G
    LH_ATTR(k_attn);
#undef LH_ATTR
    return hipSuccess;
}

hipError_t launch_add(const float *a, const float *b, float *c, long n, hipStream_t st) {
    hipLaunchKernelGGL(k_add, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, a, b, c, n);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_repack(const uint8_t *src_aos, uint8_t *dst, int M, int K, hipStream_t st) {
    const int nb = K / 32, ngroups = (M + 7) / 8, nchunks = (nb + 7) / 8;
    const long total = (long) ngroups * nchunks * 64;
    const int bs = 256;
    hipLaunchKernelGGL(k_repack_q4, dim3((unsigned) ((total + bs - 1) / bs)), dim3(bs), 0, st, src_aos, dst, M, nb, ngroups, nchunks);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_embed(const int32_t *tokens, const uint8_t *emb, float *x, int d, int N, hipStream_t st) {
    hipLaunchKernelGGL(k_embed, dim3(N), dim3(256), 0, st, tokens, emb, x, d);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

size_t prep_lds_bytes(int K) { return 32 * sizeof(double) + ((size_t) K + K / 32 + 64) * sizeof(float); }

hipError_t launch_prep(int mode, const float *in0, const float *in1, long in_stride, long in1_stride, int K, int N,
                       uint32_t *qa_A, float *qa_d, float *y_out, uint8_t *raw_out, const uint16_t *T_silu,
                       int nsum, long sum_stride, hipStream_t st) {
    const int Kp = (K + 255) / 256 * 256;
    const size_t lds = prep_lds_bytes(K);
#define LH_PREP(MODE) hipLaunchKernelGGL(k_prep_qa<MODE>, dim3(N), dim3(256), lds, st, in0, in1, in_stride, in1_stride, K, Kp, qa_A, qa_d, y_out, raw_out, T_silu, nsum, sum_stride)
    switch (mode) {
        case PREP_PLAIN:    LH_PREP(PREP_PLAIN); break;
        case PREP_NORM:     LH_PREP(PREP_NORM); break;
        case PREP_SILU_MUL: LH_PREP(PREP_SILU_MUL); break;
        case PREP_SUM:      LH_PREP(PREP_SUM); break;
        default: return hipErrorInvalidValue;
    }
#undef LH_PREP
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

static int pick_waves(int ngroups) {
    // aim for >= 2 workgroups per CU (256 CUs) before growing the workgroup
    if (ngroups >= 4 * 512) return 4;
    if (ngroups >= 2 * 512) return 2;
    return 1;
}

// ring depth for a row of `nchunks` chunks: whole row in flight when it fits 16 slots, else the
// depth in {8..16} that wastes the fewest padded (zero-tile) chunks
static int pick_depth(int nchunks) {
    if (nchunks <= 16) return 16;
    static const int cand[] = { 16, 13, 11, 10, 8 };
    int best = 16, best_waste = 1 << 30;
    for (int d : cand) {
        const int waste = (nchunks + d - 1) / d * d - nchunks;
        if (waste < best_waste) { best_waste = waste; best = d; }
    }
    return best;
}

template <int PRE, int EPI>
static hipError_t launch_gemv_t(const QMat &w, const uint32_t *qa_A, const float *qa_d,
                                const float *in0, const float *in1, float *y, const float *resid,
                                const uint16_t *T_silu, int nsum, long sum_stride, hipStream_t st) {
    const int nw = pick_waves(w.ngroups);
    const int grid = (w.ngroups + nw - 1) / nw;
    size_t lds = (size_t) w.nchunks * 64 * 4 + (size_t) w.nchunks * 8 * 4;
    if (PRE != PRE_QA) lds += prep_lds_bytes(w.K);
    lds = (lds + 15) & ~(size_t) 15;
#define LH_GO(D, RING) hipLaunchKernelGGL((k_gemv<PRE, EPI, D, RING>), dim3(grid), dim3(nw * 64), lds, st, w.tiles, w.ngroups, w.nchunks, w.M, qa_A, qa_d, in0, in1, w.K, y, resid, T_silu, nsum, sum_stride)
    if (w.nchunks <= 16) {
        LH_GO(16, false);
    } else {
        switch (pick_depth(w.nchunks)) {
            case 8:  LH_GO(8, true); break;
            case 10: LH_GO(10, true); break;
            case 11: LH_GO(11, true); break;
            case 13: LH_GO(13, true); break;
            default: LH_GO(16, true); break;
        }
    }
#undef LH_GO
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_gemv(const QMat &w, int pre, int epi, const uint32_t *qa_A, const float *qa_d,
                       const float *in0, const float *in1, float *y, const float *resid,
                       const uint16_t *T_silu, int nsum, long sum_stride, hipStream_t st) {
#define LH_ARGS w, qa_A, qa_d, in0, in1, y, resid, T_silu, nsum, sum_stride, st
    // only the (prologue, epilogue) pairs the forward pass uses are instantiated
    if (pre == PRE_QA && epi == EPI_STORE)        return launch_gemv_t<PRE_QA, EPI_STORE>(LH_ARGS);
    if (pre == PRE_QA && epi == EPI_RESID)        return launch_gemv_t<PRE_QA, EPI_RESID>(LH_ARGS);
    if (pre == PREP_NORM && epi == EPI_STORE)     return launch_gemv_t<PREP_NORM, EPI_STORE>(LH_ARGS);
    if (pre == PREP_PLAIN && epi == EPI_RESID)    return launch_gemv_t<PREP_PLAIN, EPI_RESID>(LH_ARGS);
    if (pre == PREP_SILU_MUL && epi == EPI_RESID) return launch_gemv_t<PREP_SILU_MUL, EPI_RESID>(LH_ARGS);
    if (pre == PREP_SUM && epi == EPI_RESID)      return launch_gemv_t<PREP_SUM, EPI_RESID>(LH_ARGS);
#undef LH_ARGS
    return hipErrorInvalidValue;
}

template <int NC>
static hipError_t launch_gemm_nc_t(const QMat &w, int epi, const uint32_t *qa_A, const float *qa_d,
                                   float *y, long y_stride, const float *resid, long resid_stride, hipStream_t st) {
    const int nw = pick_waves(w.ngroups);
    const int grid = (w.ngroups + nw - 1) / nw;
    if (epi == EPI_RESID)
        hipLaunchKernelGGL((k_gemm_nc<NC, EPI_RESID>), dim3(grid), dim3(nw * 64), 0, st, w.tiles, w.ngroups, w.nchunks, w.M, qa_A, qa_d, y, y_stride, resid, resid_stride);
    else
        hipLaunchKernelGGL((k_gemm_nc<NC, EPI_STORE>), dim3(grid), dim3(nw * 64), 0, st, w.tiles, w.ngroups, w.nchunks, w.M, qa_A, qa_d, y, y_stride, resid, resid_stride);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

// N activation rows (QA precomputed, row stride = Kp bytes / Kp/32 floats): tiles of 16/8/4/2/1 columns
hipError_t launch_gemm(const QMat &w, int epi, const uint32_t *qa_A, const float *qa_d, int N,
                       float *y, long y_stride, const float *resid, long resid_stride, hipStream_t st) {
    const long strideA = (long) w.nchunks * 64, strideD = (long) w.nchunks * 8;
    int n0 = 0;
    while (n0 < N) {
        const int rem = N - n0;
        const uint32_t *A = qa_A + n0 * strideA;
        const float *D = qa_d + n0 * strideD;
        float *yy = y + (size_t) n0 * y_stride;
        const float *rr = resid ? resid + (size_t) n0 * resid_stride : nullptr;
        hipError_t e;
        int step;
        if (rem >= 16)     { step = 16; e = launch_gemm_nc_t<16>(w, epi, A, D, yy, y_stride, rr, resid_stride, st); }
        else if (rem >= 8) { step = 8;  e = launch_gemm_nc_t<8>(w, epi, A, D, yy, y_stride, rr, resid_stride, st); }
        else if (rem >= 4) { step = 4;  e = launch_gemm_nc_t<4>(w, epi, A, D, yy, y_stride, rr, resid_stride, st); }
        else if (rem >= 2) { step = 2;  e = launch_gemm_nc_t<2>(w, epi, A, D, yy, y_stride, rr, resid_stride, st); }
        else               { step = 1;  e = launch_gemv(w, PRE_QA, epi, A, D, nullptr, nullptr, yy, rr, nullptr, 0, 0, st); }
        if (e != hipSuccess) return e;
        n0 += step;
    }
    return hipSuccess;
}

hipError_t launch_rope_kv(const float *qkv, long qkv_stride, int d, int dh, const double *tab,
                          float *qr, float *Kc, float *Vc, int n_past, int N, hipStream_t st) {
    hipLaunchKernelGGL(k_rope_kv, dim3(N), dim3(256), 0, st, qkv, qkv_stride, d, dh, tab, qr, Kc, Vc, n_past);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_attn(const float *qr, const float *Kc, const float *Vc, float *merged, float *dbg_p, float *dbg_kqv,
                       int n_past, int N, int d, int H, int nth, const uint16_t *T_exp, hipStream_t st) {
    const int dh = d / H, T = n_past + N;
    const float kq_scale = 1.0f / sqrtf((float) d / (float) H);          // .mm:620
    const size_t lds = 32 * sizeof(double) + ((size_t) T + (size_t) nth * dh + dh + 16) * sizeof(float);
    hipLaunchKernelGGL(k_attn, dim3(H, N), dim3(256), lds, st, qr, Kc, Vc, merged, dbg_p, dbg_kqv, n_past, N, d, dh, nth, kq_scale, T_exp);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_argmax(const float *logits, int V, int32_t *out, int out_idx, int32_t *next_token, hipStream_t st) {
    hipLaunchKernelGGL(k_argmax, dim3(1), dim3(1024), 0, st, logits, V, out, out_idx, next_token);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace lh
