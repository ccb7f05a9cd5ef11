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
//     dword i of a lane covers blocks (2i, 2i+1) of the chunk: byte p = e_p(block 2i) | e_p(block 2i+1) << 4,
//     each e stored as the signed nibble (q - 8) & 0xF.
//   Quantized activations ("QA") for one row x[K]:
//       A  : uint32 [chunk c][chain k][block j]  4 signed nibbles (q-8) of chain k, in the LOW nibble
//            of each byte for even j, HIGH nibble for odd j  -> one v_dot8_i32_i4 per block
//       da : float  [block b]
//
// Contents, in file order:
//   helpers (fp16 bit conversions, DPP / shuffle reductions, quad broadcast)
//   load time      k_repack_q4, k_quantize_offline, k_embed
//   activations    make_y / quantize_y / k_prep_qa (norm, plain, SiLU*up -> QA), k_prep_fast (register-resident)
//   decode         k_gemv (fused prologues / epilogues, register ring)
//   prompt GEMMs   k_gemm_lds (decode tiles), k_gemm_skinny (2..32 rows: decode tiles, column groups; epilogues
//                  with RoPE + KV append / SiLU*up -> QA), k_gemm_rows (row-lane tiles, SGPR operands),
//                  k_gemm_mfma (+ k_tiles_to_rows, k_tiles_to_mtiles, k_qa_to_qb)
//   attention      k_rope_kv, k_attn (per row), k_attnq_* (lane = query), k_dec_scores, k_decn_scores (short
//                  evals), k_dec_pv_blk (decode and short evals), k_dec_attn_x, k_qkv_attn
//   misc           k_argmax, k_advance, k_add
//   launchers      init_kernel_attrs, launch_* (kernel selection rules live next to the launch)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "llamahip_internal.h"

namespace lh {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float    f32x4 __attribute__((ext_vector_type(4)));
typedef float    f32x2 __attribute__((ext_vector_type(2)));
typedef double   f64x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint16_t f2h_bits(float f) {          // _cvtss_sh(x, 0): RNE (ggml.c:162)
    return __half_as_ushort(__float2half_rn(f));
}
__device__ __forceinline__ float h2f_bits(uint16_t h) {          // _cvtsh_ss / table_f32_f16 (ggml.c:161,263-267)
    return __half2float(__ushort_as_half(h));
}

// The reference's fp16 look-up tables evaluated instead of gathered: table[i] = f2h((float) g((double) h2f(i))) with
// g = x / (1 + exp(-x)) (SiLU, ggml.c:2387) or exp (ggml.c:2386), built by the HOST's libm.  A table has 65 536 entries,
// so whether the device's double-precision exp reproduces every one of them is CHECKED exhaustively at load time
// (launch_check_lut_math); only then do the decode kernels take this path -- it replaces a dependent gather from
// global memory (a full round trip under load) at the tail of the w1|w3 mat-vec and in the middle of soft_max.
__device__ __forceinline__ uint16_t silu_math_bits(uint16_t h) {
    const float f = h2f_bits(h);
    return f2h_bits((float) ((double) f / (1.0 + exp((double) -f))));
}
__device__ __forceinline__ uint16_t exp_math_bits(uint16_t h) {
    return f2h_bits((float) exp((double) h2f_bits(h)));
}

// the same granule for readers on ANY XCD (another launch of the overlapped decode schedule): one write-through (sc1) store
__device__ __forceinline__ void store_tagged_agent(uint64_t *p, uint32_t bits, uint32_t tag) {
    __hip_atomic_store(p, (uint64_t) bits | ((uint64_t) tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one more look of a bounded poll: true = stop looking.  Running out raises the sticky fault word (results are invalid from there on);
// a fault somebody else raised is noticed every 1024 looks, so that one lost hand-off does not make every later poll of the forward
// pass wait out its own bound.
__device__ __forceinline__ bool poll_give_up(int &spins, int limit, uint32_t *fault) {
    if (++spins > limit) { __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return true; }
    return (spins & 1023) == 0 && __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
}
// the granule of a pipeline MAILBOX: the producer may be another device (peer-mapped memory, xGMI) -> system scope both ways
__device__ __forceinline__ void store_tagged_sys(uint64_t *p, uint32_t bits, uint32_t tag) {
    __hip_atomic_store(p, (uint64_t) bits | ((uint64_t) tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint64_t load_granule_sys(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Tag of a granule = {epoch of the forward pass : 24 bits | slot : 8 bits}.  slot = 0 for the embedding row, il + 1 for everything layer
// il (counted from the handle's first layer) produces; a residual-stream row is therefore tagged with the index of the layer that
// CONSUMES it.  The host refuses the tagged hand-offs on handles with more than TAG_MAX_LAYERS layers (llamahip_internal.h); k_bump_epoch skips the epoch
// whose 24 low bits are zero, so no tag ever equals the zero-filled state of a fresh buffer.
#ifndef LH_WATCH
#define LH_WATCH 4          // granules a waiting workgroup looks at per poll (one lane each)
#endif
__device__ __forceinline__ uint32_t make_tag(uint32_t epoch, int slot) { return (epoch << 8) | (uint32_t) slot; }
__device__ __forceinline__ uint32_t next_epoch(uint32_t e) { e += 1u; if ((e & 0xFFFFFFu) == 0u) e += 1u; return e; }

template <int Q>
__device__ __forceinline__ float quad_bcast(float v) {           // lane Q of every quad (DPP quad_perm:[Q,Q,Q,Q])
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), Q | (Q << 2) | (Q << 4) | (Q << 6), 0xF, 0xF, true));
}

__device__ __forceinline__ double shfl_xor_d(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return __hiloint2double(hi, lo);
}

// DPP lane permutations (VALU-only, a few cycles; __shfl goes through the LDS pipe)
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_QUAD_XOR1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141; // lane i <-> 7-i within each 8
constexpr int DPP_ROW_MIRROR = 0x140;      // lane i <-> 15-i within each 16

// sum over the 64 lanes of a wave (any association order: callers only use it where the order is
// immaterial, i.e. double accumulation of fp32 data, see DESIGN.md "norm")
__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp_d<DPP_QUAD_XOR1>(v);
    v += dpp_d<DPP_QUAD_XOR2>(v);
    v += dpp_d<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_d<DPP_ROW_MIRROR>(v);                       // every lane of a 16-lane row holds its row sum
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max_f(float v) {
    v = fmaxf(v, dpp_f<DPP_QUAD_XOR1>(v));
    v = fmaxf(v, dpp_f<DPP_QUAD_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v));
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// The reference's reduction of its 32 FMA chains (GGML_F32x8_REDUCE, ggml.c:872-887: xor 8, 16, 4, 1, 2 over
// 32 lanes), delivered to lane 0 of each 32-lane half only -- which is all the callers use.  Shifts instead
// of exchanges (lane i reads i + 8 / i + 4: DPP within a 16-lane row) leave one step that crosses rows.
__device__ __forceinline__ float tree32_to_lane0(float s) {
    s += dpp_f<0x108>(s);                     // row_shl:8
    s += __shfl_xor(s, 16);
    s += dpp_f<0x104>(s);                     // row_shl:4
    s += dpp_f<DPP_QUAD_XOR1>(s);
    s += dpp_f<DPP_QUAD_XOR2>(s);
    return s;
}

// max over lanes 0..31 of a wave (the result is wave-uniform; lanes 32..63 may be inactive): four DPP
// steps and two readlanes -- no LDS round trips (the epilogues that quantize one 32-element block sit at
// the tail of a launch)
__device__ __forceinline__ float max_lanes_0_31(float v) {
    v = fmaxf(v, dpp_f<DPP_QUAD_XOR1>(v));
    v = fmaxf(v, dpp_f<DPP_QUAD_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v));
    const int b = __builtin_bit_cast(int, v);
    return fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16)));
}

// block-wide sums / max; `red` is LDS scratch of >= 32 doubles.  All threads get the result.
// Successive calls alternate between the two halves of `red`, so one barrier per call suffices
// (a slot is rewritten only two calls later, after every wave passed the barrier in between).
__device__ double block_sum_d(double v, double *red, int phase = 0) {
    v = wave_sum_d(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    double *r = red + (phase & 1) * 16;
    if ((threadIdx.x & 63) == 0) r[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < nw; i++) s += r[i];
    return s;
}
// two sums with one barrier (at most 8 waves: 16 doubles per phase)
__device__ void block_sum_d2(double &a, double &b, double *red, int phase = 0) {
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    double *r = red + (phase & 1) * 16;
    if ((threadIdx.x & 63) == 0) { r[2 * w] = a; r[2 * w + 1] = b; }
    __syncthreads();
    double sa = 0.0, sb = 0.0;
    for (int i = 0; i < nw; i++) { sa += r[2 * i]; sb += r[2 * i + 1]; }
    a = sa; b = sb;
}
__device__ float block_max_f(float v, double *red, int phase = 0) {
    v = wave_max_f(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    float *r = (float *) (red + (phase & 1) * 16);
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
// gmap: tile group of logical row-group lg is  (lg / 4) * 8 + goff + lg % 4  when gmap != 0 (the
// w1|w3 interleave: every 8 consecutive tile groups hold 32 rows of w1 then the same 32 rows of w3,
// so one 8-wave workgroup owns gate and up of one whole Q4_0 block of the FFN activation), else lg.
__global__ void k_repack_q4(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                            int M, int nb, int ngroups, int nchunks, int gmap, int goff) {
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
    const int tg = gmap ? (g >> 2) * 8 + goff + (g & 3) : g;
    uint8_t *t = dst + ((size_t) tg * (nchunks + 1) + c) * TILE_BYTES;
    // stored as signed 4-bit values: (q - 8) & 0xF == q ^ 8, i.e. the dword ^ 0x88888888 -- ready for v_dot8_i32_i4
    u32x4 v = { out[0] ^ 0x88888888u, out[1] ^ 0x88888888u, out[2] ^ 0x88888888u, out[3] ^ 0x88888888u };
    *(u32x4 *) (t + lane * 16) = v;
    // scale of block c*8 + k of row m
    const int bs = c * 8 + k;
    float d = 0.0f;
    if (m < M && bs < nb) {
        const uint8_t *p = src + ((size_t) m * nb + bs) * 20;
        uint32_t bits = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t) p[3] << 24);
        d = __builtin_bit_cast(float, bits);
    }
    // scales of a row are stored as [s0,s4,s1,s5,s2,s6,s3,s7]: lane (r,k) later loads the pair
    // (s[k&3], s[4+(k&3)]), so every QUAD of lanes holds all 8 scales of its row and the per-block
    // scale is one v_mul_f32 with a quad_perm DPP broadcast (no LDS-pipe swizzle)
    *(float *) (t + 1024 + (r * 8 + (k & 3) * 2 + (k >> 2)) * 4) = d;
    if (c == 0) {   // the zero tile closing this row-group (values 0, scales 0)
        uint8_t *z = dst + ((size_t) tg * (nchunks + 1) + nchunks) * TILE_BYTES;
        u32x4 zv = { 0u, 0u, 0u, 0u };
        *(u32x4 *) (z + lane * 16) = zv;
        *(float *) (z + 1024 + lane * 4) = 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// offline quantizer: ggml_quantize_q4_0 (utils.cpp:431-485) -- one thread per 32-element block.
// NOT the runtime activation quantizer: d = amax / 7, id = d ? 1 / d : 0, round half away from zero.
// src: fp32 (f16 = 0) or IEEE half (f16 = 1, widened exactly as ggml_fp16_to_fp32 does).
// ------------------------------------------------------------------------------------------------
__global__ void k_quantize_offline(const void *__restrict__ src, int f16, uint8_t *__restrict__ dst, long nblocks) {
    const long b = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    float x[32];
    if (f16) {
        const uint16_t *p = (const uint16_t *) src + b * 32;
#pragma unroll
        for (int i = 0; i < 32; i++) x[i] = h2f_bits(p[i]);
    } else {
        const float *p = (const float *) src + b * 32;
#pragma unroll
        for (int i = 0; i < 32; i++) x[i] = p[i];
    }
    float amax = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; i++) amax = fmaxf(amax, fabsf(x[i]));
    const float d = amax / 7.0f;
    const float id = d != 0.0f ? 1.0f / d : 0.0f;
    uint8_t *o = dst + b * 20;
    const uint32_t db = __builtin_bit_cast(uint32_t, d);
    o[0] = db & 0xFF; o[1] = (db >> 8) & 0xFF; o[2] = (db >> 16) & 0xFF; o[3] = db >> 24;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const float v0 = x[2 * j] * id, v1 = x[2 * j + 1] * id;
        const int q0 = (int) (int8_t) roundf(v0) + 8, q1 = (int) (int8_t) roundf(v1) + 8;     // C round(): half away from zero
        o[4 + j] = (uint8_t) (q0 | (q1 << 4));
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
    // grid.y slices the row (one dependent round trip per workgroup instead of d/512 per thread)
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < d / 2; i += gridDim.y * blockDim.x) {       // one byte = two elements
        const int b = i >> 4, j = i & 15;
        const uint8_t *blk = row + b * 20;
        const uint32_t bits = blk[0] | (blk[1] << 8) | (blk[2] << 16) | ((uint32_t) blk[3] << 24);
        const float dd = __builtin_bit_cast(float, bits);
        const uint32_t q = blk[4 + j];
        x[(size_t) n * d + 2 * i + 0] = (float) ((int) (q & 0xF) - 8) * dd;
        x[(size_t) n * d + 2 * i + 1] = (float) ((int) (q >> 4) - 8) * dd;
    }
}

// decode: the embedding row of one token, plus the {sum x, sum x^2} pair (double) the first layer's norm-fused
// mat-vec folds instead of reducing the row itself (PREP_NORMP).  One workgroup; same dequantization.
__global__ void __launch_bounds__(256)
k_embed_part(const int32_t *__restrict__ tokens, const uint8_t *__restrict__ emb, float *__restrict__ x, int d,
             f64x2 *__restrict__ part_out, uint32_t *__restrict__ epoch, uint64_t *__restrict__ xt,
             const uint64_t *token_mb, const int32_t *__restrict__ st, uint32_t *fault, int n_vocab) {
    __shared__ double red[32];
    __shared__ int tok_s;
    // token_mb (first stage of a pipeline with device-side mailboxes): the token arrives as one tagged granule from the last stage's
    // pick kernel (tag: the position it is for, st[0] + 1); one thread polls, bounded
    if (token_mb) {
        if (threadIdx.x == 0) {
            const uint32_t want = make_tag((uint32_t) st[0] + 1u, 0);
            int spins = 0;
            uint64_t g;
            for (;;) {
                g = load_granule_sys(token_mb);
                if ((uint32_t) (g >> 32) == want) break;
                __builtin_amdgcn_s_sleep(16);
                if (poll_give_up(spins, 1 << 27, fault)) break;
            }
            const uint32_t t = (uint32_t) g;
            tok_s = t < (uint32_t) n_vocab ? (int) t : 0;        // (a poll that ran out: the fault word is up, keep the gather in bounds)
        }
        __syncthreads();
    }
    const int tok = token_mb ? tok_s : tokens[0];
    // xt (overlapped decode schedule): the row also leaves as tagged granules, slot 0 of the epoch k_bump_epoch set before this launch
    const uint32_t tag = xt ? make_tag(epoch[0], 0) : 0u;
    const uint8_t *row = emb + (size_t) tok * (d / 32) * 20;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < d / 2; i += blockDim.x) {       // one byte = two elements
        const int b = i >> 4, j = i & 15;
        const uint8_t *blk = row + b * 20;
        const uint32_t bits = blk[0] | (blk[1] << 8) | (blk[2] << 16) | ((uint32_t) blk[3] << 24);
        const float dd = __builtin_bit_cast(float, bits);
        const uint32_t q = blk[4 + j];
        const float v0 = (float) ((int) (q & 0xF) - 8) * dd, v1 = (float) ((int) (q >> 4) - 8) * dd;
        x[2 * i + 0] = v0;
        x[2 * i + 1] = v1;
        if (xt) { store_tagged_agent(xt + 2 * i, __builtin_bit_cast(uint32_t, v0), tag); store_tagged_agent(xt + 2 * i + 1, __builtin_bit_cast(uint32_t, v1), tag); }
        s1 += (double) v0; s1 += (double) v1;
        s2 += (double) v0 * (double) v0; s2 += (double) v1 * (double) v1;
    }
    s1 = block_sum_d(s1, red, 0);
    s2 = block_sum_d(s2, red, 1);
    if (threadIdx.x == 0) part_out[0] = f64x2{ s1, s2 };
    if (epoch && !xt && threadIdx.x == 0) epoch[0] = next_epoch(epoch[0]);       // one forward pass = one epoch of the tagged hand-offs (k_qkv_attn)
}

// a residual-stream row that arrived behind a kernel boundary (pipeline stage input) re-published as tagged granules, slot 0
__global__ void __launch_bounds__(256)
k_tag_row(const float *__restrict__ x, int d, const uint32_t *__restrict__ epoch, uint64_t *__restrict__ xt) {
    const uint32_t tag = make_tag(epoch[0], 0);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d; i += gridDim.x * blockDim.x) store_tagged_agent(xt + i, __builtin_bit_cast(uint32_t, x[i]), tag);
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
// Global loads are issued in batches of LB float4 per thread before anything consumes them (indices
// clamped, never branched around), so a prologue costs a couple of L2 round trips instead of one per
// element: these prologues run inside the GEMV kernels, in front of the weight stream.
constexpr int LB_DEFAULT = 8;

// Optional phase-timing probe (tools/gemv_phases.py), compiled in only with -DLH_PHASE_PROBE=1
// (`make probe` -> libllamahip_probe.so): one thread of the middle workgroup of every k_gemv launch
// stores s_memtime at the phase boundaries.  It is NOT in the product build: merely carrying the
// probe pointer through the kernel cost the 22-deep ring variant 60 VGPRs and pushed it into scratch.
// layout: [0] = launch counter, [1] = capacity, entry e at 8*(1+e): {5 stamps, ngroups, nchunks, PRE*16+EPI}
__device__ unsigned long long *g_phase_probe = nullptr;
#if LH_PHASE_PROBE
#if LH_PHASE_PROBE == 3      /* timeline: EVERY workgroup appends {5 stamps, kind << 32 | block, ngroups << 32 | nchunks, wall clock} */
#define LH_STAMP(IDX) do { probe_t[IDX] = __builtin_readcyclecounter(); } while (0)
#define LH_STAMP2(IDX) do { } while (0)
#elif LH_PHASE_PROBE == 2      /* prologue detail: entry | ring issued | mean known | scale known | prologue done */
#define LH_STAMP(IDX) do { if (probe_e && (IDX) < 2) probe_e[IDX] = __builtin_readcyclecounter(); } while (0)
#define LH_STAMP2(IDX) do { if (probe_e) probe_e[IDX] = __builtin_readcyclecounter(); } while (0)
#else
#define LH_STAMP(IDX) do { if (probe_e) probe_e[IDX] = __builtin_readcyclecounter(); } while (0)
#define LH_STAMP2(IDX) do { } while (0)
#endif
#else
#define LH_STAMP(IDX) do { } while (0)
#define LH_STAMP2(IDX) do { } while (0)
#endif


template <int MODE, int LB = LB_DEFAULT>
__device__ void make_y(float *ybuf, double *red, const float *__restrict__ in0, const float *__restrict__ in1,
                       int K, const uint16_t *__restrict__ T_silu) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n4 = K >> 2;                                   // K is a multiple of 32
    const f32x4 *a4 = (const f32x4 *) in0;
    const f32x4 *b4 = (const f32x4 *) in1;
    if (MODE == PREP_PLAIN) {
        for (int base = tid; base < n4; base += nt * LB) {
            f32x4 v[LB];
#pragma unroll
            for (int u = 0; u < LB; u++) v[u] = a4[min(base + u * nt, n4 - 1)];
#pragma unroll
            for (int u = 0; u < LB; u++) {
                const int g = base + u * nt;
                if (g < n4) { float *o = ybuf + pidx(4 * g); o[0] = v[u].x; o[1] = v[u].y; o[2] = v[u].z; o[3] = v[u].w; }
            }
        }
    } else if (MODE == PREP_SILU_MUL) {
        for (int base = tid; base < n4; base += nt * LB) {
            f32x4 ga[LB], up[LB];
#pragma unroll
            for (int u = 0; u < LB; u++) { const int g = min(base + u * nt, n4 - 1); ga[u] = a4[g]; up[u] = b4[g]; }
            uint16_t lut[LB][4];
#pragma unroll
            for (int u = 0; u < LB; u++) {
                lut[u][0] = T_silu[f2h_bits(ga[u].x)]; lut[u][1] = T_silu[f2h_bits(ga[u].y)];
                lut[u][2] = T_silu[f2h_bits(ga[u].z)]; lut[u][3] = T_silu[f2h_bits(ga[u].w)];
            }
#pragma unroll
            for (int u = 0; u < LB; u++) {
                const int g = base + u * nt;
                if (g < n4) {
                    float *o = ybuf + pidx(4 * g);
                    o[0] = h2f_bits(lut[u][0]) * up[u].x; o[1] = h2f_bits(lut[u][1]) * up[u].y;
                    o[2] = h2f_bits(lut[u][2]) * up[u].z; o[3] = h2f_bits(lut[u][3]) * up[u].w;
                }
            }
        }
    } else {  // PREP_NORM
        double s = 0.0;
        for (int base = tid; base < n4; base += nt * LB) {
            f32x4 v[LB];
#pragma unroll
            for (int u = 0; u < LB; u++) v[u] = a4[min(base + u * nt, n4 - 1)];
#pragma unroll
            for (int u = 0; u < LB; u++) {
                const int g = base + u * nt;
                if (g < n4) {
                    float *o = ybuf + pidx(4 * g);
                    o[0] = v[u].x; o[1] = v[u].y; o[2] = v[u].z; o[3] = v[u].w;
                    s += (double) v[u].x; s += (double) v[u].y; s += (double) v[u].z; s += (double) v[u].w;
                }
            }
        }
        const double mean = block_sum_d(s, red, 0) / (double) K;      // (block_sum_d syncs: ybuf is visible)
        double s2 = 0.0;
        for (int i = tid; i < K; i += nt) {
            const double v = (double) ybuf[pidx(i)] - mean;
            ybuf[pidx(i)] = (float) v;
            s2 += v * v;
        }
        const double sum2 = block_sum_d(s2, red, 1);
        const float scale = (float) (1.0 / sqrt(sum2 / (double) K + (double) 1e-5f));
        for (int base = tid; base < n4; base += nt * LB) {
            f32x4 w[LB];
#pragma unroll
            for (int u = 0; u < LB; u++) w[u] = b4[min(base + u * nt, n4 - 1)];
#pragma unroll
            for (int u = 0; u < LB; u++) {
                const int g = base + u * nt;
                if (g < n4) {
                    float *o = ybuf + pidx(4 * g);
                    o[0] = w[u].x * (o[0] * scale); o[1] = w[u].y * (o[1] * scale);
                    o[2] = w[u].z * (o[2] * scale); o[3] = w[u].w * (o[3] * scale);
                }
            }
        }
    }
    __syncthreads();
}

// cooperative global -> LDS copy of n4 16-byte granules, all loads of a batch in flight together
__device__ __forceinline__ void copy_g2l(uint32_t *dst, const uint32_t *__restrict__ src, int n4) {
    constexpr int LB = LB_DEFAULT;
    const u32x4 *s4 = (const u32x4 *) src;
    u32x4 *d4 = (u32x4 *) dst;
    for (int base = threadIdx.x; base < n4; base += blockDim.x * LB) {
        u32x4 v[LB];
#pragma unroll
        for (int u = 0; u < LB; u++) v[u] = s4[min(base + u * (int) blockDim.x, n4 - 1)];
#pragma unroll
        for (int u = 0; u < LB; u++) { const int g = base + u * (int) blockDim.x; if (g < n4) d4[g] = v[u]; }
    }
}

// Quantize ybuf[K] into QA operands at (A, da) -- generic pointers (global or LDS).  One thread per
// block; values are re-read from LDS (once for amax, once per chain) instead of being held in 64
// registers, because this runs inside the GEMV kernels while the weight ring is live.
__device__ void quantize_y(const float *ybuf, int K, int Kp, uint32_t *A, float *da, uint8_t *raw_out) {
    const int nb = K / 32, nbp = Kp / 32;
    for (int b = threadIdx.x; b < nbp; b += blockDim.x) {
        const int c = b >> 3, j = b & 7;
        const float *v = ybuf + b * 33;
        float d = 0.0f, id = 0.0f;
        if (b < nb) {
            float amax = 0.0f;
#pragma unroll 8
            for (int l = 0; l < 32; l++) amax = fmaxf(amax, fabsf(v[l]));
            d = amax / 7.0f;                                   // ggml.c:479
            id = (amax != 0.0f) ? 7.0f / amax : 0.0f;          // ggml.c:482
        }
        uint8_t *o = raw_out ? raw_out + (size_t) b * 20 : nullptr;
        if (o && b < nb) {
            const uint32_t bits = __builtin_bit_cast(uint32_t, d);
            o[0] = bits & 0xFF; o[1] = (bits >> 8) & 0xFF; o[2] = (bits >> 16) & 0xFF; o[3] = bits >> 24;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint32_t dw = 0;
            if (b < nb) {
                const int q0 = (int) __builtin_rintf(v[2 * k] * id), q1 = (int) __builtin_rintf(v[2 * k + 1] * id);
                const int q2 = (int) __builtin_rintf(v[16 + 2 * k] * id), q3 = (int) __builtin_rintf(v[17 + 2 * k] * id);
                dw = ((uint32_t) (q0 & 0xF) | ((uint32_t) (q1 & 0xF) << 8) | ((uint32_t) (q2 & 0xF) << 16) | ((uint32_t) (q3 & 0xF) << 24)) << (4 * (j & 1));
                if (o) {            // file-layout bytes: qs[k] = elements (2k, 2k+1), qs[8+k] = (16+2k, 17+2k), each q + 8
                    o[4 + k] = (uint8_t) ((q0 + 8) | ((q1 + 8) << 4));
                    o[12 + k] = (uint8_t) ((q2 + 8) | ((q3 + 8) << 4));
                }
            }
            A[(c * 8 + k) * 8 + j] = dw;
        }
        da[b] = d;
    }
}

// grid.x = rows; dynamic LDS = (K + K/32 + 64) floats + 32 doubles
template <int MODE>
__global__ void k_prep_qa(const float *__restrict__ in0, const float *__restrict__ in1, long in_stride, long in1_stride,
                          int K, int Kp, uint32_t *__restrict__ qa_A, float *__restrict__ qa_d,
                          float *__restrict__ y_out, uint8_t *__restrict__ raw_out,
                          const uint16_t *__restrict__ T_silu) {
    extern __shared__ double smem_d[];
    double *red = smem_d;
    float *ybuf = (float *) (smem_d + 32);
    const int n = blockIdx.x;
    make_y<MODE>(ybuf, red, in0 + (size_t) n * in_stride, in1 ? in1 + (size_t) n * in1_stride : nullptr,
                 K, T_silu);
    if (y_out)
        for (int i = threadIdx.x; i < K; i += blockDim.x) y_out[(size_t) n * K + i] = ybuf[pidx(i)];
    quantize_y(ybuf, K, Kp, qa_A + (size_t) n * Kp / 4, qa_d + (size_t) n * (Kp / 32),
               raw_out ? raw_out + (size_t) n * (K / 32) * 20 : nullptr);
}

// Register-resident variant of k_prep_qa (same arithmetic, the production path whenever no fp32 / raw
// side output is wanted): one thread owns one HALF-BLOCK (16 contiguous elements), the two halves of a
// Q4_0 block sit in lanes t and t^1 and exchange through DPP -- no LDS staging of y and no
// one-thread-per-block serial quantizer.  PLAIN and SILU_MUL have no row-wide reduction, so a row is
// spread over gridDim.y workgroups (a 9-row chunk of F = 11008 used to run on 9 workgroups);
// NORM keeps the whole row in one workgroup (blockDim >= K/16, host-checked).
//   grid (rows, slices); block = multiple of 64
template <int MODE>
__global__ void __launch_bounds__(1024)
k_prep_fast(const float *__restrict__ in0, const float *__restrict__ in1, long in_stride, long in1_stride,
            int K, int Kp, uint32_t *__restrict__ qa_A, float *__restrict__ qa_d, const uint16_t *__restrict__ T_silu) {
    __shared__ double red[32];
    const int n = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int nh = K >> 4, nbp = Kp >> 5;
    const int hi = blockIdx.y * nt + tid;                    // half-block index; block = hi >> 1, half = hi & 1
    const bool live = hi < nh;
    const int hc = min(hi, nh - 1);
    const f32x4 *a4 = (const f32x4 *) (in0 + (size_t) n * in_stride) + hc * 4;
    f32x4 xa[4], xb[4];
#pragma unroll
    for (int v = 0; v < 4; v++) xa[v] = a4[v];
    if (MODE == PREP_NORM) {
#pragma unroll
        for (int v = 0; v < 4; v++) xb[v] = ((const f32x4 *) in1)[hc * 4 + v];
    } else if (MODE == PREP_SILU_MUL) {
        const f32x4 *b4 = (const f32x4 *) (in1 + (size_t) n * in1_stride) + hc * 4;
#pragma unroll
        for (int v = 0; v < 4; v++) xb[v] = b4[v];
    }
    uint32_t *A = qa_A + (size_t) n * (Kp / 4);
    float *da = qa_d + (size_t) n * nbp;
    if (MODE == PREP_NORM) {
        // ggml_norm + ggml_mul (ggml.c:5327-5385, :4555)
        double s1 = 0.0;
        if (live) {
#pragma unroll
            for (int v = 0; v < 4; v++) { s1 += (double) xa[v].x; s1 += (double) xa[v].y; s1 += (double) xa[v].z; s1 += (double) xa[v].w; }
        }
        const double mean = block_sum_d(s1, red, 0) / (double) K;
        double s2 = 0.0;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const double v0 = (double) xa[v].x - mean, v1 = (double) xa[v].y - mean;
            const double v2 = (double) xa[v].z - mean, v3 = (double) xa[v].w - mean;
            xa[v].x = (float) v0; xa[v].y = (float) v1; xa[v].z = (float) v2; xa[v].w = (float) v3;
            if (live) { s2 += v0 * v0; s2 += v1 * v1; s2 += v2 * v2; s2 += v3 * v3; }
        }
        const double sum2 = block_sum_d(s2, red, 1);
        const float scale = (float) (1.0 / sqrt(sum2 / (double) K + (double) 1e-5f));
#pragma unroll
        for (int v = 0; v < 4; v++) {
            xa[v].x = xb[v].x * (xa[v].x * scale); xa[v].y = xb[v].y * (xa[v].y * scale);
            xa[v].z = xb[v].z * (xa[v].z * scale); xa[v].w = xb[v].w * (xa[v].w * scale);
        }
    } else if (MODE == PREP_SILU_MUL) {
        // silu through the fp16 table (ggml.c:1956-1963), then ggml_mul (.mm:678-680)
        uint16_t lut[4][4];
#pragma unroll
        for (int v = 0; v < 4; v++) {
            lut[v][0] = T_silu[f2h_bits(xa[v].x)]; lut[v][1] = T_silu[f2h_bits(xa[v].y)];
            lut[v][2] = T_silu[f2h_bits(xa[v].z)]; lut[v][3] = T_silu[f2h_bits(xa[v].w)];
        }
#pragma unroll
        for (int v = 0; v < 4; v++) {
            xa[v].x = h2f_bits(lut[v][0]) * xb[v].x; xa[v].y = h2f_bits(lut[v][1]) * xb[v].y;
            xa[v].z = h2f_bits(lut[v][2]) * xb[v].z; xa[v].w = h2f_bits(lut[v][3]) * xb[v].w;
        }
    }
    // quantize_row_q4_0, AVX2 branch (ggml.c:456-523), two lanes per block
    float amax = 0.0f;
#pragma unroll
    for (int v = 0; v < 4; v++)
        amax = fmaxf(fmaxf(fmaxf(amax, fabsf(xa[v].x)), fabsf(xa[v].y)), fmaxf(fabsf(xa[v].z), fabsf(xa[v].w)));
    amax = fmaxf(amax, dpp_f<DPP_QUAD_XOR1>(amax));          // partner half (lane ^ 1); K/16 is even: both live or both dead
    const float dd = amax / 7.0f;
    const float id = (amax != 0.0f) ? 7.0f / amax : 0.0f;
    uint32_t pr[8];                                          // pair p = elements (2p, 2p+1) of this half -> one 16-bit field
#pragma unroll
    for (int v = 0; v < 4; v++) {
        const uint32_t n0 = (uint32_t) ((int) __builtin_rintf(xa[v].x * id)) & 0xF, n1 = (uint32_t) ((int) __builtin_rintf(xa[v].y * id)) & 0xF;
        const uint32_t n2 = (uint32_t) ((int) __builtin_rintf(xa[v].z * id)) & 0xF, n3 = (uint32_t) ((int) __builtin_rintf(xa[v].w * id)) & 0xF;
        pr[2 * v] = n0 | (n1 << 8);
        pr[2 * v + 1] = n2 | (n3 << 8);
    }
    // chain k of the block = pair k of half 0 (low 16 bits) | pair k of half 1 (high 16 bits)
    const int half = hi & 1, b = hi >> 1, c = b >> 3, j = b & 7;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t other = (uint32_t) __builtin_amdgcn_mov_dpp((int) pr[k], DPP_QUAD_XOR1, 0xF, 0xF, true);
        const uint32_t dw = (half ? (other | (pr[k] << 16)) : (pr[k] | (other << 16))) << (4 * (j & 1));
        if (live && (k >> 2) == half) A[(c * 8 + k) * 8 + j] = dw;          // half 0 stores chains 0..3, half 1 chains 4..7
    }
    if (live && half == 0) da[b] = dd;
    // zero the padded blocks (K not a multiple of 256)
    if (blockIdx.y == 0)
        for (int pb = K / 32 + tid; pb < nbp; pb += nt) {
            const int pc = pb >> 3, pj = pb & 7;
#pragma unroll
            for (int k = 0; k < 8; k++) A[(pc * 8 + k) * 8 + pj] = 0;
            da[pb] = 0.0f;
        }
}

// ------------------------------------------------------------------------------------------------
// Q4_0 x Q4_0 mat-vec / mat-mat:  ggml_compute_forward_mul_mat_q4_0_f32 (ggml.c:5987-6285) with
// ggml_vec_dot_q4_0's AVX2 arithmetic (ggml.c:1415-1466):
//     acc_k = fma(d_w*d_a, (float) isum_k, acc_k)   block after block, k = 0..7
//     y     = ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7))
// One wave = one row-group (8 rows x 8 chains).  Weights stream HBM -> VGPR (non-temporal dwordx4,
// register ring of DEPTH chunks), activations come from LDS (decode) or L1/L2 (multi-column).
// ------------------------------------------------------------------------------------------------
// acc = fma(p_j, q_j, acc) for the 8 blocks of a chunk, where lane t of every quad holds p_t (PLO) and
// p_{t+4} (PHI): v_fmac_f32_dpp reads its first source through the DPP quad broadcast, so the d_w * d_a
// product is computed twice per lane and chunk instead of eight times.  hipcc keeps v_mov_dpp + v_fmac for
// the equivalent source, hence inline assembly; the one hazard (a VALU write of the DPP source needs two
// wait states before the DPP read) is padded inside the statement.
#define LH_FMAC8_DPP(ACC, PLO, PHI, Q01, Q23, Q45, Q67)                                            \
    asm("s_nop 1\n\t"                                                                              \
        "v_fmac_f32_dpp %0, %1, %3 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %0, %1, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %0, %1, %5 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %0, %1, %6 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %0, %2, %7 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %0, %2, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %0, %2, %9 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %0, %2, %10 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"                 \
        : "+v"(ACC)                                                                                \
        : "v"(PLO), "v"(PHI), "v"((Q01).x), "v"((Q01).y), "v"((Q23).x), "v"((Q23).y),              \
          "v"((Q45).x), "v"((Q45).y), "v"((Q67).x), "v"((Q67).y))

#define LH_STEP(J, WD, AD, DA)                                                                     \
    {                                                                                              \
        const float sc_ = quad_bcast<((J) & 3)>((J) < 4 ? sw.x : sw.y) * (DA);                     \
        /* int -> float without v_cvt: the dot accumulates onto the bit pattern of 1.5 * 2^23 (ulp 1), so  */ \
        /* its result IS the float 12582912 + isum; the exact subtraction pairs up as v_pk_add_f32.        */ \
        /* clamp: VOP3P form, |isum| <= 512 never saturates                                                 */ \
        const int p_ = __builtin_amdgcn_sdot8((int) (WD), (int) (AD), 0x4B400000, true);           \
        acc = fmaf(sc_, __builtin_bit_cast(float, p_) - 12582912.0f, acc);                         \
    }

__device__ __forceinline__ float fold8(float acc) {
    // ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)) in the lane with chain index 0 of every row (the only lane whose
    // result the callers use).  Float add commutes, so a butterfly of pairwise adds is exact; it is done with
    // DPP (row_shl:4 -> lane i reads lane i + 4, then the two quad swaps): three VALU instructions instead of
    // three dependent ds_bpermute round trips at the tail of every launch.
    acc += dpp_f<0x104>(acc);                 // lanes 0..3 of each 8: a_i + a_{i+4}
    acc += dpp_f<DPP_QUAD_XOR2>(acc);         // lanes 0, 1: (a0+a4)+(a2+a6), (a1+a5)+(a3+a7)
    acc += dpp_f<DPP_QUAD_XOR1>(acc);         // lane 0
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
//   gmapF8 : 0, or F/8 for the interleaved w1|w3 matrix (tile group -> logical row-group, see k_repack_q4)
//   EPI_SILU_QA (w1|w3 only, 8 waves per workgroup = 32 gate rows + the same 32 up rows): the
//         workgroup applies silu_lut(gate)*up (ggml.c:1956-1963, .mm:678-680) to its 32 outputs and
//         quantizes them as one Q4_0 activation block (ggml.c:456-523) straight into the QA operand of
//         the following w2 mat-vec: out_A / out_d.  y, if non-null, receives silu*up as fp32.
//   PG  : prologue granules (16 B) kept in registers per thread; the host sizes the workgroup so
//         that PG * blockDim covers the activation row (fp32 modes) or the QA "A" array (PRE_QA)
// Measured on MI355X, 7B decode in situ (tools/ab_libs.sh): SGPR-base weight addressing (saves the 64-bit
// per-lane address arithmetic) made the decode kernels 0.1-0.4 us SLOWER per launch, the zero-padded LDS
// operand tail (clamp-free `base + immediate` reads) helps the ring kernels (wq|wk|wv 9.05 -> 8.58 us) and
// costs the whole-row-in-flight one 0.3 us -- so: no SGPR base here, padding for RING kernels only.
// (k_gemm_skinny keeps both: +2 % there.)
#ifndef LH_GEMV_SADDR
#define LH_GEMV_SADDR 0
#endif
#ifndef LH_GEMV_PAD
#define LH_GEMV_PAD 1
#endif
// 8-byte granule {value, tag}: written with one 8-byte store, read with one 8-byte load that bypasses the L1 (sc1), so a
// reader sees the value together with its tag or not at all.  The spin is bounded; running out raises the fault word.
__device__ __forceinline__ float poll_tagged(const uint64_t *p, uint32_t tag, uint32_t *fault, int nowait /* bit 0: pass at once (measurement), bit 1: no sleep between polls, bit 2: give up after 256 polls (fault-injection test) */) {
    uint64_t v;
    int spins = 0;
    for (;;) {
        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t) (v >> 32) == tag || (nowait & 1)) break;
        if (!(nowait & 2)) __builtin_amdgcn_s_sleep(1);
        if (poll_give_up(spins, (nowait & 4) ? (1 << 8) : (1 << 20), fault)) break;
    }
    return __builtin_bit_cast(float, (uint32_t) v);
}
__device__ __forceinline__ void store_tagged(uint64_t *p, float v, uint32_t tag) {
    __hip_atomic_store(p, (uint64_t) __builtin_bit_cast(uint32_t, v) | ((uint64_t) tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct GemvArgs {
    const uint8_t *wt; int ngroups, nchunks, M, gmapF8;
    const uint32_t *qa_A; const float *qa_d;
    const float *in0, *in1; int K;
    float *y; const float *resid;
    const uint16_t *T_silu;
    uint32_t *out_A; float *out_d;
    const f64x2 *part_in; int npart; f64x2 *part_out;
    uint32_t *sync; int sync_blocks, sync_epoch;      // hand-off words, blocks of the producer role, 1-based epoch
    int lut_math;                                     // bit 0: evaluate SiLU instead of gathering it (verified at load time)
    uint32_t *fault;                                  // tagged operands: sticky fault word (a bounded poll that ran out)
    // residual-stream rows handed between pipeline stages through a device-side mailbox: tagged granules; `sync` -> the epoch word
    const uint64_t *in_t;   int slot_in;              // PREP_NORM_TAG: the fp32 row [K] arrives tagged
    const uint64_t *resid_t; int slot_resid;          // EPI_RESID_TAG: the residual row [M] arrives tagged (null: plain `resid`)
    uint64_t *out_t;        int slot_out;             // EPI_RESID_TAG: y [M] also leaves tagged (null: plain `y` only)
    const int32_t *pos_w;                             // mailbox tags are made from the sequence position, *pos_w + 1, not from the epoch (null: epoch)
    int patience;                                     // mailbox polls wait for ANOTHER process / device: their bounds are shifted left by this
};
template <int PRE, int EPI, int D, bool RING, int PG>
__device__ __forceinline__ void gemv_body(const GemvArgs &ga, const int blk, const int nw, double *smem_d) {
    const uint8_t *__restrict__ wt = ga.wt;
    const int ngroups = ga.ngroups, nchunks = ga.nchunks, M = ga.M, gmapF8 = ga.gmapF8, K = ga.K, npart = ga.npart;
    const uint32_t *__restrict__ qa_A = ga.qa_A; const float *__restrict__ qa_d = ga.qa_d;
    const float *__restrict__ in0 = ga.in0; const float *__restrict__ in1 = ga.in1;
    float *__restrict__ y = ga.y; const float *__restrict__ resid = ga.resid;
    const uint16_t *__restrict__ T_silu = ga.T_silu;
    uint32_t *__restrict__ out_A = ga.out_A; float *__restrict__ out_d = ga.out_d;
    const f64x2 *__restrict__ part_in = ga.part_in; f64x2 *__restrict__ part_out = ga.part_out;
    // (EPI_STORE_TAG: the tag of this launch's output granules, read up front -- not a dependent load at the tail)
    constexpr bool TAGGED = (EPI == EPI_STORE_TAG || PRE == PREP_NORM_TAG || EPI == EPI_RESID_TAG);
    const uint32_t epoch_ = TAGGED ? __builtin_nontemporal_load(ga.sync) : 0u;
    const uint32_t store_tag = make_tag(epoch_, ga.sync_epoch + 1);        // EPI_STORE_TAG output of layer ga.sync_epoch
    // (mailbox rows between pipeline stages: the tag is the sequence position both sides know, st[0] + 1 -- the stages' epochs differ)
    const uint32_t mb_epoch_ = ((PRE == PREP_NORM_TAG || EPI == EPI_RESID_TAG) && ga.pos_w) ? (uint32_t) __builtin_nontemporal_load(ga.pos_w) + 1u : epoch_;
    const uint32_t tag_in = make_tag(mb_epoch_, ga.slot_in), tag_resid = make_tag(mb_epoch_, ga.slot_resid), tag_out = make_tag(mb_epoch_, ga.slot_out);
    // RING kernels: LDS holds D chunks more than the row has.  The ring's tail and its one-chunk-ahead
    // operand fetch run past the end (against the zero tile), and with zeroed padding those reads need no
    // index clamp -- their addresses are `loop base + immediate` instead of three VALU per chunk.
    uint32_t *ldsA = (uint32_t *) smem_d;
    constexpr int PADC = (LH_GEMV_PAD && RING) ? D : 0;
    float *ldsD = (float *) (ldsA + (nchunks + PADC) * 64);
    // (LH_GEMV_SADDR, off: with a provably uniform wave index the row-group base lives in SGPRs and every
    //  weight load is `global_load ... v_off, s[base]` with a constant per-lane offset -- measured slower here)
    const int tid = threadIdx.x, lane = tid & 63, wave = LH_GEMV_SADDR ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
    constexpr bool active = true;
    const int g = blk * nw + wave;
    const bool valid = active && g < ngroups;
    const uint8_t *wbase = wt + (size_t) (valid ? g : 0) * (nchunks + 1) * TILE_BYTES;
    const uint32_t voff_w = (uint32_t) lane * 16u, voff_s = 1024u + (uint32_t) ((lane >> 3) * 8 + (lane & 3) * 2) * 4u;
    // (an empty asm per loop trip keeps the 32 -> 64-bit extension of these lane offsets inside the loop
    //  block: hoisted out of it they become 64-bit VGPR pairs and the SGPR-base addressing no longer matches)
    uint32_t vw_ = voff_w, vs_ = voff_s;
#define LH_OPAQUE_OFFSETS() { vw_ = voff_w; vs_ = voff_s; asm volatile("" : "+v"(vw_), "+v"(vs_)); }
#if LH_PHASE_PROBE == 3
    unsigned long long probe_t[5] = { 0, 0, 0, 0, 0 };
    const unsigned long long probe_wall = wall_clock64();
#elif LH_PHASE_PROBE
    unsigned long long *probe_e = nullptr;
    if (g_phase_probe && blk == (int) gridDim.x / 2 && threadIdx.x == 0) {
        unsigned long long *pb = g_phase_probe;
        const unsigned long long slot = atomicAdd(pb, 1ull);
        if (slot < pb[1]) { probe_e = pb + 8 * (1 + slot); probe_e[5] = ngroups; probe_e[6] = nchunks; probe_e[7] = PRE * 16 + EPI; }
    }
#endif
    LH_STAMP(0);

    u32x4 wq[D];
    f32x2 ws[D];
#define LH_LOADW(SLOT, CH)                                                                                   \
    {                                                                                                        \
        const int ch_ = min((CH), nchunks);   /* tile `nchunks` of every row-group is the zero tile */      \
        const uint8_t *tp_ = wbase + (size_t) ch_ * TILE_BYTES;                                              \
        if (LH_GEMV_SADDR) {                                                                                 \
            wq[SLOT] = __builtin_nontemporal_load((const u32x4 *) (tp_ + (size_t) vw_));                     \
            ws[SLOT] = __builtin_nontemporal_load((const f32x2 *) (tp_ + (size_t) vs_));                     \
        } else {                                                                                             \
            wq[SLOT] = __builtin_nontemporal_load((const u32x4 *) (tp_ + lane * 16));                        \
            ws[SLOT] = __builtin_nontemporal_load((const f32x2 *) (tp_ + 1024 + ((lane >> 3) * 8 + (lane & 3) * 2) * 4)); \
        }                                                                                                    \
    }
    // ---- phase 1: the prologue's own (small, L2-resident) loads go out FIRST.  vmcnt retires in
    // order, so anything issued behind the weight prefetch would have to wait for all of it.
    // fp32 prologues own the row in HALF-BLOCK granules (16 contiguous elements = 4 float4): granule
    // h belongs to thread h % blockDim, so the two halves of a Q4_0 block sit in lanes t and t^1 and the
    // whole norm -> quantize pipeline stays in registers (no LDS staging of y, no one-thread-per-block
    // serial quantizer: the prologue is VALU work repeated by every workgroup, so its instruction
    // count matters as much as the mat-vec's).
    constexpr bool NORMTAG = (PRE == PREP_NORM_TAG);      // PREP_NORM on a row that arrives as tagged granules (gathered after phase 2)
    constexpr bool NORMLIKE = (PRE == PREP_NORM || PRE == PREP_NORMP || NORMTAG);
    constexpr bool REGPRE = (PRE == PRE_QA || NORMLIKE || PRE == PREP_PLAIN);
    constexpr int MAXH = (NORMLIKE || PRE == PREP_PLAIN) ? PG : 1;   // half-block granules per thread
    constexpr int MAXQA = (PRE == PRE_QA) ? PG : 1, MAXQD = (PG + 7) / 8;    // QA granules per thread (da is 1/8 of A)
    f32x4 xa[MAXH][4], xb[MAXH][4];
    u32x4 qg[MAXQA], qh[MAXQD];
    const int nt = nw * 64;
    const int nh = K >> 4;                                   // half-blocks in the row
    // (trip counts are wave-uniform; a skipped load only makes the compiler's vmcnt for these
    //  prologue loads stricter -- they are all older than the weight loads, which stay in flight)
    // PREP_NORMP: the producer of the row (an EPI_RESID mat-vec, or k_embed_part) left per-workgroup
    // {sum x, sum x^2} in double; wave 0 folds them (same data, same order in every workgroup -> identical
    // statistics everywhere) and the row itself is never reduced.  PNP pairs per lane cover up to 64 * PNP
    // producer workgroups.
    constexpr int PNP = (PRE == PREP_NORMP) ? 8 : 1;
    f64x2 pp[PNP];
    const int npl = (npart + 63) >> 6;
    float resid_v = 0.0f;
    uint64_t resid_g = 0;
    auto phase1 = [&]() {
    if ((NORMLIKE || PRE == PREP_PLAIN) && active) {
        const int ng = (nh + nt - 1) / nt;
#pragma unroll
        for (int u = 0; u < MAXH; u++) {
            // (waves whose granules all lie past the row issue nothing: w1|w3 runs 8 waves on a 4-wave row)
            if (u < ng && (tid & ~63) + u * nt < nh) {
                const int hi = min(tid + u * nt, nh - 1);
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    if (!NORMTAG) xa[u][v] = ((const f32x4 *) in0)[hi * 4 + v];
                    if (NORMLIKE) xb[u][v] = ((const f32x4 *) in1)[hi * 4 + v];
                }
            } else {
#pragma unroll
                for (int v = 0; v < 4; v++) { xa[u][v] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f }; xb[u][v] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f }; }
            }
        }
        if (PRE == PREP_NORMP && wave == 0) {          // one wave folds the pairs for the workgroup
#pragma unroll
            for (int u = 0; u < PNP; u++) if (u < npl) pp[u] = part_in[min(lane + u * 64, npart - 1)];
        }
    }
    // the residual operand of the epilogue is fetched here too, not at the end of the kernel where it
    // would add a memory round trip to every wave's critical path
    if ((EPI == EPI_RESID || (EPI == EPI_RESID_TAG && !ga.resid_t)) && active) {
        int lg0 = g;
        if (gmapF8) { const int b8 = g >> 3, w8 = g & 7; lg0 = w8 < 4 ? b8 * 4 + w8 : gmapF8 + b8 * 4 + (w8 - 4); }
        resid_v = resid[min(lg0 * 8 + (lane >> 3), M - 1)];
    }
    // (EPI_RESID_TAG with a mailbox residual: the granule is requested here as well -- the launch before this one gathered the same
    //  row, so it is there; the epilogue re-polls only if the tag says otherwise)
    if (EPI == EPI_RESID_TAG && ga.resid_t && active && (lane & 7) == 0) {
        int lg0 = g;
        if (gmapF8) { const int b8 = g >> 3, w8 = g & 7; lg0 = w8 < 4 ? b8 * 4 + w8 : gmapF8 + b8 * 4 + (w8 - 4); }
        resid_g = load_granule_sys(ga.resid_t + min(lg0 * 8 + (lane >> 3), M - 1));
    }
    if (PRE == PRE_QA && active) {
        const int nqa = (nchunks * 16 + nt - 1) / nt, nqd = (nchunks * 2 + nt - 1) / nt;
#pragma unroll
        for (int u = 0; u < MAXQA; u++) if (u < nqa) qg[u] = ((const u32x4 *) qa_A)[min(tid + u * nt, nchunks * 16 - 1)];
#pragma unroll
        for (int u = 0; u < MAXQD; u++) if (u < nqd) qh[u] = ((const u32x4 *) qa_d)[min(tid + u * nt, nchunks * 2 - 1)];
    }
    };
    // ---- phase 2: put the first D weight chunks in flight (they do not depend on the activations).
    // The scheduling barriers pin the issue order phase 1 -> phase 2 -> phase 3.
    phase1();
    __builtin_amdgcn_sched_barrier(0);
    if (active) {
#pragma unroll
        for (int i = 0; i < D; i++) LH_LOADW(i, i)
    }
    __builtin_amdgcn_sched_barrier(0);

    if (NORMTAG) {
        // The row comes from the launch that runs BESIDE this one (the other branch of the overlapped decode schedule) as tagged
        // granules.  This workgroup's first D weight chunks are in flight (phase 2).  Wave 0 watches LH_WATCH sample granules spread
        // over the row, sleeping between looks -- the producer's workgroups finish together, and pollers compete with its
        // weight stream -- then every thread runs the tag-checked copy of its own half-blocks, which passes on its first or second
        // round.  Correctness rests on the copy alone; the watch only keeps the polling traffic small.
        const uint64_t *__restrict__ xt = ga.in_t;
        const int give_up = (ga.lut_math & 0x1000) ? (1 << 8) : ((1 << 20) << ga.patience);      // (0x1000: fault-injection test)
        if (wave == 0) {
            int spins = 0;
            for (;;) {
                bool ok = true;
                if (lane < LH_WATCH) ok = (uint32_t) (load_granule_sys(xt + ((2 * lane + 1) * K / (2 * LH_WATCH))) >> 32) == tag_in;
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (give_up >> 2) || ((spins & 255) == 0 && __hip_atomic_load(ga.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)) break;   // (the copy below raises the fault word if the row never comes)
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < MAXH; u++) {
            const int hi = tid + u * nt;
            if (active && hi < nh) {
                uint64_t gv[16];
                int spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        gv[i] = load_granule_sys(xt + hi * 16 + i);
                        ok = ok && (uint32_t) (gv[i] >> 32) == tag_in;
                    }
                    if (ok) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (poll_give_up(spins, give_up, ga.fault)) break;
                }
#pragma unroll
                for (int v = 0; v < 4; v++)
                    xa[u][v] = f32x4{ __builtin_bit_cast(float, (uint32_t) gv[4 * v]), __builtin_bit_cast(float, (uint32_t) gv[4 * v + 1]),
                                      __builtin_bit_cast(float, (uint32_t) gv[4 * v + 2]), __builtin_bit_cast(float, (uint32_t) gv[4 * v + 3]) };
            } else {
#pragma unroll
                for (int v = 0; v < 4; v++) xa[u][v] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
            }
        }
    }

    // ---- phase 3: prologue arithmetic while the weights stream in
    LH_STAMP(1);
    double *red = (double *) (ldsD + (nchunks + PADC) * 8);
    for (int i = active ? tid : PADC * 72; i < PADC * 72; i += nt) {              // zero the padding chunks (A: 64 dwords, d: 8 floats each)
        if (i < PADC * 64) ldsA[nchunks * 64 + i] = 0u;
        else ldsD[nchunks * 8 + (i - PADC * 64)] = 0.0f;
    }
    if (PRE == PRE_QA) {
#pragma unroll
        for (int u = 0; u < MAXQA; u++) { const int gi = tid + u * nt; if (active && gi < nchunks * 16) ((u32x4 *) ldsA)[gi] = qg[u]; }
#pragma unroll
        for (int u = 0; u < MAXQD; u++) { const int gi = tid + u * nt; if (active && gi < nchunks * 2) ((u32x4 *) ldsD)[gi] = qh[u]; }
        __syncthreads();
    } else if (REGPRE) {
        if (NORMLIKE) {
            // ggml_norm + ggml_mul (ggml.c:5327-5385, :4555) on register-resident x
            // The statistics: S1 = sum x, S2 = sum x^2 (double; x^2 is exact there), either handed over by the
            // producer (PREP_NORMP) or reduced here with ONE barrier.  Then
            //     mean = S1 / K,   sum (x - mean)^2 = S2 - mean * S1.
            // Both forms of the second moment carry a few 2^-53 of rounding (the reference's own sum rounds every
            // (x - mean)^2 and every addition) and the result is narrowed to fp32 afterwards: the same class of
            // agreement as the re-ordered double sums this prologue always had (DESIGN.md "norm").  When the mean
            // dominates (K * mean^2 above a quarter of sum x^2) the subtraction would cancel, and the reference's
            // two-pass form runs instead (also with npart < 0: measurement switch).
            double S1 = 0.0, S2 = 0.0;
            if (PRE == PREP_NORMP) {
                if (wave == 0) {
#pragma unroll
                    for (int u = 0; u < PNP; u++)
                        if (u < npl && lane + u * 64 < npart) { S1 += pp[u].x; S2 += pp[u].y; }
                    S1 = wave_sum_d(S1);
                    S2 = wave_sum_d(S2);
                    if (lane == 0) { red[0] = S1; red[1] = S2; }
                }
                __syncthreads();
                S1 = red[0]; S2 = red[1];
            } else {
#pragma unroll
                for (int u = 0; u < MAXH; u++)
                    if (active && tid + u * nt < nh) {
#pragma unroll
                        for (int v = 0; v < 4; v++) {
                            const double x0 = (double) xa[u][v].x, x1 = (double) xa[u][v].y, x2 = (double) xa[u][v].z, x3 = (double) xa[u][v].w;
                            S1 += x0; S1 += x1; S1 += x2; S1 += x3;
                            S2 = __builtin_fma(x0, x0, S2); S2 = __builtin_fma(x1, x1, S2); S2 = __builtin_fma(x2, x2, S2); S2 = __builtin_fma(x3, x3, S2);
                        }
                    }
                block_sum_d2(S1, S2, red, 0);
            }
            const double mean = S1 / (double) K;
            double sum2 = __builtin_fma(-mean, S1, S2);
            const bool fast = npart >= 0 && mean * S1 <= 0.25 * S2;        // (false for NaNs too); identical in every wave of the launch
            LH_STAMP2(2);
            double s2 = 0.0;
#pragma unroll
            for (int u = 0; u < MAXH; u++)
                if (active && tid + u * nt < nh) {
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const double v0 = (double) xa[u][v].x - mean, v1 = (double) xa[u][v].y - mean;
                        const double v2 = (double) xa[u][v].z - mean, v3 = (double) xa[u][v].w - mean;
                        xa[u][v].x = (float) v0; xa[u][v].y = (float) v1; xa[u][v].z = (float) v2; xa[u][v].w = (float) v3;
                        if (!fast) { s2 += v0 * v0; s2 += v1 * v1; s2 += v2 * v2; s2 += v3 * v3; }
                    }
                }
            if (!fast) sum2 = block_sum_d(s2, red, 1);
            const float scale = (float) (1.0 / sqrt(sum2 / (double) K + (double) 1e-5f));
            LH_STAMP2(3);
#pragma unroll
            for (int u = 0; u < MAXH; u++)
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    xa[u][v].x = xb[u][v].x * (xa[u][v].x * scale); xa[u][v].y = xb[u][v].y * (xa[u][v].y * scale);
                    xa[u][v].z = xb[u][v].z * (xa[u][v].z * scale); xa[u][v].w = xb[u][v].w * (xa[u][v].w * scale);
                }
        }
        // quantize_row_q4_0, AVX2 branch (ggml.c:456-523), two lanes per block
        const int nbp = nchunks * 8;
#pragma unroll
        for (int u = 0; u < MAXH; u++) {
            const int hi = tid + u * nt;                       // half-block index; block = hi >> 1, half = hi & 1
            const bool live = active && hi < nh;
            float amax = 0.0f;
            if (live) {
#pragma unroll
                for (int v = 0; v < 4; v++)
                    amax = fmaxf(fmaxf(fmaxf(amax, fabsf(xa[u][v].x)), fabsf(xa[u][v].y)), fmaxf(fabsf(xa[u][v].z), fabsf(xa[u][v].w)));
            }
            amax = fmaxf(amax, dpp_f<DPP_QUAD_XOR1>(amax));    // partner half (lane ^ 1); both dead or both live
            const float dd = amax / 7.0f;
            const float id = (amax != 0.0f) ? 7.0f / amax : 0.0f;
            // this half's 8 element pairs: pair p = elements (2p, 2p+1) of the half -> one 16-bit field
            uint32_t pr[8];
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const uint32_t n0 = (uint32_t) ((int) __builtin_rintf(xa[u][v].x * id)) & 0xF, n1 = (uint32_t) ((int) __builtin_rintf(xa[u][v].y * id)) & 0xF;
                const uint32_t n2 = (uint32_t) ((int) __builtin_rintf(xa[u][v].z * id)) & 0xF, n3 = (uint32_t) ((int) __builtin_rintf(xa[u][v].w * id)) & 0xF;
                pr[2 * v] = n0 | (n1 << 8);
                pr[2 * v + 1] = n2 | (n3 << 8);
            }
            // chain k of the block = pair k of half 0 (low 16 bits) | pair k of half 1 (high 16 bits)
            const int half = hi & 1, b = hi >> 1, c = b >> 3, j = b & 7;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t other = (uint32_t) __builtin_amdgcn_mov_dpp((int) pr[k], DPP_QUAD_XOR1, 0xF, 0xF, true);
                const uint32_t dw = (half ? (other | (pr[k] << 16)) : (pr[k] | (other << 16))) << (4 * (j & 1));
                // half 0 stores chains 0..3, half 1 chains 4..7
                if (live && (k >> 2) == half) ldsA[(c * 8 + k) * 8 + j] = dw;
            }
            if (live && half == 0) ldsD[b] = dd;
        }
        // zero the padded blocks (K not a multiple of 256)
        for (int b = active ? K / 32 + tid : nbp; b < nbp; b += nt) {
            const int c = b >> 3, j = b & 7;
#pragma unroll
            for (int k = 0; k < 8; k++) ldsA[(c * 8 + k) * 8 + j] = 0;
            ldsD[b] = 0.0f;
        }
        __syncthreads();
    } else {
        float *ybuf = (float *) (red + 32);
        make_y<PRE>(ybuf, red, in0, in1, K, T_silu);
        quantize_y(ybuf, K, nchunks * 256, ldsA, ldsD, nullptr);
        __syncthreads();
    }

    const int k = lane & 7;
    float acc = 0.0f;
    // LDS operands (activation nibbles + scales of one chunk) are fetched one chunk ahead into the
    // other half of a two-entry register buffer, so their ~100-cycle latency is off the FMA chain
    static_assert(D % 2 == 0, "the two-entry LDS operand buffer alternates by slot parity: ring depth must be even");
    u32x4 la0[2], la1[2];
    float ldl[2], ldh[2];
    const int tq = lane & 3;                 // this lane's weight scales are those of blocks tq and tq + 4
#define LH_LDSLOAD(BUF, CH)                                                                        \
    {                                                                                              \
        const int cl_ = PADC ? (CH) : min((CH), nchunks - 1);                                      \
        const u32x4 *pa = (const u32x4 *) (ldsA + (cl_ * 8 + k) * 8);                              \
        la0[BUF] = pa[0]; la1[BUF] = pa[1];                                                        \
        ldl[BUF] = ldsD[cl_ * 8 + tq]; ldh[BUF] = ldsD[cl_ * 8 + 4 + tq];                          \
    }
#define LH_CONSUME(SLOT, CH)                                                                       \
    {                                                                                              \
        const u32x4 w = wq[SLOT];                                                                  \
        const f32x2 sw = ws[SLOT];                                                                 \
        const u32x4 a0 = la0[(SLOT) & 1], a1 = la1[(SLOT) & 1];                                    \
        const float plo_ = sw.x * ldl[(SLOT) & 1], phi_ = sw.y * ldh[(SLOT) & 1];                  \
        LH_LDSLOAD(((SLOT) + 1) & 1, (CH) + 1)                                                     \
        /* 8 blocks: integer dots first, accumulated onto the bit pattern of 1.5 * 2^23 (ulp 1) so each  */ \
        /* result IS the float 12582912 + isum (|isum| <= 512; clamp selects the VOP3P form and never    */ \
        /* saturates); the exact subtraction is done two at a time (v_pk_add_f32) instead of 8 v_cvt;    */ \
        /* then the block-ordered FMA chain with the DPP-broadcast scales.                                */ \
        const int i0_ = __builtin_amdgcn_sdot8((int) w.x, (int) a0.x, 0x4B400000, true);           \
        const int i1_ = __builtin_amdgcn_sdot8((int) w.x, (int) a0.y, 0x4B400000, true);           \
        const int i2_ = __builtin_amdgcn_sdot8((int) w.y, (int) a0.z, 0x4B400000, true);           \
        const int i3_ = __builtin_amdgcn_sdot8((int) w.y, (int) a0.w, 0x4B400000, true);           \
        const int i4_ = __builtin_amdgcn_sdot8((int) w.z, (int) a1.x, 0x4B400000, true);           \
        const int i5_ = __builtin_amdgcn_sdot8((int) w.z, (int) a1.y, 0x4B400000, true);           \
        const int i6_ = __builtin_amdgcn_sdot8((int) w.w, (int) a1.z, 0x4B400000, true);           \
        const int i7_ = __builtin_amdgcn_sdot8((int) w.w, (int) a1.w, 0x4B400000, true);           \
        const f32x2 mg_ = { 12582912.0f, 12582912.0f };                                            \
        const f32x2 q01_ = f32x2{ __builtin_bit_cast(float, i0_), __builtin_bit_cast(float, i1_) } - mg_; \
        const f32x2 q23_ = f32x2{ __builtin_bit_cast(float, i2_), __builtin_bit_cast(float, i3_) } - mg_; \
        const f32x2 q45_ = f32x2{ __builtin_bit_cast(float, i4_), __builtin_bit_cast(float, i5_) } - mg_; \
        const f32x2 q67_ = f32x2{ __builtin_bit_cast(float, i6_), __builtin_bit_cast(float, i7_) } - mg_; \
        LH_FMAC8_DPP(acc, plo_, phi_, q01_, q23_, q45_, q67_);                                     \
    }
    LH_STAMP(2);
    LH_STAMP2(4);
    if (active) {
    LH_LDSLOAD(0, 0)
    int c0 = 0;
    if (RING) {
        do {
            if (LH_GEMV_SADDR) LH_OPAQUE_OFFSETS()
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
    }
#undef LH_LDSLOAD
#undef LH_CONSUME
#undef LH_LOADW
#undef LH_OPAQUE_OFFSETS

    LH_STAMP(3);
    acc = fold8(acc);
    int lg = g;
    if (gmapF8) { const int b8 = g >> 3, w8 = g & 7; lg = w8 < 4 ? b8 * 4 + w8 : gmapF8 + b8 * 4 + (w8 - 4); }
    const int m = lg * 8 + (lane >> 3);
    if (EPI == EPI_SILU_QA) {
        // 8 waves: waves 0-3 hold gate rows b*32 .. b*32+31, waves 4-7 the matching up rows (b = blockIdx.x)
        float *gu = (float *) red;                      // prologue scratch is free again
        if (!(ga.lut_math & 8)) __syncthreads();
        if (k == 0) gu[wave * 8 + (lane >> 3)] = acc;
        __syncthreads();
        if (wave == 0 && !(ga.lut_math & 4)) {
            const int i = lane & 31;
            const uint16_t gh = f2h_bits(gu[i]);
            const float act = h2f_bits((ga.lut_math & 1) ? silu_math_bits(gh) : T_silu[gh]) * gu[32 + i];
            float amax = fabsf(act);
            amax = max_lanes_0_31(amax);
            const float dd = amax / 7.0f;
            const float id = (amax != 0.0f) ? 7.0f / amax : 0.0f;
            const uint32_t nib = ((uint32_t) ((int) __builtin_rintf(act * id) + 8) - 8) & 0xF;     // signed nibble of (q - 8)
            const int kk = lane & 7;
            const uint32_t e0 = __shfl(nib, 2 * kk), e1 = __shfl(nib, 2 * kk + 1);
            const uint32_t e2 = __shfl(nib, 16 + 2 * kk), e3 = __shfl(nib, 17 + 2 * kk);
            const int b = blk, c = b >> 3, j = b & 7;
            const uint32_t dw = (e0 | (e1 << 8) | (e2 << 16) | (e3 << 24)) << (4 * (j & 1));
            {
                if (lane < 8) out_A[(c * 8 + kk) * 8 + j] = dw;
                if (lane == 0) out_d[b] = dd;
            }
            if (y && lane < 32) y[b * 32 + i] = act;
        }
    } else {
        const bool live = valid && k == 0 && m < M;
        if (EPI == EPI_RESID) acc = acc + resid_v;
        if (EPI == EPI_RESID_TAG) {
            float rv = resid_v;
            if (ga.resid_t) {
                rv = __builtin_bit_cast(float, (uint32_t) resid_g);
                if (live && (uint32_t) (resid_g >> 32) != tag_resid) {
                    int spins = 0;
                    uint64_t gq;
                    for (;;) {
                        gq = load_granule_sys(ga.resid_t + m);
                        if ((uint32_t) (gq >> 32) == tag_resid) break;
                        __builtin_amdgcn_s_sleep(4);
                        if (poll_give_up(spins, (ga.lut_math & 0x1000) ? (1 << 8) : ((1 << 20) << ga.patience), ga.fault)) break;
                    }
                    rv = __builtin_bit_cast(float, (uint32_t) gq);
                }
            }
            acc = acc + rv;
        }
        if (EPI == EPI_STORE_TAG) {                     // ga.sync -> the epoch word, ga.sync_epoch = layer (k_qkv_attn)
            if (live) store_tagged((uint64_t *) y + m, acc, store_tag ^ ((ga.lut_math & 0x1000) ? 1u : 0u));      // (0x1000: fault-injection test -- a tag nobody waits for)
        } else if (EPI == EPI_RESID_TAG) {              // the row leaves for the next pipeline stage's mailbox (and / or plain)
            if (live) {
                if (ga.out_t) store_tagged_sys(ga.out_t + m, __builtin_bit_cast(uint32_t, acc), tag_out ^ ((ga.lut_math & 0x2000) ? 1u : 0u));      // (0x2000: fault-injection test)
                if (y) y[m] = acc;
            }
        } else
        if (live) y[m] = acc;
        if ((EPI == EPI_RESID || EPI == EPI_RESID_TAG) && part_out) {
            // this workgroup's share of the next norm's statistics (consumed by a PREP_NORMP prologue): sum y and
            // sum y^2 over its rows, in double (y^2 is exact there), folded in a fixed order
            const double yd = live ? (double) acc : 0.0;
            const double s1 = wave_sum_d(yd), s2 = wave_sum_d(yd * yd);
            if (nw == 1) {
                if (lane == 0) part_out[blk] = f64x2{ s1, s2 };
            } else {
                if (lane == 0) { red[2 * wave] = s1; red[2 * wave + 1] = s2; }
                __syncthreads();
                if (tid == 0) {
                    double t1 = red[0], t2 = red[1];
                    for (int w2_ = 1; w2_ < nw; w2_++) { t1 += red[2 * w2_]; t2 += red[2 * w2_ + 1]; }
                    part_out[blk] = f64x2{ t1, t2 };
                }
            }
        }
    }
    LH_STAMP(4);
#if LH_PHASE_PROBE == 3
    if (g_phase_probe && threadIdx.x == 0) {
        unsigned long long *pb = g_phase_probe;
        const unsigned long long slot = atomicAdd(pb, 1ull);
        if (slot < pb[1]) {
            unsigned long long *e = pb + 8 * (1 + slot);
            for (int i = 0; i < 5; i++) e[i] = probe_t[i];
            e[5] = ((unsigned long long) (PRE * 16 + EPI) << 48) | ((unsigned long long) nchunks << 32) | (unsigned) blk;
            e[6] = wall_clock64();          // s_memtime is per-XCD: launches are lined up on the 100 MHz wall clock
            e[7] = probe_wall;
        }
    }
#endif
}


template <int PRE, int EPI, int D, bool RING, int PG>
__global__ void __launch_bounds__(EPI == EPI_SILU_QA ? 512 : 256, EPI == EPI_SILU_QA ? 4 : 1)
k_gemv(const GemvArgs ga) {
    extern __shared__ double smem_d[];
    gemv_body<PRE, EPI, D, RING, PG>(ga, blockIdx.x, blockDim.x >> 6, smem_d);
}

// Prompt path on the decode tiles (runs when the handle has no row-lane copy): NC activation rows
// share every weight tile; their operands for one chunk (NC x 288 B) are staged once per workgroup in
// LDS (double-buffered, one barrier per chunk) and shared by its 4 waves.  (Round 1's first variant
// read them through the texture path instead: 64 vector loads per chunk per wave, 1.8x slower.)
//   ncols <= NC: columns past ncols are clamped duplicates whose results are not stored.
template <int NC, int EPI>
__global__ void __launch_bounds__(256)
k_gemm_lds(const uint8_t *__restrict__ wt, int ngroups, int nchunks, int M, int gmapF8,
           const uint32_t *__restrict__ qa_A, const float *__restrict__ qa_d, int ncols,
           float *__restrict__ y, long y_stride, const float *__restrict__ resid, long resid_stride) {
    __shared__ u32x4 sA[2][NC * 16];          // [buf][col][chain k][2 x u32x4]  = [col][64 dwords]
    __shared__ f32x4 sD[2][NC * 2];           // [buf][col][8 floats]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6, nt = blockDim.x;
    const int g = min((int) (blockIdx.x * nw + wave), ngroups - 1);
    const bool valid = (int) (blockIdx.x * nw + wave) < ngroups;
    const uint8_t *wbase = wt + (size_t) g * (nchunks + 1) * TILE_BYTES;
    const int k = lane & 7;
    const long strideA = (long) nchunks * 16, strideD = (long) nchunks * 2;      // in 16-byte granules
    const int soff = 1024 + ((lane >> 3) * 8 + (lane & 3) * 2) * 4;
    constexpr int GA = (NC * 16 + 255) / 256, GD = 1;                             // granules per thread per chunk
    float accs[NC];
#pragma unroll
    for (int n = 0; n < NC; n++) accs[n] = 0.0f;

    constexpr int RD = 4;                     // weight ring depth
    u32x4 wq[RD];
    f32x2 ws[RD];
#pragma unroll
    for (int i = 0; i < RD; i++) {
        const uint8_t *tp = wbase + (size_t) min(i, nchunks) * TILE_BYTES;
        wq[i] = __builtin_nontemporal_load((const u32x4 *) (tp + lane * 16));
        ws[i] = __builtin_nontemporal_load((const f32x2 *) (tp + soff));
    }
    // QA granule (col n, piece p) of chunk c lives at qa_A4[n * strideA + c * 16 + p]
    u32x4 ga[GA];
    f32x4 gd[GD];
    auto fetch = [&](int c) {
#pragma unroll
        for (int u = 0; u < GA; u++) {
            const int gi = min(tid + u * nt, NC * 16 - 1), n = min(gi >> 4, ncols - 1), pce = gi & 15;
            ga[u] = ((const u32x4 *) qa_A)[n * strideA + (long) c * 16 + pce];
        }
        const int gj = min(tid, NC * 2 - 1), n = min(gj >> 1, ncols - 1);
        gd[0] = ((const f32x4 *) qa_d)[n * strideD + (long) c * 2 + (gj & 1)];
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < GA; u++) { const int gi = tid + u * nt; if (gi < NC * 16) sA[buf][gi] = ga[u]; }
        if (tid < NC * 2) sD[buf][tid] = gd[0];
    };
    fetch(0);
    stash(0);
    for (int c0 = 0; c0 < nchunks; c0 += RD) {
#pragma unroll
        for (int i = 0; i < RD; i++) {
            const int c = c0 + i;                        // chunks past the row end read the zero tile: no effect
            __syncthreads();
            const int buf = c & 1;
            if (c + 1 < nchunks) fetch(c + 1);
            const u32x4 w = wq[i];
            const f32x2 sw = ws[i];
            {
                const uint8_t *tp = wbase + (size_t) min(c + RD, nchunks) * TILE_BYTES;
                wq[i] = __builtin_nontemporal_load((const u32x4 *) (tp + lane * 16));
                ws[i] = __builtin_nontemporal_load((const f32x2 *) (tp + soff));
            }
            if (c < nchunks) {
                const uint32_t w0 = w.x, w1 = w.y, w2 = w.z, w3 = w.w;
                const float s0 = quad_bcast<0>(sw.x), s1 = quad_bcast<1>(sw.x), s2 = quad_bcast<2>(sw.x), s3 = quad_bcast<3>(sw.x);
                const float s4 = quad_bcast<0>(sw.y), s5 = quad_bcast<1>(sw.y), s6 = quad_bcast<2>(sw.y), s7 = quad_bcast<3>(sw.y);
#pragma unroll
                for (int n = 0; n < NC; n++) {
                    const u32x4 a0 = sA[buf][n * 16 + k * 2], a1 = sA[buf][n * 16 + k * 2 + 1];
                    const f32x4 d0 = sD[buf][n * 2], d1 = sD[buf][n * 2 + 1];
                    float acc = accs[n];
#define LH_STEPN(SW, WD, AD, DA) { const float sc_ = (SW) * (DA); const int p_ = __builtin_amdgcn_sdot8((int) (WD), (int) (AD), 0, true); acc = fmaf(sc_, (float) p_, acc); }
                    LH_STEPN(s0, w0, a0.x, d0.x) LH_STEPN(s1, w0, a0.y, d0.y)
                    LH_STEPN(s2, w1, a0.z, d0.z) LH_STEPN(s3, w1, a0.w, d0.w)
                    LH_STEPN(s4, w2, a1.x, d1.x) LH_STEPN(s5, w2, a1.y, d1.y)
                    LH_STEPN(s6, w3, a1.z, d1.z) LH_STEPN(s7, w3, a1.w, d1.w)
#undef LH_STEPN
                    accs[n] = acc;
                }
            }
            if (c + 1 < nchunks) stash((c + 1) & 1);
        }
    }
    int lg = g;
    if (gmapF8) { const int blk = g >> 3, w8 = g & 7; lg = w8 < 4 ? blk * 4 + w8 : gmapF8 + blk * 4 + (w8 - 4); }
    const int m = lg * 8 + (lane >> 3);
#pragma unroll
    for (int n = 0; n < NC; n++) {
        float acc = fold8(accs[n]);
        if (valid && k == 0 && m < M && n < ncols) {
            if (EPI == EPI_RESID) acc = acc + resid[(size_t) n * resid_stride + m];
            y[(size_t) n * y_stride + m] = acc;
        }
    }
}

// Short prompt chunks (2 <= N <= ~32 columns; the reference feeds prompts n_batch = 8 tokens at a time): the
// decode kernel's work distribution -- lane = (row, chain), weights streamed once per wave through a
// register ring -- with NC activation columns per wave.  The row-per-lane kernel below needs 64 rows per
// wave, which leaves a 4096-row matrix with 64 waves per column and makes every column re-read the
// weights from L2; here a 4096-row matrix is 512 / RG waves per column GROUP and the weights are read once
// per group.  The QA operands of the workgroup's NC columns are staged whole in LDS before the main loop
// (no barriers inside it); the weight ring is put in flight before the staging so the two latencies
// overlap.  Same arithmetic and order as k_gemv.  What bounds this kernel is VALU issue and LDS read
// bandwidth together (a 16-byte broadcast read still delivers 1 KiB per wave), so:
//   * a wave owns RG row-groups (lane = row r of each, chain k): every activation read serves RG rows;
//   * the d_w * d_a products are computed once per quad lane (lane t of a quad holds the weight scales of
//     blocks t and t + 4 -- the tile's scale layout -- and reads the two matching activation scales with
//     one 4-byte LDS read each), and the FMA takes them through the DPP quad broadcast of v_fmac_f32_dpp:
//     8 dots + 4 packed subtractions + 2 products + 8 FMAs = 22 VALU per (lane, chunk, column), not 28.
//     The DPP form is written as inline assembly (the compiler keeps v_mov_dpp + v_fmac); its one hazard
//     -- a VALU write of the DPP source needs two wait states before the read -- is padded inside.
//   grid: XCD-aware, blockIdx -> (row-block of 4 * RG row-groups, column group), column groups of a row-block on one XCD
//   dynamic LDS: [NC][(nchunks + 4) * 64] dwords A, then [NC][(nchunks + 4) * 8] floats d (4 zeroed padding chunks per column)
// two independent chains interleaved (a dependent v_fmac issues ~1.7x slower than an independent one)
#define LH_FMAC8_DPP2(ACC0, PLO0, PHI0, A01, A23, A45, A67, ACC1, PLO1, PHI1, B01, B23, B45, B67)  \
    asm("s_nop 1\n\t"                                                                              \
        "v_fmac_f32_dpp %0, %2, %4 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %1, %12, %14 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"           \
        "v_fmac_f32_dpp %0, %2, %5 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %1, %12, %15 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"           \
        "v_fmac_f32_dpp %0, %2, %6 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %1, %12, %16 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"           \
        "v_fmac_f32_dpp %0, %2, %7 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %1, %12, %17 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"           \
        "v_fmac_f32_dpp %0, %3, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %1, %13, %18 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"           \
        "v_fmac_f32_dpp %0, %3, %9 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %1, %13, %19 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"           \
        "v_fmac_f32_dpp %0, %3, %10 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"            \
        "v_fmac_f32_dpp %1, %13, %20 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"           \
        "v_fmac_f32_dpp %0, %3, %11 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"            \
        "v_fmac_f32_dpp %1, %13, %21 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"                \
        : "+v"(ACC0), "+v"(ACC1)                                                                   \
        : "v"(PLO0), "v"(PHI0), "v"((A01).x), "v"((A01).y), "v"((A23).x), "v"((A23).y),            \
          "v"((A45).x), "v"((A45).y), "v"((A67).x), "v"((A67).y),                                  \
          "v"(PLO1), "v"(PHI1), "v"((B01).x), "v"((B01).y), "v"((B23).x), "v"((B23).y),            \
          "v"((B45).x), "v"((B45).y), "v"((B67).x), "v"((B67).y))

//   EPI_ROPE_KV (the wq|wk|wv matrix): the epilogue is k_rope_kv -- outputs 2i, 2i+1 of a row sit in lanes 8
//         apart of one DPP row, so the pair is rotated in place (double arithmetic, host-built cos/sin table)
//         and q goes to qr, k and v straight into the cache rows n_past + column: no fp32 qkv round trip,
//         no RoPE launch
//   EPI_SILU_QA (the interleaved w1|w3 matrix only, RG = 1): 8 waves per workgroup = 32 gate rows + the same 32
//         up rows; wave n of the workgroup then turns column n's 64 outputs into silu_lut(gate) * up
//         (ggml.c:1956-1963, .mm:678-680) and quantizes them as one Q4_0 activation block (ggml.c:456-523)
//         of the w2 mat-mul's operand: out_A / out_d, row strides out_strideA dwords / out_strideD floats
//         (no fp32 round trip, no preparation launch in between)
template <int NC, int RG, int EPI>
__global__ void __launch_bounds__(EPI == EPI_SILU_QA ? 512 : 256)
k_gemm_skinny(const uint8_t *__restrict__ wt, int ngroups, int nchunks, int M, int gmapF8,
              const uint32_t *__restrict__ qa_A, const float *__restrict__ qa_d, int ncols, int ncg,
              float *__restrict__ y, long y_stride, const float *__restrict__ resid, long resid_stride,
              const uint16_t *__restrict__ T_silu, uint32_t *__restrict__ out_A, float *__restrict__ out_d,
              long out_strideA, long out_strideD, RopeKvArgs ra) {
    constexpr int D = 4;
    constexpr int NW = EPI == EPI_SILU_QA ? 8 : 4, NT = NW * 64;
    static_assert(EPI != EPI_SILU_QA || RG == 1, "the fused FFN epilogue pairs one gate wave with one up wave");
    extern __shared__ double smem_d[];
    // every column's operand is padded with D zeroed chunks: the ring tail and the one-step-ahead operand
    // fetch run past the row end (against the zero tile) without an index clamp (see k_gemv)
    const int npad = nchunks + D;
    u32x4 *sA = (u32x4 *) smem_d;                            // [NC][npad * 16]
    f32x4 *sD = (f32x4 *) (sA + (size_t) NC * npad * 16);    // [NC][npad * 2]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x, xcd = b & 7, q = b >> 3, cg = q % ncg, wgi = (q / ncg) * 8 + xcd;
    const int g0 = (wgi * NW + wave) * RG;                    // first of this wave's RG consecutive row-groups
    const int n0 = cg * NC;
    const int k = lane & 7, t = lane & 3;
    const uint32_t voff_w = (uint32_t) lane * 16u, voff_s = 1024u + (uint32_t) ((lane >> 3) * 8 + t * 2) * 4u;
    uint32_t vw_ = voff_w, vs_ = voff_s;                     // (see k_gemv: SGPR base + per-lane offset addressing)
#define LH_OPAQUE_OFFSETS() { vw_ = voff_w; vs_ = voff_s; asm volatile("" : "+v"(vw_), "+v"(vs_)); }
    const uint8_t *wbase[RG];
#pragma unroll
    for (int rg = 0; rg < RG; rg++) wbase[rg] = wt + (size_t) min(g0 + rg, ngroups - 1) * (nchunks + 1) * TILE_BYTES;

    u32x4 wq[RG][D];
    f32x2 ws[RG][D];
#define LH_LOADW(SLOT, CH)                                                                         \
    _Pragma("unroll")                                                                              \
    for (int rg = 0; rg < RG; rg++) {                                                              \
        const uint8_t *tp_ = wbase[rg] + (size_t) min((CH), nchunks) * TILE_BYTES;                 \
        wq[rg][SLOT] = __builtin_nontemporal_load((const u32x4 *) (tp_ + (size_t) vw_));           \
        ws[rg][SLOT] = __builtin_nontemporal_load((const f32x2 *) (tp_ + (size_t) vs_));           \
    }
#pragma unroll
    for (int i = 0; i < D; i++) { LH_LOADW(i, i) }
    __builtin_amdgcn_sched_barrier(0);
    // stage the NC columns' operands (columns past ncols are clamped duplicates, never stored);
    // 8 loads per thread in flight per pass: a pass is one L2 round trip
    {
        constexpr int LB = 8;
        const int perA = nchunks * 16, perD = nchunks * 2;     // QA row strides in 16-byte granules
        const int totA = NC * perA, totD = NC * perD;
        for (int base = tid; base < totA; base += NT * LB) {
            u32x4 v[LB];
#pragma unroll
            for (int u = 0; u < LB; u++) {
                const int i = min(base + u * NT, totA - 1), n = i / perA, r = i - n * perA;
                v[u] = ((const u32x4 *) qa_A)[(long) min(n0 + n, ncols - 1) * perA + r];
            }
#pragma unroll
            for (int u = 0; u < LB; u++) {
                const int i = base + u * NT, n = i / perA, r = i - n * perA;
                if (i < totA) sA[n * npad * 16 + r] = v[u];
            }
        }
        for (int base = tid; base < totD; base += NT * LB) {
            f32x4 v[LB];
#pragma unroll
            for (int u = 0; u < LB; u++) {
                const int i = min(base + u * NT, totD - 1), n = i / perD, r = i - n * perD;
                v[u] = ((const f32x4 *) qa_d)[(long) min(n0 + n, ncols - 1) * perD + r];
            }
#pragma unroll
            for (int u = 0; u < LB; u++) {
                const int i = base + u * NT, n = i / perD, r = i - n * perD;
                if (i < totD) sD[n * npad * 2 + r] = v[u];
            }
        }
        for (int i = tid; i < NC * D * 18; i += NT) {         // zero the padding chunks
            const int n = i / (D * 18), r = i - n * (D * 18);
            if (r < D * 16) sA[(n * npad + nchunks) * 16 + r] = u32x4{ 0u, 0u, 0u, 0u };
            else sD[(n * npad + nchunks) * 2 + (r - D * 16)] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
        }
    }
    // The staging loops have run-time trip counts, after which the compiler's waitcnt pass no longer knows
    // how old the ring loads are and would put a vmcnt(0) at the top of the single-block main loop, i.e. in
    // EVERY iteration (no prefetch left).  Draining explicitly here makes the loop's entry state exact, and
    // the waits inside become the counted vmcnt(2 * RG * (D - 1)) of the back edge.
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0), nothing else
    __syncthreads();

    const float *sDf = (const float *) sD;
    float accs[RG][NC];
#pragma unroll
    for (int rg = 0; rg < RG; rg++)
#pragma unroll
        for (int n = 0; n < NC; n++) accs[rg][n] = 0.0f;
    // LDS operands of step (slot, column) are fetched one step ahead into the other half of a two-entry
    // register buffer (D * NC steps per loop trip is even, so the parity is a compile-time constant)
    u32x4 la0[2], la1[2];
    float ldl[2], ldh[2];
#define LH_LDSLOAD(BUF, N, CH)                                                                     \
    {                                                                                              \
        const u32x4 *pa_ = sA + ((N) * npad + (CH)) * 16 + k * 2;                                  \
        la0[BUF] = pa_[0]; la1[BUF] = pa_[1];                                                      \
        const float *pd_ = sDf + ((N) * npad + (CH)) * 8 + t;                                      \
        ldl[BUF] = pd_[0]; ldh[BUF] = pd_[4];                                                      \
    }
#define LH_CONSUME(SLOT, CH)                                                                       \
    {                                                                                              \
        _Pragma("unroll")                                                                          \
        for (int n = 0; n < NC; n++) {                                                             \
            const int pb_ = ((SLOT) * NC + n) & 1;                                                 \
            const u32x4 a0 = la0[pb_], a1 = la1[pb_];                                              \
            const float dlo_ = ldl[pb_], dhi_ = ldh[pb_];                                          \
            if (n + 1 < NC) LH_LDSLOAD(pb_ ^ 1, n + 1, (CH))                                       \
            else LH_LDSLOAD(pb_ ^ 1, 0, (CH) + 1)                                                  \
            __builtin_amdgcn_sched_barrier(0);     /* reads for the next step go out before this step's arithmetic */ \
            float plo_[RG], phi_[RG];                                                              \
            f32x2 q01_[RG], q23_[RG], q45_[RG], q67_[RG];                                          \
            _Pragma("unroll")                                                                      \
            for (int rg = 0; rg < RG; rg++) {                                                      \
                const u32x4 w = wq[rg][SLOT];                                                      \
                plo_[rg] = ws[rg][SLOT].x * dlo_; phi_[rg] = ws[rg][SLOT].y * dhi_;                \
                const int i0_ = __builtin_amdgcn_sdot8((int) w.x, (int) a0.x, 0x4B400000, true);   \
                const int i1_ = __builtin_amdgcn_sdot8((int) w.x, (int) a0.y, 0x4B400000, true);   \
                const int i2_ = __builtin_amdgcn_sdot8((int) w.y, (int) a0.z, 0x4B400000, true);   \
                const int i3_ = __builtin_amdgcn_sdot8((int) w.y, (int) a0.w, 0x4B400000, true);   \
                const int i4_ = __builtin_amdgcn_sdot8((int) w.z, (int) a1.x, 0x4B400000, true);   \
                const int i5_ = __builtin_amdgcn_sdot8((int) w.z, (int) a1.y, 0x4B400000, true);   \
                const int i6_ = __builtin_amdgcn_sdot8((int) w.w, (int) a1.z, 0x4B400000, true);   \
                const int i7_ = __builtin_amdgcn_sdot8((int) w.w, (int) a1.w, 0x4B400000, true);   \
                const f32x2 mg_ = { 12582912.0f, 12582912.0f };                                    \
                q01_[rg] = f32x2{ __builtin_bit_cast(float, i0_), __builtin_bit_cast(float, i1_) } - mg_; \
                q23_[rg] = f32x2{ __builtin_bit_cast(float, i2_), __builtin_bit_cast(float, i3_) } - mg_; \
                q45_[rg] = f32x2{ __builtin_bit_cast(float, i4_), __builtin_bit_cast(float, i5_) } - mg_; \
                q67_[rg] = f32x2{ __builtin_bit_cast(float, i6_), __builtin_bit_cast(float, i7_) } - mg_; \
            }                                                                                      \
            if (RG == 2) {                                                                         \
                LH_FMAC8_DPP2(accs[0][n], plo_[0], phi_[0], q01_[0], q23_[0], q45_[0], q67_[0],    \
                              accs[RG - 1][n], plo_[RG - 1], phi_[RG - 1], q01_[RG - 1], q23_[RG - 1], q45_[RG - 1], q67_[RG - 1]); \
            } else {                                                                               \
                LH_FMAC8_DPP(accs[0][n], plo_[0], phi_[0], q01_[0], q23_[0], q45_[0], q67_[0]);    \
            }                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
    }
    LH_LDSLOAD(0, 0, 0)
    // straight-line ring body (see k_gemv): chunks past the row end read the zero tile (scale 0)
    for (int c0 = 0; c0 < nchunks; c0 += D) {
        LH_OPAQUE_OFFSETS()
#pragma unroll
        for (int i = 0; i < D; i++) {
            LH_CONSUME(i, c0 + i)
            LH_LOADW(i, c0 + D + i)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef LH_CONSUME
#undef LH_LDSLOAD
#undef LH_LOADW
#undef LH_OPAQUE_OFFSETS

    if (EPI == EPI_SILU_QA) {
        // waves 0-3 hold gate rows wgi*32 .. +31, waves 4-7 the matching up rows
        float *gu = (float *) smem_d;                          // [NC][64]; the operand staging area is free again
        __syncthreads();
#pragma unroll
        for (int n = 0; n < NC; n++) {
            const float acc = fold8(accs[0][n]);
            if (k == 0) gu[n * 64 + wave * 8 + (lane >> 3)] = acc;
        }
        __syncthreads();
        if (wave < NC && n0 + wave < ncols && wgi * 8 < ngroups) {
            const int i = lane & 31;
            const float act = h2f_bits(T_silu[f2h_bits(gu[wave * 64 + i])]) * gu[wave * 64 + 32 + i];
            const float amax = max_lanes_0_31(fabsf(act));
            const float dd = amax / 7.0f;
            const float id = (amax != 0.0f) ? 7.0f / amax : 0.0f;
            const uint32_t nib = (uint32_t) ((int) __builtin_rintf(act * id)) & 0xF;       // signed nibble of (q - 8)
            const int kk = lane & 7;
            const uint32_t e0 = __shfl(nib, 2 * kk), e1 = __shfl(nib, 2 * kk + 1);
            const uint32_t e2 = __shfl(nib, 16 + 2 * kk), e3 = __shfl(nib, 17 + 2 * kk);
            const int bb = wgi, c = bb >> 3, j = bb & 7;
            uint32_t *oA = out_A + (size_t) (n0 + wave) * out_strideA;
            float *oD = out_d + (size_t) (n0 + wave) * out_strideD;
            if (lane < 8) oA[(c * 8 + kk) * 8 + j] = (e0 | (e1 << 8) | (e2 << 16) | (e3 << 24)) << (4 * (j & 1));
            if (lane == 0) oD[bb] = dd;
        }
        return;
    }
#pragma unroll
    for (int rg = 0; rg < RG; rg++) {
        const int g = g0 + rg;
        int lg = g;
        if (gmapF8) { const int blk = g >> 3, w8 = g & 7; lg = w8 < 4 ? blk * 4 + w8 : gmapF8 + blk * 4 + (w8 - 4); }
        const int m = lg * 8 + (lane >> 3);
#pragma unroll
        for (int n = 0; n < NC; n++) {
            float acc = fold8(accs[rg][n]);
            if (EPI == EPI_ROPE_KV) {
                // (ggml.c:7076-7131, .mm:586-611; see k_rope_kv) rows m, m^1 = lanes 8 apart; m is even iff the lane's row is
                const float up = dpp_f<0x108>(acc), dn = dpp_f<0x118>(acc);        // row_shl:8 / row_shr:8
                if (g < ngroups && k == 0 && m < M && n0 + n < ncols) {
                    const int which = m / ra.d, c = m - which * ra.d, pos = ra.n_past + n0 + n;
                    if (which == 2) {
                        ra.Vc[(size_t) pos * ra.d + c] = acc;
                    } else {
                        const int pe = (c % ra.dh) & ~1;
                        const double cs = ra.tab[(size_t) pos * ra.dh + pe], sn = ra.tab[(size_t) pos * ra.dh + pe + 1];
                        const double x0 = (double) ((c & 1) ? dn : acc), x1 = (double) ((c & 1) ? acc : up);
                        const float val = (c & 1) ? (float) (x0 * sn + x1 * cs) : (float) (x0 * cs - x1 * sn);
                        if (which == 0) ra.qr[(size_t) (n0 + n) * ra.d + c] = val;
                        else ra.Kc[(size_t) pos * ra.d + c] = val;
                    }
                }
                continue;
            }
            if (g < ngroups && k == 0 && m < M && n0 + n < ncols) {
                if (EPI == EPI_RESID) acc = acc + resid[(size_t) (n0 + n) * resid_stride + m];
                y[(size_t) (n0 + n) * y_stride + m] = acc;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Prompt path, row-per-lane: second resident copy of a matrix in ROW-LANE tiles.
//   tile (row-block R of 64 rows, chunk c) = 10240 B; a row-block is nchunks + 1 tiles, the last all-zero:
//     vector k = 0..7 : [64 lanes x 16 B]  lane = row: chain k of the row's 8 blocks (same 4 dwords a
//                       decode tile holds for lane (r, k))
//     vector 8, 9     : [64 lanes x 16 B]  the row's block scales s0..s3 | s4..s7
// A wave owns 64 rows x NC activation columns; every lane runs all 8 chains of ITS row, so the
// activation operand (column n, chunk c: 64 dwords + 8 scales) is the same for the whole wave: it is
// fetched with scalar loads and fed to v_dot8_i32_i4 / v_mul_f32 as an SGPR operand -- no LDS, no
// barriers, no cross-lane traffic, and the d_w*d_a product is shared by the 8 chains of a block
// (25 VALU instructions per row x block x column instead of 32).  Same arithmetic, same order.
// ------------------------------------------------------------------------------------------------
constexpr int ROWTILE_BYTES = 10240;

// decode tiles -> row-lane tiles (load time).  One thread per (row-block, chunk, vector, lane).
__global__ void k_tiles_to_rows(const uint8_t *__restrict__ tiles, uint8_t *__restrict__ rows,
                                int ngroups, int nchunks, int nrb, int gmapF8) {
    const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long) nrb * (nchunks + 1) * 10 * 64;
    if (gid >= total) return;
    const int lane = (int) (gid & 63);
    const long t = gid >> 6;
    const int v = (int) (t % 10);
    const int c = (int) ((t / 10) % (nchunks + 1));      // c == nchunks: the all-zero tile closing the row-block
    const int rb = (int) (t / 10 / (nchunks + 1));
    const int m = rb * 64 + lane, lg = m >> 3, r = m & 7;
    u32x4 out = { 0u, 0u, 0u, 0u };
    if (lg < ngroups && c < nchunks) {
        int tg = lg;
        if (gmapF8) tg = lg < gmapF8 ? (lg >> 2) * 8 + (lg & 3) : ((lg - gmapF8) >> 2) * 8 + 4 + ((lg - gmapF8) & 3);
        const uint8_t *tp = tiles + ((size_t) tg * (nchunks + 1) + c) * TILE_BYTES;
        if (v < 8) {
            out = *(const u32x4 *) (tp + (r * 8 + v) * 16);
        } else {
            const uint32_t *sp = (const uint32_t *) (tp + 1024 + r * 32);     // stored [s0,s4,s1,s5,s2,s6,s3,s7]
            const int o = (v - 8);
            out = u32x4{ sp[0 + o], sp[2 + o], sp[4 + o], sp[6 + o] };
        }
    }
    *(u32x4 *) (rows + ((size_t) rb * (nchunks + 1) + c) * ROWTILE_BYTES + v * 1024 + lane * 16) = out;
}

//   DB  : double-buffer the weight chunk in registers (next chunk in flight during the arithmetic);
//         without it the wave stalls on every chunk and the other waves of the SIMD cover -- fewer
//         registers, more waves
//   WPE : occupancy target (waves per SIMD) the register allocator must honour
template <int NC, int EPI, bool DB, int WPE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE)))
k_gemm_rows(const uint8_t *__restrict__ wr, int nrb, int nchunks, int M,
            const uint32_t *__restrict__ qa_A, const float *__restrict__ qa_d, int ncols, int ncg,
            float *__restrict__ y, long y_stride, const float *__restrict__ resid, long resid_stride) {
    // XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs; give every XCD its own
    // row-blocks (rb % 8) and walk the column groups of one row-block back to back, so the weight
    // tiles a column group streams are still in that XCD's L2 for the next one
    // (one wave per workgroup: grouping 4 row-blocks of the same columns into a workgroup, to share the
    // scalar-cache lines of the operand, measured 3 % slower)
    const int b = blockIdx.x, xcd = b & 7, q = b >> 3;
    const int cg = q % ncg, rb = (q / ncg) * 8 + xcd;
    if (rb >= nrb) return;
    const int lane = threadIdx.x;
    const int n0 = cg * NC;
    const uint8_t *wbase = wr + (size_t) rb * (nchunks + 1) * ROWTILE_BYTES + lane * 16;
    const long strideA = (long) nchunks * 64, strideD = (long) nchunks * 8;      // per column, in dwords / floats
    float acc[NC][8];
#pragma unroll
    for (int n = 0; n < NC; n++)
#pragma unroll
        for (int k = 0; k < 8; k++) acc[n][k] = 0.0f;

// one chunk (registers WQ[8], scales SA/SB) against the NC columns at activation chunk CA
#define LH_ROWS_CONSUME(WQ, SA, SB, CA)                                                                        \
    {                                                                                                          \
        const float sw_[8] = { (SA).x, (SA).y, (SA).z, (SA).w, (SB).x, (SB).y, (SB).z, (SB).w };               \
        _Pragma("unroll") for (int n = 0; n < NC; n++) {                                                       \
            const int col_ = min(n0 + n, ncols - 1);           /* wave-uniform: scalar loads below */          \
            const uint32_t *Ap_ = qa_A + col_ * strideA + (long) (CA) * 64;                                    \
            const float *Dp_ = qa_d + col_ * strideD + (long) (CA) * 8;                                        \
            _Pragma("unroll") for (int j = 0; j < 8; j++) {                                                    \
                const float sc_ = sw_[j] * Dp_[j];                                                             \
                _Pragma("unroll") for (int k = 0; k < 8; k++) {                                                \
                    const uint32_t wd_ = (j >> 1) == 0 ? (WQ)[k].x : (j >> 1) == 1 ? (WQ)[k].y : (j >> 1) == 2 ? (WQ)[k].z : (WQ)[k].w; \
                    /* int -> float without v_cvt: accumulate onto the bit pattern of 1.5 * 2^23 (ulp 1), so the  */ \
                    /* result IS the float 12582912 + isum; subtracting the constant is exact and pairs up as   */ \
                    /* v_pk_add_f32 (|isum| <= 8 * 7 * 8 * 8 never leaves the binade)                           */ \
                    const int p_ = __builtin_amdgcn_sdot8((int) wd_, (int) Ap_[k * 8 + j], 0x4B400000, true);  \
                    acc[n][k] = fmaf(sc_, __builtin_bit_cast(float, p_) - 12582912.0f, acc[n][k]);             \
                }                                                                                              \
            }                                                                                                  \
        }                                                                                                      \
    }
#define LH_ROWS_LOAD(WQ, SA, SB, CH)                                                                           \
    {                                                                                                          \
        const uint8_t *tp_ = wbase + (size_t) (CH) * ROWTILE_BYTES;                                            \
        _Pragma("unroll") for (int k = 0; k < 8; k++) (WQ)[k] = __builtin_nontemporal_load((const u32x4 *) (tp_ + k * 1024)); \
        (SA) = __builtin_nontemporal_load((const f32x4 *) (tp_ + 8192));                                       \
        (SB) = __builtin_nontemporal_load((const f32x4 *) (tp_ + 9216));                                       \
    }
    if constexpr (NC >= 2 && DB) {
        // column groups: accumulators take the registers (8 * NC) and one chunk of arithmetic
        // (>= 1000 VALU instructions) covers the next chunk's load latency: double buffer
        u32x4 w[8], wn[8];
        f32x4 s0, s1, s0n, s1n;
        LH_ROWS_LOAD(wn, s0n, s1n, 0)
        for (int c = 0; c < nchunks; c++) {
#pragma unroll
            for (int k = 0; k < 8; k++) w[k] = wn[k];
            s0 = s0n; s1 = s1n;
            LH_ROWS_LOAD(wn, s0n, s1n, min(c + 1, nchunks - 1))
            LH_ROWS_CONSUME(w, s0, s1, c)
        }
    } else if constexpr (NC >= 2) {
        u32x4 w[8];
        f32x4 s0, s1;
        for (int c = 0; c < nchunks; c++) {
            LH_ROWS_LOAD(w, s0, s1, c)
            LH_ROWS_CONSUME(w, s0, s1, c)
        }
    } else {
        // single columns (short prompts: the reference feeds 9 tokens at a time): the wave is alone on
        // its SIMD and walks K serially, so nothing may sit on its critical path but the arithmetic:
        //   * weights: a ring of RD chunks with RD - 1 in flight.  Straight-line body (no branch around loads,
        //     see k_gemv): chunks past the row end are the zero tile closing the row-block (scale 0);
        //   * the column's whole operand (K + K/8 bytes) is copied to LDS once and read back with broadcast
        //     ds_reads one block pair ahead -- scalar loads per chunk cannot be prefetched (a chunk's operand
        //     is 72 of the ~100 SGPRs) and cost this kernel a scalar-cache round trip per chunk.
        extern __shared__ __attribute__((aligned(16))) uint32_t lds_op[];   // [nchunks * 64] A dwords | [nchunks * 8] da
        {
            const int col = min(n0, ncols - 1);
            const u32x4 *ga = (const u32x4 *) (qa_A + col * strideA);
            const f32x4 *gd = (const f32x4 *) (qa_d + col * strideD);
            for (int i = lane; i < nchunks * 16; i += 64) ((u32x4 *) lds_op)[i] = ga[i];
            for (int i = lane; i < nchunks * 2; i += 64) ((f32x4 *) (lds_op + nchunks * 64))[i] = gd[i];
        }
        constexpr int RD = 3;                                     // 3 x 40 VGPRs; a 4th slot spills
        u32x4 w[RD][8];
        f32x4 s0[RD], s1[RD];
#pragma unroll
        for (int i = 0; i < RD; i++) LH_ROWS_LOAD(w[i], s0[i], s1[i], min(i, nchunks))
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        const float *lds_d = (const float *) (lds_op + nchunks * 64);
        for (int c0 = 0; c0 < nchunks; c0 += RD) {
#pragma unroll
            for (int i = 0; i < RD; i++) {
                const int c = c0 + i, ca = min(c, nchunks - 1);
                const float sw[8] = { s0[i].x, s0[i].y, s0[i].z, s0[i].w, s1[i].x, s1[i].y, s1[i].z, s1[i].w };
                const f32x4 d0 = *(const f32x4 *) (lds_d + ca * 8), d1 = *(const f32x4 *) (lds_d + ca * 8 + 4);
                const float da[8] = { d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w };
#pragma unroll
                for (int h = 0; h < 2; h++) {                     // blocks 4h .. 4h + 3 of every chain: 8 ds_read_b128
                    u32x4 a[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) a[k] = *(const u32x4 *) (lds_op + ca * 64 + k * 8 + 4 * h);
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        const int j = 4 * h + jj;
                        const float sc = sw[j] * da[j];
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const uint32_t wd = (j >> 1) == 0 ? w[i][k].x : (j >> 1) == 1 ? w[i][k].y : (j >> 1) == 2 ? w[i][k].z : w[i][k].w;
                            const uint32_t ad = jj == 0 ? a[k].x : jj == 1 ? a[k].y : jj == 2 ? a[k].z : a[k].w;
                            const int p = __builtin_amdgcn_sdot8((int) wd, (int) ad, 0x4B400000, true);
                            acc[0][k] = fmaf(sc, __builtin_bit_cast(float, p) - 12582912.0f, acc[0][k]);
                        }
                        if (jj & 1) __builtin_amdgcn_sched_barrier(0);      // bound the live temporaries
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                LH_ROWS_LOAD(w[i], s0[i], s1[i], min(c + RD, nchunks))
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#undef LH_ROWS_CONSUME
#undef LH_ROWS_LOAD
    const int m = rb * 64 + lane;
#pragma unroll
    for (int n = 0; n < NC; n++) {
        // the reference's lane fold: ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7))  (ggml.c:872-887 tree)
        float r = ((acc[n][0] + acc[n][4]) + (acc[n][2] + acc[n][6])) + ((acc[n][1] + acc[n][5]) + (acc[n][3] + acc[n][7]));
        if (m < M && n0 + n < ncols) {
            if (EPI == EPI_RESID) r = r + resid[(size_t) (n0 + n) * resid_stride + m];
            y[(size_t) (n0 + n) * y_stride + m] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Prompt path on the matrix cores, still bit-exact.
// The reference needs, per output and Q4_0 block, EIGHT separate 4-element integer sums (one per
// lane of its AVX accumulator), each scaled and FMA-accumulated on its own chain -- an MFMA sums
// over its whole K.  So the activation operand is MASKED: v_mfma_i32_32x32x32_i8 (K = 32 = one
// block) is issued once per chain with every byte of B zeroed except that chain's 4 elements; the
// product is that chain's exact integer sum for a 32 x 32 tile of outputs.  7/8 of the MACs multiply
// zeros, which the matrix pipe has to spare, and the VALU is left with what cannot be avoided: the
// conversion and the scaled FMA, both packed (v_pk_add_f32 / v_pk_fma_f32), 16 outputs per lane.
//   * A = weights as int8 = signed nibble << 4 (two VALU per 8 nibbles); the x16 is undone for free
//     by accumulating onto the bit pattern of 1.5 * 2^19 (ulp 1/16): D IS the float 786432 + isum.
//   * d_w * d_a: 16 products per lane and block, shared by the 8 chains.
// Third resident copy of a matrix ("mtiles"): tile (row-block of 32, quad of 4 blocks) = 2560 B:
//   [j 0..3][lane 0..63][8 B]  lane = m + 32 * kg (kg = K-half: elements 16kg..16kg+15 of block 4q+j);
//                              dword 0 = signed nibbles of (k = 0..3, p = 0,1), dword 1 = k = 4..7,
//                              nibble index 2 * (k & 3) + p  <->  element 2k + p + 16kg
//   [j][32 rows] fp32 scales
// Activation operand "QB" (k_qa_to_qb): per column [block][kg][16 int8] in the matching K order:
//   dword t = (k >> 2) * 2 + p, byte k & 3.
// Workgroup = 4 waves = 64 rows x 64 columns, operands of one quad staged in LDS (double-buffered).
// ------------------------------------------------------------------------------------------------
constexpr int MTILE_BYTES = 2560;
typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef int i32x16v __attribute__((ext_vector_type(16)));

// decode tiles -> mtiles (load time).  One thread per (row-block, quad, j, lane) for the nibbles,
// plus the scales.
__global__ void k_tiles_to_mtiles(const uint8_t *__restrict__ tiles, uint8_t *__restrict__ mt,
                                  int ngroups, int nchunks, int nrb32, int gmapF8) {
    const int nq = nchunks * 2;
    const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long) nrb32 * nq * 4 * 64;
    if (gid >= total) return;
    const int lane = (int) (gid & 63), j = (int) ((gid >> 6) & 3);
    const long t = gid >> 8;
    const int q = (int) (t % nq), rb = (int) (t / nq);
    const int m = lane & 31, kg = lane >> 5;
    const int row = rb * 32 + m, lg = row >> 3, r = row & 7;
    const int b = q * 4 + j, c = b >> 3, jj = b & 7, i = jj >> 1, half = jj & 1;
    uint32_t x0 = 0, x1 = 0;
    float d = 0.0f;
    if (lg < ngroups) {
        int tg = lg;
        if (gmapF8) tg = lg < gmapF8 ? (lg >> 2) * 8 + (lg & 3) : ((lg - gmapF8) >> 2) * 8 + 4 + ((lg - gmapF8) & 3);
        const uint8_t *tp = tiles + ((size_t) tg * (nchunks + 1) + c) * TILE_BYTES;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t dw = ((const uint32_t *) (tp + (r * 8 + k) * 16))[i];       // chain k, blocks (2i, 2i+1)
#pragma unroll
            for (int pp = 0; pp < 2; pp++) {
                const uint32_t nib = (dw >> (8 * (2 * kg + pp) + 4 * half)) & 0xF;        // element 2k + pp + 16kg, already signed
                if (k < 4) x0 |= nib << (4 * (2 * k + pp)); else x1 |= nib << (4 * (2 * (k - 4) + pp));
            }
        }
        // scales of a row are stored [s0,s4,s1,s5,s2,s6,s3,s7]
        d = ((const float *) (tp + 1024 + r * 32))[(jj & 3) * 2 + (jj >> 2)];
    }
    uint8_t *o = mt + ((size_t) rb * nq + q) * MTILE_BYTES;
    ((uint32_t *) (o + j * 512 + lane * 8))[0] = x0;
    ((uint32_t *) (o + j * 512 + lane * 8))[1] = x1;
    if (kg == 0) ((float *) (o + 2048))[j * 32 + m] = d;
}

// QA (chain-major signed nibbles) -> QB (int8, MFMA K order).  One thread per (column, block, kg).
__global__ void k_qa_to_qb(const uint32_t *__restrict__ qa_A, uint8_t *__restrict__ qb, int nchunks, int N) {
    const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
    const int nbp = nchunks * 8;                           // blocks per column incl. padding
    const long total = (long) N * nbp * 2;
    if (gid >= total) return;
    const int kg = (int) (gid & 1);
    const long t = gid >> 1;
    const int b = (int) (t % nbp), n = (int) (t / nbp);
    const int c = b >> 3, jj = b & 7;
    const uint32_t *src = qa_A + (size_t) n * nchunks * 64 + c * 64 + jj;
    uint32_t out[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t dw = src[k * 8];
#pragma unroll
        for (int pp = 0; pp < 2; pp++) {
            const int nib = (int) ((dw >> (8 * (2 * kg + pp) + 4 * (jj & 1))) & 0xF);
            const uint32_t v = (uint32_t) ((nib ^ 8) - 8) & 0xFF;                           // sign-extend 4 -> 8 bits
            out[(k >> 2) * 2 + pp] |= v << (8 * (k & 3));
        }
    }
    *(u32x4 *) (qb + ((size_t) n * nbp + b) * 32 + kg * 16) = u32x4{ out[0], out[1], out[2], out[3] };
}

//   FAST (LLAMAHIP_FLAG_FAST_PREFILL, opt-in, NOT the reference's arithmetic): one unmasked MFMA per Q4_0 block -- the
//        whole 32-element integer sum -- and ONE fp32 FMA chain per output instead of eight: 8x fewer MFMAs, conversions
//        and FMAs.  Sums are re-associated (the 8 lane partials of _mm256_madd_epi16 are added as integers before the
//        scale), so logits agree with the exact path only to rounding and the next activation quantization can flip
//        codes; never used for parity claims.
template <int EPI, bool FAST>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
k_gemm_mfma(const uint8_t *__restrict__ mt, int nrb32, int nq, int M,
            const uint8_t *__restrict__ qb, const float *__restrict__ qa_d, int ncols, int nct,
            float *__restrict__ y, long y_stride, const float *__restrict__ resid, long resid_stride) {
    constexpr int BSTRIDE = 144;                            // 128 B of a column's quad + 16 B pad: conflict-free b128 reads
    __shared__ __attribute__((aligned(16))) uint8_t sW[2][2][MTILE_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t sB[2][64 * BSTRIDE];
    __shared__ __attribute__((aligned(16))) float sDa[2][64 * 4];
    // XCD-aware: a row-pair's column tiles run back to back on one XCD (its weights stay in that L2)
    const int bid = blockIdx.x, xcd = bid & 7, qq = bid >> 3;
    const int ct = qq % nct, rp = (qq / nct) * 8 + xcd;
    if (rp * 2 >= nrb32) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave & 1, wc = wave >> 1;
    const int n0 = ct * 64;
    const int nbp = nq * 4;
    const long strideD = (long) nq * 4;                    // floats per column in qa_d

    // ---- global -> registers -> LDS staging of one quad
    u32x4 gw[2], gb[2];
    f32x4 gd;
    auto fetch = [&](int q) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int g = tid + u * 256;                   // 320 granules of weights (2 tiles x 160)
            const int tile = min(g / 160, 1), off = (g % 160) * 16;
            const int rb = min(rp * 2 + tile, nrb32 - 1);
            gw[u] = *(const u32x4 *) (mt + ((size_t) rb * nq + q) * MTILE_BYTES + off);
            const int col = min(n0 + (g >> 3), ncols - 1), part = g & 7;     // 512 granules of activations
            gb[u] = *(const u32x4 *) (qb + ((size_t) col * nbp + q * 4) * 32 + part * 16);
        }
        gd = *(const f32x4 *) (qa_d + (size_t) min(n0 + (tid & 63), ncols - 1) * strideD + q * 4);
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int g = tid + u * 256;
            if (g < 320) *(u32x4 *) (&sW[buf][g / 160][(g % 160) * 16]) = gw[u];
            *(u32x4 *) (&sB[buf][(g >> 3) * BSTRIDE + (g & 7) * 16]) = gb[u];
        }
        if (tid < 64) *(f32x4 *) (&sDa[buf][tid * 4]) = gd;
    };

    constexpr int NCH = FAST ? 1 : 8;
    f32x2 acc[NCH][8];                                      // [chain][pair of adjacent C/D registers]
#pragma unroll
    for (int k = 0; k < NCH; k++)
#pragma unroll
        for (int r = 0; r < 8; r++) acc[k][r] = f32x2{ 0.0f, 0.0f };
    i32x16v cm;
#pragma unroll
    for (int r = 0; r < 16; r++) cm[r] = 0x49400000;       // 1.5 * 2^19: ulp 1/16

    const bool second_tile_real = rp * 2 + 1 < nrb32;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int q = 0; q < nq; q++) {
        const int buf = q & 1;
        if (q + 1 < nq) fetch(q + 1);
        const uint8_t *wt_ = sW[buf][wr];
        const uint8_t *bt_ = &sB[buf][(wc * 32 + (lane & 31)) * BSTRIDE + (lane >> 5) * 16];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t x0 = ((const uint32_t *) (wt_ + j * 512 + lane * 8))[0];
            const uint32_t x1 = ((const uint32_t *) (wt_ + j * 512 + lane * 8))[1];
            const i32x4v A = { (int) ((x0 << 4) & 0xF0F0F0F0u), (int) (x0 & 0xF0F0F0F0u), (int) ((x1 << 4) & 0xF0F0F0F0u), (int) (x1 & 0xF0F0F0F0u) };
            const u32x4 B = *(const u32x4 *) (bt_ + j * 32);
            const float da = sDa[buf][(wc * 32 + (lane & 31)) * 4 + j];
            f32x2 sc[8];
            const f32x2 da2 = { da, da };
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x4 dw = *(const f32x4 *) (wt_ + 2048 + (j * 32 + 8 * g + 4 * (lane >> 5)) * 4);
                sc[2 * g + 0] = f32x2{ dw.x, dw.y } * da2;
                sc[2 * g + 1] = f32x2{ dw.z, dw.w } * da2;
            }
            if constexpr (FAST) {
                const i32x4v Bi = { (int) B.x, (int) B.y, (int) B.z, (int) B.w };
                const i32x16v Df = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, Bi, cm, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int d0 = Df[2 * r], d1 = Df[2 * r + 1];
                    const f32x2 qv = f32x2{ __builtin_bit_cast(float, d0), __builtin_bit_cast(float, d1) } - f32x2{ 786432.0f, 786432.0f };
                    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[0][r]) : "v"(sc[r]), "v"(qv));
                }
                continue;
            }
            // chain k: B with every byte but that chain's 4 elements zeroed.  Two MFMAs stay in flight
            // ahead of the packed conversion + FMA of a chain.  Left alone, the scheduler issues all 32
            // MFMAs of a quad first and spills their 512 result registers, so the order is pinned with
            // empty volatile asms (they keep their program order): "use" all 16 accumulators of chain k,
            // then "define" the operand of chain k + 2.
            if constexpr (!FAST) {
            i32x16v D[2];
#define LH_MFMA(K, PIN)                                                                            \
            {                                                                                      \
                const uint32_t mask_ = 0xFFu << (8 * ((K) & 3));                                   \
                i32x4v Bk_ = { 0, 0, 0, 0 };                                                       \
                if ((K) < 4) { Bk_.x = (int) (B.x & mask_); Bk_.y = (int) (B.y & mask_); }         \
                else         { Bk_.z = (int) (B.z & mask_); Bk_.w = (int) (B.w & mask_); }         \
                if (PIN) { if ((K) < 4) asm volatile("" : "+v"(Bk_.x)); else asm volatile("" : "+v"(Bk_.z)); } \
                D[(K) % 2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, Bk_, cm, 0, 0, 0);           \
            }
            LH_MFMA(0, true)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (k + 1 < 8) LH_MFMA(k + 1, true)             // in flight behind the consumption of chain k
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    // (scalar copies: __builtin_bit_cast on a vector ELEMENT reads element 0)
                    const int d0 = D[k % 2][2 * r], d1 = D[k % 2][2 * r + 1];
                    const f32x2 qv = f32x2{ __builtin_bit_cast(float, d0), __builtin_bit_cast(float, d1) } - f32x2{ 786432.0f, 786432.0f };
                    // in-place packed FMA (tied operand): left to the register allocator, the 128 accumulators
                    // come out of the loop body in other registers than they went in (~100 copies per block)
                    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[k][r]) : "v"(sc[r]), "v"(qv));
                }
                asm volatile("" :: "v"(acc[k][0]), "v"(acc[k][1]), "v"(acc[k][2]), "v"(acc[k][3]),
                             "v"(acc[k][4]), "v"(acc[k][5]), "v"(acc[k][6]), "v"(acc[k][7]));
            }
            }
#undef LH_MFMA
        }
        if (q + 1 < nq) stash(buf ^ 1);
        __syncthreads();
    }
    (void) second_tile_real;
    // ---- fold the 8 chains (ggml.c:872-887 tree) and store: lane = column, 16 rows (C/D layout)
    const int n = n0 + wc * 32 + (lane & 31);
    const int mb = (rp * 2 + wr) * 32 + 4 * (lane >> 5);
    if (n < ncols && rp * 2 + wr < nrb32) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = mb + (r & 3) + 8 * (r >> 2);
#define LH_A(K) ((r & 1) ? acc[(K) % NCH][r >> 1].y : acc[(K) % NCH][r >> 1].x)
            float v = FAST ? LH_A(0) : ((LH_A(0) + LH_A(4)) + (LH_A(2) + LH_A(6))) + ((LH_A(1) + LH_A(5)) + (LH_A(3) + LH_A(7)));
#undef LH_A
            if (m < M) {
                if (EPI == EPI_RESID) v = v + resid[(size_t) n * resid_stride + m];
                y[(size_t) n * y_stride + m] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same exact product on v_mfma_f32_32x32x4_2b_f16: K = 4 is exactly one chain of a Q4_0 block (the 4 elements
// {2k, 2k+1, 16+2k, 17+2k} that one lane of the reference's _mm256_madd_epi16 sums), the instruction carries TWO
// independent 32 x 32 x 4 products ("blocks" = lane halves on the operand side, result registers 0-15 / 16-31 --
// checked on the hardware, tools/mfma_layout_probe.hip), so one issue returns two chains' sums for a 32 x 32 tile:
//   * nothing is masked (the int8 kernel above issues one 32 x 32 x 32 MFMA per chain with 7/8 of its operand zeroed);
//   * the sums arrive as FLOATS (small integers are exact in fp16 operands and fp32 accumulation), so the
//     integer -> float conversion of the int8 kernel (one packed subtraction per pair of outputs, as many VALU issues
//     as the FMAs themselves) disappears: what is left per output and chain is the one FMA the reference defines.
// Matrix-pipe time per Q4_0 block and 32 x 32 tile is the same (4 issues of 16 passes = 8 of 8), VALU work drops from
// ~190 to ~110 instructions.
// Weight copy "mt16" (same size as the int8 tiles, replaces them): tile (row-block of 32, quad of 4 blocks) = 2560 B:
//   [j 0..3][lane 0..63][8 B]   lane = m + 32 * g: the 16 nibbles of chains {g, 2 + g, 4 + g, 6 + g} of block 4q + j, row m,
//                               BIASED (q = n + 8, 0..15) and placed so that `(x >> 4s) & 0x000F000F | 0x64006400` is the
//                               fp16 pair (1024 + q_lo, 1024 + q_hi): dword 0 serves issues 0, 1 (chains g, 2 + g), dword 1
//                               issues 2, 3; within a dword s = 0: (e0, e1) of the even issue, s = 1: (e2, e3), s = 2, 3: odd issue
//   [j][32 rows] fp32 scales
// Activation operand "QB16" (k_qa_to_qb16): per column [block][g][issue 0..3][4 fp16] = 64 B, exact integers -8..7.
// Workgroup = 4 waves = 64 rows x 64 columns, operands of one quad staged in LDS (double-buffered), as above.
// ------------------------------------------------------------------------------------------------
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef float f32x32v __attribute__((ext_vector_type(32)));

__global__ void k_tiles_to_mt16(const uint8_t *__restrict__ tiles, uint8_t *__restrict__ mt,
                                int ngroups, int nchunks, int nrb32, int gmapF8) {
    const int nq = nchunks * 2;
    const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long) nrb32 * nq * 4 * 64;
    if (gid >= total) return;
    const int lane = (int) (gid & 63), j = (int) ((gid >> 6) & 3);
    const long t = gid >> 8;
    const int q = (int) (t % nq), rb = (int) (t / nq);
    const int m = lane & 31, g = lane >> 5;
    const int row = rb * 32 + m, lg = row >> 3, r = row & 7;
    const int b = q * 4 + j, c = b >> 3, jj = b & 7, i = jj >> 1, half = jj & 1;
    uint32_t x[2] = { 0x88888888u ^ 0x88888888u, 0u };     // biased zero is 8: padding rows get q = 8 everywhere below
    x[0] = 0x88888888u; x[1] = 0x88888888u;
    float d = 0.0f;
    if (lg < ngroups) {
        int tg = lg;
        if (gmapF8) tg = lg < gmapF8 ? (lg >> 2) * 8 + (lg & 3) : ((lg - gmapF8) >> 2) * 8 + 4 + ((lg - gmapF8) & 3);
        const uint8_t *tp = tiles + ((size_t) tg * (nchunks + 1) + c) * TILE_BYTES;
        x[0] = x[1] = 0u;
#pragma unroll
        for (int ii = 0; ii < 4; ii++) {
            const int kc = 2 * ii + g;
            const uint32_t dw = ((const uint32_t *) (tp + (r * 8 + kc) * 16))[i];       // chain kc, blocks (2i, 2i+1): byte p = element e_p
            uint32_t e[4];
#pragma unroll
            for (int pp = 0; pp < 4; pp++) e[pp] = (((dw >> (8 * pp + 4 * half)) & 0xFu) ^ 8u);     // signed nibble -> biased q
            const int sh = (ii & 1) * 8;
            x[ii >> 1] |= (e[0] << sh) | (e[1] << (16 + sh)) | (e[2] << (4 + sh)) | (e[3] << (20 + sh));
        }
        d = ((const float *) (tp + 1024 + r * 32))[(jj & 3) * 2 + (jj >> 2)];
    }
    uint8_t *o = mt + ((size_t) rb * nq + q) * MTILE_BYTES;
    ((uint32_t *) (o + j * 512 + lane * 8))[0] = x[0];
    ((uint32_t *) (o + j * 512 + lane * 8))[1] = x[1];
    if (g == 0) ((float *) (o + 2048))[j * 32 + m] = d;
}

// QA (chain-major signed nibbles) -> QB16.  One thread per (column, block, chain).
__global__ void k_qa_to_qb16(const uint32_t *__restrict__ qa_A, uint8_t *__restrict__ qb, int nchunks, int N) {
    const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
    const int nbp = nchunks * 8;
    const long total = (long) N * nbp * 8;
    if (gid >= total) return;
    const int kc = (int) (gid & 7);
    const long t = gid >> 3;
    const int b = (int) (t % nbp), n = (int) (t / nbp);
    const int c = b >> 3, jj = b & 7;
    const uint32_t dw = qa_A[(size_t) n * nchunks * 64 + c * 64 + kc * 8 + jj];
    h4v v;
#pragma unroll
    for (int pp = 0; pp < 4; pp++) {
        const int nib = (int) ((dw >> (8 * pp + 4 * (jj & 1))) & 0xF);
        v[pp] = (_Float16) (float) ((nib ^ 8) - 8);
    }
    const int g = kc & 1, ii = kc >> 1;
    *(h4v *) (qb + ((size_t) n * nbp + b) * 64 + g * 32 + ii * 8) = v;
}

template <int EPI>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_gemm_mfma16(const uint8_t *__restrict__ mt, int nrb32, int nq, int M,
              const uint8_t *__restrict__ qb, const float *__restrict__ qa_d, int ncols, int nct,
              float *__restrict__ y, long y_stride, const float *__restrict__ resid, long resid_stride) {
    constexpr int CS = 272;                                 // 256 B of a column's quad + 16 B pad: conflict-free b128 reads
    __shared__ __attribute__((aligned(16))) uint8_t sW[2][2][MTILE_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t sB[2][64 * CS];
    __shared__ __attribute__((aligned(16))) float sDa[2][64 * 4];
    const int bid = blockIdx.x, xcd = bid & 7, qq = bid >> 3;
    const int ct = qq % nct, rp = (qq / nct) * 8 + xcd;
    if (rp * 2 >= nrb32) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave & 1, wc = wave >> 1;
    const int n0 = ct * 64;
    const int nbp = nq * 4;
    const long strideD = (long) nq * 4;

    u32x4 gw[2], gb[4];
    f32x4 gd;
    auto fetch = [&](int q) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int g = tid + u * 256;                   // 320 granules of weights (2 tiles x 160)
            const int tile = min(g / 160, 1), off = (g % 160) * 16;
            const int rb = min(rp * 2 + tile, nrb32 - 1);
            gw[u] = *(const u32x4 *) (mt + ((size_t) rb * nq + q) * MTILE_BYTES + off);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int g = tid + u * 256;                   // 1024 granules of activations: 64 columns x 16
            const int col = min(n0 + (g >> 4), ncols - 1), part = g & 15;
            gb[u] = *(const u32x4 *) (qb + ((size_t) col * nbp + q * 4) * 64 + part * 16);
        }
        gd = *(const f32x4 *) (qa_d + (size_t) min(n0 + (tid & 63), ncols - 1) * strideD + q * 4);
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int g = tid + u * 256;
            if (g < 320) *(u32x4 *) (&sW[buf][g / 160][(g % 160) * 16]) = gw[u];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int g = tid + u * 256;
            *(u32x4 *) (&sB[buf][(g >> 4) * CS + (g & 15) * 16]) = gb[u];
        }
        if (tid < 64) *(f32x4 *) (&sDa[buf][tid * 4]) = gd;
    };

    f32x2 acc[8][8];                                        // [chain][pair of adjacent C/D registers]
#pragma unroll
    for (int k = 0; k < 8; k++)
#pragma unroll
        for (int r = 0; r < 8; r++) acc[k][r] = f32x2{ 0.0f, 0.0f };
    f32x32v zero32;
#pragma unroll
    for (int r = 0; r < 32; r++) zero32[r] = 0.0f;

    fetch(0);
    stash(0);
    __syncthreads();
    for (int q = 0; q < nq; q++) {
        const int buf = q & 1;
        if (q + 1 < nq) fetch(q + 1);
        const uint8_t *wt_ = sW[buf][wr];
        const uint8_t *bt_ = &sB[buf][(wc * 32 + (lane & 31)) * CS + (lane >> 5) * 32];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t x0 = ((const uint32_t *) (wt_ + j * 512 + lane * 8))[0];
            const uint32_t x1 = ((const uint32_t *) (wt_ + j * 512 + lane * 8))[1];
            const u32x4 B0 = *(const u32x4 *) (bt_ + j * 64), B1 = *(const u32x4 *) (bt_ + j * 64 + 16);      // issues 0,1 | 2,3
            const float da = sDa[buf][(wc * 32 + (lane & 31)) * 4 + j];
            f32x2 sc[8];
            const f32x2 da2 = { da, da };
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x4 dw = *(const f32x4 *) (wt_ + 2048 + (j * 32 + 8 * g + 4 * (lane >> 5)) * 4);
                sc[2 * g + 0] = f32x2{ dw.x, dw.y } * da2;
                sc[2 * g + 1] = f32x2{ dw.z, dw.w } * da2;
            }
            // Two waves per SIMD (256 registers each: 128 accumulators + one result set + operands + the next quad's
            // staging): while one wave's issue is in the matrix pipe the other runs its FMA chains.  Measured alternatives,
            // one wave per SIMD owning the whole file with two or four result sets in flight: 334 ms against 236 ms for
            // 2048 tokens of the 7B -- the in-order wave stalls on every MFMA result, and above 256 registers the compiler
            // parks results in AGPRs (+128 v_accvgpr_read per block).
            typedef _Float16 h2v __attribute__((ext_vector_type(2)));
            const h2v bias = { (_Float16) 1032.0f, (_Float16) 1032.0f };
#pragma unroll
            for (int ii = 0; ii < 4; ii++) {
                const uint32_t xs = (ii < 2 ? x0 : x1) >> ((ii & 1) * 8);
                uint32_t p0, p1;                                // biased nibbles -> fp16 1024 + q (exact), then - 1032
                asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(p0) : "v"(xs), "v"(0x000F000Fu), "v"(0x64006400u));
                asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(p1) : "v"(xs >> 4), "v"(0x000F000Fu), "v"(0x64006400u));
                const h2v a01 = __builtin_bit_cast(h2v, p0) - bias, a23 = __builtin_bit_cast(h2v, p1) - bias;
                const h4v A = { a01.x, a01.y, a23.x, a23.y };
                const u32x4 Bq = ii < 2 ? B0 : B1;
                struct { uint32_t a, b; } bw = { (ii & 1) ? Bq.z : Bq.x, (ii & 1) ? Bq.w : Bq.y };
                const f32x32v D = __builtin_amdgcn_mfma_f32_32x32x4f16(A, __builtin_bit_cast(h4v, bw), zero32, 0, 0, 0);
                // The FMA chains are volatile asm so that they stay in this order with one result set live (written as plain
                // C++ the compiler sinks them below later MFMAs and spills 2 KB per lane).  The MFMA -> VALU read needs software
                // wait states which the compiler only inserts for instructions it can see: the first FMA of the issue is a
                // visible one, and its result is a (dummy) input of the first asm FMA, which orders every asm FMA after it.
                const float first = __builtin_fmaf(sc[0].x, D[0], acc[2 * ii][0].x);
#pragma unroll
                for (int hb = 0; hb < 2; hb++) {
                    const int k = 2 * ii + hb;                 // result registers 16 hb .. 16 hb + 15 = chain k (lane halves carried chains 2 ii, 2 ii + 1)
#pragma unroll
                    for (int r = 0; r < 8; r++) {
                        const float d0 = D[16 * hb + 2 * r], d1 = D[16 * hb + 2 * r + 1];
                        // (two plain FMAs issue faster than one packed one: tools/mfma_rate_probe.hip)
                        if (hb == 0 && r == 0) {
                            acc[k][r].x = first;
                            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[k][r].y) : "v"(sc[r].y), "v"(d1), "v"(first));
                        } else {
                            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[k][r].x) : "v"(sc[r].x), "v"(d0));
                            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[k][r].y) : "v"(sc[r].y), "v"(d1));
                        }
                    }
                }
            }
        }
        if (q + 1 < nq) stash(buf ^ 1);
        __syncthreads();
    }
    // ---- fold the 8 chains (ggml.c:872-887 tree) and store: lane = column, 16 rows (C/D layout)
    const int n = n0 + wc * 32 + (lane & 31);
    const int mb = (rp * 2 + wr) * 32 + 4 * (lane >> 5);
    if (n < ncols && rp * 2 + wr < nrb32) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = mb + (r & 3) + 8 * (r >> 2);
#define LH_A(K) ((r & 1) ? acc[K][r >> 1].y : acc[K][r >> 1].x)
            float v = ((LH_A(0) + LH_A(4)) + (LH_A(2) + LH_A(6))) + ((LH_A(1) + LH_A(5)) + (LH_A(3) + LH_A(7)));
#undef LH_A
            if (m < M) {
                if (EPI == EPI_RESID) v = v + resid[(size_t) n * resid_stride + m];
                y[(size_t) n * y_stride + m] = v;
            }
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
            s = tree32_to_lane0(s);
            if (l == 0) sc[t] = s * kq_scale;
        }
    }
    __syncthreads();

    // ---- soft_max over keys 0..tmax (masked keys are -inf -> 0)
    float mx = -INFINITY;
    for (int t = tid; t <= tmax; t += blockDim.x) mx = fmaxf(mx, sc[t]);
    mx = block_max_f(mx, red, 0);
    double sum = 0.0;
    for (int t = tid; t <= tmax; t += blockDim.x) {
        const float e = h2f_bits(T_exp[f2h_bits(sc[t] - mx)]);
        sc[t] = e;
        sum += (double) e;
    }
    sum = block_sum_d(sum, red, 1);
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
// Prompt attention for many query rows (N >= 32, head size 128): lane = QUERY ROW.
// k_attn above gives every (head, query) its own workgroup and re-reads the head's whole K and V for
// each query: 2048 rows stream 69 GB per layer through L2.  Here a wave owns 64 consecutive queries
// of one head and walks the keys; the key row (scores) / value row (V*P) is the same for all 64 lanes,
// so it is fetched with SCALAR loads and used as an SGPR operand of v_fma_f32 -- no LDS, no
// cross-lane reduction (each lane runs the reference's 32 FMA chains and its reduction tree itself),
// and K / V are read once per 64 queries.  Scores are materialised like the reference's KQ tensor
// (.mm:614), in a [head][key][query] workspace so that lanes read and write it coalesced; queries
// are processed in batches of NB rows to bound it.
//   k_attnq_scores   grid (NB/64, H, KS): KQ * scale for its key slice, running max      -> S, pmax
//   k_attnq_softmax  grid (NB/64, H) x (64 queries x 16 key phases): exp LUT, double sum  -> S = e, inv
//   k_attnq_pv       grid (NB/64, H, 4 column groups x nth): p = e * inv; one FMA chain per chunk of the
//                    reference's nth-way key split                                       -> part
//   k_attnq_merge    the ordered add of the nth partials                                 -> merged
// Arithmetic is identical to k_attn.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3)))
k_attnq_scores(const float *__restrict__ qr, const float *__restrict__ Kc, float *__restrict__ S, float *__restrict__ pmax,
               int n_past, int N, int nb0, int NB, int d, int T, float kq_scale, int KS) {
    // (one wave per workgroup: grouping four query blocks of a head into a workgroup so that they share
    // the K rows in the scalar cache measured 7x SLOWER)
    const int lane = threadIdx.x, h = blockIdx.y, ks = blockIdx.z;
    const int nl = blockIdx.x * 64 + lane, n = nb0 + nl;
    const bool valid = n < N;
    const int nq = valid ? n : N - 1;
    const int nb_end = min(nb0 + (int) (blockIdx.x + 1) * 64, N);
    const int Tb = n_past + nb_end;                       // keys any query of this block can see
    const int per = (Tb + KS - 1) / KS, t0 = ks * per, t1 = min(Tb, t0 + per);
    const int tq = n_past + nq;                           // last key this lane's query sees
    float q[128];
    {
        const f32x4 *qp = (const f32x4 *) (qr + (size_t) nq * d + h * 128);
#pragma unroll
        for (int i = 0; i < 32; i++) { const f32x4 v = qp[i]; q[4 * i] = v.x; q[4 * i + 1] = v.y; q[4 * i + 2] = v.z; q[4 * i + 3] = v.w; }
    }
    float mx = -INFINITY;
    // ggml_vec_dot_f32 (ggml.c:1223-1258): chain l (0..31) = elements l, l+32, l+64, l+96 by FMA from 0;
    // reduction tree (ggml.c:872-887) = lanes xor 8, 16, 4, 1, 2.  Chains 0..15 first, then 16..31.
    // The key row reaches the FMAs as SGPR operands, 16 floats (one s_load_dwordx16) per PIECE; a key is
    // 8 pieces, consumed in the order (half, j): piece pi covers elements 32 * (pi & 3) + 16 * (pi >> 2) + e.
    // A whole row (128 SGPRs) cannot be resident, so the pieces of consecutive keys form one stream that is
    // software-pipelined three pieces ahead through four 16-SGPR buffers (the scalar-cache round trip
    // per piece was this kernel's critical path).
    float kb[4][16];
#define LH_LOADP(BUF, TT, PI)                                                                      \
    {                                                                                              \
        const float *p_ = Kc + (size_t) min((TT), t1 - 1) * d + h * 128 + 32 * ((PI) & 3) + 16 * ((PI) >> 2);   /* wave-uniform */ \
        _Pragma("unroll") for (int e = 0; e < 16; e++) kb[BUF][e] = p_[e];                         \
    }
    if (t0 < t1) { LH_LOADP(0, t0, 0) LH_LOADP(1, t0, 1) LH_LOADP(2, t0, 2) }
    for (int t = t0; t < t1; t++) {
        float r1[2][8];
        float c[16];
#pragma unroll
        for (int pi = 0; pi < 8; pi++) {
            LH_LOADP((pi + 3) & 3, t + ((pi + 3) >> 3), (pi + 3) & 7)
            const int base = 32 * (pi & 3) + 16 * (pi >> 2);
#pragma unroll
            for (int e = 0; e < 16; e++) c[e] = fmaf(kb[pi & 3][e], q[base + e], (pi & 3) == 0 ? 0.0f : c[e]);
            if ((pi & 3) == 3) {
#pragma unroll
                for (int l = 0; l < 8; l++) r1[pi >> 2][l] = c[l] + c[l + 8];
            }
        }
        float u[8];
#pragma unroll
        for (int l = 0; l < 8; l++) u[l] = r1[0][l] + r1[1][l];
        const float v0 = u[0] + u[4], v1 = u[1] + u[5], v2 = u[2] + u[6], v3 = u[3] + u[7];
        const float sc = ((v0 + v1) + (v2 + v3)) * kq_scale;
        if (t <= tq) mx = fmaxf(mx, sc);
        S[((size_t) h * T + t) * NB + nl] = sc;
    }
#undef LH_LOADP
    pmax[((size_t) h * KS + ks) * NB + nl] = mx;
}

// The same scores on the fp32 matrix cores (round 2, after k_attnq_scores_lds).  v_mfma_f32_16x16x4_f32 is bit-for-bit the k-ordered
// fmaf chain fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, C)))) per output: with A = query elements {l, l + 32, l + 64, l + 96} and
// B = the same elements of a key it IS chain l of ggml_vec_dot_f32 (ggml.c:1223-1258) for a 16 x 16 tile of (query, key) pairs.  Lane
// (m = lane % 16, kk = lane / 16) supplies element l + 32 kk of row m: the 32 operands a lane needs for the 32 chains are the 32
// CONSECUTIVE floats [32 kk, 32 kk + 32) of its query / key row -- loaded straight into registers, no LDS, no scalar loads.  One wave =
// 16 queries (registers, loaded once) against its key slice, 16 keys per step = 32 independent MFMAs (C = 0) + the reduction tree
// (ggml.c:872-887) on the VALU, 31 additions per pair.  Result registers: lane holds key n = lane % 16, queries 4 kk + r.
// 153 (LDS variant) -> 100 us per launch at 2 048 tokens, logits bit-identical; requesting the next step's key rows a step ahead
// (+32 registers) measured 109 us: two waves per SIMD already cover the load.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))
k_attnq_scores_mfma(const float *__restrict__ qr, const float *__restrict__ Kc, float *__restrict__ S, float *__restrict__ pmax,
                    int n_past, int N, int nb0, int NB, int d, int T, float kq_scale, int KS) {
    const int lane = threadIdx.x, m = lane & 15, kk = lane >> 4, h = blockIdx.y, ks = blockIdx.z;
    const int nl0 = blockIdx.x * 16;
    const int Tb = n_past + min(nb0 + nl0 + 16, N);                    // keys any query of this tile can see
    const int per = (Tb + KS - 1) / KS, t0 = ks * per, t1 = min(Tb, t0 + per);
    float mx[4] = { -INFINITY, -INFINITY, -INFINITY, -INFINITY };
    if (t0 < t1) {
        float aq[32];
        {
            const int nq = min(nb0 + nl0 + m, N - 1);
            const f32x4 *qp = (const f32x4 *) (qr + (size_t) nq * d + h * 128 + 32 * kk);
#pragma unroll
            for (int j = 0; j < 8; j++) { const f32x4 v = qp[j]; aq[4 * j] = v.x; aq[4 * j + 1] = v.y; aq[4 * j + 2] = v.z; aq[4 * j + 3] = v.w; }
        }
        int tq[4];
#pragma unroll
        for (int r = 0; r < 4; r++) tq[r] = n_past + min(nb0 + nl0 + 4 * kk + r, N - 1);      // last key query 4 kk + r sees
        const f32x4v zero4 = { 0.0f, 0.0f, 0.0f, 0.0f };
        for (int tb = t0; tb < t1; tb += 16) {
            float bk[32];
            {
                const f32x4 *kp = (const f32x4 *) (Kc + (size_t) min(tb + m, t1 - 1) * d + h * 128 + 32 * kk);
#pragma unroll
                for (int j = 0; j < 8; j++) { const f32x4 v = kp[j]; bk[4 * j] = v.x; bk[4 * j + 1] = v.y; bk[4 * j + 2] = v.z; bk[4 * j + 3] = v.w; }
            }
            float r1[2][8][4];
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                f32x4v D[16];
#pragma unroll
                for (int l = 0; l < 16; l++) D[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[16 * hf + l], bk[16 * hf + l], zero4, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 8; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) r1[hf][j][r] = D[j][r] + D[j + 8][r];
            }
            f32x4 out;
            float scv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float u[8];
#pragma unroll
                for (int j = 0; j < 8; j++) u[j] = r1[0][j][r] + r1[1][j][r];
                const float v0 = u[0] + u[4], v1 = u[1] + u[5], v2 = u[2] + u[6], v3 = u[3] + u[7];
                scv[r] = ((v0 + v1) + (v2 + v3)) * kq_scale;
            }
            const int t = tb + m;                                       // this lane's key
            if (t < t1) {
#pragma unroll
                for (int r = 0; r < 4; r++) if (t <= tq[r]) mx[r] = fmaxf(mx[r], scv[r]);
                out.x = scv[0]; out.y = scv[1]; out.z = scv[2]; out.w = scv[3];
                *(f32x4 *) (S + ((size_t) h * T + t) * NB + nl0 + 4 * kk) = out;
            }
        }
    }
    // running maximum of each query over this key slice: across the 16 lanes (keys) of a DPP row
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float v = mx[r];
        v = fmaxf(v, dpp_f<DPP_QUAD_XOR1>(v));
        v = fmaxf(v, dpp_f<DPP_QUAD_XOR2>(v));
        v = fmaxf(v, dpp_f<DPP_ROW_HALF_MIRROR>(v));
        v = fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v));
        if (m == 0) pmax[((size_t) h * KS + ks) * NB + nl0 + 4 * kk + r] = v;
    }
}

// The same scores with the K rows staged in LDS (round 2): a workgroup = 4 waves = 256 consecutive queries of one head; a
// tile of 32 keys (16 KB) is fetched with coalesced vector loads (next tile in registers while this one is consumed,
// two LDS buffers, one barrier per tile) and every lane reads the key's elements as LDS BROADCASTS (wave-uniform
// address, ds_read_b128).  k_attnq_scores above brings the key row in through the scalar cache instead: its loop carries
// 134 s_mov + 170 v_mov per key for the SGPR buffer rotation, spills, and every piece waits on lgkmcnt(0) because
// scalar loads return out of order -- 201 us per launch at 2 048 tokens where the FMAs need ~60.  Measured here: 152 us
// (2 048-token eval 212.5 -> 202.7 ms): now LDS-bound -- a broadcast ds_read_b128 costs the LDS pipe as much as a spread one
// (8 clocks per wave), i.e. 2 clocks per key element and wave, the price of the FMA it feeds, and the LDS is shared by the
// CU's four SIMDs.  Arithmetic identical (same chains, same tree); a wave skips the tiles none of its 64 queries can see
// (k_attnq_softmax never reads them).
constexpr int AQ_TK = 32;
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_attnq_scores_lds(const float *__restrict__ qr, const float *__restrict__ Kc, float *__restrict__ S, float *__restrict__ pmax,
                   int n_past, int N, int nb0, int NB, int d, int T, float kq_scale, int KS) {
    __shared__ f32x4 sK[2][AQ_TK * 32];
    const int tid = threadIdx.x, wave = tid >> 6, h = blockIdx.y, ks = blockIdx.z;
    const int q0 = nb0 + (int) blockIdx.x * 256;
    const int nl = blockIdx.x * 256 + tid, n = nb0 + nl;
    const int nq = n < N ? n : N - 1;
    const int Tb = n_past + min(q0 + 256, N);                          // keys any query of this workgroup can see
    const int Tw = n_past + min(q0 + (wave + 1) * 64, N);               // ... of this wave (its 64-query block, as k_attnq_softmax counts)
    const int per = (Tb + KS - 1) / KS, t0 = ks * per, t1 = min(Tb, t0 + per);
    const int tq = n_past + nq;
    float mx = -INFINITY;
    if (t0 < t1) {
        float q[128];
        {
            const f32x4 *qp = (const f32x4 *) (qr + (size_t) nq * d + h * 128);
#pragma unroll
            for (int i = 0; i < 32; i++) { const f32x4 v = qp[i]; q[4 * i] = v.x; q[4 * i + 1] = v.y; q[4 * i + 2] = v.z; q[4 * i + 3] = v.w; }
        }
        f32x4 g[4];
#define LH_GLOAD(TBASE)                                                                                     \
        _Pragma("unroll") for (int u = 0; u < 4; u++) {                                                     \
            const int idx_ = tid + u * 256;                                                                 \
            g[u] = ((const f32x4 *) (Kc + (size_t) min((TBASE) + (idx_ >> 5), t1 - 1) * d + h * 128))[idx_ & 31]; \
        }
        LH_GLOAD(t0)
#pragma unroll
        for (int u = 0; u < 4; u++) sK[0][tid + u * 256] = g[u];
        __syncthreads();
        int buf = 0;
        for (int tbase = t0; tbase < t1; tbase += AQ_TK, buf ^= 1) {
            const bool more = tbase + AQ_TK < t1;
            if (more) LH_GLOAD(tbase + AQ_TK)
            const int kend = min(AQ_TK, min(t1, Tw) - tbase);              // (<= 0: nothing of this tile is visible to this wave)
            for (int kk = 0; kk < kend; kk++) {
                const int t = tbase + kk;
                const f32x4 *kr = &sK[buf][kk * 32];
                float r1[2][8];
                float c[16];
#pragma unroll
                for (int pi = 0; pi < 8; pi++) {
                    const int base = 32 * (pi & 3) + 16 * (pi >> 2);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const f32x4 kv = kr[base / 4 + j];                  // wave-uniform address: LDS broadcast
                        c[4 * j + 0] = fmaf(kv.x, q[base + 4 * j + 0], (pi & 3) == 0 ? 0.0f : c[4 * j + 0]);
                        c[4 * j + 1] = fmaf(kv.y, q[base + 4 * j + 1], (pi & 3) == 0 ? 0.0f : c[4 * j + 1]);
                        c[4 * j + 2] = fmaf(kv.z, q[base + 4 * j + 2], (pi & 3) == 0 ? 0.0f : c[4 * j + 2]);
                        c[4 * j + 3] = fmaf(kv.w, q[base + 4 * j + 3], (pi & 3) == 0 ? 0.0f : c[4 * j + 3]);
                    }
                    if ((pi & 3) == 3) {
#pragma unroll
                        for (int l = 0; l < 8; l++) r1[pi >> 2][l] = c[l] + c[l + 8];
                    }
                    asm volatile("" ::: "memory");                          // one or two pieces (16 key elements each) live at a time: left alone the scheduler
                }                                                           // hoists a whole key's 32 LDS reads and spills the query row
                float u[8];
#pragma unroll
                for (int l = 0; l < 8; l++) u[l] = r1[0][l] + r1[1][l];
                const float v0 = u[0] + u[4], v1 = u[1] + u[5], v2 = u[2] + u[6], v3 = u[3] + u[7];
                const float sc = ((v0 + v1) + (v2 + v3)) * kq_scale;
                if (t <= tq) mx = fmaxf(mx, sc);
                S[((size_t) h * T + t) * NB + nl] = sc;
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < 4; u++) sK[buf ^ 1][tid + u * 256] = g[u];
            }
            __syncthreads();
        }
#undef LH_GLOAD
    }
    pmax[((size_t) h * KS + ks) * NB + nl] = mx;
}

__global__ void __launch_bounds__(1024)
k_attnq_softmax(float *__restrict__ S, const float *__restrict__ pmax, float *__restrict__ inv,
                int n_past, int N, int nb0, int NB, int T, int KS, const uint16_t *__restrict__ T_exp) {
    constexpr int PH = 16;                                // key phases: a thread takes every 16th key of its query
    __shared__ double part[PH][64];
    const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6, h = blockIdx.y;
    const int nl = blockIdx.x * 64 + lane, n = nb0 + nl;
    const int nb_end = min(nb0 + (int) (blockIdx.x + 1) * 64, N);
    const int Tb = n_past + nb_end;
    const int tq = n_past + (n < N ? n : N - 1);
    float mx = -INFINITY;
    for (int k = 0; k < KS; k++) mx = fmaxf(mx, pmax[((size_t) h * KS + k) * NB + nl]);
    double sum = 0.0;
    for (int t = ph; t < Tb; t += PH) {
        float *sp = S + ((size_t) h * T + t) * NB + nl;
        float e = 0.0f;                                   // masked keys (-inf in the reference) contribute 0
        if (t <= tq) { e = h2f_bits(T_exp[f2h_bits(*sp - mx)]); sum += (double) e; }
        *sp = e;
    }
    part[ph][lane] = sum;
    __syncthreads();
    if (ph == 0) {
        // every term is a multiple of 2^-24 and <= 1: the double sum is exact in any order
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < PH; k++) tot += part[k][lane];
        inv[(size_t) h * NB + nl] = (float) (1.0 / tot);
    }
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4)))
k_attnq_pv(const float *__restrict__ S, const float *__restrict__ inv, const float *__restrict__ Vc, float *__restrict__ part,
           int n_past, int N, int nb0, int NB, int d, int T, int nth) {
    // one wave = 64 queries x 32 columns x ONE chunk of the reference's nth-way key split; the chunks'
    // partial sums are added in thread order by k_attnq_merge (ggml.c:5553-5577)
    const int lane = threadIdx.x, h = blockIdx.y, cg = blockIdx.z & 3, th = blockIdx.z >> 2, c0 = cg * 32;
    const int nl = blockIdx.x * 64 + lane;
    const int nb_end = min(nb0 + (int) (blockIdx.x + 1) * 64, N);
    const int Tb = n_past + nb_end;
    const float iv = inv[(size_t) h * NB + nl];
    const int dc = (T + nth - 1) / nth;
    const int t0 = dc * th;
    const int t1 = min(min(t0 + dc, T), Tb);              // beyond Tb every P of this block is 0: fma(v, 0, acc) == acc
    float acc[32];
#pragma unroll
    for (int c = 0; c < 32; c++) acc[c] = 0.0f;
    const float *sp = S + ((size_t) h * T + t0) * NB + nl;
    // Two keys per trip, the next trip's operands requested before this trip's FMAs (round 2): the loop used to be one
    // dependent round trip per key -- the lane's probability (vector load) and the value row (two s_load_dwordx16, which only
    // lgkmcnt(0) can wait for) were requested and awaited inside the same iteration.  The FMA order per column is unchanged.
    if (t0 < t1) {
        float pa = sp[0], pb = sp[(size_t) min(1, t1 - 1 - t0) * NB];
        float va[32], vb[32];
        {
            const float *r0 = Vc + (size_t) t0 * d + h * 128 + c0, *r1 = Vc + (size_t) min(t0 + 1, t1 - 1) * d + h * 128 + c0;      // wave-uniform: scalar loads
#pragma unroll
            for (int c = 0; c < 32; c++) { va[c] = r0[c]; vb[c] = r1[c]; }
        }
        for (int t = t0; t < t1; t += 2) {
            const float p0 = pa * iv, p1 = (t + 1 < t1) ? pb * iv : 0.0f;        // soft_max's final scale (ggml.c:7036-7041); a clamped re-read is weighted 0: fma(v, 0, acc) == acc
            float na[32], nb[32];
            const int ta = min(t + 2, t1 - 1), tb = min(t + 3, t1 - 1);
            const float npa = sp[(size_t) (ta - t0) * NB], npb = sp[(size_t) (tb - t0) * NB];
            {
                const float *r0 = Vc + (size_t) ta * d + h * 128 + c0, *r1 = Vc + (size_t) tb * d + h * 128 + c0;
#pragma unroll
                for (int c = 0; c < 32; c++) { na[c] = r0[c]; nb[c] = r1[c]; }
            }
#pragma unroll
            for (int c = 0; c < 32; c++) acc[c] = fmaf(va[c], p0, acc[c]);
#pragma unroll
            for (int c = 0; c < 32; c++) acc[c] = fmaf(vb[c], p1, acc[c]);
#pragma unroll
            for (int c = 0; c < 32; c++) { va[c] = na[c]; vb[c] = nb[c]; }
            pa = npa; pb = npb;
        }
    }
    // part[th][h][nl][128]
    f32x4 *o = (f32x4 *) (part + (((size_t) th * gridDim.y + h) * NB + nl) * 128 + c0);
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = f32x4{ acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3] };
}

// merged[n][h*128 + c] = part[0] + part[1] + ... in thread order.  grid (NB/2, H), 256 threads = 2 queries x 128 columns
// The same V*P partial sums on the matrix cores (round 2).  v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fp32 fmaf chain per
// output -- D = fma(a1, b1, fma(a0, b0, C)), one rounding per product, subnormals kept (cdna_hip_programming.md, "Numerics" of the
// f32 MFMAs) -- i.e. exactly acc = fma(v, p, acc) over two consecutive keys, which is what the chain of a (chunk, query, column)
// is.  One wave = 64 queries x the head's 128 columns x ONE chunk of the nth-way key split: 8 accumulator tiles (128 registers),
// per pair of keys 2 loads of probabilities (lane = query) and 4 of values (lane = column), 8 MFMAs = 512 cycles of the matrix
// pipe at the fp32 FMA peak, and no VALU work but the soft_max scale.  A chunk with an odd number of keys is padded with a zero
// pair at the FRONT: fma(0, 0, +0) = +0 leaves the chain's start unchanged (a trailing pad could turn a -0 sum into +0).
typedef float f32x16v __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3)))
k_attnq_pv_mfma(const float *__restrict__ S, const float *__restrict__ inv, const float *__restrict__ Vc, float *__restrict__ part,
                int n_past, int N, int nb0, int NB, int d, int T, int nth) {
    const int lane = threadIdx.x, i = lane & 31, kk = lane >> 5, h = blockIdx.y, th = blockIdx.z;
    const int q0 = blockIdx.x * 64;
    const int nb_end = min(nb0 + (int) (blockIdx.x + 1) * 64, N);
    const int Tb = n_past + nb_end;
    const int dc = (T + nth - 1) / nth;
    const int t0 = dc * th;
    const int t1 = min(min(t0 + dc, T), Tb);              // beyond Tb every P of this block is 0: fma(v, 0, acc) == acc
    f32x16v D[2][4];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) D[a][b][r] = 0.0f;
    const float iv0 = inv[(size_t) h * NB + q0 + i], iv1 = inv[(size_t) h * NB + q0 + 32 + i];
    const int nk = t1 - t0;
    // operands of PF pair-steps in flight: a step is 512 matrix-pipe cycles, a load round trip several times that
    constexpr int PF = 4;
    float pa0[PF], pa1[PF], pb0[PF], pb1[PF], pb2[PF], pb3[PF];
    const int tstart = t0 - (nk > 0 ? (nk & 1) : 0);
#define LH_PVLOAD(ST, TP)                                                                           \
    {                                                                                               \
        const int key_ = (TP) + kk;                                                                 \
        const bool real_ = key_ >= t0 && key_ < t1;      /* (the front pad, and steps past the end) */ \
        const int kc_ = min(max(key_, t0), max(t1 - 1, t0));                                        \
        const float *sp_ = S + ((size_t) h * T + kc_) * NB + q0 + i;                                \
        const float *vp_ = Vc + (size_t) kc_ * d + h * 128 + i;                                     \
        const float s0_ = sp_[0], s1_ = sp_[32], v0_ = vp_[0], v1_ = vp_[32], v2_ = vp_[64], v3_ = vp_[96]; \
        pa0[ST] = real_ ? s0_ * iv0 : 0.0f; pa1[ST] = real_ ? s1_ * iv1 : 0.0f;     /* soft_max's final scale (ggml.c:7036-7041) */ \
        pb0[ST] = real_ ? v0_ : 0.0f; pb1[ST] = real_ ? v1_ : 0.0f; pb2[ST] = real_ ? v2_ : 0.0f; pb3[ST] = real_ ? v3_ : 0.0f; \
    }
    if (nk > 0) {
#pragma unroll
        for (int st = 0; st < PF; st++) LH_PVLOAD(st, tstart + 2 * st)
        for (int tp = tstart; tp < t1; tp += 2 * PF) {
#pragma unroll
            for (int st = 0; st < PF; st++) {
                if (tp + 2 * st < t1) {
                    const float a0 = pa0[st], a1 = pa1[st], b0 = pb0[st], b1 = pb1[st], b2 = pb2[st], b3 = pb3[st];
                    LH_PVLOAD(st, tp + 2 * (st + PF))
                    D[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, D[0][0], 0, 0, 0);
                    D[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, D[0][1], 0, 0, 0);
                    D[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b2, D[0][2], 0, 0, 0);
                    D[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b3, D[0][3], 0, 0, 0);
                    D[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, D[1][0], 0, 0, 0);
                    D[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, D[1][1], 0, 0, 0);
                    D[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b2, D[1][2], 0, 0, 0);
                    D[1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b3, D[1][3], 0, 0, 0);
                }
            }
        }
    }
#undef LH_PVLOAD
    // part[th][h][nl][128]; D register r of tile (a, b): query q0 + 32 a + (r & 3) + 8 (r >> 2) + 4 kk, column 32 b + i
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int nl = q0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kk;
            float *o = part + (((size_t) th * gridDim.y + h) * NB + nl) * 128 + i;
#pragma unroll
            for (int b = 0; b < 4; b++) o[32 * b] = D[a][b][r];
        }
}

__global__ void __launch_bounds__(256)
k_attnq_merge(const float *__restrict__ part, float *__restrict__ merged, int N, int nb0, int NB, int d, int nth) {
    const int c = threadIdx.x & 127, nl = blockIdx.x * 2 + (threadIdx.x >> 7), h = blockIdx.y, H = gridDim.y;
    const int n = nb0 + nl;
    if (n >= N) return;
    float s = part[(((size_t) 0 * H + h) * NB + nl) * 128 + c];
    for (int th = 1; th < nth; th++) s += part[(((size_t) th * H + h) * NB + nl) * 128 + c];
    merged[(size_t) n * d + h * 128 + c] = s;
}

// ------------------------------------------------------------------------------------------------
// Decode attention (one query row), split so that every CU works and replayable from a hipGraph:
// the context position lives in device memory (st[0] = n_past), never in a kernel argument.
//
//   k_dec_scores  grid (H, ceil(n_ctx/32)): RoPE of q (every workgroup, 64 pairs), RoPE + append of
//                 the new K row and copy of the new V row (the workgroup whose key slice contains
//                 n_past), then KQ*scale for its 32 keys            -> sc[H][n_ctx]
//   k_dec_pv_blk  grid (H, dh/32), nth*32 threads: soft_max over the head's row (exact in any order, see
//                 k_attn), the nth partial V*P sums of the reference's nth-way key split, their
//                 addition in thread order (ggml.c:5553-5577) and the quantization of the head's dh
//                 outputs to Q4_0 activation blocks for the wo mat-vec -> QA (and fp32 merged row)
// Arithmetic is identical to k_attn / k_rope_kv; only the work distribution differs.
// ------------------------------------------------------------------------------------------------
constexpr int DEC_TS = 32;      // keys per workgroup: 8 half-waves x 4 keys

__global__ void __launch_bounds__(256)
k_dec_scores(const float *__restrict__ qkv, int d, int dh, const double *__restrict__ sincos_tab,
             float *__restrict__ Kc, float *__restrict__ Vc, float *__restrict__ sc, int n_ctx,
             float kq_scale, const int32_t *__restrict__ st) {
    extern __shared__ double smem_d[];
    float *qs = (float *) smem_d;          // roped q of this head
    float *kn = qs + dh;                   // roped new k of this head
    const int h = blockIdx.x;
    const int n_past = st[0];
    const int t0 = blockIdx.y * DEC_TS;
    if (t0 > n_past) return;
    const int tid = threadIdx.x;
    const bool owns_new = n_past < t0 + DEC_TS;          // this slice contains key n_past
    const double *tab = sincos_tab + (size_t) n_past * dh;
    const float *q = qkv + h * dh, *kk = qkv + d + h * dh, *vv = qkv + 2 * d + h * dh;
    if (tid < dh / 2) {
        const int e = 2 * tid;
        const double cs = tab[e], sn = tab[e + 1];
        const double x0 = (double) q[e], x1 = (double) q[e + 1];
        qs[e] = (float) (x0 * cs - x1 * sn);
        qs[e + 1] = (float) (x0 * sn + x1 * cs);
        if (owns_new) {
            const double k0 = (double) kk[e], k1 = (double) kk[e + 1];
            const float r0 = (float) (k0 * cs - k1 * sn), r1 = (float) (k0 * sn + k1 * cs);
            kn[e] = r0; kn[e + 1] = r1;
            Kc[(size_t) n_past * d + h * dh + e] = r0;
            Kc[(size_t) n_past * d + h * dh + e + 1] = r1;
            Vc[(size_t) n_past * d + h * dh + e] = vv[e];
            Vc[(size_t) n_past * d + h * dh + e + 1] = vv[e + 1];
        }
    }
    __syncthreads();
    // each half-wave owns DEC_TS/8 consecutive keys and keeps all their loads in flight at once
    const int hw = tid >> 5, l = tid & 31;
    constexpr int KPH = DEC_TS / 8;
    const int tb = t0 + hw * KPH;
    float kv[KPH][8];
#pragma unroll
    for (int u = 0; u < KPH; u++) {
        const int t = min(tb + u, n_past);
        const float *kr = Kc + (size_t) t * d + h * dh;
#pragma unroll
        for (int i = 0; i < 8; i++) kv[u][i] = (i * 32 < dh) ? kr[min(i * 32, dh - 32) + l] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < KPH; u++) {
        const int t = tb + u;
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (i * 32 < dh) {
                const float kval = (t == n_past) ? kn[i * 32 + l] : kv[u][i];   // own row from LDS: the global store above is not yet visible
                s = fmaf(kval, qs[i * 32 + l], s);
            }
        }
        s = tree32_to_lane0(s);
        if (l == 0 && t <= n_past) sc[(size_t) h * n_ctx + t] = s * kq_scale;
    }
}

// Short prompt chunks (2 <= N <= 16 rows, the reference's n_batch = 8 flow): the decode work distribution with
// one more grid dimension.  Row n (blockIdx.z) is the query at position n_past + n and sees keys
// 0 .. n_past + n; q is already rotated and K / V already appended by k_rope_kv.  Same 32 FMA chains
// and reduction tree as k_dec_scores / k_attn.   sc: [row][head][n_ctx]
__global__ void __launch_bounds__(256)
k_decn_scores(const float *__restrict__ qr, int d, int dh, const float *__restrict__ Kc, float *__restrict__ sc,
              int n_ctx, float kq_scale, int n_past) {
    const int h = blockIdx.x, n = blockIdx.z, H = gridDim.x;
    const int np = n_past + n;
    const int t0 = blockIdx.y * DEC_TS;
    if (t0 > np) return;
    const int tid = threadIdx.x, hw = tid >> 5, l = tid & 31;
    constexpr int KPH = DEC_TS / 8;
    const int tb = t0 + hw * KPH;
    const float *q = qr + (size_t) n * d + h * dh;
    float qv[8], kv[KPH][8];
#pragma unroll
    for (int i = 0; i < 8; i++) qv[i] = (i * 32 < dh) ? q[min(i * 32, dh - 32) + l] : 0.0f;
#pragma unroll
    for (int u = 0; u < KPH; u++) {
        const float *kr = Kc + (size_t) min(tb + u, np) * d + h * dh;
#pragma unroll
        for (int i = 0; i < 8; i++) kv[u][i] = (i * 32 < dh) ? kr[min(i * 32, dh - 32) + l] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < KPH; u++) {
        const int t = tb + u;
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (i * 32 < dh) s = fmaf(kv[u][i], qv[i], s);
        s = tree32_to_lane0(s);
        if (l == 0 && t <= np) sc[((size_t) n * H + h) * n_ctx + t] = s * kq_scale;
    }
}

// One workgroup per (head, 32-column block of the head): soft_max of the head's score row
// (recomputed by each of the head's dh/32 workgroups -- exact in any order), the nth partial V*P sums
// for its 32 columns (one sequential FMA chain per (chunk, column), all nth*32 chains in parallel),
// their addition in thread order, and the Q4_0 quantization of exactly one activation block.
// Splitting a head by columns needs no cross-workgroup hand-off: the ordered combine is per column.
// block = 32 * min(nth, 32) threads; dynamic LDS: [32 doubles][n_ctx p][nth*32 partials]
//   MULTI (short prompt chunks): blockIdx.z = row n of the chunk, position n_past0 + n (host value, `st` unused);
//         sc is [row][head][n_ctx], merged / QA are per row (strides d, qa_strideA dwords, qa_strideD floats) and
//         the last workgroup of a row zeroes the QA blocks that pad K up to a multiple of 256.
template <bool MULTI>
__global__ void __launch_bounds__(1024)
k_dec_pv_blk(const float *__restrict__ sc, const float *__restrict__ Vc, int d, int dh, int n_ctx, int nth,
             float *__restrict__ merged, uint32_t *__restrict__ qa_A, float *__restrict__ qa_d,
             const uint16_t *__restrict__ T_exp, const int32_t *__restrict__ st,
             int n_past0, long qa_strideA, long qa_strideD, int lut_math) {
    extern __shared__ double smem_d[];
    double *red = smem_d;
    float *p = (float *) (smem_d + 32);
    float *part = p + n_ctx;
    const int h = blockIdx.x, cb = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
    const int n_past = MULTI ? n_past0 + (int) blockIdx.z : st[0];
    const int T = n_past + 1;
    const float *row = sc + (size_t) h * n_ctx;
    if (MULTI) {
        const int n = blockIdx.z;
        row = sc + ((size_t) n * gridDim.x + h) * n_ctx;
        if (merged) merged += (size_t) n * d;
        qa_A += (size_t) n * qa_strideA;
        qa_d += (size_t) n * qa_strideD;
        if (h == (int) gridDim.x - 1 && cb == (int) gridDim.y - 1)
            for (int pb = d / 32 + tid; pb < (int) qa_strideD; pb += nt) {
                const int pc = pb >> 3, pj = pb & 7;
#pragma unroll
                for (int kk = 0; kk < 8; kk++) qa_A[(pc * 8 + kk) * 8 + pj] = 0;
                qa_d[pb] = 0.0f;
            }
    }
    float mx = -INFINITY;
    for (int t = tid; t < T; t += nt) { const float v = row[t]; p[t] = v; mx = fmaxf(mx, v); }
    mx = block_max_f(mx, red, 0);
    double sum = 0.0;
    for (int t = tid; t < T; t += nt) {
        const uint16_t xh = f2h_bits(p[t] - mx);
        const float e = h2f_bits((lut_math & 2) ? exp_math_bits(xh) : T_exp[xh]);
        p[t] = e;
        sum += (double) e;
    }
    sum = block_sum_d(sum, red, 1);
    const float inv = (float) (1.0 / sum);
    for (int t = tid; t < T; t += nt) p[t] *= inv;
    // a chunk row is as long as the whole chunk's context (ggml.c:5459-5480 splits n_past + N keys over the
    // threads for every row); the masked tail has weight exp(-inf) = 0 and is walked like the reference does
    const int Tpv = MULTI ? n_past0 + (int) gridDim.z : T;
    if (MULTI)
        for (int t = T + tid; t < Tpv; t += nt) p[t] = 0.0f;
    __syncthreads();

    const int c = tid & 31, sub = tid >> 5, nsub = nt >> 5;
    const int dc = (Tpv + nth - 1) / nth;
    const int col = h * dh + cb * 32 + c;
    const float *vcol = Vc + col;
    for (int th = sub; th < nth; th += nsub) {
        const int t0 = dc * th, t1 = min(t0 + dc, Tpv);
        // The chain is sequential but its loads are not: two register batches of 16 rows, the next one
        // in flight while the current one is consumed (a chain walks T/nth rows 16 KB apart; at a
        // 2 000-token context the single-batch loop spent one memory round trip per 16 rows).
        // Rows past the chunk end are clamped re-reads weighted by 0: fma(v, 0, acc) == acc.
        float acc = 0.0f;
        float va[16], vb[16];
#pragma unroll
        for (int u = 0; u < 16; u++) va[u] = vcol[(size_t) min(t0 + u, t1 - 1) * d];
        for (int tb = t0; tb < t1; tb += 32) {
#pragma unroll
            for (int u = 0; u < 16; u++) vb[u] = vcol[(size_t) min(tb + 16 + u, t1 - 1) * d];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const float pe = (tb + u < t1) ? p[min(tb + u, Tpv - 1)] : 0.0f;
                acc = fmaf(va[u], pe, acc);
            }
#pragma unroll
            for (int u = 0; u < 16; u++) va[u] = vcol[(size_t) min(tb + 32 + u, t1 - 1) * d];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const float pe = (tb + 16 + u < t1) ? p[min(tb + 16 + u, Tpv - 1)] : 0.0f;
                acc = fmaf(vb[u], pe, acc);
            }
        }
        part[th * 32 + c] = acc;
    }
    __syncthreads();
    if (tid < 32) {
        float s = part[tid];
        for (int th = 1; th < nth; th++) s += part[th * 32 + tid];          // thread order (ggml.c:5553-5577)
        if (merged) merged[col] = s;
        // quantize this 32-element block (ggml.c:456-523), one element per lane
        float amax = fabsf(s);
        amax = max_lanes_0_31(amax);
        const float dd = amax / 7.0f;
        const float id = (amax != 0.0f) ? 7.0f / amax : 0.0f;
        const uint32_t nib = (uint32_t) ((int) __builtin_rintf(s * id)) & 0xF;
        const int kk = tid & 7;
        const uint32_t e0 = __shfl(nib, 2 * kk), e1 = __shfl(nib, 2 * kk + 1);
        const uint32_t e2 = __shfl(nib, 16 + 2 * kk), e3 = __shfl(nib, 17 + 2 * kk);
        const int b = h * (dh / 32) + cb, cc = b >> 3, j = b & 7;
        if (tid < 8) qa_A[(cc * 8 + kk) * 8 + j] = (e0 | (e1 << 8) | (e2 << 16) | (e3 << 24)) << (4 * (j & 1));
        if (tid == 0) qa_d[b] = dd;
    }
}

// ------------------------------------------------------------------------------------------------
// k_dec_attn_x: k_dec_scores + k_dec_pv_blk<false> in ONE launch with a hand-off that stays inside one XCD.
// The scores -> soft_max . V seam is per head, and the workgroups of head h (grid (H, n_ctx / 32), linear id
// h + H * y, H a multiple of 8) all sit on XCD h % 8 (round-robin dispatch, checked by the load-time self-test
// k_xcd_selftest and by tools/xcd_barrier_probe.hip), i.e. behind ONE L2.  So the hand-off needs no device-scope
// traffic (that is what makes a cross-XCD hand-off cost 6-12 us): the score workgroups' stores are acknowledged by
// that L2 (s_waitcnt vmcnt(0)), their arrival is an atomic add WITHOUT scope bits (executes in the L2), the waiting
// workgroups poll it and then read the scores with sc1 loads (bypass the per-CU L1, hit the L2): ~1 us for the
// round trip against 2.3-2.7 us for a kernel boundary plus the second kernel's ramp (profiles/r02_g_xcd_barrier.txt).
//   grid (H, dh / 32 + n_ctx / 32).  Workgroup (h, y), y < dh / 32: soft_max . V for columns [32 y, 32 y + 32): requests
//                     the first V rows of its chains, waits for the head's n_past / 32 + 1 arrivals, then the
//                     k_dec_pv_blk body; the last of them to finish clears the head's two counters.
//                     y >= dh / 32: the k_dec_scores body for keys [32 (y - dh / 32), +32) if that slice starts at or
//                     before n_past, then arrive.
// The waiting workgroups have the LOWEST linear ids of the grid (dispatched first) and wait only for workgroups
// that need no resources they hold (4 H waiters of 256 threads against a chip that holds 2 048 such workgroups), so
// the spin always ends; it is bounded anyway and a time-out raises the sticky fault word in pinned host memory
// (results of that launch are then invalid; the host reports PredictionFailed after the next synchronisation).
// Arithmetic identical to the two kernels.  256 threads (nth <= 8); dynamic LDS as k_dec_pv_blk.
//   sync: [H][32] dwords (arrivals, finished waiters, padding to one 128-byte line per head)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float load_f32_sc1(const float *p) {
    return __builtin_bit_cast(float, __hip_atomic_load((const uint32_t *) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

struct AttnXArgs {
    const float *qkv; int d, dh; const double *sincos_tab; float *Kc, *Vc, *sc; int n_ctx, nth; float kq_scale;
    float *merged; uint32_t *qa_A; float *qa_d; const uint16_t *T_exp; const int32_t *st; uint32_t *sync, *fault; int lut_math;
    // k_qkv_attn only: data-tagged hand-offs.  qkv2[3 d] / sc2[H][n_ctx] hold {fp32 bits, tag} 8-byte granules,
    // tag = make_tag(epoch[0], layer + 1): a reader polls the granule itself until the tag is this launch's
    const uint64_t *qkv2; uint64_t *sc2; const uint32_t *epoch; int layer;
};
// role of workgroup (h, yy): yy < ncb: soft_max . V for column block yy; else scores for key slice yy - ncb.
// QKV_WAIT (k_qkv_attn): the head's q / k / v rows come from mat-vec workgroups of the SAME launch as tagged granules
// (qkv2); the K rows of the slice are requested first, then the rotating threads poll their own q / k / v granules.  The
// scores go to the soft_max . V workgroups as tagged granules too (sc2): no counters, no store-acknowledge wait, no
// separate poll -- a hand-off is one store and one load that sees it.
template <bool QKV_WAIT>
__device__ __forceinline__ void attn_x_body(const AttnXArgs &aa, const int h, const int yy, double *smem_d) {
    const float *__restrict__ qkv = aa.qkv; const int d = aa.d, dh = aa.dh; const double *__restrict__ sincos_tab = aa.sincos_tab;
    float *__restrict__ Kc = aa.Kc, *__restrict__ Vc = aa.Vc; float *sc = aa.sc; const int n_ctx = aa.n_ctx, nth = aa.nth; const float kq_scale = aa.kq_scale;
    float *__restrict__ merged = aa.merged; uint32_t *__restrict__ qa_A = aa.qa_A; float *__restrict__ qa_d = aa.qa_d;
    const uint16_t *__restrict__ T_exp = aa.T_exp; const int32_t *__restrict__ st = aa.st; uint32_t *sync = aa.sync, *fault = aa.fault; const int lut_math = aa.lut_math;
    const int ncb = dh / 32, tid = threadIdx.x;
#if LH_PHASE_PROBE == 3        /* timeline probe (tools/attn_timeline.py): kind 0xA0 = score workgroup, 0xA1 = soft_max . V workgroup */
    unsigned long long probe_t[5] = { 0, 0, 0, 0, 0 };
    const unsigned long long probe_wall = wall_clock64();
#define LH_ASTAMP(IDX) do { probe_t[IDX] = __builtin_readcyclecounter(); } while (0)
#define LH_AFLUSH(KIND) do { if (g_phase_probe && tid == 0) { unsigned long long *pb = g_phase_probe; const unsigned long long slot = atomicAdd(pb, 1ull); \
        if (slot < pb[1]) { unsigned long long *e = pb + 8 * (1 + slot); for (int i = 0; i < 5; i++) e[i] = probe_t[i]; \
            e[5] = ((unsigned long long) (KIND) << 48) | ((unsigned long long) yy << 32) | (unsigned) h; e[6] = wall_clock64(); e[7] = probe_wall; } } } while (0)
#else
#define LH_ASTAMP(IDX) do { } while (0)
#define LH_AFLUSH(KIND) do { } while (0)
#endif
    LH_ASTAMP(0);
    const int n_past = st[0];
    uint32_t *cnt = sync + h * 32;
    if (yy >= ncb) {
        // ---- score workgroup: keys [t0, t0 + 32)
        const int t0 = (yy - ncb) * DEC_TS;
        if (t0 > n_past) return;
        float *qs = (float *) smem_d, *kn = qs + dh;
        const bool owns_new = n_past < t0 + DEC_TS;
        const double *tab = sincos_tab + (size_t) n_past * dh;
        const float *q = qkv + h * dh, *kk = qkv + d + h * dh, *vv = qkv + 2 * d + h * dh;
        const int hw = tid >> 5, l = tid & 31;
        constexpr int KPH = DEC_TS / 8;
        const int tb = t0 + hw * KPH;
        float kv[KPH][8];
        auto load_keys = [&]() {
#pragma unroll
            for (int u = 0; u < KPH; u++) {
                const int t = min(tb + u, n_past);      // (row n_past itself comes from LDS below: whatever this returns for it is not used)
                const float *kr = Kc + (size_t) t * d + h * dh;
#pragma unroll
                for (int i = 0; i < 8; i++) kv[u][i] = (i * 32 < dh) ? kr[min(i * 32, dh - 32) + l] : 0.0f;
            }
        };
        const int nowait = ((lut_math & 0x400) ? 1 : 0) | ((lut_math >> 8) & 2) | ((lut_math & 0x1000) ? 4 : 0);     // (measurement-only switches: 0x400 this hop does not wait, results invalid; 0x200 polls without sleep; 0x1000 fault-injection test)
        const uint32_t tag = QKV_WAIT ? (make_tag(aa.epoch[0], aa.layer + 1)) : 0u;
        if (QKV_WAIT) load_keys();                              // in flight while the mat-vec workgroups finish
        if (tid < dh / 2) {
            const int e = 2 * tid;
            const double cs = tab[e], sn = tab[e + 1];
            const uint64_t *q2 = aa.qkv2 + h * dh, *k2 = q2 + d, *v2 = q2 + 2 * d;
            const double x0 = (double) (QKV_WAIT ? poll_tagged(q2 + e, tag, fault, nowait) : q[e]), x1 = (double) (QKV_WAIT ? poll_tagged(q2 + e + 1, tag, fault, nowait) : q[e + 1]);
            qs[e] = (float) (x0 * cs - x1 * sn);
            qs[e + 1] = (float) (x0 * sn + x1 * cs);
            if (owns_new) {
                const double k0 = (double) (QKV_WAIT ? poll_tagged(k2 + e, tag, fault, nowait) : kk[e]), k1 = (double) (QKV_WAIT ? poll_tagged(k2 + e + 1, tag, fault, nowait) : kk[e + 1]);
                const float r0 = (float) (k0 * cs - k1 * sn), r1 = (float) (k0 * sn + k1 * cs);
                kn[e] = r0; kn[e + 1] = r1;
                Kc[(size_t) n_past * d + h * dh + e] = r0;
                Kc[(size_t) n_past * d + h * dh + e + 1] = r1;
                Vc[(size_t) n_past * d + h * dh + e] = QKV_WAIT ? poll_tagged(v2 + e, tag, fault, nowait) : vv[e];
                Vc[(size_t) n_past * d + h * dh + e + 1] = QKV_WAIT ? poll_tagged(v2 + e + 1, tag, fault, nowait) : vv[e + 1];
                // the new V row must be in the L2 before any score of this workgroup is (a soft_max . V workgroup reads it once it
                // has seen the tagged scores): drain these stores on this side of the barrier
                if (QKV_WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
        LH_ASTAMP(1);
        if (!QKV_WAIT) load_keys();
#pragma unroll
        for (int u = 0; u < KPH; u++) {
            const int t = tb + u;
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (i * 32 < dh) {
                    const float kval = (t == n_past) ? kn[i * 32 + l] : kv[u][i];
                    s = fmaf(kval, qs[i * 32 + l], s);
                }
            }
            s = tree32_to_lane0(s);
            if (l == 0 && t <= n_past) {
                if (QKV_WAIT) store_tagged(aa.sc2 + (size_t) h * n_ctx + t, s * kq_scale, tag);
                else sc[(size_t) h * n_ctx + t] = s * kq_scale;
            }
        }
        if (QKV_WAIT) { LH_ASTAMP(2); LH_ASTAMP(3); LH_ASTAMP(4); LH_AFLUSH(0xA0); return; }
        // publish: every wave's stores (scores; the new K / V rows) are acknowledged by the L2, then one arrival
        LH_ASTAMP(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        LH_ASTAMP(3);
        if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#if LH_PHASE_PROBE == 3
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        LH_ASTAMP(4);
        LH_AFLUSH(0xA0);
        return;
    }
    // ---- soft_max . V workgroup for columns [32 cb, 32 cb + 32): k_dec_pv_blk<false> with 256 threads.  The first 32
    // rows of every V*P chain do not depend on the scores: they are requested BEFORE the wait and arrive while the score
    // workgroups run (the row with the new token's V is the last one of the last chain: never among them unless the
    // context is shorter than the batch, in which case the batch is fetched after the wait instead).
    const int cb = yy, nt = 256;
    const int T = n_past + 1;
    const int c = tid & 31, sub = tid >> 5, nsub = nt >> 5;
    const int dc = (T + nth - 1) / nth;
    const int col = h * dh + cb * 32 + c;
    const float *vcol = Vc + col;
    const int ta0 = dc * sub, t10 = min(ta0 + dc, T);
    // rows [ta0, ta0 + 32) of chain `sub` are old rows (< n_past) iff ta0 + 32 <= n_past or they are clamped below t10 - 1 < n_past
    // VB rows per register batch.  (Measured: 20 -- the most that keeps k_qkv_attn at 128 registers -- changes nothing at
    // contexts 288 and 400; 24 and 32 drop the launch to 3 and 2 waves per SIMD and the score workgroups lose their slots.)
    constexpr int VB = 16;
    const bool early = sub < nth && ta0 < t10 && min(ta0 + 2 * VB - 1, t10 - 1) < n_past;      // wave-uniform per 32-lane half; both halves of a wave differ only in `sub`
    float va[VB], vb[VB];
    if (early) {
#pragma unroll
        for (int u = 0; u < VB; u++) va[u] = vcol[(size_t) min(ta0 + u, t10 - 1) * d];
#pragma unroll
        for (int u = 0; u < VB; u++) vb[u] = vcol[(size_t) min(ta0 + VB + u, t10 - 1) * d];
    }
    if (!QKV_WAIT) {
        if (tid == 0) {
            const uint32_t need = (uint32_t) (n_past / DEC_TS + 1);
            int spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 20)) { __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            }
        }
        __syncthreads();
    }
    LH_ASTAMP(1);
    double *red = smem_d;
    float *p = (float *) (smem_d + 32);
    float *part = p + n_ctx;
    const float *row = sc + (size_t) h * n_ctx;
    float mx = -INFINITY;
    if (QKV_WAIT) {
        const uint32_t tag = make_tag(aa.epoch[0], aa.layer + 1);
        const int nowait = ((lut_math & 0x800) ? 1 : 0) | ((lut_math >> 8) & 2) | ((lut_math & 0x1000) ? 4 : 0);
        for (int t = tid; t < T; t += nt) { const float v = poll_tagged(aa.sc2 + (size_t) h * n_ctx + t, tag, fault, nowait); p[t] = v; mx = fmaxf(mx, v); }
    } else
    for (int t = tid; t < T; t += nt) { const float v = load_f32_sc1(row + t); p[t] = v; mx = fmaxf(mx, v); }
    mx = block_max_f(mx, red, 0);
    double sum = 0.0;
    for (int t = tid; t < T; t += nt) {
        const uint16_t xh = f2h_bits(p[t] - mx);
        const float e = h2f_bits((lut_math & 2) ? exp_math_bits(xh) : T_exp[xh]);
        p[t] = e;
        sum += (double) e;
    }
    sum = block_sum_d(sum, red, 1);
    const float inv = (float) (1.0 / sum);
    for (int t = tid; t < T; t += nt) p[t] *= inv;
    __syncthreads();
    LH_ASTAMP(2);
    for (int th = sub; th < nth; th += nsub) {
        const int ta = dc * th, t1 = min(ta + dc, T);
        float acc = 0.0f;
        if (!(early && th == sub)) {
#pragma unroll
            for (int u = 0; u < VB; u++) va[u] = vcol[(size_t) min(ta + u, t1 - 1) * d];
#pragma unroll
            for (int u = 0; u < VB; u++) vb[u] = vcol[(size_t) min(ta + VB + u, t1 - 1) * d];
        }
        for (int tb = ta; tb < t1; tb += 2 * VB) {
#pragma unroll
            for (int u = 0; u < VB; u++) {
                const float pe = (tb + u < t1) ? p[min(tb + u, T - 1)] : 0.0f;
                acc = fmaf(va[u], pe, acc);
            }
#pragma unroll
            for (int u = 0; u < VB; u++) va[u] = vcol[(size_t) min(tb + 2 * VB + u, t1 - 1) * d];
#pragma unroll
            for (int u = 0; u < VB; u++) {
                const float pe = (tb + VB + u < t1) ? p[min(tb + VB + u, T - 1)] : 0.0f;
                acc = fmaf(vb[u], pe, acc);
            }
#pragma unroll
            for (int u = 0; u < VB; u++) vb[u] = vcol[(size_t) min(tb + 3 * VB + u, t1 - 1) * d];
        }
        part[th * 32 + c] = acc;
    }
    __syncthreads();
    LH_ASTAMP(3);
    if (tid < 32) {
        float s = part[tid];
        for (int th = 1; th < nth; th++) s += part[th * 32 + tid];          // thread order (ggml.c:5553-5577)
        if (merged) merged[col] = s;
        float amax = fabsf(s);
        amax = max_lanes_0_31(amax);
        const float dd = amax / 7.0f;
        const float id = (amax != 0.0f) ? 7.0f / amax : 0.0f;
        const uint32_t nib = (uint32_t) ((int) __builtin_rintf(s * id)) & 0xF;
        const int kk = tid & 7;
        const uint32_t e0 = __shfl(nib, 2 * kk), e1 = __shfl(nib, 2 * kk + 1);
        const uint32_t e2 = __shfl(nib, 16 + 2 * kk), e3 = __shfl(nib, 17 + 2 * kk);
        const int b = h * (dh / 32) + cb, cc = b >> 3, j = b & 7;
        const uint32_t dwq = (e0 | (e1 << 8) | (e2 << 16) | (e3 << 24)) << (4 * (j & 1));
        if (tid < 8) qa_A[(cc * 8 + kk) * 8 + j] = dwq;
        if (tid == 0) qa_d[b] = dd;
    }
    // the last soft_max . V workgroup of the head to get here clears the counters for the next launch (every one of them
    // has passed the poll, every score workgroup has arrived: nobody touches them again in this launch)
    if (!QKV_WAIT && tid == 0) {
        const uint32_t done = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (done == (uint32_t) (ncb - 1)) {
            __hip_atomic_exchange(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_exchange(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    LH_ASTAMP(4);
    LH_AFLUSH(0xA1);
}
#undef LH_ASTAMP
#undef LH_AFLUSH

__global__ void __launch_bounds__(256)
k_dec_attn_x(const AttnXArgs aa) {
    extern __shared__ double smem_d[];
    attn_x_body<false>(aa, blockIdx.x, blockIdx.y, smem_d);
}

// wq|wk|wv mat-vec AND the attention in one launch.  The seam is per head as well: head h's scores need only head h's
// 3 dh output rows.  Blocks [0, gridA) are the mat-vec's workgroups (4 waves = 32 rows), PERMUTED so that the 3 dh / 32
// workgroups that own head h's q, k and v rows sit on XCD h % 8 (block b: XCD b % 8, slot b / 8 -> (head of that XCD, part)):
// they store their rows as tagged 8-byte granules {value, make_tag(epoch, layer + 1)} (EPI_STORE_TAG) which the readers poll.  Blocks [gridA, ...) are the attention workgroups
// of k_dec_attn_x in the same order (soft_max . V, then scores; gridA is a multiple of 8, so head h's stay on XCD h % 8); they
// request their V / K rows first, then wait.  The mat-vec workgroups never wait and are dispatched first; the 4 H
// soft_max . V workgroups are the only ones that wait for HIGHER block indices, and they cannot fill the chip.
template <int PRE, int D, int PG>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PG == 1 ? 4 : 3)))
k_qkv_attn(const GemvArgs ga, const AttnXArgs aa, const int gridA, const int H) {
    extern __shared__ double smem_d[];
    const int b = blockIdx.x;
    if (b < gridA) {
        const int ncb = aa.dh / 32, wph = 3 * ncb;
        const int xcd = b & 7, slot = b >> 3, j = slot / wph, part = slot % wph, mat = part / ncb, sub = part % ncb;
        const int h = xcd + 8 * j;
        gemv_body<PRE, EPI_STORE_TAG, D, true, PG>(ga, mat * (aa.d / 32) + h * ncb + sub, 4, smem_d);    // y = tagged granules
        return;
    }
    const int a = b - gridA, h = a % H, y = a / H;            // y < dh / 32: soft_max . V (their V prefetch starts with the mat-vec), then the score slices
    attn_x_body<true>(aa, h, y, smem_d);
}

// load-time self-test of the assumption above: out[b] = XCC_ID of workgroup b of a (H, Y) grid
__global__ void k_xcd_selftest(uint32_t *out) {
    if (threadIdx.x == 0) {
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        out[blockIdx.x + gridDim.x * blockIdx.y] = id & 0xf;
    }
}

// ------------------------------------------------------------------------------------------------
// greedy argmax, lowest index on ties (harness definition of temperature 0)
// ------------------------------------------------------------------------------------------------
// st (optional): st[0] = n_past, st[1] = decode step index -- both advanced here so a captured
// decode graph can be replayed without touching kernel arguments; out[st[1]] receives the token.
// One workgroup of 1024 threads: 32 loads in flight per thread (a 32 000-entry row is one round trip, not
// four), then the (value, index) pair is reduced inside each wave with DPP exchanges + readlane and across
// the 16 waves through LDS with a single barrier (a 10-level LDS tree with a barrier per level before).
__device__ __forceinline__ void argmax_take(float &v, int &i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
template <int CTRL>
__device__ __forceinline__ void argmax_dpp(float &v, int &i) {
    const float ov = dpp_f<CTRL>(v);
    const int oi = __builtin_amdgcn_mov_dpp(i, CTRL, 0xF, 0xF, true);
    argmax_take(v, i, ov, oi);
}
__global__ void __launch_bounds__(1024)
k_argmax(const float *__restrict__ logits, int V, int32_t *__restrict__ out, int out_idx,
         int32_t *__restrict__ next_token, int32_t *__restrict__ st, uint64_t *token_mb) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const int tid = threadIdx.x, nt = blockDim.x;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i0 = tid; i0 < V; i0 += 32 * nt) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; u++) v[u] = logits[min(i0 + u * nt, V - 1)];
#pragma unroll
        for (int u = 0; u < 32; u++) {
            const int i = i0 + u * nt;
            if (i < V) argmax_take(best, idx, v[u], i);       // ascending i: a tie keeps the lower index
        }
    }
    argmax_dpp<DPP_QUAD_XOR1>(best, idx);
    argmax_dpp<DPP_QUAD_XOR2>(best, idx);
    argmax_dpp<DPP_ROW_HALF_MIRROR>(best, idx);
    argmax_dpp<DPP_ROW_MIRROR>(best, idx);                    // every lane of a 16-lane row holds the row's pick
    {
        const int vb = __builtin_bit_cast(int, best);
        float wv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vb, 0));
        int wi = __builtin_amdgcn_readlane(idx, 0);
        argmax_take(wv, wi, __builtin_bit_cast(float, __builtin_amdgcn_readlane(vb, 16)), __builtin_amdgcn_readlane(idx, 16));
        argmax_take(wv, wi, __builtin_bit_cast(float, __builtin_amdgcn_readlane(vb, 32)), __builtin_amdgcn_readlane(idx, 32));
        argmax_take(wv, wi, __builtin_bit_cast(float, __builtin_amdgcn_readlane(vb, 48)), __builtin_amdgcn_readlane(idx, 48));
        if ((tid & 63) == 0) { bv[tid >> 6] = wv; bi[tid >> 6] = wi; }
    }
    __syncthreads();
    if (tid == 0) {
        float v = bv[0];
        int i = bi[0];
        for (int w = 1; w < (nt >> 6); w++) argmax_take(v, i, bv[w], bi[w]);
        const int r = i == 0x7fffffff ? 0 : i;
        out[st ? st[1] : out_idx] = r;
        if (next_token) *next_token = r;
        // (pipeline mailbox: the pick is the token of the NEXT position -- tagged with it -- stored into the first stage's memory)
        if (token_mb && st) store_tagged_sys(token_mb, (uint32_t) r, make_tag((uint32_t) st[0] + 2u, 0));
        if (st) { st[0] += 1; st[1] += 1; }
    }
}

// ------------------------------------------------------------------------------------------------
// Sampler front end on the device: llama_sample_top_p_top_k's candidate scores and its top-k selection
// (utils.cpp:345-395), so that a sampled decode step returns k (score, id) pairs instead of n_vocab logits.
//   score_i = logit_i * (1 / temp) [* or / repeat_penalty for ids in the last-n window]      in double, as the host does
//   the k largest, sorted descending (std::partial_sort with a.first > b.first)
// std::partial_sort is not stable: where two scores are EQUAL the reference's order (and which of two equal
// scores at the k-th place survives) is whatever libstdc++'s heap does with the whole 32 000-entry sequence.
// That cannot be reproduced from a candidate set, so the kernel reports `exact` = 0 whenever an equality could
// matter (a tie among the k + at the boundary, or a NaN) and the caller falls back to the host path on the full
// logits; with exact = 1 the k pairs are unambiguous and identical to the reference's cand[0..k).
// One workgroup of 1024 threads, <= 32 values per thread (n_vocab <= 32768), order-preserving 64-bit keys:
//   1. the maximum of every group of 16 threads (512 values, DPP row reduction): 64 group maxima.  Their minimum T is a
//      LOWER bound of the k-th largest value overall for any k <= 64 (64 values >= T exist), so each of the k best is
//      >= T -- and only a few hundred other values are;
//   2. the values >= T are collected (at most 768, else `exact` = 0) and ranked against each other; the k + 1 best
//      decide the answer and whether an equality is in play.
// (Measured and dropped: radix select -- its top-byte histogram is 32 000 atomics on a handful of LDS words, 48 us;
//  ranking 1024 per-thread maxima against each other -- a million LDS reads, 55 us.)
// flags[0] = exact, flags[1] = number of values collected.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long dpp_max_u64(unsigned long long v, unsigned long long o) { return o > v ? o : v; }
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
    const int lo = __builtin_amdgcn_mov_dpp((int) (uint32_t) v, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int) (uint32_t) (v >> 32), CTRL, 0xF, 0xF, true);
    return ((unsigned long long) (uint32_t) hi << 32) | (uint32_t) lo;
}
__global__ void __launch_bounds__(1024)
k_topk_candidates(const float *__restrict__ logits, int V, const int32_t *__restrict__ window, int n_window,
                  double scale, double repeat_penalty, int k,
                  double *__restrict__ out_score, int32_t *__restrict__ out_id, int32_t *__restrict__ flags) {
    constexpr int NPT = 32, LCAP = 768;
    __shared__ uint32_t seen[1024];                    // bitmap of the last-n window (n_vocab <= 32768)
    __shared__ unsigned long long gmax[64];
    __shared__ unsigned long long list_key[LCAP];
    __shared__ int32_t list_id[LCAP];
    __shared__ uint32_t n_list, bad;
    const int tid = threadIdx.x;
    seen[tid] = 0u;
    if (tid == 0) { n_list = 0u; bad = 0u; }
    __syncthreads();
    if (tid < n_window) { const int id = window[tid]; if (id >= 0 && id < V) atomicOr(&seen[id >> 5], 1u << (id & 31)); }
    __syncthreads();
    unsigned long long key[NPT], best = 0ull;
    float lv[NPT];
#pragma unroll
    for (int u = 0; u < NPT; u++) lv[u] = logits[min(tid + u * 1024, V - 1)];
#pragma unroll
    for (int u = 0; u < NPT; u++) {
        const int i = tid + u * 1024;
        unsigned long long kk = 0ull;                  // below every real key
        if (i < V) {
            const float lf = lv[u];
            double sc;
            if ((seen[i >> 5] >> (i & 31)) & 1u) sc = lf < 0.0f ? (double) lf * scale * repeat_penalty : (double) lf * scale / repeat_penalty;   // utils.cpp:363-368
            else sc = (double) lf * scale;
            if (sc != sc) bad = 1u;
            const unsigned long long b = (unsigned long long) __double_as_longlong(sc);
            kk = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
            if (kk == 0ull) kk = 1ull;
        }
        key[u] = kk;
        best = kk > best ? kk : best;
    }
    // maximum over each 16-lane DPP row, then the minimum of the 64 row maxima
    best = dpp_max_u64(best, dpp_u64<DPP_QUAD_XOR1>(best));
    best = dpp_max_u64(best, dpp_u64<DPP_QUAD_XOR2>(best));
    best = dpp_max_u64(best, dpp_u64<DPP_ROW_HALF_MIRROR>(best));
    best = dpp_max_u64(best, dpp_u64<DPP_ROW_MIRROR>(best));
    if ((tid & 15) == 0) gmax[tid >> 4] = best;
    __syncthreads();
    unsigned long long T = gmax[0];
#pragma unroll
    for (int j = 1; j < 64; j++) { const unsigned long long o = gmax[j]; T = o < T ? o : T; }      // (unrolled: the 63 LDS reads go out together)
    if (T == 0ull) {                                   // a group without a real value: V < 1024 * ... (tiny vocabularies) -- host path
        if (tid == 0) { flags[0] = 0; flags[1] = 0; }
        return;
    }
#pragma unroll
    for (int u = 0; u < NPT; u++) {
        if (key[u] >= T) {
            const uint32_t at = atomicAdd(&n_list, 1u);
            if (at < (uint32_t) LCAP) { list_key[at] = key[u]; list_id[at] = tid + u * 1024; }
        }
    }
    __syncthreads();
    const int n = (int) (n_list < (uint32_t) LCAP ? n_list : (uint32_t) LCAP);
    if (n_list > (uint32_t) LCAP) bad = 1u;            // (a flood of equal values at T)
    if (tid < n) {
        const unsigned long long mine = list_key[tid];
        const int my_id = list_id[tid];
        int rank = 0;
        bool dup = false;
        // (eight entries per trip so that their LDS reads are in flight together: rolled, every entry was a dependent LDS round
        //  trip -- ~300 of them, the largest part of this kernel's 39 us)
        int j = 0;
        for (; j + 8 <= n; j += 8) {
            unsigned long long o[8]; int oid[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { o[u] = list_key[j + u]; oid[u] = list_id[j + u]; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                dup = dup || (j + u != tid && o[u] == mine);
                rank += (o[u] > mine || (o[u] == mine && oid[u] < my_id)) ? 1 : 0;
            }
        }
        for (; j < n; j++) {
            const unsigned long long o = list_key[j];
            dup = dup || (j != tid && o == mine);
            rank += (o > mine || (o == mine && list_id[j] < my_id)) ? 1 : 0;
        }
        if (rank <= k && dup) bad = 1u;                // an equality among the k best or between the k-th and its runner-up
        if (rank < k) {
            const unsigned long long b = (mine >> 63) ? (mine & 0x7fffffffffffffffull) : ~mine;
            out_score[rank] = __longlong_as_double((long long) b);
            out_id[rank] = my_id;
        }
    }
    __syncthreads();
    if (tid == 0) { flags[0] = (bad == 0u && n >= k) ? 1 : 0; flags[1] = n; }
}

// a pipeline stage that does not pick the token still has to advance its device-resident position
__global__ void k_advance(int32_t *__restrict__ st) {
    if (threadIdx.x == 0) { st[0] += 1; st[1] += 1; }
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

hipError_t set_phase_probe(unsigned long long *dev_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase_probe), &dev_buf, sizeof(dev_buf));
}

hipError_t init_kernel_attrs() {
    // fused prologues of wide models (K = 22016) need more than the default 64 KB of dynamic LDS
    const int cap = 160 * 1024;
#define LH_ATTR(KERNEL) do { hipError_t e_ = hipFuncSetAttribute((const void *) KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, cap); if (e_ != hipSuccess) return e_; } while (0)
    LH_ATTR(k_prep_qa<PREP_PLAIN>); LH_ATTR(k_prep_qa<PREP_NORM>); LH_ATTR(k_prep_qa<PREP_SILU_MUL>);
#define LH_ATTR_G1(PRE, EPI, PG) LH_ATTR((k_gemv<PRE, EPI, 16, false, PG>)); LH_ATTR((k_gemv<PRE, EPI, 16, true, PG>)); LH_ATTR((k_gemv<PRE, EPI, 22, true, PG>)); \
    LH_ATTR((k_gemv<PRE, EPI, 18, true, PG>)); LH_ATTR((k_gemv<PRE, EPI, 14, true, PG>)); LH_ATTR((k_gemv<PRE, EPI, 10, true, PG>)); LH_ATTR((k_gemv<PRE, EPI, 8, true, PG>)); LH_ATTR((k_gemv<PRE, EPI, 4, true, PG>))
    LH_ATTR_G1(PRE_QA, EPI_STORE, 4); LH_ATTR_G1(PRE_QA, EPI_STORE, 12); LH_ATTR_G1(PRE_QA, EPI_RESID, 4); LH_ATTR_G1(PRE_QA, EPI_RESID, 12);
    LH_ATTR_G1(PREP_NORM, EPI_STORE, 1); LH_ATTR_G1(PREP_NORM, EPI_STORE, 2); LH_ATTR_G1(PREP_PLAIN, EPI_RESID, 1); LH_ATTR_G1(PREP_PLAIN, EPI_RESID, 2);
    LH_ATTR_G1(PREP_SILU_MUL, EPI_RESID, 1); LH_ATTR_G1(PREP_NORM, EPI_SILU_QA, 1);
    LH_ATTR_G1(PREP_NORMP, EPI_STORE, 1); LH_ATTR_G1(PREP_NORMP, EPI_STORE, 2); LH_ATTR_G1(PREP_NORMP, EPI_SILU_QA, 1);
    LH_ATTR_G1(PRE_QA, EPI_SILU_QA, 1);
    LH_ATTR_G1(PREP_NORM_TAG, EPI_STORE, 1); LH_ATTR_G1(PREP_NORM_TAG, EPI_STORE, 2); LH_ATTR_G1(PRE_QA, EPI_RESID_TAG, 4); LH_ATTR_G1(PRE_QA, EPI_RESID_TAG, 12);
#undef LH_ATTR_G1
#undef LH_ATTR_G
#define LH_ATTR_SK(NC) LH_ATTR((k_gemm_skinny<NC, 1, EPI_STORE>)); LH_ATTR((k_gemm_skinny<NC, 1, EPI_RESID>)); LH_ATTR((k_gemm_skinny<NC, 2, EPI_STORE>)); LH_ATTR((k_gemm_skinny<NC, 2, EPI_RESID>))
    LH_ATTR_SK(1); LH_ATTR_SK(2); LH_ATTR_SK(3); LH_ATTR_SK(4); LH_ATTR_SK(5);
    LH_ATTR((k_gemm_skinny<5, 1, EPI_ROPE_KV>)); LH_ATTR((k_gemm_skinny<5, 1, EPI_SILU_QA>));
    LH_ATTR((k_gemm_skinny<1, 1, EPI_ROPE_KV>)); LH_ATTR((k_gemm_skinny<2, 1, EPI_ROPE_KV>)); LH_ATTR((k_gemm_skinny<3, 1, EPI_ROPE_KV>)); LH_ATTR((k_gemm_skinny<4, 1, EPI_ROPE_KV>));
    LH_ATTR((k_gemm_skinny<1, 1, EPI_SILU_QA>)); LH_ATTR((k_gemm_skinny<2, 1, EPI_SILU_QA>)); LH_ATTR((k_gemm_skinny<3, 1, EPI_SILU_QA>)); LH_ATTR((k_gemm_skinny<4, 1, EPI_SILU_QA>));
    LH_ATTR(k_attn); LH_ATTR(k_dec_pv_blk<false>); LH_ATTR(k_dec_pv_blk<true>); LH_ATTR(k_dec_attn_x); LH_ATTR((k_qkv_attn<PREP_NORMP, 8, 1>)); LH_ATTR((k_qkv_attn<PREP_NORM, 8, 1>)); LH_ATTR((k_qkv_attn<PREP_NORMP, 10, 2>)); LH_ATTR((k_qkv_attn<PREP_NORM, 10, 2>)); LH_ATTR((k_qkv_attn<PREP_NORMP, 4, 2>)); LH_ATTR((k_qkv_attn<PREP_NORM, 4, 2>));
    LH_ATTR((k_qkv_attn<PREP_NORM_TAG, 8, 1>)); LH_ATTR((k_qkv_attn<PREP_NORM_TAG, 10, 2>)); LH_ATTR((k_qkv_attn<PREP_NORM_TAG, 4, 2>));
#undef LH_ATTR
    return hipSuccess;
}

// counts[0] / counts[1]: entries of the SiLU / exp table (non-NaN inputs) the device formulas do NOT reproduce
__global__ void k_check_lut_math(const uint16_t *__restrict__ T_silu, const uint16_t *__restrict__ T_exp, uint32_t *__restrict__ counts) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 65536u) return;
    const uint16_t h = (uint16_t) i;
    if ((h & 0x7C00u) == 0x7C00u && (h & 0x03FFu)) return;          // NaN inputs: payloads are not compared
    if (silu_math_bits(h) != T_silu[i]) atomicAdd(counts, 1u);
    if (exp_math_bits(h) != T_exp[i]) atomicAdd(counts + 1, 1u);
}
int g_lut_math = 0;          // bit 0: SiLU, bit 1: exp (bits 2, 3: epilogue ablation switches of LLAMAHIP_EPI_ABLATE, measurement only) -- set by launch_check_lut_math (process-wide: the tables are the same for every model)
hipError_t launch_check_lut_math(const uint16_t *T_silu, const uint16_t *T_exp, hipStream_t st) {
    static const bool off = getenv("LLAMAHIP_NO_LUT_MATH") != nullptr;       // measurement only
    uint32_t *d_counts = nullptr, h[2] = { 1, 1 };
    hipError_t e = hipMalloc((void **) &d_counts, 8);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d_counts, 0, 8, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_check_lut_math, dim3(256), dim3(256), 0, st, T_silu, T_exp, d_counts);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h, d_counts, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void) hipFree(d_counts);
    if (e != hipSuccess) return e;
    g_lut_math = off ? 0 : ((h[0] == 0 ? 1 : 0) | (h[1] == 0 ? 2 : 0));
    if (const char *ab = getenv("LLAMAHIP_EPI_ABLATE")) g_lut_math |= (atoi(ab) & 3) << 2;
    return hipSuccess;
}

hipError_t launch_add(const float *a, const float *b, float *c, long n, hipStream_t st) {
    hipLaunchKernelGGL(k_add, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, a, b, c, n);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_repack(const uint8_t *src_aos, uint8_t *dst, int M, int K, int gmap, int goff, hipStream_t st) {
    const int nb = K / 32, ngroups = (M + 7) / 8, nchunks = (nb + 7) / 8;
    const long total = (long) ngroups * nchunks * 64;
    const int bs = 256;
    hipLaunchKernelGGL(k_repack_q4, dim3((unsigned) ((total + bs - 1) / bs)), dim3(bs), 0, st, src_aos, dst, M, nb, ngroups, nchunks, gmap, goff);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_embed(const int32_t *tokens, const uint8_t *emb, float *x, int d, int N, hipStream_t st) {
    hipLaunchKernelGGL(k_embed, dim3(N, N <= 64 ? (d / 2 + 255) / 256 : 1), dim3(256), 0, st, tokens, emb, x, d);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

size_t prep_lds_bytes(int K) { return 32 * sizeof(double) + ((size_t) K + K / 32 + 64) * sizeof(float); }

hipError_t launch_embed_part(const int32_t *token, const uint8_t *emb, float *x, int d, double *part_out, hipStream_t st, uint32_t *epoch, uint64_t *xt,
                             const uint64_t *token_mb, const int32_t *state, uint32_t *fault, int n_vocab) {
    hipLaunchKernelGGL(k_embed_part, dim3(1), dim3(256), 0, st, token, emb, x, d, (f64x2 *) part_out, epoch, xt, token_mb, state, fault, n_vocab);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_tag_row(const float *x, int d, const uint32_t *epoch, uint64_t *xt, hipStream_t st) {
    hipLaunchKernelGGL(k_tag_row, dim3((d + 1023) / 1024), dim3(256), 0, st, x, d, epoch, xt);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_prep(int mode, const float *in0, const float *in1, long in_stride, long in1_stride, int K, int N,
                       uint32_t *qa_A, float *qa_d, float *y_out, uint8_t *raw_out, const uint16_t *T_silu,
                       hipStream_t st) {
    const int Kp = (K + 255) / 256 * 256;
    static const bool slow_only = getenv("LLAMAHIP_PREP_LDS") != nullptr;      // measurement: the LDS-staged kernel for everything
    const int nh = K / 16;
    if (!y_out && !raw_out && !slow_only && (mode != PREP_NORM || nh <= 1024)) {
        // register-resident kernel: NORM = one workgroup per row, the others sliced 256 half-blocks per workgroup
        const int nt = mode == PREP_NORM ? (nh + 63) / 64 * 64 : 256;
        const dim3 grid(N, mode == PREP_NORM ? 1 : (nh + nt - 1) / nt);
#define LH_PREPF(MODE) hipLaunchKernelGGL(k_prep_fast<MODE>, grid, dim3(nt), 0, st, in0, in1, in_stride, in1_stride, K, Kp, qa_A, qa_d, T_silu)
        switch (mode) {
            case PREP_PLAIN:    LH_PREPF(PREP_PLAIN); break;
            case PREP_NORM:     LH_PREPF(PREP_NORM); break;
            case PREP_SILU_MUL: LH_PREPF(PREP_SILU_MUL); break;
            default: return hipErrorInvalidValue;
        }
#undef LH_PREPF
        LH_LAUNCH_CHECK();
        return hipSuccess;
    }
    const size_t lds = prep_lds_bytes(K);
#define LH_PREP(MODE) hipLaunchKernelGGL(k_prep_qa<MODE>, dim3(N), dim3(256), lds, st, in0, in1, in_stride, in1_stride, K, Kp, qa_A, qa_d, y_out, raw_out, T_silu)
    switch (mode) {
        case PREP_PLAIN:    LH_PREP(PREP_PLAIN); break;
        case PREP_NORM:     LH_PREP(PREP_NORM); break;
        case PREP_SILU_MUL: LH_PREP(PREP_SILU_MUL); break;
        default: return hipErrorInvalidValue;
    }
#undef LH_PREP
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

static int pick_waves(int ngroups) {
    static const int ovr = getenv("LLAMAHIP_WAVES") ? atoi(getenv("LLAMAHIP_WAVES")) : 0;      // tuning override (measurement only)
    if (ovr == 1 || ovr == 2 || ovr == 4) return ovr;
    // aim for >= 2 workgroups per CU (256 CUs) before growing the workgroup
    if (ngroups >= 4 * 512) return 4;
    if (ngroups >= 2 * 512) return 2;
    return 1;
}

// Ring depth for a row of `nchunks` chunks (always even, see k_gemv).  Launches with >= 4 waves per
// CU keep the ring shallow (8 or 10 slots: 128 VGPRs, 4 waves per SIMD); small launches (2 waves per
// CU) need the depth for bytes in flight.  Among the candidates the one padding the fewest zero-tile
// chunks wins, ties go to the deeper ring.
static int pick_depth(int nchunks, int ngroups) {
    // tuning override (measurement only): LLAMAHIP_DEPTH="small,mid,big" ring depths by launch size
    static int ovr[3] = { -1, -1, -1 };
    if (ovr[0] == -1) {
        ovr[0] = ovr[1] = ovr[2] = 0;
        if (const char *e = getenv("LLAMAHIP_DEPTH")) sscanf(e, "%d,%d,%d", &ovr[0], &ovr[1], &ovr[2]);
    }
    const int o = ovr[ngroups >= 2048 ? 2 : ngroups >= 1024 ? 1 : 0];
    if (o == 4 || o == 8 || o == 10 || o == 14 || o == 16 || o == 18 || o == 22) return o;
    static const int very_shallow[] = { 4 }, shallow[] = { 10, 8 }, deep[] = { 10, 14, 16, 18, 22, 8 };
    // >= 8 waves per CU: a 4-deep ring (24 VGPRs) still keeps > 40 KB per CU in flight.
    // Small launches: measured on MI355X (w2, 43 chunks) the 10-deep ring beats 14..22 although it
    // pads 7 zero-tile chunks -- smaller code and fewer live registers win; take the first candidate
    // that wastes <= 20 %, else the least wasteful.
    const int *cand = ngroups >= 2048 ? very_shallow : ngroups >= 1024 ? shallow : deep;
    const int n = ngroups >= 2048 ? 1 : ngroups >= 1024 ? 2 : 6;
    int best = cand[0], best_waste = 1 << 30;
    for (int i = 0; i < n; i++) {
        const int d = cand[i];
        const int waste = (nchunks + d - 1) / d * d - nchunks;
        if (waste * 5 <= nchunks) return d;
        if (waste < best_waste) { best_waste = waste; best = d; }
    }
    return best;
}

template <int PRE, int EPI, int PG>
static hipError_t launch_gemv_pg(const QMat &w, int nw, const uint32_t *qa_A, const float *qa_d,
                                 const float *in0, const float *in1, float *y, const float *resid,
                                 const uint16_t *T_silu,
                                 uint32_t *out_A, float *out_d, const NormPart &np, hipStream_t st, const MailboxIO *mb = nullptr) {
    const int grid = (w.ngroups + nw - 1) / nw;
    size_t lds = (size_t) w.nchunks * 64 * 4 + (size_t) w.nchunks * 8 * 4 + 32 * sizeof(double);
    if (PRE == PREP_SILU_MUL) lds += prep_lds_bytes(w.K);      // only the LDS-staged prologues need y scratch
    lds = (lds + 15) & ~(size_t) 15;
    GemvArgs ga = { w.tiles, w.ngroups, w.nchunks, w.M, w.gmapF8, qa_A, qa_d, in0, in1, w.K, y, resid, T_silu, out_A, out_d,
                    (const f64x2 *) np.in, np.n_in, (f64x2 *) np.out, nullptr, 0, 0, g_lut_math };
    if (mb) {       // a row of a pipeline mailbox on one side of this launch
        ga.in_t = mb->in_t; ga.resid_t = mb->resid_t; ga.out_t = mb->out_t; ga.slot_in = ga.slot_resid = ga.slot_out = 0;
        ga.pos_w = mb->pos_w; ga.patience = 7; ga.fault = mb->fault; ga.sync = mb->epoch; ga.lut_math |= mb->test_bits;
    }
#define LH_GO(D, RING) hipLaunchKernelGGL((k_gemv<PRE, EPI, D, RING, PG>), dim3(grid), dim3(nw * 64), lds + ((LH_GEMV_PAD && (RING)) ? (D) * 288 : 0), st, ga)
    if (np.out && grid > NORM_PART_MAX) return hipErrorInvalidValue;
    // rows that fit 16 slots: whole row in flight (latency-bound small matrices) unless the launch
    // already has >= 4 waves per CU, where an 8-deep ring saves 48 VGPRs and keeps 4 waves/SIMD resident
    static const bool no_full = getenv("LLAMAHIP_NO_FULL") != nullptr;      // tuning override (measurement only)
    if (w.nchunks <= 16 && !(w.nchunks == 16 && w.ngroups >= 1024) && !(no_full && w.nchunks == 16)) {
        LH_GO(16, false);
    } else {
        switch (pick_depth(w.nchunks, w.ngroups)) {
            case 4:  LH_GO(4, true); break;
            case 8:  LH_GO(8, true); break;
            case 10: LH_GO(10, true); break;
            case 14: LH_GO(14, true); break;
            case 18: LH_GO(18, true); break;
            case 22: LH_GO(22, true); break;
            default: LH_GO(16, true); break;
        }
    }
#undef LH_GO
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

// Workgroup size and prologue register budget.  fp32 prologues (norm / plain) keep K/4 float4
// granules in registers, PRE_QA keeps the nchunks*16 granules of the A array: small budgets (PG 4)
// keep the kernel near 128 VGPRs so 4 waves per SIMD stay resident; the large budget covers wide rows.
// workgroup size of the decode mat-vec for a (prologue, matrix) pair -- also what sizes the partial-sum
// array an EPI_RESID launch writes (gemv_resid_parts)
static int gemv_pick_nw_qa(const QMat &w, int *pg) {
    static const int resid_waves = getenv("LLAMAHIP_QA_WAVES") ? atoi(getenv("LLAMAHIP_QA_WAVES")) : 0;      // tuning override (measurement only)
    // Workgroups of ngroups / 256 waves (1, 2 or 4): ONE workgroup per CU where the matrix has fewer than 1024
    // row-groups.  Measured on the 7B decode step: w2 (512 row-groups) as 256 x 2 waves with the 12-granule operand
    // budget 7.60 us, as 128 x 4 waves with the 4-granule budget 8.52 us, as 512 x 1 wave 7.99 us; wo as 256 x 2 waves
    // 5.15 us against 5.41 us as 512 x 1 (profiles/r02_d_small_matvec_ab.txt).  The operand budget (4 or 12 granules
    // of 16 B per thread) follows from the workgroup size, not the other way round.
    int nw = w.ngroups >= 1024 ? 4 : w.ngroups >= 512 ? 2 : 1;
    if (resid_waves == 1 || resid_waves == 2 || resid_waves == 4) nw = resid_waves;
    const int need = w.nchunks * 16;
    for (; nw <= 4; nw *= 2) {
        if (need <= 4 * nw * 64) { *pg = 4; return nw; }
        if (need <= 12 * nw * 64) { *pg = 12; return nw; }
    }
    *pg = 0;
    return 0;
}
int gemv_resid_parts(const QMat &w) {
    int pg = 0;
    const int nw = gemv_pick_nw_qa(w, &pg);
    return nw ? (w.ngroups + nw - 1) / nw : 0;
}

template <int PRE, int EPI>
static hipError_t launch_gemv_t(const QMat &w, const uint32_t *qa_A, const float *qa_d,
                                const float *in0, const float *in1, float *y, const float *resid,
                                const uint16_t *T_silu,
                                uint32_t *out_A, float *out_d, const NormPart &np, hipStream_t st, const MailboxIO *mb = nullptr) {
#define LH_PGARGS w, nw, qa_A, qa_d, in0, in1, y, resid, T_silu, out_A, out_d, np, st, mb
    if constexpr (EPI == EPI_SILU_QA) {
        // 8 waves = 4 gate row-groups + the 4 matching up row-groups (interleaved layout)
        const int nw = 8;
        if (!w.gmapF8 || w.ngroups % 8 != 0 || w.K / 16 > 1 * 512) return hipErrorInvalidValue;
        if (PRE == PRE_QA && w.nchunks * 16 > 512) return hipErrorInvalidValue;      // (measurement variant only)
        return launch_gemv_pg<PRE, EPI, 1>(LH_PGARGS);
    } else if constexpr (PRE == PREP_SILU_MUL) {
        const int nw = pick_waves(w.ngroups);
        return launch_gemv_pg<PRE, EPI, 1>(LH_PGARGS);
    } else {
        int nw = pick_waves(w.ngroups);
        if constexpr (PRE == PRE_QA) {
            int pg = 0;
            nw = gemv_pick_nw_qa(w, &pg);
            if (np.out && (w.ngroups + std::max(nw, 1) - 1) / std::max(nw, 1) > NORM_PART_MAX) return hipErrorInvalidValue;
            if (pg == 4) return launch_gemv_pg<PRE, EPI, 4>(LH_PGARGS);
            if (pg == 12) return launch_gemv_pg<PRE, EPI, 12>(LH_PGARGS);
        } else {
            const int need = w.K / 16;            // half-block granules (32 VGPRs each with the norm weight)
            while (nw < 4 && need > 1 * nw * 64) nw *= 2;
            if (need <= 1 * nw * 64) return launch_gemv_pg<PRE, EPI, 1>(LH_PGARGS);
            if (need <= 2 * nw * 64) return launch_gemv_pg<PRE, EPI, 2>(LH_PGARGS);
        }
        return hipErrorInvalidValue;          // caller falls back to the unfused path
    }
#undef LH_PGARGS
}

static size_t gemv_lds_bytes(const QMat &w, int depth_pad) {
    size_t lds = (size_t) w.nchunks * 64 * 4 + (size_t) w.nchunks * 8 * 4 + 32 * sizeof(double);
    lds = (lds + 15) & ~(size_t) 15;
    return lds + (LH_GEMV_PAD ? (size_t) depth_pad * 288 : 0);
}
hipError_t launch_gemv(const QMat &w, int pre, int epi, const uint32_t *qa_A, const float *qa_d,
                       const float *in0, const float *in1, float *y, const float *resid,
                       const uint16_t *T_silu,
                       uint32_t *out_A, float *out_d, hipStream_t st, const NormPart *npp, const MailboxIO *mb) {
    // LLAMAHIP_NORM_MODE (measurement only): 0 = the reference's two-pass statistics in the prologue, 1 = one-pass
    // statistics in the prologue, 2 (default) = statistics handed over by the producer where the caller offers them
    static const int norm_mode = getenv("LLAMAHIP_NORM_MODE") ? atoi(getenv("LLAMAHIP_NORM_MODE")) : 2;
    NormPart np = npp ? *npp : NormPart();
    if (norm_mode < 2) np = NormPart();
    if (pre == PREP_NORM && np.in && np.n_in > 0 && np.n_in <= NORM_PART_MAX) pre = PREP_NORMP;
    else { np.in = nullptr; np.n_in = norm_mode == 0 ? -1 : 0; }
    // pipeline mailbox on one side of the launch: the row arrives tagged (first layer's wq|wk|wv; its wo takes the residual from
    // the same granules) or leaves tagged (last layer's w2)
    if (mb && mb->in_t && (pre == PREP_NORM || pre == PREP_NORMP) && epi == EPI_STORE) return launch_gemv_t<PREP_NORM_TAG, EPI_STORE>(w, qa_A, qa_d, in0, in1, y, resid, T_silu, out_A, out_d, np, st, mb);
    if (mb && (mb->resid_t || mb->out_t) && pre == PRE_QA && epi == EPI_RESID) return launch_gemv_t<PRE_QA, EPI_RESID_TAG>(w, qa_A, qa_d, in0, in1, y, resid, T_silu, out_A, out_d, np, st, mb);
    if (mb) return hipErrorInvalidValue;
#define LH_ARGS w, qa_A, qa_d, in0, in1, y, resid, T_silu, out_A, out_d, np, st
    // only the (prologue, epilogue) pairs the forward pass uses are instantiated
    if (pre == PRE_QA && epi == EPI_STORE)        return launch_gemv_t<PRE_QA, EPI_STORE>(LH_ARGS);
    if (pre == PRE_QA && epi == EPI_RESID)        return launch_gemv_t<PRE_QA, EPI_RESID>(LH_ARGS);
    if (pre == PREP_NORM && epi == EPI_STORE)     return launch_gemv_t<PREP_NORM, EPI_STORE>(LH_ARGS);
    if (pre == PREP_NORM && epi == EPI_SILU_QA)   return launch_gemv_t<PREP_NORM, EPI_SILU_QA>(LH_ARGS);
    if (pre == PREP_NORMP && epi == EPI_STORE)    return launch_gemv_t<PREP_NORMP, EPI_STORE>(LH_ARGS);
    if (pre == PREP_NORMP && epi == EPI_SILU_QA)  return launch_gemv_t<PREP_NORMP, EPI_SILU_QA>(LH_ARGS);
    if (pre == PRE_QA && epi == EPI_SILU_QA)      return launch_gemv_t<PRE_QA, EPI_SILU_QA>(LH_ARGS);        // llamahip_bench_gemv variant
    if (pre == PREP_PLAIN && epi == EPI_RESID)    return launch_gemv_t<PREP_PLAIN, EPI_RESID>(LH_ARGS);
    if (pre == PREP_SILU_MUL && epi == EPI_RESID) return launch_gemv_t<PREP_SILU_MUL, EPI_RESID>(LH_ARGS);
#undef LH_ARGS
    return hipErrorInvalidValue;
}

template <int NC>
static hipError_t launch_gemm_lds_t(const QMat &w, int epi, const uint32_t *qa_A, const float *qa_d, int ncols,
                                    float *y, long y_stride, const float *resid, long resid_stride, hipStream_t st) {
    const int nw = 4;
    const int grid = (w.ngroups + nw - 1) / nw;
    if (epi == EPI_RESID)
        hipLaunchKernelGGL((k_gemm_lds<NC, EPI_RESID>), dim3(grid), dim3(nw * 64), 0, st, w.tiles, w.ngroups, w.nchunks, w.M, w.gmapF8, qa_A, qa_d, ncols, y, y_stride, resid, resid_stride);
    else
        hipLaunchKernelGGL((k_gemm_lds<NC, EPI_STORE>), dim3(grid), dim3(nw * 64), 0, st, w.tiles, w.ngroups, w.nchunks, w.M, w.gmapF8, qa_A, qa_d, ncols, y, y_stride, resid, resid_stride);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

template <int NC, int RG>
static hipError_t launch_gemm_skinny_t(const QMat &w, int epi, const uint32_t *qa_A, const float *qa_d, int ncols, int ncg,
                                       float *y, long y_stride, const float *resid, long resid_stride, hipStream_t st) {
    const int nwg = (w.ngroups + 4 * RG - 1) / (4 * RG);
    const int grid = ((nwg + 7) / 8) * ncg * 8;
    const size_t lds = (size_t) NC * (w.nchunks + 4) * 288;
    if (epi == EPI_RESID)
        hipLaunchKernelGGL((k_gemm_skinny<NC, RG, EPI_RESID>), dim3(grid), dim3(256), lds, st, w.tiles, w.ngroups, w.nchunks, w.M, w.gmapF8, qa_A, qa_d, ncols, ncg, y, y_stride, resid, resid_stride,
                           (const uint16_t *) nullptr, (uint32_t *) nullptr, (float *) nullptr, 0L, 0L, RopeKvArgs{});
    else
        hipLaunchKernelGGL((k_gemm_skinny<NC, RG, EPI_STORE>), dim3(grid), dim3(256), lds, st, w.tiles, w.ngroups, w.nchunks, w.M, w.gmapF8, qa_A, qa_d, ncols, ncg, y, y_stride, resid, resid_stride,
                           (const uint16_t *) nullptr, (uint32_t *) nullptr, (float *) nullptr, 0L, 0L, RopeKvArgs{});
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

template <int NC>
static hipError_t launch_gemm_skinny_silu_t(const QMat &w, const uint32_t *qa_A, const float *qa_d, int ncols, int ncg,
                                            const uint16_t *T_silu, uint32_t *out_A, float *out_d, long out_strideA, long out_strideD, hipStream_t st) {
    const int nwg = (w.ngroups + 7) / 8;
    const int grid = ((nwg + 7) / 8) * ncg * 8;
    const size_t lds = (size_t) NC * (w.nchunks + 4) * 288;
    hipLaunchKernelGGL((k_gemm_skinny<NC, 1, EPI_SILU_QA>), dim3(grid), dim3(512), lds, st, w.tiles, w.ngroups, w.nchunks, w.M, w.gmapF8, qa_A, qa_d, ncols, ncg,
                       (float *) nullptr, 0L, (const float *) nullptr, 0L, T_silu, out_A, out_d, out_strideA, out_strideD, RopeKvArgs{});
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

template <int NC>
static hipError_t launch_gemm_skinny_rope_t(const QMat &w, const uint32_t *qa_A, const float *qa_d, int ncols, int ncg, const RopeKvArgs &ra, hipStream_t st) {
    const int nwg = (w.ngroups + 3) / 4;
    const int grid = ((nwg + 7) / 8) * ncg * 8;
    const size_t lds = (size_t) NC * (w.nchunks + 4) * 288;
    hipLaunchKernelGGL((k_gemm_skinny<NC, 1, EPI_ROPE_KV>), dim3(grid), dim3(256), lds, st, w.tiles, w.ngroups, w.nchunks, w.M, w.gmapF8, qa_A, qa_d, ncols, ncg,
                       (float *) nullptr, 0L, (const float *) nullptr, 0L, (const uint16_t *) nullptr, (uint32_t *) nullptr, (float *) nullptr, 0L, 0L, ra);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

static int skinny_max_rows() {
    // measured crossover against the row-per-lane kernel at 7B shapes: +25 % at 33 rows, +7 % at 56, -1 % at 63
    static const int v = getenv("LLAMAHIP_SKINNY_MAX") ? atoi(getenv("LLAMAHIP_SKINNY_MAX")) : 60;
    return v;
}
// column-group width of k_gemm_skinny for N rows: the widest (<= 4) that still leaves ~1500 waves, balanced
static int skinny_pick_nc(const QMat &w, int N) {
    static const int skinny_nc = getenv("LLAMAHIP_SKINNY_NC") ? atoi(getenv("LLAMAHIP_SKINNY_NC")) : 0;
    int nc = 4;
    while (nc > 1 && (long) w.ngroups * ((N + nc - 1) / nc) < 1536) nc--;
    if (nc > N) nc = N;
    if (skinny_nc >= 1 && skinny_nc <= 5) nc = skinny_nc;
    while (nc > 1 && (size_t) nc * (w.nchunks + 4) * 288 > 150 * 1024) nc--;
    const int ncg = (N + nc - 1) / nc;
    return (N + ncg - 1) / ncg;                            // balance the groups (9 columns: 3 + 3 + 3, not 4 + 4 + 1)
}

// Short evals, wq|wk|wv: mat-mul + RoPE + KV append in one launch (k_gemm_skinny<EPI_ROPE_KV>)
bool gemm_rope_kv_applies(const QMat &wqkv, int N, int d) {
    static const bool off = getenv("LLAMAHIP_NO_SKINNY_ROPE") != nullptr;      // measurement
    return !off && N >= 2 && N <= skinny_max_rows() && wqkv.gmapF8 == 0 && wqkv.M == 3 * d && d % 8 == 0;
}
hipError_t launch_gemm_rope_kv(const QMat &wqkv, const uint32_t *qa_A, const float *qa_d, int N, const RopeKvArgs &ra, hipStream_t st) {
    const int nc = skinny_pick_nc(wqkv, N), ncg = (N + nc - 1) / nc;
    switch (nc) {
    case 5:  return launch_gemm_skinny_rope_t<5>(wqkv, qa_A, qa_d, N, ncg, ra, st);
    case 4:  return launch_gemm_skinny_rope_t<4>(wqkv, qa_A, qa_d, N, ncg, ra, st);
    case 3:  return launch_gemm_skinny_rope_t<3>(wqkv, qa_A, qa_d, N, ncg, ra, st);
    case 2:  return launch_gemm_skinny_rope_t<2>(wqkv, qa_A, qa_d, N, ncg, ra, st);
    default: return launch_gemm_skinny_rope_t<1>(wqkv, qa_A, qa_d, N, ncg, ra, st);
    }
}

// Short evals on the interleaved w1|w3 matrix: mat-mul + SiLU * up + Q4_0 quantization of the result in one
// launch (k_gemm_skinny<EPI_SILU_QA>).  false = not applicable (row count, layout): use the separate steps.
bool gemm_silu_qa_applies(const QMat &w13, int N) {
    static const bool off = getenv("LLAMAHIP_NO_SKINNY_SILU") != nullptr;      // measurement
    return !off && N >= 2 && N <= skinny_max_rows() && w13.gmapF8 != 0 && w13.ngroups % 8 == 0;
}
hipError_t launch_gemm_silu_qa(const QMat &w13, const uint32_t *qa_A, const float *qa_d, int N, const uint16_t *T_silu,
                               uint32_t *out_A, float *out_d, long out_strideA, long out_strideD, hipStream_t st) {
    const int nc = skinny_pick_nc(w13, N), ncg = (N + nc - 1) / nc;
    switch (nc) {
    case 5:  return launch_gemm_skinny_silu_t<5>(w13, qa_A, qa_d, N, ncg, T_silu, out_A, out_d, out_strideA, out_strideD, st);
    case 4:  return launch_gemm_skinny_silu_t<4>(w13, qa_A, qa_d, N, ncg, T_silu, out_A, out_d, out_strideA, out_strideD, st);
    case 3:  return launch_gemm_skinny_silu_t<3>(w13, qa_A, qa_d, N, ncg, T_silu, out_A, out_d, out_strideA, out_strideD, st);
    case 2:  return launch_gemm_skinny_silu_t<2>(w13, qa_A, qa_d, N, ncg, T_silu, out_A, out_d, out_strideA, out_strideD, st);
    default: return launch_gemm_skinny_silu_t<1>(w13, qa_A, qa_d, N, ncg, T_silu, out_A, out_d, out_strideA, out_strideD, st);
    }
}

template <int NC, bool DB, int WPE>
static hipError_t launch_gemm_rows_t(const QMat &w, int epi, const uint32_t *qa_A, const float *qa_d, int ncols,
                                     float *y, long y_stride, const float *resid, long resid_stride, hipStream_t st) {
    const int ncg = (ncols + NC - 1) / NC;
    const int grid = ((w.nrb + 7) / 8) * ncg * 8;
    const size_t lds = NC == 1 ? (size_t) w.nchunks * 288 : 0;       // the single-column variant keeps its operand in LDS
    if (epi == EPI_RESID)
        hipLaunchKernelGGL((k_gemm_rows<NC, EPI_RESID, DB, WPE>), dim3(grid), dim3(64), lds, st, w.rows, w.nrb, w.nchunks, w.M, qa_A, qa_d, ncols, ncg, y, y_stride, resid, resid_stride);
    else
        hipLaunchKernelGGL((k_gemm_rows<NC, EPI_STORE, DB, WPE>), dim3(grid), dim3(64), lds, st, w.rows, w.nrb, w.nchunks, w.M, qa_A, qa_d, ncols, ncg, y, y_stride, resid, resid_stride);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_tiles_to_mtiles(const QMat &w, hipStream_t st) {
    const long total = (long) w.nrb32 * w.nchunks * 2 * 4 * 64;
    hipLaunchKernelGGL(k_tiles_to_mtiles, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, w.tiles, w.mt, w.ngroups, w.nchunks, w.nrb32, w.gmapF8);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_qa_to_qb(const uint32_t *qa_A, uint8_t *qb, int nchunks, int N, hipStream_t st) {
    const long total = (long) N * nchunks * 8 * 2;
    hipLaunchKernelGGL(k_qa_to_qb, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, qa_A, qb, nchunks, N);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

static hipError_t launch_gemm_mfma(const QMat &w, int epi, const uint8_t *qb, const float *qa_d, int ncols,
                                   float *y, long y_stride, const float *resid, long resid_stride, hipStream_t st, bool fast) {
    const int nct = (ncols + 63) / 64, nq = w.nchunks * 2;
    const int nrp = (w.nrb32 + 1) / 2;
    const int grid = ((nrp + 7) / 8) * nct * 8;
#define LH_MF(E, F) hipLaunchKernelGGL((k_gemm_mfma<E, F>), dim3(grid), dim3(256), 0, st, w.mt, w.nrb32, nq, w.M, qb, qa_d, ncols, nct, y, y_stride, resid, resid_stride)
    if (epi == EPI_RESID) { if (fast) LH_MF(EPI_RESID, true); else LH_MF(EPI_RESID, false); }
    else                  { if (fast) LH_MF(EPI_STORE, true); else LH_MF(EPI_STORE, false); }
#undef LH_MF
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_tiles_to_mt16(const QMat &w, hipStream_t st) {
    const long total = (long) w.nrb32 * w.nchunks * 2 * 4 * 64;
    hipLaunchKernelGGL(k_tiles_to_mt16, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, w.tiles, w.mt16, w.ngroups, w.nchunks, w.nrb32, w.gmapF8);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}
static hipError_t launch_gemm_mfma16(const QMat &w, int epi, const uint32_t *qa_A, uint8_t *qb16, const float *qa_d, int ncols,
                                     float *y, long y_stride, const float *resid, long resid_stride, hipStream_t st) {
    const long tot = (long) ncols * w.nchunks * 8 * 8;
    hipLaunchKernelGGL(k_qa_to_qb16, dim3((unsigned) ((tot + 255) / 256)), dim3(256), 0, st, qa_A, qb16, w.nchunks, ncols);
    LH_LAUNCH_CHECK();
    const int nct = (ncols + 63) / 64, nq = w.nchunks * 2;
    const int nrp = (w.nrb32 + 1) / 2;
    const int grid = ((nrp + 7) / 8) * nct * 8;
    if (epi == EPI_RESID) hipLaunchKernelGGL((k_gemm_mfma16<EPI_RESID>), dim3(grid), dim3(256), 0, st, w.mt16, w.nrb32, nq, w.M, qb16, qa_d, ncols, nct, y, y_stride, resid, resid_stride);
    else                  hipLaunchKernelGGL((k_gemm_mfma16<EPI_STORE>), dim3(grid), dim3(256), 0, st, w.mt16, w.nrb32, nq, w.M, qb16, qa_d, ncols, nct, y, y_stride, resid, resid_stride);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_tiles_to_rows(const QMat &w, hipStream_t st) {
    const long total = (long) w.nrb * (w.nchunks + 1) * 10 * 64;
    hipLaunchKernelGGL(k_tiles_to_rows, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, w.tiles, w.rows, w.ngroups, w.nchunks, w.nrb, w.gmapF8);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

// N activation rows (QA precomputed, row stride = Kp bytes / Kp/32 floats).
//   matrix has a row-lane copy: one k_gemm_rows launch (column-group width: see below)
//   else: LDS-staged column tiles of 16 (the last one clamped), small remainders as 8 / 4 columns
//   a single row always goes through the decode GEMV
// which kernel family served a mat-mul (tests assert that the full-size shapes take the path they are meant to)
long g_gemm_path_counts[GEMM_PATH_COUNT] = { 0, 0, 0, 0, 0 };

hipError_t launch_gemm(const QMat &w, int epi, const uint32_t *qa_A, const float *qa_d, int N,
                       float *y, long y_stride, const float *resid, long resid_stride, hipStream_t st, uint8_t *qb_ws, bool fast) {
    // Matrix-core path when its 64 x 64-output workgroups fill the chip twice over (measured crossover
    // against the row-per-lane kernel on MI355X: N ~ 256 for the 7B matrices; 1.25x faster at N = 1024).
    static const int mfma_min = getenv("LLAMAHIP_MFMA_MIN") ? atoi(getenv("LLAMAHIP_MFMA_MIN")) : 0;     // measurement override
    const long mfma_wgs = (long) ((w.nrb32 + 1) / 2) * ((N + 63) / 64);
    if (w.mt16 && !fast && qb_ws && (mfma_min ? N >= mfma_min : (N >= 64 && mfma_wgs >= 512))) {
        // matrix-core path, exact: fp16 operands (QB16: 2 bytes per element of these N activation rows), two chains per MFMA
        g_gemm_path_counts[GEMM_PATH_MFMA]++;
        return launch_gemm_mfma16(w, epi, qa_A, qb_ws, qa_d, N, y, y_stride, resid, resid_stride, st);
    }
    if (w.mt && qb_ws && (mfma_min ? N >= mfma_min : (N >= 64 && mfma_wgs >= 512))) {
        // matrix-core path on the int8 tiles (the opt-in fast path; LLAMAHIP_MFMA_I8: the round-1 exact kernel): needs the int8 operand (QB)
        g_gemm_path_counts[GEMM_PATH_MFMA]++;
        hipError_t e = launch_qa_to_qb(qa_A, qb_ws, w.nchunks, N, st);
        if (e != hipSuccess) return e;
        return launch_gemm_mfma(w, epi, qb_ws, qa_d, N, y, y_stride, resid, resid_stride, st, fast);
    }
    const long strideA = (long) w.nchunks * 64, strideD = (long) w.nchunks * 8;
    // short chunks: decode-shaped kernel, NC columns per wave; as many column groups as it takes to put
    // ~1500 waves on the chip (LLAMAHIP_SKINNY_MAX = 0 switches it off, LLAMAHIP_SKINNY_NC forces the width)
    if (N >= 2 && N <= skinny_max_rows()) {
        // two row-groups per wave (halves the LDS operand reads per row) measured 3-7 % slower at 9 columns
        static const int skinny_rg = getenv("LLAMAHIP_SKINNY_RG") ? atoi(getenv("LLAMAHIP_SKINNY_RG")) : 0;
        const int nc = skinny_pick_nc(w, N), ncg = (N + nc - 1) / nc;
        const int rg = skinny_rg == 2 ? 2 : 1;
        g_gemm_path_counts[GEMM_PATH_SKINNY]++;
#define LH_SK_ARGS w, epi, qa_A, qa_d, N, ncg, y, y_stride, resid, resid_stride, st
#define LH_SK_CASE(NCV) case NCV: return rg == 2 ? launch_gemm_skinny_t<NCV, 2>(LH_SK_ARGS) : launch_gemm_skinny_t<NCV, 1>(LH_SK_ARGS)
        switch (nc) {
        LH_SK_CASE(5);
        LH_SK_CASE(4);
        LH_SK_CASE(3);
        LH_SK_CASE(2);
        default: return rg == 2 ? launch_gemm_skinny_t<1, 2>(LH_SK_ARGS) : launch_gemm_skinny_t<1, 1>(LH_SK_ARGS);
        }
#undef LH_SK_CASE
#undef LH_SK_ARGS
    }
    static const bool no_rows = getenv("LLAMAHIP_GEMM_LDS") != nullptr;     // measurement: skip the row-lane kernel
    static const int force_nc = getenv("LLAMAHIP_GEMM_ROWS_NC") ? atoi(getenv("LLAMAHIP_GEMM_ROWS_NC")) : 0;
    if (w.rows && N >= 2 && !no_rows) {
        // widest column group that still gives the chip >= 2 waves per SIMD.  Wider groups (8, 16
        // columns: 191 / 249 VGPRs, 2 waves per SIMD) measured 10-16 % slower than 4 columns at 3 waves
        // per SIMD on a 512-token prompt: the kernel runs at ~85 % of its VALU issue limit and the third
        // wave is what hides the scalar-load latency of the operand.
        int nc = 1;
        for (int cand : { 4, 2 })
            if ((long) w.nrb * ((N + cand - 1) / cand) >= 2048) { nc = cand; break; }
        if (force_nc) nc = force_nc;
        g_gemm_path_counts[GEMM_PATH_ROWS]++;
#define LH_ROWS_ARGS w, epi, qa_A, qa_d, N, y, y_stride, resid, resid_stride, st
        switch (nc) {
        case 16: return launch_gemm_rows_t<16, true, 2>(LH_ROWS_ARGS);
        case 8:  return launch_gemm_rows_t<8, true, 2>(LH_ROWS_ARGS);
        case 4:  return launch_gemm_rows_t<4, true, 3>(LH_ROWS_ARGS);
        case 2:  return launch_gemm_rows_t<2, true, 3>(LH_ROWS_ARGS);
        default: return launch_gemm_rows_t<1, true, 2>(LH_ROWS_ARGS);
        }
#undef LH_ROWS_ARGS
    }
    g_gemm_path_counts[N == 1 ? GEMM_PATH_GEMV : GEMM_PATH_LDS]++;
    int n0 = 0;
    while (n0 < N) {
        const int rem = N - n0;
        const uint32_t *A = qa_A + n0 * strideA;
        const float *D = qa_d + n0 * strideD;
        float *yy = y + (size_t) n0 * y_stride;
        const float *rr = resid ? resid + (size_t) n0 * resid_stride : nullptr;
        hipError_t e;
        int step;
        if (rem == 1)   { step = 1;  e = launch_gemv(w, PRE_QA, epi, A, D, nullptr, nullptr, yy, rr, nullptr, nullptr, nullptr, st); }
        else if (rem > 8)      { step = rem < 16 ? rem : 16; e = launch_gemm_lds_t<16>(w, epi, A, D, step, yy, y_stride, rr, resid_stride, st); }
        else if (rem > 4)      { step = rem;                 e = launch_gemm_lds_t<8>(w, epi, A, D, step, yy, y_stride, rr, resid_stride, st); }
        else                   { step = rem;                 e = launch_gemm_lds_t<4>(w, epi, A, D, step, yy, y_stride, rr, resid_stride, st); }
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
                       int n_past, int N, int d, int H, int nth, const uint16_t *T_exp, const AttnWs *ws, hipStream_t st) {
    const int dh = d / H, T = n_past + N;
    const float kq_scale = 1.0f / sqrtf((float) d / (float) H);          // .mm:620
    static const bool old_only = getenv("LLAMAHIP_ATTN_ROWWISE") != nullptr;     // measurement: per-row kernel for every N
    static const int attnq_min = getenv("LLAMAHIP_ATTNQ_MIN") ? atoi(getenv("LLAMAHIP_ATTNQ_MIN")) : 2;
    if (ws && ws->S && dh == 128 && N >= attnq_min && !dbg_p && !dbg_kqv && T <= ws->T_cap && nth <= ws->nth_cap && !old_only) {
        for (int nb0 = 0; nb0 < N; nb0 += ws->NB) {
            const int nb = min(ws->NB, N - nb0), qb = (nb + 63) / 64;
            int KS = (6144 + qb * H - 1) / (qb * H);      // ~2 rounds of 3 waves per SIMD: the waves are latency-bound
            KS = KS < 1 ? 1 : KS > ws->KS_cap ? ws->KS_cap : KS;
            // scores: K rows through LDS broadcasts (k_attnq_scores_lds, 256 queries per workgroup) unless LLAMAHIP_ATTNQ_SCALAR
            // asks for round 1's scalar-cache variant; ~3 workgroups per CU: KS key slices
            static const bool scalar_k = getenv("LLAMAHIP_ATTNQ_SCALAR") != nullptr;
            static const bool lds_k = getenv("LLAMAHIP_ATTNQ_LDS") != nullptr;
            if (!scalar_k && !lds_k) {
                // scores on the fp32 matrix cores (k_attnq_scores_mfma): 16 queries per wave, ~4 waves per SIMD over key slices
                const int qt = (nb + 15) / 16;
                KS = (4096 + qt * H - 1) / (qt * H);
                KS = KS < 1 ? 1 : KS > ws->KS_cap ? ws->KS_cap : KS;
                hipLaunchKernelGGL(k_attnq_scores_mfma, dim3(qt, H, KS), dim3(64), 0, st, qr, Kc, ws->S, ws->pmax, n_past, N, nb0, ws->NB, d, T, kq_scale, KS);
            } else if (!scalar_k) {
                const int qb4 = (nb + 255) / 256;
                KS = (768 + qb4 * H - 1) / (qb4 * H);
                KS = KS < 1 ? 1 : KS > ws->KS_cap ? ws->KS_cap : KS;
                hipLaunchKernelGGL(k_attnq_scores_lds, dim3(qb4, H, KS), dim3(256), 0, st, qr, Kc, ws->S, ws->pmax, n_past, N, nb0, ws->NB, d, T, kq_scale, KS);
            } else
            hipLaunchKernelGGL(k_attnq_scores, dim3(qb, H, KS), dim3(64), 0, st, qr, Kc, ws->S, ws->pmax, n_past, N, nb0, ws->NB, d, T, kq_scale, KS);
            LH_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_attnq_softmax, dim3(qb, H), dim3(1024), 0, st, ws->S, ws->pmax, ws->inv, n_past, N, nb0, ws->NB, T, KS, T_exp);
            LH_LAUNCH_CHECK();
            // V*P on the fp32 matrix cores (k_attnq_pv_mfma: bit-identical fmaf chains) unless LLAMAHIP_ATTNQ_PV_VALU asks for round 1's kernel
            static const bool pv_valu = getenv("LLAMAHIP_ATTNQ_PV_VALU") != nullptr;
            if (!pv_valu) hipLaunchKernelGGL(k_attnq_pv_mfma, dim3(qb, H, nth), dim3(64), 0, st, ws->S, ws->inv, Vc, ws->part, n_past, N, nb0, ws->NB, d, T, nth);
            else
            hipLaunchKernelGGL(k_attnq_pv, dim3(qb, H, 4 * nth), dim3(64), 0, st, ws->S, ws->inv, Vc, ws->part, n_past, N, nb0, ws->NB, d, T, nth);
            LH_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_attnq_merge, dim3((nb + 1) / 2, H), dim3(256), 0, st, ws->part, merged, N, nb0, ws->NB, d, nth);
            LH_LAUNCH_CHECK();
        }
        return hipSuccess;
    }
    const size_t lds = 32 * sizeof(double) + ((size_t) T + (size_t) nth * dh + dh + 16) * sizeof(float);
    hipLaunchKernelGGL(k_attn, dim3(H, N), dim3(256), lds, st, qr, Kc, Vc, merged, dbg_p, dbg_kqv, n_past, N, d, dh, nth, kq_scale, T_exp);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

// Short prompt chunk (see k_decn_scores): scores -> soft_max + V*P + ordered combine + Q4_0 quantization of the
// merged rows straight into the QA operand of the wo mat-mul (no separate preparation launch).
//   sc : scratch of N * H * n_ctx floats
hipError_t launch_attn_short(const float *qr, const float *Kc, const float *Vc, float *sc, float *merged,
                             uint32_t *qa_A, float *qa_d, int n_past, int N, int d, int H, int n_ctx, int nth,
                             const uint16_t *T_exp, hipStream_t st) {
    const int dh = d / H, T = n_past + N;
    const float kq_scale = 1.0f / sqrtf((float) d / (float) H);          // .mm:620
    const int Kp = (d + 255) / 256 * 256;
    hipLaunchKernelGGL(k_decn_scores, dim3(H, (T + DEC_TS - 1) / DEC_TS, N), dim3(256), 0, st, qr, d, dh, Kc, sc, n_ctx, kq_scale, n_past);
    LH_LAUNCH_CHECK();
    const int nt = (32 * (nth < 32 ? nth : 32) + 63) / 64 * 64;
    const size_t lds = 32 * sizeof(double) + ((size_t) n_ctx + (size_t) nth * 32 + 16) * sizeof(float);
    hipLaunchKernelGGL(k_dec_pv_blk<true>, dim3(H, dh / 32, N), dim3(nt), lds, st, sc, Vc, d, dh, n_ctx, nth, merged, qa_A, qa_d, T_exp,
                       (const int32_t *) nullptr, n_past, (long) Kp / 4, (long) Kp / 32, g_lut_math);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

// true when every workgroup of a (H, Y) grid that shares blockIdx.x also shares an XCD (what k_dec_attn_x relies on)
bool xcd_selftest(int H, int Y, hipStream_t st) {
    uint32_t *d_out = nullptr;
    const size_t n = (size_t) H * Y;
    if (hipMalloc((void **) &d_out, n * 4) != hipSuccess) return false;
    std::vector<uint32_t> out(n, 0xffffffffu);
    bool ok = hipMemsetAsync(d_out, 0xff, n * 4, st) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_xcd_selftest, dim3(H, Y), dim3(64), 0, st, d_out);
        ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(out.data(), d_out, n * 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipStreamSynchronize(st) == hipSuccess;
    }
    if (ok) {
        for (int h = 0; h < H; h++)
            for (int y = 0; y < Y; y++)
                if (out[h + (size_t) H * y] != out[h] || out[h] > 15u) ok = false;
    }
    // ... and the same for a one-dimensional grid of that many workgroups of 256 threads (k_qkv_attn): block b on the XCD of block b % 8
    if (ok) {
        std::fill(out.begin(), out.end(), 0xffffffffu);
        ok = hipMemsetAsync(d_out, 0xff, n * 4, st) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(k_xcd_selftest, dim3((unsigned) n), dim3(256), 0, st, d_out);
            ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(out.data(), d_out, n * 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
                 hipStreamSynchronize(st) == hipSuccess;
        }
        for (size_t b = 0; ok && b < n; b++)
            if (out[b] != out[b & 7] || out[b] > 15u) ok = false;
    }
    (void) hipFree(d_out);
    return ok;
}

hipError_t launch_dec_attn(const float *qkv, int d, int H, int n_ctx, int nth, const double *tab, float *Kc, float *Vc,
                           float *sc, float *part, float *merged, uint32_t *qa_A, float *qa_d,
                           const uint16_t *T_exp, const int32_t *state, hipStream_t st, uint32_t *xsync, uint32_t *fault) {
    const int dh = d / H;
    const float kq_scale = 1.0f / sqrtf((float) d / (float) H);          // .mm:620
    (void) part;
    // scores and soft_max . V in one launch with an XCD-local hand-off (k_dec_attn_x); the caller passes xsync only
    // after xcd_selftest() confirmed the placement it relies on
    // (contexts beyond 1 024: the score slices no longer fit the chip next to the waiting workgroups at this kernel's 4 waves per
    //  SIMD -- 525 against 570 tokens/s at context 1 024, 429 against 481 at 2 048 on the 7B -- the two launches below take over)
    if (xsync && fault && nth <= 8 && dh % 32 == 0 && dh <= 256 && H % 8 == 0 && n_ctx <= 1024) {
        const int nsl = (n_ctx + DEC_TS - 1) / DEC_TS;
        const size_t lds_pv = 32 * sizeof(double) + ((size_t) n_ctx + (size_t) nth * 32 + 16) * sizeof(float);
        const size_t lds = std::max(lds_pv, (size_t) 2 * dh * sizeof(float));
        const AttnXArgs aa = { qkv, d, dh, tab, Kc, Vc, sc, n_ctx, nth, kq_scale, merged, qa_A, qa_d, T_exp, state, xsync, fault, g_lut_math, nullptr, nullptr, nullptr, 0 };
        hipLaunchKernelGGL(k_dec_attn_x, dim3(H, dh / 32 + nsl), dim3(256), lds, st, aa);
        LH_LAUNCH_CHECK();
        return hipSuccess;
    }
    // (a variant with one 16-wave workgroup per (head, column block) doing scores and soft_max . V measured slower in round 1 --
    //  13.3 us against 4.9 + 5.7 per layer at 7B, n_ctx 512 -- and was removed in round 3)
    const int nsl = (n_ctx + DEC_TS - 1) / DEC_TS;
    hipLaunchKernelGGL(k_dec_scores, dim3(H, nsl), dim3(256), 2 * dh * sizeof(float), st, qkv, d, dh, tab, Kc, Vc, sc, n_ctx, kq_scale, state);
    LH_LAUNCH_CHECK();
    const int nt = (32 * (nth < 32 ? nth : 32) + 63) / 64 * 64;      // whole waves: the DPP reductions need every lane live
    const size_t lds = 32 * sizeof(double) + ((size_t) n_ctx + (size_t) nth * 32 + 16) * sizeof(float);
    hipLaunchKernelGGL(k_dec_pv_blk<false>, dim3(H, dh / 32), dim3(nt), lds, st, sc, Vc, d, dh, n_ctx, nth, merged, qa_A, qa_d, T_exp, state, 0, 0L, 0L, g_lut_math);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

// wq|wk|wv mat-vec + decode attention in ONE launch (k_qkv_attn).  Applies to the shapes its mat-vec role is instantiated
// for (the 8-deep ring, 4-wave, one-granule variant: K = 4096) and head layouts whose workgroups line up with heads.
// the mat-vec role of k_qkv_attn is instantiated for (ring depth, granules per thread) = (8, 1) [7B], (10, 2) [13B], (4, 2) [65B]:
// what launch_gemv_t / launch_gemv_pg pick for wq|wk|wv of those models; returns 0 when the shape takes another variant
static int qkv_attn_variant(const QMat &w) {
    int nw = pick_waves(w.ngroups);
    const int need = w.K / 16;                  // as launch_gemv_t
    while (nw < 4 && need > nw * 64) nw *= 2;
    if (nw != 4) return 0;
    const int pg = need <= 256 ? 1 : need <= 512 ? 2 : 0;
    if (!pg || (w.nchunks <= 16 && !(w.nchunks == 16 && w.ngroups >= 1024))) return 0;          // (whole-row-in-flight variant: small models)
    const int D = pick_depth(w.nchunks, w.ngroups);
    if (D == 8 && pg == 1) return 1;
    if (D == 10 && pg == 2) return 2;
    if (D == 4 && pg == 2) return 3;
    return 0;
}
bool qkv_attn_applies(const QMat &w, int d, int H, int nth) {
    static const bool off = getenv("LLAMAHIP_NO_QKV_ATTN") != nullptr;
    if (off || H % 8 != 0 || d % H != 0) return false;
    const int dh = d / H;
    if (dh % 32 != 0 || dh > 256 || nth > 8 || w.gmapF8 || w.M != 3 * d || w.K != d || w.ngroups != 3 * d / 8) return false;
    return qkv_attn_variant(w) != 0;
}
hipError_t launch_qkv_attn(const QMat &w, const float *x, const float *norm_w, const NormPart &np, uint64_t *qkv2, uint64_t *sc2, uint32_t *epoch, int layer,
                           int d, int H, int n_ctx, int nth, const double *tab, float *Kc, float *Vc, float *merged, uint32_t *qa_A, float *qa_d,
                           const uint16_t *T_silu, const uint16_t *T_exp, const int32_t *state, uint32_t *fault, hipStream_t st,
                           const MailboxIO *mb) {
    const uint64_t *x_t = mb ? mb->in_t : nullptr;
    // x_t (first layer of a pipeline stage fed through a device-side mailbox): the input row arrives as tagged granules, slot 0
    const int dh = d / H, nsl = (n_ctx + DEC_TS - 1) / DEC_TS, gridA = w.ngroups / 4;
    const float kq_scale = 1.0f / sqrtf((float) d / (float) H);          // .mm:620
    static const int norm_mode = getenv("LLAMAHIP_NORM_MODE") ? atoi(getenv("LLAMAHIP_NORM_MODE")) : 2;      // as launch_gemv
    const bool normp = norm_mode >= 2 && np.in && np.n_in > 0 && np.n_in <= NORM_PART_MAX;
    const int variant = qkv_attn_variant(w);
    const size_t lds_mv = gemv_lds_bytes(w, variant == 1 ? 8 : variant == 2 ? 10 : 4);
    const size_t lds_pv = 32 * sizeof(double) + ((size_t) n_ctx + (size_t) nth * 32 + 16) * sizeof(float);
    const size_t lds = std::max(std::max(lds_mv, lds_pv), (size_t) 2 * dh * sizeof(float));
    // the mat-vec role writes tagged granules: y -> qkv2, sync -> the epoch word, sync_epoch = layer
    // measurement only, RESULTS ARE INVALID: LLAMAHIP_ATTN_NOWAIT=1 no poll waits, =2 only the soft_max . V role does not wait, =3 only the score role
    static const int nw_mode = getenv("LLAMAHIP_ATTN_NOWAIT") ? atoi(getenv("LLAMAHIP_ATTN_NOWAIT")) : 0;
    static const int nowait = nw_mode == 1 ? (0x100 | 0x400 | 0x800) : nw_mode == 2 ? 0x800 : nw_mode == 3 ? 0x400 : 0;
    static const int nosleep = (getenv("LLAMAHIP_POLL_SLEEP") && atoi(getenv("LLAMAHIP_POLL_SLEEP")) == 0) ? 0x200 : 0;     // measurement only
    // test only (tests/test_gpu_parity.py): the mat-vec role publishes a wrong tag and every poll gives up after 256 looks -> the
    // sticky fault word must come back as an error
    static const int fault_test = (getenv("LLAMAHIP_HANDOFF_FAULT_TEST") && atoi(getenv("LLAMAHIP_HANDOFF_FAULT_TEST")) < 2) ? 0x1000 : 0;     // (2: the wo launch of the overlapped schedule misbehaves instead)
    GemvArgs ga = { w.tiles, w.ngroups, w.nchunks, w.M, w.gmapF8, nullptr, nullptr, x, norm_w, w.K, (float *) qkv2, nullptr, T_silu, nullptr, nullptr,
                    (const f64x2 *) (normp ? np.in : nullptr), normp ? np.n_in : (norm_mode == 0 ? -1 : 0), nullptr, epoch, 0, layer, g_lut_math | fault_test, fault };
    if (x_t) { ga.in_t = x_t; ga.slot_in = 0; ga.part_in = nullptr; ga.npart = norm_mode == 0 ? -1 : 0; ga.pos_w = mb->pos_w; ga.patience = 7; ga.lut_math |= mb->test_bits; }
    const AttnXArgs aa = { nullptr, d, dh, tab, Kc, Vc, nullptr, n_ctx, nth, kq_scale, merged, qa_A, qa_d, T_exp, state, nullptr, fault, g_lut_math | nowait | nosleep | fault_test,
                           qkv2, sc2, epoch, layer };
    const int grid = gridA + H * (nsl + dh / 32);
#define LH_GOX(D, PG) { if (x_t) hipLaunchKernelGGL((k_qkv_attn<PREP_NORM_TAG, D, PG>), dim3(grid), dim3(256), lds, st, ga, aa, gridA, H); \
                        else if (normp) hipLaunchKernelGGL((k_qkv_attn<PREP_NORMP, D, PG>), dim3(grid), dim3(256), lds, st, ga, aa, gridA, H); \
                        else hipLaunchKernelGGL((k_qkv_attn<PREP_NORM, D, PG>), dim3(grid), dim3(256), lds, st, ga, aa, gridA, H); }
    if (variant == 1) LH_GOX(8, 1) else if (variant == 2) LH_GOX(10, 2) else if (variant == 3) LH_GOX(4, 2) else return hipErrorInvalidValue;
#undef LH_GOX
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

__global__ void k_bump_epoch(uint32_t *epoch) { epoch[0] = next_epoch(epoch[0]); }
hipError_t launch_bump_epoch(uint32_t *epoch, hipStream_t st) {
    hipLaunchKernelGGL(k_bump_epoch, dim3(1), dim3(1), 0, st, epoch);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

// The same selection in two launches (round 2): k_topk_candidates is ONE workgroup, i.e. 16 waves x ~5 000 instructions of
// fp64 key arithmetic on a single CU -- 34-39 us.  k_topk_keys spreads the per-logit work (penalty, score, order-preserving
// key, the 64 group maxima) over V / 1024 workgroups; k_topk_select (one workgroup) only compares the finished keys with the
// threshold and ranks the survivors.  Same groups (element i belongs to group (i % 1024) / 16), same threshold, same flags.
//   ws: keys[32768] u64 | gmax[64] u64 (zero between calls: k_topk_select clears it) | bad u32
__global__ void __launch_bounds__(1024)
k_topk_keys(const float *__restrict__ logits, int V, const int32_t *__restrict__ window, int n_window, double scale, double repeat_penalty,
            unsigned long long *__restrict__ keys, unsigned long long *__restrict__ gmax, uint32_t *__restrict__ badw) {
    __shared__ uint32_t seen[1024];
    const int tid = threadIdx.x, i = blockIdx.x * 1024 + tid;
    seen[tid] = 0u;
    __syncthreads();
    if (tid < n_window) { const int id = window[tid]; if (id >= 0 && id < V) atomicOr(&seen[id >> 5], 1u << (id & 31)); }
    __syncthreads();
    unsigned long long kk = 0ull;                      // below every real key
    if (i < V) {
        const float lf = logits[i];
        double sc;
        if ((seen[i >> 5] >> (i & 31)) & 1u) sc = lf < 0.0f ? (double) lf * scale * repeat_penalty : (double) lf * scale / repeat_penalty;   // utils.cpp:363-368
        else sc = (double) lf * scale;
        if (sc != sc) atomicOr(badw, 1u);
        const unsigned long long b = (unsigned long long) __double_as_longlong(sc);
        kk = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
        if (kk == 0ull) kk = 1ull;
    }
    keys[i] = kk;
    unsigned long long best = kk;
    best = dpp_max_u64(best, dpp_u64<DPP_QUAD_XOR1>(best));
    best = dpp_max_u64(best, dpp_u64<DPP_QUAD_XOR2>(best));
    best = dpp_max_u64(best, dpp_u64<DPP_ROW_HALF_MIRROR>(best));
    best = dpp_max_u64(best, dpp_u64<DPP_ROW_MIRROR>(best));
    if ((tid & 15) == 0 && best != 0ull) atomicMax(&gmax[tid >> 4], best);
}

__global__ void __launch_bounds__(1024)
k_topk_select(int V, int k, unsigned long long *__restrict__ keys, unsigned long long *__restrict__ gmax, uint32_t *__restrict__ badw,
              double *__restrict__ out_score, int32_t *__restrict__ out_id, int32_t *__restrict__ flags) {
    constexpr int NPT = 32, LCAP = 768;
    __shared__ unsigned long long list_key[LCAP];
    __shared__ int32_t list_id[LCAP];
    __shared__ uint32_t n_list, bad;
    const int tid = threadIdx.x;
    if (tid == 0) { n_list = 0u; bad = badw[0]; }
    unsigned long long key[NPT];
#pragma unroll
    for (int u = 0; u < NPT; u++) key[u] = keys[tid + u * 1024];              // (entries past V are 0: below every threshold)
    // threshold = the k-th LARGEST of the 64 group maxima (k <= 64): at least k logits are >= it, so the k best and anything tied
    // with the k-th are among the survivors -- and only a few more (the minimum of the maxima, as k_topk_candidates uses, lets
    // 300-700 through, and the rank pass below is quadratic in that)
    __shared__ unsigned long long s_T;
    if (tid == 0) s_T = 0ull;
    __syncthreads();
    if (tid < 64) {
        const unsigned long long v = gmax[tid];
        int r = 0;
#pragma unroll
        for (int j = 0; j < 64; j++) { const unsigned long long o = gmax[j]; r += (o > v || (o == v && j < tid)) ? 1 : 0; }
        if (r == k - 1) s_T = v;
    }
    __syncthreads();                                                           // everybody has read gmax / badw: clear them for the next call
    const unsigned long long T = s_T;
    if (tid < 64) gmax[tid] = 0ull;
    if (tid == 0) badw[0] = 0u;
    if (T == 0ull) {                                   // a group without a real value (tiny vocabularies) -- host path
        if (tid == 0) { flags[0] = 0; flags[1] = 0; }
        return;
    }
    // collect: one LDS atomic per WAVE (its survivor count), slots inside the wave's range by ballot prefix -- 512 same-address
    // atomics (one per wave and key slot) were most of this kernel's time
    {
        uint32_t cnt = 0;
        unsigned long long pass[NPT];
#pragma unroll
        for (int u = 0; u < NPT; u++) { pass[u] = __ballot(key[u] >= T); cnt += (uint32_t) __popcll(pass[u]); }
        uint32_t base = 0;
        if ((tid & 63) == 0 && cnt) base = atomicAdd(&n_list, cnt);
        base = (uint32_t) __builtin_amdgcn_readfirstlane((int) base);
        const unsigned long long lt = (1ull << (tid & 63)) - 1ull;
#pragma unroll
        for (int u = 0; u < NPT; u++) {
            if (key[u] >= T) {
                const uint32_t at = base + (uint32_t) __popcll(pass[u] & lt);
                if (at < (uint32_t) LCAP) { list_key[at] = key[u]; list_id[at] = tid + u * 1024; }
            }
            base += (uint32_t) __popcll(pass[u]);
        }
    }
    __syncthreads();
    const int n = (int) (n_list < (uint32_t) LCAP ? n_list : (uint32_t) LCAP);
    if (n_list > (uint32_t) LCAP) bad = 1u;            // (a flood of equal values at T)
    if (tid < n) {
        const unsigned long long mine = list_key[tid];
        const int my_id = list_id[tid];
        int rank = 0;
        bool dup = false;
        int j = 0;
        for (; j + 8 <= n; j += 8) {
            unsigned long long o[8]; int oid[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { o[u] = list_key[j + u]; oid[u] = list_id[j + u]; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                dup = dup || (j + u != tid && o[u] == mine);
                rank += (o[u] > mine || (o[u] == mine && oid[u] < my_id)) ? 1 : 0;
            }
        }
        for (; j < n; j++) {
            const unsigned long long o = list_key[j];
            dup = dup || (j != tid && o == mine);
            rank += (o > mine || (o == mine && list_id[j] < my_id)) ? 1 : 0;
        }
        if (rank <= k && dup) bad = 1u;                // an equality among the k best or between the k-th and its runner-up
        if (rank < k) {
            const unsigned long long b = (mine >> 63) ? (mine & 0x7fffffffffffffffull) : ~mine;
            out_score[rank] = __longlong_as_double((long long) b);
            out_id[rank] = my_id;
        }
    }
    __syncthreads();
    if (tid == 0) { flags[0] = (bad == 0u && n >= k) ? 1 : 0; flags[1] = n; }
}

hipError_t launch_topk_candidates(const float *logits, int V, const int32_t *window, int n_window, double scale, double repeat_penalty, int k,
                                  double *out_score, int32_t *out_id, int32_t *flags, hipStream_t st, void *ws) {
    if (V > 32768 || k < 1 || k > 64 || n_window > 1024) return hipErrorInvalidValue;
    static const bool one_launch = getenv("LLAMAHIP_TOPK_ONE") != nullptr;          // measurement: round-2a single-workgroup kernel
    if (ws && !one_launch) {
        unsigned long long *keys = (unsigned long long *) ws, *gmax = keys + 32768;
        uint32_t *badw = (uint32_t *) (gmax + 64);
        hipLaunchKernelGGL(k_topk_keys, dim3((V + 1023) / 1024), dim3(1024), 0, st, logits, V, window, n_window, scale, repeat_penalty, keys, gmax, badw);
        LH_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_topk_select, dim3(1), dim3(1024), 0, st, V, k, keys, gmax, badw, out_score, out_id, flags);
        LH_LAUNCH_CHECK();
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_topk_candidates, dim3(1), dim3(1024), 0, st, logits, V, window, n_window, scale, repeat_penalty, k, out_score, out_id, flags);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_argmax(const float *logits, int V, int32_t *out, int out_idx, int32_t *next_token, int32_t *state, hipStream_t st, uint64_t *token_mb) {
    hipLaunchKernelGGL(k_argmax, dim3(1), dim3(1024), 0, st, logits, V, out, out_idx, next_token, state, token_mb);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_quantize_offline(const void *src, int f16, uint8_t *dst, long nblocks, hipStream_t st) {
    hipLaunchKernelGGL(k_quantize_offline, dim3((unsigned) ((nblocks + 127) / 128)), dim3(128), 0, st, src, f16, dst, nblocks);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_advance(int32_t *state, hipStream_t st) {
    hipLaunchKernelGGL(k_advance, dim3(1), dim3(64), 0, st, state);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace lh
