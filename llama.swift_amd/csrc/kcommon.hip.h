// kcommon.hip.h -- device helpers shared by the HIP translation units of the Q4_0 hot path (included by prep.hip, decode.hip,
// prompt_gemm.hip, prompt_attn.hip).  Everything here is `static` / inline device code: each translation unit gets its own copy
// (no relocatable device code).  See decode.hip for the conventions (arithmetic order of the reference, HBM layouts).
#pragma once
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

#define LH_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)


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
static __device__ double block_sum_d(double v, double *red, int phase = 0) {
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
static __device__ void block_sum_d2(double &a, double &b, double *red, int phase = 0) {
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
static __device__ float block_max_f(float v, double *red, int phase = 0) {
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
#ifdef LH_DEFINE_PHASE_PROBE
__device__ unsigned long long *g_phase_probe = nullptr;
#endif
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
static __device__ void make_y(float *ybuf, double *red, const float *__restrict__ in0, const float *__restrict__ in1,
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
static __device__ void quantize_y(const float *ybuf, int K, int Kp, uint32_t *A, float *da, uint8_t *raw_out) {
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


}  // namespace lh
