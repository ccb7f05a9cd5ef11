// gemv_set.hip -- the Q4_0 x Q4_0 mat-mul for a FEW activation rows (2 .. 16): the rows of a batched decode step
// (llamahip_stage_step_set: one row per sequence) and of the reference's own prompt flow (nine tokens per llama_eval,
// .mm:880-888).  Same arithmetic and order as k_gemv (ggml_compute_forward_mul_mat_q4_0_f32, ggml.c:5987-6285, N > 1: 6134-6152,
// 6182-6222; vec_dot ggml.c:1415-1466), bit-exact.
//
// Why another kernel.  k_gemm_skinny gives a wave one row-group (8 rows x 8 chains) and <= 4 columns, and puts the other columns
// into more workgroups that stream the SAME weight tiles again.  Round 4's tables fit one model for every matrix at 9 rows: time =
// bytes pulled through the CUs' load paths / ~5.5 TB/s, whether the re-reads hit the L2 or not (w2: 3 x 28 MB -> 20 us; one row:
// 7.7 us).  Reading the weights once with all columns in one wave loses more than it gains on the 4 096-row matrices: 512 waves are
// half a wave per SIMD, and a wave alone on its SIMD issues one VALU instruction per ~8-12 cycles.  So here the waves that share a
// row-group share its weight bytes through LDS:
//   workgroup = RGW row-groups x CW "column waves".  Wave (rgi, ci) owns row-group blk * RGW + rgi and NC of the columns.
//   Weights: the CW waves of a row-group take turns fetching its chunks HBM -> VGPR (chunk c by wave c % CW, non-temporal, a
//     register ring of DR chunks each: CW * DR chunks of the row-group in flight, nothing staged twice) and publish each chunk to a
//     two-step LDS stage; ONE workgroup barrier per step of CW chunks, then every wave runs the CW chunks of the step against its own
//     columns.  Every weight byte crosses a CU's load path once per step of the model, whatever the number of rows.
//   Activations: the QA operands of all NC * CW columns are staged whole in LDS (as k_gemm_skinny), zero-padded to the step grid.
//   CW = 1 is the plain form (no stage, weights straight from the ring): k_gemm_skinny's loop with the epilogues below.
// 22 VALU per (lane, chunk, column) as k_gemv; LDS traffic per (wave, chunk): 1.25 KiB of weights + NC x 2.6 KiB of operands.
//
// Epilogues (per column):
//   EPI_STORE / EPI_RESID   y = acc (+ resid)                                             lm head; wo, w2 (.mm:649-654, 682-687)
//   EPI_ROPE_KV             wq|wk|wv: RoPE of q and k, append of k and v at the row's own position (k_gemm_skinny's epilogue;
//                           ggml.c:7076-7131, .mm:586-611)
//   EPI_SILU_QAH            interleaved w1|w3 in HALF-block workgroups (RGW = 4: 16 gate rows + the same 16 up rows): SiLU * up
//                           (ggml.c:1956-1963, .mm:678-680) and the Q4_0 quantization of the block for w2 (ggml.c:456-523); the two
//                           halves of a block exchange their partial amax per column as tagged granules inside one XCD's L2 (k_gemv's
//                           EPI_SILU_QAH: fmaxf is exact in any order).  Half blocks because whole blocks are F / 32 = 344 workgroups
//                           on 256 CUs at 7B.
#include <cstring>

#include "kcommon.hip.h"

namespace lh {

// (decode.hip) 8-byte granule {value, tag}: one store, one L1-bypassing load that sees both or neither; bounded spin, sticky fault word
__device__ __forceinline__ float set_poll_tagged(const uint64_t *p, uint32_t tag, uint32_t *fault, bool short_fuse) {
    uint64_t v;
    int spins = 0;
    for (;;) {
        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t) (v >> 32) == tag) break;
        __builtin_amdgcn_s_sleep(1);
        if (poll_give_up(spins, short_fuse ? (1 << 8) : (1 << 20), fault)) break;
    }
    return __builtin_bit_cast(float, (uint32_t) v);
}
__device__ __forceinline__ void set_store_tagged(uint64_t *p, float v, uint32_t tag) {
    __hip_atomic_store(p, (uint64_t) __builtin_bit_cast(uint32_t, v) | ((uint64_t) tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct GemvSetArgs {
    const uint8_t *wt; int ngroups, nchunks, M, gmapF8;
    const uint32_t *qa_A; const float *qa_d;        // [ncols] operand rows: nchunks * 64 dwords / nchunks * 8 floats apart
    int ncols, rgw;                                  // rgw: row-groups per workgroup (waves = rgw * CW)
    float *y; long y_stride; const float *resid; long resid_stride;
    const uint16_t *T_silu; uint32_t *out_A; float *out_d; long out_strideA, out_strideD;
    RopeKvArgs ra;
    uint64_t *amax_t; const uint32_t *epoch; int layer; uint32_t *fault; int lut_math;      // EPI_SILU_QAH (lut_math as GemvArgs: bit 0 SiLU evaluated, 0x1000 fault-injection test)
};

constexpr int SET_DR = 4;                            // register ring of a wave: chunks in flight (2 loads each)

// threads a launch may ask for: half-block workgroups are 4 row-groups x CW waves, the others at most 8 waves
template <int CW, int EPI> constexpr int set_max_threads() { return EPI == EPI_SILU_QAH ? (CW * 256 > 512 ? CW * 256 : 512) : 512; }

template <int NC, int CW, int EPI>
__global__ void __launch_bounds__((set_max_threads<CW, EPI>()))
k_gemv_set(const GemvSetArgs a) {
    constexpr int DR = SET_DR, NCW = NC * CW;
    constexpr bool SHARE = CW > 1;
    static_assert((DR & 1) == 0, "stage parity and operand-buffer parity are taken from the unrolled step index");
    extern __shared__ double smem_d[];
    const int nchunks = a.nchunks, ncols = a.ncols, rgw = a.rgw;
    const int steps = (nchunks + CW - 1) / CW;
    const int npad = steps * CW + 1;                                // chunks per column in LDS: the step grid + the one-ahead operand fetch
    u32x4 *sA = (u32x4 *) smem_d;                                   // [ncols][npad][16]
    f32x2 *sD = (f32x2 *) (sA + (size_t) ncols * npad * 16);        // [ncols][npad][4]: {d[t], d[t + 4]} -- lane t of a quad owns blocks t and t + 4
    uint8_t *stage = (uint8_t *) (sD + (size_t) ncols * npad * 4);  // [2][rgw][CW][TILE_BYTES]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nt = (int) blockDim.x;
    const int rgi = wave / CW, ci = wave - rgi * CW;
    const int blk = blockIdx.x;
    // (EPI_SILU_QAH: workgroup blk = half `(blk >> 3) & 1` of activation block `(blk >> 4) * 8 + (blk & 7)` -- the halves of a block are 8
    //  apart in the grid, one XCD; rgi 0, 1 = the half's two gate row-groups, rgi 2, 3 = the matching up row-groups of the interleaved order)
    const int qah_block = (blk >> 4) * 8 + (blk & 7), qah_half = (blk >> 3) & 1;
    const int g = EPI == EPI_SILU_QAH ? qah_block * 8 + (rgi >> 1) * 4 + qah_half * 2 + (rgi & 1) : blk * rgw + rgi;
    const bool valid = g < a.ngroups;
    const uint8_t *wbase = a.wt + (size_t) (valid ? g : a.ngroups - 1) * (nchunks + 1) * TILE_BYTES;
    const int k = lane & 7, t = lane & 3;
    const int woff = lane * 16, soff = 1024 + ((lane >> 3) * 8 + t * 2) * 4;
    const uint32_t store_tag = EPI == EPI_SILU_QAH ? make_tag(__builtin_nontemporal_load(a.epoch), a.layer + 1) : 0u;

    u32x4 wq[DR];
    f32x2 ws[DR];
#define LH_LOADW(SLOT, CH)                                                                         \
    {                                                                                              \
        const uint8_t *tp_ = wbase + (size_t) min((CH), nchunks) * TILE_BYTES;   /* tile `nchunks` is the zero tile */ \
        wq[SLOT] = __builtin_nontemporal_load((const u32x4 *) (tp_ + woff));                       \
        ws[SLOT] = __builtin_nontemporal_load((const f32x2 *) (tp_ + soff));                       \
    }
    // this wave's chunks are ci, ci + CW, ci + 2 CW, ...: the first DR of them go out before anything else
#pragma unroll
    for (int i = 0; i < DR; i++) LH_LOADW(i, i * CW + ci)
    __builtin_amdgcn_sched_barrier(0);
    // stage the columns' operands
    {
        constexpr int LB = 8;
        const int perA = nchunks * 16, perD = nchunks * 2;        // granules of 16 bytes per operand row
        const int totA = ncols * perA, totD = ncols * perD;
        for (int base = tid; base < totA; base += nt * LB) {
            u32x4 v[LB];
#pragma unroll
            for (int u = 0; u < LB; u++) {
                const int i = min(base + u * nt, totA - 1), n = i / perA, r = i - n * perA;
                v[u] = ((const u32x4 *) a.qa_A)[(long) n * perA + r];
            }
#pragma unroll
            for (int u = 0; u < LB; u++) {
                const int i = base + u * nt, n = i / perA, r = i - n * perA;
                if (i < totA) sA[(size_t) n * npad * 16 + r] = v[u];
            }
        }
        float *sDf = (float *) sD;
        for (int base = tid; base < totD; base += nt * LB) {
            f32x4 v[LB];
#pragma unroll
            for (int u = 0; u < LB; u++) {
                const int i = min(base + u * nt, totD - 1), n = i / perD, r = i - n * perD;
                v[u] = ((const f32x4 *) a.qa_d)[(long) n * perD + r];
            }
#pragma unroll
            for (int u = 0; u < LB; u++) {
                const int i = base + u * nt, n = i / perD, r = i - n * perD;
                if (i < totD) {                                   // granule r = blocks 4 (r & 1) .. + 3 of chunk r >> 1
                    float *o = sDf + ((size_t) n * npad + (r >> 1)) * 8 + (r & 1);
                    o[0] = v[u].x; o[2] = v[u].y; o[4] = v[u].z; o[6] = v[u].w;
                }
            }
        }
        const int zc = npad - nchunks;                             // zeroed chunks behind every column: A 16 granules, d 8 floats each
        for (int i = tid; i < ncols * zc * 18; i += nt) {
            const int n = i / (zc * 18), r = i - n * (zc * 18);
            if (r < zc * 16) sA[((size_t) n * npad + nchunks) * 16 + r] = u32x4{ 0u, 0u, 0u, 0u };
            else ((f32x4 *) sD)[((size_t) n * npad + nchunks) * 2 + (r - zc * 16)] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
        }
    }
    // (see k_gemm_skinny: after loops with run-time trip counts the compiler no longer knows the age of the ring loads; draining here
    //  makes the loop's entry state exact and the waits inside become the counted ones of the back edge.  The staging loads were
    //  issued behind the ring's and vmcnt retires in order, so this waits for nothing the first step would not wait for.)
    __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0), nothing else
    __syncthreads();

    float accs[NC];
#pragma unroll
    for (int n = 0; n < NC; n++) accs[n] = 0.0f;
    const int ncol0 = ci * NC;
    // (only the ncols real columns are staged; a wave's columns past them are clamped duplicates of the last one)
    int colofs[NC];
#pragma unroll
    for (int n = 0; n < NC; n++) colofs[n] = min(ncol0 + n, ncols - 1) * npad;
    // operands of item (chunk, column) are fetched one item ahead into the other half of a two-entry register buffer; the weights of
    // chunk c + 1 of a step while chunk c is consumed
    u32x4 la0[2], la1[2], wb[2];
    f32x2 ldd[2], wsb[2];
#define LH_LDSA(BUF, N, CH)                                                                        \
    {                                                                                              \
        const u32x4 *pa_ = sA + (colofs[N] + (CH)) * 16 + k * 2;                                   \
        la0[BUF] = pa_[0]; la1[BUF] = pa_[1];                                                      \
        ldd[BUF] = sD[(colofs[N] + (CH)) * 4 + t];                                                 \
    }
#define LH_LDSW(BUF, PAR, C)                                                                       \
    {                                                                                              \
        const uint8_t *sp_ = stage + (size_t) ((((PAR) * rgw + rgi) * CW) + (C)) * TILE_BYTES;     \
        wb[BUF] = *(const u32x4 *) (sp_ + woff); wsb[BUF] = *(const f32x2 *) (sp_ + soff);         \
    }
    // one (chunk, column) item: 8 integer dots onto the bits of 1.5 * 2^23, 4 packed subtractions, 2 scale products, the block-ordered
    // FMA chain with DPP-broadcast scales (k_gemv's LH_CONSUME)
#define LH_ITEM(W, SW, PB, ACC)                                                                    \
    {                                                                                              \
        const u32x4 a0 = la0[PB], a1 = la1[PB];                                                    \
        const float plo_ = (SW).x * ldd[PB].x, phi_ = (SW).y * ldd[PB].y;                          \
        const int i0_ = __builtin_amdgcn_sdot8((int) (W).x, (int) a0.x, 0x4B400000, true);         \
        const int i1_ = __builtin_amdgcn_sdot8((int) (W).x, (int) a0.y, 0x4B400000, true);         \
        const int i2_ = __builtin_amdgcn_sdot8((int) (W).y, (int) a0.z, 0x4B400000, true);         \
        const int i3_ = __builtin_amdgcn_sdot8((int) (W).y, (int) a0.w, 0x4B400000, true);         \
        const int i4_ = __builtin_amdgcn_sdot8((int) (W).z, (int) a1.x, 0x4B400000, true);         \
        const int i5_ = __builtin_amdgcn_sdot8((int) (W).z, (int) a1.y, 0x4B400000, true);         \
        const int i6_ = __builtin_amdgcn_sdot8((int) (W).w, (int) a1.z, 0x4B400000, true);         \
        const int i7_ = __builtin_amdgcn_sdot8((int) (W).w, (int) a1.w, 0x4B400000, true);         \
        const f32x2 mg_ = { 12582912.0f, 12582912.0f };                                            \
        const f32x2 q01_ = f32x2{ __builtin_bit_cast(float, i0_), __builtin_bit_cast(float, i1_) } - mg_; \
        const f32x2 q23_ = f32x2{ __builtin_bit_cast(float, i2_), __builtin_bit_cast(float, i3_) } - mg_; \
        const f32x2 q45_ = f32x2{ __builtin_bit_cast(float, i4_), __builtin_bit_cast(float, i5_) } - mg_; \
        const f32x2 q67_ = f32x2{ __builtin_bit_cast(float, i6_), __builtin_bit_cast(float, i7_) } - mg_; \
        LH_FMAC8_DPP(ACC, plo_, phi_, q01_, q23_, q45_, q67_);                                     \
    }
    // step S (I = S % DR, compile time): this wave's chunk of the step sits in ring slot I.  Publish it, refill the slot with this wave's
    // chunk of step S + DR, barrier, then the CW chunks of the step in block order against this wave's NC columns.  The stage has
    // two parities: step S + 1 writes the other one, and a wave reaches the barrier of step S + 1 only after it has read everything of
    // step S, so the writes of step S + 2 (behind that barrier) cannot overtake a reader.
#define LH_SSTEP(I, S)                                                                              \
    {                                                                                              \
        const int par_ = (I) & 1;                                                                  \
        if (SHARE) {                                                                               \
            uint8_t *sp_ = stage + (size_t) (((par_ * rgw + rgi) * CW) + ci) * TILE_BYTES;         \
            *(u32x4 *) (sp_ + woff) = wq[I]; *(f32x2 *) (sp_ + soff) = ws[I];                      \
            LH_LOADW(I, ((S) + DR) * CW + ci)                                                      \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                        \
            LH_LDSW(0, par_, 0)                                                                    \
        }                                                                                          \
        _Pragma("unroll")                                                                          \
        for (int c = 0; c < CW; c++) {                                                             \
            if (SHARE && c + 1 < CW) LH_LDSW((c + 1) & 1, par_, c + 1)                             \
            _Pragma("unroll")                                                                      \
            for (int n = 0; n < NC; n++) {                                                         \
                const int pb_ = ((I) * NCW + c * NC + n) & 1;                                      \
                if (n + 1 < NC) LH_LDSA(pb_ ^ 1, n + 1, (S) * CW + c)                              \
                else LH_LDSA(pb_ ^ 1, 0, (S) * CW + c + 1)                                         \
                __builtin_amdgcn_sched_barrier(0);      /* the reads for the next item go out before this item's arithmetic */ \
                if (SHARE) LH_ITEM(wb[c & 1], wsb[c & 1], pb_, accs[n])                            \
                else LH_ITEM(wq[I], ws[I], pb_, accs[n])                                           \
                __builtin_amdgcn_sched_barrier(0);                                                 \
            }                                                                                      \
        }                                                                                          \
        if (!SHARE) LH_LOADW(I, ((S) + DR) * CW + ci)                                              \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
    LH_LDSA(0, 0, 0)
    int s0 = 0;
    for (; s0 + DR <= steps; s0 += DR) {
#pragma unroll
        for (int i = 0; i < DR; i++) LH_SSTEP(i, s0 + i)
    }
    // the last steps % DR steps (the same unrolled sequence, cut short; `steps` is uniform over the workgroup)
#pragma unroll
    for (int i = 0; i < DR - 1; i++)
        if (s0 + i < steps) LH_SSTEP(i, s0 + i)
#undef LH_SSTEP
#undef LH_ITEM
#undef LH_LDSW
#undef LH_LDSA
#undef LH_LOADW

    int lg = g;
    if (a.gmapF8) { const int b8 = g >> 3, w8 = g & 7; lg = w8 < 4 ? b8 * 4 + w8 : a.gmapF8 + b8 * 4 + (w8 - 4); }
    const int m = lg * 8 + (lane >> 3);
    if (EPI == EPI_SILU_QAH) {
        // the workgroup's 32 outputs per column: gu[column][gate 0 .. 15 | up 0 .. 15]; the operand staging area is free again
        float *gu = (float *) smem_d;
        __syncthreads();
#pragma unroll
        for (int n = 0; n < NC; n++) {
            const float acc = fold8(accs[n]);
            if (k == 0) gu[(ncol0 + n) * 32 + rgi * 8 + (lane >> 3)] = acc;
        }
        __syncthreads();
        const bool inject = (a.lut_math & 0x1000) != 0;                       // fault-injection test: a tag nobody waits for, short polls
        for (int col = wave; col < NCW; col += (int) (blockDim.x >> 6)) {
            if (col >= ncols || qah_block * 8 >= a.ngroups) continue;
            const int i = lane & 15;
            const uint16_t gh = f2h_bits(gu[col * 32 + i]);
            const float act = h2f_bits((a.lut_math & 1) ? silu_math_bits(gh) : a.T_silu[gh]) * gu[col * 32 + 16 + i];
            float amax = wave_max_f(lane < 16 ? fabsf(act) : 0.0f);
            // the other half's partial amax of this column: one tagged granule each way inside this XCD's L2
            uint64_t *at = a.amax_t + (size_t) col * ((size_t) (a.ngroups / 8) * 2 + 16);
            const int hb = qah_block * 2 + qah_half;
            float other = 0.0f;
            if (lane == 0) {
                set_store_tagged(at + hb, amax, store_tag ^ (inject ? 1u : 0u));
                other = set_poll_tagged(at + (hb ^ 1), store_tag, a.fault, inject);
            }
            other = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, other)));
            amax = fmaxf(amax, other);
            const float dd = amax / 7.0f;                                      // ggml.c:479
            const float id = (amax != 0.0f) ? 7.0f / amax : 0.0f;              // ggml.c:482
            const uint32_t nib = (uint32_t) ((int) __builtin_rintf(act * id)) & 0xF;      // signed nibble of (q - 8)
            const int kk = lane & 7;
            const uint32_t e0 = __shfl(nib, 2 * kk), e1 = __shfl(nib, 2 * kk + 1);
            const int b = qah_block, c = b >> 3, j = b & 7;
            uint32_t *oA = a.out_A + (size_t) col * a.out_strideA;
            if (lane < 8) ((uint16_t *) (oA + (c * 8 + kk) * 8 + j))[qah_half] = (uint16_t) ((e0 | (e1 << 8)) << (4 * (j & 1)));
            if (lane == 0 && qah_half == 0) a.out_d[(size_t) col * a.out_strideD + b] = dd;
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < NC; n++) {
        float acc = fold8(accs[n]);
        const int col = ncol0 + n;
        if (EPI == EPI_ROPE_KV) {
            // (ggml.c:7076-7131, .mm:586-611; k_rope_kv) rows m, m ^ 1 = lanes 8 apart; m is even iff the lane's row is
            const RopeKvArgs &ra = a.ra;
            const float up = dpp_f<0x108>(acc), dn = dpp_f<0x118>(acc);        // row_shl:8 / row_shr:8
            if (valid && k == 0 && m < a.M && col < ncols) {
                const int which = m / ra.d, c = m - which * ra.d;
                // (batched decode step: the row's own position and cache)
                const int pos = ra.set ? ra.set->state[col][0] : ra.n_past + col;
                const long kvo = ra.set ? ra.set->kv_off[col] : 0L;
                if (which == 2) {
                    ra.Vc[kvo + (size_t) pos * ra.d + c] = acc;
                } else {
                    const int pe = (c % ra.dh) & ~1;
                    const double cs = ra.tab[(size_t) pos * ra.dh + pe], sn = ra.tab[(size_t) pos * ra.dh + pe + 1];
                    const double x0 = (double) ((c & 1) ? dn : acc), x1 = (double) ((c & 1) ? acc : up);
                    const float val = (c & 1) ? (float) (x0 * sn + x1 * cs) : (float) (x0 * cs - x1 * sn);
                    if (which == 0) ra.qr[(size_t) col * ra.d + c] = val;
                    else ra.Kc[kvo + (size_t) pos * ra.d + c] = val;
                }
            }
            continue;
        }
        if (valid && k == 0 && m < a.M && col < ncols) {
            if (EPI == EPI_RESID) acc = acc + a.resid[(size_t) col * a.resid_stride + m];
            a.y[(size_t) col * a.y_stride + m] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side: the (NC, CW) plan per row count, launchers
// ------------------------------------------------------------------------------------------------
struct SetPlan { int nc = 0, cw = 0, rgw = 0; size_t lds = 0; };

static size_t set_lds_bytes(const QMat &w, int N, int nc, int cw, int rgw) {
    const int steps = (w.nchunks + cw - 1) / cw, npad = steps * cw + 1;
    size_t lds = (size_t) N * npad * 288 + (cw > 1 ? (size_t) 2 * rgw * cw * TILE_BYTES : 0);
    return std::max(lds, (size_t) nc * cw * 64 * 4);
}

// Columns per wave and waves per row-group for N rows.  Small matrices (512 row-groups at 7B: wo, w2) need CW waves per row-group to put
// two waves on every SIMD; on the large ones (wq|wk|wv, w1|w3, the lm head) the row-groups alone do that and wider waves save LDS reads
// of the weights.  LLAMAHIP_SET_PLAN="nc,cw[,rgw]" overrides (measurement).
static SetPlan set_plan(const QMat &w, int N, int epi) {
    static const char *env = getenv("LLAMAHIP_SET_PLAN");
    SetPlan p;
    int nc, cw;
    const bool big = w.ngroups >= 1024;
    if (N <= 4) { if (big) { nc = N <= 2 ? N : 2; cw = (N + nc - 1) / nc; } else { nc = 1; cw = N; } }
    else if (N <= 6) { nc = 2; cw = 3; }
    else if (N <= 8) { nc = 2; cw = 4; }
    else if (N == 9) { nc = 3; cw = 3; }
    else if (N <= 12) { nc = 3; cw = 4; }
    else { nc = 4; cw = 4; }
    int rgw = epi == EPI_SILU_QAH ? 4 : (cw >= 3 ? 2 : 4);
    if (env) {
        int e_nc = 0, e_cw = 0, e_rgw = 0;
        const int got = sscanf(env, "%d,%d,%d", &e_nc, &e_cw, &e_rgw);
        if (got >= 2 && e_nc >= 1 && e_nc <= 4 && e_cw >= 1 && e_cw <= 4 && e_nc * e_cw >= N) { nc = e_nc; cw = e_cw; }
        if (got >= 3 && e_rgw >= 1 && e_rgw <= 8 && epi != EPI_SILU_QAH) rgw = e_rgw;
    }
    if (epi != EPI_SILU_QAH && rgw * cw > 8) rgw = 8 / cw;            // (set_max_threads)
    p.nc = nc; p.cw = cw; p.rgw = rgw;
    p.lds = set_lds_bytes(w, N, nc, cw, rgw);
    return p;
}

static bool set_disabled() {
    static const bool off = getenv("LLAMAHIP_NO_GEMV_SET") != nullptr;
    return off;
}
constexpr int SET_ROWS_MAX = 16;
constexpr size_t SET_LDS_CAP = 160 * 1024;

bool gemv_set_applies(const QMat &w, int N, int epi) {
    if (set_disabled() || N < 2 || N > SET_ROWS_MAX || !w.tiles) return false;
    if (epi == EPI_ROPE_KV && w.gmapF8 != 0) return false;
    if (epi == EPI_SILU_QAH && (w.gmapF8 == 0 || w.ngroups % 8 != 0)) return false;
    if (epi != EPI_STORE && epi != EPI_RESID && epi != EPI_ROPE_KV && epi != EPI_SILU_QAH) return false;
    return set_plan(w, N, epi).lds <= SET_LDS_CAP;
}

template <int NC, int CW>
static hipError_t launch_set_t(const GemvSetArgs &a, int epi, int grid, int nthreads, size_t lds, hipStream_t st) {
    switch (epi) {
    case EPI_STORE:    hipLaunchKernelGGL((k_gemv_set<NC, CW, EPI_STORE>), dim3(grid), dim3(nthreads), lds, st, a); break;
    case EPI_RESID:    hipLaunchKernelGGL((k_gemv_set<NC, CW, EPI_RESID>), dim3(grid), dim3(nthreads), lds, st, a); break;
    case EPI_ROPE_KV:  hipLaunchKernelGGL((k_gemv_set<NC, CW, EPI_ROPE_KV>), dim3(grid), dim3(nthreads), lds, st, a); break;
    case EPI_SILU_QAH: hipLaunchKernelGGL((k_gemv_set<NC, CW, EPI_SILU_QAH>), dim3(grid), dim3(nthreads), lds, st, a); break;
    default: return hipErrorInvalidValue;
    }
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

static hipError_t launch_set_any(const QMat &w, GemvSetArgs a, int epi, hipStream_t st) {
    const SetPlan p = set_plan(w, a.ncols, epi);
    if (p.lds > SET_LDS_CAP) return hipErrorInvalidValue;
    a.rgw = p.rgw;
    const int grid = epi == EPI_SILU_QAH ? (w.ngroups / 8 + 7) / 8 * 16 : (w.ngroups + p.rgw - 1) / p.rgw;
    const int nthreads = p.rgw * p.cw * 64;
#define LH_SP(NCV, CWV) if (p.nc == NCV && p.cw == CWV) return launch_set_t<NCV, CWV>(a, epi, grid, nthreads, p.lds, st)
    LH_SP(1, 2); LH_SP(1, 3); LH_SP(1, 4);
    LH_SP(2, 1); LH_SP(2, 2); LH_SP(2, 3); LH_SP(2, 4);
    LH_SP(3, 1); LH_SP(3, 3); LH_SP(3, 4);
    LH_SP(4, 1); LH_SP(4, 2); LH_SP(4, 4);
#undef LH_SP
    return hipErrorInvalidValue;
}

static GemvSetArgs set_args(const QMat &w, const uint32_t *qa_A, const float *qa_d, int N) {
    GemvSetArgs a;
    memset(&a, 0, sizeof(a));
    a.wt = w.tiles; a.ngroups = w.ngroups; a.nchunks = w.nchunks; a.M = w.M; a.gmapF8 = w.gmapF8;
    a.qa_A = qa_A; a.qa_d = qa_d; a.ncols = N;
    return a;
}

hipError_t launch_gemv_set(const QMat &w, int epi, const uint32_t *qa_A, const float *qa_d, int N,
                           float *y, long y_stride, const float *resid, long resid_stride, hipStream_t st) {
    GemvSetArgs a = set_args(w, qa_A, qa_d, N);
    a.y = y; a.y_stride = y_stride; a.resid = resid; a.resid_stride = resid_stride;
    return launch_set_any(w, a, epi, st);
}
hipError_t launch_gemv_set_rope_kv(const QMat &wqkv, const uint32_t *qa_A, const float *qa_d, int N, const RopeKvArgs &ra, hipStream_t st) {
    GemvSetArgs a = set_args(wqkv, qa_A, qa_d, N);
    a.ra = ra;
    return launch_set_any(wqkv, a, EPI_ROPE_KV, st);
}
hipError_t launch_gemv_set_silu(const QMat &w13, const uint32_t *qa_A, const float *qa_d, int N, const uint16_t *T_silu,
                                uint32_t *out_A, float *out_d, long out_strideA, long out_strideD, const SiluHalfIO &hx, hipStream_t st) {
    static const int fault_test = (getenv("LLAMAHIP_HANDOFF_FAULT_TEST") && atoi(getenv("LLAMAHIP_HANDOFF_FAULT_TEST")) == 7) ? 0x1000 : 0;
    GemvSetArgs a = set_args(w13, qa_A, qa_d, N);
    a.T_silu = T_silu; a.out_A = out_A; a.out_d = out_d; a.out_strideA = out_strideA; a.out_strideD = out_strideD;
    a.amax_t = hx.amax_t; a.epoch = hx.epoch; a.layer = hx.layer; a.fault = hx.fault; a.lut_math = g_lut_math | fault_test;
    return launch_set_any(w13, a, EPI_SILU_QAH, st);
}

hipError_t init_attrs_gemv_set() {
    const int cap = (int) SET_LDS_CAP;
#define LH_ATTR1(NCV, CWV, E) do { hipError_t e_ = hipFuncSetAttribute((const void *) k_gemv_set<NCV, CWV, E>, hipFuncAttributeMaxDynamicSharedMemorySize, cap); if (e_ != hipSuccess) return e_; } while (0)
#define LH_ATTR(NCV, CWV) do { LH_ATTR1(NCV, CWV, EPI_STORE); LH_ATTR1(NCV, CWV, EPI_RESID); LH_ATTR1(NCV, CWV, EPI_ROPE_KV); LH_ATTR1(NCV, CWV, EPI_SILU_QAH); } while (0)
    LH_ATTR(1, 2); LH_ATTR(1, 3); LH_ATTR(1, 4);
    LH_ATTR(2, 1); LH_ATTR(2, 2); LH_ATTR(2, 3); LH_ATTR(2, 4);
    LH_ATTR(3, 1); LH_ATTR(3, 3); LH_ATTR(3, 4);
    LH_ATTR(4, 1); LH_ATTR(4, 2); LH_ATTR(4, 4);
#undef LH_ATTR
#undef LH_ATTR1
    return hipSuccess;
}

}  // namespace lh
