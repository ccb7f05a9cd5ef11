// gemv_set.hip -- the Q4_0 x Q4_0 mat-mul for a FEW activation rows (2 .. 16): the rows of a batched decode step
// (llamahip_stage_step_set: one row per sequence) and of the reference's own prompt flow (nine tokens per llama_eval,
// .mm:880-888).  Same arithmetic and order as k_gemv (ggml_compute_forward_mul_mat_q4_0_f32, ggml.c:5987-6285, N > 1: 6134-6152,
// 6182-6222; vec_dot ggml.c:1415-1466), bit-exact.
//
// Why another kernel.  k_gemm_skinny (rounds 1-4, removed in round 5) gave a wave one row-group (8 rows x 8 chains) and <= 4 columns through a 4-deep REGISTER ring, staged
// the operand rows behind that ring and drained everything before its loop.  What the in-kernel timelines of this file's first versions
// showed (profiles/r05_*timeline*.txt, tools/set_timeline.py) and what the kernel does about it:
//   * a wave spends ~0.08 us per (chunk, column) item whatever shares its SIMD (21 VALU + 3 LDS reads at ~9 cycles per instruction and
//     wave), so a launch ends when the wave with the most items does: the 4 096-row matrices (512 row-groups) want SEVERAL waves per
//     row-group with one or two columns each.  Those waves share the row-group's weight bytes through LDS:
//       workgroup = RGW row-groups x CW "column waves"; wave (rgi, ci) owns row-group blk * RGW + rgi and NC of the columns; the CW waves
//       of a row-group take turns fetching its chunks (chunk c by wave c % CW) into an LDS ring of SET_DR steps by LDS-DMA
//       (global_load_lds_dwordx4: no VGPRs, no ds_write), one workgroup barrier per step of CW chunks, then every wave runs the step's
//       chunks against its own columns.  Every weight byte crosses a CU's load path once.  CW = 1: the same ring, private, no barrier.
//   * with the ring in registers the compiler puts register copies in front of the loop that carries it; they wait for a wave's WHOLE
//     prefill (with CW x DR chunks in flight that is the far end of the row), and operand rows requested behind the ring arrive behind it
//     (vmcnt retires in order): 3 - 6.5 us between "loads issued" and step 0.  Now: epilogue operands first, operand rows next (LDS-DMA,
//     permuted into the LDS layout by per-lane source addresses), then the ring; one counted wait, the loop starts on the first chunk.
//   * the operand rows' LDS layout is [chunk][half][chain] -- eight chains x 16 B = all 32 banks once per read (chain-major put chains k
//     and k + 4 on the same banks) --, operands are requested SET_PF items ahead (1: three measured the same).
//   * matrices with >= 1 536 row-groups have waves enough without sharing: up to five columns per wave through k_gemv's REGISTER ring (CW = 1),
//     and for more rows column GROUPS at grid level (the groups of a row block sit on one XCD, 8 apart in dispatch order).
//   * from ~8 columns on a launch is VALU bound (22 instructions per (lane, chunk, column), 12 of them half rate: 57 ns of a SIMD per item),
//     and a shared ring's barrier per step is then pure loss: the small matrices take unshared column groups from 9 rows on (set_plan).
// LDS traffic per (wave, chunk): 1.25 KiB of weights (CW > 1) + NC x 288 B of operands.
// Measured (7B, MI355X, every figure from a fresh process, profiles/r05_w_fresh_ab_final_plan.txt): set step of 2 / 4 / 8 sequences 1.94 / 2.30 /
// 2.99 ms (k_gemm_skinny) -> 1.79 / 2.03 / 2.74 ms; evals of 4 / 9 / 10 / 16 tokens 2.39 / 3.31 / 3.86 / 4.34 -> 2.16 / 3.16 / 3.18 / 4.14 ms;
// bit-identical.  HBM traffic 1.05 - 1.10 x algorithmic per launch (profiles/r05_final_set_pmc_S4.txt, _S8.txt).
//
// Epilogues (per column):
//   EPI_STORE / EPI_RESID   y = acc (+ resid)                                             lm head; wo, w2 (.mm:649-654, 682-687)
//   EPI_ROPE_KV             wq|wk|wv: RoPE of q and k, append of k and v at the row's own position (k_rope_kv's arithmetic;
//                           ggml.c:7076-7131, .mm:586-611)
//   EPI_SILU_QAH            interleaved w1|w3 in HALF-block workgroups (RGW = 4: 16 gate rows + the same 16 up rows): SiLU * up
//                           (ggml.c:1956-1963, .mm:678-680) and the Q4_0 quantization of the block for w2 (ggml.c:456-523); the two
//                           halves of a block exchange their partial amax per column as tagged granules inside one XCD's L2 (k_gemv's
//                           EPI_SILU_QAH: fmaxf is exact in any order).  Half blocks because whole blocks are F / 32 = 344 workgroups
//                           on 256 CUs at 7B.
//   EPI_SILU_QA             the same in WHOLE-block workgroups (RGW = 8: the block's four gate row-groups + its four up row-groups, as
//                           k_gemv's 8-wave w1|w3 launch): no exchange.  Taken from two column groups on (6+ rows): the groups double
//                           the workgroups, so the CU balance that asked for halves is there anyway, and with more columns per step a
//                           half-block workgroup waited 4 - 11 us for its partner (profiles/r05_final_set_timeline.txt: epilogue 1.6 /
//                           3.8 / 10.7 us at 4 / 8 / 9 rows).
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
    int ncols, rgw, ncg;                             // rgw: row-groups per workgroup (waves = rgw * CW); ncg: column groups of NC * CW columns (grid level)
    float *y; long y_stride; const float *resid; long resid_stride;
    const uint16_t *T_silu; uint32_t *out_A; float *out_d; long out_strideA, out_strideD;
    RopeKvArgs ra;
    uint64_t *amax_t; const uint32_t *epoch; int layer; uint32_t *fault; int lut_math;      // EPI_SILU_QAH (lut_math as GemvArgs: bit 0 SiLU evaluated, 0x1000 fault-injection test)
    unsigned long long *probe;                      // LH_SET_PROBE builds: [0] record counter, [1] capacity, records of 32 words from [32]
};

// LH_SET_ABLATE (measurement builds only, tools/build_set_variants.sh; results are wrong): 1 = no arithmetic (what the launch shape streams),
// 2 = no weight loads inside the loop (arithmetic, LDS traffic and barriers alone)
#ifndef LH_SET_ABLATE
#define LH_SET_ABLATE 0
#endif
// LH_SET_PROBE (measurement build, tools/set_timeline.py): wave 0 of every workgroup stamps s_memtime at the phase boundaries and inside its
// first steps into a record of 32 words: [0] wall clock at entry | [1] entry | [2] loads issued | [3] operands staged | [4 + 3 s + {0, 1, 2}] step s
// (s < 4): start, its chunk has landed, behind the barrier (the items end where the next step starts) | [24] loop done | [25] exit | [26] wall clock at exit | [27] id
#ifndef LH_SET_PROBE
#define LH_SET_PROBE 0
#endif
#if LH_SET_PROBE
#define LH_PSTAMP(IDX) do { if (probe_on) probe_t[IDX] = __builtin_readcyclecounter(); } while (0)      /* probe_t: LDS, behind the ring */
#else
#define LH_PSTAMP(IDX) do { } while (0)
#endif
// The weight ring lives in LDS and is filled by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B -> 1 KiB of LDS, no VGPRs, no ds_write):
// SET_DR slots of CW chunks per row-group, SET_DR - 1 steps in flight.  (The first version kept the ring in registers and published
// every chunk with a ds_write.  Its in-kernel timeline, profiles/r05_t_timeline_register_ring.txt: 3.0 - 6.5 us between "ring issued" and the first
// step -- the operand rows were requested BEHIND the ring and vmcnt retires in order, and the register copies the compiler places in
// front of a loop that carries a ring made every wave wait for ALL of its prefill, i.e. for the far end of its rows, before step 0.)
#ifndef LH_SET_DR
#define LH_SET_DR 4
#endif
constexpr int SET_DR = LH_SET_DR;
// LDS operands of an item (chunk, column) are requested SET_PF items ahead into a ring of SET_PF + 1 register buffers.  Three items ahead
// measured the same as one (profiles/r05_c_pf.txt: the loop is not LDS-latency bound) and costs 20 VGPRs: one.
#ifndef LH_SET_PF
#define LH_SET_PF 1
#endif
constexpr int SET_PF = LH_SET_PF, SET_NB = SET_PF + 1;
static_assert((SET_DR % SET_NB) == 0 || SET_NB == 2, "the operand-buffer index is the item's position in the unrolled block of SET_DR steps");

// one wave instruction of LDS-DMA: lane l's 16 (4) bytes at `base + voff` land in LDS at `lds_dst + 16 l` (`+ 4 l`); lds_dst, base wave-uniform
__device__ __forceinline__ void set_dma16(uint32_t lds_dst, uint64_t base, uint32_t voff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(base) : "memory");
}
__device__ __forceinline__ void set_dma16_cached(uint32_t lds_dst, uint64_t base, uint32_t voff) {      // operand rows: every workgroup reads them
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(base) : "memory");
}
__device__ __forceinline__ void set_dma4(uint32_t lds_dst, uint64_t base, uint32_t voff, bool nt) {
    uint32_t keep;
    if (nt) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3 nt\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(base) : "memory");
    else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                      : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(base) : "memory");
}
template <int N> __device__ __forceinline__ void set_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// the epilogues of k_gemv_set / k_gemv_set_ar (file header): accs[n] = the lane's chain sums of the wave's n-th column; NCW = columns per
// workgroup; resid_v / rope_pos / rope_kvo were requested in the prologue
struct SetEpiCtx { int g, blk, rgi, ncol0, col0, ncols, nw, wave, lane, qah_block, qah_half; bool valid; uint32_t store_tag; };
template <int NC, int NCW, int EPI>
__device__ __forceinline__ void set_epilogue(const GemvSetArgs &a, double *smem_d, float (&accs)[NC], const SetEpiCtx &x,
                                             const float (&resid_v)[NC], const int (&rope_pos)[NC], const long (&rope_kvo)[NC]) {
    const int g = x.g, rgi = x.rgi, ncol0 = x.ncol0, col0 = x.col0, ncols = x.ncols, nw = x.nw, wave = x.wave, lane = x.lane;
    const int qah_block = x.qah_block, qah_half = x.qah_half, k = lane & 7;
    const bool valid = x.valid;
    const uint32_t store_tag = x.store_tag;
    int lg = g;
    if (a.gmapF8) { const int b8 = g >> 3, w8 = g & 7; lg = w8 < 4 ? b8 * 4 + w8 : a.gmapF8 + b8 * 4 + (w8 - 4); }
    const int m = lg * 8 + (lane >> 3);
    if (EPI == EPI_SILU_QA) {
        // whole block: gu[column][gate 0 .. 31 | up 0 .. 31] (row-groups 0 - 3 gate, 4 - 7 up); k_gemv's EPI_SILU_QA per column
        float *gu = (float *) smem_d;
        __syncthreads();
#pragma unroll
        for (int n = 0; n < NC; n++) {
            const float acc = fold8(accs[n]);
            if (k == 0) gu[(ncol0 + n) * 64 + rgi * 8 + (lane >> 3)] = acc;
        }
        __syncthreads();
        for (int col = wave; col < NCW; col += nw) {
            if (col >= ncols || x.blk * 8 >= a.ngroups) continue;
            const int i = lane & 31;
            const uint16_t gh = f2h_bits(gu[col * 64 + i]);
            const float act = h2f_bits((a.lut_math & 1) ? silu_math_bits(gh) : a.T_silu[gh]) * gu[col * 64 + 32 + i];
            const float amax = wave_max_f(fabsf(act));                         // (lanes 32 .. 63 repeat lanes 0 .. 31)
            const float dd = amax / 7.0f;                                      // ggml.c:479
            const float id = (amax != 0.0f) ? 7.0f / amax : 0.0f;              // ggml.c:482
            const uint32_t nib = (uint32_t) ((int) __builtin_rintf(act * id)) & 0xF;      // signed nibble of (q - 8)
            const int kk = lane & 7;
            const uint32_t e0 = __shfl(nib, 2 * kk), e1 = __shfl(nib, 2 * kk + 1);
            const uint32_t e2 = __shfl(nib, 16 + 2 * kk), e3 = __shfl(nib, 17 + 2 * kk);
            const int gcol = col0 + col, b = x.blk, c = b >> 3, j = b & 7;
            if (lane < 8) a.out_A[(size_t) gcol * a.out_strideA + (c * 8 + kk) * 8 + j] = (e0 | (e1 << 8) | (e2 << 16) | (e3 << 24)) << (4 * (j & 1));
            if (lane == 0) a.out_d[(size_t) gcol * a.out_strideD + b] = dd;
        }
        return;
    }
    if (EPI == EPI_SILU_QAH) {
        // the workgroup's 32 outputs per column: gu[column][gate 0 .. 15 | up 0 .. 15]; the operand area is free again
        float *gu = (float *) smem_d;
        __syncthreads();
#pragma unroll
        for (int n = 0; n < NC; n++) {
            const float acc = fold8(accs[n]);
            if (k == 0) gu[(ncol0 + n) * 32 + rgi * 8 + (lane >> 3)] = acc;
        }
        __syncthreads();
        const bool inject = (a.lut_math & 0x1000) != 0;                       // fault-injection test: a tag nobody waits for, short polls
        for (int col = wave; col < NCW; col += nw) {
            if (col >= ncols || qah_block * 8 >= a.ngroups) continue;
            const int i = lane & 15;
            const uint16_t gh = f2h_bits(gu[col * 32 + i]);
            const float act = h2f_bits((a.lut_math & 1) ? silu_math_bits(gh) : a.T_silu[gh]) * gu[col * 32 + 16 + i];
            float amax = wave_max_f(lane < 16 ? fabsf(act) : 0.0f);
            // the other half's partial amax of this column: one tagged granule each way inside this XCD's L2
            const int gcol = col0 + col;
            uint64_t *at = a.amax_t + (size_t) gcol * ((size_t) (a.ngroups / 8) * 2 + 16);
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
            uint32_t *oA = a.out_A + (size_t) gcol * a.out_strideA;
            if (lane < 8) ((uint16_t *) (oA + (c * 8 + kk) * 8 + j))[qah_half] = (uint16_t) ((e0 | (e1 << 8)) << (4 * (j & 1)));
            if (lane == 0 && qah_half == 0) a.out_d[(size_t) gcol * a.out_strideD + b] = dd;
        }
        return;
    }
    if (EPI == EPI_ROPE_KV) {
        // (ggml.c:7076-7131, .mm:586-611; k_rope_kv) rows m, m ^ 1 = lanes 8 apart; m is even iff the lane's row is.  The table entries of
        // all columns are requested together (one round trip), then the rotations
        const RopeKvArgs &ra = a.ra;
        const int which = m / ra.d, c = m - which * ra.d, pe = (c % ra.dh) & ~1;
        const bool live = valid && k == 0 && m < a.M;
        double cs[NC], sn[NC];
#pragma unroll
        for (int n = 0; n < NC; n++) {
            cs[n] = 1.0; sn[n] = 0.0;
            if (live && which != 2) { cs[n] = ra.tab[(size_t) rope_pos[n] * ra.dh + pe]; sn[n] = ra.tab[(size_t) rope_pos[n] * ra.dh + pe + 1]; }
        }
#pragma unroll
        for (int n = 0; n < NC; n++) {
            const float acc = fold8(accs[n]);
            const int col = ncol0 + n;
            const float up = dpp_f<0x108>(acc), dn = dpp_f<0x118>(acc);        // row_shl:8 / row_shr:8
            if (live && col < ncols) {
                const int pos = rope_pos[n];
                const long kvo = rope_kvo[n];
                if (which == 2) {
                    ra.Vc[kvo + (size_t) pos * ra.d + c] = acc;
                } else {
                    const double x0 = (double) ((c & 1) ? dn : acc), x1 = (double) ((c & 1) ? acc : up);
                    const float val = (c & 1) ? (float) (x0 * sn[n] + x1 * cs[n]) : (float) (x0 * cs[n] - x1 * sn[n]);
                    if (which == 0) ra.qr[(size_t) (col0 + col) * ra.d + c] = val;
                    else ra.Kc[kvo + (size_t) pos * ra.d + c] = val;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < NC; n++) {
        float acc = fold8(accs[n]);
        const int col = ncol0 + n;
        if (valid && k == 0 && m < a.M && col < ncols) {
            if (EPI == EPI_RESID) acc = acc + resid_v[n];
            a.y[(size_t) (col0 + col) * a.y_stride + m] = acc;
        }
    }
}

// threads a launch may ask for: half-block workgroups are 4 row-groups x CW waves, the others at most 8 waves
template <int CW, int EPI> constexpr int set_max_threads() { return EPI == EPI_SILU_QAH ? (CW * 256 > 512 ? CW * 256 : 512) : 512; }

template <int NC, int CW, int EPI>
__global__ void __launch_bounds__((set_max_threads<CW, EPI>()))
k_gemv_set(const GemvSetArgs a) {
    constexpr int DR = SET_DR, NCW = NC * CW;
    constexpr bool SHARE = CW > 1;
    extern __shared__ double smem_d[];
    // grid: blockIdx -> (row block `blk`, column group `cg`); the column groups of a row block sit on one XCD, 8 apart in dispatch order
    // (their weight tiles meet in that XCD's L2).  One group (ncg = 1): blk = blockIdx.
    const int bid = blockIdx.x, cg = (bid >> 3) % a.ncg, blk = ((bid >> 3) / a.ncg) * 8 + (bid & 7);
    const int col0 = cg * NCW;
    const int nchunks = a.nchunks, ncols = min(NCW, a.ncols - col0), rgw = a.rgw;      // this group's columns: col0 .. col0 + ncols - 1
    if (ncols <= 0) return;                                         // (a plan never has an empty group -- set_plan; uniform per workgroup, before any barrier)
    const uint32_t *qa_A = a.qa_A + (size_t) col0 * nchunks * 64;
    const float *qa_d = a.qa_d + (size_t) col0 * nchunks * 8;
    const int steps = (nchunks + CW - 1) / CW;
    const int npad = steps * CW + SET_PF;                           // chunks per column in LDS: the step grid + the operand fetches SET_PF items ahead
    u32x4 *sA = (u32x4 *) smem_d;                                   // [ncols][npad][16]: a chunk is [half][chain] (the eight chains' 16-byte reads of one half are 128 contiguous bytes: all 32 banks once)
    f32x2 *sD = (f32x2 *) (sA + (size_t) ncols * npad * 16);        // [ncols][npad][4]: {d[t], d[t + 4]} -- lane t of a quad owns blocks t and t + 4
    uint8_t *ring = (uint8_t *) (sD + (size_t) ncols * npad * 4);   // (CW > 1) [rgw][DR][CW][TILE_BYTES]: slot s % DR of a row-group holds the CW chunks of step s
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = (int) (blockDim.x >> 6), nt = (int) blockDim.x;
    const int rgi = wave / CW, ci = wave - rgi * CW;
    // (EPI_SILU_QAH: workgroup blk = half `(blk >> 3) & 1` of activation block `(blk >> 4) * 8 + (blk & 7)` -- the halves of a block are 8
    //  apart in the grid, one XCD; rgi 0, 1 = the half's two gate row-groups, rgi 2, 3 = the matching up row-groups of the interleaved order)
    const int qah_block = (blk >> 4) * 8 + (blk & 7), qah_half = (blk >> 3) & 1;
    const int g = EPI == EPI_SILU_QAH ? qah_block * 8 + (rgi >> 1) * 4 + qah_half * 2 + (rgi & 1) : blk * rgw + rgi;
    const bool valid = g < a.ngroups;
    const uint64_t wbase = (uint64_t) (uintptr_t) (a.wt + (size_t) (valid ? g : a.ngroups - 1) * (nchunks + 1) * TILE_BYTES);
    const int k = lane & 7, t = lane & 3;
    const int woff = lane * 16, soff = 1024 + ((lane >> 3) * 8 + t * 2) * 4;
    const uint32_t store_tag = EPI == EPI_SILU_QAH ? make_tag(__builtin_nontemporal_load(a.epoch), a.layer + 1) : 0u;
    const uint32_t ring_lds = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint8_t *) ring + (uint32_t) rgi * (DR * CW * TILE_BYTES);
    const uint8_t *ring_rg = ring + (size_t) rgi * (DR * CW * TILE_BYTES);
#if LH_SET_PROBE
    const bool probe_on = a.probe != nullptr && tid == 0;
    unsigned long long *probe_t = (unsigned long long *) (ring + (SHARE ? (size_t) rgw * DR * CW * TILE_BYTES : 0));
    if (probe_on) { for (int i = 0; i < 28; i++) probe_t[i] = 0; probe_t[0] = wall_clock64(); }
#endif
    LH_PSTAMP(1);
    // this wave's chunk of step S goes into slot (S % DR, ci): two DMA instructions (1 KiB of nibbles, 256 B of scales); chunks past the row
    // end read the zero tile that closes every row-group (scale 0 -> fma(0 * d_a, p, acc) == acc)
#define LH_DMAW(SLOTI, S)                                                                          \
    {                                                                                              \
        const uint64_t tp_ = wbase + (uint64_t) min((S) * CW + ci, nchunks) * TILE_BYTES;          \
        const uint32_t dst_ = ring_lds + (uint32_t) (((SLOTI) * CW + ci) * TILE_BYTES);            \
        set_dma16(dst_, tp_, (uint32_t) woff);                                                     \
        set_dma4(dst_ + 1024u, tp_ + 1024u, (uint32_t) lane * 4u, true);                           \
    }
    // CW = 1: nobody shares this wave's weights, so its ring stays in REGISTERS (non-temporal loads, counted waits placed by the compiler) and
    // LDS holds the operand rows only: occupancy is then bounded by registers (5 waves per SIMD), not by 5 KiB of LDS ring per wave -- the
    // many small workgroups of the large matrices (w1|w3 at 8 rows: 1 376) fit the chip in one round instead of two.  (The register copies in
    // front of the loop wait for chunks 0 .. DR - 1 here, consecutive chunks that arrive together -- not for the far end of the row.)
    u32x4 wq[DR];
    f32x2 ws[DR];
#define LH_LOADW(SLOT, CH)                                                                         \
    {                                                                                              \
        const uint8_t *tp_ = (const uint8_t *) (uintptr_t) wbase + (size_t) min((CH), nchunks) * TILE_BYTES;      \
        wq[SLOT] = __builtin_nontemporal_load((const u32x4 *) (tp_ + woff));                       \
        ws[SLOT] = __builtin_nontemporal_load((const f32x2 *) (tp_ + soff));                       \
    }
    // ---- prologue.  vmcnt retires in order, so the order of issue is the order of arrival: the epilogue's operands (residual values, the
    // rows' positions) first, then the operand rows of all columns (LDS-DMA, permuted into the LDS layout by per-lane source addresses),
    // then the first DR - 1 steps of the weight ring; ONE counted wait leaves the ring in flight and the loop starts on the first chunk.
    float resid_v[NC];
    int rope_pos[NC];
    long rope_kvo[NC];
#pragma unroll
    for (int n = 0; n < NC; n++) { resid_v[n] = 0.0f; rope_pos[n] = 0; rope_kvo[n] = 0; }
    {
        int lg0 = g;
        if (a.gmapF8) { const int b8 = g >> 3, w8 = g & 7; lg0 = w8 < 4 ? b8 * 4 + w8 : a.gmapF8 + b8 * 4 + (w8 - 4); }
        const int m0 = min(lg0 * 8 + (lane >> 3), a.M - 1);
#pragma unroll
        for (int n = 0; n < NC; n++) {
            const int col = col0 + min(ci * NC + n, ncols - 1);
            if (EPI == EPI_RESID) resid_v[n] = a.resid[(size_t) col * a.resid_stride + m0];
            if (EPI == EPI_ROPE_KV) {
                // (batched decode step: the row's own position -- gathered into the descriptor by the step's first launch -- and cache)
                rope_pos[n] = a.ra.set ? a.ra.set->pos[col] : a.ra.n_past + col;      // (col: the global column)
                rope_kvo[n] = a.ra.set ? a.ra.set->kv_off[col] : 0L;
            }
        }
    }
    {
        // operand rows: LDS granule L of a column = (chunk L >> 4, half (L >> 3) & 1, chain L & 7)  <-  QA granule (chunk, chain, half);
        // d pairs: LDS float f of a chunk = (t = f >> 1, h = f & 1)  <-  d[4 h + t].  A wave instruction fills 64 consecutive LDS granules
        // (floats) of one column: the permutation is a per-lane constant, everything else is wave-uniform (scalar); the instructions are
        // dealt to the waves round-robin (lanes past a row's end are masked).
        const int perA = nchunks * 16, perDf = nchunks * 8;
        const int ipcA = (perA + 63) >> 6, ipcD = (perDf + 63) >> 6;
        const uint32_t sA_lds = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint8_t *) (uint8_t *) sA;
        const uint32_t sD_lds = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint8_t *) (uint8_t *) sD;
        const uint32_t laneA = (uint32_t) (((lane & ~15) + (lane & 7) * 2 + ((lane >> 3) & 1)) * 16);
        const uint32_t laneD = (uint32_t) (((lane & ~7) + (lane & 1) * 4 + ((lane >> 1) & 3)) * 4);
        int n = wave / ipcA, b = wave - n * ipcA;
        for (int q = wave; q < ncols * ipcA; q += nw) {
            if (b * 64 + lane < perA)
                set_dma16_cached(sA_lds + (uint32_t) ((n * npad * 16 + b * 64) * 16), (uint64_t) (uintptr_t) qa_A + ((uint64_t) n * perA + (uint64_t) b * 64) * 16, laneA);
            b += nw;
            while (b >= ipcA) { b -= ipcA; n++; }
        }
        n = wave / ipcD; b = wave - n * ipcD;
        for (int q = wave; q < ncols * ipcD; q += nw) {
            if (b * 64 + lane < perDf)
                set_dma4(sD_lds + (uint32_t) ((n * npad * 8 + b * 64) * 4), (uint64_t) (uintptr_t) qa_d + ((uint64_t) n * perDf + (uint64_t) b * 64) * 4, laneD, false);
            b += nw;
            while (b >= ipcD) { b -= ipcD; n++; }
        }
    }
    if (SHARE) {
#pragma unroll
        for (int i = 0; i < DR - 1; i++) LH_DMAW(i, i)
    } else {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < DR; i++) LH_LOADW(i, i)
        __builtin_amdgcn_sched_barrier(0);
    }
    LH_PSTAMP(2);
    {
        const int zc = npad - nchunks;                             // zeroed chunks behind every column: A 16 granules, d 8 floats each
        for (int i = tid; i < ncols * zc * 18; i += nt) {
            const int n = i / (zc * 18), r = i - n * (zc * 18);
            if (r < zc * 16) sA[((size_t) n * npad + nchunks) * 16 + r] = u32x4{ 0u, 0u, 0u, 0u };
            else ((f32x4 *) sD)[((size_t) n * npad + nchunks) * 2 + (r - zc * 16)] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
        }
    }
    if (SHARE) set_wait_vmcnt<2 * (DR - 1)>();                     // everything older than the ring: this wave's share of the operand rows
    else set_wait_vmcnt<2 * DR>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... and everybody else's (a __syncthreads() would wait for the ring as well)
    LH_PSTAMP(3);

    float accs[NC];
#pragma unroll
    for (int n = 0; n < NC; n++) accs[n] = 0.0f;
    const int ncol0 = ci * NC;
    // (only the ncols real columns are staged; a wave's columns past them are clamped duplicates of the last one)
    int colofs[NC];
#pragma unroll
    for (int n = 0; n < NC; n++) colofs[n] = min(ncol0 + n, ncols - 1) * npad;
    u32x4 la0[SET_NB], la1[SET_NB], wb[2];
    f32x2 ldd[SET_NB], wsb[2];
#define LH_LDSA(BUF, N, CH)                                                                        \
    {                                                                                              \
        const u32x4 *pa_ = sA + (colofs[N] + (CH)) * 16 + k;                                       \
        la0[BUF] = pa_[0]; la1[BUF] = pa_[8];                                                      \
        ldd[BUF] = sD[(colofs[N] + (CH)) * 4 + t];                                                 \
    }
#define LH_LDSW(BUF, SLOTI, C)                                                                     \
    {                                                                                              \
        const uint8_t *sp_ = ring_rg + (size_t) (((SLOTI) * CW) + (C)) * TILE_BYTES;               \
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
        if (LH_SET_ABLATE != 1) LH_FMAC8_DPP(ACC, plo_, phi_, q01_, q23_, q45_, q67_);              \
    }
    // step S (I = S % DR, compile time).  CW > 1: this wave's chunk of the step is DR - 1 steps old: wait for it (counted: the DR - 2 younger
    // steps stay in flight), then one workgroup barrier: every wave's chunk of step S has landed AND every wave is done reading step S - 1,
    // whose slot the DMA for step S + DR - 1 may now overwrite; then the CW chunks of the step in block order against this wave's NC
    // columns.  CW = 1: chunk S from ring register I, refilled with chunk S + DR behind its last use.
#define LH_SSTEP(I, S)                                                                             \
    {                                                                                              \
        if (s0 == 0) LH_PSTAMP(4 + 3 * (I));                                                       \
        if (SHARE) {                                                                               \
            set_wait_vmcnt<2 * (DR - 2)>();                                                        \
            if (s0 == 0) LH_PSTAMP(5 + 3 * (I));                                                   \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                        \
            if (s0 == 0) LH_PSTAMP(6 + 3 * (I));                                                   \
            if (LH_SET_ABLATE != 2) LH_DMAW(((I) + DR - 1) % DR, (S) + DR - 1)                     \
            LH_LDSW(0, (I), 0)                                                                     \
        }                                                                                          \
        _Pragma("unroll")                                                                          \
        for (int c = 0; c < CW; c++) {                                                             \
            if (SHARE && c + 1 < CW) LH_LDSW((c + 1) & 1, (I), c + 1)                              \
            _Pragma("unroll")                                                                      \
            for (int n = 0; n < NC; n++) {                                                         \
                const int it_ = (I) * NCW + c * NC + n;        /* item index in the unrolled block */   \
                const int pb_ = it_ % SET_NB;                                                      \
                LH_LDSA((it_ + SET_PF) % SET_NB, (n + SET_PF) % NC, (S) * CW + c + (n + SET_PF) / NC) \
                __builtin_amdgcn_sched_barrier(0);      /* the reads for the items ahead go out before this item's arithmetic */ \
                if (SHARE) LH_ITEM(wb[c & 1], wsb[c & 1], pb_, accs[n])                            \
                else LH_ITEM(wq[I], ws[I], pb_, accs[n])                                           \
                __builtin_amdgcn_sched_barrier(0);                                                 \
            }                                                                                      \
        }                                                                                          \
        if (!SHARE && LH_SET_ABLATE != 2) LH_LOADW(I, (S) + DR)                                    \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
#pragma unroll
    for (int i = 0; i < SET_PF; i++) LH_LDSA(i, i % NC, i / NC)
    int s0 = 0;
    for (; s0 + DR <= steps; s0 += DR) {
#pragma unroll
        for (int i = 0; i < DR; i++) LH_SSTEP(i, s0 + i)
    }
    // the last steps % DR steps (the same unrolled sequence, cut short; `steps` is uniform over the workgroup)
#pragma unroll
    for (int i = 0; i < DR - 1; i++)
        if (s0 + i < steps) LH_SSTEP(i, s0 + i)
    set_wait_vmcnt<0>();                                           // (the ring's last requests are zero tiles: nothing may still be landing in LDS below)
    LH_PSTAMP(24);
#undef LH_SSTEP
#undef LH_ITEM
#undef LH_LDSW
#undef LH_LDSA
#undef LH_DMAW
#undef LH_LOADW

#if LH_SET_PROBE
#define LH_PFLUSH() do { if (probe_on) { probe_t[25] = __builtin_readcyclecounter(); probe_t[26] = wall_clock64();                       \
        probe_t[27] = ((unsigned long long) NC << 56) | ((unsigned long long) CW << 48) | ((unsigned long long) EPI << 40) | ((unsigned long long) nchunks << 24) | (unsigned) blk; \
        const unsigned long long slot_ = atomicAdd(a.probe, 1ull);                                                                      \
        if (slot_ < a.probe[1]) { unsigned long long *e_ = a.probe + 32 * (1 + slot_); for (int i_ = 0; i_ < 28; i_++) e_[i_] = probe_t[i_]; } } } while (0)
#else
#define LH_PFLUSH() do { } while (0)
#endif
    {
        const SetEpiCtx ex = { g, blk, rgi, ncol0, col0, ncols, nw, wave, lane, qah_block, qah_half, valid, store_tag };
        set_epilogue<NC, NCW, EPI>(a, smem_d, accs, ex, resid_v, rope_pos, rope_kvo);
    }
    LH_PFLUSH();
}

// ------------------------------------------------------------------------------------------------
// host side: the (NC, CW) plan per row count, launchers
// ------------------------------------------------------------------------------------------------
struct SetPlan { int nc = 0, cw = 0, rgw = 0, ncg = 1; size_t lds = 0; };

static size_t set_lds_bytes(const QMat &w, int ncols_group, int nc, int cw, int rgw) {
    const int steps = (w.nchunks + cw - 1) / cw, npad = steps * cw + SET_PF;
    size_t lds = (size_t) ncols_group * npad * 288 + (cw > 1 ? (size_t) rgw * SET_DR * cw * TILE_BYTES : 0) + (LH_SET_PROBE ? 256 : 0);
    return std::max(lds, (size_t) nc * cw * 64 * 4);
}

// Columns per wave (nc), waves per row-group (cw) and column groups (ncg) for N rows.  What the in-kernel timelines say (profiles/r05_*):
// a wave spends ~0.07 - 0.09 us per (chunk, column) item whatever shares its SIMD, a step of a shared ring costs a barrier (~0.3 us in a
// 16-wave workgroup), and a launch ends when its longest wave does.  So: matrices with few row-groups (wo, w2: 512 at 7B) take cw waves per
// row-group with one or two columns each (items per wave = chunks x nc); matrices with >= 1 024 row-groups (wq|wk|wv, w1|w3, the lm head)
// have waves enough: up to four columns per wave, no sharing, and for more than four rows column GROUPS at grid level (the groups of a
// row block run side by side on one XCD; the second read of a tile is an L2 hit hidden behind arithmetic -- at 8 rows these launches
// are VALU bound: sharing the ring there AND streaming the operand rows through a second ring, k_gemv_set_ar of round 5, read the weights once
// and measured the same -- profiles/r05_o_operand_ring_ab.txt; removed).  LLAMAHIP_SET_PLAN[_BIG|_SMALL]="nc,cw[,ncg[,rgw]]" overrides (measurement; _BIG: the unshared class).
// the short evals: up to 60 rows (measured crossover against the row-per-lane kernel at 7B shapes, round 1: +25 % at 33 rows, +7 % at 56, -1 % at
// 63; k_gemv_set against k_gemm_skinny at 20 / 24 / 32 / 48 / 60 rows: -5.5 / -5.5 / -0.7 / -5.2 / -6.3 % per eval, profiles/r05_y_rows_max.txt)
constexpr int SET_ROWS_MAX = 60;
constexpr size_t SET_LDS_CAP = 160 * 1024;
// the (NC, CW) pairs k_gemv_set is instantiated for (launch_set_any's dispatch list)
static bool set_plan_instantiated(int nc, int cw) {
    static const int pairs[][2] = { {1, 2}, {1, 3}, {1, 4}, {2, 1}, {2, 2}, {2, 3}, {2, 4}, {3, 1}, {3, 3}, {3, 4}, {4, 1}, {4, 2}, {4, 4}, {5, 1} };
    for (const auto &pr : pairs) if (pr[0] == nc && pr[1] == cw) return true;
    return false;
}

// LLAMAHIP_SET_PLAN[_BIG|_SMALL] are parsed once per process (the switches of this library are read once; set_plan runs several times per launch)
struct SetPlanEnv { int got = 0, nc = 0, cw = 0, ncg = 0, rgw = 0; };
static SetPlanEnv set_plan_env_parse(const char *name) {
    SetPlanEnv e;
    const char *env = getenv(name);
    if (!env) return e;
    e.got = sscanf(env, "%d,%d,%d,%d", &e.nc, &e.cw, &e.ncg, &e.rgw);
    if (e.got < 2 || e.nc < 1 || e.nc > 5 || e.cw < 1 || e.cw > 4 || (e.nc == 5 && e.cw != 1)) e.got = 0;
    return e;
}
static bool set_plan_env(int which, int N, int epi, int &nc, int &cw, int &ncg, int &rgw) {
    static const SetPlanEnv envs[3] = { set_plan_env_parse("LLAMAHIP_SET_PLAN_BIG"), set_plan_env_parse("LLAMAHIP_SET_PLAN_SMALL"), set_plan_env_parse("LLAMAHIP_SET_PLAN") };
    const SetPlanEnv &e = envs[which];
    if (e.got < 2) return false;
    int e_ncg = e.ncg;
    if (e.got < 3 || e_ncg < 1) e_ncg = (N + e.nc * e.cw - 1) / (e.nc * e.cw);
    if (e.nc * e.cw * e_ncg < N || (e_ncg - 1) * e.nc * e.cw >= N) return false;
    nc = e.nc; cw = e.cw; ncg = e_ncg;
    if (e.got >= 4 && e.rgw >= 1 && e.rgw <= 8 && epi != EPI_SILU_QAH && epi != EPI_SILU_QA) rgw = e.rgw;
    return true;
}
static SetPlan set_plan(const QMat &w, int N, int epi) {
    SetPlan p;
    // waves per row-group: enough to put about two waves on every SIMD of the chip (ngroups x cw >= ~2 000), w1|w3 always unshared (its
    // half-block workgroups are many); then <= 4 columns per wave, column groups at grid level beyond 4 cw columns
    const bool big = w.ngroups >= 1536 || epi == EPI_SILU_QAH || epi == EPI_SILU_QA;
    int cw = big ? 1 : w.ngroups >= 768 ? 2 : 4;
    if (cw > N) cw = N;
    const int ncmax = cw == 1 ? 5 : 4;                                // (five columns per wave are instantiated for unshared rings only: 9 rows = 5 + 4)
    int ncg = (N + ncmax * cw - 1) / (ncmax * cw);
    const int ng = (N + ncg - 1) / ncg;                               // columns per group, balanced (9 rows, unshared: 3 + 3 + 3)
    int nc = (ng + cw - 1) / cw, rgw = 0;
    cw = (ng + nc - 1) / nc;
    // the small class by row count, measured (7B wo / w2, fresh processes, profiles/r05_u_fresh_ab.txt, r05_v_small_plans.txt): three
    // column-waves leave 16 chunks a ragged last step (5 - 6 rows: <2,4> -1.5 %); from 9 rows on the launches are VALU bound and a step's
    // barrier is pure loss: unshared rings in column groups of 3 (9 - 12 rows: -4 ... -6 % per eval) or 4 (13 - 16: -2 %)
    // (the 768 .. 1 535 row-group class -- wo of the 30B / 65B -- takes the same rule from 9 rows on: the argument does not depend on the size)
    if (!big && N >= 5) {
        if (N <= 8) { if (w.ngroups < 768) { nc = 2; cw = 4; ncg = 1; } }
        else if (N <= 12) { nc = 3; cw = 1; ncg = (N + 2) / 3; }
        else { nc = 4; cw = 1; ncg = (N + 3) / 4; }
    }
    (void) (set_plan_env(big ? 0 : 1, N, epi, nc, cw, ncg, rgw) || set_plan_env(2, N, epi, nc, cw, ncg, rgw));
    auto finish = [&]() {
        // (instantiated: <1,2> <1,3> <1,4> <2,1> <2,2> <2,3> <2,4> <3,1> <3,3> <3,4> <4,1> <4,2> <4,4> <5,1>)
        if (nc == 1 && cw == 1) nc = 2;
        if (nc == 3 && cw == 2) nc = 4;
        if (nc == 4 && cw == 3) cw = 4;
        ncg = (N + nc * cw - 1) / (nc * cw);                          // (the promotions above widen a group: no column group may start past row N)
        if (!rgw) rgw = epi == EPI_SILU_QAH ? 4 : (cw >= 3 ? 2 : 4);
        if (epi == EPI_SILU_QAH) rgw = 4;
        else if (epi == EPI_SILU_QA) rgw = 8;
        else if (rgw * cw > 8) rgw = 8 / cw;                          // (set_max_threads)
        p.nc = nc; p.cw = cw; p.rgw = rgw; p.ncg = ncg;
        p.lds = set_lds_bytes(w, std::min(N, nc * cw), nc, cw, rgw);
    };
    finish();
    if (p.lds > SET_LDS_CAP) {
        // (a forced plan that cannot fit degrades to this rule for that shape too: a measurement variant never sends a shape to the generic kernel)
        // wide rows (w2 of the 13B / 65B: 54 / 86 chunks): the operand rows of a whole group do not fit next to a shared ring.  Unshared
        // column groups of as many columns (<= 4) as leave two workgroups' worth of LDS per CU where possible
        const size_t per_col = set_lds_bytes(w, 1, 1, 1, 4);
        int fit = (int) std::min<size_t>(4, (SET_LDS_CAP / 2) / per_col);
        if (fit < 2) fit = (int) std::min<size_t>(4, SET_LDS_CAP / per_col);
        if (fit >= 1) {
            cw = 1; rgw = 0;
            ncg = (N + fit - 1) / fit;
            nc = (N + ncg - 1) / ncg;
            finish();
        }
    }
    return p;
}

bool gemv_set_applies(const QMat &w, int N, int epi) {
    if (N < 2 || N > SET_ROWS_MAX || !w.tiles) return false;
    if (epi == EPI_ROPE_KV && w.gmapF8 != 0) return false;
    if ((epi == EPI_SILU_QAH || epi == EPI_SILU_QA) && (w.gmapF8 == 0 || w.ngroups % 8 != 0)) return false;
    if (epi != EPI_STORE && epi != EPI_RESID && epi != EPI_ROPE_KV && epi != EPI_SILU_QAH && epi != EPI_SILU_QA) return false;
    const SetPlan p = set_plan(w, N, epi);
    if (epi == EPI_SILU_QA && p.cw != 1) return false;                // (whole-block workgroups are instantiated for unshared rings)
    return p.lds <= SET_LDS_CAP && set_plan_instantiated(p.nc, p.cw);
}
// host-only query (tests, no device needed): the plan for N rows against an M x K matrix -- out = { nc, cw, ncg, rgw, LDS bytes }
bool gemv_set_plan_query(int M, int K, bool interleaved, int N, int epi, long out[5]) {
    QMat w;
    w.tiles = (uint8_t *) (uintptr_t) 16; w.M = M; w.K = K; w.ngroups = (M + 7) / 8; w.nchunks = (K + 255) / 256; w.gmapF8 = interleaved ? w.ngroups / 2 : 0;
    if (!gemv_set_applies(w, N, epi)) return false;
    const SetPlan p = set_plan(w, N, epi);
    out[0] = p.nc; out[1] = p.cw; out[2] = p.ncg; out[3] = p.rgw; out[4] = (long) p.lds;
    return true;
}

template <int NC, int CW>
static hipError_t launch_set_t(const GemvSetArgs &a, int epi, int grid, int nthreads, size_t lds, hipStream_t st) {
    switch (epi) {
    case EPI_STORE:    hipLaunchKernelGGL((k_gemv_set<NC, CW, EPI_STORE>), dim3(grid), dim3(nthreads), lds, st, a); break;
    case EPI_RESID:    hipLaunchKernelGGL((k_gemv_set<NC, CW, EPI_RESID>), dim3(grid), dim3(nthreads), lds, st, a); break;
    case EPI_ROPE_KV:  hipLaunchKernelGGL((k_gemv_set<NC, CW, EPI_ROPE_KV>), dim3(grid), dim3(nthreads), lds, st, a); break;
    case EPI_SILU_QAH: hipLaunchKernelGGL((k_gemv_set<NC, CW, EPI_SILU_QAH>), dim3(grid), dim3(nthreads), lds, st, a); break;
    case EPI_SILU_QA:
        if constexpr (CW == 1) { hipLaunchKernelGGL((k_gemv_set<NC, 1, EPI_SILU_QA>), dim3(grid), dim3(nthreads), lds, st, a); break; }
        else return hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
    }
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

// LH_SET_PROBE builds: LLAMAHIP_SET_PROBE=<records> allocates the record buffer at the first launch; llamahip_debug_set_probe reads it
static unsigned long long *g_set_probe = nullptr;
static long g_set_probe_cap = 0;
long set_probe_dump(unsigned long long *out, long cap_records, bool reset) {
    if (!g_set_probe) return 0;
    unsigned long long hdr[2] = { 0, 0 };
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(hdr, g_set_probe, sizeof(hdr), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    const long n = (long) std::min<unsigned long long>(hdr[0], (unsigned long long) std::min(cap_records, g_set_probe_cap));
    if (out && n > 0 && hipMemcpy(out, g_set_probe + 32, (size_t) n * 32 * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (reset) { hdr[0] = 0; hdr[1] = (unsigned long long) g_set_probe_cap; (void) hipMemcpy(g_set_probe, hdr, sizeof(hdr), hipMemcpyHostToDevice); }
    return n;
}
static hipError_t launch_set_any(const QMat &w, GemvSetArgs a, int epi, hipStream_t st) {
#if LH_SET_PROBE
    if (!g_set_probe && getenv("LLAMAHIP_SET_PROBE")) {
        g_set_probe_cap = std::max(1L, atol(getenv("LLAMAHIP_SET_PROBE")));
        if (hipMalloc((void **) &g_set_probe, (size_t) (g_set_probe_cap + 1) * 32 * 8) != hipSuccess) { g_set_probe = nullptr; g_set_probe_cap = 0; }
        else { const unsigned long long hdr[2] = { 0, (unsigned long long) g_set_probe_cap }; (void) hipMemset(g_set_probe, 0, (size_t) (g_set_probe_cap + 1) * 32 * 8); (void) hipMemcpy(g_set_probe, hdr, sizeof(hdr), hipMemcpyHostToDevice); }
    }
    a.probe = g_set_probe;
#endif
    const SetPlan p = set_plan(w, a.ncols, epi);
    if (p.lds > SET_LDS_CAP || !set_plan_instantiated(p.nc, p.cw)) return hipErrorInvalidValue;
    a.rgw = p.rgw; a.ncg = p.ncg;
    const int nwg = epi == EPI_SILU_QAH ? (w.ngroups / 8 + 7) / 8 * 16 : (w.ngroups + p.rgw - 1) / p.rgw;      // (EPI_SILU_QA: rgw = 8, a workgroup per block)
    const int grid = p.ncg == 1 ? nwg : (nwg + 7) / 8 * 8 * p.ncg;
    const int nthreads = p.rgw * p.cw * 64;
#define LH_SP(NCV, CWV) if (p.nc == NCV && p.cw == CWV) return launch_set_t<NCV, CWV>(a, epi, grid, nthreads, p.lds, st)
    LH_SP(1, 2); LH_SP(1, 3); LH_SP(1, 4);
    LH_SP(2, 1); LH_SP(2, 2); LH_SP(2, 3); LH_SP(2, 4);
    LH_SP(3, 1); LH_SP(3, 3); LH_SP(3, 4);
    LH_SP(4, 1); LH_SP(4, 2); LH_SP(4, 4); LH_SP(5, 1);
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

// w1|w3: whole-block workgroups (no exchange) from two column groups on, or where the half-block exchange has no buffers;
// LLAMAHIP_SET_W13_BLOCKS=0|1 forces halves / whole blocks (parity tests: both epilogues at every row count)
bool gemv_set_silu_whole_blocks(const QMat &w13, int N, bool have_exchange) {
    static const int force = getenv("LLAMAHIP_SET_W13_BLOCKS") ? atoi(getenv("LLAMAHIP_SET_W13_BLOCKS")) : -1;
    if (!gemv_set_applies(w13, N, EPI_SILU_QA)) return false;
    if (!have_exchange || !gemv_set_applies(w13, N, EPI_SILU_QAH)) return true;
    if (force == 0 || force == 1) return force == 1;
    return set_plan(w13, N, EPI_SILU_QA).ncg >= 2;
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
    return launch_set_any(w13, a, gemv_set_silu_whole_blocks(w13, N, hx.amax_t && hx.epoch) ? EPI_SILU_QA : EPI_SILU_QAH, st);
}

hipError_t init_attrs_gemv_set() {
    const int cap = (int) SET_LDS_CAP;
#define LH_ATTR1(NCV, CWV, E) do { hipError_t e_ = hipFuncSetAttribute((const void *) k_gemv_set<NCV, CWV, E>, hipFuncAttributeMaxDynamicSharedMemorySize, cap); if (e_ != hipSuccess) return e_; } while (0)
#define LH_ATTR(NCV, CWV) do { LH_ATTR1(NCV, CWV, EPI_STORE); LH_ATTR1(NCV, CWV, EPI_RESID); LH_ATTR1(NCV, CWV, EPI_ROPE_KV); LH_ATTR1(NCV, CWV, EPI_SILU_QAH); if (CWV == 1) LH_ATTR1(NCV, 1, EPI_SILU_QA); } while (0)
    LH_ATTR(1, 2); LH_ATTR(1, 3); LH_ATTR(1, 4);
    LH_ATTR(2, 1); LH_ATTR(2, 2); LH_ATTR(2, 3); LH_ATTR(2, 4);
    LH_ATTR(3, 1); LH_ATTR(3, 3); LH_ATTR(3, 4);
    LH_ATTR(4, 1); LH_ATTR(4, 2); LH_ATTR(4, 4); LH_ATTR(5, 1);
#undef LH_ATTR
#undef LH_ATTR1
    return hipSuccess;
}

}  // namespace lh
