// decode.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the single-token DECODE step of the quantized-LLaMA hot path, and
// their launchers.  (Round 3 split the former kernels.hip: kcommon.hip.h shared device helpers | prep.hip load-time + activation
// preparation | decode.hip | prompt_gemm.hip multi-row Q4_0 mat-muls | prompt_attn.hip multi-row attention.)
//
// Every kernel reproduces the ARITHMETIC ORDER of the reference's x86 AVX2+FMA+F16C build of
// Sources/cpp/ggml.c (file:line cited per kernel), so results are bit-identical to it, not merely
// close: the reference quantizes activations to Q4_0 before every mat-mul (ggml.c:6134-6152), so a
// 1-ulp difference upstream can flip a 4-bit activation code downstream and move a logit by 1e-3.
// All translation units are compiled with -ffp-contract=off; FMAs appear only where the reference issues
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
// Contents of this file, in order:
//   tagged hand-off helpers (poll_tagged, store_tagged), GemvArgs
//   gemv_body / k_gemv          the decode mat-vec: fused prologues (QA copy | norm | plain | SiLU*up | norm on a tagged row) and
//                               epilogues (store | +residual | SiLU*up -> Q4_0 | tagged rows), register ring of weight chunks
//   k_dec_scores, k_decn_scores, k_dec_pv_blk   attention of one row (decode fallback) / of a short eval (2..60 rows)
//   attn_x_body, k_dec_attn_x, k_qkv_attn       attention in one launch; wq|wk|wv mat-vec + attention in one launch (XCD-local tagged hand-offs)
//   k_xcd_selftest, k_argmax, k_advance, k_bump_epoch, k_topk_keys, k_topk_select
//   launchers: set_phase_probe, launch_gemv (+ kernel selection rules), launch_attn_short, xcd_selftest, launch_dec_attn,
//              launch_qkv_attn, launch_bump_epoch, launch_topk_candidates, launch_argmax, launch_advance, init_kernel_attrs
#define LH_DEFINE_PHASE_PROBE 1
#include <cmath>

#include "kcommon.hip.h"

namespace lh {

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
// (the few-row kernels keep both: +2 % there.)
#ifndef LH_GEMV_SADDR
#define LH_GEMV_SADDR 0
#endif
#ifndef LH_GEMV_PAD
#define LH_GEMV_PAD 1
#endif

// 8-byte granule {value, tag}: written with one 8-byte store, read with one 8-byte load that bypasses the L1 (sc1), so a
// reader sees the value together with its tag or not at all.  The spin is bounded; running out raises the fault word.
__device__ __forceinline__ float poll_tagged(const uint64_t *p, uint32_t tag, uint32_t *fault, bool short_fuse /* give up after 256 polls: fault-injection test */) {
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
    PickIO pick;                                      // EPI_STORE_PICK (llamahip_internal.h)
    int patience;                                     // mailbox polls wait for ANOTHER process / device: their bounds are shifted left by this (3: ~20 s of
                                                      // looks at an uncached / remote granule -- legitimate waits are milliseconds, and a mapping that does not carry
                                                      // the stores must cost the bench's one-token handshake seconds, not minutes, before it falls back to RCCL)
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
    constexpr bool TAGGED = (EPI == EPI_STORE_TAG || PRE == PREP_NORM_TAG || EPI == EPI_RESID_TAG || EPI == EPI_SILU_QAH);
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
    // (EPI_SILU_QAH: workgroup blk = half `(blk >> 3) & 1` of activation block `(blk >> 4) * 8 + (blk & 7)` -- the two halves of a block
    //  are 8 apart in the grid, i.e. on one XCD; waves 0, 1 own the half's two gate row-groups, waves 2, 3 the matching up row-groups
    //  of the interleaved tile order)
    const int qah_block = (blk >> 4) * 8 + (blk & 7), qah_half = (blk >> 3) & 1;
    const int g = EPI == EPI_SILU_QAH ? qah_block * 8 + (wave >> 1) * 4 + qah_half * 2 + (wave & 1) : blk * nw + wave;
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
    if (EPI == EPI_SILU_QAH) {
        // 4 waves: waves 0, 1 hold gate rows 16 half .. + 15 of the block, waves 2, 3 the matching up rows
        float *gu = (float *) red;                      // prologue scratch is free again
        __syncthreads();
        if (k == 0) gu[wave * 8 + (lane >> 3)] = acc;
        __syncthreads();
        if (wave == 0 && valid) {
            const int i = lane & 15;
            const uint16_t gh = f2h_bits(gu[i]);
            const float act = h2f_bits((ga.lut_math & 1) ? silu_math_bits(gh) : T_silu[gh]) * gu[16 + i];
            float amax = wave_max_f(lane < 16 ? fabsf(act) : 0.0f);
            // the other half's partial amax: one tagged granule each way inside this XCD's L2
            uint64_t *at = (uint64_t *) ga.out_t;
            const int hb = qah_block * 2 + qah_half;
            float other = 0.0f;
            if (lane == 0) {
                store_tagged(at + hb, amax, store_tag ^ ((ga.lut_math & 0x1000) ? 1u : 0u));      // (0x1000: fault-injection test)
                other = poll_tagged(at + (hb ^ 1), store_tag, ga.fault, (ga.lut_math & 0x1000) != 0);
            }
            other = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, other)));
            amax = fmaxf(amax, other);
            const float dd = amax / 7.0f;
            const float id = (amax != 0.0f) ? 7.0f / amax : 0.0f;
            const uint32_t nib = (uint32_t) ((int) __builtin_rintf(act * id)) & 0xF;           // signed nibble of (q - 8)
            const int kk = lane & 7;
            const uint32_t e0 = __shfl(nib, 2 * kk), e1 = __shfl(nib, 2 * kk + 1);
            const int b = qah_block, c = b >> 3, j = b & 7;
            if (lane < 8) ((uint16_t *) (out_A + (c * 8 + kk) * 8 + j))[qah_half] = (uint16_t) ((e0 | (e1 << 8)) << (4 * (j & 1)));
            if (lane == 0 && qah_half == 0) out_d[b] = dd;
        }
    } else
    if (EPI == EPI_SILU_QA) {
        // 8 waves: waves 0-3 hold gate rows b*32 .. b*32+31, waves 4-7 the matching up rows (b = blockIdx.x)
        float *gu = (float *) red;                      // prologue scratch is free again
        __syncthreads();
        if (k == 0) gu[wave * 8 + (lane >> 3)] = acc;
        __syncthreads();
        if (wave == 0) {
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
        if (EPI == EPI_STORE_PICK) {
            if (live) y[m] = acc;
            // order-preserving key {value, ~index}: the largest value wins, then the lower index; NaN (key 0) never does; -0 == +0
            unsigned long long key = 0ull;
            if (live && acc == acc) {
                const uint32_t b = __builtin_bit_cast(uint32_t, acc == 0.0f ? 0.0f : acc);
                const uint32_t ob = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
                key = ((unsigned long long) ob << 32) | (unsigned long long) (0xFFFFFFFFu - (uint32_t) m);
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const uint32_t lo_ = (uint32_t) __shfl_xor((int) (uint32_t) key, o), hi_ = (uint32_t) __shfl_xor((int) (uint32_t) (key >> 32), o);
                const unsigned long long ok = ((unsigned long long) hi_ << 32) | lo_;
                key = ok > key ? ok : key;
            }
            unsigned long long *wk = (unsigned long long *) red;      // [nw] wave keys, then [15] = "this workgroup finished last", [14] = the pick
            __syncthreads();
            if (lane == 0) wk[wave] = key;
            __syncthreads();
            // Arrival: this workgroup's key goes to its own slot (write-through store, acknowledged before the ticket), tickets are
            // taken per shard of 8 (block % 8: one atomic per workgroup on 8 different words -- a single counter costs ~12 ns per
            // arrival, 1 000 workgroups = 12 us), the last of a shard takes a ticket of the top counter, the last of those reduces.
            if (tid == 0) {
                unsigned long long k2 = wk[0];
                for (int w2_ = 1; w2_ < nw; w2_++) k2 = wk[w2_] > k2 ? wk[w2_] : k2;
                __hip_atomic_store(ga.pick.key + blk, k2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int nwg = (int) gridDim.x, shard = blk & 7, nsh = (nwg + 7 - shard) / 8, ntop = nwg < 8 ? nwg : 8;
                int lastwg = 0;
                if (atomicAdd(ga.pick.count + shard * 16, 1u) == (uint32_t) (nsh - 1))
                    lastwg = atomicAdd(ga.pick.count + 128, 1u) == (uint32_t) (ntop - 1);
                wk[15] = (unsigned long long) lastwg;
            }
            __syncthreads();
            if (wk[15] != 0ull) {
                // the last workgroup: every key is visible (each was acknowledged before its ticket); fold them in index order
                unsigned long long best = 0ull;
                for (int i = tid; i < (int) gridDim.x; i += nw * 64) {
                    const unsigned long long kv = __hip_atomic_load(ga.pick.key + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    best = kv > best ? kv : best;
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    const uint32_t lo_ = (uint32_t) __shfl_xor((int) (uint32_t) best, o), hi_ = (uint32_t) __shfl_xor((int) (uint32_t) (best >> 32), o);
                    const unsigned long long ok = ((unsigned long long) hi_ << 32) | lo_;
                    best = ok > best ? ok : best;
                }
                __syncthreads();
                if (lane == 0) wk[wave] = best;
                __syncthreads();
                if (tid == 0) {
                    unsigned long long fin = wk[0];
                    for (int w2_ = 1; w2_ < nw; w2_++) fin = wk[w2_] > fin ? wk[w2_] : fin;
                    const int r = (fin >> 32) == 0ull ? 0 : (int) (0xFFFFFFFFu - (uint32_t) fin);
                    int32_t *st = ga.pick.state;
                    ga.pick.out[st[1]] = r;
                    if (ga.pick.next_token) *ga.pick.next_token = r;
                    st[0] += 1; st[1] += 1;
                    for (int sh = 0; sh < 8; sh++) __hip_atomic_store(ga.pick.count + sh * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(ga.pick.count + 128, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    wk[14] = (unsigned long long) (r < ga.pick.n_vocab ? r : 0);
                }
                __syncthreads();
            }
            if (wk[15] != 0ull && ga.pick.emb) {
                // the picked token's embedding row for the next step: k_embed_part's arithmetic, thread for thread (nw * 64 = 256 threads)
                const int tok = (int) wk[14];
                const int dE = K;
                __syncthreads();
                const uint8_t *row = ga.pick.emb + (size_t) tok * (dE / 32) * 20;
                double s1 = 0.0, s2 = 0.0;
                for (int i = tid; i < dE / 2; i += nw * 64) {
                    const int b = i >> 4, j = i & 15;
                    const uint8_t *blk_ = row + b * 20;
                    const uint32_t bits = blk_[0] | (blk_[1] << 8) | (blk_[2] << 16) | ((uint32_t) blk_[3] << 24);
                    const float dd = __builtin_bit_cast(float, bits);
                    const uint32_t q = blk_[4 + j];
                    const float v0 = (float) ((int) (q & 0xF) - 8) * dd, v1 = (float) ((int) (q >> 4) - 8) * dd;
                    ga.pick.x_next[2 * i + 0] = v0;
                    ga.pick.x_next[2 * i + 1] = v1;
                    s1 += (double) v0; s1 += (double) v1;
                    s2 += (double) v0 * (double) v0; s2 += (double) v1 * (double) v1;
                }
                s1 = block_sum_d(s1, red, 0);
                s2 = block_sum_d(s2, red, 1);
                if (tid == 0) {
                    ((f64x2 *) ga.pick.part_next)[0] = f64x2{ s1, s2 };
                    if (ga.pick.epoch) ga.pick.epoch[0] = next_epoch(ga.pick.epoch[0]);
                }
            }
        } else
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
#if LH_PHASE_PROBE == 3        /* timeline probe (tools/pv_stream_timeline.py): kind 0xB2 */
    const unsigned long long probe_wall = wall_clock64(), probe_t0 = __builtin_readcyclecounter();
    unsigned long long probe_t1 = 0, probe_t2 = 0;
#endif
    const int n_past = st[0];
    const int t0 = blockIdx.y * DEC_TS;
    if (t0 > n_past) return;
    const int tid = threadIdx.x;
    const bool owns_new = n_past < t0 + DEC_TS;          // this slice contains key n_past
    const double *tab = sincos_tab + (size_t) n_past * dh;
    const float *q = qkv + h * dh, *kk = qkv + d + h * dh, *vv = qkv + 2 * d + h * dh;
    // the rotation's operands first, then the K rows, then the rotation: loads return in order, so the rotation waits for its few
    // operands only while the K rows -- at long contexts this launch is one round trip of K -- are already on their way (the rows
    // used to be requested behind the rotation and its barrier)
    const bool rot = tid < dh / 2;
    const int e = rot ? 2 * tid : 0;
    const double cs = tab[e], sn = tab[e + 1];
    const float q0 = q[e], q1 = q[e + 1];
    const float kr0 = kk[e], kr1 = kk[e + 1], vr0 = vv[e], vr1 = vv[e + 1];
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
    if (rot) {
        const double x0 = (double) q0, x1 = (double) q1;
        qs[e] = (float) (x0 * cs - x1 * sn);
        qs[e + 1] = (float) (x0 * sn + x1 * cs);
        if (owns_new) {
            const double k0 = (double) kr0, k1 = (double) kr1;
            const float r0 = (float) (k0 * cs - k1 * sn), r1 = (float) (k0 * sn + k1 * cs);
            kn[e] = r0; kn[e + 1] = r1;
            Kc[(size_t) n_past * d + h * dh + e] = r0;
            Kc[(size_t) n_past * d + h * dh + e + 1] = r1;
            Vc[(size_t) n_past * d + h * dh + e] = vr0;
            Vc[(size_t) n_past * d + h * dh + e + 1] = vr1;
        }
    }
    __syncthreads();
#if LH_PHASE_PROBE == 3
    probe_t1 = __builtin_readcyclecounter();
#endif
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
#if LH_PHASE_PROBE == 3
    probe_t2 = __builtin_readcyclecounter();
    if (g_phase_probe && tid == 0) {
        unsigned long long *pb = g_phase_probe; const unsigned long long slot = atomicAdd(pb, 1ull);
        if (slot < pb[1]) { unsigned long long *e = pb + 8 * (1 + slot); e[0] = probe_t0; e[1] = probe_t1; e[2] = probe_t2; e[3] = e[4] = probe_t2;
            e[5] = (0xB2ull << 48) | ((unsigned long long) blockIdx.y << 32) | (unsigned) h; e[6] = wall_clock64(); e[7] = probe_wall; }
    }
#endif
}

// Short prompt chunks (2 <= N <= 16 rows, the reference's n_batch = 8 flow): the decode work distribution with
// one more grid dimension.  Row n (blockIdx.z) is the query at position n_past + n and sees keys
// 0 .. n_past + n; q is already rotated and K / V already appended by k_rope_kv.  Same 32 FMA chains
// and reduction tree as k_dec_scores / k_attn.   sc: [row][head][n_ctx]
__global__ void __launch_bounds__(256)
k_decn_scores(const float *__restrict__ qr, int d, int dh, const float *__restrict__ Kc, float *__restrict__ sc,
              int n_ctx, float kq_scale, int n_past, const SeqSet *__restrict__ set) {
    const int h = blockIdx.x, n = blockIdx.z, H = gridDim.x;
    const int np = set ? set->state[n][0] : n_past + n;         // (set: row n is the next token of its own sequence)
    if (set) Kc += set->kv_off[n];
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
// soft_max of one head's score row into LDS (p[0..T)): max, fp16-table exp, double sum, scale (ggml.c:5619-5665); every thread
// of the workgroup calls it, the caller synchronizes before reading p
static __device__ __forceinline__ void pv_soft_max(const float *__restrict__ row, float *p, int T, int tid, int nt, double *red,
                                                   const uint16_t *__restrict__ T_exp, int lut_math) {
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
}
// the nth partial sums of 32 columns added in thread order, stored (fp32 row, if asked) and quantized to one Q4_0 activation
// block of the QA layout; lanes 0..31 of the workgroup's first wave work
static __device__ __forceinline__ void pv_store_block(const float *part, int nth, int tid, int col, int h, int dh, int cb,
                                                      float *__restrict__ merged, uint32_t *__restrict__ qa_A, float *__restrict__ qa_d) {
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

template <bool MULTI>
__global__ void __launch_bounds__(1024)
k_dec_pv_blk(const float *__restrict__ sc, const float *__restrict__ Vc, int d, int dh, int n_ctx, int nth,
             float *__restrict__ merged, uint32_t *__restrict__ qa_A, float *__restrict__ qa_d,
             const uint16_t *__restrict__ T_exp, const int32_t *__restrict__ st,
             int n_past0, long qa_strideA, long qa_strideD, int lut_math, int chunk, const SeqSet *__restrict__ set) {
    extern __shared__ double smem_d[];
    double *red = smem_d;
    float *p = (float *) (smem_d + 32);
    float *part = p + n_ctx;
    const int h = blockIdx.x, cb = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
    // (MULTI with a set: row blockIdx.z is a single-row eval of its own sequence -- own position, own cache, own key split)
    const int n_past = MULTI ? (set ? set->state[blockIdx.z][0] : n_past0 + (int) blockIdx.z) : st[0];
    if (MULTI && set) Vc += set->kv_off[blockIdx.z];
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
    pv_soft_max(row, p, T, tid, nt, red, T_exp, lut_math);
    // a chunk row is as long as the whole chunk's context (ggml.c:5459-5480 splits n_past + N keys over the
    // threads for every row); the masked tail has weight exp(-inf) = 0 and is walked like the reference does
    // (a chunked pass -- prompt_attn.hip split_keys -- stands for successive evals of `chunk` rows: the row's own eval ends with its chunk)
    const int Tpv = (MULTI && !set) ? (chunk > 0 ? n_past0 + min((int) gridDim.z, ((int) blockIdx.z / chunk + 1) * chunk) : n_past0 + (int) gridDim.z) : T;
    if (MULTI && !set)
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
    pv_store_block(part, nth, tid, col, h, dh, cb, merged, qa_A, qa_d);
}

// k_dec_pv_stream: k_dec_pv_blk<false> for long contexts (decode after a long prompt).  A chain walks T / nth rows of V that lie
// 4 d bytes apart, and the register batches of k_dec_pv_blk keep 32 of them in flight per chain: at 2 048 keys that is 8 dependent
// memory round trips with 4 MB in flight chip-wide (H dh/32 workgroups x nth 32 chains x 32 rows x 4 B).  Here ALL 1 024 threads of the
// workgroup fetch (16 B each, 8 lanes per 128-byte row segment) and only the chain owners do arithmetic: a stage is SR consecutive
// rows of every chain of the workgroup ([chain][SR][32] floats in LDS), two stages are in flight in registers (NL 16-byte loads per
// thread each) while the owners consume the one before from the LDS double buffer -- one barrier per stage.  The chains themselves
// are the reference's: rows dc th .. dc th + dc - 1 in order, one fp32 FMA each (ggml.c:5459-5480), so the sums are bit-identical
// to k_dec_pv_blk's.
// What a CU can pull from memory is bounded (about 20 GB/s each when all 256 stream; 17.6 us per launch at 2 048 keys with the 7B's
// H dh/32 = 128 workgroups on 128 CUs), so the chains of a (head, column block) are SPLIT over `split` workgroups where that fills
// the chip: workgroup z owns chains [z cpw, z cpw + cpw), the ones with z < split - 1 leave their sums as tagged granules
// {sum, make_tag(epoch, layer + 1)} in xpart, the last one (dispatched last: the others are resident or done when it looks) takes
// them with bounded polls and adds all nth in thread order.  Workgroups of a (head, column block) are 8 apart in the 1-D grid:
// one XCD, one L2 (the placement xcd_selftest checked at load) -- workgroup-scope stores, L1-bypassing loads, no fences.
// grid H dh/32 x split (1-D), 1 024 threads, LDS = red + p[n_ctx] + part + 2 stages.
template <int NL>
__global__ void __launch_bounds__(1024)
k_dec_pv_stream(const float *__restrict__ sc, const float *__restrict__ Vc, int d, int dh, int n_ctx, int nth, int SR,
                float *__restrict__ merged, uint32_t *__restrict__ qa_A, float *__restrict__ qa_d,
                const uint16_t *__restrict__ T_exp, const int32_t *__restrict__ st, int lut_math,
                int H, int split, uint64_t *__restrict__ xpart, const uint32_t *__restrict__ epoch, int layer, uint32_t *__restrict__ fault) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ double smem_d[];
    double *red = smem_d;
    float *p = (float *) (smem_d + 32);
    float *part = p + ((n_ctx + 3) & ~3);
    f32x4 *stage = (f32x4 *) (part + nth * 32);                          // 2 x [cpw][SR][8] quads
    const int tid = threadIdx.x;
#if LH_PHASE_PROBE == 3        /* timeline probe (tools/pv_stream_timeline.py): kinds 0xB0 / 0xB1 = first / second half of a workgroup's stamps */
    unsigned long long probe_t[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    const unsigned long long probe_wall = wall_clock64();
#define LH_PSTAMP(IDX) do { probe_t[IDX] = __builtin_readcyclecounter(); } while (0)
#define LH_PFLUSH() do { if (g_phase_probe && tid == 0) { unsigned long long *pb = g_phase_probe; const unsigned long long wall_ = wall_clock64(); \
        for (int half_ = 0; half_ < 2; half_++) { const unsigned long long slot = atomicAdd(pb, 1ull); \
        if (slot < pb[1]) { unsigned long long *e = pb + 8 * (1 + slot); for (int i = 0; i < 5; i++) e[i] = probe_t[5 * half_ + i]; \
            e[5] = ((unsigned long long) (0xB0 + half_) << 48) | ((unsigned long long) z << 32) | (unsigned) base; e[6] = wall_; e[7] = probe_wall; } } } } while (0)
#else
#define LH_PSTAMP(IDX) do { } while (0)
#define LH_PFLUSH() do { } while (0)
#endif
    const int bid = blockIdx.x, z = (bid >> 3) % split, base = (bid / (8 * split)) * 8 + (bid & 7);
    const int h = base % H, cb = base / H;
    const int cpw = (nth + split - 1) / split, th_lo = z * cpw, nloc = min(nth, th_lo + cpw) - th_lo;      // this workgroup's chains
    LH_PSTAMP(0);
    const int T = st[0] + 1;
    const int dc = (T + nth - 1) / nth;
    const int nstage = (dc + SR - 1) / SR;
    const int col0 = h * dh + cb * 32;
    const int stage_quads = nloc * SR * 8;                              // <= NL * 1024
    // the head's score row first (registers; PV_ROW values per thread cover n_ctx <= 4 096): loads complete in order, so the
    // soft_max below waits for these and not for the V stages issued right after them
    constexpr int PV_ROW = 4;
    const float *row = sc + (size_t) h * n_ctx;
    float rv[PV_ROW];
#pragma unroll
    for (int i = 0; i < PV_ROW; i++) rv[i] = row[min(i * 1024 + tid, T - 1)];
    // quad e of a stage: row r = e / 8 of the stage (local chain r / SR, its u-th row), columns 4 (e % 8) ..; rows past the chain's
    // end are clamped re-reads nobody consumes
    auto fetch = [&](int sidx, f32x4 (&v)[NL]) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int e = min(i * 1024 + tid, stage_quads - 1);
            const int r = e >> 3, q = e & 7;
            const int lc = r / SR, u = r - lc * SR, th = th_lo + lc;
            const int t = min(min(dc * th + sidx * SR + u, dc * th + dc - 1), T - 1);
            v[i] = __builtin_nontemporal_load((const f32x4 *) (Vc + (size_t) t * d + col0 + q * 4));
        }
    };
    auto put = [&](int buf, const f32x4 (&v)[NL]) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int e = i * 1024 + tid;
            if (e < stage_quads) stage[(size_t) buf * stage_quads + e] = v[i];
        }
    };
    f32x4 va[NL], vb[NL];
    {   // soft_max (pv_soft_max's arithmetic on the register copy: max, fp16-table exp, double sum, scale -- ggml.c:5619-5665)
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < PV_ROW; i++) if (i * 1024 + tid < T) mx = fmaxf(mx, rv[i]);
        mx = block_max_f(mx, red, 0);
        LH_PSTAMP(1);
        // The V stages are requested only now, with the score row in hand: every workgroup of the launch asks for its whole share of
        // V within the first microsecond, and a row requested alongside arrives behind all of it (7.3 us after entry at 2 048 keys,
        // in-kernel timeline) with the soft_max still to do; asked for alone it is back in ~2 us and the soft_max runs under the V stream.
        fetch(0, va);
        if (nstage > 1) fetch(1, vb);
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < PV_ROW; i++) {
            if (i * 1024 + tid < T) {
                const uint16_t xh = f2h_bits(rv[i] - mx);
                rv[i] = h2f_bits((lut_math & 2) ? exp_math_bits(xh) : T_exp[xh]);
                sum += (double) rv[i];
            }
        }
        sum = block_sum_d(sum, red, 1);
        const float inv = (float) (1.0 / sum);
#pragma unroll
        for (int i = 0; i < PV_ROW; i++) if (i * 1024 + tid < T) p[i * 1024 + tid] = rv[i] * inv;
    }
    LH_PSTAMP(2);
    const int lc = tid >> 5, c = tid & 31, th = th_lo + lc;
    const bool owner = lc < nloc;
    const int t0 = dc * th, t1 = min(t0 + dc, T);
    float acc = 0.0f;
    auto consume = [&](int sidx, int buf) {
        if (!owner) return;
        const float *vs = (const float *) (stage + (size_t) buf * stage_quads) + (size_t) lc * SR * 32 + c;
        const int tb = t0 + sidx * SR, n = min(SR, t1 - tb);
        int u = 0;
        for (; u + 16 <= n; u += 16) {                                   // LDS reads batched, the FMA chain in key order
            float v[16], w[16];
#pragma unroll
            for (int k = 0; k < 16; k++) { v[k] = vs[(u + k) * 32]; w[k] = p[tb + u + k]; }
#pragma unroll
            for (int k = 0; k < 16; k++) acc = fmaf(v[k], w[k], acc);
        }
        for (; u < n; u++) acc = fmaf(vs[u * 32], p[tb + u], acc);
    };
    for (int sidx = 0; sidx < nstage; sidx += 2) {
        put(0, va);
        __syncthreads();                                                // (the first one also orders p before the owners' reads)
        if (sidx == 0) LH_PSTAMP(3); else if (sidx == 2) LH_PSTAMP(7);
        if (sidx + 2 < nstage) fetch(sidx + 2, va);
        consume(sidx, 0);
        if (sidx == 0) LH_PSTAMP(4);
        if (sidx + 1 < nstage) {
            put(1, vb);
            __syncthreads();
            if (sidx == 0) LH_PSTAMP(5);
            if (sidx + 3 < nstage) fetch(sidx + 3, vb);
            consume(sidx + 1, 1);
            if (sidx == 0) LH_PSTAMP(6);
        }
    }
    LH_PSTAMP(8);
    if (z + 1 < split) {                                                // not the last workgroup of this column block: publish and leave
        if (owner) store_tagged(xpart + ((size_t) base * nth + th) * 32 + c, acc, make_tag(epoch[0], layer + 1) ^ ((lut_math & 0x1000) ? 1u : 0u));      // (0x1000: fault-injection test)
        LH_PSTAMP(9);
        LH_PFLUSH();
        return;
    }
    if (owner) part[th * 32 + c] = acc;
    if (split > 1 && tid < th_lo * 32)                                  // the chains before this workgroup's, from their owners
        part[tid] = poll_tagged(xpart + (size_t) base * nth * 32 + tid, make_tag(epoch[0], layer + 1), fault, (lut_math & 0x1000) != 0);
    __syncthreads();
    pv_store_block(part, nth, tid, col0 + tid, h, dh, cb, merged, qa_A, qa_d);
    LH_PSTAMP(9);
    LH_PFLUSH();
}
#undef LH_PSTAMP
#undef LH_PFLUSH

// ------------------------------------------------------------------------------------------------
// k_dec_pv_dma (round 4): the long-context soft_max . V with the V rows moved HBM -> LDS by LDS-DMA from two loader waves instead of
// through the registers of all 1 024 threads, and with a different decomposition: a workgroup owns `cpw` chunks of the reference's
// n_threads-way key split for ALL 128 columns of a head (k_dec_pv_stream: 32 columns), so it reads 512 contiguous bytes of every V
// row instead of 128.  In k_dec_pv_stream the soft_max took 6 us at 2 048 keys although it needs 1.8 us alone: every thread was also
// issuing V loads, and a wave's exp / sum instructions queue behind its own load issue.  Here waves 0 and 1 do nothing but
// global_load_lds_dwordx4 (1 KiB per instruction = 2 consecutive rows x 512 bytes of one chain, per-lane addresses; no VGPR round trip,
// no barrier per stage; a lone wave issues ~one instruction per 8 cycles, hence two loaders and incremental addresses), starting when
// the score row is back (issued earlier, V would delay the row behind megabytes of requests: k_dec_pv_stream's timeline), into a
// ring of stages [chain][8 rows][128 floats]; waves 2 .. 7 run the soft_max (ggml.c:5619-5665) and then the owners of the workgroup's
// chains (two waves per chain: 128 columns) walk their rows in key order out of the ring -- the same fp32 FMA chains
// (ggml.c:5459-5480), so the sums are bit-identical to k_dec_pv_blk / k_dec_pv_stream.  Flow control: `landed` per loader wave (its own
// vmcnt) and one `done` word per consumer wave.  The chunks of a head are split over `split` workgroups on one XCD; all but the last
// leave their sums as tagged granules, the last adds all n_threads in thread order and quantizes the head's four Q4_0 blocks.
// Workgroup barriers: one raw s_barrier before the loaders start (a __syncthreads would make them wait for their DMA), then an LDS
// counter among the six worker waves.  grid H x split, 512 threads; LDS = red + p[n_ctx] + part[nth][128] + flags + ring.  dh = 128.
// ------------------------------------------------------------------------------------------------
#ifndef LH_PVD_ABLATE
#define LH_PVD_ABLATE 0
#endif
constexpr int PVD_SR = 8;            // rows per chain and stage: four DMA instructions of two rows
template <int N> __device__ __forceinline__ void pvd_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void pvd_barrier() {      // LDS writes of this wave done, then the hardware barrier; nothing waits for VMEM
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// barrier of the 6 worker waves only (the two loader waves take part in the first hardware barrier and in nothing after it): an LDS counter
__device__ __forceinline__ void pvd_worker_barrier(uint32_t *ctr, uint32_t &round, int lane, uint32_t *fault) {
    round += 6;
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < round) { __builtin_amdgcn_s_sleep(0); if (poll_give_up(spins, 1 << 24, fault)) break; }
}
__global__ void __launch_bounds__(512)
k_dec_pv_dma(const float *__restrict__ sc, const float *__restrict__ Vc, int d, int n_ctx, int nth, int NS,
             float *__restrict__ merged, uint32_t *__restrict__ qa_A, float *__restrict__ qa_d,
             const uint16_t *__restrict__ T_exp, const int32_t *__restrict__ st, int lut_math,
             int H, int split, uint64_t *__restrict__ xpart, const uint32_t *__restrict__ epoch, int layer, uint32_t *__restrict__ fault) {
    constexpr int dh = 128;
    extern __shared__ double smem_d[];
    double *red = smem_d;                                               // [32]
    float *p = (float *) (smem_d + 32);                                 // [n_ctx rounded to 4]
    float *part = p + ((n_ctx + 3) & ~3);                               // [nth][128]
    uint32_t *flags = (uint32_t *) (part + nth * dh);                   // [0], [1] stages landed per loader wave | [2 .. 7] done per consumer wave | [8] worker barrier
    float *ring = (float *) (flags + 16);                               // [NS][nloc][8][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x, z = (bid >> 3) % split, h = (bid / (8 * split)) * 8 + (bid & 7);      // the workgroups of a head are 8 apart: one XCD
    const int cpw = (nth + split - 1) / split, th_lo = z * cpw, nloc = min(nth, th_lo + cpw) - th_lo;      // this workgroup's chunks (1 .. 3)
    const int T = st[0] + 1;
    const int dc = (T + nth - 1) / nth;
    const int nstage = (dc + PVD_SR - 1) / PVD_SR;
    const int col0 = h * dh;
    const int ncw = 2 * nloc;                                           // consumer waves (2 .. 1 + ncw): two per chain
    const int limit = (lut_math & 0x1000) ? (1 << 8) : (1 << 22);
    if (tid < 16) flags[tid] = 0u;
    // the head's score row: the 6 worker waves only (a loader wave must have no load of its own in flight next to its DMA)
    constexpr int PV_ROW = 11, NWT = 384;                               // 384 threads x 11 cover n_ctx <= 4 096
    const float *row = sc + (size_t) h * n_ctx;
    float rv[PV_ROW];
    const int wt = tid - 128;
    float mx = -INFINITY;
    if (wave > 1) {
#pragma unroll
        for (int i = 0; i < PV_ROW; i++) rv[i] = row[min(i * NWT + wt, T - 1)];
#pragma unroll
        for (int i = 0; i < PV_ROW; i++) if (i * NWT + wt < T) mx = fmaxf(mx, rv[i]);
        mx = wave_max_f(mx);
        if (lane == 0) ((float *) red)[wave] = mx;
    }
    pvd_barrier();                                                      // the row is back (and the flags are zero): the loaders may start
    if (wave < 2) {
        // ====== loaders: a chain-stage is 4 instructions (rows 0-1, 2-3, 4-5, 6-7); loader w issues instructions w and w + 2 of every chain ======
        const uint32_t ring_lds = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) float *) ring;
        const uint64_t vbase = (uint64_t) (uintptr_t) (Vc + col0);
        const int u = lane >> 5, q = lane & 31;
        // per-lane byte offset of my first row of stage 0 of chain j (instruction w: rows 2 w + u; instruction w + 2: 4 rows further);
        // a stage further is PVD_SR rows further.  Rows beyond a chain's end are fetched and never consumed; only the cache's end bounds them.
        uint32_t off[3];
#pragma unroll
        for (int j = 0; j < 3; j++) off[j] = (uint32_t) ((size_t) min(dc * (th_lo + min(j, nloc - 1)) + 2 * wave + u, n_ctx - 1) * d * 4 + q * 16);
        const uint32_t step = (uint32_t) ((size_t) PVD_SR * d * 4), four = (uint32_t) ((size_t) 4 * d * 4);
        const uint32_t off_max = (uint32_t) ((size_t) (n_ctx - 1) * d * 4 + q * 16);
        uint32_t pub = 0;
        int slot = 0;
        auto min_done = [&]() { uint32_t m = 0xffffffffu; for (int w = 0; w < ncw; w++) m = min(m, __hip_atomic_load(flags + 2 + w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)); return m; };
        const int F = 60 / (2 * nloc);                                  // stages whose DMA instructions (2 nloc per stage and loader) may be outstanding
        for (int s = 0; s < nstage; s++) {
            if (s >= NS && min_done() < (uint32_t) (s - NS + 1)) {
                pvd_wait_vmcnt<0>();                                    // the ring is full: everything issued has to land anyway
                if ((uint32_t) s > pub) { pub = (uint32_t) s; __hip_atomic_store(flags + wave, pub, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
                int spins = 0;
                while (min_done() < (uint32_t) (s - NS + 1)) { __builtin_amdgcn_s_sleep(1); if (poll_give_up(spins, limit, fault)) break; }
            }
            const bool tail = (s + 2) * PVD_SR + dc * nth > n_ctx;      // (uniform) only the last stages of the last chains can run past the cache
#pragma unroll
            for (int j = 0; j < 3; j++) {
                if (j < nloc) {
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        const uint32_t o = off[j] + (i ? four : 0u);
                        const uint32_t voff = tail ? min(o, off_max) : o;
                        const uint32_t dst = ring_lds + (uint32_t) ((slot * nloc + j) * 4096 + (wave + 2 * i) * 1024);
                        uint32_t keep;
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep) : "v"(voff), "s"(dst), "s"(vbase) : "memory");
                    }
                    off[j] += step;
                }
            }
            slot = slot + 1 == NS ? 0 : slot + 1;
            if (s + 1 > F) {
                if (nloc == 1) pvd_wait_vmcnt<60>(); else if (nloc == 2) pvd_wait_vmcnt<60>(); else pvd_wait_vmcnt<60>();       // F * 2 nloc = 60 for 1, 2, 3 chains
                if ((uint32_t) (s + 1 - F) > pub) { pub = (uint32_t) (s + 1 - F); __hip_atomic_store(flags + wave, pub, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
            }
        }
        pvd_wait_vmcnt<0>();
        __hip_atomic_store(flags + wave, (uint32_t) nstage, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;                                                         // the loaders are done; the workers go on among themselves
    }
    uint32_t wround = 0;
    {
        // soft_max (pv_soft_max's arithmetic on the register copy): max, fp16-table exp, double sum, scale
        float m2 = ((float *) red)[2];
        for (int w = 3; w < 8; w++) m2 = fmaxf(m2, ((float *) red)[w]);
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < PV_ROW; i++) {
#if LH_PVD_ABLATE == 1     /* measurement only (wrong sums): exponentiate this workgroup's OWN keys only -- the upper bound of what splitting the soft_max over a head's workgroups could save, before its exchange costs (profiles/r05_q_pv_softmax_ablation.txt: 15.23 -> 15.74 us, i.e. nothing) */
            if (i * NWT + wt < T && (i * NWT + wt < dc * th_lo || i * NWT + wt >= dc * (th_lo + nloc))) rv[i] = 0.0f; else
#endif
            if (i * NWT + wt < T) {
                const uint16_t xh = f2h_bits(rv[i] - m2);
                rv[i] = h2f_bits((lut_math & 2) ? exp_math_bits(xh) : T_exp[xh]);
                sum += (double) rv[i];
            }
        }
        sum = wave_sum_d(sum);
        if (lane == 0) red[8 + wave] = sum;
    }
    pvd_worker_barrier(flags + 8, wround, lane, fault);
    float acc = 0.0f;
    const int lc = (tid - 128) >> 7, c = (tid - 128) & 127, th = th_lo + lc;      // chain lc = worker threads [128 lc, 128 lc + 128): its 128 columns
    const bool owner = lc < nloc;
    {
        double tot = red[10];
        for (int w = 3; w < 8; w++) tot += red[8 + w];                  // (exact in any order: every term is a multiple of 2^-24, at most 2^12 terms)
        const float inv = (float) (1.0 / tot);
#pragma unroll
        for (int i = 0; i < PV_ROW; i++) if (i * NWT + wt < T) p[i * NWT + wt] = rv[i] * inv;
    }
    pvd_worker_barrier(flags + 8, wround, lane, fault);                 // p is complete
    if (wave - 2 < ncw) {
        // ---- chain owners: rows t0 .. t1 - 1 of chunk th in key order, from the ring
        const int t0 = dc * th, t1 = min(t0 + dc, T);
        uint32_t landed_seen = 0;
        int slot = 0;
        for (int s = 0; s < nstage; s++) {
            if ((uint32_t) s >= landed_seen) {
                int spins = 0;
                for (;;) {
                    landed_seen = min(__hip_atomic_load(flags, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP), __hip_atomic_load(flags + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                    if (landed_seen > (uint32_t) s) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (poll_give_up(spins, limit, fault)) break;
                }
            }
            {
                const float *vs = ring + (size_t) (slot * nloc + lc) * 1024 + c;
                const int tb = t0 + s * PVD_SR, n = min(PVD_SR, t1 - tb);
                if (n == PVD_SR) {
                    float v[PVD_SR], w[PVD_SR];
#pragma unroll
                    for (int k = 0; k < PVD_SR; k++) { v[k] = vs[k * dh]; w[k] = p[tb + k]; }
#pragma unroll
                    for (int k = 0; k < PVD_SR; k++) acc = fmaf(v[k], w[k], acc);
                } else {
                    for (int k = 0; k < n; k++) acc = fmaf(vs[k * dh], p[tb + k], acc);
                }
            }
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_store(flags + wave, (uint32_t) (s + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            slot = slot + 1 == NS ? 0 : slot + 1;
        }
    }
    const uint32_t ptag = make_tag(epoch[0], layer + 1);
    if (z + 1 < split) {                                                // not the last workgroup of this head: publish and leave
        if (owner) store_tagged(xpart + ((size_t) h * nth + th) * dh + c, acc, ptag ^ ((lut_math & 0x1000) ? 1u : 0u));      // (0x1000: fault-injection test)
        return;
    }
    if (owner) part[th * dh + c] = acc;
    for (int i = wt; i < th_lo * dh; i += NWT)                          // the chunks before this workgroup's, from their owners
        part[i] = poll_tagged(xpart + (size_t) h * nth * dh + i, ptag, fault, (lut_math & 0x1000) != 0);
    pvd_worker_barrier(flags + 8, wround, lane, fault);
    // the head's four 32-column blocks: partial sums added in thread order, quantized (pv_store_block, one block per wave 2 .. 5)
    if (wave >= 2 && wave < 6 && lane < 32) {
        const int cb = wave - 2, cc0 = cb * 32 + lane;
        float s_ = part[cc0];
        for (int t2 = 1; t2 < nth; t2++) s_ += part[t2 * dh + cc0];         // thread order (ggml.c:5553-5577)
        if (merged) merged[col0 + cc0] = s_;
        float amax = fabsf(s_);
        amax = max_lanes_0_31(amax);
        const float dd = amax / 7.0f;
        const float id = (amax != 0.0f) ? 7.0f / amax : 0.0f;
        const uint32_t nib = (uint32_t) ((int) __builtin_rintf(s_ * id)) & 0xF;
        const int kk = lane & 7;
        const uint32_t e0 = __shfl(nib, 2 * kk), e1 = __shfl(nib, 2 * kk + 1);
        const uint32_t e2 = __shfl(nib, 16 + 2 * kk), e3 = __shfl(nib, 17 + 2 * kk);
        const int b = h * (dh / 32) + cb, cc = b >> 3, j = b & 7;
        if (lane < 8) qa_A[(cc * 8 + kk) * 8 + j] = (e0 | (e1 << 8) | (e2 << 16) | (e3 << 24)) << (4 * (j & 1));
        if (lane == 0) qa_d[b] = dd;
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

#ifndef LH_ATTN_ABLATE
#define LH_ATTN_ABLATE 0
#endif
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
        const bool nowait = (lut_math & 0x1000) != 0;     // (fault-injection test: short poll bound)
        const uint32_t tag = QKV_WAIT ? (make_tag(aa.epoch[0], aa.layer + 1)) : 0u;
        if (QKV_WAIT) load_keys();                              // in flight while the mat-vec workgroups finish
        if (tid < dh / 2) {
            const int e = 2 * tid;
            const double cs = tab[e], sn = tab[e + 1];
            const uint64_t *q2 = aa.qkv2 + h * dh, *k2 = q2 + d, *v2 = q2 + 2 * d;
            // the granules this thread needs -- 2 (q) or, in the workgroup whose slice holds the new key, 6 (q, k, v) -- are polled TOGETHER:
            // one look = all loads in flight at once, one L2 round trip (they used to be polled one after the other: six dependent
            // round trips in the one workgroup every soft_max . V workgroup of the head waits for)
            float gq0, gq1, gk0 = 0.0f, gk1 = 0.0f, gv0 = 0.0f, gv1 = 0.0f;
            if (QKV_WAIT) {
                uint64_t g_[6] = { 0, 0, 0, 0, 0, 0 };
                int spins = 0;
                for (;;) {
                    g_[0] = __hip_atomic_load(q2 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    g_[1] = __hip_atomic_load(q2 + e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (owns_new) {
                        g_[2] = __hip_atomic_load(k2 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        g_[3] = __hip_atomic_load(k2 + e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        g_[4] = __hip_atomic_load(v2 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        g_[5] = __hip_atomic_load(v2 + e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    bool ok = (uint32_t) (g_[0] >> 32) == tag && (uint32_t) (g_[1] >> 32) == tag;
                    if (owns_new) ok = ok && (uint32_t) (g_[2] >> 32) == tag && (uint32_t) (g_[3] >> 32) == tag && (uint32_t) (g_[4] >> 32) == tag && (uint32_t) (g_[5] >> 32) == tag;
                    if (ok) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (poll_give_up(spins, nowait ? (1 << 8) : (1 << 20), fault)) break;
                }
                gq0 = __builtin_bit_cast(float, (uint32_t) g_[0]); gq1 = __builtin_bit_cast(float, (uint32_t) g_[1]);
                gk0 = __builtin_bit_cast(float, (uint32_t) g_[2]); gk1 = __builtin_bit_cast(float, (uint32_t) g_[3]);
                gv0 = __builtin_bit_cast(float, (uint32_t) g_[4]); gv1 = __builtin_bit_cast(float, (uint32_t) g_[5]);
            } else {
                gq0 = q[e]; gq1 = q[e + 1];
                if (owns_new) { gk0 = kk[e]; gk1 = kk[e + 1]; gv0 = vv[e]; gv1 = vv[e + 1]; }
            }
            const double x0 = (double) gq0, x1 = (double) gq1;
            qs[e] = (float) (x0 * cs - x1 * sn);
            qs[e + 1] = (float) (x0 * sn + x1 * cs);
            if (owns_new) {
                const double k0 = (double) gk0, k1 = (double) gk1;
                const float r0 = (float) (k0 * cs - k1 * sn), r1 = (float) (k0 * sn + k1 * cs);
                kn[e] = r0; kn[e + 1] = r1;
                Kc[(size_t) n_past * d + h * dh + e] = r0;
                Kc[(size_t) n_past * d + h * dh + e + 1] = r1;
                Vc[(size_t) n_past * d + h * dh + e] = gv0;
                Vc[(size_t) n_past * d + h * dh + e + 1] = gv1;
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
        const bool nowait = (lut_math & 0x1000) != 0;
        // (a thread's granules, up to four at a time, are polled together: one round trip per look instead of one per granule)
#if LH_ATTN_ABLATE == 2        /* measurement build, results wrong: the soft_max . V workgroups do not wait for the scores (bound of the hop + soft_max) */
        for (int t0 = tid; t0 < T; t0 += nt) p[t0] = 0.0f;
        for (int t0 = T; t0 < T; t0 += 4 * nt) {
#else
        for (int t0 = tid; t0 < T; t0 += 4 * nt) {
#endif
            uint64_t g_[4] = { 0, 0, 0, 0 };
            int spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int t = t0 + i * nt;
                    if (t < T) { g_[i] = __hip_atomic_load(aa.sc2 + (size_t) h * n_ctx + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = ok && (uint32_t) (g_[i] >> 32) == tag; }
                }
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
                if (poll_give_up(spins, nowait ? (1 << 8) : (1 << 20), fault)) break;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int t = t0 + i * nt;
                if (t < T) { const float v = __builtin_bit_cast(float, (uint32_t) g_[i]); p[t] = v; mx = fmaxf(mx, v); }
            }
        }
    } else
    for (int t = tid; t < T; t += nt) { const float v = load_f32_sc1(row + t); p[t] = v; mx = fmaxf(mx, v); }
#if LH_ATTN_ABLATE == 1 || LH_ATTN_ABLATE == 2      /* measurement build, results wrong: no soft_max arithmetic between the scores and the V*P chains (bound of every soft_max restructuring) */
    if (!QKV_WAIT) {
#endif
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
#if LH_ATTN_ABLATE == 1 || LH_ATTN_ABLATE == 2
    }
#endif
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
// (register budget: 4 / 3 waves per SIMD.  The self-contained norm prologue PREP_NORM -- the FIRST layer of a later pipeline stage that takes its row
//  from a plain buffer, one launch per stage step -- needs ~20 registers more and gets one wave less instead of spilling them
//  (tools/kernel_scratch_report.py).  The mailbox prologue PREP_NORM_TAG keeps 4 / 3 and its few spilled registers: its workgroups WAIT for
//  another stage's row, and when stages share one GPU (tests, one-GPU smoke runs) the fewer slots they hold while waiting the better.)
template <int PRE, int D, int PG>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PRE == PREP_NORM ? (PG == 1 ? 3 : 2) : (PG == 1 ? 4 : 3))))
k_qkv_attn(const GemvArgs ga, const AttnXArgs aa, const int gridA, const int H) {
    extern __shared__ double smem_d[];
    const int b = blockIdx.x;
    if (b < gridA) {
        const int ncb = aa.dh / 32, wph = 3 * ncb;
        // (round 5 measured the other dispatch order -- the q and k row-groups of every head of an XCD before any v row-group: a head's score
        //  chain needs q and k, only its last V*P chain needs v -- and found no difference, 14.97 against 14.99 us per launch:
        //  profiles/r05_m_qkv_order_ab.txt; the switch is gone)
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
// (Round 2a's first version was ONE workgroup of 1024 threads doing all of the below, 39 us; the two launches further down replaced it
//  -- 17.6 us -- and it was removed in round 3.)  <= 32 values per thread (n_vocab <= 32768), order-preserving 64-bit keys:
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

// a pipeline stage that does not pick the token still has to advance its device-resident position
__global__ void k_advance(int32_t *__restrict__ st) {
    if (threadIdx.x == 0) { st[0] += 1; st[1] += 1; }
}

// batched decode step (SeqSet): one workgroup per row -- k_argmax on the row's logits with the row's slot state, trace and pick buffer
__global__ void __launch_bounds__(1024)
k_argmax_set(const float *__restrict__ logits, int V, const SeqSet *__restrict__ set) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    logits += (size_t) b * V;
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
    argmax_dpp<DPP_ROW_MIRROR>(best, idx);
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
        int32_t *st = set->state[b];
        set->trace[b][st[1]] = r;
        if (set->tok_out[b]) *set->tok_out[b] = r;
        st[0] += 1; st[1] += 1;
    }
}
__global__ void k_advance_set(const SeqSet *__restrict__ set, int n) {
    const int b = threadIdx.x;
    if (b < n) { int32_t *st = set->state[b]; st[0] += 1; st[1] += 1; }
}

hipError_t set_phase_probe(unsigned long long *dev_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase_probe), &dev_buf, sizeof(dev_buf));
}


static int pick_waves(int ngroups) {
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
        ga.pos_w = mb->pos_w; ga.patience = 3; ga.fault = mb->fault; ga.sync = mb->epoch; ga.lut_math |= mb->test_bits;
    }
#define LH_GO(D, RING) hipLaunchKernelGGL((k_gemv<PRE, EPI, D, RING, PG>), dim3(grid), dim3(nw * 64), lds + ((LH_GEMV_PAD && (RING)) ? (D) * 288 : 0), st, ga)
    if (np.out && grid > NORM_PART_MAX) return hipErrorInvalidValue;
    // rows that fit 16 slots: whole row in flight (latency-bound small matrices) unless the launch
    // already has >= 4 waves per CU, where an 8-deep ring saves 48 VGPRs and keeps 4 waves/SIMD resident
    if (w.nchunks <= 16 && !(w.nchunks == 16 && w.ngroups >= 1024)) {
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
    // Workgroups of ngroups / 256 waves (1, 2 or 4): ONE workgroup per CU where the matrix has fewer than 1024
    // row-groups.  Measured on the 7B decode step: w2 (512 row-groups) as 256 x 2 waves with the 12-granule operand
    // budget 7.60 us, as 128 x 4 waves with the 4-granule budget 8.52 us, as 512 x 1 wave 7.99 us; wo as 256 x 2 waves
    // 5.15 us against 5.41 us as 512 x 1 (profiles/r02_d_small_matvec_ab.txt).  The operand budget (4 or 12 granules
    // of 16 B per thread) follows from the workgroup size, not the other way round.
    int nw = w.ngroups >= 1024 ? 4 : w.ngroups >= 512 ? 2 : 1;
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


hipError_t launch_gemv_pick(const QMat &w, const float *in0, const float *in1, float *y, const uint16_t *T_silu, hipStream_t st,
                            const NormPart *npp, const PickIO &pick) {
    static const int norm_mode = getenv("LLAMAHIP_NORM_MODE") ? atoi(getenv("LLAMAHIP_NORM_MODE")) : 2;
    NormPart np = npp ? *npp : NormPart();
    const bool normp = norm_mode >= 2 && np.in && np.n_in > 0 && np.n_in <= NORM_PART_MAX;
    if (!normp) { np.in = nullptr; np.n_in = norm_mode == 0 ? -1 : 0; }
    // the workgroup shape of launch_gemv_t's fp32 prologues; the embedding in the epilogue is written for 256 threads
    int nw = pick_waves(w.ngroups);
    const int need = w.K / 16;
    while (nw < 4 && need > nw * 64) nw *= 2;
    const int pg = need <= nw * 64 ? 1 : need <= 2 * nw * 64 ? 2 : 0;
    if (nw != 4 || !pg || (w.nchunks <= 16 && !(w.nchunks == 16 && w.ngroups >= 1024))) return hipErrorInvalidValue;      // (the caller checks gemv_pick_applies)
    const int D = pick_depth(w.nchunks, w.ngroups);
    const int grid = (w.ngroups + nw - 1) / nw;
    size_t lds = (size_t) w.nchunks * 64 * 4 + (size_t) w.nchunks * 8 * 4 + 32 * sizeof(double);
    lds = ((lds + 15) & ~(size_t) 15) + (size_t) D * 288;
    GemvArgs ga = { w.tiles, w.ngroups, w.nchunks, w.M, w.gmapF8, nullptr, nullptr, in0, in1, w.K, y, nullptr, T_silu, nullptr, nullptr,
                    (const f64x2 *) np.in, np.n_in, nullptr, nullptr, 0, 0, g_lut_math };
    ga.pick = pick;
#define LH_GOP(PRE, DD, PG) hipLaunchKernelGGL((k_gemv<PRE, EPI_STORE_PICK, DD, true, PG>), dim3(grid), dim3(nw * 64), lds, st, ga)
#define LH_GOPD(PRE, PG) { if (D == 4) LH_GOP(PRE, 4, PG); else if (D == 8) LH_GOP(PRE, 8, PG); else if (D == 10) LH_GOP(PRE, 10, PG); else return hipErrorInvalidValue; }
    if (normp) { if (pg == 1) LH_GOPD(PREP_NORMP, 1) else LH_GOPD(PREP_NORMP, 2) }
    else { if (pg == 1) LH_GOPD(PREP_NORM, 1) else LH_GOPD(PREP_NORM, 2) }
#undef LH_GOPD
#undef LH_GOP
    LH_LAUNCH_CHECK();
    return hipSuccess;
}
// w1|w3 in half-block workgroups (EPI_SILU_QAH, llamahip_internal.h)
// (QUARTER-block workgroups of two waves -- VERDICT r04 item 4c: 13B's 432 blocks would go from max / mean 1.19 to 1.04 workgroups per CU --
//  were built in round 5, parity green, and measured: w1|w3 at 13B 16.60 us in block workgroups, 21.48 in halves, 25.01 in quarters; decode
//  466 / 427 / 405 tokens/s (profiles/r05_q_w13_quarter_ab.txt).  Every workgroup repeats the norm -> Q4_0 prologue of the whole row, so
//  halving the workgroup doubles that work per CU, and at K = 5120 a 128-thread workgroup needs three prologue passes.  Removed.)
bool gemv_silu_half_applies(const QMat &w) {
    static const bool off = getenv("LLAMAHIP_NO_W13_HALF") != nullptr;
    static const bool force = getenv("LLAMAHIP_W13_HALF") && atoi(getenv("LLAMAHIP_W13_HALF")) == 1;      // tests: shapes the balance rule leaves with block workgroups
    static const int ncu = [] { int dev = 0; hipDeviceProp_t p; return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256; }();
    if (off || !w.gmapF8 || w.ngroups % 8 != 0 || w.K / 16 > 512 || w.nchunks <= 4 || pick_depth(w.nchunks, w.ngroups) != 4 || w.ngroups < 2048) return false;
    if (force) return true;
    // halves only where they leave the busiest CU relatively less to stream than whole blocks do: max / mean workgroups per CU
    // (7B: 344 blocks 2 / 1.34 = 1.49 against 688 halves 3 / 2.69 = 1.12; 13B 432 / 864 and 65B 688 / 1 376: equal, blocks stay)
    const double nb = w.ngroups / 8.0;
    const double r_blocks = std::ceil(nb / ncu) / (nb / ncu), r_halves = std::ceil(2.0 * nb / ncu) / (2.0 * nb / ncu);
    return r_halves < r_blocks - 0.02;
}
hipError_t launch_gemv_silu_half(const QMat &w, const float *in0, const float *in1, const uint16_t *T_silu, uint32_t *out_A, float *out_d, hipStream_t st,
                                 const NormPart *npp, uint64_t *amax_t, uint32_t *epoch, int layer, uint32_t *fault) {
    static const int norm_mode = getenv("LLAMAHIP_NORM_MODE") ? atoi(getenv("LLAMAHIP_NORM_MODE")) : 2;
    static const int fault_test = (getenv("LLAMAHIP_HANDOFF_FAULT_TEST") && atoi(getenv("LLAMAHIP_HANDOFF_FAULT_TEST")) == 6) ? 0x1000 : 0;
    NormPart np = npp ? *npp : NormPart();
    const bool normp = norm_mode >= 2 && np.in && np.n_in > 0 && np.n_in <= NORM_PART_MAX;
    if (!normp) { np.in = nullptr; np.n_in = norm_mode == 0 ? -1 : 0; }
    const int nblocks = w.ngroups / 8, grid = (nblocks + 7) / 8 * 16;       // two halves per block, blocks of one XCD 16 apart
    const int need = w.K / 16, pg = need <= 256 ? 1 : 2;
    size_t lds = (size_t) w.nchunks * 64 * 4 + (size_t) w.nchunks * 8 * 4 + 32 * sizeof(double);
    lds = ((lds + 15) & ~(size_t) 15) + (size_t) 4 * 288;
    GemvArgs ga = { w.tiles, w.ngroups, w.nchunks, w.M, w.gmapF8, nullptr, nullptr, in0, in1, w.K, nullptr, nullptr, T_silu, out_A, out_d,
                    (const f64x2 *) np.in, np.n_in, nullptr, epoch, 0, layer, g_lut_math | fault_test, fault };
    ga.out_t = amax_t;
#define LH_GOH(PRE, PG) hipLaunchKernelGGL((k_gemv<PRE, EPI_SILU_QAH, 4, true, PG>), dim3(grid), dim3(256), lds, st, ga)
    if (normp) { if (pg == 1) LH_GOH(PREP_NORMP, 1); else LH_GOH(PREP_NORMP, 2); }
    else { if (pg == 1) LH_GOH(PREP_NORM, 1); else LH_GOH(PREP_NORM, 2); }
#undef LH_GOH
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

// the shapes launch_gemv_pick is instantiated for (ring kernels with 4-wave workgroups: every LLaMA lm head)
bool gemv_pick_applies(const QMat &w) {
    int nw = pick_waves(w.ngroups);
    const int need = w.K / 16;
    while (nw < 4 && need > nw * 64) nw *= 2;
    const int pg = need <= nw * 64 ? 1 : need <= 2 * nw * 64 ? 2 : 0;
    if (nw != 4 || !pg || (w.nchunks <= 16 && !(w.nchunks == 16 && w.ngroups >= 1024))) return false;
    const int D = pick_depth(w.nchunks, w.ngroups);
    return D == 4 || D == 8 || D == 10;
}

// Short prompt chunk (see k_decn_scores): scores -> soft_max + V*P + ordered combine + Q4_0 quantization of the
// merged rows straight into the QA operand of the wo mat-mul (no separate preparation launch).
//   sc : scratch of N * H * n_ctx floats
hipError_t launch_attn_short(const float *qr, const float *Kc, const float *Vc, float *sc, float *merged,
                             uint32_t *qa_A, float *qa_d, int n_past, int N, int d, int H, int n_ctx, int nth,
                             const uint16_t *T_exp, hipStream_t st, int chunk, const SeqSet *set, int set_keys) {
    // (set: positions live on the device -- the key slices up to `set_keys`, the host's bound on every row's position + 1 over the life of the
    //  captured step (0: n_ctx), are launched; those beyond a row's position return at once)
    const int dh = d / H, T = set ? (set_keys > 0 && set_keys < n_ctx ? set_keys : n_ctx) : n_past + N;
    const float kq_scale = 1.0f / sqrtf((float) d / (float) H);          // .mm:620
    const int Kp = (d + 255) / 256 * 256;
    hipLaunchKernelGGL(k_decn_scores, dim3(H, (T + DEC_TS - 1) / DEC_TS, N), dim3(256), 0, st, qr, d, dh, Kc, sc, n_ctx, kq_scale, n_past, set);
    LH_LAUNCH_CHECK();
    const int nt = (32 * (nth < 32 ? nth : 32) + 63) / 64 * 64;
    const size_t lds = 32 * sizeof(double) + ((size_t) n_ctx + (size_t) nth * 32 + 16) * sizeof(float);
    hipLaunchKernelGGL(k_dec_pv_blk<true>, dim3(H, dh / 32, N), dim3(nt), lds, st, sc, Vc, d, dh, n_ctx, nth, merged, qa_A, qa_d, T_exp,
                       (const int32_t *) nullptr, n_past, (long) Kp / 4, (long) Kp / 32, g_lut_math, chunk, set);
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

// the streaming soft_max . V (k_dec_pv_stream): one chain owner per (thread of the split, column) in a 1 024-thread workgroup,
// the score row in 4 registers per thread, 16-byte column quads
// (LLAMAHIP_HANDOFF_FAULT_TEST=4: the polls of k_dec_pv_stream give up after 256 looks and its publishers use a tag nobody waits for)
static const int g_pv_fault_test = (getenv("LLAMAHIP_HANDOFF_FAULT_TEST") && atoi(getenv("LLAMAHIP_HANDOFF_FAULT_TEST")) == 4) ? 0x1000 : 0;
static bool th_split_ok(int nth, int split, int cpw) { return split >= 1 && cpw >= 1 && (split - 1) * cpw < nth; }
bool pv_stream_applies(int dh, int n_ctx, int nth) { return nth >= 1 && nth <= 32 && n_ctx <= 4096 && dh % 32 == 0; }

hipError_t launch_dec_attn(const float *qkv, int d, int H, int n_ctx, int nth, const double *tab, float *Kc, float *Vc,
                           float *sc, float *part, float *merged, uint32_t *qa_A, float *qa_d,
                           const uint16_t *T_exp, const int32_t *state, hipStream_t st, uint32_t *xsync, uint32_t *fault, bool long_ctx,
                           uint64_t *xpart, const uint32_t *epoch, int layer) {
    const int dh = d / H;
    const float kq_scale = 1.0f / sqrtf((float) d / (float) H);          // .mm:620
    (void) part;
    // long contexts (the caller knows the position): scores, then the streaming soft_max . V
    if (long_ctx && pv_stream_applies(dh, n_ctx, nth)) {
        const int nsl = (n_ctx + DEC_TS - 1) / DEC_TS;
        hipLaunchKernelGGL(k_dec_scores, dim3(H, nsl), dim3(256), 2 * dh * sizeof(float), st, qkv, d, dh, tab, Kc, Vc, sc, n_ctx, kq_scale, state);
        LH_LAUNCH_CHECK();
        // chains of a (head, column block) over `split` workgroups until the chip is full (needs the XCD placement: xpart is only
        // passed when the load-time self-test confirmed it); every workgroup must own at least one chain
        static const int split_env = getenv("LLAMAHIP_PV_SPLIT") ? atoi(getenv("LLAMAHIP_PV_SPLIT")) : 0;
        const int W = H * (dh / 32);
        auto valid = [&](int s) { const int cpw = (nth + s - 1) / s; return s >= 1 && s <= nth && (s - 1) * cpw < nth; };
        int split = 1;
        if (xpart && epoch && fault && W % 8 == 0) {
            if (split_env > 0) { if (valid(split_env)) split = split_env; }
            else while (W * split * 2 <= 256 && valid(split * 2)) split *= 2;       // (13B: 160 workgroups stay unsplit -- 320 would need two rounds: 322 against 298 tokens/s at 2 048 keys)
        }
        const int cpw = (nth + split - 1) / split;
        constexpr int nl = 4;                   // 16-byte loads per thread and stage: 64 KB stages, 128 KB of LDS, one workgroup per CU
        // (LLAMAHIP_PV_STAGE_ROWS shortens the stages so that small test contexts run many of them; tests only)
        static const int sr_cap = getenv("LLAMAHIP_PV_STAGE_ROWS") ? std::max(1, atoi(getenv("LLAMAHIP_PV_STAGE_ROWS"))) : 1 << 20;
        const int SR = std::min(nl * 128 / cpw, sr_cap);
        const size_t lds = 32 * sizeof(double) + ((size_t) ((n_ctx + 3) & ~3) + (size_t) nth * 32) * sizeof(float) + (size_t) 2 * cpw * SR * 128;
        // the LDS-DMA variant (k_dec_pv_dma): head size 128, a workgroup = (head, up to 3 chunks of the key split) over all 128 columns,
        // the heads' chunks split over as many workgroups as fill the chip once; LLAMAHIP_PV_DMA=0 keeps k_dec_pv_stream
        static const bool no_dma = getenv("LLAMAHIP_PV_DMA") && atoi(getenv("LLAMAHIP_PV_DMA")) == 0;
        if (!no_dma && dh == 128 && xpart && epoch && fault && H % 8 == 0 && !getenv("LLAMAHIP_PV_STAGE_ROWS")) {
            int sp = 1;
            if (split_env > 0 && valid(split_env)) sp = split_env;
            else while (H * sp * 2 <= 256 && valid(sp * 2)) sp *= 2;
            const int cpwd = (nth + sp - 1) / sp;
            if (cpwd <= 3) {
                const size_t fixed = 32 * sizeof(double) + ((size_t) ((n_ctx + 3) & ~3) + (size_t) nth * 128) * sizeof(float) + 64;
                int NS = (int) (((size_t) 156 * 1024 - fixed) / ((size_t) cpwd * 4096));
                NS = std::max(2, std::min(NS, 256));
                const size_t ldsd = fixed + (size_t) NS * cpwd * 4096;
                hipLaunchKernelGGL(k_dec_pv_dma, dim3(H * sp), dim3(512), ldsd, st, sc, Vc, d, n_ctx, nth, NS, merged, qa_A, qa_d, T_exp, state, g_lut_math | g_pv_fault_test, H, sp, xpart, epoch, layer, fault);
                LH_LAUNCH_CHECK();
                return hipSuccess;
            }
        }
        hipLaunchKernelGGL(k_dec_pv_stream<4>, dim3(W * split), dim3(1024), lds, st, sc, Vc, d, dh, n_ctx, nth, SR, merged, qa_A, qa_d, T_exp, state, g_lut_math | g_pv_fault_test, H, split, xpart, epoch, layer, fault);
        LH_LAUNCH_CHECK();
        return hipSuccess;
    }
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
    hipLaunchKernelGGL(k_dec_pv_blk<false>, dim3(H, dh / 32), dim3(nt), lds, st, sc, Vc, d, dh, n_ctx, nth, merged, qa_A, qa_d, T_exp, state, 0, 0L, 0L, g_lut_math, 0, (const SeqSet *) nullptr);
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
    // test only (tests/test_gpu_parity.py): the mat-vec role publishes a wrong tag and every poll gives up after 256 looks -> the
    // sticky fault word must come back as an error
    static const int fault_test = (getenv("LLAMAHIP_HANDOFF_FAULT_TEST") && atoi(getenv("LLAMAHIP_HANDOFF_FAULT_TEST")) < 2) ? 0x1000 : 0;     // (2: the wo launch of the overlapped schedule misbehaves instead)
    GemvArgs ga = { w.tiles, w.ngroups, w.nchunks, w.M, w.gmapF8, nullptr, nullptr, x, norm_w, w.K, (float *) qkv2, nullptr, T_silu, nullptr, nullptr,
                    (const f64x2 *) (normp ? np.in : nullptr), normp ? np.n_in : (norm_mode == 0 ? -1 : 0), nullptr, epoch, 0, layer, g_lut_math | fault_test, fault };
    if (x_t) { ga.in_t = x_t; ga.slot_in = 0; ga.part_in = nullptr; ga.npart = norm_mode == 0 ? -1 : 0; ga.pos_w = mb->pos_w; ga.patience = 3; ga.lut_math |= mb->test_bits; }
    const AttnXArgs aa = { nullptr, d, dh, tab, Kc, Vc, nullptr, n_ctx, nth, kq_scale, merged, qa_A, qa_d, T_exp, state, nullptr, fault, g_lut_math | fault_test,
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


// The selection in two launches (round 2; the single-workgroup first version, k_topk_candidates, was removed in round 3): that was ONE workgroup, i.e. 16 waves x ~5 000 instructions of
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
    // with the k-th are among the survivors -- and only a few more (the minimum of the maxima, as the single-workgroup version used, lets
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
    if (!ws) return hipErrorInvalidValue;
    unsigned long long *keys = (unsigned long long *) ws, *gmax = keys + 32768;
    uint32_t *badw = (uint32_t *) (gmax + 64);
    hipLaunchKernelGGL(k_topk_keys, dim3((V + 1023) / 1024), dim3(1024), 0, st, logits, V, window, n_window, scale, repeat_penalty, keys, gmax, badw);
    LH_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_topk_select, dim3(1), dim3(1024), 0, st, V, k, keys, gmax, badw, out_score, out_id, flags);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_argmax(const float *logits, int V, int32_t *out, int out_idx, int32_t *next_token, int32_t *state, hipStream_t st, uint64_t *token_mb) {
    hipLaunchKernelGGL(k_argmax, dim3(1), dim3(1024), 0, st, logits, V, out, out_idx, next_token, state, token_mb);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}


hipError_t launch_argmax_set(const float *logits, int V, const SeqSet *set, int n, hipStream_t st) {
    hipLaunchKernelGGL(k_argmax_set, dim3(n), dim3(1024), 0, st, logits, V, set);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_advance_set(const SeqSet *set, int n, hipStream_t st) {
    hipLaunchKernelGGL(k_advance_set, dim3(1), dim3(64), 0, st, set, n);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_advance(int32_t *state, hipStream_t st) {
    hipLaunchKernelGGL(k_advance, dim3(1), dim3(64), 0, st, state);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}


hipError_t init_attrs_prep();
hipError_t init_attrs_prompt_attn();
hipError_t init_attrs_decode();
hipError_t init_kernel_attrs() {
    hipError_t e;
    if ((e = init_attrs_prep()) != hipSuccess) return e;
    if ((e = init_attrs_decode()) != hipSuccess) return e;
    if ((e = init_attrs_gemv_set()) != hipSuccess) return e;
    return init_attrs_prompt_attn();
}

hipError_t init_attrs_decode() {
    const int cap = 160 * 1024;          // fused prologues / wide rows need more than the default 64 KB of dynamic LDS
#define LH_ATTR(KERNEL) do { hipError_t e_ = hipFuncSetAttribute((const void *) KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, cap); if (e_ != hipSuccess) return e_; } while (0)
#define LH_ATTR_G1(PRE, EPI, PG) LH_ATTR((k_gemv<PRE, EPI, 16, false, PG>)); LH_ATTR((k_gemv<PRE, EPI, 16, true, PG>)); LH_ATTR((k_gemv<PRE, EPI, 22, true, PG>)); \
    LH_ATTR((k_gemv<PRE, EPI, 18, true, PG>)); LH_ATTR((k_gemv<PRE, EPI, 14, true, PG>)); LH_ATTR((k_gemv<PRE, EPI, 10, true, PG>)); LH_ATTR((k_gemv<PRE, EPI, 8, true, PG>)); LH_ATTR((k_gemv<PRE, EPI, 4, true, PG>))
    LH_ATTR_G1(PRE_QA, EPI_STORE, 4); LH_ATTR_G1(PRE_QA, EPI_STORE, 12); LH_ATTR_G1(PRE_QA, EPI_RESID, 4); LH_ATTR_G1(PRE_QA, EPI_RESID, 12);
    LH_ATTR_G1(PREP_NORM, EPI_STORE, 1); LH_ATTR_G1(PREP_NORM, EPI_STORE, 2); LH_ATTR_G1(PREP_PLAIN, EPI_RESID, 1); LH_ATTR_G1(PREP_PLAIN, EPI_RESID, 2);
    LH_ATTR_G1(PREP_SILU_MUL, EPI_RESID, 1); LH_ATTR_G1(PREP_NORM, EPI_SILU_QA, 1);
    LH_ATTR_G1(PREP_NORMP, EPI_STORE, 1); LH_ATTR_G1(PREP_NORMP, EPI_STORE, 2); LH_ATTR_G1(PREP_NORMP, EPI_SILU_QA, 1);
    LH_ATTR_G1(PRE_QA, EPI_SILU_QA, 1);
    LH_ATTR((k_gemv<PREP_NORMP, EPI_SILU_QAH, 4, true, 1>)); LH_ATTR((k_gemv<PREP_NORMP, EPI_SILU_QAH, 4, true, 2>)); LH_ATTR((k_gemv<PREP_NORM, EPI_SILU_QAH, 4, true, 1>)); LH_ATTR((k_gemv<PREP_NORM, EPI_SILU_QAH, 4, true, 2>));
    LH_ATTR((k_gemv<PREP_NORMP, EPI_STORE_PICK, 4, true, 1>)); LH_ATTR((k_gemv<PREP_NORMP, EPI_STORE_PICK, 8, true, 1>)); LH_ATTR((k_gemv<PREP_NORMP, EPI_STORE_PICK, 10, true, 1>));
    LH_ATTR((k_gemv<PREP_NORMP, EPI_STORE_PICK, 4, true, 2>)); LH_ATTR((k_gemv<PREP_NORMP, EPI_STORE_PICK, 8, true, 2>)); LH_ATTR((k_gemv<PREP_NORMP, EPI_STORE_PICK, 10, true, 2>));
    LH_ATTR((k_gemv<PREP_NORM, EPI_STORE_PICK, 4, true, 1>)); LH_ATTR((k_gemv<PREP_NORM, EPI_STORE_PICK, 8, true, 1>)); LH_ATTR((k_gemv<PREP_NORM, EPI_STORE_PICK, 10, true, 1>));
    LH_ATTR((k_gemv<PREP_NORM, EPI_STORE_PICK, 4, true, 2>)); LH_ATTR((k_gemv<PREP_NORM, EPI_STORE_PICK, 8, true, 2>)); LH_ATTR((k_gemv<PREP_NORM, EPI_STORE_PICK, 10, true, 2>));
    LH_ATTR_G1(PREP_NORM_TAG, EPI_STORE, 1); LH_ATTR_G1(PREP_NORM_TAG, EPI_STORE, 2); LH_ATTR_G1(PRE_QA, EPI_RESID_TAG, 4); LH_ATTR_G1(PRE_QA, EPI_RESID_TAG, 12);
#undef LH_ATTR_G1
    LH_ATTR(k_dec_pv_blk<false>); LH_ATTR(k_dec_pv_blk<true>); LH_ATTR(k_dec_pv_stream<4>); LH_ATTR(k_dec_pv_dma); LH_ATTR(k_dec_attn_x); LH_ATTR((k_qkv_attn<PREP_NORMP, 8, 1>)); LH_ATTR((k_qkv_attn<PREP_NORM, 8, 1>)); LH_ATTR((k_qkv_attn<PREP_NORMP, 10, 2>)); LH_ATTR((k_qkv_attn<PREP_NORM, 10, 2>)); LH_ATTR((k_qkv_attn<PREP_NORMP, 4, 2>)); LH_ATTR((k_qkv_attn<PREP_NORM, 4, 2>));
    LH_ATTR((k_qkv_attn<PREP_NORM_TAG, 8, 1>)); LH_ATTR((k_qkv_attn<PREP_NORM_TAG, 10, 2>)); LH_ATTR((k_qkv_attn<PREP_NORM_TAG, 4, 2>));
#undef LH_ATTR
    return hipSuccess;
}

}  // namespace lh
