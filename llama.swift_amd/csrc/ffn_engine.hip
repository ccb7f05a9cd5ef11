// ffn_engine.hip -- the feed-forward half of a decode layer as ONE persistent launch (round 4):
//
//     wo mat-vec -> + residual -> norm * w -> Q4_0 -> w1 | w3 mat-vec -> SiLU(gate) * up -> Q4_0 -> w2 mat-vec -> + residual
//     (.mm:649-690; ggml.c:5987-6285 mul_mat_q4_0_f32, :1415-1466 vec_dot_q4_0, :5327-5385 norm, :456-523 quantize_row_q4_0,
//      :1956-1963 silu)
//
// One workgroup per CU (`G` of them), 5 waves: wave 0 is the LOADER, waves 1-4 are CONSUMERS.
//   * The loader streams this CU's share of the three matrices, in the order the consumers will need it, HBM -> LDS with LDS-DMA
//     (global_load_lds_dwordx4 ... nt, 1 KiB per wave instruction) into a ring of `S` QUADS (5 KiB = four 1280-byte decode tiles:
//     4 x 1 KiB nibbles, then 4 x 256 B scales).  It never waits for a dependency edge: while the consumers wait for the
//     all-to-all hand-off of an operator's result it keeps landing the NEXT operator's weights until the ring is full.
//     Flow control is two kinds of LDS words: `landed` (quads whose DMA has completed: the loader's own vmcnt) and `done[slot]`
//     (the consumer that finished a quad releases its slot).
//   * A consumer wave owns one row-group (8 rows x 8 chains, kcommon.hip.h) at a time and runs the decode mat-vec's arithmetic
//     (v_dot8_i32_i4, the block-ordered v_fmac_f32_dpp chain of gemv_body) with the weights coming from the ring instead of a
//     register ring.  Arithmetic order per output is exactly gemv_body's, i.e. the reference's.
//   * Work split: wo / w2 row-group g belongs to workgroup g % G (so the rows a workgroup finishes in w2 are the rows whose
//     residual operand it produced in wo); w1|w3 is split in UNITS (gate row-group u + up row-group u: 8 FFN activations),
//     contiguous and balanced to +-1 unit.  A Q4_0 block of the FFN activation (4 units) may straddle workgroups: every unit
//     publishes its partial amax, fmaxf is exact in any order, so every owner quantizes its 8 values with the block's amax.
//   * Hand-offs are the data-tagged 8-byte granules of k_qkv_attn ({payload, tag}, one write-through store, polled with L1-bypassing
//     loads): edge 1 = the row h = x + wo.attn (d granules), edge 2 = the FFN activation's nibbles (F/8 granules, 8 nibbles
//     each) + block scales (F/32).  tag = (epoch of the forward pass, layer + 1).  Every poll is bounded; one that runs out raises
//     the sticky fault word (-> PredictionFailed), it never hangs.
//   * The consumers synchronise among themselves through an LDS counter (the loader takes part in no barrier after kernel entry).
//
// The norm statistics are the reference's two-pass form on the gathered row (mean, then sum (x - mean)^2, both in double; only the
// association order of the double additions differs from the reference's sequential loop -- DESIGN.md "norm").
//
// LDS-DMA notes (MI355X_MICROARCH.md): M0 carries the LDS destination; a ds_read is ordered behind a pending DMA only by the
// issuing wave's vmcnt, hence `landed` is published by the loader after its own s_waitcnt.
#include "kcommon.hip.h"

namespace lh {

constexpr int QUAD_BYTES = 5120;      // 4 tiles: [4][1024 B nibbles][4][256 B scales]
constexpr int ENG_F = 11;             // quads that may be in flight behind the one whose landing is awaited (5 DMA instructions each; vmcnt holds 63)
constexpr int ENG_CW = 4;             // consumer waves
constexpr int ENG_NT = 64 * (1 + ENG_CW);

#if LH_PHASE_PROBE == 3
__device__ unsigned long long *g_phase_probe_eng = nullptr;
#define ENG_STAMP(ARR, IDX) do { ARR[IDX] = __builtin_readcyclecounter(); } while (0)
#else
#define ENG_STAMP(ARR, IDX) do { } while (0)
#endif

struct FfnEngArgs {
    const uint8_t *eng;                 // this layer's weights in engine order: quad q of workgroup c at ((size_t) q * G + c) * QUAD_BYTES
    int G;                              // workgroups
    int d, F;                           // n_embd, n_ff
    int ncd, nqd, ncF, nqF;             // chunks (256 columns) and quads (4 chunks) of K = d / K = F
    int R, U;                           // row-groups of wo / w2 (d / 8), units of w1|w3 (F / 8)
    int S;                              // ring slots (quads)
    int maxrg, maxu;                    // bounds of row-groups / units per workgroup (LDS carve)
    const int32_t *utab;                // [G][maxu + 1]: {n, unit ids in processing order} of each workgroup (ffn_engine_geometry)
    const uint32_t *qa_A; const float *qa_d;    // QA operand of wo (the attention output, K = d)
    const float *x_in; float *x_out;    // residual stream into / out of the layer (may be the same buffer: a workgroup reads and writes its own rows only)
    const float *norm_w;                // ffn_norm
    const uint16_t *T_silu; int lut_math;      // bit 0: evaluate SiLU instead of gathering; 0x1000: fault-injection test (short polls, wrong edge-1 tag)
    uint64_t *h_t, *amax_t, *act_t, *d2_t;      // tagged hand-off buffers: [d], [U], [U], [F / 32]
    const uint32_t *epoch; int layer;
    f64x2 *part_out;                    // [G] {sum y, sum y^2} of this workgroup's output rows (the next norm's statistics, PREP_NORMP)
    uint32_t *fault;
};

typedef __attribute__((address_space(3))) uint32_t lds_u32;

__device__ __forceinline__ uint32_t ld_acq(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_rel(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint64_t ld_granule(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// one 1 KiB piece HBM -> LDS: lane l moves 16 bytes from gsrc + 16 l to lds_dst + 16 l (M0 = destination, restored afterwards)
__device__ __forceinline__ void dma_1k(uint32_t lds_dst, uint64_t gsrc, uint32_t voff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(gsrc) : "memory");
}

// the tail of the loader: the last ENG_F quads land one after the other
template <int K>
__device__ __forceinline__ void drain_tail(uint32_t *landed, uint32_t &pub, int Q) {
    wait_vmcnt<5 * K>();
    if (Q - K > (int) pub) { pub = (uint32_t) (Q - K); st_rel(landed, pub); }
    if constexpr (K > 0) drain_tail<K - 1>(landed, pub, Q);
}

struct EngLds {
    uint8_t *ring;                      // S quads
    uint32_t *qa1A; float *qa1D;        // QA operand with K = d: first wo's (copied), then w1|w3's (norm -> quantize)
    uint8_t *un;                        // union: fp32 row h [d]  |  QA operand of w2 {A [nqF * 4 * 64 dwords], d [nqF * 32]}
    float *hown;                        // [maxrg][8] this workgroup's rows of h (w2's residual operand)
    float *actb; float *amaxb;          // [maxu][8] SiLU(gate) * up of this workgroup's units, [maxu] their partial amax
    float *gub;                         // [maxu][2][8] gate / up results of a unit (its two row-groups run on different waves)
    uint32_t *ucnt;                     // [maxu] row-groups of the unit finished (the second finisher completes the unit)
    double *red;                        // [2][ENG_CW] reduction scratch
    uint32_t *landed, *cbar, *done;     // flags: quads landed | consumer barrier counter | [S] release words
};

// consumer-only barrier: every consumer wave adds one, all wait for the round's total (LDS operations of a wave are performed in
// order, so what a wave wrote to LDS before its add is there when another wave sees the add)
__device__ __forceinline__ void cons_barrier(uint32_t *ctr, uint32_t &round, int lane, uint32_t *fault) {
    round += ENG_CW;
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    int spins = 0;
    while (ld_acq(ctr) < round) { __builtin_amdgcn_s_sleep(0); if (poll_give_up(spins, 1 << 24, fault)) break; }      // (bounded like every wait of the launch)
}

// ---- one row-group (8 rows x 8 chains) against the QA operand (A, D) in LDS; weights = quads [qb, qb + nq) of the ring ----
// Arithmetic of gemv_body's LH_CONSUME (ggml.c:1415-1466): per block the exact integer dot, then acc = fma(d_w * d_a, isum, acc) in
// block order on this lane's chain.  Operands of the next chunk are fetched while the current one is computed.
struct ChunkOps { u32x4 w; f32x2 sw; u32x4 a0, a1; float dl, dh; };

__device__ __forceinline__ void load_chunk(ChunkOps &o, const uint8_t *quad, int t, const uint32_t *A, const float *D, int ch, int lane) {
    const int k = lane & 7, tq = lane & 3;
    o.w = *(const u32x4 *) (quad + t * 1024 + lane * 16);
    o.sw = *(const f32x2 *) (quad + 4096 + t * 256 + ((lane >> 3) * 8 + tq * 2) * 4);
    const u32x4 *pa = (const u32x4 *) (A + (ch * 8 + k) * 8);
    o.a0 = pa[0]; o.a1 = pa[1];
    o.dl = D[ch * 8 + tq]; o.dh = D[ch * 8 + 4 + tq];
}
__device__ __forceinline__ void compute_chunk(float &acc, const ChunkOps &o) {
    const float plo_ = o.sw.x * o.dl, phi_ = o.sw.y * o.dh;
    const int i0_ = __builtin_amdgcn_sdot8((int) o.w.x, (int) o.a0.x, 0x4B400000, true);
    const int i1_ = __builtin_amdgcn_sdot8((int) o.w.x, (int) o.a0.y, 0x4B400000, true);
    const int i2_ = __builtin_amdgcn_sdot8((int) o.w.y, (int) o.a0.z, 0x4B400000, true);
    const int i3_ = __builtin_amdgcn_sdot8((int) o.w.y, (int) o.a0.w, 0x4B400000, true);
    const int i4_ = __builtin_amdgcn_sdot8((int) o.w.z, (int) o.a1.x, 0x4B400000, true);
    const int i5_ = __builtin_amdgcn_sdot8((int) o.w.z, (int) o.a1.y, 0x4B400000, true);
    const int i6_ = __builtin_amdgcn_sdot8((int) o.w.w, (int) o.a1.z, 0x4B400000, true);
    const int i7_ = __builtin_amdgcn_sdot8((int) o.w.w, (int) o.a1.w, 0x4B400000, true);
    const f32x2 mg_ = { 12582912.0f, 12582912.0f };
    const f32x2 q01_ = f32x2{ __builtin_bit_cast(float, i0_), __builtin_bit_cast(float, i1_) } - mg_;
    const f32x2 q23_ = f32x2{ __builtin_bit_cast(float, i2_), __builtin_bit_cast(float, i3_) } - mg_;
    const f32x2 q45_ = f32x2{ __builtin_bit_cast(float, i4_), __builtin_bit_cast(float, i5_) } - mg_;
    const f32x2 q67_ = f32x2{ __builtin_bit_cast(float, i6_), __builtin_bit_cast(float, i7_) } - mg_;
    LH_FMAC8_DPP(acc, plo_, phi_, q01_, q23_, q45_, q67_);
}

struct EngCtx {
    EngLds L; int S; uint32_t landed_seen; uint32_t *fault; int limit;
    __device__ __forceinline__ bool is_landed(int q) {
        if ((uint32_t) q < landed_seen) return true;
        landed_seen = ld_acq(L.landed);
        return (uint32_t) q < landed_seen;
    }
    __device__ __forceinline__ void wait_landed(int q) {
        int spins = 0;
        while (!is_landed(q)) { __builtin_amdgcn_s_sleep(1); if (poll_give_up(spins, limit, fault)) break; }
    }
};

// Two chunks ahead: operand set t belongs to chunk t of the current quad; while chunk t is computed the loads of chunk t + 2 are in
// flight (the next quad's chunks 0 / 1 once it has landed).  The release of a slot is a plain LDS store: a wave's LDS operations
// are performed in order, so the store follows the reads of the quad it frees.
__device__ __forceinline__ void release_slot(uint32_t *done, uint32_t v) {
    asm volatile("" ::: "memory");
    __hip_atomic_store(done, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ float run_rowgroup(EngCtx &cx, int qb, int nq, const uint32_t *A, const float *D, int lane) {
    float acc = 0.0f;
    ChunkOps o0, o1, o2, o3;
    int slot = qb % cx.S;
    cx.wait_landed(qb);
    const uint8_t *quad = cx.L.ring + slot * QUAD_BYTES;
    load_chunk(o0, quad, 0, A, D, 0, lane);
    load_chunk(o1, quad, 1, A, D, 1, lane);
    for (int qq = 0; qq < nq; qq++) {
        const int ch = qq * 4;
        load_chunk(o2, quad, 2, A, D, ch + 2, lane);
        compute_chunk(acc, o0);
        load_chunk(o3, quad, 3, A, D, ch + 3, lane);
        compute_chunk(acc, o1);
        const int nslot = slot + 1 == cx.S ? 0 : slot + 1;
        const uint8_t *nquad = cx.L.ring + nslot * QUAD_BYTES;
        const bool more = qq + 1 < nq;
        if (more && cx.is_landed(qb + qq + 1)) {
            load_chunk(o0, nquad, 0, A, D, ch + 4, lane);
            compute_chunk(acc, o2);
            load_chunk(o1, nquad, 1, A, D, ch + 5, lane);
            compute_chunk(acc, o3);
            release_slot(cx.L.done + slot, (uint32_t) (qb + qq + 1));
        } else {
            compute_chunk(acc, o2);
            compute_chunk(acc, o3);
            release_slot(cx.L.done + slot, (uint32_t) (qb + qq + 1));
            if (more) {
                cx.wait_landed(qb + qq + 1);
                load_chunk(o0, nquad, 0, A, D, ch + 4, lane);
                load_chunk(o1, nquad, 1, A, D, ch + 5, lane);
            }
        }
        slot = nslot; quad = nquad;
    }
    return acc;
}

// gather `n` granules {fp32 / u32 payload, tag} spread over the consumer lanes (index ctid + 256 i), `NB` at a time; `sink(index, payload)`
template <int NB, typename Sink>
__device__ __forceinline__ void gather_granules(const uint64_t *gt, int n, uint32_t tag, int ctid, uint32_t *fault, int limit, Sink sink) {
    for (int base = 0; base < n; base += NB * 64 * ENG_CW) {
        uint64_t gv[NB];
        int spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < NB; i++) {
                const int idx = base + ctid + i * 64 * ENG_CW;
                gv[i] = ld_granule(gt + min(idx, n - 1));
                ok = ok && (uint32_t) (gv[i] >> 32) == tag;
            }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(2);
            if (poll_give_up(spins, limit, fault)) break;
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int idx = base + ctid + i * 64 * ENG_CW;
            if (idx < n) sink(idx, (uint32_t) gv[i]);
        }
    }
}

template <int PG>
__global__ void __launch_bounds__(ENG_NT, 1)
k_ffn_engine(const FfnEngArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = blockIdx.x, G = a.G, S = a.S;
    // ---- LDS carve (every offset a multiple of 16)
    EngLds L;
    {
        uint8_t *p = lds;
        L.ring = p; p += (size_t) S * QUAD_BYTES;
        L.qa1A = (uint32_t *) p; p += (size_t) a.nqd * 4 * 256;
        L.qa1D = (float *) p; p += (size_t) a.nqd * 4 * 32;
        L.un = p; p += (size_t) max(a.d * 4, a.nqF * 4 * 288);
        L.hown = (float *) p; p += (size_t) a.maxrg * 32;
        L.actb = (float *) p; p += (size_t) a.maxu * 32;
        L.amaxb = (float *) p; p += (size_t) ((a.maxu * 4 + 15) & ~15);
        L.gub = (float *) p; p += (size_t) a.maxu * 64;
        L.ucnt = (uint32_t *) p; p += (size_t) ((a.maxu * 4 + 15) & ~15);
        L.red = (double *) p; p += 2 * ENG_CW * 8;
        L.landed = (uint32_t *) p; L.cbar = L.landed + 1; L.done = L.landed + 4;
    }
    // ---- this workgroup's work
    const int nrg = c < a.R ? (a.R - c + G - 1) / G : 0;                   // wo / w2 row-groups c, c + G, ...
    const int32_t *ut = a.utab + (size_t) c * (a.maxu + 1);
    const int nu = ut[0];                                                  // w1|w3 units ut[1 .. nu], in processing order
    const int Q0 = nrg * a.nqd, Q1 = Q0 + nu * 2 * a.nqd, Q2 = Q1 + nrg * a.nqF;      // quad ranges: wo | w1|w3 | w2
    const int limit = (a.lut_math & 0x1000) ? (1 << 8) : (1 << 20);
    if (tid < 4 + S) L.landed[tid] = 0u;
    if (tid < a.maxu) L.ucnt[tid] = 0u;
    __syncthreads();                        // the only workgroup barrier: from here on the loader and the consumers run apart
#if LH_PHASE_PROBE == 3
    unsigned long long pt[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    const unsigned long long probe_wall = wall_clock64();
    ENG_STAMP(pt, 0);
#endif

    if (wave == 0) {
        // =================================================== loader ===================================================
        const uint32_t ring_lds = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint8_t *) L.ring;
        const uint32_t voff = (uint32_t) lane * 16u;
        const uint64_t stride = (uint64_t) G * QUAD_BYTES;
        uint64_t g = (uint64_t) (uintptr_t) (a.eng + (size_t) c * QUAD_BYTES);
        uint32_t pub = 0;
        int slot = 0;
#if LH_PHASE_PROBE == 3
        unsigned long long blocked = 0;
#endif
        for (int q = 0; q < Q2; q++) {
            if (q >= S && ld_acq(L.done + slot) != (uint32_t) (q - S + 1)) {
                // the ring is full.  Nothing more can be issued, so waiting for everything in flight costs nothing -- and the consumers
                // may be waiting for exactly those quads before they can release a slot.
                wait_vmcnt<0>();
                if ((uint32_t) q > pub) { pub = (uint32_t) q; st_rel(L.landed, pub); }
#if LH_PHASE_PROBE == 3
                const unsigned long long b0 = __builtin_readcyclecounter();
#endif
                int spins = 0;
                while (ld_acq(L.done + slot) != (uint32_t) (q - S + 1)) { __builtin_amdgcn_s_sleep(1); if (poll_give_up(spins, limit << 2, a.fault)) break; }
#if LH_PHASE_PROBE == 3
                blocked += __builtin_readcyclecounter() - b0;
#endif
            }
            const uint32_t dst = ring_lds + (uint32_t) slot * QUAD_BYTES;
#pragma unroll
            for (int i = 0; i < 5; i++) dma_1k(dst + i * 1024, g + i * 1024, voff);
            g += stride;
            slot = slot + 1 == S ? 0 : slot + 1;
            if (q + 1 > ENG_F) {
                wait_vmcnt<5 * ENG_F>();
                if ((uint32_t) (q + 1 - ENG_F) > pub) { pub = (uint32_t) (q + 1 - ENG_F); st_rel(L.landed, pub); }
            }
#if LH_PHASE_PROBE == 3
            if (q + 1 == Q0) ENG_STAMP(pt, 1);
            if (q + 1 == Q1) ENG_STAMP(pt, 2);
#endif
        }
        drain_tail<ENG_F - 1>(L.landed, pub, Q2);
#if LH_PHASE_PROBE == 3
        ENG_STAMP(pt, 3);
        if (g_phase_probe_eng && lane == 0) {
            unsigned long long *pb = g_phase_probe_eng;
            const unsigned long long s_ = atomicAdd(pb, 1ull);
            if (s_ < pb[1]) { unsigned long long *e = pb + 8 * (1 + s_); e[0] = pt[0]; e[1] = pt[1]; e[2] = pt[2]; e[3] = pt[3]; e[4] = blocked;
                              e[5] = (0xE2ull << 48) | ((unsigned long long) Q2 << 32) | (unsigned) c; e[6] = wall_clock64(); e[7] = probe_wall; }
        }
#endif
        return;
    }

    // ===================================================== consumers =====================================================
    const int cw = wave - 1, ctid = tid - 64, k = lane & 7, r = lane >> 3;
    EngCtx cx = { L, S, 0u, a.fault, limit };
    uint32_t round = 0;
    const uint32_t epoch_ = __builtin_nontemporal_load(a.epoch);
    const uint32_t tag = make_tag(epoch_, a.layer + 1);
    const int nh = a.d >> 4;                                 // half-blocks of a K = d row
    // the norm weights of the half-blocks this lane will own after edge 1 (L2-resident; requested now, used much later)
    f32x4 nw[PG][4];
#pragma unroll
    for (int u = 0; u < PG; u++) {
        const int hi = min(ctid + u * 64 * ENG_CW, nh - 1);
#pragma unroll
        for (int v = 0; v < 4; v++) nw[u][v] = ((const f32x4 *) a.norm_w)[hi * 4 + v];
    }
    if (nrg > 0) {
        // ---- phase A: wo.  QA operand (the attention launch's output) global -> LDS; pad chunks beyond the row zeroed
        const int nA = a.ncd * 16, nD = a.ncd * 2;          // 16-byte granules of A / d
        for (int i = ctid; i < nA; i += 64 * ENG_CW) ((u32x4 *) L.qa1A)[i] = ((const u32x4 *) a.qa_A)[i];
        for (int i = ctid; i < nD; i += 64 * ENG_CW) ((u32x4 *) L.qa1D)[i] = ((const u32x4 *) a.qa_d)[i];
        for (int i = a.ncd * 64 + ctid; i < a.nqd * 4 * 64; i += 64 * ENG_CW) L.qa1A[i] = 0u;
        for (int i = a.ncd * 8 + ctid; i < a.nqd * 4 * 8; i += 64 * ENG_CW) L.qa1D[i] = 0.0f;
        cons_barrier(L.cbar, round, lane, a.fault);
        for (int j = cw; j < nrg; j += ENG_CW) {
            const int m = (c + G * j) * 8 + r;
            const float xr = a.x_in[m];                          // residual operand (requested before the chain, used after it)
            float acc = run_rowgroup(cx, j * a.nqd, a.nqd, L.qa1A, L.qa1D, lane);
            acc = fold8(acc);
            if (k == 0) {
                const float h = acc + xr;                     // .mm:654  inpFF = cur + inpSA
                L.hown[j * 8 + r] = h;
                store_tagged_agent(a.h_t + m, __builtin_bit_cast(uint32_t, h), tag ^ ((a.lut_math & 0x1000) ? 1u : 0u));
            }
        }
    }
    ENG_STAMP(pt, 1);
    if (nu > 0) {
        // ---- edge 1: the whole row h, from every workgroup
        float *row = (float *) L.un;
        gather_granules<16>(a.h_t, a.d, tag, ctid, a.fault, limit, [&](int idx, uint32_t bits) { row[idx] = __builtin_bit_cast(float, bits); });
        cons_barrier(L.cbar, round, lane, a.fault);
        ENG_STAMP(pt, 2);
        // ---- norm * w -> Q4_0 (ggml.c:5327-5385, :4555, :456-523), register-resident half-blocks as gemv_body's prologue
        f32x4 xa[PG][4];
        double s1 = 0.0;
#pragma unroll
        for (int u = 0; u < PG; u++) {
            const int hi = ctid + u * 64 * ENG_CW;
            if (hi < nh) {
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    xa[u][v] = ((const f32x4 *) row)[hi * 4 + v];
                    s1 += (double) xa[u][v].x; s1 += (double) xa[u][v].y; s1 += (double) xa[u][v].z; s1 += (double) xa[u][v].w;
                }
            } else {
#pragma unroll
                for (int v = 0; v < 4; v++) xa[u][v] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
            }
        }
        s1 = wave_sum_d(s1);
        if (lane == 0) L.red[cw] = s1;
        cons_barrier(L.cbar, round, lane, a.fault);
        const double mean = (((L.red[0] + L.red[1]) + L.red[2]) + L.red[3]) / (double) a.d;
        double s2 = 0.0;
#pragma unroll
        for (int u = 0; u < PG; u++)
            if (ctid + u * 64 * ENG_CW < nh) {
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const double v0 = (double) xa[u][v].x - mean, v1 = (double) xa[u][v].y - mean;
                    const double v2 = (double) xa[u][v].z - mean, v3 = (double) xa[u][v].w - mean;
                    xa[u][v].x = (float) v0; xa[u][v].y = (float) v1; xa[u][v].z = (float) v2; xa[u][v].w = (float) v3;
                    s2 += v0 * v0; s2 += v1 * v1; s2 += v2 * v2; s2 += v3 * v3;
                }
            }
        s2 = wave_sum_d(s2);
        if (lane == 0) L.red[ENG_CW + cw] = s2;
        cons_barrier(L.cbar, round, lane, a.fault);
        const double sum2 = ((L.red[ENG_CW] + L.red[ENG_CW + 1]) + L.red[ENG_CW + 2]) + L.red[ENG_CW + 3];
        const float scale = (float) (1.0 / sqrt(sum2 / (double) a.d + (double) 1e-5f));
#pragma unroll
        for (int u = 0; u < PG; u++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                xa[u][v].x = nw[u][v].x * (xa[u][v].x * scale); xa[u][v].y = nw[u][v].y * (xa[u][v].y * scale);
                xa[u][v].z = nw[u][v].z * (xa[u][v].z * scale); xa[u][v].w = nw[u][v].w * (xa[u][v].w * scale);
            }
#pragma unroll
        for (int u = 0; u < PG; u++) {
            const int hi = ctid + u * 64 * ENG_CW;          // half-block index; block = hi >> 1, half = hi & 1
            const bool live = hi < nh;
            float amax = 0.0f;
            if (live) {
#pragma unroll
                for (int v = 0; v < 4; v++)
                    amax = fmaxf(fmaxf(fmaxf(amax, fabsf(xa[u][v].x)), fabsf(xa[u][v].y)), fmaxf(fabsf(xa[u][v].z), fabsf(xa[u][v].w)));
            }
            amax = fmaxf(amax, dpp_f<DPP_QUAD_XOR1>(amax));    // partner half (lane ^ 1); both dead or both live
            const float dd = amax / 7.0f;
            const float id = (amax != 0.0f) ? 7.0f / amax : 0.0f;
            uint32_t pr[8];
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const uint32_t n0 = (uint32_t) ((int) __builtin_rintf(xa[u][v].x * id)) & 0xF, n1 = (uint32_t) ((int) __builtin_rintf(xa[u][v].y * id)) & 0xF;
                const uint32_t n2 = (uint32_t) ((int) __builtin_rintf(xa[u][v].z * id)) & 0xF, n3 = (uint32_t) ((int) __builtin_rintf(xa[u][v].w * id)) & 0xF;
                pr[2 * v] = n0 | (n1 << 8);
                pr[2 * v + 1] = n2 | (n3 << 8);
            }
            const int half = hi & 1, b = hi >> 1, cc = b >> 3, j = b & 7;
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                const uint32_t other = (uint32_t) __builtin_amdgcn_mov_dpp((int) pr[kk], DPP_QUAD_XOR1, 0xF, 0xF, true);
                const uint32_t dw = (half ? (other | (pr[kk] << 16)) : (pr[kk] | (other << 16))) << (4 * (j & 1));
                if (live && (kk >> 2) == half) L.qa1A[(cc * 8 + kk) * 8 + j] = dw;
            }
            if (live && half == 0) L.qa1D[b] = dd;
        }
        // blocks that pad the row to whole quads
        for (int b = a.d / 32 + ctid; b < a.nqd * 32; b += 64 * ENG_CW) {
            const int cc = b >> 3, j = b & 7;
#pragma unroll
            for (int kk = 0; kk < 8; kk++) L.qa1A[(cc * 8 + kk) * 8 + j] = 0u;
            L.qa1D[b] = 0.0f;
        }
        cons_barrier(L.cbar, round, lane, a.fault);
        ENG_STAMP(pt, 3);
        // ---- phase C: w1 | w3.  Task i = row-group i of this workgroup's stream (gate, up of its units in turn), one per consumer wave
        // at a time; the wave that finishes a unit's second row-group completes the unit: SiLU(gate) * up (ggml.c:1956-1963,
        // .mm:678-680) and the unit's partial amax, published at once (units of a Q4_0 block shared with another workgroup come first
        // in the stream, so the partner has the partial amax long before it quantizes)
        for (int i = cw; i < 2 * nu; i += ENG_CW) {
            const int ul = i >> 1, part = i & 1;
            float av = run_rowgroup(cx, Q0 + i * a.nqd, a.nqd, L.qa1A, L.qa1D, lane);
            av = fold8(av);
            if (k == 0) L.gub[(ul * 2 + part) * 8 + r] = av;
            uint32_t prev = 0;
            if (lane == 0) prev = __hip_atomic_fetch_add(L.ucnt + ul, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            prev = (uint32_t) __builtin_amdgcn_readfirstlane((int) prev);
            if (prev == 1u) {
                float act = 0.0f;
                if (lane < 8) {
                    const float gv = L.gub[(ul * 2) * 8 + lane], uv = L.gub[(ul * 2 + 1) * 8 + lane];
                    const uint16_t gh = f2h_bits(gv);
                    act = h2f_bits((a.lut_math & 1) ? silu_math_bits(gh) : a.T_silu[gh]) * uv;
                    L.actb[ul * 8 + lane] = act;
                }
                const float am = wave_max_f(lane < 8 ? fabsf(act) : 0.0f);
                if (lane == 0) { L.amaxb[ul] = am; store_tagged_agent(a.amax_t + ut[1 + ul], __builtin_bit_cast(uint32_t, am), tag); }
            }
        }
        cons_barrier(L.cbar, round, lane, a.fault);
        ENG_STAMP(pt, 4);
        // ---- quantize this workgroup's FFN activations (ggml.c:456-523) with the amax of their whole Q4_0 block: lane = (unit, row).
        // Partial amaxes of the block's other units: this workgroup's from LDS, the others' granules polled together.
        for (int base = 0; base < nu * 8; base += 64 * ENG_CW) {
            const int i = base + ctid;
            const bool live = i < nu * 8;
            const int ul = live ? i >> 3 : 0, rr = i & 7, u = ut[1 + ul], b = u >> 2;
            float amax = 0.0f;
            bool need[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int uu = b * 4 + t;
                int li = -1;
                for (int j = 0; j < nu; j++) if (ut[1 + j] == uu) li = j;
                need[t] = li < 0;
                if (li >= 0) amax = fmaxf(amax, L.amaxb[li]);
            }
            int spins = 0;
            for (;;) {
                bool ok = true;
                uint64_t gv[4];
#pragma unroll
                for (int t = 0; t < 4; t++) gv[t] = need[t] ? ld_granule(a.amax_t + b * 4 + t) : 0ull;
                float am2 = amax;
#pragma unroll
                for (int t = 0; t < 4; t++)
                    if (need[t]) { ok = ok && (uint32_t) (gv[t] >> 32) == tag; am2 = fmaxf(am2, __builtin_bit_cast(float, (uint32_t) gv[t])); }
                if (ok) { amax = am2; break; }
                __builtin_amdgcn_s_sleep(1);
                if (poll_give_up(spins, limit, a.fault)) break;
            }
            const float act = L.actb[ul * 8 + rr];
            const float dd = amax / 7.0f;
            const float id = (amax != 0.0f) ? 7.0f / amax : 0.0f;
            const uint32_t nib = (uint32_t) ((int) __builtin_rintf(act * id)) & 0xF;
            uint32_t pay = 0;
#pragma unroll
            for (int t = 0; t < 8; t++) pay |= (uint32_t) __shfl((int) nib, (lane & ~7) + t) << (4 * t);
            if (live && rr == 0) {
                store_tagged_agent(a.act_t + u, pay, tag);
                if ((u & 3) == 0) store_tagged_agent(a.d2_t + b, __builtin_bit_cast(uint32_t, dd), tag);
            }
        }
    }
    ENG_STAMP(pt, 5);
    double ps1 = 0.0, ps2 = 0.0;
    if (nrg > 0) {
        // ---- edge 2: the FFN activation's QA operand, from every workgroup (the fp32 row in the union buffer is dead: every consumer
        // passed the barrier behind the quantizer)
        uint32_t *A2 = (uint32_t *) L.un;
        float *D2 = (float *) (L.un + (size_t) a.nqF * 4 * 256);
        const int nb = a.F / 32;
        for (int b = nb + ctid; b < a.nqF * 32; b += 64 * ENG_CW) {      // padding blocks: zero nibbles, zero scale
            const int cc = b >> 3, j = b & 7;
#pragma unroll
            for (int kk = 0; kk < 8; kk++) A2[(cc * 8 + kk) * 8 + j] = 0u;
            D2[b] = 0.0f;
        }
        gather_granules<8>(a.act_t, a.U, tag, ctid, a.fault, limit, [&](int u, uint32_t pay) {
            // unit u = rows 8 v .. 8 v + 7 of block b: elements e = 8 v + i -> chain (e % 16) / 2, byte pair (e / 16)
            const int b = u >> 2, v = u & 3, cc = b >> 3, j = b & 7, kb = 4 * (v & 1), hf = v >> 1;
#pragma unroll
            for (int mm = 0; mm < 4; mm++) {
                const uint32_t n0 = (pay >> (8 * mm)) & 0xF, n1 = (pay >> (8 * mm + 4)) & 0xF;
                const uint16_t piece = (uint16_t) ((n0 | (n1 << 8)) << (4 * (j & 1)));
                ((uint16_t *) (A2 + (cc * 8 + kb + mm) * 8 + j))[hf] = piece;
            }
        });
        gather_granules<2>(a.d2_t, nb, tag, ctid, a.fault, limit, [&](int b, uint32_t bits) { D2[b] = __builtin_bit_cast(float, bits); });
        cons_barrier(L.cbar, round, lane, a.fault);
        ENG_STAMP(pt, 6);
        // ---- phase E: w2 + residual (.mm:682-687)
        for (int j = cw; j < nrg; j += ENG_CW) {
            const int m = (c + G * j) * 8 + r;
            float acc = run_rowgroup(cx, Q1 + j * a.nqF, a.nqF, A2, D2, lane);
            acc = fold8(acc);
            const float y = acc + L.hown[j * 8 + r];
            if (k == 0) a.x_out[m] = y;
            const double yd = k == 0 ? (double) y : 0.0;
            ps1 += wave_sum_d(yd); ps2 += wave_sum_d(yd * yd);
        }
    }
    ENG_STAMP(pt, 7);
    // this workgroup's share of the next norm's statistics (PREP_NORMP of the following launch): fixed order wave 1..4
    if (lane == 0) { L.red[cw] = ps1; L.red[ENG_CW + cw] = ps2; }
    cons_barrier(L.cbar, round, lane, a.fault);
    if (ctid == 0) {
        const double t1 = ((L.red[0] + L.red[1]) + L.red[2]) + L.red[3];
        const double t2 = ((L.red[ENG_CW] + L.red[ENG_CW + 1]) + L.red[ENG_CW + 2]) + L.red[ENG_CW + 3];
        a.part_out[c] = f64x2{ t1, t2 };
    }
#if LH_PHASE_PROBE == 3
    ENG_STAMP(pt, 8);
    if (g_phase_probe_eng && ctid == 0) {
        unsigned long long *pb = g_phase_probe_eng;
        const unsigned long long wall_ = wall_clock64();
        const unsigned long long s_ = atomicAdd(pb, 2ull);
        if (s_ + 1 < pb[1]) {
            unsigned long long *e = pb + 8 * (1 + s_);
            e[0] = pt[0]; e[1] = pt[1]; e[2] = pt[2]; e[3] = pt[3]; e[4] = pt[4]; e[5] = (0xE0ull << 48) | ((unsigned long long) nu << 40) | ((unsigned long long) nrg << 32) | (unsigned) c; e[6] = wall_; e[7] = probe_wall;
            e += 8;
            e[0] = pt[0]; e[1] = pt[5]; e[2] = pt[6]; e[3] = pt[7]; e[4] = pt[8]; e[5] = (0xE1ull << 48) | (unsigned) c; e[6] = wall_; e[7] = probe_wall;
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------------------------
// decode tiles -> engine order (load time).  One workgroup per (workgroup c of the engine, quad q): copies 4 tiles' nibbles and scales.
// ------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_tiles_to_engine(const uint8_t *__restrict__ wo, const uint8_t *__restrict__ w13, const uint8_t *__restrict__ w2, uint8_t *__restrict__ eng,
                  int G, int R, const int32_t *__restrict__ utab, int maxu, int ncd, int nqd, int ncF, int nqF) {
    const int c = blockIdx.x, q = blockIdx.y, t = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nrg = c < R ? (R - c + G - 1) / G : 0;
    const int32_t *ut = utab + (size_t) c * (maxu + 1);
    const int nu = ut[0];
    const int Q0 = nrg * nqd, Q1 = Q0 + nu * 2 * nqd, Q2 = Q1 + nrg * nqF;
    if (q >= Q2) return;
    const uint8_t *tiles; int tg, nch, ch;                // source matrix, tile group (row-group in tile order), its chunk count, first chunk
    if (q < Q0) { tiles = wo; tg = c + G * (q / nqd); nch = ncd; ch = (q % nqd) * 4; }
    else if (q < Q1) {
        const int qq = q - Q0, ul = qq / (2 * nqd), part = (qq / nqd) & 1, u = ut[1 + ul];     // part 0: gate row-group u, 1: up row-group u
        tiles = w13; tg = (u >> 2) * 8 + part * 4 + (u & 3); nch = ncd; ch = (qq % nqd) * 4;      // (interleaved w1|w3 tile order, k_repack_q4)
    } else { const int qq = q - Q1; tiles = w2; tg = c + G * (qq / nqF); nch = ncF; ch = (qq % nqF) * 4; }
    uint8_t *dst = eng + ((size_t) q * G + c) * QUAD_BYTES;
    const int cht = ch + t;
    u32x4 nib = { 0u, 0u, 0u, 0u };
    float sc = 0.0f;
    if (cht < nch) {
        const uint8_t *src = tiles + ((size_t) tg * (nch + 1) + cht) * TILE_BYTES;
        nib = *(const u32x4 *) (src + lane * 16);
        sc = *(const float *) (src + 1024 + lane * 4);
    }
    *(u32x4 *) (dst + t * 1024 + lane * 16) = nib;
    *(float *) (dst + 4096 + t * 256 + lane * 4) = sc;
}

// ------------------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------------------
static int eng_ncu() {
    static int n = [] { int dev = 0; hipDeviceProp_t p; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0; return p.multiProcessorCount; }();
    return n;
}

// ---- the split of w1|w3 over the workgroups: UNITS (gate row-group u + up row-group u) in the order each workgroup processes them.
// A Q4_0 block of the FFN activation is 4 consecutive units; a workgroup that holds only part of a block exchanges partial amaxes
// with the holders of the rest, so (i) as few blocks as possible should be shared and (ii) shared units come FIRST in a workgroup's
// order: the exchange then hides behind the rest of its units.
//   scheme 1 (half blocks): workgroups get whole HALF blocks (2 units); those with an odd number of halves sit in adjacent pairs
//            that share exactly one block -- at most one shared block per workgroup, its 2 units first.  7B: 88 pairs x 3 halves
//            + 80 x 2 halves.
//   scheme 0 (units): contiguous, balanced to +-1 unit; up to two shared blocks per workgroup (head and tail), both first.
// The scheme with the smaller maximum load wins (ties: 1); LLAMAHIP_ENGINE_SPLIT forces one (tests).
static int build_units(int U, int G, int scheme, std::vector<std::vector<int>> &out) {
    out.assign(G, {});
    int maxu = 0;
    if (scheme == 1) {
        const int H2 = U / 2, base = H2 / G, extra = H2 % G;        // halves per workgroup: base or base + 1
        std::vector<int> cnt(G);
        // workgroups with an ODD number of halves first (there is an even number of them), so that every pair starts block-aligned
        const int n_hi = extra, n_lo = G - extra;
        const bool hi_odd = ((base + 1) & 1) != 0;
        for (int c = 0; c < G; c++) cnt[c] = hi_odd ? (c < n_hi ? base + 1 : base) : (c < n_lo ? base : base + 1);
        int h = 0;
        for (int c = 0; c < G; c++) {
            const int h0 = h, h1 = h + cnt[c];
            h = h1;
            std::vector<int> halves;
            // the shared half first: the LAST half when this workgroup's range ends inside a block, the FIRST when it starts inside one
            if (cnt[c] > 0 && (h1 & 1)) halves.push_back(h1 - 1);
            if (cnt[c] > 0 && (h0 & 1)) halves.push_back(h0);
            for (int x = h0; x < h1; x++) if (std::find(halves.begin(), halves.end(), x) == halves.end()) halves.push_back(x);
            for (int x : halves) { out[c].push_back(2 * x); out[c].push_back(2 * x + 1); }
            maxu = std::max(maxu, (int) out[c].size());
        }
    } else {
        for (int c = 0; c < G; c++) {
            const int u0 = (int) ((long) c * U / G), u1 = (int) ((long) (c + 1) * U / G);
            std::vector<int> &o = out[c];
            // units of the partial block at the tail, then at the head, then the whole blocks
            for (int u = u0; u < u1; u++) if ((u >> 2) == ((u1 - 1) >> 2) && (u1 & 3) != 0) o.push_back(u);
            for (int u = u0; u < u1; u++) if ((u >> 2) == (u0 >> 2) && (u0 & 3) != 0 && std::find(o.begin(), o.end(), u) == o.end()) o.push_back(u);
            for (int u = u0; u < u1; u++) if (std::find(o.begin(), o.end(), u) == o.end()) o.push_back(u);
            maxu = std::max(maxu, (int) o.size());
        }
    }
    return maxu;
}

// geometry of the engine for a layer shape; G = 0: the engine does not apply.  utab: [G][maxu + 1] {n, units...} for the device
FfnEngGeom ffn_engine_geometry(int d, int F, std::vector<int32_t> *utab) {
    FfnEngGeom g = {};
    static const int grid_env = getenv("LLAMAHIP_ENGINE_GRID") ? atoi(getenv("LLAMAHIP_ENGINE_GRID")) : 0;      // tests: other work splits
    static const int split_env = getenv("LLAMAHIP_ENGINE_SPLIT") ? atoi(getenv("LLAMAHIP_ENGINE_SPLIT")) : -1;
    const int ncu = eng_ncu();
    if (d % 32 != 0 || F % 32 != 0 || d < 32 || F < 32 || d > 8192 || ncu < 1) return g;
    g.d = d; g.F = F;
    g.ncd = (d + 255) / 256; g.nqd = (g.ncd + 3) / 4; g.ncF = (F + 255) / 256; g.nqF = (g.ncF + 3) / 4;
    g.R = d / 8; g.U = F / 8;
    int G = std::min(ncu, std::max(g.R, g.U));
    if (grid_env > 0) G = std::min(ncu, grid_env);
    g.G = G;
    std::vector<std::vector<int>> u0s, u1s;
    const int m0 = build_units(g.U, G, 0, u0s), m1 = build_units(g.U, G, 1, u1s);
    const int scheme = split_env >= 0 ? (split_env ? 1 : 0) : (m1 <= m0 ? 1 : 0);
    const std::vector<std::vector<int>> &us = scheme ? u1s : u0s;
    g.scheme = scheme;
    g.maxrg = (g.R + G - 1) / G; g.maxu = std::max(1, scheme ? m1 : m0);
    if (utab) {
        utab->assign((size_t) G * (g.maxu + 1), -1);
        for (int c = 0; c < G; c++) {
            (*utab)[(size_t) c * (g.maxu + 1)] = (int32_t) us[c].size();
            for (size_t j = 0; j < us[c].size(); j++) (*utab)[(size_t) c * (g.maxu + 1) + 1 + j] = us[c][j];
        }
    }
    g.Qmax = 0;
    for (int c = 0; c < G; c++) {
        const int nrg = c < g.R ? (g.R - c + G - 1) / G : 0;
        g.Qmax = std::max(g.Qmax, nrg * g.nqd + (int) us[c].size() * 2 * g.nqd + nrg * g.nqF);
    }
    const size_t fixed = (size_t) g.nqd * 4 * 288 + (size_t) std::max(d * 4, g.nqF * 4 * 288) + (size_t) g.maxrg * 32 + (size_t) g.maxu * 32 +
                         (size_t) ((g.maxu * 4 + 15) & ~15) + (size_t) g.maxu * 64 + (size_t) ((g.maxu * 4 + 15) & ~15) + 2 * ENG_CW * 8 + 16;
    const size_t cap = 160 * 1024;
    // ring: what is left, at most 60 slots (the release words), never more than the longest stream
    long S = ((long) cap - (long) fixed - 64 * 4) / QUAD_BYTES;
    S = std::min<long>(S, 60);
    if (S < ENG_F + 4) { g.G = 0; return g; }
    g.S = (int) std::min<long>(S, std::max(g.Qmax, ENG_F + 4));
    g.lds = (size_t) g.S * QUAD_BYTES + fixed + (size_t) g.S * 4;
    g.lds = (g.lds + 15) & ~(size_t) 15;
    return g;
}

hipError_t launch_tiles_to_engine(const FfnEngGeom &g, const int32_t *d_utab, const QMat &wo, const QMat &w13, const QMat &w2, uint8_t *eng, hipStream_t st) {
    if (g.G < 1 || !d_utab || wo.nchunks != g.ncd || w13.nchunks != g.ncd || w2.nchunks != g.ncF || !w13.gmapF8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_tiles_to_engine, dim3(g.G, g.Qmax), dim3(256), 0, st, wo.tiles, w13.tiles, w2.tiles, eng, g.G, g.R, d_utab, g.maxu, g.ncd, g.nqd, g.ncF, g.nqF);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_ffn_engine(const FfnEngGeom &g, const FfnEngIO &io, hipStream_t st) {
    static const int fault_test = (getenv("LLAMAHIP_HANDOFF_FAULT_TEST") && atoi(getenv("LLAMAHIP_HANDOFF_FAULT_TEST")) == 5) ? 0x1000 : 0;
    const FfnEngArgs a = { io.eng, g.G, g.d, g.F, g.ncd, g.nqd, g.ncF, g.nqF, g.R, g.U, g.S, g.maxrg, g.maxu, io.utab, io.qa_A, io.qa_d, io.x_in, io.x_out, io.norm_w,
                           io.T_silu, g_lut_math | fault_test, io.h_t, io.amax_t, io.act_t, io.d2_t, io.epoch, io.layer, (f64x2 *) io.part_out, io.fault };
    if (g.d <= 4096) hipLaunchKernelGGL(k_ffn_engine<1>, dim3(g.G), dim3(ENG_NT), g.lds, st, a);
    else hipLaunchKernelGGL(k_ffn_engine<2>, dim3(g.G), dim3(ENG_NT), g.lds, st, a);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t init_attrs_engine() {
    const int cap = 160 * 1024;
    hipError_t e = hipFuncSetAttribute((const void *) k_ffn_engine<1>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void *) k_ffn_engine<2>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
}

hipError_t set_phase_probe_engine(unsigned long long *dev_buf) {
#if LH_PHASE_PROBE == 3
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase_probe_eng), &dev_buf, sizeof(dev_buf));
#else
    (void) dev_buf;
    return hipSuccess;
#endif
}

}  // namespace lh
