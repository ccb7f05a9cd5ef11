// prompt_gemm.hip -- the Q4_0 x Q4_0 mat-mul for MULTI-ROW evals (prompt chunks), bit-exact with ggml_compute_forward_mul_mat_q4_0_f32
// (ggml.c:5987-6285, vec_dot :1415-1466): k_gemm_lds (decode tiles, QA in LDS), k_gemm_rows (row-lane tiles), k_gemm_mfma / k_gemm_mfma4 (integer sums on the matrix cores), the tile converters,
// and launch_gemm with its kernel selection rules.  Conventions and layouts: decode.hip / DESIGN.md.
#include "kcommon.hip.h"

namespace lh {

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

// (k_gemm_skinny, the 2 .. 60-row kernel of rounds 1-4 -- a wave = one row-group x <= 4 columns through a register ring, the epilogues with
//  RoPE + KV append / SiLU * up -> Q4_0 -- was replaced by k_gemv_set (gemv_set.hip) in round 5: faster at every row count from 2 to 60,
//  profiles/r05_w_fresh_ab_final_plan.txt, r05_y_rows_max.txt; removed.)

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

typedef _Float16 h4v __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// k_gemm_mfma4 -- the long-prompt kernel: the same exact product on the fp16 K = 4 matrix instructions.
// K = 4 is exactly one chain of a Q4_0 block (the 4 elements {2k, 2k+1, 16+2k, 17+2k} that one lane of the reference's
// _mm256_madd_epi16 sums), and v_mfma_f32_16x16x4_4b_f16 carries FOUR independent 16 x 16 x 4 products ("blocks" = groups of 16
// lanes on the operand side, result registers 4 b .. 4 b + 3; checked on the hardware, tools/mfma_layout_probe4.hip), so one issue
// returns four chains' sums for a 16 x 16 sub-tile:
//   * nothing is masked (the int8 kernel above issues one 32 x 32 x 32 MFMA per chain with 7/8 of its operand zeroed);
//   * the sums arrive as FLOATS (small integers are exact in fp16 operands and fp32 accumulation), so the integer -> float
//     conversion of the int8 kernel disappears: what is left per output and chain is the one FMA the reference defines.
// Rounds 2-3 ran this on v_mfma_f32_32x32x4_2b_f16 with 128 accumulators per lane (k_gemm_mfma16: a 32 x 32 tile x 8 chains per
// wave), i.e. two waves per SIMD -- and tools/valu_rate_probe.hip / tools/chain_probe.hip show what that costs: a wave alone on its
// SIMD issues one VALU instruction per ~12 cycles, so whenever one of the two waits (for its MFMA result, for the barrier) the other
// runs at half rate; it sat at 94 ns per MFMA issue and SIMD against 60 ns of VALU work (2 048 tokens of the 7B: 192 ms; removed in
// round 4, profiles/r04_p_gemm4_ab.txt .. r04_s_*).  Here a wave owns a 16-row x 32-column output tile with all 8 chains (a lane holds
// 4 rows x 1 column of each of its two sub-tiles, so the 8 scale products d_w * d_a of a block serve all 8 chains): 64 accumulators +
// 16 results, <= 128 registers, FOUR waves per SIMD, 8 waves per workgroup (64 x 64 outputs), two workgroups per CU.  The operands of
// the next quad go global -> LDS by DMA (global_load_lds_dwordx4: no staging registers, no ds_write), the activation columns
// XOR-swizzled by the loader so that the 256-byte column stride reads conflict-free.
// The FMA chains are volatile asm so that they stay in this order with one result set live (written as plain C++ the compiler sinks
// them below later MFMAs and spills).  The MFMA -> VALU read needs software wait states which the compiler only inserts for
// instructions it can see: the first FMA of a result set is a visible one, and its result is a (dummy) input of the first asm FMA,
// which orders every asm FMA after it (without it: wrong logits, 5 % faster).
// What bounds it (tools/mfma_overlap_probe.hip, profiles/r04_p_mfma_overlap.txt): on gfx950 an MFMA and the VALU instructions of
// OTHER waves of the same SIMD do not run side by side -- an 8-pass MFMA keeps the VALU out for its 14 ns, whatever the instruction
// type (f16, i8, f32; only the 16-pass 32x32x4_2b lets the VALU in for its second half) -- so a block of a sub-tile costs the MFMA's
// 14 ns PLUS its 16 FMAs (19 ns) and their operand preparation: 39 ns measured per block, sub-tile and SIMD.  (That 16-pass form at three
// waves per SIMD -- a wave = 32 x 32 outputs x one side of the tree, 4-wave workgroups -- was built and measured: bit-identical, 190.5
// against 167.1 ms for 2 048 tokens, profiles/r04_z_gemm2b_ab.txt; a wave waits 28 ns for each of its results and three do not cover that.)
// Weight copy "mt4" (ONE BYTE per weight, 1.8 x the size of the int8 tiles): tile (row-block of 32, quad of 4 blocks) = MT4_BYTES = 4608 B:
//   [s 0..1][lane 0..63][32 B]   lane = i + 16 b, row 16 s + i: dword 2 j + h = block 4 q + j, chain CH(h, b) = h + {0, 4, 2, 6}[b]: its four
//                                weights as the HIGH BYTES of their fp16 values (every integer -8..7 has an fp16 low byte of zero), so
//                                that two v_perm_b32 (bytes {0, w0, 0, w1} and {0, w2, 0, w3}) ARE the MFMA operand -- no bias, no
//                                masks, no shifts (the nibble form cost 11 VALU instructions per block and lane, this one 4)
//   [j][32 rows] fp32 scales
// Activation operand "QB4" (k_qa_to_qb4): per column and quad 256 B = [b][j 0..3][h][4 fp16], exact integers -8..7.
// The chains of an operand, {h, 4 + h, 2 + h, 6 + h}, are one side of the reference's final add tree (ggml.c:872-887).
// ------------------------------------------------------------------------------------------------
typedef float f32x16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int mt4_chain(int h, int b) { return h + 2 * (((b & 1) << 1) | (b >> 1)); }

constexpr int MT4_BYTES = MT4_TILE_BYTES;
__global__ void k_tiles_to_mt4(const uint8_t *__restrict__ tiles, uint8_t *__restrict__ mt,
                               int ngroups, int nchunks, int nrb32, int gmapF8) {
    const int nq = nchunks * 2;
    const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long) nrb32 * nq * 2 * 64;
    if (gid >= total) return;
    const int lane = (int) (gid & 63), s = (int) ((gid >> 6) & 1);
    const long t = gid >> 7;
    const int q = (int) (t % nq), rb = (int) (t / nq);
    const int i = lane & 15, b = lane >> 4;
    auto tile_of = [&](int row, int c) -> const uint8_t * {
        const int lg = row >> 3;
        if (lg >= ngroups) return nullptr;
        int tg = lg;
        if (gmapF8) tg = lg < gmapF8 ? (lg >> 2) * 8 + (lg & 3) : ((lg - gmapF8) >> 2) * 8 + 4 + ((lg - gmapF8) & 3);
        return tiles + ((size_t) tg * (nchunks + 1) + c) * TILE_BYTES;
    };
    const int row = rb * 32 + 16 * s + i, r = row & 7;
    uint32_t x[8];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int bk = q * 4 + j, c = bk >> 3, jj = bk & 7, i2 = jj >> 1, half = jj & 1;
        const uint8_t *tp = tile_of(row, c);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int kc = mt4_chain(h, b);
            const uint32_t dw = tp ? ((const uint32_t *) (tp + (r * 8 + kc) * 16))[i2] : 0u;     // chain kc, blocks (2 i2, 2 i2 + 1): byte p = element e_p
            uint32_t v = 0u;
#pragma unroll
            for (int pp = 0; pp < 4; pp++) {
                const int n = (int) (((dw >> (8 * pp + 4 * half)) & 0xFu) ^ 8u) - 8;           // the weight, -8..7 (padding rows: 0)
                const uint32_t hb = (uint32_t) (__builtin_bit_cast(uint16_t, (_Float16) (float) n) >> 8);
                v |= hb << (8 * pp);
            }
            x[2 * j + h] = v;
        }
    }
    uint8_t *o = mt + ((size_t) rb * nq + q) * MT4_BYTES;
    *(u32x4 *) (o + s * 2048 + lane * 32) = u32x4{ x[0], x[1], x[2], x[3] };
    *(u32x4 *) (o + s * 2048 + lane * 32 + 16) = u32x4{ x[4], x[5], x[6], x[7] };
    if (s == 0) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int idx = lane + 64 * u, j = idx >> 5, m = idx & 31;
            const int bk = q * 4 + j, c = bk >> 3, jj = bk & 7, row2 = rb * 32 + m;
            const uint8_t *tp = tile_of(row2, c);
            ((float *) (o + 4096))[idx] = tp ? ((const float *) (tp + 1024 + (row2 & 7) * 32))[(jj & 3) * 2 + (jj >> 2)] : 0.0f;
        }
    }
}

// QA (chain-major signed nibbles) -> QB4.  One thread per (column, block, chain).
__global__ void k_qa_to_qb4(const uint32_t *__restrict__ qa_A, uint8_t *__restrict__ qb, int nchunks, int N) {
    const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
    const int nbp = nchunks * 8;
    const long total = (long) N * nbp * 8;
    if (gid >= total) return;
    const int kc = (int) (gid & 7);
    const long t = gid >> 3;
    const int bk = (int) (t % nbp), n = (int) (t / nbp);
    const int c = bk >> 3, jj = bk & 7;
    const uint32_t dw = qa_A[(size_t) n * nchunks * 64 + c * 64 + kc * 8 + jj];
    h4v v;
#pragma unroll
    for (int pp = 0; pp < 4; pp++) {
        const int nib = (int) ((dw >> (8 * pp + 4 * (jj & 1))) & 0xF);
        v[pp] = (_Float16) (float) ((nib ^ 8) - 8);
    }
    const int h = kc & 1, k2 = kc >> 1, b = ((k2 & 1) << 1) | (k2 >> 1);          // mt4_chain(h, b) == kc
    const int q = bk >> 2, j = bk & 3;
    *(h4v *) (qb + ((size_t) n * nbp + q * 4) * 64 + (b * 4 + j) * 16 + h * 8) = v;
}

// one wave instruction: lane l's 16 bytes at `base + voff` land in LDS at `lds_dst + 16 l` (lds_dst, base: wave-uniform)
__device__ __forceinline__ void gemm4_dma16(uint32_t lds_dst, uint64_t base, uint32_t voff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(base) : "memory");
}

template <int EPI>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_gemm_mfma4(const uint8_t *__restrict__ mt, int nrb32, int nq, int M,
             const uint8_t *__restrict__ qb, const float *__restrict__ qa_d, int ncols, int nct,
             float *__restrict__ y, long y_stride, const float *__restrict__ resid, long resid_stride) {
    __shared__ __attribute__((aligned(1024))) uint8_t sB[2][64 * 256];          // [column][16 granules, XOR-swizzled by column & 15]
    __shared__ __attribute__((aligned(1024))) uint8_t sW[2][2 * MT4_BYTES];
    __shared__ __attribute__((aligned(1024))) float sDa[2][64 * 4];
    const int bid = blockIdx.x, xcd = bid & 7, qq = bid >> 3;
    const int ct = qq % nct, rp = (qq / nct) * 8 + xcd;
    if (rp * 2 >= nrb32) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = wave & 1, wr = (wave >> 1) & 1, wc = wave >> 2;             // rows 32 wr + 16 s .. + 15, columns 32 wc .. + 31 of the 64 x 64 tile
    const int n0 = ct * 64;
    const int nbp = nq * 4;
    const int cc = lane & 15, b = lane >> 4;

    // ---- loader role of this wave: activation instructions 2 wave, 2 wave + 1 (4 columns each); weights: waves 0..4; scales of the columns: wave 5
    const uint32_t ldsB = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint8_t *) &sB[0][0];
    const uint32_t ldsW = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint8_t *) &sW[0][0];
    const uint32_t ldsD = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) float *) &sDa[0][0];
    // (weights: two tiles = 9 instructions of 1 KiB: instruction w by wave w, the ninth by wave 6; the columns' scales: wave 5)
    uint32_t offB[2], offW, offX = 0u;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int col = 4 * (2 * wave + u) + (lane >> 4), p = (lane & 15) ^ (col & 15);
        offB[u] = (uint32_t) ((size_t) min(n0 + col, ncols - 1) * nbp * 64 + p * 16);
    }
    auto w_off = [&](int g) { return (uint32_t) ((size_t) min(rp * 2 + g / 288, nrb32 - 1) * nq * MT4_BYTES + (g % 288) * 16); };
    offW = w_off(wave * 64 + lane);
    if (wave == 6) offX = w_off(8 * 64 + lane);
    if (wave == 5) offX = (uint32_t) ((size_t) min(n0 + lane, ncols - 1) * nbp * 4);
    auto issue = [&](int q, int buf) {
#pragma unroll
        for (int u = 0; u < 2; u++) gemm4_dma16(ldsB + buf * 16384 + (2 * wave + u) * 1024, (uint64_t) (uintptr_t) qb, offB[u] + (uint32_t) q * 256u);
        gemm4_dma16(ldsW + buf * (2 * MT4_BYTES) + wave * 1024, (uint64_t) (uintptr_t) mt, offW + (uint32_t) q * MT4_BYTES);
        if (wave == 6) gemm4_dma16(ldsW + buf * (2 * MT4_BYTES) + 8 * 1024, (uint64_t) (uintptr_t) mt, offX + (uint32_t) q * MT4_BYTES);
        if (wave == 5) gemm4_dma16(ldsD + buf * 1024, (uint64_t) (uintptr_t) qa_d, offX + (uint32_t) q * 16u);
    };

    float acc[2][2][4][4];                                  // [column sub-tile t][operand h][chain slot b][row r]: chain CH(h, b)
#pragma unroll
    for (int i = 0; i < 64; i++) (&acc[0][0][0][0])[i] = 0.0f;
    f32x16v zero16;
#pragma unroll
    for (int r = 0; r < 16; r++) zero16[r] = 0.0f;

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int q = 0; q < nq; q++) {
        const int buf = q & 1;
        if (q + 1 < nq) issue(q + 1, buf ^ 1);              // that buffer was last read in quad q - 1: every wave is past its closing barrier
        const uint8_t *wt_ = &sW[buf][wr * MT4_BYTES];
        const u32x4 x01 = *(const u32x4 *) (wt_ + s * 2048 + lane * 32), x23 = *(const u32x4 *) (wt_ + s * 2048 + lane * 32 + 16);
        const uint8_t *bc0 = &sB[buf][(wc * 32 + cc) * 256], *bc1 = bc0 + 16 * 256;
        const float *dc0 = &sDa[buf][(wc * 32 + cc) * 4], *dc1 = dc0 + 16 * 4;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int g = ((b * 4 + j) ^ cc) * 16;
            const u32x4 Bt0 = *(const u32x4 *) (bc0 + g), Bt1 = *(const u32x4 *) (bc1 + g);        // (h = 0 | h = 1) operands of column sub-tiles 0, 1
            const u32x4 xq = j < 2 ? x01 : x23;
            uint32_t pa[4];                                 // operand h = (pa[2 h], pa[2 h + 1]): the chain's four fp16 weights
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t xd = (j & 1) ? (h ? xq.w : xq.z) : (h ? xq.y : xq.x);
                pa[2 * h] = __builtin_amdgcn_perm(0u, xd, 0x010C000Cu);
                pa[2 * h + 1] = __builtin_amdgcn_perm(0u, xd, 0x030C020Cu);
            }
            const float da0 = dc0[j], da1 = dc1[j];
            const f32x4 dw = *(const f32x4 *) (wt_ + 4096 + (j * 32 + 16 * s + 4 * b) * 4);
            const float sc[2][4] = { { dw.x * da0, dw.y * da0, dw.z * da0, dw.w * da0 }, { dw.x * da1, dw.y * da1, dw.z * da1, dw.w * da1 } };
#pragma unroll
            for (int h = 0; h < 2; h++) {
                struct { uint32_t a, b; } aw = { pa[2 * h], pa[2 * h + 1] };
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const u32x4 Bq = t ? Bt1 : Bt0;
                    struct { uint32_t a, b; } bw = { h ? Bq.z : Bq.x, h ? Bq.w : Bq.y };
                    const f32x16v D = __builtin_amdgcn_mfma_f32_16x16x4f16(__builtin_bit_cast(h4v, aw), __builtin_bit_cast(h4v, bw), zero16, 0, 0, 0);
                    // (volatile asm FMAs in this order with one result set live; the first one is a plain FMA so that the compiler
                    // inserts the MFMA -> VALU wait states, and its result is a dummy input of the second: see the header.
                    // Packed FMAs -- v_pk_fma_f32 + v_pk_mul_f32 with op_sel broadcasts, no register copies -- measured SLOWER
                    // here, 184.8 -> 192.0 ms for 2048 tokens: profiles/r04_q_gemm4_pk_ab.txt)
                    float (&a)[4][4] = acc[t][h];
                    const float first = __builtin_fmaf(sc[t][0], D[0], a[0][0]);
                    a[0][0] = first;
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[0][1]) : "v"(sc[t][1]), "v"(D[1]), "v"(first));
#pragma unroll
                    for (int e = 2; e < 16; e++)
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[e >> 2][e & 3]) : "v"(sc[t][e & 3]), "v"(D[e]));
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // my DMA instructions for quad q + 1 have landed
        __syncthreads();
    }
    // ---- fold the 8 chains (ggml.c:872-887 tree: ((A0 + A4) + (A2 + A6)) + ((A1 + A5) + (A3 + A7))) and store: lane = column, 4 rows
    if (rp * 2 + wr >= nrb32) return;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int n = n0 + wc * 32 + 16 * t + cc;
        const int m0 = (rp * 2 + wr) * 32 + 16 * s + 4 * b;
        if (n >= ncols) continue;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float v = ((acc[t][0][0][r] + acc[t][0][1][r]) + (acc[t][0][2][r] + acc[t][0][3][r]))
                    + ((acc[t][1][0][r] + acc[t][1][1][r]) + (acc[t][1][2][r] + acc[t][1][3][r]));
            if (m0 + r < M) {
                if (EPI == EPI_RESID) v = v + resid[(size_t) n * resid_stride + m0 + r];
                y[(size_t) n * y_stride + m0 + r] = v;
            }
        }
    }
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

// Short evals (2 .. 60 rows), wq|wk|wv: mat-mul + RoPE + KV append in one launch (k_gemv_set<EPI_ROPE_KV>)
bool gemm_rope_kv_applies(const QMat &wqkv, int N, int d) {
    return wqkv.M == 3 * d && d % 8 == 0 && gemv_set_applies(wqkv, N, EPI_ROPE_KV);
}
hipError_t launch_gemm_rope_kv(const QMat &wqkv, const uint32_t *qa_A, const float *qa_d, int N, const RopeKvArgs &ra, hipStream_t st) {
    if (!gemv_set_applies(wqkv, N, EPI_ROPE_KV)) return hipErrorInvalidValue;
    g_gemm_path_counts[GEMM_PATH_SET]++;
    return launch_gemv_set_rope_kv(wqkv, qa_A, qa_d, N, ra, st);
}

// Short evals, the interleaved w1|w3: mat-mul + SiLU * up + Q4_0 quantization of the w2 operand in one launch (k_gemv_set<EPI_SILU_QAH> in
// half-block workgroups where the exchange buffers exist and one column group covers the rows, else <EPI_SILU_QA> in whole-block
// workgroups).  false = not applicable (row count, layout): use the separate steps.
bool gemm_silu_qa_applies(const QMat &w13, int N) {
    return gemv_set_applies(w13, N, EPI_SILU_QA);
}
hipError_t launch_gemm_silu_qa(const QMat &w13, const uint32_t *qa_A, const float *qa_d, int N, const uint16_t *T_silu,
                               uint32_t *out_A, float *out_d, long out_strideA, long out_strideD, hipStream_t st, const SiluHalfIO *hx) {
    const bool have_hx = hx && hx->amax_t && hx->epoch;
    if (!((have_hx && gemv_set_applies(w13, N, EPI_SILU_QAH)) || gemv_set_silu_whole_blocks(w13, N, have_hx))) return hipErrorInvalidValue;
    g_gemm_path_counts[GEMM_PATH_SET]++;
    return launch_gemv_set_silu(w13, qa_A, qa_d, N, T_silu, out_A, out_d, out_strideA, out_strideD, have_hx ? *hx : SiluHalfIO(), st);
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

hipError_t launch_tiles_to_mt4(const QMat &w, hipStream_t st) {
    static_assert(MT4_BYTES == MT4_TILE_BYTES, "QMat::mt4_bytes() sizes the copy");
    const long tot4 = (long) w.nrb32 * w.nchunks * 2 * 2 * 64;
    hipLaunchKernelGGL(k_tiles_to_mt4, dim3((unsigned) ((tot4 + 255) / 256)), dim3(256), 0, st, w.tiles, w.mt4, w.ngroups, w.nchunks, w.nrb32, w.gmapF8);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}
static hipError_t launch_gemm_mfma4(const QMat &w, int epi, const uint32_t *qa_A, uint8_t *qb4, const float *qa_d, int ncols,
                                    float *y, long y_stride, const float *resid, long resid_stride, hipStream_t st) {
    const long tot = (long) ncols * w.nchunks * 8 * 8;
    const int nct = (ncols + 63) / 64, nq = w.nchunks * 2;
    const int nrp = (w.nrb32 + 1) / 2;
    const int grid = ((nrp + 7) / 8) * nct * 8;
    hipLaunchKernelGGL(k_qa_to_qb4, dim3((unsigned) ((tot + 255) / 256)), dim3(256), 0, st, qa_A, qb4, w.nchunks, ncols);
    LH_LAUNCH_CHECK();
    if (epi == EPI_RESID) hipLaunchKernelGGL((k_gemm_mfma4<EPI_RESID>), dim3(grid), dim3(512), 0, st, w.mt4, w.nrb32, nq, w.M, qb4, qa_d, ncols, nct, y, y_stride, resid, resid_stride);
    else                  hipLaunchKernelGGL((k_gemm_mfma4<EPI_STORE>), dim3(grid), dim3(512), 0, st, w.mt4, w.nrb32, nq, w.M, qb4, qa_d, ncols, nct, y, y_stride, resid, resid_stride);
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
    // Matrix-core path when its 64 x 64-output workgroups fill the chip twice over, or from 128 rows when the last 64-column tile is
    // not mostly padding (measured against the row-per-lane kernel, 7B, every matrix on one kernel, profiles/r04_y_midN_ab.txt: 64 rows
    // -3 %, 96 -7 %, 128 +19 %, 192 +10 %, 240 +32 %; 1.5x at 1 024).
    static const int mfma_min = getenv("LLAMAHIP_MFMA_MIN") ? atoi(getenv("LLAMAHIP_MFMA_MIN")) : 0;     // tests: the small models through this kernel
    const long mfma_wgs = (long) ((w.nrb32 + 1) / 2) * ((N + 63) / 64);
    const bool mfma_rows = N >= 128 && N * 10 >= (N + 63) / 64 * 64 * 7;
    // (k_gemm_mfma4 addresses its operands as base + 32-bit byte offset)
    const bool mfma_fits = w.mt4_bytes() < ((size_t) 1 << 32) && (size_t) N * w.nchunks * 8 * 64 < ((size_t) 1 << 32);
    if (w.mt4 && !fast && qb_ws && mfma_fits && (mfma_min ? N >= mfma_min : (mfma_rows || (N >= 64 && mfma_wgs >= 512)))) {
        // matrix-core path, exact: fp16 operands (QB4: 2 bytes per element of these N activation rows), four chains per MFMA
        g_gemm_path_counts[GEMM_PATH_MFMA]++;
        return launch_gemm_mfma4(w, epi, qa_A, qb_ws, qa_d, N, y, y_stride, resid, resid_stride, st);
    }
    if (w.mt && qb_ws && (mfma_min ? N >= mfma_min : (N >= 64 && mfma_wgs >= 512))) {
        // matrix-core path on the int8 tiles (the opt-in fast path; LLAMAHIP_MFMA_I8: the round-1 exact kernel): needs the int8 operand (QB)
        g_gemm_path_counts[GEMM_PATH_MFMA]++;
        hipError_t e = launch_qa_to_qb(qa_A, qb_ws, w.nchunks, N, st);
        if (e != hipSuccess) return e;
        return launch_gemm_mfma(w, epi, qb_ws, qa_d, N, y, y_stride, resid, resid_stride, st, fast);
    }
    const long strideA = (long) w.nchunks * 64, strideD = (long) w.nchunks * 8;
    // 2 .. 60 rows (a batched decode step, the reference's 9-token evals, short prompt chunks): k_gemv_set
    if ((epi == EPI_STORE || epi == EPI_RESID) && gemv_set_applies(w, N, epi)) {
        g_gemm_path_counts[GEMM_PATH_SET]++;
        return launch_gemv_set(w, epi, qa_A, qa_d, N, y, y_stride, resid, resid_stride, st);
    }
    if (w.rows && N >= 2) {
        // widest column group that still gives the chip >= 2 waves per SIMD.  Wider groups (8, 16
        // columns: 191 / 249 VGPRs, 2 waves per SIMD) measured 10-16 % slower than 4 columns at 3 waves
        // per SIMD on a 512-token prompt: the kernel runs at ~85 % of its VALU issue limit and the third
        // wave is what hides the scalar-load latency of the operand.
        int nc = 1;
        for (int cand : { 4, 2 })
            if ((long) w.nrb * ((N + cand - 1) / cand) >= 2048) { nc = cand; break; }
        g_gemm_path_counts[GEMM_PATH_ROWS]++;
#define LH_ROWS_ARGS w, epi, qa_A, qa_d, N, y, y_stride, resid, resid_stride, st
        switch (nc) {
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


}  // namespace lh
