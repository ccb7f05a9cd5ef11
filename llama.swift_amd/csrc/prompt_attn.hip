// prompt_attn.hip -- RoPE + KV append and attention for MULTI-ROW evals: k_rope_kv, k_attn (one workgroup per (head, query): any head size /
// n_threads, debug dumps), k_attnq_* (lane = query: scores on the fp32 matrix cores / LDS-staged keys / scalar, soft_max, V*P, merge), and
// launch_rope_kv / launch_attn.  Conventions: decode.hip / DESIGN.md.
#include "kcommon.hip.h"

namespace lh {

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

// An eval of N rows can stand for the reference's SEQUENCE of evals of `chunk` rows each (its prompt flow, .mm:880-888: n_batch + 1 = 9
// tokens per llama_eval): every row-wise operator gives the same bits either way, and the one place where the reference's arithmetic
// depends on the eval a row belongs to is the V*P key split -- dc = ceil(keys / n_threads) with keys = n_past + N OF THAT EVAL
// (ggml.c:5459-5480).  Row n of a chunked pass therefore splits n_past + (its chunk's last row + 1) keys; chunk = 0: one eval.
__device__ __host__ __forceinline__ int split_keys(int n_past, int N, int n, int chunk) {
    return chunk > 0 ? n_past + min(N, (n / chunk + 1) * chunk) : n_past + N;
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
       int n_past, int N, int d, int dh, int nth, float kq_scale, const uint16_t *__restrict__ T_exp, int chunk) {
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
        const int Ts = split_keys(n_past, N, n, chunk);      // the key count the reference splits for this row (chunk_keys.h rule below)
        const int dc = (Ts + nth - 1) / nth;
        for (int th = sub; th < nth; th += nsub) {
            const int t0 = dc * th;
            int t1 = t0 + dc < Ts ? t0 + dc : Ts;
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
// Prompt attention for many query rows (head size 128) on the fp32 matrix cores.
// k_attn above gives every (head, query) its own workgroup and re-reads the head's whole K and V for each query: 2 048 rows stream
// 69 GB per layer through L2.  Here scores are materialised like the reference's KQ tensor (.mm:614), in a [head][key][query]
// workspace so that lanes read and write it coalesced; queries are processed in batches of NB rows to bound it.
//   k_attnq_scores_lds   grid (NB/64, H, KS) x 4 waves: KQ * scale for its key slice, running max -> S, pmax
//   k_attnq_softmax      grid (NB/64, H) x (64 queries x 16 key phases): exp LUT, double sum -> S = e, inv
//   k_attnq_pv_mfma      grid (NB/64, H, nth): p = e * inv; the FMA chains of ONE chunk of the reference's nth-way key split -> part
//   k_attnq_merge        the ordered add of the nth partials                              -> merged
// (Round 5 kept e as the 16 bits of its table entry in a second workspace -- half the bytes V*P reads back; bit-identical; 2 048-token eval
// 165.6 -> 166.4 ms, i.e. nothing: the workspace round trips are not what these launches wait for.  profiles/r05_r_prefill_e16_ab.txt; reverted.)
// Arithmetic is identical to k_attn.  (Rounds 1-2 had three more score kernels / one more V*P kernel with lane = query row and the key /
// value row as SGPR operands, through LDS broadcasts, or on the VALU: 201 / 152 us per score launch at 2 048 tokens against 100 here;
// removed in round 3, DESIGN_HISTORY.md.)
// ------------------------------------------------------------------------------------------------
// Scores.  ggml_vec_dot_f32 (ggml.c:1223-1258): chain l (0..31) = elements l, l + 32, l + 64, l + 96 of the two rows by FMA from 0; reduction tree
// (ggml.c:872-887) = lanes xor 8, 16, 4, 1, 2 of the AVX accumulators, i.e. ((c_l + c_{l+8}) halves added, then u0+u4 .. as written below.
// v_mfma_f32_16x16x4_f32 is bit-for-bit the k-ordered
// fmaf chain fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, C)))) per output: with A = query elements {l, l + 32, l + 64, l + 96} and
// B = the same elements of a key it IS chain l of ggml_vec_dot_f32 (ggml.c:1223-1258) for a 16 x 16 tile of (query, key) pairs.  Lane
// (m = lane % 16, kk = lane / 16) supplies element l + 32 kk of row m: the 32 operands a lane needs for the 32 chains are the 32
// CONSECUTIVE floats [32 kk, 32 kk + 32) of its query / key row -- loaded straight into registers, no LDS, no scalar loads.  One wave =
// 16 queries (registers, loaded once) against its key slice, 16 keys per step = 32 independent MFMAs (C = 0) + the reduction tree
// (ggml.c:872-887) on the VALU, 31 additions per pair.  Result registers: lane holds key n = lane % 16, queries 4 kk + r.
// Requesting the next step's key rows a step ahead (+32 registers) measured slower.  FOUR waves per SIMD (the MFMAs four at a time, <= 128
// registers, spills) measured slower too, 102 -> 126 us per launch (profiles/r04_u_attn_ab.txt): each group of four waits for its own results.
typedef float f32x4v __attribute__((ext_vector_type(4)));
// THREE waves per SIMD: the 32 MFMAs of a step go in four groups of eight (inline asm: the builtin's results go to
// AGPRs and the scheduler issues all 32 first -- 122 + 128 registers), every group folded into the tree's first levels before the next
// one's results arrive: 32 + 32 operands, 32 results, 32 partial sums.  Group order (hf 0, j 0-3), (hf 1, j 0-3), (hf 0, j 4-7),
// (hf 1, j 4-7): r1[hf][j] = D[j] + D[j + 8], u[j] = r1[0][j] + r1[1][j], then v = u[j] + u[j + 4] -- the tree (ggml.c:872-887) level by level.
// (Rounds 2-3: all 32 MFMAs first, 186 registers, two waves per SIMD: 2 048-token eval 162.5 -> 160.5 ms, profiles/r04_w_scores3_ab.txt.)
// An 8-pass MFMA result may be read by the VALU 11 wait states after its issue; only the compiler's own MFMAs get those automatically.
#define LH_SC_GROUP(D, HF, J0)                                                                                                     \
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %8, %16, 0\n\tv_mfma_f32_16x16x4_f32 %1, %9, %17, 0\n\t"                            \
                 "v_mfma_f32_16x16x4_f32 %2, %10, %18, 0\n\tv_mfma_f32_16x16x4_f32 %3, %11, %19, 0\n\t"                          \
                 "v_mfma_f32_16x16x4_f32 %4, %12, %20, 0\n\tv_mfma_f32_16x16x4_f32 %5, %13, %21, 0\n\t"                          \
                 "v_mfma_f32_16x16x4_f32 %6, %14, %22, 0\n\tv_mfma_f32_16x16x4_f32 %7, %15, %23, 0\n\ts_nop 11"                 \
                 : "=&v"(D[0]), "=&v"(D[1]), "=&v"(D[2]), "=&v"(D[3]), "=&v"(D[4]), "=&v"(D[5]), "=&v"(D[6]), "=&v"(D[7])          \
                 : "v"(aq[16 * (HF) + (J0)]), "v"(aq[16 * (HF) + (J0) + 1]), "v"(aq[16 * (HF) + (J0) + 2]), "v"(aq[16 * (HF) + (J0) + 3]),   \
                   "v"(aq[16 * (HF) + (J0) + 8]), "v"(aq[16 * (HF) + (J0) + 9]), "v"(aq[16 * (HF) + (J0) + 10]), "v"(aq[16 * (HF) + (J0) + 11]), \
                   "v"(bk[16 * (HF) + (J0)]), "v"(bk[16 * (HF) + (J0) + 1]), "v"(bk[16 * (HF) + (J0) + 2]), "v"(bk[16 * (HF) + (J0) + 3]),   \
                   "v"(bk[16 * (HF) + (J0) + 8]), "v"(bk[16 * (HF) + (J0) + 9]), "v"(bk[16 * (HF) + (J0) + 10]), "v"(bk[16 * (HF) + (J0) + 11]))
// The score kernel: the key tile SHARED by QW query tiles through LDS (round 6, VERDICT r05 item 5 / DESIGN 11.8 "the step that pays first").
// Rounds 2-5 (k_attnq_scores_mfma, removed) gave every wave its own copy of each 16-key tile straight from L2: 8 KB per 256 (query, key) pairs,
// 2.1 GB per layer at 2 048 tokens -- 95.5 us per launch; four tiles sharing: 76.3 us, eight: 83.9 (profiles/r06_d_prefill_scores_ab.txt; logits CRC
// equal).  A workgroup is QW waves = QW x 16 consecutive queries walking ONE key slice together: the tile (16 rows x 512 B)
// is fetched once per workgroup by LDS-DMA (global_load_lds_dwordx4: no staging registers -- with them the kernel spilled at three waves
// per SIMD), double buffered, one barrier per step; a wave whose queries cannot see the step's keys (causal mask: its Tb is lower than the
// workgroup's) skips the arithmetic, not the barrier.  LDS layout: rows 33 units of 16 bytes apart (32 + one unit of padding: the 16 rows of
// a quarter-wave's ds_read_b128 -- same column unit, m = 0 .. 15 -- fall on 16 distinct bank groups; unpadded rows would all hit one; a lane reads
// its 8 units through ONE address and immediates).  A DMA instruction writes lane l's 16 bytes at LDS unit 64 i + l, so the lane FETCHES
// column unit (64 i + l) % 33 of row (64 i + l) / 33 (the pad unit and the overshoot of the ninth instruction fetch something harmless).  The arithmetic per
// (query, key) pair is the macro above, unchanged: bit-identical scores, same S / pmax layout; k_attnq_softmax / _pv_mfma / _merge follow as they are.
__device__ __forceinline__ void sc_dma16(uint32_t lds_dst, uint64_t base, uint32_t voff) {      // (gemv_set.hip set_dma16_cached: K rows are read by every query tile)
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(base) : "memory");
}
template <int QW>
__global__ void __launch_bounds__(64 * QW) __attribute__((amdgpu_waves_per_eu(3)))
k_attnq_scores_lds(const float *__restrict__ qr, const float *__restrict__ Kc, float *__restrict__ S, float *__restrict__ pmax,
                   int n_past, int N, int nb0, int NB, int nb, int d, int T, float kq_scale, int KS) {
    constexpr int TILE_B = 9 * 1024;                                    // 16 rows x 528 B = 8 448 B, fetched as nine instructions of 1 KiB
    __shared__ __attribute__((aligned(1024))) uint8_t ktile[2][TILE_B];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, m = lane & 15, kk = lane >> 4, h = blockIdx.y, ks = blockIdx.z;
    const int nl0 = (blockIdx.x * QW + w) * 16;
    const bool active = nl0 < nb;                                       // (the last workgroup of a batch may hold tiles past its end)
    const int Tb = n_past + min(nb0 + nl0 + 16, N);                    // keys any query of this wave's tile can see
    const int Tg = n_past + min(nb0 + min((int) (blockIdx.x * QW + QW) * 16, nb), N);      // ... of the workgroup's last tile: the slice is the workgroup's
    const int per = (Tg + KS - 1) / KS, t0 = ks * per, t1 = min(Tg, t0 + per);
    float mx[4] = { -INFINITY, -INFINITY, -INFINITY, -INFINITY };
    if (t0 < t1) {                                                      // (uniform per workgroup)
        float aq[32];
        {
            const int nq = min(nb0 + nl0 + m, N - 1);
            const f32x4 *qp = (const f32x4 *) (qr + (size_t) nq * d + h * 128 + 32 * kk);
#pragma unroll
            for (int j = 0; j < 8; j++) { const f32x4 v = qp[j]; aq[4 * j] = v.x; aq[4 * j + 1] = v.y; aq[4 * j + 2] = v.z; aq[4 * j + 3] = v.w; }
        }
        const int tq0 = n_past + nb0 + nl0 + 4 * kk, tqmax = n_past + N - 1;                 // last key query 4 kk + r sees: min(tq0 + r, tqmax)
        const float *kbase = Kc + h * 128;
        float *sbase = S + (size_t) h * T * NB + nl0;
        const uint32_t lds0 = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint8_t *) (uint8_t *) &ktile[0][0];
        // nine DMA instructions of 64 units per tile, instruction i by wave i % QW
        auto fetch = [&](int tb, int buf) {
#pragma unroll
            for (int f = 0; f < (9 + QW - 1) / QW; f++) {
                const int i = w + f * QW;
                if (i < 9) {
                    const int U = 64 * i + lane, row = min((U * 993) >> 15, 15), cu = min(U - 33 * ((U * 993) >> 15), 31);      // (U / 33 for U < 640)
                    const uint32_t voff = (uint32_t) (min(tb + row, t1 - 1) * d + 4 * cu) * 4u;
                    sc_dma16(lds0 + (uint32_t) buf * (uint32_t) TILE_B + (uint32_t) i * 1024u, (uint64_t) (uintptr_t) kbase, voff);
                }
            }
        };
        fetch(t0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int buf = 0;
        for (int tb = t0; tb < t1; tb += 16, buf ^= 1) {
            if (tb + 16 < t1) fetch(tb + 16, buf ^ 1);                  // in flight behind this step's arithmetic (the other buffer's readers passed the previous barrier)
            if (active && tb < Tb) {
                float bk[32];
                {
                    const f32x4 *kp = (const f32x4 *) (&ktile[buf][(m * 33 + 8 * kk) * 16]);
#pragma unroll
                    for (int j = 0; j < 8; j++) { const f32x4 v = kp[j]; bk[4 * j] = v.x; bk[4 * j + 1] = v.y; bk[4 * j + 2] = v.z; bk[4 * j + 3] = v.w; }
                }
                f32x4v D[8];
                float ra[4][4], u[4][4], vsum[4][4];
#pragma unroll
                for (int half = 0; half < 2; half++) {                       // j 0-3, then j 4-7
                    LH_SC_GROUP(D, 0, 4 * half);
#pragma unroll
                    for (int j = 0; j < 4; j++)
#pragma unroll
                        for (int r = 0; r < 4; r++) ra[j][r] = D[j][r] + D[j + 4][r];
                    LH_SC_GROUP(D, 1, 4 * half);
#pragma unroll
                    for (int j = 0; j < 4; j++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const float uj = ra[j][r] + (D[j][r] + D[j + 4][r]);
                            if (half == 0) u[j][r] = uj; else vsum[j][r] = u[j][r] + uj;
                        }
                }
                float scv[4];
#pragma unroll
                for (int r = 0; r < 4; r++) scv[r] = ((vsum[0][r] + vsum[1][r]) + (vsum[2][r] + vsum[3][r])) * kq_scale;
                const int t = tb + m;                                       // this lane's key
                const int te = min(t1, Tb);
#pragma unroll
                for (int r = 0; r < 4; r++) mx[r] = fmaxf(mx[r], (t < te && t <= min(tq0 + r, tqmax)) ? scv[r] : -INFINITY);
                if (t < te) {
                    f32x4 out;
                    out.x = scv[0]; out.y = scv[1]; out.z = scv[2]; out.w = scv[3];
                    *(f32x4 *) (sbase + (uint32_t) (t * NB + 4 * kk)) = out;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's share of the next tile has landed (and its S stores are out)
            __syncthreads();
        }
    }
    if (!active) return;
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
#undef LH_SC_GROUP


__global__ void __launch_bounds__(1024)
k_attnq_softmax(float *__restrict__ S, const float *__restrict__ pmax, float *__restrict__ inv,
                int n_past, int N, int nb0, int NB, int T, int KS, const uint16_t *__restrict__ T_exp) {
    constexpr int PH = 16;                                // key phases: a thread takes every 16th key of its query
    // the exp table's negative half (score - max <= 0: 64 KB of the 128) in LDS: a wave's 64 table reads are 64 different cache lines
    // for the texture path, a few bank conflicts for LDS
    extern __shared__ double smem_d[];
    double (*part)[64] = (double (*)[64]) smem_d;         // [PH][64]
    uint16_t *lut = (uint16_t *) (smem_d + PH * 64);      // [32768]: lut[i] = T_exp[0x8000 | i]
    for (int i = threadIdx.x; i < 32768 / 8; i += 1024) ((u32x4 *) lut)[i] = ((const u32x4 *) (T_exp + 0x8000))[i];
    __syncthreads();
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
        if (t <= tq) {
            const uint16_t xh = f2h_bits(*sp - mx);       // (exp(+0) = exp(-0): the row's maximum itself reads entry 0 of the negative half)
            e = h2f_bits(((xh & 0x8000) || xh == 0) ? lut[xh & 0x7FFF] : T_exp[xh]);
            sum += (double) e;
        }
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

// V*P partial sums on the matrix cores.  v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fp32 fmaf chain per
// output -- D = fma(a1, b1, fma(a0, b0, C)), one rounding per product, subnormals kept (cdna_hip_programming.md, "Numerics" of the
// f32 MFMAs) -- i.e. exactly acc = fma(v, p, acc) over two consecutive keys, which is what the chain of a (chunk, query, column)
// is.  One wave = 64 queries x the head's 128 columns x ONE chunk of the nth-way key split: 8 accumulator tiles (128 registers),
// per pair of keys 2 loads of probabilities (lane = query) and 4 of values (lane = column), 8 MFMAs = 512 cycles of the matrix
// pipe at the fp32 FMA peak, and no VALU work but the soft_max scale.  A chunk with an odd number of keys is padded with a zero
// pair at the FRONT: fma(0, 0, +0) = +0 leaves the chain's start unchanged (a trailing pad could turn a -0 sum into +0).
// PERROW: a chunked pass (split_keys, chunk > 0) gives the queries of a block different key ranges; a plain eval (one split for every
// row) does not pay for the per-lane range tests (the 2 048-token eval: 200.4 -> see profiles/r04_*prefill*).
typedef float f32x16v __attribute__((ext_vector_type(16)));
// NCB = 32-column tiles per wave: 2 (grid z = 2 nth: the head's columns in two halves, 64 accumulators, four waves per SIMD; rounds 2-3: 4)
// puts twice the waves on the chip -- profiles/r04_t_prefill_pmc.txt: with one wave per (64 queries, head, chunk) a CU held 1.4 waves on average, each
// alone on its SIMD waiting for its own loads, the matrix pipe 41 % busy.
template <bool PERROW, int NCB>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NCB == 4 ? 3 : 4)))
k_attnq_pv_mfma(const float *__restrict__ S, const float *__restrict__ inv, const float *__restrict__ Vc, float *__restrict__ part,
                int n_past, int N, int nb0, int NB, int d, int T, int nth, int chunk) {
    const int lane = threadIdx.x, i = lane & 31, kk = lane >> 5, h = blockIdx.y, th = blockIdx.z % nth, c0 = (blockIdx.z / nth) * (NCB * 32);
    const int q0 = blockIdx.x * 64;
    const int nb_end = min(nb0 + (int) (blockIdx.x + 1) * 64, N);
    const int Tb = n_past + nb_end;
    // key range of chunk th of the split, per QUERY: a chunked pass (split_keys) gives the queries of a block different ranges.  The
    // ranges are monotone in the query index, so the wave walks [lo of its first query, hi of its last) and a lane's probability is
    // zeroed outside its own query's range: fma(v, 0, acc) == acc wherever the zero falls (before the range acc is still +0).
    const int qlast = min(nb0 + q0 + 63, N - 1);
    auto lo_of = [&](int n) { const int Ts = split_keys(n_past, N, n, chunk); return ((Ts + nth - 1) / nth) * th; };
    auto hi_of = [&](int n) { const int Ts = split_keys(n_past, N, n, chunk); const int dcq = (Ts + nth - 1) / nth; return min(min(dcq * th + dcq, Ts), Tb); };   // beyond Tb every P of this block is 0
    const int t0 = lo_of(min(nb0 + q0, N - 1));
    const int t1 = hi_of(qlast);
    const int nA = min(nb0 + q0 + i, N - 1), nB = min(nb0 + q0 + 32 + i, N - 1);
    const int loA = lo_of(nA), hiA = hi_of(nA), loB = lo_of(nB), hiB = hi_of(nB);
    f32x16v D[2][NCB];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < NCB; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) D[a][b][r] = 0.0f;
    const float iv0 = inv[(size_t) h * NB + q0 + i], iv1 = inv[(size_t) h * NB + q0 + 32 + i];
    const int nk = t1 - t0;
    // operands of PF pair-steps in flight: a step is 512 matrix-pipe cycles, a load round trip several times that (8 and 12 steps at two
    // waves per SIMD measured the same 2 048-token eval: 191.9 / 191.0 / 190.8 ms)
    constexpr int PF = 4;
    float pa0[PF], pa1[PF], pb[PF][NCB];
    const int tstart = t0 - (nk > 0 ? (nk & 1) : 0);          // (an even number of steps; the extra front key is outside every range)
#define LH_PVLOAD(ST, TP)                                                                           \
    {                                                                                               \
        const int key_ = (TP) + kk;                                                                 \
        const bool real_ = key_ >= t0 && key_ < t1;      /* (the front pad, and steps past the end) */ \
        const int kc_ = min(max(key_, t0), max(t1 - 1, t0));                                        \
        const float *sp_ = S + ((size_t) h * T + kc_) * NB + q0 + i;                                \
        const float *vp_ = Vc + (size_t) kc_ * d + h * 128 + c0 + i;                                \
        const float s0_ = sp_[0], s1_ = sp_[32];                                                    \
        float v_[NCB];                                                                              \
        _Pragma("unroll") for (int b = 0; b < NCB; b++) v_[b] = vp_[32 * b];                        \
        pa0[ST] = (real_ && (!PERROW || (key_ >= loA && key_ < hiA))) ? s0_ * iv0 : 0.0f;          /* soft_max's final scale (ggml.c:7036-7041) */ \
        pa1[ST] = (real_ && (!PERROW || (key_ >= loB && key_ < hiB))) ? s1_ * iv1 : 0.0f;          \
        _Pragma("unroll") for (int b = 0; b < NCB; b++) pb[ST][b] = real_ ? v_[b] : 0.0f;           \
    }
    if (nk > 0) {
#pragma unroll
        for (int st = 0; st < PF; st++) LH_PVLOAD(st, tstart + 2 * st)
        for (int tp = tstart; tp < t1; tp += 2 * PF) {
#pragma unroll
            for (int st = 0; st < PF; st++) {
                if (tp + 2 * st < t1) {
                    const float a0 = pa0[st], a1 = pa1[st];
                    float bv[NCB];
#pragma unroll
                    for (int b = 0; b < NCB; b++) bv[b] = pb[st][b];
                    LH_PVLOAD(st, tp + 2 * (st + PF))
#pragma unroll
                    for (int b = 0; b < NCB; b++) D[0][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[b], D[0][b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < NCB; b++) D[1][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[b], D[1][b], 0, 0, 0);
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
            float *o = part + (((size_t) th * gridDim.y + h) * NB + nl) * 128 + c0 + i;
#pragma unroll
            for (int b = 0; b < NCB; b++) o[32 * b] = D[a][b][r];
        }
}

// merged[n][h*128 + c] = part[0] + part[1] + ... in thread order (ggml.c:5553-5577).  grid (NB/2, H), 256 threads = 2 queries x 128 columns
__global__ void __launch_bounds__(256)
k_attnq_merge(const float *__restrict__ part, float *__restrict__ merged, int N, int nb0, int NB, int d, int nth) {
    const int c = threadIdx.x & 127, nl = blockIdx.x * 2 + (threadIdx.x >> 7), h = blockIdx.y, H = gridDim.y;
    const int n = nb0 + nl;
    if (n >= N) return;
    float s = part[(((size_t) 0 * H + h) * NB + nl) * 128 + c];
    for (int th = 1; th < nth; th++) s += part[(((size_t) th * H + h) * NB + nl) * 128 + c];
    merged[(size_t) n * d + h * 128 + c] = s;
}


hipError_t launch_rope_kv(const float *qkv, long qkv_stride, int d, int dh, const double *tab,
                          float *qr, float *Kc, float *Vc, int n_past, int N, hipStream_t st) {
    hipLaunchKernelGGL(k_rope_kv, dim3(N), dim3(256), 0, st, qkv, qkv_stride, d, dh, tab, qr, Kc, Vc, n_past);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_attn(const float *qr, const float *Kc, const float *Vc, float *merged, float *dbg_p, float *dbg_kqv,
                       int n_past, int N, int d, int H, int nth, const uint16_t *T_exp, const AttnWs *ws, hipStream_t st, int chunk) {
    const int dh = d / H, T = n_past + N;
    const float kq_scale = 1.0f / sqrtf((float) d / (float) H);          // .mm:620
    if (ws && ws->S && dh == 128 && N >= 2 && !dbg_p && !dbg_kqv && T <= ws->T_cap && nth <= ws->nth_cap) {
        for (int nb0 = 0; nb0 < N; nb0 += ws->NB) {
            const int nb = min(ws->NB, N - nb0), qb = (nb + 63) / 64;
            // scores: 16 queries per wave, ~4 waves per SIMD over key slices
            const int qt = (nb + 15) / 16;
            int KS = (4096 + qt * H - 1) / (qt * H);         // (8 192 / 16 384 waves per launch measured the same: profiles/r04_u_attn_ab.txt)
            KS = KS < 1 ? 1 : KS > ws->KS_cap ? ws->KS_cap : KS;
            {
                constexpr int QW = 4;                         // query tiles sharing a key tile (8 measured slower: fewer, larger workgroups)
                hipLaunchKernelGGL((k_attnq_scores_lds<QW>), dim3((qt + QW - 1) / QW, H, KS), dim3(64 * QW), 0, st, qr, Kc, ws->S, ws->pmax, n_past, N, nb0, ws->NB, nb, d, T, kq_scale, KS);
            }
            LH_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_attnq_softmax, dim3(qb, H), dim3(1024), 16 * 64 * sizeof(double) + 65536, st, ws->S, ws->pmax, ws->inv, n_past, N, nb0, ws->NB, T, KS, T_exp);
            LH_LAUNCH_CHECK();
            // (column quarters, and 1 024 / 2 048 queries per launch: measured level / slower, profiles/r04_v_attn_shapes_ab.txt)
            if (chunk > 0) hipLaunchKernelGGL((k_attnq_pv_mfma<true, 2>), dim3(qb, H, nth * 2), dim3(64), 0, st, ws->S, ws->inv, Vc, ws->part, n_past, N, nb0, ws->NB, d, T, nth, chunk);
            else hipLaunchKernelGGL((k_attnq_pv_mfma<false, 2>), dim3(qb, H, nth * 2), dim3(64), 0, st, ws->S, ws->inv, Vc, ws->part, n_past, N, nb0, ws->NB, d, T, nth, 0);
            LH_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_attnq_merge, dim3((nb + 1) / 2, H), dim3(256), 0, st, ws->part, merged, N, nb0, ws->NB, d, nth);
            LH_LAUNCH_CHECK();
        }
        return hipSuccess;
    }
    const size_t lds = 32 * sizeof(double) + ((size_t) T + (size_t) nth * dh + dh + 16) * sizeof(float);
    hipLaunchKernelGGL(k_attn, dim3(H, N), dim3(256), lds, st, qr, Kc, Vc, merged, dbg_p, dbg_kqv, n_past, N, d, dh, nth, kq_scale, T_exp, chunk);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}


hipError_t init_attrs_prompt_attn() {
    const int cap = 160 * 1024;          // fused prologues / wide rows need more than the default 64 KB of dynamic LDS
#define LH_ATTR(KERNEL) do { hipError_t e_ = hipFuncSetAttribute((const void *) KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, cap); if (e_ != hipSuccess) return e_; } while (0)
    LH_ATTR(k_attn);
    LH_ATTR(k_attnq_softmax);
#undef LH_ATTR
    return hipSuccess;
}

}  // namespace lh
