// llamahip_internal.h -- shared between the HIP kernels (kernels.hip) and the C-ABI host code.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace lh {

constexpr int MT4_TILE_BYTES = 4608;   // k_gemm_mfma4's weight tile: 32 rows x 4 Q4_0 blocks, one byte per weight + 128 fp32 scales
constexpr int TILE_BYTES = 1280;   // 8 rows x 8 Q4_0 blocks: 1024 B nibbles + 64 fp32 scales

// activation-preparation modes (also the fused-prologue selector of k_gemv)
// (PREP_NORMP: PREP_NORM with the row's {sum x, sum x^2} supplied by its producer -- k_gemv only, selected by launch_gemv)
// (the *_TAG forms: the operand arrives / the result leaves as 8-byte {fp32 bits, tag} granules that the consumer polls -- the
//  hand-offs inside k_qkv_attn, and the residual-stream row between pipeline stages through a device-side mailbox)
enum { PRE_QA = 0, PREP_PLAIN = 1, PREP_NORM = 2, PREP_SILU_MUL = 3, PREP_NORMP = 4, PREP_NORM_TAG = 6 };
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_SILU_QA = 2, EPI_ROPE_KV = 3, EPI_STORE_TAG = 4, EPI_RESID_TAG = 5, EPI_STORE_PICK = 6, EPI_SILU_QAH = 7 };
// The rows of a BATCHED decode step (llamahip_stage_step_set): row b is the next token of sequence slot b' -- its own position (device
// resident, so a captured step replays unchanged), its own KV cache, its own token / pick / residual-stream buffers.  Lives in device
// memory; kernels that take a `const SeqSet *` treat null as "one sequence, consecutive positions" (the prompt-chunk meaning).
constexpr int SET_MAX = 16;
struct SeqSet {
    int32_t *state[SET_MAX];            // {position, step} of the row's slot
    const int32_t *tok_in[SET_MAX];     // first stage: where the row's token is read
    int32_t *tok_out[SET_MAX];          // last stage: where the greedy pick goes (may be null)
    int32_t *trace[SET_MAX];            // last stage: the slot's list of picks
    const float *hid_in[SET_MAX];       // stages after the first: the row of the residual stream coming in
    float *hid_out[SET_MAX];            // stages before the last: ... going out
    long kv_off[SET_MAX];               // elements from slot 0's KV cache to the row's slot's
    int32_t pos[SET_MAX];               // the rows' positions of THIS step (= *state[i]), gathered by the step's first launch (k_embed_set / k_rows_set): one
                                        // load instead of a pointer chase at the tail of the wq|wk|wv launch (written by the device, every step)
    int n;
};
// operands of the EPI_ROPE_KV epilogue (short evals, wq|wk|wv): rotate q / k, append k / v to the cache
// (set: row n is at position set->state[n][0] of the cache at Kc / Vc + set->kv_off[n] instead of n_past + n)
struct RopeKvArgs { const double *tab; float *qr, *Kc, *Vc; int n_past, d, dh; const SeqSet *set = nullptr; };
// what the half-block w1|w3 epilogue of k_gemv_set needs (EPI_SILU_QAH): granules for the partial amaxes, the epoch word its tags are made from
struct SiluHalfIO { uint64_t *amax_t = nullptr; const uint32_t *epoch = nullptr; int layer = 0; uint32_t *fault = nullptr; };

// a Q4_0 weight matrix resident in HBM in chain-major tile layout
struct QMat {
    uint8_t *tiles = nullptr;   // ngroups * (nchunks + 1) * TILE_BYTES; the last tile of a row-group is all-zero
    int M = 0, K = 0;           // logical rows / columns
    int ngroups = 0;            // ceil(M / 8)
    int nchunks = 0;            // ceil(K / 256)
    int gmapF8 = 0;             // F/8 for the interleaved w1|w3 matrix, else 0 (see k_repack_q4)
    // optional second copy for the prompt path: row-lane tiles, nrb * (nchunks + 1) * 10240 B (see k_gemm_rows);
    // rows are in LOGICAL order here (no w1|w3 interleave)
    uint8_t *rows = nullptr;
    int nrb = 0;                // ceil(M / 64)
    // optional third copy for long prompts: MFMA tiles, nrb32 * (2 * nchunks) * 2560 B (see k_gemm_mfma)
    uint8_t *mt = nullptr;      // int8 operand order (the opt-in fast path and LLAMAHIP_MFMA_I8)
    uint8_t *mt4 = nullptr;     // one byte per weight, four-chain operand order (k_gemm_mfma4: the exact path): nrb32 * (2 * nchunks) * 4608 B
    int nrb32 = 0;              // ceil(M / 32)
    size_t mt_bytes() const { return (size_t) nrb32 * nchunks * 2 * 2560; }
    size_t mt4_bytes() const { return (size_t) nrb32 * nchunks * 2 * MT4_TILE_BYTES; }
    size_t rows_bytes() const { return (size_t) nrb * (nchunks + 1) * 10240; }
    size_t bytes() const { return (size_t) ngroups * (nchunks + 1) * TILE_BYTES; }
    int Kp() const { return nchunks * 256; }
};

// an f16 / f32 weight matrix (file layout: row-major [M][K]) of a dense model file (dense.hip)
struct DMat {
    void *w = nullptr;          // fp32 / fp16: the rows; Q4_1: {min, d} pairs [row-block of 64][block][lane]
    void *w2 = nullptr;         // Q4_1: the 16 nibble bytes in the same order
    int M = 0, K = 0;
    int wtype = 0;              // 0 fp32, 1 fp16, 3 Q4_1
    size_t bytes() const { return wtype == 3 ? (size_t) ((M + 63) / 64) * 64 * (K / 32) * 8 : (size_t) M * K * (wtype == 1 ? 2 : 4); }
    size_t bytes2() const { return wtype == 3 ? (size_t) ((M + 63) / 64) * 64 * (K / 32) * 16 : 0; }
};
// scratch: N * K floats (the permuted / rounded activation operand; Q4_1: the expanded one)
hipError_t launch_dense_mm(const DMat &w, int epi, const float *x, long x_stride, int N, float *y, long y_stride,
                           const float *resid, long resid_stride, hipStream_t st, float *scratch = nullptr);
hipError_t launch_q41_repack(const uint8_t *raw_rows, DMat &w, hipStream_t st);
bool dense_prep_applies(int wtype, int mode, int K);
hipError_t launch_dense_prep(int mode, int wtype, const float *in0, const float *in1, long in_stride, long in1_stride,
                             int K, int N, float *scratch, const uint16_t *T_silu, hipStream_t st);
hipError_t launch_dense_perm_rows(const void *raw_rows, DMat &w, int row0, int rows, hipStream_t st);
hipError_t launch_quantize_q41_offline(const void *src, int f16, uint8_t *dst, long nrows, int nb, hipStream_t st);
hipError_t launch_embed_dense(const int32_t *tokens, const void *emb, int wtype, float *x, int d, int N, hipStream_t st);

hipError_t init_kernel_attrs();
size_t prep_lds_bytes(int K);                                       // dynamic LDS of the LDS-staged activation preparation (prep.hip)
// exhaustive check (all 65 536 entries) that the device's double-precision formulas reproduce the host-built SiLU /
// exp tables; enables the gather-free paths of the decode kernels when they do (g_lut_math)
extern int g_lut_math;
hipError_t launch_check_lut_math(const uint16_t *T_silu, const uint16_t *T_exp, hipStream_t st);
hipError_t set_phase_probe(unsigned long long *dev_buf);

hipError_t launch_add(const float *a, const float *b, float *c, long n, hipStream_t st);
hipError_t launch_repack(const uint8_t *src_aos, uint8_t *dst, int M, int K, int gmap, int goff, hipStream_t st);
hipError_t launch_tiles_to_rows(const QMat &w, hipStream_t st);   // w.rows / w.nrb set by the caller
hipError_t launch_tiles_to_mtiles(const QMat &w, hipStream_t st); // w.mt / w.nrb32 set by the caller
hipError_t launch_tiles_to_mt4(const QMat &w, hipStream_t st);    // w.mt4 / w.nrb32 set by the caller
hipError_t launch_embed(const int32_t *tokens, const uint8_t *emb, float *x, int d, int N, hipStream_t st);
hipError_t launch_prep(int mode, const float *in0, const float *in1, long in_stride, long in1_stride, int K, int N,
                       uint32_t *qa_A, float *qa_d, float *y_out, uint8_t *raw_out, const uint16_t *T_silu,
                       hipStream_t st);
// Norm statistics handed from the producer of a residual-stream row to the norm-fused mat-vec that reads it
// (decode only): an EPI_RESID launch with `out` set writes one {sum y, sum y^2} pair of doubles per workgroup
// (gemv_resid_parts(w) of them); a PREP_NORM launch with `in` / `n_in` set folds them instead of reducing the
// row itself.  n_in <= NORM_PART_MAX.
constexpr int NORM_PART_MAX = 512;
struct NormPart { const double *in = nullptr; int n_in = 0; double *out = nullptr; };
// Pipeline mailbox on one side of a decode launch: the residual-stream row as tagged 8-byte granules in memory the NEIGHBOUR stage
// reads / writes (peer-mapped).  Tags are made from the sequence position (*pos_w + 1), which both stages know.
struct MailboxIO {
    const uint64_t *in_t = nullptr;      // the row the first layer's wq|wk|wv launch normalises arrives here ...
    const uint64_t *resid_t = nullptr;   // ... and the same granules are the residual operand of the first layer's wo launch
    uint64_t *out_t = nullptr;           // the last layer's w2 launch stores its row here (the next stage's inbox)
    const int32_t *pos_w = nullptr;      // device word holding the position (the slot's st[0])
    uint32_t *epoch = nullptr, *fault = nullptr;
    int test_bits = 0;                   // 0x1000 short polls, 0x2000 wrong tag on out_t (fault-injection tests)
};
// EPI_STORE_PICK (the lm head of the device-resident greedy loop): the launch that writes the logits also picks the token (k_argmax's
// rule: the largest value, the LOWEST index on ties, NaN never wins) -- every workgroup folds its rows into one 64-bit atomic max, the
// last workgroup to finish records the pick, advances the position and embeds the picked token for the next step (k_embed_part's
// arithmetic: row, {sum x, sum x^2}, epoch bump), so a decode step has no single-workgroup launches left.
struct PickIO {
    unsigned long long *key; uint32_t *count;       // key: one slot per workgroup of the launch; count: 8 shard tickets (16 dwords apart) + the top ticket at [128]; zeroed once, re-zeroed by the last workgroup
    int32_t *out; int32_t *next_token; int32_t *state;      // as launch_argmax: out[state[1]] = pick, *next_token = pick, state advances
    const uint8_t *emb; float *x_next; double *part_next; uint32_t *epoch; int n_vocab;      // the next step's embedding row
};
int gemv_resid_parts(const QMat &w);
// EPI_SILU_QAH: the w1|w3 decode mat-vec in HALF-block workgroups (4 waves: 16 gate rows + the same 16 up rows).  The 8-wave workgroups of
// EPI_SILU_QA are F / 32 = 344 on 256 CUs at 7B: 88 CUs stream two workgroups' weights, the others one, and a CU's load path bounds what
// it can pull -- the launch ends with a third of the chip streaming alone.  688 half-block workgroups spread 3 / 2 per CU.  The two halves
// of a Q4_0 activation block sit 8 blocks apart in the grid (one XCD: xcd_selftest) and exchange their partial amax as one tagged granule
// each way (fmaxf is exact in any order), then each writes its 16-bit halves of the block's eight QA dwords.
bool gemv_silu_half_applies(const QMat &w);
hipError_t launch_gemv_silu_half(const QMat &w, const float *in0, const float *in1, const uint16_t *T_silu, uint32_t *out_A, float *out_d, hipStream_t st,
                                 const NormPart *np, uint64_t *amax_t, uint32_t *epoch, int layer, uint32_t *fault);
bool gemv_pick_applies(const QMat &w);
hipError_t launch_gemv_pick(const QMat &w, const float *in0, const float *in1, float *y, const uint16_t *T_silu, hipStream_t st,
                            const NormPart *np, const PickIO &pick);
hipError_t launch_gemv(const QMat &w, int pre, int epi, const uint32_t *qa_A, const float *qa_d,
                       const float *in0, const float *in1, float *y, const float *resid,
                       const uint16_t *T_silu, uint32_t *out_A, float *out_d, hipStream_t st,
                       const NormPart *np = nullptr, const MailboxIO *mb = nullptr);
// embedding row of ONE token (decode) + its {sum x, sum x^2} pair for the first norm (part_out[0])
hipError_t launch_embed_part(const int32_t *token, const uint8_t *emb, float *x, int d, double *part_out, hipStream_t st, uint32_t *epoch = nullptr, uint64_t *xt = nullptr,
                             const uint64_t *token_mb = nullptr, const int32_t *state = nullptr, uint32_t *fault = nullptr, int n_vocab = 0);      // token_mb: the token arrives as a mailbox granule
enum { GEMM_PATH_MFMA = 0, GEMM_PATH_ROWS = 1, GEMM_PATH_LDS = 2, GEMM_PATH_GEMV = 3, GEMM_PATH_SET = 4, GEMM_PATH_COUNT = 5 };
extern long g_gemm_path_counts[GEMM_PATH_COUNT];     // launches per kernel family of launch_gemm (process-wide; tests)
// qb_ws: scratch for the int8 operand of the matrix-core path (N * nchunks * 256 B), or nullptr
hipError_t launch_gemm(const QMat &w, int epi, const uint32_t *qa_A, const float *qa_d, int N,
                       float *y, long y_stride, const float *resid, long resid_stride, hipStream_t st,
                       uint8_t *qb_ws = nullptr, bool fast = false);
hipError_t launch_rope_kv(const float *qkv, long qkv_stride, int d, int dh, const double *tab,
                          float *qr, float *Kc, float *Vc, int n_past, int N, hipStream_t st);
// workspace of the many-row prompt attention (k_attnq_*): scores [H][T_cap][NB] fp32 + per-query max / 1/sum
struct AttnWs {
    float *S = nullptr, *pmax = nullptr, *inv = nullptr, *part = nullptr;   // part: [nth_cap][H][NB][128]
    int NB = 0;        // query rows per batch (multiple of 64)
    int T_cap = 0;     // keys the workspace can hold
    int KS_cap = 32;   // key slices of the score pass
    int nth_cap = 8;   // chunks of the V*P key split the workspace can hold (larger n_threads: per-row kernel)
};
hipError_t launch_attn(const float *qr, const float *Kc, const float *Vc, float *merged, float *dbg_p, float *dbg_kqv,
                       int n_past, int N, int d, int H, int nth, const uint16_t *T_exp, const AttnWs *ws, hipStream_t st,
                       int chunk = 0);       // chunk > 0: the pass stands for successive evals of `chunk` rows (prompt_attn.hip split_keys)
bool gemm_rope_kv_applies(const QMat &wqkv, int N, int d);
hipError_t launch_gemm_rope_kv(const QMat &wqkv, const uint32_t *qa_A, const float *qa_d, int N, const RopeKvArgs &ra, hipStream_t st);
bool gemm_silu_qa_applies(const QMat &w13, int N);
hipError_t launch_gemm_silu_qa(const QMat &w13, const uint32_t *qa_A, const float *qa_d, int N, const uint16_t *T_silu,
                               uint32_t *out_A, float *out_d, long out_strideA, long out_strideD, hipStream_t st,
                               const SiluHalfIO *hx = nullptr);      // hx: the half-block exchange of k_gemv_set may be used (2 .. 16 rows)
// k_gemv_set (gemv_set.hip): the mat-mul for 2 .. 16 activation rows -- a batched decode step's rows, the reference's 9-token evals.  The waves
// that share a row-group share its weight bytes through LDS, so every weight byte crosses a CU's load path once per launch.
//   gemv_set_applies(w, N, epi): EPI_STORE / EPI_RESID (launch_gemv_set), EPI_ROPE_KV (launch_gemv_set_rope_kv), EPI_SILU_QAH (launch_gemv_set_silu:
//   interleaved w1|w3 in half-block workgroups whose halves exchange their partial amax per column as tagged granules -- needs the XCD
//   placement xcd_selftest confirmed, the epoch word and SET_AMAX_GRANULES(F) * 16 zeroed granules)
inline size_t set_amax_granules(int F) { return (size_t) F / 16 + 16; }      // per column
bool gemv_set_applies(const QMat &w, int N, int epi);
bool gemv_set_plan_query(int M, int K, bool interleaved, int N, int epi, long out[5]);      // host-only: the (nc, cw, ncg, rgw, LDS) plan, false = the kernel does not take the shape
bool gemv_set_silu_whole_blocks(const QMat &w13, int N, bool have_exchange);      // EPI_SILU_QA (whole-block workgroups, no exchange) instead of EPI_SILU_QAH
hipError_t launch_gemv_set(const QMat &w, int epi, const uint32_t *qa_A, const float *qa_d, int N,
                           float *y, long y_stride, const float *resid, long resid_stride, hipStream_t st);
hipError_t launch_gemv_set_rope_kv(const QMat &wqkv, const uint32_t *qa_A, const float *qa_d, int N, const RopeKvArgs &ra, hipStream_t st);
hipError_t launch_gemv_set_silu(const QMat &w13, const uint32_t *qa_A, const float *qa_d, int N, const uint16_t *T_silu,
                                uint32_t *out_A, float *out_d, long out_strideA, long out_strideD, const SiluHalfIO &hx, hipStream_t st);
hipError_t init_attrs_gemv_set();
long set_probe_dump(unsigned long long *out, long cap_records, bool reset);      // LH_SET_PROBE builds (tools/set_timeline.py): records of 32 words
hipError_t launch_attn_short(const float *qr, const float *Kc, const float *Vc, float *sc, float *merged,
                             uint32_t *qa_A, float *qa_d, int n_past, int N, int d, int H, int n_ctx, int nth,
                             const uint16_t *T_exp, hipStream_t st, int chunk = 0, const SeqSet *set = nullptr,      // set: N independent single-row evals (batched decode step)
                             int set_keys = 0);                                                                   // ... whose positions are all < set_keys (0: n_ctx): bounds the score grid
// batched decode step: embedding rows / residual rows in and out / greedy picks of the set's rows
// (the step's FIRST launch -- embed, or rows with gather -- also writes set->pos and, given the epoch word, opens the step's epoch)
hipError_t launch_embed_set(SeqSet *set, int n, const uint8_t *emb, float *x, int d, hipStream_t st, uint32_t *epoch = nullptr);
hipError_t launch_rows_set(SeqSet *set, int n, float *x, int d, bool gather, hipStream_t st, uint32_t *epoch = nullptr);      // gather: hid_in -> x rows; else x rows -> hid_out
hipError_t launch_argmax_set(const float *logits, int V, const SeqSet *set, int n, hipStream_t st);   // + trace, tok_out, position advance
hipError_t launch_advance_set(const SeqSet *set, int n, hipStream_t st);
hipError_t launch_dec_attn(const float *qkv, int d, int H, int n_ctx, int nth, const double *tab, float *Kc, float *Vc,
                           float *sc, float *part, float *merged, uint32_t *qa_A, float *qa_d,
                           const uint16_t *T_exp, const int32_t *state, hipStream_t st,
                           uint32_t *xsync = nullptr, uint32_t *fault = nullptr,      // xsync: H * 32 zeroed dwords -> single-launch k_dec_attn_x
                           bool long_ctx = false,                                     // long_ctx: k_dec_scores + k_dec_pv_stream (pv_stream_applies) ...
                           uint64_t *xpart = nullptr, const uint32_t *epoch = nullptr, int layer = 0);      // ... its chains split over workgroups: [H dh/32][nth][32] granules, tag = (epoch, layer + 1)
bool pv_stream_applies(int dh, int n_ctx, int nth);
bool xcd_selftest(int H, int Y, hipStream_t st);
// wq|wk|wv mat-vec + decode attention as one launch (k_qkv_attn); xsync / fault as launch_dec_attn
constexpr int TAG_MAX_LAYERS = 250;                                 // hand-off tags carry the layer index in 8 bits (kernels.hip make_tag)
bool qkv_attn_applies(const QMat &w, int d, int H, int nth);
hipError_t launch_qkv_attn(const QMat &w, const float *x, const float *norm_w, const NormPart &np, uint64_t *qkv2, uint64_t *sc2, uint32_t *epoch, int layer,
                           int d, int H, int n_ctx, int nth, const double *tab, float *Kc, float *Vc, float *merged, uint32_t *qa_A, float *qa_d,
                           const uint16_t *T_silu, const uint16_t *T_exp, const int32_t *state, uint32_t *fault, hipStream_t st,
                           const MailboxIO *mb = nullptr);      // mb->in_t: the row arrives as tagged granules (pipeline mailbox)
hipError_t launch_bump_epoch(uint32_t *epoch, hipStream_t st);
// a residual-stream row re-published as tagged granules, slot 0 of the current epoch (pipeline mailbox tests, single-GPU stage chains)
hipError_t launch_quantize_offline(const void *src, int f16, uint8_t *dst, long nblocks, hipStream_t st);
hipError_t launch_advance(int32_t *state, hipStream_t st);
// sampler front end (utils.cpp:345-395): the k best candidate scores of the last row of logits, on the device
hipError_t launch_topk_candidates(const float *logits, int V, const int32_t *window, int n_window, double scale, double repeat_penalty, int k,
                                  double *out_score, int32_t *out_id, int32_t *flags, hipStream_t st, void *ws = nullptr);      // ws: TOPK_WS_BYTES zeroed once -> the two-launch variant
constexpr size_t TOPK_WS_BYTES = 32768 * 8 + 64 * 8 + 64;
hipError_t launch_argmax(const float *logits, int V, int32_t *out, int out_idx, int32_t *next_token, int32_t *state, hipStream_t st, uint64_t *token_mb = nullptr);

}  // namespace lh
