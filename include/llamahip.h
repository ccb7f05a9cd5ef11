/* llamahip.h -- C ABI of libllamahip.so: the MI355X (gfx950) drop-in for the quantized-LLaMA hot
 * path of alexrozanski/llama.swift.
 *
 * The two entry points the reference's Objective-C++ bridge binds are
 *     llama_model_load()  Sources/llamaObjCxx/bridge/LlamaPredictOperation.mm:98
 *     llama_eval()        Sources/llamaObjCxx/bridge/LlamaPredictOperation.mm:510-518
 * (called from -[LlamaPredictOperation main], .mm:790, :822, :840; model released at :900).
 * Everything else here is either an accessor the caller needs because the model is now an opaque
 * handle (vocab, hparams), the host-side text utilities the same caller uses
 * (Sources/cpp/utils.cpp:275-311 tokenizer, :345-428 sampler), or measurement / debug hooks.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types cross this boundary;
 *   - return 0 on success, LLAMAHIP_ERR_LOAD (-1000) / LLAMAHIP_ERR_PREDICT (-1001) on failure --
 *     the values of LlamaErrorCodeFailedToLoadModel / LlamaErrorCodePredictionFailed
 *     (Sources/llamaObjCxx/headers/LlamaError.h:14-19); a UTF-8 message is written to `err`;
 *   - calls are synchronous; a handle is used by one thread at a time; distinct handles are
 *     independent (no process-wide scratch, unlike the reference's static buffer, .mm:532-533);
 *   - the library never falls back to a CPU path: without a HIP device every compute entry point
 *     fails with an error.
 *
 * Numerics: results are computed with the arithmetic order of the reference's x86 AVX2+FMA+F16C
 * build, so logits are expected to be bit-identical to that build for the same
 * `n_threads` (the reference's attention V*P product depends on its thread count,
 * Sources/cpp/ggml.c:5619-5665 / 5553-5577; `n_threads` selects the same split here).
 * One operator is exact BY TEST, NOT BY CONSTRUCTION: the norm (ggml.c:5327-5385).  Its two double-precision sums are formed in a
 * different association order than the reference's sequential loops, and the single-token decode kernels by default take the second
 * moment from the producer's partial sums in one pass (sum x^2 - mean * sum x, guarded against cancellation) instead of the reference's
 * two-pass form: both carry a few 2^-53 of rounding before the result is narrowed to fp32, where a difference survives with
 * probability ~2^-29 per row.  No parity test (504-token full-depth traces, rows with a DC offset on both sides of the guard) has ever
 * observed one; LLAMAHIP_NORM_MODE=0 selects the reference's two-pass formula in every decode prologue (prompt evals always use it).
 */
#ifndef LLAMAHIP_H
#define LLAMAHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the declarations of this header are its whole dynamic symbol table. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define LLAMAHIP_OK            0
#define LLAMAHIP_ERR_UNKNOWN   (-1)      /* LlamaErrorCodeUnknown            (LlamaError.h:15) */
#define LLAMAHIP_ERR_LOAD      (-1000)   /* LlamaErrorCodeFailedToLoadModel  (LlamaError.h:17) */
#define LLAMAHIP_ERR_PREDICT   (-1001)   /* LlamaErrorCodePredictionFailed   (LlamaError.h:18) */

typedef struct llamahip_model llamahip_model;

/* Optional load-time options (pass NULL for defaults).  Not part of the reference surface: these
 * are the knobs SURVEY.md section 5 "Config / flags" routes through an extended struct so the
 * Swift API stays unchanged. */
typedef struct llamahip_opts {
    int32_t struct_size;   /* sizeof(llamahip_opts) */
    int32_t device;        /* HIP device ordinal; -1 = current device */
    int32_t layer_begin;   /* pipeline stage: first layer held by this handle (0) */
    int32_t layer_end;     /* one past the last layer; -1 = n_layer */
    int32_t n_parts;       /* 0 = by n_embd as the reference (.mm:33-38; unknown widths -> 1) */
    int32_t flags;         /* LLAMAHIP_FLAG_* */
    int32_t n_seq;         /* independent KV caches held by this handle (pipeline micro-batching); 0 = 1 */
    int32_t n_devices;     /* > 1: an in-process layer pipeline, stage s = an even share of the layers on devices[s] (below); 0 / 1 = `device` */
    int32_t devices[8];    /* HIP device ordinals of the stages, in layer order (an ordinal may repeat: several stages on one GPU) */
} llamahip_opts;
#define LLAMAHIP_MAX_DEVICES 8
/* In-process layer pipeline (SURVEY.md section 8e behind the reference's own surface): the bridge makes ONE llama_model_load call from ONE
 * process (.mm:790; LlamaRunnerBridge.mm:18-26).  A handle loaded with n_devices > 1 -- or, for a caller that passes no options such as the
 * replacement bridge, with the environment variable LLAMAHIP_DEVICES="0,1,...,7" (or a count: "8" = devices 0 .. 7) -- holds one stage per
 * device; llamahip_eval / llamahip_eval_chunks / llamahip_eval_topk / llamahip_decode_greedy / llamahip_kv_read / llamahip_get_stats and the
 * llama_runner_* driver work on it unchanged, the residual stream (.mm:563-564, 687-690) crosses devices as stream-ordered peer copies.
 * Waiting for a stage is bounded: LLAMAHIP_PIPE_WATCHDOG_S seconds (default 600) without the stage's stream completing is LLAMAHIP_ERR_PREDICT, not a hang.
 * Results are bit for bit the single-device handle's.  The stage-level entry points (llamahip_eval_stage, llamahip_stage_*) and
 * llamahip_eval_debug's dumps refuse such a handle.  LLAMAHIP_DEVICES never applies to a handle loaded with an explicit device, layer range
 * or LLAMAHIP_FLAG_HOST_ONLY. */

#define LLAMAHIP_FLAG_NO_GRAPH   1   /* launch decode kernels eagerly instead of via hipGraph */
#define LLAMAHIP_FLAG_UNFUSED    2   /* use the separate prepare+GEMV kernels for decode */
#define LLAMAHIP_FLAG_HOST_ONLY  4   /* parse + validate the file and vocab only (no device work): the
                                        handle serves tokenize / token_text / tensor_bytes / sampler;
                                        every compute call on it fails with LLAMAHIP_ERR_PREDICT */
#define LLAMAHIP_FLAG_NO_PREFILL_COPY 8 /* never build the two extra copies of the layer matrices (row-lane
                                        tiles, MFMA tiles) that the multi-token prompt GEMMs read (they are
                                        built lazily, by the first eval of 61+ rows): a third of the weight
                                        memory, long prompt evals fall back to the slower LDS-staged GEMM.
                                        Results are bit-identical either way. */

#define LLAMAHIP_FLAG_FAST_PREFILL 16 /* OPT-IN, NOT the reference's arithmetic: multi-token evals that take the matrix-core
                                        GEMM (>= 64 rows at the 7B shapes) add each Q4_0 block's 32 integer products in one
                                        MFMA and run ONE fp32 accumulation chain per output instead of the reference's eight
                                        (ggml.c:1415-1466).  Same weights, same activation codes; sums re-associated, so a mat-mul
                                        agrees to rounding only -- and every flipped 4-bit activation code downstream amplifies
                                        that: after 32 layers of the thin-margin synthetic 7B the logits differ by 0.9 on
                                        average (cosine 0.986).  2.2x the exact path.  Decode and short evals are unaffected.
                                        Never the default, never used for a parity claim. */

/* ---- the drop-in boundary ------------------------------------------------------------------ */

/* Replaces llama_model_load(fname, model, vocab, n_ctx, &error)  (.mm:98).
 * Parses ggml-model-q4_0.bin[.1 ...] (magic 0x67676d6c, 7 int32 hparams, vocab, tensors; multi-part
 * shards merged as .mm:312-495), uploads and repacks the weights, allocates the fp32 KV cache
 * (.mm:290-304) on the device. */
int llamahip_model_load(const char *path, int32_t n_ctx, const llamahip_opts *opts,
                        llamahip_model **out, char *err, size_t err_cap);

/* Replaces llama_eval(model, n_threads, n_past, embd_inp, embd_w, mem_per_token, &error) (.mm:510).
 * Runs `n_tokens` tokens at context offset `n_past`; writes the n_vocab fp32 logits of the LAST
 * token to `logits_out` (.mm:724-725).  Fails (PREDICT) if n_past + n_tokens > n_ctx. */
int llamahip_eval(llamahip_model *m, int32_t n_threads, int32_t n_past,
                  const int32_t *tokens, int32_t n_tokens, float *logits_out,
                  char *err, size_t err_cap);

/* The caller's prompt loop in one call: the reference feeds a prompt to llama_eval n_batch + 1 = 9 tokens at a time
 * (-[LlamaPredictOperation main], .mm:840-848 + 880-888).  Afterwards the KV cache and `logits_out` (may be NULL) are bit for bit
 * what ceil(n_tokens / chunk_tokens) successive llamahip_eval calls of chunk_tokens tokens (the last one shorter) leave behind:
 * every operator of llama_eval's graph works row by row except the V*P key split, which depends on the eval a row belongs to
 * (ggml.c:5459-5480) and is applied per row.  One pass over all rows runs the matrix-core GEMMs instead of 9-row ones
 * (about 4x the tokens/s of the chunk-by-chunk loop on a 500-token prompt).  n_tokens <= chunk_tokens: llamahip_eval. */
int llamahip_eval_chunks(llamahip_model *m, int32_t n_threads, int32_t n_past,
                         const int32_t *tokens, int32_t n_tokens, int32_t chunk_tokens, float *logits_out,
                         char *err, size_t err_cap);

/* Replaces ggml_free(model.ctx)  (.mm:900). */
void llamahip_model_free(llamahip_model *m);

/* ---- accessors (the reference reads these fields off llama_model / gpt_vocab directly) ------ */
int32_t llamahip_n_vocab(const llamahip_model *m);
int32_t llamahip_n_ctx(const llamahip_model *m);
int32_t llamahip_n_embd(const llamahip_model *m);
int32_t llamahip_n_head(const llamahip_model *m);
int32_t llamahip_n_layer(const llamahip_model *m);
int32_t llamahip_n_ff(const llamahip_model *m);
int32_t llamahip_n_parts(const llamahip_model *m);
/* vocab.id_to_token[id] (utils.h:49-55); returns NULL for an id out of range */
const char *llamahip_token_text(const llamahip_model *m, int32_t id, uint32_t *len);

/* ---- host-side text utilities used by the same caller --------------------------------------- */
/* llama_tokenize(vocab, text, bos)  (utils.cpp:275-311).  Returns the token count (may exceed cap). */
int32_t llamahip_tokenize(const llamahip_model *m, const char *text, int32_t bos,
                          int32_t *out, int32_t cap);

/* Sampler state = std::mt19937 rng(seed) + the last_n_tokens window (.mm:773, :827-829). */
typedef struct llamahip_sampler llamahip_sampler;
llamahip_sampler *llamahip_sampler_new(int32_t seed, int32_t repeat_last_n);
void              llamahip_sampler_free(llamahip_sampler *s);
void              llamahip_sampler_accept(llamahip_sampler *s, int32_t id);   /* .mm:867-868, 882-883 */
/* gpt_random_prompt(rng)  (utils.cpp:102-119): the prompt the caller substitutes for an empty one (.mm:774-776);
 * consumes one draw of the sampler's rng, exactly as the reference's shared std::mt19937 does. */
const char       *llamahip_sampler_random_prompt(llamahip_sampler *s);
/* llama_sample_top_p_top_k  (utils.cpp:345-428) */
int32_t llamahip_sample_top_p_top_k(const llamahip_model *m, llamahip_sampler *s, const float *logits,
                                    double repeat_penalty, int32_t top_k, double top_p, double temp);

/* The same sampler with its first half on the device ("avoids the 128 KB logits copy per token"):
 * llamahip_eval_topk = llamahip_eval + candidate scores (temperature, repetition penalty over `last_n_tokens`) + the
 * top_k selection of utils.cpp:345-395.  With *exact == 1, cand_scores / cand_ids [0, min(top_k, n_vocab)) are the
 * reference's candidates after its partial_sort, and llamahip_sample_from_candidates finishes the draw (soft-max,
 * top-p cut, std::discrete_distribution on the sampler's mt19937: utils.cpp:397-428) -- same ids, same rng draws.
 * With *exact == 0 (two equal scores whose order only libstdc++'s partial_sort defines, a NaN, n_vocab > 32768 or
 * top_k > 64) logits_out holds the n_vocab logits and the caller uses llamahip_sample_top_p_top_k as before.
 * cand_scores / cand_ids: room for 64 entries; logits_out: n_vocab floats. */
int llamahip_eval_topk(llamahip_model *m, int32_t n_threads, int32_t n_past, const int32_t *tokens, int32_t n_tokens,
                       const int32_t *last_n_tokens, int32_t n_last, double repeat_penalty, int32_t top_k, double temp,
                       double *cand_scores, int32_t *cand_ids, int32_t *exact, float *logits_out, char *err, size_t err_cap);
int32_t llamahip_sample_from_candidates(llamahip_sampler *s, const double *scores, const int32_t *ids, int32_t n, double top_p);
/* the sampler's last_n_tokens window (oldest first); returns its length */
int32_t llamahip_sampler_window(const llamahip_sampler *s, int32_t *out, int32_t cap);

/* ---- extensions -------------------------------------------------------------------------------- */

/* Greedy decode loop kept on the device: step i evaluates one token at n_past + i, takes
 * argmax(logits) (lowest index on ties -- the harness' definition of "temperature 0", SURVEY.md
 * fact 8) and feeds it to step i+1 without a host round trip.  out_tokens[i] = token produced by
 * step i.  If logits_last is non-NULL it receives the final step's logits. */
int llamahip_decode_greedy(llamahip_model *m, int32_t n_threads, int32_t n_past, int32_t first_token,
                           int32_t n_steps, int32_t *out_tokens, float *logits_last,
                           char *err, size_t err_cap);

/* Greedy decode of n_seqs INDEPENDENT sequences at once (beyond the reference's surface, whose bridge holds one conversation; SURVEY.md section 8e:
 * "throughput scales only with independent sequences in flight").  Sequence i lives in KV slot i of the handle (llamahip_opts.n_seq >= n_seqs; its
 * context was evaluated with llamahip_set_seq(i) + llamahip_eval / llamahip_eval_chunks), continues at position n_past[i] with first_tokens[i], and
 * gets out_tokens[i * n_steps + t], t < n_steps: bit for bit the tokens of llamahip_decode_greedy on that sequence alone.  The slots are stepped in
 * groups of up to 16 as sets (llamahip_stage_step_set: the weights are streamed once per step for a whole group); on a pipeline handle
 * (n_devices > 1) the groups -- at least one per stage -- are additionally pipelined over the stages: in steady state every stage (GPU) works on a
 * different group, rows and picks move between stages as stream-ordered copies.  The native form of the schedule bench.py --gpus N runs over RCCL. */
int llamahip_decode_greedy_multi(llamahip_model *m, int32_t n_threads, int32_t n_seqs, const int32_t *n_past, const int32_t *first_tokens,
                                 int32_t n_steps, int32_t *out_tokens, char *err, size_t err_cap);

/* llamahip_eval + every token's logits (n_tokens * n_vocab) and, for dump_layer >= 0, that layer's
 * 17 intermediates in the order documented in DESIGN.md ("debug dump order").  Parity tooling. */
int llamahip_eval_debug(llamahip_model *m, int32_t n_threads, int32_t n_past,
                        const int32_t *tokens, int32_t n_tokens, float *logits_last, float *logits_all,
                        int32_t dump_layer, float *dump, int64_t dump_cap, int64_t *dump_sizes,
                        char *err, size_t err_cap);

/* Pipeline-stage evaluation for layer-sharded models (handle loaded with layer_begin/layer_end).
 * hidden_in / hidden_out are DEVICE pointers to n_tokens * n_embd fp32 (the residual stream that
 * crosses layers, .mm:563-564, 687-690).  The first stage ignores hidden_in and embeds `tokens`;
 * the last stage also writes logits (host pointer, may be NULL) . */
int llamahip_eval_stage(llamahip_model *m, int32_t n_threads, int32_t n_past,
                        const int32_t *tokens, int32_t n_tokens,
                        const void *hidden_in, void *hidden_out, float *logits_out,
                        char *err, size_t err_cap);

/* Asynchronous, stream-ordered single-token stage steps: the decode loop of the layer pipeline
 *  with no host round trip per token.
 *
 * llamahip_stage_bind fixes, for sequence slot `seq`, the context position of the next token and the
 * caller-owned DEVICE buffers the step reads and writes:
 *   token_in   int32[1]        the token to evaluate            (first stage; NULL elsewhere)
 *   hidden_in  fp32[n_embd]    residual stream from the previous stage (NULL on the first stage)
 *   hidden_out fp32[n_embd]    residual stream for the next stage      (NULL on the last stage)
 *   token_out  int32[1]        greedy pick (argmax, lowest index on ties) of the last stage; may be
 *                              NULL, and may alias token_in on a whole-model handle
 * llamahip_stage_step enqueues one token step for that slot on `stream` (a hipStream_t; NULL = the
 * null stream) and returns without waiting: the caller orders its receives before and its
 * sends after the step on the same stream.  The position advances on the device after every step
 * (the KV cache must already hold positions [0, n_past): llamahip_eval_stage with the same slot
 * selected fills it).  Stepping past n_ctx is refused.
 * llamahip_stage_trace waits for the device, returns the number of steps taken since the bind,
 * stores the current position in *n_past and (last stage) the picked tokens in tokens[0..min(cap,n)). */
int llamahip_stage_bind(llamahip_model *m, int32_t seq, int32_t n_past,
                        void *token_in, const void *hidden_in, void *hidden_out, void *token_out,
                        char *err, size_t err_cap);
int llamahip_stage_step(llamahip_model *m, int32_t seq, int32_t n_threads, void *stream,
                        char *err, size_t err_cap);
int llamahip_stage_trace(llamahip_model *m, int32_t seq, int32_t *n_past, int32_t *tokens, int32_t cap,
                         char *err, size_t err_cap);

/* One decode step for a SET of bound slots at once: the result is bit for bit that of n_seqs llamahip_stage_step calls, one per slot
 * (every operator of llama_eval's graph works row by row, .mm:563-705; each row keeps its own position, KV cache and V*P key split of
 * its own single-token eval, ggml.c:5459-5480) -- but every weight matrix is streamed ONCE per step for all the sequences, which is
 * what a decode step costs ("throughput scales only with independent sequences in flight").  2 .. 16 distinct
 * slots, each bound with plain buffers (no mailbox); one graph is captured per (set of slots, n_threads), at most 32 are kept (least
 * recently used first out).  n_seqs = 1 is llamahip_stage_step.
 * (Norm statistics: a set step runs the reference's two-pass form (ggml.c:5327-5385), a single step takes them one-pass from its
 * producer's partial sums -- both narrow to the same fp32 bits except with probability ~2^-29 per value; the equality
 * of set steps and single steps is therefore tested (tests/test_pipeline.py), not structural.)
 * The score launch of a set covers the key slices up to the set's highest position rounded up to 128 (the host tracks every slot's position;
 * a captured step is keyed by that bucket too and re-captured when a row crosses into the next one). */
int llamahip_stage_step_set(llamahip_model *m, const int32_t *seqs, int32_t n_seqs, int32_t n_threads, void *stream,
                            char *err, size_t err_cap);
/* 1 if llamahip_stage_step_set can step n_seqs slots of this handle with this n_threads as ONE set (Q4_0 handle with layers, head size a
 * multiple of 32, n_threads <= 32); 0: step the slots one by one with llamahip_stage_step (up to 64 threads, every handle shape). */
int32_t llamahip_stage_set_applies(const llamahip_model *m, int32_t n_seqs, int32_t n_threads);

/* Device-side mailboxes between pipeline stages: instead of the caller moving hidden_out -> hidden_in (and token_out -> token_in)
 * between stages with a collective per token, the LAST kernel of a stage step stores the residual-stream row (.mm:563-564, 687-690)
 * straight into the NEXT stage's inbox -- memory of the next stage's process / GPU, peer-mapped through HIP IPC (xGMI stores between
 * GPUs) -- as 8-byte {fp32 bits, tag} granules, and the FIRST kernel of the next stage's step polls them; the last stage's pick
 * kernel stores the token into the first stage's token inbox the same way.  The tag is the sequence position, which every stage
 * knows: no host call, no RCCL launch and no stream ordering between stages per token; every poll is bounded (a lost neighbour
 * surfaces as LLAMAHIP_ERR_PREDICT from the next llamahip_stage_trace).
 *   llamahip_stage_mailbox         creates (once) slot `seq`'s inboxes -- n_embd granules on every stage but the first, one token
 *                                  granule on the first stage of a multi-stage pipeline -- and returns their device pointers
 *                                  (same-process neighbours) and / or 64-byte hipIpcMemHandle_t's (other processes); outputs may be NULL.
 *   llamahip_stage_mailbox_connect gives the slot the NEXT stage's hidden inbox (every stage but the last) and / or the first stage's
 *                                  token inbox (last stage), each as an IPC handle or as a raw device pointer.
 * A slot with an inbox / a connected peer takes NULL for llamahip_stage_bind's hidden_in / hidden_out; on the first stage token_in
 * still carries the FIRST token (bind publishes it to the inbox), later tokens arrive through the mailbox. */
int llamahip_stage_mailbox(llamahip_model *m, int32_t seq, void **hidden_inbox, void **token_inbox,
                           void *hidden_handle64, void *token_handle64, char *err, size_t err_cap);
int llamahip_stage_mailbox_connect(llamahip_model *m, int32_t seq, const void *next_hidden_handle64, void *next_hidden_ptr,
                                   const void *token_handle64, void *token_ptr, char *err, size_t err_cap);

/* Row `row` of the logits the most recent step / eval left on the device (last stage; waits for the device): row 0 after
 * llamahip_stage_step, row i = the i-th slot of the set (in the caller's order) after llamahip_stage_step_set, row i = the i-th row of the
 * eval after llamahip_eval / llamahip_eval_stage.  Every entry point that rewrites the logits resets the row map; a caller that steps
 * a set's slots one by one instead (llamahip_stage_set_applies() == 0) finds the LAST stepped slot's logits in row 0.  Parity tooling. */
int llamahip_stage_logits(llamahip_model *m, int32_t row, float *logits_out, char *err, size_t err_cap);

/* Select which of the handle's n_seq KV caches subsequent evals read and write (default 0). */
int llamahip_set_seq(llamahip_model *m, int32_t seq, char *err, size_t err_cap);

/* Raw fp32 KV rows of layer il, positions [0, n_pos) copied to host (n_pos * n_embd floats each). */
int llamahip_kv_read(llamahip_model *m, int32_t il, int32_t n_pos, float *out_k, float *out_v,
                     char *err, size_t err_cap);

/* Copy a weight tensor's merged file-format bytes (Q4_0 blocks or fp32) back to the host:
 * loader / multi-part merge parity.  Returns the byte count, or -1 for an unknown name. */
int64_t llamahip_tensor_bytes(llamahip_model *m, const char *name, void *out, int64_t cap);

/* ---- the step before the path: f32 / f16 model file -> Q4_0 model file ------------------------
 * Replaces llama_model_quantize (Sources/cpp/quantize.cpp:32-286): same
 * container handling (every 2-D tensor named "*weight" is quantized, everything else is copied, the
 * header's f16 field becomes `itype`), the reference's offline quantizer (utils.cpp:431-485) run on the
 * device.  itype 2 = Q4_0, 3 = Q4_1.  Byte-identical output. */
int llamahip_quantize_file(const char *fname_inp, const char *fname_out, int32_t itype,
                           char *err, size_t err_cap);

/* ---- single-op entry points (parity tests and kernel benchmarks; host buffers in/out) ---------- */
/* y[n][m] = W . quantize_q4_0(x[n])  with W = M rows of K/32 Q4_0 blocks in file layout
 * (replaces ggml_compute_forward_mul_mat_q4_0_f32, ggml.c:5987-6285). */
int llamahip_op_mul_mat_q4_0(const void *w_q4_0, int32_t M, int32_t K, const float *x, int32_t N,
                             float *y, char *err, size_t err_cap);
/* the device half of llamahip_eval_topk on caller-supplied logits (n_vocab <= 32768, top_k <= 64) */
int llamahip_op_topk(const float *logits, int32_t n_vocab, const int32_t *last_n_tokens, int32_t n_last, double repeat_penalty,
                     int32_t top_k, double temp, double *cand_scores, int32_t *cand_ids, int32_t *exact, char *err, size_t err_cap);
/* runtime activation quantizer (ggml.c:456-523): x[k] -> k/32 blocks of 20 bytes */
int llamahip_op_quantize_row_q4_0(const float *x, int32_t k, void *y, char *err, size_t err_cap);

typedef struct llamahip_gemv_bench {
    int32_t M, K;            /* shape */
    int32_t iters;           /* timed launches */
    float   ms_total;        /* HIP-event time over the timed launches, on the launch stream */
    double  algo_bytes;      /* M*(K/32)*20 + (K/32)*20 + 4*M  per launch */
} llamahip_gemv_bench;
/* Times the decode GEMV kernel on the model's resident matrices:
 * which = 0 fused wq|wk|wv, 1 wo, 2 fused w1|w3, 3 w2, 4 output; layer = layer index, or -1 to cycle
 * over every layer so each launch streams different weights from HBM (`iters` = cycles; out->iters =
 * timed launches).  For `output` the matrix is evicted from the Infinity Cache between launches. */
int llamahip_bench_gemv(llamahip_model *m, int32_t which, int32_t layer, int32_t warmup, int32_t iters,
                        llamahip_gemv_bench *out, char *err, size_t err_cap);

/* In-kernel phase probe of the decode GEMVs: runs n_steps greedy decode steps with the probe armed;
 * one record of 8 uint64 per GEMV launch {s_memtime at entry, loads issued, prologue done, weights
 * consumed, exit; ngroups; nchunks; PRE*16+EPI}.  Returns the record count.  Measurement tooling only. */
int64_t llamahip_debug_decode_phases(llamahip_model *m, int32_t n_past, int32_t first_token, int32_t n_steps,
                                     uint64_t *records, int64_t cap, char *err, size_t err_cap);

/* Launch counts of the multi-row mat-mul kernel families since process start, in the order
 * {matrix-core (k_gemm_mfma4), row-per-lane (k_gemm_rows), LDS-staged (k_gemm_lds), mat-vec (k_gemv), few rows (k_gemv_set)}:
 * lets a test assert that a shape took the path it is meant to.  Returns the number of families. */
int32_t llamahip_debug_gemm_paths(int64_t *out, int32_t cap);
/* Host-only (no device needed): how the few-row mat-mul would take n_rows activation rows against an m x k Q4_0 matrix (interleaved:
 * the w1|w3 layout; epi: 0 store, 1 +residual, 2 / 7 SiLU*up -> Q4_0 in whole- / half-block workgroups, 3 RoPE + KV append) --
 * out = {columns per wave, column-waves per row-group, column groups, row-groups per workgroup, LDS bytes}; 0 = the kernel does not
 * take that shape (the caller's generic path runs).  Lets a CPU test walk every LLaMA shape and row count. */
int32_t llamahip_debug_set_plan(int32_t m, int32_t k, int32_t interleaved, int32_t n_rows, int32_t epi, int64_t out[5]);
/* in-kernel phase records of the few-row mat-mul (measurement builds only; 0 records in the product build) */
int64_t llamahip_debug_set_probe(uint64_t *records, int64_t cap, int32_t reset);

/* Bit 0 / bit 1: the decode kernels evaluate the reference's SiLU / exp fp16 tables with device arithmetic instead of
 * gathering them -- enabled only after a load-time check that all 65 536 entries are reproduced.  0 before any load. */
int32_t llamahip_debug_lut_math(void);

typedef struct llamahip_stats {
    int32_t struct_size;
    int64_t weight_bytes_device;   /* repacked Q4_0 bytes resident in HBM */
    int64_t kv_bytes_device;
    int64_t n_evals;
    double  t_load_ms;             /* the reference measures t_load_us/t_predict_us and drops them (.mm:778,845) */
    double  t_eval_ms_total;
    int32_t n_stages;              /* in-process layer pipeline: stages of this handle (1: a plain handle) */
    int32_t hand_off;              /* ... 1 once llamahip_decode_greedy has run on it: the row and the token move between its stages as stream-ordered copies */
} llamahip_stats;
int llamahip_get_stats(const llamahip_model *m, llamahip_stats *out);

const char *llamahip_version(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* LLAMAHIP_H */
