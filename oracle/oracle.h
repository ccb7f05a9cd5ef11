/* oracle/oracle.h -- TEST INFRASTRUCTURE ONLY (parity checker), never linked into the product.
 *
 * Standalone CPU restatement (no ggml) of the quantized-LLaMA hot path of alexrozanski/llama.swift
 * for the x86 AVX2+FMA+F16C build of the reference.  Pinned bit-for-bit against the reference's own
 * ggml.c compiled in place (oracle/_ref, tests/test_oracle_vs_ref.py) and against the committed
 * golden vectors generated from it (tests/golden).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_model orc_model;

/* --- kernels ------------------------------------------------------------------------------- */
void     orc_tables_init(void);                                      /* ggml.c:2376-2389 */
void     orc_tables_get(uint16_t *silu, uint16_t *expt);             /* 65536 entries each */
uint16_t orc_f32_to_f16(float f);                                    /* ggml.c:162 (F16C, RNE) */
float    orc_f16_to_f32(uint16_t h);                                 /* ggml.c:161 */
void     orc_quantize_row_q4_0(const float *x, uint8_t *y, int k);   /* ggml.c:456-523 (AVX2 branch) */
void     orc_dequantize_row_q4_0(const uint8_t *x, float *y, int k); /* ggml.c:651-684 */
void     orc_quantize_q4_0_offline(const float *src, uint8_t *dst, long n, int k); /* utils.cpp:431-485 */
float    orc_vec_dot_q4_0(int n, const uint8_t *x, const uint8_t *y);            /* ggml.c:1415-1466 */
float    orc_vec_dot_q4_0_scalar(int n, const uint8_t *x, const uint8_t *y);     /* same, lane-emulating scalar C */
float    orc_vec_dot_f32(int n, const float *x, const float *y);                 /* ggml.c:1223-1258, :872-887 */
void     orc_mul_mat_q4_0(const uint8_t *w, const float *x, float *y, int M, int K, int N, int n_threads); /* ggml.c:5987-6285 */
void     orc_norm_rows(const float *x, float *y, int ncols, int nrows);          /* ggml.c:5327-5385 */
void     orc_silu_rows(const float *x, float *y, int ncols, int nrows);          /* ggml.c:1956-1963 */
void     orc_softmax_rows(const float *x, float *y, int ncols, int nrows);       /* ggml.c:6982-7050 */
void     orc_rope(float *x, int dh, int H, int n, int n_past, int mode);         /* ggml.c:7076-7131 */

/* --- model --------------------------------------------------------------------------------- */
orc_model *orc_load(const char *path, int n_ctx, int force_parts, char *err, size_t err_cap); /* .mm:98-498 */
void       orc_free(orc_model *m);
int        orc_hparam(const orc_model *m, int which);  /* 0 n_vocab 1 n_ctx 2 n_embd 3 n_mult 4 n_head 5 n_layer 6 n_rot 7 f16 8 n_ff 9 n_parts */
long       orc_tensor_bytes(const orc_model *m, const char *name, void *out, long cap);
void       orc_kv(const orc_model *m, int il, int n_pos, float *out_k, float *out_v);
/* forward pass (.mm:510-735); n_threads reproduces the reference's thread-count-dependent
 * summation split in the V*P product (ggml.c:5619-5665, 5553-5577) and sets the OpenMP team size.
 * dump_layer >= 0 copies that layer's intermediates (17 tensors, same order as oracle/ref_driver.cpp). */
/* tests only: rows of one orc_eval call split their V*P keys as if they had arrived in successive calls of `chunk` rows (0: off) */
void       orc_set_split_chunk(int chunk);
int        orc_eval(orc_model *m, int n_threads, int n_past, const int32_t *tokens, int N,
                    float *logits_last, float *logits_all,
                    int dump_layer, float *dump, long dump_cap, long *dump_sizes);

/* layers [l0, l1) only -- emulates one stage of the layer pipeline */
int        orc_eval_range(orc_model *m, int n_threads, int n_past, const int32_t *tokens, int N, int l0, int l1,
                          const float *hidden_in, float *hidden_out, float *logits_last);

#ifdef __cplusplus
}
#endif
#endif
