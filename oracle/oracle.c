/* oracle/oracle.c -- TEST INFRASTRUCTURE ONLY (parity checker), never linked into the product.
 *
 * Standalone CPU restatement of the quantized-LLaMA hot path of alexrozanski/llama.swift, written
 * from the reference's behaviour (file:line cited per function, paths relative to /root/reference).
 * Numerics follow the x86 AVX2+FMA+F16C build of Sources/cpp/ggml.c compiled with the flags of
 * tools/Makefile (-O3 -std=c11 => no implicit FP contraction; explicit FMA only where the reference
 * uses _mm256_fmadd_ps).  This file is compiled with -ffp-contract=off for the same reason.
 *
 * Parity status: PINNED.  Every kernel here is checked bit-for-bit against the reference's own
 * ggml.c compiled in place (oracle/_ref/libggml_ref.so, tests/test_oracle_vs_ref.py) and against
 * committed golden vectors generated from that build (tests/golden/).
 *
 * Conventions: a Q4_0 block is 20 bytes {float d; uint8 qs[16]}, qs[j] = q[2j] | q[2j+1] << 4,
 * value = (q - 8) * d  (ggml.c:2026-2046, utils.cpp:447-480).
 */
#define _GNU_SOURCE
#include "oracle.h"

#include <immintrin.h>
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define QK 32
#define BLK 20

/* ------------------------------------------------------------------------------------------- */
/* fp16 conversion + lookup tables (ggml.c:159-162, 248-257, 2376-2389)                        */
/* ------------------------------------------------------------------------------------------- */
static uint16_t T_silu[1 << 16];
static uint16_t T_exp[1 << 16];
static float    T_f32[1 << 16];
static int      tables_ready = 0;

uint16_t orc_f32_to_f16(float f) { return (uint16_t) _cvtss_sh(f, 0); }   /* round-to-nearest-even */
float    orc_f16_to_f32(uint16_t h) { return _cvtsh_ss(h); }

void orc_tables_init(void) {
    if (tables_ready) return;
    for (int i = 0; i < (1 << 16); i++) {
        const float f = T_f32[i] = orc_f16_to_f32((uint16_t) i);
        /* ggml_silu_f32 (ggml.c:1944-1946): x/(1.0 + exp(-x)) evaluated in double, returned as float */
        const float s = (float) ((double) f / (1.0 + exp((double) -f)));
        T_silu[i] = orc_f32_to_f16(s);
        /* table_exp_f16 (ggml.c:2388): exp() in double, narrowed to float, then to fp16 */
        T_exp[i] = orc_f32_to_f16((float) exp((double) f));
    }
    tables_ready = 1;
}

void orc_tables_get(uint16_t *silu, uint16_t *expt) {
    orc_tables_init();
    memcpy(silu, T_silu, sizeof(T_silu));
    memcpy(expt, T_exp, sizeof(T_exp));
}

/* ------------------------------------------------------------------------------------------- */
/* Q4_0 quantizers                                                                             */
/* ------------------------------------------------------------------------------------------- */

/* Runtime activation quantizer, AVX2 branch (ggml.c:456-523):
 *   d = amax/7.0f ; id = amax != 0 ? 7.0f/amax : 0 ; q = rint_RNE(x*id) + 8.                    */
void orc_quantize_row_q4_0(const float *x, uint8_t *y, int k) {
    const int nb = k / QK;
    for (int b = 0; b < nb; b++) {
        const float *xb = x + b * QK;
        float amax = 0.0f;
        for (int l = 0; l < QK; l++) {
            const float a = fabsf(xb[l]);
            if (a > amax) amax = a;
        }
        const float d = amax / 7.0f;
        const float id = (amax != 0.0f) ? 7.0f / amax : 0.0f;
        memcpy(y + b * BLK, &d, 4);
        uint8_t *qs = y + b * BLK + 4;
        for (int j = 0; j < QK / 2; j++) {
            /* _mm256_round_ps(NEAREST) then _mm256_cvtps_epi32: both RNE; nearbyintf under the
             * default rounding mode is the same function */
            const int q0 = (int) nearbyintf(xb[2 * j + 0] * id) + 8;
            const int q1 = (int) nearbyintf(xb[2 * j + 1] * id) + 8;
            qs[j] = (uint8_t) (q0 | (q1 << 4));
        }
    }
}

/* ggml.c:651-684 */
void orc_dequantize_row_q4_0(const uint8_t *x, float *y, int k) {
    const int nb = k / QK;
    for (int b = 0; b < nb; b++) {
        float d;
        memcpy(&d, x + b * BLK, 4);
        const uint8_t *qs = x + b * BLK + 4;
        for (int j = 0; j < QK / 2; j++) {
            y[b * QK + 2 * j + 0] = (float) ((int) (qs[j] & 0xF) - 8) * d;
            y[b * QK + 2 * j + 1] = (float) ((int) (qs[j] >> 4) - 8) * d;
        }
    }
}

/* Offline (model-file) quantizer, utils.cpp:431-485: id = 1.0f/d, C round() (half away from 0). */
void orc_quantize_q4_0_offline(const float *src, uint8_t *dst, long n, int k) {
    const int nb = k / QK;
    const long rows = n / k;
    for (long r = 0; r < rows; r++) {
        for (int b = 0; b < nb; b++) {
            const float *xb = src + r * k + b * QK;
            uint8_t *out = dst + (r * nb + b) * BLK;
            float amax = 0.0f;
            for (int l = 0; l < QK; l++) amax = fmaxf(amax, fabsf(xb[l]));
            const float d = amax / 7.0f;
            const float id = d ? 1.0f / d : 0.0f;
            memcpy(out, &d, 4);
            for (int j = 0; j < QK / 2; j++) {
                const float v0 = xb[2 * j + 0] * id, v1 = xb[2 * j + 1] * id;
                const uint8_t q0 = (uint8_t) ((int8_t) round((double) v0) + 8);
                const uint8_t q1 = (uint8_t) ((int8_t) round((double) v1) + 8);
                out[4 + j] = (uint8_t) (q0 | (q1 << 4));
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------- */
/* dot products                                                                                */
/* ------------------------------------------------------------------------------------------- */

/* ggml_vec_dot_q4_0, AVX2 branch (ggml.c:1415-1466), written as scalar C that reproduces the
 * vector lanes: lane k (0..7) owns elements {2k, 2k+1, 16+2k, 17+2k} of every block
 * (_mm256_madd_epi16 pairs of the low then high 16 bytes), accumulates
 * acc[k] = fma(d_w*d_a, (float) isum_k, acc[k]) block after block, and the 8 lanes are folded as
 * ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)).                                                        */
float orc_vec_dot_q4_0_scalar(int n, const uint8_t *x, const uint8_t *y) {
    const int nb = n / QK;
    float acc[8] = { 0 };
    for (int b = 0; b < nb; b++) {
        float dx, dy;
        memcpy(&dx, x + b * BLK, 4);
        memcpy(&dy, y + b * BLK, 4);
        const float scale = dx * dy;
        const uint8_t *px = x + b * BLK + 4, *py = y + b * BLK + 4;
        for (int k = 0; k < 8; k++) {
            const int x0 = (px[k] & 0xF) - 8, x1 = (px[k] >> 4) - 8;
            const int x2 = (px[8 + k] & 0xF) - 8, x3 = (px[8 + k] >> 4) - 8;
            const int y0 = (py[k] & 0xF) - 8, y1 = (py[k] >> 4) - 8;
            const int y2 = (py[8 + k] & 0xF) - 8, y3 = (py[8 + k] >> 4) - 8;
            const int isum = x0 * y0 + x1 * y1 + x2 * y2 + x3 * y3;
            acc[k] = fmaf(scale, (float) isum, acc[k]);
        }
    }
    const float r0 = acc[0] + acc[4], r1 = acc[1] + acc[5], r2 = acc[2] + acc[6], r3 = acc[3] + acc[7];
    return (r0 + r2) + (r1 + r3);
}

/* Same arithmetic with AVX2 intrinsics (own formulation: maddubs on |x| / sign-adjusted y, then
 * fold the two 128-bit halves) so full-size models evaluate in reasonable time.  Produces the same
 * eight int32 lane sums, hence the same floats, as the scalar version above.                      */
float orc_vec_dot_q4_0(int n, const uint8_t *x, const uint8_t *y) {
    const int nb = n / QK;
    const __m128i m4 = _mm_set1_epi8(0x0F);
    const __m256i off = _mm256_set1_epi8(8);
    __m256 acc = _mm256_setzero_ps();
    for (int b = 0; b < nb; b++) {
        float dx, dy;
        memcpy(&dx, x + b * BLK, 4);
        memcpy(&dy, y + b * BLK, 4);
        const __m256 scale = _mm256_set1_ps(dx * dy);
        const __m128i vx = _mm_loadu_si128((const __m128i *) (x + b * BLK + 4));
        const __m128i vy = _mm_loadu_si128((const __m128i *) (y + b * BLK + 4));
        const __m128i xl = _mm_and_si128(vx, m4), xh = _mm_and_si128(_mm_srli_epi16(vx, 4), m4);
        const __m128i yl = _mm_and_si128(vy, m4), yh = _mm_and_si128(_mm_srli_epi16(vy, 4), m4);
        /* element order 0..31 */
        __m256i ex = _mm256_set_m128i(_mm_unpackhi_epi8(xl, xh), _mm_unpacklo_epi8(xl, xh));
        __m256i ey = _mm256_set_m128i(_mm_unpackhi_epi8(yl, yh), _mm_unpacklo_epi8(yl, yh));
        ex = _mm256_sub_epi8(ex, off);
        ey = _mm256_sub_epi8(ey, off);
        /* int16 lane j = e[2j]*f[2j] + e[2j+1]*f[2j+1], j = 0..15 */
        const __m256i p16 = _mm256_maddubs_epi16(_mm256_abs_epi8(ex), _mm256_sign_epi8(ey, ex));
        /* int32 lane k = lane16[k] + lane16[k+8] */
        const __m128i s16 = _mm_add_epi16(_mm256_castsi256_si128(p16), _mm256_extracti128_si256(p16, 1));
        const __m256 p = _mm256_cvtepi32_ps(_mm256_cvtepi16_epi32(s16));
        acc = _mm256_fmadd_ps(scale, p, acc);
    }
    float a[8];
    _mm256_storeu_ps(a, acc);
    const float r0 = a[0] + a[4], r1 = a[1] + a[5], r2 = a[2] + a[6], r3 = a[3] + a[7];
    return (r0 + r2) + (r1 + r3);
}

/* ggml_vec_dot_f32 with the AVX macro layer (ggml.c:1223-1258; GGML_F32_STEP 32, 4 accumulator
 * vectors of 8 lanes, FMA; reduce ggml.c:872-887).  Leftovers (n % 32) accumulate in double.    */
float orc_vec_dot_f32(int n, const float *x, const float *y) {
    const int np = n & ~31;
    float s[4][8];
    memset(s, 0, sizeof(s));
    for (int i = 0; i < np; i += 32)
        for (int j = 0; j < 4; j++)
            for (int l = 0; l < 8; l++)
                s[j][l] = fmaf(x[i + 8 * j + l], y[i + 8 * j + l], s[j][l]);
    float u[8];
    for (int l = 0; l < 8; l++) u[l] = (s[0][l] + s[1][l]) + (s[2][l] + s[3][l]);
    float t0[4];
    for (int l = 0; l < 4; l++) t0[l] = u[l] + u[l + 4];
    double sumf = (double) ((t0[0] + t0[1]) + (t0[2] + t0[3]));
    for (int i = np; i < n; i++) sumf += (double) (x[i] * y[i]);
    return (float) sumf;
}

/* y[n][m] = W[m,:] . quantize(x[n,:])   (ggml.c:5987-6285: INIT quantizes every src1 row
 * :6134-6152, COMPUTE is row-parallel :6182-6222; results do not depend on the thread count)    */
void orc_mul_mat_q4_0(const uint8_t *w, const float *x, float *y, int M, int K, int N, int n_threads) {
    const size_t rb = (size_t) (K / QK) * BLK;
    uint8_t *qa = (uint8_t *) malloc(rb * (size_t) N);
    for (int n = 0; n < N; n++) orc_quantize_row_q4_0(x + (size_t) n * K, qa + rb * n, K);
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for num_threads(n_threads) schedule(static)
    for (int m = 0; m < M; m++)
        for (int n = 0; n < N; n++)
            y[(size_t) n * M + m] = orc_vec_dot_q4_0(K, w + rb * m, qa + rb * n);
    free(qa);
}

/* ------------------------------------------------------------------------------------------- */
/* row ops                                                                                     */
/* ------------------------------------------------------------------------------------------- */

/* ggml_compute_forward_norm_f32 (ggml.c:5327-5385): mean-centred, no affine, eps 1e-5f widened
 * to double, sums in double in index order, scale narrowed to float before the multiply.         */
void orc_norm_rows(const float *x, float *y, int ncols, int nrows) {
    const double eps = (double) 1e-5f;
    for (int r = 0; r < nrows; r++) {
        const float *xr = x + (size_t) r * ncols;
        float *yr = y + (size_t) r * ncols;
        double mean = 0.0;
        for (int i = 0; i < ncols; i++) mean += (double) xr[i];
        mean /= ncols;
        double sum2 = 0.0;
        for (int i = 0; i < ncols; i++) {
            const double v = (double) xr[i] - mean;
            yr[i] = (float) v;
            sum2 += v * v;
        }
        const float scale = (float) (1.0 / sqrt(sum2 / ncols + eps));
        for (int i = 0; i < ncols; i++) yr[i] *= scale;
    }
}

/* ggml_vec_silu_f32 with GGML_SILU_FP16 (ggml.c:86, 1956-1963) */
void orc_silu_rows(const float *x, float *y, int ncols, int nrows) {
    orc_tables_init();
    const size_t n = (size_t) ncols * nrows;
    for (size_t i = 0; i < n; i++) y[i] = T_f32[T_silu[orc_f32_to_f16(x[i])]];
}

/* ggml_compute_forward_soft_max_f32 (ggml.c:6982-7050) */
void orc_softmax_rows(const float *x, float *y, int ncols, int nrows) {
    orc_tables_init();
    for (int r = 0; r < nrows; r++) {
        const float *xr = x + (size_t) r * ncols;
        float *p = y + (size_t) r * ncols;
        double mx = -INFINITY;
        for (int i = 0; i < ncols; i++) mx = mx > (double) xr[i] ? mx : (double) xr[i];
        const float max = (float) mx;
        double sum = 0.0;
        for (int i = 0; i < ncols; i++) {
            if (xr[i] == -INFINITY) {
                p[i] = 0.0f;
            } else {
                const float val = T_f32[T_exp[orc_f32_to_f16(xr[i] - max)]];
                sum += (double) val;
                p[i] = val;
            }
        }
        const float inv = (float) (1.0 / sum);
        for (int i = 0; i < ncols; i++) p[i] *= inv;
    }
}

/* ggml_compute_forward_rope_f32 (ggml.c:7076-7131).  x is [n][H][dh]; mode 0 rotates every row
 * with position n_past + row, mode 1 rotates rows >= n_past with position = row.                */
void orc_rope(float *x, int dh, int H, int n, int n_past, int mode) {
    for (int i2 = (mode == 0 ? 0 : n_past); i2 < n; i2++) {
        const int p = (mode == 0 ? n_past + i2 : i2);
        for (int i1 = 0; i1 < H; i1++) {
            for (int i0 = 0; i0 < dh; i0 += 2) {
                const double theta = pow(10000.0, ((double) -i0) / dh);
                double sn, cs;
                sincos(p * theta, &sn, &cs);      /* the -O3 reference build calls sincos() */
                float *v = x + ((size_t) i2 * H + i1) * dh + i0;
                const double x0 = v[0], x1 = v[1];
                v[0] = (float) (x0 * cs - x1 * sn);
                v[1] = (float) (x0 * sn + x1 * cs);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------- */
/* model file (LlamaPredictOperation.mm:98-498)                                                */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    char name[96];
    int n_dims;
    int ne0, ne1;        /* ne0 = input dim (columns), ne1 = rows */
    int is_q4;           /* 1: Q4_0 blocks, 0: f32 */
    size_t nbytes;
    uint8_t *data;
} orc_tensor;

typedef struct {
    orc_tensor *attention_norm, *wq, *wk, *wv, *wo, *ffn_norm, *w1, *w2, *w3;
} orc_layer;

struct orc_model {
    int n_vocab, n_ctx, n_embd, n_mult, n_head, n_layer, n_rot, f16, n_ff, n_parts;
    int n_tensors;
    orc_tensor *tensors;
    orc_tensor *tok_embeddings, *norm, *output;
    orc_layer *layers;
    float *memory_k, *memory_v;    /* [n_layer][n_ctx][n_embd] fp32 each (.mm:290-304) */
};

static orc_tensor *add_tensor(orc_model *m, const char *name, int n_dims, int ne0, int ne1, int is_q4) {
    orc_tensor *t = &m->tensors[m->n_tensors++];
    snprintf(t->name, sizeof(t->name), "%s", name);
    t->n_dims = n_dims; t->ne0 = ne0; t->ne1 = ne1; t->is_q4 = is_q4;
    t->nbytes = is_q4 ? (size_t) ne1 * (ne0 / QK) * BLK : (size_t) ne0 * ne1 * 4;
    t->data = (uint8_t *) calloc(1, t->nbytes);
    return t;
}

static orc_tensor *find_tensor(const orc_model *m, const char *name) {
    for (int i = 0; i < m->n_tensors; i++)
        if (strcmp(m->tensors[i].name, name) == 0) return &m->tensors[i];
    return NULL;
}

static int split_type_of(const char *name) {          /* .mm:358-388 */
    if (strstr(name, "tok_embeddings")) return 0;
    if (strstr(name, "layers")) {
        if (strstr(name, "attention.wo.weight")) return 0;
        if (strstr(name, "feed_forward.w2.weight")) return 0;
        return 1;
    }
    if (strstr(name, "output")) return 1;
    return 0;
}

static int parts_for_width(int d) {                   /* .mm:33-38 (unknown widths: 1) */
    return d == 5120 ? 2 : d == 6656 ? 4 : d == 8192 ? 8 : 1;
}

#define FAIL(...) do { if (err && err_cap) snprintf(err, err_cap, __VA_ARGS__); if (f) fclose(f); orc_free(m); return NULL; } while (0)

orc_model *orc_load(const char *path, int n_ctx, int force_parts, char *err, size_t err_cap) {
    orc_model *m = NULL;
    FILE *f = fopen(path, "rb");
    if (!f) FAIL("failed to open '%s'", path);
    uint32_t magic = 0;
    if (fread(&magic, 4, 1, f) != 1 || magic != 0x67676d6c) FAIL("invalid model file '%s' (bad magic)", path);
    m = (orc_model *) calloc(1, sizeof(*m));
    int32_t hp[7];
    if (fread(hp, 4, 7, f) != 7) FAIL("invalid model file '%s' (truncated header)", path);
    m->n_vocab = hp[0]; m->n_embd = hp[1]; m->n_mult = hp[2]; m->n_head = hp[3];
    m->n_layer = hp[4]; m->n_rot = hp[5]; m->f16 = hp[6]; m->n_ctx = n_ctx;
    m->n_ff = ((2 * (4 * m->n_embd) / 3 + m->n_mult - 1) / m->n_mult) * m->n_mult;
    m->n_parts = force_parts > 0 ? force_parts : parts_for_width(m->n_embd);
    if (m->f16 != 2) FAIL("oracle supports Q4_0 files only (f16 = %d)", m->f16);
    for (int i = 0; i < m->n_vocab; i++) {            /* vocab strings are skipped by the oracle */
        uint32_t len = 0;
        if (fread(&len, 4, 1, f) != 1) FAIL("invalid model file '%s' (truncated vocab)", path);
        fseek(f, len, SEEK_CUR);
    }
    const long tensors_at = ftell(f);
    fclose(f); f = NULL;

    const int d = m->n_embd, V = m->n_vocab, F = m->n_ff, L = m->n_layer;
    m->tensors = (orc_tensor *) calloc(3 + 9 * (size_t) L, sizeof(orc_tensor));
    m->layers = (orc_layer *) calloc(L, sizeof(orc_layer));
    m->tok_embeddings = add_tensor(m, "tok_embeddings.weight", 2, d, V, 1);
    m->norm = add_tensor(m, "norm.weight", 1, d, 1, 0);
    m->output = add_tensor(m, "output.weight", 2, d, V, 1);
    for (int i = 0; i < L; i++) {
        char nm[96];
        orc_layer *l = &m->layers[i];
#define T2(field, suffix, a, b) snprintf(nm, sizeof(nm), "layers.%d." suffix, i); l->field = add_tensor(m, nm, 2, a, b, 1)
#define T1(field, suffix) snprintf(nm, sizeof(nm), "layers.%d." suffix, i); l->field = add_tensor(m, nm, 1, d, 1, 0)
        T1(attention_norm, "attention_norm.weight");
        T2(wq, "attention.wq.weight", d, d); T2(wk, "attention.wk.weight", d, d);
        T2(wv, "attention.wv.weight", d, d); T2(wo, "attention.wo.weight", d, d);
        T1(ffn_norm, "ffn_norm.weight");
        T2(w1, "feed_forward.w1.weight", d, F); T2(w2, "feed_forward.w2.weight", F, d);
        T2(w3, "feed_forward.w3.weight", d, F);
#undef T1
#undef T2
    }
    m->memory_k = (float *) calloc((size_t) L * n_ctx * d, 4);
    m->memory_v = (float *) calloc((size_t) L * n_ctx * d, 4);

    for (int part = 0; part < m->n_parts; part++) {
        char fname[4096];
        if (part == 0) snprintf(fname, sizeof(fname), "%s", path);
        else snprintf(fname, sizeof(fname), "%s.%d", path, part);
        f = fopen(fname, "rb");
        if (!f) FAIL("failed to open '%s'", fname);
        fseek(f, tensors_at, SEEK_SET);
        for (;;) {
            int32_t hdr[3];
            if (fread(hdr, 4, 3, f) != 3) break;      /* EOF ends the tensor list (.mm:338-340) */
            const int n_dims = hdr[0], name_len = hdr[1], ftype = hdr[2];
            int32_t ne[2] = { 1, 1 };
            long nelements = 1;
            for (int i = 0; i < n_dims && i < 2; i++) { if (fread(&ne[i], 4, 1, f) != 1) FAIL("truncated tensor header"); nelements *= ne[i]; }
            char name[128] = { 0 };
            if (name_len >= (int) sizeof(name) || fread(name, 1, name_len, f) != (size_t) name_len) FAIL("bad tensor name");
            orc_tensor *t = find_tensor(m, name);
            if (!t) FAIL("unknown tensor '%s' in model file", name);
            const int np = (n_dims == 1) ? 1 : m->n_parts;
            const int split = split_type_of(name);
            if ((long) t->ne0 * t->ne1 / np != nelements) FAIL("tensor '%s' has wrong size in model file", name);
            int ok;
            if (n_dims == 1) ok = t->ne0 == ne[0] && t->ne1 == ne[1];
            else if (split == 0) ok = t->ne0 / np == ne[0] && t->ne1 == ne[1];
            else ok = t->ne0 == ne[0] && t->ne1 / np == ne[1];
            if (!ok) FAIL("tensor '%s' has wrong shape in model file", name);
            if (ftype != (t->is_q4 ? 2 : 0)) {
                if (ftype < 0 || ftype > 3) FAIL("unknown ftype %d in model file", ftype);
                FAIL("tensor '%s' has wrong size in model file", name);
            }
            const size_t row_bytes = t->is_q4 ? (size_t) (t->ne0 / QK) * BLK : (size_t) t->ne0 * 4;
            if (np == 1) {
                if (part == 0) { if (fread(t->data, 1, t->nbytes, f) != t->nbytes) FAIL("truncated tensor '%s'", name); }
                else fseek(f, (long) t->nbytes, SEEK_CUR);
            } else if (split == 0) {
                const size_t slice = row_bytes / np;
                const size_t at = ((size_t) part * ne[0] / QK) * BLK;
                for (int r = 0; r < ne[1]; r++)
                    if (fread(t->data + (size_t) r * row_bytes + at, 1, slice, f) != slice) FAIL("truncated tensor '%s'", name);
            } else {
                for (int r = 0; r < ne[1]; r++)
                    if (fread(t->data + ((size_t) r + (size_t) part * ne[1]) * row_bytes, 1, row_bytes, f) != row_bytes) FAIL("truncated tensor '%s'", name);
            }
        }
        fclose(f); f = NULL;
    }
    orc_tables_init();
    return m;
}

void orc_free(orc_model *m) {
    if (!m) return;
    for (int i = 0; i < m->n_tensors; i++) free(m->tensors[i].data);
    free(m->tensors); free(m->layers); free(m->memory_k); free(m->memory_v);
    free(m);
}

int orc_hparam(const orc_model *m, int which) {
    const int v[10] = { m->n_vocab, m->n_ctx, m->n_embd, m->n_mult, m->n_head, m->n_layer, m->n_rot, m->f16, m->n_ff, m->n_parts };
    return (which >= 0 && which < 10) ? v[which] : -1;
}

long orc_tensor_bytes(const orc_model *m, const char *name, void *out, long cap) {
    const orc_tensor *t = find_tensor(m, name);
    if (!t) return -1;
    if (out && cap >= (long) t->nbytes) memcpy(out, t->data, t->nbytes);
    return (long) t->nbytes;
}

void orc_kv(const orc_model *m, int il, int n_pos, float *out_k, float *out_v) {
    const size_t off = (size_t) il * m->n_ctx * m->n_embd;
    memcpy(out_k, m->memory_k + off, sizeof(float) * (size_t) n_pos * m->n_embd);
    memcpy(out_v, m->memory_v + off, sizeof(float) * (size_t) n_pos * m->n_embd);
}

/* ------------------------------------------------------------------------------------------- */
/* forward pass (LlamaPredictOperation.mm:510-735)                                             */
/* ------------------------------------------------------------------------------------------- */
/* OpenMP threads of the loops whose iterations are independent (rows of a mat-mul, (head, query) pairs of the attention): the
 * reference's n_threads only enters the ARITHMETIC through the V*P key split below, so the worker count of these loops is free.
 * ORC_OMP_THREADS overrides it (full-size tests: a 2048-token eval of the 32-layer model on all host cores). */
static int omp_workers(int nth) {
    static int ovr = -1;
    if (ovr < 0) { const char *e = getenv("ORC_OMP_THREADS"); ovr = e ? atoi(e) : 0; }
    return ovr > 0 ? ovr : nth;
}

static void matmul_q4(const orc_tensor *w, const uint8_t *qa, float *y, int N, int nth) {
    const int M = w->ne1, K = w->ne0;
    const size_t rb = (size_t) (K / QK) * BLK;
    nth = omp_workers(nth);
#pragma omp parallel for num_threads(nth) schedule(static)
    for (int mrow = 0; mrow < M; mrow++)
        for (int n = 0; n < N; n++)
            y[(size_t) n * M + mrow] = orc_vec_dot_q4_0(K, w->data + rb * mrow, qa + rb * n);
}

static void quantize_rows(const float *x, uint8_t *q, int K, int N) {
    const size_t rb = (size_t) (K / QK) * BLK;
    for (int n = 0; n < N; n++) orc_quantize_row_q4_0(x + (size_t) n * K, q + rb * n, K);
}

static void norm_mul(const float *x, const float *w, float *y, int d, int N) {
    orc_norm_rows(x, y, d, N);                         /* ggml_norm */
    for (int n = 0; n < N; n++)                        /* ggml_mul(ggml_repeat(w), cur) (ggml.c:4555) */
        for (int i = 0; i < d; i++) y[(size_t) n * d + i] = w[i] * y[(size_t) n * d + i];
}

#define DUMP(idx, ptr, count) do { if (dmp && dump && dump_sizes) { const long c_ = (long) (count); \
    if (dump_used + c_ <= dump_cap) { memcpy(dump + dump_used, (ptr), sizeof(float) * c_); dump_sizes[idx] = c_; dump_used += c_; } } } while (0)

static int g_split_chunk = 0;
void orc_set_split_chunk(int chunk) { g_split_chunk = chunk > 0 ? chunk : 0; }

/* layers [l0, l1): a pipeline stage.  The first stage embeds `tokens`, later stages start from
 * hidden_in (the fp32 residual stream, .mm:563-564, 687-690); the last stage applies the final norm
 * and lm head, earlier stages return the residual stream in hidden_out. */
static int eval_range(orc_model *m, int n_threads, int n_past, const int32_t *tokens, int N, int l0, int l1,
                      const float *hidden_in, float *hidden_out,
                      float *logits_last, float *logits_all,
                      int dump_layer, float *dump, long dump_cap, long *dump_sizes) {
    const int d = m->n_embd, C = m->n_ctx, H = m->n_head, V = m->n_vocab, F = m->n_ff;
    const int dh = d / H, T = n_past + N;
    const int nth = n_threads < 1 ? 1 : n_threads;
    if (T > C || N < 1) return -1001;
    long dump_used = 0;
    if (dump_sizes) for (int i = 0; i < 17; i++) dump_sizes[i] = 0;

    const size_t Nd = (size_t) N * d, NF = (size_t) N * F;
    float *x = (float *) malloc(Nd * 4), *cur = (float *) malloc(Nd * 4), *q = (float *) malloc(Nd * 4);
    float *k = (float *) malloc(Nd * 4), *v = (float *) malloc(Nd * 4), *ffin = (float *) malloc(Nd * 4);
    float *merged = (float *) malloc(Nd * 4), *up = (float *) malloc(NF * 4), *gate = (float *) malloc(NF * 4);
    float *kq = (float *) malloc((size_t) H * N * T * 4), *kqv = (float *) malloc(Nd * 4);
    float *part = (float *) malloc((size_t) nth * dh * 4);
    uint8_t *qa = (uint8_t *) malloc((size_t) N * ((F > d ? F : d) / QK) * BLK);
    const float kq_scale = 1.0f / sqrtf((float) d / H);         /* .mm:620 */

    if (l0 == 0) {
        /* ggml_get_rows on a Q4_0 matrix (ggml.c:6760-6785) */
        for (int n = 0; n < N; n++)
            orc_dequantize_row_q4_0(m->tok_embeddings->data + (size_t) tokens[n] * (d / QK) * BLK, x + (size_t) n * d, d);
    } else {
        memcpy(x, hidden_in, Nd * 4);
    }

    for (int il = l0; il < l1; il++) {
        const orc_layer *l = &m->layers[il];
        const int dmp = (il == dump_layer);
        float *Kc = m->memory_k + (size_t) il * C * d, *Vc = m->memory_v + (size_t) il * C * d;

        DUMP(0, x, Nd);
        norm_mul(x, (const float *) l->attention_norm->data, cur, d, N);          /* .mm:570-575 */
        DUMP(1, cur, Nd);
        quantize_rows(cur, qa, d, N);
        matmul_q4(l->wq, qa, q, N, nth); matmul_q4(l->wk, qa, k, N, nth); matmul_q4(l->wv, qa, v, N, nth);
        DUMP(2, q, Nd); DUMP(3, k, Nd); DUMP(4, v, Nd);

        memcpy(Kc + (size_t) n_past * d, k, Nd * 4);                              /* .mm:586-590 */
        memcpy(Vc + (size_t) n_past * d, v, Nd * 4);
        orc_rope(q, dh, H, N, n_past, 0);                                         /* .mm:594-601 */
        orc_rope(Kc, dh, H, T, n_past, 1);                                        /* .mm:604-611, in cache */
        DUMP(5, q, Nd);

        /* KQ[h][n][t] = K[t,h,:] . Q[n,h,:]  (.mm:614; ggml.c:5579-5618) then scale, mask, softmax */
#pragma omp parallel for num_threads(omp_workers(nth)) collapse(2) schedule(static)
        for (int h = 0; h < H; h++)
            for (int n = 0; n < N; n++) {
                float *row = kq + ((size_t) h * N + n) * T;
                for (int t = 0; t < T; t++) {
                    float s = orc_vec_dot_f32(dh, Kc + (size_t) t * d + h * dh, q + (size_t) n * d + h * dh);
                    s *= kq_scale;                                                /* .mm:617-621 */
                    if (t > n_past + n) s = -INFINITY;                            /* ggml.c:6946-6953 */
                    row[t] = s;
                }
                orc_softmax_rows(row, row, T, 1);                                 /* .mm:627 */
            }
        DUMP(6, kq, (size_t) H * N * T);

        /* KQV[h][n][c] = sum_t V[t,h,c] * P[h][n][t]  (.mm:638).  src0 is "transposed", so the
         * reference splits t into n_threads contiguous ranges, each accumulated with FMA into a
         * private zeroed buffer (ggml.c:5619-5665), then adds the buffers in thread order
         * (ggml.c:5553-5577).  The split therefore is part of the numerics.                     */
        {
            const int nwk = omp_workers(nth);
            float *part_all = (float *) malloc((size_t) nwk * nth * dh * 4);      /* one set of n_threads buffers per OpenMP worker */
#pragma omp parallel for num_threads(nwk) collapse(2) schedule(static)
            for (int h = 0; h < H; h++)
                for (int n = 0; n < N; n++) {
                    const float *P = kq + ((size_t) h * N + n) * T;
                    float *part = part_all + (size_t) omp_get_thread_num() * nth * dh;
                    memset(part, 0, (size_t) nth * dh * 4);
                    /* keys the reference splits for this row: n_past + N of the llama_eval call the row belongs to.  g_split_chunk > 0
                     * (orc_set_split_chunk, tests only) evaluates the rows as if they had arrived in successive calls of that many
                     * rows -- the claim behind llamahip_eval_chunks, checked against real successive calls by tests/test_oracle_chunks.py */
                    const int Ts = g_split_chunk > 0 ? n_past + ((n / g_split_chunk + 1) * g_split_chunk < N ? (n / g_split_chunk + 1) * g_split_chunk : N) : T;
                    const int dc = (Ts + nth - 1) / nth;
                    for (int th = 0; th < nth; th++) {
                        const int t0 = dc * th, t1 = (t0 + dc < Ts) ? t0 + dc : Ts;
                        float *acc = part + (size_t) th * dh;
                        for (int t = t0; t < t1; t++) {
                            const float *vr = Vc + (size_t) t * d + h * dh;
                            for (int c = 0; c < dh; c++) acc[c] = fmaf(vr[c], P[t], acc[c]);
                        }
                    }
                    float *out = kqv + ((size_t) h * N + n) * dh;
                    for (int c = 0; c < dh; c++) {
                        float s = part[c];
                        for (int th = 1; th < nth; th++) s += part[(size_t) th * dh + c];
                        out[c] = s;
                    }
                }
            free(part_all);
        }
        DUMP(7, kqv, Nd);
        for (int n = 0; n < N; n++)                                               /* .mm:641-646 */
            for (int h = 0; h < H; h++)
                memcpy(merged + (size_t) n * d + h * dh, kqv + ((size_t) h * N + n) * dh, (size_t) dh * 4);
        DUMP(8, merged, Nd);
        quantize_rows(merged, qa, d, N);
        matmul_q4(l->wo, qa, cur, N, nth);                                        /* .mm:649-651 */
        DUMP(9, cur, Nd);
        for (size_t i = 0; i < Nd; i++) ffin[i] = cur[i] + x[i];                  /* .mm:654 */
        DUMP(10, ffin, Nd);

        norm_mul(ffin, (const float *) l->ffn_norm->data, cur, d, N);             /* .mm:660-665 */
        DUMP(11, cur, Nd);
        quantize_rows(cur, qa, d, N);
        matmul_q4(l->w3, qa, up, N, nth);                                         /* .mm:668-670 */
        matmul_q4(l->w1, qa, gate, N, nth);                                       /* .mm:673-675 */
        DUMP(12, up, NF); DUMP(13, gate, NF);
        orc_silu_rows(gate, gate, F, N);                                          /* .mm:678 */
        for (size_t i = 0; i < NF; i++) gate[i] = gate[i] * up[i];                /* .mm:680 */
        DUMP(14, gate, NF);
        quantize_rows(gate, qa, F, N);
        matmul_q4(l->w2, qa, cur, N, nth);                                        /* .mm:682-684 */
        DUMP(15, cur, Nd);
        for (size_t i = 0; i < Nd; i++) x[i] = cur[i] + ffin[i];                  /* .mm:687 */
        DUMP(16, x, Nd);
    }

    float *logits = NULL;
    if (l1 == m->n_layer) {
        norm_mul(x, (const float *) m->norm->data, cur, d, N);                    /* .mm:695-700 */
        quantize_rows(cur, qa, d, N);
        logits = (float *) malloc((size_t) N * V * 4);
        matmul_q4(m->output, qa, logits, N, nth);                                 /* .mm:705 */
        if (logits_last) memcpy(logits_last, logits + (size_t) (N - 1) * V, (size_t) V * 4);   /* .mm:724-725 */
        if (logits_all) memcpy(logits_all, logits, (size_t) N * V * 4);
    } else if (hidden_out) {
        memcpy(hidden_out, x, Nd * 4);
    }

    free(logits); free(x); free(cur); free(q); free(k); free(v); free(ffin); free(merged);
    free(up); free(gate); free(kq); free(kqv); free(part); free(qa);
    return 0;
}

int orc_eval(orc_model *m, int n_threads, int n_past, const int32_t *tokens, int N,
             float *logits_last, float *logits_all,
             int dump_layer, float *dump, long dump_cap, long *dump_sizes) {
    return eval_range(m, n_threads, n_past, tokens, N, 0, m->n_layer, NULL, NULL, logits_last, logits_all,
                      dump_layer, dump, dump_cap, dump_sizes);
}

int orc_eval_range(orc_model *m, int n_threads, int n_past, const int32_t *tokens, int N, int l0, int l1,
                   const float *hidden_in, float *hidden_out, float *logits_last) {
    if (l0 < 0 || l1 > m->n_layer || l0 >= l1) return -1001;
    return eval_range(m, n_threads, n_past, tokens, N, l0, l1, hidden_in, hidden_out, logits_last, NULL, -1, NULL, 0, NULL);
}
