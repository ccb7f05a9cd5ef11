// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Reference-backed oracle: a restatement of the reference's model loader and forward pass that
// links the reference's OWN ggml (compiled in place from /root/reference/Sources/cpp/ggml.c into
// oracle/_ref/, never copied into this repo).  Every arithmetic operation below therefore executes
// inside the reference's ggml_graph_compute, so the logits it returns *are* reference logits for
// the x86 AVX2+FMA+F16C build (the ISA variant SURVEY.md section 8c names as the parity target).
//
// What is restated here (the .mm file cannot be compiled without an Objective-C toolchain):
//   * file format + multi-part merge : Sources/llamaObjCxx/bridge/LlamaPredictOperation.mm:98-498
//   * forward graph                  : LlamaPredictOperation.mm:510-735
//   * n_parts table (relaxed: widths not in the table load as 1 part so tiny fixtures work)
//                                    : LlamaPredictOperation.mm:33-38
// Also exported: thin wrappers over the reference's non-static kernels (quantize_row_q4_0,
// dequantize_row_q4_0, ggml_quantize_q4_0, llama_tokenize, llama_sample_top_p_top_k) and single-op
// graphs (mul_mat, norm, soft_max, silu, rope) so tests can pin the standalone restatement
// (oracle/oracle.c) op by op.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.

#include "ggml.h"
#include "utils.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <random>
#include <string>
#include <vector>

extern "C" {
// non-static reference kernels (ggml.c:404, ggml.c:651)
void quantize_row_q4_0(const float * x, void * y, int k);
void dequantize_row_q4_0(const void * x, float * y, int k);
}

namespace {

struct RefLayer {
    ggml_tensor *attention_norm, *wq, *wk, *wv, *wo, *ffn_norm, *w1, *w2, *w3;
};

struct RefModel {
    int32_t n_vocab = 0, n_ctx = 0, n_embd = 0, n_mult = 0, n_head = 0, n_layer = 0, n_rot = 0, f16 = 0;
    int32_t n_ff = 0, n_parts = 1;
    ggml_context * ctx = nullptr;
    ggml_tensor *tok_embeddings = nullptr, *norm = nullptr, *output = nullptr;
    ggml_tensor *memory_k = nullptr, *memory_v = nullptr;
    std::vector<RefLayer> layers;
    std::map<std::string, ggml_tensor *> by_name;
    gpt_vocab vocab;
    // eval scratch (the reference keeps a process-wide static 512 MB buffer, .mm:532-533;
    // here it is per model so two oracles can coexist)
    std::vector<uint8_t> scratch;
    size_t mem_per_token = 0;
};

void set_err(char * err, size_t cap, const char * fmt, const char * a = "", long b = 0) {
    if (err && cap) snprintf(err, cap, fmt, a, b);
}

int parts_for_width(int n_embd) {
    // LlamaPredictOperation.mm:33-38; unknown widths -> 1 (relaxation for tiny fixtures)
    switch (n_embd) {
        case 4096: return 1;
        case 5120: return 2;
        case 6656: return 4;
        case 8192: return 8;
        default:   return 1;
    }
}

// split rule of LlamaPredictOperation.mm:358-388: 0 = shard along ne[0] (columns), 1 = along ne[1]
int split_type_of(const std::string & name) {
    if (name.find("tok_embeddings") != std::string::npos) return 0;
    if (name.find("layers") != std::string::npos) {
        if (name.find("attention.wo.weight") != std::string::npos) return 0;
        if (name.find("feed_forward.w2.weight") != std::string::npos) return 0;
        return 1;
    }
    if (name.find("output") != std::string::npos) return 1;
    return 0;
}

} // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------
// model load / free
// ---------------------------------------------------------------------------------------------
void * refllama_load(const char * path, int n_ctx, int force_parts, char * err, size_t err_cap) {
    std::ifstream fin(path, std::ios::binary);
    if (!fin) { set_err(err, err_cap, "failed to open '%s'", path); return nullptr; }

    uint32_t magic = 0;
    fin.read((char *) &magic, 4);
    if (magic != 0x67676d6c) { set_err(err, err_cap, "invalid model file '%s' (bad magic)", path); return nullptr; }

    RefModel * m = new RefModel();
    int32_t hp[7];
    fin.read((char *) hp, sizeof(hp));
    m->n_vocab = hp[0]; m->n_embd = hp[1]; m->n_mult = hp[2]; m->n_head = hp[3];
    m->n_layer = hp[4]; m->n_rot = hp[5]; m->f16 = hp[6];
    m->n_ctx = n_ctx;
    m->n_ff = ((2*(4*m->n_embd)/3 + m->n_mult - 1)/m->n_mult)*m->n_mult;   // .mm:135
    m->n_parts = force_parts > 0 ? force_parts : parts_for_width(m->n_embd);

    for (int i = 0; i < m->n_vocab; i++) {                                   // .mm:149-163
        uint32_t len = 0;
        fin.read((char *) &len, 4);
        std::string word(len, '\0');
        if (len) fin.read(&word[0], len);
        m->vocab.token_to_id[word] = i;
        m->vocab.id_to_token[i] = word;
    }

    ggml_type wtype;
    switch (m->f16) {                                                        // .mm:168-180
        case 0: wtype = GGML_TYPE_F32;  break;
        case 1: wtype = GGML_TYPE_F16;  break;
        case 2: wtype = GGML_TYPE_Q4_0; break;
        case 3: wtype = GGML_TYPE_Q4_1; break;
        default:
            set_err(err, err_cap, "invalid model file '%s' (bad f16 value %ld)", path, m->f16);
            delete m; return nullptr;
    }

    const int64_t d = m->n_embd, L = m->n_layer, V = m->n_vocab, F = m->n_ff, C = m->n_ctx;
    double bytes = 0;                                                        // .mm:186-219
    bytes += 2.0*d*V*ggml_type_sizef(wtype) + d*4.0;
    bytes += L*(2.0*d*4.0 + 4.0*d*d*ggml_type_sizef(wtype) + 3.0*F*d*ggml_type_sizef(wtype));
    bytes += 2.0*C*L*d*4.0;
    bytes += (5 + 10*L)*256 + 4096;

    ggml_init_params ip = { (size_t) bytes, nullptr };
    m->ctx = ggml_init(ip);
    if (!m->ctx) { set_err(err, err_cap, "ggml_init() failed"); delete m; return nullptr; }

    ggml_context * ctx = m->ctx;
    m->tok_embeddings = ggml_new_tensor_2d(ctx, wtype, d, V);
    m->norm           = ggml_new_tensor_1d(ctx, GGML_TYPE_F32, d);
    m->output         = ggml_new_tensor_2d(ctx, wtype, d, V);
    m->by_name["tok_embeddings.weight"] = m->tok_embeddings;
    m->by_name["norm.weight"]           = m->norm;
    m->by_name["output.weight"]         = m->output;
    m->layers.resize(L);
    for (int i = 0; i < L; i++) {
        RefLayer & l = m->layers[i];
        const std::string p = "layers." + std::to_string(i) + ".";
        m->by_name[p + "attention_norm.weight"]  = l.attention_norm = ggml_new_tensor_1d(ctx, GGML_TYPE_F32, d);
        m->by_name[p + "attention.wq.weight"]    = l.wq = ggml_new_tensor_2d(ctx, wtype, d, d);
        m->by_name[p + "attention.wk.weight"]    = l.wk = ggml_new_tensor_2d(ctx, wtype, d, d);
        m->by_name[p + "attention.wv.weight"]    = l.wv = ggml_new_tensor_2d(ctx, wtype, d, d);
        m->by_name[p + "attention.wo.weight"]    = l.wo = ggml_new_tensor_2d(ctx, wtype, d, d);
        m->by_name[p + "ffn_norm.weight"]        = l.ffn_norm = ggml_new_tensor_1d(ctx, GGML_TYPE_F32, d);
        m->by_name[p + "feed_forward.w1.weight"] = l.w1 = ggml_new_tensor_2d(ctx, wtype, d, F);
        m->by_name[p + "feed_forward.w2.weight"] = l.w2 = ggml_new_tensor_2d(ctx, wtype, F, d);
        m->by_name[p + "feed_forward.w3.weight"] = l.w3 = ggml_new_tensor_2d(ctx, wtype, d, F);
    }
    m->memory_k = ggml_new_tensor_1d(ctx, GGML_TYPE_F32, d*L*C);             // .mm:297-301
    m->memory_v = ggml_new_tensor_1d(ctx, GGML_TYPE_F32, d*L*C);
    // the reference leaves the cache uninitialised (malloc); zero it so runs are reproducible
    memset(m->memory_k->data, 0, ggml_nbytes(m->memory_k));
    memset(m->memory_v->data, 0, ggml_nbytes(m->memory_v));

    const std::streamoff tensors_at = fin.tellg();
    fin.close();

    for (int part = 0; part < m->n_parts; part++) {                          // .mm:312-495
        std::string fname = path;
        if (part > 0) fname += "." + std::to_string(part);
        std::ifstream fp(fname, std::ios::binary);
        if (!fp) { set_err(err, err_cap, "failed to open '%s'", fname.c_str()); ggml_free(m->ctx); delete m; return nullptr; }
        fp.seekg(tensors_at);

        for (;;) {
            int32_t n_dims, name_len, ftype;
            fp.read((char *) &n_dims, 4);
            fp.read((char *) &name_len, 4);
            fp.read((char *) &ftype, 4);
            if (fp.eof()) break;

            int32_t ne[2] = { 1, 1 };
            int64_t nelements = 1;
            for (int i = 0; i < n_dims; i++) { fp.read((char *) &ne[i], 4); nelements *= ne[i]; }
            std::string name(name_len, '\0');
            fp.read(&name[0], name_len);

            auto it = m->by_name.find(name);
            if (it == m->by_name.end()) {
                set_err(err, err_cap, "unknown tensor '%s' in model file", name.c_str());
                ggml_free(m->ctx); delete m; return nullptr;
            }
            ggml_tensor * t = it->second;
            const int split = split_type_of(name);
            const int np = (n_dims == 1) ? 1 : m->n_parts;

            if (ggml_nelements(t)/np != nelements) {
                set_err(err, err_cap, "tensor '%s' has wrong size in model file", name.c_str());
                ggml_free(m->ctx); delete m; return nullptr;
            }
            bool shape_ok;
            if (n_dims == 1)      shape_ok = t->ne[0] == ne[0] && t->ne[1] == ne[1];
            else if (split == 0)  shape_ok = t->ne[0]/np == ne[0] && t->ne[1] == ne[1];
            else                  shape_ok = t->ne[0] == ne[0] && t->ne[1]/np == ne[1];
            if (!shape_ok) {
                set_err(err, err_cap, "tensor '%s' has wrong shape in model file", name.c_str());
                ggml_free(m->ctx); delete m; return nullptr;
            }

            size_t bpe;
            switch (ftype) {                                                 // .mm:434-444
                case 0: bpe = ggml_type_size(GGML_TYPE_F32);  break;
                case 1: bpe = ggml_type_size(GGML_TYPE_F16);  break;
                case 2: bpe = ggml_type_size(GGML_TYPE_Q4_0); break;
                case 3: bpe = ggml_type_size(GGML_TYPE_Q4_1); break;
                default:
                    set_err(err, err_cap, "unknown ftype %ld in model file", "", ftype);
                    ggml_free(m->ctx); delete m; return nullptr;
            }
            if ((nelements*bpe)/ggml_blck_size(t->type) != ggml_nbytes(t)/np) {
                set_err(err, err_cap, "tensor '%s' has wrong size in model file", name.c_str());
                ggml_free(m->ctx); delete m; return nullptr;
            }

            const size_t row_bytes = (t->ne[0]/ggml_blck_size(t->type))*ggml_type_size(t->type);
            if (np == 1) {
                // 1-D tensors (and single-part files): part 0 supplies the data (.mm:446-459)
                if (part == 0) fp.read((char *) t->data, ggml_nbytes(t));
                else           fp.seekg(ggml_nbytes(t), std::ios::cur);
            } else if (split == 0) {
                // column shard: every row receives a contiguous slice (.mm:467-477)
                const size_t slice = row_bytes/np;
                const size_t at = ((size_t) part*ne[0]/ggml_blck_size(t->type))*ggml_type_size(t->type);
                for (int r = 0; r < ne[1]; r++) fp.read((char *) t->data + r*row_bytes + at, slice);
            } else {
                // row shard: rows [part*ne1, (part+1)*ne1) (.mm:478-487)
                for (int r = 0; r < ne[1]; r++) fp.read((char *) t->data + ((size_t) r + (size_t) part*ne[1])*row_bytes, row_bytes);
            }
        }
    }

    m->scratch.resize(512u*1024*1024);
    return m;
}

void refllama_free(void * h) {
    RefModel * m = (RefModel *) h;
    if (!m) return;
    if (m->ctx) ggml_free(m->ctx);
    delete m;
}

int refllama_hparam(void * h, int which) {
    RefModel * m = (RefModel *) h;
    switch (which) {
        case 0: return m->n_vocab; case 1: return m->n_ctx;  case 2: return m->n_embd;
        case 3: return m->n_mult;  case 4: return m->n_head; case 5: return m->n_layer;
        case 6: return m->n_rot;   case 7: return m->f16;    case 8: return m->n_ff;
        case 9: return m->n_parts;
    }
    return -1;
}

// copy one named weight tensor's raw bytes (merged across parts) -- loader parity tests
long refllama_tensor_bytes(void * h, const char * name, void * out, long cap) {
    RefModel * m = (RefModel *) h;
    auto it = m->by_name.find(name);
    if (it == m->by_name.end()) return -1;
    const long n = (long) ggml_nbytes(it->second);
    if (out && cap >= n) memcpy(out, it->second->data, n);
    return n;
}

// raw fp32 KV cache rows for layer il, positions [0, n_pos): out_k/out_v = n_pos*n_embd floats
void refllama_kv(void * h, int il, int n_pos, float * out_k, float * out_v) {
    RefModel * m = (RefModel *) h;
    const size_t off = (size_t) il*m->n_ctx*m->n_embd;
    memcpy(out_k, (float *) m->memory_k->data + off, sizeof(float)*(size_t) n_pos*m->n_embd);
    memcpy(out_v, (float *) m->memory_v->data + off, sizeof(float)*(size_t) n_pos*m->n_embd);
}

// ---------------------------------------------------------------------------------------------
// forward pass (LlamaPredictOperation.mm:510-735).
//   logits_last : n_vocab floats (what the reference returns, .mm:724-725)
//   logits_all  : optional, N*n_vocab floats
//   dump_layer  : >= 0 -> copy that layer's intermediates into dump (see DUMP_* order below);
//                 dump_sizes[i] receives element counts.  -1 = off.
// ---------------------------------------------------------------------------------------------
enum { DUMP_COUNT = 17 };

int refllama_eval(void * h, int n_threads, int n_past, const int32_t * tokens, int N,
                  float * logits_last, float * logits_all,
                  int dump_layer, float * dump, long dump_cap, long * dump_sizes,
                  char * err, size_t err_cap) {
    RefModel * m = (RefModel *) h;
    const int d = m->n_embd, L = m->n_layer, C = m->n_ctx, H = m->n_head, V = m->n_vocab;
    const int dh = d/H;
    const int n_rot = dh;                                                    // .mm:528

    if (m->mem_per_token > 0 && m->mem_per_token*N > m->scratch.size()) {    // .mm:535-547
        m->scratch.resize((size_t)(1.1*(m->mem_per_token*N)));
    }
    ggml_init_params ip = { m->scratch.size(), m->scratch.data() };
    ggml_context * c0 = ggml_init(ip);
    if (!c0) { set_err(err, err_cap, "ggml_init() failed"); return -1001; }
    ggml_cgraph gf = {};
    gf.n_threads = n_threads;

    ggml_tensor * embd = ggml_new_tensor_1d(c0, GGML_TYPE_I32, N);
    memcpy(embd->data, tokens, sizeof(int32_t)*N);
    ggml_tensor * x = ggml_get_rows(c0, m->tok_embeddings, embd);

    ggml_tensor * dumps[DUMP_COUNT] = { nullptr };
    const size_t fsz = sizeof(float);

    for (int il = 0; il < L; il++) {
        const RefLayer & l = m->layers[il];
        const bool dmp = (il == dump_layer);
        ggml_tensor * resid = x;

        ggml_tensor * cur = ggml_norm(c0, x);                                // .mm:570-575
        cur = ggml_mul(c0, ggml_repeat(c0, l.attention_norm, cur), cur);
        if (dmp) { dumps[0] = x; dumps[1] = cur; }

        ggml_tensor * Qc = ggml_mul_mat(c0, l.wq, cur);                      // .mm:580-582
        ggml_tensor * Kc = ggml_mul_mat(c0, l.wk, cur);
        ggml_tensor * Vc = ggml_mul_mat(c0, l.wv, cur);
        if (dmp) { dumps[2] = Qc; dumps[3] = Kc; dumps[4] = Vc; }

        {   // append K,V rows to the cache; scheduled before anything that reads the cache (.mm:585-591)
            const size_t at = fsz*d*((size_t) il*C + n_past);
            ggml_tensor * kdst = ggml_view_1d(c0, m->memory_k, (int64_t) N*d, at);
            ggml_tensor * vdst = ggml_view_1d(c0, m->memory_v, (int64_t) N*d, at);
            ggml_build_forward_expand(&gf, ggml_cpy(c0, Kc, kdst));
            ggml_build_forward_expand(&gf, ggml_cpy(c0, Vc, vdst));
        }

        const int T = n_past + N;
        const size_t layer_at = fsz*d*(size_t) il*C;

        ggml_tensor * Qr = ggml_rope(c0, ggml_cpy(c0, Qc, ggml_new_tensor_3d(c0, GGML_TYPE_F32, dh, H, N)),
                                     n_past, n_rot, 0);                      // .mm:594-601
        ggml_tensor * Q = ggml_permute(c0, Qr, 0, 2, 1, 3);
        ggml_tensor * Kr = ggml_rope(c0, ggml_reshape_3d(c0, ggml_view_1d(c0, m->memory_k, (int64_t) T*d, layer_at), dh, H, T),
                                     n_past, n_rot, 1);                      // .mm:604-611 (in-cache)
        ggml_tensor * K = ggml_permute(c0, Kr, 0, 2, 1, 3);
        if (dmp) dumps[5] = Qr;

        ggml_tensor * KQ = ggml_mul_mat(c0, K, Q);                           // .mm:614
        KQ = ggml_scale(c0, KQ, ggml_new_f32(c0, 1.0f/sqrt(float(d)/H)));    // .mm:617-621
        KQ = ggml_diag_mask_inf(c0, KQ, n_past);                             // .mm:624
        KQ = ggml_soft_max(c0, KQ);                                          // .mm:627
        if (dmp) dumps[6] = KQ;

        ggml_tensor * Vt = ggml_permute(c0, ggml_reshape_3d(c0, ggml_view_1d(c0, m->memory_v, (int64_t) T*d, layer_at), dh, H, T),
                                        1, 2, 0, 3);                         // .mm:630-635
        ggml_tensor * KQV = ggml_mul_mat(c0, Vt, KQ);                        // .mm:638
        if (dmp) dumps[7] = KQV;
        cur = ggml_cpy(c0, ggml_permute(c0, KQV, 0, 2, 1, 3), ggml_new_tensor_2d(c0, GGML_TYPE_F32, d, N));
        if (dmp) dumps[8] = cur;
        cur = ggml_mul_mat(c0, l.wo, cur);                                   // .mm:649-651
        if (dmp) dumps[9] = cur;

        ggml_tensor * ff_in = ggml_add(c0, cur, resid);                      // .mm:654
        if (dmp) dumps[10] = ff_in;

        cur = ggml_norm(c0, ff_in);                                          // .mm:660-665
        cur = ggml_mul(c0, ggml_repeat(c0, l.ffn_norm, cur), cur);
        if (dmp) dumps[11] = cur;
        ggml_tensor * up = ggml_mul_mat(c0, l.w3, cur);                      // .mm:668-670
        ggml_tensor * gate = ggml_mul_mat(c0, l.w1, cur);                    // .mm:673-675
        if (dmp) { dumps[12] = up; dumps[13] = gate; }
        cur = ggml_mul(c0, ggml_silu(c0, gate), up);                         // .mm:678-680
        if (dmp) dumps[14] = cur;
        cur = ggml_mul_mat(c0, l.w2, cur);                                   // .mm:682-684
        if (dmp) dumps[15] = cur;
        x = ggml_add(c0, cur, ff_in);                                        // .mm:687
        if (dmp) dumps[16] = x;
    }

    x = ggml_norm(c0, x);                                                    // .mm:695-700
    x = ggml_mul(c0, ggml_repeat(c0, m->norm, x), x);
    x = ggml_mul_mat(c0, m->output, x);                                      // .mm:705

    ggml_build_forward_expand(&gf, x);
    ggml_graph_compute(c0, &gf);

    if (logits_last) memcpy(logits_last, (float *) ggml_get_data(x) + (size_t) V*(N - 1), fsz*V);
    if (logits_all)  memcpy(logits_all, ggml_get_data(x), fsz*(size_t) V*N);

    if (dump_layer >= 0 && dump && dump_sizes) {
        long used = 0;
        for (int i = 0; i < DUMP_COUNT; i++) {
            dump_sizes[i] = 0;
            if (!dumps[i]) continue;
            const long n = (long) ggml_nelements(dumps[i]);
            if (used + n > dump_cap) break;
            memcpy(dump + used, dumps[i]->data, fsz*n);
            dump_sizes[i] = n;
            used += n;
        }
    }

    if (m->mem_per_token == 0) m->mem_per_token = ggml_used_mem(c0)/N;       // .mm:727-729
    ggml_free(c0);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// single-op entry points (each runs the reference kernel through a one-node ggml graph)
// ---------------------------------------------------------------------------------------------
static std::vector<uint8_t> & op_arena(size_t need) {
    static std::vector<uint8_t> buf;
    if (buf.size() < need) buf.resize(need);
    return buf;
}

void ref_init_tables(void) {   // forces ggml_init's first-call table construction (ggml.c:2376-2389)
    ggml_init_params ip = { 1024, nullptr };
    ggml_context * c = ggml_init(ip);
    ggml_free(c);
}

void ref_quantize_row_q4_0(const float * x, void * y, int k) { quantize_row_q4_0(x, y, k); }       // ggml.c:404 (AVX2 branch :456-523)
void ref_dequantize_row_q4_0(const void * x, float * y, int k) { dequantize_row_q4_0(x, y, k); }   // ggml.c:651
long ref_quantize_q4_0_offline(float * src, void * dst, int n, int k, int64_t * hist) {           // utils.cpp:431-485
    return (long) ggml_quantize_q4_0(src, dst, n, k, 32, hist);
}
float    ref_fp16_to_fp32(uint16_t hbits) { return ggml_fp16_to_fp32(hbits); }                     // ggml.c:276
uint16_t ref_fp32_to_fp16(float f) { return ggml_fp32_to_fp16(f); }                                // ggml.c:280

// y[N][M] = W(Q4_0, M rows of K) x X(f32, N rows of K)   -- ggml.c:5987-6285
void ref_mul_mat_q4_0(const void * w, const float * x, float * y, int M, int K, int N, int n_threads) {
    const size_t wbytes = (size_t) M*(K/32)*20;
    auto & buf = op_arena(wbytes + (size_t) N*K*4 + (size_t) N*M*4 + (size_t) N*K + (1u << 20));
    ggml_init_params ip = { buf.size(), buf.data() };
    ggml_context * c = ggml_init(ip);
    ggml_tensor * tw = ggml_new_tensor_2d(c, GGML_TYPE_Q4_0, K, M);
    ggml_tensor * tx = ggml_new_tensor_2d(c, GGML_TYPE_F32, K, N);
    memcpy(tw->data, w, wbytes);
    memcpy(tx->data, x, sizeof(float)*(size_t) N*K);
    ggml_tensor * ty = ggml_mul_mat(c, tw, tx);
    ggml_cgraph gf = {};
    gf.n_threads = n_threads;
    ggml_build_forward_expand(&gf, ty);
    ggml_graph_compute(c, &gf);
    memcpy(y, ty->data, sizeof(float)*(size_t) N*M);
    ggml_free(c);
}

// op: 0 = norm (ggml.c:5327), 1 = silu (ggml.c:5261), 2 = soft_max over rows of length ncols (ggml.c:6982)
void ref_unary_rows(int op, const float * x, float * y, int ncols, int nrows, int n_threads) {
    auto & buf = op_arena((size_t) ncols*nrows*8 + (1u << 20));
    ggml_init_params ip = { buf.size(), buf.data() };
    ggml_context * c = ggml_init(ip);
    ggml_tensor * tx = ggml_new_tensor_2d(c, GGML_TYPE_F32, ncols, nrows);
    memcpy(tx->data, x, sizeof(float)*(size_t) ncols*nrows);
    ggml_tensor * ty = op == 0 ? ggml_norm(c, tx) : op == 1 ? ggml_silu(c, tx) : ggml_soft_max(c, tx);
    ggml_cgraph gf = {};
    gf.n_threads = n_threads;
    ggml_build_forward_expand(&gf, ty);
    ggml_graph_compute(c, &gf);
    memcpy(y, ty->data, sizeof(float)*(size_t) ncols*nrows);
    ggml_free(c);
}

// rope on a [dh, H, n] tensor in place (ggml.c:7076-7131); mode as in the reference
void ref_rope(float * x, int dh, int H, int n, int n_past, int mode) {
    auto & buf = op_arena((size_t) dh*H*n*4 + (1u << 20));
    ggml_init_params ip = { buf.size(), buf.data() };
    ggml_context * c = ggml_init(ip);
    ggml_tensor * tx = ggml_new_tensor_3d(c, GGML_TYPE_F32, dh, H, n);
    memcpy(tx->data, x, sizeof(float)*(size_t) dh*H*n);
    ggml_tensor * ty = ggml_rope(c, tx, n_past, dh, mode);
    ggml_cgraph gf = {};
    gf.n_threads = 1;
    ggml_build_forward_expand(&gf, ty);
    ggml_graph_compute(c, &gf);
    memcpy(x, ty->data, sizeof(float)*(size_t) dh*H*n);
    ggml_free(c);
}

// ---------------------------------------------------------------------------------------------
// tokenizer / sampler (utils.cpp:275-311, 345-428) against the model's vocab
// ---------------------------------------------------------------------------------------------
int refllama_tokenize(void * h, const char * text, int bos, int32_t * out, int cap) {
    RefModel * m = (RefModel *) h;
    std::vector<gpt_vocab::id> ids = llama_tokenize(m->vocab, text, bos != 0);
    const int n = (int) ids.size();
    for (int i = 0; i < n && i < cap; i++) out[i] = ids[i];
    return n;
}

// one sampling call with explicit rng state: seeds a fresh mt19937(seed), discards `skip` draws
// (each reference sample consumes draws inside discrete_distribution), returns the sampled id.
// For sequence parity use refllama_sampler_* below.
struct RefSampler { std::mt19937 rng; std::vector<gpt_vocab::id> last_n; };

void * refllama_sampler_new(int32_t seed, int repeat_last_n) {
    RefSampler * s = new RefSampler();
    s->rng = std::mt19937(seed);                       // .mm:773
    s->last_n.assign(repeat_last_n, 0);                // .mm:827-829
    return s;
}
void refllama_sampler_free(void * s) { delete (RefSampler *) s; }
void refllama_sampler_accept(void * sp, int32_t id) { // .mm:867-868 / 882-883
    RefSampler * s = (RefSampler *) sp;
    s->last_n.erase(s->last_n.begin());
    s->last_n.push_back(id);
}
// gpt_random_prompt(rng) on the sampler's rng: what -[LlamaPredictOperation main] does for an empty prompt
// (.mm:774-776) -- the draw comes out of the same mt19937 the sampler uses afterwards
int refllama_sampler_random_prompt(void * sp, char * out, int cap) {
    RefSampler * s = (RefSampler *) sp;
    const std::string p = gpt_random_prompt(s->rng);
    snprintf(out, cap, "%s", p.c_str());
    return (int) p.size();
}
int32_t refllama_sampler_sample(void * h, void * sp, const float * logits,
                                double repeat_penalty, int top_k, double top_p, double temp) {
    RefModel * m = (RefModel *) h;
    RefSampler * s = (RefSampler *) sp;
    return llama_sample_top_p_top_k(m->vocab, logits, s->last_n, repeat_penalty, top_k, top_p, temp, s->rng);
}

} // extern "C"
