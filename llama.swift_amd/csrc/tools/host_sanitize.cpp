// host_sanitize.cpp -- the host half of the drop-in (model-file reader, tokenizer, sampler, generation driver) under
// AddressSanitizer + UndefinedBehaviorSanitizer ("race / memory-error detection").  Built by
// `make asan` with g++ -fsanitize=address,undefined from model_file.cpp + runner.cpp as they are; the device-side
// entry points the driver calls are replaced by stubs that parse the file and then fail like a box without a GPU, so
// the run exercises: header / vocabulary / tensor-directory parsing (valid, truncated and corrupted files), multi-part
// shard merging through read_tensor, llamahip_tokenize, llamahip_sample_top_p_top_k / _from_candidates, and the
// bridge's failure path, and -- with evals that succeed on made-up logits -- the driver's whole control flow (prompt taken at once,
// chunk-exact pass + last chunk, one eval per generated token, event counts).  usage: host_sanitize <model file> [n_parts]   (exit code 0 = clean)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../../include/llama_runner.h"
#include "../../../include/llamahip.h"
#include "../model_file.h"

struct llamahip_model { lh::ModelFile file; };

extern "C" {
int llamahip_model_load(const char *path, int32_t n_ctx, const llamahip_opts *opts, llamahip_model **out, char *err, size_t err_cap) {
    llamahip_model *m = new llamahip_model();
    std::string e;
    if (!m->file.open(path, n_ctx, opts ? opts->n_parts : 0, e)) { snprintf(err, err_cap, "%s", e.c_str()); delete m; *out = nullptr; return LLAMAHIP_ERR_LOAD; }
    *out = m;
    return 0;
}
void llamahip_model_free(llamahip_model *m) { delete m; }
int32_t llamahip_n_vocab(const llamahip_model *m) { return m ? m->file.hp.n_vocab : 0; }
const char *llamahip_token_text(const llamahip_model *m, int32_t id, uint32_t *len) {
    if (!m || id < 0 || id >= m->file.hp.n_vocab) return nullptr;
    if (len) *len = (uint32_t) m->file.id_to_token[id].size();
    return m->file.id_to_token[id].data();
}
// Two modes.  Default: every eval fails like a box without a GPU.  g_drive: evals SUCCEED with deterministic made-up logits and are
// logged, so that the generation driver's control flow (warm-up, the prompt taken at once and evaluated chunk-exactly, the last
// chunk's sampled logits, one eval per generated token, the event sequence) runs end to end on the host, under the sanitizers.
struct EvalCall { int kind /* 0 eval, 1 eval_chunks, 2 eval_topk */, n_past, n, chunk; };
static bool g_drive = false;
static std::vector<EvalCall> g_calls;
static int fake_eval(llamahip_model *m, int kind, int32_t np, const int32_t *t, int32_t n, int32_t chunk, float *lg, char *err, size_t err_cap) {
    if (!g_drive) { snprintf(err, err_cap, "no HIP device available: libllamahip has no CPU fallback"); return LLAMAHIP_ERR_PREDICT; }
    if (!m || !t || n < 1 || np < 0 || np + n > m->file.hp.n_ctx) { snprintf(err, err_cap, "context overflow: n_past (%d) + n_tokens (%d) > n_ctx (%d)", np, n, m ? m->file.hp.n_ctx : 0); return LLAMAHIP_ERR_PREDICT; }
    g_calls.push_back({ kind, np, n, chunk });
    const int V = m->file.hp.n_vocab;
    for (int i = 0; lg && i < V; i++) lg[i] = (float) (((uint32_t) i * 2654435761u + (uint32_t) (np + n) * 97u + (uint32_t) t[n - 1] * 13u) >> 20 & 0xffu) * 0.03125f;
    return 0;
}
int llamahip_eval(llamahip_model *m, int32_t, int32_t np, const int32_t *t, int32_t n, float *lg, char *err, size_t err_cap) { return fake_eval(m, 0, np, t, n, 0, lg, err, err_cap); }
int llamahip_eval_chunks(llamahip_model *m, int32_t, int32_t np, const int32_t *t, int32_t n, int32_t chunk, float *lg, char *err, size_t err_cap) { return fake_eval(m, 1, np, t, n, chunk, lg, err, err_cap); }
int llamahip_eval_topk(llamahip_model *m, int32_t, int32_t np, const int32_t *t, int32_t n, const int32_t *, int32_t, double, int32_t, double,
                       double *, int32_t *, int32_t *exact, float *lg, char *err, size_t err_cap) { *exact = 0; return fake_eval(m, 2, np, t, n, 0, lg, err, err_cap); }
}

static int failures = 0;
#define EXPECT(c) do { if (!(c)) { fprintf(stderr, "host_sanitize: FAILED %s (line %d)\n", #c, __LINE__); failures++; } } while (0)

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: host_sanitize <model file> [n_parts]\n"); return 2; }
    const std::string path = argv[1];
    const int parts = argc > 2 ? atoi(argv[2]) : 0;
    char err[512] = { 0 };
    llamahip_opts o; memset(&o, 0, sizeof(o)); o.struct_size = sizeof(o); o.device = -1; o.layer_end = -1; o.n_parts = parts;
    llamahip_model *m = nullptr;
    EXPECT(llamahip_model_load(path.c_str(), 64, &o, &m, err, sizeof(err)) == 0);
    if (!m) { fprintf(stderr, "%s\n", err); return 1; }
    // every tensor through the shard merge
    size_t total = 0;
    for (const auto &kv : m->file.tensors) {
        std::vector<uint8_t> buf((size_t) kv.second.nbytes());
        std::string e;
        EXPECT(m->file.read_tensor(kv.first, buf.data(), e));
        total += buf.size();
    }
    // tokenizer on vocabulary pieces, odd bytes and the empty string
    std::mt19937 rng(5);
    const int V = llamahip_n_vocab(m);
    for (int it = 0; it < 200; it++) {
        std::string text;
        const int n = (int) (rng() % 12);
        for (int i = 0; i < n; i++) {
            uint32_t len = 0;
            const char *t = llamahip_token_text(m, (int32_t) (rng() % V), &len);
            if (t && len && !memchr(t, 0, len)) text.append(t, len);
            if (rng() % 7 == 0) text.push_back((char) (rng() % 255 + 1));
        }
        std::vector<int32_t> out(text.size() + 2);
        const int32_t nt = llamahip_tokenize(m, text.c_str(), it & 1, out.data(), (int32_t) out.size());
        EXPECT(nt >= 0 && nt <= (int32_t) out.size());
        EXPECT(llamahip_tokenize(m, text.c_str(), 1, out.data(), 1) >= 0);          // cap smaller than the result
    }
    // sampler: both halves, tiny and oversized top_k, ties, a seen-everything window
    llamahip_sampler *s = llamahip_sampler_new(-1, 64);
    std::vector<float> lg(V);
    for (int it = 0; it < 300; it++) {
        for (int i = 0; i < V; i++) lg[i] = (float) ((int) (rng() % 41) - 20) * (it % 2 ? 0.25f : 0.37f);
        const int32_t id = llamahip_sample_top_p_top_k(m, s, lg.data(), 1.3, (int32_t) (it % 5 == 0 ? V + 7 : 1 + rng() % 40), it % 3 ? 0.95 : 1.0, 0.8);
        EXPECT(id >= 0 && id < V);
        llamahip_sampler_accept(s, id);
        double sc[4] = { 3.0, 2.0, 1.0, 0.5 }; int32_t ids[4] = { 1, 2, 3, 4 };
        EXPECT(llamahip_sample_from_candidates(s, sc, ids, 1 + it % 4, 0.95) >= 1);
    }
    int32_t win[64];
    EXPECT(llamahip_sampler_window(s, win, 64) == 64);
    EXPECT(llamahip_sampler_random_prompt(s) != nullptr);
    llamahip_sampler_free(s);
    llamahip_model_free(m);
    // corrupted copies of the file: every header field and a sweep of truncations must be load errors, never crashes
    std::vector<uint8_t> raw;
    { FILE *f = fopen(path.c_str(), "rb"); fseek(f, 0, SEEK_END); raw.resize((size_t) ftell(f)); fseek(f, 0, SEEK_SET); EXPECT(fread(raw.data(), 1, raw.size(), f) == raw.size()); fclose(f); }
    const std::string tmp = path + ".sanitize.tmp";
    auto try_load = [&](const std::vector<uint8_t> &bytes) {
        FILE *f = fopen(tmp.c_str(), "wb"); if (!bytes.empty()) fwrite(bytes.data(), 1, bytes.size(), f); fclose(f);
        llamahip_model *mm = nullptr;
        const int rc = llamahip_model_load(tmp.c_str(), 64, &o, &mm, err, sizeof(err));
        if (mm) llamahip_model_free(mm);
        return rc;
    };
    if (parts <= 1) {
        for (int field = 0; field < 8; field++)
            for (uint32_t v : { 0u, 0x7fffffffu, 0xffffffffu, 1u << 24 }) {
                std::vector<uint8_t> b = raw;
                memcpy(b.data() + 4 * field, &v, 4);
                (void) try_load(b);
            }
        for (size_t cut : { (size_t) 0, (size_t) 3, (size_t) 31, (size_t) 40, raw.size() / 3, raw.size() / 2, raw.size() - 1 }) {
            std::vector<uint8_t> b(raw.begin(), raw.begin() + (long) std::min(cut, raw.size()));
            EXPECT(try_load(b) != 0 || cut >= raw.size() - 1);
        }
    }
    remove(tmp.c_str());
    // the bridge without a GPU: load succeeds (stub), the first eval fails -> exactly one `failed` event, no leak
    struct Ev { int failed = 0, completed = 0; } ev;
    llama_runner_bridge *b = llama_runner_bridge_new(path.c_str());
    llama_runner_config c; llama_runner_config_default(&c); c.numberOfTokens = 4; c.n_ctx = 64;
    const int32_t rc = llama_runner_bridge_run(b, "", &c, [](void *u, llama_event_type t, const char *, uint32_t, int32_t) {
        Ev *e = (Ev *) u; if (t == LLAMA_EVENT_FAILED) e->failed++; if (t == LLAMA_EVENT_COMPLETED) e->completed++; }, &ev);
    EXPECT((rc == LLAMAHIP_ERR_PREDICT || rc == LLAMAHIP_ERR_LOAD) && ev.failed == 1 && ev.completed == 0);   // (a multi-part file of a non-LLaMA width fails in the loader: the bridge cannot force the part count)
    llama_runner_bridge_free(b);
    // the bridge's control flow with evals that succeed (made-up logits): prompts of 1 ... 40 tokens (one-part files: the bridge derives the
    // part count from n_embd, .mm:33-38, and cannot be told that a test file of another width has two)
    if (parts <= 1) {
        g_drive = true;
        llamahip_model *mv = nullptr;
        EXPECT(llamahip_model_load(path.c_str(), 64, &o, &mv, err, sizeof(err)) == 0);
        for (int want_tokens : { 1, 5, 9, 10, 18, 19, 27, 40 }) {
            if (!mv) break;
            // a prompt text made of vocabulary pieces; P = what the tokenizer (with BOS) makes of it
            std::string text;
            std::vector<int32_t> toks(512);
            int32_t P = 1;
            for (int tries = 0; tries < 400 && P < want_tokens; tries++) {
                uint32_t len = 0;
                const char *t = llamahip_token_text(mv, (int32_t) (3 + rng() % (uint32_t) (V - 3)), &len);
                if (!t || !len || memchr(t, 0, len)) continue;
                const std::string cand = text + std::string(t, len);
                const int32_t n = llamahip_tokenize(mv, cand.c_str(), 1, toks.data(), (int32_t) toks.size());
                if (n <= want_tokens) { text = cand; P = n; }
            }
            g_calls.clear();
            struct Ev2 { int tokens = 0, failed = 0, completed = 0; } e2;
            llama_runner_bridge *b2 = llama_runner_bridge_new(path.c_str());
            llama_runner_config c2; llama_runner_config_default(&c2); c2.numberOfTokens = 6; c2.n_ctx = 64;
            const int32_t rc2 = llama_runner_bridge_run(b2, text.c_str(), &c2, [](void *u, llama_event_type t, const char *, uint32_t, int32_t) {
                Ev2 *e = (Ev2 *) u; if (t == LLAMA_EVENT_OUTPUT_TOKEN) e->tokens++; if (t == LLAMA_EVENT_FAILED) e->failed++; if (t == LLAMA_EVENT_COMPLETED) e->completed++; }, &e2);
            llama_runner_bridge_free(b2);
            const int n_predict = std::min(6, 64 - P), chunk = 9;
            const int n_full = P > chunk ? ((P - 1) / chunk) * chunk : 0;
            EXPECT(rc2 == 0 && e2.failed == 0 && e2.completed == 1);
            EXPECT(e2.tokens == P + n_predict);                                        // the prompt echoed, then the generated tokens (.mm:892-895)
            size_t k = 0;
            EXPECT(g_calls.size() == (size_t) (1 + (n_full ? 1 : 0) + 1 + (n_predict - 1)));
            if (g_calls.size() == (size_t) (1 + (n_full ? 1 : 0) + 1 + (n_predict - 1))) {
                EXPECT(g_calls[k].kind == 0 && g_calls[k].n_past == 0 && g_calls[k].n == 4); k++;                      // warm-up (.mm:820-822)
                if (n_full) { EXPECT(g_calls[k].kind == 1 && g_calls[k].n_past == 0 && g_calls[k].n == n_full && g_calls[k].chunk == chunk); k++; }
                EXPECT(g_calls[k].kind == 2 && g_calls[k].n_past == n_full && g_calls[k].n == P - n_full); k++;          // the last chunk: its logits are sampled
                for (int g = 0; g + 1 < n_predict; g++, k++) EXPECT(g_calls[k].n_past == P + g && g_calls[k].n == 1);
            }
            printf("host_sanitize: driver with a %d-token prompt: %zu evals (%d tokens in one chunk-exact pass, %d in the last chunk), %d token events\n",
                   P, g_calls.size(), n_full, P - n_full, e2.tokens);
        }
        if (mv) llamahip_model_free(mv);
        g_drive = false;
    }
    printf("host_sanitize: %zu tensor bytes merged, tokenizer + sampler + bridge failure path exercised: %s\n", total, failures ? "FAILED" : "clean");
    return failures ? 1 : 0;
}
