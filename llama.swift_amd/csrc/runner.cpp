// runner.cpp -- host-side text utilities and the generation driver above the C ABI.
//
//   llamahip_tokenize            <- llama_tokenize              Sources/cpp/utils.cpp:275-311
//   llamahip_sample_top_p_top_k  <- llama_sample_top_p_top_k    Sources/cpp/utils.cpp:345-428
//   llama_runner_bridge_run      <- -[LlamaPredictOperation main]
//                                   Sources/llamaObjCxx/bridge/LlamaPredictOperation.mm:768-901
//
// The sampler deliberately uses the same standard-library facilities as the reference
// (std::partial_sort, std::discrete_distribution, std::mt19937): the draw sequence of
// discrete_distribution is implementation-defined, so parity is pinned against libstdc++.
#include "../../include/llama_runner.h"
#include "../../include/llamahip.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <utility>
#include <mutex>
#include <vector>

struct llamahip_sampler {
    std::mt19937 rng;
    std::vector<int32_t> last_n_tokens;
    // scratch kept between calls: "is token i among the last n" as a byte per vocabulary entry (the reference
    // runs std::find over the window for every one of the 32 000 logits: 0.3 ms per token, nothing next to its
    // 55 ms evals, a fifth of a 1.5 ms GPU token), and the candidate array
    std::vector<uint8_t> seen;
    std::vector<std::pair<double, int32_t>> cand;
};

struct llama_runner_bridge {
    std::string model_path;
    std::mutex run_lock;                 // one generation at a time per bridge
    llamahip_model *kept = nullptr;      // config.keepModel: the model of the previous run ...
    int32_t kept_n_ctx = 0;              // ... and the context size it was loaded with
    int64_t loads = 0;                   // model loads performed by this bridge (tests)
};

// llama_sample_top_p_top_k from the soft-max on (utils.cpp:397-428): `cand` = the top_k candidates, best first
static int32_t finish_sample(llamahip_sampler *s, std::vector<std::pair<double, int32_t>> &cand, double top_p) {
    double maxl = -INFINITY;
    for (const auto &c : cand) maxl = std::max(maxl, c.first);
    std::vector<double> probs;
    probs.reserve(cand.size());
    double sum = 0.0;
    for (const auto &c : cand) {
        const double p = exp(c.first - maxl);
        probs.push_back(p);
        sum += p;
    }
    for (auto &p : probs) p /= sum;

    if (top_p < 1.0f) {
        double cumsum = 0.0f;
        for (int i = 0; i < (int) probs.size(); i++) {
            cumsum += probs[i];
            if (cumsum >= top_p) {
                probs.resize(i + 1);
                cand.resize(i + 1);
                break;
            }
        }
        cumsum = 1.0 / cumsum;
        for (auto &p : probs) p *= cumsum;
    }
    std::discrete_distribution<> dist(probs.begin(), probs.end());
    const int idx = dist(s->rng);
    return cand[idx].second;
}

extern "C" {

int32_t llamahip_tokenize(const llamahip_model *m, const char *text_c, int32_t bos, int32_t *out, int32_t cap) {
    if (!m || !text_c) return 0;
    const std::string text(text_c);
    const int32_t n_vocab = llamahip_n_vocab(m);
    std::vector<int32_t> res;
    if (bos) res.push_back(1);
    // longest match at every position; among equally long matches the highest id wins because the
    // reference walks id_to_token in ascending id order and only skips strictly shorter tokens
    // (utils.cpp:275-311).  The reference tries the whole vocabulary at every position (32 000 string
    // compares per token: ~0.1 s for a 2 000-character prompt, as long as evaluating it here); only tokens
    // that start with the byte at `pos` can match, so they are bucketed by first byte once per call, in
    // ascending id order -- the same candidates in the same order, the same result.
    std::vector<std::vector<int32_t>> by_first(256);
    for (int32_t id = 0; id < n_vocab; id++) {
        uint32_t len = 0;
        const char *tok = llamahip_token_text(m, id, &len);
        if (len > 0) by_first[(unsigned char) tok[0]].push_back(id);
    }
    size_t pos = 0;
    while (pos < text.size()) {
        size_t best_len = 0;
        int32_t best_id = 0;
        for (int32_t id : by_first[(unsigned char) text[pos]]) {
            uint32_t len = 0;
            const char *tok = llamahip_token_text(m, id, &len);
            if (len < best_len) continue;
            if (len > text.size() - pos) continue;
            if (text.compare(pos, len, tok, len) == 0) { best_len = len; best_id = id; }
        }
        if (best_len == 0) break;
        res.push_back(best_id);
        pos += best_len;
    }
    for (size_t i = 0; i < res.size() && (int32_t) i < cap; i++) out[i] = res[i];
    return (int32_t) res.size();
}

llamahip_sampler *llamahip_sampler_new(int32_t seed, int32_t repeat_last_n) {
    llamahip_sampler *s = new llamahip_sampler();
    s->rng = std::mt19937(seed);                                   // .mm:773 (seed -1 -> 4294967295)
    s->last_n_tokens.assign(repeat_last_n > 0 ? repeat_last_n : 0, 0);   // .mm:827-829
    return s;
}

void llamahip_sampler_free(llamahip_sampler *s) { delete s; }

// gpt_random_prompt (utils.cpp:102-119): the prompt -[LlamaPredictOperation main] substitutes for an empty one
// (.mm:774-776).  It draws from the SAME mt19937 the sampler uses afterwards, so the draw is part of the
// sampled-token sequence of an empty-prompt run.
const char *llamahip_sampler_random_prompt(llamahip_sampler *s) {
    if (!s) return "The";
    static const char *const prompts[10] = { "So", "Once upon a time", "When", "The", "After", "If", "import", "He", "She", "They" };
    return prompts[s->rng() % 10];
}

void llamahip_sampler_accept(llamahip_sampler *s, int32_t id) {
    if (!s || s->last_n_tokens.empty()) return;
    s->last_n_tokens.erase(s->last_n_tokens.begin());
    s->last_n_tokens.push_back(id);
}

int32_t llamahip_sample_top_p_top_k(const llamahip_model *m, llamahip_sampler *s, const float *logits,
                                    double repeat_penalty, int32_t top_k, double top_p, double temp) {
    const int n_logits = llamahip_n_vocab(m);
    if (!s || !logits || n_logits <= 0) return -1;
    // the reference indexes cand.begin() + top_k unchecked (utils.cpp:389-395: undefined behaviour for a
    // vocabulary smaller than top_k); clamping changes nothing for real vocabularies
    top_k = std::min(std::max(top_k, 1), (int32_t) n_logits);
    std::vector<std::pair<double, int32_t>> &cand = s->cand;
    cand.clear();
    cand.reserve(n_logits);
    if ((int) s->seen.size() != n_logits) s->seen.assign(n_logits, 0);
    for (int32_t id : s->last_n_tokens) if (id >= 0 && id < n_logits) s->seen[id] = 1;
    const double scale = 1.0 / temp;
    for (int i = 0; i < n_logits; i++) {
        const bool seen = s->seen[i] != 0;                  // == std::find(last_n_tokens, i) != end   (utils.cpp:361)
        if (seen) {
            // CTRL-style repetition penalty: negative scores are multiplied, positive ones divided
            if (logits[i] < 0.0) cand.emplace_back(logits[i] * scale * repeat_penalty, i);
            else                 cand.emplace_back(logits[i] * scale / repeat_penalty, i);
        } else {
            cand.emplace_back(logits[i] * scale, i);
        }
    }
    for (int32_t id : s->last_n_tokens) if (id >= 0 && id < n_logits) s->seen[id] = 0;
    // (the same std::partial_sort call on the same sequence as the reference: how ties at the top_k boundary
    //  fall is the library's business, and parity is pinned against it)
    std::partial_sort(cand.begin(), cand.begin() + top_k, cand.end(),
                      [](const std::pair<double, int32_t> &a, const std::pair<double, int32_t> &b) { return a.first > b.first; });
    cand.resize(top_k);

    return finish_sample(s, cand, top_p);
}

// the same, from candidates selected on the device (llamahip_eval_topk with *exact == 1): cand[0..n) of the reference
// after its partial_sort + resize (utils.cpp:389-395)
int32_t llamahip_sample_from_candidates(llamahip_sampler *s, const double *scores, const int32_t *ids, int32_t n, double top_p) {
    if (!s || !scores || !ids || n < 1) return -1;
    std::vector<std::pair<double, int32_t>> &cand = s->cand;
    cand.clear();
    for (int i = 0; i < n; i++) cand.emplace_back(scores[i], ids[i]);
    return finish_sample(s, cand, top_p);
}

int32_t llamahip_sampler_window(const llamahip_sampler *s, int32_t *out, int32_t cap) {
    if (!s) return 0;
    if (!out) cap = 0;
    const int32_t n = (int32_t) s->last_n_tokens.size();
    for (int32_t i = 0; i < n && i < cap; i++) out[i] = s->last_n_tokens[i];
    return n;
}

// ------------------------------------------------------------------------------------------------
// bridge + generation driver
// ------------------------------------------------------------------------------------------------
void llama_runner_config_default(llama_runner_config *c) {
    if (!c) return;
    c->numberOfThreads = 8;       // LlamaRunner.swift:17
    c->numberOfTokens = 512;
    c->reversePrompt = nullptr;
    c->n_ctx = 0;
    c->greedy = 0;
    c->seed = -1;                 // utils.h:16
    c->keepModel = 0;             // the reference reloads the model on every run
}

llama_runner_bridge *llama_runner_bridge_new(const char *model_path) {
    llama_runner_bridge *b = new llama_runner_bridge();
    b->model_path = model_path ? model_path : "";
    return b;
}
void llama_runner_bridge_free(llama_runner_bridge *b) {
    if (!b) return;
    if (b->kept) llamahip_model_free(b->kept);
    delete b;
}
int64_t llama_runner_bridge_loads(const llama_runner_bridge *b) { return b ? b->loads : 0; }
const char *llama_runner_bridge_model_path(const llama_runner_bridge *b) { return b ? b->model_path.c_str() : nullptr; }

int32_t llama_runner_bridge_run(llama_runner_bridge *b, const char *prompt_c, const llama_runner_config *config,
                                llama_event_handler handler, void *user) {
    if (!b) return LLAMAHIP_ERR_LOAD;
    std::lock_guard<std::mutex> guard(b->run_lock);
    llama_runner_config cfg;
    llama_runner_config_default(&cfg);
    if (config) cfg = *config;
    auto post = [&](llama_event_type t, const char *text, uint32_t len, int32_t code) {
        if (handler) handler(user, t, text, len, code);
    };
    // gpt_params defaults that the bridge does not override (utils.h:15-37)
    const int32_t repeat_last_n = 64, top_k = 40, n_batch = 8;
    const double top_p = 0.95f, temp = 0.80f, repeat_penalty = 1.30f;
    const int32_t n_ctx = cfg.n_ctx > 0 ? cfg.n_ctx : 512;                          // .mm:790
    const int32_t n_threads = (int32_t) cfg.numberOfThreads;
    std::string prompt = prompt_c ? prompt_c : "";

    char err[512] = { 0 };
    post(LLAMA_EVENT_STARTED_LOADING_MODEL, nullptr, 0, 0);                         // .mm:785
    llamahip_model *model = nullptr;
    if (b->kept && (!cfg.keepModel || b->kept_n_ctx != n_ctx)) {                    // a kept model that no longer fits the request
        llamahip_model_free(b->kept);
        b->kept = nullptr;
    }
    if (b->kept) {
        model = b->kept;                                                            // every position the run reads is rewritten first
    } else {
        int rc = llamahip_model_load(b->model_path.c_str(), n_ctx, nullptr, &model, err, sizeof(err));
        if (rc != 0) {                                                              // .mm:790-793
            post(LLAMA_EVENT_FAILED, err, (uint32_t) strlen(err), LLAMAHIP_ERR_LOAD);
            return LLAMAHIP_ERR_LOAD;
        }
        b->loads++;
        if (cfg.keepModel) { b->kept = model; b->kept_n_ctx = n_ctx; }
    }
    auto release = [&](void) { if (model != b->kept) llamahip_model_free(model); };
    post(LLAMA_EVENT_FINISHED_LOADING_MODEL, nullptr, 0, 0);                        // .mm:797
    post(LLAMA_EVENT_STARTED_GENERATING_OUTPUT, nullptr, 0, 0);                     // .mm:800

    const int32_t n_vocab = llamahip_n_vocab(model);
    llamahip_sampler *sampler = llamahip_sampler_new(cfg.seed, repeat_last_n);       // rng: .mm:773, window: .mm:827-829
    if (prompt.empty()) prompt = llamahip_sampler_random_prompt(sampler);             // .mm:774-776 (one rng draw)
    std::vector<int32_t> embd_inp(prompt.size() + 2);
    const int32_t n_inp = llamahip_tokenize(model, prompt.c_str(), 1, embd_inp.data(), (int32_t) embd_inp.size());   // .mm:810
    embd_inp.resize(n_inp);
    int32_t n_predict = std::min((int32_t) cfg.numberOfTokens, n_ctx - n_inp);     // .mm:812
    if (cfg.reversePrompt) {                                                        // .mm:815 (result unused there too)
        std::vector<int32_t> anti(strlen(cfg.reversePrompt) + 2);
        (void) llamahip_tokenize(model, cfg.reversePrompt, 0, anti.data(), (int32_t) anti.size());
    }

    std::vector<float> logits(n_vocab);
    double cand_scores[64];
    int32_t cand_ids[64];
    static const bool host_sampler = getenv("LLAMAHIP_HOST_SAMPLER") != nullptr;      // measurement: always copy the logits and select on the host
    auto fail = [&](void) {
        post(LLAMA_EVENT_FAILED, err, (uint32_t) strlen(err), LLAMAHIP_ERR_PREDICT);
        llamahip_sampler_free(sampler);
        release();
        return (int32_t) LLAMAHIP_ERR_PREDICT;
    };
    {   // warm-up eval that sizes the reference's scratch buffer (.mm:820-825); kept because it
        // writes KV rows 0..3 of every layer exactly as the reference does
        const int32_t warm[4] = { 0, 1, 2, 3 };
        if (n_ctx >= 4 && llamahip_eval(model, n_threads, 0, warm, 4, logits.data(), err, sizeof(err)) != 0) return fail();
    }

    std::vector<int32_t> embd;
    int32_t n_past = 0, remaining = n_predict;
    size_t consumed = 0;
    while (remaining > 0) {                                                         // .mm:834
        bool have_cand = false;
        if ((int32_t) embd.size() > n_batch + 1) {
            // a run of prompt chunks (gathered below): everything but the last chunk in ONE pass that leaves the KV cache exactly as the
            // reference's chunk-by-chunk evals do (llamahip_eval_chunks); the last chunk takes the usual route, its logits may be sampled
            const int32_t chunk = n_batch + 1, n_full = (((int32_t) embd.size() - 1) / chunk) * chunk;
            if (llamahip_eval_chunks(model, n_threads, n_past, embd.data(), n_full, chunk, nullptr, err, sizeof(err)) != 0) return fail();
            n_past += n_full;
            embd.erase(embd.begin(), embd.begin() + n_full);
        }
        if (!embd.empty()) {
            // the logits of this eval are sampled next iff the prompt is used up (.mm:851): then the candidate scores and
            // the top-k selection run on the device and 816 bytes come back instead of 128 KB (exact = 0: a tie only
            // libstdc++'s partial_sort can order -- the full row came back and the host path below decides)
            const bool samples_next = embd_inp.size() <= consumed && !cfg.greedy && !host_sampler;
            if (samples_next) {
                // (the whole repetition window goes along, whatever repeat_last_n is: the device path takes up to 1024 ids and
                //  llamahip_eval_topk falls back to the host sampler beyond that)
                int32_t exact = 0;
                std::vector<int32_t> winv((size_t) std::max(repeat_last_n, 1));
                int32_t *win = winv.data();
                const int32_t nw = std::min(llamahip_sampler_window(sampler, win, (int32_t) winv.size()), (int32_t) winv.size());
                if (llamahip_eval_topk(model, n_threads, n_past, embd.data(), (int32_t) embd.size(), win, nw, repeat_penalty, top_k, temp,
                                       cand_scores, cand_ids, &exact, logits.data(), err, sizeof(err)) != 0) return fail();
                have_cand = exact == 1;
            } else if (llamahip_eval(model, n_threads, n_past, embd.data(), (int32_t) embd.size(), logits.data(), err, sizeof(err)) != 0) return fail();
        }
        n_past += (int32_t) embd.size();
        embd.clear();
        if (embd_inp.size() <= consumed) {                                          // .mm:851
            int32_t id;
            if (cfg.greedy) {
                id = (int32_t) (std::max_element(logits.begin(), logits.end()) - logits.begin());   // first maximum = lowest index
            } else if (have_cand) {
                id = llamahip_sample_from_candidates(sampler, cand_scores, cand_ids, std::min(top_k, n_vocab), top_p);
            } else {
                id = llamahip_sample_top_p_top_k(model, sampler, logits.data(), repeat_penalty, top_k, top_p, temp);
            }
            llamahip_sampler_accept(sampler, id);
            embd.push_back(id);
            --remaining;
        } else {
            // .mm:880-888 hands the prompt over in chunks of n_batch + 1 tokens, one llama_eval each; here the whole rest of the prompt is
            // taken at once (same tokens, same order, same sampler window) and evaluated chunk-exactly above
            while (embd_inp.size() > consumed) {
                embd.push_back(embd_inp[consumed]);
                llamahip_sampler_accept(sampler, embd_inp[consumed]);
                ++consumed;
            }
        }
        for (int32_t id : embd) {                                                   // .mm:892-895
            uint32_t len = 0;
            const char *tok = llamahip_token_text(model, id, &len);
            post(LLAMA_EVENT_OUTPUT_TOKEN, tok, len, 0);
        }
    }
    llamahip_sampler_free(sampler);
    post(LLAMA_EVENT_COMPLETED, nullptr, 0, 0);                                     // .mm:898
    release();                                                                      // .mm:900
    return 0;
}

}  // extern "C"
