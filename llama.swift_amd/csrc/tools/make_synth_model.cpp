// make_synth_model.cpp -- writes a synthetic ggml-model-q4_0.bin[.k] in the reference's container
// format (reader: Sources/llamaObjCxx/bridge/LlamaPredictOperation.mm:98-498; writer side:
// tools/convert-pth-to-ggml.py:92-169 + Sources/cpp/quantize.cpp:62-260).  There are no real
// LLaMA weights in this environment, so benchmarks and full-size parity runs use files made here:
// fp32 weights ~ N(0, sigma^2) pushed through a restatement of the reference's OFFLINE quantizer
// ggml_quantize_q4_0 (Sources/cpp/utils.cpp:431-485: d = amax/7, id = 1/d, q = round_half_away(x*id)+8),
// norm weights 1 + 0.1*N(0,1).  Values come from a counter-based generator keyed by
// (seed, tensor, row, column) so any shard regenerates independently (multi-part files agree with
// the single-part file of the same seed).
//
//   make_synth_model --out PATH [--preset 7B|13B|30B|65B] [--n_vocab V --n_embd D --n_mult M --n_head H
//                    --n_layer L] [--parts P] [--seed S] [--sigma X] [--threads T] [--emb_offset C]
//   --emb_offset C : embedding row r gets the constant C * (r % 4) / 2 added to every element (rows whose mean is 0, small, near and
//                    well above sigma / sqrt(3): the norm prologue of the decode kernels switches from the one-pass second moment to the
//                    reference's two-pass form when the mean dominates -- tests need rows on both sides of that threshold)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// approximately standard-normal from one 64-bit draw (sum of four 16-bit uniforms, variance-matched)
static inline float gauss(uint64_t key) {
    const uint64_t r = splitmix64(key);
    const float s = (float) ((r & 0xFFFF) + ((r >> 16) & 0xFFFF) + ((r >> 32) & 0xFFFF) + (r >> 48)) * (1.0f / 65536.0f);
    return (s - 2.0f) * 1.7320508f;
}

static uint64_t name_hash(const std::string &s) {
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}

struct Args {
    std::string out;
    int n_vocab = 32000, n_embd = 4096, n_mult = 256, n_head = 32, n_layer = 32, parts = 0, threads = 0;
    uint64_t seed = 20230312;
    float sigma = 0.02f, emb_offset = 0.0f;
};

// quantize one row slice [col0, col0+n) of tensor `tkey` row `row` to Q4_0 blocks
static void quant_row(uint64_t tkey, int64_t row, int64_t ncols_total, int64_t col0, int64_t n, float sigma, uint8_t *dst, float offset = 0.0f) {
    const float dc = offset * (float) (row & 3) * 0.5f;
    float v[32];
    for (int64_t b = 0; b < n / 32; b++) {
        float amax = 0.0f;
        for (int l = 0; l < 32; l++) {
            const int64_t col = col0 + b * 32 + l;
            v[l] = sigma * gauss(tkey ^ splitmix64((uint64_t) (row * ncols_total + col))) + dc;
            amax = fmaxf(amax, fabsf(v[l]));
        }
        const float d = amax / 7.0f;
        const float id = d ? 1.0f / d : 0.0f;
        memcpy(dst + b * 20, &d, 4);
        for (int j = 0; j < 16; j++) {
            const uint8_t q0 = (uint8_t) ((int8_t) round((double) (v[2 * j] * id)) + 8);
            const uint8_t q1 = (uint8_t) ((int8_t) round((double) (v[2 * j + 1] * id)) + 8);
            dst[b * 20 + 4 + j] = (uint8_t) (q0 | (q1 << 4));
        }
    }
}

static int split_type(const std::string &name) {      // .mm:358-388
    if (name.find("tok_embeddings") != std::string::npos) return 0;
    if (name.find("layers") != std::string::npos) {
        if (name.find("attention.wo.weight") != std::string::npos) return 0;
        if (name.find("feed_forward.w2.weight") != std::string::npos) return 0;
        return 1;
    }
    if (name.find("output") != std::string::npos) return 1;
    return 0;
}

static void write_q4(FILE *f, const Args &a, const std::string &name, int64_t rows, int64_t cols, int part, int parts) {
    const uint64_t tkey = splitmix64(a.seed ^ name_hash(name));
    int64_t r0 = 0, nr = rows, c0 = 0, nc = cols;
    if (parts > 1) {
        if (split_type(name) == 0) { nc = cols / parts; c0 = part * nc; }
        else { nr = rows / parts; r0 = part * nr; }
    }
    const int32_t hdr[3] = { 2, (int32_t) name.size(), 2 };
    const int32_t ne[2] = { (int32_t) nc, (int32_t) nr };
    fwrite(hdr, 4, 3, f); fwrite(ne, 4, 2, f); fwrite(name.data(), 1, name.size(), f);
    const int64_t row_bytes = nc / 32 * 20;
    std::vector<uint8_t> buf((size_t) (nr * row_bytes));
    const int T = a.threads;
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t]() {
            for (int64_t r = t; r < nr; r += T) quant_row(tkey, r0 + r, cols, c0, nc, a.sigma, buf.data() + r * row_bytes, name.find("tok_embeddings") != std::string::npos ? a.emb_offset : 0.0f);
        });
    for (auto &x : th) x.join();
    fwrite(buf.data(), 1, buf.size(), f);
}

static void write_f32(FILE *f, const Args &a, const std::string &name, int64_t n) {
    const uint64_t tkey = splitmix64(a.seed ^ name_hash(name));
    const int32_t hdr[3] = { 1, (int32_t) name.size(), 0 };
    const int32_t ne = (int32_t) n;
    fwrite(hdr, 4, 3, f); fwrite(&ne, 4, 1, f); fwrite(name.data(), 1, name.size(), f);
    std::vector<float> v((size_t) n);
    for (int64_t i = 0; i < n; i++) v[i] = 1.0f + 0.1f * gauss(tkey ^ splitmix64((uint64_t) i));
    fwrite(v.data(), 4, v.size(), f);
}

int main(int argc, char **argv) {
    Args a;
    for (int i = 1; i < argc; i++) {
        const std::string k = argv[i];
        auto val = [&]() -> const char * { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", k.c_str()); exit(2); } return argv[++i]; };
        if (k == "--out") a.out = val();
        else if (k == "--preset") {
            const std::string p = val();
            if (p == "7B")       { a.n_embd = 4096; a.n_head = 32; a.n_layer = 32; }
            else if (p == "13B") { a.n_embd = 5120; a.n_head = 40; a.n_layer = 40; }
            else if (p == "30B") { a.n_embd = 6656; a.n_head = 52; a.n_layer = 60; }
            else if (p == "65B") { a.n_embd = 8192; a.n_head = 64; a.n_layer = 80; }
            else { fprintf(stderr, "unknown preset %s\n", p.c_str()); return 2; }
            a.n_vocab = 32000; a.n_mult = 256;
        }
        else if (k == "--n_vocab") a.n_vocab = atoi(val());
        else if (k == "--n_embd") a.n_embd = atoi(val());
        else if (k == "--n_mult") a.n_mult = atoi(val());
        else if (k == "--n_head") a.n_head = atoi(val());
        else if (k == "--n_layer") a.n_layer = atoi(val());
        else if (k == "--parts") a.parts = atoi(val());
        else if (k == "--seed") a.seed = strtoull(val(), nullptr, 10);
        else if (k == "--sigma") a.sigma = (float) atof(val());
        else if (k == "--emb_offset") a.emb_offset = (float) atof(val());
        else if (k == "--threads") a.threads = atoi(val());
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    if (a.out.empty()) { fprintf(stderr, "usage: make_synth_model --out PATH [--preset 7B|13B|30B|65B] [...]\n"); return 2; }
    if (a.threads <= 0) a.threads = (int) std::thread::hardware_concurrency();
    if (a.threads <= 0) a.threads = 8;
    if (a.parts <= 0) a.parts = a.n_embd == 5120 ? 2 : a.n_embd == 6656 ? 4 : a.n_embd == 8192 ? 8 : 1;   // .mm:33-38
    const int64_t d = a.n_embd, V = a.n_vocab;
    const int64_t F = ((2 * (4 * d) / 3 + a.n_mult - 1) / a.n_mult) * a.n_mult;                              // .mm:135

    for (int part = 0; part < a.parts; part++) {
        const std::string fname = part == 0 ? a.out : a.out + "." + std::to_string(part);
        FILE *f = fopen(fname.c_str(), "wb");
        if (!f) { perror(fname.c_str()); return 1; }
        static char iobuf[1 << 22];
        setvbuf(f, iobuf, _IOFBF, sizeof(iobuf));
        const uint32_t magic = 0x67676d6c;
        const int32_t hp[7] = { a.n_vocab, a.n_embd, a.n_mult, a.n_head, a.n_layer, a.n_embd / a.n_head, 2 };
        fwrite(&magic, 4, 1, f); fwrite(hp, 4, 7, f);
        for (int i = 0; i < a.n_vocab; i++) {
            char w[16]; uint32_t len;
            if (i < 3) len = 0;
            else if (i < 29) { w[0] = (char) ('a' + i - 3); len = 1; }
            else if (i == 29) { w[0] = ' '; len = 1; }
            else len = (uint32_t) snprintf(w, sizeof(w), "tok%05d", i);
            fwrite(&len, 4, 1, f); fwrite(w, 1, len, f);
        }
        write_q4(f, a, "tok_embeddings.weight", V, d, part, a.parts);
        write_f32(f, a, "norm.weight", d);
        write_q4(f, a, "output.weight", V, d, part, a.parts);
        for (int l = 0; l < a.n_layer; l++) {
            const std::string p = "layers." + std::to_string(l) + ".";
            write_q4(f, a, p + "attention.wq.weight", d, d, part, a.parts);
            write_q4(f, a, p + "attention.wk.weight", d, d, part, a.parts);
            write_q4(f, a, p + "attention.wv.weight", d, d, part, a.parts);
            write_q4(f, a, p + "attention.wo.weight", d, d, part, a.parts);
            write_q4(f, a, p + "feed_forward.w1.weight", F, d, part, a.parts);
            write_q4(f, a, p + "feed_forward.w2.weight", d, F, part, a.parts);
            write_q4(f, a, p + "feed_forward.w3.weight", F, d, part, a.parts);
            write_f32(f, a, p + "attention_norm.weight", d);
            write_f32(f, a, p + "ffn_norm.weight", d);
        }
        if (fclose(f) != 0) { perror("fclose"); return 1; }
    }
    fprintf(stderr, "wrote %s (%d part%s): n_vocab %d n_embd %d n_head %d n_layer %d n_ff %lld seed %llu\n", a.out.c_str(), a.parts,
            a.parts > 1 ? "s" : "", a.n_vocab, a.n_embd, a.n_head, a.n_layer, (long long) F, (unsigned long long) a.seed);
    return 0;
}
