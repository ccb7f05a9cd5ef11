// model_file.cpp -- see model_file.h.  Error strings follow the reference loader's
// (LlamaPredictOperation.mm:101-102,111-112,176-177,353-354,394-395,408-409,441,448-449).
#include "model_file.h"

#include <cinttypes>
#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace lh {

namespace {

struct File {
    FILE *f = nullptr;
    explicit File(const std::string &p) { f = fopen(p.c_str(), "rb"); }
    ~File() { if (f) fclose(f); }
    bool ok() const { return f != nullptr; }
    bool read(void *dst, size_t n) { return fread(dst, 1, n, f) == n; }
    bool seek(int64_t off) { return fseeko(f, (off_t) off, SEEK_SET) == 0; }
    bool skip(int64_t n) { return fseeko(f, (off_t) n, SEEK_CUR) == 0; }
    int64_t tell() { return (int64_t) ftello(f); }
    int64_t size() {                       // -1 if the stream is not seekable
        const off_t at = ftello(f);
        if (at < 0 || fseeko(f, 0, SEEK_END) != 0) return -1;
        const off_t end = ftello(f);
        (void) fseeko(f, at, SEEK_SET);
        return (int64_t) end;
    }
};

std::string fmt(const char *f, ...) __attribute__((format(printf, 1, 2)));
std::string fmt(const char *f, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof(buf), f, ap);
    va_end(ap);
    return buf;
}

int parts_for_width(int n_embd) {
    // LLAMA_N_PARTS (.mm:33-38).  The reference throws for any other width; tiny test models need
    // to load, so unknown widths are single-part here.
    switch (n_embd) {
        case 4096: return 1;
        case 5120: return 2;
        case 6656: return 4;
        case 8192: return 8;
        default:   return 1;
    }
}

int split_type_of(const std::string &name) {       // .mm:358-388
    if (name.find("tok_embeddings") != std::string::npos) return 0;
    if (name.find("layers") != std::string::npos) {
        if (name.find("attention.wo.weight") != std::string::npos) return 0;
        if (name.find("feed_forward.w2.weight") != std::string::npos) return 0;
        return 1;
    }
    if (name.find("output") != std::string::npos) return 1;
    return 0;
}

void expect(std::map<std::string, TensorInfo> &m, const std::string &name, int n_dims, int64_t ne0, int64_t ne1, int wtype, int n_parts) {
    TensorInfo t;
    t.name = name; t.n_dims = n_dims; t.ne0 = ne0; t.ne1 = ne1; t.wtype = wtype; t.q4 = wtype == 2;
    t.split = split_type_of(name);
    t.shards.resize(n_parts);
    m[name] = t;
}

}  // namespace

std::string ModelFile::part_name(int part) const {
    return part == 0 ? path_ : path_ + "." + std::to_string(part);     // .mm:316-319
}

bool ModelFile::open(const std::string &path, int32_t n_ctx, int32_t force_parts, std::string &err) {
    path_ = path;
    File fin(path);
    if (!fin.ok()) { err = fmt("failed to open '%s'", path.c_str()); return false; }

    uint32_t magic = 0;
    if (!fin.read(&magic, 4) || magic != 0x67676d6c) {
        err = fmt("invalid model file '%s' (bad magic)", path.c_str());
        return false;
    }
    int32_t h[7];
    if (!fin.read(h, sizeof(h))) { err = fmt("invalid model file '%s' (truncated header)", path.c_str()); return false; }
    hp.n_vocab = h[0]; hp.n_embd = h[1]; hp.n_mult = h[2]; hp.n_head = h[3];
    hp.n_layer = h[4]; hp.n_rot = h[5]; hp.f16 = h[6];
    hp.n_ctx = n_ctx;
    // bounds before anything is sized from the header (a corrupt n_vocab used to reach std::vector::resize):
    // the same cap as the quantize tool's, and every vocabulary entry needs at least its 4-byte length
    const int64_t fsize = fin.size();
    if (hp.n_vocab <= 0 || hp.n_embd <= 0 || hp.n_mult <= 0 || hp.n_head <= 0 || hp.n_layer <= 0 ||
        hp.n_vocab > (1 << 24) || hp.n_embd > (1 << 20) || hp.n_mult > (1 << 20) || hp.n_head > hp.n_embd || hp.n_layer > (1 << 16) ||
        (fsize >= 0 && (int64_t) hp.n_vocab * 4 > fsize)) {
        err = fmt("invalid model file '%s' (bad hyper-parameters)", path.c_str());
        return false;
    }
    hp.n_ff = ((2 * (4 * hp.n_embd) / 3 + hp.n_mult - 1) / hp.n_mult) * hp.n_mult;
    hp.n_parts = force_parts > 0 ? force_parts : parts_for_width(hp.n_embd);

    id_to_token.resize(hp.n_vocab);
    for (int i = 0; i < hp.n_vocab; i++) {                              // .mm:149-163
        uint32_t len = 0;
        if (!fin.read(&len, 4) || len > (1u << 20)) { err = fmt("invalid model file '%s' (truncated vocab)", path.c_str()); return false; }
        std::string word(len, '\0');
        if (len && !fin.read(&word[0], len)) { err = fmt("invalid model file '%s' (truncated vocab)", path.c_str()); return false; }
        token_to_id[word] = i;
        id_to_token[i] = word;
    }

    switch (hp.f16) {                                                   // .mm:168-180
        case 0: case 1: case 2: case 3: break;                           // fp32, fp16, Q4_0, Q4_1 weights
        default:
            err = fmt("invalid model file '%s' (bad f16 value %d)", path.c_str(), hp.f16);
            return false;
    }

    const int64_t d = hp.n_embd, F = hp.n_ff, V = hp.n_vocab;
    const int np = hp.n_parts;
    const int wt = hp.f16;                                              // wtype of every 2-D tensor (.mm:168-180)
    expect(tensors, "tok_embeddings.weight", 2, d, V, wt, np);          // .mm:246-286
    expect(tensors, "norm.weight", 1, d, 1, 0, np);
    expect(tensors, "output.weight", 2, d, V, wt, np);
    for (int i = 0; i < hp.n_layer; i++) {
        const std::string p = "layers." + std::to_string(i) + ".";
        expect(tensors, p + "attention_norm.weight", 1, d, 1, 0, np);
        expect(tensors, p + "attention.wq.weight", 2, d, d, wt, np);
        expect(tensors, p + "attention.wk.weight", 2, d, d, wt, np);
        expect(tensors, p + "attention.wv.weight", 2, d, d, wt, np);
        expect(tensors, p + "attention.wo.weight", 2, d, d, wt, np);
        expect(tensors, p + "ffn_norm.weight", 1, d, 1, 0, np);
        expect(tensors, p + "feed_forward.w1.weight", 2, d, F, wt, np);
        expect(tensors, p + "feed_forward.w2.weight", 2, F, d, wt, np);
        expect(tensors, p + "feed_forward.w3.weight", 2, d, F, wt, np);
    }

    const int64_t tensors_at = fin.tell();                              // .mm:306 (same offset in every part, :322)

    for (int part = 0; part < np; part++) {
        const std::string fname = part_name(part);
        File fp(fname);
        if (!fp.ok()) { err = fmt("failed to open '%s'", fname.c_str()); return false; }
        if (!fp.seek(tensors_at)) { err = fmt("failed to open '%s'", fname.c_str()); return false; }

        for (;;) {
            int32_t hdr[3];
            if (!fp.read(hdr, sizeof(hdr))) break;                      // EOF ends the list (.mm:338-340)
            const int32_t n_dims = hdr[0], name_len = hdr[1], ftype = hdr[2];
            if (n_dims < 1 || n_dims > 2 || name_len < 0 || name_len > 4096) {
                err = fmt("invalid model file '%s' (corrupt tensor header)", fname.c_str());
                return false;
            }
            int32_t ne[2] = { 1, 1 };
            int64_t nelements = 1;
            for (int i = 0; i < n_dims; i++) {
                if (!fp.read(&ne[i], 4)) { err = fmt("invalid model file '%s' (corrupt tensor header)", fname.c_str()); return false; }
                nelements *= ne[i];
            }
            std::string name(name_len, '\0');
            if (name_len && !fp.read(&name[0], name_len)) { err = fmt("invalid model file '%s' (corrupt tensor header)", fname.c_str()); return false; }

            auto it = tensors.find(name);
            if (it == tensors.end()) { err = fmt("unknown tensor '%s' in model file", name.c_str()); return false; }
            TensorInfo &t = it->second;
            const int tp = (n_dims == 1) ? 1 : np;

            if ((t.ne0 * t.ne1) / tp != nelements) {                    // .mm:392-404
                err = fmt("tensor '%s' has wrong size in model file", name.c_str());
                return false;
            }
            int64_t e0 = t.ne0, e1 = t.ne1;                             // .mm:406-426
            if (n_dims == 2) { if (t.split == 0) e0 /= tp; else e1 /= tp; }
            if (e0 != ne[0] || e1 != ne[1]) {
                err = fmt("tensor '%s' has wrong shape in model file: got [%d, %d], expected [%d, %d]",
                          name.c_str(), (int) e0, (int) e1, ne[0], ne[1]);
                return false;
            }
            int64_t bpe_num, bpe_den;                                   // bytes per element as a fraction (.mm:432-444)
            switch (ftype) {
                case 0: bpe_num = 4;  bpe_den = 1;  break;
                case 1: bpe_num = 2;  bpe_den = 1;  break;
                case 2: bpe_num = 20; bpe_den = 32; break;
                case 3: bpe_num = 24; bpe_den = 32; break;
                default: err = fmt("unknown ftype %d in model file", ftype); return false;
            }
            const int64_t got = nelements * bpe_num / bpe_den;
            if (got != t.nbytes() / tp) {                               // .mm:446-465
                err = fmt("tensor '%s' has wrong size in model file: got %zu, expected %zu",
                          name.c_str(), (size_t) (t.nbytes() / tp), (size_t) got);
                return false;
            }
            if ((t.q4 || t.wtype == 3) && (ne[0] % 64) != 0) {          // assert(ne[0] % 64 == 0), .mm:437
                err = fmt("tensor '%s' has wrong shape in model file: ne[0] = %d is not a multiple of 64", name.c_str(), ne[0]);
                return false;
            }
            t.shards[part].offset = fp.tell();
            t.shards[part].ne0 = ne[0];
            t.shards[part].ne1 = ne[1];
            if (!fp.skip(got)) { err = fmt("invalid model file '%s' (truncated tensor '%s')", fname.c_str(), name.c_str()); return false; }
        }
    }

    for (const auto &kv : tensors) {
        const TensorInfo &t = kv.second;
        const int need = (t.n_dims == 1) ? 1 : np;
        for (int p = 0; p < need; p++) {
            if (t.shards[p].offset < 0) {
                err = fmt("tensor '%s' is missing from model file '%s'", t.name.c_str(), part_name(p).c_str());
                return false;
            }
        }
    }
    return true;
}

bool ModelFile::read_tensor(const std::string &name, uint8_t *dst, std::string &err) const {
    auto it = tensors.find(name);
    if (it == tensors.end()) { err = fmt("unknown tensor '%s' in model file", name.c_str()); return false; }
    const TensorInfo &t = it->second;
    const int np = (t.n_dims == 1) ? 1 : hp.n_parts;
    const int64_t row_bytes = t.row_bytes();
    for (int part = 0; part < np; part++) {
        File fp(part_name(part));
        if (!fp.ok() || !fp.seek(t.shards[part].offset)) { err = fmt("failed to open '%s'", part_name(part).c_str()); return false; }
        bool ok = true;
        if (np == 1) {
            ok = fp.read(dst, (size_t) t.nbytes());
        } else if (t.split == 0) {
            // column shard: each row receives a contiguous slice of row_bytes/np (.mm:467-477)
            const int64_t slice = row_bytes / np;
            const int64_t at = (int64_t) part * slice;                  // shards are equal slices of a row
            std::vector<uint8_t> buf((size_t) slice * t.ne1);
            ok = fp.read(buf.data(), buf.size());
            if (ok) for (int64_t r = 0; r < t.ne1; r++) memcpy(dst + r * row_bytes + at, buf.data() + r * slice, (size_t) slice);
        } else {
            // row shard: rows [part*ne1, (part+1)*ne1) (.mm:478-487)
            ok = fp.read(dst + (int64_t) part * t.shards[part].ne1 * row_bytes, (size_t) (t.shards[part].ne1 * row_bytes));
        }
        if (!ok) { err = fmt("invalid model file '%s' (truncated tensor '%s')", part_name(part).c_str(), name.c_str()); return false; }
    }
    return true;
}

}  // namespace lh
