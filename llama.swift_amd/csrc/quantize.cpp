// quantize.cpp -- llamahip_quantize_file: the step before the hot path.
// Replaces llama_model_quantize (Sources/cpp/quantize.cpp:32-286): copies the container (magic,
// hparams with the f16 field set to the target type, vocabulary), quantizes every 2-D tensor whose name
// matches ".*weight" from f32 / f16 to Q4_0 or Q4_1 with the reference's OFFLINE quantizers
// (utils.cpp:431-485, :487-544) and copies everything else verbatim.  The arithmetic runs on the device
// (k_quantize_offline); the host only moves bytes.  No CPU fallback.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "../../include/llamahip.h"
#include "llamahip_internal.h"

namespace {

void set_err(char *err, size_t cap, const char *fmt, ...) {
    if (!err || cap == 0) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, cap, fmt, ap);
    va_end(ap);
}

struct File {
    FILE *f = nullptr;
    ~File() { if (f) fclose(f); }
};

bool ends_with_weight(const std::string &name) {          // std::regex_match(name, ".*weight")
    return name.size() >= 6 && name.compare(name.size() - 6, 6, "weight") == 0;
}

}  // namespace

static int quantize_file_impl(const char *fname_inp, const char *fname_out, int32_t itype, char *err, size_t err_cap);

// No C++ exception may cross the C ABI.
extern "C" int llamahip_quantize_file(const char *fname_inp, const char *fname_out, int32_t itype, char *err, size_t err_cap) {
    try {
        return quantize_file_impl(fname_inp, fname_out, itype, err, err_cap);
    } catch (const std::exception &ex) {
        set_err(err, err_cap, "quantize failed: %s", ex.what());
    } catch (...) {
        set_err(err, err_cap, "quantize failed: unknown exception");
    }
    return LLAMAHIP_ERR_LOAD;
}

static int quantize_file_impl(const char *fname_inp, const char *fname_out, int32_t itype, char *err, size_t err_cap) {
    using namespace lh;
    if (!fname_inp || !fname_out) { set_err(err, err_cap, "null file name"); return LLAMAHIP_ERR_LOAD; }
    if (itype != 2 && itype != 3) {                        // quantize.cpp:35-39: 2 = Q4_0, 3 = Q4_1
        set_err(err, err_cap, "invalid quantization type %d", itype);
        return LLAMAHIP_ERR_LOAD;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_err(err, err_cap, "no HIP device available: libllamahip has no CPU fallback");
        return LLAMAHIP_ERR_LOAD;
    }
    File in, out;
    in.f = fopen(fname_inp, "rb");
    if (!in.f) { set_err(err, err_cap, "failed to open '%s' for reading", fname_inp); return LLAMAHIP_ERR_LOAD; }      // quantize.cpp:49-53
    out.f = fopen(fname_out, "wb");
    if (!out.f) { set_err(err, err_cap, "failed to open '%s' for writing", fname_out); return LLAMAHIP_ERR_LOAD; }     // quantize.cpp:55-59
    auto rd = [&](void *p, size_t n) { return fread(p, 1, n, in.f) == n; };
    auto wr = [&](const void *p, size_t n) { return fwrite(p, 1, n, out.f) == n; };
#define IO_TRY(x) do { if (!(x)) { set_err(err, err_cap, "i/o error while quantizing '%s' (truncated file?)", fname_inp); return LLAMAHIP_ERR_LOAD; } } while (0)

    uint32_t magic = 0;
    IO_TRY(rd(&magic, 4));
    if (magic != 0x67676d6c) { set_err(err, err_cap, "invalid model file '%s' (bad magic)", fname_inp); return LLAMAHIP_ERR_LOAD; }   // quantize.cpp:62-70
    IO_TRY(wr(&magic, 4));
    int32_t hp[7];                                          // n_vocab n_embd n_mult n_head n_layer n_rot f16
    IO_TRY(rd(hp, sizeof(hp)));
    const int32_t n_vocab = hp[0];
    hp[6] = itype;                                          // quantize.cpp:100
    IO_TRY(wr(hp, sizeof(hp)));
    if (n_vocab < 0 || n_vocab > (1 << 24)) { set_err(err, err_cap, "invalid model file '%s' (bad vocab size %d)", fname_inp, n_vocab); return LLAMAHIP_ERR_LOAD; }
    std::string word;
    for (int32_t i = 0; i < n_vocab; i++) {                 // quantize.cpp:104-127
        uint32_t len = 0;
        IO_TRY(rd(&len, 4));
        if (len > (1u << 20)) { set_err(err, err_cap, "invalid model file '%s' (token %d is %u bytes long)", fname_inp, i, len); return LLAMAHIP_ERR_LOAD; }
        word.resize(len);
        IO_TRY(len == 0 || rd(&word[0], len));
        IO_TRY(wr(&len, 4));
        IO_TRY(len == 0 || wr(word.data(), len));
    }

    std::vector<uint8_t> h_in, h_out;
    uint8_t *d_in = nullptr, *d_out = nullptr;
    size_t cap_in = 0, cap_out = 0;
    int rc = LLAMAHIP_OK;
    for (;;) {                                              // quantize.cpp:142-262
        int32_t n_dims = 0, length = 0, ftype = 0;
        if (fread(&n_dims, 1, 4, in.f) != 4) break;         // clean end of file
        if (!rd(&length, 4) || !rd(&ftype, 4)) { set_err(err, err_cap, "i/o error while quantizing '%s' (truncated tensor header)", fname_inp); rc = LLAMAHIP_ERR_LOAD; break; }
        if (n_dims < 1 || n_dims > 2 || length < 0 || length > 4096) { set_err(err, err_cap, "invalid tensor header in '%s'", fname_inp); rc = LLAMAHIP_ERR_LOAD; break; }
        int32_t ne[2] = { 1, 1 };
        int64_t nelements = 1;
        bool ok = true;
        for (int i = 0; i < n_dims; i++) { ok = ok && rd(&ne[i], 4); nelements *= ne[i]; }
        std::string name((size_t) length, 0);
        ok = ok && (length == 0 || rd(&name[0], (size_t) length));
        if (!ok || nelements < 0) { set_err(err, err_cap, "i/o error while quantizing '%s' (truncated tensor header)", fname_inp); rc = LLAMAHIP_ERR_LOAD; break; }
        const bool quantize = ends_with_weight(name) && n_dims == 2;          // quantize.cpp:171-185
        const int32_t ftype_in = ftype;
        if (quantize) {
            if (ftype != 0 && ftype != 1) { set_err(err, err_cap, "unsupported ftype %d for integer quantization (tensor '%s')", ftype, name.c_str()); rc = LLAMAHIP_ERR_LOAD; break; }   // quantize.cpp:188-191
            if (ne[0] % 32 != 0) { set_err(err, err_cap, "tensor '%s': row length %d is not a multiple of the Q4_0 block size", name.c_str(), ne[0]); rc = LLAMAHIP_ERR_LOAD; break; }
            ftype = itype;
        } else if (ftype != 0 && ftype != 1) {
            // the reference copies "nelements * 2 bytes" for every non-f32 type (quantize.cpp:207); only f32 / f16 are meaningful
            set_err(err, err_cap, "tensor '%s' has ftype %d: the input must be an f32 / f16 model", name.c_str(), ftype); rc = LLAMAHIP_ERR_LOAD; break;
        }
        const size_t bpe = ftype_in == 0 ? 4 : 2;
        const size_t in_bytes = (size_t) nelements * bpe;
        h_in.resize(in_bytes);
        if (!rd(h_in.data(), in_bytes)) { set_err(err, err_cap, "i/o error while quantizing '%s' (tensor '%s' is truncated)", fname_inp, name.c_str()); rc = LLAMAHIP_ERR_LOAD; break; }
        ok = wr(&n_dims, 4) && wr(&length, 4) && wr(&ftype, 4);
        for (int i = 0; i < n_dims; i++) ok = ok && wr(&ne[i], 4);
        ok = ok && (length == 0 || wr(name.data(), (size_t) length));
        if (quantize) {
            const long nblocks = (long) (nelements / 32);
            const size_t out_bytes = (size_t) nblocks * (itype == 2 ? 20 : 24);
            if (in_bytes > cap_in) { if (d_in) (void) hipFree(d_in); d_in = nullptr; if (hipMalloc((void **) &d_in, in_bytes) != hipSuccess) { rc = LLAMAHIP_ERR_LOAD; set_err(err, err_cap, "hipMalloc failed"); break; } cap_in = in_bytes; }
            if (out_bytes > cap_out) { if (d_out) (void) hipFree(d_out); d_out = nullptr; if (hipMalloc((void **) &d_out, out_bytes) != hipSuccess) { rc = LLAMAHIP_ERR_LOAD; set_err(err, err_cap, "hipMalloc failed"); break; } cap_out = out_bytes; }
            h_out.resize(out_bytes);
            if (hipMemcpy(d_in, h_in.data(), in_bytes, hipMemcpyHostToDevice) != hipSuccess ||
                (itype == 2 ? launch_quantize_offline(d_in, ftype_in == 1, d_out, nblocks, nullptr)
                            : launch_quantize_q41_offline(d_in, ftype_in == 1, d_out, (long) ne[1], ne[0] / 32, nullptr)) != hipSuccess ||
                hipMemcpy(h_out.data(), d_out, out_bytes, hipMemcpyDeviceToHost) != hipSuccess) {
                set_err(err, err_cap, "HIP error while quantizing tensor '%s': %s", name.c_str(), hipGetErrorString(hipGetLastError()));
                rc = LLAMAHIP_ERR_LOAD;
                break;
            }
            ok = ok && wr(h_out.data(), out_bytes);
        } else {
            ok = ok && wr(h_in.data(), in_bytes);
        }
        if (!ok) { set_err(err, err_cap, "failed to write '%s'", fname_out); rc = LLAMAHIP_ERR_LOAD; break; }
    }
#undef IO_TRY
    if (d_in) (void) hipFree(d_in);
    if (d_out) (void) hipFree(d_out);
    if (rc == LLAMAHIP_OK && fflush(out.f) != 0) { set_err(err, err_cap, "failed to write '%s'", fname_out); rc = LLAMAHIP_ERR_LOAD; }
    return rc;
}
