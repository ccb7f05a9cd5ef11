// model_file.h -- host-side reader for the reference's ggml-model-q4_0.bin[.k] container
// (Sources/llamaObjCxx/bridge/LlamaPredictOperation.mm:98-498; writer side
// tools/convert-pth-to-ggml.py:92-169 + Sources/cpp/quantize.cpp:62-260).  The file format is kept
// byte for byte; only what happens after the bytes are read is new.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace lh {

struct HParams {
    int32_t n_vocab = 0, n_embd = 0, n_mult = 0, n_head = 0, n_layer = 0, n_rot = 0, f16 = 0;
    int32_t n_ctx = 0;     // supplied by the caller, not stored in the file (.mm:125,133)
    int32_t n_ff = 0;      // .mm:135
    int32_t n_parts = 1;   // .mm:33-38,136
};

// where one tensor's shard lives inside one part file
struct ShardLoc {
    int64_t offset = -1;   // byte offset of the raw data
    int32_t ne0 = 0, ne1 = 1;
};

struct TensorInfo {
    std::string name;
    int n_dims = 0;
    int64_t ne0 = 0, ne1 = 1;        // full (merged) shape; ne0 = input dimension
    int wtype = 0;                   // storage: 0 fp32, 1 fp16, 2 Q4_0 (20 B / 32 elements), 3 Q4_1 (24 B / 32) -- the file's ftype codes
    bool q4 = false;                 // wtype == 2
    int split = 0;                   // 0: shards along ne0 (columns), 1: along ne1 (rows)  (.mm:358-388)
    std::vector<ShardLoc> shards;    // one per part (1-D tensors: only part 0 is used, .mm:446-459)
    int64_t row_bytes() const { return wtype == 2 ? (ne0 / 32) * 20 : wtype == 3 ? (ne0 / 32) * 24 : wtype == 1 ? ne0 * 2 : ne0 * 4; }
    int64_t nbytes() const { return ne1 * row_bytes(); }
};

class ModelFile {
public:
    HParams hp;
    std::vector<std::string> id_to_token;            // gpt_vocab::id_to_token (utils.h:49-55)
    std::map<std::string, int32_t> token_to_id;
    std::map<std::string, TensorInfo> tensors;

    // Parses header, vocab and the tensor directory of every part; validates shapes exactly as the
    // reference loader does.  Returns false and fills `err` with the reference's message text.
    bool open(const std::string &path, int32_t n_ctx, int32_t force_parts, std::string &err);

    // Reads one tensor and merges its shards into file-layout bytes (dst has nbytes()).
    bool read_tensor(const std::string &name, uint8_t *dst, std::string &err) const;

    const std::string &path() const { return path_; }

private:
    std::string path_;
    std::string part_name(int part) const;
};

}  // namespace lh
