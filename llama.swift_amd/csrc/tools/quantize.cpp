// quantize -- counterpart of the reference's tools "quantize" binary (Sources/cpp/quantize.cpp:289-338):
//   ./quantize models/llama/ggml-model-f16.bin models/llama/ggml-model-q4_0.bin 2
// on top of llamahip_quantize_file (device-side offline quantizer).
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "../../../include/llamahip.h"

int main(int argc, char **argv) {
    if (argc != 4) {                                        // quantize.cpp:292-297
        fprintf(stderr, "usage: %s model-f32.bin model-quant.bin type\n", argv[0]);
        fprintf(stderr, "  type = 2 - q4_0\n");
        fprintf(stderr, "  type = 3 - q4_1\n");
        return 1;
    }
    char err[512] = { 0 };
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = llamahip_quantize_file(argv[1], argv[2], atoi(argv[3]), err, sizeof(err));
    if (rc != 0) {
        fprintf(stderr, "%s: failed to quantize model from '%s': %s\n", argv[0], argv[1], err);
        return 1;
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("\n%s: quantize time = %8.2f ms\n", argv[0], ms);
    return 0;
}
