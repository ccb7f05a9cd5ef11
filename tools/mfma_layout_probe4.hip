// mfma_layout_probe4.hip -- register layout of v_mfma_f32_16x16x4_4b_f16 on gfx950 (bring-up tooling for k_gemm_mfma4):
// A(block g, row m, k) = (k == 0) * (m + 16 g), B = (k == 0): every result register holds row + 16 block; then the same for columns.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_layout_probe4.hip -o tools/mfma_layout_probe4
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16v __attribute__((ext_vector_type(16)));
__global__ void k(float *out) {
    const int lane = threadIdx.x, m = lane & 15, g = lane >> 4;
    h4 a = { (_Float16) (float) (m + 16 * g), 0, 0, 0 };
    h4 b = { (_Float16) 1.0f, 0, 0, 0 };
    f32x16v c = {};
    c = __builtin_amdgcn_mfma_f32_16x16x4f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) out[lane * 16 + r] = c[r];
    h4 a2 = { (_Float16) 1.0f, 0, 0, 0 }, b2 = { (_Float16) (float) (m + 16 * g), 0, 0, 0 };
    f32x16v c2 = {};
    c2 = __builtin_amdgcn_mfma_f32_16x16x4f16(a2, b2, c2, 0, 0, 0);
    for (int r = 0; r < 16; r++) out[1024 + lane * 16 + r] = c2[r];
}
int main() {
    float *d; hipMalloc((void **) &d, 2048 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[2048]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; lane++)
        for (int r = 0; r < 16; r++) {
            const int want_row = (4 * (lane >> 4) + (r & 3)) + 16 * (r >> 2), want_col = (lane & 15) + 16 * (r >> 2);
            if ((int) h[lane * 16 + r] != want_row || (int) h[1024 + lane * 16 + r] != want_col) bad++;
        }
    printf("16x16x4_4b layout as k_gemm_mfma4 assumes (register 4 blk + r of lane l = block blk, row 4 (l >> 4) + r, column l & 15): %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    for (int lane : { 0, 5, 16, 37 }) {
        printf("lane %2d rows  :", lane); for (int r = 0; r < 16; r++) printf(" %3.0f", h[lane * 16 + r]); printf("\n");
        printf("lane %2d cols  :", lane); for (int r = 0; r < 16; r++) printf(" %3.0f", h[1024 + lane * 16 + r]); printf("\n");
    }
    return 0;
}
