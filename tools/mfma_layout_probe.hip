// mfma_layout_probe.hip -- register layout of v_mfma_f32_32x32x4_2b_f16 on gfx950 (bring-up tooling for k_gemm_mfma16):
// A(block g, row m, k) = (k == 0) * (m + 64 g), B(block g, col n, k) = (k == 0) * (n == N0 ? 1 : 0) ... prints, for lane N0
// and lane N0 + 32, which (block, row) each of the 32 result registers holds.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x32v __attribute__((ext_vector_type(32)));
__global__ void k(float *out) {
    const int lane = threadIdx.x, m = lane & 31, g = lane >> 5;
    h4 a = { (_Float16) (float) (m + 64 * g), 0, 0, 0 };
    h4 b = { (_Float16) 1.0f, 0, 0, 0 };                      // every column: D[blk][row][col] = row + 64 blk
    f32x32v c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x4f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 32; r++) out[lane * 32 + r] = c[r];
    // column check: A = 1, B(block g, col n) = n + 64 g
    h4 a2 = { (_Float16) 1.0f, 0, 0, 0 }, b2 = { (_Float16) (float) (m + 64 * g), 0, 0, 0 };
    f32x32v c2 = {};
    c2 = __builtin_amdgcn_mfma_f32_32x32x4f16(a2, b2, c2, 0, 0, 0);
    for (int r = 0; r < 32; r++) out[2048 + lane * 32 + r] = c2[r];
}
int main() {
    float *d; hipMalloc((void **) &d, 4096 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int lane : { 0, 5, 32, 37 }) {
        printf("lane %2d rows  :", lane); for (int r = 0; r < 32; r++) printf(" %3.0f", h[lane * 32 + r]); printf("\n");
        printf("lane %2d cols  :", lane); for (int r = 0; r < 32; r++) printf(" %3.0f", h[2048 + lane * 32 + r]); printf("\n");
    }
    return 0;
}
