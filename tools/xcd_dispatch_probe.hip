// xcd_dispatch_probe.hip -- which XCD does workgroup 0 of a launch land on? (measurement tooling)
// Launches a sequence of grids of different sizes on one stream, then interleaved on two streams, then as a captured graph replayed
// several times, and prints HW_REG_XCC_ID of workgroups 0..7 of every launch: does the round-robin over the 8 XCDs restart with
// every launch, or carry on from where the previous launch (of that queue? of any queue?) stopped?
// build: hipcc --offload-arch=gfx950 -O3 tools/xcd_dispatch_probe.hip -o tools/xcd_dispatch_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_xcc(uint32_t *out, int cap) {
    if (threadIdx.x == 0 && (int) blockIdx.x < cap) { uint32_t id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); out[blockIdx.x] = id & 0xf; }
}
int main() {
    const int grids[] = { 8, 8, 1, 8, 3, 8, 16, 256, 1, 1, 344, 8, 1000, 8, 5, 8 };
    const int n = sizeof(grids) / sizeof(grids[0]);
    uint32_t *d; CHECK(hipMalloc(&d, n * 3 * 8 * 4 + 8 * 4 * 64)); CHECK(hipMemset(d, 0xff, n * 3 * 8 * 4 + 8 * 4 * 64));
    hipStream_t s1, s2; CHECK(hipStreamCreate(&s1)); CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    std::vector<uint32_t> h(n * 3 * 8 + 8 * 64);
    auto show = [&](const char *what, int base, int count, const int *g) {
        CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
        printf("%s\n", what);
        for (int i = 0; i < count; i++) { printf("  grid %4d: XCC of workgroups 0..7:", g ? g[i] : 8); for (int j = 0; j < 8; j++) { uint32_t v = h[(base + i) * 8 + j]; if (v > 15) printf("  -"); else printf(" %2u", v); } printf("\n"); }
        return 0;
    };
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_xcc, dim3(grids[i]), dim3(64), 0, s1, d + i * 8, 8);
    show("one stream, launches back to back:", 0, n, grids);
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_xcc, dim3(grids[i]), dim3(64), 0, (i & 1) ? s2 : s1, d + (n + i) * 8, 8);
    show("two streams alternating (even launches stream 1, odd stream 2):", n, n, grids);
    // graph: 8, 1, 8, 1, 8 captured once, replayed 3 times
    hipGraph_t g; hipGraphExec_t ge;
    const int gg[] = { 8, 1, 8, 1, 8, 8, 1, 8, 1, 8, 8, 1, 8, 1, 8 };
    CHECK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    // (a replay writes the same slots: capture three copies with different slots instead)
    for (int r = 0; r < 3; r++) for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k_xcc, dim3(gg[i]), dim3(64), 0, s1, d + (2 * n + r * 5 + i) * 8, 8);
    CHECK(hipStreamEndCapture(s1, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(ge, s1));
    show("one captured graph of 15 launches (8, 1, 8, 1, 8 three times):", 2 * n, 15, gg);
    return 0;
}
