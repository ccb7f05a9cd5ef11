// kv_layout_probe.hip -- how fast the decode attention's K (or V) rows come out of HBM in two cache layouts
// (measurement tooling).  Every launch reads T rows of all H heads of one layer's cache exactly like k_dec_scores does
// (workgroup = (head, 32 keys), a half-wave owns 4 keys and has all their loads in flight), from
//   pos-major  [pos][H][dh]   (the reference's layout: a head's rows are 16 KB apart at 7B)
//   head-major [H][pos][dh]   (a head's rows are contiguous)
// Launches cycle over 40 layer-sized buffers (> the 256 MB Infinity Cache with room to spare), so rows are cold.
// build: hipcc --offload-arch=gfx950 -O3 tools/kv_layout_probe.hip -o tools/kv_layout_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool HEAD_MAJOR>
__global__ void __launch_bounds__(256) k_rows(const float *__restrict__ Kc, int T, int n_ctx, int H, int dh, float *__restrict__ out) {
    const int h = blockIdx.x, t0 = blockIdx.y * 32, tid = threadIdx.x, hw = tid >> 5, l = tid & 31;
    if (t0 >= T) return;
    const int d = H * dh;
    float kv[4][4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int t = min(t0 + hw * 4 + u, T - 1);
        const float *kr = HEAD_MAJOR ? Kc + ((size_t) h * n_ctx + t) * dh : Kc + (size_t) t * d + h * dh;
#pragma unroll
        for (int i = 0; i < 4; i++) kv[u][i] = kr[i * 32 + l];
    }
    float s = 0.0f;
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int i = 0; i < 4; i++) s += kv[u][i];
    if (s == 123.456f) out[h] = s;
}

// the V side: thread = (chunk of T/8 rows, column), 32 rows in flight (k_dec_pv_blk)
template <bool HEAD_MAJOR>
__global__ void __launch_bounds__(256) k_cols(const float *__restrict__ Vc, int T, int n_ctx, int H, int dh, float *__restrict__ out) {
    const int h = blockIdx.x, cb = blockIdx.y, tid = threadIdx.x, c = tid & 31, sub = tid >> 5;
    const int d = H * dh, dc = (T + 7) / 8, ta = dc * sub, t1 = min(ta + dc, T);
    float acc = 0.0f;
    for (int tb = ta; tb < t1; tb += 32) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; u++) {
            const int t = min(tb + u, t1 - 1);
            v[u] = HEAD_MAJOR ? Vc[((size_t) h * n_ctx + t) * dh + cb * 32 + c] : Vc[(size_t) t * d + h * dh + cb * 32 + c];
        }
#pragma unroll
        for (int u = 0; u < 32; u++) acc += v[u];
    }
    if (acc == 123.456f) out[h] = acc;
}

__global__ void __launch_bounds__(256) k_stream(const float4 *__restrict__ p, size_t n16, float *__restrict__ out) {
    float acc = 0.0f;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t) gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char **argv) {
    const int H = 32, dh = 128, n_ctx = argc > 1 ? atoi(argv[1]) : 512, NB = 40;          // usage: kv_layout_probe [n_ctx = 512]
    const size_t layer = (size_t) n_ctx * H * dh;
    float *buf, *d_out;
    CHECK(hipMalloc((void **) &buf, layer * 4 * NB)); CHECK(hipMemset(buf, 0, layer * 4 * NB)); CHECK(hipMalloc((void **) &d_out, 4096));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int T : { n_ctx / 8, n_ctx / 4, n_ctx / 2, n_ctx == 512 ? 512 : n_ctx * 4 / 5 }) {
        for (int var = 0; var < 4; var++) {
            auto launch = [&](int i) {
                const float *p = buf + (size_t) (i % NB) * layer;
                if (var == 0) hipLaunchKernelGGL(k_rows<false>, dim3(H, n_ctx / 32), dim3(256), 0, 0, p, T, n_ctx, H, dh, d_out);
                if (var == 1) hipLaunchKernelGGL(k_rows<true>, dim3(H, n_ctx / 32), dim3(256), 0, 0, p, T, n_ctx, H, dh, d_out);
                if (var == 2) hipLaunchKernelGGL(k_cols<false>, dim3(H, dh / 32), dim3(256), 0, 0, p, T, n_ctx, H, dh, d_out);
                if (var == 3) hipLaunchKernelGGL(k_cols<true>, dim3(H, dh / 32), dim3(256), 0, 0, p, T, n_ctx, H, dh, d_out);
            };
            for (int i = 0; i < NB; i++) launch(i);
            CHECK(hipDeviceSynchronize());
            const int iters = 200;
            CHECK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; i++) launch(i);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const char *names[] = { "K rows (scores pattern), pos-major ", "K rows (scores pattern), head-major", "V columns (V.P pattern), pos-major ", "V columns (V.P pattern), head-major" };
            printf("T = %4d   %s  %6.2f us per launch   (%5.2f MB -> %5.2f TB/s)\n", T, names[var], ms * 1e3 / iters, T * H * dh * 4 / 1e6, T * H * dh * 4.0 / (ms * 1e3 / iters) * 1e-6);
        }
    }
    // the same launches with 140 MB of OTHER memory streamed in between (a layer's weights), out of a 4.4 GB buffer: does the
    // page-table walk of the cache rows show (the translations of 270 MB of K / V cache are evicted by a token's 4 GB)?
    {
        const size_t wbytes = (size_t) 4400 << 20, chunk = (size_t) 140 << 20;
        float4 *w; CHECK(hipMalloc((void **) &w, wbytes)); CHECK(hipMemset(w, 0, wbytes));
        hipEvent_t ea[64], eb[64];
        for (int i = 0; i < 64; i++) { CHECK(hipEventCreate(&ea[i])); CHECK(hipEventCreate(&eb[i])); }
        for (int T : { 256, 512 })
          for (int traffic = 0; traffic < 2; traffic++)
            for (int var = 0; var < 4; var += 2) {
                double tot = 0;
                for (int rep = 0; rep < 3; rep++) {
                    for (int i = 0; i < 31; i++) {
                        if (traffic) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, w + (size_t) i * (chunk / 16), chunk / 16, d_out);
                        const float *p = buf + (size_t) (i % NB) * layer;
                        CHECK(hipEventRecord(ea[i], 0));
                        if (var == 0) hipLaunchKernelGGL(k_rows<false>, dim3(H, n_ctx / 32), dim3(256), 0, 0, p, T, n_ctx, H, dh, d_out);
                        else hipLaunchKernelGGL(k_cols<false>, dim3(H, dh / 32), dim3(256), 0, 0, p, T, n_ctx, H, dh, d_out);
                        CHECK(hipEventRecord(eb[i], 0));
                    }
                    CHECK(hipDeviceSynchronize());
                    if (rep == 2) for (int i = 0; i < 31; i++) { float ms; CHECK(hipEventElapsedTime(&ms, ea[i], eb[i])); tot += ms; }
                }
                printf("T = %3d   %-22s %s: %6.2f us per launch (event to event)\n", T, var == 0 ? "K rows, pos-major" : "V columns, pos-major", traffic ? "after 140 MB of other traffic (4.4 GB per pass)" : "back to back                                  ", tot * 1e3 / 31);
            }
    }
    return 0;
}
