// exp_lut_probe.hip -- can the reference's fp16 look-up tables be EVALUATED in fp32 instead of fp64?  (measurement tooling, round 3)
// The decode kernels replace the gathers from table_exp_f16 / table_silu_f16 (ggml.c:2381-2389, used at :7024-7036 and :1956-1963) by
// f2h((float) g((double) h2f(i))) computed in double precision on the device (kcommon.hip.h exp_math_bits / silu_math_bits, checked
// exhaustively against the host tables at load).  fp64 exp costs ~40 double-rate instructions per value; this probe tries an fp32
// evaluation that is EXACT by construction: compute g in fp32 with a known error bound (Cody-Waite reduction, degree-7 polynomial,
// <= ~2 ulp), then round the two ends of a +-W ulp window to fp16 -- if they agree, every float in the window, the reference's
// (float) g((double) x) included, rounds to that half; if not (the value sits near a rounding boundary of the fp16 grid, ~0.1 %
// of inputs) the lane falls back to the fp64 formula.  Reported: mismatches against the host tables over all 65 536 inputs
// (NaN payloads excluded, as in launch_check_lut_math) for the fp64 formula, the fp32 formula alone and fp32 + fallback; the
// fallback rate; and the time per value of each on a soft_max-like input distribution (1 M values in [-20, 0]).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/exp_lut_probe.hip -o tools/exp_lut_probe
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static inline float h2f_host(uint16_t h) {          // bit-exact fp16 -> fp32, host side
    const uint32_t s = (uint32_t) (h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = s;
        else { int sh = 0; uint32_t mm = m; while (!(mm & 0x400u)) { mm <<= 1; sh++; } u = s | ((uint32_t) (113 - sh) << 23) | ((mm & 0x3FFu) << 13); }
    } else if (e == 31) u = s | 0x7F800000u | (m << 13);
    else u = s | ((e + 112u) << 23) | (m << 13);
    float f; __builtin_memcpy(&f, &u, 4); return f;
}
static inline uint16_t f2h_host(float f) {                             // round to nearest even, subnormals kept (as _cvtss_sh)
    uint32_t u; __builtin_memcpy(&u, &f, 4);
    const uint32_t s = (u >> 16) & 0x8000u;
    u &= 0x7FFFFFFFu;
    if (u >= 0x7F800000u) return (uint16_t) (s | (u > 0x7F800000u ? 0x7E00u : 0x7C00u));
    if (u >= 0x477FF000u) return (uint16_t) (s | 0x7C00u);                       // rounds to >= 65520 -> inf
    if (u < 0x33000001u) return (uint16_t) s;                                    // <= 2^-25 -> 0 (tie to even)
    const int e = (int) (u >> 23) - 127;
    uint32_t m = (u & 0x7FFFFFu) | 0x800000u;
    int shift = e >= -14 ? 13 : 13 + (-14 - e);
    uint32_t r = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1u))) r++;
    if (e >= -14) return (uint16_t) (s | (((uint32_t) (e + 15) << 10) + (r - 0x400u)));
    return (uint16_t) (s | r);
}

__device__ __forceinline__ uint16_t f2h_bits(float f) { return __half_as_ushort(__float2half_rn(f)); }
__device__ __forceinline__ float h2f_bits(uint16_t h) { return __half2float(__ushort_as_half(h)); }

// ---- the fp64 formulas of the product (kcommon.hip.h)
__device__ __forceinline__ uint16_t exp_f64_bits(uint16_t h) { return f2h_bits((float) exp((double) h2f_bits(h))); }
__device__ __forceinline__ uint16_t silu_f64_bits(uint16_t h) { const float f = h2f_bits(h); return f2h_bits((float) ((double) f / (1.0 + exp((double) -f)))); }

// ---- fp32: exp(x) with Cody-Waite reduction and a degree-7 polynomial on |r| <= ln2 / 2; returns the value BEFORE the fp16 rounding
__device__ __forceinline__ float exp_f32(float x) {
    const float t = x * 1.44269504088896341f;
    const float n = __builtin_rintf(t);
    float r = __builtin_fmaf(-n, 0.693145751953125f, x);                  // ln2 high part (exact product for |n| < 2^11)
    r = __builtin_fmaf(-n, 1.428606765330187045e-06f, r);                 // ln2 low part
    float p = 1.0f / 5040.0f;
    p = __builtin_fmaf(p, r, 1.0f / 720.0f);
    p = __builtin_fmaf(p, r, 1.0f / 120.0f);
    p = __builtin_fmaf(p, r, 1.0f / 24.0f);
    p = __builtin_fmaf(p, r, 1.0f / 6.0f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    const float nn = fminf(fmaxf(n, -200.0f), 200.0f);
    return ldexpf(p, (int) nn);                                           // (overflow -> inf, underflow -> subnormal / 0: what the reference's float is)
}
// window test: do all floats within +-W ulp of v round to the same fp16?
template <int W>
__device__ __forceinline__ bool window_ok(float v, uint16_t *out) {
    const float lo = v * (1.0f - (float) W * 5.9604645e-8f), hi = v * (1.0f + (float) W * 5.9604645e-8f);
    const uint16_t a = f2h_bits(lo), b = f2h_bits(hi);
    *out = a;
    return a == b;
}
template <bool FALLBACK>
__device__ __forceinline__ uint16_t exp_f32_bits(uint16_t h, uint32_t *fell) {
    const float x = h2f_bits(h);
    uint16_t r;
    if (window_ok<6>(exp_f32(x), &r)) return r;
    if (!FALLBACK) return f2h_bits(exp_f32(x));
    if (fell) atomicAdd(fell, 1u);
    return exp_f64_bits(h);
}
template <bool FALLBACK>
__device__ __forceinline__ uint16_t silu_f32_bits(uint16_t h, uint32_t *fell) {
    const float x = h2f_bits(h);
    const float den = 1.0f + exp_f32(-x);
    const float q = x / den;                                              // correctly rounded fp32 division (no -ffast-math)
    uint16_t r;
    if (window_ok<10>(q, &r)) return r;
    if (!FALLBACK) return f2h_bits(q);
    if (fell) atomicAdd(fell, 1u);
    return silu_f64_bits(h);
}

__global__ void k_check(const uint16_t *T_exp, const uint16_t *T_silu, uint32_t *cnt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 65536u) return;
    const uint16_t h = (uint16_t) i;
    if ((h & 0x7C00u) == 0x7C00u && (h & 0x03FFu)) return;              // NaN inputs
    if (exp_f64_bits(h) != T_exp[i]) atomicAdd(cnt + 0, 1u);
    if (exp_f32_bits<false>(h, nullptr) != T_exp[i]) atomicAdd(cnt + 1, 1u);
    if (exp_f32_bits<true>(h, cnt + 3) != T_exp[i]) atomicAdd(cnt + 2, 1u);
    if (silu_f64_bits(h) != T_silu[i]) atomicAdd(cnt + 4, 1u);
    if (silu_f32_bits<false>(h, nullptr) != T_silu[i]) atomicAdd(cnt + 5, 1u);
    if (silu_f32_bits<true>(h, cnt + 7) != T_silu[i]) atomicAdd(cnt + 6, 1u);
}
template <int MODE>
__global__ void k_time(const uint16_t *in, uint16_t *out, int n, int reps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint16_t h = in[i], acc = 0;
    for (int r = 0; r < reps; r++) {
        uint16_t v;
        if (MODE == 0) v = exp_f64_bits(h);
        else if (MODE == 1) v = exp_f32_bits<true>(h, nullptr);
        else if (MODE == 2) v = silu_f64_bits(h);
        else v = silu_f32_bits<true>(h, nullptr);
        acc ^= v;
        h = (uint16_t) (h + (v & 1u));                                   // (dependent: keeps the compiler from hoisting)
    }
    out[i] = acc;
}

int main() {
    std::vector<uint16_t> Te(65536), Ts(65536);
    for (uint32_t i = 0; i < 65536; i++) {
        const float f = h2f_host((uint16_t) i);
        Te[i] = f2h_host((float) exp((double) f));
        Ts[i] = f2h_host((float) ((double) f / (1.0 + exp((double) -f))));
    }
    uint16_t *dTe, *dTs, *din, *dout; uint32_t *dcnt;
    CHECK(hipMalloc(&dTe, 131072)); CHECK(hipMalloc(&dTs, 131072)); CHECK(hipMalloc(&dcnt, 64)); CHECK(hipMemset(dcnt, 0, 64));
    CHECK(hipMemcpy(dTe, Te.data(), 131072, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dTs, Ts.data(), 131072, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_check, dim3(256), dim3(256), 0, 0, dTe, dTs, dcnt);
    uint32_t c[8]; CHECK(hipMemcpy(c, dcnt, 32, hipMemcpyDeviceToHost));
    printf("exp : mismatches vs host table over 65 536 inputs: fp64 formula %u | fp32 alone %u | fp32 + window fallback %u (fell back on %u inputs)\n", c[0], c[1], c[2], c[3]);
    printf("silu: mismatches vs host table over 65 536 inputs: fp64 formula %u | fp32 alone %u | fp32 + window fallback %u (fell back on %u inputs)\n", c[4], c[5], c[6], c[7]);
    const int n = 1 << 20, reps = 16;
    std::vector<uint16_t> in(n);
    srand(7);
    for (int i = 0; i < n; i++) in[i] = f2h_host(-20.0f * (float) rand() / (float) RAND_MAX);
    CHECK(hipMalloc(&din, n * 2)); CHECK(hipMalloc(&dout, n * 2)); CHECK(hipMemcpy(din, in.data(), n * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char *names[4] = { "exp  fp64", "exp  fp32 + fallback", "silu fp64", "silu fp32 + fallback" };
    for (int mode = 0; mode < 4; mode++) {
        for (int it = 0; it < 2; it++) {
            CHECK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(k_time<0>, dim3(n / 256), dim3(256), 0, 0, din, dout, n, reps);
            if (mode == 1) hipLaunchKernelGGL(k_time<1>, dim3(n / 256), dim3(256), 0, 0, din, dout, n, reps);
            if (mode == 2) hipLaunchKernelGGL(k_time<2>, dim3(n / 256), dim3(256), 0, 0, din, dout, n, reps);
            if (mode == 3) hipLaunchKernelGGL(k_time<3>, dim3(n / 256), dim3(256), 0, 0, din, dout, n, reps);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (it == 1) printf("%-22s %8.3f ms for %d x %d values = %.1f G values/s chip-wide = %.1f ns per wave-wide evaluation and SIMD\n", names[mode], ms, n, reps,
                                (double) n * reps / (ms * 1e6), ms * 1e6 / ((double) n * reps / 64.0) * 1024.0);
        }
    }
    return 0;
}
