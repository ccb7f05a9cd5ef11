// chain_probe.hip -- how fast ONE wave walks the Q4_0 x Q4_0 chain of a row-group (measurement tooling, round 4).
// The decode mat-vecs with few row-groups (wo, w2: 2 waves per CU) and the consumers of k_ffn_engine are a lone wave per SIMD
// executing, per 256-column chunk: 8 v_dot8_i32_i4, 4 v_pk_add_f32, 2 v_mul_f32 and the 8-deep block-ordered FMA chain
// acc = fma(d_w d_a, isum, acc).  In situ that costs ~240 cycles per chunk, far more than its 22 instructions.  Variants, all with
// operands in LDS (one wave, one workgroup per CU, so nothing competes):
//   v0  the production form: v_fmac_f32_dpp (scale product read through the DPP quad broadcast), compiler-scheduled around the asm block
//   v1  scale products broadcast with v_mov_b32_dpp first (off the chain), then a plain v_fmac_f32 chain
//   v2  v1 with the chain of chunk c hand-interleaved with the independent work of chunk c + 1 (one asm block)
//   v3  chain only (8 dependent plain FMAs per chunk, operands in registers): the floor
//   v4  chain only with v_fmac_f32_dpp: the DPP form's dependent latency
// Prints cycles per chunk (s_memtime) for each.  build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/chain_probe.hip -o tools/chain_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int Q> __device__ __forceinline__ float quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), Q | (Q << 2) | (Q << 4) | (Q << 6), 0xF, 0xF, true));
}
#define FMAC8_DPP(ACC, PLO, PHI, Q01, Q23, Q45, Q67)                                               \
    asm("s_nop 1\n\t"                                                                              \
        "v_fmac_f32_dpp %0, %1, %3 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %0, %1, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %0, %1, %5 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %0, %1, %6 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %0, %2, %7 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %0, %2, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %0, %2, %9 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"             \
        "v_fmac_f32_dpp %0, %2, %10 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"                 \
        : "+v"(ACC)                                                                                \
        : "v"(PLO), "v"(PHI), "v"((Q01).x), "v"((Q01).y), "v"((Q23).x), "v"((Q23).y),              \
          "v"((Q45).x), "v"((Q45).y), "v"((Q67).x), "v"((Q67).y))

struct Ops { u32x4 w; f32x2 sw; u32x4 a0, a1; float dl, dh; };
__device__ __forceinline__ void load_ops(Ops &o, const uint8_t *tile, const uint32_t *A, const float *D, int ch, int lane) {
    const int k = lane & 7, tq = lane & 3;
    o.w = *(const u32x4 *) (tile + lane * 16);
    o.sw = *(const f32x2 *) (tile + 1024 + ((lane >> 3) * 8 + tq * 2) * 4);
    const u32x4 *pa = (const u32x4 *) (A + (ch * 8 + k) * 8);
    o.a0 = pa[0]; o.a1 = pa[1];
    o.dl = D[ch * 8 + tq]; o.dh = D[ch * 8 + 4 + tq];
}
#define DOTS(o)                                                                                                   \
    const int i0_ = __builtin_amdgcn_sdot8((int) o.w.x, (int) o.a0.x, 0x4B400000, true), i1_ = __builtin_amdgcn_sdot8((int) o.w.x, (int) o.a0.y, 0x4B400000, true); \
    const int i2_ = __builtin_amdgcn_sdot8((int) o.w.y, (int) o.a0.z, 0x4B400000, true), i3_ = __builtin_amdgcn_sdot8((int) o.w.y, (int) o.a0.w, 0x4B400000, true); \
    const int i4_ = __builtin_amdgcn_sdot8((int) o.w.z, (int) o.a1.x, 0x4B400000, true), i5_ = __builtin_amdgcn_sdot8((int) o.w.z, (int) o.a1.y, 0x4B400000, true); \
    const int i6_ = __builtin_amdgcn_sdot8((int) o.w.w, (int) o.a1.z, 0x4B400000, true), i7_ = __builtin_amdgcn_sdot8((int) o.w.w, (int) o.a1.w, 0x4B400000, true); \
    const f32x2 mg_ = { 12582912.0f, 12582912.0f };                                                                \
    const f32x2 q01_ = f32x2{ __builtin_bit_cast(float, i0_), __builtin_bit_cast(float, i1_) } - mg_;              \
    const f32x2 q23_ = f32x2{ __builtin_bit_cast(float, i2_), __builtin_bit_cast(float, i3_) } - mg_;              \
    const f32x2 q45_ = f32x2{ __builtin_bit_cast(float, i4_), __builtin_bit_cast(float, i5_) } - mg_;              \
    const f32x2 q67_ = f32x2{ __builtin_bit_cast(float, i6_), __builtin_bit_cast(float, i7_) } - mg_;              \
    const float plo_ = o.sw.x * o.dl, phi_ = o.sw.y * o.dh;

__device__ __forceinline__ void chunk_v0(float &acc, const Ops &o) { DOTS(o) FMAC8_DPP(acc, plo_, phi_, q01_, q23_, q45_, q67_); }
__device__ __forceinline__ void chunk_v1(float &acc, const Ops &o) {
    DOTS(o)
    const float s0 = quad_bcast<0>(plo_), s1 = quad_bcast<1>(plo_), s2 = quad_bcast<2>(plo_), s3 = quad_bcast<3>(plo_);
    const float s4 = quad_bcast<0>(phi_), s5 = quad_bcast<1>(phi_), s6 = quad_bcast<2>(phi_), s7 = quad_bcast<3>(phi_);
    acc = fmaf(s0, q01_.x, acc); acc = fmaf(s1, q01_.y, acc); acc = fmaf(s2, q23_.x, acc); acc = fmaf(s3, q23_.y, acc);
    acc = fmaf(s4, q45_.x, acc); acc = fmaf(s5, q45_.y, acc); acc = fmaf(s6, q67_.x, acc); acc = fmaf(s7, q67_.y, acc);
}
// the independent part of a chunk: everything but the chain -> 8 products s[], 8 floats q[]
struct Pre { float s[8], q[8]; };
__device__ __forceinline__ void pre_chunk(Pre &p, const Ops &o) {
    DOTS(o)
    p.s[0] = quad_bcast<0>(plo_); p.s[1] = quad_bcast<1>(plo_); p.s[2] = quad_bcast<2>(plo_); p.s[3] = quad_bcast<3>(plo_);
    p.s[4] = quad_bcast<0>(phi_); p.s[5] = quad_bcast<1>(phi_); p.s[6] = quad_bcast<2>(phi_); p.s[7] = quad_bcast<3>(phi_);
    p.q[0] = q01_.x; p.q[1] = q01_.y; p.q[2] = q23_.x; p.q[3] = q23_.y; p.q[4] = q45_.x; p.q[5] = q45_.y; p.q[6] = q67_.x; p.q[7] = q67_.y;
}
__device__ __forceinline__ void chain8(float &acc, const Pre &p) {
#pragma unroll
    for (int j = 0; j < 8; j++) acc = fmaf(p.s[j], p.q[j], acc);
}

template <int V>
__global__ void __launch_bounds__(64) k_chain(const uint32_t *__restrict__ src, int nchunks, int reps, float *out, unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    // LDS: [nchunks tiles of 1280 B][A: nchunks * 256 B][D: nchunks * 32 B]
    uint8_t *tiles = lds;
    uint32_t *A = (uint32_t *) (lds + (size_t) nchunks * 1280);
    float *D = (float *) (A + nchunks * 64);
    const int lane = threadIdx.x;
    const int total = nchunks * (1280 + 256 + 32) / 4;
    for (int i = lane; i < total; i += 64) {
        uint32_t v = src[i];
        if (i >= nchunks * (1280 + 256) / 4) v = 0x3c000000u + (v & 0x7fffffu);           // activation scales: small positive floats
        else if (i < nchunks * 320 && (i % 320) >= 256) v = 0x3c000000u + (v & 0x7fffffu);  // weight scales
        ((uint32_t *) lds)[i] = v;
    }
    __syncthreads();
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; rep++) {
        if (V == 0 || V == 1) {
            Ops o0, o1;
            load_ops(o0, tiles, A, D, 0, lane);
            for (int c = 0; c < nchunks; c += 2) {
                load_ops(o1, tiles + (c + 1) * 1280, A, D, c + 1, lane);
                if (V == 0) chunk_v0(acc, o0); else chunk_v1(acc, o0);
                load_ops(o0, tiles + ((c + 2) % nchunks) * 1280, A, D, (c + 2) % nchunks, lane);
                if (V == 0) chunk_v0(acc, o1); else chunk_v1(acc, o1);
            }
        } else if (V == 2) {
            // software pipeline: operands two chunks ahead, independent work one chunk ahead of the chain
            Ops o0, o1; Pre p0, p1;
            load_ops(o0, tiles, A, D, 0, lane);
            load_ops(o1, tiles + 1280, A, D, 1, lane);
            pre_chunk(p0, o0);
            for (int c = 0; c < nchunks; c += 2) {
                load_ops(o0, tiles + ((c + 2) % nchunks) * 1280, A, D, (c + 2) % nchunks, lane);
                pre_chunk(p1, o1);
                chain8(acc, p0);
                load_ops(o1, tiles + ((c + 3) % nchunks) * 1280, A, D, (c + 3) % nchunks, lane);
                pre_chunk(p0, o0);
                chain8(acc, p1);
            }
        } else if (V == 3) {
            Pre p; for (int j = 0; j < 8; j++) { p.s[j] = 1.0f + lane * 1e-3f * j; p.q[j] = 0.5f + j; }
            for (int c = 0; c < nchunks; c++) { chain8(acc, p); asm volatile("" : "+v"(acc)); }
        } else if (V == 5) {          // 8 INDEPENDENT FMAs per "chunk": the lone wave's issue rate
            float b[8]; for (int j = 0; j < 8; j++) b[j] = acc + j;
            for (int c = 0; c < nchunks; c++) {
#pragma unroll
                for (int j = 0; j < 8; j++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(b[j]) : "v"(1.0001f + lane), "v"(0.5f));
            }
            for (int j = 0; j < 8; j++) acc += b[j];
        } else if (V == 6) {          // 8 independent v_dot8_i32_i4
            int b[8]; for (int j = 0; j < 8; j++) b[j] = lane + j;
            for (int c = 0; c < nchunks; c++) {
#pragma unroll
                for (int j = 0; j < 8; j++) asm volatile("v_dot8_i32_i4 %0, %1, %2, %0" : "+v"(b[j]) : "v"(0x12345678 + lane), "v"(0x01010101));
            }
            for (int j = 0; j < 8; j++) acc += (float) b[j];
        } else if (V == 7) {          // 8 independent v_fmac_f32_dpp
            float b[8]; for (int j = 0; j < 8; j++) b[j] = acc + j;
            float pl = 1.0f + lane;
            for (int c = 0; c < nchunks; c++) {
#pragma unroll
                for (int j = 0; j < 8; j++) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(b[j]) : "v"(pl), "v"(0.5f));
            }
            for (int j = 0; j < 8; j++) acc += b[j];
        } else if (V == 8) {          // dependent FMA chain with 2 independent dots in every gap
            int b[8]; for (int j = 0; j < 8; j++) b[j] = lane + j;
            for (int c = 0; c < nchunks; c++) {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    asm volatile("v_fmac_f32 %0, %2, %3\n\tv_dot8_i32_i4 %1, %4, %5, %1\n\tv_add_f32 %1, 0xcb400000, %1" : "+v"(acc), "+v"(b[j]) : "v"(1.0001f), "v"(0.5f), "v"(0x12345678 + lane), "v"(0x01010101));
            }
            for (int j = 0; j < 8; j++) acc += (float) b[j];
        } else if (V == 9) {          // dependent DPP-FMA chain with 2 independent instructions in every gap
            int b[8]; for (int j = 0; j < 8; j++) b[j] = lane + j;
            float pl = 1.0f + lane * 1e-3f;
            for (int c = 0; c < nchunks; c++) {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    asm volatile("v_fmac_f32_dpp %0, %2, %3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\tv_dot8_i32_i4 %1, %4, %5, %1\n\tv_add_f32 %1, 0xcb400000, %1" : "+v"(acc), "+v"(b[j]) : "v"(pl), "v"(0.5f), "v"(0x12345678 + lane), "v"(0x01010101));
            }
            for (int j = 0; j < 8; j++) acc += (float) b[j];
        } else {
            f32x2 q01 = { 1.0f, 2.0f }, q23 = { 3.0f, 4.0f }, q45 = { 5.0f, 6.0f }, q67 = { 7.0f, 8.0f };
            float plo = 1.0f + lane * 1e-3f, phi = 0.5f;
            for (int c = 0; c < nchunks; c++) { FMAC8_DPP(acc, plo, phi, q01, q23, q45, q67); asm volatile("" : "+v"(acc)); }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int nchunks = 44, reps = 200;
    const size_t n = (size_t) nchunks * (1280 + 256 + 32) / 4;
    uint32_t *h = (uint32_t *) malloc(n * 4);
    uint32_t x = 12345u;
    for (size_t i = 0; i < n; i++) { x = x * 1664525u + 1013904223u; h[i] = x; }
    uint32_t *d_src; float *d_out; unsigned long long *d_cyc;
    CHECK(hipMalloc((void **) &d_src, n * 4)); CHECK(hipMemcpy(d_src, h, n * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc((void **) &d_out, 256 * 64 * 4)); CHECK(hipMalloc((void **) &d_cyc, 256 * 8));
    const size_t lds = n * 4;
    auto run = [&](const char *label, auto kern, int grid) {
        CHECK(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        for (int it = 0; it < 2; it++) { hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, 0, d_src, nchunks, reps, d_out, d_cyc); CHECK(hipDeviceSynchronize()); }
        unsigned long long c[256]; float o[64];
        CHECK(hipMemcpy(c, d_cyc, grid * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(o, d_out, 256, hipMemcpyDeviceToHost));
        double s = 0; for (int i = 0; i < grid; i++) s += (double) c[i];
        printf("%-58s grid %3d: %7.1f cycles per chunk   (acc[0] = %g)\n", label, grid, s / grid / ((double) nchunks * reps), o[0]);
    };
    for (int grid : { 1 }) {
        run("v0 v_fmac_f32_dpp chain (production)", k_chain<0>, grid);
        run("v1 v_mov_dpp broadcasts + plain v_fmac chain", k_chain<1>, grid);
        run("v2 v1, chain of chunk c beside the dots of chunk c+1", k_chain<2>, grid);
        run("v3 chain only, plain v_fmac (registers)", k_chain<3>, grid);
        run("v4 chain only, v_fmac_f32_dpp (registers)", k_chain<4>, grid);
        run("v5 8 independent v_fmac_f32", k_chain<5>, grid);
        run("v6 8 independent v_dot8_i32_i4", k_chain<6>, grid);
        run("v7 8 independent v_fmac_f32_dpp", k_chain<7>, grid);
        run("v8 8 x {dependent v_fmac, independent dot8 + v_add}", k_chain<8>, grid);
        run("v9 8 x {dependent v_fmac_dpp, independent dot8 + v_add}", k_chain<9>, grid);
    }
    return 0;
}
