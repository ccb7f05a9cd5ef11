// prep.hip -- load-time kernels (repack to chain-major tiles, offline quantizer, embedding gather) and the activation preparation
// kernels of multi-row evals (norm / plain / SiLU*up -> Q4_0 activation operands), with their launchers.  Part of libllamahip.so;
// conventions and layouts: decode.hip / DESIGN.md.
#include "kcommon.hip.h"

namespace lh {

// ------------------------------------------------------------------------------------------------
// repack: file-layout Q4_0 rows -> chain-major tiles (load time only)
// ------------------------------------------------------------------------------------------------
// one thread per (row-group, chunk, lane); src = M rows of nb blocks of 20 bytes (unaligned floats)
// gmap: tile group of logical row-group lg is  (lg / 4) * 8 + goff + lg % 4  when gmap != 0 (the
// w1|w3 interleave: every 8 consecutive tile groups hold 32 rows of w1 then the same 32 rows of w3,
// so one 8-wave workgroup owns gate and up of one whole Q4_0 block of the FFN activation), else lg.
__global__ void k_repack_q4(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                            int M, int nb, int ngroups, int nchunks, int gmap, int goff) {
    const long gid = (long) blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long) ngroups * nchunks * 64;
    if (gid >= total) return;
    const int lane = (int) (gid & 63);
    const long tile = gid >> 6;
    const int c = (int) (tile % nchunks);
    const int g = (int) (tile / nchunks);
    const int r = lane >> 3, k = lane & 7;
    const int m = g * 8 + r;
    uint32_t out[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t dw = 0;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int b = c * 8 + 2 * i + half;
            uint32_t e0 = 8, e1 = 8, e2 = 8, e3 = 8;      // q = 8 -> value 0 (padding)
            if (m < M && b < nb) {
                const uint8_t *blk = src + ((size_t) m * nb + b) * 20 + 4;
                const uint32_t lo = blk[k], hi = blk[8 + k];
                e0 = lo & 0xF; e1 = lo >> 4; e2 = hi & 0xF; e3 = hi >> 4;
            }
            const uint32_t packed = e0 | (e1 << 8) | (e2 << 16) | (e3 << 24);
            dw |= packed << (4 * half);
        }
        out[i] = dw;
    }
    const int tg = gmap ? (g >> 2) * 8 + goff + (g & 3) : g;
    uint8_t *t = dst + ((size_t) tg * (nchunks + 1) + c) * TILE_BYTES;
    // stored as signed 4-bit values: (q - 8) & 0xF == q ^ 8, i.e. the dword ^ 0x88888888 -- ready for v_dot8_i32_i4
    u32x4 v = { out[0] ^ 0x88888888u, out[1] ^ 0x88888888u, out[2] ^ 0x88888888u, out[3] ^ 0x88888888u };
    *(u32x4 *) (t + lane * 16) = v;
    // scale of block c*8 + k of row m
    const int bs = c * 8 + k;
    float d = 0.0f;
    if (m < M && bs < nb) {
        const uint8_t *p = src + ((size_t) m * nb + bs) * 20;
        uint32_t bits = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t) p[3] << 24);
        d = __builtin_bit_cast(float, bits);
    }
    // scales of a row are stored as [s0,s4,s1,s5,s2,s6,s3,s7]: lane (r,k) later loads the pair
    // (s[k&3], s[4+(k&3)]), so every QUAD of lanes holds all 8 scales of its row and the per-block
    // scale is one v_mul_f32 with a quad_perm DPP broadcast (no LDS-pipe swizzle)
    *(float *) (t + 1024 + (r * 8 + (k & 3) * 2 + (k >> 2)) * 4) = d;
    if (c == 0) {   // the zero tile closing this row-group (values 0, scales 0)
        uint8_t *z = dst + ((size_t) tg * (nchunks + 1) + nchunks) * TILE_BYTES;
        u32x4 zv = { 0u, 0u, 0u, 0u };
        *(u32x4 *) (z + lane * 16) = zv;
        *(float *) (z + 1024 + lane * 4) = 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// offline quantizer: ggml_quantize_q4_0 (utils.cpp:431-485) -- one thread per 32-element block.
// NOT the runtime activation quantizer: d = amax / 7, id = d ? 1 / d : 0, round half away from zero.
// src: fp32 (f16 = 0) or IEEE half (f16 = 1, widened exactly as ggml_fp16_to_fp32 does).
// ------------------------------------------------------------------------------------------------
__global__ void k_quantize_offline(const void *__restrict__ src, int f16, uint8_t *__restrict__ dst, long nblocks) {
    const long b = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    float x[32];
    if (f16) {
        const uint16_t *p = (const uint16_t *) src + b * 32;
#pragma unroll
        for (int i = 0; i < 32; i++) x[i] = h2f_bits(p[i]);
    } else {
        const float *p = (const float *) src + b * 32;
#pragma unroll
        for (int i = 0; i < 32; i++) x[i] = p[i];
    }
    float amax = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; i++) amax = fmaxf(amax, fabsf(x[i]));
    const float d = amax / 7.0f;
    const float id = d != 0.0f ? 1.0f / d : 0.0f;
    uint8_t *o = dst + b * 20;
    const uint32_t db = __builtin_bit_cast(uint32_t, d);
    o[0] = db & 0xFF; o[1] = (db >> 8) & 0xFF; o[2] = (db >> 16) & 0xFF; o[3] = db >> 24;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const float v0 = x[2 * j] * id, v1 = x[2 * j + 1] * id;
        const int q0 = (int) (int8_t) roundf(v0) + 8, q1 = (int) (int8_t) roundf(v1) + 8;     // C round(): half away from zero
        o[4 + j] = (uint8_t) (q0 | (q1 << 4));
    }
}

// ------------------------------------------------------------------------------------------------
// embedding gather: ggml_get_rows on a Q4_0 matrix (ggml.c:6760-6785 -> dequantize_row_q4_0 :651-684)
// ------------------------------------------------------------------------------------------------
__global__ void k_embed(const int32_t *__restrict__ tokens, const uint8_t *__restrict__ emb,
                        float *__restrict__ x, int d) {
    const int n = blockIdx.x;
    const int tok = tokens[n];
    const uint8_t *row = emb + (size_t) tok * (d / 32) * 20;
    // grid.y slices the row (one dependent round trip per workgroup instead of d/512 per thread)
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < d / 2; i += gridDim.y * blockDim.x) {       // one byte = two elements
        const int b = i >> 4, j = i & 15;
        const uint8_t *blk = row + b * 20;
        const uint32_t bits = blk[0] | (blk[1] << 8) | (blk[2] << 16) | ((uint32_t) blk[3] << 24);
        const float dd = __builtin_bit_cast(float, bits);
        const uint32_t q = blk[4 + j];
        x[(size_t) n * d + 2 * i + 0] = (float) ((int) (q & 0xF) - 8) * dd;
        x[(size_t) n * d + 2 * i + 1] = (float) ((int) (q >> 4) - 8) * dd;
    }
}

// batched decode step: row n = the token at *set->tok_in[n] (same dequantization as k_embed)
// (the step's first launch also gathers the rows' positions into the descriptor and opens the step's epoch: see SeqSet::pos)
__global__ void k_embed_set(SeqSet *set, const uint8_t *__restrict__ emb, float *__restrict__ x, int d, uint32_t *epoch) {
    const int n = blockIdx.x;
    if (blockIdx.y == 0 && threadIdx.x == 0) { set->pos[n] = set->state[n][0]; if (epoch && n == 0) epoch[0] = next_epoch(epoch[0]); }
    const int tok = *set->tok_in[n];
    const uint8_t *row = emb + (size_t) tok * (d / 32) * 20;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < d / 2; i += gridDim.y * blockDim.x) {
        const int b = i >> 4, j = i & 15;
        const uint8_t *blk = row + b * 20;
        const uint32_t bits = blk[0] | (blk[1] << 8) | (blk[2] << 16) | ((uint32_t) blk[3] << 24);
        const float dd = __builtin_bit_cast(float, bits);
        const uint32_t q = blk[4 + j];
        x[(size_t) n * d + 2 * i + 0] = (float) ((int) (q & 0xF) - 8) * dd;
        x[(size_t) n * d + 2 * i + 1] = (float) ((int) (q >> 4) - 8) * dd;
    }
}
// ... and the residual-stream rows of a pipeline stage's set: hid_in[n] -> x row n (gather) or x row n -> hid_out[n]
__global__ void k_rows_set(SeqSet *set, float *__restrict__ x, int d, int gather, uint32_t *epoch) {
    const int n = blockIdx.x;
    if (gather && blockIdx.y == 0 && threadIdx.x == 0) { set->pos[n] = set->state[n][0]; if (epoch && n == 0) epoch[0] = next_epoch(epoch[0]); }
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < d; i += gridDim.y * blockDim.x) {
        if (gather) x[(size_t) n * d + i] = set->hid_in[n][i];
        else set->hid_out[n][i] = x[(size_t) n * d + i];
    }
}

// decode: the embedding row of one token, plus the {sum x, sum x^2} pair (double) the first layer's norm-fused
// mat-vec folds instead of reducing the row itself (PREP_NORMP).  One workgroup; same dequantization.
__global__ void __launch_bounds__(256)
k_embed_part(const int32_t *__restrict__ tokens, const uint8_t *__restrict__ emb, float *__restrict__ x, int d,
             f64x2 *__restrict__ part_out, uint32_t *__restrict__ epoch, uint64_t *__restrict__ xt,
             const uint64_t *token_mb, const int32_t *__restrict__ st, uint32_t *fault, int n_vocab) {
    __shared__ double red[32];
    __shared__ int tok_s;
    // token_mb (first stage of a pipeline with device-side mailboxes): the token arrives as one tagged granule from the last stage's
    // pick kernel (tag: the position it is for, st[0] + 1); one thread polls, bounded
    if (token_mb) {
        if (threadIdx.x == 0) {
            const uint32_t want = make_tag((uint32_t) st[0] + 1u, 0);
            int spins = 0;
            uint64_t g;
            for (;;) {
                g = load_granule_sys(token_mb);
                if ((uint32_t) (g >> 32) == want) break;
                __builtin_amdgcn_s_sleep(16);
                if (poll_give_up(spins, 1 << 23, fault)) break;
            }
            const uint32_t t = (uint32_t) g;
            tok_s = t < (uint32_t) n_vocab ? (int) t : 0;        // (a poll that ran out: the fault word is up, keep the gather in bounds)
        }
        __syncthreads();
    }
    const int tok = token_mb ? tok_s : tokens[0];
    // xt (overlapped decode schedule): the row also leaves as tagged granules, slot 0 of the epoch k_bump_epoch set before this launch
    const uint32_t tag = xt ? make_tag(epoch[0], 0) : 0u;
    const uint8_t *row = emb + (size_t) tok * (d / 32) * 20;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < d / 2; i += blockDim.x) {       // one byte = two elements
        const int b = i >> 4, j = i & 15;
        const uint8_t *blk = row + b * 20;
        const uint32_t bits = blk[0] | (blk[1] << 8) | (blk[2] << 16) | ((uint32_t) blk[3] << 24);
        const float dd = __builtin_bit_cast(float, bits);
        const uint32_t q = blk[4 + j];
        const float v0 = (float) ((int) (q & 0xF) - 8) * dd, v1 = (float) ((int) (q >> 4) - 8) * dd;
        x[2 * i + 0] = v0;
        x[2 * i + 1] = v1;
        if (xt) { store_tagged_agent(xt + 2 * i, __builtin_bit_cast(uint32_t, v0), tag); store_tagged_agent(xt + 2 * i + 1, __builtin_bit_cast(uint32_t, v1), tag); }
        s1 += (double) v0; s1 += (double) v1;
        s2 += (double) v0 * (double) v0; s2 += (double) v1 * (double) v1;
    }
    s1 = block_sum_d(s1, red, 0);
    s2 = block_sum_d(s2, red, 1);
    if (threadIdx.x == 0) part_out[0] = f64x2{ s1, s2 };
    if (epoch && !xt && threadIdx.x == 0) epoch[0] = next_epoch(epoch[0]);       // one forward pass = one epoch of the tagged hand-offs (k_qkv_attn)
}

// grid.x = rows; dynamic LDS = (K + K/32 + 64) floats + 32 doubles
template <int MODE>
__global__ void k_prep_qa(const float *__restrict__ in0, const float *__restrict__ in1, long in_stride, long in1_stride,
                          int K, int Kp, uint32_t *__restrict__ qa_A, float *__restrict__ qa_d,
                          float *__restrict__ y_out, uint8_t *__restrict__ raw_out,
                          const uint16_t *__restrict__ T_silu) {
    extern __shared__ double smem_d[];
    double *red = smem_d;
    float *ybuf = (float *) (smem_d + 32);
    const int n = blockIdx.x;
    make_y<MODE>(ybuf, red, in0 + (size_t) n * in_stride, in1 ? in1 + (size_t) n * in1_stride : nullptr,
                 K, T_silu);
    if (y_out)
        for (int i = threadIdx.x; i < K; i += blockDim.x) y_out[(size_t) n * K + i] = ybuf[pidx(i)];
    quantize_y(ybuf, K, Kp, qa_A + (size_t) n * Kp / 4, qa_d + (size_t) n * (Kp / 32),
               raw_out ? raw_out + (size_t) n * (K / 32) * 20 : nullptr);
}

// Register-resident variant of k_prep_qa (same arithmetic, the production path whenever no fp32 / raw
// side output is wanted): one thread owns one HALF-BLOCK (16 contiguous elements), the two halves of a
// Q4_0 block sit in lanes t and t^1 and exchange through DPP -- no LDS staging of y and no
// one-thread-per-block serial quantizer.  PLAIN and SILU_MUL have no row-wide reduction, so a row is
// spread over gridDim.y workgroups (a 9-row chunk of F = 11008 used to run on 9 workgroups);
// NORM keeps the whole row in one workgroup (blockDim >= K/16, host-checked).
//   grid (rows, slices); block = multiple of 64
template <int MODE>
__global__ void __launch_bounds__(1024)
k_prep_fast(const float *__restrict__ in0, const float *__restrict__ in1, long in_stride, long in1_stride,
            int K, int Kp, uint32_t *__restrict__ qa_A, float *__restrict__ qa_d, const uint16_t *__restrict__ T_silu) {
    __shared__ double red[32];
    const int n = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int nh = K >> 4, nbp = Kp >> 5;
    const int hi = blockIdx.y * nt + tid;                    // half-block index; block = hi >> 1, half = hi & 1
    const bool live = hi < nh;
    const int hc = min(hi, nh - 1);
    const f32x4 *a4 = (const f32x4 *) (in0 + (size_t) n * in_stride) + hc * 4;
    f32x4 xa[4], xb[4];
#pragma unroll
    for (int v = 0; v < 4; v++) xa[v] = a4[v];
    if (MODE == PREP_NORM) {
#pragma unroll
        for (int v = 0; v < 4; v++) xb[v] = ((const f32x4 *) in1)[hc * 4 + v];
    } else if (MODE == PREP_SILU_MUL) {
        const f32x4 *b4 = (const f32x4 *) (in1 + (size_t) n * in1_stride) + hc * 4;
#pragma unroll
        for (int v = 0; v < 4; v++) xb[v] = b4[v];
    }
    uint32_t *A = qa_A + (size_t) n * (Kp / 4);
    float *da = qa_d + (size_t) n * nbp;
    if (MODE == PREP_NORM) {
        // ggml_norm + ggml_mul (ggml.c:5327-5385, :4555)
        double s1 = 0.0;
        if (live) {
#pragma unroll
            for (int v = 0; v < 4; v++) { s1 += (double) xa[v].x; s1 += (double) xa[v].y; s1 += (double) xa[v].z; s1 += (double) xa[v].w; }
        }
        const double mean = block_sum_d(s1, red, 0) / (double) K;
        double s2 = 0.0;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const double v0 = (double) xa[v].x - mean, v1 = (double) xa[v].y - mean;
            const double v2 = (double) xa[v].z - mean, v3 = (double) xa[v].w - mean;
            xa[v].x = (float) v0; xa[v].y = (float) v1; xa[v].z = (float) v2; xa[v].w = (float) v3;
            if (live) { s2 += v0 * v0; s2 += v1 * v1; s2 += v2 * v2; s2 += v3 * v3; }
        }
        const double sum2 = block_sum_d(s2, red, 1);
        const float scale = (float) (1.0 / sqrt(sum2 / (double) K + (double) 1e-5f));
#pragma unroll
        for (int v = 0; v < 4; v++) {
            xa[v].x = xb[v].x * (xa[v].x * scale); xa[v].y = xb[v].y * (xa[v].y * scale);
            xa[v].z = xb[v].z * (xa[v].z * scale); xa[v].w = xb[v].w * (xa[v].w * scale);
        }
    } else if (MODE == PREP_SILU_MUL) {
        // silu through the fp16 table (ggml.c:1956-1963), then ggml_mul (.mm:678-680)
        uint16_t lut[4][4];
#pragma unroll
        for (int v = 0; v < 4; v++) {
            lut[v][0] = T_silu[f2h_bits(xa[v].x)]; lut[v][1] = T_silu[f2h_bits(xa[v].y)];
            lut[v][2] = T_silu[f2h_bits(xa[v].z)]; lut[v][3] = T_silu[f2h_bits(xa[v].w)];
        }
#pragma unroll
        for (int v = 0; v < 4; v++) {
            xa[v].x = h2f_bits(lut[v][0]) * xb[v].x; xa[v].y = h2f_bits(lut[v][1]) * xb[v].y;
            xa[v].z = h2f_bits(lut[v][2]) * xb[v].z; xa[v].w = h2f_bits(lut[v][3]) * xb[v].w;
        }
    }
    // quantize_row_q4_0, AVX2 branch (ggml.c:456-523), two lanes per block
    float amax = 0.0f;
#pragma unroll
    for (int v = 0; v < 4; v++)
        amax = fmaxf(fmaxf(fmaxf(amax, fabsf(xa[v].x)), fabsf(xa[v].y)), fmaxf(fabsf(xa[v].z), fabsf(xa[v].w)));
    amax = fmaxf(amax, dpp_f<DPP_QUAD_XOR1>(amax));          // partner half (lane ^ 1); K/16 is even: both live or both dead
    const float dd = amax / 7.0f;
    const float id = (amax != 0.0f) ? 7.0f / amax : 0.0f;
    uint32_t pr[8];                                          // pair p = elements (2p, 2p+1) of this half -> one 16-bit field
#pragma unroll
    for (int v = 0; v < 4; v++) {
        const uint32_t n0 = (uint32_t) ((int) __builtin_rintf(xa[v].x * id)) & 0xF, n1 = (uint32_t) ((int) __builtin_rintf(xa[v].y * id)) & 0xF;
        const uint32_t n2 = (uint32_t) ((int) __builtin_rintf(xa[v].z * id)) & 0xF, n3 = (uint32_t) ((int) __builtin_rintf(xa[v].w * id)) & 0xF;
        pr[2 * v] = n0 | (n1 << 8);
        pr[2 * v + 1] = n2 | (n3 << 8);
    }
    // chain k of the block = pair k of half 0 (low 16 bits) | pair k of half 1 (high 16 bits)
    const int half = hi & 1, b = hi >> 1, c = b >> 3, j = b & 7;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t other = (uint32_t) __builtin_amdgcn_mov_dpp((int) pr[k], DPP_QUAD_XOR1, 0xF, 0xF, true);
        const uint32_t dw = (half ? (other | (pr[k] << 16)) : (pr[k] | (other << 16))) << (4 * (j & 1));
        if (live && (k >> 2) == half) A[(c * 8 + k) * 8 + j] = dw;          // half 0 stores chains 0..3, half 1 chains 4..7
    }
    if (live && half == 0) da[b] = dd;
    // zero the padded blocks (K not a multiple of 256)
    if (blockIdx.y == 0)
        for (int pb = K / 32 + tid; pb < nbp; pb += nt) {
            const int pc = pb >> 3, pj = pb & 7;
#pragma unroll
            for (int k = 0; k < 8; k++) A[(pc * 8 + k) * 8 + pj] = 0;
            da[pb] = 0.0f;
        }
}


// elementwise add (ggml_add, ggml.c:4425-4476) -- only the debug/dump path uses it; the production
// path fuses the residual add into the GEMV epilogue (same single fp32 add)
__global__ void k_add(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ c, long n) {
    const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c[i] = a[i] + b[i];
}


// counts[0] / counts[1]: entries of the SiLU / exp table (non-NaN inputs) the device formulas do NOT reproduce
__global__ void k_check_lut_math(const uint16_t *__restrict__ T_silu, const uint16_t *__restrict__ T_exp, uint32_t *__restrict__ counts) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 65536u) return;
    const uint16_t h = (uint16_t) i;
    if ((h & 0x7C00u) == 0x7C00u && (h & 0x03FFu)) return;          // NaN inputs: payloads are not compared
    if (silu_math_bits(h) != T_silu[i]) atomicAdd(counts, 1u);
    if (exp_math_bits(h) != T_exp[i]) atomicAdd(counts + 1, 1u);
}
int g_lut_math = 0;          // bit 0: SiLU, bit 1: exp (0x1000 / 0x2000 are OR-ed in per launch by the fault-injection tests) -- set by launch_check_lut_math (process-wide: the tables are the same for every model)
hipError_t launch_check_lut_math(const uint16_t *T_silu, const uint16_t *T_exp, hipStream_t st) {
    static const bool off = getenv("LLAMAHIP_NO_LUT_MATH") != nullptr;       // measurement only
    uint32_t *d_counts = nullptr, h[2] = { 1, 1 };
    hipError_t e = hipMalloc((void **) &d_counts, 8);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d_counts, 0, 8, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_check_lut_math, dim3(256), dim3(256), 0, st, T_silu, T_exp, d_counts);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h, d_counts, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void) hipFree(d_counts);
    if (e != hipSuccess) return e;
    g_lut_math = off ? 0 : ((h[0] == 0 ? 1 : 0) | (h[1] == 0 ? 2 : 0));
    return hipSuccess;
}


hipError_t launch_add(const float *a, const float *b, float *c, long n, hipStream_t st) {
    hipLaunchKernelGGL(k_add, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, a, b, c, n);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_repack(const uint8_t *src_aos, uint8_t *dst, int M, int K, int gmap, int goff, hipStream_t st) {
    const int nb = K / 32, ngroups = (M + 7) / 8, nchunks = (nb + 7) / 8;
    const long total = (long) ngroups * nchunks * 64;
    const int bs = 256;
    hipLaunchKernelGGL(k_repack_q4, dim3((unsigned) ((total + bs - 1) / bs)), dim3(bs), 0, st, src_aos, dst, M, nb, ngroups, nchunks, gmap, goff);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_embed(const int32_t *tokens, const uint8_t *emb, float *x, int d, int N, hipStream_t st) {
    hipLaunchKernelGGL(k_embed, dim3(N, N <= 64 ? (d / 2 + 255) / 256 : 1), dim3(256), 0, st, tokens, emb, x, d);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_embed_set(SeqSet *set, int n, const uint8_t *emb, float *x, int d, hipStream_t st, uint32_t *epoch) {
    hipLaunchKernelGGL(k_embed_set, dim3(n, (d / 2 + 255) / 256), dim3(256), 0, st, set, emb, x, d, epoch);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_rows_set(SeqSet *set, int n, float *x, int d, bool gather, hipStream_t st, uint32_t *epoch) {
    hipLaunchKernelGGL(k_rows_set, dim3(n, (d + 1023) / 1024), dim3(256), 0, st, set, x, d, gather ? 1 : 0, epoch);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}

size_t prep_lds_bytes(int K) { return 32 * sizeof(double) + ((size_t) K + K / 32 + 64) * sizeof(float); }

hipError_t launch_embed_part(const int32_t *token, const uint8_t *emb, float *x, int d, double *part_out, hipStream_t st, uint32_t *epoch, uint64_t *xt,
                             const uint64_t *token_mb, const int32_t *state, uint32_t *fault, int n_vocab) {
    hipLaunchKernelGGL(k_embed_part, dim3(1), dim3(256), 0, st, token, emb, x, d, (f64x2 *) part_out, epoch, xt, token_mb, state, fault, n_vocab);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_prep(int mode, const float *in0, const float *in1, long in_stride, long in1_stride, int K, int N,
                       uint32_t *qa_A, float *qa_d, float *y_out, uint8_t *raw_out, const uint16_t *T_silu,
                       hipStream_t st) {
    const int Kp = (K + 255) / 256 * 256;
    const int nh = K / 16;
    if (!y_out && !raw_out && (mode != PREP_NORM || nh <= 1024)) {
        // register-resident kernel: NORM = one workgroup per row, the others sliced 256 half-blocks per workgroup
        const int nt = mode == PREP_NORM ? (nh + 63) / 64 * 64 : 256;
        const dim3 grid(N, mode == PREP_NORM ? 1 : (nh + nt - 1) / nt);
#define LH_PREPF(MODE) hipLaunchKernelGGL(k_prep_fast<MODE>, grid, dim3(nt), 0, st, in0, in1, in_stride, in1_stride, K, Kp, qa_A, qa_d, T_silu)
        switch (mode) {
            case PREP_PLAIN:    LH_PREPF(PREP_PLAIN); break;
            case PREP_NORM:     LH_PREPF(PREP_NORM); break;
            case PREP_SILU_MUL: LH_PREPF(PREP_SILU_MUL); break;
            default: return hipErrorInvalidValue;
        }
#undef LH_PREPF
        LH_LAUNCH_CHECK();
        return hipSuccess;
    }
    const size_t lds = prep_lds_bytes(K);
#define LH_PREP(MODE) hipLaunchKernelGGL(k_prep_qa<MODE>, dim3(N), dim3(256), lds, st, in0, in1, in_stride, in1_stride, K, Kp, qa_A, qa_d, y_out, raw_out, T_silu)
    switch (mode) {
        case PREP_PLAIN:    LH_PREP(PREP_PLAIN); break;
        case PREP_NORM:     LH_PREP(PREP_NORM); break;
        case PREP_SILU_MUL: LH_PREP(PREP_SILU_MUL); break;
        default: return hipErrorInvalidValue;
    }
#undef LH_PREP
    LH_LAUNCH_CHECK();
    return hipSuccess;
}


hipError_t launch_quantize_offline(const void *src, int f16, uint8_t *dst, long nblocks, hipStream_t st) {
    hipLaunchKernelGGL(k_quantize_offline, dim3((unsigned) ((nblocks + 127) / 128)), dim3(128), 0, st, src, f16, dst, nblocks);
    LH_LAUNCH_CHECK();
    return hipSuccess;
}


hipError_t init_attrs_prep() {
    const int cap = 160 * 1024;          // fused prologues / wide rows need more than the default 64 KB of dynamic LDS
#define LH_ATTR(KERNEL) do { hipError_t e_ = hipFuncSetAttribute((const void *) KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, cap); if (e_ != hipSuccess) return e_; } while (0)
    LH_ATTR(k_prep_qa<PREP_PLAIN>); LH_ATTR(k_prep_qa<PREP_NORM>); LH_ATTR(k_prep_qa<PREP_SILU_MUL>);
#undef LH_ATTR
    return hipSuccess;
}

}  // namespace lh
