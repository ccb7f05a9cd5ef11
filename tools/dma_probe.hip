// dma_probe.hip -- semantics check of the gfx950 LDS-DMA forms k_gemv_dma relies on (measurement / bring-up tooling):
//   global_load_lds_dwordx4 v_off, s[base:base+1]            LDS dst = M0 + lane * 16
//   global_load_lds_dword   v_off, s[base:base+1] offset:1024 LDS dst = M0 + 1024 + lane * 4, global src = base + v_off + 1024
// build: hipcc --offload-arch=gfx950 -O2 tools/dma_probe.hip -o tools/dma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

__global__ void k_probe(const uint8_t *__restrict__ src, uint32_t *__restrict__ out, int nslots) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < (int) (blockDim.x / 64) * nslots * 1280 / 4; i += blockDim.x) ((uint32_t *) lds)[i] = 0xDEADBEEFu;
    __syncthreads();
    uint8_t *ring = lds + wave * nslots * 1280;
    const uint32_t ring_lds = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint8_t *) ring;
    const uint8_t *base = src + (size_t) wave * nslots * 1280;
    const uint32_t voff_n = lane * 16, voff_s = lane * 4;
    for (int c = 0; c < nslots; c++) {
        const uint64_t gb = (uint64_t) (base + (size_t) c * 1280);
        const uint32_t glo = __builtin_amdgcn_readfirstlane((uint32_t) gb), ghi = __builtin_amdgcn_readfirstlane((uint32_t) (gb >> 32));
        const uint32_t dst = __builtin_amdgcn_readfirstlane(ring_lds + c * 1280);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %3\n\t"
                     "s_nop 4\n\t"
                     "global_load_lds_dwordx4 %1, %4 nt\n\t"
                     "global_load_lds_dword %2, %4 offset:1024 nt\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff_n), "v"(voff_s), "s"(dst), "s"(((uint64_t) ghi << 32) | glo) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < (int) (blockDim.x / 64) * nslots * 1280 / 4; i += blockDim.x) out[i] = ((uint32_t *) lds)[i];
}

int main() {
    const int nw = 4, nslots = 6, bytes = nw * nslots * 1280;
    std::vector<uint8_t> h(bytes);
    for (int i = 0; i < bytes; i++) h[i] = (uint8_t) ((i * 2654435761u) >> 13);
    uint8_t *d_src; uint32_t *d_out;
    hipMalloc((void **) &d_src, bytes); hipMalloc((void **) &d_out, bytes);
    hipMemcpy(d_src, h.data(), bytes, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(nw * 64), bytes, 0, d_src, d_out, nslots);
    std::vector<uint8_t> o(bytes);
    hipError_t e = hipMemcpy(o.data(), d_out, bytes, hipMemcpyDeviceToHost);
    int bad = 0, first = -1;
    for (int i = 0; i < bytes; i++) if (o[i] != h[i]) { if (first < 0) first = i; bad++; }
    printf("dma_probe: %s, %d / %d bytes differ (first at %d: slot %d off %d)\n", hipGetErrorString(e), bad, bytes, first, first >= 0 ? first / 1280 : -1, first >= 0 ? first % 1280 : -1);
    if (bad) { printf("  got:"); for (int i = first; i < first + 16 && i < bytes; i++) printf(" %02x", o[i]); printf("\n want:"); for (int i = first; i < first + 16 && i < bytes; i++) printf(" %02x", h[i]); printf("\n"); }
    return bad != 0;
}
