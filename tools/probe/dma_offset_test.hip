// dma_offset_test.hip — does the immediate offset of global_load_lds_dwordx4 move the LDS destination too (M0 + offset + lane * 16)?
// If so, the contiguous pieces of a wave's share of a tile need one M0 write per 4 KiB instead of one per KiB.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/dma_offset_test tools/probe/dma_offset_test.hip && tools/probe/dma_offset_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const uint8_t *src, uint32_t *out) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[8192];
    for (int i = threadIdx.x; i < 2048; i += 64) ((uint32_t *)lds)[i] = 0xdeadbeefu;
    __syncthreads();
    const uint32_t dst = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)lds;
    const uint32_t voff = threadIdx.x * 16u;
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2 nt\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:1024 nt\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:2048 nt\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:3072 nt\n\t"
        "s_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
        : "=&s"(keep)
        : "v"(voff), "s"(src), "s"(dst)
        : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = ((uint32_t *)lds)[i];
}
int main() {
    std::vector<uint32_t> h(2048), o(2048);
    for (int i = 0; i < 2048; i++) h[i] = i * 2654435761u;
    uint8_t *d; uint32_t *dout;
    hipMalloc(&d, 8192); hipMalloc(&dout, 8192);
    hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout);
    hipMemcpy(o.data(), dout, 8192, hipMemcpyDeviceToHost);
    int ok4 = 1, same0 = 1;
    for (int i = 0; i < 1024; i++) ok4 &= o[i] == h[i];
    for (int i = 256; i < 1024; i++) same0 &= o[i] == 0xdeadbeefu;
    printf("offset moves the LDS destination: %s; (pieces 1-3 untouched: %s); lds[0..3]=%08x %08x, expect piece0 %08x piece3 %08x\n", ok4 ? "YES" : "no", same0 ? "yes" : "no", o[0], o[1], h[0], h[768]);
    return 0;
}
