// Probe: does gfx950 execute scalar stores (s_store_dwordx2 + s_dcache_wb)?  Each wave writes (wave id, 0xC0FFEE00 + i) pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint2 *out, int per_wave) {
    const uint32_t w = blockIdx.x * (blockDim.x / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = 0; i < per_wave; i++) {
        const uint64_t data = ((uint64_t)(0xC0FFEE00u + (uint32_t)i) << 32) | w;
        uint2 *dst = out + (size_t)w * per_wave + i;
        const uint64_t a = (uint64_t)(uintptr_t)dst;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        const uint64_t au = ((uint64_t)hi << 32) | lo;
        const uint32_t dlo = __builtin_amdgcn_readfirstlane((uint32_t)data), dhi = __builtin_amdgcn_readfirstlane((uint32_t)(data >> 32));
        const uint64_t du = ((uint64_t)dhi << 32) | dlo;
        asm volatile("s_nop 4\n\ts_store_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" ::"s"(du), "s"(au) : "memory");
    }
    asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}
int main() {
    const int blocks = 512, per_wave = 37, waves = blocks * 4;
    uint2 *d;
    hipMalloc(&d, sizeof(uint2) * waves * per_wave);
    hipMemset(d, 0, sizeof(uint2) * waves * per_wave);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, per_wave);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    uint2 *h = new uint2[waves * per_wave];
    hipMemcpy(h, d, sizeof(uint2) * waves * per_wave, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int w = 0; w < waves; w++)
        for (int i = 0; i < per_wave; i++)
            if (h[w * per_wave + i].x != (uint32_t)w || h[w * per_wave + i].y != 0xC0FFEE00u + i) bad++;
    printf("scalar stores: %s (%ld bad of %d)\n", bad ? "WRONG" : "OK", bad, waves * per_wave);
    return bad != 0;
}
