// Probe: scalar stores issued back to back WITHOUT waiting (the data / address SGPRs are rewritten by the next iteration right
// away): are the operands read at issue?  Each wave writes 4096 (wave id, i) pairs; any stale or torn value shows up as a mismatch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint2 *out, int per_wave) {
    const uint32_t w = blockIdx.x * (blockDim.x / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint2 *base = out + (size_t)w * per_wave;
    for (int i = 0; i < per_wave; i++) {
        const uint32_t dlo = __builtin_amdgcn_readfirstlane(w), dhi = __builtin_amdgcn_readfirstlane(0xC0FFEE00u ^ (uint32_t)i * 2654435761u);
        const uint64_t du = ((uint64_t)dhi << 32) | dlo;
        uint2 *dst = base + i;
        asm volatile("s_store_dwordx2 %0, %1, 0x0" ::"s"(du), "s"(dst) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}
int main() {
    const int blocks = 1024, per_wave = 4096, waves = blocks * 4;
    uint2 *d;
    hipMalloc(&d, sizeof(uint2) * (size_t)waves * per_wave);
    long bad = 0;
    for (int rep = 0; rep < 5; rep++) {
        hipMemset(d, 0, sizeof(uint2) * (size_t)waves * per_wave);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, per_wave);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("sync: %s\n", hipGetErrorString(e)); return 2; }
        uint2 *h = new uint2[(size_t)waves * per_wave];
        hipMemcpy(h, d, sizeof(uint2) * (size_t)waves * per_wave, hipMemcpyDeviceToHost);
        for (int w = 0; w < waves; w++)
            for (int i = 0; i < per_wave; i++)
                if (h[(size_t)w * per_wave + i].x != (uint32_t)w || h[(size_t)w * per_wave + i].y != (0xC0FFEE00u ^ (uint32_t)i * 2654435761u)) bad++;
        delete[] h;
    }
    printf("scalar stores without waits: %s (%ld bad of %ld)\n", bad ? "WRONG" : "OK", bad, 5L * waves * per_wave);
    return bad != 0;
}
