// pvs_wg_select.hpp — workgroup-wide selection helpers over keys resident in LDS (gfx950), shared by the device-side page rankings
// (pvs_groups.hip: per-item pages; pvs_rrf_device.hip: the bounded RRF fusion).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

// The kth smallest (1-based) of n 64-bit keys in LDS: eight 8-bit digits from the top, one LDS histogram per digit; the bin that
// holds the rank is found by the first wave (4 bins per lane, a shuffle scan).  Workgroup-wide call; hist: 256 words, misc: 4 words.
__device__ static inline unsigned long long wg_radix_kth_u64(const unsigned long long *keys, uint32_t n, uint32_t kth, uint32_t *hist, uint32_t *misc) {
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    unsigned long long prefix = 0, mask = 0;
    uint32_t kk = kth;
    for (int shift = 56; shift >= 0; shift -= 8) {
        for (uint32_t i = tid; i < 256; i += nt) hist[i] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += nt) {
            const unsigned long long k = keys[i];
            if ((k & mask) == prefix) atomicAdd(&hist[(uint32_t)(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            const uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            uint32_t v = h0 + h1 + h2 + h3;
            const uint32_t own = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)v, off, 64);
                if ((int)tid >= off) v += up;
            }
            const uint32_t before = v - own;
            if (before < kk && v >= kk) {  // exactly one lane
                uint32_t r = kk - before, bin = 4 * tid;
                if (r > h0) {
                    r -= h0;
                    bin++;
                    if (r > h1) {
                        r -= h1;
                        bin++;
                        if (r > h2) {
                            r -= h2;
                            bin++;
                        }
                    }
                }
                misc[0] = bin;
                misc[1] = r;
            }
        }
        __syncthreads();
        prefix |= (unsigned long long)misc[0] << shift;
        mask |= 0xffull << shift;
        kk = misc[1];
        __syncthreads();
    }
    return prefix;
}
