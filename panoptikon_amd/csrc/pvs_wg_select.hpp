// pvs_wg_select.hpp — workgroup-wide selection helpers over keys resident in LDS (gfx950), shared by the device-side page rankings
// (pvs_groups.hip: per-item pages; pvs_rrf_device.hip: the bounded RRF fusion).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

// The kth smallest (1-based) of n 64-bit keys in LDS: eight 8-bit digits from the top, one LDS histogram per digit; the bin that
// holds the rank is found by the first wave (4 bins per lane, a shuffle scan).  Workgroup-wide call; hist: 256 words, misc: 4 words.
// slack > 0: stop at the first digit whose chosen bin holds at most `slack` keys and return the UPPER END of that bin (>= the exact
// key, at most `slack` - 1 keys too many at or below it) — enough for a page threshold, and a pass or two fewer.
__device__ static inline unsigned long long wg_radix_kth_u64(const unsigned long long *keys, uint32_t n, uint32_t kth, uint32_t *hist, uint32_t *misc, uint32_t slack = 0) {
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    // The digits every key shares are skipped (the AND and the OR of all keys differ from the first bit in which two keys differ), and
    // a digit whose chosen bin holds ONE key ends the search — that key is fetched by a last scan: three or four passes instead of eight
    // for keys that spread over a few dozen bits (round 4).
    __shared__ unsigned long long s_and, s_or, s_one;
    if (tid == 0) {
        s_and = ~0ull;
        s_or = 0;
    }
    __syncthreads();
    {
        unsigned long long a = ~0ull, o = 0;
        for (uint32_t i = tid; i < n; i += nt) {
            const unsigned long long k = keys[i];
            a &= k;
            o |= k;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            a &= __shfl_xor(a, off, 64);
            o |= __shfl_xor(o, off, 64);
        }
        if ((tid & 63u) == 0) {
            atomicAnd(&s_and, a);
            atomicOr(&s_or, o);
        }
    }
    __syncthreads();
    const unsigned long long diff = s_and ^ s_or;
    if (diff == 0) return s_and;  // (all keys equal)
    const int shift0 = ((63 - __builtin_clzll(diff)) / 8) * 8;
    unsigned long long mask = shift0 >= 56 ? 0ull : ~0ull << (shift0 + 8);
    unsigned long long prefix = s_and & mask;
    uint32_t kk = kth;
    for (int shift = shift0; shift >= 0; shift -= 8) {
        for (uint32_t i = tid; i < 256; i += nt) hist[i] = 0;
        __syncthreads();
        // Keys of one search share their upper bits (distances, aggregates and window keys of a narrow range): in the first passes
        // every key falls into ONE bin and a plain LDS atomic per key serialises — 8k keys: ~10 us per pass, most of a 27-us select.
        // A wave whose live lanes all hold the same digit adds their count with one atomic (round 4); mixed digits keep the plain form.
        for (uint32_t i0 = 0; i0 < n; i0 += nt) {
            const uint32_t i = i0 + tid;
            unsigned long long k = 0;
            bool in = false;
            if (i < n) {
                k = keys[i];
                in = (k & mask) == prefix;
            }
            const uint32_t digit = (uint32_t)(k >> shift) & 255u;
            const unsigned long long act = __builtin_amdgcn_ballot_w64(in);
            if (act) {
                const int first = __builtin_ctzll(act);
                const uint32_t d0 = (uint32_t)__shfl((int)digit, first, 64);
                const unsigned long long same = __builtin_amdgcn_ballot_w64(in && digit == d0);
                if (same == act) {
                    if ((int)(tid & 63u) == first) atomicAdd(&hist[d0], (uint32_t)__popcll(act));
                } else if (in) {
                    atomicAdd(&hist[digit], 1u);
                }
            }
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
                misc[2] = hist[bin];
            }
        }
        __syncthreads();
        prefix |= (unsigned long long)misc[0] << shift;
        mask |= 0xffull << shift;
        kk = misc[1];
        const bool single = misc[2] == 1 && shift > 0;
        const bool near = slack && misc[2] <= slack && shift > 0;
        __syncthreads();
        if (near) return prefix | (~0ull >> (64 - shift));
        if (single) {
            for (uint32_t i = tid; i < n; i += nt) {
                const unsigned long long k = keys[i];
                if ((k & mask) == prefix) s_one = k;  // exactly one key
            }
            __syncthreads();
            return s_one;
        }
    }
    return prefix;
}
