// pvs_direct_kernel.hpp — k_direct_topk<DT, METRIC, NQ>: ONE launch for 1..8 queries over a small or medium corpus: exact distance
// of every stored row in the reference's arithmetic (vec_distance_cosine / vec_distance_L2 of sqlite-vec 0.1.9 per row,
// db/sql_functions.rs:105-128, filters/image_embeddings.rs:321-362) and the page `ORDER BY d LIMIT k` (pql/builder.rs:578-582)
// of every query, selected while the rows stream.  Instantiated per element type in pvs_direct_{i8,f16,f32}.hip.
//
// The filter scan (pvs_scan_kernel.hpp) answers a batch with five dependent launches — query prep, pass A, k-th select, pass B,
// pass C — whose fixed cost (~0.1 ms with the host round trip) is most of a search at the reference's own scale (its measured
// index holds 690k vectors; the API's default page is 10 rows; a PQL request carries one to a handful of vector filters,
// pql/builder.rs:638-661).  For a few queries the exact in-order chain runs at HBM speed anyway (k_dense_exact: one lane per row,
// NQ chains per lane), so nothing has to be filtered:
//   * rows stream HBM -> LDS by LDS-DMA, 64 rows x one 256-B k-slab (16 KiB) per wave and stage, two stages per wave, no workgroup
//     barrier in the loop.  WORK IS DEQUEUED, not dealt: after `static_rounds` dealt units a wave takes its next unit (a few
//     consecutive 64-row pairs) from one device-wide counter, one unit ahead of need (the atomic's latency rides under a stage).
//     Round 4 dealt every wave the same share and the workgroups finished 65 / 80 / 117 us (min / mean / max) into a 690k x 768
//     int8 search: the mean is the HBM time, the max was the kernel's;
//   * every wave keeps, per query, the best rows it has seen in an LDS list of 64-bit keys (distance sort key | tie rank or row):
//     a row enters when its key is below the wave's current k-th best (one ballot per 64 rows and query, usually empty), a full
//     list is cut back to its k smallest by a wave-wide rank sort;
//   * at the end the workgroup drops every key above the smallest of its waves' k-th-best bounds, rank-sorts what is left (about
//     2k keys) and publishes its k best per query, sorted, padded with ~0; a ticket;
//   * the LAST min(G, NQ) workgroups to arrive each finalise queries: the first c0 keys of every list go to an LDS pool; the m-th
//     smallest of the lists' d-th keys (m d >= k) bounds the page from above, the handful of pool keys at or below it are
//     rank-sorted into the page; lists whose head is exhausted below the bound hand over more.  Pools that hold too many keys
//     below the bound (massive ties, one workgroup holding the whole page) fall back to round 4's radix select over the pool.
// Semantics are pass C's: (distance, tie rank | row) order, NULL distances never on the finite part of a page; a page that the
// finite distances cannot fill gets flag 3 (the host appends the head of the index's NULL list, pvs_sparse.hip) when that list
// is query-independent, else flag 1 (dense path); int8 rows whose sums leave the closed form's range send the query to the dense
// path (flag 1) as pass C and k_score_i8_direct do.
//
// Roofline: HBM (rows x row pitch per launch, whatever NQ) for large N, launch + merge latency for small N.
#pragma once
#include <hip/hip_ext.h>

#include "pvs_kernels.hpp"
#include "pvs_lds_dma.hpp"
#include <algorithm>
#include <atomic>

namespace pvs_direct {

struct DirectK {
    const uint8_t *rows;
    const float *norm2;
    const void *qexact;  // [nq][dim] int8 codes (int8 rows) or f32
    const QInfo *qinfo;  // [nq]
    const uint32_t *trank, *tinv;  // second sort key (pvs_index_set_order_keys) or nullptr
    const int64_t *ids;
    const uint8_t *mask;          // candidate mask or nullptr
    unsigned long long *wg_keys;  // [nq][grid][k]: a workgroup's best keys of a query, ascending, padded with ~0
    uint32_t *wg_cnt;             // [nq][grid]
    uint32_t *ctl;                // control words (CTL_*), zero between launches (the last finaliser resets them)
    int64_t *out_ids;             // [nq][k]
    float *out_dist;
    uint32_t *out_count, *need_dense, *h_flags, *h_seen;  // [nq]
    int64_t *h_out_ids;  // pinned mirror of the pages (or nullptr): [nq][k]
    float *h_out_dist;
    uint32_t *h_out_count;  // [nq]
    uint32_t *h_out_rows;   // [nq][k]
    uint64_t n_rows;
    uint32_t stride, kslabs, dim, qpad_ld, n_pairs, n_waves, k, capw, nq;
    uint32_t unit, n_units, static_rounds, dyn;  // work distribution: pairs per unit, units, dealt rounds, then dequeue?
    int null_ok;
};

// control words in DirectK::ctl.  The dequeue counters: one per wave slot (wave w of every workgroup takes the units
// static_rounds * n_waves + 4 j + w), 256 B apart — one word serves ~88 returning atomics per microsecond (MI355X guide, "dequeue"),
// a 690k x 768 int8 search asks for 133 units per microsecond; the four groups hold one wave of every CU each, so they run dry together
enum { CTL_TICKET = 0, CTL_DONE = 1, CTL_ALLOWED = 3, CTL_TOT = 8, CTL_BAD = 16, CTL_NEXT = 64, CTL_NEXT_STEP = 64, CTL_WORDS = 320 };

constexpr int DIR_WAVE_LDS = 2 * 16384;
constexpr int DIR_RING_LDS = 4 * DIR_WAVE_LDS;  // 128 KiB: two 16-KiB stages per wave
constexpr int DIR_MISC_LDS = 2 * 1024;          // the merges' small arrays (MiscLds)
constexpr int DIR_QSEL_LDS = 160 * 1024 - DIR_RING_LDS - DIR_MISC_LDS;  // 30 KiB: the queries, then 4 x NQ wave lists
constexpr int DIR_LDS = DIR_RING_LDS + DIR_QSEL_LDS + DIR_MISC_LDS;
static_assert(DIR_LDS <= 160 * 1024, "LDS per CU");
// the final merge lives in the (idle) ring: [pool DIR_POOL keys | list cursors 2 KiB | d-th heads 3 KiB (upper words, keys) | candidates FIN_CAP keys | page 256 keys]
constexpr uint32_t DIR_FIN_CAP = 1024;
constexpr uint32_t DIR_POOL = (DIR_RING_LDS - 2048 - 3072 - DIR_FIN_CAP * 8 - 2048) / 8;  // 13,952 keys
constexpr uint32_t DIR_CHUNK = 64;  // keys a list hands over at a time

struct MiscLds {
    uint32_t hist[256];
    unsigned long long wthr[4][8];  // the waves' k-th-best bounds per query
    uint32_t wgn[8];                // keys the workgroup kept per query
    uint32_t misc[8];
    uint32_t ticket, pool_n, more, total, outn, have, allowed, pad;
    unsigned long long kmin, kmax, slot, ubound;
};
static_assert(sizeof(MiscLds) <= DIR_MISC_LDS, "misc LDS");

template <int DT>
__device__ static inline float dir_elem(const uint4 &v, int e) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    if constexpr (DT == PVS_I8)
        return (float)(int)(int8_t)(w[e >> 2] >> ((e & 3) * 8));
    else if constexpr (DT == PVS_F16)
        return h2f((uint16_t)(w[e >> 1] >> ((e & 1) * 16)));
    else
        return __builtin_bit_cast(float, w[e]);
}

__device__ static inline uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    return v;
}
__device__ static inline unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long w = __shfl_xor(v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}
__device__ static inline unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long w = __shfl_xor(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}

// LDS traffic of ONE wave: its instructions reach the LDS in order, so a read sees every earlier write of the same wave once the
// compiler keeps them apart
__device__ static inline void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// (row_scalars_async / row_scalars_wait / dequeue_async / wait_all_and_dequeued: pvs_lds_dma.hpp — values that travel in front of a
//  stage's DMA instructions live in named accumulation registers)

// count of the n (a multiple of 8, padded with ~0) keys of `keys` that are smaller than m: broadcast 16-byte reads, eight keys per step
__device__ static inline uint32_t rank_in(const unsigned long long *keys, uint32_t n, unsigned long long m) {
    const ulonglong2 *rd = (const ulonglong2 *)keys;
    uint32_t r = 0;
    for (uint32_t i = 0; i < n; i += 8) {
        const ulonglong2 x0 = rd[(i >> 1) + 0], x1 = rd[(i >> 1) + 1], x2 = rd[(i >> 1) + 2], x3 = rd[(i >> 1) + 3];
        r += (x0.x < m ? 1u : 0u) + (x0.y < m ? 1u : 0u) + (x1.x < m ? 1u : 0u) + (x1.y < m ? 1u : 0u) + (x2.x < m ? 1u : 0u) + (x2.y < m ? 1u : 0u) +
             (x3.x < m ? 1u : 0u) + (x3.y < m ? 1u : 0u);
    }
    return r;
}
// The rank sort of a list by a group of `gs` threads (thread gi of the group): out[rank] = key for every key of rank < keep.  A
// thread takes FOUR keys per walk over the list (the walk is bound by the latency of its LDS reads: one key per walk made a
// 300-key workgroup merge cost 12 us at k = 100, the same walk serves four keys for the price of one).  n8: n rounded up to 8, pads ~0.
__device__ static inline void rank_sort_into(const unsigned long long *keys, uint32_t n, uint32_t n8, uint32_t gi, uint32_t gs, uint32_t keep,
                                             unsigned long long *out) {
    const ulonglong2 *rd = (const ulonglong2 *)keys;
    if (n <= gs) {  // at most one key per thread: a walk that serves one
        if (gi < n) {
            const unsigned long long m = keys[gi];
            const uint32_t r = rank_in(keys, n8, m);
            if (r < keep) out[r] = m;
        }
        return;
    }
    for (uint32_t b0 = gi; b0 < n; b0 += 4 * gs) {
        unsigned long long m[4];
        uint32_t r[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; j++) m[j] = b0 + j * gs < n ? keys[b0 + j * gs] : ~0ull;
        for (uint32_t i = 0; i < n8; i += 8) {
            const ulonglong2 x0 = rd[(i >> 1) + 0], x1 = rd[(i >> 1) + 1], x2 = rd[(i >> 1) + 2], x3 = rd[(i >> 1) + 3];
#pragma unroll
            for (int j = 0; j < 4; j++)
                r[j] += (x0.x < m[j] ? 1u : 0u) + (x0.y < m[j] ? 1u : 0u) + (x1.x < m[j] ? 1u : 0u) + (x1.y < m[j] ? 1u : 0u) + (x2.x < m[j] ? 1u : 0u) +
                        (x2.y < m[j] ? 1u : 0u) + (x3.x < m[j] ? 1u : 0u) + (x3.y < m[j] ? 1u : 0u);
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (m[j] != ~0ull && r[j] < keep) out[r[j]] = m[j];
    }
}
// the m-th smallest (1-based) of the upper words of 256 keys, by rank: hi[] holds the 256 words (0xffffffff = no key); returns it to
// the thread(s) holding it through *slot (untouched when fewer than m keys exist).  32-bit compares: half the work of the 64-bit
// walk; a bound needs no tie-break (every key with this upper word counts as "at or below").
__device__ static inline void mth_upper_word(const uint32_t *hi, uint32_t mine, uint32_t m, uint32_t *slot) {
    const uint4 *rd = (const uint4 *)hi;
    uint32_t less = 0, le = 0;
#pragma unroll 4
    for (uint32_t i = 0; i < 64; i += 2) {
        const uint4 a = rd[i], b = rd[i + 1];
        less += (a.x < mine ? 1u : 0u) + (a.y < mine ? 1u : 0u) + (a.z < mine ? 1u : 0u) + (a.w < mine ? 1u : 0u) + (b.x < mine ? 1u : 0u) + (b.y < mine ? 1u : 0u) +
                (b.z < mine ? 1u : 0u) + (b.w < mine ? 1u : 0u);
        le += (a.x <= mine ? 1u : 0u) + (a.y <= mine ? 1u : 0u) + (a.z <= mine ? 1u : 0u) + (a.w <= mine ? 1u : 0u) + (b.x <= mine ? 1u : 0u) + (b.y <= mine ? 1u : 0u) +
              (b.z <= mine ? 1u : 0u) + (b.w <= mine ? 1u : 0u);
    }
    if (mine != 0xffffffffu && less < m && m <= le) *slot = mine;
}

// A pivot P = M << 32 with k <= #{keys < P} <= kmax among the keys a wave holds in registers (NKL per lane, ~0 = no key), searched
// on the upper word (the distance): regula falsi on the counts with every other step a bisection (the best distances of a corpus are
// smoothly distributed: five or six counts instead of twenty-five; the bisection steps bound the worst case).  A count is one
// ballot per register.  false: a run of equal distances straddles the window — the caller sorts exactly.
template <int NKL>
__device__ static inline bool wave_pivot(const unsigned long long (&mk)[NKL], uint32_t per, uint32_t k, uint32_t kmax, unsigned long long &P) {
    uint32_t lo = 0xffffffffu, hi = 0, total = 0;
#pragma unroll
    for (int j = 0; j < NKL; j++)
        if ((uint32_t)j < per) {
            const bool real = mk[j] != ~0ull;
            const uint32_t h = (uint32_t)(mk[j] >> 32);
            if (real) {
                lo = min(lo, h);
                hi = max(hi, h);
            }
            total += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(real));
        }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, o, 64));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, o, 64));
    }
    if (total <= kmax) return false;  // (nothing to drop)
    // count(X) = #{keys < X << 32}: cL = count(L) < k and cH = count(H) > kmax throughout (count(lo) = 0; H = hi + 1 counts every
    // key; 64-bit so that hi = 2^32 - 1 needs no special case: +inf distances enter the lists, NaN ones never do)
    unsigned long long L = lo, H = (unsigned long long)hi + 1;
    uint32_t cL = 0, cH = total;
    const uint32_t target = (k + kmax + 1) / 2;
    bool bisect = false;
    while (H - L > 1) {
        unsigned long long M;
        if (bisect) {
            M = L + (H - L) / 2;
        } else {
            M = L + (H - L) * (unsigned long long)(target - cL) / (unsigned long long)(cH - cL);
            M = M <= L ? L + 1 : M >= H ? H - 1 : M;
        }
        bisect = !bisect;
        const unsigned long long X = M << 32;
        uint32_t n = 0;
#pragma unroll
        for (int j = 0; j < NKL; j++)
            if ((uint32_t)j < per) n += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(mk[j] < X));
        if (n < k) {
            L = M;
            cL = n;
        } else if (n > kmax) {
            H = M;
            cH = n;
        } else {
            P = X;
            return true;
        }
    }
    return false;
}

// kth smallest (1-based) of the keys of `keys` that are not ~0, as an offset from kmin: 8-bit digits of (key - kmin) from byte
// `shift0 / 8` down (every key's offset is below 2^(shift0 + 8)).  Workgroup-wide; hist: 256 words, misc: 2 words.
__device__ static inline unsigned long long wg_radix_kth_range(const unsigned long long *keys, uint32_t n, uint32_t kth, unsigned long long kmin, int shift0,
                                                               uint32_t *hist, uint32_t *misc) {
    const uint32_t tid = threadIdx.x;
    unsigned long long prefix = 0, mask = 0;
    uint32_t kk = kth;
    for (int shift = shift0; shift >= 0; shift -= 8) {
        hist[tid] = 0;
        __syncthreads();
        for (uint32_t i0 = 0; i0 < n; i0 += 256) {  // (a wave whose live lanes share the digit adds their count with one atomic: pvs_wg_select.hpp)
            const uint32_t i = i0 + tid;
            unsigned long long o = 0;
            bool in = false;
            if (i < n) {
                const unsigned long long k = keys[i];
                o = k - kmin;
                in = k != ~0ull && (o & mask) == prefix;
            }
            const uint32_t digit = (uint32_t)(o >> shift) & 255u;
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

#ifdef PVS_DIR_PROF  // tuning build: wall clock (100 MHz s_memrealtime) of every workgroup's phases
__device__ unsigned long long g_dir_prof[256][12];
#define DIR_STAMP(i)                                                         \
    do {                                                                     \
        if (threadIdx.x == 0) g_dir_prof[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define DIR_STAMP(i) do { } while (0)
#endif

typedef float v2f __attribute__((ext_vector_type(2)));
// the wave lists, addressed as LDS by type: through a generic pointer hipcc emitted flat loads here (vmcnt waits — on the very
// counter the prefetched stage is in flight on)
typedef volatile __attribute__((address_space(3))) unsigned long long lds_vu64;
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) u64x2 lds_cu64x2;

template <int DT, int METRIC, int NQ>
__global__ __launch_bounds__(256, 1) void k_direct_topk(DirectK a) {
    DIR_STAMP(0);
    constexpr int PER = DT == PVS_I8 ? 16 : DT == PVS_F16 ? 8 : 4;  // components per 16-B chunk
    constexpr int EPS = 16 * PER;                                    // components per 256-B slab row
    constexpr bool PK = DT != PVS_I8 && NQ >= 2;                     // float rows, pairs of queries on the packed f32 pipe (queries interleaved by pairs in LDS)
    constexpr int NP = NQ >= 2 ? NQ / 2 : 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t G = gridDim.x;
    uint8_t *const qsel = smem + DIR_RING_LDS;
    float *const qlds = (float *)qsel;
    const uint32_t qbytes = (uint32_t)NQ * (DT == PVS_I8 ? a.stride : a.qpad_ld * 4u);
    lds_vu64 *const sel_all = (lds_vu64 *)(__attribute__((address_space(3))) uint8_t *)(qsel + qbytes);  // [4 waves][NQ][capw]
    MiscLds &ml = *(MiscLds *)(smem + DIR_RING_LDS + DIR_QSEL_LDS);

    // ---- the wave's row stream starts before anything else: the first stage needs nothing but the row pointer
    const uint32_t gw = blockIdx.x * 4 + wave;
    uint8_t *const wbuf = smem + wave * DIR_WAVE_LDS;
    const uint32_t wlds = lds_addr(wbuf);
    const uint32_t voff = lane * 16u;
    auto uni = [](const uint8_t *p) {
        const uint64_t v = (uint64_t)(uintptr_t)p;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return (const uint8_t *)(uintptr_t)(((uint64_t)hi << 32) | lo);
    };
    // issue cursor: the unit being issued [ipair, iend), its k-slab; the unit after it (dealt, or dequeued one unit ahead)
    uint32_t ipair = 0, iend = 0, islab = 0, round = 0, nxt_unit = 0, issued = 0, consumed = 0, last_pair = 0;
    bool nxt_ok = false, pending = false, more_dyn = a.dyn != 0;
    auto unit_range = [&](uint32_t u) {
        ipair = u * a.unit;
        iend = min(ipair + a.unit, a.n_pairs);
        islab = 0;
    };
    auto fetch_next = [&]() {  // decide the unit after the current one
        round++;
        if (round < a.static_rounds) {
            nxt_unit = gw + round * a.n_waves;
            nxt_ok = nxt_unit < a.n_units;
        } else if (more_dyn) {
            if (lane == 0) dequeue_async(a.ctl + CTL_NEXT + CTL_NEXT_STEP * wave);
            pending = true;  // (resolved behind the next wait_vm<0>: the atomic returns in front of the stage issued after it)
            nxt_ok = false;
        } else {
            nxt_ok = false;
        }
    };
    auto issue_one = [&]() -> bool {  // the next 16-KiB stage of this wave's stream, if there is one
        if (ipair == iend) {
            if (!nxt_ok) return false;
            unit_range(nxt_unit);
            fetch_next();
        }
        const uint8_t *bA = uni(a.rows + (uint64_t)ipair * 64 * a.stride + (uint64_t)islab * 8192);
        const uint8_t *bB = uni(bA + 32ull * a.stride);
        const uint32_t dst = wlds + (issued & 1u) * 16384u;
#pragma unroll
        for (int e = 0; e < 8; e++) dma16(bA + e * 1024, voff, dst + e * 1024);
#pragma unroll
        for (int e = 0; e < 8; e++) dma16(bB + e * 1024, voff, dst + 8192 + e * 1024);
        last_pair = ipair;
        issued++;
        if (++islab == a.kslabs) {
            islab = 0;
            ipair++;
        }
        return true;
    };
    if (gw < a.n_units) {
        unit_range(gw);
        fetch_next();
        (void)issue_one();
    }
    uint32_t cpair = last_pair;  // consume cursor: the pair of the stage being consumed

    // ---- the queries into LDS (their global loads overlap the first stage)
    if constexpr (DT == PVS_I8) {  // codes stay codes: the integer dot product below is exact, the distance its closed form
        int8_t *qb = (int8_t *)qlds;
        for (uint32_t i = tid; i < (uint32_t)NQ * a.stride; i += 256) {
            const uint32_t q = i / a.stride, x = i - q * a.stride;
            qb[i] = (x < a.dim && q < a.nq) ? ((const int8_t *)a.qexact)[(size_t)q * a.dim + x] : (int8_t)0;
        }
    } else if constexpr (PK) {  // [pair][component][2]
        for (uint32_t i = tid; i < (uint32_t)NQ * a.qpad_ld; i += 256) {
            const uint32_t q = i / a.qpad_ld, x = i - q * a.qpad_ld;
            qlds[((size_t)(q >> 1) * a.qpad_ld + x) * 2 + (q & 1)] = (x < a.dim && q < a.nq) ? ((const float *)a.qexact)[(size_t)q * a.dim + x] : 0.f;
        }
    } else {
        for (uint32_t i = tid; i < a.qpad_ld; i += 256) qlds[i] = i < a.dim ? ((const float *)a.qexact)[i] : 0.f;
    }
    float bb[NQ];
    double sqb[NQ];
    bool qnull[NQ];
    bool all_null = true;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        bb[q] = a.qinfo[(uint32_t)q < a.nq ? q : 0].bb;  // (uniform address: scalar loads)
        sqb[q] = __dsqrt_rn((double)bb[q]);
        // a query that makes every distance NULL (zero / NaN-bearing): its page is the head of ALL rows in tie order — the NULL-tail
        // step writes it (flag 3, as pass C says it); none of its rows enters a list
        qnull[q] = (uint32_t)q >= a.nq || pvs_query_all_null(METRIC, bb[q]);
        all_null = all_null && qnull[q];
    }
    __syncthreads();
    DIR_STAMP(1);
    if (all_null && a.null_ok) {  // nothing to scan
        wait_vm<0>();
        if (blockIdx.x == 0 && tid < a.nq) {
            a.need_dense[tid] = 3;
            if (a.h_flags) a.h_flags[tid] = 3;
            if (a.h_seen) a.h_seen[tid] = 0;
            a.out_count[tid] = 0;
        }
        return;
    }

    // ---- the wave's lists: cnt[q] keys, every one below thr[q] once k are held
    uint32_t cnt[NQ];
    unsigned long long thr[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        cnt[q] = 0;
        thr[q] = ~0ull;
    }
    // Make room in a full list.  Only TWO things are needed of a cut: at least the k smallest keys stay, and the bound `t` under
    // which a later row must lie is one that k held keys lie under.  So the fast form does not sort: it bisects on the upper 32
    // bits of the key (the distance) for a pivot P with k <= #{keys < P} <= kmax (every lane holds its <= 8 keys in registers, a
    // count is eight ballots), keeps the keys below P wherever they land and makes P the bound: ~25 rounds of ~40 instructions
    // where the rank sort of round 4 walked capw^2 / 64 compares per lane (k = 100: 256 slots, three or four cuts per wave, 14 us
    // of a 92-us stream).  A cluster of equal distances across the window (duplicates) falls back to the exact rank sort.
    auto cut = [&](uint32_t list, uint32_t &c, unsigned long long &t, bool tight) {  // (the list by number: a pointer parameter loses its LDS address space — flat loads, vmcnt waits)
        lds_vu64 *const sel = sel_all + list * a.capw;
        unsigned long long mk[8];
        const uint32_t per = (a.capw + 63u) >> 6;  // <= 8
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t slot = lane + 64u * (uint32_t)j;
            mk[j] = ((uint32_t)j < per && slot < c) ? sel[slot] : ~0ull;
        }
        const uint32_t kmax = tight ? a.k + max(8u, a.k / 4u) : a.k + (a.capw - 64u - a.k) / 2u;  // (tight: the trim in front of the workgroup merge)
        {
            unsigned long long P;
            if (wave_pivot<8>(mk, per, a.k, kmax, P)) {
                wave_lds_sync();  // (every lane has its keys in registers)
                uint32_t base = 0;
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if ((uint32_t)j < per) {
                        const bool keep = mk[j] < P;
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
                        if (keep) sel[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = mk[j];
                        base += (uint32_t)__popcll(m);
                    }
                wave_lds_sync();
                c = base;
                t = P;
                return;
            }
        }
        // exact form: keep the k smallest, ascending.  A rank sort: the keys are distinct, so a key's slot is the number of smaller
        // ones — every lane walks the list once with broadcast reads (capw^2 / 64 compares per lane)
        uint32_t rk[8];
#pragma unroll
        for (int j = 0; j < 8; j++) rk[j] = 0;
        for (uint32_t i = c + lane; i < ((c + 7u) & ~7u); i += 64) sel[i] = ~0ull;  // (pads compare "not smaller")
        wave_lds_sync();
        lds_cu64x2 *rd = (lds_cu64x2 *)sel;  // (nothing is written while the ranks are counted)
        for (uint32_t i = 0; i < c; i += 8) {  // eight keys per step, four broadcast 16-byte reads in flight
            const u64x2 x0 = rd[(i >> 1) + 0], x1 = rd[(i >> 1) + 1], x2 = rd[(i >> 1) + 2], x3 = rd[(i >> 1) + 3];
#pragma unroll
            for (int j = 0; j < 8; j++)
                if ((uint32_t)j < per) {
                    const unsigned long long m = mk[j];
                    rk[j] += (x0.x < m ? 1u : 0u) + (x0.y < m ? 1u : 0u) + (x1.x < m ? 1u : 0u) + (x1.y < m ? 1u : 0u) + (x2.x < m ? 1u : 0u) +
                             (x2.y < m ? 1u : 0u) + (x3.x < m ? 1u : 0u) + (x3.y < m ? 1u : 0u);
                }
        }
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < 8; j++)
            if ((uint32_t)j < per && mk[j] != ~0ull && rk[j] < a.k) sel[rk[j]] = mk[j];
        wave_lds_sync();
        if (c > a.k) c = a.k;
        if (c == a.k) t = sel[a.k - 1] + 1;  // (the bound is exclusive: a row passes when its key is BELOW it)
    };

    float acc[NQ];
    v2f acc2[NP];
    int acci[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        acc[q] = 0.0f;
        acci[q] = 0;
    }
#pragma unroll
    for (int p = 0; p < NP; p++) acc2[p] = v2f{0.0f, 0.0f};
    uint32_t badq = 0;         // int8: bit q = a row whose sums leave the closed form's range
    uint32_t allowed_rows = 0;  // rows the candidate mask lets through (this lane's)
    const uint32_t row_in = (lane >> 5) * 8192u + (lane & 31) * 256u;
    const uint32_t jx = lane & 15u;
    uint32_t cs = 0;
    const bool need_aa = DT == PVS_I8 || METRIC == PVS_COSINE;
    while (consumed < issued) {
        const uint32_t deq = wait_all_and_dequeued();  // stage `consumed` has landed (and nothing else is outstanding: a dequeued unit number has arrived)
        if (pending) {
            const uint32_t u = a.static_rounds * a.n_waves + 4u * (uint32_t)__builtin_amdgcn_readfirstlane((int)deq) + wave;
            pending = false;
            nxt_unit = u;
            nxt_ok = u < a.n_units;
            more_dyn = nxt_ok;  // (the counter ran past the last unit: nothing left to take)
        }
        // the row scalars of a pair travel in front of the next stage's DMA (in-order return: they are there when 16 DMA
        // instructions are still in flight), issued while the pair's last k-slab is consumed
        const bool last_slab = cs + 1 == a.kslabs;
        const uint64_t row = (uint64_t)cpair * 64 + lane;
        if (last_slab && row < a.n_rows) row_scalars_async(need_aa ? a.norm2 + row : nullptr, a.mask ? a.mask + row : nullptr, a.trank ? a.trank + row : nullptr);
        const bool fed = issue_one();  // streams in while this stage is consumed
        const uint8_t *tile = wbuf + (consumed & 1u) * 16384u + row_in;
        if constexpr (DT == PVS_I8) {
            // int8 rows: v_dot4_i32_i8 against the queries' codes (16 components per chunk in four instructions per query); the
            // reference's f32 chain of integer-valued terms equals the integer sum while it stays below 2^24 (checked per row below)
            const uint8_t *q0b = (const uint8_t *)qlds + (size_t)cs * 256;
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const uint4 v = *(const uint4 *)(tile + ((((uint32_t)c) ^ jx) << 4));
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const uint4 qv = *(const uint4 *)(q0b + (size_t)q * a.stride + c * 16);  // broadcast
                    acci[q] = __builtin_amdgcn_sdot4((int)v.x, (int)qv.x, acci[q], false);
                    acci[q] = __builtin_amdgcn_sdot4((int)v.y, (int)qv.y, acci[q], false);
                    acci[q] = __builtin_amdgcn_sdot4((int)v.z, (int)qv.z, acci[q], false);
                    acci[q] = __builtin_amdgcn_sdot4((int)v.w, (int)qv.w, acci[q], false);
                }
            }
        } else if constexpr (PK) {
            const float *q0p = qlds + (size_t)cs * EPS * 2;
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const uint4 v = *(const uint4 *)(tile + ((((uint32_t)c) ^ jx) << 4));
                float4 qv4[NP][PER / 2];  // pair p, components 2x, 2x+1: (q0 c0, q1 c0, q0 c1, q1 c1)
#pragma unroll
                for (int p = 0; p < NP; p++)
#pragma unroll
                    for (int x = 0; x < PER / 2; x++) qv4[p][x] = *(const float4 *)(q0p + ((size_t)p * a.qpad_ld + c * PER + 2 * x) * 2);  // broadcast
#pragma unroll
                for (int e = 0; e < PER; e++) {
                    const float av = dir_elem<DT>(v, e);
                    const v2f av2 = v2f{av, av};
#pragma unroll
                    for (int p = 0; p < NP; p++) {
                        const float4 &t4 = qv4[p][e >> 1];
                        const v2f qv = (e & 1) == 0 ? v2f{t4.x, t4.y} : v2f{t4.z, t4.w};
                        if (METRIC == PVS_COSINE) {
                            acc2[p] = acc2[p] + av2 * qv;  // (-ffp-contract=off: one rounding per multiply, one per add)
                        } else {
                            const v2f t = av2 - qv;
                            acc2[p] = acc2[p] + t * t;
                        }
                    }
                }
            }
        } else {
            const float *q0 = qlds + (size_t)cs * EPS;
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const uint4 v = *(const uint4 *)(tile + ((((uint32_t)c) ^ jx) << 4));
                float4 qv4[PER / 4];
#pragma unroll
                for (int x = 0; x < PER / 4; x++) qv4[x] = *(const float4 *)(q0 + c * PER + 4 * x);  // broadcast
#pragma unroll
                for (int e = 0; e < PER; e++) {
                    const float av = dir_elem<DT>(v, e);
                    const float4 &t4 = qv4[e >> 2];
                    const float qv = (e & 3) == 0 ? t4.x : (e & 3) == 1 ? t4.y : (e & 3) == 2 ? t4.z : t4.w;
                    if (METRIC == PVS_COSINE) {
                        acc[0] = __fadd_rn(acc[0], __fmul_rn(av, qv));
                    } else {
                        const float t = __fsub_rn(av, qv);
                        acc[0] = __fadd_rn(acc[0], __fmul_rn(t, t));
                    }
                }
            }
        }
        consumed++;
        if (++cs == a.kslabs) {
            uint32_t r_aa, r_mask, r_rank;  // (garbage where the load was not issued: rows beyond the index, absent arrays)
            if (fed)
                row_scalars_wait<16>(r_aa, r_mask, r_rank);
            else
                row_scalars_wait<0>(r_aa, r_mask, r_rank);
            bool valid = row < a.n_rows;
            if (valid && a.mask) {
                valid = (r_mask & 0xffu) != 0;
                allowed_rows += valid ? 1u : 0u;
            }
            const float aa = __builtin_bit_cast(float, r_aa);
            const uint32_t low = a.trank ? r_rank : (uint32_t)row;
            const double sa = need_aa && METRIC == PVS_COSINE ? __dsqrt_rn((double)aa) : 0.0;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                lds_vu64 *const sel = sel_all + (wave * NQ + q) * a.capw;
                unsigned long long key = ~0ull;
                if (valid && !qnull[q]) {
                    float d;
                    if constexpr (DT == PVS_I8) {
                        const float lim = 16777216.0f;
                        if (METRIC == PVS_COSINE) {
                            d = (float)(1.0 - (double)(float)acci[q] / (sa * sqb[q]));  // ref_cosine_finish
                            if (!(aa < lim && bb[q] < lim)) badq |= 1u << q;
                        } else {
                            const double ss = (double)aa + (double)bb[q] - 2.0 * (double)acci[q];
                            d = ref_l2_finish((float)ss);
                            if (!(aa < lim && bb[q] < lim && ss < (double)lim)) badq |= 1u << q;
                        }
                    } else {
                        const float sum = PK ? acc2[q >> 1][q & 1] : acc[q];
                        d = METRIC == PVS_COSINE ? (float)(1.0 - (double)sum / (sa * sqb[q])) : ref_l2_finish(sum);
                    }
                    if (d == d) key = ((unsigned long long)f32_sort_key(d) << 32) | low;  // a NULL distance is never on the finite part of a page
                }
                cnt[q] = __builtin_amdgcn_readfirstlane(cnt[q]);
                if (cnt[q] + 64 > a.capw) cut(wave * NQ + q, cnt[q], thr[q], false);
                const bool pass = key < thr[q];
                const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
                if (m) {
                    if (pass) sel[cnt[q] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = key;
                    cnt[q] += (uint32_t)__popcll(m);
                }
                acc[q] = 0.0f;
                acci[q] = 0;
            }
#pragma unroll
            for (int p = 0; p < NP; p++) acc2[p] = v2f{0.0f, 0.0f};
            cs = 0;
            cpair = last_pair;  // (the stage issued above opens the next pair)
        }
    }
    wait_vm<0>();
    DIR_STAMP(2);
#ifdef PVS_DIR_PROF
    if (lane == 0) atomicMax(&g_dir_prof[blockIdx.x][6], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif

    // ---- the workgroup's k best per query.  Every wave holds >= k keys below its bound thr (~0 while it holds fewer), so the
    // workgroup's k-th best lies below the smallest of the four: keys at or above that bound are dropped, what is left (the keys
    // of the tightest wave plus a few of the others) is rank-sorted by the whole workgroup into the idle ring.
    // (a list that holds far more than k keys — a wave that saw few rows never cut — is trimmed first: the four waves do that side
    //  by side, the workgroup's merge below is quadratic in what they hand it)
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        cnt[q] = __builtin_amdgcn_readfirstlane(cnt[q]);
        if (cnt[q] > 2 * a.k + 16) cut(wave * NQ + q, cnt[q], thr[q], true);
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NQ; q++) ml.wthr[wave][q] = thr[q];
    }
    if (tid < 8) ml.wgn[tid] = 0;
    __syncthreads();
    unsigned long long *const ring = (unsigned long long *)smem;
    const uint32_t wcap = 4 * a.capw;  // slots per query in the ring
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const unsigned long long w0 = ml.wthr[0][q], w1 = ml.wthr[1][q], w2 = ml.wthr[2][q], w3 = ml.wthr[3][q];
        const unsigned long long wt = min(min(w0, w1), min(w2, w3));
        lds_vu64 *const sel = sel_all + (wave * NQ + q) * a.capw;
        const uint32_t c = __builtin_amdgcn_readfirstlane(cnt[q]);
        for (uint32_t i0 = 0; i0 < c; i0 += 64) {
            const uint32_t i = i0 + lane;
            const unsigned long long key = i < c ? sel[i] : ~0ull;
            const bool keep = key < wt;  // (wt <= ~0: pads never pass)
            const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
            if (m) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&ml.wgn[q], (uint32_t)__popcll(m));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (keep) ring[(size_t)q * wcap + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = key;
            }
        }
    }
    __syncthreads();
    unsigned long long *const srt = (unsigned long long *)(qsel + qbytes);  // [NQ][k]: the wave lists are dead now
    // A rank sort costs n^2 compares: at k = 100 the ~400 keys the bound leaves took the workgroup 12 us.  One wave per query first
    // finds a pivot with k .. k + max(8, k / 4) keys below it (wave_pivot: the keys in registers, a count is a ballot per
    // register) and moves those keys to a second buffer; the rank sort then walks ~1.2 k keys.
    constexpr int NKL = NQ <= 2 ? 32 : NQ == 4 ? 16 : 8;  // keys per lane: 4 capw / 64 at the widest lists the instance takes
    unsigned long long *const ring2 = ring + (size_t)NQ * wcap;  // [NQ][wcap2]
    const uint32_t kmaxw = a.k + max(8u, a.k / 4u), wcap2 = (kmaxw + 15u) & ~7u;
    for (uint32_t q = wave; q < (uint32_t)NQ; q += 4) {
        const uint32_t n = ml.wgn[q];
        ml.misc[q] = 0;  // (every lane writes the same word)
        if (n > kmaxw && n <= (uint32_t)NKL * 64u && n * n > 64u * (256u / NQ)) {  // (a short list is sorted faster than it is searched)
            const unsigned long long *const buf = ring + (size_t)q * wcap;
            unsigned long long mk[NKL];
            const uint32_t per = (n + 63u) >> 6;
#pragma unroll
            for (int j = 0; j < NKL; j++) {
                const uint32_t slot = lane + 64u * (uint32_t)j;
                mk[j] = ((uint32_t)j < per && slot < n) ? buf[slot] : ~0ull;
            }
            unsigned long long P;
            if (wave_pivot<NKL>(mk, per, a.k, kmaxw, P)) {
                unsigned long long *const b2 = ring2 + (size_t)q * wcap2;
                uint32_t base = 0;
#pragma unroll
                for (int j = 0; j < NKL; j++)
                    if ((uint32_t)j < per) {
                        const bool keep = mk[j] < P;
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
                        if (keep) b2[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = mk[j];
                        base += (uint32_t)__popcll(m);
                    }
                ml.misc[q] = base;  // keys in the second buffer (>= k)
            }
        }
    }
    __syncthreads();
    {
        constexpr uint32_t GS = 256 / NQ;  // threads per query
        const uint32_t q = tid / GS, gi = tid - q * GS;
        const uint32_t n2 = ml.misc[q];
        const uint32_t n = n2 ? n2 : ml.wgn[q], n8 = (n + 7u) & ~7u;
        unsigned long long *const buf = n2 ? ring2 + (size_t)q * wcap2 : ring + (size_t)q * wcap;
        for (uint32_t i = n + gi; i < n8; i += GS) buf[i] = ~0ull;
        __syncthreads();
        rank_sort_into(buf, n, n8, gi, GS, a.k, srt + (size_t)q * a.k);
    }
    __syncthreads();
    for (uint32_t x = tid; x < a.nq * a.k; x += 256) {
        const uint32_t q = x / a.k, i = x - q * a.k;
        const unsigned long long v = i < ml.wgn[q] ? srt[x] : ~0ull;
        __hip_atomic_store(a.wg_keys + ((size_t)q * G + blockIdx.x) * a.k + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (write-through)
    }
    if (tid < a.nq) __hip_atomic_store(a.wg_cnt + (size_t)tid * G + blockIdx.x, min(ml.wgn[tid], a.k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {
        const unsigned long long anybad = __builtin_amdgcn_ballot_w64(badq != 0);
        if (anybad) {
            uint32_t b = badq;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) b |= (uint32_t)__shfl_xor((int)b, o, 64);
            if (lane == 0)
                for (uint32_t q = 0; q < a.nq; q++)
                    if (b & (1u << q)) atomicOr(a.ctl + CTL_BAD + q, 1u);
        }
        if (a.mask) {
            const uint32_t s = wave_sum_u32(allowed_rows);
            if (lane == 0 && s) atomicAdd(a.ctl + CTL_ALLOWED, s);
        }
    }
    // (the lists went out as write-through stores: once they are acknowledged they are visible device-wide — no L2 write-back
    //  fence in front of the ticket; the finalisers read them with device-scope loads, no L2 invalidate behind it)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    DIR_STAMP(3);
    if (tid == 0) ml.ticket = atomicAdd(a.ctl + CTL_TICKET, 1u);
    __syncthreads();
    const uint32_t n_fin = min(G, a.nq);  // finalisers: the last n_fin workgroups to arrive
    const uint32_t ticket = ml.ticket;
    if (ticket + n_fin < G) return;
    const uint32_t fin_id = ticket - (G - n_fin);
    if (ticket != G - 1) {  // every list must be published: wait for the last arrival (all workgroups are running or done)
        if (tid == 0)
            while (__hip_atomic_load(a.ctl + CTL_TICKET, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < G) __builtin_amdgcn_s_sleep(8);
        __syncthreads();
    }
    DIR_STAMP(4);

    // ---- finalisers: one query each (round robin when there are fewer workgroups than queries)
    unsigned long long *const pool = (unsigned long long *)smem;
    uint32_t *const s_len = (uint32_t *)(smem + (size_t)DIR_POOL * 8);  // [256] keys of list g already in the pool
    uint32_t *const s_cntg = s_len + 256;                               // [256] its length
    unsigned long long *const heads = (unsigned long long *)(smem + (size_t)DIR_POOL * 8 + 2048);  // [256] the lists' d-th keys
    unsigned long long *const cand = heads + 384;                                                   // [DIR_FIN_CAP]
    unsigned long long *const page = cand + DIR_FIN_CAP;                                            // [256]
    auto ld_u32 = [](const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto ld_key = [](const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const uint32_t allowed_total = a.mask ? ld_u32(a.ctl + CTL_ALLOWED) : 0xffffffffu;
    for (uint32_t q = fin_id; q < a.nq; q += n_fin) {
        const unsigned long long *const lists = a.wg_keys + (size_t)q * G * a.k;
        const uint32_t *const lcnt = a.wg_cnt + (size_t)q * G;
        const uint64_t rows_allowed = a.mask ? (uint64_t)allowed_total : a.n_rows;
        const uint32_t want = (uint32_t)((uint64_t)a.k < rows_allowed ? a.k : rows_allowed);
        bool overflow = ld_u32(a.ctl + CTL_BAD + q) != 0;
        const uint32_t c0 = min(a.k, min(DIR_CHUNK, max(8u, 4u * a.k / G + 4u)));
        __syncthreads();
        if (tid == 0) {
            ml.have = 0;
            ml.total = 0;
            ml.outn = 0;
            ml.kmin = ~0ull;
            ml.kmax = 0;
            ml.ubound = ~0ull;
            ml.pool_n = G * c0;
        }
        page[tid] = ~0ull;
        // the first c0 keys of every list (a list holds k / G of the page on average; lists are padded with ~0)
        for (uint32_t x0 = 0; x0 < G * c0; x0 += 256 * 8) {  // (eight device-scope loads per thread in flight, then the LDS writes)
            unsigned long long v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t x = x0 + tid + 256u * (uint32_t)u;
                const uint32_t g = x / c0, i = x - g * c0;
                v[u] = x < G * c0 ? ld_key(lists + (size_t)g * a.k + i) : ~0ull;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t x = x0 + tid + 256u * (uint32_t)u;
                if (x < G * c0) pool[x] = v[u];
            }
        }
        {
            const uint32_t cg = tid < G ? ld_u32(lcnt + tid) : 0;
            s_len[tid] = min(cg, c0);
            s_cntg[tid] = cg;
            const uint32_t hsum = wave_sum_u32(min(cg, c0)), tsum = wave_sum_u32(cg);
            __syncthreads();
            if (lane == 0) {
                atomicAdd(&ml.have, hsum);
                atomicAdd(&ml.total, tsum);
            }
        }
        __syncthreads();
        DIR_STAMP(7);
        const uint32_t kk = min(ml.total, a.k);  // rows on the finite part of the page
        bool done = false;
        unsigned long long T = ~0ull;  // keys below T are the page
        if (kk && !overflow) {
            // the bound: the m-th smallest of the lists' d-th keys — those m lists alone hold m d >= kk keys at or below it.  d is
            // chosen so that m is about half the lists: the bound then sits low in the distribution of the d-th keys and few keys
            // beyond the page lie below it (k = 10: ~10; k = 100 of 256 lists: ~126; k = 256: ~430)
            const uint32_t d = min(c0, max(1u, (2u * kk + G - 1) / G));
            const uint32_t m = (kk + d - 1) / d;
            uint32_t *const hd = (uint32_t *)heads;  // the upper words (distances) of the lists' d-th keys
            const unsigned long long my = tid < G ? pool[(size_t)tid * c0 + d - 1] : ~0ull;
            hd[tid] = (uint32_t)(my >> 32);  // (~0: no such key — a real key's upper word is never 0xffffffff: NaN distances stay out of the lists)
            heads[128 + tid] = my;  // (the keys themselves behind the 1 KiB of upper words, for wave 0)
            if (tid == 0) {
                ml.misc[4] = 0xffffffffu;
                ml.misc[5] = 0;
            }
            __syncthreads();
            // wave 0: a pivot with m .. m + max(2, m / 8) of the 256 d-th keys below it (four keys per lane, a count is four ballots);
            // equal distances across the window: the rank-based form below, by everybody
            if (wave == 0) {
                unsigned long long hk[4];
#pragma unroll
                for (int j = 0; j < 4; j++) hk[j] = heads[128 + lane + 64 * j];
                unsigned long long P;
                if (wave_pivot<4>(hk, 4, m, m + max(2u, m / 8u), P)) {
                    ml.misc[4] = (uint32_t)(P >> 32) - 1u;  // keys below P = keys whose upper word is at most this
                    ml.misc[5] = 1;
                }
            }
            __syncthreads();
            if (!ml.misc[5]) {
                mth_upper_word(hd, (uint32_t)(my >> 32), m, &ml.misc[4]);
                __syncthreads();
            }
            DIR_STAMP(8);
            const uint32_t uw = ml.misc[4];
            const unsigned long long U = uw == 0xffffffffu ? ~0ull : (((unsigned long long)uw << 32) | 0xffffffffull);  // every key with that distance or a smaller one
            if (U != ~0ull) {
                const uint32_t pn = G * c0;
                for (uint32_t i0 = 0; i0 < pn; i0 += 256) {
                    const uint32_t i = i0 + tid;
                    const unsigned long long v = i < pn ? pool[i] : ~0ull;
                    const bool in = v <= U;
                    const unsigned long long mm = __builtin_amdgcn_ballot_w64(in);
                    if (mm) {
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(&ml.outn, (uint32_t)__popcll(mm));
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        const uint32_t at = base + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
                        if (in && at < DIR_FIN_CAP) cand[at] = v;
                    }
                }
                // a list whose head went in entirely may hold more keys at or below the bound
                if (tid < G && c0 < a.k && pool[(size_t)tid * c0 + c0 - 1] <= U) {
                    const unsigned long long *lst = lists + (size_t)tid * a.k;
                    for (uint32_t i = c0; i < a.k; i++) {
                        const unsigned long long v = ld_key(lst + i);
                        if (v > U) break;
                        const uint32_t at = atomicAdd(&ml.outn, 1u);
                        if (at < DIR_FIN_CAP) cand[at] = v;
                    }
                }
                __syncthreads();
                DIR_STAMP(9);
                const uint32_t C = ml.outn;
                if (C <= DIR_FIN_CAP) {  // (C >= kk: the m lists behind the bound hold that many)
                    const uint32_t C8 = (C + 7u) & ~7u;
                    for (uint32_t i = C + tid; i < C8; i += 256) cand[i] = ~0ull;
                    __syncthreads();
                    rank_sort_into(cand, C, C8, tid, 256, kk, page);
                    done = true;
                }
                __syncthreads();
                DIR_STAMP(10);
            }
        }
        if (kk && !overflow && !done) {
            // ---- round 4's form: the pool's k-th key by a radix select, lists extended while they hold keys below it
            if (tid == 0) ml.outn = 0;
            uint32_t scanned = 0;  // pool entries already folded into kmin / kmax
            for (;;) {
                const uint32_t pn = ml.pool_n;
                {
                    unsigned long long lo = ~0ull, hi = 0;
                    for (uint32_t x = scanned + tid; x < pn; x += 256) {
                        const unsigned long long v = pool[x];
                        if (v != ~0ull) {
                            lo = v < lo ? v : lo;
                            hi = v > hi ? v : hi;
                        }
                    }
                    lo = wave_min_u64(lo);
                    hi = wave_max_u64(hi);
                    if (lane == 0 && lo != ~0ull) {
                        atomicMin(&ml.kmin, lo);
                        atomicMax(&ml.kmax, hi);
                    }
                    scanned = pn;
                }
                __syncthreads();
                const uint32_t have = ml.have;  // real keys in the pool
                if (have >= kk) {
                    const unsigned long long kmin = ml.kmin, range = ml.kmax - kmin;
                    const int top = range ? 63 - __builtin_clzll(range) : 0;
                    T = kmin + wg_radix_kth_range(pool, pn, kk, kmin, (top / 8) * 8, ml.hist, ml.misc) + 1;
                }
                if (tid == 0) ml.more = 0;
                __syncthreads();
                // every list whose next unread key is below T hands over DIR_CHUNK more
                if (tid < G) {
                    const uint32_t len = s_len[tid], cg = s_cntg[tid];
                    const unsigned long long *lst = lists + (size_t)tid * a.k;
                    if (len < cg && ld_key(lst + len) < T) {
                        const uint32_t take = min(DIR_CHUNK, cg - len);
                        const uint32_t at = atomicAdd(&ml.pool_n, take);
                        if (at + take <= DIR_POOL) {
                            for (uint32_t i = 0; i < take; i++) pool[at + i] = ld_key(lst + len + i);
                            s_len[tid] = len + take;
                            atomicAdd(&ml.have, take);
                        }
                        atomicAdd(&ml.more, 1u);
                    }
                }
                __syncthreads();
                if (ml.pool_n > DIR_POOL) {
                    overflow = true;
                    break;
                }
                const uint32_t more = ml.more;
                __syncthreads();  // (ml.more is reset in the next round)
                if (more == 0) break;
            }
            if (!overflow) {  // the page: the pool's keys below T (exactly kk of them: the keys are distinct), rank-sorted
                const uint32_t pn = ml.pool_n;
                for (uint32_t i = tid; i < pn; i += 256) {
                    const unsigned long long v = pool[i];
                    if (v < T) {
                        const uint32_t at = atomicAdd(&ml.outn, 1u);
                        if (at < 256) cand[at] = v;
                    }
                }
                __syncthreads();
                const uint32_t kk8 = (kk + 7u) & ~7u;
                for (uint32_t i = kk + tid; i < kk8; i += 256) cand[i] = ~0ull;
                __syncthreads();
                if (tid < kk) {
                    const unsigned long long v = cand[tid];
                    page[rank_in(cand, kk8, v)] = v;
                }
            }
            __syncthreads();
        }
        const float bbq = a.qinfo[q].bb;
        const bool can_complete = a.null_ok && bbq < __builtin_inff() && (METRIC == PVS_L2 || bbq > 0.f);
        const bool q_all_null = a.null_ok && pvs_query_all_null(METRIC, bbq);  // (no row entered a list: the page is all NULL tail)
        const bool tail = kk < want;
        if (overflow || (tail && !can_complete && !q_all_null)) {
            if (tid == 0) {
                a.need_dense[q] = 1;
                if (a.h_flags) a.h_flags[q] = 1;
                if (a.h_seen) a.h_seen[q] = 0;
                a.out_count[q] = 0;
            }
            continue;
        }
        __syncthreads();
        for (uint32_t i = tid; i < a.k; i += 256) {
            const size_t o = (size_t)q * a.k + i;
            if (i < kk) {
                const unsigned long long v = page[i];
                const uint32_t r = (uint32_t)v;
                const uint32_t srow = a.tinv ? a.tinv[r] : r;
                const int64_t id = a.ids[srow];
                const float d = f32_from_sort_key((uint32_t)(v >> 32));
                a.out_ids[o] = id;
                a.out_dist[o] = d;
                if (a.h_out_ids && !tail) {
                    a.h_out_ids[o] = id;
                    a.h_out_dist[o] = d;
                    if (a.h_out_rows) a.h_out_rows[o] = srow;
                }
            } else {
                a.out_ids[o] = -1;
                a.out_dist[o] = __builtin_nanf("");
            }
        }
        if (tid == 0) {
            if (a.h_out_count && !tail) a.h_out_count[q] = kk;
            a.out_count[q] = kk;
            a.need_dense[q] = tail ? 3 : 0;
            if (a.h_seen) a.h_seen[q] = 0;
        }
        if (a.h_flags) {
            // the host polls this query's flag word in pinned memory instead of waiting for the kernel's completion signal; the fence
            // puts the page in front of the flag as far as this stack honours it — the host checks every word of the page itself
            // (search_host: poisoned words), so a late line delays the caller, never misleads it
            __threadfence_system();
            __syncthreads();
            if (tid == 0) a.h_flags[q] = tail ? 3 : 0;
        }
    }
    DIR_STAMP(5);
    __syncthreads();
    if (tid == 0) {
        const uint32_t dn = atomicAdd(a.ctl + CTL_DONE, 1u);
        if (dn == n_fin - 1) {  // every finaliser has read what it needs: ready for the next launch
            for (uint32_t w = 0; w < 32; w++) a.ctl[w] = 0;
            for (uint32_t w = 0; w < 4; w++) a.ctl[CTL_NEXT + CTL_NEXT_STEP * w] = 0;
#ifdef PVS_DIR_PROF
            unsigned long long t0 = ~0ull, mx[6] = {0, 0, 0, 0, 0, 0}, mn[6], sm[6] = {0, 0, 0, 0, 0, 0};
            for (int j = 0; j < 6; j++) mn[j] = ~0ull;
            for (uint32_t g = 0; g < G; g++) t0 = g_dir_prof[g][0] < t0 ? g_dir_prof[g][0] : t0;
            for (uint32_t g = 0; g < G; g++) {
                for (int j = 0; j < 4; j++) {
                    const unsigned long long v = g_dir_prof[g][j] - t0;
                    mx[j] = v > mx[j] ? v : mx[j];
                    mn[j] = v < mn[j] ? v : mn[j];
                    sm[j] += v;
                }
                const unsigned long long v = g_dir_prof[g][6] - t0;  // the workgroup's LAST wave out of the stream
                mx[4] = v > mx[4] ? v : mx[4];
                mn[4] = v < mn[4] ? v : mn[4];
                sm[4] += v;
                g_dir_prof[g][6] = 0;
            }
            printf("dirprof G %u k %u nq %u (x10 ns after the first start; min/avg/max): start %llu/%llu/%llu queries-in-lds %llu/%llu/%llu stream-end %llu/%llu/%llu last-wave-out %llu/%llu/%llu published %llu/%llu/%llu; this finaliser: all-published %llu pool %llu bound %llu candidates %llu ranked %llu pages-written %llu\n",
                   G, a.k, a.nq, mn[0], sm[0] / G, mx[0], mn[1], sm[1] / G, mx[1], mn[2], sm[2] / G, mx[2], mn[4], sm[4] / G, mx[4], mn[3], sm[3] / G, mx[3], g_dir_prof[blockIdx.x][4] - t0,
                   g_dir_prof[blockIdx.x][7] - t0, g_dir_prof[blockIdx.x][8] - t0, g_dir_prof[blockIdx.x][9] - t0, g_dir_prof[blockIdx.x][10] - t0, g_dir_prof[blockIdx.x][5] - t0);
#endif
        }
    }
}

template <int DT, int METRIC, int NQ>
hipError_t direct_launch(const DirectK &k, uint32_t grid, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
    static std::atomic<bool> configured{false};
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_direct_topk<DT, METRIC, NQ>, hipFuncAttributeMaxDynamicSharedMemorySize, DIR_LDS);
        if (e != hipSuccess) return e;
        configured.store(true, std::memory_order_release);
    }
    if (ev_start || ev_stop)
        hipExtLaunchKernelGGL((k_direct_topk<DT, METRIC, NQ>), dim3(grid), dim3(256), DIR_LDS, s, ev_start, ev_stop, 0, k);
    else
        hipLaunchKernelGGL((k_direct_topk<DT, METRIC, NQ>), dim3(grid), dim3(256), DIR_LDS, s, k);
    return hipGetLastError();
}
template <int DT, int NQ>
hipError_t direct_metric(const DirectK &k, int metric, uint32_t grid, hipStream_t s, hipEvent_t a, hipEvent_t b) {
    return metric == PVS_COSINE ? direct_launch<DT, PVS_COSINE, NQ>(k, grid, s, a, b) : direct_launch<DT, PVS_L2, NQ>(k, grid, s, a, b);
}

// per element type (pvs_direct_{i8,f16,f32}.hip): nq_inst in {1, 2, 4, 8} (float rows: up to 4)
hipError_t launch_i8(const DirectK &k, int metric, uint32_t nq_inst, uint32_t grid, hipStream_t s, hipEvent_t a, hipEvent_t b);
hipError_t launch_f16(const DirectK &k, int metric, uint32_t nq_inst, uint32_t grid, hipStream_t s, hipEvent_t a, hipEvent_t b);
hipError_t launch_f32(const DirectK &k, int metric, uint32_t nq_inst, uint32_t grid, hipStream_t s, hipEvent_t a, hipEvent_t b);

}  // namespace pvs_direct
