// pvs_direct.hip — ONE launch for a single query over a small or medium corpus: exact distance of every stored row in the
// reference's arithmetic (vec_distance_cosine / vec_distance_L2 of sqlite-vec 0.1.9 per row, db/sql_functions.rs:105-128,
// filters/image_embeddings.rs:321-362) and the page `ORDER BY d LIMIT k` (pql/builder.rs:578-582), selected while the rows stream.
//
// The filter scan (pvs_scan_kernel.hpp) answers a query with five dependent launches — query prep, pass A, k-th select, pass B,
// pass C — whose fixed cost (~0.1 ms with the host round trip) is most of a search at the reference's own scale (its measured
// index holds 690k vectors; the API's default page is 10 rows).  For ONE query the exact in-order chain runs at HBM speed anyway
// (k_dense_exact: one lane per row), so nothing has to be filtered: this kernel streams the rows like k_dense_exact (LDS-DMA,
// 64 rows per wave, no workgroup barrier in the loop), and
//   * every wave keeps the best rows it has seen in an LDS list of 64-bit keys (distance sort key | tie rank or row): a row enters
//     when its key is below the wave's current k-th best (one ballot per 64 rows, usually empty), a full list is cut back to its
//     k smallest by a wave-wide rank sort (a handful of times per wave);
//   * at the end the four waves' lists merge into the workgroup's k best, the workgroup publishes them (sorted) and takes a
//     ticket; the LAST workgroup merges all lists: the first few keys of each go to an LDS pool, a radix select finds the pool's
//     k-th key T, a list whose next unread key is below T hands over 64 more, until no list has anything below T — exact whatever
//     the placement of the best rows (a run of near-duplicates stored side by side sits in one workgroup), and the page is the
//     pool's k smallest, sorted; written with the same key order as pass C ((distance, tie rank | row), NULL distances never on it).
// A page that the finite distances cannot fill ends in NULL rows: flag 3 (the host appends the head of the index's NULL list,
// pvs_sparse.hip) when that list is query-independent, else flag 1 (dense path) — pass C's rules.
//
// Roofline: HBM (rows x row pitch per launch) for large N, launch + merge latency (~25 us) for small N.
#include <hip/hip_ext.h>

#include "pvs_kernels.hpp"
#include "pvs_lds_dma.hpp"
#include "pvs_wg_select.hpp"
#include <algorithm>
#include <atomic>

namespace {

struct DirectK {
    const uint8_t *rows;
    const float *norm2;
    const void *qexact;  // [dim] int8 codes (q_is_i8) or f32
    const QInfo *qinfo;
    const uint32_t *trank, *tinv;  // second sort key (pvs_index_set_order_keys) or nullptr
    const int64_t *ids;
    const uint8_t *mask;          // candidate mask or nullptr
    unsigned long long *wg_keys;  // [grid][k]: a workgroup's best keys, ascending
    uint32_t *wg_cnt;             // [grid]
    uint32_t *ticket;             // zero between launches (the last workgroup resets it)
    uint32_t *bad;                // int8: raised when a row's sums left the closed form's range (the page goes to the dense path)
    int64_t *out_ids;
    float *out_dist;
    uint32_t *out_count, *need_dense, *h_flags, *h_seen;
    int64_t *h_out_ids;  // pinned mirror of the page (or nullptr)
    float *h_out_dist;
    uint32_t *h_out_count;
    uint32_t *h_out_rows;
    uint64_t n_rows;
    uint32_t stride, kslabs, dim, qpad_ld, n_pairs, n_waves, k, kp, capw;
    int null_ok, q_is_i8;
};

constexpr int DIR_WAVE_LDS = 2 * 16384;
constexpr int DIR_RING_LDS = 4 * DIR_WAVE_LDS;      // 128 KiB: two 16-KiB stages per wave
constexpr int DIR_Q_LDS = 12 * 1024;                // the zero-padded f32 query (dim <= 3072)
constexpr int DIR_SEL_LDS = 16 * 1024;              // four wave lists of <= 512 keys
constexpr int DIR_MISC_LDS = 4 * 1024;              // the merges' small arrays (MiscLds)
constexpr int DIR_LDS = DIR_RING_LDS + DIR_Q_LDS + DIR_SEL_LDS + DIR_MISC_LDS;
static_assert(DIR_LDS <= 160 * 1024, "LDS per CU");
constexpr uint32_t DIR_POOL = (DIR_RING_LDS + DIR_Q_LDS) / 8 - 512;  // keys of the final merge's pool (the ring and the query are free then)
constexpr uint32_t DIR_CHUNK = 64;                  // keys a list hands over at a time

struct MiscLds {
    unsigned long long fin[256];  // the page's keys
    uint32_t hist[256];
    uint32_t misc[8], wcnt[4];
    uint32_t ticket, pool_n, more, total, outn, have;
    unsigned long long kmin, kmax, slot;
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
// kth smallest (1-based) of the keys of `keys` that are not ~0, as an offset from kmin: 8-bit digits of (key - kmin) from byte
// `shift0 / 8` down (every key's offset is below 2^(shift0 + 8)).  Workgroup-wide; hist: 256 words, misc: 2 words.
// A digit whose bin holds ONE key ends the search: that key is fetched by a last scan (three or four passes instead of seven for
// keys that spread over 50 bits).  slot: one 64-bit LDS word.
__device__ static inline unsigned long long wg_radix_kth_range(const unsigned long long *keys, uint32_t n, uint32_t kth, unsigned long long kmin, int shift0,
                                                               uint32_t *hist, uint32_t *misc, unsigned long long *slot) {
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

#ifdef PVS_DIR_PROF  // tuning build: wall clock (100 MHz s_memrealtime) at the phase boundaries of the LAST workgroup
#define DIR_STAMP(i) dp[i] = __builtin_amdgcn_s_memrealtime()
#else
#define DIR_STAMP(i) do { } while (0)
#endif

template <int DT, int METRIC>
__global__ __launch_bounds__(256, 1) void k_direct_topk(DirectK a) {
#ifdef PVS_DIR_PROF
    unsigned long long dp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    DIR_STAMP(0);
    constexpr int PER = DT == PVS_I8 ? 16 : DT == PVS_F16 ? 8 : 4;  // components per 16-B chunk
    constexpr int EPS = 16 * PER;                                    // components per 256-B slab row
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *const qlds = (float *)(smem + DIR_RING_LDS);
    volatile unsigned long long *const sel = (volatile unsigned long long *)(smem + DIR_RING_LDS + DIR_Q_LDS) + (size_t)wave * a.capw;
    if constexpr (DT == PVS_I8) {  // codes stay codes: the integer dot product below is exact, the distance its closed form
        int8_t *qb = (int8_t *)qlds;
        for (uint32_t i = tid; i < a.stride; i += 256) qb[i] = i < a.dim ? ((const int8_t *)a.qexact)[i] : (int8_t)0;
    } else {
        for (uint32_t i = tid; i < a.qpad_ld; i += 256) qlds[i] = i < a.dim ? ((const float *)a.qexact)[i] : 0.f;
    }
    __syncthreads();
    DIR_STAMP(1);
    const float bb = a.qinfo[0].bb;
    // a query that makes every distance NULL (zero / NaN-bearing): the whole page is the head of ALL rows in tie order — the
    // NULL-tail step writes it (flag 3, as pass C says it); nothing to scan
    if (a.null_ok && pvs_query_all_null(METRIC, bb)) {
        if (blockIdx.x == 0 && tid == 0) {
            a.need_dense[0] = 3;
            if (a.h_flags) a.h_flags[0] = 3;
            if (a.h_seen) a.h_seen[0] = 0;
            a.out_count[0] = 0;
        }
        return;
    }

    const uint32_t gw = blockIdx.x * 4 + wave;  // this wave's first pair of row tiles
    const uint32_t my_pairs = gw < a.n_pairs ? (a.n_pairs - gw + a.n_waves - 1) / a.n_waves : 0;
    const uint32_t n_items = my_pairs * a.kslabs;
    uint8_t *const wbuf = smem + wave * DIR_WAVE_LDS;
    const uint32_t wlds = lds_addr(wbuf);
    const uint32_t voff = lane * 16u;
    auto uni = [](const uint8_t *p) {
        const uint64_t v = (uint64_t)(uintptr_t)p;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return (const uint8_t *)(uintptr_t)(((uint64_t)hi << 32) | lo);
    };
    uint32_t ip = 0, is = 0;
    auto issue = [&](uint32_t buf) {
        const uint64_t pair = gw + (uint64_t)ip * a.n_waves;
        const uint8_t *bA = uni(a.rows + pair * 64 * a.stride + (uint64_t)is * 8192);
        const uint8_t *bB = uni(bA + 32ull * a.stride);
        const uint32_t dst = wlds + buf * 16384u;
#pragma unroll
        for (int e = 0; e < 8; e++) dma16(bA + e * 1024, voff, dst + e * 1024);
#pragma unroll
        for (int e = 0; e < 8; e++) dma16(bB + e * 1024, voff, dst + 8192 + e * 1024);
        if (++is == a.kslabs) {
            is = 0;
            ip++;
        }
    };

    // the wave's list: cnt keys, every one below thr once k are held
    uint32_t cnt = 0;
    unsigned long long thr = ~0ull;
    // keep the k smallest, ascending.  A rank sort: the keys are distinct, so a key's slot is the number of smaller ones — every lane
    // holds up to 8 keys in registers and walks the list once with broadcast reads (capw^2 / 64 compares per lane: 1 us at 128
    // slots where the bitonic network's 28 wait-separated steps took 6)
    auto cut = [&]() {
        unsigned long long mk[8];
        uint32_t rk[8];
        const uint32_t per = a.capw >> 6;  // 2, 4 or 8
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t slot = lane + 64u * (uint32_t)j;
            mk[j] = ((uint32_t)j < per && slot < cnt) ? sel[slot] : ~0ull;
            rk[j] = 0;
        }
        for (uint32_t i = cnt + lane; i < ((cnt + 7u) & ~7u); i += 64) sel[i] = ~0ull;  // (pads compare "not smaller")
        wave_lds_sync();
        const ulonglong2 *rd = (const ulonglong2 *)sel;  // (nothing is written while the ranks are counted)
        for (uint32_t i = 0; i < cnt; i += 8) {  // eight keys per step, four broadcast 16-byte reads in flight
            const ulonglong2 x0 = rd[(i >> 1) + 0], x1 = rd[(i >> 1) + 1], x2 = rd[(i >> 1) + 2], x3 = rd[(i >> 1) + 3];
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
        if (cnt > a.k) cnt = a.k;
        if (cnt == a.k) thr = sel[a.k - 1];
    };

    float acc = 0.0f;
    int acci = 0;
    bool bad = false;  // int8: a row whose sums leave the closed form's range
    const uint32_t row_in = (lane >> 5) * 8192u + (lane & 31) * 256u;
    const uint32_t jx = lane & 15u;
    uint32_t cp = 0, cs = 0;
    if (n_items) issue(0);
    for (uint32_t it = 0; it < n_items; it++) {
        wait_vm<0>();
        if (it + 1 < n_items) issue((it + 1) & 1u);
        const uint8_t *tile = wbuf + (it & 1u) * 16384u + row_in;
        if constexpr (DT == PVS_I8) {
            // int8 rows: v_dot4_i32_i8 against the query's codes (16 components per chunk in four instructions instead of sixteen
            // convert / multiply / add triples); the reference's f32 chain of integer-valued terms equals the integer sum while it
            // stays below 2^24 (checked per row below, as pass C and k_score_i8_direct do)
            const uint8_t *q0b = (const uint8_t *)qlds + (size_t)cs * 256;
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const uint4 v = *(const uint4 *)(tile + ((((uint32_t)c) ^ jx) << 4));
                const uint4 qv = *(const uint4 *)(q0b + c * 16);  // broadcast
                acci = __builtin_amdgcn_sdot4((int)v.x, (int)qv.x, acci, false);
                acci = __builtin_amdgcn_sdot4((int)v.y, (int)qv.y, acci, false);
                acci = __builtin_amdgcn_sdot4((int)v.z, (int)qv.z, acci, false);
                acci = __builtin_amdgcn_sdot4((int)v.w, (int)qv.w, acci, false);
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
                    acc = __fadd_rn(acc, __fmul_rn(av, qv));
                } else {
                    const float t = __fsub_rn(av, qv);
                    acc = __fadd_rn(acc, __fmul_rn(t, t));
                }
            }
        }
        }
        if (++cs == a.kslabs) {
            const uint64_t row = (gw + (uint64_t)cp * a.n_waves) * 64 + lane;
            bool valid = row < a.n_rows;
            if (valid && a.mask) valid = a.mask[row] != 0;
            unsigned long long key = ~0ull;
            if (valid) {
                float d;
                if constexpr (DT == PVS_I8) {
                    const float aa = a.norm2[row], lim = 16777216.0f;
                    if (METRIC == PVS_COSINE) {
                        d = ref_cosine_finish((float)acci, aa, bb);
                        bad |= !(aa < lim && bb < lim);
                    } else {
                        const double ss = (double)aa + (double)bb - 2.0 * (double)acci;
                        d = ref_l2_finish((float)ss);
                        bad |= !(aa < lim && bb < lim && ss < (double)lim);
                    }
                } else {
                    const float aa = METRIC == PVS_COSINE ? a.norm2[row] : 0.f;
                    d = METRIC == PVS_COSINE ? ref_cosine_finish(acc, aa, bb) : ref_l2_finish(acc);
                }
                valid = d == d;  // a NULL distance is never on the finite part of a page
                key = ((unsigned long long)f32_sort_key(d) << 32) | (a.trank ? a.trank[row] : (uint32_t)row);
            }
            cnt = __builtin_amdgcn_readfirstlane(cnt);
            if (cnt + 64 > a.capw) cut();
            const bool pass = valid && key < thr;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
            if (m) {
                if (pass) sel[cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = key;
                cnt += (uint32_t)__popcll(m);
            }
            acc = 0.0f;
            acci = 0;
            cs = 0;
            cp++;
        }
    }
    wait_vm<0>();
    DIR_STAMP(2);
    cut();  // ascending, cnt <= k
    DIR_STAMP(3);
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicOr(a.bad, 1u);
    __syncthreads();

    // ---- the workgroup's k best: the four sorted lists merge by rank — a key's slot is its index in its own list plus, for each
    // other list, the number of smaller keys there (a binary search; the keys are distinct) — into the (now idle) ring
    unsigned long long *const mb = (unsigned long long *)smem;
    MiscLds &ml = *(MiscLds *)(smem + DIR_RING_LDS + DIR_Q_LDS + DIR_SEL_LDS);
    if (lane == 0) ml.wcnt[wave] = cnt;
    __syncthreads();
    {
        const unsigned long long *lists = (const unsigned long long *)(smem + DIR_RING_LDS + DIR_Q_LDS);
        for (uint32_t x = tid; x < 4 * a.kp; x += 256) {
            const uint32_t w = x / a.kp, i = x - w * a.kp;
            if (i >= ml.wcnt[w]) continue;
            const unsigned long long key = lists[(size_t)w * a.capw + i];
            uint32_t rank = i;
            for (uint32_t o = 0; o < 4; o++) {
                if (o == w) continue;
                const unsigned long long *lo_ = lists + (size_t)o * a.capw;
                uint32_t lo = 0, hi = ml.wcnt[o];
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (lo_[mid] < key)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                rank += lo;
            }
            if (rank < a.k) mb[rank] = key;
        }
    }
    __syncthreads();
    const uint32_t wtot = min(ml.wcnt[0] + ml.wcnt[1] + ml.wcnt[2] + ml.wcnt[3], a.k);
    unsigned long long *const mine = a.wg_keys + (size_t)blockIdx.x * a.k;
    for (uint32_t i = tid; i < wtot; i += 256) mine[i] = mb[i];
    if (tid == 0) a.wg_cnt[blockIdx.x] = wtot;
    __threadfence();
    __syncthreads();
    DIR_STAMP(4);
    if (tid == 0) ml.ticket = atomicAdd(a.ticket, 1u);
    __syncthreads();
    DIR_STAMP(5);
    if (ml.ticket != gridDim.x - 1) return;
    __threadfence();

    // ---- the last workgroup: every list's head in an LDS pool, lists extended while they still hold keys at or below the pool's k-th
    const uint32_t G = gridDim.x;
    unsigned long long *const pool = (unsigned long long *)smem;
    uint32_t *const s_len = (uint32_t *)(smem + (size_t)DIR_POOL * 8);  // [256] keys of list g already in the pool
    uint32_t *const s_cntg = s_len + 256;                               // [256] its length
    if (tid == 0) {
        ml.total = 0;
        ml.have = 0;
        ml.outn = 0;
        ml.kmin = ~0ull;
        ml.kmax = 0;
        *a.ticket = 0;  // (every workgroup has taken its ticket: ready for the next launch)
    }
    ml.fin[tid] = ~0ull;
    __syncthreads();
    // first round, all threads: the first c0 keys of every list (a list holds k / G of the page on average), ~0 where a list is shorter
    const uint32_t c0 = min(DIR_CHUNK, max(8u, 4u * a.k / G + 4u));
    {
        const uint32_t cg = tid < G ? a.wg_cnt[tid] : 0;
        s_len[tid] = min(cg, c0);
        s_cntg[tid] = cg;
        const uint32_t tsum = wave_sum_u32(cg), hsum = wave_sum_u32(min(cg, c0));  // (one LDS atomic per wave, not per thread)
        if (lane == 0) {
            atomicAdd(&ml.total, tsum);
            atomicAdd(&ml.have, hsum);
        }
    }
    for (uint32_t x = tid; x < G * c0; x += 256) {
        const uint32_t g = x / c0, i = x - g * c0;
        pool[x] = i < a.wg_cnt[g] ? a.wg_keys[(size_t)g * a.k + i] : ~0ull;
    }
    if (tid == 0) ml.pool_n = G * c0;
    __syncthreads();
    DIR_STAMP(6);
    const uint32_t total = ml.total;
    const uint32_t want = (uint32_t)(a.k < a.n_rows ? a.k : a.n_rows);
    const uint32_t kk = min(total, a.k);  // rows on the finite part of the page
    bool overflow = *a.bad != 0;  // (read behind the ticket's fence)
    __syncthreads();
    if (tid == 0) *a.bad = 0;
    unsigned long long T = ~0ull;  // keys below T are wanted: +inf, then the pool's k-th key + 1
    uint32_t scanned = 0;          // pool entries already folded into kmin / kmax
    while (kk && !overflow) {
        // the pool's k-th key, when it holds that many: a radix select over (key - kmin), from the highest byte in which two keys
        // differ (all keys of a search share their upper bits: an LDS histogram of those would be one contended bin)
        const uint32_t pn = ml.pool_n;
        {
            unsigned long long lo = ~0ull, hi = 0;
            uint32_t real = 0;
            for (uint32_t x = scanned + tid; x < pn; x += 256) {
                const unsigned long long v = pool[x];
                if (v != ~0ull) {
                    lo = v < lo ? v : lo;
                    hi = v > hi ? v : hi;
                    real++;
                }
            }
            (void)real;
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
            T = kmin + wg_radix_kth_range(pool, pn, kk, kmin, (top / 8) * 8, ml.hist, ml.misc, &ml.slot) + 1;
        }
        if (tid == 0) ml.more = 0;
        __syncthreads();
        // every list whose next unread key is below T hands over DIR_CHUNK more
        if (tid < G) {
            const uint32_t len = s_len[tid], cg = s_cntg[tid];
            const unsigned long long *lst = a.wg_keys + (size_t)tid * a.k;
            if (len < cg && lst[len] < T) {
                const uint32_t take = min(DIR_CHUNK, cg - len);
                const uint32_t at = atomicAdd(&ml.pool_n, take);
                if (at + take <= DIR_POOL) {
                    for (uint32_t i = 0; i < take; i++) pool[at + i] = lst[len + i];
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
    DIR_STAMP(7);
    const bool can_complete = a.null_ok && bb < __builtin_inff() && (METRIC == PVS_L2 || bb > 0.f);
    const bool tail = kk < want;
    if (overflow || (tail && !can_complete)) {
        if (tid == 0) {
            a.need_dense[0] = 1;
            if (a.h_flags) a.h_flags[0] = 1;
            if (a.h_seen) a.h_seen[0] = 0;
            a.out_count[0] = 0;
        }
        return;
    }
    // the page: the pool's keys below T (exactly kk of them: the keys are distinct), rank-sorted
    const uint32_t pn = ml.pool_n;
    for (uint32_t i = tid; i < pn; i += 256) {
        const unsigned long long v = pool[i];
        if (v < T) {
            const uint32_t at = atomicAdd(&ml.outn, 1u);
            if (at < 256) ml.fin[at] = v;
        }
    }
    __syncthreads();
    {
        const unsigned long long v = ml.fin[tid];
        uint32_t rank = 0;
        const ulonglong2 *f2 = (const ulonglong2 *)ml.fin;  // (entries beyond the page's kk keys are ~0: never smaller)
        for (uint32_t j = 0; j < kk; j += 4) {
            const ulonglong2 x0 = f2[(j >> 1)], x1 = f2[(j >> 1) + 1];
            rank += (x0.x < v ? 1u : 0u) + (x0.y < v ? 1u : 0u) + (x1.x < v ? 1u : 0u) + (x1.y < v ? 1u : 0u);
        }
        __syncthreads();
        if (v != ~0ull) ml.fin[rank] = v;
        __syncthreads();
    }
    for (uint32_t i = tid; i < a.k; i += 256) {
        if (i < kk) {
            const unsigned long long v = ml.fin[i];
            const uint32_t r = (uint32_t)v;
            const uint32_t srow = a.tinv ? a.tinv[r] : r;
            const int64_t id = a.ids[srow];
            const float d = f32_from_sort_key((uint32_t)(v >> 32));
            a.out_ids[i] = id;
            a.out_dist[i] = d;
            if (a.h_out_ids && !tail) {
                a.h_out_ids[i] = id;
                a.h_out_dist[i] = d;
                if (a.h_out_rows) a.h_out_rows[i] = srow;
            }
        } else {
            a.out_ids[i] = -1;
            a.out_dist[i] = __builtin_nanf("");
        }
    }
#ifdef PVS_DIR_PROF
    DIR_STAMP(8);
    if (tid == 0)
        printf("dirprof G %u k %u: fill %llu stream %llu cut %llu wgmerge+publish %llu ticket %llu pool-load %llu select-loop %llu page %llu (x10 ns)\n", G, a.k,
               dp[1] - dp[0], dp[2] - dp[1], dp[3] - dp[2], dp[4] - dp[3], dp[5] - dp[4], dp[6] - dp[5], dp[7] - dp[6], dp[8] - dp[7]);
#endif
    if (tid == 0) {
        if (a.h_out_count && !tail) a.h_out_count[0] = kk;
        a.out_count[0] = kk;
        a.need_dense[0] = tail ? 3 : 0;
        if (a.h_flags) a.h_flags[0] = tail ? 3 : 0;
        if (a.h_seen) a.h_seen[0] = 0;
    }
}

template <int DT, int METRIC>
hipError_t direct_launch(const DirectK &k, uint32_t grid, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
    static std::atomic<bool> configured{false};
    if (!configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_direct_topk<DT, METRIC>, hipFuncAttributeMaxDynamicSharedMemorySize, DIR_LDS);
        if (e != hipSuccess) return e;
        configured.store(true, std::memory_order_release);
    }
    if (ev_start || ev_stop)
        hipExtLaunchKernelGGL((k_direct_topk<DT, METRIC>), dim3(grid), dim3(256), DIR_LDS, s, ev_start, ev_stop, 0, k);
    else
        hipLaunchKernelGGL((k_direct_topk<DT, METRIC>), dim3(grid), dim3(256), DIR_LDS, s, k);
    return hipGetLastError();
}
template <int DT>
hipError_t direct_metric(const DirectK &k, int metric, uint32_t grid, hipStream_t s, hipEvent_t a, hipEvent_t b) {
    return metric == PVS_COSINE ? direct_launch<DT, PVS_COSINE>(k, grid, s, a, b) : direct_launch<DT, PVS_L2>(k, grid, s, a, b);
}
}  // namespace

bool pvs_direct_supported(uint32_t stride, uint32_t esz, uint32_t k) {
    return k >= 1 && k <= PVS_DIRECT_MAX_K && (uint64_t)(stride / esz) * 4 <= (uint64_t)DIR_Q_LDS;
}
uint64_t pvs_direct_work_bytes(uint32_t n_cu) { return (uint64_t)n_cu * PVS_DIRECT_MAX_K * 8 + (uint64_t)n_cu * 4 + 64; }

hipError_t pvs_launch_direct_topk(const DirectArgs &d, hipStream_t s) {
    if (d.n_rows == 0 || !pvs_direct_supported(d.stride, pvs_esz((uint32_t)d.dtype), d.k)) return hipErrorInvalidValue;
    const uint32_t esz = pvs_esz((uint32_t)d.dtype);
    DirectK k;
    k.rows = d.rows;
    k.norm2 = d.norm2;
    k.qexact = d.qexact;
    k.qinfo = d.qinfo;
    k.trank = d.trank;
    k.tinv = d.tinv;
    k.ids = d.ids;
    k.mask = d.mask;
    const uint32_t n_pairs = (uint32_t)((d.n_rows + 63) / 64);
    const uint32_t grid = std::min<uint32_t>({(n_pairs + 3) / 4, std::max<uint32_t>(d.n_cu, 1), 256u});
    uint8_t *w = (uint8_t *)d.work;
    k.wg_keys = (unsigned long long *)w;
    k.wg_cnt = (uint32_t *)(w + (size_t)std::max<uint32_t>(d.n_cu, 1) * PVS_DIRECT_MAX_K * 8);
    k.ticket = k.wg_cnt + std::max<uint32_t>(d.n_cu, 1);
    k.bad = k.ticket + 1;
    k.out_ids = d.out_ids;
    k.out_dist = d.out_dist;
    k.out_count = d.out_count;
    k.need_dense = d.need_dense;
    k.h_flags = d.h_flags;
    k.h_seen = d.h_seen;
    k.h_out_ids = d.h_out_ids;
    k.h_out_dist = d.h_out_dist;
    k.h_out_count = d.h_out_count;
    k.h_out_rows = d.h_out_rows;
    k.n_rows = d.n_rows;
    k.stride = d.stride;
    k.kslabs = d.stride / PVS_KSLAB_BYTES;
    k.dim = d.dim;
    k.qpad_ld = d.stride / esz;
    k.n_pairs = n_pairs;
    k.n_waves = grid * 4;
    k.k = d.k;
    uint32_t kp = 16;
    while (kp < d.k) kp <<= 1;
    k.kp = kp;
    k.capw = std::max<uint32_t>(2 * kp, 128);
    k.null_ok = d.null_ok;
    k.q_is_i8 = d.dtype == PVS_I8 ? 1 : 0;
    return d.dtype == PVS_I8    ? direct_metric<PVS_I8>(k, d.metric, grid, s, d.ev_start, d.ev_stop)
           : d.dtype == PVS_F16 ? direct_metric<PVS_F16>(k, d.metric, grid, s, d.ev_start, d.ev_stop)
                                : direct_metric<PVS_F32>(k, d.metric, grid, s, d.ev_start, d.ev_stop);
}
