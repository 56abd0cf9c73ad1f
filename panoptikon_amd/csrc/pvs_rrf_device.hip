// pvs_rrf_device.hip — one round of the bounded RRF fusion (pvs_items.hip: rrf_bounded; HISTORY.md §4.4) with every step on the
// device and ONE synchronisation: the reference's `row_number() OVER (ORDER BY agg)` per branch, UNION, `SUM(w / (k + rank))`,
// `ORDER BY score DESC LIMIT k` (pql/builder.rs:757-771, 1284-1301) for the files that can reach the page.
//
// The host form of the round makes seven round trips (sample, page count, page, candidate keys, counts per branch; sorts on the host
// in between): 0.9-1.2 ms of a 7.5-ms composed query at configs[4] for ~0.15 ms of kernels.  Here, on one stream behind the branches'
// scoring:
//   per branch   k_rd_threshold   sample of 8,192 window keys in LDS -> the key T_b at or below which ~1.5 x target files lie
//                k_rd_compact     every file with key <= T_b -> a list of slots (any order)
//                k_rd_cut         cut back to the files at or below the list's own target-th smallest key: the page, R_b files
//   once         k_rd_union       candidates = sorted union of the pages' file ids
//   per branch   k_rd_lookup      each candidate's slot and key in this branch (binary search in the id-ordered file list)
//                k_rd_order       the candidates present in the branch, sorted by (key, file id) — the window order
//                k_rd_count       one pass over the branch's keys: files strictly before each candidate
//                k_rd_ranks       prefix sums -> exact window ranks
//   once         k_rd_fuse        SQLite's arithmetic for the fused score, sort by (score DESC, file id), first k -> pinned memory
// Exactness is the host form's: every step computes the same set or number; only where it runs changed.  Anything outside the
// sizes one LDS sort takes (more than 4,096 candidates, a compacted list above 16,384, massive ties at a threshold) raises a flag
// and the caller runs the host form.
#include <algorithm>
#include <atomic>

#include "pvs_kernels.hpp"
#include "pvs_wg_select.hpp"

namespace {
constexpr uint32_t RD_M = 8192;      // sampled keys per branch
constexpr uint32_t RD_CUT = 16384;   // compacted list the cut takes (its keys sit in LDS)
constexpr uint32_t RD_PAGE = 4096;   // page entries per branch
constexpr uint32_t RD_CAND = 4096;   // candidates of a round
constexpr uint32_t RD_T = 1024;      // threads of the single-workgroup kernels

struct RdBranch {
    const unsigned long long *keys;  // [n] window keys of the branch's files
    const int64_t *gids;             // [n] file ids, ascending
    uint32_t n;
    // work arrays of the branch
    unsigned long long *thr;         // [1]
    uint32_t *cnt;                   // [1] compacted entries (may exceed RD_CUT)
    uint32_t *slots;                 // [RD_CUT]
    int64_t *page_g;                 // [RD_PAGE]
    uint32_t *cslot;                 // [RD_CAND] slot of every candidate (~0: absent)
    unsigned long long *ckey;        // [RD_CAND] its key
    unsigned long long *ok;          // [RD_CAND] present candidates in window order: key
    int64_t *og;                     //           file id
    uint32_t *oi;                    //           candidate index
    uint32_t *mp;                    // [1] how many are present
    unsigned long long *hist;        // [RD_CAND + 1]
    int64_t *rank;                   // [RD_CAND] exact window rank of every candidate (-1: absent)
};
struct RdArgs {
    RdBranch br[PVS_RRF_MAX_BRANCHES];
    uint32_t nb, target, k;
    PvsRrfParams p;
    int64_t *cand;    // [RD_CAND]
    uint32_t *m;      // [1]
    // pinned, device-mapped output block
    uint32_t *flags;  // [0]: != 0 -> the host form answers this round
    uint32_t *R;      // [nb] page sizes
    uint32_t *out_m;  // [1] candidates
    int64_t *out_g;   // [k]
    double *out_s;    // [k]
    uint32_t *out_n;  // [1] entries written (min(k, candidates))
};

// ascending bitonic sort of n2 (a power of two) records of up to three parallel LDS arrays, ordered by (a, b) — b, c may be null;
// workgroup-wide
template <typename A, typename B, typename C>
__device__ inline void wg_bitonic(A *a, B *b, C *c, uint32_t n2) {
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    for (uint32_t sz = 2; sz <= n2; sz <<= 1)
        for (uint32_t st = sz >> 1; st > 0; st >>= 1) {
            for (uint32_t i = tid; i < n2 / 2; i += nt) {
                const uint32_t lo = 2 * i - (i & (st - 1)), hi = lo + st;
                const bool up = (lo & sz) == 0;
                const bool gt = a[lo] != a[hi] ? a[lo] > a[hi] : (b ? b[lo] > b[hi] : false);
                if (gt == up) {
                    const A x = a[lo];
                    a[lo] = a[hi];
                    a[hi] = x;
                    if (b) {
                        const B y = b[lo];
                        b[lo] = b[hi];
                        b[hi] = y;
                    }
                    if (c) {
                        const C z = c[lo];
                        c[lo] = c[hi];
                        c[hi] = z;
                    }
                }
            }
            __syncthreads();
        }
}
__device__ inline uint32_t pow2_at_least(uint32_t n) {
    uint32_t p = 64;
    while (p < n) p <<= 1;
    return p;
}

__global__ __launch_bounds__(1024) void k_rd_threshold(RdArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t rd_smem[];
    unsigned long long *s = (unsigned long long *)rd_smem;  // [RD_M]
    __shared__ uint32_t hist[256], misc[4];
    const RdBranch &b = a.br[blockIdx.x];
    const uint32_t tid = threadIdx.x;
    if (b.n == 0 || (uint64_t)a.target * 2 >= b.n) {  // a page of (nearly) everything: no threshold
        if (tid == 0) *b.thr = ~0ull;
        return;
    }
    for (uint32_t i = tid; i < RD_M; i += RD_T) s[i] = b.keys[(uint64_t)i * b.n / RD_M];
    __syncthreads();
    uint64_t j = (uint64_t)((double)RD_M * 1.5 * (double)a.target / (double)b.n) + 1;
    if (j >= RD_M) j = RD_M - 1;
    const unsigned long long t = wg_radix_kth_u64(s, RD_M, (uint32_t)j + 1, hist, misc);  // the sample's j-th smallest (0-based)
    if (tid == 0) *b.thr = t;
}
// every file with key <= T_b: its slot, in any order (hits collected per workgroup in LDS, one global atomic per workgroup)
__global__ __launch_bounds__(256) void k_rd_compact(RdBranch b) {
    constexpr uint32_t LIST = 1024;
    __shared__ uint32_t s_n, s_base, s_list[LIST];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const unsigned long long thr = *b.thr;
    const uint32_t step = gridDim.x * 256;
    for (uint32_t i0 = blockIdx.x * 256 + threadIdx.x; i0 < b.n; i0 += 4 * step) {
        unsigned long long k4[4];  // four independent loads in flight per lane (one per iteration leaves the pass latency-bound)
#pragma unroll
        for (int u = 0; u < 4; u++) k4[u] = i0 + u * step < b.n ? b.keys[i0 + u * step] : ~0ull;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = i0 + u * step;
            if (i < b.n && k4[u] <= thr) {
                const uint32_t p = atomicAdd(&s_n, 1u);
                if (p < LIST) {
                    s_list[p] = i;
                } else {
                    const uint32_t gp = atomicAdd(b.cnt, 1u);
                    if (gp < RD_CUT) b.slots[gp] = i;
                }
            }
        }
    }
    __syncthreads();
    const uint32_t m = s_n < LIST ? s_n : LIST;
    if (threadIdx.x == 0 && m) s_base = atomicAdd(b.cnt, m);
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < m; j += 256)
        if (s_base + j < RD_CUT) b.slots[s_base + j] = s_list[j];
}
// the sampled threshold is a noisy order statistic: cut the list back to the files at or below its own target-th smallest key —
// still "every file with key <= T'", only with a smaller T' (rrf_bounded does the same with nth_element)
__global__ __launch_bounds__(1024) void k_rd_cut(RdArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t rd_smem[];
    unsigned long long *s = (unsigned long long *)rd_smem;  // [RD_CUT]
    __shared__ uint32_t hist[256], misc[4], s_n;
    const RdBranch &b = a.br[blockIdx.x];
    const uint32_t tid = threadIdx.x, cnt = *b.cnt;
    if (tid == 0) s_n = 0;
    if (cnt > RD_CUT) {  // more than the cut takes (massive ties at the threshold, or an unlucky sample): the host form
        if (tid == 0) {
            atomicOr(a.flags, 1u);
            a.R[blockIdx.x] = 0;
        }
        return;
    }
    for (uint32_t i = tid; i < cnt; i += RD_T) s[i] = b.keys[b.slots[i]];
    __syncthreads();
    unsigned long long t2 = ~0ull;
    if (cnt > a.target) t2 = wg_radix_kth_u64(s, cnt, a.target, hist, misc);
    __syncthreads();
    for (uint32_t i = tid; i < cnt; i += RD_T)
        if (s[i] <= t2) {
            const uint32_t p = atomicAdd(&s_n, 1u);
            if (p < RD_PAGE) b.page_g[p] = b.gids[b.slots[i]];
        }
    __syncthreads();
    if (tid == 0) {
        if (s_n > RD_PAGE) atomicOr(a.flags, 2u);  // ties at the cut: more than a page holds
        a.R[blockIdx.x] = s_n;
    }
}
// candidates = sorted union of the pages
__global__ __launch_bounds__(1024) void k_rd_union(RdArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t rd_smem[];
    int64_t *s = (int64_t *)rd_smem;               // [RD_CAND]
    uint32_t *head = (uint32_t *)(s + RD_CAND);    // [RD_CAND] 1 where a new id starts -> its output position
    __shared__ uint32_t s_wave[16];
    const uint32_t tid = threadIdx.x;
    uint32_t total = 0;
    for (uint32_t b = 0; b < a.nb; b++) total += a.R[b] < RD_PAGE ? a.R[b] : RD_PAGE;
    if (*a.flags || total > RD_CAND) {
        if (tid == 0) {
            atomicOr(a.flags, 4u);
            *a.m = 0;
            *a.out_m = 0;
        }
        return;
    }
    const uint32_t n2 = pow2_at_least(total);
    for (uint32_t e = tid; e < n2; e += RD_T) {
        int64_t g = 0x7fffffffffffffffll;
        uint32_t base = 0;
        for (uint32_t b = 0; b < a.nb; b++) {
            const uint32_t r = a.R[b];
            if (e >= base && e < base + r) g = a.br[b].page_g[e - base];
            base += r;
        }
        s[e] = g;
    }
    __syncthreads();
    wg_bitonic<int64_t, int64_t, int64_t>(s, nullptr, nullptr, n2);
    // heads of runs, and an exclusive prefix over them: 4 consecutive entries per thread, a shuffle scan per wave, the wave totals in LDS
    uint32_t mine[4], cntm = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const uint32_t e = 4 * tid + u;
        mine[u] = e < total && (e == 0 || s[e] != s[e - 1]) ? 1u : 0u;
        cntm += mine[u];
    }
    uint32_t v = cntm;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)v, off, 64);
        if (lane >= off) v += up;
    }
    if (lane == 63) s_wave[wave] = v;
    __syncthreads();
    uint32_t before = v - cntm;
    for (int w = 0; w < wave; w++) before += s_wave[w];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const uint32_t e = 4 * tid + u;
        if (e < RD_CAND) head[e] = mine[u] ? before : 0xffffffffu;
        before += mine[u];
    }
    __syncthreads();
    for (uint32_t e = tid; e < total; e += RD_T)
        if (head[e] != 0xffffffffu) a.cand[head[e]] = s[e];
    if (tid == RD_T - 1) {
        *a.m = before;  // (the last thread's running count is the number of distinct ids)
        *a.out_m = before;
    }
}
// candidate file ids -> their slot in this branch (files are stored in id order) and window key; absent: slot = ~0
__global__ __launch_bounds__(256) void k_rd_lookup(RdArgs a) {
    const RdBranch &b = a.br[blockIdx.y];
    const uint32_t c = blockIdx.x * 256 + threadIdx.x, m = *a.m;
    if (c >= m) return;
    const int64_t g = a.cand[c];
    uint32_t lo = 0, hi = b.n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (b.gids[mid] < g) lo = mid + 1;
        else hi = mid;
    }
    const bool found = lo < b.n && b.gids[lo] == g;
    b.cslot[c] = found ? lo : 0xffffffffu;
    b.ckey[c] = found ? b.keys[lo] : 0ull;
    b.rank[c] = -1;
}
// the candidates present in the branch in window order: (key, file id) ascending
__global__ __launch_bounds__(1024) void k_rd_order(RdArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t rd_smem[];
    unsigned long long *sk = (unsigned long long *)rd_smem;  // [RD_CAND]
    int64_t *sg = (int64_t *)(sk + RD_CAND);                 // [RD_CAND]
    uint32_t *si = (uint32_t *)(sg + RD_CAND);               // [RD_CAND]
    __shared__ uint32_t s_n;
    const RdBranch &b = a.br[blockIdx.x];
    const uint32_t tid = threadIdx.x, m = *a.m;
    if (tid == 0) s_n = 0;
    __syncthreads();
    const uint32_t n2 = pow2_at_least(m);
    for (uint32_t c = tid; c < n2; c += RD_T) {
        const bool present = c < m && b.cslot[c] != 0xffffffffu;
        sk[c] = present ? b.ckey[c] : ~0ull;
        sg[c] = present ? a.cand[c] : 0x7fffffffffffffffll;  // (absent ones sort behind every present one)
        si[c] = c;
        if (present) atomicAdd(&s_n, 1u);
    }
    __syncthreads();
    wg_bitonic<unsigned long long, int64_t, uint32_t>(sk, sg, si, n2);
    const uint32_t mp = s_n;
    for (uint32_t i = tid; i < mp; i += RD_T) {
        b.ok[i] = sk[i];
        b.og[i] = sg[i];
        b.oi[i] = si[i];
    }
    for (uint32_t i = tid; i <= mp; i += RD_T) b.hist[i] = 0;
    if (tid == 0) *b.mp = mp;
}
// hist[p]++ with p = number of present candidates whose (key, file id) is <= the file's: one pass over the branch's keys, binary
// search in LDS; files behind every candidate — nearly all of them — are dismissed by one compare and never read their id
__global__ __launch_bounds__(1024) void k_rd_count(RdBranch b) {  // (80 KB of candidates per workgroup: one workgroup of 16 waves per CU)
    extern __shared__ __attribute__((aligned(16))) uint8_t rd_smem[];
    const uint32_t m = *b.mp;
    unsigned long long *sk = (unsigned long long *)rd_smem;  // [m]
    int64_t *ss = (int64_t *)(sk + RD_CAND);                 // [m]
    uint32_t *sh = (uint32_t *)(ss + RD_CAND);               // [m + 1]
    if (m == 0) return;
    for (uint32_t i = threadIdx.x; i < m; i += RD_T) {
        sk[i] = b.ok[i];
        ss[i] = b.og[i];
    }
    for (uint32_t i = threadIdx.x; i <= m; i += RD_T) sh[i] = 0;
    __syncthreads();
    const unsigned long long kmax = sk[m - 1];
    const uint32_t step = gridDim.x * RD_T;
    for (uint32_t g0 = blockIdx.x * RD_T + threadIdx.x; g0 < b.n; g0 += 4 * step) {
      unsigned long long k4[4];  // four independent loads in flight per lane
#pragma unroll
      for (int u = 0; u < 4; u++) k4[u] = g0 + u * step < b.n ? b.keys[g0 + u * step] : ~0ull;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t g = g0 + u * step;
        const unsigned long long k = k4[u];
        if (g >= b.n || k > kmax) continue;
        const int64_t gid = b.gids[g];
        uint32_t lo = 0, hi = m;  // first candidate with (key, file id) > this file's
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            const bool le = sk[mid] < k || (sk[mid] == k && ss[mid] <= gid);
            if (le) lo = mid + 1;
            else hi = mid;
        }
        if (lo < m) atomicAdd(&sh[lo], 1u);
      }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= m; i += RD_T)
        if (sh[i]) atomicAdd(&b.hist[i], (unsigned long long)sh[i]);
}
// files strictly before candidate j of the window order = sum_{p <= j} hist[p]; its rank = that + 1
__global__ __launch_bounds__(1024) void k_rd_ranks(RdArgs a) {
    __shared__ unsigned long long s_wave[16];
    const RdBranch &b = a.br[blockIdx.x];
    const uint32_t tid = threadIdx.x, mp = *b.mp;
    unsigned long long mine[4], tot = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const uint32_t e = 4 * tid + u;
        mine[u] = e < mp ? b.hist[e] : 0ull;
        tot += mine[u];
    }
    unsigned long long v = tot;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long up = (unsigned long long)__shfl_up((long long)v, off, 64);
        if (lane >= off) v += up;
    }
    if (lane == 63) s_wave[wave] = v;
    __syncthreads();
    unsigned long long run = v - tot;
    for (int w = 0; w < wave; w++) run += s_wave[w];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const uint32_t e = 4 * tid + u;
        run += mine[u];
        if (e < mp) b.rank[b.oi[e]] = (int64_t)run + 1;
    }
}
// SQLite's arithmetic for one term (pql/builder.rs:1284-1301; rrf_score_host in pvs_items.hip, k_rrf_score in pvs_rrf.hip)
__device__ inline double rd_term(int32_t k, int64_t rank, double w) {
    const int64_t BIG = 9223372036854775805LL;
    if (rank < 0) rank = BIG;
    int64_t di;
    const double denom = __builtin_add_overflow((int64_t)k, rank, &di) ? (double)k + (double)rank : (double)di;
    return (1.0 / denom) * w;
}
__global__ __launch_bounds__(1024) void k_rd_fuse(RdArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t rd_smem[];
    unsigned long long *sk = (unsigned long long *)rd_smem;  // [RD_CAND] ~(sortable score): ascending = score descending
    int64_t *sg = (int64_t *)(sk + RD_CAND);                 // [RD_CAND]
    double *sc = (double *)(sg + RD_CAND);                   // [RD_CAND] the score as computed (travels with its record)
    const uint32_t tid = threadIdx.x, m = *a.m;
    if (*a.flags) {
        if (tid == 0) *a.out_n = 0;
        return;
    }
    const uint32_t n2 = pow2_at_least(m);
    for (uint32_t c = tid; c < n2; c += RD_T) {
        if (c < m) {
            double tot = 0.0;
            for (uint32_t b = 0; b < a.nb; b++) {
                const double t = rd_term(a.p.k[b], a.br[b].rank[c], a.p.w[b]);
                tot = b == 0 ? t : tot + t;
            }
            unsigned long long u = (unsigned long long)__double_as_longlong(tot);
            u = (u >> 63) ? ~u : (u | 0x8000000000000000ull);  // ascending in the score
            sk[c] = ~u;                                        // descending
            sg[c] = a.cand[c];
            sc[c] = tot;
        } else {
            sk[c] = ~0ull;
            sg[c] = 0x7fffffffffffffffll;
            sc[c] = 0.0;
        }
    }
    __syncthreads();
    wg_bitonic<unsigned long long, int64_t, double>(sk, sg, sc, n2);
    const uint32_t nout = m < a.k ? m : a.k;
    for (uint32_t i = tid; i < nout; i += RD_T) {
        a.out_g[i] = sg[i];
        a.out_s[i] = sc[i];
    }
    if (tid == 0) *a.out_n = nout;
}

size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
std::atomic<bool> g_rd_configured{false};
}  // namespace

bool pvs_rrf_round_device_supported(uint32_t nb, uint64_t target, uint32_t k) {
    return nb >= 1 && nb <= (uint32_t)PVS_RRF_MAX_BRANCHES && target <= RD_PAGE / 2 && (uint64_t)nb * target <= RD_CAND && k <= RD_CAND;
}
size_t pvs_rrf_round_device_work_bytes(uint32_t nb) {
    const size_t per = al(8) + al(4) + al(RD_CUT * 4) + al(RD_PAGE * 8) + al(RD_CAND * 4) + al(RD_CAND * 8) * 3 + al(RD_CAND * 4) + al(4) + al((RD_CAND + 1) * 8) + al(RD_CAND * 8);
    return per * nb + al(RD_CAND * 8) + al(4) + 256;
}
size_t pvs_rrf_round_device_out_bytes(uint32_t nb, uint32_t k) { return 64 + al((size_t)nb * 4) + al((size_t)k * 8) * 2 + 256; }

// One round on stream s (which already waits for the branches' window keys).  h_out: a pinned, device-mapped block of
// pvs_rrf_round_device_out_bytes; the caller synchronises s and reads it with pvs_rrf_round_device_result.
hipError_t pvs_rrf_round_device(const unsigned long long *const *d_keys, const int64_t *const *d_gids, const uint32_t *n, uint32_t nb, const PvsRrfParams &p,
                                uint32_t target, uint32_t k, void *d_work, uint8_t *h_out, hipStream_t s) {
    if (!g_rd_configured.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_rd_threshold, hipFuncAttributeMaxDynamicSharedMemorySize, RD_M * 8);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_rd_cut, hipFuncAttributeMaxDynamicSharedMemorySize, RD_CUT * 8);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_rd_union, hipFuncAttributeMaxDynamicSharedMemorySize, RD_CAND * 12);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_rd_order, hipFuncAttributeMaxDynamicSharedMemorySize, RD_CAND * 20);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_rd_count, hipFuncAttributeMaxDynamicSharedMemorySize, RD_CAND * 20 + 16);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_rd_fuse, hipFuncAttributeMaxDynamicSharedMemorySize, RD_CAND * 24);
        if (e != hipSuccess) return e;
        g_rd_configured.store(true, std::memory_order_release);
    }
    RdArgs a;
    memset(&a, 0, sizeof a);
    a.nb = nb;
    a.target = target;
    a.k = k;
    a.p = p;
    uint8_t *w = (uint8_t *)d_work;
    auto take = [&](size_t bytes) {
        uint8_t *r = w;
        w += al(bytes);
        return r;
    };
    for (uint32_t b = 0; b < nb; b++) {
        RdBranch &r = a.br[b];
        r.keys = d_keys[b];
        r.gids = d_gids[b];
        r.n = n[b];
        r.thr = (unsigned long long *)take(8);
        r.cnt = (uint32_t *)take(4);
        r.slots = (uint32_t *)take(RD_CUT * 4);
        r.page_g = (int64_t *)take(RD_PAGE * 8);
        r.cslot = (uint32_t *)take(RD_CAND * 4);
        r.ckey = (unsigned long long *)take(RD_CAND * 8);
        r.ok = (unsigned long long *)take(RD_CAND * 8);
        r.og = (int64_t *)take(RD_CAND * 8);
        r.oi = (uint32_t *)take(RD_CAND * 4);
        r.mp = (uint32_t *)take(4);
        r.hist = (unsigned long long *)take((RD_CAND + 1) * 8);
        r.rank = (int64_t *)take(RD_CAND * 8);
    }
    a.cand = (int64_t *)take(RD_CAND * 8);
    a.m = (uint32_t *)take(4);
    uint8_t *o = h_out;
    a.flags = (uint32_t *)o;
    a.out_m = (uint32_t *)(o + 16);
    a.out_n = (uint32_t *)(o + 32);
    a.R = (uint32_t *)(o + 64);
    a.out_g = (int64_t *)(o + 64 + al((size_t)nb * 4));
    a.out_s = (double *)((uint8_t *)a.out_g + al((size_t)k * 8));
    *(volatile uint32_t *)a.flags = 0;
    *(volatile uint32_t *)a.out_m = 0;
    *(volatile uint32_t *)a.out_n = 0;
    for (uint32_t b = 0; b < nb; b++) {
        hipError_t e = hipMemsetAsync(a.br[b].cnt, 0, 4, s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_rd_threshold, dim3(nb), dim3(RD_T), RD_M * 8, s, a);
    for (uint32_t b = 0; b < nb; b++)
        if (a.br[b].n) {
            const unsigned g = (unsigned)std::min<uint64_t>(((uint64_t)a.br[b].n + 255) / 256, 2048);
            hipLaunchKernelGGL(k_rd_compact, dim3(g), dim3(256), 0, s, a.br[b]);
        }
    hipLaunchKernelGGL(k_rd_cut, dim3(nb), dim3(RD_T), RD_CUT * 8, s, a);
    hipLaunchKernelGGL(k_rd_union, dim3(1), dim3(RD_T), RD_CAND * 12, s, a);
    hipLaunchKernelGGL(k_rd_lookup, dim3(RD_CAND / 256, nb), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_rd_order, dim3(nb), dim3(RD_T), RD_CAND * 20, s, a);
    for (uint32_t b = 0; b < nb; b++)
        if (a.br[b].n) {
            const unsigned g = (unsigned)std::min<uint64_t>(((uint64_t)a.br[b].n + RD_T - 1) / RD_T, 256);
            hipLaunchKernelGGL(k_rd_count, dim3(g), dim3(RD_T), RD_CAND * 20 + 16, s, a.br[b]);
        }
    hipLaunchKernelGGL(k_rd_ranks, dim3(nb), dim3(RD_T), 0, s, a);
    hipLaunchKernelGGL(k_rd_fuse, dim3(1), dim3(RD_T), RD_CAND * 24, s, a);
    return hipGetLastError();
}
// after the stream drained: flags != 0 -> the round must be redone by the host form
void pvs_rrf_round_device_result(const uint8_t *h_out, uint32_t nb, uint32_t k, uint32_t *flags, uint32_t *R, uint32_t *m, uint32_t *n_out, const int64_t **groups,
                                 const double **scores) {
    *flags = *(const volatile uint32_t *)h_out;
    *m = *(const volatile uint32_t *)(h_out + 16);
    *n_out = *(const volatile uint32_t *)(h_out + 32);
    for (uint32_t b = 0; b < nb; b++) R[b] = ((const volatile uint32_t *)(h_out + 64))[b];
    *groups = (const int64_t *)(h_out + 64 + al((size_t)nb * 4));
    *scores = (const double *)((const uint8_t *)*groups + al((size_t)k * 8));
}
