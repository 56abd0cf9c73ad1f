// pvs_sparse.hip — searches whose cost follows the CANDIDATE SET, not the corpus.
//
// In the reference the vector filter is joined to the context CTE of the query's other filters
// (pql/builder/filters/image_embeddings.rs:140-199: `LEFT JOIN begin_cte ... WHERE begin_cte.item_id IS NOT NULL`): only the rows
// the other filters left are scored.  With a selective tag / path filter that is a few hundred rows of millions, and the reference
// prefetches up to 4,096 of them (api/search.rs:51) — more than there are.  Round 3 streamed the whole corpus for such a query
// (masked rows as NaN scalars) and, when fewer than k rows were allowed, fell to the dense path: every one of N rows scored and
// selected, ~8x the filter scan.  Here:
//   * a row list (pvs_search_rows) or a sparse mask (pvs_search_filtered counts it) is answered by GATHER-AND-SCORE: the exact
//     distance of every allowed row in the reference's order (rerank_distance: the arithmetic pass C uses), one lane per (row,
//     query) pair, then one in-LDS sort per query (<= 8,192 rows) or the radix select over the gathered matrix — no corpus pass,
//     no dense fallback, NULL rows included where the reference puts them (last, by tie order);
//   * the per-item form (pvs_search_groups_filtered with MAX / AVG / weights) aggregates the gathered rows per group through a
//     CSR built for the list;
//   * a page that ends in NULL rows (pql/builder.rs:1201-1205: NULLS LAST) no longer sends a cosine query to the dense path: the
//     NULL set of a cosine search is query-independent — zero vectors and rows with non-finite components, found once per index
//     state — and the tail of the page is the head of that list in tie order (k_null_tail).
#include <hipcub/hipcub.hpp>

#include "pvs_index.hpp"
#include "pvs_rerank.hpp"

namespace {
// ------------------------------------------------------------------ NULL rows of a cosine search
// suspects: rows whose |a|^2 is not a positive, moderate number
__device__ inline bool norm_suspect(float aa) { return !(aa > 0.f && aa < 1e30f); }
__global__ void k_null_count(const float *norm2, uint64_t n, uint32_t *count) {
    uint32_t mine = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) mine += norm_suspect(norm2[r]);
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(count, mine);
}
// Class of every suspect, per metric (appended unordered as (sort key, row); sort key = tie rank when a second sort key is set, else
// the row).  Cosine (1 - dot / (|a||q|)): NULL for EVERY query when every component is +-0 (0/0) or some component is NaN / inf
// (dot or |a| not finite: NaN or inf/inf); query-dependent ("weird") when |a|^2 under- or overflowed although the components are
// ordinary numbers.  L2 (sqrt(sum (a-q)^2)): NULL for every query when some component is NaN; weird when some component is inf
// (inf - inf only against a query that is inf there).  An index with weird rows of a metric keeps the dense fallback for it.
// counters: [0] cosine NULL rows, [1] L2 NULL rows, [2] cosine weird, [3] L2 weird
template <int DT>
__global__ void k_null_classify(const uint8_t *rows, uint32_t stride, int dim, const float *norm2, uint64_t n, const uint32_t *trank,
                                uint32_t *counters, uint32_t *cos_key, uint32_t *cos_row, uint32_t *l2_key, uint32_t *l2_row, uint32_t cap) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n || !norm_suspect(norm2[r])) return;
    bool all_zero = true, has_nan = false, has_inf = false;
    if constexpr (DT == PVS_I8) {
        row_foreach<DT>(rows, stride, r, dim, [&](int, int v) { all_zero &= v == 0; });
    } else {
        row_foreach<DT>(rows, stride, r, dim, [&](int, float v) {
            all_zero &= v == 0.f;
            has_nan |= v != v;
            has_inf |= v == __builtin_inff() || v == -__builtin_inff();
        });
    }
    const uint32_t key = trank ? trank[r] : (uint32_t)r;
    if (all_zero || has_nan || has_inf) {
        const uint32_t p = atomicAdd(&counters[0], 1u);
        if (p < cap) {
            cos_key[p] = key;
            cos_row[p] = (uint32_t)r;
        }
    } else {
        atomicAdd(&counters[2], 1u);
    }
    if (has_nan) {
        const uint32_t p = atomicAdd(&counters[1], 1u);
        if (p < cap) {
            l2_key[p] = key;
            l2_row[p] = (uint32_t)r;
        }
    } else if (has_inf) {
        atomicAdd(&counters[3], 1u);
    }
}

// The tail of a page that ends in NULL rows (flag 3 from pass C): the first allowed rows of the NULL list in tie order (or, for a
// query that makes every distance NULL — cosine: a zero or NaN-bearing query, L2: a NaN-bearing one — of ALL rows in tie order).
// One workgroup per query.
__global__ __launch_bounds__(256) void k_null_tail(uint32_t *flags, const QInfo *qinfo, int metric, const uint32_t *null_rows, uint32_t n_null,
                                                   const uint32_t *tinv, uint64_t n_rows, const uint8_t *mask, const int64_t *ids, uint32_t k, int64_t *out_ids,
                                                   float *out_dist, uint32_t *out_count, uint32_t *h_flags) {
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    if (flags[q] != 3) return;
    __shared__ uint32_t s_wave[4], s_base;
    const float bb = qinfo[q].bb;
    const bool all_null = pvs_query_all_null(metric, bb);
    const uint64_t L = all_null ? n_rows : (uint64_t)n_null;
    int64_t *oi = out_ids + (size_t)q * k;
    float *od = out_dist + (size_t)q * k;
    uint32_t have = out_count[q];
    if (tid == 0) s_base = have;
    __syncthreads();
    for (uint64_t i0 = 0; i0 < L && have < k; i0 += 256) {
        const uint64_t i = i0 + tid;
        uint32_t row = 0;
        bool ok = false;
        if (i < L) {
            row = all_null ? (tinv ? tinv[i] : (uint32_t)i) : null_rows[i];
            ok = !mask || mask[row] != 0;
        }
        const unsigned long long b = __ballot(ok);
        const int lane = tid & 63, wave = tid >> 6;
        const uint32_t before = (uint32_t)__popcll(b & ((1ull << lane) - 1));
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(b);
        __syncthreads();
        uint32_t off = s_base;
        for (int w = 0; w < wave; w++) off += s_wave[w];
        const uint32_t p = off + before;
        if (ok && p < k) {
            oi[p] = ids[row];
            od[p] = __builtin_nanf("");
        }
        const uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
        if (tid == 0) s_base += total;
        have += total;
        __syncthreads();
    }
    if (tid == 0) {
        out_count[q] = have < k ? have : k;
        flags[q] = 0;
        if (h_flags) h_flags[q] = 0;
    }
}

// ------------------------------------------------------------------ candidate lists
// non-zero bytes of the mask: 16 bytes per load where the pointer allows it, ONE global atomic per workgroup (the first form — a byte
// per load, an atomic per wave on one address, 2,700 of them for 690k rows — took 34 us for 690 KB)
__global__ __launch_bounds__(256) void k_mask_count(const uint8_t *mask, uint64_t n, uint32_t *count) {
    __shared__ uint32_t s_part[4];
    uint32_t mine = 0;
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x, nt = (uint64_t)gridDim.x * 256;
    uint64_t done = 0;
    if (((uintptr_t)mask & 15u) == 0) {
        const uint64_t n16 = n / 16;
        const uint4 *m4 = (const uint4 *)mask;
        auto nz = [](uint32_t w) { return (uint32_t)__popc((((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u); };
        for (uint64_t i = tid; i < n16; i += nt) {
            const uint4 v = m4[i];
            mine += nz(v.x) + nz(v.y) + nz(v.z) + nz(v.w);
        }
        done = n16 * 16;
    }
    for (uint64_t r = done + tid; r < n; r += nt) mine += mask[r] != 0;
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (t) atomicAdd(count, t);
    }
}
struct NonZero {
    __host__ __device__ bool operator()(const uint8_t &v) const { return v != 0; }
};
// ascending and inside the index?  (flag bit 0: not strictly ascending, bit 1: a row beyond the index)
__global__ void k_list_check(const uint32_t *list, uint32_t m, uint64_t n_rows, uint32_t *flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    if (list[i] >= n_rows) atomicOr(flag, 2u);
    if (i && list[i - 1] >= list[i]) atomicOr(flag, 1u);
}
__global__ void k_list_scatter_mask(const uint32_t *list, uint32_t m, uint64_t n_rows, uint8_t *mask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m && list[i] < n_rows) mask[list[i]] = 1;
}
__global__ void k_list_gather_ids(const uint32_t *list, uint32_t m, const int64_t *ids, const uint32_t *trank, int64_t *sub_ids, uint32_t *sub_key, uint32_t *sub_pos) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    sub_ids[i] = ids[list[i]];
    if (sub_key) {
        sub_key[i] = trank[list[i]];
        sub_pos[i] = i;
    }
}

// ------------------------------------------------------------------ gather-and-score
// One lane per (list row, query) pair: QT adjacent lanes share a row (their loads coalesce into one), the workgroup's QT queries sit
// in LDS zero-padded to whole 16-byte chunks.  out[pos * ld + q].  Exactly the reference's arithmetic (rerank_distance).
template <int DT>
__global__ __launch_bounds__(256) void k_sparse_score(const uint8_t *rows, uint32_t stride, int dim, int metric, const float *norm2, const uint32_t *list,
                                                      uint32_t m, uint64_t n_rows, const void *qexact, const QInfo *qinfo, uint32_t nb, uint32_t qt, uint32_t qpad,
                                                      float *out, uint32_t ld, uint32_t *flag) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_q[];  // [qt][qpad]
    const uint32_t tid = threadIdx.x, q0 = blockIdx.y * qt;
    const uint32_t qbytes = (uint32_t)dim * (DT == PVS_I8 ? 1u : 4u);
    for (uint32_t i = tid; i < qt * (qpad / 4); i += 256) {
        const uint32_t ql = i / (qpad / 4), w = i % (qpad / 4);
        uint32_t v = 0;
        if (q0 + ql < nb) {
            const uint8_t *src = (const uint8_t *)qexact + (size_t)(q0 + ql) * qbytes;
            if ((qbytes & 3u) == 0 && 4 * w + 4 <= qbytes) {
                v = *(const uint32_t *)(src + 4 * w);
            } else {  // (an int8 query of a dimension that is no multiple of 4: bytewise, the words would be misaligned)
                for (uint32_t b = 0; b < 4; b++)
                    if (4 * w + b < qbytes) v |= (uint32_t)src[4 * w + b] << (8 * b);
            }
        }
        ((uint32_t *)s_q)[(size_t)ql * (qpad / 4) + w] = v;
    }
    __syncthreads();
    const uint32_t rows_per_wg = 256 / qt;
    const uint32_t ql = tid % qt, pos = blockIdx.x * rows_per_wg + tid / qt, q = q0 + ql;
    if (pos >= m || q >= nb) return;
    const uint32_t row = list[pos];
    // the list is validated here (no separate pass, no round trip before the gather): a position beyond the index is never
    // dereferenced, an unsorted list is reported; the caller discards the page when the flag is raised
    if (row >= n_rows) {
        if (q == 0) flag[1] = 2u;  // (plain stores: the words live in pinned host memory, an atomic there needs PCIe atomics)
        out[(size_t)pos * ld + q] = __builtin_nanf("");
        return;
    }
    if (q == 0 && pos && list[pos - 1] >= row) flag[0] = 1u;
    const float aa = norm2[row];
    out[(size_t)pos * ld + q] = rerank_distance<DT>(rows, stride, row, s_q + (size_t)ql * qpad, dim, metric, aa, qinfo[q].bb);
}

// One workgroup per query: the column's m <= 8,192 distances -> (distance key, tie position) records, bitonic sort in LDS, the first
// k with their ids.  NULL distances sort last, among themselves by tie position, and ARE part of the page (NULLS LAST).
__global__ __launch_bounds__(256) void k_sparse_sort(const float *d, uint32_t ld, const uint32_t *list, uint32_t m, uint64_t n_rows, const int64_t *ids,
                                                     const uint32_t *trank, const uint32_t *tinv, uint32_t k, int64_t *out_ids, float *out_dist,
                                                     uint32_t *out_count) {
    extern __shared__ unsigned long long s_sort[];
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    uint32_t m2 = 1;
    while (m2 < m) m2 <<= 1;
    for (uint32_t i = tid; i < m2; i += 256) {
        unsigned long long v = ~0ull;
        if (i < m) v = ((unsigned long long)f32_sort_key(d[(size_t)i * ld + q]) << 32) | (trank ? (list[i] < n_rows ? trank[list[i]] : 0u) : i);  // (a bad list: flagged by the scorer, the page is discarded)
        s_sort[i] = v;
    }
    __syncthreads();
    for (uint32_t sz = 2; sz <= m2; sz <<= 1)
        for (uint32_t st = sz >> 1; st > 0; st >>= 1) {
            for (uint32_t i = tid; i < m2 / 2; i += 256) {
                const uint32_t lo = 2 * i - (i & (st - 1)), hi = lo + st;
                const bool up = (lo & sz) == 0;
                const unsigned long long x = s_sort[lo], y = s_sort[hi];
                if ((x > y) == up) {
                    s_sort[lo] = y;
                    s_sort[hi] = x;
                }
            }
            __syncthreads();
        }
    const uint32_t nout = m < k ? m : k;
    int64_t *oi = out_ids + (size_t)q * k;
    float *od = out_dist + (size_t)q * k;
    for (uint32_t i = tid; i < k; i += 256) {
        if (i < nout) {
            const unsigned long long v = s_sort[i];
            const uint32_t row = trank ? tinv[(uint32_t)v] : list[(uint32_t)v];
            oi[i] = row < n_rows ? ids[row] : -1;
            od[i] = f32_from_sort_key((uint32_t)(v >> 32));
        } else {
            oi[i] = -1;
            od[i] = __builtin_nanf("");
        }
    }
    if (tid == 0) out_count[q] = nout;
}

// per-item form: the group slot of every listed row (the sort key that brings a group's rows together, in row order), its position
__global__ void k_list_group_keys(const uint32_t *list, uint32_t m, const uint32_t *row_gidx, const float *weights, uint32_t *key, uint32_t *pos, float *w_sub) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    key[i] = row_gidx[list[i]];
    pos[i] = i;
    if (w_sub) w_sub[i] = weights[list[i]];
}
// sorted keys -> 1 where a new group starts
__global__ void k_group_heads(const uint32_t *key_s, uint32_t m, uint8_t *head) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) head[i] = i == 0 || key_s[i] != key_s[i - 1];
}
__global__ void k_sub_slots(const uint32_t *key_s, const uint32_t *sub_off, uint32_t n_sub, uint32_t m, uint32_t *sub_slot, uint32_t *sub_off_end) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_sub) sub_slot[j] = key_s[sub_off[j]];
    if (j == 0) *sub_off_end = m;  // the CSR's closing offset
}

// Candidate FILES (group slots, any order, m_f <= 8,192) -> the row list grouped by file with its CSR: one workgroup.  A file's rows
// keep their row order (grp_rows lists them so: SQLite's aggregates are order dependent); rows the mask leaves out are dropped.
__global__ __launch_bounds__(1024) void k_files_layout(const uint32_t *files, uint32_t m_f, const uint32_t *grp_off, const uint32_t *grp_rows, const uint8_t *mask,
                                                       const float *weights, uint32_t *sub_off, uint32_t *sub_slot, uint32_t *list, uint32_t *pos, float *w_sub, uint32_t list_cap) {
    __shared__ uint32_t s_part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (m_f + 1023) / 1024;
    const uint32_t i0 = tid * per, i1 = min(i0 + per, m_f);
    uint32_t mine = 0;
    for (uint32_t i = i0; i < i1; i++) {
        const uint32_t f = files[i];
        uint32_t c = 0;
        for (uint32_t e = grp_off[f]; e < grp_off[f + 1]; e++) c += (!mask || mask[grp_rows[e]]) ? 1u : 0u;
        mine += c;
    }
    s_part[tid] = mine;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {  // inclusive scan of the threads' totals
        const uint32_t v = tid >= off ? s_part[tid - off] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t at = s_part[tid] - mine;
    for (uint32_t i = i0; i < i1; i++) {
        const uint32_t f = files[i];
        sub_off[i] = at;
        sub_slot[i] = f;
        for (uint32_t e = grp_off[f]; e < grp_off[f + 1]; e++) {
            const uint32_t row = grp_rows[e];
            if (mask && !mask[row]) continue;
            if (at < list_cap) {
                list[at] = row;
                pos[at] = at;
                if (w_sub) w_sub[at] = weights[row];
            }
            at++;
        }
    }
    if (tid == 1023) sub_off[m_f] = s_part[1023];
}

constexpr uint32_t SPARSE_SORT_MAX = 8192;  // rows one in-LDS sort takes (64 KiB of records)
constexpr uint32_t SPARSE_SELECT_KMAX = 8192;  // pvs_select.hip's largest page
}  // namespace

// ---------------------------------------------------------------- NULL lists
// Built once per index state (rows, order keys) on first need, under a lock; searches only read them.
pvs_status pvs_ensure_null_rows(pvs_index *ix) {
    if (ix->null_built_n.load(std::memory_order_acquire) == ix->n && ix->null_built_epoch == ix->order_epoch) return PVS_OK;
    std::lock_guard<std::mutex> lk(ix->null_mu);
    if (ix->null_built_n.load(std::memory_order_acquire) == ix->n && ix->null_built_epoch == ix->order_epoch) return PVS_OK;
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t s = ix->admin_stream;
    uint32_t *d_cnt = nullptr, *d_key[2] = {nullptr, nullptr}, *d_row[2] = {nullptr, nullptr}, *d_key2 = nullptr;
    void *tmp = nullptr;
    const uint64_t n = ix->n;
    const uint32_t *trank = ix->order_rows == n && n ? ix->d_trank : nullptr;
    uint32_t h[4] = {0, 0, 0, 0};
    uint32_t *fresh[2] = {nullptr, nullptr};
    auto body = [&]() -> pvs_status {
        if (n == 0) return PVS_OK;
        HIP_TRY(pvs_scratch_alloc((void **)&d_cnt, 16));
        HIP_TRY(hipMemsetAsync(d_cnt, 0, 16, s));
        hipLaunchKernelGGL(k_null_count, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 2048)), dim3(256), 0, s, ix->d_norm2, n, d_cnt);
        HIP_TRY(hipMemcpyAsync(h, d_cnt, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        const uint32_t suspects = h[0];
        h[0] = 0;
        if (suspects == 0) return PVS_OK;  // the usual case: one pass over 4 bytes per row, nothing else
        for (int m = 0; m < 2; m++) {
            HIP_TRY(pvs_scratch_alloc((void **)&d_key[m], (size_t)suspects * 4));
            HIP_TRY(pvs_scratch_alloc((void **)&d_row[m], (size_t)suspects * 4));
        }
        HIP_TRY(pvs_scratch_alloc((void **)&d_key2, (size_t)suspects * 4));
        HIP_TRY(hipMemsetAsync(d_cnt, 0, 16, s));
        const dim3 g((unsigned)((n + 255) / 256));
#define PVS_NULL_CLASSIFY(DT) \
    hipLaunchKernelGGL(k_null_classify<DT>, g, dim3(256), 0, s, ix->d_rows, ix->stride, (int)ix->dim, ix->d_norm2, n, trank, d_cnt, d_key[0], d_row[0], d_key[1], d_row[1], suspects)
        if (ix->dtype == PVS_I8)
            PVS_NULL_CLASSIFY(PVS_I8);
        else if (ix->dtype == PVS_F16)
            PVS_NULL_CLASSIFY(PVS_F16);
        else
            PVS_NULL_CLASSIFY(PVS_F32);
#undef PVS_NULL_CLASSIFY
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h, d_cnt, 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (int m = 0; m < 2; m++) {
            if (!h[m]) continue;
            HIP_TRY(pvs_malloc_retry((void **)&fresh[m], (size_t)h[m] * 4));
            size_t tb = 0;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, d_key[m], d_key2, d_row[m], fresh[m], (int)h[m]));
            pvs_scratch_free_on(tmp, s);
            tmp = nullptr;
            HIP_TRY(pvs_scratch_alloc(&tmp, tb ? tb : 16));
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, d_key[m], d_key2, d_row[m], fresh[m], (int)h[m], 0, 32, s));  // tie order
            HIP_TRY(hipStreamSynchronize(s));
        }
        return PVS_OK;
    };
    pvs_status st = body();
    for (void *p : {(void *)d_cnt, (void *)d_key[0], (void *)d_row[0], (void *)d_key[1], (void *)d_row[1], (void *)d_key2, tmp}) pvs_scratch_free_on(p, s);
    if (st != PVS_OK) {
        hipFree(fresh[0]);
        hipFree(fresh[1]);
        return st;
    }
    // (no search holds a pointer to the old lists: searches fetch them after this function returned on their own thread, and
    //  pvs_index_add / set_order_keys — the only calls that invalidate them — are exclusive)
    for (int m = 0; m < 2; m++) {
        hipFree(ix->d_null_rows[m]);
        ix->d_null_rows[m] = fresh[m];
        ix->n_null[m] = h[m];
        ix->null_weird[m] = h[2 + m];
    }
    ix->null_built_epoch = ix->order_epoch;
    ix->null_built_n.store(n, std::memory_order_release);
    return PVS_OK;
}

// flags: device [nq] verdicts of pass C for the chunk whose queries are prepared in c (prep_chunk); queries with flag 3 get their
// tail and flag 0 (also in h_flags, the pinned mirror, when given)
pvs_status pvs_launch_null_tails(pvs_index *ix, SearchCtx &c, int metric, uint32_t *d_flags, uint32_t *h_flags, uint32_t nq, uint32_t k, int64_t *d_out_ids,
                                 float *d_out_dist, uint32_t *d_out_count) {
    const uint32_t *tinv = ix->order_rows == ix->n && ix->n ? ix->d_tinv : nullptr;
    const int m = metric == PVS_L2 ? 1 : 0;
    hipLaunchKernelGGL(k_null_tail, dim3(nq), dim3(256), 0, c.stream, d_flags, c.d_qinfo, metric, ix->d_null_rows[m], ix->n_null[m], tinv, ix->n, c.cur_mask,
                       ix->d_ids, k, d_out_ids, d_out_dist, d_out_count, h_flags);
    HIP_TRY(hipGetLastError());
    return PVS_OK;
}

// ---------------------------------------------------------------- candidate lists
// allowed rows of a device mask (a 4-byte read-back: the caller decides which path serves the query)
pvs_status pvs_mask_count(const uint8_t *d_mask, uint64_t n, uint32_t *out_count, hipStream_t s) {
    *out_count = 0;
    if (n == 0) return PVS_OK;
    uint32_t *d_cnt = nullptr;
    HIP_TRY(pvs_scratch_alloc((void **)&d_cnt, 4));
    hipError_t e = hipMemsetAsync(d_cnt, 0, 4, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_mask_count, dim3((unsigned)std::min<uint64_t>((n + 16383) / 16384, 512)), dim3(256), 0, s, d_mask, n, d_cnt);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_count, d_cnt, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    pvs_scratch_free(d_cnt);
    if (e != hipSuccess) return pvs_fail(PVS_ERR_DEVICE, "mask count: %s", hipGetErrorString(e));
    return PVS_OK;
}
// the allowed rows in ascending order: d_list [count] (count = pvs_mask_count's answer); stream-ordered
pvs_status pvs_mask_compact(const uint8_t *d_mask, uint64_t n, uint32_t *d_list, uint32_t count, hipStream_t s) {
    if (n == 0 || count == 0) return PVS_OK;
    if (n > 0x7fffffffull) return pvs_fail(PVS_ERR_UNSUPPORTED, "mask compaction is limited to 2^31-1 rows");
    uint32_t *d_num = nullptr;
    void *tmp = nullptr;
    auto body = [&]() -> pvs_status {
        HIP_TRY(pvs_scratch_alloc((void **)&d_num, 4));
        hipcub::CountingInputIterator<uint32_t> it(0);
        hipcub::TransformInputIterator<bool, NonZero, const uint8_t *> fl(d_mask, NonZero());
        size_t tb = 0;
        HIP_TRY(hipcub::DeviceSelect::Flagged(nullptr, tb, it, fl, d_list, d_num, (int)n, s));
        HIP_TRY(pvs_scratch_alloc(&tmp, tb ? tb : 16));
        HIP_TRY(hipcub::DeviceSelect::Flagged(tmp, tb, it, fl, d_list, d_num, (int)n, s));
        return PVS_OK;
    };
    pvs_status st = body();
    pvs_scratch_free_on(d_num, s);
    pvs_scratch_free_on(tmp, s);
    return st;
}

// a (long) candidate list as the byte mask the filter scan takes; validates the list (ascending, inside the index); synchronous
pvs_status pvs_list_to_mask(const uint32_t *d_list, uint32_t m, uint64_t n, uint8_t *d_mask, hipStream_t s) {
    uint32_t *d_flag = nullptr;
    HIP_TRY(pvs_scratch_alloc((void **)&d_flag, 4));
    uint32_t flag = 0;
    auto body = [&]() -> pvs_status {
        HIP_TRY(hipMemsetAsync(d_flag, 0, 4, s));
        HIP_TRY(hipMemsetAsync(d_mask, 0, n, s));
        if (m) {
            hipLaunchKernelGGL(k_list_check, dim3((m + 255) / 256), dim3(256), 0, s, d_list, m, n, d_flag);
            hipLaunchKernelGGL(k_list_scatter_mask, dim3((m + 255) / 256), dim3(256), 0, s, d_list, m, n, d_mask);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        return PVS_OK;
    };
    pvs_status st = body();
    pvs_scratch_free_on(d_flag, s);
    PVS_TRY(st);
    if (flag & 2u) return pvs_fail(PVS_ERR_INVALID_ARG, "candidate rows must be row positions below the index's row count (%llu)", (unsigned long long)n);
    if (flag & 1u) return pvs_fail(PVS_ERR_INVALID_ARG, "candidate rows must be strictly ascending");
    return PVS_OK;
}

// does the gather-and-score path take a list of m rows for `batch` queries at page size k?
bool pvs_sparse_eligible(const pvs_index *ix, uint64_t m, uint32_t batch, uint32_t k) {
    if (pvs_dbg(PVS_DBG_NO_SPARSE)) return false;
    if (m > 0xffffffffull) return false;
    if (m > SPARSE_SORT_MAX && k > SPARSE_SELECT_KMAX) return false;
    // A pair costs one row read by one lane in the reference's order (~2-3k VALU, a row's bytes mostly out of L2 when queries
    // share it); the corpus pass it replaces costs N rows at HBM speed: the crossover is ~N/32 pairs (and a floor below which the
    // corpus pass's fixed ~0.1 ms of five launches is already more)
    const uint64_t lim = pvs_dbg(PVS_DBG_SPARSE_MAX) > 0 ? (uint64_t)pvs_dbg(PVS_DBG_SPARSE_MAX) : std::max<uint64_t>(ix->n / 32, 16384);
    return m * std::max<uint32_t>(batch, 1) <= lim;
}

// The page of every query over the rows of d_list (device, strictly ascending row positions; validated): gather-and-score +
// sort / select, enqueued on c.stream and waited for.  Outputs are device buffers [batch][k].
pvs_status pvs_sparse_search(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t batch, uint32_t k, int metric, const uint32_t *d_list,
                             uint32_t m, int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count) {
    hipStream_t s = c.stream;
    if (m == 0) {
        HIP_TRY(hipMemsetAsync(d_out_count, 0, 4 * (size_t)batch, s));
        HIP_TRY(hipMemsetAsync(d_out_ids, 0xff, 8 * (size_t)batch * k, s));
        HIP_TRY(pvs_launch_fill_f32(d_out_dist, (uint64_t)batch * k, __builtin_nanf(""), s));
        HIP_TRY(hipStreamSynchronize(s));
        ix->sparse_queries += batch;
        return PVS_OK;
    }
    const bool keyed = ix->order_rows == ix->n && ix->n;
    const bool big = m > SPARSE_SORT_MAX;
    float *d_m = nullptr;
    uint32_t *d_flag = nullptr, *d_skey = nullptr, *d_spos = nullptr, *d_skey2 = nullptr, *d_stinv = nullptr;
    int64_t *d_sids = nullptr;
    void *tmp = nullptr;
    // the validity flag of the list: a word of the context's pinned block (written by the scorer, read here after the one
    // synchronisation at the end: no copy, no extra round trip)
    PVS_TRY(ctx_pinned_io(c, 64));
    volatile uint32_t *h_flag = (volatile uint32_t *)c.h_io;  // word 0: unsorted, word 1: beyond the index (one word per condition: plain stores)
    h_flag[0] = 0;
    h_flag[1] = 0;
    auto bad_list = [&](uint32_t flag) -> pvs_status {
        if (flag & 2u) return pvs_fail(PVS_ERR_INVALID_ARG, "candidate rows must be row positions below the index's row count (%llu)", (unsigned long long)ix->n);
        if (flag & 1u) return pvs_fail(PVS_ERR_INVALID_ARG, "candidate rows must be strictly ascending");
        return PVS_OK;
    };
    auto body = [&]() -> pvs_status {
        if (big) {  // (the select walks ids and tie order of the list: validate before those gathers)
            HIP_TRY(pvs_scratch_alloc((void **)&d_flag, 4));
            HIP_TRY(hipMemsetAsync(d_flag, 0, 4, s));
            hipLaunchKernelGGL(k_list_check, dim3((m + 255) / 256), dim3(256), 0, s, d_list, m, ix->n, d_flag);
            uint32_t flag = 0;
            HIP_TRY(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            PVS_TRY(bad_list(flag));
            HIP_TRY(pvs_scratch_alloc((void **)&d_sids, (size_t)m * 8));
            if (keyed) {
                HIP_TRY(pvs_scratch_alloc((void **)&d_skey, (size_t)m * 4));
                HIP_TRY(pvs_scratch_alloc((void **)&d_spos, (size_t)m * 4));
                HIP_TRY(pvs_scratch_alloc((void **)&d_skey2, (size_t)m * 4));
                HIP_TRY(pvs_scratch_alloc((void **)&d_stinv, (size_t)m * 4));
            }
            hipLaunchKernelGGL(k_list_gather_ids, dim3((m + 255) / 256), dim3(256), 0, s, d_list, m, ix->d_ids, keyed ? ix->d_trank : nullptr, d_sids, d_skey, d_spos);
            if (keyed) {  // list positions in tie order: what the select walks instead of 0..m-1
                size_t tb = 0;
                HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, d_skey, d_skey2, d_spos, d_stinv, (int)m));
                HIP_TRY(pvs_scratch_alloc(&tmp, tb ? tb : 16));
                HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, d_skey, d_skey2, d_spos, d_stinv, (int)m, 0, 32, s));
            }
        }
        const uint32_t chunk = std::min<uint32_t>(batch, PVS_MAX_BATCH);
        HIP_TRY(pvs_scratch_alloc((void **)&d_m, (size_t)m * chunk * 4));
        const uint32_t qbytes = ix->dim * (ix->dtype == PVS_I8 ? 1u : 4u), qpad = (qbytes + 63u) & ~63u;
        for (uint32_t q0 = 0; q0 < batch; q0 += chunk) {
            const uint32_t nb = std::min(chunk, batch - q0);
            const uint32_t pad = nb <= 32 ? 32 : nb <= 64 ? 64 : 128;
            PVS_TRY(prep_chunk(ix, c, d_queries, qdtype, q0, nb, pad, metric));
            uint32_t qt = 1;  // queries per workgroup: a power of two, <= 32 KiB of LDS, <= 64
            while (qt * 2 <= nb && qt * 2 * qpad <= 32768 && qt * 2 <= 64) qt *= 2;
            const dim3 g((m + 256 / qt - 1) / (256 / qt), (nb + qt - 1) / qt);
            const size_t lds = (size_t)qt * qpad;
#define PVS_SPARSE_SCORE(DT)                                                                                                                                  \
    hipLaunchKernelGGL(k_sparse_score<DT>, g, dim3(256), lds, s, ix->d_rows, ix->stride, (int)ix->dim, metric, ix->d_norm2, d_list, m, ix->n, c.d_qexact, c.d_qinfo, \
                       nb, qt, qpad, d_m, nb, (uint32_t *)c.h_io)
            if (ix->dtype == PVS_I8)
                PVS_SPARSE_SCORE(PVS_I8);
            else if (ix->dtype == PVS_F16)
                PVS_SPARSE_SCORE(PVS_F16);
            else
                PVS_SPARSE_SCORE(PVS_F32);
#undef PVS_SPARSE_SCORE
            HIP_TRY(hipGetLastError());
            int64_t *oi = d_out_ids + (size_t)q0 * k;
            float *od = d_out_dist + (size_t)q0 * k;
            uint32_t *oc = d_out_count + q0;
            if (!big) {
                uint32_t m2 = 1;
                while (m2 < m) m2 <<= 1;
                static std::atomic<bool> configured{false};
                if (!configured.load(std::memory_order_acquire)) {
                    HIP_TRY(hipFuncSetAttribute((const void *)k_sparse_sort, hipFuncAttributeMaxDynamicSharedMemorySize, SPARSE_SORT_MAX * 8));
                    configured.store(true, std::memory_order_release);
                }
                hipLaunchKernelGGL(k_sparse_sort, dim3(nb), dim3(256), (size_t)m2 * 8, s, d_m, nb, d_list, m, ix->n, ix->d_ids, keyed ? ix->d_trank : nullptr,
                                   keyed ? ix->d_tinv : nullptr, k, oi, od, oc);
                HIP_TRY(hipGetLastError());
            } else {
                PVS_TRY(pvs_select_topk(d_m, m, nb, nb, k, nullptr, d_sids, nullptr, oi, od, oc, s, keyed ? d_stinv : nullptr));
            }
        }
        HIP_TRY(hipStreamSynchronize(s));
        return bad_list(h_flag[0] | h_flag[1]);
    };
    pvs_status st = body();
    if (st != PVS_OK) (void)hipStreamSynchronize(s);
    for (void *p : {(void *)d_m, (void *)d_flag, (void *)d_skey, (void *)d_spos, (void *)d_skey2, (void *)d_stinv, (void *)d_sids, tmp}) pvs_scratch_free(p);
    if (st == PVS_OK) ix->sparse_queries += batch;
    return st;
}

// The per-item page over the rows of d_list (device, ascending, from a compacted candidate mask): gather-and-score, the listed rows
// brought together per group (a stable sort by group slot keeps a group's rows in row order: SQLite's aggregates are order
// dependent), k_group_aggregate over that CSR, one LDS sort per query column.  *handled = false: more sub-groups than one LDS
// sort takes, or massive ties at the page's edge — the caller runs the corpus pass instead.  Outputs: host arrays [batch][k].
pvs_status pvs_sparse_search_groups(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t batch, uint32_t k, int metric, int agg,
                                    const float *d_weights, const uint32_t *d_list, uint32_t m, int64_t *out_groups, double *out_values, uint32_t *out_count,
                                    bool *handled) {
    *handled = false;
    hipStream_t s = c.stream;
    if (m == 0) {
        for (size_t i = 0; i < (size_t)batch * k; i++) {
            out_groups[i] = -1;
            out_values[i] = __builtin_nan("");
        }
        for (uint32_t q = 0; q < batch; q++) out_count[q] = 0;
        *handled = true;
        ix->sparse_queries += batch;
        return PVS_OK;
    }
    if (!ix->d_row_gidx) return PVS_OK;
    uint32_t *d_key = nullptr, *d_pos = nullptr, *d_key_s = nullptr, *d_pos_s = nullptr, *d_sub_off = nullptr, *d_sub_slot = nullptr, *d_num = nullptr;
    uint8_t *d_head = nullptr;
    float *d_wsub = nullptr, *d_m = nullptr;
    double *d_vals = nullptr;
    void *tmp = nullptr, *d_work = nullptr;
    const bool keyed = ix->d_grp_tinv && ix->d_grp_trank;
    auto body = [&]() -> pvs_status {
        for (uint32_t **p : {&d_key, &d_pos, &d_key_s, &d_pos_s}) HIP_TRY(pvs_scratch_alloc((void **)p, (size_t)m * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_sub_off, ((size_t)m + 1) * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_sub_slot, (size_t)m * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_head, m));
        HIP_TRY(pvs_scratch_alloc((void **)&d_num, 4));
        if (d_weights) HIP_TRY(pvs_scratch_alloc((void **)&d_wsub, (size_t)m * 4));
        hipLaunchKernelGGL(k_list_group_keys, dim3((m + 255) / 256), dim3(256), 0, s, d_list, m, ix->d_row_gidx, d_weights, d_key, d_pos, d_wsub);
        size_t tb = 0, tb2 = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, d_key, d_key_s, d_pos, d_pos_s, (int)m));
        hipcub::CountingInputIterator<uint32_t> it(0);
        HIP_TRY(hipcub::DeviceSelect::Flagged(nullptr, tb2, it, d_head, d_sub_off, d_num, (int)m, s));
        HIP_TRY(pvs_scratch_alloc(&tmp, std::max<size_t>(std::max(tb, tb2), 16)));
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, d_key, d_key_s, d_pos, d_pos_s, (int)m, 0, 32, s));  // stable: positions ascending inside a group
        hipLaunchKernelGGL(k_group_heads, dim3((m + 255) / 256), dim3(256), 0, s, d_key_s, m, d_head);
        HIP_TRY(hipcub::DeviceSelect::Flagged(tmp, tb2, it, d_head, d_sub_off, d_num, (int)m, s));
        uint32_t n_sub = 0;
        HIP_TRY(hipMemcpyAsync(&n_sub, d_num, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (!pvs_sub_rank_supported(n_sub)) return PVS_OK;  // (not handled)
        hipLaunchKernelGGL(k_sub_slots, dim3((n_sub + 255) / 256), dim3(256), 0, s, d_key_s, d_sub_off, n_sub, m, d_sub_slot, d_sub_off + n_sub);
        const uint32_t chunk = std::min<uint32_t>(batch, PVS_MAX_BATCH);
        HIP_TRY(pvs_scratch_alloc((void **)&d_m, (size_t)m * chunk * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_vals, (size_t)n_sub * chunk * 8));
        HIP_TRY(pvs_scratch_alloc(&d_work, pvs_gm_rank_work_bytes(chunk)));
        const size_t off_g = 64, off_v = off_g + (size_t)chunk * k * 8, off_f = off_v + (size_t)chunk * k * 8, off_c = off_f + (size_t)chunk * 4, need = off_c + (size_t)chunk * 4;
        if (need != pvs_group_pages_bytes(chunk, k)) return pvs_fail(PVS_ERR_STATE, "per-item page layout and pvs_group_pages_bytes disagree");
        PVS_TRY(ctx_pinned_io(c, need));
        const uint32_t qbytes = ix->dim * (ix->dtype == PVS_I8 ? 1u : 4u), qpad = (qbytes + 63u) & ~63u;
        ((volatile uint32_t *)c.h_io)[0] = 0;  // (the scorer's list-validity words
        ((volatile uint32_t *)c.h_io)[1] = 0;  //: the list comes from a compacted mask, it stays 0)
        for (uint32_t q0 = 0; q0 < batch; q0 += chunk) {
            const uint32_t nb = std::min(chunk, batch - q0);
            const uint32_t pad = nb <= 32 ? 32 : nb <= 64 ? 64 : 128;
            PVS_TRY(prep_chunk(ix, c, d_queries, qdtype, q0, nb, pad, metric));
            uint32_t qt = 1;
            while (qt * 2 <= nb && qt * 2 * qpad <= 32768 && qt * 2 <= 64) qt *= 2;
            const dim3 g((m + 256 / qt - 1) / (256 / qt), (nb + qt - 1) / qt);
            const size_t lds = (size_t)qt * qpad;
#define PVS_SPARSE_SCORE(DT)                                                                                                                                  \
    hipLaunchKernelGGL(k_sparse_score<DT>, g, dim3(256), lds, s, ix->d_rows, ix->stride, (int)ix->dim, metric, ix->d_norm2, d_list, m, ix->n, c.d_qexact, c.d_qinfo, \
                       nb, qt, qpad, d_m, nb, (uint32_t *)c.h_io)
            if (ix->dtype == PVS_I8)
                PVS_SPARSE_SCORE(PVS_I8);
            else if (ix->dtype == PVS_F16)
                PVS_SPARSE_SCORE(PVS_F16);
            else
                PVS_SPARSE_SCORE(PVS_F32);
#undef PVS_SPARSE_SCORE
            HIP_TRY(hipGetLastError());
            HIP_TRY(pvs_launch_group_aggregate(d_m, nb, nb, 0, d_sub_off, d_pos_s, n_sub, d_wsub, nullptr, agg, d_vals, s));
            HIP_TRY(pvs_sub_rank(d_vals, n_sub, nb, k, d_sub_slot, ix->d_grp_ids, keyed ? ix->d_grp_trank : nullptr, keyed ? ix->d_grp_tinv : nullptr, d_work,
                                 (int64_t *)(c.h_io + off_g), (double *)(c.h_io + off_v), (uint32_t *)(c.h_io + off_f), (uint32_t *)(c.h_io + off_c), s));
            HIP_TRY(hipStreamSynchronize(s));
            const uint32_t *fl = (const uint32_t *)(c.h_io + off_f), *cn = (const uint32_t *)(c.h_io + off_c);
            for (uint32_t q = 0; q < nb; q++)
                if (!fl[q]) return PVS_OK;  // (not handled: the caller's corpus pass answers the whole call)
            memcpy(out_groups + (size_t)q0 * k, c.h_io + off_g, (size_t)nb * k * 8);
            memcpy(out_values + (size_t)q0 * k, c.h_io + off_v, (size_t)nb * k * 8);
            memcpy(out_count + q0, cn, (size_t)nb * 4);
        }
        *handled = true;
        return PVS_OK;
    };
    pvs_status st = body();
    if (st != PVS_OK) (void)hipStreamSynchronize(s);
    for (void *p : {(void *)d_key, (void *)d_pos, (void *)d_key_s, (void *)d_pos_s, (void *)d_sub_off, (void *)d_sub_slot, (void *)d_head, (void *)d_num, (void *)d_wsub,
                    (void *)d_m, (void *)d_vals, tmp, d_work})
        pvs_scratch_free(p);
    if (st == PVS_OK && *handled) ix->sparse_queries += batch;
    return st;
}


// The per-item pages of `nb` queries over the rows of m_f candidate FILES (d_files: group slots, any order, no duplicates; m_rows =
// their rows the mask allows): the lean form of pvs_sparse_search_groups for the certified float route (pvs_items_float.hip) — the
// list is laid out grouped by file by one workgroup (no sort, no select, one synchronisation), then gather-and-score with the
// reference's in-order chain, SQLite's aggregates per file, one LDS ranking per query column.  *handled = false: more files than
// one LDS ranking takes, or massive ties at a page's edge.  skip[q] != 0: that column's verdict is not looked at (its page is
// garbage the caller overwrites).  Outputs: host arrays [nb][k].
pvs_status pvs_sparse_groups_of_files(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t nb, uint32_t k, int metric, int agg, const float *d_weights,
                                      const uint8_t *d_mask, const uint32_t *d_files, uint32_t m_f, uint32_t m_rows, const uint8_t *skip, int64_t *out_groups,
                                      double *out_values, uint32_t *out_count, bool *handled) {
    *handled = false;
    hipStream_t s = c.stream;
    if (m_f == 0 || m_rows == 0 || !pvs_sub_rank_supported(m_f) || nb > PVS_MAX_BATCH) return PVS_OK;
    uint32_t *d_sub_off = nullptr, *d_sub_slot = nullptr, *d_list = nullptr, *d_pos = nullptr;
    float *d_wsub = nullptr, *d_m = nullptr;
    double *d_vals = nullptr;
    void *d_work = nullptr;
    const bool keyed = ix->d_grp_tinv && ix->d_grp_trank;
    auto body = [&]() -> pvs_status {
        HIP_TRY(pvs_scratch_alloc((void **)&d_sub_off, ((size_t)m_f + 1) * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_sub_slot, (size_t)m_f * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_list, (size_t)m_rows * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_pos, (size_t)m_rows * 4));
        if (d_weights) HIP_TRY(pvs_scratch_alloc((void **)&d_wsub, (size_t)m_rows * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_m, (size_t)m_rows * nb * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_vals, (size_t)m_f * nb * 8));
        HIP_TRY(pvs_scratch_alloc(&d_work, pvs_gm_rank_work_bytes(nb)));
        hipLaunchKernelGGL(k_files_layout, dim3(1), dim3(1024), 0, s, d_files, m_f, ix->d_grp_off, ix->d_grp_rows, d_mask, d_weights, d_sub_off, d_sub_slot, d_list, d_pos, d_wsub, m_rows);
        HIP_TRY(hipGetLastError());
        const size_t off_g = 64, off_v = off_g + (size_t)nb * k * 8, off_f = off_v + (size_t)nb * k * 8, off_c = off_f + (size_t)nb * 4, need = off_c + (size_t)nb * 4;
        if (need != pvs_group_pages_bytes(nb, k)) return pvs_fail(PVS_ERR_STATE, "per-item page layout and pvs_group_pages_bytes disagree");
        PVS_TRY(ctx_pinned_io(c, need));
        const uint32_t qbytes = ix->dim * (ix->dtype == PVS_I8 ? 1u : 4u), qpad = (qbytes + 63u) & ~63u;
        const uint32_t pad = nb <= 32 ? 32 : nb <= 64 ? 64 : 128;
        PVS_TRY(prep_chunk(ix, c, d_queries, qdtype, 0, nb, pad, metric));
        uint32_t qt = 1;
        while (qt * 2 <= nb && qt * 2 * qpad <= 32768 && qt * 2 <= 64) qt *= 2;
        const dim3 g((m_rows + 256 / qt - 1) / (256 / qt), (nb + qt - 1) / qt);
        const size_t lds = (size_t)qt * qpad;
        // (the scorer's list-validity words: this list is grouped by file, not ascending — its "unsorted" word is not looked at; every
        //  row comes out of the index's own CSR, none lies beyond the index)
#define PVS_SPARSE_SCORE(DT)                                                                                                                                       \
    hipLaunchKernelGGL(k_sparse_score<DT>, g, dim3(256), lds, s, ix->d_rows, ix->stride, (int)ix->dim, metric, ix->d_norm2, d_list, m_rows, ix->n, c.d_qexact, c.d_qinfo, \
                       nb, qt, qpad, d_m, nb, (uint32_t *)c.h_io)
        if (ix->dtype == PVS_I8)
            PVS_SPARSE_SCORE(PVS_I8);
        else if (ix->dtype == PVS_F16)
            PVS_SPARSE_SCORE(PVS_F16);
        else
            PVS_SPARSE_SCORE(PVS_F32);
#undef PVS_SPARSE_SCORE
        HIP_TRY(hipGetLastError());
        HIP_TRY(pvs_launch_group_aggregate(d_m, nb, nb, 0, d_sub_off, d_pos, m_f, d_wsub, nullptr, agg, d_vals, s));
        HIP_TRY(pvs_sub_rank(d_vals, m_f, nb, k, d_sub_slot, ix->d_grp_ids, keyed ? ix->d_grp_trank : nullptr, keyed ? ix->d_grp_tinv : nullptr, d_work,
                             (int64_t *)(c.h_io + off_g), (double *)(c.h_io + off_v), (uint32_t *)(c.h_io + off_f), (uint32_t *)(c.h_io + off_c), s));
        HIP_TRY(hipStreamSynchronize(s));
        const uint32_t *fl = (const uint32_t *)(c.h_io + off_f), *cn = (const uint32_t *)(c.h_io + off_c);
        for (uint32_t q = 0; q < nb; q++)
            if (!fl[q] && !(skip && skip[q])) return PVS_OK;  // (not handled: the caller's corpus pass answers the chunk)
        memcpy(out_groups, c.h_io + off_g, (size_t)nb * k * 8);
        memcpy(out_values, c.h_io + off_v, (size_t)nb * k * 8);
        memcpy(out_count, cn, (size_t)nb * 4);
        *handled = true;
        return PVS_OK;
    };
    pvs_status st = body();
    if (st != PVS_OK) (void)hipStreamSynchronize(s);
    for (void *p : {(void *)d_sub_off, (void *)d_sub_slot, (void *)d_list, (void *)d_pos, (void *)d_wsub, (void *)d_m, (void *)d_vals, d_work}) pvs_scratch_free(p);
    return st;
}
