// pvs_items_float.hip — per-item pages over FLOAT rows without computing every distance exactly: bound, certify, rescan.
//
// The reference's exact mode is f32 (filters/exact.rs:106-165, image_embeddings.rs:412-438) and similar_to defaults to AVG
// (item_similarity.rs:432-581): `GROUP BY file_id` + rank_aggregate over the `d` column, then ORDER BY ... LIMIT k — a page of k
// files out of millions.  Until round 6 the build answered that for f16 / f32 indexes by running the reference's in-order f32
// chain on EVERY (row, query) pair (k_exact_wide: 0.18 of the HBM rate the rows stream at, bound by the packed-f32 VALU rate:
// 98 M exact chains to return 32 pages of 100 files).  The row search never did that: it filters on the matrix cores with
// rigorous error bounds and rescans only survivors.  This file does the same for per-item pages:
//
//   1. k_scan MODE 4 (pvs_scan_kernel.hpp): one corpus pass on the matrix cores writes the scan KEY of every (row, query) pair,
//      |key - kappa| <= err = eA + eR |a|^2, kappa = the reference key the distance D is a monotone function of (cosine:
//      D = fl32(1 + kappa / sqrt(bb)); L2: D = fl32(sqrt(kappa)); HISTORY.md section 4.2 — the algebra passes A/B/C rest on);
//   2. k_group_bounds: per (file, query) the bracket [lo_i, hi_i] of every row's distance is folded into a bracket [L, U] of the
//      file's aggregate — AVG: means of the brackets; MIN / MAX: min / max of the ends; SUM(d w)/SUM(w) with positive weights: the
//      weighted means — widened for every rounding on the way.  A file with a row whose distance may be NULL (zero / non-finite
//      norm, non-finite key) or a non-positive weight is FORCED: L = -inf, it never lowers the threshold.  U goes into one of
//      16,384 per-query buckets (atomic minimum); L is stored;
//   3. k_kth (pass A's select): T = the k-th smallest bucket minimum >= the k-th smallest U >= the k-th smallest exact value;
//   4. k_candidates / k_union: every file with L <= T for a query is that query's candidate (every file of its true page is one:
//      exact value <= the k-th exact value <= T, and L <= exact); a query that names more than 1,024 ties with everything and is
//      set aside; the union of the others' candidates is the file list of the chunk;
//   5. pvs_sparse_groups_of_files (pvs_sparse.hip): the exact in-order chain on the rows of those files only, SQLite's KBN sums per
//      file, ranking under the page order (value, [order key,] file id, NULL last) — for every query over the UNION of the
//      candidates, a superset of each query's own, all of it exact: the page is the reference's, bit for bit.
//
// Whatever cannot be certified — too many candidates (ties, a page deeper than the files with a finite bracket, a NULL query,
// non-finite components), k beyond what the buckets resolve — is handed back (*handled = false) and the exact-everywhere route
// (k_exact_wide + k_group_aggregate8) answers as before.  pvs_score_all / pvs_score_batch (the SQL seam's full `d` column) keep
// the exact kernels: they return every distance.
#include "pvs_index.hpp"

namespace {
constexpr uint32_t BUCKETS = 16384;  // per query: minima of U over disjoint sets of files (file index mod BUCKETS) -> k_kth

struct BoundsK {
    const float *keys;       // [n][ld]
    uint32_t ld, nb;
    const QInfo *qinfo;
    const float *norm2;      // [n] |a|^2 (sequential f32: the reference's aMag)
    const uint32_t *grp_off, *grp_rows;
    uint32_t n_groups;
    const float *weights;    // optional [n]
    const uint8_t *mask;     // optional [n]: 0 = the row takes part in nothing
    int metric, agg;
    float *lo;               // [n_groups][nb]
    uint32_t *bucket_min;    // [nb][BUCKETS] bit patterns of non-negative floats
    uint32_t *bad_query;     // [nb] 1: nothing of this query can be bracketed (every distance NULL, a non-finite norm): it names no candidate
                             // and the caller answers it through the exact-everywhere route
};

// one thread per (file, query), query fastest: the 32 keys of a row are one 128-byte line.  Per-row arithmetic in f32 (every
// operation's rounding is inside the 1e-6 (1 + |d|) the brackets are widened by: five operations of 6e-8 each plus the half ulp of the
// reference's own f32 distance), sums of brackets in f64; no division or square root in f64 (42 M threads: 0.95 ms with them, round 6).
__global__ __launch_bounds__(256) void k_group_bounds(BoundsK a) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t q = (uint32_t)(t % a.nb);
    const uint64_t f = t / a.nb;
    if (f >= a.n_groups) return;
    const QInfo qi = a.qinfo[q];
    const bool cosine = a.metric == PVS_COSINE;
    // a query whose distances are all NULL, or whose norm is not finite: nothing can be bracketed — it names no candidate here and is
    // answered by the exact-everywhere route (its page is the first k files in (order key, id) order, all NULL, or worse)
    const bool q_ok = qi.bb == qi.bb && qi.bb < __builtin_inff() && (!cosine || qi.bb > 0.f) && qi.dscale > 0.f;
    if (f == 0) a.bad_query[q] = q_ok ? 0u : 1u;
    if (!q_ok) {
        a.lo[f * a.nb + q] = __builtin_nanf("");  // (compares false against any threshold, +inf included)
        return;
    }
    const float inv_sb = cosine ? 1.0f / sqrtf(qi.bb) : 0.f;  // (two roundings: inside the margin)
    const uint32_t e0 = a.grp_off[f], e1 = a.grp_off[f + 1];
    const bool weighted = a.weights != nullptr;
    const bool want_min = a.agg == PVS_AGG_MIN, want_max = a.agg == PVS_AGG_MAX;
    double s_lo = 0.0, s_hi = 0.0, s_w = 0.0;
    float x_lo = want_min ? __builtin_inff() : -__builtin_inff(), x_hi = x_lo;
    uint32_t cnt = 0;
    bool forced = false;
    for (uint32_t e = e0; e < e1; e++) {
        const uint32_t row = a.grp_rows[e];
        if (a.mask && !a.mask[row]) continue;
        cnt++;
        const float aa = a.norm2[row];
        const float key = a.keys[(size_t)row * a.ld + q];
        // a row whose distance may be NULL or infinite (cosine: zero vector, |a|^2 under / overflow; NaN / inf components), or whose
        // key is not a number: no bracket.  (L2: a zero vector is an ordinary row at distance |q|.)
        if (!((cosine ? aa > 1e-30f : aa >= 0.f) && aa < 1e30f) || !(key == key) || fabsf(key) > 1e30f) {
            forced = true;
            continue;
        }
        float lo, hi;
        if (cosine) {
            lo = 1.0f + (key - qi.eA) * inv_sb;
            hi = 1.0f + (key + qi.eA) * inv_sb;
        } else {
            const float err = qi.eA + qi.eR * aa;
            lo = sqrtf(fmaxf(key - err, 0.f));
            hi = sqrtf(fmaxf(key + err, 0.f));
        }
        lo -= 1e-6f * (1.0f + fabsf(lo));
        hi += 1e-6f * (1.0f + fabsf(hi));
        if (weighted) {
            const float w = a.weights[row];
            if (!(w > 0.f && w < 1e30f)) {
                forced = true;
                continue;
            }
            s_lo += (double)lo * (double)w;
            s_hi += (double)hi * (double)w;
            s_w += (double)w;
        } else if (want_min) {
            x_lo = fminf(x_lo, lo);
            x_hi = fminf(x_hi, hi);
        } else if (want_max) {
            x_lo = fmaxf(x_lo, lo);
            x_hi = fmaxf(x_hi, hi);
        } else {
            s_lo += (double)lo;
            s_hi += (double)hi;
        }
    }
    float L, U;
    if (cnt == 0) {  // no candidate row: the file is not part of the result at all
        L = __builtin_inff();
        U = __builtin_inff();
    } else if (forced) {
        L = -__builtin_inff();
        U = __builtin_inff();
    } else {
        if (want_min || want_max) {
            L = x_lo;
            U = x_hi;
        } else {
            // (the f64 sums are exact to 1e-16 relative per term, SQLite's compensated sum within an ulp of the true one; the f32
            //  reciprocal, product and conversion: three roundings — 1e-6 relative covers all of it many times over)
            const float inv = 1.0f / (weighted ? (float)s_w : (float)cnt);
            L = (float)s_lo * inv;
            U = (float)s_hi * inv;
            L -= 1e-6f * (1.0f + fabsf(L));
            U += 1e-6f * (1.0f + fabsf(U));
        }
        if (!(L == L) || !(U == U)) {
            L = -__builtin_inff();
            U = __builtin_inff();
        }
    }
    a.lo[f * a.nb + q] = L;
    if (U < __builtin_inff()) {
        if (U < 0.f) U = 0.f;  // (raising an upper bound keeps it one; non-negative floats order like their bit patterns)
        uint32_t *slot = a.bucket_min + (size_t)q * BUCKETS + (uint32_t)(f % BUCKETS);
        const uint32_t ub = __builtin_bit_cast(uint32_t, U);
        if (ub < *(volatile uint32_t *)slot) atomicMin(slot, ub);  // (the minimum settles after a few files per bucket: most threads only look)
    }
}

constexpr uint32_t QCAP = 1024;   // candidate files one query may name before it counts as "cannot be certified" (ties with everything)
constexpr uint32_t UCAP = 8192;   // files in the union of a chunk's candidates (one LDS ranking per query column: pvs_sub_rank)

// one thread per (file, query): a candidate (L <= T) is appended to its query's list
__global__ __launch_bounds__(256) void k_candidates(const float *lo, uint32_t nb, uint32_t n_groups, const float *thr, uint32_t *qcnt, uint32_t *qlist) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (uint64_t)n_groups * nb) return;
    const uint32_t q = (uint32_t)(t % nb);
    if (!(lo[t] <= thr[q])) return;
    if (*(volatile uint32_t *)(qcnt + q) > 4 * QCAP) return;  // (already beyond saving: the count only has to say so)
    const uint32_t slot = atomicAdd(qcnt + q, 1u);
    if (slot < QCAP) qlist[(size_t)q * QCAP + slot] = (uint32_t)(t / nb);
}
// the union of the lists of the queries that stayed below QCAP: a file enters once (a bit per file), with its allowed rows counted
__global__ __launch_bounds__(256) void k_union(const uint32_t *qcnt, const uint32_t *qlist, uint32_t nb, uint32_t *bits, const uint32_t *grp_off, const uint32_t *grp_rows,
                                               const uint8_t *mask, uint32_t *ucnt, uint32_t *ufiles) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, q = blockIdx.y;
    const uint32_t c = qcnt[q];
    if (c > QCAP || i >= c) return;
    const uint32_t f = qlist[(size_t)q * QCAP + i];
    const uint32_t bit = 1u << (f & 31u);
    if (atomicOr(bits + (f >> 5), bit) & bit) return;
    const uint32_t slot = atomicAdd(ucnt, 1u);
    if (slot < UCAP) ufiles[slot] = f;
    uint32_t rows = 0;
    for (uint32_t e = grp_off[f]; e < grp_off[f + 1]; e++) rows += (!mask || mask[grp_rows[e]]) ? 1u : 0u;
    atomicAdd(ucnt + 1, rows);
}
}  // namespace

bool pvs_float_certify_applies(const pvs_index *ix, uint32_t nb, uint32_t k) {
    if (pvs_dbg(PVS_DBG_NO_FLOAT_CERTIFY) || ix->forced_path == 1) return false;
    if (ix->dtype == PVS_I8 || ix->n == 0 || ix->n_groups == 0 || !ix->d_row_gidx) return false;
    if (!pvs_scan_supported((int)ix->dtype, ix->stride / PVS_KSLAB_BYTES)) return false;
    if (nb > 128) return false;
    // the buckets must resolve the k-th smallest upper bound: many more buckets (and files) than k
    if ((uint64_t)k * 16 > BUCKETS || (uint64_t)k * 64 > ix->n_groups) return false;
    // a handful of rows: the exact kernels are cheaper than five launches
    return ix->n >= 16384;
}

// The queries [q0, q0 + nb) of d_queries were prepared in c (prep_chunk with batch_pad).  d_keys: scratch of >= n * nb floats.
// *handled: the pages of the chunk are in out_*; (*redo)[q] != 0: except this query's, which the caller answers through the
// exact-everywhere route (nothing of it can be bracketed, or it ties with more than QCAP files).
pvs_status pvs_float_groups_certified(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t q0, uint32_t nb, uint32_t batch_pad, uint32_t k,
                                      int metric, int agg, const float *d_w, const uint8_t *d_mask, float *d_keys, int64_t *out_groups, double *out_values,
                                      uint32_t *out_count, bool *handled, std::vector<uint8_t> *redo) {
    *handled = false;
    redo->assign(nb, 0);
    hipStream_t s = c.stream;
    const uint32_t G = ix->n_groups;
    float *d_lo = nullptr, *d_thr = nullptr;
    uint32_t *d_bmin = nullptr, *d_small = nullptr, *d_qlist = nullptr, *d_bits = nullptr, *d_ufiles = nullptr;
    // d_small: [bad query flags nb | candidate counts nb | union files, union rows] — one copy to the host
    std::vector<uint32_t> h_small(2 * (size_t)nb + 2, 0);
    const size_t bits_bytes = ((size_t)G + 31) / 32 * 4;
    auto body = [&]() -> pvs_status {
        HIP_TRY(pvs_scratch_alloc((void **)&d_lo, (size_t)G * nb * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_bmin, (size_t)nb * BUCKETS * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_thr, (size_t)nb * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_small, h_small.size() * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_qlist, (size_t)nb * QCAP * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_bits, bits_bytes));
        HIP_TRY(pvs_scratch_alloc((void **)&d_ufiles, (size_t)UCAP * 4));
        uint32_t *d_badq = d_small, *d_qcnt = d_small + nb, *d_ucnt = d_small + 2 * (size_t)nb;
        // 1. the scan keys of every (row, query) pair: one corpus pass on the matrix cores
        ScanArgs a;
        a.dtype = (int)ix->dtype;
        a.metric = metric;
        a.kslabs = ix->stride / PVS_KSLAB_BYTES;
        a.qgroups = batch_pad / 32;
        a.rows = ix->d_rows;
        a.aux = metric == PVS_COSINE ? ix->d_scan_cos : ix->d_scan_l2;
        a.stride = ix->stride;
        a.n_rows = ix->n;
        a.qmat = c.d_qmat;
        a.qinfo = c.d_qinfo;
        a.thr = c.d_thr;
        a.gmin = c.d_gmin;
        a.groups_per_query = 0;
        a.mode = 4;
        a.tile_step = 1;
        const uint32_t wg_rows = 32u * pvs_scan_row_tiles(a.qgroups);
        const uint32_t n_wgtiles = (uint32_t)((ix->n + wg_rows - 1) / wg_rows);
        a.grid = std::min<uint32_t>(n_wgtiles, (uint32_t)ix->n_cu * pvs_scan_wg_per_cu(a.dtype, a.qgroups, a.kslabs));
        a.dense_out = d_keys;
        a.dense_ld = nb;
        a.batch = nb;
        HIP_TRY(pvs_launch_fill_f32((float *)d_bmin, (uint64_t)nb * BUCKETS, __builtin_inff(), s));
        HIP_TRY(hipMemsetAsync(d_small, 0, h_small.size() * 4, s));
        HIP_TRY(hipMemsetAsync(d_bits, 0, bits_bytes, s));
        if (!span_bound(ix, c, 1, ix->n, &a.ev_start, &a.ev_stop)) a.ev_start = a.ev_stop = nullptr;
        HIP_TRY(pvs_launch_scan(a, s));
        // 2. brackets per (file, query)
        BoundsK b;
        b.keys = d_keys;
        b.ld = nb;
        b.nb = nb;
        b.qinfo = c.d_qinfo;
        b.norm2 = ix->d_norm2;
        b.grp_off = ix->d_grp_off;
        b.grp_rows = ix->d_grp_rows;
        b.n_groups = G;
        b.weights = d_w;
        b.mask = d_mask;
        b.metric = metric;
        b.agg = agg;
        b.lo = d_lo;
        b.bucket_min = d_bmin;
        b.bad_query = d_badq;
        const uint64_t threads = (uint64_t)G * nb;
        hipLaunchKernelGGL(k_group_bounds, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, b);
        HIP_TRY(hipGetLastError());
        // 3. the thresholds, 4. every query's candidate files and their union
        HIP_TRY(pvs_launch_kth((const float *)d_bmin, BUCKETS, nb, k, d_thr, s));
        hipLaunchKernelGGL(k_candidates, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, d_lo, nb, G, d_thr, d_qcnt, d_qlist);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_union, dim3(QCAP / 256, nb), dim3(256), 0, s, d_qcnt, d_qlist, nb, d_bits, ix->d_grp_off, ix->d_grp_rows, d_mask, d_ucnt, d_ufiles);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_small.data(), d_small, h_small.size() * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        spans_collect(ix, c);
        uint32_t n_redo = 0;
        std::vector<uint8_t> skip(nb, 0);
        for (uint32_t q = 0; q < nb; q++) {
            skip[q] = (h_small[q] || h_small[nb + q] > QCAP) ? 1 : 0;
            n_redo += skip[q];
        }
        const uint32_t m_f = h_small[2 * (size_t)nb], m_rows = h_small[2 * (size_t)nb + 1];
        const bool trace = pvs_dbg(PVS_DBG_FLOAT_CERTIFY_TRACE) != 0;
        if (trace) {
            std::vector<float> ht(nb);
            (void)hipMemcpy(ht.data(), d_thr, (size_t)nb * 4, hipMemcpyDeviceToHost);
            fprintf(stderr, "[float certify] n=%llu files=%u nb=%u k=%u metric=%d agg=%d: union of %u candidate files (%u rows), %u queries not certifiable, candidates of q0 %u, thresholds %g %g ...\n",
                    (unsigned long long)ix->n, G, nb, k, metric, agg, m_f, m_rows, n_redo, h_small[nb], ht[0], ht[nb > 1 ? 1 : 0]);
        }
        // Not worth it, or not possible: every query uncertifiable; more than a couple of them (each one is a corpus pass of its own
        // through the exact route: the whole chunk through k_exact_wide is cheaper); more files than one LDS ranking takes; so many
        // rows that the rescan is no bargain
        if (n_redo == nb || n_redo > 2 || m_f == 0 || m_f > UCAP || (uint64_t)m_rows * 8 > ix->n) return PVS_OK;
        pvs_dbg_add(PVS_DBG_FLOAT_CERTIFY_ROWS, m_rows);
        // 5. the candidates, exactly
        const size_t qbytes = (size_t)ix->dim * (qdtype == PVS_I8 ? 1 : 4);
        bool done = false;
        PVS_TRY(pvs_sparse_groups_of_files(ix, c, (const uint8_t *)d_queries + (size_t)q0 * qbytes, qdtype, nb, k, metric, agg, d_w, d_mask, d_ufiles, m_f, m_rows, skip.data(),
                                           out_groups, out_values, out_count, &done));
        if (trace) fprintf(stderr, "[float certify] exact stage over %u files, %u rows: %s\n", m_f, m_rows, done ? "answered" : "handed back");
        if (done) {
            pvs_dbg_add(PVS_DBG_FLOAT_CERTIFY_QUERIES, nb - n_redo);
            *redo = skip;  // (their slots hold pages over the wrong files: the caller overwrites them)
            *handled = true;
        }
        return PVS_OK;
    };
    pvs_status st = body();
    if (st != PVS_OK) (void)hipStreamSynchronize(s);
    for (void *p : {(void *)d_lo, (void *)d_bmin, (void *)d_thr, (void *)d_small, (void *)d_qlist, (void *)d_bits, (void *)d_ufiles}) pvs_scratch_free_on(p, s);
    return st;
}
