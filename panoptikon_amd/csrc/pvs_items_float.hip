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
//   4. k_flag_rows: every file with L <= T for SOME query of the chunk is a candidate (every file of the true page is one: its
//      exact value <= the k-th exact value <= T, and L <= exact); its rows go into a row list;
//   5. pvs_sparse_search_groups (pvs_sparse.hip, round 4): the exact in-order chain on the listed rows only, SQLite's KBN sums per
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

// one thread per (file, query), query fastest: the 32 keys of a row are one 128-byte line
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
    const double sb = cosine ? sqrt((double)qi.bb) : 0.0;
    const uint32_t e0 = a.grp_off[f], e1 = a.grp_off[f + 1];
    const bool weighted = a.weights != nullptr;
    const bool want_min = a.agg == PVS_AGG_MIN, want_max = a.agg == PVS_AGG_MAX;
    double s_lo = 0.0, s_hi = 0.0, s_w = 0.0;
    double x_lo = want_min ? __builtin_inf() : -__builtin_inf(), x_hi = x_lo;
    uint32_t cnt = 0;
    bool forced = false;
    for (uint32_t e = e0; e < e1; e++) {
        const uint32_t row = a.grp_rows[e];
        if (a.mask && !a.mask[row]) continue;
        cnt++;
        const float aa = a.norm2[row];
        const float key = a.keys[(size_t)row * a.ld + q];
        // a row whose distance may be NULL or infinite (zero vector, |a|^2 under / overflow, NaN / inf components), or whose key is
        // not a number: no bracket
        // (L2: a zero vector is an ordinary row at distance |q|)
        if (!((cosine ? aa > 1e-30f : aa >= 0.f) && aa < 1e30f) || !(key == key) || fabsf(key) > 1e30f) {
            forced = true;
            continue;
        }
        double lo, hi;
        if (cosine) {
            const double err = (double)qi.eA;
            lo = 1.0 + ((double)key - err) / sb;
            hi = 1.0 + ((double)key + err) / sb;
        } else {
            const double err = (double)qi.eA + (double)qi.eR * (double)aa;
            const double kl = (double)key - err, kh = (double)key + err;
            lo = kl > 0.0 ? sqrt(kl) : 0.0;
            hi = kh > 0.0 ? sqrt(kh) : 0.0;
        }
        // the f32 rounding of the reference's distance itself (half an ulp) and the roundings above
        lo -= 4e-7 * (1.0 + fabs(lo));
        hi += 4e-7 * (1.0 + fabs(hi));
        if (weighted) {
            const float w = a.weights[row];
            if (!(w > 0.f && w < 1e30f)) {
                forced = true;
                continue;
            }
            s_lo += lo * (double)w;
            s_hi += hi * (double)w;
            s_w += (double)w;
        } else if (want_min) {
            x_lo = fmin(x_lo, lo);
            x_hi = fmin(x_hi, hi);
        } else if (want_max) {
            x_lo = fmax(x_lo, lo);
            x_hi = fmax(x_hi, hi);
        } else {
            s_lo += lo;
            s_hi += hi;
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
        double l, u;
        if (weighted) {
            l = s_lo / s_w;
            u = s_hi / s_w;
        } else if (want_min || want_max) {
            l = x_lo;
            u = x_hi;
        } else {
            l = s_lo / (double)cnt;
            u = s_hi / (double)cnt;
        }
        // (sums of <= a few thousand brackets in f64, SQLite's compensated sum within 1 ulp of the true one: 1e-12 covers both)
        l -= 1e-12 * (1.0 + fabs(l));
        u += 1e-12 * (1.0 + fabs(u));
        L = nextafterf((float)l, -__builtin_inff());
        U = nextafterf((float)u, __builtin_inff());
        if (!(L == L) || !(U == U)) {
            L = -__builtin_inff();
            U = __builtin_inff();
        }
    }
    a.lo[f * a.nb + q] = L;
    if (U < __builtin_inff()) {
        if (U < 0.f) U = 0.f;  // (raising an upper bound keeps it one; non-negative floats order like their bit patterns)
        atomicMin(a.bucket_min + (size_t)q * BUCKETS + (uint32_t)(f % BUCKETS), __builtin_bit_cast(uint32_t, U));
    }
}

// one thread per file: a candidate for some query of the chunk -> its (allowed) rows are flagged
__global__ __launch_bounds__(256) void k_flag_rows(const float *lo, uint32_t nb, const float *thr, const uint32_t *grp_off, const uint32_t *grp_rows, uint32_t n_groups,
                                                   const uint8_t *mask, uint8_t *flag) {
    const uint32_t f = blockIdx.x * 256 + threadIdx.x;
    if (f >= n_groups) return;
    const float *l = lo + (size_t)f * nb;
    bool cand = false;
    for (uint32_t q = 0; q < nb; q++) cand |= l[q] <= thr[q];
    if (!cand) return;
    for (uint32_t e = grp_off[f]; e < grp_off[f + 1]; e++) {
        const uint32_t row = grp_rows[e];
        if (!mask || mask[row]) flag[row] = 1;
    }
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
pvs_status pvs_float_groups_certified(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t q0, uint32_t nb, uint32_t batch_pad, uint32_t k,
                                      int metric, int agg, const float *d_w, const uint8_t *d_mask, float *d_keys, int64_t *out_groups, double *out_values,
                                      uint32_t *out_count, bool *handled, std::vector<uint8_t> *redo) {
    *handled = false;
    redo->assign(nb, 0);
    hipStream_t s = c.stream;
    const uint32_t G = ix->n_groups;
    float *d_lo = nullptr, *d_thr = nullptr;
    uint32_t *d_bmin = nullptr, *d_list = nullptr, *d_badq = nullptr;
    std::vector<uint32_t> h_badq(nb, 0);
    uint8_t *d_flag = nullptr;
    auto body = [&]() -> pvs_status {
        HIP_TRY(pvs_scratch_alloc((void **)&d_lo, (size_t)G * nb * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_bmin, (size_t)nb * BUCKETS * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_thr, (size_t)nb * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&d_flag, ix->n + 64));
        HIP_TRY(pvs_scratch_alloc((void **)&d_badq, (size_t)nb * 4));
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
        HIP_TRY(hipMemsetAsync(d_flag, 0, ix->n, s));
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
        // 3. the threshold, 4. the candidates' rows
        HIP_TRY(pvs_launch_kth((const float *)d_bmin, BUCKETS, nb, k, d_thr, s));
        hipLaunchKernelGGL(k_flag_rows, dim3((G + 255) / 256), dim3(256), 0, s, d_lo, nb, d_thr, ix->d_grp_off, ix->d_grp_rows, G, d_mask, d_flag);
        HIP_TRY(hipGetLastError());
        uint32_t m = 0;
        HIP_TRY(hipMemcpyAsync(h_badq.data(), d_badq, (size_t)nb * 4, hipMemcpyDeviceToHost, s));
        PVS_TRY(pvs_mask_count(d_flag, ix->n, &m, s));  // (synchronises)
        uint32_t n_bad = 0;
        for (uint32_t q = 0; q < nb; q++) n_bad += h_badq[q] ? 1u : 0u;
        const bool trace = pvs_dbg(PVS_DBG_FLOAT_CERTIFY_TRACE) != 0;
        if (trace) {
            std::vector<float> ht(nb);
            (void)hipMemcpy(ht.data(), d_thr, (size_t)nb * 4, hipMemcpyDeviceToHost);
            fprintf(stderr, "[float certify] n=%llu files=%u nb=%u k=%u metric=%d agg=%d: candidate rows %u, bad queries %u, thresholds %g %g ...\n", (unsigned long long)ix->n, G, nb, k,
                    metric, agg, m, n_bad, ht[0], ht[nb > 1 ? 1 : 0]);
        }
        if (n_bad == nb) return PVS_OK;  // (nothing to certify)
        spans_collect(ix, c);
        pvs_dbg_add(PVS_DBG_FLOAT_CERTIFY_ROWS, m);
        // too many candidates to be worth it (ties, a page deeper than the bracketed files, an all-NULL query): the exact-everywhere
        // route answers — it costs n x nb chains, the rescan m x nb
        if (m == 0 || (uint64_t)m * 8 > ix->n || m > (4u << 20)) return PVS_OK;
        HIP_TRY(pvs_scratch_alloc((void **)&d_list, (size_t)m * 4));
        PVS_TRY(pvs_mask_compact(d_flag, ix->n, d_list, m, s));
        // 5. the candidates, exactly
        const size_t qbytes = (size_t)ix->dim * (qdtype == PVS_I8 ? 1 : 4);
        bool done = false;
        PVS_TRY(pvs_sparse_search_groups(ix, c, (const uint8_t *)d_queries + (size_t)q0 * qbytes, qdtype, nb, k, metric, agg, d_w, d_list, m, out_groups, out_values,
                                         out_count, &done));
        if (trace) fprintf(stderr, "[float certify] exact stage over %u rows: %s\n", m, done ? "answered" : "handed back");
        if (done) {
            ix->sparse_queries -= nb;  // (counted there as mask-driven sparse searches; these are not)
            pvs_dbg_add(PVS_DBG_FLOAT_CERTIFY_QUERIES, nb - n_bad);
            for (uint32_t q = 0; q < nb; q++) (*redo)[q] = h_badq[q] ? 1 : 0;  // (their slots hold pages over the wrong files: the caller overwrites them)
            *handled = true;
        }
        return PVS_OK;
    };
    pvs_status st = body();
    if (st != PVS_OK) (void)hipStreamSynchronize(s);
    for (void *p : {(void *)d_lo, (void *)d_bmin, (void *)d_thr, (void *)d_flag, (void *)d_list, (void *)d_badq}) pvs_scratch_free_on(p, s);
    return st;
}
