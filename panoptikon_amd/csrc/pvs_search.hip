// pvs_search.hip — C ABI of libpvs, part 2: search orchestration over HIP streams (filter scan passes A/B/C,
// dense fallbacks, candidate masks), the stream-ordered and sharded entry points, the dense `d` column.
#include <chrono>
#include <cstring>
#include <new>
#include <string>

#include "pvs_index.hpp"

// ------------------------------------------------------------------- search
pvs_status validate_search(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k,
                                  pvs_metric metric) {
    if (!ix) return pvs_fail(PVS_ERR_INVALID_ARG, "null index");
    if (batch && !queries) return pvs_fail(PVS_ERR_INVALID_ARG, "null queries");
    if (k < 1) return pvs_fail(PVS_ERR_INVALID_ARG, "k must be a positive integer");  // preprocess.rs:441-444
    if (k > (1u << 20)) return pvs_fail(PVS_ERR_INVALID_ARG, "k too large");
    if (metric != PVS_COSINE && metric != PVS_L2) return pvs_fail(PVS_ERR_INVALID_ARG, "unknown metric");
    if (qdtype == PVS_I8) {
        if (ix->dtype != PVS_I8) return pvs_fail(PVS_ERR_DIM_MISMATCH, "int8 query against a float index (element type mismatch)");
    } else if (qdtype == PVS_F32) {
        if (ix->dtype == PVS_I8 && !ix->scale_set)
            return pvs_fail(PVS_ERR_STATE, "f32 query on an int8 index needs the scale artifact");
    } else {
        return pvs_fail(PVS_ERR_INVALID_ARG, "queries must be f32 or int8");
    }
    return PVS_OK;
}

// the tie order of the second sort key, when it covers the index's rows (pvs_index_set_order_keys)
static inline const uint32_t *order_tinv(const pvs_index *ix) { return ix->order_rows == ix->n && ix->n ? ix->d_tinv : nullptr; }

bool fast_path_ok(const pvs_index *ix, uint32_t k) {
    if (ix->forced_path == 1) return false;
    if (!pvs_scan_supported((int)ix->dtype, ix->stride / PVS_KSLAB_BYTES)) return false;
    if (k > PVS_MAX_K) return false;
    return ix->n > 0;
}

// one query through the dense path; q is the query's index inside the current chunk
static pvs_status dense_one(pvs_index *ix, SearchCtx &c, uint32_t q, uint32_t k, int metric, int64_t *out_ids, float *out_dist,
                            uint32_t *out_count, DenseBounds bounds = DenseBounds()) {
    PVS_TRY(pvs_dense_reserve(c.dense, ix->n));
    const uint8_t *qe = c.d_qexact + (size_t)q * ix->dim * (ix->dtype == PVS_I8 ? 1 : 4);
    HIP_TRY(pvs_launch_dense_exact((int)ix->dtype, metric, ix->d_rows, ix->stride, ix->dim, ix->n, ix->d_norm2, qe, c.d_qinfo + q, 1,
                                   c.d_qpad, c.dense.d_dist, 1, 0, (uint32_t)ix->n_cu, c.stream));
    PVS_TRY(pvs_dense_topk(c.dense, ix->n, k, ix->d_ids, out_ids, out_dist, out_count, c.stream, c.cur_mask, bounds, order_tinv(ix)));
    ix->dense_queries++;
    return PVS_OK;
}

pvs_status prep_chunk(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t qoff, uint32_t nb,
                             uint32_t batch_pad, int metric, hipStream_t on) {
    const size_t qesz = qdtype == PVS_I8 ? 1 : 4;
    const uint8_t *qsrc = (const uint8_t *)d_queries + (size_t)qoff * ix->dim * qesz;
    HIP_TRY(pvs_launch_prep_queries((int)ix->dtype, qdtype, qsrc, nb, batch_pad, ix->dim, ix->stride, ix->scale, metric, c.d_qmat,
                                    c.d_qexact, c.d_qinfo, c.d_need_dense + qoff, on ? on : c.stream));
    return PVS_OK;
}

// The filter-scan passes of one chunk of <= pass_max prepared queries (prep_chunk already ran): pass A (sample -> group minima),
// the k-th select, pass B (every row once), pass C (exact page).  flat_rerun: the chunk is run again for the queries pass C handed
// back with need_dense == 2 — a candidate segment overflowed although the query's candidates fit one list (ties clustered in a
// few tile streams) — with pass B appending to per-query flat lists through atomic counters; the other queries' thresholds
// are voided so that they emit nothing, and pass C finalises the handed-back ones only.
static pvs_status enqueue_fast_chunk(pvs_index *ix, SearchCtx &c, uint32_t qoff, uint32_t nb, uint32_t batch_pad, uint32_t k, int metric,
                                     int64_t *oid, float *od, uint32_t *oc, bool flat_rerun, bool side = false, hipStream_t prelude = nullptr) {
    // prelude: pass A and the k-th select go to that stream (the caller queued the query prep there), pass B waits for them
    if ((ix->multi_stream || side || pvs_dbg(PVS_DBG_FORCE_LIGHT_FINALIZE)) && !pvs_dbg(PVS_DBG_NO_LIGHT_FINALIZE)) PVS_TRY(ctx_fin_buffers(c));  // (first such search of the context)
    ScanArgs a;
    a.dtype = (int)ix->dtype;
    a.metric = metric;
    a.kslabs = ix->stride / PVS_KSLAB_BYTES;
    a.qgroups = batch_pad / 32;
    a.rows = ix->d_rows;
    a.aux = metric == PVS_COSINE ? ix->d_scan_cos : ix->d_scan_l2;
    if (c.cur_mask) {  // filtered search: rows outside the mask stream a NaN scalar and never pass
        HIP_TRY(pvs_launch_mask_aux(a.aux, c.cur_mask, ix->n, ix->cap, c.d_aux_masked, c.stream));
        a.aux = c.d_aux_masked;
    }
    a.stride = ix->stride;
    a.n_rows = ix->n;
    a.qmat = c.d_qmat;
    a.qinfo = c.d_qinfo;
    a.thr = c.d_thr;
    a.seg = c.d_seg;
    a.seg_cnt = c.d_seg_cnt;
    a.gmin = c.d_gmin;
    const uint32_t wg_rows = pvs_scan_wg_rows(a.dtype, a.qgroups, a.kslabs);
    const uint32_t n_wgtiles = (uint32_t)((ix->n + wg_rows - 1) / wg_rows);
    // pass A: strided sample of row tiles -> group minima -> threshold
    // Sample size.  Per wave-tile (32 rows x 32 queries) pass B expects 1024*k/n_sample emitted
    // candidates, and each emit costs a few hundred cycles, while pass A costs ~ n_sample/N of a
    // scan.  Measured on MI355X (10Mx768 int8 x128 and 1Mx768 f16 x32): 1/16 beats 1/8, 1/32, 1/64;
    // a handful of queries emit so little that 1/64 is enough.  The candidate list of a query
    // holds ~k/frac rows: keep that 2.5x below its capacity.
    double frac = nb <= 4 ? 1.0 / 32.0 : 1.0 / 16.0;  // (single query, 10M rows: 1/64 leaves pass C 6,400 candidates = 100 us (f16) for 10 us of pass A)
    if (pvs_dbg(PVS_DBG_SAMPLE_DIV) > 0) frac = 1.0 / (double)pvs_dbg(PVS_DBG_SAMPLE_DIV);  // tuning experiments
    // The threshold need not be the sample's k-th value: its j-th value (j < k) from a sample j/k the size expects the same number
    // of candidates (j / frac') at j/k of pass A's cost, with a relative spread of 1/sqrt(j).  What is lost is the guarantee that k
    // rows lie below T — pass C certifies that from the candidates' upper bounds (FinalizeArgs.thr) and hands the query to the
    // dense path otherwise (never seen: it needs T below the corpus' k-th value while j sample rows lie below it).
    const uint32_t j_div = pvs_dbg(PVS_DBG_SAMPLE_J_DIV) > 0 ? (uint32_t)pvs_dbg(PVS_DBG_SAMPLE_J_DIV) : 4u;  // tuning experiments
    uint32_t k_sel = k;
    if (ix->n >= (1ull << 18) && !flat_rerun) k_sel = std::min(k, std::max<uint32_t>(8, k / j_div));
    frac *= (double)k_sel / (double)k;
    frac = std::min(0.5, std::max(frac, 2.5 * (double)k_sel / (double)PVS_CAND_CAP));
    const uint64_t target_rows = std::min<uint64_t>(ix->n, std::max<uint64_t>((uint64_t)((double)ix->n * frac), 32768));
    const uint32_t want_tiles = (uint32_t)std::max<uint64_t>(1, (target_rows + wg_rows - 1) / wg_rows);
    a.tile_step = std::max<uint32_t>(1, n_wgtiles / want_tiles);
    const uint32_t n_samp = (n_wgtiles + a.tile_step - 1) / a.tile_step;
    if (k_sel < k) {
        // the sample that is actually taken (tile granularity, the 32,768-row floor) may be a larger share of the corpus than planned:
        // keep the expected number of candidates, k_sel / share, at 8 k (or what the candidate list holds with its 2.5x reserve) —
        // pass C's certificate needs k of them below T with a margin of many standard deviations (1 / sqrt(k_sel) each)
        const double share = std::min(1.0, (double)n_samp * wg_rows / (double)ix->n);
        const double expect = std::min(8.0 * k, (double)PVS_CAND_CAP / 2.5);
        k_sel = std::min<uint32_t>(k, std::max<uint32_t>(k_sel, (uint32_t)std::ceil(expect * share)));
    }
    const uint32_t per_cu_a = pvs_scan_wg_per_cu(a.dtype, a.qgroups, a.kslabs);
    const uint32_t spp = pvs_scan_segs_per_stream(a.dtype, a.qgroups, a.kslabs);  // lanes per query and workgroup stream (= its candidate segments)
    a.grid = std::min<uint32_t>({n_samp, (uint32_t)ix->n_cu * per_cu_a, GMAX / (spp * pvs_scan_gmin_max(a.dtype, a.qgroups, a.kslabs))});
    a.mode = 0;
    // row groups per query: >= 16k keeps the threshold within ~3 % of the finest partition (two of the
    // k best rows rarely share a group) and >= 1024; each lane can supply 1..16
    a.gmin_per_lane = pvs_scan_gmin_max(a.dtype, a.qgroups, a.kslabs);
    while (a.gmin_per_lane > 1 && (uint64_t)a.grid * spp * (a.gmin_per_lane / 2) >= std::max<uint64_t>(16ull * k_sel, 1024)) a.gmin_per_lane /= 2;
    a.groups_per_query = a.grid * spp * a.gmin_per_lane;
    hipStream_t ps = prelude ? prelude : c.stream;
    const bool bound_ev = pvs_dbg(PVS_DBG_MARKER_EVENTS) == 0;  // (1: the round-3 form, two hipEventRecord around every kernel)
    if (bound_ev) {
        (void)span_bound(ix, c, 0, (uint64_t)n_samp * wg_rows, &a.ev_start, &a.ev_stop);
        HIP_TRY(pvs_launch_scan(a, ps));
        a.ev_start = a.ev_stop = nullptr;
    } else {
        span_begin(ix, c, 0, (uint64_t)n_samp * wg_rows, ps);
        HIP_TRY(pvs_launch_scan(a, ps));
        span_end(ix, c, ps);
    }
    HIP_TRY(pvs_launch_kth(c.d_gmin, a.groups_per_query, nb, k_sel, c.d_thr, ps, c.d_qinfo, metric));
    if (prelude) {
        HIP_TRY(hipEventRecord(c.preluded, prelude));
        HIP_TRY(hipStreamWaitEvent(c.stream, c.preluded, 0));
    }
    if (flat_rerun) {
        HIP_TRY(pvs_launch_void_thresholds(c.d_thr, c.d_need_dense + qoff, nb, c.stream));  // queries not handed back emit nothing
        HIP_TRY(hipMemsetAsync(c.d_flat_cnt, 0, 4 * (size_t)PVS_SCAN_MAX_BATCH, c.stream));
        a.flat = c.d_cand;
        a.flat_cnt = c.d_flat_cnt;
        a.flat_cap = PVS_CAND_CAP;
    }
    // pass B: every row once (candidate counters were zeroed by the prep kernel)
    a.mode = 1;
    a.tile_step = 1;
    const uint32_t per_cu = pvs_scan_wg_per_cu(a.dtype, a.qgroups, a.kslabs);
    a.grid = std::min<uint32_t>({n_wgtiles, (uint32_t)ix->n_cu * per_cu,
                                 (uint32_t)((uint64_t)PVS_SEG_PAIRS * PVS_SEG_CAP / ((uint64_t)batch_pad * spp * pvs_scan_seg_cap(a.dtype, a.qgroups, a.kslabs)))});
    a.n_segments = a.grid * spp;
    // side: pass C waits for pass B on another stream — the event it waits for is bound to pass B's dispatch as well
    const bool side_c = side && c.d_fin_ub && pvs_dbg(PVS_DBG_NO_LIGHT_FINALIZE) == 0;
    hipEvent_t scanned = nullptr;
    if (bound_ev) {
        if (!span_bound(ix, c, 1, ix->n, &a.ev_start, &a.ev_stop) && side_c) a.ev_stop = c.scanned;
        scanned = a.ev_stop;
        HIP_TRY(pvs_launch_scan(a, c.stream));
        a.ev_start = a.ev_stop = nullptr;
    } else {
        span_begin(ix, c, 1, ix->n);
        HIP_TRY(pvs_launch_scan(a, c.stream));
        span_end(ix, c);
    }
    // pass C
    FinalizeArgs f;
    f.dtype = (int)ix->dtype;
    f.metric = metric;
    f.rows = ix->d_rows;
    f.norm2 = ix->d_norm2;
    f.ids = ix->d_ids;
    f.stride = ix->stride;
    f.dim = ix->dim;
    f.n_rows = ix->n;
    f.qexact = c.d_qexact;
    f.qinfo = c.d_qinfo;
    f.seg = c.d_seg;
    f.seg_cnt = c.d_seg_cnt;
    f.n_segments = a.n_segments;
    f.seg_queries = batch_pad;
    f.seg_cap = pvs_scan_seg_cap(a.dtype, a.qgroups, a.kslabs);
    f.cand = c.d_cand;
    const bool no_light = pvs_dbg(PVS_DBG_NO_LIGHT_FINALIZE) != 0, force_light = pvs_dbg(PVS_DBG_FORCE_LIGHT_FINALIZE) != 0;  // tuning / tests
    if ((ix->multi_stream || force_light || side) && c.d_fin_ub && !no_light) {
        f.w_ub = c.d_fin_ub;
        f.w_surv = c.d_fin_surv;
        f.w_sort = c.d_fin_sort;
    }
    f.cand_cap = PVS_CAND_CAP;
    f.batch = nb;
    f.k = k;
    f.out_ids = oid;
    f.out_dist = od;
    f.out_count = oc;
    f.need_dense = c.d_need_dense + qoff;
    f.cand_seen = c.d_need_dense + c.flags_cap + qoff;
    if (flat_rerun) f.flat_cnt = c.d_flat_cnt;
    f.h_flags = c.h_need_dense + qoff;  // (hipHostMalloc: the same address on the device)
    f.h_seen = c.h_need_dense + c.flags_cap + qoff;
    if (k_sel < k) f.thr = c.d_thr;  // thresholds below the k-th sample value: pass C certifies them
    f.thr_all = c.d_thr;
    // pages that end in NULL rows are completed from the index's NULL list of the metric (search_fallbacks, flag 3) when that set
    // does not depend on the query (no "weird" row, pvs_sparse.hip): pvs_ensure_null_rows ran in search_enqueue
    f.null_ok = ix->null_built_n.load(std::memory_order_acquire) == ix->n && ix->null_weird[metric == PVS_L2 ? 1 : 0] == 0;
    if (order_tinv(ix)) {
        f.trank = ix->d_trank;
        f.tinv = ix->d_tinv;
    }
    // Pass C of a pipelined caller's search goes to the index's side stream, behind an event recorded after pass B: the LDS-light
    // finaliser (6 KB, ~100 registers) fits beside k_scan_wide's one workgroup per CU (148 KB, 2 x 92 registers per SIMD), so it
    // runs under the scan of the caller's NEXT search instead of in front of it (38 us of a 1.3-ms step at configs[2]).
    hipStream_t fs = c.stream;
    if (side && f.w_ub) {
        if (!scanned) {
            HIP_TRY(hipEventRecord(c.scanned, c.stream));
            scanned = c.scanned;
        }
        HIP_TRY(hipStreamWaitEvent(ix->fin_stream, scanned, 0));
        fs = ix->fin_stream;
        c.side_finalize = true;
    }
    if (bound_ev) {
        (void)span_bound(ix, c, 2, 0, &f.ev_start, &f.ev_stop);
        HIP_TRY(pvs_launch_finalize(f, fs));
    } else {
        span_begin(ix, c, 2, 0, fs);
        HIP_TRY(pvs_launch_finalize(f, fs));
        span_end(ix, c, fs);
    }
    return PVS_OK;
}

// One to eight queries, pages of <= 256 rows, over a corpus below the crossover: ONE launch scores every row exactly and selects
// the pages on the way (pvs_direct.hip) — the filter scan's five dependent launches are most of such a search's latency.
// Measured (round 4, tools/direct_crossover.py, 768-d, p50 of pvs_search, k = 10 / 100): one query wins at every size tried — int8 1M rows 0.197 / 0.229 ms against
// 0.239 / 0.236, 8M 1.000 / 1.000 against 1.017 / 1.017; f16 4M 1.01 / 1.00 against 1.09 / 1.10; f32 4M (11.7 GB) 1.94 / 1.88 against 2.01 / 2.04 — by the
// fixed cost it saves; the crossover keeps the north-star shape (10M x 768 f16, 15 GB, filter scan at 0.82 of HBM) where it was.
// Round 5: 2..8 queries (a PQL `or` of a few vector filters over one space, pql/builder.rs:638-661; coalesced callers) share the
// launch: the stream stays HBM-bound (a lane's row chunk feeds NQ chains), the pages are finalised by NQ workgroups at once.
bool pvs_direct_route(const pvs_index *ix, uint32_t k, uint32_t batch) {
    if (ix->forced_path != 0 || ix->n == 0 || pvs_dbg(PVS_DBG_NO_DIRECT_TOPK)) return false;
    if (batch > 1 && pvs_dbg(PVS_DBG_DIRECT_MAX_NQ) > 0 && (int64_t)batch > pvs_dbg(PVS_DBG_DIRECT_MAX_NQ)) return false;
    if (!pvs_direct_supported((int)ix->dtype, ix->stride, ix->esz, k, batch)) return false;
    const uint64_t lim_mb = pvs_dbg(PVS_DBG_DIRECT_MAX_MB) > 0 ? (uint64_t)pvs_dbg(PVS_DBG_DIRECT_MAX_MB) : PVS_DIRECT_CROSSOVER_MB;
    return ix->n * (uint64_t)ix->stride <= (lim_mb << 20);
}
static bool direct_ok(const pvs_index *ix, const SearchCtx &c, uint32_t batch, uint32_t k) { return batch >= 1 && batch <= PVS_DIRECT_MAX_NQ && pvs_direct_route(ix, k, batch); }
// h_page: the context's pinned block for the pages [ids batch x k x 8 | distances batch x k x 4 | counts (64 B) | stored rows batch x k x 4]
static pvs_status enqueue_direct(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t batch, uint32_t k, int metric, int64_t *oid, float *od,
                                 uint32_t *oc, uint8_t *h_page = nullptr) {
    if (!c.d_direct) {
        const uint64_t bytes = pvs_direct_work_bytes((uint32_t)ix->n_cu);
        HIP_TRY(pvs_malloc_retry(&c.d_direct, bytes));
        HIP_TRY(hipMemsetAsync(c.d_direct, 0, bytes, c.stream));
    }
    PVS_TRY(prep_chunk(ix, c, d_queries, qdtype, 0, batch, 32, metric));
    DirectArgs d;
    d.dtype = (int)ix->dtype;
    d.metric = metric;
    d.rows = ix->d_rows;
    d.norm2 = ix->d_norm2;
    d.ids = ix->d_ids;
    d.stride = ix->stride;
    d.dim = ix->dim;
    d.n_rows = ix->n;
    d.qexact = c.d_qexact;
    d.qinfo = c.d_qinfo;
    if (order_tinv(ix)) {
        d.trank = ix->d_trank;
        d.tinv = ix->d_tinv;
    }
    d.mask = c.cur_mask;  // (pvs_search_filtered: rows outside the mask are skipped)
    d.k = k;
    d.nq = batch;
    d.work = c.d_direct;
    d.out_ids = oid;
    d.out_dist = od;
    d.out_count = oc;
    d.need_dense = c.d_need_dense;
    d.h_flags = c.h_need_dense;
    d.h_seen = c.h_need_dense + c.flags_cap;
    d.null_ok = ix->null_built_n.load(std::memory_order_acquire) == ix->n && ix->null_weird[metric == PVS_L2 ? 1 : 0] == 0;
    d.n_cu = (uint32_t)ix->n_cu;
    if (h_page) {
        const size_t bk = (size_t)batch * k;
        d.h_out_ids = (int64_t *)h_page;
        d.h_out_dist = (float *)(h_page + bk * 8);
        d.h_out_count = (uint32_t *)(h_page + bk * 12);
        d.h_out_rows = (uint32_t *)(h_page + bk * 12 + 64);
    }
    (void)span_bound(ix, c, 1, ix->n, &d.ev_start, &d.ev_stop);
    HIP_TRY(pvs_launch_direct_topk(d, c.stream));
    ix->direct_queries += batch;
    pvs_dbg_add(PVS_DBG_DIRECT_QUERIES, batch);
    return PVS_OK;
}

// Enqueues the whole search on c.stream.  Outputs are device buffers.
pvs_status search_enqueue(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t batch, uint32_t k,
                                 int metric, int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count, bool *used_fast, bool side_finalize) {
    const bool fast = fast_path_ok(ix, k);
    *used_fast = fast;
    if (!fast && ix->forced_path == 2) return pvs_fail(PVS_ERR_UNSUPPORTED, "filter-scan path not available for this index / k");
    if (ix->n == 0) {
        HIP_TRY(hipMemsetAsync(c.d_need_dense, 0, 4 * (size_t)batch, c.stream));
        HIP_TRY(hipMemsetAsync(d_out_count, 0, 4 * (size_t)batch, c.stream));
        HIP_TRY(hipMemsetAsync(d_out_ids, 0xff, 8 * (size_t)batch * k, c.stream));
        HIP_TRY(pvs_launch_fill_f32(d_out_dist, (uint64_t)batch * k, __builtin_nanf(""), c.stream));
        HIP_TRY(hipEventRecord(c.done, c.stream));
        return PVS_OK;
    }
    if (fast) PVS_TRY(pvs_ensure_null_rows(ix));  // (one pass over |a|^2 per index state; a no-op afterwards)
    if (direct_ok(ix, c, batch, k)) {  // (whether or not a scan instance exists for the row pitch)
        if (!fast) PVS_TRY(pvs_ensure_null_rows(ix));
        *used_fast = true;
        PVS_TRY(enqueue_direct(ix, c, d_queries, qdtype, batch, k, metric, d_out_ids, d_out_dist, d_out_count));
        HIP_TRY(hipEventRecord(c.done, c.stream));
        return PVS_OK;
    }
    const uint32_t pass_max = fast ? pvs_scan_max_batch((int)ix->dtype, ix->stride / PVS_KSLAB_BYTES) : PVS_MAX_BATCH;
    for (uint32_t qoff = 0; qoff < batch; qoff += pass_max) {
        const uint32_t nb = std::min(pass_max, batch - qoff);
        const uint32_t batch_pad = nb <= 32 ? 32 : nb <= 64 ? 64 : nb <= 128 ? 128 : 256;
        // Experiment, off by default (pvs_debug_set("prelude_stream", 1)): query prep, pass A and the k-th select of a pipelined
        // caller's search on a stream of their own, so that pass A's workgroups (a whole CU's LDS each, like pass B's) start where the
        // PREVIOUS search's pass B has finished its share (its workgroups end 80-140 us apart) and this search's pass B follows that
        // one directly.  Measured at configs[2] (same box, alternating): pass B 1.20 ms instead of 1.22, but the step 1.354-1.374 ms
        // instead of 1.303-1.312 (93.2-94.6 k q/s against 97.6-98.2 k; 3 or 4 searches in flight: the same) — the two cross-queue
        // event waits per search cost more than the ~60 us of sample + select they hide.
        const bool side = fast && side_finalize && batch <= pass_max && !ix->multi_stream && c.stream == ix->search_stream &&
                          !pvs_dbg(PVS_DBG_NO_SIDE_FINALIZE);
        hipStream_t prelude = side && !c.cur_mask && pvs_dbg(PVS_DBG_PRELUDE_STREAM) ? ix->pre_stream : nullptr;
        PVS_TRY(prep_chunk(ix, c, d_queries, qdtype, qoff, nb, batch_pad, metric, prelude));
        int64_t *oid = d_out_ids + (size_t)qoff * k;
        float *od = d_out_dist + (size_t)qoff * k;
        uint32_t *oc = d_out_count + qoff;
        if (!fast) {
            const bool per_query = pvs_dbg(PVS_DBG_DENSE_PER_QUERY) != 0;
            if (nb >= 2 && pvs_select_supported(k) && !per_query) {  // all of the chunk's queries per corpus pass, pages by radix select
                const uint32_t per = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)nb, (1ull << 31) / (4 * std::max<uint64_t>(ix->n, 1))));
                float *d_m = nullptr;
                HIP_TRY(pvs_scratch_alloc((void **)&d_m, (size_t)ix->n * per * 4));
                pvs_status st = PVS_OK;
                for (uint32_t off = 0; off < nb && st == PVS_OK; off += per) {
                    const uint32_t nbb = std::min(per, nb - off);
                    const uint32_t pad = nbb <= 32 ? 32 : nbb <= 64 ? 64 : 128;
                    st = prep_chunk(ix, c, d_queries, qdtype, qoff + off, nbb, pad, metric);
                    if (st == PVS_OK) st = dense_chunk(ix, c, nbb, pad, metric, d_m);
                    if (st == PVS_OK)
                        st = pvs_select_topk(d_m, ix->n, nbb, nbb, k, c.cur_mask, ix->d_ids, nullptr, oid + (size_t)off * k, od + (size_t)off * k, oc + off, c.stream, order_tinv(ix));
                }
                pvs_scratch_free_on(d_m, c.stream);  // (scoring and select are queued, not finished)
                PVS_TRY(st);
                ix->dense_queries += nb;
                continue;
            }
            for (uint32_t q = 0; q < nb; q++) PVS_TRY(dense_one(ix, c, q, k, metric, oid + (size_t)q * k, od + (size_t)q * k, oc + q));
            continue;
        }
        // (side: one chunk only — a second chunk's query prep would overwrite what the first one's pass C still reads)
        PVS_TRY(enqueue_fast_chunk(ix, c, qoff, nb, batch_pad, k, metric, oid, od, oc, false, side, prelude));
    }
    // (pass C wrote its verdicts and candidate counts straight into c.h_need_dense: FinalizeArgs.h_flags)
    HIP_TRY(hipEventRecord(c.done, c.side_finalize ? ix->fin_stream : c.stream));
    return PVS_OK;
}

// After the stream drained: answer the queries the filter path handed back.
pvs_status search_fallbacks(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t batch, uint32_t k,
                                   int metric, int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count) {
    uint32_t n_dense = 0, n_rerun = 0;
    uint64_t seen = 0;
    for (uint32_t q = 0; q < batch; q++) {
        n_rerun += c.h_need_dense[q] == 2 ? 1 : 0;
        seen += c.h_need_dense[c.flags_cap + q];
    }
    if (n_rerun && fast_path_ok(ix, k)) {
        // A candidate segment overflowed although the query's candidates fit one list (ties clustered in a few tile streams):
        // the chunks that hold such queries go through the scan once more, pass B appending to per-query flat lists (the scan
        // costs the same for one query as for a chunk of them; 10M x 768 int8, 106 of 128 queries affected: 1.6 + ~1.5 ms
        // instead of the dense path's 12.4).
        const uint32_t pass_max = pvs_scan_max_batch((int)ix->dtype, ix->stride / PVS_KSLAB_BYTES);
        for (uint32_t qoff = 0; qoff < batch; qoff += pass_max) {
            const uint32_t nb = std::min(pass_max, batch - qoff);
            bool any = false;
            for (uint32_t q = 0; q < nb; q++) any |= c.h_need_dense[qoff + q] == 2;
            if (!any) continue;
            const uint32_t batch_pad = nb <= 32 ? 32 : nb <= 64 ? 64 : nb <= 128 ? 128 : 256;
            PVS_TRY(prep_chunk(ix, c, d_queries, qdtype, qoff, nb, batch_pad, metric));
            // (prep_chunk rewrites this chunk's flags: put the hand-back marks where pass C and the threshold mask read them)
            HIP_TRY(hipMemcpyAsync(c.d_need_dense + qoff, c.h_need_dense + qoff, 4 * (size_t)nb, hipMemcpyHostToDevice, c.stream));
            PVS_TRY(enqueue_fast_chunk(ix, c, qoff, nb, batch_pad, k, metric, d_out_ids + (size_t)qoff * k, d_out_dist + (size_t)qoff * k,
                                       d_out_count + qoff, true));
        }
        HIP_TRY(hipMemcpyAsync(c.h_need_dense, c.d_need_dense, 4 * (size_t)batch, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
        ix->flat_reruns += n_rerun;
    }
    // pages that end in NULL rows (flag 3): the finite part is written; the tail is the head of the index's NULL list in tie order
    uint32_t n_tail = 0;
    for (uint32_t q = 0; q < batch; q++) n_tail += c.h_need_dense[q] == 3 ? 1 : 0;
    if (n_tail) {
        const uint32_t pass_max = pvs_scan_max_batch((int)ix->dtype, ix->stride / PVS_KSLAB_BYTES);
        for (uint32_t qoff = 0; qoff < batch; qoff += pass_max) {
            const uint32_t nb = std::min(pass_max, batch - qoff);
            bool any = false;
            for (uint32_t q = 0; q < nb; q++) any |= c.h_need_dense[qoff + q] == 3;
            if (!any) continue;
            const uint32_t batch_pad = nb <= 32 ? 32 : nb <= 64 ? 64 : nb <= 128 ? 128 : 256;
            PVS_TRY(prep_chunk(ix, c, d_queries, qdtype, qoff, nb, batch_pad, metric));  // (the tail kernel reads the queries' norms: c.d_qinfo of THIS chunk)
            HIP_TRY(hipMemcpyAsync(c.d_need_dense + qoff, c.h_need_dense + qoff, 4 * (size_t)nb, hipMemcpyHostToDevice, c.stream));
            PVS_TRY(pvs_launch_null_tails(ix, c, metric, c.d_need_dense + qoff, c.h_need_dense + qoff, nb, k, d_out_ids + (size_t)qoff * k, d_out_dist + (size_t)qoff * k,
                                          d_out_count + qoff));
        }
        HIP_TRY(hipStreamSynchronize(c.stream));
        ix->null_tail_queries += n_tail;
    }
    for (uint32_t q = 0; q < batch; q++) n_dense += c.h_need_dense[q] ? 1 : 0;
    ix->last_candidates = seen;
    ix->fast_queries += batch - n_dense;
    if (!n_dense) return PVS_OK;
    if (ix->forced_path == 2) return pvs_fail(PVS_ERR_UNSUPPORTED, "%u queries need the dense path but path=2 forbids it", n_dense);
    const bool no_batched = pvs_dbg(PVS_DBG_DENSE_PER_QUERY) != 0;  // tests: the round-1 form (one query per pass, full sort)
    if (n_dense >= 2 && pvs_select_supported(k) && !no_batched) {
        // Several queries at once: scored together into one [rows][queries] matrix (int8: up to 128 per corpus pass on the
        // matrix cores), pages by an exact radix select over all columns (pvs_select.hip) — not one corpus pass + one full
        // sort per query.
        std::vector<uint32_t> dq;
        for (uint32_t q = 0; q < batch; q++)
            if (c.h_need_dense[q]) dq.push_back(q);
        const size_t qbytes = (size_t)ix->dim * (qdtype == PVS_I8 ? 1 : 4);
        const uint32_t per = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)PVS_MAX_BATCH, (1ull << 31) / (4 * std::max<uint64_t>(ix->n, 1)), (uint64_t)n_dense}));
        uint8_t *d_qd = nullptr;
        float *d_m = nullptr;
        uint32_t *d_qmap = nullptr;
        auto body = [&]() -> pvs_status {
            HIP_TRY(pvs_scratch_alloc((void **)&d_qd, qbytes * n_dense));
            HIP_TRY(pvs_scratch_alloc((void **)&d_m, (size_t)ix->n * per * 4));
            HIP_TRY(pvs_scratch_alloc((void **)&d_qmap, (size_t)n_dense * 4));
            for (uint32_t i = 0; i < n_dense; i++)
                HIP_TRY(hipMemcpyAsync(d_qd + (size_t)i * qbytes, (const uint8_t *)d_queries + (size_t)dq[i] * qbytes, qbytes, hipMemcpyDefault, c.stream));
            HIP_TRY(hipMemcpyAsync(d_qmap, dq.data(), (size_t)n_dense * 4, hipMemcpyHostToDevice, c.stream));
            for (uint32_t off = 0; off < n_dense; off += per) {
                const uint32_t nb = std::min(per, n_dense - off);
                const uint32_t pad = nb <= 32 ? 32 : nb <= 64 ? 64 : 128;
                PVS_TRY(prep_chunk(ix, c, d_qd, qdtype, off, nb, pad, metric));
                PVS_TRY(dense_chunk(ix, c, nb, pad, metric, d_m));
                PVS_TRY(pvs_select_topk(d_m, ix->n, nb, nb, k, c.cur_mask, ix->d_ids, d_qmap + off, d_out_ids, d_out_dist, d_out_count, c.stream, order_tinv(ix)));
            }
            HIP_TRY(hipStreamSynchronize(c.stream));
            return PVS_OK;
        };
        pvs_status st = body();
        if (st != PVS_OK) (void)hipStreamSynchronize(c.stream);
        pvs_scratch_free(d_qd);
        pvs_scratch_free(d_m);
        pvs_scratch_free(d_qmap);
        if (st == PVS_OK) ix->dense_queries += n_dense;
        return st;
    }
    for (uint32_t qoff = 0; qoff < batch; qoff += PVS_MAX_BATCH) {
        const uint32_t nb = std::min(PVS_MAX_BATCH, batch - qoff);
        bool any = false;
        for (uint32_t q = 0; q < nb; q++) any |= c.h_need_dense[qoff + q] != 0;
        if (!any) continue;
        PVS_TRY(prep_chunk(ix, c, d_queries, qdtype, qoff, nb, 32 * ((nb + 31) / 32), metric));
        for (uint32_t q = 0; q < nb; q++) {
            if (!c.h_need_dense[qoff + q]) continue;
            PVS_TRY(dense_one(ix, c, q, k, metric, d_out_ids + (size_t)(qoff + q) * k, d_out_dist + (size_t)(qoff + q) * k,
                              d_out_count + qoff + q));
        }
    }
    HIP_TRY(hipStreamSynchronize(c.stream));
    return PVS_OK;
}

// Contexts = searches in flight on one index (NCTX, the size of the reference's read pool).  Synchronous entry
// points block on a condition variable until one is free; the stream-ordered ones (block == false) fail with
// PVS_ERR_STATE instead — their caller may be the very thread that has to pvs_wait() to free one.
SearchCtx *ctx_acquire(pvs_index *ix, uint32_t *ticket, bool block) {
    std::unique_lock<std::mutex> lk(ix->mu);
    for (;;) {
        for (uint32_t i = 0; i < NCTX; i++)
            if (!ix->ctx[i].busy) {
                ix->ctx[i].busy = true;
                *ticket = i;
                return &ix->ctx[i];
            }
        if (!block) {
            pvs_fail(PVS_ERR_STATE, "too many searches in flight on this index (limit %u): pvs_wait() one first", NCTX);
            return nullptr;
        }
        ix->ctx_cv.wait(lk);
    }
}
void ctx_done(pvs_index *ix, SearchCtx *c) {
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        c->pending = false;
        c->busy = false;
    }
    ix->ctx_cv.notify_one();
}

static pvs_status search_host_any(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                                  int64_t *out_ids, float *out_dist, uint32_t *out_count) {
    if (is_multi(ix)) return multi_search_host(ix, queries, qdtype, batch, k, metric, out_ids, out_dist, out_count);
    return search_host(ix, queries, qdtype, batch, k, metric, nullptr, PVS_HOST, out_ids, out_dist, out_count);
}

// ---- request coalescing.  The reference host answers one query per SQL statement from a pool of up to 16 read connections
// (db/connection.rs:235,320-357): sixteen threads each asking for ONE query's page.  A corpus pass costs the same for 1 query as
// for 32 (HBM-bound), so callers that arrive within `window_us` of each other are answered by ONE pass: the first caller
// becomes the leader, waits out the window (or until `max_batch` queries are waiting), takes every pending request with its own
// metric and query dtype, runs one search with the largest k of the group and hands each request the head of its page — the
// page for a smaller k is a prefix of the page for a larger one (same ordering: distance, then id, NULLs last).  Requests with
// another metric / dtype stay queued for the next leader.
PVS_EXPORT pvs_status pvs_index_set_coalescing(pvs_index *ix, uint32_t window_us, uint32_t max_batch) {
    if (!ix) return pvs_fail(PVS_ERR_INVALID_ARG, "null index");
    std::lock_guard<std::mutex> lk(ix->co.mu);
    ix->co.max_batch = max_batch ? std::min<uint32_t>(max_batch, PVS_MAX_BATCH) : 32;
    ix->co.window_us.store(window_us);
    return PVS_OK;
}
PVS_EXPORT pvs_status pvs_index_coalescing_stats(pvs_index *ix, uint64_t *out_calls, uint64_t *out_passes) {
    if (!ix || !out_calls || !out_passes) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    *out_calls = ix->co.calls.load();
    *out_passes = ix->co.passes.load();
    return PVS_OK;
}

bool coalescing_applies(pvs_index *ix, uint32_t batch) { return ix && ix->co.window_us.load() && batch && batch * 2 <= ix->co.max_batch; }

static pvs_status coalesce_run(pvs_index *ix, int kind, int agg, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                               int64_t *out_a, void *out_b, uint32_t *out_count) {
    if (kind == 0) return search_host_any(ix, queries, qdtype, batch, k, metric, out_a, (float *)out_b, out_count);
    return search_groups_impl(ix, queries, qdtype, batch, k, metric, (pvs_agg)agg, nullptr, nullptr, PVS_HOST, out_a, (double *)out_b, out_count);
}

pvs_status coalesce_call(pvs_index *ix, int kind, int agg, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                         int64_t *out_a, void *out_b, uint32_t *out_count) {
    using Req = pvs_index::CoalesceReq;
    auto &co = ix->co;
    Req me;
    me.queries = queries;
    me.qdtype = qdtype;
    me.batch = batch;
    me.k = k;
    me.metric = metric;
    me.kind = kind;
    me.agg = agg;
    me.out_a = out_a;
    me.out_b = out_b;
    me.out_count = out_count;
    co.calls++;
    const size_t bsz = kind == 0 ? 4 : 8;  // bytes per entry of out_b
    auto same = [&](const Req *r) { return r->metric == me.metric && r->qdtype == me.qdtype && r->kind == me.kind && r->agg == me.agg; };
    std::unique_lock<std::mutex> lk(co.mu);
    co.pending.push_back(&me);
    for (;;) {
        if (me.done) break;
        if (co.leader_active) {
            co.cv_leader.notify_one();  // (the leader may be waiting for the batch to fill)
            co.cv_done.wait(lk);
            continue;
        }
        // ---- this caller leads one pass
        co.leader_active = true;
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(co.window_us.load());
        auto waiting = [&]() {
            uint32_t n = 0;
            for (Req *r : co.pending)
                if (same(r)) n += r->batch;
            return n;
        };
        while (waiting() < co.max_batch && co.cv_leader.wait_until(lk, deadline) != std::cv_status::timeout) {
        }
        std::vector<Req *> group, rest;
        group.push_back(&me);
        uint32_t total = me.batch, kmax = me.k;
        for (Req *r : co.pending) {
            if (r == &me) continue;
            if (same(r) && total + r->batch <= co.max_batch) {
                group.push_back(r);
                total += r->batch;
                kmax = std::max(kmax, r->k);
            } else {
                rest.push_back(r);
            }
        }
        co.pending.swap(rest);
        lk.unlock();
        // one pass for the group
        pvs_status st = PVS_OK;
        std::string err;
        // (nothing may unwind past this point: the group would never be marked done and leader_active never cleared — every
        //  coalesced caller on the index, present and future, would wait forever)
        try {
            if (group.size() == 1) {
                st = coalesce_run(ix, kind, agg, me.queries, me.qdtype, me.batch, me.k, me.metric, me.out_a, me.out_b, me.out_count);
                if (st != PVS_OK) err = pvs_last_error();
            } else {
                const size_t qbytes = (size_t)ix->dim * (me.qdtype == PVS_I8 ? 1 : 4);
                std::vector<uint8_t> q((size_t)total * qbytes), vb((size_t)total * kmax * bsz);
                std::vector<int64_t> va((size_t)total * kmax);
                std::vector<uint32_t> cnt(total);
                size_t off = 0;
                for (Req *r : group) {
                    memcpy(q.data() + off * qbytes, r->queries, (size_t)r->batch * qbytes);
                    off += r->batch;
                }
                st = coalesce_run(ix, kind, agg, q.data(), me.qdtype, total, kmax, me.metric, va.data(), vb.data(), cnt.data());
                if (st != PVS_OK) err = pvs_last_error();
                off = 0;
                const float nan32 = __builtin_nanf("");
                const double nan64 = __builtin_nan("");
                for (Req *r : group) {
                    if (st == PVS_OK)
                        for (uint32_t b = 0; b < r->batch; b++) {
                            const uint32_t have = std::min(cnt[off + b], r->k);
                            int64_t *oa = r->out_a + (size_t)b * r->k;
                            uint8_t *ob = (uint8_t *)r->out_b + (size_t)b * r->k * bsz;
                            memcpy(oa, va.data() + (off + b) * kmax, (size_t)have * 8);
                            memcpy(ob, vb.data() + (off + b) * kmax * bsz, (size_t)have * bsz);
                            for (uint32_t i = have; i < r->k; i++) {
                                oa[i] = -1;
                                if (bsz == 4)
                                    memcpy(ob + (size_t)i * 4, &nan32, 4);
                                else
                                    memcpy(ob + (size_t)i * 8, &nan64, 8);
                            }
                            r->out_count[b] = have;
                        }
                    off += r->batch;
                }
            }
        } catch (const std::bad_alloc &) {
            st = PVS_ERR_OOM;
            err = "out of host memory while coalescing requests";
        } catch (...) {
            st = PVS_ERR_STATE;
            err = "unexpected failure while coalescing requests";
        }
        co.passes++;
        lk.lock();
        for (Req *r : group) {
            r->st = st;
            r->err = err;
            r->done = true;
        }
        co.leader_active = false;
        co.cv_done.notify_all();  // the group is served; one of the callers left in `pending` leads the next pass
    }
    lk.unlock();
    if (me.st != PVS_OK) return pvs_fail(me.st, "%s", me.err.c_str());
    return PVS_OK;
}

PVS_EXPORT pvs_status pvs_search(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k,
                                 pvs_metric metric, int64_t *out_ids, float *out_dist, uint32_t *out_count) {
    if (coalescing_applies(ix, batch)) {
        // (arguments are checked before the request is queued: a bad call fails alone)
        PVS_TRY(validate_search(ix, queries, qdtype, batch, k, metric));
        if (!out_ids || !out_dist || !out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null output");
        return coalesce_call(ix, 0, 0, queries, qdtype, batch, k, metric, out_ids, out_dist, out_count);
    }
    if (ix && is_multi(ix)) return multi_search_host(ix, queries, qdtype, batch, k, metric, out_ids, out_dist, out_count);
    return search_host(ix, queries, qdtype, batch, k, metric, nullptr, PVS_HOST, out_ids, out_dist, out_count);
}

// ---- pagination: one search at k = offset + limit, the tail handed out
template <typename V, typename Run>
static pvs_status search_page_impl(uint32_t batch, uint64_t offset, uint32_t limit, int64_t *out_a, V *out_b, uint32_t *out_count, V nan, Run &&run) {
    if (!out_a || !out_b || !out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null output");
    if (limit < 1) return pvs_fail(PVS_ERR_INVALID_ARG, "k must be a positive integer");
    if (offset + limit > 0x7fffffffull) return pvs_fail(PVS_ERR_INVALID_ARG, "offset + limit too large");
    const uint32_t k = (uint32_t)(offset + limit);
    std::vector<int64_t> a((size_t)batch * k);
    std::vector<V> b((size_t)batch * k);
    std::vector<uint32_t> c(batch);
    PVS_TRY(run(k, a.data(), b.data(), c.data()));
    for (uint32_t q = 0; q < batch; q++) {
        const uint32_t have = c[q] > offset ? (uint32_t)std::min<uint64_t>(c[q] - offset, limit) : 0u;
        for (uint32_t i = 0; i < limit; i++) {
            out_a[(size_t)q * limit + i] = i < have ? a[(size_t)q * k + offset + i] : -1;
            out_b[(size_t)q * limit + i] = i < have ? b[(size_t)q * k + offset + i] : nan;
        }
        out_count[q] = have;
    }
    return PVS_OK;
}
PVS_EXPORT pvs_status pvs_search_page(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint64_t offset, uint32_t limit,
                                      pvs_metric metric, int64_t *out_ids, float *out_dist, uint32_t *out_count) {
    if (offset == 0) return pvs_search(ix, queries, qdtype, batch, limit, metric, out_ids, out_dist, out_count);
    return search_page_impl<float>(batch, offset, limit, out_ids, out_dist, out_count, __builtin_nanf(""), [&](uint32_t k, int64_t *a, float *b, uint32_t *c) {
        return pvs_search(ix, queries, qdtype, batch, k, metric, a, b, c);
    });
}
PVS_EXPORT pvs_status pvs_search_groups_page(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint64_t offset, uint32_t limit,
                                             pvs_metric metric, pvs_agg agg, const float *row_weights, int64_t *out_groups, double *out_values,
                                             uint32_t *out_count) {
    if (offset == 0) return pvs_search_groups(ix, queries, qdtype, batch, limit, metric, agg, row_weights, out_groups, out_values, out_count);
    return search_page_impl<double>(batch, offset, limit, out_groups, out_values, out_count, __builtin_nan(""), [&](uint32_t k, int64_t *a, double *b, uint32_t *c) {
        return pvs_search_groups(ix, queries, qdtype, batch, k, metric, agg, row_weights, a, b, c);
    });
}

PVS_EXPORT pvs_status pvs_search_filtered(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k,
                                          pvs_metric metric, const uint8_t *allowed_rows, pvs_space mask_space, int64_t *out_ids,
                                          float *out_dist, uint32_t *out_count) {
    if (!allowed_rows) return pvs_fail(PVS_ERR_INVALID_ARG, "null candidate mask");
    if (ix && is_multi(ix)) return multi_search_filtered(ix, queries, qdtype, batch, k, metric, allowed_rows, mask_space, out_ids, out_dist, out_count);
    return search_host(ix, queries, qdtype, batch, k, metric, allowed_rows, mask_space, out_ids, out_dist, out_count);
}

// mask: candidate mask over the rows (or nullptr); rows / n_listed: the candidates as a strictly ascending list of row positions
// instead (pvs_search_rows) — answered by gather-and-score when it is short, turned into a mask for the filter scan otherwise
// The page over an explicit candidate set: `rows` = strictly ascending row positions (add order, like a mask's index) — what the
// reference's join against the context CTE leaves (filters/image_embeddings.rs:140-199).  A short list costs what the list costs.
PVS_EXPORT pvs_status pvs_search_rows(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                                      const uint32_t *rows, uint64_t n_listed, pvs_space rows_space, int64_t *out_ids, float *out_dist,
                                      uint32_t *out_count) {
    if (!ix) return pvs_fail(PVS_ERR_INVALID_ARG, "null index");
    if (n_listed && !rows) return pvs_fail(PVS_ERR_INVALID_ARG, "null candidate rows");
    if (is_multi(ix)) return multi_search_rows(ix, queries, qdtype, batch, k, metric, rows, n_listed, rows_space, out_ids, out_dist, out_count);
    static const uint32_t empty = 0;
    return search_host(ix, queries, qdtype, batch, k, metric, nullptr, PVS_HOST, out_ids, out_dist, out_count, rows ? rows : &empty, n_listed,
                       rows ? rows_space : PVS_HOST);
}

pvs_status search_host(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                       const uint8_t *mask, pvs_space mask_space, int64_t *out_ids, float *out_dist, uint32_t *out_count, const uint32_t *rows,
                       uint64_t n_listed, pvs_space rows_space, uint32_t *out_row_idx) {
    PVS_TRY(validate_search(ix, queries, qdtype, batch, k, metric));
    if (out_row_idx) memset(out_row_idx, 0xff, (size_t)batch * k * 4);
    if (!out_ids || !out_dist || !out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null output");
    if (rows && mask) return pvs_fail(PVS_ERR_INVALID_ARG, "a candidate mask or a candidate row list, not both");
    if (!rows && n_listed) return pvs_fail(PVS_ERR_INVALID_ARG, "null candidate rows");
    if (n_listed > ix->n) return pvs_fail(PVS_ERR_INVALID_ARG, "%llu candidate rows for an index of %llu rows", (unsigned long long)n_listed, (unsigned long long)ix->n);
    const bool listed = rows != nullptr;
    if (batch == 0) return PVS_OK;
    HIP_TRY(hipSetDevice(ix->device));
    uint32_t t;
    SearchCtx *c = ctx_acquire(ix, &t);
    // A short candidate list from host memory — the reference's everyday shape: one query, the few hundred rows its other filters
    // left — goes through the context's pinned block end to end: queries and list are read from it by the kernels, the page is
    // written into it; two host memcpys and ONE synchronisation instead of six staged copies and three.
    if (listed && rows_space == PVS_HOST && ix->forced_path == 0 && (ix->n == 0 || pvs_sparse_eligible(ix, n_listed, batch, k))) {
        const size_t qb = (size_t)ix->dim * (qdtype == PVS_I8 ? 1 : 4) * batch;
        const uint32_t m = ix->n ? (uint32_t)n_listed : 0u;
        const size_t off_q = 64, off_l = pvs_round_up(off_q + qb, 64), off_i = pvs_round_up(off_l + (size_t)m * 4, 64), off_d = off_i + (size_t)batch * k * 8,
                     off_c = off_d + (size_t)batch * k * 4, need = off_c + (size_t)batch * 4;
        if (need <= ((size_t)8 << 20)) {
            pvs_status st = ctx_prepare(ix, *c, batch, k, false);
            if (st == PVS_OK) st = ctx_pinned_io(*c, need);
            if (st == PVS_OK && m == 0) {  // no candidate: empty pages, no device work
                for (size_t i = 0; i < (size_t)batch * k; i++) {
                    out_ids[i] = -1;
                    out_dist[i] = __builtin_nanf("");
                }
                for (uint32_t q = 0; q < batch; q++) out_count[q] = 0;
                ix->sparse_queries += batch;
            } else if (st == PVS_OK) {
                uint8_t *io = c->h_io;
                memcpy(io + off_q, queries, qb);
                if (m) memcpy(io + off_l, rows, (size_t)m * 4);
                st = pvs_sparse_search(ix, *c, io + off_q, qdtype, batch, k, metric, (const uint32_t *)(io + off_l), m, (int64_t *)(io + off_i), (float *)(io + off_d),
                                       (uint32_t *)(io + off_c));  // (returns after its one synchronisation)
                if (st == PVS_OK) {
                    memcpy(out_ids, io + off_i, (size_t)batch * k * 8);
                    memcpy(out_dist, io + off_d, (size_t)batch * k * 4);
                    memcpy(out_count, io + off_c, (size_t)batch * 4);
                }
            }
            ix->searches++;
            ctx_done(ix, c);
            return st;
        }
    }
    pvs_status st = ctx_prepare(ix, *c, batch, k, true);
    const size_t qbytes = (size_t)ix->dim * (qdtype == PVS_I8 ? 1 : 4);
    if (st == PVS_OK && !mask && !listed && direct_ok(ix, *c, batch, k)) {
        // One to eight queries over a small or medium corpus (pvs_direct.hip): the queries are read from this context's pinned,
        // device-mapped block, the pages are mirrored into it — no staging copy either way, one synchronisation.  Pages that need the
        // fallbacks (NULL tail, dense path) take the ordinary route below from the device copy of what the kernel wrote.
        const size_t bk = (size_t)batch * k;
        const size_t off_p = pvs_round_up(64 + qbytes * batch, 64), need = off_p + bk * 16 + 128;  // [ids | distances | counts .. | rows]
        auto run = [&]() -> pvs_status {
            PVS_TRY(ctx_pinned_io(*c, need));
            uint8_t *io = c->h_io;
            memcpy(io + 64, queries, qbytes * batch);
            PVS_TRY(pvs_ensure_null_rows(ix));
            // The kernel raises one flag word per query in pinned memory behind a system-scope fence (pages and counts first): the
            // caller polls those words instead of sleeping on the stream's completion signal — the wake-up through the runtime costs
            // more than the page's trip over PCIe.  Profiling (event spans) and pvs_debug_set("no_flag_poll", 1) keep the event wait.
            const bool poll = !ix->profiling && !pvs_dbg(PVS_DBG_NO_FLAG_POLL);
            volatile uint32_t *hf = c->h_need_dense;
            if (poll)
                for (uint32_t q = 0; q < batch; q++) hf[q] = 0xffffffffu;
            PVS_TRY(enqueue_direct(ix, *c, io + 64, qdtype, batch, k, metric, c->d_out_ids, c->d_out_dist, c->d_out_count, io + off_p));
            bool seen = false;
            if (poll) {
                const auto t0 = std::chrono::steady_clock::now();
                for (uint64_t spin = 0;; spin++) {
                    bool all = true;
                    for (uint32_t q = 0; q < batch; q++) all = all && hf[q] != 0xffffffffu;
                    if (all) {
                        seen = true;
                        break;
                    }
                    __builtin_ia32_pause();
                    if ((spin & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;  // (a long search: sleep on the event)
                }
                std::atomic_thread_fence(std::memory_order_acquire);
            }
            if (!seen) {
                HIP_TRY(hipEventRecord(c->done, c->stream));
                HIP_TRY(hipEventSynchronize(c->done));
            }
            spans_collect(ix, *c);
            bool complete = true;
            for (uint32_t q = 0; q < batch; q++) complete = complete && c->h_need_dense[q] == 0;
            if (complete) {
                memcpy(out_ids, io + off_p, bk * 8);
                memcpy(out_dist, io + off_p + bk * 8, bk * 4);
                const uint32_t *cnts = (const uint32_t *)(io + off_p + bk * 12);
                for (uint32_t q = 0; q < batch; q++) {
                    const uint32_t cnt = cnts[q];
                    out_count[q] = cnt;
                    if (out_row_idx) memcpy(out_row_idx + (size_t)q * k, io + off_p + bk * 12 + 64 + (size_t)q * k * 4, (size_t)cnt * 4);
                    for (uint32_t i = cnt; i < k; i++) {  // (k > rows: the page's unused tail)
                        out_ids[(size_t)q * k + i] = -1;
                        out_dist[(size_t)q * k + i] = __builtin_nanf("");
                    }
                }
                ix->fast_queries += batch;
                ix->last_candidates = 0;
                return PVS_OK;
            }
            // (the queries are needed on the device by the fallbacks: the pinned block is device-addressable)
            PVS_TRY(search_fallbacks(ix, *c, io + 64, qdtype, batch, k, metric, c->d_out_ids, c->d_out_dist, c->d_out_count));
            HIP_TRY(hipMemcpyAsync(out_ids, c->d_out_ids, 8 * bk, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(out_dist, c->d_out_dist, 4 * bk, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(out_count, c->d_out_count, 4 * (size_t)batch, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            return PVS_OK;
        };
        st = run();
        if (st != PVS_OK) (void)hipStreamSynchronize(c->stream);
        ix->searches++;
        ctx_done(ix, c);
        return st;
    }
    // The page of a host caller is written by the kernels straight into the context's pinned, device-mapped block (pass C, the merges
    // and the fallbacks address it like HBM) and copied out by the CPU after the search's one synchronisation — three staged D2H copies
    // and a second synchronisation less per call; small query batches are read from that block too instead of an H2D copy.
    int64_t *o_ids = c->d_out_ids;
    float *o_dist = c->d_out_dist;
    uint32_t *o_cnt = c->d_out_count;
    bool pinned_out = false, pinned_q = false;
    const size_t page_bytes = (size_t)batch * k * 12 + (size_t)batch * 4;
    const size_t q_all = qbytes * batch;
    if (st == PVS_OK && page_bytes <= ((size_t)2 << 20)) {
        const size_t off_q = pvs_round_up(8192 + page_bytes, 256);
        st = ctx_pinned_io(*c, off_q + (q_all <= ((size_t)256 << 10) ? q_all : 0) + 256);
        if (st == PVS_OK) {
            uint8_t *io = c->h_io;
            o_ids = (int64_t *)(io + 8192);
            o_dist = (float *)(io + 8192 + (size_t)batch * k * 8);
            o_cnt = (uint32_t *)(io + 8192 + (size_t)batch * k * 12);
            pinned_out = true;
            if (q_all <= ((size_t)256 << 10)) {
                memcpy(io + off_q, queries, q_all);
                pinned_q = true;
            }
        }
    }
    void *d_q = nullptr;
    if (st == PVS_OK && pinned_q) {
        d_q = c->h_io + pvs_round_up(8192 + page_bytes, 256);
    } else if (st == PVS_OK) {
        // (hipMalloc/hipFree per call would cost ~0.1 ms and hipFree synchronises the whole device,
        // stalling the other host threads' searches)
        if (qbytes * batch > c->qstage_cap) {
            hipFree(c->d_qstage);
            c->d_qstage = nullptr;
            c->qstage_cap = 0;
            const size_t cap = pvs_round_up(qbytes * batch, 1 << 16);
            hipError_t e = pvs_malloc_retry(&c->d_qstage, cap);
            if (e != hipSuccess)
                st = pvs_fail(PVS_ERR_OOM, "hipMalloc queries: %s", hipGetErrorString(e));
            else
                c->qstage_cap = cap;
        }
        d_q = c->d_qstage;
    }
    bool fast = false;
    if (st == PVS_OK && !pinned_q) {
        hipError_t e = hipMemcpyAsync(d_q, queries, qbytes * batch, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) st = pvs_fail(PVS_ERR_DEVICE, "H2D queries: %s", hipGetErrorString(e));
    }
    if (st == PVS_OK && (mask || listed) && ix->n) {
        auto setup = [&]() -> pvs_status {
            if (listed && pvs_sparse_eligible(ix, n_listed, batch, k) && ix->forced_path == 0) return PVS_OK;  // (no mask needed: gather-and-score below)
            if (ix->cap > c->mask_cap) {
                hipFree(c->d_mask);
                hipFree(c->d_aux_masked);
                c->d_mask = nullptr;
                c->d_aux_masked = nullptr;
                c->mask_cap = 0;
                HIP_TRY(pvs_malloc_retry((void **)&c->d_mask, ix->cap));
                HIP_TRY(pvs_malloc_retry((void **)&c->d_aux_masked, ix->cap / 32 * PVS_AUX_REC * 4));
                c->mask_cap = ix->cap;
            }
            if (listed) return PVS_OK;  // (the mask is filled from the list below)
            if (mask_space == PVS_HOST) {
                HIP_TRY(hipMemcpyAsync(c->d_mask, mask, ix->n, hipMemcpyHostToDevice, c->stream));
                c->cur_mask = c->d_mask;
            } else {
                c->cur_mask = mask;
            }
            return PVS_OK;
        };
        st = setup();
    }
    // A candidate mask that leaves few rows: gather-and-score over the allowed rows only (pvs_sparse.hip) — cost proportional to the
    // candidate set, as in the reference, where the vector filter is joined to the context CTE (filters/image_embeddings.rs:140-199)
    bool sparse = false;
    uint32_t *d_list = nullptr;
    if (st == PVS_OK && listed) {
        const uint32_t m = (uint32_t)n_listed;
        const uint32_t *dl = rows;
        uint32_t *d_up = nullptr;
        auto run = [&]() -> pvs_status {
            if (rows_space == PVS_HOST && m) {
                HIP_TRY(pvs_scratch_alloc((void **)&d_up, (size_t)m * 4));
                HIP_TRY(hipMemcpyAsync(d_up, rows, (size_t)m * 4, hipMemcpyHostToDevice, c->stream));
                dl = d_up;
            }
            if (ix->n == 0 || (pvs_sparse_eligible(ix, m, batch, k) && ix->forced_path == 0)) {
                sparse = true;
                return pvs_sparse_search(ix, *c, d_q, qdtype, batch, k, metric, dl, ix->n ? m : 0, o_ids, o_dist, o_cnt);
            }
            PVS_TRY(pvs_list_to_mask(dl, m, ix->n, c->d_mask, c->stream));  // (validates the list like the gather path does)
            c->cur_mask = c->d_mask;
            return PVS_OK;
        };
        st = run();
        pvs_scratch_free_on(d_up, c->stream);
    }
    // (one query over a corpus the one-launch search streams in ~0.1 ms: it skips masked rows at no cost, counting the mask first — a
    //  kernel, a 4-byte copy and a round trip, 75 us — could only find a gather path that is no faster there)
    const bool direct_small = direct_ok(ix, *c, batch, k) && ix->n * (uint64_t)ix->stride <= ((uint64_t)1 << 30) && !pvs_dbg(PVS_DBG_SPARSE_MAX);
    if (st == PVS_OK && !sparse && !listed && c->cur_mask && ix->n && ix->forced_path == 0 && !direct_small) {
        uint32_t allowed = 0;
        st = pvs_mask_count(c->cur_mask, ix->n, &allowed, c->stream);
        if (st == PVS_OK && pvs_sparse_eligible(ix, allowed, batch, k)) {
            sparse = true;
            hipError_t e = pvs_scratch_alloc((void **)&d_list, (size_t)std::max<uint32_t>(allowed, 1) * 4);
            if (e != hipSuccess) st = pvs_fail(PVS_ERR_OOM, "candidate list: %s", hipGetErrorString(e));
            if (st == PVS_OK) st = pvs_mask_compact(c->cur_mask, ix->n, d_list, allowed, c->stream);
            if (st == PVS_OK) st = pvs_sparse_search(ix, *c, d_q, qdtype, batch, k, metric, d_list, allowed, o_ids, o_dist, o_cnt);
            pvs_scratch_free_on(d_list, c->stream);
        }
    }
    if (st == PVS_OK && !sparse) st = search_enqueue(ix, *c, d_q, qdtype, batch, k, metric, o_ids, o_dist, o_cnt, &fast);
    if (st == PVS_OK && !sparse) {
        hipError_t e = hipEventSynchronize(c->done);
        if (e != hipSuccess) st = pvs_fail(PVS_ERR_DEVICE, "search failed on device: %s", hipGetErrorString(e));
    }
    if (st == PVS_OK) spans_collect(ix, *c);
    if (st == PVS_OK && !sparse && fast && ix->n)
        st = search_fallbacks(ix, *c, d_q, qdtype, batch, k, metric, o_ids, o_dist, o_cnt);
    if (st == PVS_OK) {
        if (pinned_out) {
            hipError_t e = sparse ? hipStreamSynchronize(c->stream) : hipSuccess;  // (the other routes have waited for c->done / their fallbacks)
            if (e != hipSuccess) st = pvs_fail(PVS_ERR_DEVICE, "search failed on device: %s", hipGetErrorString(e));
            if (st == PVS_OK) {
                memcpy(out_ids, o_ids, 8 * (size_t)batch * k);
                memcpy(out_dist, o_dist, 4 * (size_t)batch * k);
                memcpy(out_count, o_cnt, 4 * (size_t)batch);
            }
        } else {
            hipError_t e = hipMemcpyAsync(out_ids, c->d_out_ids, 8 * (size_t)batch * k, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(out_dist, c->d_out_dist, 4 * (size_t)batch * k, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(out_count, c->d_out_count, 4 * (size_t)batch, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) st = pvs_fail(PVS_ERR_DEVICE, "D2H results: %s", hipGetErrorString(e));
        }
    }
    ix->searches++;
    ctx_done(ix, c);
    return st;
}

// pvs_search restricted by apply_sort_bounds (pql/builder.rs:781-815) on the distance: page 1 of the rows with gt < d < lt.
// An upper bound alone (the usual similarity cut-off) changes nothing about WHICH rows are best: the k smallest distances
// among the rows with d < lt are the k smallest of all rows, cut where d reaches lt — the plain search (filter scan) with the
// page truncated; NULL distances never satisfy a comparison.  With a lower bound `gt`: growing pages of the plain ordering
// first (see below), the dense path — every row scored exactly, rows outside the bounds leave the sort — for deep bounds.
PVS_EXPORT pvs_status pvs_search_bounded(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                                         int32_t have_gt, double gt, int32_t have_lt, double lt, int64_t *out_ids, float *out_dist,
                                         uint32_t *out_count) {
    if (!ix) return pvs_fail(PVS_ERR_INVALID_ARG, "null index");
    if (!out_ids || !out_dist || !out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null output");
    if ((have_gt && gt != gt) || (have_lt && lt != lt)) return pvs_fail(PVS_ERR_INVALID_ARG, "bounds must be numbers");
    if (!have_gt) {
        PVS_TRY(search_host_any(ix, queries, qdtype, batch, k, metric, out_ids, out_dist, out_count));
        const float nan32 = __builtin_nanf("");
        for (uint32_t q = 0; q < batch; q++) {
            uint32_t keep = 0;
            const float *d = out_dist + (size_t)q * k;
            while (keep < out_count[q] && (!have_lt || (d[keep] == d[keep] && (double)d[keep] < lt))) keep++;  // sorted ascending, NULLs last
            for (uint32_t i = keep; i < out_count[q]; i++) {
                out_ids[(size_t)q * k + i] = -1;
                out_dist[(size_t)q * k + i] = nan32;
            }
            out_count[q] = keep;
        }
        return PVS_OK;
    }
    // A multi-device index: every shard answers the bounded search over its rows (growing pages, then ITS dense path for a deep
    // bound), the pages merge under the shared order — rows outside the bounds are no candidates on any shard, so the merge of the
    // shards' first k is the first k of the whole index.
    if (is_multi(ix)) return multi_search_bounded(ix, queries, qdtype, batch, k, metric, have_gt, gt, have_lt, lt, out_ids, out_dist, out_count);
    PVS_TRY(validate_search(ix, queries, qdtype, batch, k, metric));
    if (batch == 0) return PVS_OK;
    // Lower bound.  The rows with d > gt are a SUFFIX of the plain ordering (distance asc, NULL last; ties keep their order), and
    // `gt` is in practice the last distance of an earlier page: few rows lie at or below it.  So: pages of the plain ordering
    // (filter scan) of growing size until k rows inside the bounds are on the page, the page ran into `lt` / the NULL rows (no
    // later row satisfies a comparison), or the page is everything.  Only a query whose bound lies deeper than PVS_MAX_K rows
    // goes on to the dense path below.
    std::vector<uint32_t> pending(batch);
    for (uint32_t q = 0; q < batch; q++) pending[q] = q;
    const size_t qbytes_h = (size_t)ix->dim * (qdtype == PVS_I8 ? 1 : 4);
    const uint64_t kmax = std::min<uint64_t>(PVS_MAX_K, std::max<uint64_t>(ix->n, 1));
    for (uint64_t kp = std::min<uint64_t>(kmax, std::max<uint64_t>(2ull * k, 64)); !pending.empty(); kp = std::min<uint64_t>(kmax, kp * 4)) {
        const uint32_t nb = (uint32_t)pending.size();
        std::vector<uint8_t> pq(qbytes_h * nb);
        for (uint32_t i = 0; i < nb; i++) memcpy(pq.data() + qbytes_h * i, (const uint8_t *)queries + qbytes_h * pending[i], qbytes_h);
        std::vector<int64_t> pi((size_t)nb * kp);
        std::vector<float> pd((size_t)nb * kp);
        std::vector<uint32_t> pc(nb);
        PVS_TRY(search_host_any(ix, pq.data(), qdtype, nb, (uint32_t)kp, metric, pi.data(), pd.data(), pc.data()));
        std::vector<uint32_t> still;
        for (uint32_t i = 0; i < nb; i++) {
            const uint32_t q = pending[i];
            const int64_t *ids = pi.data() + (size_t)i * kp;
            const float *d = pd.data() + (size_t)i * kp;
            uint32_t got = 0;
            bool closed = pc[i] < kp || kp >= ix->n;  // the page is every row there is
            for (uint32_t e = 0; e < pc[i] && got < k; e++) {
                if (d[e] != d[e] || (have_lt && !((double)d[e] < lt))) {  // NULL, or at / beyond lt: nothing later qualifies
                    closed = true;
                    break;
                }
                if (!((double)d[e] > gt)) continue;
                out_ids[(size_t)q * k + got] = ids[e];
                out_dist[(size_t)q * k + got] = d[e];
                got++;
            }
            if (got < k && !closed) {  // grow the page; past kmax: deeper than the filter path pages, the dense path below
                still.push_back(q);
                continue;
            }
            for (uint32_t e = got; e < k; e++) {
                out_ids[(size_t)q * k + e] = -1;
                out_dist[(size_t)q * k + e] = __builtin_nanf("");
            }
            out_count[q] = got;
        }
        pending.swap(still);
        if (kp >= kmax) break;
    }
    if (pending.empty()) return PVS_OK;
    HIP_TRY(hipSetDevice(ix->device));
    uint32_t t;
    SearchCtx *c = ctx_acquire(ix, &t);
    DenseBounds b;
    b.have_gt = have_gt != 0;
    b.have_lt = have_lt != 0;
    b.gt = gt;
    b.lt = lt;
    auto body = [&]() -> pvs_status {
        PVS_TRY(ctx_prepare(ix, *c, batch, k, true));
        const size_t qbytes = (size_t)ix->dim * (qdtype == PVS_I8 ? 1 : 4);
        if (qbytes * batch > c->qstage_cap) {
            hipFree(c->d_qstage);
            c->d_qstage = nullptr;
            c->qstage_cap = 0;
            const size_t cap = pvs_round_up(qbytes * batch, 1 << 16);
            HIP_TRY(pvs_malloc_retry(&c->d_qstage, cap));
            c->qstage_cap = cap;
        }
        HIP_TRY(hipMemcpyAsync(c->d_qstage, queries, qbytes * batch, hipMemcpyHostToDevice, c->stream));
        if (ix->n == 0) {
            HIP_TRY(hipMemsetAsync(c->d_out_count, 0, 4 * (size_t)batch, c->stream));
            HIP_TRY(hipMemsetAsync(c->d_out_ids, 0xff, 8 * (size_t)batch * k, c->stream));
            HIP_TRY(pvs_launch_fill_f32(c->d_out_dist, (uint64_t)batch * k, __builtin_nanf(""), c->stream));
        }
        std::vector<uint8_t> is_pending(batch, 0);
        for (uint32_t q : pending) is_pending[q] = 1;
        for (uint32_t qoff = 0; qoff < batch && ix->n; qoff += PVS_MAX_BATCH) {
            const uint32_t nb = std::min(PVS_MAX_BATCH, batch - qoff);
            bool any = false;
            for (uint32_t q = 0; q < nb; q++) any |= is_pending[qoff + q] != 0;
            if (!any) continue;
            PVS_TRY(prep_chunk(ix, *c, c->d_qstage, qdtype, qoff, nb, 32 * ((nb + 31) / 32), metric));
            for (uint32_t q = 0; q < nb; q++)
                if (is_pending[qoff + q])
                    PVS_TRY(dense_one(ix, *c, q, k, metric, c->d_out_ids + (size_t)(qoff + q) * k, c->d_out_dist + (size_t)(qoff + q) * k,
                                      c->d_out_count + qoff + q, b));
        }
        for (uint32_t q : pending) {  // (the other queries were answered from their pages above)
            HIP_TRY(hipMemcpyAsync(out_ids + (size_t)q * k, c->d_out_ids + (size_t)q * k, 8 * (size_t)k, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(out_dist + (size_t)q * k, c->d_out_dist + (size_t)q * k, 4 * (size_t)k, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(out_count + q, c->d_out_count + q, 4, hipMemcpyDeviceToHost, c->stream));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        return PVS_OK;
    };
    pvs_status st = body();
    if (st != PVS_OK) (void)hipStreamSynchronize(c->stream);
    ix->searches++;
    ctx_done(ix, c);
    return st;
}

PVS_EXPORT pvs_status pvs_search_device(pvs_index *ix, const void *d_queries, pvs_dtype qdtype, uint32_t batch, uint32_t k,
                                        pvs_metric metric, int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count,
                                        uint32_t *out_ticket) {
    if (ix && is_multi(ix)) return multi_search_device(ix, d_queries, qdtype, batch, k, metric, d_out_ids, d_out_dist, d_out_count, out_ticket);
    PVS_TRY(validate_search(ix, d_queries, qdtype, batch, k, metric));
    if (!d_out_ids || !d_out_dist || !d_out_count || !out_ticket) return pvs_fail(PVS_ERR_INVALID_ARG, "null output");
    if (batch == 0) return pvs_fail(PVS_ERR_INVALID_ARG, "empty batch");
    HIP_TRY(hipSetDevice(ix->device));
    uint32_t t;
    SearchCtx *c = ctx_acquire(ix, &t, false);
    if (!c) return PVS_ERR_STATE;
    pvs_status st = ctx_prepare(ix, *c, batch, k, false);
    bool fast = false;
    if (st == PVS_OK) st = search_enqueue(ix, *c, d_queries, qdtype, batch, k, metric, d_out_ids, d_out_dist, d_out_count, &fast, true);
    if (st != PVS_OK) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamSynchronize(ix->pre_stream);
        (void)hipStreamSynchronize(ix->fin_stream);
        ctx_done(ix, c);
        return st;
    }
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        c->pending = true;
    }
    c->p_queries = d_queries;
    c->p_qdtype = qdtype;
    c->p_metric = metric;
    c->p_batch = batch;
    c->p_k = k;
    c->p_out_ids = d_out_ids;
    c->p_out_dist = d_out_dist;
    c->p_out_count = d_out_count;
    c->p_fast = fast;
    ix->searches++;
    *out_ticket = t;
    return PVS_OK;
}

// The shard exchange of a row search: this rank's flags join its page record, ONE all-gather of the records over xGMI, the
// merge on every rank, the gathered flags to the host (every rank sees the same flags and so agrees on a redo).  Stream-ordered.
static pvs_status exchange_pages(pvs_index *ix, SearchCtx &c, pvs_comm *comm, uint32_t batch, uint32_t k, uint32_t world, int64_t *d_out_ids, float *d_out_dist,
                                 uint32_t *d_out_count, hipStream_t cs) {
    PVS_TRY(ctx_finish_local_page(ix, c, batch, k, cs));
    PVS_TRY(pvs_comm_gather_records_(comm, c.d_loc_rec, c.d_all_rec, c.rec_bytes, cs));
    HIP_TRY(pvs_launch_merge_packed(c.d_all_rec, c.rec_bytes, world, batch, k, d_out_ids, d_out_dist, d_out_count, cs, c.h_all_flags));
    return PVS_OK;
}

PVS_EXPORT pvs_status pvs_wait(pvs_index *ix, uint32_t ticket) {
    if (!ix || ticket >= NCTX) return pvs_fail(PVS_ERR_INVALID_ARG, "bad ticket");
    if (is_multi(ix)) return multi_wait(ix, ticket);
    SearchCtx *c = &ix->ctx[ticket];
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        if (!c->busy || !c->pending) return pvs_fail(PVS_ERR_STATE, "ticket %u has no search in flight", ticket);
    }
    HIP_TRY(hipSetDevice(ix->device));
    pvs_status st = PVS_OK;
    hipError_t e = hipEventSynchronize(c->done);
    if (e != hipSuccess) st = pvs_fail(PVS_ERR_DEVICE, "search failed on device: %s", hipGetErrorString(e));
    if (st == PVS_OK) spans_collect(ix, *c);
    if (st == PVS_OK && c->p_comm) {
        // every rank sees the same gathered flags, so they all agree on whether to redo
        bool redo = false;
        for (uint64_t i = 0; i < (uint64_t)c->sh_world * c->p_batch; i++) redo |= (c->h_all_flags[i] & ~PVS_PAGE_KEYED) != 0;
        if (!redo) {
            ix->fast_queries += c->p_fast ? c->p_batch : 0;
        } else {
            if (c->p_fast && ix->n)
                st = search_fallbacks(ix, *c, c->p_queries, c->p_qdtype, c->p_batch, c->p_k, c->p_metric, c->d_loc_ids, c->d_loc_dist,
                                      c->d_loc_cnt);
            // (search_fallbacks drained c->stream; the redo's collective goes where all the others go)
            hipStream_t cs = ix->comm_stream;
            if (st == PVS_OK) {
                hipError_t e2 = hipMemsetAsync(c->d_need_dense, 0, 4 * (size_t)c->p_batch, cs);
                if (e2 != hipSuccess) st = pvs_fail(PVS_ERR_DEVICE, "memset: %s", hipGetErrorString(e2));
            }
            if (st == PVS_OK) st = exchange_pages(ix, *c, c->p_comm, c->p_batch, c->p_k, c->sh_world, c->p_final_ids, c->p_final_dist, c->p_final_count, cs);
            if (st == PVS_OK) {
                hipError_t e2 = hipStreamSynchronize(cs);
                if (e2 != hipSuccess) st = pvs_fail(PVS_ERR_DEVICE, "sharded redo: %s", hipGetErrorString(e2));
            }
        }
        c->p_comm = nullptr;
    } else if (st == PVS_OK && c->p_fast && ix->n) {
        st = search_fallbacks(ix, *c, c->p_queries, c->p_qdtype, c->p_batch, c->p_k, c->p_metric, c->p_out_ids, c->p_out_dist,
                              c->p_out_count);
    }
    ctx_done(ix, c);
    return st;
}

// this context's own page (a rank's / a shard's local result before the exchange): one record, its three views repointed for
// the current (batch, k)
pvs_status ctx_reserve_local_pages(SearchCtx &c, uint32_t batch, uint32_t k) {
    const size_t need = pvs_page_record_bytes(batch, k);
    if (need > c.loc_rec_cap) {
        hipFree(c.d_loc_rec);
        c.d_loc_rec = nullptr;
        c.loc_rec_cap = 0;
        HIP_TRY(pvs_malloc_retry((void **)&c.d_loc_rec, need));
        c.loc_rec_cap = need;
    }
    c.rec_bytes = need;
    c.d_loc_ids = (int64_t *)c.d_loc_rec;
    c.d_loc_dist = (float *)(c.d_loc_rec + pvs_page_record_off_dist(batch, k));
    c.d_loc_cnt = (uint32_t *)(c.d_loc_rec + pvs_page_record_off_cnt(batch, k));
    c.d_loc_keys = (int64_t *)(c.d_loc_rec + pvs_page_record_off_keys(batch, k));
    return PVS_OK;
}
pvs_status ctx_finish_local_page(pvs_index *ix, SearchCtx &c, uint32_t batch, uint32_t k, hipStream_t s) {
    const bool keyed = ix->order_rows == ix->n && ix->n && ix->d_order_keys;
    HIP_TRY(pvs_launch_page_finish(c.d_loc_rec, batch, k, c.d_need_dense, ix->d_ids, ix->n, keyed ? ix->d_order_keys : nullptr, s));
    return PVS_OK;
}

PVS_EXPORT pvs_status pvs_search_sharded_async(pvs_index *ix, pvs_comm *comm, const void *d_queries, pvs_dtype qdtype, uint32_t batch,
                                               uint32_t k, pvs_metric metric, int64_t *d_out_ids, float *d_out_dist,
                                               uint32_t *d_out_count, uint32_t *out_ticket) {
    PVS_TRY(validate_search(ix, d_queries, qdtype, batch, k, metric));
    if (!comm || !d_out_ids || !d_out_dist || !d_out_count || !out_ticket) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (batch == 0) return pvs_fail(PVS_ERR_INVALID_ARG, "empty batch");
    if (is_multi(ix)) return pvs_fail(PVS_ERR_UNSUPPORTED, "a multi-device index shards inside one process: use pvs_search / pvs_search_device");
    if (pvs_comm_device_(comm) != ix->device) return pvs_fail(PVS_ERR_INVALID_ARG, "index and communicator live on different devices");
    HIP_TRY(hipSetDevice(ix->device));
    const uint32_t world = (uint32_t)pvs_comm_world_(comm);
    uint32_t t;
    SearchCtx *c = ctx_acquire(ix, &t, false);
    if (!c) return PVS_ERR_STATE;
    auto body = [&]() -> pvs_status {
        PVS_TRY(ctx_prepare(ix, *c, batch, k, false));
        PVS_TRY(ctx_reserve_local_pages(*c, batch, k));
        if (c->rec_bytes * world > c->all_rec_cap) {
            hipFree(c->d_all_rec);
            c->d_all_rec = nullptr;
            c->all_rec_cap = 0;
            HIP_TRY(pvs_malloc_retry((void **)&c->d_all_rec, c->rec_bytes * world));
            c->all_rec_cap = c->rec_bytes * world;
        }
        if ((size_t)batch * world > c->h_all_flags_cap) {
            if (c->h_all_flags) hipHostFree(c->h_all_flags);
            c->h_all_flags = nullptr;
            c->h_all_flags_cap = 0;
            HIP_TRY(hipHostMalloc((void **)&c->h_all_flags, (size_t)batch * 4 * world, hipHostMallocDefault));
            c->h_all_flags_cap = (size_t)batch * world;
        }
        c->sh_world = world;
        bool fast = false;
        // 1. this shard's page (row ids in the index are global ids)
        PVS_TRY(search_enqueue(ix, *c, d_queries, qdtype, batch, k, metric, c->d_loc_ids, c->d_loc_dist, c->d_loc_cnt, &fast));
        // 2. one grouped all-gather over xGMI, 3. merge on every rank — stream-ordered, no host sync.
        // With one stream per context (pvs_index_set_streams) the local scans of several searches
        // overlap, but their collectives still go out on ONE stream in program order: a communicator
        // is never driven from two streams at once.
        // (Both stream modes: every collective of an index goes out on comm_stream, also the per-item pages of
        // pvs_search_groups_sharded.)
        hipStream_t cs = ix->comm_stream;
        HIP_TRY(hipStreamWaitEvent(cs, c->done, 0));  // c->done was just recorded behind the local search
        span_begin(ix, *c, 3, 0, cs);
        PVS_TRY(exchange_pages(ix, *c, comm, batch, k, world, d_out_ids, d_out_dist, d_out_count, cs));
        span_end(ix, *c, cs);
        HIP_TRY(hipEventRecord(c->done, cs));
        {
            std::lock_guard<std::mutex> lk(ix->mu);
            c->pending = true;
        }
        c->p_comm = comm;
        c->p_queries = d_queries;
        c->p_qdtype = qdtype;
        c->p_metric = metric;
        c->p_batch = batch;
        c->p_k = k;
        c->p_out_ids = c->d_loc_ids;
        c->p_out_dist = c->d_loc_dist;
        c->p_out_count = c->d_loc_cnt;
        c->p_final_ids = d_out_ids;
        c->p_final_dist = d_out_dist;
        c->p_final_count = d_out_count;
        c->p_fast = fast;
        return PVS_OK;
    };
    pvs_status st = body();
    if (st != PVS_OK) {
        (void)hipStreamSynchronize(c->stream);
        ctx_done(ix, c);
        return st;
    }
    ix->searches++;
    *out_ticket = t;
    return PVS_OK;
}

PVS_EXPORT pvs_status pvs_search_sharded(pvs_index *ix, pvs_comm *comm, const void *d_queries, pvs_dtype qdtype, uint32_t batch, uint32_t k,
                                         pvs_metric metric, int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count) {
    uint32_t t = 0;
    PVS_TRY(pvs_search_sharded_async(ix, comm, d_queries, qdtype, batch, k, metric, d_out_ids, d_out_dist, d_out_count, &t));
    return pvs_wait(ix, t);
}

PVS_EXPORT pvs_status pvs_sync(pvs_index *ix) {
    if (!ix) return pvs_fail(PVS_ERR_INVALID_ARG, "null index");
    if (is_multi(ix)) return multi_sync(ix);
    pvs_status st = PVS_OK;
    for (uint32_t i = 0; i < NCTX; i++) {
        bool live;
        {
            std::lock_guard<std::mutex> lk(ix->mu);
            live = ix->ctx[i].busy && ix->ctx[i].pending;
        }
        if (live) {
            pvs_status s = pvs_wait(ix, i);
            if (s != PVS_OK) st = s;
        }
    }
    return st;
}

PVS_EXPORT pvs_status pvs_score_all(pvs_index *ix, const void *query, pvs_dtype qdtype, pvs_metric metric, float *out_dist,
                                    pvs_space out_space) {
    if (ix && is_multi(ix)) return multi_score_all(ix, query, qdtype, metric, out_dist, out_space);
    PVS_TRY(validate_search(ix, query, qdtype, 1, 1, metric));
    if (!out_dist) return pvs_fail(PVS_ERR_INVALID_ARG, "null output");
    if (ix->n == 0) return PVS_OK;
    HIP_TRY(hipSetDevice(ix->device));
    uint32_t t;
    SearchCtx *c = ctx_acquire(ix, &t);
    pvs_status st = ctx_prepare(ix, *c, 1, 1, false);
    auto body = [&]() -> pvs_status {
        // the query is read from the context's pinned, device-mapped block (no staged H2D copy), the int8 scorer's out-of-range flag is a
        // word of that block, and a host-space column is copied back behind the scorer without waiting for the flag first: ONE
        // synchronisation per call (three before: flag round trip, column copy)
        const size_t qbytes = (size_t)ix->dim * (qdtype == PVS_I8 ? 1 : 4);
        PVS_TRY(ctx_pinned_io(*c, 4096 + qbytes));
        uint8_t *io = c->h_io;
        memcpy(io + 64, query, qbytes);
        PVS_TRY(prep_chunk(ix, *c, io + 64, qdtype, 0, 1, 32, metric));
        float *dst = out_dist;
        if (out_space == PVS_HOST) {
            PVS_TRY(pvs_dense_reserve(c->dense, ix->n));
            dst = c->dense.d_dist;
        }
        if (ix->dtype == PVS_I8 && (uint64_t)ix->dim * 127 * 127 < (1u << 24)) {
            // int8 codes: the closed form of the exact integer sums straight from HBM (pvs_score_direct.hip, 6.3-6.6 TB/s against
            // 5.0 for the in-order chains); an L2 sum beyond 2^24 raises the flag and the in-order scorer below answers instead
            volatile uint32_t *hf = (volatile uint32_t *)(io + 40);
            *hf = 0;
            HIP_TRY(pvs_launch_score_i8_direct(metric, ix->d_rows, ix->stride, ix->dim, ix->n, ix->d_norm2, c->d_qexact, c->d_qinfo, 1, dst, 1,
                                               (uint32_t *)(io + 40), (uint32_t)ix->n_cu, c->stream));
            if (out_space == PVS_HOST) HIP_TRY(hipMemcpyAsync(out_dist, dst, ix->n * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (*hf == 0) return PVS_OK;
        }
        HIP_TRY(pvs_launch_dense_exact((int)ix->dtype, metric, ix->d_rows, ix->stride, ix->dim, ix->n, ix->d_norm2, c->d_qexact,
                                       c->d_qinfo, 1, c->d_qpad, dst, 1, 0, (uint32_t)ix->n_cu, c->stream));
        if (out_space == PVS_HOST) HIP_TRY(hipMemcpyAsync(out_dist, dst, ix->n * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return PVS_OK;
    };
    if (st == PVS_OK) st = body();
    ctx_done(ix, c);
    return st;
}

// ---- the same column as a handle read in windows (pvs_sqlite.cpp's scalar drop-ins: one device pass per statement, then
// one lookup per row; the statement must not hold the whole column on the host)
struct pvs_column {
    uint64_t rows = 0;
    int device = -1;
    float *d_dev = nullptr;     // single-device index: the column stays in HBM
    std::vector<float> host;    // multi-device index: multi_score_all gathers on the host
};
PVS_EXPORT pvs_status pvs_score_column_create(pvs_index *ix, const void *query, pvs_dtype qdtype, pvs_metric metric, pvs_column **out) {
    if (!ix || !out) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    pvs_column *c = new (std::nothrow) pvs_column();
    if (!c) return pvs_fail(PVS_ERR_OOM, "out of host memory");
    pvs_status st = PVS_OK;
    if (is_multi(ix)) {
        c->rows = ix->n;
        try {
            c->host.assign(c->rows, 0.f);
        } catch (...) {
            delete c;
            return pvs_fail(PVS_ERR_OOM, "out of host memory for a %llu-row column", (unsigned long long)ix->n);
        }
        if (c->rows) st = multi_score_all(ix, query, qdtype, metric, c->host.data(), PVS_HOST);
    } else {
        c->rows = ix->n;
        c->device = ix->device;
        if (c->rows) {
            hipError_t e = hipSetDevice(ix->device);
            if (e == hipSuccess) e = pvs_malloc_retry((void **)&c->d_dev, c->rows * 4);
            if (e != hipSuccess) {
                delete c;
                return pvs_fail(e == hipErrorOutOfMemory ? PVS_ERR_OOM : PVS_ERR_DEVICE, "hipMalloc of a %llu-row column: %s", (unsigned long long)ix->n, hipGetErrorString(e));
            }
            st = pvs_score_all(ix, query, qdtype, metric, c->d_dev, PVS_DEVICE);
        } else {
            st = validate_search(ix, query, qdtype, 1, 1, metric);
        }
    }
    if (st != PVS_OK) {
        pvs_score_column_destroy(c);
        return st;
    }
    *out = c;
    return PVS_OK;
}
PVS_EXPORT pvs_status pvs_score_column_rows(const pvs_column *c, uint64_t *out_rows) {
    if (!c || !out_rows) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    *out_rows = c->rows;
    return PVS_OK;
}
PVS_EXPORT pvs_status pvs_score_column_read(pvs_column *c, uint64_t row0, uint64_t n, float *out_host) {
    if (!c || (n && !out_host)) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (row0 > c->rows || n > c->rows - row0) return pvs_fail(PVS_ERR_INVALID_ARG, "rows [%llu, +%llu) outside the column (%llu rows)", (unsigned long long)row0, (unsigned long long)n, (unsigned long long)c->rows);
    if (!n) return PVS_OK;
    if (c->d_dev) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipMemcpy(out_host, c->d_dev + row0, n * 4, hipMemcpyDeviceToHost));
    } else {
        memcpy(out_host, c->host.data() + row0, n * 4);
    }
    return PVS_OK;
}
PVS_EXPORT void pvs_score_column_destroy(pvs_column *c) {
    if (!c) return;
    if (c->d_dev) {
        (void)hipSetDevice(c->device);
        (void)hipFree(c->d_dev);
    }
    delete c;
}

