// pvs_search.hip — search orchestration over HIP streams: what a search enqueues (the one-launch exact search, the filter scan's
// passes A / B / C, the dense path) and what answers the queries a route hands back; the contexts.  The host-buffer entry points
// (pvs_search, pages, masks, row lists, bounds, coalescing): pvs_search_host.hip; the stream-ordered and sharded ones, pvs_wait and the
// dense `d` column: pvs_search_device.hip.
#include <chrono>
#include <cstring>
#include <new>
#include <string>

#include "pvs_index.hpp"

// ------------------------------------------------------------------- search
pvs_status validate_search(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k,
                                  pvs_metric metric) {
    if (!ix) return pvs_fail(PVS_ERR_INVALID_ARG, "null index");
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, PVS_POISONED_MSG);
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


bool fast_path_ok(const pvs_index *ix, uint32_t k) {
    if (ix->forced_path == 1) return false;
    if (!pvs_scan_supported((int)ix->dtype, ix->stride / PVS_KSLAB_BYTES)) return false;
    if (k > PVS_MAX_K) return false;
    return ix->n > 0;
}

// one query through the dense path; q is the query's index inside the current chunk
pvs_status dense_one(pvs_index *ix, SearchCtx &c, uint32_t q, uint32_t k, int metric, int64_t *out_ids, float *out_dist, uint32_t *out_count,
                     DenseBounds bounds) {
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
bool direct_ok(const pvs_index *ix, const SearchCtx &c, uint32_t batch, uint32_t k) { return batch >= 1 && batch <= PVS_DIRECT_MAX_NQ && pvs_direct_route(ix, k, batch); }
// h_page: the context's pinned block for the pages [ids batch x k x 8 | distances batch x k x 4 | counts (64 B) | stored rows batch x k x 4]
pvs_status enqueue_direct(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t batch, uint32_t k, int metric, int64_t *oid, float *od,
                          uint32_t *oc, uint8_t *h_page) {
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
        c->draining = false;
        c->finished = false;
        c->busy = false;
    }
    ix->ctx_cv.notify_all();  // (context waiters AND a writer waiting at the gate share this condition variable)
}
