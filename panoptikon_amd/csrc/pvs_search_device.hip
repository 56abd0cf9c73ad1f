// pvs_search_device.hip — C ABI of libpvs: the stream-ordered entry points (pvs_search_device + pvs_wait), the one-process-per-GPU
// sharded search (one packed all-gather + merge), pvs_sync, and the dense `d` column (pvs_score_all, pvs_score_column_*:
// what fills dist_{cte}, filters/exact.rs:106-134).  Split out of pvs_search.hip in round 5.
#include <chrono>
#include <cstring>
#include <new>
#include <string>

#include "pvs_index.hpp"

PVS_EXPORT pvs_status pvs_search_device(pvs_index *ix, const void *d_queries, pvs_dtype qdtype, uint32_t batch, uint32_t k,
                                        pvs_metric metric, int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count,
                                        uint32_t *out_ticket) {
    GateShared gate(ix);  // (pvs_gate.hip: a mutation waits for this call, a search never sees one half done)
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

// Everything pvs_wait does for a pending ticket except releasing its context: the device work, the deferred fallbacks, the redo of
// a sharded exchange.  Called by the ticket's owner (pvs_wait) or by a writer that drains the index (pvs_gate.hip); the caller
// holds the context's `draining` claim.
static pvs_status ticket_complete_body(pvs_index *ix, SearchCtx *c);
pvs_status pvs_ticket_complete_(pvs_index *ix, uint32_t ticket) {
    SearchCtx *c = &ix->ctx[ticket];
    const pvs_status st = ticket_complete_body(ix, c);
    c->p_comm = nullptr;  // (whatever happened: the context's next search may not be a sharded one)
    c->p_local_status = PVS_OK;
    return st;
}
static pvs_status ticket_complete_body(pvs_index *ix, SearchCtx *c) {
    HIP_TRY(hipSetDevice(ix->device));
    pvs_status st = PVS_OK;
    if (c->p_comm) {  // behind a collective: the other ranks may never arrive — bounded (pvs_comm.hip)
        st = pvs_comm_wait_event_(c->p_comm, c->done, "shard exchange (all-gather + merge)");
    } else {
        hipError_t e = hipEventSynchronize(c->done);
        if (e != hipSuccess) st = pvs_fail(PVS_ERR_DEVICE, "search failed on device: %s", hipGetErrorString(e));
    }
    if (st == PVS_OK) spans_collect(ix, *c);
    if (st == PVS_OK && c->p_comm) {
        // a rank that failed before the exchange sent a failure record: every rank fails this search (the failing one with its own error)
        int failed_rank = -1;
        for (uint64_t i = 0; i < (uint64_t)c->sh_world * c->p_batch && failed_rank < 0; i++)
            if (c->h_all_flags[i] & PVS_PAGE_FAILED) failed_rank = (int)(i / c->p_batch);
        if (c->p_local_status != PVS_OK) {
            st = pvs_fail(c->p_local_status, "%s", c->p_local_err.c_str());
        } else if (failed_rank >= 0) {
            st = pvs_fail(PVS_ERR_COMM, "rank %d of %u failed before the shard exchange of this search: every rank fails it together", failed_rank, c->sh_world);
        }
        c->p_local_status = PVS_OK;
        if (st != PVS_OK) {
            c->p_comm = nullptr;
            return st;
        }
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
                hipError_t e2 = hipEventRecord(c->done, cs);
                if (e2 != hipSuccess)
                    st = pvs_fail(PVS_ERR_DEVICE, "sharded redo: %s", hipGetErrorString(e2));
                else
                    st = pvs_comm_wait_event_(c->p_comm, c->done, "shard exchange (redo after dense fallbacks)");
            }
        }
        c->p_comm = nullptr;
    } else if (st == PVS_OK && c->p_fast && ix->n) {
        st = search_fallbacks(ix, *c, c->p_queries, c->p_qdtype, c->p_batch, c->p_k, c->p_metric, c->p_out_ids, c->p_out_dist,
                              c->p_out_count);
    }
    return st;
}

PVS_EXPORT pvs_status pvs_wait(pvs_index *ix, uint32_t ticket) {
    if (!ix || ticket >= NCTX) return pvs_fail(PVS_ERR_INVALID_ARG, "bad ticket");
    if (is_multi(ix)) return multi_wait(ix, ticket);
    SearchCtx *c = &ix->ctx[ticket];
    pvs_status st = PVS_OK;
    bool mine = false;
    {
        std::unique_lock<std::mutex> lk(ix->mu);
        if (!c->busy || !(c->pending || c->draining || c->finished)) return pvs_fail(PVS_ERR_STATE, "ticket %u has no search in flight", ticket);
        while (c->draining) ix->ctx_cv.wait(lk);  // a writer is completing it on our behalf (pvs_gate.hip)
        if (!c->busy || !(c->pending || c->finished)) return pvs_fail(PVS_ERR_STATE, "ticket %u has no search in flight", ticket);
        if (c->finished) {  // ... or has: take its verdict
            st = c->fin_status;
            if (st != PVS_OK) pvs_fail(st, "%s", c->fin_err.c_str());
        } else {
            c->draining = true;
            mine = true;
        }
    }
    if (mine) st = pvs_ticket_complete_(ix, ticket);
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
    GateShared gate(ix);  // (pvs_gate.hip: a mutation waits for this call, a search never sees one half done)
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
    bool exchanged = false;  // the all-gather of this search has been queued (a failure after that point is every rank's failure)
    hipStream_t cs = ix->comm_stream;
    auto park = [&](bool fast) {  // the search is in flight: pvs_wait completes it
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
    };
    auto buffers = [&]() -> pvs_status {
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
        return PVS_OK;
    };
    bool fast = false;
    auto body = [&]() -> pvs_status {
        if (pvs_dbg(PVS_DBG_COMM_FAIL_LOCAL) > 0) {  // tests: a rank that fails before its exchange
            pvs_dbg_add(PVS_DBG_COMM_FAIL_LOCAL, -1);
            return pvs_fail(PVS_ERR_OOM, "injected local failure (pvs_debug comm_fail_local)");
        }
        // 1. this shard's page (row ids in the index are global ids)
        PVS_TRY(search_enqueue(ix, *c, d_queries, qdtype, batch, k, metric, c->d_loc_ids, c->d_loc_dist, c->d_loc_cnt, &fast));
        // 2. one grouped all-gather over xGMI, 3. merge on every rank — stream-ordered, no host sync.
        // With one stream per context (pvs_index_set_streams) the local scans of several searches
        // overlap, but their collectives still go out on ONE stream in program order: a communicator
        // is never driven from two streams at once.
        // (Both stream modes: every collective of an index goes out on comm_stream, also the per-item pages of
        // pvs_search_groups_sharded.)
        HIP_TRY(hipStreamWaitEvent(cs, c->done, 0));  // c->done was just recorded behind the local search
        span_begin(ix, *c, 3, 0, cs);
        exchanged = true;
        PVS_TRY(exchange_pages(ix, *c, comm, batch, k, world, d_out_ids, d_out_dist, d_out_count, cs));
        span_end(ix, *c, cs);
        HIP_TRY(hipEventRecord(c->done, cs));
        return PVS_OK;
    };
    pvs_status st = buffers();
    if (st != PVS_OK) {
        // not even the exchange buffers: this rank cannot send a failure record.  Abort the communicator so that the peers' waits
        // end with an error at their deadline at the latest, instead of every later collective queueing behind this one.
        const std::string why = pvs_last_error();
        (void)hipStreamSynchronize(c->stream);
        ctx_done(ix, c);
        pvs_comm_abort_(comm);
        return pvs_fail(st, "%s (this rank could not take part in the shard exchange: communicator aborted)", why.c_str());
    }
    st = body();
    if (st != PVS_OK && !exchanged) {
        // Failed locally BEFORE the exchange: the other ranks are (or will be) inside the all-gather of this search.  Take part in it
        // with a failure record — every flag word carries PVS_PAGE_FAILED, counts 0 — so that every rank fails this search at its
        // pvs_wait instead of waiting for this rank for ever.  The error is reported there.
        const pvs_status local = st;
        const std::string why = pvs_last_error();
        (void)hipStreamSynchronize(c->stream);
        auto send_failure = [&]() -> pvs_status {
            HIP_TRY(hipMemsetAsync(c->d_loc_rec, 0, c->rec_bytes, cs));
            HIP_TRY(hipMemsetAsync(c->d_loc_rec + pvs_page_record_off_flags(batch, k), 0x40, (size_t)batch * 4, cs));
            PVS_TRY(pvs_comm_gather_records_(comm, c->d_loc_rec, c->d_all_rec, c->rec_bytes, cs));
            HIP_TRY(pvs_launch_merge_packed(c->d_all_rec, c->rec_bytes, world, batch, k, d_out_ids, d_out_dist, d_out_count, cs, c->h_all_flags));
            HIP_TRY(hipEventRecord(c->done, cs));
            return PVS_OK;
        };
        if (send_failure() == PVS_OK) {
            c->p_local_status = local;
            c->p_local_err = why;
            park(false);
            ix->searches++;
            *out_ticket = t;
            return PVS_OK;  // (stream-ordered: the failure surfaces at pvs_wait, on every rank)
        }
        pvs_comm_abort_(comm);
        ctx_done(ix, c);
        return pvs_fail(local, "%s (and the failure record could not be sent: communicator aborted)", why.c_str());
    }
    if (st != PVS_OK) {
        (void)hipStreamSynchronize(c->stream);
        ctx_done(ix, c);
        return st;
    }
    park(fast);
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
            live = ix->ctx[i].busy && (ix->ctx[i].pending || ix->ctx[i].draining || ix->ctx[i].finished);
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
    GateShared gate(ix);  // (pvs_gate.hip: a mutation waits for this call, a search never sees one half done)
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
    GateShared gate(ix);  // (pvs_gate.hip: a mutation waits for this call, a search never sees one half done)
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
