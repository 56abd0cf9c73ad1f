// pvs_search_host.hip — C ABI of libpvs: the host-buffer search entry points — pvs_search (the reference's request: queries and pages in
// host memory, api/search.rs:524-694), request coalescing (16 read connections, db/connection.rs:235), pagination
// (pql/builder.rs:578-582), candidate masks and row lists (filters/image_embeddings.rs:140-199), sort bounds (builder.rs:781-815).
// Split out of pvs_search.hip in round 5.
#include <chrono>
#include <cstring>
#include <new>
#include <string>

#include "pvs_index.hpp"

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
    GateShared gate(ix);  // (pvs_gate.hip: a mutation waits for this call, a search never sees one half done)
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
    GateShared gate(ix);  // (pvs_gate.hip: a mutation waits for this call, a search never sees one half done)
    if (offset == 0) return pvs_search(ix, queries, qdtype, batch, limit, metric, out_ids, out_dist, out_count);
    return search_page_impl<float>(batch, offset, limit, out_ids, out_dist, out_count, __builtin_nanf(""), [&](uint32_t k, int64_t *a, float *b, uint32_t *c) {
        return pvs_search(ix, queries, qdtype, batch, k, metric, a, b, c);
    });
}
PVS_EXPORT pvs_status pvs_search_groups_page(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint64_t offset, uint32_t limit,
                                             pvs_metric metric, pvs_agg agg, const float *row_weights, int64_t *out_groups, double *out_values,
                                             uint32_t *out_count) {
    GateShared gate(ix);  // (pvs_gate.hip: a mutation waits for this call, a search never sees one half done)
    if (offset == 0) return pvs_search_groups(ix, queries, qdtype, batch, limit, metric, agg, row_weights, out_groups, out_values, out_count);
    return search_page_impl<double>(batch, offset, limit, out_groups, out_values, out_count, __builtin_nan(""), [&](uint32_t k, int64_t *a, double *b, uint32_t *c) {
        return pvs_search_groups(ix, queries, qdtype, batch, k, metric, agg, row_weights, a, b, c);
    });
}

PVS_EXPORT pvs_status pvs_search_filtered(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k,
                                          pvs_metric metric, const uint8_t *allowed_rows, pvs_space mask_space, int64_t *out_ids,
                                          float *out_dist, uint32_t *out_count) {
    GateShared gate(ix);  // (pvs_gate.hip: a mutation waits for this call, a search never sees one half done)
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
    GateShared gate(ix);  // (pvs_gate.hip: a mutation waits for this call, a search never sees one half done)
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
            // The kernel raises one flag word per query in pinned memory once its page is written: the caller polls those words instead
            // of sleeping on the stream's completion signal (the wake-up through the runtime costs more than the page's trip over
            // PCIe: 690k x 768 int8 p50 0.117 -> 0.111 ms).  A flag may reach host memory BEFORE some of its page (the kernel's
            // system-scope fence does not keep every line of the page in front of the flag: 246 of 400,000 searches, tools/soak_poll.py), so the
            // caller does not trust the flag for that: it poisons every word the kernel will write — ids, distances, counts, stored
            // rows — with values the kernel never writes and reads the page only when no poisoned word is left among the entries the
            // count announces (aligned 4- and 8-byte words arrive whole).  Anything else — a flag that asks for a fallback, a poisoned
            // word that stays (a caller whose row ids really hold the poison value), 20 ms without an answer — waits for the event.
            // Profiling (event spans) and pvs_debug_set("no_flag_poll", 1) keep the event wait.
            const bool poll = !ix->profiling && !pvs_dbg(PVS_DBG_NO_FLAG_POLL);
            volatile uint32_t *hf = c->h_need_dense;
            volatile int64_t *p_ids = (volatile int64_t *)(io + off_p);
            volatile uint32_t *p_dist = (volatile uint32_t *)(io + off_p + bk * 8), *p_cnt = (volatile uint32_t *)(io + off_p + bk * 12),
                              *p_rows = (volatile uint32_t *)(io + off_p + bk * 12 + 64);
            constexpr int64_t POISON_ID = INT64_MIN + 0x5EA1;
            constexpr uint32_t POISON_DIST = 0x7fc5ea1du, POISON_U32 = 0xffffffffu;  // (a NaN payload no arithmetic produces; rows and counts are < 2^32 - 1)
            if (poll) {
                for (size_t i = 0; i < bk; i++) {
                    p_ids[i] = POISON_ID;
                    p_dist[i] = POISON_DIST;
                    p_rows[i] = POISON_U32;
                }
                for (uint32_t q = 0; q < batch; q++) {
                    p_cnt[q] = POISON_U32;
                    hf[q] = POISON_U32;
                }
            }
            PVS_TRY(enqueue_direct(ix, *c, io + 64, qdtype, batch, k, metric, c->d_out_ids, c->d_out_dist, c->d_out_count, io + off_p));
            bool seen = false, late = false;
            if (poll) {
                const auto t0 = std::chrono::steady_clock::now();
                for (uint64_t spin = 0;; spin++) {
                    bool all = true, ok = true;
                    for (uint32_t q = 0; q < batch; q++) {
                        all = all && hf[q] != POISON_U32;
                        ok = ok && hf[q] == 0;
                    }
                    if (all && !ok) break;  // (a fallback is needed: the event below says when the kernel is through)
                    if (all) {
                        bool landed = true;
                        for (uint32_t q = 0; q < batch && landed; q++) {
                            const uint32_t cnt = p_cnt[q];
                            landed = cnt != POISON_U32 && cnt <= k;
                            for (uint32_t i = 0; i < cnt && landed; i++) {
                                const size_t o = (size_t)q * k + i;
                                landed = p_ids[o] != POISON_ID && p_dist[o] != POISON_DIST && p_rows[o] != POISON_U32;
                            }
                        }
                        if (landed) {
                            seen = true;
                            break;
                        }
                        if (!late) {
                            late = true;
                            pvs_dbg_add(PVS_DBG_POLL_LATE_PAGES, 1);  // (the flag was there before its page: counted, pvs_debug_get("poll_late_pages"))
                        }
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
    GateShared gate(ix);  // (pvs_gate.hip: a mutation waits for this call, a search never sees one half done)
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
