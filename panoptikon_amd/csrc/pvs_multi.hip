// pvs_multi.hip — one host process, several GPUs (pvs_index_desc.n_devices > 1).
//
// The reference host is ONE process with a pool of read connections (db/connection.rs:235,320-357), so the natural
// drop-in is one index object that owns every GPU of the node: the rows shard across the devices, a search fans
// out to all shards from the calling thread (every enqueue is asynchronous, the shards scan concurrently), each
// shard's page travels to devices[0] with peer copies over xGMI (hipMemcpyPeerAsync: <= 0.31 MB per shard at
// 256 x 100, latency-bound like the RCCL all-gather of the one-process-per-GPU form in pvs_comm.hip) and the same
// k_merge kernel merges them there.  Stream-ordered end to end: the root stream waits on one event per shard, no
// host synchronisation until pvs_wait.  Exactness is the single-device argument per shard plus the merge under
// the shared (distance, id) order; shards hold disjoint ids.
//
// Sharding rule: every pvs_index_add call splits its rows into n_devices contiguous pieces.  Row ids stay strictly
// increasing inside every shard (all a shard needs); global "row order" (pvs_index_read_rows / read_ids /
// pvs_score_all) is the order of the add calls, kept as a segment table.
//
// Per-item work (placement BY GROUP: every row of a file on one shard, multi_add): device-resident rows are placed by a pick
// kernel per shard on the source device + a peer copy; candidate masks resident on devices[0] are split there by one gather per
// shard over the expanded segment table (ensure_shard_rows) and reach the shards by peer copies; the shards write their per-item
// pages into a pinned block, look up the second sort key of the entries on their own device, and devices[0] merges the pages in
// one LDS sort per query (k_merge_group_pages; S * k <= 4,096) or by rank (k_rankmerge_group_pages, <= 32,768; beyond: the host).  Row weights arrive in host memory by the
// ABI and are split there.
#include <thread>

#include "pvs_multi.hpp"

namespace {

void mctx_release(MultiCtx &m) {
    hipFree(m.d_all_rec);
    hipFree(m.d_qroot);
    hipFree(m.d_out_ids);
    hipFree(m.d_out_dist);
    hipFree(m.d_out_cnt);
    if (m.done) hipEventDestroy(m.done);
    if (m.stream) hipStreamDestroy(m.stream);
}

MultiCtx *mctx_acquire(pvs_index *ix, uint32_t *ticket, bool block) {
    std::unique_lock<std::mutex> lk(ix->mu);
    for (;;) {
        for (uint32_t i = 0; i < NCTX; i++)
            if (!ix->mctx[i].busy) {
                ix->mctx[i].busy = true;
                *ticket = i;
                return &ix->mctx[i];
            }
        if (!block) {
            pvs_fail(PVS_ERR_STATE, "too many searches in flight on this index (limit %u): pvs_wait() one first", NCTX);
            return nullptr;
        }
        ix->ctx_cv.wait(lk);
    }
}
void mctx_done(pvs_index *ix, MultiCtx *m) {
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        m->pending = false;
        m->draining = false;
        m->finished = false;
        m->busy = false;
    }
    ix->ctx_cv.notify_all();  // (context waiters and a writer at the gate, pvs_gate.hip)
}

// the shard contexts a multi search holds are released together with it
void release_shard_ctxs(pvs_index *ix, MultiCtx &m) {
    for (size_t s = 0; s < m.tickets.size(); s++) ctx_done(ix->shards[s], &ix->shards[s]->ctx[m.tickets[s]]);
    m.tickets.clear();
}

pvs_status mctx_prepare(pvs_index *ix, MultiCtx &m, uint32_t batch, uint32_t k) {
    const uint32_t S = (uint32_t)ix->shards.size();
    HIP_TRY(hipSetDevice(root_device(ix)));
    if (!m.stream) HIP_TRY(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
    if (!m.done) HIP_TRY(hipEventCreateWithFlags(&m.done, hipEventDisableTiming));
    const size_t need = pvs_page_record_bytes(batch, k) * S;
    if (need > m.all_rec_cap) {
        hipFree(m.d_all_rec);
        m.d_all_rec = nullptr;
        m.all_rec_cap = 0;
        HIP_TRY(pvs_malloc_retry((void **)&m.d_all_rec, need));
        m.all_rec_cap = need;
    }
    m.d_q.resize(S, nullptr);
    m.q_cap.resize(S, 0);
    return PVS_OK;
}

// shard s's page record (flags and order keys completed) -> its slot of the root's gather buffer: one peer copy on the shard's stream
pvs_status ship_page(pvs_index *ix, MultiCtx &m, uint32_t s, SearchCtx &c, uint32_t batch, uint32_t k) {
    pvs_index *sh = ix->shards[s];
    const int root = root_device(ix);
    PVS_TRY(ctx_finish_local_page(sh, c, batch, k, c.stream));
    HIP_TRY(hipMemcpyPeerAsync(m.d_all_rec + (size_t)s * c.rec_bytes, root, c.d_loc_rec, sh->device, c.rec_bytes, c.stream));
    return PVS_OK;
}

// Enqueues one search over every shard plus the gather and the merge; outputs on the root device.
pvs_status multi_enqueue(pvs_index *ix, MultiCtx &m, const void *d_queries, pvs_dtype qdtype, uint32_t batch, uint32_t k,
                         pvs_metric metric, int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count) {
    const uint32_t S = (uint32_t)ix->shards.size();
    const int root = root_device(ix);
    const size_t qbytes = (size_t)ix->dim * (qdtype == PVS_I8 ? 1 : 4) * batch;
    m.p_fast.assign(S, 0);
    m.tickets.clear();
    for (uint32_t s = 0; s < S; s++) {
        pvs_index *sh = ix->shards[s];
        HIP_TRY(hipSetDevice(sh->device));
        uint32_t t;
        SearchCtx *c = ctx_acquire(sh, &t, true);  // never blocks: a shard serves multi searches only, one context each
        m.tickets.push_back(t);
        PVS_TRY(ctx_prepare(sh, *c, batch, k, false));
        PVS_TRY(ctx_reserve_local_pages(*c, batch, k));
        const void *q_local = d_queries;
        if (sh->device != root) {  // replicate the queries on the shard's device (<= 0.8 MB at 256 x 768 f32)
            if (qbytes > m.q_cap[s]) {
                hipFree(m.d_q[s]);
                m.d_q[s] = nullptr;
                m.q_cap[s] = 0;
                const size_t cap = pvs_round_up(qbytes, 1 << 16);
                HIP_TRY(pvs_malloc_retry(&m.d_q[s], cap));
                m.q_cap[s] = cap;
            }
            HIP_TRY(hipMemcpyPeerAsync(m.d_q[s], sh->device, d_queries, root, qbytes, c->stream));
            q_local = m.d_q[s];
        }
        bool fast = false;
        PVS_TRY(search_enqueue(sh, *c, q_local, qdtype, batch, k, metric, c->d_loc_ids, c->d_loc_dist, c->d_loc_cnt, &fast));
        m.p_fast[s] = fast;
        c->p_queries = q_local;
        PVS_TRY(ship_page(ix, m, s, *c, batch, k));
        HIP_TRY(hipEventRecord(c->done, c->stream));
        HIP_TRY(hipSetDevice(root));
        HIP_TRY(hipStreamWaitEvent(m.stream, c->done, 0));
    }
    HIP_TRY(hipSetDevice(root));
    HIP_TRY(pvs_launch_merge_packed(m.d_all_rec, pvs_page_record_bytes(batch, k), S, batch, k, d_out_ids, d_out_dist, d_out_count, m.stream));
    HIP_TRY(hipEventRecord(m.done, m.stream));
    m.p_queries = d_queries;
    m.p_qdtype = qdtype;
    m.p_metric = metric;
    m.p_batch = batch;
    m.p_k = k;
    m.p_out_ids = d_out_ids;
    m.p_out_dist = d_out_dist;
    m.p_out_count = d_out_count;
    return PVS_OK;
}

// After the merge drained: shards that handed queries to the dense path answer them, re-ship, and the root merges again.
pvs_status multi_complete(pvs_index *ix, MultiCtx &m) {
    const uint32_t S = (uint32_t)ix->shards.size();
    const int root = root_device(ix);
    HIP_TRY(hipSetDevice(root));
    hipError_t e = hipEventSynchronize(m.done);
    if (e != hipSuccess) return pvs_fail(PVS_ERR_DEVICE, "search failed on device: %s", hipGetErrorString(e));
    bool redo = false;
    std::vector<uint8_t> dense_q(m.p_batch, 0);
    for (uint32_t s = 0; s < S; s++) {
        pvs_index *sh = ix->shards[s];
        SearchCtx &c = sh->ctx[m.tickets[s]];
        spans_collect(sh, c);
        if (!m.p_fast[s] || sh->n == 0) {
            if (sh->n) std::fill(dense_q.begin(), dense_q.end(), 1);
            continue;
        }
        bool any = false;
        for (uint32_t q = 0; q < m.p_batch; q++)
            if (c.h_need_dense[q]) {
                any = true;
                dense_q[q] = 1;
            }
        if (!any) continue;
        HIP_TRY(hipSetDevice(sh->device));
        PVS_TRY(search_fallbacks(sh, c, c.p_queries, m.p_qdtype, m.p_batch, m.p_k, m.p_metric, c.d_loc_ids, c.d_loc_dist, c.d_loc_cnt));
        PVS_TRY(ship_page(ix, m, s, c, m.p_batch, m.p_k));
        HIP_TRY(hipStreamSynchronize(c.stream));
        redo = true;
    }
    uint32_t nd = 0;
    for (uint8_t f : dense_q) nd += f;
    ix->dense_queries += nd;
    ix->fast_queries += m.p_batch - nd;
    if (redo) {
        HIP_TRY(hipSetDevice(root));
        HIP_TRY(pvs_launch_merge_packed(m.d_all_rec, pvs_page_record_bytes(m.p_batch, m.p_k), S, m.p_batch, m.p_k, m.p_out_ids, m.p_out_dist,
                                        m.p_out_count, m.stream));
        HIP_TRY(hipStreamSynchronize(m.stream));
    }
    return PVS_OK;
}

// shard of a group under BY-GROUP placement: a 64-bit finaliser (splitmix64) of the group id, so that runs of consecutive
// file ids spread evenly
inline uint32_t multi_group_shard(int64_t g, uint32_t S) {
    uint64_t x = (uint64_t)g + 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    x ^= x >> 31;
    return (uint32_t)(x % S);
}
}  // namespace

pvs_status multi_create(const pvs_index_desc *desc, pvs_index **out) {
    if (desc->n_devices > PVS_MAX_DEVICES) return pvs_fail(PVS_ERR_INVALID_ARG, "at most %d devices per index", PVS_MAX_DEVICES);
    if (!desc->devices) return pvs_fail(PVS_ERR_INVALID_ARG, "n_devices > 1 needs the device list");
    pvs_index *ix = new (std::nothrow) pvs_index();
    if (!ix) return pvs_fail(PVS_ERR_OOM, "host allocation failed");
    ix->dtype = desc->dtype;
    ix->dim = desc->dim;
    ix->id_base = desc->id_base;
    const uint32_t S = desc->n_devices;
    for (uint32_t s = 0; s < S; s++) {
        pvs_index_desc d1;
        memset(&d1, 0, sizeof d1);
        d1.struct_size = sizeof d1;
        d1.device = desc->devices[s];
        d1.dtype = desc->dtype;
        d1.dim = desc->dim;
        d1.capacity_rows = (desc->capacity_rows + S - 1) / S;
        d1.id_base = desc->id_base;
        if (d1.device < 0) {
            multi_destroy(ix);
            return pvs_fail(PVS_ERR_INVALID_ARG, "devices[%u] must be an explicit ordinal", s);
        }
        pvs_index *sh = nullptr;
        pvs_status st = pvs_index_create(&d1, &sh);
        if (st != PVS_OK) {
            multi_destroy(ix);
            return st;
        }
        ix->shards.push_back(sh);
    }
    ix->esz = ix->shards[0]->esz;
    ix->stride = ix->shards[0]->stride;
    ix->device = ix->shards[0]->device;
    // direct xGMI peer copies between the root and every other shard's device (without peer access the runtime
    // stages hipMemcpyPeerAsync through host memory: still correct)
    const int root = root_device(ix);
    for (uint32_t s = 1; s < S; s++) {
        const int dev = ix->shards[s]->device;
        if (dev == root) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, root, dev) == hipSuccess && can) {
            (void)hipSetDevice(root);
            (void)hipDeviceEnablePeerAccess(dev, 0);
            (void)hipSetDevice(dev);
            (void)hipDeviceEnablePeerAccess(root, 0);
        }
        (void)hipGetLastError();  // "already enabled" is fine
    }
    (void)hipSetDevice(root);
    *out = ix;
    return PVS_OK;
}

void multi_destroy(pvs_index *ix) {
    if (!ix->shards.empty()) {
        (void)multi_sync(ix);
        (void)hipSetDevice(root_device(ix));
        (void)hipDeviceSynchronize();
    }
    for (auto &m : ix->mctx) {
        for (size_t s = 0; s < m.d_q.size(); s++)
            if (m.d_q[s]) {
                (void)hipSetDevice(ix->shards[s]->device);
                hipFree(m.d_q[s]);
            }
        if (!ix->shards.empty()) (void)hipSetDevice(root_device(ix));
        mctx_release(m);
    }
    if (!ix->shards.empty()) {
        (void)hipSetDevice(root_device(ix));
        for (uint32_t *p : ix->d_shard_rows) hipFree(p);
        for (auto &b : ix->page_blocks)
            if (b.p) hipHostFree(b.p);
    }
    for (pvs_index *sh : ix->shards) pvs_index_destroy(sh);
    ix->shards.clear();
    delete ix;
}

pvs_status multi_add(pvs_index *ix, const void *rows, bool from_f32, uint64_t n, const int64_t *row_ids, const int64_t *group_ids,
                     pvs_space space) {
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, "this multi-device index lost its row order in a failed pvs_index_add: destroy and rebuild it");
    std::lock_guard<std::mutex> lk(ix->mu);
    const uint32_t S = (uint32_t)ix->shards.size();
    if (from_f32 && ix->dtype == PVS_I8 && !ix->scale_set)
        return pvs_fail(PVS_ERR_STATE, "int8 index has no scale artifact: set it before adding f32 rows");
    int64_t last = 0;
    PVS_TRY(check_ids(ix, row_ids, n, &last));  // increasing over the WHOLE index, like a single-device one
    const size_t row_bytes = (size_t)ix->dim * (from_f32 ? 4 : ix->esz);
    int src_dev = -1;
    if (space == PVS_DEVICE) {
        hipPointerAttribute_t at;
        HIP_TRY(hipPointerGetAttributes(&at, rows));
        src_dev = at.device;
    }
    if (group_ids || ix->by_group) {
        // placement BY GROUP: every row of a group lives on shard mix(group) % S, so the per-item operators
        // (MAX / AVG / weighted aggregates, candidate masks, RRF) stay shard-local.  The rows of one call are gathered per shard
        // in their order (ids stay increasing inside every shard); the segment table records the runs.
        if (!group_ids) return pvs_fail(PVS_ERR_INVALID_ARG, "this multi-device index places rows by group: every pvs_index_add needs group_ids");
        if (ix->n && !ix->by_group)
            return pvs_fail(PVS_ERR_STATE, "group ids must be given from the first pvs_index_add of a multi-device index (earlier rows were placed without them)");
        ix->by_group = true;
        const uint64_t chunk = std::max<uint64_t>(1, (256ull << 20) / row_bytes);
        // Shards and the segment table advance chunk by chunk, ix->n only at the end: ANY failure after the first chunk was
        // committed (a HIP error of the staging copy, host memory, a shard's add) leaves them ahead of ix->n — the next add would
        // reuse the same global rows.  Such an index is poisoned (every later call fails), whatever the failure was.
        bool committed = false;
        auto body = [&]() -> pvs_status {
        // Device-resident rows never visit the host: the host decides the shard of every row from the group ids it is given
        // anyway (a list of row positions per shard, 4 bytes a row), a kernel on the source device picks each shard's rows into
        // a dense block, and the block goes to the shard by a peer copy (or stays, when the shard lives on the source device).
        std::vector<std::vector<uint8_t>> buf(S);
        std::vector<std::vector<uint32_t>> pick(S);
        std::vector<std::vector<int64_t>> ids(S), grp(S);
        for (uint64_t off = 0; off < n; off += chunk) {
            const uint64_t m = std::min(chunk, n - off);
            const uint8_t *src = (const uint8_t *)rows + off * row_bytes;
            for (uint32_t s = 0; s < S; s++) {
                buf[s].clear();
                pick[s].clear();
                ids[s].clear();
                grp[s].clear();
            }
            std::vector<MultiSegment> runs;
            std::vector<uint64_t> taken(S, 0);
            for (uint64_t i = 0; i < m; i++) {
                const uint32_t s = multi_group_shard(group_ids[off + i], S);
                if (space == PVS_DEVICE)
                    pick[s].push_back((uint32_t)i);
                else
                    buf[s].insert(buf[s].end(), src + i * row_bytes, src + (i + 1) * row_bytes);
                ids[s].push_back(row_ids ? row_ids[off + i] : ix->id_base + (int64_t)(ix->n + off + i));
                grp[s].push_back(group_ids[off + i]);
                const uint64_t local = ix->shards[s]->n + taken[s]++;
                if (!runs.empty() && runs.back().shard == s && runs.back().row0 + runs.back().n == ix->n + off + i)
                    runs.back().n++;
                else
                    runs.push_back({ix->n + off + i, 1, s, local});
            }
            for (uint32_t s = 0; s < S; s++) {
                if (ids[s].empty()) continue;
                pvs_status st;
                if (space == PVS_DEVICE) {
                    const uint64_t ms = ids[s].size();
                    uint32_t *d_pick = nullptr;
                    void *d_block = nullptr, *d_there = nullptr;
                    auto place = [&]() -> pvs_status {
                        HIP_TRY(hipSetDevice(src_dev));
                        HIP_TRY(pvs_scratch_alloc((void **)&d_pick, ms * 4));
                        HIP_TRY(pvs_scratch_alloc(&d_block, ms * row_bytes));
                        HIP_TRY(hipMemcpy(d_pick, pick[s].data(), ms * 4, hipMemcpyHostToDevice));
                        HIP_TRY(pvs_launch_pick_rows(src, (uint32_t)row_bytes, d_pick, ms, d_block, nullptr));
                        HIP_TRY(hipStreamSynchronize(nullptr));
                        const void *there = d_block;
                        if (ix->shards[s]->device != src_dev) {
                            HIP_TRY(hipSetDevice(ix->shards[s]->device));
                            HIP_TRY(pvs_scratch_alloc(&d_there, ms * row_bytes));
                            HIP_TRY(hipMemcpyPeer(d_there, ix->shards[s]->device, d_block, src_dev, ms * row_bytes));
                            there = d_there;
                        }
                        return add_impl(ix->shards[s], there, from_f32, ms, ids[s].data(), grp[s].data(), PVS_DEVICE);
                    };
                    st = place();
                    // (add_impl returns with the rows ingested: the blocks are idle — after a failure, once the devices drained)
                    if (d_there) {
                        (void)hipSetDevice(ix->shards[s]->device);
                        if (st != PVS_OK) (void)hipDeviceSynchronize();
                        pvs_scratch_free(d_there);
                    }
                    (void)hipSetDevice(src_dev);
                    if (st != PVS_OK) (void)hipDeviceSynchronize();
                    pvs_scratch_free(d_pick);
                    pvs_scratch_free(d_block);
                } else {
                    st = add_impl(ix->shards[s], buf[s].data(), from_f32, ids[s].size(), ids[s].data(), grp[s].data(), PVS_HOST);
                }
                committed = true;  // (a failed add may have taken part of its rows too)
                if (st != PVS_OK) return st;
            }
            // consecutive calls extend the last run when they can
            for (const MultiSegment &r : runs) {
                if (!ix->segs.empty() && ix->segs.back().shard == r.shard && ix->segs.back().row0 + ix->segs.back().n == r.row0 &&
                    ix->segs.back().local0 + ix->segs.back().n == r.local0)
                    ix->segs.back().n += r.n;
                else
                    ix->segs.push_back(r);
            }
        }
        return PVS_OK;
        };
        pvs_status st;
        try {
            st = body();
        } catch (const std::bad_alloc &) {
            st = pvs_fail(PVS_ERR_OOM, "out of host memory while placing rows by group");
        }
        if (st != PVS_OK) {
            if (committed) ix->poisoned = true;
            return st;
        }
        ix->n += n;
        ix->last_id = last;
        ix->ids_epoch++;
        return PVS_OK;
    }
    const uint64_t base = n / S, rem = n % S;
    uint64_t off = 0;
    for (uint32_t s = 0; s < S; s++) {
        const uint64_t m = base + (s < rem ? 1 : 0);
        if (m == 0) continue;
        pvs_index *sh = ix->shards[s];
        const uint8_t *piece = (const uint8_t *)rows + off * row_bytes;
        const int64_t id0 = ix->id_base + (int64_t)(ix->n + off);
        const uint64_t local0 = sh->n;
        pvs_status st;
        if (space == PVS_DEVICE && src_dev != sh->device) {
            // rows resident on another GPU: stage the piece on the shard's device in bounded chunks
            HIP_TRY(hipSetDevice(sh->device));
            const uint64_t chunk = std::max<uint64_t>(1, (256ull << 20) / row_bytes);
            void *stage = nullptr;
            HIP_TRY(pvs_malloc_retry(&stage, std::min(chunk, m) * row_bytes));
            st = PVS_OK;
            for (uint64_t o = 0; o < m && st == PVS_OK; o += chunk) {
                const uint64_t mm = std::min(chunk, m - o);
                hipError_t e = hipMemcpyPeer(stage, sh->device, piece + o * row_bytes, src_dev, mm * row_bytes);
                if (e != hipSuccess) {
                    st = pvs_fail(PVS_ERR_DEVICE, "hipMemcpyPeer: %s", hipGetErrorString(e));
                    break;
                }
                st = add_impl(sh, stage, from_f32, mm, row_ids ? row_ids + off + o : nullptr, group_ids ? group_ids + off + o : nullptr,
                              PVS_DEVICE, id0 + (int64_t)o);
            }
            (void)hipSetDevice(sh->device);
            hipFree(stage);
        } else {
            st = add_impl(sh, piece, from_f32, m, row_ids ? row_ids + off : nullptr, group_ids ? group_ids + off : nullptr, space, id0);
        }
        if (st != PVS_OK) {
            // earlier shards already hold their pieces of this call while the parent's row count does not: the global row order is
            // gone.  Every later call fails instead of reading rows at the wrong offsets; the caller rebuilds the index.
            ix->poisoned = true;
            return st;
        }
        ix->segs.push_back({ix->n + off, m, s, local0});
        off += m;
    }
    ix->n += n;
    ix->last_id = last;
    ix->ids_epoch++;
    return PVS_OK;
}

pvs_status multi_set_scale(pvs_index *ix, float scale) {
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, "this multi-device index lost its row order in a failed pvs_index_add: destroy and rebuild it");
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->n && ix->scale_set && ix->scale != scale)
        return pvs_fail(PVS_ERR_STATE, "scale is frozen once rows exist (artifact_rev semantics): rebuild the index");
    for (pvs_index *sh : ix->shards) PVS_TRY(pvs_index_set_scale(sh, scale));
    ix->scale = scale;
    ix->scale_set = true;
    return PVS_OK;
}

// pvs_index_set_order_keys on a multi-device index: the keys (global row order) split into the shards' row orders; every shard
// orders its own pages with them and attaches the keys to its page records, the root's merge compares (distance, key DESC, id)
pvs_status multi_set_order_keys(pvs_index *ix, const int64_t *keys, uint64_t n, pvs_space space) {
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, "this multi-device index lost its row order in a failed pvs_index_add: destroy and rebuild it");
    std::lock_guard<std::mutex> lk(ix->mu);  // (the caller holds the gate exclusively: no search in flight)
    ix->order_rows = 0;
    ix->h_order_keys.clear();
    if (!keys) {
        for (pvs_index *sh : ix->shards) PVS_TRY(pvs_index_set_order_keys(sh, nullptr, 0, PVS_HOST));
        return PVS_OK;
    }
    if (n != ix->n) return pvs_fail(PVS_ERR_INVALID_ARG, "%llu order keys for %llu rows: one key per stored row", (unsigned long long)n, (unsigned long long)ix->n);
    ix->h_order_keys.resize(n);
    if (space == PVS_HOST)
        memcpy(ix->h_order_keys.data(), keys, n * 8);
    else if (n)
        HIP_TRY(hipMemcpy(ix->h_order_keys.data(), keys, n * 8, hipMemcpyDeviceToHost));
    const auto per = split_rows<int64_t>(ix, ix->h_order_keys.data());
    for (size_t s = 0; s < ix->shards.size(); s++) PVS_TRY(pvs_index_set_order_keys(ix->shards[s], per[s].data(), per[s].size(), PVS_HOST));
    ix->order_rows = n;
    return PVS_OK;
}

pvs_status multi_stats(pvs_index *ix, pvs_stats *out, size_t out_bytes) {
    pvs_stats s;
    memset(&s, 0, sizeof s);
    s.struct_size = sizeof s;
    s.dtype = ix->dtype;
    s.dim = ix->dim;
    s.row_stride_bytes = ix->stride;
    s.scale = ix->scale_set ? ix->scale : 0.f;
    for (pvs_index *sh : ix->shards) {
        pvs_stats p;
        PVS_TRY(pvs_index_stats_ex(sh, &p, sizeof p));
        s.rescanned_queries += p.rescanned_queries;
        s.sparse_queries += p.sparse_queries;
        s.null_tail_queries += p.null_tail_queries;
        s.rows += p.rows;
        s.capacity_rows += p.capacity_rows;
        s.hbm_bytes += p.hbm_bytes;
        s.last_candidates += p.last_candidates;
    }
    s.searches = ix->searches.load();
    s.fast_queries = ix->fast_queries.load();
    s.dense_queries = ix->dense_queries.load();
    const size_t want = std::min(out_bytes, sizeof s);
    s.struct_size = (uint32_t)want;
    memcpy(out, &s, want);
    return PVS_OK;
}

pvs_status multi_read_rows(pvs_index *ix, uint64_t row0, uint64_t n, void *out_host) {
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, "this multi-device index lost its row order in a failed pvs_index_add: destroy and rebuild it");
    if (row0 + n > ix->n) return pvs_fail(PVS_ERR_INVALID_ARG, "row range [%llu, %llu) exceeds %llu rows", (unsigned long long)row0,
                                          (unsigned long long)(row0 + n), (unsigned long long)ix->n);
    const size_t w = (size_t)ix->dim * ix->esz;
    for (const SegRange &r : locate(ix, row0, n))
        PVS_TRY(pvs_index_read_rows(ix->shards[r.shard], r.local0, r.n, (uint8_t *)out_host + r.out_off * w));
    return PVS_OK;
}

pvs_status multi_read_ids(pvs_index *ix, uint64_t row0, uint64_t n, int64_t *out_row_ids, int64_t *out_group_ids) {
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, "this multi-device index lost its row order in a failed pvs_index_add: destroy and rebuild it");
    if (row0 + n > ix->n) return pvs_fail(PVS_ERR_INVALID_ARG, "row range [%llu, %llu) exceeds %llu rows", (unsigned long long)row0,
                                          (unsigned long long)(row0 + n), (unsigned long long)ix->n);
    for (const SegRange &r : locate(ix, row0, n))
        PVS_TRY(pvs_index_read_ids(ix->shards[r.shard], r.local0, r.n, out_row_ids + r.out_off, out_group_ids ? out_group_ids + r.out_off : nullptr));
    return PVS_OK;
}

pvs_status multi_search_device(pvs_index *ix, const void *d_queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                               int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count, uint32_t *out_ticket) {
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, "this multi-device index lost its row order in a failed pvs_index_add: destroy and rebuild it");
    PVS_TRY(validate_search(ix->shards[0], d_queries, qdtype, batch, k, metric));
    if (!d_out_ids || !d_out_dist || !d_out_count || !out_ticket) return pvs_fail(PVS_ERR_INVALID_ARG, "null output");
    if (batch == 0) return pvs_fail(PVS_ERR_INVALID_ARG, "empty batch");
    uint32_t t;
    MultiCtx *m = mctx_acquire(ix, &t, false);
    if (!m) return PVS_ERR_STATE;
    pvs_status st = mctx_prepare(ix, *m, batch, k);
    if (st == PVS_OK) st = multi_enqueue(ix, *m, d_queries, qdtype, batch, k, metric, d_out_ids, d_out_dist, d_out_count);
    if (st != PVS_OK) {
        for (size_t s = 0; s < m->tickets.size(); s++) {
            (void)hipSetDevice(ix->shards[s]->device);
            (void)hipStreamSynchronize(ix->shards[s]->ctx[m->tickets[s]].stream);
        }
        release_shard_ctxs(ix, *m);
        mctx_done(ix, m);
        return st;
    }
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        m->pending = true;
    }
    ix->searches++;
    *out_ticket = t;
    return PVS_OK;
}

// the pvs_wait work of a pending multi-device ticket, its shard contexts released; the multi context stays held (pvs_gate.hip)
pvs_status multi_ticket_complete_(pvs_index *ix, uint32_t ticket) {
    MultiCtx *m = &ix->mctx[ticket];
    pvs_status st = multi_complete(ix, *m);
    release_shard_ctxs(ix, *m);
    return st;
}

pvs_status multi_wait(pvs_index *ix, uint32_t ticket) {
    MultiCtx *m = &ix->mctx[ticket];
    pvs_status st = PVS_OK;
    bool mine = false;
    {
        std::unique_lock<std::mutex> lk(ix->mu);
        if (!m->busy || !(m->pending || m->draining || m->finished)) return pvs_fail(PVS_ERR_STATE, "ticket %u has no search in flight", ticket);
        while (m->draining) ix->ctx_cv.wait(lk);  // a writer is completing it on our behalf
        if (!m->busy || !(m->pending || m->finished)) return pvs_fail(PVS_ERR_STATE, "ticket %u has no search in flight", ticket);
        if (m->finished) {
            st = m->fin_status;
            if (st != PVS_OK) pvs_fail(st, "%s", m->fin_err.c_str());
        } else {
            m->draining = true;
            mine = true;
        }
    }
    if (mine) st = multi_ticket_complete_(ix, ticket);
    mctx_done(ix, m);
    return st;
}

pvs_status multi_sync(pvs_index *ix) {
    pvs_status st = PVS_OK;
    for (uint32_t i = 0; i < NCTX; i++) {
        bool live;
        {
            std::lock_guard<std::mutex> lk(ix->mu);
            live = ix->mctx[i].busy && (ix->mctx[i].pending || ix->mctx[i].draining || ix->mctx[i].finished);
        }
        if (live) {
            pvs_status s = multi_wait(ix, i);
            if (s != PVS_OK) st = s;
        }
    }
    return st;
}

pvs_status multi_search_host(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                             int64_t *out_ids, float *out_dist, uint32_t *out_count) {
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, "this multi-device index lost its row order in a failed pvs_index_add: destroy and rebuild it");
    PVS_TRY(validate_search(ix->shards[0], queries, qdtype, batch, k, metric));
    if (!out_ids || !out_dist || !out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null output");
    if (batch == 0) return PVS_OK;
    uint32_t t;
    MultiCtx *m = mctx_acquire(ix, &t, true);
    auto body = [&]() -> pvs_status {
        PVS_TRY(mctx_prepare(ix, *m, batch, k));
        const size_t qbytes = (size_t)ix->dim * (qdtype == PVS_I8 ? 1 : 4) * batch;
        if (qbytes > m->qroot_cap) {
            hipFree(m->d_qroot);
            m->d_qroot = nullptr;
            m->qroot_cap = 0;
            const size_t cap = pvs_round_up(qbytes, 1 << 16);
            HIP_TRY(pvs_malloc_retry(&m->d_qroot, cap));
            m->qroot_cap = cap;
        }
        const uint64_t need = (uint64_t)batch * k;
        if (need > m->out_cap || batch > m->out_batch_cap) {
            hipFree(m->d_out_ids);
            hipFree(m->d_out_dist);
            hipFree(m->d_out_cnt);
            m->d_out_ids = nullptr;
            m->d_out_dist = nullptr;
            m->d_out_cnt = nullptr;
            m->out_cap = 0;
            HIP_TRY(pvs_malloc_retry((void **)&m->d_out_ids, need * 8));
            HIP_TRY(pvs_malloc_retry((void **)&m->d_out_dist, need * 4));
            HIP_TRY(pvs_malloc_retry((void **)&m->d_out_cnt, (size_t)batch * 4));
            m->out_cap = need;
            m->out_batch_cap = batch;
        }
        HIP_TRY(hipMemcpyAsync(m->d_qroot, queries, qbytes, hipMemcpyHostToDevice, m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));  // the shard streams read (or peer-copy) the staged queries
        pvs_status st = multi_enqueue(ix, *m, m->d_qroot, qdtype, batch, k, metric, m->d_out_ids, m->d_out_dist, m->d_out_cnt);
        if (st == PVS_OK) st = multi_complete(ix, *m);
        if (st != PVS_OK) {
            for (size_t s = 0; s < m->tickets.size(); s++) {
                (void)hipSetDevice(ix->shards[s]->device);
                (void)hipStreamSynchronize(ix->shards[s]->ctx[m->tickets[s]].stream);
            }
            return st;
        }
        HIP_TRY(hipSetDevice(root_device(ix)));
        HIP_TRY(hipMemcpyAsync(out_ids, m->d_out_ids, need * 8, hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipMemcpyAsync(out_dist, m->d_out_dist, need * 4, hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipMemcpyAsync(out_count, m->d_out_cnt, (size_t)batch * 4, hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));
        return PVS_OK;
    };
    pvs_status st = body();
    release_shard_ctxs(ix, *m);
    ix->searches++;
    mctx_done(ix, m);
    return st;
}

pvs_status multi_score_all(pvs_index *ix, const void *query, pvs_dtype qdtype, pvs_metric metric, float *out_dist, pvs_space out_space) {
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, "this multi-device index lost its row order in a failed pvs_index_add: destroy and rebuild it");
    PVS_TRY(validate_search(ix->shards[0], query, qdtype, 1, 1, metric));
    if (!out_dist) return pvs_fail(PVS_ERR_INVALID_ARG, "null output");
    if (out_space != PVS_HOST) return pvs_fail(PVS_ERR_UNSUPPORTED, "pvs_score_all on a multi-device index writes host memory only");
    // one dense column per shard (they run one after the other: this is the build-side / SQL-seam entry point, not the hot one),
    // scattered into global row order
    std::vector<float> col;
    for (uint32_t s = 0; s < ix->shards.size(); s++) {
        pvs_index *sh = ix->shards[s];
        if (sh->n == 0) continue;
        col.resize(sh->n);
        PVS_TRY(pvs_score_all(sh, query, qdtype, metric, col.data(), PVS_HOST));
        for (const MultiSegment &g : ix->segs)
            if (g.shard == s) memcpy(out_dist + g.row0, col.data() + g.local0, g.n * 4);
    }
    return PVS_OK;
}

// key of group g under pvs_index_set_order_keys (the key of the group's first row; built with the group CSR, ensure_groups)
bool index_group_key(const pvs_index *ix, int64_t g, int64_t *key) {
    if (is_multi(ix)) {
        if (!ix->by_group || ix->order_rows != ix->n) return false;
        return index_group_key(ix->shards[multi_group_shard(g, (uint32_t)ix->shards.size())], g, key);
    }
    if (ix->h_grp_key.empty()) return false;
    auto it = std::lower_bound(ix->h_grp_ids.begin(), ix->h_grp_ids.end(), g);
    if (it == ix->h_grp_ids.end() || *it != g) return false;
    *key = ix->h_grp_key[(size_t)(it - ix->h_grp_ids.begin())];
    return true;
}
