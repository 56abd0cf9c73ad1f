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
// one LDS sort per query (k_merge_group_pages; S * k <= 4,096, else the host merge).  Row weights arrive in host memory by the
// ABI and are split there.
#include <thread>

#include "pvs_index.hpp"

namespace {
int root_device(const pvs_index *ix) { return ix->shards[0]->device; }

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
        m->busy = false;
    }
    ix->ctx_cv.notify_one();
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
// a global per-row host array split into the shards' row orders (segments are in global order and, per shard, in local order)
template <typename T>
std::vector<std::vector<T>> split_rows(const pvs_index *ix, const T *global) {
    std::vector<std::vector<T>> out(ix->shards.size());
    for (size_t s = 0; s < out.size(); s++) out[s].reserve(ix->shards[s]->n);
    for (const MultiSegment &g : ix->segs) out[g.shard].insert(out[g.shard].end(), global + g.row0, global + g.row0 + g.n);
    return out;
}
struct SegRange {
    uint32_t shard;
    uint64_t local0, n, out_off;  // out_off: offset (rows) inside the caller's range
};
// the pieces of global rows [row0, row0 + n)
std::vector<SegRange> locate(const pvs_index *ix, uint64_t row0, uint64_t n) {
    std::vector<SegRange> out;
    auto it = std::upper_bound(ix->segs.begin(), ix->segs.end(), row0, [](uint64_t r, const MultiSegment &g) { return r < g.row0; });
    if (it != ix->segs.begin()) --it;
    for (; it != ix->segs.end() && it->row0 < row0 + n; ++it) {
        const uint64_t a = std::max(row0, it->row0), b = std::min(row0 + n, it->row0 + it->n);
        if (a < b) out.push_back({it->shard, it->local0 + (a - it->row0), b - a, a - row0});
    }
    return out;
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
    PVS_TRY(multi_sync(ix));
    std::lock_guard<std::mutex> lk(ix->mu);
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

pvs_status multi_wait(pvs_index *ix, uint32_t ticket) {
    MultiCtx *m = &ix->mctx[ticket];
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        if (!m->busy || !m->pending) return pvs_fail(PVS_ERR_STATE, "ticket %u has no search in flight", ticket);
    }
    pvs_status st = multi_complete(ix, *m);
    release_shard_ctxs(ix, *m);
    mctx_done(ix, m);
    return st;
}

pvs_status multi_sync(pvs_index *ix) {
    pvs_status st = PVS_OK;
    for (uint32_t i = 0; i < NCTX; i++) {
        bool live;
        {
            std::lock_guard<std::mutex> lk(ix->mu);
            live = ix->mctx[i].busy && ix->mctx[i].pending;
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

// Host merge of the shards' per-item pages [S][batch][k]: duplicates of a group folded to their minimum, then (value asc, NULL
// last, order key DESC when the index carries keys, group id asc) -> first k
static void merge_group_pages_host(const pvs_index *ix, const int64_t *g, const double *v, const uint32_t *c, uint32_t S, uint32_t batch, uint32_t k,
                                   int64_t *out_groups, double *out_values, uint32_t *out_count) {
    struct GV {
        double v;
        int64_t g, key;
    };
    const size_t elems = (size_t)batch * k;
    const bool keyed = ix->order_rows == ix->n && ix->n;
    std::vector<GV> all;
    for (uint32_t q = 0; q < batch; q++) {
        all.clear();
        for (uint32_t s = 0; s < S; s++)
            for (uint32_t i = 0; i < c[(size_t)s * batch + q]; i++) {
                GV e{v[s * elems + (size_t)q * k + i], g[s * elems + (size_t)q * k + i], 0};
                if (keyed) (void)index_group_key(ix->shards[s], e.g, &e.key);
                all.push_back(e);
            }
        std::sort(all.begin(), all.end(), [](const GV &a, const GV &b) {
            if (a.g != b.g) return a.g < b.g;
            const bool na = a.v != a.v, nb = b.v != b.v;
            if (na != nb) return nb;
            return a.v < b.v;
        });
        size_t w = 0;
        for (size_t i = 0; i < all.size(); i++)
            if (i == 0 || all[i].g != all[i - 1].g) all[w++] = all[i];
        all.resize(w);
        std::sort(all.begin(), all.end(), [](const GV &a, const GV &b) {
            const bool na = a.v != a.v, nb = b.v != b.v;
            if (na != nb) return nb;
            if (!na && a.v != b.v) return a.v < b.v;
            if (a.key != b.key) return a.key > b.key;
            return a.g < b.g;
        });
        const uint32_t nout = (uint32_t)std::min<size_t>(k, all.size());
        for (uint32_t i = 0; i < k; i++) {
            out_groups[(size_t)q * k + i] = i < nout ? all[i].g : -1;
            out_values[(size_t)q * k + i] = i < nout ? all[i].v : __builtin_nan("");
        }
        out_count[q] = nout;
    }
}

// a candidate mask over the global rows as host bytes (device-space masks are read back: the per-shard masks are gathers)
static pvs_status host_mask(pvs_index *ix, const uint8_t *mask, pvs_space space, std::vector<uint8_t> &stage, const uint8_t **out) {
    *out = mask;
    if (!mask || space == PVS_HOST) return PVS_OK;
    stage.resize(ix->n);
    if (ix->n) HIP_TRY(hipMemcpy(stage.data(), mask, ix->n, hipMemcpyDeviceToHost));
    *out = stage.data();
    return PVS_OK;
}
template <typename F>
static pvs_status per_shard(pvs_index *ix, F f) {
    const uint32_t S = (uint32_t)ix->shards.size();
    std::vector<pvs_status> st(S, PVS_OK);
    std::vector<std::string> err(S);
    std::vector<std::thread> th;
    for (uint32_t s = 0; s < S; s++)
        th.emplace_back([&, s]() {
            try {  // (an exception leaving a std::thread is std::terminate)
                st[s] = f(s);
            } catch (const std::bad_alloc &) {
                st[s] = pvs_fail(PVS_ERR_OOM, "out of host memory");
            } catch (...) {
                st[s] = pvs_fail(PVS_ERR_STATE, "unexpected failure");
            }
            if (st[s] != PVS_OK) err[s] = pvs_last_error();
        });
    for (auto &t : th) t.join();
    for (uint32_t s = 0; s < S; s++)
        if (st[s] != PVS_OK) return pvs_fail(st[s], "shard %u: %s", s, err[s].c_str());
    return PVS_OK;
}
static const char *k_need_groups = "needs every row of a group on one device: give group_ids to every pvs_index_add of a multi-device index";

// Host merge of the shards' row pages [S][batch][k] under (distance asc, NULL last, [order key DESC,] id asc); with order keys a
// row's key is found by its id through the global id list (ids increase in global row order)
static pvs_status merge_row_pages_host(pvs_index *ix, const std::vector<int64_t> &ids, const std::vector<float> &dist, const std::vector<uint32_t> &cnt,
                                       uint32_t S, uint32_t batch, uint32_t k, int64_t *out_ids, float *out_dist, uint32_t *out_count) {
    if (!(ix->order_rows == ix->n && ix->n)) return pvs_merge_topk(ids.data(), dist.data(), cnt.data(), S, batch, k, out_ids, out_dist, out_count);
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        if (ix->h_ids_cache.size() != ix->n) {
            ix->h_ids_cache.resize(ix->n);
            PVS_TRY(multi_read_ids(ix, 0, ix->n, ix->h_ids_cache.data(), nullptr));
        }
    }
    const size_t elems = (size_t)batch * k;
    std::vector<int64_t> keys(S * elems, 0);
    for (uint32_t s = 0; s < S; s++)
        for (uint32_t q = 0; q < batch; q++)
            for (uint32_t i = 0; i < cnt[(size_t)s * batch + q] && i < k; i++) {
                const size_t e = s * elems + (size_t)q * k + i;
                const size_t row = (size_t)(std::lower_bound(ix->h_ids_cache.begin(), ix->h_ids_cache.end(), ids[e]) - ix->h_ids_cache.begin());
                if (row < ix->h_order_keys.size()) keys[e] = ix->h_order_keys[row];
            }
    return pvs_merge_topk_keyed(ids.data(), dist.data(), keys.data(), cnt.data(), S, batch, k, out_ids, out_dist, out_count);
}

// pvs_search_filtered on a multi-device index: the mask split into the shards' row orders, one masked search per shard (threads),
// pages merged on the host under (distance asc, id asc, NULL last)
// ---- per-item work of a multi-device index on the devices --------------------------------------------------------------------
// Every shard's global rows in its local order, resident on devices[0] (built from the segment table once per index state): a
// per-row array the caller holds on devices[0] — a candidate mask — is split by one gather per shard there and travels to the
// shard by a peer copy; it never visits the host.
static pvs_status ensure_shard_rows(pvs_index *ix) {
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->shard_rows_n == ix->n && ix->d_shard_rows.size() == ix->shards.size()) return PVS_OK;
    HIP_TRY(hipSetDevice(root_device(ix)));
    for (uint32_t *p : ix->d_shard_rows) hipFree(p);
    ix->d_shard_rows.assign(ix->shards.size(), nullptr);
    ix->shard_rows_n = 0;
    std::vector<std::vector<uint32_t>> rows(ix->shards.size());
    for (size_t s = 0; s < rows.size(); s++) rows[s].reserve(ix->shards[s]->n);
    for (const MultiSegment &g : ix->segs)
        for (uint64_t i = 0; i < g.n; i++) rows[g.shard].push_back((uint32_t)(g.row0 + i));
    for (size_t s = 0; s < rows.size(); s++) {
        HIP_TRY(pvs_malloc_retry((void **)&ix->d_shard_rows[s], std::max<size_t>(rows[s].size(), 1) * 4));
        if (!rows[s].empty()) HIP_TRY(hipMemcpy(ix->d_shard_rows[s], rows[s].data(), rows[s].size() * 4, hipMemcpyHostToDevice));
    }
    ix->shard_rows_n = ix->n;
    return PVS_OK;
}
// is this device pointer resident on devices[0]?
static bool on_root(const pvs_index *ix, const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.device == root_device(ix);
}
// shard s's part of a per-row byte array on devices[0], on the shard's device (scratch: *to_free holds what to give back)
static pvs_status shard_mask_device(pvs_index *ix, uint32_t s, const uint8_t *d_mask_root, const uint8_t **out, void **to_free_root, void **to_free_there) {
    pvs_index *sh = ix->shards[s];
    const int root = root_device(ix);
    uint8_t *lm = nullptr;
    HIP_TRY(hipSetDevice(root));
    HIP_TRY(pvs_scratch_alloc((void **)&lm, sh->n + 64));
    *to_free_root = lm;
    HIP_TRY(pvs_launch_take_rows(d_mask_root, 1, ix->d_shard_rows[s], sh->n, lm, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    *out = lm;
    if (sh->device != root) {
        uint8_t *there = nullptr;
        HIP_TRY(hipSetDevice(sh->device));
        HIP_TRY(pvs_scratch_alloc((void **)&there, sh->n + 64));
        *to_free_there = there;
        HIP_TRY(hipMemcpyPeer(there, sh->device, lm, root, sh->n));
        *out = there;
    }
    return PVS_OK;
}
// Pinned page blocks (host memory every device reads and writes at the same address): the shards' per-item pages land in one,
// devices[0] merges from it.  A small pool: concurrent callers each hold their own.
struct PageLease {
    pvs_index *ix = nullptr;
    int slot = -1;
    uint8_t *p = nullptr;
    ~PageLease() {
        if (slot >= 0) {
            std::lock_guard<std::mutex> lk(ix->mu);
            ix->page_blocks[(size_t)slot].busy = false;
        }
    }
};
static pvs_status page_lease(pvs_index *ix, size_t bytes, PageLease &l) {
    std::lock_guard<std::mutex> lk(ix->mu);
    int slot = -1;
    for (size_t i = 0; i < ix->page_blocks.size(); i++)
        if (!ix->page_blocks[i].busy) {
            slot = (int)i;
            break;
        }
    if (slot < 0) {
        ix->page_blocks.emplace_back();
        slot = (int)ix->page_blocks.size() - 1;
    }
    auto &b = ix->page_blocks[(size_t)slot];
    if (b.cap < bytes) {
        HIP_TRY(hipSetDevice(root_device(ix)));
        if (b.p) hipHostFree(b.p);
        b.p = nullptr;
        b.cap = 0;
        const size_t cap = pvs_round_up(bytes, 1 << 16);
        HIP_TRY(hipHostMalloc((void **)&b.p, cap, hipHostMallocPortable | hipHostMallocMapped));
        b.cap = cap;
    }
    b.busy = true;
    l.ix = ix;
    l.slot = slot;
    l.p = b.p;
    return PVS_OK;
}
// [S][batch][k] groups | values | keys, [S][batch] counts, then the merged page
struct GroupPages {
    int64_t *g = nullptr, *key = nullptr, *og = nullptr;
    double *v = nullptr, *ov = nullptr;
    uint32_t *c = nullptr, *oc = nullptr;
    static size_t bytes(uint32_t S, uint32_t batch, uint32_t k) { return ((size_t)S * 24 + 16) * batch * k + ((size_t)S + 1) * batch * 4 + 64; }
    void carve(uint8_t *p, uint32_t S, uint32_t batch, uint32_t k) {
        const size_t e = (size_t)batch * k;
        g = (int64_t *)p;
        v = (double *)(g + S * e);
        key = (int64_t *)(v + S * e);
        og = key + S * e;
        ov = (double *)(og + e);
        c = (uint32_t *)(ov + e);
        oc = c + (size_t)S * batch;
    }
};
// the shards' pages -> the caller's page: on devices[0] when one LDS sort takes them (S * k <= 4,096), else on the host
static pvs_status merge_group_pages(pvs_index *ix, GroupPages &pg, bool keyed, uint32_t S, uint32_t batch, uint32_t k, int64_t *out_groups, double *out_values,
                                    uint32_t *out_count) {
    if (!pvs_merge_group_pages_supported(S, k) || pvs_dbg(PVS_DBG_MULTI_HOST_PAGES)) {
        merge_group_pages_host(ix, pg.g, pg.v, pg.c, S, batch, k, out_groups, out_values, out_count);
        return PVS_OK;
    }
    HIP_TRY(hipSetDevice(root_device(ix)));
    HIP_TRY(pvs_launch_merge_group_pages(pg.g, pg.v, keyed ? pg.key : nullptr, pg.c, S, batch, k, pg.og, pg.ov, pg.oc, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    memcpy(out_groups, pg.og, (size_t)batch * k * 8);
    memcpy(out_values, pg.ov, (size_t)batch * k * 8);
    memcpy(out_count, pg.oc, (size_t)batch * 4);
    return PVS_OK;
}
// the second sort key of every entry of shard s's page, looked up on the shard's device
static pvs_status shard_page_keys(pvs_index *ix, uint32_t s, GroupPages &pg, uint32_t batch, uint32_t k) {
    pvs_index *sh = ix->shards[s];
    const size_t e = (size_t)batch * k;
    if (!sh->d_grp_key || !sh->d_grp_ids) {
        memset(pg.key + s * e, 0, e * 8);
        return PVS_OK;
    }
    HIP_TRY(hipSetDevice(sh->device));
    HIP_TRY(pvs_launch_page_group_keys(pg.g + s * e, pg.c + (size_t)s * batch, batch, k, sh->d_grp_ids, sh->n_groups, sh->d_grp_key, pg.key + s * e, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return PVS_OK;
}

pvs_status multi_search_filtered(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                                 const uint8_t *mask, pvs_space mask_space, int64_t *out_ids, float *out_dist, uint32_t *out_count) {
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, "this multi-device index lost its row order in a failed pvs_index_add: destroy and rebuild it");
    PVS_TRY(validate_search(ix->shards[0], queries, qdtype, batch, k, metric));
    if (!out_ids || !out_dist || !out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null output");
    if (batch == 0) return PVS_OK;
    // (a mask resident on devices[0] is split there and reaches the shards by peer copies: see shard_mask_device)
    const bool dev_mask = mask && mask_space == PVS_DEVICE && ix->n && ix->n < (1ull << 32) && on_root(ix, mask) && !pvs_dbg(PVS_DBG_MULTI_HOST_PAGES);
    std::vector<uint8_t> stage;
    const uint8_t *hm = nullptr;
    std::vector<std::vector<uint8_t>> masks;
    if (dev_mask) {
        PVS_TRY(ensure_shard_rows(ix));
    } else {
        PVS_TRY(host_mask(ix, mask, mask_space, stage, &hm));
        masks = split_rows<uint8_t>(ix, hm);
    }
    const uint32_t S = (uint32_t)ix->shards.size();
    const size_t elems = (size_t)batch * k;
    std::vector<int64_t> ids(S * elems);
    std::vector<float> dist(S * elems);
    std::vector<uint32_t> cnt((size_t)S * batch, 0);
    PVS_TRY(per_shard(ix, [&](uint32_t s) -> pvs_status {
        if (ix->shards[s]->n == 0) return PVS_OK;
        if (!dev_mask)
            return search_host(ix->shards[s], queries, qdtype, batch, k, metric, masks[s].data(), PVS_HOST, ids.data() + s * elems, dist.data() + s * elems,
                               cnt.data() + (size_t)s * batch);
        const uint8_t *m = nullptr;
        void *free_root = nullptr, *free_there = nullptr;
        pvs_status st = shard_mask_device(ix, s, mask, &m, &free_root, &free_there);
        if (st == PVS_OK)
            st = search_host(ix->shards[s], queries, qdtype, batch, k, metric, m, PVS_DEVICE, ids.data() + s * elems, dist.data() + s * elems,
                             cnt.data() + (size_t)s * batch);
        if (free_there) {
            (void)hipSetDevice(ix->shards[s]->device);
            pvs_scratch_free(free_there);
        }
        if (free_root) {
            (void)hipSetDevice(root_device(ix));
            pvs_scratch_free(free_root);
        }
        return st;
    }));
    ix->searches++;
    return merge_row_pages_host(ix, ids, dist, cnt, S, batch, k, out_ids, out_dist, out_count);
}

// pvs_search_rows on a multi-device index: the global row list split into the shards' row orders (the segment table: both are
// ascending), one search per shard over its own list, pages merged on the host
pvs_status multi_search_rows(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric, const uint32_t *rows,
                             uint64_t n_listed, pvs_space rows_space, int64_t *out_ids, float *out_dist, uint32_t *out_count) {
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, "this multi-device index lost its row order in a failed pvs_index_add: destroy and rebuild it");
    PVS_TRY(validate_search(ix->shards[0], queries, qdtype, batch, k, metric));
    if (!out_ids || !out_dist || !out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null output");
    if (n_listed > ix->n) return pvs_fail(PVS_ERR_INVALID_ARG, "%llu candidate rows for an index of %llu rows", (unsigned long long)n_listed, (unsigned long long)ix->n);
    if (batch == 0) return PVS_OK;
    std::vector<uint32_t> stage;
    const uint32_t *hl = rows;
    if (rows_space == PVS_DEVICE && n_listed) {
        stage.resize(n_listed);
        HIP_TRY(hipMemcpy(stage.data(), rows, n_listed * 4, hipMemcpyDeviceToHost));
        hl = stage.data();
    }
    const uint32_t S = (uint32_t)ix->shards.size();
    std::vector<std::vector<uint32_t>> lists(S);
    size_t seg = 0;
    for (uint64_t i = 0; i < n_listed; i++) {
        const uint64_t r = hl[i];
        if (r >= ix->n) return pvs_fail(PVS_ERR_INVALID_ARG, "candidate rows must be row positions below the index's row count (%llu)", (unsigned long long)ix->n);
        if (i && hl[i - 1] >= hl[i]) return pvs_fail(PVS_ERR_INVALID_ARG, "candidate rows must be strictly ascending");
        while (seg < ix->segs.size() && ix->segs[seg].row0 + ix->segs[seg].n <= r) seg++;
        const MultiSegment &g = ix->segs[seg];
        lists[g.shard].push_back((uint32_t)(g.local0 + (r - g.row0)));
    }
    const size_t elems = (size_t)batch * k;
    std::vector<int64_t> ids(S * elems, -1);
    std::vector<float> dist(S * elems, __builtin_nanf(""));
    std::vector<uint32_t> cnt((size_t)S * batch, 0);
    static const uint32_t empty = 0;
    PVS_TRY(per_shard(ix, [&](uint32_t s) -> pvs_status {
        if (ix->shards[s]->n == 0) return PVS_OK;
        return search_host(ix->shards[s], queries, qdtype, batch, k, metric, nullptr, PVS_HOST, ids.data() + s * elems, dist.data() + s * elems,
                           cnt.data() + (size_t)s * batch, lists[s].empty() ? &empty : lists[s].data(), lists[s].size(), PVS_HOST);
    }));
    ix->searches++;
    return merge_row_pages_host(ix, ids, dist, cnt, S, batch, k, out_ids, out_dist, out_count);
}

// pvs_search_bounded with a lower bound on a multi-device index (pql/builder.rs:781-815): one bounded search per shard — growing
// pages of the filter scan, then that shard's dense path for a bound deeper than PVS_MAX_K rows — merged on the host.  Rows
// outside (gt, lt) are candidates on no shard, so the first k of the merged pages are the first k of the whole index.
pvs_status multi_search_bounded(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric, int32_t have_gt,
                                double gt, int32_t have_lt, double lt, int64_t *out_ids, float *out_dist, uint32_t *out_count) {
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, "this multi-device index lost its row order in a failed pvs_index_add: destroy and rebuild it");
    PVS_TRY(validate_search(ix->shards[0], queries, qdtype, batch, k, metric));
    if (batch == 0) return PVS_OK;
    const uint32_t S = (uint32_t)ix->shards.size();
    const size_t elems = (size_t)batch * k;
    std::vector<int64_t> ids(S * elems, -1);
    std::vector<float> dist(S * elems, __builtin_nanf(""));
    std::vector<uint32_t> cnt((size_t)S * batch, 0);
    PVS_TRY(per_shard(ix, [&](uint32_t s) -> pvs_status {
        if (ix->shards[s]->n == 0) return PVS_OK;
        return pvs_search_bounded(ix->shards[s], queries, qdtype, batch, k, metric, have_gt, gt, have_lt, lt, ids.data() + s * elems, dist.data() + s * elems,
                                  cnt.data() + (size_t)s * batch);
    }));
    ix->searches++;
    return merge_row_pages_host(ix, ids, dist, cnt, S, batch, k, out_ids, out_dist, out_count);
}

// pvs_score_batch on a multi-device index: one dense matrix per shard, scattered into global row order
pvs_status multi_score_batch(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, pvs_metric metric, float *out_dist,
                             pvs_space out_space) {
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, "this multi-device index lost its row order in a failed pvs_index_add: destroy and rebuild it");
    PVS_TRY(validate_search(ix->shards[0], queries, qdtype, batch, 1, metric));
    if (!out_dist) return pvs_fail(PVS_ERR_INVALID_ARG, "null output");
    if (out_space != PVS_HOST) return pvs_fail(PVS_ERR_UNSUPPORTED, "pvs_score_batch on a multi-device index writes host memory only");
    if (batch == 0) return PVS_OK;
    std::vector<float> m;
    for (uint32_t s = 0; s < ix->shards.size(); s++) {
        pvs_index *sh = ix->shards[s];
        if (sh->n == 0) continue;
        m.resize(sh->n * (size_t)batch);
        PVS_TRY(pvs_score_batch(sh, queries, qdtype, batch, metric, m.data(), PVS_HOST));
        for (const MultiSegment &g : ix->segs)
            if (g.shard == s) memcpy(out_dist + g.row0 * batch, m.data() + g.local0 * batch, g.n * (size_t)batch * 4);
    }
    return PVS_OK;
}

// Per-item pages across row shards.  Rows are placed BY GROUP (group_ids given to every add), so every aggregate, row weights
// and candidate masks are shard-local; the shards' pages hold disjoint groups and merge under (value asc, group id asc, NULL
// last).  (The merge below also folds a group that appears in two pages to its minimum: harmless here, and what MIN over
// row-wise shards would need.)
pvs_status multi_search_groups(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                               pvs_agg agg, const float *row_weights, const uint8_t *mask, pvs_space mask_space, int64_t *out_groups,
                               double *out_values, uint32_t *out_count) {
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, "this multi-device index lost its row order in a failed pvs_index_add: destroy and rebuild it");
    PVS_TRY(validate_search(ix->shards[0], queries, qdtype, batch, k, metric));
    if (!out_groups || !out_values || !out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null output");
    if (ix->n && !ix->by_group) return pvs_fail(PVS_ERR_UNSUPPORTED, "per-item search %s", k_need_groups);
    if (batch == 0) return PVS_OK;
    // a candidate mask resident on devices[0] is split there (one gather per shard) and reaches the shards by peer copies; any other
    // mask goes through the host as before
    const bool dev_mask = mask && mask_space == PVS_DEVICE && ix->n && ix->n < (1ull << 32) && on_root(ix, mask) && !pvs_dbg(PVS_DBG_MULTI_HOST_PAGES);
    std::vector<uint8_t> stage;
    const uint8_t *hm = nullptr;
    if (!dev_mask) PVS_TRY(host_mask(ix, mask, mask_space, stage, &hm));
    if (dev_mask) PVS_TRY(ensure_shard_rows(ix));
    std::vector<std::vector<uint8_t>> masks;
    std::vector<std::vector<float>> weights;
    if (hm) masks = split_rows<uint8_t>(ix, hm);
    if (row_weights) weights = split_rows<float>(ix, row_weights);  // (the ABI takes the weights in host memory: each shard uploads its part)
    const uint32_t S = (uint32_t)ix->shards.size();
    const size_t elems = (size_t)batch * k;
    const bool keyed = ix->order_rows == ix->n && ix->n;
    PageLease lease;
    PVS_TRY(page_lease(ix, GroupPages::bytes(S, batch, k), lease));
    GroupPages pg;
    pg.carve(lease.p, S, batch, k);
    memset(pg.c, 0, (size_t)S * batch * 4);
    const bool dev_merge = pvs_merge_group_pages_supported(S, k) && !pvs_dbg(PVS_DBG_MULTI_HOST_PAGES);
    PVS_TRY(per_shard(ix, [&](uint32_t s) -> pvs_status {
        if (ix->shards[s]->n == 0) return PVS_OK;
        const uint8_t *m = hm ? masks[s].data() : nullptr;
        pvs_space ms = PVS_HOST;
        void *free_root = nullptr, *free_there = nullptr;
        pvs_status st = PVS_OK;
        if (dev_mask) {
            st = shard_mask_device(ix, s, mask, &m, &free_root, &free_there);
            ms = PVS_DEVICE;
        }
        if (st == PVS_OK)
            st = search_groups_impl(ix->shards[s], queries, qdtype, batch, k, metric, agg, row_weights ? weights[s].data() : nullptr, m, ms, pg.g + s * elems,
                                    pg.v + s * elems, pg.c + (size_t)s * batch);
        if (st == PVS_OK && keyed && dev_merge) st = shard_page_keys(ix, s, pg, batch, k);
        if (free_there) {
            (void)hipSetDevice(ix->shards[s]->device);
            pvs_scratch_free(free_there);  // (the search returned: nothing in flight reads the mask)
        }
        if (free_root) {
            (void)hipSetDevice(root_device(ix));
            pvs_scratch_free(free_root);
        }
        return st;
    }));
    PVS_TRY(merge_group_pages(ix, pg, keyed, S, batch, k, out_groups, out_values, out_count));
    ix->searches++;
    return PVS_OK;
}

// similar_to on a multi-device index placed BY GROUP: the target vectors (read from the shards that own them) are scored against
// every shard; the target rows are left out on their own shard; the shards' pages hold disjoint groups and merge on the host.
pvs_status multi_similar_to(pvs_index *ix, const int64_t *target_row_ids, uint32_t n_targets, uint32_t k, pvs_metric metric, const SimilarArgs &a,
                            int64_t *out_groups, double *out_values, uint32_t *out_count) {
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, "this multi-device index lost its row order in a failed pvs_index_add: destroy and rebuild it");
    if (ix->n && !ix->by_group) return pvs_fail(PVS_ERR_UNSUPPORTED, "similar_to %s", k_need_groups);
    std::vector<uint64_t> trow;
    SimilarTargets tg;
    PVS_TRY(similar_targets(ix, target_row_ids, n_targets, a, trow, tg));
    const uint32_t S = (uint32_t)ix->shards.size();
    std::vector<std::vector<uint32_t>> excluded(S);
    for (uint32_t i = 0; i < n_targets; i++)
        for (const SegRange &r : locate(ix, trow[i], 1)) excluded[r.shard].push_back((uint32_t)r.local0);
    std::vector<std::vector<double>> conf, lang;
    std::vector<std::vector<uint8_t>> kind;
    if (a.row_conf) conf = split_rows<double>(ix, a.row_conf);
    if (a.row_lang) lang = split_rows<double>(ix, a.row_lang);
    if (a.row_kind) kind = split_rows<uint8_t>(ix, a.row_kind);
    const size_t elems = k;
    std::vector<int64_t> g(S * elems);
    std::vector<double> v(S * elems);
    std::vector<uint32_t> c(S, 0);
    PVS_TRY(per_shard(ix, [&](uint32_t s) -> pvs_status {
        if (ix->shards[s]->n == 0) return PVS_OK;
        SimilarArgs mine = a;
        mine.row_conf = a.row_conf ? conf[s].data() : nullptr;
        mine.row_lang = a.row_lang ? lang[s].data() : nullptr;
        mine.row_kind = a.row_kind ? kind[s].data() : nullptr;
        return similar_core(ix->shards[s], tg, n_targets, excluded[s], k, metric, mine, g.data() + s * elems, v.data() + s * elems, &c[s]);
    }));
    ix->searches++;
    {
        const bool keyed = ix->order_rows == ix->n && ix->n;
        PageLease lease;
        PVS_TRY(page_lease(ix, GroupPages::bytes(S, 1, k), lease));
        GroupPages pg;
        pg.carve(lease.p, S, 1, k);
        memcpy(pg.g, g.data(), S * elems * 8);
        memcpy(pg.v, v.data(), S * elems * 8);
        memcpy(pg.c, c.data(), (size_t)S * 4);
        if (keyed && pvs_merge_group_pages_supported(S, k) && !pvs_dbg(PVS_DBG_MULTI_HOST_PAGES))
            for (uint32_t s = 0; s < S; s++) PVS_TRY(shard_page_keys(ix, s, pg, 1, k));
        PVS_TRY(merge_group_pages(ix, pg, keyed, S, 1, k, out_groups, out_values, out_count));
    }
    return PVS_OK;
}

// pvs_rrf_search over multi-device branches placed BY GROUP: shard s of every branch is rank s of the sharded protocol
// (pvs_rrf_search_sharded: pages of each branch's ranking, candidates' exact keys and ranks summed over the ranks); the ranks
// are threads of this process and the all-gather is a rendezvous in host memory.  Every rank returns the same page.
namespace {
struct Rendezvous {
    std::mutex mu;
    std::condition_variable cv;
    uint32_t world = 0, arrived = 0, readers = 0;
    uint64_t gen = 0;
    bool aborted = false;  // a rank left the protocol with an error: nobody may wait for it any more
    std::vector<uint8_t> buf;
};
struct RankCtx {
    Rendezvous *z;
    uint32_t rank;
};
int32_t rendezvous_gather(void *ctx, const void *send, void *recv, uint64_t bytes) {
    RankCtx *r = (RankCtx *)ctx;
    Rendezvous &z = *r->z;
    std::unique_lock<std::mutex> lk(z.mu);
    if (z.aborted) return 1;
    if (z.arrived == 0) z.buf.resize((size_t)z.world * bytes);  // (the previous round's readers are all gone: second wait below)
    if (z.buf.size() != (size_t)z.world * bytes) return 1;     // ranks disagree on the message size
    memcpy(z.buf.data() + (size_t)r->rank * bytes, send, bytes);
    const uint64_t g = z.gen;
    if (++z.arrived == z.world) {
        z.arrived = 0;
        z.readers = z.world;
        z.gen++;
        z.cv.notify_all();
    } else {
        z.cv.wait(lk, [&] { return z.gen != g || z.aborted; });
        if (z.gen == g) return 1;  // released by a departing rank, not by the round completing
    }
    memcpy(recv, z.buf.data(), (size_t)z.world * bytes);
    if (--z.readers == 0)
        z.cv.notify_all();
    else
        z.cv.wait(lk, [&] { return z.readers == 0 || z.aborted; });
    return z.aborted ? 1 : 0;
}
}  // namespace

pvs_status multi_rrf_search(const pvs_rrf_branch *br, uint32_t nb, uint32_t k, int64_t *out_groups, double *out_scores, uint32_t *out_count) {
    const uint32_t S = (uint32_t)br[0].idx->shards.size();
    for (uint32_t b = 0; b < nb; b++) {
        pvs_index *ix = br[b].idx;
        if (!ix || !is_multi(ix) || ix->shards.size() != S)
            return pvs_fail(PVS_ERR_INVALID_ARG, "pvs_rrf_search: either every branch is a single-device index or every branch a multi-device index over the same number of devices");
        if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, "this multi-device index lost its row order in a failed pvs_index_add: destroy and rebuild it");
        if (ix->n && !ix->by_group) return pvs_fail(PVS_ERR_UNSUPPORTED, "pvs_rrf_search %s", k_need_groups);
    }
    std::vector<std::vector<std::vector<float>>> weights(nb);
    for (uint32_t b = 0; b < nb; b++)
        if (br[b].row_weights) weights[b] = split_rows<float>(br[b].idx, br[b].row_weights);
    Rendezvous z;
    z.world = S;
    std::vector<RankCtx> rc(S);
    std::vector<std::vector<int64_t>> og(S, std::vector<int64_t>(k));
    std::vector<std::vector<double>> os(S, std::vector<double>(k));
    std::vector<uint32_t> oc(S, 0);
    PVS_TRY(per_shard(br[0].idx, [&](uint32_t s) -> pvs_status {
        std::vector<pvs_rrf_branch> mine(br, br + nb);
        for (uint32_t b = 0; b < nb; b++) {
            mine[b].idx = br[b].idx->shards[s];
            mine[b].row_weights = br[b].row_weights ? weights[b][s].data() : nullptr;
        }
        rc[s] = {&z, s};
        pvs_status st = pvs_rrf_search_sharded(mine.data(), nb, k, nullptr, S, rendezvous_gather, &rc[s], og[s].data(), os[s].data(), &oc[s]);
        if (st != PVS_OK) {  // this rank is out of the protocol: release whoever waits for it in the rendezvous
            std::lock_guard<std::mutex> lk(z.mu);
            z.aborted = true;
            z.cv.notify_all();
        }
        return st;
    }));
    memcpy(out_groups, og[0].data(), (size_t)k * 8);
    memcpy(out_scores, os[0].data(), (size_t)k * 8);
    *out_count = oc[0];
    return PVS_OK;
}

