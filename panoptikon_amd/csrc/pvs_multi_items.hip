// pvs_multi_items.hip — the multi-device index beyond the plain row search (split from pvs_multi.hip in round 5): candidate
// masks and row lists, bounded pages, the distance matrix, per-item pages (the shards' pages merged on devices[0] or the host),
// similar_to and the OR arm's fusion with the shards as ranks.  See pvs_multi.hip for the placement and the design.
#include <thread>

#include "pvs_multi.hpp"

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
        PVS_TRY(pvs_host_ids_locked(ix));
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
// the shards' pages -> the caller's page: on devices[0] (one LDS sort for S * k <= 4,096, a merge by rank up to 32,768), else on the host
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

