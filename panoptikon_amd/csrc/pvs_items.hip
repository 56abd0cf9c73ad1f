// pvs_items.hip — C ABI of libpvs, part 3: per-item results on the device — the group table, dense score matrices, GROUP BY
// aggregates and their ranking (pvs_search_groups*), the sharded per-item search.  similar_to: pvs_similar.hip; the OR arm (RRF):
// pvs_rrf_search.hip.
#include <chrono>
#include <string>
#include <thread>

#include "pvs_index.hpp"

// ------------------------------------------------- groups, dense scores, similar_to
pvs_status ensure_groups(pvs_index *ix) {
    if (ix->groups_built_n == ix->n) return PVS_OK;
    if (!ix->h_groups.empty() && ix->h_groups.size() != ix->n) return pvs_fail(PVS_ERR_STATE, "group ids missing for some rows");
    hipFree(ix->d_grp_off);
    hipFree(ix->d_grp_rows);
    hipFree(ix->d_grp_ids);
    hipFree(ix->d_grp_tinv);
    hipFree(ix->d_grp_trank);
    hipFree(ix->d_tile_grp);
    hipFree(ix->d_straddlers);
    hipFree(ix->d_row_gidx);
    hipFree(ix->d_grp_key);
    ix->d_grp_key = nullptr;
    ix->d_grp_off = ix->d_grp_rows = ix->d_grp_tinv = ix->d_grp_trank = ix->d_straddlers = ix->d_row_gidx = nullptr;
    ix->d_grp_ids = nullptr;
    ix->d_tile_grp = nullptr;
    ix->groups_are_runs = false;
    ix->n_straddlers = 0;
    ix->h_grp_ids.clear();
    ix->h_grp_key.clear();
    const uint64_t n = ix->n;
    std::vector<uint32_t> off, rows(n);
    std::vector<int64_t> gids;
    if (ix->h_groups.empty()) {  // identity: one group per row, the group id is the row id
        gids.resize(n);
        if (n) HIP_TRY(hipMemcpy(gids.data(), ix->d_ids, n * 8, hipMemcpyDeviceToHost));
        off.resize(n + 1);
        for (uint64_t i = 0; i <= n; i++) off[i] = (uint32_t)i;
        for (uint64_t i = 0; i < n; i++) rows[i] = (uint32_t)i;
    } else {
        std::vector<uint32_t> order(n);
        for (uint64_t i = 0; i < n; i++) order[i] = (uint32_t)i;
        const int64_t *g = ix->h_groups.data();
        std::stable_sort(order.begin(), order.end(), [g](uint32_t a, uint32_t b) { return g[a] < g[b]; });  // rows stay ascending inside a group
        for (uint64_t i = 0; i < n; i++) {
            if (i == 0 || g[order[i]] != g[order[i - 1]]) {
                gids.push_back(g[order[i]]);
                off.push_back((uint32_t)i);
            }
            rows[i] = order[i];
        }
        off.push_back((uint32_t)n);
    }
    ix->n_groups = (uint32_t)gids.size();
    HIP_TRY(pvs_malloc_retry((void **)&ix->d_grp_off, (off.size() + 1) * 4));
    HIP_TRY(pvs_malloc_retry((void **)&ix->d_grp_rows, (n + 1) * 4));
    HIP_TRY(pvs_malloc_retry((void **)&ix->d_grp_ids, (gids.size() + 1) * 8));
    HIP_TRY(hipMemcpy(ix->d_grp_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    if (n) HIP_TRY(hipMemcpy(ix->d_grp_rows, rows.data(), n * 4, hipMemcpyHostToDevice));
    if (!gids.empty()) HIP_TRY(hipMemcpy(ix->d_grp_ids, gids.data(), gids.size() * 8, hipMemcpyHostToDevice));
    if (n && n < (1ull << 32)) {  // row -> group slot
        std::vector<uint32_t> gidx(n);
        for (uint32_t g = 0; g + 1 < off.size(); g++)
            for (uint32_t e = off[g]; e < off[g + 1]; e++) gidx[rows[e]] = g;
        HIP_TRY(pvs_malloc_retry((void **)&ix->d_row_gidx, n * 4));
        HIP_TRY(hipMemcpy(ix->d_row_gidx, gidx.data(), n * 4, hipMemcpyHostToDevice));
    }
    // Are the groups runs of consecutive rows (the CSR order is the row order)?  Then the per-item scorer can fold a group in
    // the epilogue of the tile that holds it (k_scan MODE 2 + ScanK.tile_grp): per 32-row tile a record {group of its first row,
    // rows that end their group, rows whose group crosses a tile boundary}, and the list of those crossing groups.
    {
        bool runs = n < (1ull << 31);
        for (uint64_t i = 0; i < n && runs; i++) runs = rows[i] == (uint32_t)i;
        if (runs && n) {
            const uint64_t n_tiles = (n + 31) / 32;
            std::vector<uint32_t> rec(n_tiles * 4, 0), strad;
            const uint32_t G = (uint32_t)gids.size();
            for (uint32_t g = 0; g < G; g++) {
                const uint32_t a = off[g], b = off[g + 1] - 1;  // first and last row
                rec[(size_t)(b >> 5) * 4 + 1] |= 1u << (b & 31);
                if ((a >> 5) != (b >> 5)) {
                    strad.push_back(g);
                    for (uint32_t r = a; r <= b; r++) rec[(size_t)(r >> 5) * 4 + 2] |= 1u << (r & 31);
                }
                // tiles whose first row belongs to this group
                for (uint64_t t = (a + 31) >> 5; t * 32 <= b; t++) rec[t * 4] = g;
            }
            HIP_TRY(pvs_malloc_retry((void **)&ix->d_tile_grp, rec.size() * 4));
            HIP_TRY(hipMemcpy(ix->d_tile_grp, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
            HIP_TRY(pvs_malloc_retry((void **)&ix->d_straddlers, (strad.size() + 1) * 4));
            if (!strad.empty()) HIP_TRY(hipMemcpy(ix->d_straddlers, strad.data(), strad.size() * 4, hipMemcpyHostToDevice));
            ix->n_straddlers = (uint32_t)strad.size();
            ix->groups_are_runs = true;
        }
    }
    if (ix->order_rows == n && n) {
        // second sort key of the per-item pages (pql/model.rs:547-553: ORDER BY order_rank, last_modified DESC): a group's key is
        // its first row's (the rows of a file share files.last_modified); groups in (key DESC, group id ASC) order feed the stable
        // value sort of pvs_group_rank
        const uint32_t G = (uint32_t)gids.size();
        std::vector<int64_t> gkey(G);
        for (uint32_t g = 0; g < G; g++) gkey[g] = ix->h_order_keys[rows[off[g]]];
        std::vector<uint32_t> tinv(G);
        for (uint32_t g = 0; g < G; g++) tinv[g] = g;
        std::stable_sort(tinv.begin(), tinv.end(), [&](uint32_t a, uint32_t b) { return gkey[a] > gkey[b]; });
        HIP_TRY(pvs_malloc_retry((void **)&ix->d_grp_tinv, (size_t)G * 4));
        HIP_TRY(hipMemcpy(ix->d_grp_tinv, tinv.data(), (size_t)G * 4, hipMemcpyHostToDevice));
        std::vector<uint32_t> trank(G);  // its inverse: the tie position of a group (the device-side page ranking sorts by it)
        for (uint32_t i = 0; i < G; i++) trank[tinv[i]] = i;
        HIP_TRY(pvs_malloc_retry((void **)&ix->d_grp_trank, (size_t)G * 4));
        HIP_TRY(hipMemcpy(ix->d_grp_trank, trank.data(), (size_t)G * 4, hipMemcpyHostToDevice));
        HIP_TRY(pvs_malloc_retry((void **)&ix->d_grp_key, (size_t)std::max<uint32_t>(G, 1) * 8));
        HIP_TRY(hipMemcpy(ix->d_grp_key, gkey.data(), (size_t)G * 8, hipMemcpyHostToDevice));
        ix->h_grp_ids = gids;
        ix->h_grp_key = std::move(gkey);
    }
    ix->groups_built_n = n;
    return PVS_OK;
}

// d_out[row * nb + q], nb <= PVS_MAX_BATCH queries already prepared in ctx c (prep_chunk)
pvs_status dense_chunk(pvs_index *ix, SearchCtx &c, uint32_t nb, uint32_t batch_pad, int metric, float *d_out, uint32_t *h_flag, bool inorder_only) {
    const uint32_t kslabs = ix->stride / PVS_KSLAB_BYTES;
    const bool no_direct = pvs_dbg(PVS_DBG_NO_DIRECT_SCORE) != 0;  // tuning: compare with the matrix-core scorer
    if (inorder_only) {
        // (the caller saw the out-of-range flag of an earlier, unwaited call)
    } else if (ix->dtype == PVS_I8 && nb <= 4 && !no_direct && (uint64_t)ix->dim * 127 * 127 < (1u << 24)) {
        // a handful of queries: a pure HBM stream, v_dot4 straight from global memory (pvs_score_direct.hip); same closed form
        if (h_flag)
            *(volatile uint32_t *)h_flag = 0;
        else
            HIP_TRY(hipMemsetAsync(c.d_cand_cnt, 0, 4, c.stream));
        span_begin(ix, c, 1, ix->n);
        HIP_TRY(pvs_launch_score_i8_direct(metric, ix->d_rows, ix->stride, ix->dim, ix->n, ix->d_norm2, c.d_qexact, c.d_qinfo, nb, d_out, nb,
                                           h_flag ? h_flag : c.d_cand_cnt, (uint32_t)ix->n_cu, c.stream));
        span_end(ix, c);
        if (h_flag) return PVS_OK;
        uint32_t flag = 0;
        HIP_TRY(hipMemcpyAsync(&flag, c.d_cand_cnt, 4, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
        spans_collect(ix, c);
        if (!flag) return PVS_OK;  // else: some L2 sum left the exact range -> score in order below
    } else if (ix->dtype == PVS_I8 && pvs_scan_supported(PVS_I8, kslabs) && (uint64_t)ix->dim * 127 * 127 < (1u << 24)) {
        // matrix-core path: exact integer dots, closed-form finish (valid below 2^24)
        ScanArgs a;
        a.dtype = PVS_I8;
        a.metric = metric;
        a.kslabs = kslabs;
        a.qgroups = batch_pad / 32;
        a.rows = ix->d_rows;
        a.aux = ix->d_scan_l2;  // MODE 2 streams |a|^2
        a.stride = ix->stride;
        a.n_rows = ix->n;
        a.qmat = c.d_qmat;
        a.qinfo = c.d_qinfo;
        a.thr = c.d_thr;
        a.gmin = c.d_gmin;
        a.groups_per_query = 0;
        a.mode = 2;
        a.tile_step = 1;
        const uint32_t wg_rows = 32u * pvs_scan_row_tiles(a.qgroups);  // (MODE 2 always runs on k_scan)
        const uint32_t n_wgtiles = (uint32_t)((ix->n + wg_rows - 1) / wg_rows);
        const uint32_t per_cu = (a.kslabs > 4 || (a.qgroups == 1 && a.kslabs == 4)) ? 1 : 2;  // (MODE 2: two workgroups per CU at 32 queries up to 768-B rows, scan_fold2)
        a.grid = std::min<uint32_t>(n_wgtiles, (uint32_t)ix->n_cu * per_cu);
        a.dense_out = d_out;
        a.dense_ld = nb;
        a.batch = nb;
        a.dense_flag = h_flag ? h_flag : c.d_cand_cnt;  // reused as the out-of-range flag word
        if (h_flag)
            *(volatile uint32_t *)h_flag = 0;
        else
            HIP_TRY(hipMemsetAsync(c.d_cand_cnt, 0, 4, c.stream));
        span_begin(ix, c, 1, ix->n);  // (profiling: the dense scorer is the dominant kernel of the per-item paths)
        HIP_TRY(pvs_launch_scan(a, c.stream));
        span_end(ix, c);
        if (h_flag) return PVS_OK;
        uint32_t flag = 0;
        HIP_TRY(hipMemcpyAsync(&flag, c.d_cand_cnt, 4, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
        spans_collect(ix, c);
        if (!flag) return PVS_OK;  // else: some L2 sum left the exact range -> score in order below
    }
    span_begin(ix, c, 1, ix->n);
    HIP_TRY(pvs_launch_dense_exact((int)ix->dtype, metric, ix->d_rows, ix->stride, ix->dim, ix->n, ix->d_norm2, c.d_qexact, c.d_qinfo, nb,
                                   c.d_qpad, d_out, nb, 0, (uint32_t)ix->n_cu, c.stream));
    span_end(ix, c);
    if (ix->profiling) {
        HIP_TRY(hipStreamSynchronize(c.stream));
        spans_collect(ix, c);
    }
    return PVS_OK;
}

// queries per dense chunk so that the [n][nb] f32 matrix stays <= 2 GiB
static uint32_t dense_chunk_queries(const pvs_index *ix, uint32_t batch) {
    const uint64_t cap = (1ull << 31) / (4 * std::max<uint64_t>(ix->n, 1));
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>({cap, (uint64_t)PVS_MAX_BATCH, (uint64_t)batch}));
}

PVS_EXPORT pvs_status pvs_score_batch(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, pvs_metric metric,
                                      float *out_dist, pvs_space out_space) {
    GateShared gate(ix);  // (pvs_gate.hip: a mutation waits for this call, a search never sees one half done)
    if (ix && is_multi(ix)) return multi_score_batch(ix, queries, qdtype, batch, metric, out_dist, out_space);
    PVS_TRY(validate_search(ix, queries, qdtype, batch, 1, metric));
    if (!out_dist) return pvs_fail(PVS_ERR_INVALID_ARG, "null output");
    if (batch == 0 || ix->n == 0) return PVS_OK;
    HIP_TRY(hipSetDevice(ix->device));
    uint32_t t;
    SearchCtx *c = ctx_acquire(ix, &t);
    void *d_q = nullptr;
    float *d_m = nullptr;
    auto body = [&]() -> pvs_status {
        PVS_TRY(ctx_prepare(ix, *c, batch, 1, false));
        const size_t qbytes = (size_t)ix->dim * (qdtype == PVS_I8 ? 1 : 4);
        HIP_TRY(pvs_scratch_alloc(&d_q, qbytes * batch));  // (cached blocks: hipMalloc / hipFree per call would stall every other host thread's searches)
        HIP_TRY(hipMemcpyAsync(d_q, queries, qbytes * batch, hipMemcpyHostToDevice, c->stream));
        const uint32_t cq = dense_chunk_queries(ix, batch);
        HIP_TRY(pvs_scratch_alloc((void **)&d_m, (size_t)ix->n * cq * 4));
        for (uint32_t q0 = 0; q0 < batch; q0 += cq) {
            const uint32_t nb = std::min(cq, batch - q0);
            const uint32_t pad = nb <= 32 ? 32 : nb <= 64 ? 64 : 128;
            PVS_TRY(prep_chunk(ix, *c, d_q, qdtype, q0, nb, pad, metric));
            PVS_TRY(dense_chunk(ix, *c, nb, pad, metric, d_m));
            // scatter the chunk's columns into out[row * batch + q0 + j]
            const hipMemcpyKind kind = out_space == PVS_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
            if (nb == batch)  // one chunk: the matrix IS the output layout — one linear copy (a 2-D copy of 4M rows of 32 bytes ran at 2 GB/s)
                HIP_TRY(hipMemcpyAsync(out_dist, d_m, (size_t)ix->n * nb * 4, kind, c->stream));
            else
                HIP_TRY(hipMemcpy2DAsync(out_dist + q0, (size_t)batch * 4, d_m, (size_t)nb * 4, (size_t)nb * 4, ix->n, kind, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        return PVS_OK;
    };
    pvs_status st = body();
    pvs_scratch_free_on(d_q, c->stream);
    pvs_scratch_free_on(d_m, c->stream);
    ctx_done(ix, c);
    return st;
}

// Columns of group values -> each column's first k groups, through pages of the groups at or below a sampled threshold (see
// aggregate_and_rank).  done[q] = 0: the caller ranks every group of that column instead.  Three host round trips for all columns.
static pvs_status rank_groups_page_first(pvs_index *ix, SearchCtx &c, const double *d_vals, const double *d_vals_t, uint32_t G, uint32_t ncol, uint32_t k,
                                         int64_t *out_groups, double *out_values, uint32_t *out_count, std::vector<uint8_t> &done) {
    done.assign(ncol, 0);
    const uint64_t target = std::max<uint64_t>(4ull * k, 2048);
    if (target * 4 >= G) return PVS_OK;
    unsigned long long *d_keys = nullptr;
    auto body = [&]() -> pvs_status {
        HIP_TRY(pvs_scratch_alloc((void **)&d_keys, (size_t)G * ncol * 8));
        if (d_vals)
            HIP_TRY(pvs_group_page_keys(d_vals, G * ncol, d_keys, c.stream));  // (elementwise: the columns are contiguous)
        else
            HIP_TRY(pvs_group_page_keys_t(d_vals_t, G, ncol, d_keys, c.stream));  // group-major values (the fused scorer): transposed on the way
        const uint32_t M = 4096;
        std::vector<unsigned long long> sample((size_t)M * ncol), thr(ncol);
        PVS_TRY(pvs_rrf_sample_keys_cols(d_keys, G, ncol, M, sample.data(), c.stream));
        // the sample's quantile that should admit ~2 x target groups (a noisy order statistic: the page is exact whatever it is)
        const size_t j = (size_t)std::min<uint64_t>(M - 1, (uint64_t)((double)M * 2.0 * (double)target / (double)G) + 4);
        for (uint32_t q = 0; q < ncol; q++) {
            unsigned long long *sq = sample.data() + (size_t)q * M;
            std::nth_element(sq, sq + j, sq + M);
            thr[q] = sq[j] >= ~0ull - 1 ? 0ull : sq[j];  // a threshold among the NULL / absent groups: page of nothing -> full ranking
        }
        const uint32_t cap = (uint32_t)std::min<uint64_t>(G, 8 * target + 16384);
        std::vector<int64_t> pg((size_t)cap * ncol);
        std::vector<unsigned long long> pk((size_t)cap * ncol);
        std::vector<uint32_t> cnt(ncol, 0);
        PVS_TRY(pvs_rrf_pages_cols(d_keys, ix->d_grp_ids, G, ncol, thr.data(), cap, pg.data(), pk.data(), cnt.data(), c.stream));
        struct E {
            unsigned long long key;
            int64_t gkey, g;
        };
        std::vector<E> e;
        const bool keyed = !ix->h_grp_key.empty();
        for (uint32_t q = 0; q < ncol; q++) {
            if (cnt[q] > cap || cnt[q] < k) continue;  // massive ties at the threshold / an unlucky sample: rank everything
            e.resize(cnt[q]);
            for (uint32_t i = 0; i < cnt[q]; i++) {
                int64_t gk = 0;
                const int64_t g = pg[(size_t)q * cap + i];
                if (keyed) (void)index_group_key(ix, g, &gk);
                e[i] = {pk[(size_t)q * cap + i], gk, g};
            }
            std::partial_sort(e.begin(), e.begin() + k, e.end(), [](const E &a, const E &b) {
                if (a.key != b.key) return a.key < b.key;
                if (a.gkey != b.gkey) return a.gkey > b.gkey;
                return a.g < b.g;
            });
            for (uint32_t i = 0; i < k; i++) {
                out_groups[(size_t)q * k + i] = e[i].g;
                out_values[(size_t)q * k + i] = pvs_group_value_of_key(e[i].key);
            }
            out_count[q] = k;
            done[q] = 1;
        }
        return PVS_OK;
    };
    pvs_status st = body();
    pvs_scratch_free_on(d_keys, c.stream);
    return st;
}

static pvs_status rank_values(pvs_index *ix, SearchCtx &c, const double *d_vals, const double *d_vals_t, uint32_t ncol, uint32_t k, int64_t *d_og, double *d_ov,
                              uint32_t *d_oc, int64_t *out_groups, double *out_values, uint32_t *out_count);
// shared tail: d_m [n][nb] (fanout == 0: nb output columns; else one) -> ranked groups on the host (declared in pvs_index.hpp:
// pvs_similar.hip aggregates its fan-out through it)
pvs_status aggregate_and_rank(pvs_index *ix, SearchCtx &c, const float *d_m, uint32_t nb, uint32_t fanout, int agg,
                              const float *d_weights, const uint8_t *d_exclude, uint32_t k, int64_t *out_groups,
                              double *out_values, uint32_t *out_count, FanoutWeights fw, uint32_t skip_when) {
    const uint32_t G = ix->n_groups, ncol = fanout ? 1u : nb;
    double *d_vals = nullptr;
    int64_t *d_og = nullptr;
    double *d_ov = nullptr;
    uint32_t *d_oc = nullptr;
    auto body = [&]() -> pvs_status {
        HIP_TRY(pvs_scratch_alloc((void **)&d_vals, (size_t)std::max<uint32_t>(G, 1) * ncol * 8));
        HIP_TRY(pvs_scratch_alloc((void **)&d_og, (size_t)k * 8));
        HIP_TRY(pvs_scratch_alloc((void **)&d_ov, (size_t)k * 8));
        HIP_TRY(pvs_scratch_alloc((void **)&d_oc, 4));
        HIP_TRY(pvs_launch_group_aggregate(d_m, nb, nb, fanout, ix->d_grp_off, ix->d_grp_rows, G, d_weights, d_exclude, agg, d_vals,
                                           c.stream, fw, skip_when, ix->groups_are_runs));
        return rank_values(ix, c, d_vals, nullptr, ncol, k, d_og, d_ov, d_oc, out_groups, out_values, out_count);
    };
    pvs_status st = body();
    for (void *p : {(void *)d_vals, (void *)d_og, (void *)d_ov, (void *)d_oc}) pvs_scratch_free_on(p, c.stream);
    return st;
}

// The device ranking writes its pages into the context's pinned block.  The host does not sleep on the stream for them: it poisons
// every word the kernels will write (values they never write) and reads the pages when no poisoned word is left — the wake-up through
// the runtime costs more than the pages' trip over PCIe (as pvs_search's one-launch route, pvs_search_host.hip; round 5).  A column
// whose flag comes back 0 goes to the full ranking; 20 ms without an answer, profiling and pvs_debug_set("no_flag_poll", 1) wait for the
// stream.  Whatever earlier kernels of the stream wrote to the pinned block (the scorers' out-of-range flag) is final once a later
// kernel's page has landed: a kernel's stores are performed before the next kernel of its stream starts.
namespace {
constexpr int64_t PAGE_POISON_G = INT64_MIN + 0x5EA1;
constexpr uint64_t PAGE_POISON_V = 0x7ff85ea15ea15ea1ull;  // (a NaN payload no aggregate produces)
constexpr uint32_t PAGE_POISON_F = 0xffffffffu;
bool pages_polled(const pvs_index *ix) { return !ix->profiling && !pvs_dbg(PVS_DBG_NO_FLAG_POLL); }
void pages_poison(uint8_t *io, size_t off_g, size_t off_v, size_t off_f, uint32_t ncol, uint32_t k) {
    volatile int64_t *g = (volatile int64_t *)(io + off_g);
    volatile uint64_t *v = (volatile uint64_t *)(io + off_v);
    volatile uint32_t *f = (volatile uint32_t *)(io + off_f);
    for (size_t i = 0; i < (size_t)ncol * k; i++) {
        g[i] = PAGE_POISON_G;
        v[i] = PAGE_POISON_V;
    }
    for (uint32_t q = 0; q < ncol; q++) f[q] = PAGE_POISON_F;
    std::atomic_thread_fence(std::memory_order_release);
}
// true: every column's flag is there and every page with flag 1 has landed
bool pages_wait(uint8_t *io, size_t off_g, size_t off_v, size_t off_f, uint32_t ncol, uint32_t k) {
    volatile int64_t *g = (volatile int64_t *)(io + off_g);
    volatile uint64_t *v = (volatile uint64_t *)(io + off_v);
    volatile uint32_t *f = (volatile uint32_t *)(io + off_f);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spin = 0;; spin++) {
        bool all = true;
        for (uint32_t q = 0; q < ncol && all; q++) {
            const uint32_t fl = f[q];
            all = fl != PAGE_POISON_F;
            if (all && fl)
                for (size_t i = (size_t)q * k; i < (size_t)(q + 1) * k && all; i++) all = g[i] != PAGE_POISON_G && v[i] != PAGE_POISON_V;
        }
        if (all) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return true;
        }
        __builtin_ia32_pause();
        if ((spin & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) return false;
    }
}
}  // namespace

// Columns of group values, column-major d_vals [ncol][G] or group-major d_vals_t [G][ncol] (the fused scorer's layout; exactly
// one of the two is given) -> each column's first k groups on the host.
static pvs_status rank_values(pvs_index *ix, SearchCtx &c, const double *d_vals, const double *d_vals_t, uint32_t ncol, uint32_t k, int64_t *d_og, double *d_ov,
                              uint32_t *d_oc, int64_t *out_groups, double *out_values, uint32_t *out_count) {
    const uint32_t G = ix->n_groups;
    const bool no_page_rank = pvs_dbg(PVS_DBG_NO_PAGE_RANK) != 0;  // tuning: always sort every group
    double *d_cm = nullptr;
    void *d_work = nullptr;
    auto body = [&]() -> pvs_status {
        // Page first: the k best of millions of groups do not need all of them sorted (a stable 64-bit radix sort of 1.3M groups
        // is 0.3 ms per query column, ten times the aggregation).  A threshold key from a sample admits a few thousand groups per
        // column; where at least k come back, the k best are among them: sorted on the host under the page order (value, second
        // key DESC, group id).  Columns where that fails (an unlucky sample, fewer than k groups with a value) are fully sorted.
        std::vector<uint8_t> paged(ncol, 0);
        if (d_vals_t && !no_page_rank && pvs_gm_rank_supported(G, ncol, k)) {
            // group-major values (the fused scorer): thresholds, pages and the per-column sort all on the device, the pages land
            // in the context's pinned block — one synchronisation, no copies (round 3's page step: three host round trips, 64
            // staged copies and a host sort per column = 2 of the 3.5 ms of a 32-query per-item search)
            const size_t off_g = 64, off_v = off_g + (size_t)ncol * k * 8, off_f = off_v + (size_t)ncol * k * 8, need = off_f + (size_t)ncol * 4;
            PVS_TRY(ctx_pinned_io(c, need));
            HIP_TRY(pvs_scratch_alloc(&d_work, pvs_gm_rank_work_bytes(ncol)));
            const bool keyed = ix->d_grp_tinv && ix->d_grp_trank;
            const bool poll = pages_polled(ix);
            if (poll) pages_poison(c.h_io, off_g, off_v, off_f, ncol, k);
            HIP_TRY(pvs_gm_rank(d_vals_t, G, ncol, k, ix->d_grp_ids, keyed ? ix->d_grp_trank : nullptr, keyed ? ix->d_grp_tinv : nullptr, d_work,
                                (int64_t *)(c.h_io + off_g), (double *)(c.h_io + off_v), (uint32_t *)(c.h_io + off_f), c.stream));
            if (!poll || !pages_wait(c.h_io, off_g, off_v, off_f, ncol, k)) HIP_TRY(hipStreamSynchronize(c.stream));
            const uint32_t *fl = (const uint32_t *)(c.h_io + off_f);
            for (uint32_t q = 0; q < ncol; q++)
                if (fl[q]) {
                    memcpy(out_groups + (size_t)q * k, c.h_io + off_g + (size_t)q * k * 8, (size_t)k * 8);
                    memcpy(out_values + (size_t)q * k, c.h_io + off_v + (size_t)q * k * 8, (size_t)k * 8);
                    out_count[q] = k;
                    paged[q] = 1;
                }
        } else if (d_vals && !no_page_rank && ncol <= 256 && pvs_gm_rank_supported(G, ncol, k)) {
            // column-major values (the dense-matrix route: float indexes, similar_to's fan-out): the same device page ranking with the
            // strides swapped — three launches for all columns (until round 5: four per column, 62 us each, 2 of the 10 ms of a
            // 32-query per-item search over 4M float rows)
            const size_t off_g = 64, off_v = off_g + (size_t)ncol * k * 8, off_f = off_v + (size_t)ncol * k * 8, need = off_f + (size_t)ncol * 4;
            PVS_TRY(ctx_pinned_io(c, need));
            HIP_TRY(pvs_scratch_alloc(&d_work, pvs_gm_rank_work_bytes(ncol)));
            const bool keyed = ix->d_grp_tinv && ix->d_grp_trank;
            const bool poll = pages_polled(ix);
            if (poll) pages_poison(c.h_io, off_g, off_v, off_f, ncol, k);
            HIP_TRY(pvs_gm_rank(d_vals, G, ncol, k, ix->d_grp_ids, keyed ? ix->d_grp_trank : nullptr, keyed ? ix->d_grp_tinv : nullptr, d_work,
                                (int64_t *)(c.h_io + off_g), (double *)(c.h_io + off_v), (uint32_t *)(c.h_io + off_f), c.stream, true));
            if (!poll || !pages_wait(c.h_io, off_g, off_v, off_f, ncol, k)) HIP_TRY(hipStreamSynchronize(c.stream));
            const uint32_t *fl = (const uint32_t *)(c.h_io + off_f);
            for (uint32_t q = 0; q < ncol; q++)
                if (fl[q]) {
                    memcpy(out_groups + (size_t)q * k, c.h_io + off_g + (size_t)q * k * 8, (size_t)k * 8);
                    memcpy(out_values + (size_t)q * k, c.h_io + off_v + (size_t)q * k * 8, (size_t)k * 8);
                    out_count[q] = k;
                    paged[q] = 1;
                }
        } else if (G >= 65536 && (uint64_t)G * ncol >= (2u << 20) && k <= 4096 && !no_page_rank) {
            // (three host round trips, ~0.35 ms whatever the size: worth it from ~2M group values to sort — one column of 230k groups,
            //  similar_to at the reference's scale, sorts in 0.1 ms)
            PVS_TRY(rank_groups_page_first(ix, c, d_vals, d_vals_t, G, ncol, k, out_groups, out_values, out_count, paged));
        }
        for (uint32_t q = 0; q < ncol; q++) {
            if (paged[q]) continue;
            if (!d_vals) {  // a column is sorted whole: it wants its values contiguous
                HIP_TRY(pvs_scratch_alloc((void **)&d_cm, (size_t)std::max<uint32_t>(G, 1) * ncol * 8));
                HIP_TRY(pvs_launch_group_transpose(d_vals_t, G, ncol, d_cm, c.stream));
                d_vals = d_cm;
            }
            PVS_TRY(pvs_group_rank(d_vals + (size_t)q * G, ix->d_grp_ids, G, k, c.gwork, d_og, d_ov, d_oc, c.stream, ix->d_grp_tinv));
            HIP_TRY(hipMemcpyAsync(out_groups + (size_t)q * k, d_og, (size_t)k * 8, hipMemcpyDeviceToHost, c.stream));
            HIP_TRY(hipMemcpyAsync(out_values + (size_t)q * k, d_ov, (size_t)k * 8, hipMemcpyDeviceToHost, c.stream));
            HIP_TRY(hipMemcpyAsync(out_count + q, d_oc, 4, hipMemcpyDeviceToHost, c.stream));
            HIP_TRY(hipStreamSynchronize(c.stream));
        }
        return PVS_OK;
    };
    pvs_status st = body();
    pvs_scratch_free_on(d_cm, c.stream);
    pvs_scratch_free_on(d_work, c.stream);
    return st;
}

// MIN aggregation (the reference's default, filters/embedding_types.rs:14-18) without scoring every row
// into a dense matrix: a group's MIN is the distance of its best row, so the top-k groups are the
// groups of the first rows of the row ranking.  Take a row page of kp rows through the filter scan,
// keep each group's first occurrence, and accept iff the page provably contains the answer: it is the
// whole corpus, or the k-th group's value is strictly below the last row's distance (rows tied with
// the boundary could otherwise hide an unseen group).  Else grow kp; past PVS_MAX_K the caller runs
// the dense path.  Values are the same f64(f32 distance) the dense path produces.
static pvs_status groups_min_fast(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                                  const uint8_t *mask, pvs_space mask_space, int64_t *out_groups, double *out_values,
                                  uint32_t *out_count, bool *done) {
    *done = false;
    if (ix->n == 0 || ix->n_groups == 0 || ix->forced_path == 1) return PVS_OK;
    const uint64_t n = ix->n;
    const double per_group = (double)n / (double)ix->n_groups;
    uint64_t kp = std::max<uint64_t>(64, (uint64_t)(2.0 * k * std::min(std::ceil(per_group), 8.0)));
    kp = std::min<uint64_t>({kp, (uint64_t)PVS_MAX_K, n});
    if (kp < std::min<uint64_t>(k, n) || !fast_path_ok(ix, (uint32_t)kp)) return PVS_OK;
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        PVS_TRY(pvs_host_ids_locked(ix));
    }
    std::vector<int64_t> ids;
    std::vector<float> dist;
    std::vector<uint32_t> cnt(batch), rowidx;
    struct GV {
        double v;
        int64_t g;
        int64_t key;  // second sort key of the group (0 when none is set)
    };
    const bool keyed = !ix->h_grp_key.empty();
    auto group_key = [&](int64_t g) -> int64_t {
        if (!keyed) return 0;
        auto it = std::lower_bound(ix->h_grp_ids.begin(), ix->h_grp_ids.end(), g);
        return ix->h_grp_key[(size_t)(it - ix->h_grp_ids.begin())];
    };
    std::vector<GV> gv;
    std::vector<int64_t> seen;
    for (;;) {
        ids.assign((size_t)batch * kp, -1);
        dist.assign((size_t)batch * kp, 0.f);
        rowidx.resize((size_t)batch * kp);
        PVS_TRY(search_host(ix, queries, qdtype, batch, (uint32_t)kp, metric, mask, mask_space, ids.data(), dist.data(), cnt.data(), nullptr, 0, PVS_HOST,
                            rowidx.data()));
        bool all_ok = true;
        for (uint32_t q = 0; q < batch && all_ok; q++) {
            const int64_t *qi = ids.data() + (size_t)q * kp;
            const float *qd = dist.data() + (size_t)q * kp;
            gv.clear();
            seen.clear();
            for (uint32_t i = 0; i < cnt[q]; i++) {
                int64_t g = qi[i];  // identity groups: the group id is the row id
                if (!ix->h_groups.empty()) {
                    const uint32_t r = rowidx[(size_t)q * kp + i];  // (the one-launch search hands the stored row over: no search among the ids)
                    if (r < n) {
                        g = ix->h_groups[r];
                    } else {
                        const auto it = std::lower_bound(ix->h_ids_cache.begin(), ix->h_ids_cache.end(), qi[i]);
                        g = ix->h_groups[(size_t)(it - ix->h_ids_cache.begin())];
                    }
                }
                seen.push_back(g);
            }
            // first occurrence of each group, in page order
            std::vector<uint32_t> order(seen.size());
            for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return seen[a] < seen[b]; });
            for (size_t i = 0; i < order.size(); i++)
                if (i == 0 || seen[order[i]] != seen[order[i - 1]]) gv.push_back({(double)qd[order[i]], seen[order[i]], group_key(seen[order[i]])});
            std::sort(gv.begin(), gv.end(), [](const GV &a, const GV &b) {
                const bool na = a.v != a.v, nb = b.v != b.v;  // NULL last, then value, then the second key DESC, then group id
                if (na != nb) return nb;
                if (!na && a.v != b.v) return a.v < b.v;
                if (a.key != b.key) return a.key > b.key;
                return a.g < b.g;
            });
            const bool complete = cnt[q] == n || cnt[q] < kp;  // the page is the whole corpus / every candidate row
            const uint32_t want = (uint32_t)std::min<uint64_t>(k, complete ? gv.size() : (uint64_t)k);
            bool ok = complete;
            if (!ok && gv.size() >= k && cnt[q] > 0) {
                const double last = (double)qd[cnt[q] - 1];
                ok = gv[k - 1].v < last;  // false for NaN on either side
            }
            if (!ok) {
                all_ok = false;
                break;
            }
            for (uint32_t i = 0; i < k; i++) {
                out_groups[(size_t)q * k + i] = i < want ? gv[i].g : -1;
                out_values[(size_t)q * k + i] = i < want ? gv[i].v : __builtin_nan("");
            }
            out_count[q] = want;
        }
        if (all_ok) {
            *done = true;
            return PVS_OK;
        }
        if (kp >= std::min<uint64_t>(PVS_MAX_K, n)) return PVS_OK;  // give up: dense path
        kp = std::min<uint64_t>({kp * 4, (uint64_t)PVS_MAX_K, n});
    }
}

PVS_EXPORT pvs_status pvs_search_groups(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k,
                                        pvs_metric metric, pvs_agg agg, const float *row_weights, int64_t *out_groups,
                                        double *out_values, uint32_t *out_count) {
    GateShared gate(ix);  // (pvs_gate.hip: a mutation waits for this call, a search never sees one half done)
    if (!row_weights && coalescing_applies(ix, batch)) {  // (pvs_index_set_coalescing: concurrent callers share one pass)
        PVS_TRY(validate_search(ix, queries, qdtype, batch, k, metric));
        if (!out_groups || !out_values || !out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null output");
        if (agg != PVS_AGG_MIN && agg != PVS_AGG_MAX && agg != PVS_AGG_AVG) return pvs_fail(PVS_ERR_INVALID_ARG, "aggregation must be MIN, MAX or AVG");
        return coalesce_call(ix, 1, (int)agg, queries, qdtype, batch, k, metric, out_groups, out_values, out_count);
    }
    return search_groups_impl(ix, queries, qdtype, batch, k, metric, agg, row_weights, nullptr, PVS_HOST, out_groups, out_values, out_count);
}

PVS_EXPORT pvs_status pvs_search_groups_filtered(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k,
                                                 pvs_metric metric, pvs_agg agg, const float *row_weights, const uint8_t *allowed_rows,
                                                 pvs_space mask_space, int64_t *out_groups, double *out_values, uint32_t *out_count) {
    GateShared gate(ix);  // (pvs_gate.hip: a mutation waits for this call, a search never sees one half done)
    if (!allowed_rows) return pvs_fail(PVS_ERR_INVALID_ARG, "null candidate mask");
    return search_groups_impl(ix, queries, qdtype, batch, k, metric, agg, row_weights, allowed_rows, mask_space, out_groups, out_values,
                              out_count);
}

// Can the fused per-item scorer serve this chunk?  int8 rows inside the closed form's range on a row pitch k_scan has an instance
// for, groups that are runs of rows, per-row arrays the scalar loads of the fold can read in aligned 32-row pieces.
static bool fused_groups_ok(const pvs_index *ix, uint32_t nb, const uint8_t *d_mask, const float *d_w) {
    if (pvs_dbg(PVS_DBG_NO_FUSED_AGG) || !ix->groups_are_runs || !ix->d_tile_grp) return false;
    if (ix->dtype != PVS_I8 || !pvs_scan_supported(PVS_I8, ix->stride / PVS_KSLAB_BYTES) || (uint64_t)ix->dim * 127 * 127 >= (1u << 24)) return false;
    if (((uintptr_t)d_mask & 3u) || ((uintptr_t)d_w & 3u)) return false;
    (void)nb;
    return true;
}
// One corpus pass over the queries prepared in c, enqueued: d_vals_t [n_groups][nb] = every group's aggregate.  An L2 sum that left
// the closed form's range raises word 8 of c.h_io (the caller scores in order instead).  d_m: the [n][nb] matrix the rows of
// tile-crossing groups go through.
static pvs_status fused_group_chunk(pvs_index *ix, SearchCtx &c, uint32_t nb, uint32_t batch_pad, int metric, int agg, const float *d_w, const uint8_t *d_mask,
                                    float *d_m, double *d_vals_t) {
    if (nb <= 4 && !pvs_dbg(PVS_DBG_NO_DIRECT_SCORE)) {
        // one to four queries: the v_dot4 stream with the fold in its tile epilogue (pvs_score_direct.hip) — the matrix-core scorer
        // pads them to 32 and its fold walks a tile's rows serially in every query lane (690k x 768, one query: 125 us against ~90)
        uint32_t *flag = (uint32_t *)(c.h_io + 32);
        *(volatile uint32_t *)flag = 0;
        span_begin(ix, c, 1, ix->n);
        HIP_TRY(pvs_launch_score_i8_fold(metric, ix->d_rows, ix->stride, ix->dim, ix->n, ix->d_norm2, c.d_qexact, c.d_qinfo, nb, d_m, nb, flag, ix->d_tile_grp, d_w,
                                         d_mask, d_vals_t, nb, agg, (uint32_t)ix->n_cu, c.stream));
        span_end(ix, c);
        HIP_TRY(pvs_launch_group_aggregate_list(d_m, nb, nb, ix->d_grp_off, ix->d_grp_rows, ix->d_straddlers, ix->n_straddlers, d_w, d_mask, agg, d_vals_t, nb,
                                                c.stream, d_mask ? 0u : 1u));
        return PVS_OK;
    }
    ScanArgs a;
    a.dtype = PVS_I8;
    a.metric = metric;
    a.kslabs = ix->stride / PVS_KSLAB_BYTES;
    a.qgroups = batch_pad / 32;
    a.rows = ix->d_rows;
    a.aux = ix->d_scan_l2;  // MODE 2 streams |a|^2
    a.stride = ix->stride;
    a.n_rows = ix->n;
    a.qmat = c.d_qmat;
    a.qinfo = c.d_qinfo;
    a.thr = c.d_thr;
    a.gmin = c.d_gmin;
    a.groups_per_query = 0;
    a.mode = 3;
    a.tile_step = 1;
    const uint32_t wg_rows = 32u * pvs_scan_row_tiles(a.qgroups);
    const uint32_t n_wgtiles = (uint32_t)((ix->n + wg_rows - 1) / wg_rows);
    const uint32_t per_cu = a.kslabs > 3 ? 1 : 2;  // (MODE 3: two workgroups per CU up to 768-B rows, also at 32 queries: scan_fold2)
    a.grid = std::min<uint32_t>(n_wgtiles, (uint32_t)ix->n_cu * per_cu);
    a.dense_out = d_m;
    a.dense_ld = nb;
    a.batch = nb;
    // the out-of-range flag lives in the context's pinned block (word 8): the caller reads it after its own synchronisation, no
    // copy and no extra round trip (the kernel touches it only when an L2 sum leaves the closed form's range)
    uint32_t *flag = (uint32_t *)(c.h_io + 32);
    *(volatile uint32_t *)flag = 0;
    a.dense_flag = flag;
    a.tile_grp = ix->d_tile_grp;
    a.fold_weights = d_w;
    a.fold_mask = d_mask;
    a.fold_out = d_vals_t;
    a.fold_ld = nb;
    a.fold_agg = agg;
    span_begin(ix, c, 1, ix->n);
    HIP_TRY(pvs_launch_scan(a, c.stream));
    span_end(ix, c);
    HIP_TRY(pvs_launch_group_aggregate_list(d_m, nb, nb, ix->d_grp_off, ix->d_grp_rows, ix->d_straddlers, ix->n_straddlers, d_w, d_mask, agg, d_vals_t, nb,
                                            c.stream, d_mask ? 0u : 1u));
    return PVS_OK;
}

pvs_status search_groups_impl(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                              pvs_agg agg, const float *row_weights, const uint8_t *mask, pvs_space mask_space, int64_t *out_groups,
                              double *out_values, uint32_t *out_count) {
    if (ix && is_multi(ix)) {
        if (!row_weights && agg != PVS_AGG_MIN && agg != PVS_AGG_MAX && agg != PVS_AGG_AVG)
            return pvs_fail(PVS_ERR_INVALID_ARG, "aggregation must be MIN, MAX or AVG");
        return multi_search_groups(ix, queries, qdtype, batch, k, metric, agg, row_weights, mask, mask_space, out_groups, out_values, out_count);
    }
    PVS_TRY(validate_search(ix, queries, qdtype, batch, k, metric));
    if (!out_groups || !out_values || !out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null output");
    if (!row_weights && agg != PVS_AGG_MIN && agg != PVS_AGG_MAX && agg != PVS_AGG_AVG)
        return pvs_fail(PVS_ERR_INVALID_ARG, "aggregation must be MIN, MAX or AVG");
    if (batch == 0) return PVS_OK;
    HIP_TRY(hipSetDevice(ix->device));
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        PVS_TRY(ensure_groups(ix));
    }
    // MIN (the reference's default): from a handful of queries up the one-pass scorer with the per-group fold is cheaper than row
    // pages of the filter scan regrouped on the host (4M x 768 int8, 32 queries: 0.95 ms against 3.1); one to three queries keep
    // the filter scan, whose pass streams at the full HBM rate for them
    const bool min_fused = batch >= 4 && fused_groups_ok(ix, batch, nullptr, nullptr) && pvs_dbg(PVS_DBG_NO_FUSED_AGG) == 0;
    if (agg == PVS_AGG_MIN && !row_weights && !min_fused) {
        bool done = false;
        PVS_TRY(groups_min_fast(ix, queries, qdtype, batch, k, metric, mask, mask_space, out_groups, out_values, out_count, &done));
        if (done) return PVS_OK;
    }
    uint32_t t;
    SearchCtx *c = ctx_acquire(ix, &t);
    void *d_q = nullptr;
    float *d_m = nullptr, *d_w = nullptr;
    uint8_t *d_mask = nullptr;
    double *d_vt = nullptr;
    uint32_t dense_q = 0;
    auto body = [&]() -> pvs_status {
        PVS_TRY(ctx_prepare(ix, *c, batch, k, false));
        const size_t qbytes = (size_t)ix->dim * (qdtype == PVS_I8 ? 1 : 4);
        // the context's pinned block: [0, 64) flag words of the kernels, then the pages of the device-side ranking (rank_values),
        // then the queries — read by the query-prep kernel straight from host memory (no staged copy).  Sized ONCE here: kernels
        // in flight hold pointers into it.
        // (per query column: k groups + k values, a handled flag AND a count — pvs_sparse_search_groups lays its pages out as
        //  [64 | groups | values | flags | counts]; with 4 bytes per column its counts ran into the queries behind the pages)
        const size_t io_pages = pvs_group_pages_bytes(std::min<uint32_t>(batch, PVS_MAX_BATCH), k);
        const size_t io_q = pvs_round_up(io_pages, 256);
        const void *q_dev = nullptr;
        if (qbytes * batch <= ((size_t)4 << 20)) {
            PVS_TRY(ctx_pinned_io(*c, io_q + qbytes * batch));
            memcpy(c->h_io + io_q, queries, qbytes * batch);
            q_dev = c->h_io + io_q;
        } else {
            PVS_TRY(ctx_pinned_io(*c, io_pages));
            HIP_TRY(pvs_scratch_alloc(&d_q, qbytes * batch));
            HIP_TRY(hipMemcpyAsync(d_q, queries, qbytes * batch, hipMemcpyHostToDevice, c->stream));
            q_dev = d_q;
        }
        const uint8_t *dm = nullptr;  // candidate mask on the device
        if (mask && ix->n) {
            if (mask_space == PVS_HOST) {
                HIP_TRY(pvs_scratch_alloc((void **)&d_mask, ix->n));
                HIP_TRY(hipMemcpyAsync(d_mask, mask, ix->n, hipMemcpyHostToDevice, c->stream));
                dm = d_mask;
            } else if ((ix->n & 31u) || ((uintptr_t)mask & 3u)) {
                // (the fused scorer reads the mask in aligned 32-byte pieces through the scalar cache: a caller's device buffer of
                //  exactly n bytes is copied into a block with slack behind it)
                HIP_TRY(pvs_scratch_alloc((void **)&d_mask, ix->n));
                HIP_TRY(hipMemcpyAsync(d_mask, mask, ix->n, hipMemcpyDeviceToDevice, c->stream));
                dm = d_mask;
            } else {
                dm = mask;
            }
        }
        if (row_weights && ix->n) {
            HIP_TRY(pvs_scratch_alloc((void **)&d_w, ix->n * 4));
            HIP_TRY(hipMemcpyAsync(d_w, row_weights, ix->n * 4, hipMemcpyHostToDevice, c->stream));
        }
        // A candidate mask that leaves few rows: score and aggregate those rows only (pvs_sparse.hip), like the row form
        if (dm && ix->n && ix->forced_path == 0 && !pvs_dbg(PVS_DBG_NO_SPARSE)) {
            uint32_t allowed = 0;
            PVS_TRY(pvs_mask_count(dm, ix->n, &allowed, c->stream));
            if (pvs_sparse_eligible(ix, allowed, batch, k)) {
                uint32_t *d_list = nullptr;
                HIP_TRY(pvs_scratch_alloc((void **)&d_list, (size_t)std::max<uint32_t>(allowed, 1) * 4));
                pvs_status ss = pvs_mask_compact(dm, ix->n, d_list, allowed, c->stream);
                bool handled = false;
                if (ss == PVS_OK)
                    ss = pvs_sparse_search_groups(ix, *c, q_dev, qdtype, batch, k, metric, agg, d_w, d_list, allowed, out_groups, out_values, out_count, &handled);
                pvs_scratch_free_on(d_list, c->stream);
                PVS_TRY(ss);
                if (handled) return PVS_OK;
            }
        }
        const uint32_t cq = dense_chunk_queries(ix, batch);
        HIP_TRY(pvs_scratch_alloc((void **)&d_m, std::max<size_t>((size_t)ix->n * cq * 4, 16)));
        for (uint32_t q0 = 0; q0 < batch; q0 += cq) {
            const uint32_t nb = std::min(cq, batch - q0);
            const uint32_t pad = nb <= 32 ? 32 : nb <= 64 ? 64 : 128;
            if (ix->n) PVS_TRY(prep_chunk(ix, *c, q_dev, qdtype, q0, nb, pad, metric));
            // One pass: the scorer folds every group that lies inside a 32-row tile in its epilogue (k_scan MODE 2 + ScanK.tile_grp)
            // and writes [groups][queries] values; only the rows of groups that cross a tile boundary go through the matrix.
            bool fused_done = false;
            if (ix->n && fused_groups_ok(ix, nb, dm, d_w)) {
                HIP_TRY(pvs_scratch_alloc((void **)&d_vt, (size_t)std::max<uint32_t>(ix->n_groups, 1) * nb * 8));
                PVS_TRY(fused_group_chunk(ix, *c, nb, pad, metric, agg, d_w, dm, d_m, d_vt));
                {
                    int64_t *d_og = nullptr;
                    double *d_ov = nullptr;
                    uint32_t *d_oc = nullptr;
                    HIP_TRY(pvs_scratch_alloc((void **)&d_og, (size_t)k * 8));
                    HIP_TRY(pvs_scratch_alloc((void **)&d_ov, (size_t)k * 8));
                    HIP_TRY(pvs_scratch_alloc((void **)&d_oc, 4));
                    pvs_status rs = rank_values(ix, *c, nullptr, d_vt, nb, k, d_og, d_ov, d_oc, out_groups + (size_t)q0 * k, out_values + (size_t)q0 * k, out_count + q0);
                    for (void *p : {(void *)d_og, (void *)d_ov, (void *)d_oc}) pvs_scratch_free_on(p, c->stream);
                    PVS_TRY(rs);
                    // (rank_values waited for the stream: the scorer's out-of-range flag — a word of the pinned block — is final)
                    fused_done = *(volatile uint32_t *)(c->h_io + 32) == 0;
                    spans_collect(ix, *c);
                }
                pvs_scratch_free_on(d_vt, c->stream);
                d_vt = nullptr;
            }
            if (fused_done) continue;
            // float rows: bracket every file's aggregate from the matrix-core scan keys, rescan exactly only the files that can reach
            // the page (pvs_items_float.hip); whatever it cannot certify falls through to the exact-everywhere route below
            if (ix->n && pvs_float_certify_applies(ix, nb, k)) {
                bool certified = false;
                std::vector<uint8_t> redo;
                PVS_TRY(pvs_float_groups_certified(ix, *c, q_dev, qdtype, q0, nb, pad, k, metric, agg, d_w, dm, out_groups + (size_t)q0 * k,
                                                   out_values + (size_t)q0 * k, out_count + q0, &certified, &redo));
                if (certified) {
                    // queries nothing can be bracketed for (every distance NULL: a zero query under cosine, a NaN component): one by one
                    // through the exact-everywhere route
                    for (uint32_t j = 0; j < nb; j++) {
                        if (!redo[j]) continue;
                        PVS_TRY(prep_chunk(ix, *c, q_dev, qdtype, q0 + j, 1, 32, metric));
                        PVS_TRY(dense_chunk(ix, *c, 1, 32, metric, d_m));
                        PVS_TRY(aggregate_and_rank(ix, *c, d_m, 1, 0, agg, d_w, dm, k, out_groups + (size_t)(q0 + j) * k, out_values + (size_t)(q0 + j) * k,
                                                   out_count + q0 + j, FanoutWeights(), dm ? 0u : 1u));
                        dense_q++;
                    }
                    continue;
                }
                PVS_TRY(prep_chunk(ix, *c, q_dev, qdtype, q0, nb, pad, metric));  // (the rescan stage prepared the context for its own chunks)
            }
            dense_q += nb;
            if (ix->n) PVS_TRY(dense_chunk(ix, *c, nb, pad, metric, d_m));
            PVS_TRY(aggregate_and_rank(ix, *c, d_m, nb, 0, agg, d_w, dm, k, out_groups + (size_t)q0 * k, out_values + (size_t)q0 * k,
                                       out_count + q0, FanoutWeights(), dm ? 0u : 1u));
        }
        return PVS_OK;
    };
    pvs_status st = body();
    for (void *p : {d_q, (void *)d_m, (void *)d_w, (void *)d_mask, (void *)d_vt}) pvs_scratch_free_on(p, c->stream);
    ix->searches++;
    ix->dense_queries += dense_q;  // (queries answered through the N x B matrix; the one-pass scorer writes none)
    ctx_done(ix, c);
    return st;
}

PVS_EXPORT pvs_status pvs_search_groups_sharded(pvs_index *ix, pvs_comm *comm, const void *queries, pvs_dtype qdtype, uint32_t batch,
                                                uint32_t k, pvs_metric metric, pvs_agg agg, const float *row_weights, int64_t *out_groups,
                                                double *out_values, uint32_t *out_count) {
    GateShared gate(ix);  // (pvs_gate.hip: a mutation waits for this call, a search never sees one half done)
    if (!comm) return pvs_fail(PVS_ERR_INVALID_ARG, "null communicator");
    if (ix && is_multi(ix)) return pvs_fail(PVS_ERR_UNSUPPORTED, "a multi-device index shards inside one process: use pvs_search_groups");
    if (ix && pvs_comm_device_(comm) != ix->device) return pvs_fail(PVS_ERR_INVALID_ARG, "index and communicator live on different devices");
    if (!ix) return pvs_fail(PVS_ERR_INVALID_ARG, "null index");
    if (batch == 0) return PVS_OK;
    // 1. this shard's page.  Every rank must take part in the exchange below whatever happened locally: a rank
    // that returned early would leave the others inside the all-gather.  A local failure travels as a count of
    // GROUP_PAGE_FAILED and every rank fails after the exchange.
    constexpr uint32_t GROUP_PAGE_FAILED = 0xffffffffu;
    std::vector<int64_t> lg((size_t)batch * k, -1);
    std::vector<double> lv((size_t)batch * k, __builtin_nan(""));
    std::vector<uint32_t> lc(batch, 0);
    const pvs_status local_st = pvs_search_groups(ix, queries, qdtype, batch, k, metric, agg, row_weights, lg.data(), lv.data(), lc.data());
    const std::string local_err = local_st == PVS_OK ? std::string() : std::string(pvs_last_error());
    if (local_st != PVS_OK) std::fill(lc.begin(), lc.end(), GROUP_PAGE_FAILED);
    HIP_TRY(hipSetDevice(ix->device));
    const uint32_t world = (uint32_t)pvs_comm_world_(comm);
    const uint64_t elems = (uint64_t)batch * k;
    // this rank's page as ONE record [groups i64 | values f64 | order keys i64 | counts u32 | keyed u32] (16-byte padded): the
    // exchange is a single all-gather.  keyed = 1: the index carries order keys (pvs_index_set_order_keys) and the keys section
    // holds the key of every entry's group; the merge uses them when every rank says so.
    const size_t off_v = elems * 8, off_k = elems * 16, off_c = elems * 24, off_f = off_c + (size_t)batch * 4,
                 rec = (off_f + 4 + 15) / 16 * 16;
    uint8_t *d_rec = nullptr, *d_all = nullptr;
    std::vector<uint8_t> h_rec(rec, 0), h_all(rec * world);
    std::vector<int64_t> lk(elems, 0);
    uint32_t keyed = local_st == PVS_OK && ix->order_rows == ix->n && ix->n ? 1u : 0u;
    if (keyed)
        for (uint32_t q = 0; q < batch; q++)
            for (uint32_t i = 0; i < lc[q] && i < k; i++) (void)index_group_key(ix, lg[(size_t)q * k + i], &lk[(size_t)q * k + i]);
    if (local_st == PVS_OK && ix->n == 0) keyed = 2;  // an empty shard has no say
    memcpy(h_rec.data(), lg.data(), elems * 8);
    memcpy(h_rec.data() + off_v, lv.data(), elems * 8);
    memcpy(h_rec.data() + off_k, lk.data(), elems * 8);
    memcpy(h_rec.data() + off_c, lc.data(), (size_t)batch * 4);
    memcpy(h_rec.data() + off_f, &keyed, 4);
    std::vector<int64_t> ak((size_t)world * elems);
    std::vector<int64_t> ag((size_t)world * elems);
    std::vector<double> av((size_t)world * elems);
    std::vector<uint32_t> ac((size_t)world * batch);
    auto body = [&]() -> pvs_status {
        HIP_TRY(pvs_scratch_alloc((void **)&d_rec, rec));
        HIP_TRY(pvs_scratch_alloc((void **)&d_all, rec * world));
        hipStream_t s = ix->comm_stream;  // every collective of this index goes out on this one stream
        HIP_TRY(hipMemcpyAsync(d_rec, h_rec.data(), rec, hipMemcpyHostToDevice, s));
        // 2. one all-gather over xGMI
        PVS_TRY(pvs_comm_gather_records_(comm, d_rec, d_all, rec, s));
        HIP_TRY(hipMemcpyAsync(h_all.data(), d_all, rec * world, hipMemcpyDeviceToHost, s));
        PVS_TRY(pvs_comm_wait_stream_(comm, s, "per-item shard exchange (all-gather)"));  // bounded: a rank that never arrives is an error, not a hang
        for (uint32_t w = 0; w < world; w++) {
            memcpy(ag.data() + (size_t)w * elems, h_all.data() + (size_t)w * rec, elems * 8);
            memcpy(av.data() + (size_t)w * elems, h_all.data() + (size_t)w * rec + off_v, elems * 8);
            memcpy(ak.data() + (size_t)w * elems, h_all.data() + (size_t)w * rec + off_k, elems * 8);
            memcpy(ac.data() + (size_t)w * batch, h_all.data() + (size_t)w * rec + off_c, (size_t)batch * 4);
        }
        bool all_keyed = true, any_keyed = false;
        for (uint32_t w = 0; w < world; w++) {
            uint32_t f;
            memcpy(&f, h_all.data() + (size_t)w * rec + off_f, 4);
            all_keyed &= f != 0;
            any_keyed |= f == 1;
        }
        if (local_st != PVS_OK) return pvs_fail(local_st, "%s", local_err.c_str());
        for (uint32_t w = 0; w < world; w++)
            if (ac[(size_t)w * batch] == GROUP_PAGE_FAILED) return pvs_fail(PVS_ERR_COMM, "rank %u failed its shard of the per-item search", w);
        // 3. merge on every rank (tiny: world * k entries per query)
        return pvs_merge_group_pages_keyed(ag.data(), av.data(), all_keyed && any_keyed ? ak.data() : nullptr, ac.data(), world, batch, k, out_groups,
                                           out_values, out_count);
    };
    pvs_status st = body();
    for (void *p : {(void *)d_rec, (void *)d_all}) pvs_scratch_free_on(p, ix->comm_stream);  // (an early error may have left work queued)
    return st;
}
