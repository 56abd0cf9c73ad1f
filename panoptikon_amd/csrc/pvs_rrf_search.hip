// pvs_rrf_search.hip — C ABI of libpvs: the PQL `or` arm over vector filters (pql/builder.rs:638-661, 757-771, 1284-1317) — per-branch
// scoring and per-file aggregates, window keys, the bounded fusion (first round on the device: pvs_rrf_device.hip), the full ranking,
// the column handles of the sharded form (pvs_rrf_cols_*), the stage digests.  Split out of pvs_items.hip in round 5.
#include <chrono>
#include <string>
#include <thread>

#include "pvs_index.hpp"

static thread_local int32_t g_rrf_last_path = 0;
PVS_EXPORT int32_t pvs_rrf_last_path(void) { return g_rrf_last_path; }

// Race hunting (pvs_debug_set("rrf_digest", 1); tools/rrf_stage_digest.py): 64-bit digests of the stages of the last single-device
// pvs_rrf_search of the process, per branch: [0] the `d` column (every row's f32 distance), [1] the per-group aggregates (f64),
// [2] the window keys, [3] the ranks — bounded fusion: (candidate group, exact counted rank) over the candidates of the last round;
// full ranking: every group's rank in slot order.  A digest that moves between two runs of the same query names the stage.
namespace {
struct RrfDigestRec {
    uint64_t v[PVS_RRF_MAX_BRANCHES][4];
    uint32_t nb;
    int32_t path;
};
std::mutex g_rrf_dig_mu;
RrfDigestRec g_rrf_dig;
inline uint64_t host_mix(uint64_t a, uint64_t b) {
    uint64_t z = (a + 1) * 0x9E3779B97F4A7C15ull ^ b * 0xD6E8FEB86659FD93ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
}  // namespace
PVS_EXPORT pvs_status pvs_debug_rrf_digests(uint64_t *out, uint32_t *out_branches, int32_t *out_path) {
    if (!out || !out_branches || !out_path) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(g_rrf_dig_mu);
    memcpy(out, g_rrf_dig.v, sizeof g_rrf_dig.v);
    *out_branches = g_rrf_dig.nb;
    *out_path = g_rrf_dig.path;
    return PVS_OK;
}

// One branch of an OR-composition, scored (pvs_rrf_cols in the C ABI): every group's aggregate (f64, the reference's
// arithmetic) and its window key.
struct pvs_rrf_cols {
    pvs_index *ix = nullptr;
    double *d_vals = nullptr;               // [n_groups]
    unsigned long long *d_keys = nullptr;   // [n_groups] order-preserving window key (NULL placement and direction folded in)
    uint32_t n_groups = 0;
    std::vector<unsigned long long> sample;  // sorted sample of the keys (threshold proposals)
};
typedef pvs_rrf_cols RrfBranchCols;

// every row's exact distance (the dist_{cte} column), aggregated per group in row order
static pvs_status rrf_score_branch(const pvs_rrf_branch &b, RrfBranchCols *out, uint64_t *dig = nullptr) {
    pvs_index *ix = b.idx;
    out->ix = ix;
    out->n_groups = ix->n_groups;
    if (ix->n == 0) return PVS_OK;
    if (ix->n > (1ull << 31) / 4) return pvs_fail(PVS_ERR_UNSUPPORTED, "more than 2^29 rows in one dense column");
    HIP_TRY(hipSetDevice(ix->device));
    uint32_t t;
    SearchCtx *c = ctx_acquire(ix, &t);
    void *d_q = nullptr;
    float *d_m = nullptr, *d_w = nullptr;
    auto one = [&]() -> pvs_status {
        PVS_TRY(ctx_prepare(ix, *c, 1, 1, false));
        const size_t qbytes = (size_t)ix->dim * (b.query_dtype == PVS_I8 ? 1 : 4);
        HIP_TRY(pvs_scratch_alloc(&d_q, qbytes));
        HIP_TRY(hipMemcpyAsync(d_q, b.query, qbytes, hipMemcpyHostToDevice, c->stream));
        if (b.row_weights) {
            HIP_TRY(pvs_scratch_alloc((void **)&d_w, ix->n * 4));
            HIP_TRY(hipMemcpyAsync(d_w, b.row_weights, ix->n * 4, hipMemcpyHostToDevice, c->stream));
        }
        HIP_TRY(pvs_scratch_alloc((void **)&d_m, ix->n * 4));
        HIP_TRY(pvs_scratch_alloc((void **)&out->d_vals, (size_t)std::max<uint32_t>(ix->n_groups, 1) * 8));
        HIP_TRY(pvs_scratch_alloc((void **)&out->d_keys, (size_t)std::max<uint32_t>(ix->n_groups, 1) * 8));
        PVS_TRY(prep_chunk(ix, *c, d_q, b.query_dtype, 0, 1, 32, b.metric));
        // (Folding MIN / MAX per file into k_score_i8_direct's epilogue — the tile records of the one-pass per-item scorer, a 5-step
        //  segmented scan per tile — was built and measured at configs[4]: 6.81-6.88 ms per composed query against 6.68-6.72 with
        //  the two kernels below.  The scorer's waves have no slack for it, while the aggregate of one branch runs under the other
        //  branch's scoring for free.)
        PVS_TRY(dense_chunk(ix, *c, 1, 32, b.metric, d_m));
        if (dig) PVS_TRY(pvs_digest_device(d_m, ix->n, 4, &dig[0], c->stream));
        HIP_TRY(pvs_launch_group_aggregate(d_m, 1, 1, 0, ix->d_grp_off, ix->d_grp_rows, ix->n_groups, d_w, nullptr, b.agg, out->d_vals, c->stream));
        if (dig) PVS_TRY(pvs_digest_device(out->d_vals, ix->n_groups, 8, &dig[1], c->stream));
        PVS_TRY(pvs_rrf_window_keys(out->d_vals, ix->n_groups, b.row_n_descending != 0, out->d_keys, c->stream));
        if (dig) PVS_TRY(pvs_digest_device(out->d_keys, ix->n_groups, 8, &dig[2], c->stream));
        return PVS_OK;
    };
    pvs_status st = one();
    // The fusion steps that read the columns run on the index's search stream.  With one stream per context
    // (pvs_index_set_streams > 1) that is another stream than this context's: the columns must be complete before they start.
    if (st == PVS_OK && c->stream != ix->search_stream) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) st = pvs_fail(PVS_ERR_DEVICE, "scoring a branch: %s", hipGetErrorString(e));
    }
    pvs_scratch_free_on(d_q, c->stream);  // (the scoring is still queued: the blocks are reusable once the stream has passed this point)
    pvs_scratch_free_on(d_m, c->stream);
    pvs_scratch_free_on(d_w, c->stream);
    ix->searches++;
    ix->dense_queries++;
    ctx_done(ix, c);
    return st;
}

// ---- the pieces of the bounded fusion as C-ABI entry points: a host that shards a branch BY GROUP over several GPUs (or
// ranks) runs them per shard and exchanges a few thousand (group id, key) pairs between the steps (sharded.py: rrf_search_sharded)
PVS_EXPORT pvs_status pvs_rrf_cols_create(const pvs_rrf_branch *branch, pvs_rrf_cols **out) {
    if (!branch || !out) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    pvs_index *ix = branch->idx;
    GateShared gate(ix);  // (pvs_gate.hip)
    if (ix && is_multi(ix)) return pvs_fail(PVS_ERR_UNSUPPORTED, "one pvs_rrf_cols per single-device shard");
    PVS_TRY(validate_search(ix, branch->query, branch->query_dtype, 1, 1, branch->metric));
    if (!branch->row_weights && branch->agg != PVS_AGG_MIN && branch->agg != PVS_AGG_MAX && branch->agg != PVS_AGG_AVG)
        return pvs_fail(PVS_ERR_INVALID_ARG, "aggregation must be MIN, MAX or AVG");
    HIP_TRY(hipSetDevice(ix->device));
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        PVS_TRY(ensure_groups(ix));
    }
    pvs_rrf_cols *c = new (std::nothrow) pvs_rrf_cols();
    if (!c) return pvs_fail(PVS_ERR_OOM, "host allocation failed");
    pvs_status st = rrf_score_branch(*branch, c);
    if (st != PVS_OK) {
        (void)hipDeviceSynchronize();  // (whatever was queued before the failure may still write the columns)
        pvs_scratch_free(c->d_vals);
        pvs_scratch_free(c->d_keys);
        delete c;
        return st;
    }
    *out = c;
    return PVS_OK;
}
PVS_EXPORT void pvs_rrf_cols_destroy(pvs_rrf_cols *c) {
    if (!c) return;
    if (c->ix) {
        (void)hipSetDevice(c->ix->device);
        (void)hipStreamSynchronize(c->ix->search_stream);  // (the fusion steps that read the columns run there)
    }
    pvs_scratch_free(c->d_vals);
    pvs_scratch_free(c->d_keys);
    delete c;
}
PVS_EXPORT pvs_status pvs_rrf_cols_groups(pvs_rrf_cols *c, uint64_t *out_n_groups) {
    if (!c || !out_n_groups) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    *out_n_groups = c->n_groups;
    return PVS_OK;
}
// a window key at or below which about 1.5 x target_groups of this shard's groups lie (from an 8,192-key sample); all ones
// when the shard has no more groups than that
PVS_EXPORT pvs_status pvs_rrf_cols_threshold(pvs_rrf_cols *c, uint64_t target_groups, uint64_t *out_key) {
    if (!c || !out_key) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    constexpr uint32_t M = 16384;
    *out_key = ~0ull;
    if (c->n_groups == 0 || target_groups * 2 >= c->n_groups) return PVS_OK;
    HIP_TRY(hipSetDevice(c->ix->device));
    if (c->sample.empty()) {
        c->sample.resize(M);
        PVS_TRY(pvs_rrf_sample_keys(c->d_keys, c->n_groups, M, c->sample.data(), c->ix->search_stream));
    }
    uint64_t j = (uint64_t)((double)M * 1.5 * (double)target_groups / (double)c->n_groups) + 1;
    if (j >= M) j = M - 1;
    std::nth_element(c->sample.begin(), c->sample.begin() + j, c->sample.end());  // (the j-th smallest: no full sort of the sample)
    *out_key = c->sample[j];
    return PVS_OK;
}
PVS_EXPORT pvs_status pvs_rrf_cols_page(pvs_rrf_cols *c, uint64_t key, uint32_t cap, int64_t *out_gids, uint64_t *out_keys, uint32_t *out_count) {
    if (!c || !out_gids || !out_keys || !out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    *out_count = 0;
    if (c->n_groups == 0) return PVS_OK;
    HIP_TRY(hipSetDevice(c->ix->device));
    return pvs_rrf_page(c->d_keys, c->ix->d_grp_ids, c->n_groups, key, cap, out_gids, (unsigned long long *)out_keys, out_count, c->ix->search_stream);
}
PVS_EXPORT pvs_status pvs_rrf_cols_lookup(pvs_rrf_cols *c, const int64_t *gids, uint32_t m, uint64_t *out_keys, uint8_t *out_present) {
    if (!c || (m && (!gids || !out_keys || !out_present))) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (c->n_groups == 0) {
        for (uint32_t i = 0; i < m; i++) out_present[i] = 0, out_keys[i] = 0;
        return PVS_OK;
    }
    HIP_TRY(hipSetDevice(c->ix->device));
    return pvs_rrf_lookup(c->d_keys, c->ix->d_grp_ids, c->n_groups, gids, m, (unsigned long long *)out_keys, out_present, c->ix->search_stream);
}
PVS_EXPORT pvs_status pvs_rrf_cols_count_below(pvs_rrf_cols *c, const uint64_t *keys, const int64_t *gids, uint32_t m, uint64_t *out_below) {
    if (!c || (m && (!keys || !gids || !out_below))) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    for (uint32_t i = 1; i < m; i++)
        if (keys[i - 1] > keys[i] || (keys[i - 1] == keys[i] && gids[i - 1] >= gids[i]))
            return pvs_fail(PVS_ERR_INVALID_ARG, "candidates must be strictly increasing in (key, group id)");
    if (c->n_groups) HIP_TRY(hipSetDevice(c->ix->device));
    return pvs_rrf_count_below(c->d_keys, c->ix ? c->ix->d_grp_ids : nullptr, c->n_groups, (const unsigned long long *)keys, gids, m,
                               (unsigned long long *)out_below, c->ix ? c->ix->search_stream : nullptr);
}

// SQLite's arithmetic for one fused score (pql/builder.rs:1284-1301), the same expression the device kernel evaluates
static double rrf_score_host(const int64_t *ranks, const PvsRrfParams &p) {
    const int64_t BIG = 9223372036854775805LL;
    double tot = 0.0;
    for (uint32_t b = 0; b < p.n_branches; b++) {
        const int64_t rank = ranks[b] < 0 ? BIG : ranks[b];
        int64_t di;
        const double denom = __builtin_add_overflow((int64_t)p.k[b], rank, &di) ? (double)p.k[b] + (double)rank : (double)di;
        const double t = (1.0 / denom) * p.w[b];
        tot = b == 0 ? t : tot + t;
    }
    return tot;
}

// Bounded fusion.  The reference ranks EVERY group of every branch (row_number() over the whole CTE) and sorts the union;
// only groups near the top of some branch can reach the page, so:
//   1. per branch, the page of groups whose window key is at or below a threshold T_b taken from a sample of the keys:
//      R_b groups, the first R_b of the branch's ranking (NULL aggregates included where the window puts them);
//   2. candidates = union of the pages; for each candidate its EXACT window rank in every branch — one counting pass over
//      the branch's keys (k_rank_count) — and so its exact fused score, in SQLite's arithmetic;
//   3. a group outside every page has rank > R_b everywhere (or is absent, a still smaller term), so with weights >= 0 it
//      scores at most U = sum_b w_b / (k_b + R_b + 1).  When the k-th best candidate beats U the candidates' first k ARE the
//      reference's page; otherwise the thresholds move up (x4 groups) and the loop repeats; a page that would hold a
//      quarter of a branch falls back to the full ranking below.
// Cost beside the exact scoring of every row: a few passes over 8 B per group instead of three multi-pass radix sorts of
// all groups (configs[4]: 2 x 8.3M groups — the sorts were 10 of 16 ms).
// The branches of a composition live in different indexes (their own streams, contexts and scratch): their per-branch steps run
// on one host thread each, so that the round trips of one branch (uploads, the exact-range flag, page and count read-backs)
// hide behind the other's kernels.  Branches that share an index run one after the other.
template <class F>
static pvs_status per_branch(const pvs_rrf_branch *br, uint32_t nb, F &&f) {
    bool distinct = true;
    for (uint32_t a = 0; a < nb; a++)
        for (uint32_t b = a + 1; b < nb; b++) distinct &= br[a].idx != br[b].idx;
    const bool serial = pvs_dbg(PVS_DBG_RRF_SERIAL) != 0;  // tuning
    if (nb == 1 || !distinct || serial) {
        for (uint32_t b = 0; b < nb; b++) PVS_TRY(f(b));
        return PVS_OK;
    }
    std::vector<pvs_status> st(nb, PVS_OK);
    std::vector<std::string> msg(nb);
    std::vector<std::thread> th;
    for (uint32_t b = 1; b < nb; b++)
        th.emplace_back([&, b] {
            st[b] = f(b);
            if (st[b] != PVS_OK) msg[b] = pvs_last_error();
        });
    st[0] = f(0);
    for (auto &t : th) t.join();
    if (st[0] != PVS_OK) return st[0];
    for (uint32_t b = 1; b < nb; b++)
        if (st[b] != PVS_OK) return pvs_fail(st[b], "%s", msg[b].c_str());
    return PVS_OK;
}

// pvs_debug_set("rrf_trace", 1): host wall time of every phase of a composed query on stderr (tuning)
struct RrfTrace {
    bool on = pvs_dbg(PVS_DBG_RRF_TRACE) != 0;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[rrf] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};
// One round of the bounded fusion on the device (pvs_rrf_device.hip): pages, union, exact ranks, fused scores and the first k on
// one stream behind the branches' scoring, one synchronisation.  *usable = false: a size limit of the device form was hit (a
// flag) — the host form redoes the round.  Otherwise R[] holds the page sizes and, when the bound is met, the page is written.
static pvs_status rrf_round_on_device(const pvs_rrf_branch *br, std::vector<RrfBranchCols> &cols, const PvsRrfParams &p, uint64_t target, uint32_t k,
                                      int64_t *out_groups, double *out_scores, uint32_t *out_count, bool *usable, bool *done, uint32_t *out_m) {
    *usable = false;
    const uint32_t nb = p.n_branches;
    pvs_index *root = br[0].idx;
    HIP_TRY(hipSetDevice(root->device));
    hipStream_t s = root->search_stream;
    const unsigned long long *keys[PVS_RRF_MAX_BRANCHES];
    const int64_t *gids[PVS_RRF_MAX_BRANCHES];
    uint32_t n[PVS_RRF_MAX_BRANCHES];
    std::vector<hipEvent_t> evs;
    uint32_t t;
    SearchCtx *c = ctx_acquire(root, &t);  // (its pinned block takes the round's output)
    void *d_work = nullptr;
    auto body = [&]() -> pvs_status {
        for (uint32_t b = 0; b < nb; b++) {
            keys[b] = cols[b].d_keys;
            gids[b] = br[b].idx->d_grp_ids;
            n[b] = cols[b].n_groups;
            if (br[b].idx->search_stream != s) {  // the branch's keys are queued on its own index's stream
                hipEvent_t e;
                HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                evs.push_back(e);
                HIP_TRY(hipEventRecord(e, br[b].idx->search_stream));
                HIP_TRY(hipStreamWaitEvent(s, e, 0));
            }
        }
        PVS_TRY(ctx_pinned_io(*c, pvs_rrf_round_device_out_bytes(nb, k)));
        HIP_TRY(pvs_scratch_alloc(&d_work, pvs_rrf_round_device_work_bytes(nb)));
        HIP_TRY(pvs_rrf_round_device(keys, gids, n, nb, p, (uint32_t)target, k, d_work, c->h_io, s));
        HIP_TRY(hipStreamSynchronize(s));
        uint32_t flags = 0, R[PVS_RRF_MAX_BRANCHES], m = 0, n_out = 0;
        const int64_t *g = nullptr;
        const double *sc = nullptr;
        pvs_rrf_round_device_result(c->h_io, nb, k, &flags, R, &m, &n_out, &g, &sc);
        if (flags) return PVS_OK;
        *usable = true;
        *out_m = m;
        double U = 0.0;  // the most a group outside every page can score (rrf_bounded)
        for (uint32_t b = 0; b < nb; b++) U += p.w[b] / ((double)p.k[b] + (double)R[b] + 1.0);
        U *= 1.0 + 1e-12;
        if (m >= k && n_out == k && sc[k - 1] > U) {
            memcpy(out_groups, g, (size_t)k * 8);
            memcpy(out_scores, sc, (size_t)k * 8);
            *out_count = k;
            *done = true;
        }
        return PVS_OK;
    };
    pvs_status st = body();
    if (st != PVS_OK) (void)hipStreamSynchronize(s);
    for (hipEvent_t e : evs) (void)hipEventDestroy(e);
    pvs_scratch_free(d_work);
    ctx_done(root, c);
    return st;
}

static pvs_status rrf_bounded(const pvs_rrf_branch *br, std::vector<RrfBranchCols> &cols, const PvsRrfParams &p, uint32_t k,
                              int64_t *out_groups, double *out_scores, uint32_t *out_count, bool *done, RrfDigestRec *dig = nullptr) {
    *done = false;
    const uint32_t nb = p.n_branches;
    uint64_t total_groups = 0;
    for (uint32_t b = 0; b < nb; b++) {
        if (!(p.w[b] >= 0.0) || p.k[b] < 0) return PVS_OK;  // the bound needs non-negative terms (NaN weights too: full path)
        total_groups += cols[b].n_groups;
    }
    if (total_groups < 65536) return PVS_OK;  // small: the full ranking is cheap
    // first page size: R_b ~ 4k already puts the bound (sum of w/(k_b + R_b + 1)) far below the k-th fused score unless the
    // branches barely overlap near the top; every host-side step below is linear (or n log n) in the pages
    uint64_t target = std::max<uint64_t>(4ull * k, 1024);
    RrfTrace tr;
    for (int round = 0; round < 6; round++, target *= 4) {
        std::vector<uint32_t> R(nb, 0);
        std::vector<int64_t> cand;
        for (uint32_t b = 0; b < nb; b++)
            if (cols[b].n_groups && target * 4 >= cols[b].n_groups) return PVS_OK;  // a page that would hold a quarter of the branch: full ranking
        // the first round entirely on the device (the stage digests read the host form's intermediate results: that form then)
        if (round == 0 && !dig && !pvs_dbg(PVS_DBG_RRF_HOST_ROUNDS) && pvs_rrf_round_device_supported(nb, target, k)) {
            bool usable = false;
            uint32_t m = 0;
            PVS_TRY(rrf_round_on_device(br, cols, p, target, k, out_groups, out_scores, out_count, &usable, done, &m));
            tr.lap("round on the device");
            if (tr.on) fprintf(stderr, "[rrf] round 0 on the device: target %llu, %u candidates%s\n", (unsigned long long)target, m, usable ? "" : " (size limit: host form)");
            if (*done) return PVS_OK;
            if (usable) continue;  // the bound was not met: larger pages
        }
        std::vector<std::vector<int64_t>> page(nb);
        std::vector<uint8_t> overflow(nb, 0);
        PVS_TRY(per_branch(br, nb, [&](uint32_t b) -> pvs_status {
            const uint32_t n = cols[b].n_groups;
            if (n == 0) return PVS_OK;
            uint64_t thr = 0;
            PVS_TRY(pvs_rrf_cols_threshold(&cols[b], target, &thr));
            const uint32_t cap = (uint32_t)std::min<uint64_t>(n, 8 * target + 65536);
            page[b].resize(cap);
            std::vector<uint64_t> gk(cap);
            uint32_t cnt = 0;
            PVS_TRY(pvs_rrf_cols_page(&cols[b], thr, cap, page[b].data(), gk.data(), &cnt));
            if (cnt > cap) {  // many equal keys at the threshold (massive ties): full ranking
                overflow[b] = 1;
                return PVS_OK;
            }
            // the sampled threshold is a noisy order statistic (1.6k-5k files for a target of 1k): cut the page back to the files at
            // or below its own target-th smallest key — still "every file with key <= T'", only with a smaller T'
            if (cnt > target) {
                std::vector<uint64_t> ks(gk.begin(), gk.begin() + cnt);
                std::nth_element(ks.begin(), ks.begin() + (target - 1), ks.end());
                const uint64_t t2 = ks[target - 1];
                uint32_t w = 0;
                for (uint32_t i = 0; i < cnt; i++)
                    if (gk[i] <= t2) page[b][w++] = page[b][i];
                cnt = w;
            }
            R[b] = cnt;
            page[b].resize(cnt);
            return PVS_OK;
        }));
        for (uint32_t b = 0; b < nb; b++) {
            if (overflow[b]) return PVS_OK;
            cand.insert(cand.end(), page[b].begin(), page[b].end());
        }
        tr.lap("thresholds + pages");
        std::sort(cand.begin(), cand.end());
        cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
        const uint32_t m = (uint32_t)cand.size();
        tr.lap("union");
        std::vector<std::vector<int64_t>> ranks(nb, std::vector<int64_t>(m, -1));
        PVS_TRY(per_branch(br, nb, [&](uint32_t b) -> pvs_status {
            if (cols[b].n_groups == 0 || m == 0) return PVS_OK;
            std::vector<uint64_t> key(m);
            std::vector<uint8_t> present(m);
            PVS_TRY(pvs_rrf_cols_lookup(&cols[b], cand.data(), m, key.data(), present.data()));
            std::vector<uint32_t> order;
            for (uint32_t c = 0; c < m; c++)
                if (present[c]) order.push_back(c);
            std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return key[x] != key[y] ? key[x] < key[y] : cand[x] < cand[y]; });
            const uint32_t mp = (uint32_t)order.size();
            std::vector<uint64_t> ck(mp), below(mp);
            std::vector<int64_t> cg(mp);
            for (uint32_t i = 0; i < mp; i++) {
                ck[i] = key[order[i]];
                cg[i] = cand[order[i]];
            }
            PVS_TRY(pvs_rrf_cols_count_below(&cols[b], ck.data(), cg.data(), mp, below.data()));
            for (uint32_t i = 0; i < mp; i++) ranks[b][order[i]] = (int64_t)below[i] + 1;
            return PVS_OK;
        }));
        tr.lap("lookups + exact ranks");
        if (dig)
            for (uint32_t b = 0; b < nb; b++) {
                uint64_t h = 0;
                for (uint32_t i = 0; i < m; i++) h += host_mix((uint64_t)cand[i], (uint64_t)ranks[b][i]);
                dig->v[b][3] = h;
            }
        struct GS {
            double s;
            int64_t g;
        };
        std::vector<GS> gs(m);
        for (uint32_t i = 0; i < m; i++) {
            int64_t r[PVS_RRF_MAX_BRANCHES];
            for (uint32_t b = 0; b < nb; b++) r[b] = ranks[b][i];
            gs[i] = {rrf_score_host(r, p), cand[i]};
        }
        std::sort(gs.begin(), gs.end(), [](const GS &a, const GS &b) { return a.s != b.s ? a.s > b.s : a.g < b.g; });  // score DESC, group id
        tr.lap("fuse");
        if (tr.on) fprintf(stderr, "[rrf] round %d: target %llu, %u candidates\n", round, (unsigned long long)target, m);
        // the most a group outside every page can score (absent from a branch: an even smaller term)
        double U = 0.0;
        for (uint32_t b = 0; b < nb; b++) U += p.w[b] / ((double)p.k[b] + (double)R[b] + 1.0);
        U *= 1.0 + 1e-12;
        if (m >= k && gs[k - 1].s > U) {
            for (uint32_t i = 0; i < k; i++) {
                out_groups[i] = gs[i].g;
                out_scores[i] = gs[i].s;
            }
            *out_count = k;
            *done = true;
            return PVS_OK;
        }
    }
    return PVS_OK;
}

static pvs_status rrf_search_impl(const pvs_rrf_branch *br, uint32_t nb, uint32_t k, int64_t *out_groups, double *out_scores, uint32_t *out_count);

// The fused page under the reference's final ordering (`ORDER BY order_rank DESC ..., last_modified DESC`, pql/model.rs:547-553)
// when the branches' rows carry order keys (pvs_index_set_order_keys): groups that tie on the fused score — two files that swap
// places between two branches already do — come out by key descending, then group id.  The page is taken far enough past k
// that every group tying with the k-th score is on it (or the page is everything), re-ordered on the host and cut to k.  A
// group's key: from the first branch (in branch order) that carries keys and holds the group.
PVS_EXPORT pvs_status pvs_rrf_search(const pvs_rrf_branch *br, uint32_t nb, uint32_t k, int64_t *out_groups, double *out_scores,
                                     uint32_t *out_count) {
    const uint32_t nb_ok = br && nb <= (uint32_t)PVS_RRF_MAX_BRANCHES ? nb : 0;
    GateSharedMany gate(br, br + nb_ok, [](const pvs_rrf_branch &b) { return b.idx; });  // (pvs_gate.hip: every branch's index, in address order)
    bool keyed = false;
    if (br && nb >= 1 && nb <= (uint32_t)PVS_RRF_MAX_BRANCHES)
        for (uint32_t b = 0; b < nb; b++) keyed |= br[b].idx && br[b].idx->order_rows == br[b].idx->n && br[b].idx->n;
    if (!keyed) return rrf_search_impl(br, nb, k, out_groups, out_scores, out_count);
    if (!out_groups || !out_scores || !out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (k < 1) return pvs_fail(PVS_ERR_INVALID_ARG, "k must be a positive integer");
    std::vector<int64_t> g;
    std::vector<double> sc;
    uint32_t cnt = 0;
    for (uint64_t kk = std::max<uint64_t>(2ull * k, (uint64_t)k + 64);; kk *= 4) {
        kk = std::min<uint64_t>(kk, 0x7fffffffu);
        g.resize(kk);
        sc.resize(kk);
        PVS_TRY(rrf_search_impl(br, nb, (uint32_t)kk, g.data(), sc.data(), &cnt));
        if (cnt < kk || cnt <= k || sc[cnt - 1] < sc[k - 1] || kk == 0x7fffffffu) break;  // (scores are never NaN: sums of finite terms)
    }
    struct E {
        double s;
        int64_t key, g;
    };
    std::vector<E> e(cnt);
    for (uint32_t i = 0; i < cnt; i++) {
        int64_t key = INT64_MIN;
        for (uint32_t b = 0; b < nb; b++)
            if (index_group_key(br[b].idx, g[i], &key)) break;  // (the groups' keys were built by ensure_groups inside the search above)
        e[i] = {sc[i], key, g[i]};
    }
    std::sort(e.begin(), e.end(), [](const E &a, const E &b) {
        if (a.s != b.s) return a.s > b.s;
        if (a.key != b.key) return a.key > b.key;
        return a.g < b.g;
    });
    const uint32_t nout = std::min(cnt, k);
    for (uint32_t i = 0; i < k; i++) {
        out_groups[i] = i < nout ? e[i].g : -1;
        out_scores[i] = i < nout ? e[i].s : __builtin_nan("");
    }
    *out_count = nout;
    return PVS_OK;
}

static pvs_status rrf_search_impl(const pvs_rrf_branch *br, uint32_t nb, uint32_t k, int64_t *out_groups, double *out_scores,
                                  uint32_t *out_count) {
    if (!br || !out_groups || !out_scores || !out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (nb < 1 || nb > (uint32_t)PVS_RRF_MAX_BRANCHES) return pvs_fail(PVS_ERR_INVALID_ARG, "1..%d branches", PVS_RRF_MAX_BRANCHES);
    if (k < 1) return pvs_fail(PVS_ERR_INVALID_ARG, "k must be a positive integer");
    PvsRrfParams p;
    memset(&p, 0, sizeof p);
    p.n_branches = nb;
    uint64_t total = 0;
    for (uint32_t b = 0; b < nb; b++)
        if (br[b].idx && is_multi(br[b].idx)) {
            if (!br[0].idx || !is_multi(br[0].idx)) return pvs_fail(PVS_ERR_INVALID_ARG, "pvs_rrf_search: single- and multi-device branches cannot be mixed");
            for (uint32_t j = 0; j < nb; j++) {
                if (!br[j].query) return pvs_fail(PVS_ERR_INVALID_ARG, "null query");
                if (br[j].agg != PVS_AGG_MIN && br[j].agg != PVS_AGG_MAX && br[j].agg != PVS_AGG_AVG && !br[j].row_weights)
                    return pvs_fail(PVS_ERR_INVALID_ARG, "aggregation must be MIN, MAX or AVG");
            }
            return multi_rrf_search(br, nb, k, out_groups, out_scores, out_count);
        }
    for (uint32_t b = 0; b < nb; b++) {
        pvs_index *ix = br[b].idx;
        PVS_TRY(validate_search(ix, br[b].query, br[b].query_dtype, 1, 1, br[b].metric));
        if (ix->device != br[0].idx->device) return pvs_fail(PVS_ERR_INVALID_ARG, "all branches must live on one device");
        if (!br[b].row_weights && br[b].agg != PVS_AGG_MIN && br[b].agg != PVS_AGG_MAX && br[b].agg != PVS_AGG_AVG)
            return pvs_fail(PVS_ERR_INVALID_ARG, "aggregation must be MIN, MAX or AVG");
        p.k[b] = br[b].rrf_k;
        p.w[b] = br[b].weight;
        HIP_TRY(hipSetDevice(ix->device));
        std::lock_guard<std::mutex> lk(ix->mu);
        PVS_TRY(ensure_groups(ix));
        total += ix->n_groups;
    }
    std::vector<RrfBranchCols> cols(nb);
    unsigned long long *cat_key = nullptr, *cat_pay = nullptr;
    auto body = [&]() -> pvs_status {
        RrfTrace tr;
        const bool digest = pvs_dbg(PVS_DBG_RRF_DIGEST) != 0;
        RrfDigestRec dig;
        memset(&dig, 0, sizeof dig);
        dig.nb = nb;
        auto publish = [&](int path) {
            if (!digest) return;
            dig.path = path;
            std::lock_guard<std::mutex> lk(g_rrf_dig_mu);
            g_rrf_dig = dig;
        };
        PVS_TRY(per_branch(br, nb, [&](uint32_t b) { return rrf_score_branch(br[b], &cols[b], digest ? dig.v[b] : nullptr); }));
        tr.lap("score branches");
        const bool force_full = pvs_dbg(PVS_DBG_RRF_FULL) != 0;  // tests and profiles: compare the two paths
        if (!force_full) {
            bool done = false;
            PVS_TRY(rrf_bounded(br, cols, p, k, out_groups, out_scores, out_count, &done, digest ? &dig : nullptr));
            if (done) {
                publish(1);
                g_rrf_last_path = 1;
                for (uint32_t i = *out_count; i < k; i++) {
                    out_groups[i] = -1;
                    out_scores[i] = __builtin_nan("");
                }
                return PVS_OK;
            }
        }
        // full ranking: every group of every branch ranked (stable radix sorts), entries appended in branch order, fused by sort
        g_rrf_last_path = 2;
        HIP_TRY(hipSetDevice(br[0].idx->device));
        HIP_TRY(pvs_malloc_retry((void **)&cat_key, std::max<uint64_t>(total, 1) * 8));
        HIP_TRY(pvs_malloc_retry((void **)&cat_pay, std::max<uint64_t>(total, 1) * 8));
        uint64_t off = 0;
        for (uint32_t b = 0; b < nb; b++) {
            pvs_index *ix = br[b].idx;
            if (ix->n == 0) continue;
            PVS_TRY(pvs_rrf_rank_branch(cols[b].d_vals, ix->d_grp_ids, ix->n_groups, br[b].row_n_descending != 0, b, cat_key + off, cat_pay + off,
                                        ix->search_stream));
            if (digest) PVS_TRY(pvs_digest_device(cat_pay + off, ix->n_groups, 8, &dig.v[b][3], ix->search_stream));
            off += ix->n_groups;
        }
        publish(2);
        return pvs_rrf_fuse_device(cat_key, cat_pay, off, p, k, out_groups, out_scores, out_count, br[0].idx->search_stream);
    };
    pvs_status st = body();
    if (st != PVS_OK) (void)hipDeviceSynchronize();  // (a failure part-way may have left work queued on a branch's stream)
    for (auto &c : cols) {
        pvs_scratch_free(c.d_vals);
        pvs_scratch_free(c.d_keys);
    }
    hipFree(cat_key);
    hipFree(cat_pay);
    return st;
}
