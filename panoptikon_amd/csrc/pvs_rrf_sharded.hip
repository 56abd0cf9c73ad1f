// pvs_rrf_sharded.hip — the PQL `or` of vector filters with reciprocal-rank fusion over branches sharded BY GROUP across ranks
// (every row of a file on one rank): the round loop of the bounded fusion behind ONE C entry point.
//
// Reference composition: UNION of the branches' groups (pql/builder.rs:638-661), per-branch row_number() over EVERY group
// (:757-771), score = sum_b weight_b / (k_b + coalesce(rank_b, BIG)) (:1284-1301), ORDER BY score DESC LIMIT k.  A host that
// shards a 50M-row space over 8 GPUs (BASELINE configs[4]) calls pvs_rrf_search_sharded on every rank with its shard of each
// branch; every rank returns the same page, bit for bit the reference's.
//
// Protocol (what panoptikon_amd/sharded.py prototyped in Python over a host socket; HISTORY.md §4.4):
//   thresholds   each shard proposes a window key from a sample; the MINIMUM over shards is used (ncclAllReduce min), so "at or
//                below T_b" is the same set whichever shard a group lives on, and R_b = the sum of the shards' page sizes is the
//                number of groups ranked before everything outside the pages;
//   candidates   the union of all pages (one all-gather of the padded pages);
//   exact ranks  rank of a candidate in branch b = 1 + the groups strictly before it, summed over shards (one counting pass per
//                shard, ncclAllReduce sum);
//   bound        a group outside every page scores at most U = sum_b w_b / (k_b + R_b + 1): when the k-th candidate beats U the
//                first k candidates are the reference's page; otherwise the thresholds move up 4x.
// Messages are a few thousand (id, key) pairs per round: latency-bound, the xGMI link bandwidth is irrelevant.
//
// Exchange: RCCL on the communicator's device when `comm` is given; otherwise the host's own all-gather (`gather`: a Rust host
// on another transport, the tests' threads-as-ranks on a one-GPU box — RCCL refuses two ranks on one device).
#include <algorithm>
#include <numeric>

#include "pvs_index.hpp"

pvs_status pvs_comm_allgather_host_(pvs_comm *c, const void *send, void *recv, size_t bytes);          // pvs_comm.hip
pvs_status pvs_comm_allreduce_u64_host_(pvs_comm *c, uint64_t *inout, size_t n, int op /* 0 min, 1 sum */);  // pvs_comm.hip

namespace {
struct Exchange {
    pvs_comm *comm;
    pvs_allgather_fn fn;
    void *ctx;
    uint32_t world;
    // recv: world * bytes
    pvs_status gather(const void *send, void *recv, size_t bytes) const {
        if (world == 1) {
            memcpy(recv, send, bytes);
            return PVS_OK;
        }
        if (comm) return pvs_comm_allgather_host_(comm, send, recv, bytes);
        if (fn(ctx, send, recv, (uint64_t)bytes) != 0) return pvs_fail(PVS_ERR_COMM, "the host's all-gather failed");
        return PVS_OK;
    }
    pvs_status reduce_u64(uint64_t *v, size_t n, int op) const {  // 0 = min, 1 = sum; in place, same result on every rank
        if (world == 1 || n == 0) return PVS_OK;
        if (comm) return pvs_comm_allreduce_u64_host_(comm, v, n, op);
        std::vector<uint64_t> all(n * world);
        PVS_TRY(gather(v, all.data(), n * 8));
        for (size_t i = 0; i < n; i++) {
            uint64_t r = all[i];
            for (uint32_t w = 1; w < world; w++) r = op == 0 ? std::min(r, all[(size_t)w * n + i]) : r + all[(size_t)w * n + i];
            v[i] = r;
        }
        return PVS_OK;
    }
    // ragged int64 lists: sizes first, then the lists padded to the longest; out = concatenation in rank order
    pvs_status gather_ragged(const std::vector<int64_t> &mine, std::vector<int64_t> *out, uint64_t *total) const {
        std::vector<uint64_t> sizes(world);
        const uint64_t n = mine.size();
        PVS_TRY(gather(&n, sizes.data(), 8));
        const uint64_t cap = std::max<uint64_t>(*std::max_element(sizes.begin(), sizes.end()), 1);
        std::vector<int64_t> pad(cap, 0), all(cap * world);
        std::copy(mine.begin(), mine.end(), pad.begin());
        PVS_TRY(gather(pad.data(), all.data(), cap * 8));
        *total = 0;
        for (uint32_t w = 0; w < world; w++) {
            out->insert(out->end(), all.begin() + (size_t)w * cap, all.begin() + (size_t)w * cap + sizes[w]);
            *total += sizes[w];
        }
        return PVS_OK;
    }
};
}  // namespace

PVS_EXPORT pvs_status pvs_rrf_search_sharded(const pvs_rrf_branch *branches, uint32_t n_branches, uint32_t k, pvs_comm *comm, uint32_t world,
                                             pvs_allgather_fn gather, void *gather_ctx, int64_t *out_groups, double *out_scores,
                                             uint32_t *out_count) {
    if (!branches || n_branches < 1 || n_branches > (uint32_t)PVS_RRF_MAX_BRANCHES) return pvs_fail(PVS_ERR_INVALID_ARG, "1..8 branches");
    GateSharedMany gate(branches, branches + n_branches, [](const pvs_rrf_branch &b) { return b.idx; });  // (pvs_gate.hip)
    if (k < 1 || !out_groups || !out_scores || !out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "bad page arguments");
    if (comm) world = (uint32_t)pvs_comm_world_(comm);
    if (world < 1 || (world > 1 && !comm && !gather)) return pvs_fail(PVS_ERR_INVALID_ARG, "several ranks need a communicator or an all-gather callback");
    for (uint32_t b = 0; b < n_branches; b++)
        if (!(branches[b].weight >= 0.0) || branches[b].rrf_k < 0)
            return pvs_fail(PVS_ERR_UNSUPPORTED, "the sharded fusion needs non-negative RRF weights and k (its bound on the groups outside the pages)");
    const Exchange ex{comm, gather, gather_ctx, world};
    const uint32_t nb = n_branches;
    std::vector<pvs_rrf_cols *> cols(nb, nullptr);
    // Every rank walks the same sequence of collectives whatever fails locally: a local failure is carried in `bad` and agreed on
    // at the next reduction, never a reason to leave a collective early.
    uint64_t bad = 0;
    std::string bad_msg;
    auto local = [&](pvs_status st) {
        if (st != PVS_OK && !bad) {
            bad = 1;
            bad_msg = pvs_last_error();
        }
    };
    auto agree = [&]() -> pvs_status {  // has any rank failed?
        uint64_t v = bad;
        PVS_TRY(ex.reduce_u64(&v, 1, 1));
        if (v) return pvs_fail(bad ? PVS_ERR_STATE : PVS_ERR_COMM, "%s", bad ? bad_msg.c_str() : "another rank failed its part of the sharded fusion");
        return PVS_OK;
    };
    auto body = [&]() -> pvs_status {
        std::vector<uint64_t> n_loc(nb, 0), n_tot(nb, 0);
        for (uint32_t b = 0; b < nb; b++) {
            local(pvs_rrf_cols_create(&branches[b], &cols[b]));
            if (cols[b]) local(pvs_rrf_cols_groups(cols[b], &n_loc[b]));
        }
        PVS_TRY(agree());
        n_tot = n_loc;
        PVS_TRY(ex.reduce_u64(n_tot.data(), nb, 1));
        uint64_t target = std::max<uint64_t>(4ull * k, 1024);
        for (int round = 0; round < 6; round++) {
            std::vector<uint64_t> R(nb, 0);
            std::vector<int64_t> cand;
            for (uint32_t b = 0; b < nb; b++) {
                uint64_t t = ~0ull;
                if (n_loc[b]) local(pvs_rrf_cols_threshold(cols[b], std::max<uint64_t>(target / world, 64), &t));
                PVS_TRY(ex.reduce_u64(&t, 1, 0));  // the minimum proposal: the same key set on every shard
                const uint32_t cap = (uint32_t)std::min<uint64_t>(8 * target + 65536, 1u << 26);
                std::vector<int64_t> pg(cap);
                std::vector<uint64_t> pk(cap);
                uint32_t cnt = 0;
                if (n_loc[b]) local(pvs_rrf_cols_page(cols[b], t, cap, pg.data(), pk.data(), &cnt));
                if (cnt > cap && !bad) {
                    bad = 1;
                    bad_msg = "a page overflowed (massive ties at the threshold): gather the branch on one device instead";
                }
                PVS_TRY(agree());
                pg.resize(cnt);
                PVS_TRY(ex.gather_ragged(pg, &cand, &R[b]));
            }
            std::sort(cand.begin(), cand.end());
            cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
            const uint32_t m = (uint32_t)cand.size();
            std::vector<int64_t> ranks((size_t)nb * m, -1);
            for (uint32_t b = 0; b < nb && m; b++) {
                std::vector<uint64_t> keys(m, 0), allk((size_t)m * world);
                std::vector<uint8_t> present(m, 0), allp((size_t)m * world);
                if (n_loc[b]) local(pvs_rrf_cols_lookup(cols[b], cand.data(), m, keys.data(), present.data()));
                PVS_TRY(ex.gather(keys.data(), allk.data(), (size_t)m * 8));
                PVS_TRY(ex.gather(present.data(), allp.data(), m));
                std::vector<uint64_t> key(m, 0);
                std::vector<uint32_t> order;
                for (uint32_t i = 0; i < m; i++) {
                    uint32_t owners = 0;
                    for (uint32_t w = 0; w < world; w++)
                        if (allp[(size_t)w * m + i]) {
                            owners++;
                            key[i] = allk[(size_t)w * m + i];
                        }
                    if (owners > 1 && !bad) {
                        bad = 1;
                        bad_msg = "a group lives on two shards: shard the branches BY GROUP";
                    }
                    if (owners) order.push_back(i);
                }
                std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return key[x] != key[y] ? key[x] < key[y] : cand[x] < cand[y]; });  // window order
                const uint32_t mo = (uint32_t)order.size();
                std::vector<uint64_t> ok(mo), below(mo, 0);
                std::vector<int64_t> og(mo);
                for (uint32_t i = 0; i < mo; i++) {
                    ok[i] = key[order[i]];
                    og[i] = cand[order[i]];
                }
                if (n_loc[b] && mo) local(pvs_rrf_cols_count_below(cols[b], ok.data(), og.data(), mo, below.data()));
                PVS_TRY(ex.reduce_u64(below.data(), mo, 1));  // groups strictly before the candidate, over all shards
                for (uint32_t i = 0; i < mo; i++) ranks[(size_t)b * m + order[i]] = (int64_t)below[i] + 1;
            }
            PVS_TRY(agree());
            std::vector<double> score(m, 0.0);
            std::vector<int32_t> ks(nb);
            std::vector<double> ws(nb);
            for (uint32_t b = 0; b < nb; b++) {
                ks[b] = branches[b].rrf_k;
                ws[b] = branches[b].weight;
            }
            if (m) PVS_TRY(pvs_rrf_fuse(ranks.data(), nb, m, ks.data(), ws.data(), score.data()));
            // second sort key (pvs_index_set_order_keys): a candidate's key comes from the lowest branch that carries keys and holds the
            // group — on whichever rank that shard lives.  Every rank offers (branch, key) for the candidates it can answer for and
            // the lowest branch wins; without any keys all entries tie and the order below is (score DESC, group id).
            std::vector<int64_t> ckey(m, INT64_MIN);
            if (m) {
                struct BK {
                    uint64_t branch;
                    int64_t key;
                };
                std::vector<BK> mine(m, BK{~0ull, 0}), allbk((size_t)m * world);
                for (uint32_t i = 0; i < m; i++)
                    for (uint32_t b = 0; b < nb; b++)
                        if (index_group_key(branches[b].idx, cand[i], &mine[i].key)) {
                            mine[i].branch = b;
                            break;
                        }
                PVS_TRY(ex.gather(mine.data(), allbk.data(), (size_t)m * sizeof(BK)));
                for (uint32_t i = 0; i < m; i++) {
                    uint64_t best = ~0ull;
                    for (uint32_t w = 0; w < world; w++) {
                        const BK &e = allbk[(size_t)w * m + i];
                        if (e.branch < best) {
                            best = e.branch;
                            ckey[i] = e.key;
                        }
                    }
                }
            }
            std::vector<uint32_t> top(m);
            std::iota(top.begin(), top.end(), 0u);
            const uint32_t kk = std::min<uint32_t>(k, m);
            std::partial_sort(top.begin(), top.begin() + kk, top.end(), [&](uint32_t x, uint32_t y) {
                if (score[x] != score[y]) return score[x] > score[y];
                if (ckey[x] != ckey[y]) return ckey[x] > ckey[y];
                return cand[x] < cand[y];
            });
            double U = 0.0;
            for (uint32_t b = 0; b < nb; b++) U += ws[b] / ((double)ks[b] + (double)R[b] + 1.0);
            U *= 1.0 + 1e-12;
            bool all_in = true;  // do the pages already hold every group of every branch?
            for (uint32_t b = 0; b < nb; b++) all_in &= R[b] >= n_tot[b];
            if ((m >= k && score[top[kk - 1]] > U) || all_in) {
                for (uint32_t i = 0; i < k; i++) {
                    out_groups[i] = i < kk ? cand[top[i]] : -1;
                    out_scores[i] = i < kk ? score[top[i]] : __builtin_nan("");
                }
                *out_count = kk;
                return PVS_OK;
            }
            target *= 4;
        }
        return pvs_fail(PVS_ERR_UNSUPPORTED, "the bounded fusion did not converge in 6 rounds (k close to the number of groups): fuse on one device");
    };
    pvs_status st;
    try {  // (the page / key vectors hold up to 2^26 entries: no exception may cross the C ABI or end a rank thread)
        st = body();
    } catch (const std::bad_alloc &) {
        st = pvs_fail(PVS_ERR_OOM, "out of host memory in the sharded fusion");
    } catch (...) {
        st = pvs_fail(PVS_ERR_STATE, "unexpected failure in the sharded fusion");
    }
    for (pvs_rrf_cols *c : cols) pvs_rrf_cols_destroy(c);
    return st;
}
