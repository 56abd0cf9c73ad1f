// pvs_similar.hip — C ABI of libpvs: similar_to (filters/item_similarity.rs:84-142, 432-581) — the target item's vectors as a query
// batch against every other row, the fan-out aggregate per file with confidence weights and the CLIP cross-modal gates.  Split out of
// pvs_items.hip in round 5.
#include <chrono>
#include <string>
#include <thread>

#include "pvs_index.hpp"

// similar_to, second half: the target vectors (already a query batch, with their own confidence / language / kind values) against
// the rows of ONE single-device index.  `excluded`: this index's rows that are target rows (left out of the join); a.row_*: host
// arrays over this index's rows.  A multi-device index runs it once per shard (every row of a group on one shard).
pvs_status similar_core(pvs_index *ix, const SimilarTargets &tg, uint32_t n_targets, const std::vector<uint32_t> &excluded, uint32_t k,
                        pvs_metric metric, const SimilarArgs &a, int64_t *out_groups, double *out_values, uint32_t *out_count) {
    HIP_TRY(hipSetDevice(ix->device));
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        PVS_TRY(ensure_groups(ix));
    }
    if (ix->n > (1ull << 31) / (4ull * n_targets)) return pvs_fail(PVS_ERR_UNSUPPORTED, "similar_to fan-out matrix would exceed 2 GiB");
    uint32_t t;
    SearchCtx *c = ctx_acquire(ix, &t);
    void *d_q = nullptr;
    float *d_m = nullptr;
    uint8_t *d_ex = nullptr, *d_kind = nullptr, *d_tkind = nullptr;
    double *d_conf = nullptr, *d_lang = nullptr, *d_tconf = nullptr, *d_tlang = nullptr;
    const bool weighted = a.cw != 0.0 || a.lw != 0.0;
    const bool gated = a.row_kind && (a.skip_i2i || a.skip_t2t);
    auto body = [&]() -> pvs_status {
        PVS_TRY(ctx_prepare(ix, *c, n_targets, k, false));
        FanoutWeights fw;
        fw.on = weighted || gated;
        if (gated) {
            HIP_TRY(pvs_scratch_alloc((void **)&d_kind, std::max<uint64_t>(ix->n, 1)));
            HIP_TRY(hipMemcpyAsync(d_kind, a.row_kind, ix->n, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(pvs_scratch_alloc((void **)&d_tkind, n_targets));
            HIP_TRY(hipMemcpyAsync(d_tkind, tg.kind.data(), n_targets, hipMemcpyHostToDevice, c->stream));
            fw.kind = d_kind;
            fw.t_kind = d_tkind;
            fw.skip_i2i = a.skip_i2i;
            fw.skip_t2t = a.skip_t2t;
        }
        if (weighted) {
            // NULL pointer = every confidence NULL (coalesced to 1 in the kernel)
            auto upload = [&](const double *src, uint64_t n, double **dst) -> pvs_status {
                HIP_TRY(pvs_scratch_alloc((void **)dst, std::max<uint64_t>(n, 1) * 8));
                if (src)
                    HIP_TRY(hipMemcpyAsync(*dst, src, n * 8, hipMemcpyHostToDevice, c->stream));
                else
                    HIP_TRY(hipMemsetAsync(*dst, 0xff, n * 8, c->stream));  // all-ones bits = NaN
                return PVS_OK;
            };
            PVS_TRY(upload(a.row_conf, ix->n, &d_conf));
            PVS_TRY(upload(a.row_lang, ix->n, &d_lang));
            PVS_TRY(upload(tg.conf.empty() ? nullptr : tg.conf.data(), n_targets, &d_tconf));
            PVS_TRY(upload(tg.lang.empty() ? nullptr : tg.lang.data(), n_targets, &d_tlang));
            fw.conf = d_conf;
            fw.lang = d_lang;
            fw.t_conf = d_tconf;
            fw.t_lang = d_tlang;
            fw.cw = a.cw;
            fw.lw = a.lw;
        }
        // what the ranking will ask of the pinned block + the target rows' numbers (sized once: kernels in flight hold pointers into it)
        const size_t io_rows = 4096 + (size_t)n_targets * k * 16 + (size_t)n_targets * 4;
        PVS_TRY(ctx_pinned_io(*c, io_rows + (size_t)n_targets * 4 + 64));
        const size_t qesz = ix->dtype == PVS_I8 ? 1 : 4;
        HIP_TRY(pvs_scratch_alloc(&d_q, std::max<size_t>((size_t)n_targets * ix->dim * qesz, 16)));
        if (tg.hq.empty()) {
            // the targets are rows of THIS index: their vectors become the query batch on the device (one small kernel reading the row
            // numbers from pinned memory) — until round 5 they were read back to the host and uploaded again: a gather, two copies and a
            // synchronisation, ~0.1 ms of a 0.3-ms call
            uint32_t *h_rows = (uint32_t *)(c->h_io + ((io_rows + 63) & ~(size_t)63));
            for (uint32_t i = 0; i < n_targets; i++) h_rows[i] = tg.own_rows[i];
            HIP_TRY(pvs_launch_rows_to_queries((int)ix->dtype, ix->d_rows, ix->stride, ix->dim, h_rows, n_targets, d_q, c->stream));
        } else {
            HIP_TRY(hipMemcpyAsync(d_q, tg.hq.data(), tg.hq.size(), hipMemcpyHostToDevice, c->stream));
        }
        {  // the rows left out: runs of neighbours (an item's vectors are stored side by side)
            std::vector<uint32_t> ex(excluded);
            std::sort(ex.begin(), ex.end());
            std::vector<std::pair<uint32_t, uint32_t>> runs;
            for (size_t i = 0; i < ex.size();) {
                size_t j = i + 1;
                while (j < ex.size() && ex[j] <= ex[j - 1] + 1) j++;
                runs.emplace_back(ex[i], ex[j - 1]);
                i = j;
            }
            if (runs.size() <= 4) {  // ... travel as kernel arguments
                fw.n_ranges = (uint32_t)runs.size();
                for (size_t i = 0; i < runs.size(); i++) {
                    fw.r_lo[i] = runs[i].first;
                    fw.r_hi[i] = runs[i].second;
                }
            } else {  // scattered rows: a byte per row, one fill per run
                HIP_TRY(pvs_scratch_alloc((void **)&d_ex, ix->n + 1));
                HIP_TRY(hipMemsetAsync(d_ex, 0, ix->n + 1, c->stream));
                for (const auto &r : runs) HIP_TRY(hipMemsetAsync(d_ex + r.first, 1, (size_t)(r.second - r.first) + 1, c->stream));
            }
        }
        HIP_TRY(pvs_scratch_alloc((void **)&d_m, std::max<size_t>((size_t)ix->n * n_targets * 4, 4)));
        const uint32_t pad = n_targets <= 32 ? 32 : n_targets <= 64 ? 64 : 128;
        PVS_TRY(prep_chunk(ix, *c, d_q, ix->dtype == PVS_I8 ? PVS_I8 : PVS_F32, 0, n_targets, pad, metric));
        // the int8 scorers' out-of-range flag goes to a pinned word and is looked at after the ranking's own synchronisation: one
        // host round trip less per call (45 us of a 0.36-ms similar_to); raised (never with real embeddings), the call is redone in order
        uint32_t *h_flag = (uint32_t *)(c->h_io + 40);
        *(volatile uint32_t *)h_flag = 0;
        PVS_TRY(dense_chunk(ix, *c, n_targets, pad, metric, d_m, h_flag));
        PVS_TRY(aggregate_and_rank(ix, *c, d_m, n_targets, n_targets, a.agg, nullptr, d_ex, k, out_groups, out_values, out_count, fw));
        // (the page has landed — polled in pinned memory or waited for, rank_values — so the scorer's flag, written by an earlier kernel
        //  of the stream, is final)
        if (*(volatile uint32_t *)h_flag) {
            PVS_TRY(dense_chunk(ix, *c, n_targets, pad, metric, d_m, nullptr, true));
            PVS_TRY(aggregate_and_rank(ix, *c, d_m, n_targets, n_targets, a.agg, nullptr, d_ex, k, out_groups, out_values, out_count, fw));
        }
        return PVS_OK;
    };
    pvs_status st = body();
    for (void *p : {(void *)d_q, (void *)d_m, (void *)d_ex, (void *)d_conf, (void *)d_lang, (void *)d_tconf, (void *)d_tlang, (void *)d_kind, (void *)d_tkind})
        pvs_scratch_free_on(p, c->stream);  // (cached blocks: hipMalloc / hipFree per call cost more than the scoring at the reference's scale)
    ix->searches++;
    ctx_done(ix, c);
    return st;
}

// similar_to, first half: the target rows named by id -> their global row, stored vector (the query batch: int8 codes as they
// are, f16/f32 as f32) and confidence / language / kind values.  Works on both index kinds (pvs_index_read_rows / _read_ids).
pvs_status similar_targets(pvs_index *ix, const int64_t *target_row_ids, uint32_t n_targets, const SimilarArgs &a, std::vector<uint64_t> &trow,
                           SimilarTargets &tg, bool vectors_stay_on_device) {
    trow.resize(n_targets);
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        PVS_TRY(pvs_host_ids_locked(ix));
        for (uint32_t i = 0; i < n_targets; i++) {  // ids are strictly increasing: binary search
            auto it = std::lower_bound(ix->h_ids_cache.begin(), ix->h_ids_cache.end(), target_row_ids[i]);
            if (it == ix->h_ids_cache.end() || *it != target_row_ids[i])
                return pvs_fail(PVS_ERR_INVALID_ARG, "target row id %lld is not in the index", (long long)target_row_ids[i]);
            trow[i] = (uint64_t)(it - ix->h_ids_cache.begin());
        }
    }
    const size_t qesz = ix->dtype == PVS_I8 ? 1 : 4;
    if (vectors_stay_on_device) {  // (a single-device index: similar_core gathers them itself)
        tg.hq.clear();
        tg.own_rows.assign(trow.begin(), trow.end());
    } else {
    tg.hq.resize((size_t)n_targets * ix->dim * qesz);
    // the target's rows are usually consecutive (one item's vectors): one read per run of consecutive rows, not one per row (each
    // read is a gather kernel, a copy and a synchronisation: eight of them were half of a similar_to call at the reference's scale)
    const size_t row_bytes = (size_t)ix->dim * ix->esz;
    std::vector<uint8_t> rowbuf;
    for (uint32_t i = 0; i < n_targets;) {
        uint32_t run = 1;
        while (i + run < n_targets && trow[i + run] == trow[i] + run) run++;
        rowbuf.resize(row_bytes * run);
        PVS_TRY(pvs_index_read_rows(ix, trow[i], run, rowbuf.data()));
        for (uint32_t r = 0; r < run; r++) {
            const uint8_t *src = rowbuf.data() + row_bytes * r;
            uint8_t *dst = tg.hq.data() + (size_t)(i + r) * ix->dim * qesz;
            if (ix->dtype == PVS_F16) {
                for (uint32_t e = 0; e < ix->dim; e++) {
                    _Float16 hv;
                    memcpy(&hv, src + 2 * e, 2);
                    const float f = (float)hv;
                    memcpy(dst + 4 * e, &f, 4);
                }
            } else {
                memcpy(dst, src, row_bytes);
            }
        }
        i += run;
    }
    }
    const double null_v = __builtin_nan("");
    tg.conf.assign(n_targets, null_v);
    tg.lang.assign(n_targets, null_v);
    tg.kind.assign(n_targets, 0);
    for (uint32_t i = 0; i < n_targets; i++) {
        if (a.row_conf) tg.conf[i] = a.row_conf[trow[i]];
        if (a.row_lang) tg.lang[i] = a.row_lang[trow[i]];
        if (a.row_kind) tg.kind[i] = a.row_kind[trow[i]];
    }
    return PVS_OK;
}

static pvs_status similar_to_impl(pvs_index *ix, const int64_t *target_row_ids, uint32_t n_targets, uint32_t k, pvs_metric metric,
                                  pvs_agg agg, const double *row_conf, const double *row_lang, double cw, double lw,
                                  const uint8_t *row_kind, bool skip_i2i, bool skip_t2t, int64_t *out_groups, double *out_values,
                                  uint32_t *out_count) {
    if (!ix || !target_row_ids || !out_groups || !out_values || !out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (k < 1) return pvs_fail(PVS_ERR_INVALID_ARG, "k must be a positive integer");
    if (n_targets == 0 || n_targets > PVS_MAX_BATCH) return pvs_fail(PVS_ERR_INVALID_ARG, "similar_to takes 1..%u target vectors", PVS_MAX_BATCH);
    if (metric != PVS_COSINE && metric != PVS_L2) return pvs_fail(PVS_ERR_INVALID_ARG, "unknown metric");
    if (agg != PVS_AGG_MIN && agg != PVS_AGG_MAX && agg != PVS_AGG_AVG) return pvs_fail(PVS_ERR_INVALID_ARG, "aggregation must be MIN, MAX or AVG");
    const SimilarArgs a{agg, row_conf, row_lang, cw, lw, row_kind, skip_i2i, skip_t2t};
    if (is_multi(ix)) return multi_similar_to(ix, target_row_ids, n_targets, k, metric, a, out_groups, out_values, out_count);
    HIP_TRY(hipSetDevice(ix->device));
    std::vector<uint64_t> trow;
    SimilarTargets tg;
    PVS_TRY(similar_targets(ix, target_row_ids, n_targets, a, trow, tg, true));
    std::vector<uint32_t> excluded(trow.begin(), trow.end());
    return similar_core(ix, tg, n_targets, excluded, k, metric, a, out_groups, out_values, out_count);
}

PVS_EXPORT pvs_status pvs_similar_to(pvs_index *ix, const int64_t *target_row_ids, uint32_t n_targets, uint32_t k, pvs_metric metric,
                                     pvs_agg agg, int64_t *out_groups, double *out_values, uint32_t *out_count) {
    GateShared gate(ix);  // (pvs_gate.hip: a mutation waits for this call, a search never sees one half done)
    return similar_to_impl(ix, target_row_ids, n_targets, k, metric, agg, nullptr, nullptr, 0.0, 0.0, nullptr, false, false, out_groups,
                           out_values, out_count);
}

PVS_EXPORT pvs_status pvs_similar_to_ex(pvs_index *ix, const int64_t *target_row_ids, uint32_t n_targets, uint32_t k, pvs_metric metric,
                                        const pvs_similar_opts *o, int64_t *out_groups, double *out_values, uint32_t *out_count) {
    GateShared gate(ix);  // (pvs_gate.hip: a mutation waits for this call, a search never sees one half done)
    if (!o || o->struct_size < sizeof(pvs_similar_opts)) return pvs_fail(PVS_ERR_INVALID_ARG, "pvs_similar_opts.struct_size too small");
    if (o->confidence_weight != o->confidence_weight || o->language_confidence_weight != o->language_confidence_weight)
        return pvs_fail(PVS_ERR_INVALID_ARG, "confidence weights must be numbers");
    return similar_to_impl(ix, target_row_ids, n_targets, k, metric, o->agg, o->row_confidence, o->row_language_confidence,
                           o->confidence_weight, o->language_confidence_weight, o->row_kind, o->xmodal_i2i == 0, o->xmodal_t2t == 0,
                           out_groups, out_values, out_count);
}
