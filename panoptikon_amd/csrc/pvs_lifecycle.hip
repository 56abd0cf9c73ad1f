// pvs_lifecycle.hip — rows leave and change without a rebuild: pvs_index_remove_rows / pvs_index_replace_rows[_f32].
//
// The reference deletes and rewrites vectors all the time: `embeddings ... ON DELETE CASCADE`
// (migrations/index/20250117193000_init.sql:29-33) fires for every file that disappears (db/files.rs:175-192,
// db/file_scans.rs:480), quant codes are upserted per item_data id (db/vector_quants.rs:1109-1111, 1347-1438).  Until round 5 a
// deleted row meant reloading the whole index from SQLite (0.65-0.71 M rows/s: ~15 s at 10M rows).
//
// Removal COMPACTS on the device: the surviving rows behind the first removed one move up, in row order, inside the tiled
// layout (chunk by chunk through one staging block: a chunk's sources lie at or behind its destination, so chunks in
// ascending order never overwrite a row another chunk still has to read), the per-row arrays (|a|^2, 1/|a|, ids, order keys)
// follow through the same source map, the scan's tile records are rebuilt for the moved range.  No tombstones: every kernel sees
// an ordinary index afterwards, nothing on any search path tests a row for being alive, and timings are those of an index that
// never held the rows.  Cost: the moved suffix is read and written twice at HBM speed (10M x 768 int8, 1 % of the rows
// removed: ~15 GB of traffic) plus the host-side bookkeeping of the per-row vectors.  Derived state (group table, NULL-row lists, tie
// ranks) is rebuilt as after an add: lazily, by the first search that needs it.
//
// Positions: row i of the index afterwards is its i-th surviving row (order kept, ids still strictly increasing).  Per-row arrays a
// caller keeps (candidate masks, row weights, similar_to's per-row options) follow the same rule; order keys and group ids the index
// holds are carried over.
#include <numeric>

#include "pvs_index.hpp"

namespace {

// new row r (r0 <= r < r0 + m, r0 a multiple of 32) <- old row src[r - r0]: 16-byte chunks, written as the tiled image of rows
// [r0, r0 + m) into `stage` (stage row 0 = new row r0: the swizzle of a chunk follows its row's position in its 32-row tile)
__global__ __launch_bounds__(256) void k_rows_move(const uint8_t *rows, uint32_t stride, const uint32_t *src, uint64_t m, uint8_t *stage) {
    const uint32_t cpr = stride >> 4;  // 16-byte chunks per row
    const uint64_t total = m * cpr;
    for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (uint64_t)gridDim.x * 256) {
        const uint64_t r = t / cpr;
        const uint32_t ch = (uint32_t)(t - r * cpr);
        *(uint4 *)(stage + pvs_chunk_off(r, ch, stride)) = *(const uint4 *)(rows + pvs_chunk_off(src[r], ch, stride));
    }
}

// the ids to remove against the index's ids (strictly increasing): alive[pos] = 0, gone[pos] = 1 for every id found
__global__ __launch_bounds__(256) void k_mark_removed(const int64_t *ids, uint64_t n, const int64_t *rm, uint64_t m, uint8_t *alive, uint8_t *gone) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const int64_t want = rm[i];
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (ids[mid] < want)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (lo < n && ids[lo] == want) {
        alive[lo] = 0;
        gone[lo] = 1;
    }
}

template <typename T>
void erase_rows(std::vector<T> &v, const std::vector<uint32_t> &dead, uint64_t n_old) {
    if (v.size() != n_old) return;  // (not held for every row: nothing to carry over)
    uint64_t w = dead[0], d = 0;
    for (uint64_t r = dead[0]; r < n_old; r++) {
        if (d < dead.size() && dead[d] == r) {
            d++;
            continue;
        }
        v[w++] = v[r];
    }
    v.resize(w);
}

// Single-device removal.  dead_out (optional): the removed positions, ascending (the multi-device parent rebuilds its segment table
// from them).  ix->mu is taken here; searches in flight were waited for by the caller.
pvs_status remove_single(pvs_index *ix, const int64_t *row_ids, uint64_t n_ids, uint64_t *out_removed, std::vector<uint32_t> *dead_out) {
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, PVS_POISONED_MSG);
    HIP_TRY(hipSetDevice(ix->device));
    if (out_removed) *out_removed = 0;
    if (dead_out) dead_out->clear();
    if (ix->n == 0 || n_ids == 0) return PVS_OK;
    if (ix->n >= (1ull << 32)) return pvs_fail(PVS_ERR_UNSUPPORTED, "row removal: shards hold fewer than 2^32 rows");
    const uint64_t n_old = ix->n;
    hipStream_t s = ix->admin_stream;
    // which rows go: the ids to remove are looked up on the device (the index's ids ascend), the survivors' old positions and the
    // removed positions come out of two stream compactions — no pass over the rows on the host
    int64_t *d_rm = nullptr;
    uint8_t *d_alive = nullptr, *d_gone = nullptr, *stage = nullptr;
    uint32_t *d_src = nullptr, *d_dead = nullptr;
    void *tmp = nullptr;
    std::vector<uint32_t> dead;
    uint64_t n_new = n_old, r0 = 0;
    int64_t last_survivor = INT64_MIN;
    const bool keyed = ix->order_rows == n_old;
    bool mutating = false;  // the first write to the index's own arrays has been queued: a failure from here on leaves rows, ids and scalars out of step
    auto body = [&]() -> pvs_status {
        HIP_TRY(pvs_scratch_alloc((void **)&d_rm, n_ids * 8));
        HIP_TRY(pvs_scratch_alloc((void **)&d_alive, n_old + 64));
        HIP_TRY(pvs_scratch_alloc((void **)&d_gone, n_old + 64));
        HIP_TRY(hipMemcpyAsync(d_rm, row_ids, n_ids * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync(d_alive, 1, n_old, s));
        HIP_TRY(hipMemsetAsync(d_gone, 0, n_old, s));
        hipLaunchKernelGGL(k_mark_removed, dim3((unsigned)((n_ids + 255) / 256)), dim3(256), 0, s, ix->d_ids, n_old, d_rm, n_ids, d_alive, d_gone);
        HIP_TRY(hipGetLastError());
        uint32_t alive = 0;
        PVS_TRY(pvs_mask_count(d_alive, n_old, &alive, s));  // (synchronises)
        n_new = alive;
        if (n_new == n_old) return PVS_OK;
        const uint64_t n_dead = n_old - n_new;
        HIP_TRY(pvs_scratch_alloc((void **)&d_dead, n_dead * 4));
        PVS_TRY(pvs_mask_compact(d_gone, n_old, d_dead, (uint32_t)n_dead, s));
        dead.resize(n_dead);
        HIP_TRY(hipMemcpyAsync(dead.data(), d_dead, n_dead * 4, hipMemcpyDeviceToHost, s));
        if (n_new) {
            HIP_TRY(pvs_scratch_alloc((void **)&d_src, n_new * 4));
            PVS_TRY(pvs_mask_compact(d_alive, n_old, d_src, (uint32_t)n_new, s));  // new row j <- old row d_src[j]
        }
        HIP_TRY(hipStreamSynchronize(s));
        r0 = (uint64_t)dead[0] & ~31ull;  // the first tile that changes
        const uint64_t m_all = n_new > r0 ? n_new - r0 : 0;  // new rows [r0, n_new) get a (possibly new) source
        const uint64_t chunk = std::max<uint64_t>(32, ((256ull << 20) / ix->stride) & ~31ull);
        if (m_all) {
            // EVERY scratch block before the first write to the index (an allocation that failed between the row move and
            // the per-row arrays left the vectors compacted under unchanged ids and norms)
            HIP_TRY(pvs_scratch_alloc((void **)&stage, std::min<uint64_t>(chunk, (m_all + 31) & ~31ull) * ix->stride));
            HIP_TRY(pvs_scratch_alloc(&tmp, m_all * 8));
        }
        // the id of the last surviving row (later adds must ascend from IT, not from an id that left: ADVICE r5)
        if (n_new) {
            uint32_t src_last = 0;
            HIP_TRY(hipMemcpy(&src_last, d_src + (n_new - 1), 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(&last_survivor, ix->d_ids + src_last, 8, hipMemcpyDeviceToHost));
        }
        mutating = true;
        if (m_all) {
            // the rows, chunk by chunk in ascending order through one staging block
            for (uint64_t off = 0; off < m_all; off += chunk) {
                const uint64_t m = std::min(chunk, m_all - off), m32 = (m + 31) & ~31ull;
                const uint64_t total = m * (ix->stride >> 4);
                if (m != m32) HIP_TRY(hipMemsetAsync(stage, 0, m32 * ix->stride, s));  // (the last tile's rows behind the survivors: zero, like any padding row)
                hipLaunchKernelGGL(k_rows_move, dim3((unsigned)std::min<uint64_t>((total + 255) / 256, 65536)), dim3(256), 0, s, ix->d_rows, ix->stride, d_src + r0 + off, m, stage);
                HIP_TRY(hipGetLastError());
                // (whole tiles go back: the last tile's rows beyond m are rows at or behind n_new — padding from now on)
                HIP_TRY(hipMemcpyAsync(ix->d_rows + (r0 + off) * ix->stride, stage, m32 * ix->stride, hipMemcpyDeviceToDevice, s));
            }
            // the per-row arrays through the same map
            for (int which = 0; which < 4; which++) {
                void *arr = which == 0 ? (void *)ix->d_norm2 : which == 1 ? (void *)ix->d_rnorm : which == 2 ? (void *)ix->d_ids : (void *)ix->d_order_keys;
                const uint32_t eb = which < 2 ? 4u : 8u;
                if (!arr || (which == 3 && !keyed)) continue;
                HIP_TRY(pvs_launch_take_rows(arr, eb, d_src + r0, m_all, tmp, s));
                HIP_TRY(hipMemcpyAsync((uint8_t *)arr + r0 * eb, tmp, m_all * eb, hipMemcpyDeviceToDevice, s));
            }
        }
        // rows [n_new, n_old) are padding now: NaN norms make every scan comparison false; the tile records of everything that
        // moved or emptied are rebuilt
        HIP_TRY(pvs_launch_fill_f32(ix->d_norm2 + n_new, n_old - n_new, __builtin_nanf(""), s));
        HIP_TRY(pvs_launch_fill_f32(ix->d_rnorm + n_new, n_old - n_new, __builtin_nanf(""), s));
        HIP_TRY(pvs_launch_scan_aux(ix->d_norm2, ix->d_rnorm, r0, n_old - r0, ix->d_scan_cos, ix->d_scan_l2, s));
        HIP_TRY(hipStreamSynchronize(s));
        // the tie ranks of the surviving rows (their keys moved with them)
        if (keyed && n_new) PVS_TRY(pvs_build_tie_ranks(ix->d_order_keys, n_new, ix->d_trank, ix->d_tinv, s));
        return PVS_OK;
    };
    pvs_status st = body();
    if (st != PVS_OK) (void)hipStreamSynchronize(s);
    for (void *p : {(void *)d_rm, (void *)d_alive, (void *)d_gone, (void *)d_src, (void *)d_dead, (void *)stage, tmp}) pvs_scratch_free(p);
    if (st != PVS_OK && mutating) {
        // vectors, ids, norms and records no longer describe the same rows: serve nothing rather than wrong pages
        ix->poisoned = true;
        const std::string why = pvs_last_error();
        return pvs_fail(st, "row removal failed after the index had started to change (%s): the index is unusable, destroy and rebuild it", why.c_str());
    }
    PVS_TRY(st);
    if (dead.empty()) return PVS_OK;
    // host mirrors and derived state
    ix->ids_epoch++;
    ix->h_ids_cache.clear();  // (downloaded again by whoever needs it)
    erase_rows(ix->h_groups, dead, n_old);
    if (keyed) erase_rows(ix->h_order_keys, dead, n_old);
    ix->n = n_new;
    ix->last_id = last_survivor;  // (INT64_MIN when nothing is left: any id may come next)
    ix->groups_built_n = UINT64_MAX;
    ix->null_built_n.store(UINT64_MAX, std::memory_order_release);
    if (keyed) {
        ix->order_epoch++;
        ix->order_rows = n_new;
    } else if (ix->order_rows) {  // (keys that did not cover every row were unusable anyway)
        ix->order_rows = 0;
        ix->order_epoch++;
    }
    if (out_removed) *out_removed = dead.size();
    if (dead_out) *dead_out = std::move(dead);
    return PVS_OK;
}

// Single-device replacement: rows (dense [n][dim], the index dtype or f32) overwrite the vectors of the rows with the given ids.
// missing_ok: ids the index does not hold are skipped (the multi-device parent probes every shard); matched counts them.
pvs_status replace_single(pvs_index *ix, const void *rows, bool from_f32, uint64_t n, const int64_t *row_ids, pvs_space space, bool missing_ok,
                          uint64_t *matched) {
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, PVS_POISONED_MSG);
    HIP_TRY(hipSetDevice(ix->device));
    if (matched) *matched = 0;
    if (n == 0) return PVS_OK;
    if (from_f32 && ix->dtype == PVS_I8 && !ix->scale_set) return pvs_fail(PVS_ERR_STATE, "int8 index has no scale artifact: set it before writing f32 rows");
    PVS_TRY(pvs_host_ids_locked(ix));
    const std::vector<int64_t> &ids = ix->h_ids_cache;
    std::vector<uint64_t> pos, which;
    for (uint64_t i = 0; i < n; i++) {
        if (i && row_ids[i] <= row_ids[i - 1]) return pvs_fail(PVS_ERR_INVALID_ARG, "row ids of a replacement must be strictly increasing");
        const auto it = std::lower_bound(ids.begin(), ids.end(), row_ids[i]);
        if (it == ids.end() || *it != row_ids[i]) {
            if (missing_ok) continue;
            return pvs_fail(PVS_ERR_INVALID_ARG, "row id %lld is not in the index (pvs_index_replace_rows rewrites rows it already holds)", (long long)row_ids[i]);
        }
        pos.push_back((uint64_t)(it - ids.begin()));
        which.push_back(i);
    }
    if (pos.empty()) return PVS_OK;
    const size_t row_bytes = (size_t)ix->dim * (from_f32 ? 4 : ix->esz);
    hipStream_t s = ix->admin_stream;
    void *stage = nullptr;
    auto body = [&]() -> pvs_status {
        const uint8_t *src_dev = (const uint8_t *)rows;
        if (space == PVS_HOST) {  // the matched rows, packed, staged once
            std::vector<uint8_t> pack(pos.size() * row_bytes);
            for (size_t j = 0; j < pos.size(); j++) memcpy(pack.data() + j * row_bytes, (const uint8_t *)rows + which[j] * row_bytes, row_bytes);
            HIP_TRY(pvs_scratch_alloc(&stage, pack.size()));
            HIP_TRY(hipMemcpy(stage, pack.data(), pack.size(), hipMemcpyHostToDevice));
            src_dev = (const uint8_t *)stage;
            for (size_t j = 0; j < which.size(); j++) which[j] = j;
        }
        const int mode = (from_f32 && ix->dtype == PVS_I8) ? 0 : (from_f32 && ix->dtype == PVS_F16) ? 1 : 2;
        // runs of consecutive positions whose input rows are consecutive too: one ingest + norm pass each
        uint64_t a = 0;
        while (a < pos.size()) {
            uint64_t b = a + 1;
            while (b < pos.size() && pos[b] == pos[b - 1] + 1 && which[b] == which[b - 1] + 1) b++;
            const uint64_t m = b - a;
            HIP_TRY(pvs_launch_rows_ingest(mode, src_dev + which[a] * row_bytes, ix->dim, ix->esz, pos[a], m, ix->scale, ix->d_rows, ix->stride, s));
            HIP_TRY(pvs_launch_norm2((int)ix->dtype, ix->d_rows, ix->stride, ix->dim, pos[a], m, ix->d_norm2, ix->d_rnorm, s));
            HIP_TRY(pvs_launch_scan_aux(ix->d_norm2, ix->d_rnorm, pos[a], m, ix->d_scan_cos, ix->d_scan_l2, s));
            a = b;
        }
        HIP_TRY(hipStreamSynchronize(s));
        return PVS_OK;
    };
    pvs_status st = body();
    if (st != PVS_OK) (void)hipStreamSynchronize(s);
    pvs_scratch_free(stage);
    PVS_TRY(st);
    ix->null_built_n.store(UINT64_MAX, std::memory_order_release);  // (a rewritten row may have become, or stopped being, a NULL row)
    if (matched) *matched = pos.size();
    return PVS_OK;
}

pvs_status multi_remove(pvs_index *ix, const int64_t *row_ids, uint64_t n_ids, uint64_t *out_removed) {
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, "this multi-device index lost its row order in a failed pvs_index_add: destroy and rebuild it");
    std::lock_guard<std::mutex> lk(ix->mu);
    const uint32_t S = (uint32_t)ix->shards.size();
    std::vector<std::vector<uint32_t>> dead(S);
    uint64_t total = 0;
    for (uint32_t sh = 0; sh < S; sh++) {
        uint64_t r = 0;
        pvs_status st = remove_single(ix->shards[sh], row_ids, n_ids, &r, &dead[sh]);
        if (st != PVS_OK) {
            if (total || ix->shards[sh]->poisoned) ix->poisoned = true;  // (some shards compacted, this one did not — or stopped half way: the global row order is lost)
            return st;
        }
        total += r;
    }
    if (out_removed) *out_removed = total;
    if (!total) return PVS_OK;
    // the segment table: a segment loses the removed rows of its local range and stays one run in both numberings
    const uint64_t n_old = ix->n;
    std::vector<uint32_t> dead_global;
    dead_global.reserve(total);
    std::vector<MultiSegment> segs;
    std::vector<size_t> cur(S, 0);  // per shard: removed positions below the segment's local range (segments of a shard ascend)
    uint64_t row = 0;
    for (const MultiSegment &g : ix->segs) {
        const std::vector<uint32_t> &d = dead[g.shard];
        size_t &c = cur[g.shard];
        while (c < d.size() && d[c] < g.local0) c++;
        const size_t before = c;
        uint64_t gone = 0;
        while (c < d.size() && d[c] < g.local0 + g.n) {
            dead_global.push_back((uint32_t)(g.row0 + (d[c] - g.local0)));
            c++;
            gone++;
        }
        const uint64_t n2 = g.n - gone;
        if (n2) {
            const MultiSegment ng{row, n2, g.shard, g.local0 - before};
            if (!segs.empty() && segs.back().shard == ng.shard && segs.back().row0 + segs.back().n == ng.row0 && segs.back().local0 + segs.back().n == ng.local0)
                segs.back().n += n2;
            else
                segs.push_back(ng);
            row += n2;
        }
    }
    ix->segs = std::move(segs);
    std::sort(dead_global.begin(), dead_global.end());
    if (ix->order_rows == n_old && !dead_global.empty()) {
        erase_rows(ix->h_order_keys, dead_global, n_old);
        ix->order_rows = n_old - total;
    } else if (ix->order_rows) {
        ix->order_rows = 0;
        ix->h_order_keys.clear();
    }
    ix->n = n_old - total;
    ix->shard_rows_n = UINT64_MAX;  // (the shards' global rows are expanded again on next use)
    // the id -> row map of the parent (similar_to's targets, the keyed host merge) describes rows that left, and later
    // adds ascend from the largest id that is still there
    ix->ids_epoch++;
    ix->h_ids_cache.clear();
    ix->last_id = INT64_MIN;
    for (pvs_index *sh : ix->shards)
        if (sh->n) ix->last_id = std::max(ix->last_id, sh->last_id);
    return PVS_OK;
}

pvs_status multi_replace(pvs_index *ix, const void *rows, bool from_f32, uint64_t n, const int64_t *row_ids, pvs_space space) {
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, "this multi-device index lost its row order in a failed pvs_index_add: destroy and rebuild it");
    std::vector<uint8_t> staged;
    if (space != PVS_HOST) {
        // device-space rows: staged through the host once (every shard picks the rows it holds: placement is a scatter, as in
        // pvs_index_add on a multi-device index) — replacements are a few rows per call
        const size_t bytes = (size_t)n * ix->dim * (from_f32 ? 4 : pvs_esz(ix->dtype));
        try {
            staged.resize(bytes);
        } catch (const std::bad_alloc &) {
            return pvs_fail(PVS_ERR_OOM, "out of host memory staging %llu replacement rows", (unsigned long long)n);
        }
        HIP_TRY(hipMemcpy(staged.data(), rows, bytes, hipMemcpyDeviceToHost));
        rows = staged.data();
        space = PVS_HOST;
    }
    std::lock_guard<std::mutex> lk(ix->mu);
    for (uint64_t i = 1; i < n; i++)
        if (row_ids[i] <= row_ids[i - 1]) return pvs_fail(PVS_ERR_INVALID_ARG, "row ids of a replacement must be strictly increasing");
    uint64_t total = 0;
    for (pvs_index *sh : ix->shards) {
        uint64_t m = 0;
        PVS_TRY(replace_single(sh, rows, from_f32, n, row_ids, PVS_HOST, true, &m));
        total += m;
    }
    if (total != n) return pvs_fail(PVS_ERR_INVALID_ARG, "%llu of the %llu row ids are not in the index (the others were rewritten)", (unsigned long long)(n - total), (unsigned long long)n);
    return PVS_OK;
}

pvs_status replace_any(pvs_index *ix, const void *rows, bool from_f32, uint64_t n, const int64_t *row_ids, pvs_space space) {
    if (!ix) return pvs_fail(PVS_ERR_INVALID_ARG, "null index");
    if (n == 0) return PVS_OK;
    if (!rows || !row_ids) return pvs_fail(PVS_ERR_INVALID_ARG, "null rows / row ids");
    GateExcl gate(ix);  // (searches in flight read the rows: they are completed first, new ones wait)
    PVS_GATE_REFUSED(gate);
    if (is_multi(ix)) return multi_replace(ix, rows, from_f32, n, row_ids, space);
    return replace_single(ix, rows, from_f32, n, row_ids, space, false, nullptr);
}
}  // namespace

PVS_EXPORT pvs_status pvs_index_remove_rows(pvs_index *ix, const int64_t *row_ids, uint64_t n, uint64_t *out_removed) {
    if (!ix) return pvs_fail(PVS_ERR_INVALID_ARG, "null index");
    if (out_removed) *out_removed = 0;
    if (n == 0) return PVS_OK;
    if (!row_ids) return pvs_fail(PVS_ERR_INVALID_ARG, "null row ids");
    GateExcl gate(ix);  // (searches in flight read the rows that move: they are completed first, new ones wait)
    PVS_GATE_REFUSED(gate);
    if (is_multi(ix)) return multi_remove(ix, row_ids, n, out_removed);
    return remove_single(ix, row_ids, n, out_removed, nullptr);
}

PVS_EXPORT pvs_status pvs_index_replace_rows(pvs_index *ix, const void *rows, uint64_t n, const int64_t *row_ids, pvs_space rows_space) {
    return replace_any(ix, rows, false, n, row_ids, rows_space);
}
PVS_EXPORT pvs_status pvs_index_replace_rows_f32(pvs_index *ix, const float *rows, uint64_t n, const int64_t *row_ids, pvs_space rows_space) {
    return replace_any(ix, rows, true, n, row_ids, rows_space);
}
