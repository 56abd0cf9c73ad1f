// pvs_multi.hpp — helpers shared by the translation units of the multi-device index (pvs_multi.hip: contexts, placement, row
// search; pvs_multi_items.hip: masks, per-item pages, similar_to, RRF).
#pragma once
#include <algorithm>
#include <vector>

#include "pvs_index.hpp"

inline int root_device(const pvs_index *ix) { return ix->shards[0]->device; }

// a global per-row host array split into the shards' row orders (segments are in global order and, per shard, in local order)
template <typename T>
inline std::vector<std::vector<T>> split_rows(const pvs_index *ix, const T *global) {
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
inline std::vector<SegRange> locate(const pvs_index *ix, uint64_t row0, uint64_t n) {
    std::vector<SegRange> out;
    auto it = std::upper_bound(ix->segs.begin(), ix->segs.end(), row0, [](uint64_t r, const MultiSegment &g) { return r < g.row0; });
    if (it != ix->segs.begin()) --it;
    for (; it != ix->segs.end() && it->row0 < row0 + n; ++it) {
        const uint64_t a = std::max(row0, it->row0), b = std::min(row0 + n, it->row0 + it->n);
        if (a < b) out.push_back({it->shard, it->local0 + (a - it->row0), b - a, a - row0});
    }
    return out;
}
