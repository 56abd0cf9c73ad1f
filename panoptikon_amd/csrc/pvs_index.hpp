// pvs_index.hpp — internals shared by the C-ABI translation units: the index object, the per-search
// contexts, and the helpers that cross file boundaries.  Not part of the ABI (include/pvs.h is).
#pragma once
#include <sched.h>

#include <algorithm>
#include <cmath>

#include "pvs_kernels.hpp"

// ------------------------------------------------------------------- index
struct PendingChunk {
    uint32_t qoff, nb;
};

struct TimedSpan {
    hipEvent_t a, b;
    int kind;  // 0 = sample scan, 1 = full scan, 2 = finalize
    uint64_t rows;
};

struct SearchCtx {
    std::vector<TimedSpan> spans;       // recorded during the current search
    std::vector<TimedSpan> span_pool;   // recycled events
    hipStream_t stream = nullptr;      // the stream this context launches on (shared or own)
    hipStream_t own_stream = nullptr;
    hipEvent_t done = nullptr;         // recorded after the last launch of a search
    bool busy = false;
    uint8_t *d_qin = nullptr;     // host-variant query upload [MAX_BATCH][dim*4]
    uint8_t *d_qmat = nullptr;    // [MAX_BATCH][stride]
    float *d_qpad = nullptr;      // dense exact path: PVS_DENSE_NQ zero-padded f32 queries
    const uint8_t *cur_mask = nullptr;  // pvs_search_filtered: candidate mask of the search in flight (device, [rows])
    uint8_t *d_mask = nullptr;          // its staging copy when the caller's mask is in host memory
    float *d_aux_masked = nullptr;      // [cap] per-row scalar stream with NaN on rows outside the mask
    uint64_t mask_cap = 0;
    void *d_qstage = nullptr;     // pvs_search: the caller's host queries, staged (grown on demand, never freed per call)
    size_t qstage_cap = 0;
    uint8_t *d_qexact = nullptr;  // [MAX_BATCH][dim*4]
    QInfo *d_qinfo = nullptr;     // [MAX_BATCH]
    float *d_thr = nullptr;       // [MAX_BATCH]
    float *d_gmin = nullptr;      // [MAX_BATCH][GMAX]
    uint32_t *d_cand_cnt = nullptr;
    uint2 *d_cand = nullptr;      // [MAX_BATCH][CAND_CAP]
    uint32_t *d_need_dense = nullptr;  // [total batch capacity]
    uint32_t *h_need_dense = nullptr;  // pinned
    uint32_t flags_cap = 0;
    // host-variant output staging
    int64_t *d_out_ids = nullptr;
    float *d_out_dist = nullptr;
    uint32_t *d_out_count = nullptr;
    uint64_t out_cap = 0;  // elements (batch*k)
    uint32_t out_batch_cap = 0;
    DenseWork dense;
    // deferred fallback bookkeeping (device variant)
    bool pending = false;
    const void *p_queries = nullptr;
    int p_qdtype = 0, p_metric = 0;
    uint32_t p_batch = 0, p_k = 0;
    int64_t *p_out_ids = nullptr;
    float *p_out_dist = nullptr;
    uint32_t *p_out_count = nullptr;
    bool p_fast = false;
    // sharded search: this rank's page, the gathered pages and flags
    pvs_comm *p_comm = nullptr;
    int64_t *d_loc_ids = nullptr, *d_all_ids = nullptr;
    float *d_loc_dist = nullptr, *d_all_dist = nullptr;
    uint32_t *d_loc_cnt = nullptr, *d_all_cnt = nullptr, *d_all_flags = nullptr, *h_all_flags = nullptr;
    uint64_t sh_elems = 0;
    uint32_t sh_batch = 0, sh_world = 0;
    int64_t *p_final_ids = nullptr;
    float *p_final_dist = nullptr;
    uint32_t *p_final_count = nullptr;
};

struct pvs_comm;
int pvs_comm_world_(pvs_comm *c);
int pvs_comm_device_(pvs_comm *c);
pvs_status pvs_comm_gather_pages_(pvs_comm *c, const int64_t *ids, const float *dist, const uint32_t *cnt, const uint32_t *flags,
                                  int64_t *all_ids, float *all_dist, uint32_t *all_cnt, uint32_t *all_flags, uint64_t elems,
                                  uint32_t batch, hipStream_t s);

constexpr uint32_t GMAX = 16384;  // group minima per query (pass A grid * RT * 32 <= GMAX)
constexpr uint32_t NCTX = 4;

struct pvs_index {
    int device = 0;
    uint32_t dtype = 0, dim = 0, esz = 0, stride = 0;
    uint64_t n = 0, cap = 0;
    int64_t id_base = 0, last_id = INT64_MIN;
    uint8_t *d_rows = nullptr;
    float *d_norm2 = nullptr;   // |a|^2, the reference's aMag (sequential f32)
    float *d_rnorm = nullptr;   // 1/|a|
    int64_t *d_ids = nullptr;
    std::vector<int64_t> h_groups;  // optional group ids per row (host copy)
    std::vector<int64_t> h_ids_cache;  // host copy of row ids (lazy; similar_to's id -> row lookup)
    // group CSR on the device (built lazily, rebuilt after adds)
    uint64_t groups_built_n = UINT64_MAX;
    uint32_t n_groups = 0;
    uint32_t *d_grp_off = nullptr, *d_grp_rows = nullptr;
    int64_t *d_grp_ids = nullptr;
    GroupWork gwork;
    float scale = 0.f;
    bool scale_set = false;
    uint32_t forced_path = 0;
    int n_cu = 256;
    std::mutex mu;
    SearchCtx ctx[NCTX];
    hipStream_t admin_stream = nullptr;
    hipStream_t search_stream = nullptr;
    hipStream_t comm_stream = nullptr;  // multi-stream mode: every collective of every context, in program order
    bool multi_stream = false;
    std::atomic<uint64_t> searches{0}, fast_queries{0}, dense_queries{0}, last_candidates{0};
    bool profiling = false;
    std::mutex prof_mu;
    pvs_profile prof{};
};


// ---- pvs_api.hip
pvs_status use_device(int32_t device, int *resolved);
void span_begin(pvs_index *ix, SearchCtx &c, int kind, uint64_t rows);
void span_end(pvs_index *ix, SearchCtx &c);
void spans_collect(pvs_index *ix, SearchCtx &c);
pvs_status ctx_prepare(pvs_index *ix, SearchCtx &c, uint32_t batch, uint32_t k, bool host_outputs);
// ---- pvs_search.hip
pvs_status validate_search(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric);
bool fast_path_ok(const pvs_index *ix, uint32_t k);
pvs_status prep_chunk(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t qoff, uint32_t nb, uint32_t batch_pad,
                      int metric);
SearchCtx *ctx_acquire(pvs_index *ix, uint32_t *ticket);
pvs_status search_host(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                       const uint8_t *mask, pvs_space mask_space, int64_t *out_ids, float *out_dist, uint32_t *out_count);
void ctx_done(pvs_index *ix, SearchCtx *c);
// ---- pvs_items.hip
pvs_status ensure_groups(pvs_index *ix);
// ---- pvs_comm.hip
pvs_status pvs_comm_gather_group_pages_(pvs_comm *c, const int64_t *groups, const double *values, const uint32_t *cnt, int64_t *all_groups,
                                        double *all_values, uint32_t *all_cnt, uint64_t elems, uint32_t batch, hipStream_t s);
