// pvs_index.hpp — internals shared by the C-ABI translation units: the index object, the per-search
// contexts, and the helpers that cross file boundaries.  Not part of the ABI (include/pvs.h is).
#pragma once
#include <sched.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>

#include "pvs_kernels.hpp"

// ------------------------------------------------------------------- index
struct PendingChunk {
    uint32_t qoff, nb;
};

struct TimedSpan {
    hipEvent_t a, b;
    int kind;  // 0 = sample scan, 1 = full scan, 2 = finalize, 3 = exchange (all-gather + merge)
    uint64_t rows;
};

struct SearchCtx {
    std::vector<TimedSpan> spans;       // recorded during the current search
    std::vector<TimedSpan> span_pool;   // recycled events
    hipStream_t stream = nullptr;      // the stream this context launches on (shared or own)
    hipStream_t own_stream = nullptr;
    hipEvent_t done = nullptr;         // recorded after the last launch of a search
    hipEvent_t preluded = nullptr;     // recorded behind the k-th select when query prep / pass A / the select run on the index's prelude stream
    hipEvent_t scanned = nullptr;      // recorded behind pass B when pass C runs on the index's side stream
    bool side_finalize = false;        // this search's pass C went to ix->fin_stream (c.done was recorded there)
    bool busy = false;
    // (under ix->mu) a stream-ordered search whose pvs_wait work is being done right now (by its owner or by a writer draining the
    // index: pvs_gate.hip), and one a writer completed on its owner's behalf: pvs_wait hands fin_status / fin_err over and releases
    bool draining = false, finished = false;
    pvs_status fin_status = PVS_OK;
    std::string fin_err;
    uint8_t *d_qin = nullptr;     // host-variant query upload [MAX_BATCH][dim*4]
    uint8_t *d_qmat = nullptr;    // [MAX_BATCH][stride]
    float *d_qpad = nullptr;      // dense exact path: PVS_DENSE_NQ zero-padded f32 queries
    void *d_direct = nullptr;     // one-launch single-query search (pvs_direct.hip): workgroup lists + ticket, allocated on first use
    const uint8_t *cur_mask = nullptr;  // pvs_search_filtered: candidate mask of the search in flight (device, [rows])
    uint8_t *d_mask = nullptr;          // its staging copy when the caller's mask is in host memory
    float *d_aux_masked = nullptr;      // [cap/32][PVS_AUX_REC] row-scalar stream with NaN on rows outside the mask
    uint64_t mask_cap = 0;
    void *d_qstage = nullptr;     // pvs_search: the caller's host queries, staged (grown on demand, never freed per call)
    size_t qstage_cap = 0;
    uint8_t *d_qexact = nullptr;  // [MAX_BATCH][dim*4]
    QInfo *d_qinfo = nullptr;     // [MAX_BATCH]
    float *d_thr = nullptr;       // [MAX_BATCH]
    float *d_gmin = nullptr;      // [MAX_BATCH][GMAX]
    uint32_t *d_cand_cnt = nullptr;  // one flag word (dense int8 path: an L2 sum left the exact range)
    uint2 *d_seg = nullptr;          // [PVS_SEG_PAIRS][PVS_SEG_CAP] candidate segments of the filter scan
    uint32_t *d_seg_cnt = nullptr;   // their fill counts (one per list: PVS_SEG_PAIRS, twice that for the 256-query kernel's half-size lists)
    uint2 *d_cand = nullptr;      // [MAX_BATCH][CAND_CAP]
    uint32_t *d_flat_cnt = nullptr;  // [SCAN_MAX_BATCH] fill counts of the flat candidate lists (segment-overflow rerun)
    uint32_t *d_fin_ub = nullptr, *d_fin_surv = nullptr;  // pass C's global work area (FinalizeArgs.w_*), allocated in multi-stream mode
    unsigned long long *d_fin_sort = nullptr;
    uint32_t *d_need_dense = nullptr;  // [total batch capacity]
    uint32_t *h_need_dense = nullptr;  // pinned
    uint32_t flags_cap = 0;
    // host-variant output staging
    int64_t *d_out_ids = nullptr;
    float *d_out_dist = nullptr;
    uint32_t *d_out_count = nullptr;
    uint64_t out_cap = 0;  // elements (batch*k)
    uint32_t out_batch_cap = 0;
    // pinned host block, mapped into the device's address space: small searches (pvs_search_rows) read their queries / row list
    // from it and write their page into it — no staging copies, one synchronisation per call
    uint8_t *h_io = nullptr;
    size_t h_io_cap = 0;
    uint32_t *h_cert = nullptr;   // pinned: the counters of a certified per-item chunk come back here (pvs_items_float.hip), 2 * PVS_SCAN_MAX_BATCH + 3 words
    DenseWork dense;
    GroupWork gwork;              // per-context sort scratch of pvs_group_rank (searches in flight never share it)
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
    pvs_status p_local_status = PVS_OK;  // sharded search: this rank failed before the exchange and sent a failure record; pvs_wait returns this
    std::string p_local_err;
    // this rank's page is ONE record [ids | dist | counts | flags] (pvs_page_record_*: the exchange is a single all-gather);
    // d_loc_ids / d_loc_dist / d_loc_cnt are views into it for the current (batch, k)
    uint8_t *d_loc_rec = nullptr, *d_all_rec = nullptr;
    size_t loc_rec_cap = 0, all_rec_cap = 0, rec_bytes = 0;
    int64_t *d_loc_ids = nullptr;
    float *d_loc_dist = nullptr;
    uint32_t *d_loc_cnt = nullptr, *h_all_flags = nullptr;
    int64_t *d_loc_keys = nullptr;
    size_t h_all_flags_cap = 0;
    uint32_t sh_world = 0;
    int64_t *p_final_ids = nullptr;
    float *p_final_dist = nullptr;
    uint32_t *p_final_count = nullptr;
};

struct pvs_comm;
int pvs_comm_world_(pvs_comm *c);
int pvs_comm_device_(pvs_comm *c);
pvs_status pvs_comm_gather_records_(pvs_comm *c, const void *rec, void *all_rec, size_t rec_bytes, hipStream_t s);  // ONE all-gather, rec_bytes per rank
pvs_status pvs_comm_allreduce_max_(pvs_comm *c, float *d_inout, uint64_t n, hipStream_t s);
void pvs_comm_abort_(pvs_comm *c);                                               // after a deadline passed: queued collectives fail instead of waiting
pvs_status pvs_comm_wait_stream_(pvs_comm *c, hipStream_t s, const char *what);  // the same for everything queued on s so far
pvs_status pvs_comm_wait_event_(pvs_comm *c, hipEvent_t ev, const char *what);    // bounded wait (pvs_debug "comm_timeout_s"); aborts the communicator on expiry

constexpr uint32_t GMAX = 16384;  // group minima per query (pass A grid * RT * 32 <= GMAX)
constexpr uint32_t NCTX = 16;  // searches in flight per index = the reference's read pool (db/connection.rs:235)

// ---- one host process, several GPUs (pvs_multi.hip): a multi-device index owns one single-device index per
// shard; a MultiCtx is one search in flight across all of them
struct MultiCtx {
    bool busy = false, pending = false;
    bool draining = false, finished = false;  // as in SearchCtx
    pvs_status fin_status = PVS_OK;
    std::string fin_err;
    hipStream_t stream = nullptr;  // on the root device (devices[0]): merge + result copies
    hipEvent_t done = nullptr;
    uint8_t *d_all_rec = nullptr;  // root-side gather buffer: one packed page record per shard (pvs_page_record_*)
    size_t all_rec_cap = 0;
    std::vector<uint32_t> tickets;       // the shard contexts this search holds
    std::vector<void *> d_q;             // per shard: the queries copied to the shard's device (null on the root device)
    std::vector<size_t> q_cap;
    // host-buffer entry point staging (root device)
    void *d_qroot = nullptr;
    size_t qroot_cap = 0;
    int64_t *d_out_ids = nullptr;
    float *d_out_dist = nullptr;
    uint32_t *d_out_cnt = nullptr;
    uint64_t out_cap = 0;
    uint32_t out_batch_cap = 0;
    // the search in flight
    const void *p_queries = nullptr;
    int p_qdtype = 0, p_metric = 0;
    uint32_t p_batch = 0, p_k = 0;
    int64_t *p_out_ids = nullptr;
    float *p_out_dist = nullptr;
    uint32_t *p_out_count = nullptr;
    std::vector<uint8_t> p_fast;
};
struct MultiSegment {
    uint64_t row0, n;   // global rows [row0, row0 + n) ...
    uint32_t shard;
    uint64_t local0;    // ... are rows [local0, local0 + n) of this shard
};

struct pvs_index {
    // multi-device index: shards non-empty, everything below `device` unused except the counters and mu
    std::vector<pvs_index *> shards;
    std::vector<MultiSegment> segs;
    MultiCtx mctx[NCTX];
    int device = 0;
    uint32_t dtype = 0, dim = 0, esz = 0, stride = 0;
    uint64_t n = 0, cap = 0;
    int64_t id_base = 0, last_id = INT64_MIN;
    // Reader / writer gate (pvs_gate.hip; all under mu, waiters on ctx_cv).  The reference mutates while it serves: one writer actor
    // (db/index_writer.rs, db/extraction_write.rs:574-616, db/vector_quants.rs:1347-1438) beside up to 16 read connections
    // (db/connection.rs:235,320-357) under SQLite's snapshot isolation.  Here every entry point that reads the rows holds the gate
    // shared for the duration of the call (stream-ordered searches: for the enqueue; afterwards their context's `pending` flag
    // stands for them), every entry point that changes rows, ids or keys holds it exclusively: it waits for the shared holders,
    // completes the stream-ordered searches in flight on their owners' behalf, and keeps new readers out until it is done — a
    // search observes the index before or after a mutation, never a mix.
    uint32_t gate_shared = 0, gate_writers_waiting = 0;
    bool gate_writer_active = false;
    uint64_t ids_epoch = 0;                     // bumped whenever the id column changes (add, remove): validity of h_ids_cache
    uint64_t ids_cache_epoch = UINT64_MAX;
    uint8_t *d_rows = nullptr;
    float *d_norm2 = nullptr;   // |a|^2, the reference's aMag (sequential f32)
    float *d_rnorm = nullptr;   // 1/|a|
    float *d_scan_cos = nullptr, *d_scan_l2 = nullptr;  // the scan's row-scalar streams: [cap/32][PVS_AUX_REC] (k_scan_aux)
    int64_t *d_ids = nullptr;
    // second sort key (pvs_index_set_order_keys): d_trank[row] = position of the row in (key DESC, id ASC) order, d_tinv its inverse.
    // Wherever a page is ordered by (distance, row) the row is replaced by its tie rank and mapped back on output.
    uint32_t *d_trank = nullptr, *d_tinv = nullptr;
    int64_t *d_order_keys = nullptr;  // the keys themselves (device): the shard merges need the key of every page entry
    uint64_t order_rows = 0;  // rows the tie ranks cover (0: none set)
    std::vector<int64_t> h_order_keys;  // host copy of the keys (the groups' tie order is built from it in ensure_groups)
    // groups in tie order (key of the group's first row DESC, group id ASC), built with the CSR when the keys cover every row
    uint32_t *d_grp_tinv = nullptr, *d_grp_trank = nullptr;  // (d_grp_trank: the inverse)
    std::vector<int64_t> h_grp_ids, h_grp_key;  // per group, in id order (host: the page-first per-item path sorts with them)
    std::vector<int64_t> h_groups;  // optional group ids per row (host copy)
    std::vector<int64_t> h_ids_cache;  // host copy of row ids (lazy; similar_to's id -> row lookup)
    // group CSR on the device (built lazily, rebuilt after adds)
    uint64_t groups_built_n = UINT64_MAX;
    uint32_t n_groups = 0;
    uint32_t *d_grp_off = nullptr, *d_grp_rows = nullptr;
    int64_t *d_grp_ids = nullptr;
    // fused per-item scoring (k_scan MODE 2 with the per-group fold): possible when every group is one run of consecutive rows
    // (group ids non-decreasing in row order: the reference's loader streams ORDER BY item_data.id, a file's vectors adjacent)
    uint32_t *d_row_gidx = nullptr;     // [n] the group slot of every row (the per-item form of the sparse candidate path sorts by it)
    bool groups_are_runs = false;
    uint4 *d_tile_grp = nullptr;        // [ceil(n / 32)] tile records (ScanK.tile_grp)
    uint32_t *d_straddlers = nullptr;   // groups that cross a 32-row tile boundary
    uint32_t n_straddlers = 0;
    // rows whose distance is NULL for every query, per metric ([0] cosine: zero vectors and non-finite components, [1] L2: NaN
    // components), in tie order: the tail of a page that ends in NULL rows (pvs_sparse.hip: pvs_ensure_null_rows, built on first
    // need per index state).  null_weird[m]: rows whose NULL-ness depends on the query (|a|^2 under/overflow; inf components under
    // L2): with any of them the dense fallback stays for that metric.
    uint32_t *d_null_rows[2] = {nullptr, nullptr};
    uint32_t n_null[2] = {0, 0}, null_weird[2] = {0, 0};
    std::atomic<uint64_t> null_built_n{UINT64_MAX};
    uint64_t null_built_epoch = 0, order_epoch = 0;  // order_epoch: bumped by pvs_index_set_order_keys
    std::mutex null_mu;
    float scale = 0.f;
    bool scale_set = false;
    uint32_t forced_path = 0;
    int n_cu = 256;
    std::mutex mu;
    std::condition_variable ctx_cv;  // signalled when a context is released
    SearchCtx ctx[NCTX];
    hipStream_t admin_stream = nullptr;
    hipStream_t search_stream = nullptr;
    hipStream_t pre_stream = nullptr;   // query prep, pass A and the k-th select of a pipelined caller's search: they fill the tail of the PREVIOUS search's pass B
    hipStream_t fin_stream = nullptr;   // pass C of a pipelined caller's search: runs beside the NEXT search's scan (search_enqueue)
    hipStream_t comm_stream = nullptr;  // multi-stream mode: every collective of every context, in program order
    bool multi_stream = false;
    // multi-device parent, per-item work on the devices: every shard's global rows in its local order, resident on devices[0]
    // (pvs_launch_take_rows splits device-space masks there), and pinned page blocks the shards' per-item pages land in and
    // devices[0] merges from (pvs_launch_merge_group_pages)
    std::vector<uint32_t *> d_shard_rows;
    uint64_t shard_rows_n = 0;
    struct PageBlock {
        uint8_t *p = nullptr;
        size_t cap = 0;
        bool busy = false;
    };
    std::vector<PageBlock> page_blocks;
    int64_t *d_grp_key = nullptr;  // per group, in id order: its second sort key (with pvs_index_set_order_keys)
    bool by_group = false;  // multi-device parent: rows are placed by group (group_ids given to every add): per-item operators are shard-local
    bool poisoned = false;  // an add / removal failed half way (multi-device parent: global row order lost; any index: rows compacted, per-row arrays not): every later call fails
    std::atomic<uint64_t> searches{0}, fast_queries{0}, dense_queries{0}, last_candidates{0};
    std::atomic<uint64_t> direct_queries{0};  // single queries answered by the one-launch search (counted in fast_queries too; pvs_debug_get("direct_queries"))
    std::atomic<uint64_t> flat_reruns{0};  // queries that went through the scan twice (segment overflow -> flat candidate lists)
    std::atomic<uint64_t> sparse_queries{0};     // filtered / row-list queries answered by gather-and-score (pvs_sparse.hip)
    std::atomic<uint64_t> null_tail_queries{0};  // cosine pages completed from the zero-norm row list instead of the dense path
    // Request coalescing of the host-buffer entry point (pvs_index_set_coalescing): callers that arrive within a short window
    // share one corpus pass.  `pending` holds the requests not yet taken by a leader; one caller at a time is the leader.
    struct CoalesceReq {
        const void *queries;
        pvs_dtype qdtype;
        uint32_t batch, k;
        pvs_metric metric;
        int kind;       // 0 = row pages (pvs_search: ids + f32 distances), 1 = per-item pages (pvs_search_groups: group ids + f64 values)
        int agg;        // kind 1
        int64_t *out_a; // ids / group ids
        void *out_b;    // f32 distances / f64 values
        uint32_t *out_count;
        pvs_status st = PVS_OK;
        std::string err;
        bool done = false;
    };
    struct {
        std::mutex mu;
        std::condition_variable cv_leader, cv_done;
        std::vector<CoalesceReq *> pending;
        bool leader_active = false;
        std::atomic<uint32_t> window_us{0};
        uint32_t max_batch = 0;
        std::atomic<uint64_t> calls{0}, passes{0};
    } co;
    bool profiling = false;
    std::mutex prof_mu;
    pvs_profile prof{};
};


inline bool is_multi(const pvs_index *ix) { return !ix->shards.empty(); }

// ---- pvs_gate.hip: the reader / writer gate (see pvs_index)
void pvs_gate_shared_enter(pvs_index *ix);
void pvs_gate_shared_exit(pvs_index *ix);
bool pvs_gate_excl_enter(pvs_index *ix);  // waits for the shared holders, completes the stream-ordered searches in flight; false: this thread holds the gate shared (refused)
void pvs_gate_excl_exit(pvs_index *ix);
struct GateShared {
    pvs_index *ix;
    explicit GateShared(pvs_index *i) : ix(i) {
        if (ix) pvs_gate_shared_enter(ix);
    }
    ~GateShared() {
        if (ix) pvs_gate_shared_exit(ix);
    }
    GateShared(const GateShared &) = delete;
    GateShared &operator=(const GateShared &) = delete;
};
struct GateExcl {
    pvs_index *ix;
    bool ok = true;  // false: the calling thread is inside a search of this index (a mutation there would wait for itself)
    explicit GateExcl(pvs_index *i) : ix(i) {
        if (ix) ok = pvs_gate_excl_enter(ix);
    }
    ~GateExcl() {
        if (ix && ok) pvs_gate_excl_exit(ix);
    }
    GateExcl(const GateExcl &) = delete;
    GateExcl &operator=(const GateExcl &) = delete;
};
// several indexes at once (the branches of pvs_rrf_search): entered in address order, each once — two callers that name the same
// branches in different orders never wait for each other's second index while a writer waits for their first
struct GateSharedMany {
    std::vector<pvs_index *> held;
    template <typename It, typename Fn>
    GateSharedMany(It first, It last, Fn &&index_of) {
        for (It it = first; it != last; ++it)
            if (pvs_index *ix = index_of(*it)) held.push_back(ix);
        std::sort(held.begin(), held.end());
        held.erase(std::unique(held.begin(), held.end()), held.end());
        for (pvs_index *ix : held) pvs_gate_shared_enter(ix);
    }
    ~GateSharedMany() {
        for (size_t i = held.size(); i-- > 0;) pvs_gate_shared_exit(held[i]);
    }
    GateSharedMany(const GateSharedMany &) = delete;
    GateSharedMany &operator=(const GateSharedMany &) = delete;
};
// the pvs_wait work of a pending ticket without releasing its context (pvs_search_device.hip / pvs_multi.hip)
pvs_status pvs_ticket_complete_(pvs_index *ix, uint32_t ticket);
pvs_status multi_ticket_complete_(pvs_index *ix, uint32_t ticket);
// (ix->mu held) host copy of the row ids of the index's CURRENT rows in ix->h_ids_cache (keyed on ids_epoch, not on its size)
pvs_status pvs_host_ids_locked(pvs_index *ix);
#define PVS_GATE_REFUSED(gate) \
    if (!(gate).ok) return pvs_fail(PVS_ERR_STATE, "a mutation was requested by a thread that is inside a search of the same index")
#define PVS_POISONED_MSG "this index was left inconsistent by a mutation that failed half way: destroy and rebuild it"

// ---- pvs_api.hip
pvs_status use_device(int32_t device, int *resolved);
// implicit_id0: first row id when row_ids == NULL (INT64_MIN: id_base + row index, the public behaviour)
pvs_status add_impl(pvs_index *ix, const void *rows, bool from_f32, uint64_t n, const int64_t *row_ids, const int64_t *group_ids,
                    pvs_space space, int64_t implicit_id0 = INT64_MIN);
pvs_status check_ids(pvs_index *ix, const int64_t *row_ids, uint64_t n, int64_t *last, int64_t implicit_id0 = INT64_MIN);
void span_begin(pvs_index *ix, SearchCtx &c, int kind, uint64_t rows, hipStream_t on = nullptr);
void span_end(pvs_index *ix, SearchCtx &c, hipStream_t on = nullptr);
// a span whose two events are bound to ONE dispatch by the launcher (ScanArgs.ev_start / ev_stop) instead of being recorded around
// it: returns false (events untouched) when the index is not profiling
// would ONE unmasked query for a page of k rows take the one-launch search (pvs_direct.hip) on this index?
bool pvs_direct_route(const pvs_index *ix, uint32_t k, uint32_t batch = 1);
bool span_bound(pvs_index *ix, SearchCtx &c, int kind, uint64_t rows, hipEvent_t *ev_start, hipEvent_t *ev_stop);
void spans_collect(pvs_index *ix, SearchCtx &c);
pvs_status ctx_prepare(pvs_index *ix, SearchCtx &c, uint32_t batch, uint32_t k, bool host_outputs);
pvs_status ctx_fin_buffers(SearchCtx &c);  // the LDS-light pass C's work area, on first use
// ---- pvs_search.hip (shared with pvs_search_host.hip / pvs_search_device.hip)
// the tie order of the second sort key, when it covers the index's rows (pvs_index_set_order_keys)
static inline const uint32_t *order_tinv(const pvs_index *ix) { return ix->order_rows == ix->n && ix->n ? ix->d_tinv : nullptr; }
// one query through the dense path; q is the query's index inside the current chunk
pvs_status dense_one(pvs_index *ix, SearchCtx &c, uint32_t q, uint32_t k, int metric, int64_t *out_ids, float *out_dist, uint32_t *out_count,
                     DenseBounds bounds = DenseBounds());
bool direct_ok(const pvs_index *ix, const SearchCtx &c, uint32_t batch, uint32_t k);
// h_page: the context's pinned block for the pages [ids | distances | counts (64 B) | stored rows], or nullptr
pvs_status enqueue_direct(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t batch, uint32_t k, int metric, int64_t *oid, float *od,
                          uint32_t *oc, uint8_t *h_page = nullptr);
void ctx_release(SearchCtx &c);
pvs_status ctx_pinned_io(SearchCtx &c, size_t bytes);  // c.h_io holds >= bytes afterwards (contents are not preserved when it grows)
// ---- pvs_search.hip
// request coalescing (pvs_search.hip): runs the call directly or as part of a group of concurrent callers
bool coalescing_applies(pvs_index *ix, uint32_t batch);
pvs_status coalesce_call(pvs_index *ix, int kind, int agg, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                         int64_t *out_a, void *out_b, uint32_t *out_count);
pvs_status validate_search(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric);
bool fast_path_ok(const pvs_index *ix, uint32_t k);
pvs_status prep_chunk(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t qoff, uint32_t nb, uint32_t batch_pad,
                      int metric, hipStream_t on = nullptr);
// block == false: returns nullptr (and sets the error) when every context is taken
SearchCtx *ctx_acquire(pvs_index *ix, uint32_t *ticket, bool block = true);
pvs_status search_host(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                       const uint8_t *mask, pvs_space mask_space, int64_t *out_ids, float *out_dist, uint32_t *out_count,
                       const uint32_t *rows = nullptr, uint64_t n_listed = 0, pvs_space rows_space = PVS_HOST, uint32_t *out_row_idx = nullptr);
// out_row_idx (optional, [batch][k]): the stored-row number of every page entry where the route knows it for free (the one-launch
// search), 0xffffffff elsewhere
void ctx_done(pvs_index *ix, SearchCtx *c);
pvs_status search_enqueue(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t batch, uint32_t k, int metric,
                          int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count, bool *used_fast, bool side_finalize = false);
pvs_status search_fallbacks(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t batch, uint32_t k, int metric,
                            int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count);
pvs_status ctx_reserve_local_pages(SearchCtx &c, uint32_t batch, uint32_t k);
// flags + order keys of the context's local page (pvs_launch_page_finish with this index's ids and keys), on stream s
pvs_status ctx_finish_local_page(pvs_index *ix, SearchCtx &c, uint32_t batch, uint32_t k, hipStream_t s);
// key of a group under pvs_index_set_order_keys (false: the index carries none / does not hold the group)
bool index_group_key(const pvs_index *ix, int64_t g, int64_t *key);
// ---- pvs_sparse.hip
pvs_status pvs_ensure_null_rows(pvs_index *ix);
pvs_status pvs_launch_null_tails(pvs_index *ix, SearchCtx &c, int metric, uint32_t *d_flags, uint32_t *h_flags, uint32_t nq, uint32_t k, int64_t *d_out_ids,
                                 float *d_out_dist, uint32_t *d_out_count);
pvs_status pvs_mask_count(const uint8_t *d_mask, uint64_t n, uint32_t *out_count, hipStream_t s);
pvs_status pvs_mask_compact(const uint8_t *d_mask, uint64_t n, uint32_t *d_list, uint32_t count, hipStream_t s);
pvs_status pvs_list_to_mask(const uint32_t *d_list, uint32_t m, uint64_t n, uint8_t *d_mask, hipStream_t s);  // validates; synchronous
bool pvs_sparse_eligible(const pvs_index *ix, uint64_t m, uint32_t batch, uint32_t k);
// bytes of a context's pinned block that the per-item pages of `chunk` query columns occupy: [64 flag words | groups k x 8 | values k x 8 |
// handled flags | counts] per column — the ONE place that sizes it (search_groups_impl puts the queries behind it, pvs_sparse_search_groups
// and the device page ranking write into it; ADVICE r4: two copies of this sum once disagreed by 4 bytes per column)
static inline size_t pvs_group_pages_bytes(uint32_t chunk, uint32_t k) { return 64 + (size_t)chunk * ((size_t)k * 16 + 8); }
pvs_status pvs_sparse_search_groups(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t batch, uint32_t k, int metric, int agg,
                                    const float *d_weights, const uint32_t *d_list, uint32_t m, int64_t *out_groups, double *out_values, uint32_t *out_count,
                                    bool *handled);
pvs_status pvs_sparse_groups_of_files(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t nb, uint32_t k, int metric, int agg, const float *d_weights,
                                      const uint8_t *d_mask, const uint32_t *d_files, uint32_t m_f, uint32_t m_rows, const uint8_t *skip, int64_t *out_groups,
                                      double *out_values, uint32_t *out_count, bool *handled);
pvs_status pvs_sparse_search(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t batch, uint32_t k, int metric, const uint32_t *d_list,
                             uint32_t m, int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count);
// ---- pvs_items.hip
pvs_status ensure_groups(pvs_index *ix);
// d_m [n][nb] distances -> per-group aggregates -> the page (fanout != 0: ONE output column over all nb target vectors: similar_to)
pvs_status aggregate_and_rank(pvs_index *ix, SearchCtx &c, const float *d_m, uint32_t nb, uint32_t fanout, int agg, const float *d_weights,
                              const uint8_t *d_exclude, uint32_t k, int64_t *out_groups, double *out_values, uint32_t *out_count,
                              FanoutWeights fw = FanoutWeights(), uint32_t skip_when = 1);
// d_out[row * nb + q]: exact distances of the nb queries prepared in ctx c (prep_chunk) — matrix cores for int8, k_dense_exact otherwise
// h_flag (optional): a pinned, device-mapped word — the int8 scorers raise it instead of the context's device word and the call does
// not wait for them (the caller looks at it after its own synchronisation and, if raised, calls again with inorder_only)
pvs_status dense_chunk(pvs_index *ix, SearchCtx &c, uint32_t nb, uint32_t batch_pad, int metric, float *d_out, uint32_t *h_flag = nullptr,
                       bool inorder_only = false);
pvs_status search_groups_impl(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                              pvs_agg agg, const float *row_weights, const uint8_t *mask, pvs_space mask_space, int64_t *out_groups,
                              double *out_values, uint32_t *out_count);
// ---- pvs_items_float.hip: per-item pages over float rows by bound + certify + exact rescan of the candidates
bool pvs_float_certify_applies(const pvs_index *ix, uint32_t nb, uint32_t k);
pvs_status pvs_float_groups_certified(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t q0, uint32_t nb, uint32_t batch_pad, uint32_t k,
                                      int metric, int agg, const float *d_w, const uint8_t *d_mask, int64_t *out_groups, double *out_values,
                                      uint32_t *out_count, bool *handled, std::vector<uint8_t> *redo);
struct SimilarArgs {  // similar_to's options; the row_* arrays are host arrays over the rows of the index they are passed with
    pvs_agg agg;
    const double *row_conf, *row_lang;
    double cw, lw;
    const uint8_t *row_kind;
    bool skip_i2i, skip_t2t;
};
struct SimilarTargets {
    std::vector<uint8_t> hq;         // [n_targets][dim] the target vectors as a query batch (int8 codes or f32) ...
    std::vector<uint32_t> own_rows;  // ... or (hq empty) the targets are these rows of the index similar_core runs on: gathered on the device
    std::vector<double> conf, lang;  // [n_targets] NaN = NULL
    std::vector<uint8_t> kind;       // [n_targets]
};
pvs_status similar_targets(pvs_index *ix, const int64_t *target_row_ids, uint32_t n_targets, const SimilarArgs &a, std::vector<uint64_t> &trow,
                           SimilarTargets &tg, bool vectors_stay_on_device = false);
pvs_status similar_core(pvs_index *ix, const SimilarTargets &tg, uint32_t n_targets, const std::vector<uint32_t> &excluded, uint32_t k,
                        pvs_metric metric, const SimilarArgs &a, int64_t *out_groups, double *out_values, uint32_t *out_count);
// ---- pvs_multi.hip (entry points of a multi-device index; the public functions dispatch here when is_multi())
pvs_status multi_create(const pvs_index_desc *desc, pvs_index **out);
void multi_destroy(pvs_index *ix);
pvs_status multi_add(pvs_index *ix, const void *rows, bool from_f32, uint64_t n, const int64_t *row_ids, const int64_t *group_ids,
                     pvs_space space);
pvs_status multi_set_scale(pvs_index *ix, float scale);
pvs_status multi_set_order_keys(pvs_index *ix, const int64_t *keys, uint64_t n, pvs_space space);
pvs_status multi_stats(pvs_index *ix, pvs_stats *out, size_t out_bytes);
pvs_status multi_read_rows(pvs_index *ix, uint64_t row0, uint64_t n, void *out_host);
pvs_status multi_read_ids(pvs_index *ix, uint64_t row0, uint64_t n, int64_t *out_row_ids, int64_t *out_group_ids);
pvs_status multi_search_host(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                             int64_t *out_ids, float *out_dist, uint32_t *out_count);
pvs_status multi_search_device(pvs_index *ix, const void *d_queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                               int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count, uint32_t *out_ticket);
pvs_status multi_wait(pvs_index *ix, uint32_t ticket);
pvs_status multi_sync(pvs_index *ix);
pvs_status multi_score_all(pvs_index *ix, const void *query, pvs_dtype qdtype, pvs_metric metric, float *out_dist, pvs_space out_space);
pvs_status multi_search_groups(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                               pvs_agg agg, const float *row_weights, const uint8_t *mask, pvs_space mask_space, int64_t *out_groups,
                               double *out_values, uint32_t *out_count);
pvs_status multi_search_filtered(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                                 const uint8_t *mask, pvs_space mask_space, int64_t *out_ids, float *out_dist, uint32_t *out_count);
pvs_status multi_search_rows(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric, const uint32_t *rows,
                             uint64_t n_listed, pvs_space rows_space, int64_t *out_ids, float *out_dist, uint32_t *out_count);
pvs_status multi_search_bounded(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric, int32_t have_gt,
                                double gt, int32_t have_lt, double lt, int64_t *out_ids, float *out_dist, uint32_t *out_count);
pvs_status multi_score_batch(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, pvs_metric metric, float *out_dist,
                             pvs_space out_space);
pvs_status multi_similar_to(pvs_index *ix, const int64_t *target_row_ids, uint32_t n_targets, uint32_t k, pvs_metric metric, const SimilarArgs &a,
                            int64_t *out_groups, double *out_values, uint32_t *out_count);
pvs_status multi_rrf_search(const pvs_rrf_branch *br, uint32_t nb, uint32_t k, int64_t *out_groups, double *out_scores, uint32_t *out_count);
// ---- pvs_comm.hip
