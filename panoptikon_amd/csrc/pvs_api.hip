// pvs_api.hip — C ABI of libpvs: index residency in HBM, search orchestration over HIP
// streams, device codec entry points.  No torch, no CPU compute path: every distance
// is produced by a HIP kernel or the call fails (PVS_ERR_DEVICE).
#include <sched.h>

#include <algorithm>
#include <cmath>

#include "pvs_kernels.hpp"

// ------------------------------------------------------------------ errors
static thread_local std::string g_last_error;

pvs_status pvs_fail(pvs_status code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
PVS_EXPORT const char *pvs_last_error(void) { return g_last_error.c_str(); }
PVS_EXPORT uint32_t pvs_abi_version(void) { return PVS_ABI_VERSION; }

PVS_EXPORT int32_t pvs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int ok = 0;
    for (int d = 0; d < n; d++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
}

static pvs_status use_device(int32_t device, int *resolved) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        (void)hipGetLastError();
        return pvs_fail(PVS_ERR_DEVICE, "no HIP device available (%s): libpvs has no CPU path",
                        e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    }
    int d = device;
    if (d < 0) HIP_TRY(hipGetDevice(&d));
    if (d >= n) return pvs_fail(PVS_ERR_INVALID_ARG, "device %d out of range (%d visible)", d, n);
    HIP_TRY(hipSetDevice(d));
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, d));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return pvs_fail(PVS_ERR_DEVICE, "device %d is %s; libpvs is built for gfx950 (MI355X) only", d, p.gcnArchName);
    if (resolved) *resolved = d;
    return PVS_OK;
}

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

static void span_begin(pvs_index *ix, SearchCtx &c, int kind, uint64_t rows) {
    if (!ix->profiling) return;
    TimedSpan t;
    if (!c.span_pool.empty()) {
        t = c.span_pool.back();
        c.span_pool.pop_back();
    } else if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) {
        return;
    }
    t.kind = kind;
    t.rows = rows;
    (void)hipEventRecord(t.a, c.stream);
    c.spans.push_back(t);
}
static void span_end(pvs_index *ix, SearchCtx &c) {
    if (!ix->profiling || c.spans.empty()) return;
    (void)hipEventRecord(c.spans.back().b, c.stream);
}
// after the stream drained
static void spans_collect(pvs_index *ix, SearchCtx &c) {
    if (c.spans.empty()) return;
    std::lock_guard<std::mutex> lk(ix->prof_mu);
    for (auto &t : c.spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            if (t.kind == 0) {
                ix->prof.sample_launches++;
                ix->prof.sample_ms += ms;
            } else if (t.kind == 1) {
                ix->prof.scan_launches++;
                ix->prof.scan_ms += ms;
                ix->prof.scan_rows += t.rows;
            } else {
                ix->prof.finalize_launches++;
                ix->prof.finalize_ms += ms;
            }
        }
        c.span_pool.push_back(t);
    }
    c.spans.clear();
}

static void ctx_release(SearchCtx &c) {
    for (auto &t : c.spans) c.span_pool.push_back(t);
    for (auto &t : c.span_pool) {
        hipEventDestroy(t.a);
        hipEventDestroy(t.b);
    }
    c.spans.clear();
    c.span_pool.clear();
    hipFree(c.d_qin);
    hipFree(c.d_qmat);
    hipFree(c.d_qpad);
    hipFree(c.d_qstage);
    hipFree(c.d_mask);
    hipFree(c.d_aux_masked);
    hipFree(c.d_qexact);
    hipFree(c.d_qinfo);
    hipFree(c.d_thr);
    hipFree(c.d_gmin);
    hipFree(c.d_cand_cnt);
    hipFree(c.d_cand);
    hipFree(c.d_need_dense);
    if (c.h_need_dense) hipHostFree(c.h_need_dense);
    hipFree(c.d_out_ids);
    hipFree(c.d_out_dist);
    hipFree(c.d_out_count);
    pvs_dense_release(c.dense);
    if (c.done) hipEventDestroy(c.done);
    hipFree(c.d_loc_ids);
    hipFree(c.d_all_ids);
    hipFree(c.d_loc_dist);
    hipFree(c.d_all_dist);
    hipFree(c.d_loc_cnt);
    hipFree(c.d_all_cnt);
    hipFree(c.d_all_flags);
    if (c.h_all_flags) hipHostFree(c.h_all_flags);
    if (c.own_stream) hipStreamDestroy(c.own_stream);
    c = SearchCtx();
}

static pvs_status ctx_prepare(pvs_index *ix, SearchCtx &c, uint32_t batch, uint32_t k, bool host_outputs) {
    // Searches are queued on ONE stream by default: consecutive batches run back to back with no
    // host turnaround between them and their scan kernels never compete for the same CUs.
    // pvs_index_set_streams(idx, n > 1) gives every context its own stream instead.
    if (ix->multi_stream) {
        if (!c.own_stream) HIP_TRY(hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking));
        c.stream = c.own_stream;
    } else {
        c.stream = ix->search_stream;
    }
    c.cur_mask = nullptr;
    if (!c.done) HIP_TRY(hipEventCreateWithFlags(&c.done, hipEventDisableTiming));
    if (!c.d_qmat) {
        HIP_TRY(hipMalloc((void **)&c.d_qin, (size_t)PVS_SCAN_MAX_BATCH * ix->dim * 4));
        HIP_TRY(hipMalloc((void **)&c.d_qmat, (size_t)PVS_SCAN_MAX_BATCH * ix->stride));
        HIP_TRY(hipMalloc((void **)&c.d_qpad, pvs_dense_exact_scratch_bytes(ix->stride, ix->esz)));
        HIP_TRY(hipMalloc((void **)&c.d_qexact, (size_t)PVS_SCAN_MAX_BATCH * ix->dim * 4));
        HIP_TRY(hipMalloc((void **)&c.d_qinfo, sizeof(QInfo) * PVS_SCAN_MAX_BATCH));
        HIP_TRY(hipMalloc((void **)&c.d_thr, 4 * PVS_SCAN_MAX_BATCH));
        HIP_TRY(hipMalloc((void **)&c.d_gmin, (size_t)4 * PVS_SCAN_MAX_BATCH * GMAX));
        HIP_TRY(hipMalloc((void **)&c.d_cand_cnt, 4 * PVS_SCAN_MAX_BATCH));
        HIP_TRY(hipMalloc((void **)&c.d_cand, sizeof(uint2) * (size_t)PVS_SCAN_MAX_BATCH * PVS_CAND_CAP));
    }
    if (batch > c.flags_cap) {
        hipFree(c.d_need_dense);
        if (c.h_need_dense) hipHostFree(c.h_need_dense);
        c.d_need_dense = nullptr;
        c.h_need_dense = nullptr;
        uint32_t cap = (uint32_t)pvs_round_up(batch, 256);
        HIP_TRY(hipMalloc((void **)&c.d_need_dense, 4 * (size_t)cap));
        HIP_TRY(hipHostMalloc((void **)&c.h_need_dense, 4 * (size_t)cap, hipHostMallocDefault));
        c.flags_cap = cap;
    }
    if (host_outputs) {
        uint64_t need = (uint64_t)batch * k;
        if (need > c.out_cap || batch > c.out_batch_cap) {
            hipFree(c.d_out_ids);
            hipFree(c.d_out_dist);
            hipFree(c.d_out_count);
            c.d_out_ids = nullptr;
            c.d_out_dist = nullptr;
            c.d_out_count = nullptr;
            HIP_TRY(hipMalloc((void **)&c.d_out_ids, 8 * need));
            HIP_TRY(hipMalloc((void **)&c.d_out_dist, 4 * need));
            HIP_TRY(hipMalloc((void **)&c.d_out_count, 4 * (size_t)batch));
            c.out_cap = need;
            c.out_batch_cap = batch;
        }
    }
    return PVS_OK;
}

pvs_status pvs_index_reserve_(pvs_index *ix, uint64_t rows);

PVS_EXPORT pvs_status pvs_index_create(const pvs_index_desc *desc, pvs_index **out) {
    if (!desc || !out) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (desc->struct_size < sizeof(pvs_index_desc)) return pvs_fail(PVS_ERR_INVALID_ARG, "pvs_index_desc.struct_size too small");
    if (desc->dtype > PVS_I8) return pvs_fail(PVS_ERR_INVALID_ARG, "unknown dtype %u", desc->dtype);
    if (desc->dim == 0 || desc->dim > 16384) return pvs_fail(PVS_ERR_INVALID_ARG, "dim %u out of range [1, 16384]", desc->dim);
    int dev = 0;
    PVS_TRY(use_device(desc->device, &dev));
    pvs_index *ix = new (std::nothrow) pvs_index();
    if (!ix) return pvs_fail(PVS_ERR_OOM, "host allocation failed");
    ix->device = dev;
    ix->dtype = desc->dtype;
    ix->dim = desc->dim;
    ix->esz = pvs_esz(desc->dtype);
    ix->stride = (uint32_t)pvs_round_up((uint64_t)desc->dim * ix->esz, PVS_KSLAB_BYTES);
    ix->id_base = desc->id_base;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) == hipSuccess) ix->n_cu = p.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&ix->admin_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ix->search_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ix->comm_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ix;
        return pvs_fail(PVS_ERR_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    *out = ix;
    if (desc->capacity_rows) {
        pvs_status s = pvs_index_reserve_(ix, desc->capacity_rows);
        if (s != PVS_OK) {
            pvs_index_destroy(ix);
            *out = nullptr;
            return s;
        }
    }
    return PVS_OK;
}

pvs_status pvs_index_reserve_(pvs_index *ix, uint64_t rows) {
    const uint64_t cap = pvs_round_up(std::max<uint64_t>(rows, 1), PVS_ROW_ALIGN);
    if (cap <= ix->cap) return PVS_OK;
    if (cap > 0xfffffff0ull) return pvs_fail(PVS_ERR_UNSUPPORTED, "a shard holds at most 2^32-16 rows");
    uint8_t *rows_new = nullptr;
    float *norm_new = nullptr;
    int64_t *ids_new = nullptr;
    HIP_TRY(hipMalloc((void **)&rows_new, cap * (uint64_t)ix->stride));
    float *rnorm_new = nullptr;
    hipError_t e = hipMalloc((void **)&norm_new, cap * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&rnorm_new, cap * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&ids_new, cap * 8);
    if (e != hipSuccess) {
        hipFree(rows_new);
        hipFree(norm_new);
        hipFree(rnorm_new);
        return pvs_fail(PVS_ERR_OOM, "hipMalloc: %s", hipGetErrorString(e));
    }
    hipStream_t s = ix->admin_stream;
    // padding rows: zero payload, NaN norm (a NaN norm makes every scan compare fail).  The layout
    // is tiled by 32 rows, so the copy below moves whole tiles and the memset starts at a tile edge.
    const uint64_t n_tiled = pvs_round_up(ix->n, 32);
    HIP_TRY(hipMemsetAsync(rows_new + n_tiled * (uint64_t)ix->stride, 0, (cap - n_tiled) * (uint64_t)ix->stride, s));
    HIP_TRY(pvs_launch_fill_f32(norm_new + ix->n, cap - ix->n, __builtin_nanf(""), s));
    HIP_TRY(pvs_launch_fill_f32(rnorm_new + ix->n, cap - ix->n, __builtin_nanf(""), s));
    HIP_TRY(hipMemsetAsync(ids_new + ix->n, 0xff, (cap - ix->n) * 8, s));
    if (ix->n) {
        HIP_TRY(hipMemcpyAsync(rows_new, ix->d_rows, n_tiled * (uint64_t)ix->stride, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(norm_new, ix->d_norm2, ix->n * 4, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(rnorm_new, ix->d_rnorm, ix->n * 4, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(ids_new, ix->d_ids, ix->n * 8, hipMemcpyDeviceToDevice, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    hipFree(ix->d_rows);
    hipFree(ix->d_norm2);
    hipFree(ix->d_rnorm);
    hipFree(ix->d_ids);
    ix->d_rows = rows_new;
    ix->d_norm2 = norm_new;
    ix->d_rnorm = rnorm_new;
    ix->d_ids = ids_new;
    ix->cap = cap;
    return PVS_OK;
}

PVS_EXPORT void pvs_index_destroy(pvs_index *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    (void)hipDeviceSynchronize();
    for (auto &c : ix->ctx) ctx_release(c);
    hipFree(ix->d_rows);
    hipFree(ix->d_norm2);
    hipFree(ix->d_rnorm);
    hipFree(ix->d_ids);
    hipFree(ix->d_grp_off);
    hipFree(ix->d_grp_rows);
    hipFree(ix->d_grp_ids);
    pvs_group_work_release(ix->gwork);
    if (ix->admin_stream) hipStreamDestroy(ix->admin_stream);
    if (ix->search_stream) hipStreamDestroy(ix->search_stream);
    if (ix->comm_stream) hipStreamDestroy(ix->comm_stream);
    delete ix;
}

static pvs_status check_ids(pvs_index *ix, const int64_t *row_ids, uint64_t n, int64_t *last) {
    int64_t prev = ix->last_id;
    if (!row_ids) {
        int64_t first = ix->id_base + (int64_t)ix->n;
        if (ix->n && first <= prev) return pvs_fail(PVS_ERR_INVALID_ARG, "implicit row ids would not be increasing");
        *last = first + (int64_t)n - 1;
        return PVS_OK;
    }
    for (uint64_t i = 0; i < n; i++) {
        if (row_ids[i] <= prev && !(ix->n == 0 && i == 0 && prev == INT64_MIN))
            return pvs_fail(PVS_ERR_INVALID_ARG, "row_ids must be strictly increasing (row %llu: %lld after %lld)",
                            (unsigned long long)i, (long long)row_ids[i], (long long)prev);
        prev = row_ids[i];
    }
    *last = prev;
    return PVS_OK;
}

// rows_dev: [n][dim] dense device array of src_dtype (f32 when converting)
static pvs_status append_device_rows(pvs_index *ix, const void *rows_dev, bool from_f32, uint64_t n, const int64_t *row_ids,
                                     const int64_t *group_ids, int64_t last_id) {
    if (ix->n + n > ix->cap) PVS_TRY(pvs_index_reserve_(ix, std::max<uint64_t>(ix->n + n, ix->cap * 2)));
    hipStream_t s = ix->admin_stream;
    const int mode = (from_f32 && ix->dtype == PVS_I8) ? 0 : (from_f32 && ix->dtype == PVS_F16) ? 1 : 2;
    HIP_TRY(pvs_launch_rows_ingest(mode, rows_dev, ix->dim, ix->esz, ix->n, n, ix->scale, ix->d_rows, ix->stride, s));
    HIP_TRY(pvs_launch_norm2((int)ix->dtype, ix->d_rows, ix->stride, ix->dim, ix->n, n, ix->d_norm2, ix->d_rnorm, s));
    if (row_ids)
        HIP_TRY(hipMemcpyAsync(ix->d_ids + ix->n, row_ids, n * 8, hipMemcpyHostToDevice, s));
    else
        HIP_TRY(pvs_launch_iota_ids(ix->d_ids + ix->n, n, ix->id_base + (int64_t)ix->n, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (group_ids) {
        if (ix->h_groups.size() != ix->n) ix->h_groups.resize(ix->n, -1);
        ix->h_groups.insert(ix->h_groups.end(), group_ids, group_ids + n);
    } else if (!ix->h_groups.empty()) {
        ix->h_groups.resize(ix->n + n, -1);
    }
    ix->n += n;
    ix->last_id = last_id;
    ix->groups_built_n = UINT64_MAX;
    return PVS_OK;
}

static pvs_status add_impl(pvs_index *ix, const void *rows, bool from_f32, uint64_t n, const int64_t *row_ids,
                           const int64_t *group_ids, pvs_space space) {
    if (!ix) return pvs_fail(PVS_ERR_INVALID_ARG, "null index");
    if (n == 0) return PVS_OK;
    if (!rows) return pvs_fail(PVS_ERR_INVALID_ARG, "null rows");
    std::lock_guard<std::mutex> lk(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    if (from_f32 && ix->dtype == PVS_I8 && !ix->scale_set)
        return pvs_fail(PVS_ERR_STATE, "int8 index has no scale artifact: set it before adding f32 rows");
    int64_t last = 0;
    PVS_TRY(check_ids(ix, row_ids, n, &last));
    const size_t src_esz = from_f32 ? 4 : ix->esz;
    if (space == PVS_DEVICE) return append_device_rows(ix, rows, from_f32, n, row_ids, group_ids, last);
    // host rows: stage through HBM in chunks of <= 256 MiB
    const uint64_t row_bytes = (uint64_t)ix->dim * src_esz;
    const uint64_t chunk = std::max<uint64_t>(1, (256ull << 20) / row_bytes);
    void *stage = nullptr;
    HIP_TRY(hipMalloc(&stage, std::min(chunk, n) * row_bytes));
    pvs_status st = PVS_OK;
    if (ix->n + n > ix->cap) st = pvs_index_reserve_(ix, std::max<uint64_t>(ix->n + n, ix->cap * 2));
    for (uint64_t off = 0; off < n && st == PVS_OK; off += chunk) {
        const uint64_t m = std::min(chunk, n - off);
        hipError_t e = hipMemcpy(stage, (const uint8_t *)rows + off * row_bytes, m * row_bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            st = pvs_fail(PVS_ERR_DEVICE, "hipMemcpy H2D: %s", hipGetErrorString(e));
            break;
        }
        int64_t chunk_last = row_ids ? row_ids[off + m - 1] : ix->id_base + (int64_t)(ix->n + m) - 1;
        st = append_device_rows(ix, stage, from_f32, m, row_ids ? row_ids + off : nullptr, group_ids ? group_ids + off : nullptr,
                                chunk_last);
    }
    hipFree(stage);
    return st;
}

PVS_EXPORT pvs_status pvs_index_add(pvs_index *ix, const void *rows, uint64_t n, const int64_t *row_ids,
                                    const int64_t *group_ids, pvs_space space) {
    return add_impl(ix, rows, false, n, row_ids, group_ids, space);
}
PVS_EXPORT pvs_status pvs_index_add_f32(pvs_index *ix, const float *rows, uint64_t n, const int64_t *row_ids,
                                        const int64_t *group_ids, pvs_space space) {
    return add_impl(ix, rows, true, n, row_ids, group_ids, space);
}

PVS_EXPORT pvs_status pvs_index_set_scale(pvs_index *ix, float scale) {
    if (!ix) return pvs_fail(PVS_ERR_INVALID_ARG, "null index");
    if (ix->dtype != PVS_I8) return pvs_fail(PVS_ERR_INVALID_ARG, "only int8 indexes carry a scale artifact");
    // artifact_scale (db/vector_quants.rs:1456-1460): finite and > 0, nothing else
    if (!(std::isfinite(scale) && scale > 0.0f)) return pvs_fail(PVS_ERR_INVALID_ARG, "unusable scale artifact");
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->n && ix->scale_set && ix->scale != scale)
        return pvs_fail(PVS_ERR_STATE, "scale is frozen once rows exist (artifact_rev semantics): rebuild the index");
    ix->scale = scale;
    ix->scale_set = true;
    return PVS_OK;
}
PVS_EXPORT pvs_status pvs_index_set_scale_artifact(pvs_index *ix, const uint8_t *artifact, size_t len) {
    float s = 0.f;
    PVS_TRY(pvs_artifact_scale(artifact, len, &s));
    return pvs_index_set_scale(ix, s);
}

PVS_EXPORT pvs_status pvs_index_set_streams(pvs_index *ix, uint32_t n_streams) {
    if (!ix || n_streams == 0) return pvs_fail(PVS_ERR_INVALID_ARG, "bad stream count");
    PVS_TRY(pvs_sync(ix));
    ix->multi_stream = n_streams > 1;
    return PVS_OK;
}

PVS_EXPORT pvs_status pvs_index_set_path(pvs_index *ix, uint32_t path) {
    if (!ix || path > 2) return pvs_fail(PVS_ERR_INVALID_ARG, "bad path selector");
    ix->forced_path = path;
    return PVS_OK;
}

PVS_EXPORT pvs_status pvs_index_stats(pvs_index *ix, pvs_stats *out) {
    if (!ix || !out) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    pvs_stats s;
    memset(&s, 0, sizeof s);
    s.struct_size = sizeof s;
    s.dtype = ix->dtype;
    s.dim = ix->dim;
    s.rows = ix->n;
    s.capacity_rows = ix->cap;
    s.row_stride_bytes = ix->stride;
    s.hbm_bytes = ix->cap * ((uint64_t)ix->stride + 16);
    s.scale = ix->scale_set ? ix->scale : 0.f;
    s.searches = ix->searches.load();
    s.fast_queries = ix->fast_queries.load();
    s.dense_queries = ix->dense_queries.load();
    s.last_candidates = ix->last_candidates.load();
    *out = s;
    return PVS_OK;
}

PVS_EXPORT pvs_status pvs_device_synchronize(int32_t device) {
    int dev = 0;
    PVS_TRY(use_device(device, &dev));
    HIP_TRY(hipDeviceSynchronize());
    return PVS_OK;
}

PVS_EXPORT pvs_status pvs_index_read_ids(pvs_index *ix, uint64_t row0, uint64_t n, int64_t *out_row_ids, int64_t *out_group_ids) {
    if (!ix || (n && !out_row_ids)) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (row0 + n > ix->n) return pvs_fail(PVS_ERR_INVALID_ARG, "row range [%llu, %llu) exceeds %llu rows", (unsigned long long)row0,
                                          (unsigned long long)(row0 + n), (unsigned long long)ix->n);
    if (n == 0) return PVS_OK;
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipMemcpy(out_row_ids, ix->d_ids + row0, n * 8, hipMemcpyDeviceToHost));
    if (out_group_ids) {
        std::lock_guard<std::mutex> lk(ix->mu);
        if (ix->h_groups.empty())
            memcpy(out_group_ids, out_row_ids, n * 8);
        else if (ix->h_groups.size() < row0 + n)
            return pvs_fail(PVS_ERR_STATE, "group ids missing for some rows");
        else
            memcpy(out_group_ids, ix->h_groups.data() + row0, n * 8);
    }
    return PVS_OK;
}

PVS_EXPORT pvs_status pvs_index_read_rows(pvs_index *ix, uint64_t row0, uint64_t n, void *out_host) {
    if (!ix || (n && !out_host)) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (row0 + n > ix->n) return pvs_fail(PVS_ERR_INVALID_ARG, "row range [%llu, %llu) exceeds %llu rows", (unsigned long long)row0,
                                          (unsigned long long)(row0 + n), (unsigned long long)ix->n);
    if (n == 0) return PVS_OK;
    HIP_TRY(hipSetDevice(ix->device));
    const size_t w = (size_t)ix->dim * ix->esz;
    const uint64_t chunk = std::max<uint64_t>(1, (256ull << 20) / w);
    uint8_t *stage = nullptr;
    HIP_TRY(hipMalloc((void **)&stage, std::min(chunk, n) * w));
    pvs_status st = PVS_OK;
    for (uint64_t off = 0; off < n; off += chunk) {
        const uint64_t m = std::min(chunk, n - off);
        hipError_t e = pvs_launch_rows_gather(ix->d_rows, ix->stride, (uint32_t)w, row0 + off, m, stage, nullptr);
        if (e == hipSuccess) e = hipMemcpy((uint8_t *)out_host + off * w, stage, m * w, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            st = pvs_fail(PVS_ERR_DEVICE, "read_rows: %s", hipGetErrorString(e));
            break;
        }
    }
    hipFree(stage);
    return st;
}

PVS_EXPORT pvs_status pvs_index_set_profiling(pvs_index *ix, int32_t enable) {
    if (!ix) return pvs_fail(PVS_ERR_INVALID_ARG, "null index");
    ix->profiling = enable != 0;
    return PVS_OK;
}
PVS_EXPORT pvs_status pvs_index_get_profile(pvs_index *ix, pvs_profile *out, int32_t reset) {
    if (!ix || !out) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(ix->prof_mu);
    *out = ix->prof;
    out->struct_size = sizeof(pvs_profile);
    if (reset) ix->prof = pvs_profile{};
    return PVS_OK;
}

// ------------------------------------------------------------------- search
static pvs_status validate_search(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k,
                                  pvs_metric metric) {
    if (!ix) return pvs_fail(PVS_ERR_INVALID_ARG, "null index");
    if (batch && !queries) return pvs_fail(PVS_ERR_INVALID_ARG, "null queries");
    if (k < 1) return pvs_fail(PVS_ERR_INVALID_ARG, "k must be a positive integer");  // preprocess.rs:441-444
    if (k > (1u << 20)) return pvs_fail(PVS_ERR_INVALID_ARG, "k too large");
    if (metric != PVS_COSINE && metric != PVS_L2) return pvs_fail(PVS_ERR_INVALID_ARG, "unknown metric");
    if (qdtype == PVS_I8) {
        if (ix->dtype != PVS_I8) return pvs_fail(PVS_ERR_DIM_MISMATCH, "int8 query against a float index (element type mismatch)");
    } else if (qdtype == PVS_F32) {
        if (ix->dtype == PVS_I8 && !ix->scale_set)
            return pvs_fail(PVS_ERR_STATE, "f32 query on an int8 index needs the scale artifact");
    } else {
        return pvs_fail(PVS_ERR_INVALID_ARG, "queries must be f32 or int8");
    }
    return PVS_OK;
}

static bool fast_path_ok(const pvs_index *ix, uint32_t k) {
    if (ix->forced_path == 1) return false;
    if (!pvs_scan_supported((int)ix->dtype, ix->stride / PVS_KSLAB_BYTES)) return false;
    if (k > PVS_MAX_K) return false;
    return ix->n > 0;
}

// one query through the dense path; q is the query's index inside the current chunk
static pvs_status dense_one(pvs_index *ix, SearchCtx &c, uint32_t q, uint32_t k, int metric, int64_t *out_ids, float *out_dist,
                            uint32_t *out_count) {
    PVS_TRY(pvs_dense_reserve(c.dense, ix->n));
    const uint8_t *qe = c.d_qexact + (size_t)q * ix->dim * (ix->dtype == PVS_I8 ? 1 : 4);
    HIP_TRY(pvs_launch_dense_exact((int)ix->dtype, metric, ix->d_rows, ix->stride, ix->dim, ix->n, ix->d_norm2, qe, c.d_qinfo + q, 1,
                                   c.d_qpad, c.dense.d_dist, 1, 0, (uint32_t)ix->n_cu, c.stream));
    PVS_TRY(pvs_dense_topk(c.dense, ix->n, k, ix->d_ids, out_ids, out_dist, out_count, c.stream, c.cur_mask));
    ix->dense_queries++;
    return PVS_OK;
}

static pvs_status prep_chunk(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t qoff, uint32_t nb,
                             uint32_t batch_pad, int metric) {
    const size_t qesz = qdtype == PVS_I8 ? 1 : 4;
    const uint8_t *qsrc = (const uint8_t *)d_queries + (size_t)qoff * ix->dim * qesz;
    HIP_TRY(pvs_launch_prep_queries((int)ix->dtype, qdtype, qsrc, nb, batch_pad, ix->dim, ix->stride, ix->scale, metric, c.d_qmat,
                                    c.d_qexact, c.d_qinfo, c.d_cand_cnt, c.d_need_dense + qoff, c.stream));
    return PVS_OK;
}

// Enqueues the whole search on c.stream.  Outputs are device buffers.
static pvs_status search_enqueue(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t batch, uint32_t k,
                                 int metric, int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count, bool *used_fast) {
    const bool fast = fast_path_ok(ix, k);
    *used_fast = fast;
    if (!fast && ix->forced_path == 2) return pvs_fail(PVS_ERR_UNSUPPORTED, "filter-scan path not available for this index / k");
    if (ix->n == 0) {
        HIP_TRY(hipMemsetAsync(c.d_need_dense, 0, 4 * (size_t)batch, c.stream));
        HIP_TRY(hipMemsetAsync(d_out_count, 0, 4 * (size_t)batch, c.stream));
        HIP_TRY(hipMemsetAsync(d_out_ids, 0xff, 8 * (size_t)batch * k, c.stream));
        HIP_TRY(pvs_launch_fill_f32(d_out_dist, (uint64_t)batch * k, __builtin_nanf(""), c.stream));
        HIP_TRY(hipEventRecord(c.done, c.stream));
        return PVS_OK;
    }
    const uint32_t pass_max = fast ? pvs_scan_max_batch((int)ix->dtype, ix->stride / PVS_KSLAB_BYTES) : PVS_MAX_BATCH;
    for (uint32_t qoff = 0; qoff < batch; qoff += pass_max) {
        const uint32_t nb = std::min(pass_max, batch - qoff);
        const uint32_t batch_pad = nb <= 32 ? 32 : nb <= 64 ? 64 : nb <= 128 ? 128 : 256;
        PVS_TRY(prep_chunk(ix, c, d_queries, qdtype, qoff, nb, batch_pad, metric));
        int64_t *oid = d_out_ids + (size_t)qoff * k;
        float *od = d_out_dist + (size_t)qoff * k;
        uint32_t *oc = d_out_count + qoff;
        if (!fast) {
            for (uint32_t q = 0; q < nb; q++) PVS_TRY(dense_one(ix, c, q, k, metric, oid + (size_t)q * k, od + (size_t)q * k, oc + q));
            continue;
        }
        ScanArgs a;
        a.dtype = (int)ix->dtype;
        a.metric = metric;
        a.kslabs = ix->stride / PVS_KSLAB_BYTES;
        a.qgroups = batch_pad / 32;
        a.rows = ix->d_rows;
        a.aux = metric == PVS_COSINE ? ix->d_rnorm : ix->d_norm2;
        if (c.cur_mask) {  // filtered search: rows outside the mask stream a NaN scalar and never pass
            HIP_TRY(pvs_launch_mask_aux(a.aux, c.cur_mask, ix->n, ix->cap, c.d_aux_masked, c.stream));
            a.aux = c.d_aux_masked;
        }
        a.stride = ix->stride;
        a.n_rows = ix->n;
        a.qmat = c.d_qmat;
        a.qinfo = c.d_qinfo;
        a.thr = c.d_thr;
        a.cand_cnt = c.d_cand_cnt;
        a.cand = c.d_cand;
        a.cand_cap = PVS_CAND_CAP;
        a.gmin = c.d_gmin;
        const uint32_t wg_rows = pvs_scan_wg_rows(a.qgroups);
        const uint32_t n_wgtiles = (uint32_t)((ix->n + wg_rows - 1) / wg_rows);
        // pass A: strided sample of row tiles -> group minima -> threshold
        // Sample size.  Per wave-tile (32 rows x 32 queries) pass B expects 1024*k/n_sample emitted
        // candidates, and each emit costs a few hundred cycles, while pass A costs ~ n_sample/N of a
        // scan.  Measured on MI355X (10Mx768 int8 x128 and 1Mx768 f16 x32): 1/16 beats 1/8, 1/32, 1/64;
        // a handful of queries emit so little that 1/64 is enough.  The candidate list of a query
        // holds ~k/frac rows: keep that 2.5x below its capacity.
        double frac = nb <= 4 ? 1.0 / 64.0 : 1.0 / 16.0;
        static const double frac_env = getenv("PVS_SAMPLE_DIV") ? 1.0 / atof(getenv("PVS_SAMPLE_DIV")) : 0.0;  // tuning experiments
        if (frac_env > 0.0) frac = frac_env;
        frac = std::min(0.5, std::max(frac, 2.5 * (double)k / (double)PVS_CAND_CAP));
        const uint64_t target_rows = std::min<uint64_t>(ix->n, std::max<uint64_t>((uint64_t)((double)ix->n * frac), 32768));
        const uint32_t want_tiles = (uint32_t)std::max<uint64_t>(1, (target_rows + wg_rows - 1) / wg_rows);
        a.tile_step = std::max<uint32_t>(1, n_wgtiles / want_tiles);
        const uint32_t n_samp = (n_wgtiles + a.tile_step - 1) / a.tile_step;
        const uint32_t per_cu_a = (a.qgroups == 1 || a.qgroups == 8 || a.kslabs > 4) ? 1 : 2;
        const uint32_t rt = a.qgroups >= 4 ? 1 : 4 / a.qgroups;  // row sub-tiles per workgroup
        a.grid = std::min<uint32_t>({n_samp, (uint32_t)ix->n_cu * per_cu_a, GMAX / (rt * 32)});
        a.mode = 0;
        // row groups per query: >= 16k keeps the threshold within ~3 % of the finest partition (two of the
        // k best rows rarely share a group) and >= 1024; each lane can supply 1..16
        a.gmin_per_lane = 16;
        while (a.gmin_per_lane > 1 && (uint64_t)a.grid * rt * 2 * (a.gmin_per_lane / 2) >= std::max<uint64_t>(16ull * k, 1024)) a.gmin_per_lane /= 2;
        a.groups_per_query = a.grid * rt * 2 * a.gmin_per_lane;
        span_begin(ix, c, 0, (uint64_t)n_samp * wg_rows);
        HIP_TRY(pvs_launch_scan(a, c.stream));
        span_end(ix, c);
        HIP_TRY(pvs_launch_kth(c.d_gmin, a.groups_per_query, nb, k, c.d_thr, c.stream));
        // pass B: every row once (candidate counters were zeroed by the prep kernel)
        a.mode = 1;
        a.tile_step = 1;
        const uint32_t per_cu = (a.qgroups == 1 || a.qgroups == 8 || a.kslabs > 4) ? 1 : 2;
        a.grid = std::min<uint32_t>(n_wgtiles, (uint32_t)ix->n_cu * per_cu);
        span_begin(ix, c, 1, ix->n);
        HIP_TRY(pvs_launch_scan(a, c.stream));
        span_end(ix, c);
        // pass C
        FinalizeArgs f;
        f.dtype = (int)ix->dtype;
        f.metric = metric;
        f.rows = ix->d_rows;
        f.norm2 = ix->d_norm2;
        f.ids = ix->d_ids;
        f.stride = ix->stride;
        f.dim = ix->dim;
        f.n_rows = ix->n;
        f.qexact = c.d_qexact;
        f.qinfo = c.d_qinfo;
        f.cand_cnt = c.d_cand_cnt;
        f.cand = c.d_cand;
        f.cand_cap = PVS_CAND_CAP;
        f.batch = nb;
        f.k = k;
        f.out_ids = oid;
        f.out_dist = od;
        f.out_count = oc;
        f.need_dense = c.d_need_dense + qoff;
        span_begin(ix, c, 2, 0);
        HIP_TRY(pvs_launch_finalize(f, c.stream));
        span_end(ix, c);
    }
    if (fast) HIP_TRY(hipMemcpyAsync(c.h_need_dense, c.d_need_dense, 4 * (size_t)batch, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipEventRecord(c.done, c.stream));
    return PVS_OK;
}

// After the stream drained: answer the queries the filter path handed back.
static pvs_status search_fallbacks(pvs_index *ix, SearchCtx &c, const void *d_queries, int qdtype, uint32_t batch, uint32_t k,
                                   int metric, int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count) {
    uint32_t n_dense = 0;
    for (uint32_t q = 0; q < batch; q++) n_dense += c.h_need_dense[q] ? 1 : 0;
    ix->fast_queries += batch - n_dense;
    if (!n_dense) return PVS_OK;
    if (ix->forced_path == 2) return pvs_fail(PVS_ERR_UNSUPPORTED, "%u queries need the dense path but path=2 forbids it", n_dense);
    for (uint32_t qoff = 0; qoff < batch; qoff += PVS_MAX_BATCH) {
        const uint32_t nb = std::min(PVS_MAX_BATCH, batch - qoff);
        bool any = false;
        for (uint32_t q = 0; q < nb; q++) any |= c.h_need_dense[qoff + q] != 0;
        if (!any) continue;
        PVS_TRY(prep_chunk(ix, c, d_queries, qdtype, qoff, nb, 32 * ((nb + 31) / 32), metric));
        for (uint32_t q = 0; q < nb; q++) {
            if (!c.h_need_dense[qoff + q]) continue;
            PVS_TRY(dense_one(ix, c, q, k, metric, d_out_ids + (size_t)(qoff + q) * k, d_out_dist + (size_t)(qoff + q) * k,
                              d_out_count + qoff + q));
        }
    }
    HIP_TRY(hipStreamSynchronize(c.stream));
    return PVS_OK;
}

static SearchCtx *ctx_acquire(pvs_index *ix, uint32_t *ticket) {
    for (;;) {
        {
            std::lock_guard<std::mutex> lk(ix->mu);
            for (uint32_t i = 0; i < NCTX; i++)
                if (!ix->ctx[i].busy) {
                    ix->ctx[i].busy = true;
                    *ticket = i;
                    return &ix->ctx[i];
                }
        }
        sched_yield();
    }
}
static void ctx_done(pvs_index *ix, SearchCtx *c) {
    std::lock_guard<std::mutex> lk(ix->mu);
    c->pending = false;
    c->busy = false;
}

static pvs_status search_host(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                              const uint8_t *mask, pvs_space mask_space, int64_t *out_ids, float *out_dist, uint32_t *out_count);

PVS_EXPORT pvs_status pvs_search(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k,
                                 pvs_metric metric, int64_t *out_ids, float *out_dist, uint32_t *out_count) {
    return search_host(ix, queries, qdtype, batch, k, metric, nullptr, PVS_HOST, out_ids, out_dist, out_count);
}

PVS_EXPORT pvs_status pvs_search_filtered(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k,
                                          pvs_metric metric, const uint8_t *allowed_rows, pvs_space mask_space, int64_t *out_ids,
                                          float *out_dist, uint32_t *out_count) {
    if (!allowed_rows) return pvs_fail(PVS_ERR_INVALID_ARG, "null candidate mask");
    return search_host(ix, queries, qdtype, batch, k, metric, allowed_rows, mask_space, out_ids, out_dist, out_count);
}

static pvs_status search_host(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, uint32_t k, pvs_metric metric,
                              const uint8_t *mask, pvs_space mask_space, int64_t *out_ids, float *out_dist, uint32_t *out_count) {
    PVS_TRY(validate_search(ix, queries, qdtype, batch, k, metric));
    if (!out_ids || !out_dist || !out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null output");
    if (batch == 0) return PVS_OK;
    HIP_TRY(hipSetDevice(ix->device));
    uint32_t t;
    SearchCtx *c = ctx_acquire(ix, &t);
    pvs_status st = ctx_prepare(ix, *c, batch, k, true);
    const size_t qbytes = (size_t)ix->dim * (qdtype == PVS_I8 ? 1 : 4);
    void *d_q = nullptr;
    if (st == PVS_OK) {
        // (hipMalloc/hipFree per call would cost ~0.1 ms and hipFree synchronises the whole device,
        // stalling the other host threads' searches)
        if (qbytes * batch > c->qstage_cap) {
            hipFree(c->d_qstage);
            c->d_qstage = nullptr;
            c->qstage_cap = 0;
            const size_t cap = pvs_round_up(qbytes * batch, 1 << 16);
            hipError_t e = hipMalloc(&c->d_qstage, cap);
            if (e != hipSuccess)
                st = pvs_fail(PVS_ERR_OOM, "hipMalloc queries: %s", hipGetErrorString(e));
            else
                c->qstage_cap = cap;
        }
        d_q = c->d_qstage;
    }
    bool fast = false;
    if (st == PVS_OK) {
        hipError_t e = hipMemcpyAsync(d_q, queries, qbytes * batch, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) st = pvs_fail(PVS_ERR_DEVICE, "H2D queries: %s", hipGetErrorString(e));
    }
    if (st == PVS_OK && mask && ix->n) {
        auto setup = [&]() -> pvs_status {
            if (ix->cap > c->mask_cap) {
                hipFree(c->d_mask);
                hipFree(c->d_aux_masked);
                c->d_mask = nullptr;
                c->d_aux_masked = nullptr;
                c->mask_cap = 0;
                HIP_TRY(hipMalloc((void **)&c->d_mask, ix->cap));
                HIP_TRY(hipMalloc((void **)&c->d_aux_masked, ix->cap * 4));
                c->mask_cap = ix->cap;
            }
            if (mask_space == PVS_HOST) {
                HIP_TRY(hipMemcpyAsync(c->d_mask, mask, ix->n, hipMemcpyHostToDevice, c->stream));
                c->cur_mask = c->d_mask;
            } else {
                c->cur_mask = mask;
            }
            return PVS_OK;
        };
        st = setup();
    }
    if (st == PVS_OK) st = search_enqueue(ix, *c, d_q, qdtype, batch, k, metric, c->d_out_ids, c->d_out_dist, c->d_out_count, &fast);
    if (st == PVS_OK) {
        hipError_t e = hipEventSynchronize(c->done);
        if (e != hipSuccess) st = pvs_fail(PVS_ERR_DEVICE, "search failed on device: %s", hipGetErrorString(e));
    }
    if (st == PVS_OK) spans_collect(ix, *c);
    if (st == PVS_OK && fast && ix->n)
        st = search_fallbacks(ix, *c, d_q, qdtype, batch, k, metric, c->d_out_ids, c->d_out_dist, c->d_out_count);
    if (st == PVS_OK) {
        hipError_t e = hipMemcpyAsync(out_ids, c->d_out_ids, 8 * (size_t)batch * k, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(out_dist, c->d_out_dist, 4 * (size_t)batch * k, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(out_count, c->d_out_count, 4 * (size_t)batch, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) st = pvs_fail(PVS_ERR_DEVICE, "D2H results: %s", hipGetErrorString(e));
    }
    ix->searches++;
    ctx_done(ix, c);
    return st;
}

PVS_EXPORT pvs_status pvs_search_device(pvs_index *ix, const void *d_queries, pvs_dtype qdtype, uint32_t batch, uint32_t k,
                                        pvs_metric metric, int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count,
                                        uint32_t *out_ticket) {
    PVS_TRY(validate_search(ix, d_queries, qdtype, batch, k, metric));
    if (!d_out_ids || !d_out_dist || !d_out_count || !out_ticket) return pvs_fail(PVS_ERR_INVALID_ARG, "null output");
    if (batch == 0) return pvs_fail(PVS_ERR_INVALID_ARG, "empty batch");
    HIP_TRY(hipSetDevice(ix->device));
    uint32_t t;
    SearchCtx *c = ctx_acquire(ix, &t);
    pvs_status st = ctx_prepare(ix, *c, batch, k, false);
    bool fast = false;
    if (st == PVS_OK) st = search_enqueue(ix, *c, d_queries, qdtype, batch, k, metric, d_out_ids, d_out_dist, d_out_count, &fast);
    if (st != PVS_OK) {
        (void)hipStreamSynchronize(c->stream);
        ctx_done(ix, c);
        return st;
    }
    c->pending = true;
    c->p_queries = d_queries;
    c->p_qdtype = qdtype;
    c->p_metric = metric;
    c->p_batch = batch;
    c->p_k = k;
    c->p_out_ids = d_out_ids;
    c->p_out_dist = d_out_dist;
    c->p_out_count = d_out_count;
    c->p_fast = fast;
    ix->searches++;
    *out_ticket = t;
    return PVS_OK;
}

PVS_EXPORT pvs_status pvs_wait(pvs_index *ix, uint32_t ticket) {
    if (!ix || ticket >= NCTX) return pvs_fail(PVS_ERR_INVALID_ARG, "bad ticket");
    SearchCtx *c = &ix->ctx[ticket];
    if (!c->busy || !c->pending) return pvs_fail(PVS_ERR_STATE, "ticket %u has no search in flight", ticket);
    HIP_TRY(hipSetDevice(ix->device));
    pvs_status st = PVS_OK;
    hipError_t e = hipEventSynchronize(c->done);
    if (e != hipSuccess) st = pvs_fail(PVS_ERR_DEVICE, "search failed on device: %s", hipGetErrorString(e));
    if (st == PVS_OK) spans_collect(ix, *c);
    if (st == PVS_OK && c->p_comm) {
        // every rank sees the same gathered flags, so they all agree on whether to redo
        bool redo = false;
        for (uint64_t i = 0; i < (uint64_t)c->sh_world * c->p_batch; i++) redo |= c->h_all_flags[i] != 0;
        if (!redo) {
            ix->fast_queries += c->p_fast ? c->p_batch : 0;
        } else {
            if (c->p_fast && ix->n)
                st = search_fallbacks(ix, *c, c->p_queries, c->p_qdtype, c->p_batch, c->p_k, c->p_metric, c->d_loc_ids, c->d_loc_dist,
                                      c->d_loc_cnt);
            // (search_fallbacks drained c->stream; the redo's collective goes where all the others go)
            hipStream_t cs = ix->multi_stream ? ix->comm_stream : c->stream;
            if (st == PVS_OK) {
                hipError_t e2 = hipMemsetAsync(c->d_need_dense, 0, 4 * (size_t)c->p_batch, cs);
                if (e2 != hipSuccess) st = pvs_fail(PVS_ERR_DEVICE, "memset: %s", hipGetErrorString(e2));
            }
            if (st == PVS_OK)
                st = pvs_comm_gather_pages_(c->p_comm, c->d_loc_ids, c->d_loc_dist, c->d_loc_cnt, c->d_need_dense, c->d_all_ids,
                                            c->d_all_dist, c->d_all_cnt, c->d_all_flags, (uint64_t)c->p_batch * c->p_k, c->p_batch, cs);
            if (st == PVS_OK) {
                hipError_t e2 = pvs_launch_merge(c->d_all_ids, c->d_all_dist, c->d_all_cnt, c->sh_world, c->p_batch, c->p_k,
                                                 c->p_final_ids, c->p_final_dist, c->p_final_count, cs);
                if (e2 == hipSuccess) e2 = hipStreamSynchronize(cs);
                if (e2 != hipSuccess) st = pvs_fail(PVS_ERR_DEVICE, "sharded redo: %s", hipGetErrorString(e2));
            }
        }
        c->p_comm = nullptr;
    } else if (st == PVS_OK && c->p_fast && ix->n) {
        st = search_fallbacks(ix, *c, c->p_queries, c->p_qdtype, c->p_batch, c->p_k, c->p_metric, c->p_out_ids, c->p_out_dist,
                              c->p_out_count);
    }
    ctx_done(ix, c);
    return st;
}

PVS_EXPORT pvs_status pvs_search_sharded_async(pvs_index *ix, pvs_comm *comm, const void *d_queries, pvs_dtype qdtype, uint32_t batch,
                                               uint32_t k, pvs_metric metric, int64_t *d_out_ids, float *d_out_dist,
                                               uint32_t *d_out_count, uint32_t *out_ticket) {
    PVS_TRY(validate_search(ix, d_queries, qdtype, batch, k, metric));
    if (!comm || !d_out_ids || !d_out_dist || !d_out_count || !out_ticket) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (batch == 0) return pvs_fail(PVS_ERR_INVALID_ARG, "empty batch");
    if (pvs_comm_device_(comm) != ix->device) return pvs_fail(PVS_ERR_INVALID_ARG, "index and communicator live on different devices");
    HIP_TRY(hipSetDevice(ix->device));
    const uint32_t world = (uint32_t)pvs_comm_world_(comm);
    uint32_t t;
    SearchCtx *c = ctx_acquire(ix, &t);
    auto body = [&]() -> pvs_status {
        PVS_TRY(ctx_prepare(ix, *c, batch, k, false));
        const uint64_t elems = (uint64_t)batch * k;
        if (elems > c->sh_elems || batch > c->sh_batch || world != c->sh_world) {
            hipFree(c->d_loc_ids);
            hipFree(c->d_all_ids);
            hipFree(c->d_loc_dist);
            hipFree(c->d_all_dist);
            hipFree(c->d_loc_cnt);
            hipFree(c->d_all_cnt);
            hipFree(c->d_all_flags);
            if (c->h_all_flags) hipHostFree(c->h_all_flags);
            c->d_loc_ids = c->d_all_ids = nullptr;
            c->d_loc_dist = c->d_all_dist = nullptr;
            c->d_loc_cnt = c->d_all_cnt = c->d_all_flags = c->h_all_flags = nullptr;
            c->sh_elems = 0;
            HIP_TRY(hipMalloc((void **)&c->d_loc_ids, elems * 8));
            HIP_TRY(hipMalloc((void **)&c->d_loc_dist, elems * 4));
            HIP_TRY(hipMalloc((void **)&c->d_loc_cnt, (size_t)batch * 4));
            HIP_TRY(hipMalloc((void **)&c->d_all_ids, elems * 8 * world));
            HIP_TRY(hipMalloc((void **)&c->d_all_dist, elems * 4 * world));
            HIP_TRY(hipMalloc((void **)&c->d_all_cnt, (size_t)batch * 4 * world));
            HIP_TRY(hipMalloc((void **)&c->d_all_flags, (size_t)batch * 4 * world));
            HIP_TRY(hipHostMalloc((void **)&c->h_all_flags, (size_t)batch * 4 * world, hipHostMallocDefault));
            c->sh_elems = elems;
            c->sh_batch = batch;
            c->sh_world = world;
        }
        bool fast = false;
        // 1. this shard's page (row ids in the index are global ids)
        PVS_TRY(search_enqueue(ix, *c, d_queries, qdtype, batch, k, metric, c->d_loc_ids, c->d_loc_dist, c->d_loc_cnt, &fast));
        // 2. one grouped all-gather over xGMI, 3. merge on every rank — stream-ordered, no host sync.
        // With one stream per context (pvs_index_set_streams) the local scans of several searches
        // overlap, but their collectives still go out on ONE stream in program order: a communicator
        // is never driven from two streams at once.
        hipStream_t cs = c->stream;
        if (ix->multi_stream) {
            cs = ix->comm_stream;
            HIP_TRY(hipStreamWaitEvent(cs, c->done, 0));  // c->done was just recorded behind the local search
        }
        PVS_TRY(pvs_comm_gather_pages_(comm, c->d_loc_ids, c->d_loc_dist, c->d_loc_cnt, c->d_need_dense, c->d_all_ids, c->d_all_dist,
                                       c->d_all_cnt, c->d_all_flags, elems, batch, cs));
        HIP_TRY(pvs_launch_merge(c->d_all_ids, c->d_all_dist, c->d_all_cnt, world, batch, k, d_out_ids, d_out_dist, d_out_count, cs));
        HIP_TRY(hipMemcpyAsync(c->h_all_flags, c->d_all_flags, (size_t)batch * 4 * world, hipMemcpyDeviceToHost, cs));
        HIP_TRY(hipEventRecord(c->done, cs));
        c->pending = true;
        c->p_comm = comm;
        c->p_queries = d_queries;
        c->p_qdtype = qdtype;
        c->p_metric = metric;
        c->p_batch = batch;
        c->p_k = k;
        c->p_out_ids = c->d_loc_ids;
        c->p_out_dist = c->d_loc_dist;
        c->p_out_count = c->d_loc_cnt;
        c->p_final_ids = d_out_ids;
        c->p_final_dist = d_out_dist;
        c->p_final_count = d_out_count;
        c->p_fast = fast;
        return PVS_OK;
    };
    pvs_status st = body();
    if (st != PVS_OK) {
        (void)hipStreamSynchronize(c->stream);
        ctx_done(ix, c);
        return st;
    }
    ix->searches++;
    *out_ticket = t;
    return PVS_OK;
}

PVS_EXPORT pvs_status pvs_search_sharded(pvs_index *ix, pvs_comm *comm, const void *d_queries, pvs_dtype qdtype, uint32_t batch, uint32_t k,
                                         pvs_metric metric, int64_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count) {
    uint32_t t = 0;
    PVS_TRY(pvs_search_sharded_async(ix, comm, d_queries, qdtype, batch, k, metric, d_out_ids, d_out_dist, d_out_count, &t));
    return pvs_wait(ix, t);
}

PVS_EXPORT pvs_status pvs_sync(pvs_index *ix) {
    if (!ix) return pvs_fail(PVS_ERR_INVALID_ARG, "null index");
    pvs_status st = PVS_OK;
    for (uint32_t i = 0; i < NCTX; i++)
        if (ix->ctx[i].busy && ix->ctx[i].pending) {
            pvs_status s = pvs_wait(ix, i);
            if (s != PVS_OK) st = s;
        }
    return st;
}

PVS_EXPORT pvs_status pvs_score_all(pvs_index *ix, const void *query, pvs_dtype qdtype, pvs_metric metric, float *out_dist,
                                    pvs_space out_space) {
    PVS_TRY(validate_search(ix, query, qdtype, 1, 1, metric));
    if (!out_dist) return pvs_fail(PVS_ERR_INVALID_ARG, "null output");
    if (ix->n == 0) return PVS_OK;
    HIP_TRY(hipSetDevice(ix->device));
    uint32_t t;
    SearchCtx *c = ctx_acquire(ix, &t);
    pvs_status st = ctx_prepare(ix, *c, 1, 1, false);
    auto body = [&]() -> pvs_status {
        const size_t qbytes = (size_t)ix->dim * (qdtype == PVS_I8 ? 1 : 4);
        HIP_TRY(hipMemcpyAsync(c->d_qin, query, qbytes, hipMemcpyHostToDevice, c->stream));
        PVS_TRY(prep_chunk(ix, *c, c->d_qin, qdtype, 0, 1, 32, metric));
        float *dst = out_dist;
        if (out_space == PVS_HOST) {
            PVS_TRY(pvs_dense_reserve(c->dense, ix->n));
            dst = c->dense.d_dist;
        }
        HIP_TRY(pvs_launch_dense_exact((int)ix->dtype, metric, ix->d_rows, ix->stride, ix->dim, ix->n, ix->d_norm2, c->d_qexact,
                                       c->d_qinfo, 1, c->d_qpad, dst, 1, 0, (uint32_t)ix->n_cu, c->stream));
        if (out_space == PVS_HOST) HIP_TRY(hipMemcpyAsync(out_dist, dst, ix->n * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return PVS_OK;
    };
    if (st == PVS_OK) st = body();
    ctx_done(ix, c);
    return st;
}

// ------------------------------------------------- groups, dense scores, similar_to
static pvs_status ensure_groups(pvs_index *ix) {
    if (ix->groups_built_n == ix->n) return PVS_OK;
    if (!ix->h_groups.empty() && ix->h_groups.size() != ix->n) return pvs_fail(PVS_ERR_STATE, "group ids missing for some rows");
    hipFree(ix->d_grp_off);
    hipFree(ix->d_grp_rows);
    hipFree(ix->d_grp_ids);
    ix->d_grp_off = ix->d_grp_rows = nullptr;
    ix->d_grp_ids = nullptr;
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
    HIP_TRY(hipMalloc((void **)&ix->d_grp_off, (off.size() + 1) * 4));
    HIP_TRY(hipMalloc((void **)&ix->d_grp_rows, (n + 1) * 4));
    HIP_TRY(hipMalloc((void **)&ix->d_grp_ids, (gids.size() + 1) * 8));
    HIP_TRY(hipMemcpy(ix->d_grp_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    if (n) HIP_TRY(hipMemcpy(ix->d_grp_rows, rows.data(), n * 4, hipMemcpyHostToDevice));
    if (!gids.empty()) HIP_TRY(hipMemcpy(ix->d_grp_ids, gids.data(), gids.size() * 8, hipMemcpyHostToDevice));
    ix->groups_built_n = n;
    return PVS_OK;
}

// d_out[row * nb + q], nb <= PVS_MAX_BATCH queries already prepared in ctx c (prep_chunk)
static pvs_status dense_chunk(pvs_index *ix, SearchCtx &c, uint32_t nb, uint32_t batch_pad, int metric, float *d_out) {
    const uint32_t kslabs = ix->stride / PVS_KSLAB_BYTES;
    if (ix->dtype == PVS_I8 && pvs_scan_supported(PVS_I8, kslabs) && (uint64_t)ix->dim * 127 * 127 < (1u << 24)) {
        // matrix-core path: exact integer dots, closed-form finish (valid below 2^24)
        ScanArgs a;
        a.dtype = PVS_I8;
        a.metric = metric;
        a.kslabs = kslabs;
        a.qgroups = batch_pad / 32;
        a.rows = ix->d_rows;
        a.aux = ix->d_norm2;
        a.stride = ix->stride;
        a.n_rows = ix->n;
        a.qmat = c.d_qmat;
        a.qinfo = c.d_qinfo;
        a.thr = c.d_thr;
        a.gmin = c.d_gmin;
        a.groups_per_query = 0;
        a.cand_cnt = c.d_cand_cnt;
        a.cand = c.d_cand;
        a.cand_cap = PVS_CAND_CAP;
        a.mode = 2;
        a.tile_step = 1;
        const uint32_t wg_rows = pvs_scan_wg_rows(a.qgroups);
        const uint32_t n_wgtiles = (uint32_t)((ix->n + wg_rows - 1) / wg_rows);
        const uint32_t per_cu = (a.qgroups == 1 || a.kslabs > 4) ? 1 : 2;
        a.grid = std::min<uint32_t>(n_wgtiles, (uint32_t)ix->n_cu * per_cu);
        a.dense_out = d_out;
        a.dense_ld = nb;
        a.batch = nb;
        a.dense_flag = c.d_cand_cnt;  // reused as the out-of-range flag word
        HIP_TRY(hipMemsetAsync(c.d_cand_cnt, 0, 4, c.stream));
        HIP_TRY(pvs_launch_scan(a, c.stream));
        uint32_t flag = 0;
        HIP_TRY(hipMemcpyAsync(&flag, c.d_cand_cnt, 4, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
        if (!flag) return PVS_OK;  // else: some L2 sum left the exact range -> score in order below
    }
    HIP_TRY(pvs_launch_dense_exact((int)ix->dtype, metric, ix->d_rows, ix->stride, ix->dim, ix->n, ix->d_norm2, c.d_qexact, c.d_qinfo, nb,
                                   c.d_qpad, d_out, nb, 0, (uint32_t)ix->n_cu, c.stream));
    return PVS_OK;
}

// queries per dense chunk so that the [n][nb] f32 matrix stays <= 2 GiB
static uint32_t dense_chunk_queries(const pvs_index *ix, uint32_t batch) {
    const uint64_t cap = (1ull << 31) / (4 * std::max<uint64_t>(ix->n, 1));
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>({cap, (uint64_t)PVS_MAX_BATCH, (uint64_t)batch}));
}

PVS_EXPORT pvs_status pvs_score_batch(pvs_index *ix, const void *queries, pvs_dtype qdtype, uint32_t batch, pvs_metric metric,
                                      float *out_dist, pvs_space out_space) {
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
        HIP_TRY(hipMalloc(&d_q, qbytes * batch));
        HIP_TRY(hipMemcpyAsync(d_q, queries, qbytes * batch, hipMemcpyHostToDevice, c->stream));
        const uint32_t cq = dense_chunk_queries(ix, batch);
        HIP_TRY(hipMalloc((void **)&d_m, (size_t)ix->n * cq * 4));
        for (uint32_t q0 = 0; q0 < batch; q0 += cq) {
            const uint32_t nb = std::min(cq, batch - q0);
            const uint32_t pad = nb <= 32 ? 32 : nb <= 64 ? 64 : 128;
            PVS_TRY(prep_chunk(ix, *c, d_q, qdtype, q0, nb, pad, metric));
            PVS_TRY(dense_chunk(ix, *c, nb, pad, metric, d_m));
            // scatter the chunk's columns into out[row * batch + q0 + j]
            const hipMemcpyKind kind = out_space == PVS_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
            HIP_TRY(hipMemcpy2DAsync(out_dist + q0, (size_t)batch * 4, d_m, (size_t)nb * 4, (size_t)nb * 4, ix->n, kind, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        return PVS_OK;
    };
    pvs_status st = body();
    hipFree(d_q);
    hipFree(d_m);
    ctx_done(ix, c);
    return st;
}

// shared tail: d_m [n][nb] (fanout == 0: nb output columns; else one) -> ranked groups on the host
static pvs_status aggregate_and_rank(pvs_index *ix, SearchCtx &c, const float *d_m, uint32_t nb, uint32_t fanout, int agg,
                                     const float *d_weights, const uint8_t *d_exclude, uint32_t k, int64_t *out_groups,
                                     double *out_values, uint32_t *out_count, FanoutWeights fw = FanoutWeights()) {
    const uint32_t G = ix->n_groups, ncol = fanout ? 1u : nb;
    double *d_vals = nullptr;
    int64_t *d_og = nullptr;
    double *d_ov = nullptr;
    uint32_t *d_oc = nullptr;
    auto body = [&]() -> pvs_status {
        HIP_TRY(hipMalloc((void **)&d_vals, (size_t)std::max<uint32_t>(G, 1) * ncol * 8));
        HIP_TRY(hipMalloc((void **)&d_og, (size_t)k * 8));
        HIP_TRY(hipMalloc((void **)&d_ov, (size_t)k * 8));
        HIP_TRY(hipMalloc((void **)&d_oc, 4));
        HIP_TRY(pvs_launch_group_aggregate(d_m, nb, nb, fanout, ix->d_grp_off, ix->d_grp_rows, G, d_weights, d_exclude, agg, d_vals,
                                           c.stream, fw));
        for (uint32_t q = 0; q < ncol; q++) {
            PVS_TRY(pvs_group_rank(d_vals + (size_t)q * G, ix->d_grp_ids, G, k, ix->gwork, d_og, d_ov, d_oc, c.stream));
            HIP_TRY(hipMemcpyAsync(out_groups + (size_t)q * k, d_og, (size_t)k * 8, hipMemcpyDeviceToHost, c.stream));
            HIP_TRY(hipMemcpyAsync(out_values + (size_t)q * k, d_ov, (size_t)k * 8, hipMemcpyDeviceToHost, c.stream));
            HIP_TRY(hipMemcpyAsync(out_count + q, d_oc, 4, hipMemcpyDeviceToHost, c.stream));
            HIP_TRY(hipStreamSynchronize(c.stream));
        }
        return PVS_OK;
    };
    pvs_status st = body();
    hipFree(d_vals);
    hipFree(d_og);
    hipFree(d_ov);
    hipFree(d_oc);
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
                                  int64_t *out_groups, double *out_values, uint32_t *out_count, bool *done) {
    *done = false;
    if (ix->n == 0 || ix->n_groups == 0 || ix->forced_path == 1) return PVS_OK;
    const uint64_t n = ix->n;
    const double per_group = (double)n / (double)ix->n_groups;
    uint64_t kp = std::max<uint64_t>(64, (uint64_t)(2.0 * k * std::min(std::ceil(per_group), 8.0)));
    kp = std::min<uint64_t>({kp, (uint64_t)PVS_MAX_K, n});
    if (kp < std::min<uint64_t>(k, n) || !fast_path_ok(ix, (uint32_t)kp)) return PVS_OK;
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        if (ix->h_ids_cache.size() != n) {
            ix->h_ids_cache.resize(n);
            HIP_TRY(hipMemcpy(ix->h_ids_cache.data(), ix->d_ids, n * 8, hipMemcpyDeviceToHost));
        }
    }
    std::vector<int64_t> ids;
    std::vector<float> dist;
    std::vector<uint32_t> cnt(batch);
    struct GV {
        double v;
        int64_t g;
    };
    std::vector<GV> gv;
    std::vector<int64_t> seen;
    for (;;) {
        ids.assign((size_t)batch * kp, -1);
        dist.assign((size_t)batch * kp, 0.f);
        PVS_TRY(pvs_search(ix, queries, qdtype, batch, (uint32_t)kp, metric, ids.data(), dist.data(), cnt.data()));
        bool all_ok = true;
        for (uint32_t q = 0; q < batch && all_ok; q++) {
            const int64_t *qi = ids.data() + (size_t)q * kp;
            const float *qd = dist.data() + (size_t)q * kp;
            gv.clear();
            seen.clear();
            for (uint32_t i = 0; i < cnt[q]; i++) {
                int64_t g = qi[i];  // identity groups: the group id is the row id
                if (!ix->h_groups.empty()) {
                    const auto it = std::lower_bound(ix->h_ids_cache.begin(), ix->h_ids_cache.end(), qi[i]);
                    g = ix->h_groups[(size_t)(it - ix->h_ids_cache.begin())];
                }
                seen.push_back(g);
            }
            // first occurrence of each group, in page order
            std::vector<uint32_t> order(seen.size());
            for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return seen[a] < seen[b]; });
            for (size_t i = 0; i < order.size(); i++)
                if (i == 0 || seen[order[i]] != seen[order[i - 1]]) gv.push_back({(double)qd[order[i]], seen[order[i]]});
            std::sort(gv.begin(), gv.end(), [](const GV &a, const GV &b) {
                const bool na = a.v != a.v, nb = b.v != b.v;  // NULL last, then value, then group id
                if (na != nb) return nb;
                if (!na && a.v != b.v) return a.v < b.v;
                return a.g < b.g;
            });
            const bool complete = cnt[q] == n;  // the page is the whole corpus
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
    if (agg == PVS_AGG_MIN && !row_weights) {
        bool done = false;
        PVS_TRY(groups_min_fast(ix, queries, qdtype, batch, k, metric, out_groups, out_values, out_count, &done));
        if (done) return PVS_OK;
    }
    uint32_t t;
    SearchCtx *c = ctx_acquire(ix, &t);
    void *d_q = nullptr;
    float *d_m = nullptr, *d_w = nullptr;
    auto body = [&]() -> pvs_status {
        PVS_TRY(ctx_prepare(ix, *c, batch, k, false));
        const size_t qbytes = (size_t)ix->dim * (qdtype == PVS_I8 ? 1 : 4);
        HIP_TRY(hipMalloc(&d_q, qbytes * batch));
        HIP_TRY(hipMemcpyAsync(d_q, queries, qbytes * batch, hipMemcpyHostToDevice, c->stream));
        if (row_weights && ix->n) {
            HIP_TRY(hipMalloc((void **)&d_w, ix->n * 4));
            HIP_TRY(hipMemcpyAsync(d_w, row_weights, ix->n * 4, hipMemcpyHostToDevice, c->stream));
        }
        const uint32_t cq = dense_chunk_queries(ix, batch);
        HIP_TRY(hipMalloc((void **)&d_m, std::max<size_t>((size_t)ix->n * cq * 4, 16)));
        for (uint32_t q0 = 0; q0 < batch; q0 += cq) {
            const uint32_t nb = std::min(cq, batch - q0);
            const uint32_t pad = nb <= 32 ? 32 : nb <= 64 ? 64 : 128;
            if (ix->n) {
                PVS_TRY(prep_chunk(ix, *c, d_q, qdtype, q0, nb, pad, metric));
                PVS_TRY(dense_chunk(ix, *c, nb, pad, metric, d_m));
            }
            PVS_TRY(aggregate_and_rank(ix, *c, d_m, nb, 0, agg, d_w, nullptr, k, out_groups + (size_t)q0 * k, out_values + (size_t)q0 * k,
                                       out_count + q0));
        }
        return PVS_OK;
    };
    pvs_status st = body();
    hipFree(d_q);
    hipFree(d_m);
    hipFree(d_w);
    ix->searches++;
    ix->dense_queries += batch;
    ctx_done(ix, c);
    return st;
}

pvs_status pvs_comm_gather_group_pages_(pvs_comm *c, const int64_t *groups, const double *values, const uint32_t *cnt, int64_t *all_groups,
                                        double *all_values, uint32_t *all_cnt, uint64_t elems, uint32_t batch, hipStream_t s);

PVS_EXPORT pvs_status pvs_search_groups_sharded(pvs_index *ix, pvs_comm *comm, const void *queries, pvs_dtype qdtype, uint32_t batch,
                                                uint32_t k, pvs_metric metric, pvs_agg agg, const float *row_weights, int64_t *out_groups,
                                                double *out_values, uint32_t *out_count) {
    if (!comm) return pvs_fail(PVS_ERR_INVALID_ARG, "null communicator");
    if (ix && pvs_comm_device_(comm) != ix->device) return pvs_fail(PVS_ERR_INVALID_ARG, "index and communicator live on different devices");
    // 1. this shard's page (every rank must take part in the exchange below, whatever its shard holds)
    std::vector<int64_t> lg((size_t)batch * k, -1);
    std::vector<double> lv((size_t)batch * k, __builtin_nan(""));
    std::vector<uint32_t> lc(batch, 0);
    PVS_TRY(pvs_search_groups(ix, queries, qdtype, batch, k, metric, agg, row_weights, lg.data(), lv.data(), lc.data()));
    if (batch == 0) return PVS_OK;
    HIP_TRY(hipSetDevice(ix->device));
    const uint32_t world = (uint32_t)pvs_comm_world_(comm);
    const uint64_t elems = (uint64_t)batch * k;
    int64_t *d_g = nullptr, *d_ag = nullptr;
    double *d_v = nullptr, *d_av = nullptr;
    uint32_t *d_c = nullptr, *d_ac = nullptr;
    std::vector<int64_t> ag((size_t)world * elems);
    std::vector<double> av((size_t)world * elems);
    std::vector<uint32_t> ac((size_t)world * batch);
    auto body = [&]() -> pvs_status {
        HIP_TRY(hipMalloc((void **)&d_g, elems * 8));
        HIP_TRY(hipMalloc((void **)&d_v, elems * 8));
        HIP_TRY(hipMalloc((void **)&d_c, (size_t)batch * 4));
        HIP_TRY(hipMalloc((void **)&d_ag, elems * 8 * world));
        HIP_TRY(hipMalloc((void **)&d_av, elems * 8 * world));
        HIP_TRY(hipMalloc((void **)&d_ac, (size_t)batch * 4 * world));
        hipStream_t s = ix->comm_stream;  // every collective of this index goes out on this one stream
        HIP_TRY(hipMemcpyAsync(d_g, lg.data(), elems * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_v, lv.data(), elems * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_c, lc.data(), (size_t)batch * 4, hipMemcpyHostToDevice, s));
        // 2. one grouped all-gather over xGMI
        PVS_TRY(pvs_comm_gather_group_pages_(comm, d_g, d_v, d_c, d_ag, d_av, d_ac, elems, batch, s));
        HIP_TRY(hipMemcpyAsync(ag.data(), d_ag, ag.size() * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(av.data(), d_av, av.size() * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(ac.data(), d_ac, ac.size() * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        // 3. merge on every rank (tiny: world * k entries per query)
        return pvs_merge_group_pages(ag.data(), av.data(), ac.data(), world, batch, k, out_groups, out_values, out_count);
    };
    pvs_status st = body();
    hipFree(d_g);
    hipFree(d_v);
    hipFree(d_c);
    hipFree(d_ag);
    hipFree(d_av);
    hipFree(d_ac);
    return st;
}

PVS_EXPORT pvs_status pvs_rrf_search(const pvs_rrf_branch *br, uint32_t nb, uint32_t k, int64_t *out_groups, double *out_scores,
                                     uint32_t *out_count) {
    if (!br || !out_groups || !out_scores || !out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (nb < 1 || nb > (uint32_t)PVS_RRF_MAX_BRANCHES) return pvs_fail(PVS_ERR_INVALID_ARG, "1..%d branches", PVS_RRF_MAX_BRANCHES);
    if (k < 1) return pvs_fail(PVS_ERR_INVALID_ARG, "k must be a positive integer");
    PvsRrfParams p;
    memset(&p, 0, sizeof p);
    p.n_branches = nb;
    uint64_t total = 0;
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
    unsigned long long *cat_key = nullptr, *cat_pay = nullptr;
    void *d_q = nullptr;
    float *d_m = nullptr, *d_w = nullptr;
    double *d_vals = nullptr;
    auto free_branch = [&]() {
        hipFree(d_q);
        hipFree(d_m);
        hipFree(d_w);
        hipFree(d_vals);
        d_q = nullptr;
        d_m = d_w = nullptr;
        d_vals = nullptr;
    };
    auto body = [&]() -> pvs_status {
        HIP_TRY(hipMalloc((void **)&cat_key, std::max<uint64_t>(total, 1) * 8));
        HIP_TRY(hipMalloc((void **)&cat_pay, std::max<uint64_t>(total, 1) * 8));
        uint64_t off = 0;
        for (uint32_t b = 0; b < nb; b++) {
            pvs_index *ix = br[b].idx;
            if (ix->n == 0) continue;
            if (ix->n > (1ull << 31) / 4) return pvs_fail(PVS_ERR_UNSUPPORTED, "branch %u: more than 2^29 rows in one dense column", b);
            uint32_t t;
            SearchCtx *c = ctx_acquire(ix, &t);
            auto one = [&]() -> pvs_status {
                PVS_TRY(ctx_prepare(ix, *c, 1, 1, false));
                const size_t qbytes = (size_t)ix->dim * (br[b].query_dtype == PVS_I8 ? 1 : 4);
                HIP_TRY(hipMalloc(&d_q, qbytes));
                HIP_TRY(hipMemcpyAsync(d_q, br[b].query, qbytes, hipMemcpyHostToDevice, c->stream));
                if (br[b].row_weights) {
                    HIP_TRY(hipMalloc((void **)&d_w, ix->n * 4));
                    HIP_TRY(hipMemcpyAsync(d_w, br[b].row_weights, ix->n * 4, hipMemcpyHostToDevice, c->stream));
                }
                HIP_TRY(hipMalloc((void **)&d_m, ix->n * 4));
                HIP_TRY(hipMalloc((void **)&d_vals, (size_t)std::max<uint32_t>(ix->n_groups, 1) * 8));
                // every row's exact distance (the dist_{cte} column), aggregated per group in row order ...
                PVS_TRY(prep_chunk(ix, *c, d_q, br[b].query_dtype, 0, 1, 32, br[b].metric));
                PVS_TRY(dense_chunk(ix, *c, 1, 32, br[b].metric, d_m));
                HIP_TRY(pvs_launch_group_aggregate(d_m, 1, 1, 0, ix->d_grp_off, ix->d_grp_rows, ix->n_groups, d_w, nullptr, br[b].agg, d_vals,
                                                   c->stream));
                // ... ranked over ALL groups of the branch, entries appended in branch order
                PVS_TRY(pvs_rrf_rank_branch(d_vals, ix->d_grp_ids, ix->n_groups, br[b].row_n_descending != 0, b, cat_key + off, cat_pay + off,
                                            c->stream));
                return PVS_OK;
            };
            pvs_status st = one();
            free_branch();
            ix->searches++;
            ix->dense_queries++;
            ctx_done(ix, c);
            if (st != PVS_OK) return st;
            off += ix->n_groups;
        }
        return pvs_rrf_fuse_device(cat_key, cat_pay, off, p, k, out_groups, out_scores, out_count, br[0].idx->search_stream);
    };
    pvs_status st = body();
    hipFree(cat_key);
    hipFree(cat_pay);
    return st;
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
    HIP_TRY(hipSetDevice(ix->device));
    std::vector<uint32_t> trow(n_targets);
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        PVS_TRY(ensure_groups(ix));
        if (ix->h_ids_cache.size() != ix->n) {
            ix->h_ids_cache.resize(ix->n);
            if (ix->n) HIP_TRY(hipMemcpy(ix->h_ids_cache.data(), ix->d_ids, ix->n * 8, hipMemcpyDeviceToHost));
        }
        for (uint32_t i = 0; i < n_targets; i++) {  // ids are strictly increasing: binary search
            auto it = std::lower_bound(ix->h_ids_cache.begin(), ix->h_ids_cache.end(), target_row_ids[i]);
            if (it == ix->h_ids_cache.end() || *it != target_row_ids[i])
                return pvs_fail(PVS_ERR_INVALID_ARG, "target row id %lld is not in the index", (long long)target_row_ids[i]);
            trow[i] = (uint32_t)(it - ix->h_ids_cache.begin());
        }
    }
    if (ix->n > (1ull << 31) / (4ull * n_targets)) return pvs_fail(PVS_ERR_UNSUPPORTED, "similar_to fan-out matrix would exceed 2 GiB");
    uint32_t t;
    SearchCtx *c = ctx_acquire(ix, &t);
    void *d_q = nullptr;
    float *d_m = nullptr;
    uint8_t *d_ex = nullptr;
    double *d_conf = nullptr, *d_lang = nullptr;
    uint32_t *d_trows = nullptr;
    uint8_t *d_kind = nullptr;
    const bool weighted = cw != 0.0 || lw != 0.0;
    const bool gated = row_kind && (skip_i2i || skip_t2t);
    auto body = [&]() -> pvs_status {
        PVS_TRY(ctx_prepare(ix, *c, n_targets, k, false));
        FanoutWeights fw;
        if (weighted || gated) {
            HIP_TRY(hipMalloc((void **)&d_trows, (size_t)n_targets * 4));
            HIP_TRY(hipMemcpy(d_trows, trow.data(), (size_t)n_targets * 4, hipMemcpyHostToDevice));
            fw.trows = d_trows;
        }
        if (gated) {
            HIP_TRY(hipMalloc((void **)&d_kind, std::max<uint64_t>(ix->n, 1)));
            HIP_TRY(hipMemcpy(d_kind, row_kind, ix->n, hipMemcpyHostToDevice));
            fw.kind = d_kind;
            fw.skip_i2i = skip_i2i;
            fw.skip_t2t = skip_t2t;
        }
        if (weighted) {
            // NULL pointer = every confidence NULL (coalesced to 1 in the kernel)
            auto upload = [&](const double *src, double **dst) -> pvs_status {
                HIP_TRY(hipMalloc((void **)dst, std::max<uint64_t>(ix->n, 1) * 8));
                if (src)
                    HIP_TRY(hipMemcpy(*dst, src, ix->n * 8, hipMemcpyHostToDevice));
                else
                    HIP_TRY(hipMemset(*dst, 0xff, ix->n * 8));  // all-ones bits = NaN
                return PVS_OK;
            };
            PVS_TRY(upload(row_conf, &d_conf));
            PVS_TRY(upload(row_lang, &d_lang));
            fw.conf = d_conf;
            fw.lang = d_lang;
            fw.cw = cw;
            fw.lw = lw;
        }
        // the target's stored vectors become the query batch: int8 codes as they are, f16/f32 as f32
        const size_t qesz = ix->dtype == PVS_I8 ? 1 : 4;
        std::vector<uint8_t> hq((size_t)n_targets * ix->dim * qesz);
        std::vector<uint8_t> rowbuf((size_t)ix->dim * ix->esz);
        for (uint32_t i = 0; i < n_targets; i++) {
            HIP_TRY(pvs_launch_rows_gather(ix->d_rows, ix->stride, (uint32_t)rowbuf.size(), trow[i], 1, c->d_qin, c->stream));
            HIP_TRY(hipMemcpyAsync(rowbuf.data(), c->d_qin, rowbuf.size(), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            uint8_t *dst = hq.data() + (size_t)i * ix->dim * qesz;
            if (ix->dtype == PVS_F16) {
                for (uint32_t e = 0; e < ix->dim; e++) {
                    _Float16 hv;
                    memcpy(&hv, rowbuf.data() + 2 * e, 2);
                    const float f = (float)hv;
                    memcpy(dst + 4 * e, &f, 4);
                }
            } else {
                memcpy(dst, rowbuf.data(), rowbuf.size());
            }
        }
        HIP_TRY(hipMalloc(&d_q, hq.size()));
        HIP_TRY(hipMemcpy(d_q, hq.data(), hq.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&d_ex, ix->n + 1));
        HIP_TRY(hipMemsetAsync(d_ex, 0, ix->n + 1, c->stream));
        for (uint32_t i = 0; i < n_targets; i++) HIP_TRY(hipMemsetAsync(d_ex + trow[i], 1, 1, c->stream));
        HIP_TRY(hipMalloc((void **)&d_m, (size_t)ix->n * n_targets * 4));
        const uint32_t pad = n_targets <= 32 ? 32 : n_targets <= 64 ? 64 : 128;
        PVS_TRY(prep_chunk(ix, *c, d_q, ix->dtype == PVS_I8 ? PVS_I8 : PVS_F32, 0, n_targets, pad, metric));
        PVS_TRY(dense_chunk(ix, *c, n_targets, pad, metric, d_m));
        PVS_TRY(aggregate_and_rank(ix, *c, d_m, n_targets, n_targets, agg, nullptr, d_ex, k, out_groups, out_values, out_count, fw));
        return PVS_OK;
    };
    pvs_status st = body();
    hipFree(d_q);
    hipFree(d_m);
    hipFree(d_ex);
    hipFree(d_conf);
    hipFree(d_lang);
    hipFree(d_trows);
    hipFree(d_kind);
    ix->searches++;
    ctx_done(ix, c);
    return st;
}

PVS_EXPORT pvs_status pvs_similar_to(pvs_index *ix, const int64_t *target_row_ids, uint32_t n_targets, uint32_t k, pvs_metric metric,
                                     pvs_agg agg, int64_t *out_groups, double *out_values, uint32_t *out_count) {
    return similar_to_impl(ix, target_row_ids, n_targets, k, metric, agg, nullptr, nullptr, 0.0, 0.0, nullptr, false, false, out_groups,
                           out_values, out_count);
}

PVS_EXPORT pvs_status pvs_similar_to_ex(pvs_index *ix, const int64_t *target_row_ids, uint32_t n_targets, uint32_t k, pvs_metric metric,
                                        const pvs_similar_opts *o, int64_t *out_groups, double *out_values, uint32_t *out_count) {
    if (!o || o->struct_size < sizeof(pvs_similar_opts)) return pvs_fail(PVS_ERR_INVALID_ARG, "pvs_similar_opts.struct_size too small");
    if (o->confidence_weight != o->confidence_weight || o->language_confidence_weight != o->language_confidence_weight)
        return pvs_fail(PVS_ERR_INVALID_ARG, "confidence weights must be numbers");
    return similar_to_impl(ix, target_row_ids, n_targets, k, metric, o->agg, o->row_confidence, o->row_language_confidence,
                           o->confidence_weight, o->language_confidence_weight, o->row_kind, o->xmodal_i2i == 0, o->xmodal_t2t == 0,
                           out_groups, out_values, out_count);
}

// ------------------------------------------------------- codec on the device
template <typename Fn>
static pvs_status with_device_chunks(const float *x, uint64_t n, pvs_space space, Fn &&fn) {
    if (space == PVS_DEVICE) return fn(x, n, (uint64_t)0);
    const uint64_t chunk = 64ull << 20;  // elements (256 MiB)
    float *stage = nullptr;
    HIP_TRY(hipMalloc((void **)&stage, std::min(chunk, n) * 4));
    pvs_status st = PVS_OK;
    for (uint64_t off = 0; off < n && st == PVS_OK; off += chunk) {
        const uint64_t m = std::min(chunk, n - off);
        hipError_t e = hipMemcpy(stage, x + off, m * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            st = pvs_fail(PVS_ERR_DEVICE, "H2D: %s", hipGetErrorString(e));
            break;
        }
        st = fn(stage, m, off);
    }
    hipFree(stage);
    return st;
}

PVS_EXPORT pvs_status pvs_absmax(const float *x, uint64_t n, pvs_space space, int32_t device, float *out_absmax) {
    if (!out_absmax || (n && !x)) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    PVS_TRY(use_device(device, nullptr));
    float *d_out = nullptr;
    HIP_TRY(hipMalloc((void **)&d_out, 4));
    float result = 0.f;
    pvs_status st = with_device_chunks(x, n, space, [&](const float *d, uint64_t m, uint64_t) -> pvs_status {
        HIP_TRY(pvs_launch_absmax(d, m, d_out, nullptr));
        float part = 0.f;
        HIP_TRY(hipMemcpy(&part, d_out, 4, hipMemcpyDeviceToHost));
        if (part > result) result = part;  // non-negative, NaN never stored
        return PVS_OK;
    });
    hipFree(d_out);
    if (st == PVS_OK) *out_absmax = result;
    return st;
}

PVS_EXPORT pvs_status pvs_quantize_i8(const float *x, uint64_t n, float scale, int8_t *out, pvs_space space, int32_t device) {
    if (n && (!x || !out)) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    PVS_TRY(use_device(device, nullptr));
    if (space == PVS_DEVICE) {
        HIP_TRY(pvs_launch_quantize_flat(x, n, scale, out, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
        return PVS_OK;
    }
    int8_t *d_out = nullptr;
    const uint64_t chunk = 64ull << 20;
    HIP_TRY(hipMalloc((void **)&d_out, std::min(chunk, std::max<uint64_t>(n, 1))));
    pvs_status st = with_device_chunks(x, n, space, [&](const float *d, uint64_t m, uint64_t off) -> pvs_status {
        HIP_TRY(pvs_launch_quantize_flat(d, m, scale, d_out, nullptr));
        HIP_TRY(hipMemcpy(out + off, d_out, m, hipMemcpyDeviceToHost));
        return PVS_OK;
    });
    hipFree(d_out);
    return st;
}

PVS_EXPORT pvs_status pvs_merge_topk_device(int32_t device, const int64_t *d_ids, const float *d_dist, const uint32_t *d_counts,
                                            uint32_t world, uint32_t batch, uint32_t k, int64_t *d_out_ids, float *d_out_dist,
                                            uint32_t *d_out_count) {
    if (!d_ids || !d_dist || !d_counts || !d_out_ids || !d_out_dist || !d_out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (world == 0 || batch == 0 || k == 0) return pvs_fail(PVS_ERR_INVALID_ARG, "empty merge");
    PVS_TRY(use_device(device, nullptr));
    HIP_TRY(pvs_launch_merge(d_ids, d_dist, d_counts, world, batch, k, d_out_ids, d_out_dist, d_out_count, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return PVS_OK;
}

// ------------------------------------------------ device memory + synthetic
PVS_EXPORT pvs_status pvs_device_malloc(int32_t device, size_t bytes, void **out) {
    if (!out) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    PVS_TRY(use_device(device, nullptr));
    HIP_TRY(hipMalloc(out, bytes ? bytes : 16));
    return PVS_OK;
}
PVS_EXPORT pvs_status pvs_device_free(int32_t device, void *ptr) {
    PVS_TRY(use_device(device, nullptr));
    HIP_TRY(hipFree(ptr));
    return PVS_OK;
}
PVS_EXPORT pvs_status pvs_memcpy(void *dst, const void *src, size_t bytes, int32_t device) {
    PVS_TRY(use_device(device, nullptr));
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDefault));
    return PVS_OK;
}
PVS_EXPORT pvs_status pvs_synth_rows_f32(int32_t device, uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim, float *d_out) {
    if (!d_out) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    PVS_TRY(use_device(device, nullptr));
    HIP_TRY(pvs_launch_synth(seed, row0, n, dim, d_out, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return PVS_OK;
}

// exposed to pvs_comm.hip
pvs_status pvs_index_internal_(pvs_index *ix, int *device) {
    if (!ix) return pvs_fail(PVS_ERR_INVALID_ARG, "null index");
    *device = ix->device;
    return PVS_OK;
}
