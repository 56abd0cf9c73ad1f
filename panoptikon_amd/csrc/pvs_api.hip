// pvs_api.hip — C ABI of libpvs, part 1: errors, index residency in HBM (create / add / grow / read back),
// per-search contexts, device codec entry points.  No torch, no CPU compute path: every distance is
// produced by a HIP kernel or the call fails (PVS_ERR_DEVICE).  Search orchestration: pvs_search.hip;
// per-item results, similar_to, RRF: pvs_items.hip.
#include "pvs_index.hpp"

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

// ------------------------------------------------------------------ test / tuning knobs
namespace {
std::atomic<int64_t> g_dbg[PVS_DBG_COUNT];
const char *const g_dbg_names[PVS_DBG_COUNT] = {
    "sample_div",      "sample_j_div", "no_light_finalize", "force_light_finalize", "dense_per_query",     "no_direct_score",
    "no_page_rank",    "rrf_serial",   "rrf_full",          "rrf_trace",            "scan_no_wide128",     "scratch_idle_cap_mb",
    "scratch_bypass",  "rrf_digest",   "no_sparse",         "sparse_max",           "no_fused_agg",        "no_side_finalize",
    "rrf_host_rounds", "multi_host_pages", "prelude_stream", "dense_nq4", "marker_events", "no_direct_topk", "direct_max_mb", "direct_queries",
    "direct_unit", "direct_static_pct", "direct_max_nq", "dense_full_sort", "dense_page_first", "no_flag_poll", "poll_late_pages",
    "no_exact_wide", "no_agg8", "no_dense2", "comm_timeout_s", "comm_fail_local",
    "no_float_certify", "float_certify_queries", "float_certify_rows", "float_certify_trace", "float_certify_no_fold",
};
int dbg_key(const char *key) {
    if (!key) return -1;
    for (int i = 0; i < PVS_DBG_COUNT; i++)
        if (strcmp(key, g_dbg_names[i]) == 0) return i;
    return -1;
}
}  // namespace
int64_t pvs_dbg(PvsDbg key) { return g_dbg[key].load(std::memory_order_relaxed); }
void pvs_dbg_add(PvsDbg key, int64_t v) { g_dbg[key].fetch_add(v, std::memory_order_relaxed); }
PVS_EXPORT pvs_status pvs_debug_set(const char *key, int64_t value) {
    const int i = dbg_key(key);
    if (i < 0) return pvs_fail(PVS_ERR_INVALID_ARG, "unknown debug key '%s'", key ? key : "(null)");
    g_dbg[i].store(value);
    return PVS_OK;
}
PVS_EXPORT pvs_status pvs_debug_get(const char *key, int64_t *out_value) {
    const int i = dbg_key(key);
    if (i < 0 || !out_value) return pvs_fail(PVS_ERR_INVALID_ARG, "unknown debug key '%s'", key ? key : "(null)");
    *out_value = g_dbg[i].load();
    return PVS_OK;
}

// ------------------------------------------------------------------ scratch cache
// Blocks for the host-orchestrated paths, kept per (device, size class) after use.  Three rules:
//   * a block handed back while work that touches it may still be queued (pvs_scratch_free_on: the usual case — the caller
//     enqueued kernels and returns) carries an event recorded on that stream and is only handed out again once the event has
//     completed: another thread, on another stream, can never get a block a kernel is still reading or writing;
//   * idle bytes per device are capped (pvs_debug_set("scratch_idle_cap_mb"), default 16 GiB): beyond the cap the least recently used idle
//     blocks go back to the runtime — the dense fallback's n x per x 4 byte matrices come in dozens of size classes and would
//     otherwise pile up multi-GB blocks nobody asks for again;
//   * every device allocation of the library (pvs_malloc_retry) returns the idle blocks to the runtime and tries again before it
//     reports out-of-memory.
namespace {
struct ScratchBlock {
    int device;
    size_t cls;
};
struct IdleBlock {
    void *p;
    hipEvent_t ev;      // nullptr: nothing pending
    uint64_t stamp;     // LRU clock
};
std::mutex g_scratch_mu;
std::map<std::pair<int, size_t>, std::vector<IdleBlock>> g_scratch_idle;  // (device, size class) -> idle blocks
std::map<void *, ScratchBlock> g_scratch_live;
std::map<void *, int> g_scratch_bypassed;  // blocks handed out under pvs_debug_set("scratch_bypass", 1)
std::map<int, size_t> g_scratch_idle_bytes;  // per device
uint64_t g_scratch_clock = 0;
size_t scratch_class(size_t bytes) {
    size_t c = 4096;
    while (c < bytes) c <<= 1;
    if (c > (64u << 20)) c = (bytes + (16u << 20) - 1) / (16u << 20) * (16u << 20);  // big blocks: 16 MiB steps, not powers of two
    return c;
}
size_t scratch_idle_cap() {
    const int64_t mb = pvs_dbg(PVS_DBG_SCRATCH_IDLE_CAP_MB);
    return mb > 0 ? (size_t)mb << 20 : (size_t)16 << 30;
}
// (lock held) idle blocks of `device` beyond the cap, least recently used first, are moved to `drop`
void scratch_evict_locked(int device, std::vector<IdleBlock> *drop) {
    while (g_scratch_idle_bytes[device] > scratch_idle_cap()) {
        std::vector<IdleBlock> *best_v = nullptr;
        size_t best_i = 0, best_cls = 0;
        uint64_t best_stamp = ~0ull;
        for (auto &kv : g_scratch_idle)
            if (kv.first.first == device)
                for (size_t i = 0; i < kv.second.size(); i++)
                    if (kv.second[i].stamp < best_stamp) {
                        best_stamp = kv.second[i].stamp;
                        best_v = &kv.second;
                        best_i = i;
                        best_cls = kv.first.second;
                    }
        if (!best_v) break;
        drop->push_back((*best_v)[best_i]);
        best_v->erase(best_v->begin() + (long)best_i);
        g_scratch_idle_bytes[device] -= best_cls;
    }
}
void scratch_release(int device, std::vector<IdleBlock> &drop) {  // (no lock held) hipFree waits for whatever still uses the block
    if (drop.empty()) return;
    int cur = 0;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(device);
    for (IdleBlock &b : drop) {
        if (b.ev) {
            (void)hipEventSynchronize(b.ev);
            (void)hipEventDestroy(b.ev);
        }
        (void)hipFree(b.p);
    }
    (void)hipSetDevice(cur);
}
}  // namespace
hipError_t pvs_scratch_alloc(void **out, size_t bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const size_t cls = scratch_class(bytes ? bytes : 1);
    if (pvs_dbg(PVS_DBG_SCRATCH_BYPASS)) {  // (race hunting: no block is ever handed out twice)
        e = pvs_malloc_retry(out, cls);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        g_scratch_bypassed[*out] = dev;
        return hipSuccess;
    }
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        auto &v = g_scratch_idle[{dev, cls}];
        for (size_t i = v.size(); i-- > 0;) {  // most recently freed first
            if (v[i].ev) {
                if (hipEventQuery(v[i].ev) != hipSuccess) {
                    (void)hipGetLastError();  // (hipErrorNotReady is sticky in hipGetLastError)
                    continue;                // still in use on the stream it was freed on
                }
                (void)hipEventDestroy(v[i].ev);
            }
            *out = v[i].p;
            v.erase(v.begin() + (long)i);
            g_scratch_idle_bytes[dev] -= cls;
            g_scratch_live[*out] = {dev, cls};
            return hipSuccess;
        }
    }
    e = pvs_malloc_retry(out, cls);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    g_scratch_live[*out] = {dev, cls};
    return hipSuccess;
}
// `s`: the stream whose queued work may still touch the block; nullptr = the caller has already waited for it
void pvs_scratch_free_on(void *p, hipStream_t s, bool pending) {
    if (!p) return;
    hipEvent_t ev = nullptr;
    if (pending) {
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, s) != hipSuccess) {
            if (ev) (void)hipEventDestroy(ev);
            ev = nullptr;
            (void)hipStreamSynchronize(s);  // no event: wait here instead
        }
    }
    std::vector<IdleBlock> drop;
    int dev = 0;
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        auto it = g_scratch_live.find(p);
        if (it == g_scratch_live.end()) {
            if (ev) (void)hipEventDestroy(ev);
            auto bt = g_scratch_bypassed.find(p);
            if (bt != g_scratch_bypassed.end()) {  // pvs_debug_set("scratch_bypass", 1): straight back to the runtime (hipFree waits for the device)
                g_scratch_bypassed.erase(bt);
                (void)hipFree(p);
            }
            return;  // (else: not ours)
        }
        dev = it->second.device;
        g_scratch_idle[{dev, it->second.cls}].push_back({p, ev, ++g_scratch_clock});
        g_scratch_idle_bytes[dev] += it->second.cls;
        g_scratch_live.erase(it);
        scratch_evict_locked(dev, &drop);
    }
    scratch_release(dev, drop);
}
void pvs_scratch_free(void *p) { pvs_scratch_free_on(p, nullptr, false); }
void pvs_scratch_trim(int device) {
    std::vector<IdleBlock> drop;
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        for (auto &kv : g_scratch_idle)
            if (kv.first.first == device) {
                drop.insert(drop.end(), kv.second.begin(), kv.second.end());
                kv.second.clear();
            }
        g_scratch_idle_bytes[device] = 0;
    }
    scratch_release(device, drop);
}
// hipMalloc that gives the scratch cache's idle blocks back to the runtime and tries again before reporting out-of-memory
hipError_t pvs_malloc_retry(void **out, size_t bytes) {
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) {
            pvs_scratch_trim(dev);
            e = hipMalloc(out, bytes);
        }
    }
    return e;
}
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

pvs_status use_device(int32_t device, int *resolved) {
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

void span_begin(pvs_index *ix, SearchCtx &c, int kind, uint64_t rows, hipStream_t on) {
    if (!ix->profiling) return;
    hipStream_t st = on ? on : c.stream;
    TimedSpan t;
    if (!c.span_pool.empty()) {
        t = c.span_pool.back();
        c.span_pool.pop_back();
    } else if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) {
        return;
    }
    t.kind = kind;
    t.rows = rows;
    (void)hipEventRecord(t.a, st);
    c.spans.push_back(t);
}
bool span_bound(pvs_index *ix, SearchCtx &c, int kind, uint64_t rows, hipEvent_t *ev_start, hipEvent_t *ev_stop) {
    if (!ix->profiling) return false;
    TimedSpan t;
    if (!c.span_pool.empty()) {
        t = c.span_pool.back();
        c.span_pool.pop_back();
    } else if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) {
        return false;
    }
    t.kind = kind;
    t.rows = rows;
    c.spans.push_back(t);
    *ev_start = t.a;
    *ev_stop = t.b;
    return true;
}
void span_end(pvs_index *ix, SearchCtx &c, hipStream_t on) {
    if (!ix->profiling || c.spans.empty()) return;
    (void)hipEventRecord(c.spans.back().b, on ? on : c.stream);
}
// after the stream drained
void spans_collect(pvs_index *ix, SearchCtx &c) {
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
            } else if (t.kind == 2) {
                ix->prof.finalize_launches++;
                ix->prof.finalize_ms += ms;
            } else {
                ix->prof.exchange_launches++;
                ix->prof.exchange_ms += ms;
            }
        }
        c.span_pool.push_back(t);
    }
    c.spans.clear();
}

void ctx_release(SearchCtx &c) {
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
    hipFree(c.d_direct);
    hipFree(c.d_qstage);
    hipFree(c.d_mask);
    hipFree(c.d_aux_masked);
    hipFree(c.d_qexact);
    hipFree(c.d_qinfo);
    hipFree(c.d_thr);
    hipFree(c.d_gmin);
    hipFree(c.d_cand_cnt);
    hipFree(c.d_seg);
    hipFree(c.d_seg_cnt);
    hipFree(c.d_cand);
    hipFree(c.d_flat_cnt);
    hipFree(c.d_fin_ub);
    hipFree(c.d_fin_surv);
    hipFree(c.d_fin_sort);
    c.d_fin_ub = c.d_fin_surv = nullptr;
    c.d_fin_sort = nullptr;
    hipFree(c.d_need_dense);
    if (c.h_need_dense) hipHostFree(c.h_need_dense);
    hipFree(c.d_out_ids);
    hipFree(c.d_out_dist);
    hipFree(c.d_out_count);
    pvs_dense_release(c.dense);
    pvs_group_work_release(c.gwork);
    if (c.done) hipEventDestroy(c.done);
    if (c.scanned) hipEventDestroy(c.scanned);
    if (c.preluded) hipEventDestroy(c.preluded);
    hipFree(c.d_loc_rec);
    hipFree(c.d_all_rec);
    if (c.h_all_flags) hipHostFree(c.h_all_flags);
    if (c.own_stream) hipStreamDestroy(c.own_stream);
    if (c.h_io) hipHostFree(c.h_io);
    if (c.h_cert) hipHostFree(c.h_cert);
    c = SearchCtx();
}

pvs_status ctx_pinned_io(SearchCtx &c, size_t bytes) {
    if (bytes <= c.h_io_cap) return PVS_OK;
    if (c.h_io) hipHostFree(c.h_io);
    c.h_io = nullptr;
    c.h_io_cap = 0;
    const size_t cap = pvs_round_up(std::max<size_t>(bytes, 1 << 16), 1 << 16);
    HIP_TRY(hipHostMalloc((void **)&c.h_io, cap, hipHostMallocDefault));
    c.h_io_cap = cap;
    return PVS_OK;
}

// FinalizeArgs.w_*: the work area of the LDS-light pass C (beside another search's scan — several streams, or the side stream of a
// pipelined caller).  ~40 MB per context: allocated by the first search of the context that takes that route (enqueue_fast_chunk),
// not for contexts that only ever run direct, sparse or dense searches.
pvs_status ctx_fin_buffers(SearchCtx &c) {
    if (c.d_fin_ub && c.d_fin_surv && c.d_fin_sort) return PVS_OK;
    if (!c.d_fin_ub) HIP_TRY(pvs_malloc_retry((void **)&c.d_fin_ub, 4 * (size_t)PVS_SCAN_MAX_BATCH * PVS_CAND_CAP));
    if (!c.d_fin_surv) HIP_TRY(pvs_malloc_retry((void **)&c.d_fin_surv, 4 * (size_t)PVS_SCAN_MAX_BATCH * PVS_SURV_CAP));
    if (!c.d_fin_sort) HIP_TRY(pvs_malloc_retry((void **)&c.d_fin_sort, 8 * (size_t)PVS_SCAN_MAX_BATCH * PVS_SURV_CAP));
    return PVS_OK;
}

pvs_status ctx_prepare(pvs_index *ix, SearchCtx &c, uint32_t batch, uint32_t k, bool host_outputs) {
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
    if (!c.scanned) HIP_TRY(hipEventCreateWithFlags(&c.scanned, hipEventDisableTiming));
    if (!c.preluded) HIP_TRY(hipEventCreateWithFlags(&c.preluded, hipEventDisableTiming));
    c.side_finalize = false;
    if (!c.d_qmat) {
        HIP_TRY(pvs_malloc_retry((void **)&c.d_qin, (size_t)PVS_SCAN_MAX_BATCH * ix->dim * 4));
        HIP_TRY(pvs_malloc_retry((void **)&c.d_qmat, (size_t)PVS_SCAN_MAX_BATCH * ix->stride));
        HIP_TRY(pvs_malloc_retry((void **)&c.d_qpad, pvs_dense_exact_scratch_bytes(ix->stride, ix->esz)));
        HIP_TRY(pvs_malloc_retry((void **)&c.d_qexact, (size_t)PVS_SCAN_MAX_BATCH * ix->dim * 4));
        HIP_TRY(pvs_malloc_retry((void **)&c.d_qinfo, sizeof(QInfo) * PVS_SCAN_MAX_BATCH));
        HIP_TRY(pvs_malloc_retry((void **)&c.d_thr, 4 * PVS_SCAN_MAX_BATCH));
        HIP_TRY(pvs_malloc_retry((void **)&c.d_gmin, (size_t)4 * PVS_SCAN_MAX_BATCH * GMAX));
        HIP_TRY(pvs_malloc_retry((void **)&c.d_cand_cnt, 64));
        HIP_TRY(pvs_malloc_retry((void **)&c.d_seg, sizeof(uint2) * (size_t)PVS_SEG_PAIRS * PVS_SEG_CAP));
        HIP_TRY(pvs_malloc_retry((void **)&c.d_seg_cnt, 4 * (size_t)PVS_SEG_PAIRS * (PVS_SEG_CAP / PVS_WIDE_SEG_CAP)));  // (the 256-query kernel: twice the lists at half the slots)
        HIP_TRY(pvs_malloc_retry((void **)&c.d_cand, sizeof(uint2) * (size_t)PVS_SCAN_MAX_BATCH * PVS_CAND_CAP));
        HIP_TRY(pvs_malloc_retry((void **)&c.d_flat_cnt, 4 * (size_t)PVS_SCAN_MAX_BATCH));
    }
    if (batch > c.flags_cap) {
        hipFree(c.d_need_dense);
        if (c.h_need_dense) hipHostFree(c.h_need_dense);
        c.d_need_dense = nullptr;
        c.h_need_dense = nullptr;
        uint32_t cap = (uint32_t)pvs_round_up(batch, 256);
        // [0, cap): per-query "answer me on the dense path" flags; [cap, 2 cap): candidates the filter scan emitted per query
        HIP_TRY(pvs_malloc_retry((void **)&c.d_need_dense, 8 * (size_t)cap));
        HIP_TRY(hipHostMalloc((void **)&c.h_need_dense, 8 * (size_t)cap, hipHostMallocDefault));
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
            HIP_TRY(pvs_malloc_retry((void **)&c.d_out_ids, 8 * need));
            HIP_TRY(pvs_malloc_retry((void **)&c.d_out_dist, 4 * need));
            HIP_TRY(pvs_malloc_retry((void **)&c.d_out_count, 4 * (size_t)batch));
            c.out_cap = need;
            c.out_batch_cap = batch;
        }
    }
    return PVS_OK;
}

pvs_status pvs_index_reserve_(pvs_index *ix, uint64_t rows);

PVS_EXPORT pvs_status pvs_index_create(const pvs_index_desc *desc, pvs_index **out) {
    if (!desc || !out) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    constexpr size_t V1_SIZE = offsetof(pvs_index_desc, n_devices);  // ABI v1: the fields up to id_base
    if (desc->struct_size < V1_SIZE) return pvs_fail(PVS_ERR_INVALID_ARG, "pvs_index_desc.struct_size too small");
    if (desc->struct_size >= sizeof(pvs_index_desc) && desc->n_devices > 1) return multi_create(desc, out);
    if (desc->dtype > PVS_I8) return pvs_fail(PVS_ERR_INVALID_ARG, "unknown dtype %u", desc->dtype);
    if (desc->dim == 0 || desc->dim > 16384) return pvs_fail(PVS_ERR_INVALID_ARG, "dim %u out of range [1, 16384]", desc->dim);
    int dev = 0;
    const int32_t want_dev = (desc->struct_size >= sizeof(pvs_index_desc) && desc->n_devices == 1 && desc->devices) ? desc->devices[0] : desc->device;
    PVS_TRY(use_device(want_dev, &dev));
    pvs_index *ix = new (std::nothrow) pvs_index();
    if (!ix) return pvs_fail(PVS_ERR_OOM, "host allocation failed");
    ix->device = dev;
    ix->dtype = desc->dtype;
    ix->dim = desc->dim;
    ix->esz = pvs_esz(desc->dtype);
    ix->stride = (uint32_t)pvs_round_up((uint64_t)desc->dim * ix->esz, PVS_KSLAB_BYTES);
    // pad the row pitch up to the next pitch the filter scan has an instance for (zero padding is free for every
    // kernel; e.g. 640-d f16: 1280 B -> 1536 B): 9-33 % more HBM beats falling back to the dense path
    for (uint32_t ks = ix->stride / PVS_KSLAB_BYTES; ks <= 24; ks++)
        if (pvs_scan_supported((int)ix->dtype, ks)) {
            if (ks * PVS_KSLAB_BYTES <= ix->stride + (ix->stride + 2) / 3) ix->stride = ks * PVS_KSLAB_BYTES;
            break;
        }
    ix->id_base = desc->id_base;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) == hipSuccess) ix->n_cu = p.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&ix->admin_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ix->search_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ix->fin_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ix->pre_stream, hipStreamNonBlocking);
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
    HIP_TRY(pvs_malloc_retry((void **)&rows_new, cap * (uint64_t)ix->stride));
    float *rnorm_new = nullptr, *scos_new = nullptr, *sl2_new = nullptr;
    const uint64_t aux_floats = cap / 32 * PVS_AUX_REC;
    hipError_t e = pvs_malloc_retry((void **)&norm_new, cap * 4);
    if (e == hipSuccess) e = pvs_malloc_retry((void **)&rnorm_new, cap * 4);
    if (e == hipSuccess) e = pvs_malloc_retry((void **)&ids_new, cap * 8);
    if (e == hipSuccess) e = pvs_malloc_retry((void **)&scos_new, aux_floats * 4);
    if (e == hipSuccess) e = pvs_malloc_retry((void **)&sl2_new, aux_floats * 4);
    if (e != hipSuccess) {
        hipFree(rows_new);
        hipFree(norm_new);
        hipFree(rnorm_new);
        hipFree(ids_new);
        hipFree(scos_new);
        hipFree(sl2_new);
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
    // the scan's row-scalar records of every tile, old rows and padding alike, from the arrays just filled
    HIP_TRY(pvs_launch_scan_aux(norm_new, rnorm_new, 0, cap, scos_new, sl2_new, s));
    HIP_TRY(hipStreamSynchronize(s));
    hipFree(ix->d_rows);
    hipFree(ix->d_norm2);
    hipFree(ix->d_rnorm);
    hipFree(ix->d_ids);
    hipFree(ix->d_scan_cos);
    hipFree(ix->d_scan_l2);
    ix->d_rows = rows_new;
    ix->d_norm2 = norm_new;
    ix->d_rnorm = rnorm_new;
    ix->d_ids = ids_new;
    ix->d_scan_cos = scos_new;
    ix->d_scan_l2 = sl2_new;
    ix->cap = cap;
    return PVS_OK;
}

PVS_EXPORT void pvs_index_destroy(pvs_index *ix) {
    if (!ix) return;
    if (is_multi(ix)) return multi_destroy(ix);
    (void)hipSetDevice(ix->device);
    (void)hipDeviceSynchronize();
    for (auto &c : ix->ctx) ctx_release(c);
    hipFree(ix->d_rows);
    hipFree(ix->d_norm2);
    hipFree(ix->d_rnorm);
    hipFree(ix->d_ids);
    hipFree(ix->d_scan_cos);
    hipFree(ix->d_scan_l2);
    hipFree(ix->d_trank);
    hipFree(ix->d_tinv);
    hipFree(ix->d_order_keys);
    hipFree(ix->d_grp_off);
    hipFree(ix->d_grp_rows);
    hipFree(ix->d_grp_ids);
    hipFree(ix->d_grp_tinv);
    hipFree(ix->d_grp_trank);
    hipFree(ix->d_tile_grp);
    hipFree(ix->d_straddlers);
    hipFree(ix->d_row_gidx);
    hipFree(ix->d_grp_key);
    hipFree(ix->d_null_rows[0]);
    hipFree(ix->d_null_rows[1]);
    if (ix->admin_stream) hipStreamDestroy(ix->admin_stream);
    if (ix->search_stream) hipStreamDestroy(ix->search_stream);
    if (ix->fin_stream) hipStreamDestroy(ix->fin_stream);
    if (ix->pre_stream) hipStreamDestroy(ix->pre_stream);
    if (ix->comm_stream) hipStreamDestroy(ix->comm_stream);
    pvs_scratch_trim(ix->device);
    delete ix;
}

pvs_status check_ids(pvs_index *ix, const int64_t *row_ids, uint64_t n, int64_t *last, int64_t implicit_id0) {
    int64_t prev = ix->last_id;
    if (!row_ids) {
        int64_t first = implicit_id0 != INT64_MIN ? implicit_id0 : ix->id_base + (int64_t)ix->n;
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
                                     const int64_t *group_ids, int64_t last_id, int64_t implicit_first) {
    if (ix->n + n > ix->cap) PVS_TRY(pvs_index_reserve_(ix, std::max<uint64_t>(ix->n + n, ix->cap * 2)));
    hipStream_t s = ix->admin_stream;
    const int mode = (from_f32 && ix->dtype == PVS_I8) ? 0 : (from_f32 && ix->dtype == PVS_F16) ? 1 : 2;
    HIP_TRY(pvs_launch_rows_ingest(mode, rows_dev, ix->dim, ix->esz, ix->n, n, ix->scale, ix->d_rows, ix->stride, s));
    HIP_TRY(pvs_launch_norm2((int)ix->dtype, ix->d_rows, ix->stride, ix->dim, ix->n, n, ix->d_norm2, ix->d_rnorm, s));
    HIP_TRY(pvs_launch_scan_aux(ix->d_norm2, ix->d_rnorm, ix->n, n, ix->d_scan_cos, ix->d_scan_l2, s));
    if (row_ids)
        HIP_TRY(hipMemcpyAsync(ix->d_ids + ix->n, row_ids, n * 8, hipMemcpyHostToDevice, s));
    else
        HIP_TRY(pvs_launch_iota_ids(ix->d_ids + ix->n, n, implicit_first, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (group_ids) {
        if (ix->h_groups.size() != ix->n) ix->h_groups.resize(ix->n, -1);
        ix->h_groups.insert(ix->h_groups.end(), group_ids, group_ids + n);
    } else if (!ix->h_groups.empty()) {
        ix->h_groups.resize(ix->n + n, -1);
    }
    ix->n += n;
    ix->last_id = last_id;
    ix->ids_epoch++;
    ix->groups_built_n = UINT64_MAX;
    return PVS_OK;
}

pvs_status add_impl(pvs_index *ix, const void *rows, bool from_f32, uint64_t n, const int64_t *row_ids, const int64_t *group_ids,
                    pvs_space space, int64_t implicit_id0) {
    if (!ix) return pvs_fail(PVS_ERR_INVALID_ARG, "null index");
    if (n == 0) return PVS_OK;
    if (!rows) return pvs_fail(PVS_ERR_INVALID_ARG, "null rows");
    if (is_multi(ix)) return multi_add(ix, rows, from_f32, n, row_ids, group_ids, space);
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->poisoned) return pvs_fail(PVS_ERR_STATE, PVS_POISONED_MSG);
    HIP_TRY(hipSetDevice(ix->device));
    if (from_f32 && ix->dtype == PVS_I8 && !ix->scale_set)
        return pvs_fail(PVS_ERR_STATE, "int8 index has no scale artifact: set it before adding f32 rows");
    int64_t last = 0;
    PVS_TRY(check_ids(ix, row_ids, n, &last, implicit_id0));
    const size_t src_esz = from_f32 ? 4 : ix->esz;
    const int64_t first_implicit = implicit_id0 != INT64_MIN ? implicit_id0 : ix->id_base + (int64_t)ix->n;
    if (space == PVS_DEVICE) return append_device_rows(ix, rows, from_f32, n, row_ids, group_ids, last, first_implicit);
    // host rows: stage through HBM in chunks of <= 256 MiB
    const uint64_t row_bytes = (uint64_t)ix->dim * src_esz;
    const uint64_t chunk = std::max<uint64_t>(1, (256ull << 20) / row_bytes);
    void *stage = nullptr;
    HIP_TRY(pvs_malloc_retry(&stage, std::min(chunk, n) * row_bytes));
    pvs_status st = PVS_OK;
    if (ix->n + n > ix->cap) st = pvs_index_reserve_(ix, std::max<uint64_t>(ix->n + n, ix->cap * 2));
    for (uint64_t off = 0; off < n && st == PVS_OK; off += chunk) {
        const uint64_t m = std::min(chunk, n - off);
        hipError_t e = hipMemcpy(stage, (const uint8_t *)rows + off * row_bytes, m * row_bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            st = pvs_fail(PVS_ERR_DEVICE, "hipMemcpy H2D: %s", hipGetErrorString(e));
            break;
        }
        int64_t chunk_last = row_ids ? row_ids[off + m - 1] : first_implicit + (int64_t)(off + m) - 1;
        st = append_device_rows(ix, stage, from_f32, m, row_ids ? row_ids + off : nullptr, group_ids ? group_ids + off : nullptr,
                                chunk_last, first_implicit + (int64_t)off);
    }
    hipFree(stage);
    return st;
}

PVS_EXPORT pvs_status pvs_index_add(pvs_index *ix, const void *rows, uint64_t n, const int64_t *row_ids,
                                    const int64_t *group_ids, pvs_space space) {
    GateExcl gate(ix);  // (searches see the index before or after the append: the rows may move to a larger allocation)
    PVS_GATE_REFUSED(gate);
    return add_impl(ix, rows, false, n, row_ids, group_ids, space);
}
PVS_EXPORT pvs_status pvs_index_add_f32(pvs_index *ix, const float *rows, uint64_t n, const int64_t *row_ids,
                                        const int64_t *group_ids, pvs_space space) {
    GateExcl gate(ix);
    PVS_GATE_REFUSED(gate);
    return add_impl(ix, rows, true, n, row_ids, group_ids, space);
}

PVS_EXPORT pvs_status pvs_index_set_scale(pvs_index *ix, float scale) {
    if (!ix) return pvs_fail(PVS_ERR_INVALID_ARG, "null index");
    if (ix->dtype != PVS_I8) return pvs_fail(PVS_ERR_INVALID_ARG, "only int8 indexes carry a scale artifact");
    // artifact_scale (db/vector_quants.rs:1456-1460): finite and > 0, nothing else
    if (!(std::isfinite(scale) && scale > 0.0f)) return pvs_fail(PVS_ERR_INVALID_ARG, "unusable scale artifact");
    if (is_multi(ix)) return multi_set_scale(ix, scale);
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->n && ix->scale_set && ix->scale != scale)
        return pvs_fail(PVS_ERR_STATE, "scale is frozen once rows exist (artifact_rev semantics): rebuild the index");
    ix->scale = scale;
    ix->scale_set = true;
    return PVS_OK;
}
PVS_EXPORT pvs_status pvs_index_set_order_keys(pvs_index *ix, const int64_t *keys, uint64_t n, pvs_space space) {
    if (!ix) return pvs_fail(PVS_ERR_INVALID_ARG, "null index");
    GateExcl gate(ix);  // (searches in flight read the tie ranks)
    PVS_GATE_REFUSED(gate);
    if (is_multi(ix)) return multi_set_order_keys(ix, keys, n, space);
    std::lock_guard<std::mutex> lk(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    hipFree(ix->d_trank);
    hipFree(ix->d_tinv);
    hipFree(ix->d_order_keys);
    ix->d_trank = ix->d_tinv = nullptr;
    ix->d_order_keys = nullptr;
    ix->order_rows = 0;
    ix->order_epoch++;  // (the NULL-row list is kept in tie order: pvs_ensure_null_rows)
    ix->h_order_keys.clear();
    ix->groups_built_n = UINT64_MAX;  // the groups' tie order is rebuilt with the CSR
    if (!keys) return PVS_OK;
    if (n != ix->n) return pvs_fail(PVS_ERR_INVALID_ARG, "%llu order keys for %llu rows: one key per stored row", (unsigned long long)n, (unsigned long long)ix->n);
    if (n == 0) return PVS_OK;
    int64_t *d_keys = nullptr;
    const int64_t *src = keys;
    auto body = [&]() -> pvs_status {
        if (space == PVS_HOST) {
            HIP_TRY(pvs_scratch_alloc((void **)&d_keys, n * 8));
            HIP_TRY(hipMemcpy(d_keys, keys, n * 8, hipMemcpyHostToDevice));
            src = d_keys;
        }
        HIP_TRY(pvs_malloc_retry((void **)&ix->d_trank, n * 4));
        HIP_TRY(pvs_malloc_retry((void **)&ix->d_tinv, n * 4));
        HIP_TRY(pvs_malloc_retry((void **)&ix->d_order_keys, n * 8));
        HIP_TRY(hipMemcpy(ix->d_order_keys, src, n * 8, hipMemcpyDeviceToDevice));
        PVS_TRY(pvs_build_tie_ranks(src, n, ix->d_trank, ix->d_tinv, ix->admin_stream));
        return PVS_OK;
    };
    pvs_status st = body();
    pvs_scratch_free(d_keys);  // (pvs_build_tie_ranks is synchronous)
    if (st != PVS_OK) {
        hipFree(ix->d_trank);
        hipFree(ix->d_tinv);
        hipFree(ix->d_order_keys);
        ix->d_trank = ix->d_tinv = nullptr;
        ix->d_order_keys = nullptr;
        return st;
    }
    ix->h_order_keys.resize(n);
    if (space == PVS_HOST)
        memcpy(ix->h_order_keys.data(), keys, n * 8);
    else
        HIP_TRY(hipMemcpy(ix->h_order_keys.data(), keys, n * 8, hipMemcpyDeviceToHost));
    ix->order_rows = n;
    return PVS_OK;
}
PVS_EXPORT pvs_status pvs_index_set_scale_artifact(pvs_index *ix, const uint8_t *artifact, size_t len) {
    float s = 0.f;
    PVS_TRY(pvs_artifact_scale(artifact, len, &s));
    return pvs_index_set_scale(ix, s);
}

PVS_EXPORT pvs_status pvs_index_set_streams(pvs_index *ix, uint32_t n_streams) {
    if (!ix || n_streams == 0) return pvs_fail(PVS_ERR_INVALID_ARG, "bad stream count");
    GateExcl gate(ix);  // (no search in flight while the stream mode changes)
    PVS_GATE_REFUSED(gate);
    for (pvs_index *sh : ix->shards) sh->multi_stream = n_streams > 1;
    ix->multi_stream = n_streams > 1;
    return PVS_OK;
}

PVS_EXPORT pvs_status pvs_index_set_path(pvs_index *ix, uint32_t path) {
    if (!ix || path > 2) return pvs_fail(PVS_ERR_INVALID_ARG, "bad path selector");
    for (pvs_index *sh : ix->shards) sh->forced_path = path;
    ix->forced_path = path;
    return PVS_OK;
}

PVS_EXPORT pvs_status pvs_index_scan_kernel_name(pvs_index *ix, uint32_t batch, char *out, uint32_t out_len) {
    if (!ix || !out || out_len < 2 || batch < 1) return pvs_fail(PVS_ERR_INVALID_ARG, "bad argument");
    const pvs_index *sh = is_multi(ix) && !ix->shards.empty() ? ix->shards[0] : ix;
    const uint32_t ks = sh->stride / PVS_KSLAB_BYTES;
    const char *dtn = sh->dtype == PVS_I8 ? "i8" : sh->dtype == PVS_F16 ? "f16" : "f32";
    if (batch <= PVS_DIRECT_MAX_NQ && !is_multi(ix) && pvs_direct_route(sh, 10, batch)) {  // (the API's default page: search_enqueue decides per k)
        snprintf(out, out_len, "k_direct_topk<%s, %u B, %u queries> (one launch: exact distances + pages)", dtn, sh->stride, batch <= 1 ? 1u : batch <= 2 ? 2u : batch <= 4 ? 4u : 8u);
        return PVS_OK;
    }
    if (!pvs_scan_supported((int)sh->dtype, ks)) {
        snprintf(out, out_len, "dense path");
        return PVS_OK;
    }
    const uint32_t pass = std::min<uint32_t>(batch, pvs_scan_max_batch((int)sh->dtype, ks));
    const uint32_t pad = pass <= 32 ? 32 : pass <= 64 ? 64 : pass <= 128 ? 128 : 256;
    const char *dt = sh->dtype == PVS_I8 ? "i8" : sh->dtype == PVS_F16 ? "f16" : "f32";
    snprintf(out, out_len, "%s<%s, %u B, %u queries>", pvs_scan_is_wide((int)sh->dtype, pad / 32, ks) ? "k_scan_wide" : "k_scan", dt, sh->stride, pad);
    return PVS_OK;
}

static const size_t STATS_V2_BYTES = offsetof(pvs_stats, rescanned_queries);  // what callers of the earlier struct hold
PVS_EXPORT pvs_status pvs_index_stats(pvs_index *ix, pvs_stats *out) { return pvs_index_stats_ex(ix, out, STATS_V2_BYTES); }
PVS_EXPORT pvs_status pvs_index_stats_ex(pvs_index *ix, pvs_stats *out, size_t out_bytes) {
    if (!ix || !out) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (out_bytes < STATS_V2_BYTES) return pvs_fail(PVS_ERR_INVALID_ARG, "pvs_stats: out_bytes smaller than the ABI v2 struct");
    if (is_multi(ix)) return multi_stats(ix, out, out_bytes);
    pvs_stats s;
    memset(&s, 0, sizeof s);
    s.struct_size = sizeof s;
    s.dtype = ix->dtype;
    s.dim = ix->dim;
    s.rows = ix->n;
    s.capacity_rows = ix->cap;
    s.row_stride_bytes = ix->stride;
    s.hbm_bytes = ix->cap * ((uint64_t)ix->stride + 16 + 2 * PVS_AUX_REC * 4 / 32);
    s.scale = ix->scale_set ? ix->scale : 0.f;
    s.searches = ix->searches.load();
    s.fast_queries = ix->fast_queries.load();
    s.dense_queries = ix->dense_queries.load();
    s.last_candidates = ix->last_candidates.load();
    s.rescanned_queries = ix->flat_reruns.load();
    s.sparse_queries = ix->sparse_queries.load();
    s.null_tail_queries = ix->null_tail_queries.load();
    const size_t want = std::min(out_bytes, sizeof s);
    s.struct_size = (uint32_t)want;
    memcpy(out, &s, want);
    return PVS_OK;
}

PVS_EXPORT pvs_status pvs_device_mem_info(int32_t device, uint64_t *free_bytes, uint64_t *total_bytes) {
    if (!free_bytes || !total_bytes) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    int dev = 0;
    PVS_TRY(use_device(device, &dev));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    *free_bytes = f;
    *total_bytes = t;
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
    GateShared gate(ix);
    if (is_multi(ix)) return multi_read_ids(ix, row0, n, out_row_ids, out_group_ids);
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
    GateShared gate(ix);
    if (is_multi(ix)) return multi_read_rows(ix, row0, n, out_host);
    if (row0 + n > ix->n) return pvs_fail(PVS_ERR_INVALID_ARG, "row range [%llu, %llu) exceeds %llu rows", (unsigned long long)row0,
                                          (unsigned long long)(row0 + n), (unsigned long long)ix->n);
    if (n == 0) return PVS_OK;
    HIP_TRY(hipSetDevice(ix->device));
    const size_t w = (size_t)ix->dim * ix->esz;
    const uint64_t chunk = std::max<uint64_t>(1, (256ull << 20) / w);
    uint8_t *stage = nullptr;
    HIP_TRY(pvs_malloc_retry((void **)&stage, std::min(chunk, n) * w));
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
    for (pvs_index *sh : ix->shards) sh->profiling = enable != 0;
    ix->profiling = enable != 0;
    return PVS_OK;
}
PVS_EXPORT pvs_status pvs_index_get_profile(pvs_index *ix, pvs_profile *out, int32_t reset) {
    if (!ix || !out) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (is_multi(ix)) {  // sums over the shards: launches / ms stay per-device averages
        pvs_profile tot{};
        for (pvs_index *sh : ix->shards) {
            pvs_profile p;
            PVS_TRY(pvs_index_get_profile(sh, &p, reset));
            tot.scan_launches += p.scan_launches;
            tot.scan_ms += p.scan_ms;
            tot.scan_rows += p.scan_rows;
            tot.sample_launches += p.sample_launches;
            tot.sample_ms += p.sample_ms;
            tot.finalize_launches += p.finalize_launches;
            tot.finalize_ms += p.finalize_ms;
            tot.exchange_launches += p.exchange_launches;
            tot.exchange_ms += p.exchange_ms;
        }
        tot.struct_size = sizeof(pvs_profile);
        *out = tot;
        return PVS_OK;
    }
    std::lock_guard<std::mutex> lk(ix->prof_mu);
    *out = ix->prof;
    out->struct_size = sizeof(pvs_profile);
    if (reset) ix->prof = pvs_profile{};
    return PVS_OK;
}

// ------------------------------------------------------- codec on the device
template <typename Fn>
static pvs_status with_device_chunks(const float *x, uint64_t n, pvs_space space, Fn &&fn) {
    if (space == PVS_DEVICE) return fn(x, n, (uint64_t)0);
    const uint64_t chunk = 64ull << 20;  // elements (256 MiB)
    float *stage = nullptr;
    HIP_TRY(pvs_malloc_retry((void **)&stage, std::min(chunk, n) * 4));
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
    HIP_TRY(pvs_malloc_retry((void **)&d_out, 4));
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
    HIP_TRY(pvs_malloc_retry((void **)&d_out, std::min(chunk, std::max<uint64_t>(n, 1))));
    pvs_status st = with_device_chunks(x, n, space, [&](const float *d, uint64_t m, uint64_t off) -> pvs_status {
        HIP_TRY(pvs_launch_quantize_flat(d, m, scale, d_out, nullptr));
        HIP_TRY(hipMemcpy(out + off, d_out, m, hipMemcpyDeviceToHost));
        return PVS_OK;
    });
    hipFree(d_out);
    return st;
}

PVS_EXPORT pvs_status pvs_merge_topk_keyed_device(int32_t device, const int64_t *d_ids, const float *d_dist, const int64_t *d_keys,
                                                  const uint32_t *d_counts, uint32_t world, uint32_t batch, uint32_t k, int64_t *d_out_ids,
                                                  float *d_out_dist, uint32_t *d_out_count) {
    if (!d_ids || !d_dist || !d_counts || !d_out_ids || !d_out_dist || !d_out_count) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    if (world == 0 || batch == 0 || k == 0) return pvs_fail(PVS_ERR_INVALID_ARG, "empty merge");
    PVS_TRY(use_device(device, nullptr));
    HIP_TRY(pvs_launch_merge(d_ids, d_dist, d_counts, world, batch, k, d_out_ids, d_out_dist, d_out_count, nullptr, d_keys));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return PVS_OK;
}
PVS_EXPORT pvs_status pvs_merge_topk_device(int32_t device, const int64_t *d_ids, const float *d_dist, const uint32_t *d_counts,
                                            uint32_t world, uint32_t batch, uint32_t k, int64_t *d_out_ids, float *d_out_dist,
                                            uint32_t *d_out_count) {
    return pvs_merge_topk_keyed_device(device, d_ids, d_dist, nullptr, d_counts, world, batch, k, d_out_ids, d_out_dist, d_out_count);
}

// ------------------------------------------------ device memory + synthetic
PVS_EXPORT pvs_status pvs_device_malloc(int32_t device, size_t bytes, void **out) {
    if (!out) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    PVS_TRY(use_device(device, nullptr));
    HIP_TRY(pvs_malloc_retry(out, bytes ? bytes : 16));
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

PVS_EXPORT pvs_status pvs_synth_rows_clustered_f32(int32_t device, uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim, float *d_out) {
    if (!d_out) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    PVS_TRY(use_device(device, nullptr));
    HIP_TRY(pvs_launch_synth_clustered(seed, row0, n, dim, d_out, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return PVS_OK;
}

// exposed to pvs_comm.hip
pvs_status pvs_index_internal_(pvs_index *ix, int *device) {
    if (!ix) return pvs_fail(PVS_ERR_INVALID_ARG, "null index");
    *device = ix->device;
    return PVS_OK;
}
