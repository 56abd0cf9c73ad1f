// pvs_comm.hip — multi-GPU: one process per GPU, the corpus row-sharded across ranks,
// per-shard pages exchanged by ONE RCCL all-gather (one packed record per rank) over xGMI and merged on every rank
// (SURVEY.md §8e).  The reference has no distributed code at all; this is the build's
// only collective.  The payload is tiny (batch*k*16 B per rank, <= 410 KB at 256x100), so
// the step is latency-bound, not the per-link 153 GB/s ring bound.
//
// RCCL is bound at run time with dlopen so that libpvs.so loads on machines without RCCL
// (single-GPU hosts never touch it).
//
// Every wait for another rank is BOUNDED (round 6; pvs_debug_set("comm_timeout_s"), default 180 s): ncclCommInitRank runs on a
// helper thread the caller waits for with a deadline; communicator creation ends with a one-word all-reduce — the first
// collective of the communicator happens HERE, where a host can still agree with its peers on a fallback, not in the middle of
// the first search — waited for by polling the stream with the same deadline; pvs_wait polls a sharded search's completion event
// the same way (pvs_comm_wait_event_).  When a deadline passes the communicator is aborted (ncclCommAbort) and the call returns
// PVS_ERR_COMM with a message that names the rank: a rank that never arrives produces a return code, not a hung job.  A rank
// that fails LOCALLY before an exchange still takes part in it with a failure record (PVS_PAGE_FAILED in every flag word,
// pvs_search_device.hip), so that all ranks fail that search together instead of one leaving the others inside the all-gather.
// RCCL's own diagnostics: NCCL_DEBUG is set to WARN when the host has not chosen a level (RCCL prints to stdout: a host that
// owns stdout redirects fd 1 as bench.py does).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <thread>

#include "pvs_kernels.hpp"

namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // (optional)
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

pvs_status load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return PVS_OK;
    // ROCm's own RCCL first, by absolute path: a host process that imported PyTorch carries torch/lib/librccl.so
    // (another version), and a bare soname would resolve to that copy while libpvs runs on ROCm's HIP runtime.
    // PVS_RCCL_PATH overrides.
    const char *names[] = {getenv("PVS_RCCL_PATH"), "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
    (void)setenv("NCCL_DEBUG", "WARN", 0);  // (0: the host's own choice wins) a failing first run must say why
    void *h = nullptr;
    for (const char *n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) return pvs_fail(PVS_ERR_COMM, "cannot load RCCL: %s", dlerror());
    Rccl r;
    r.lib = h;
#define SYM(field, name)                                                        \
    r.field = (decltype(r.field))dlsym(h, name);                                \
    if (!r.field) return pvs_fail(PVS_ERR_COMM, "RCCL symbol %s missing", name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllGather, "ncclAllGather");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    r.CommAbort = (decltype(r.CommAbort))dlsym(h, "ncclCommAbort");
    g_rccl = r;
    return PVS_OK;
}
}  // namespace

#define NCCL_TRY(expr)                                                                                  \
    do {                                                                                                \
        ncclResult_t _r = (expr);                                                                       \
        if (_r != ncclSuccess) return pvs_fail(PVS_ERR_COMM, "%s: %s", #expr, g_rccl.GetErrorString(_r)); \
    } while (0)

struct pvs_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
    std::atomic<bool> aborted{false};
};

static_assert(sizeof(ncclUniqueId) == PVS_UNIQUE_ID_BYTES, "ncclUniqueId size");

static double comm_timeout_s() {
    const int64_t v = pvs_dbg(PVS_DBG_COMM_TIMEOUT_S);
    return v > 0 ? (double)v : 180.0;
}

// Tears the communicator down without waiting for the peers (after a deadline passed): collectives of this rank still queued
// complete with an error instead of waiting for ever.  The object stays valid for pvs_comm_destroy.
void pvs_comm_abort_(pvs_comm *c) {
    if (!c || c->aborted.exchange(true)) return;
    if (c->comm && g_rccl.CommAbort) (void)g_rccl.CommAbort(c->comm);
    c->comm = nullptr;
}

// Waits for `ev` (recorded behind a collective) with the communicator's deadline; on expiry aborts the communicator.
pvs_status pvs_comm_wait_event_(pvs_comm *c, hipEvent_t ev, const char *what) {
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = comm_timeout_s();
    for (uint32_t spins = 0;; spins++) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return PVS_OK;
        if (e != hipErrorNotReady) return pvs_fail(PVS_ERR_DEVICE, "%s: %s", what, hipGetErrorString(e));
        (void)hipGetLastError();  // (hipErrorNotReady is sticky)
        if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(spins > 20000 ? 1000 : 50));
        if ((spins & 255) == 255 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
            pvs_comm_abort_(c);
            return pvs_fail(PVS_ERR_COMM, "%s: rank %d of %d waited %.0f s for the other ranks (one never arrived, or failed before its collective); communicator aborted",
                            what, c->rank, c->world, limit);
        }
    }
}
pvs_status pvs_comm_wait_stream_(pvs_comm *c, hipStream_t s, const char *what) {
    hipEvent_t ev = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    pvs_status st = PVS_OK;
    const hipError_t e = hipEventRecord(ev, s);
    if (e != hipSuccess)
        st = pvs_fail(PVS_ERR_DEVICE, "hipEventRecord: %s", hipGetErrorString(e));
    else
        st = pvs_comm_wait_event_(c, ev, what);
    (void)hipEventDestroy(ev);
    return st;
}

PVS_EXPORT pvs_status pvs_comm_unique_id(uint8_t id[PVS_UNIQUE_ID_BYTES]) {
    if (!id) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    PVS_TRY(load_rccl());
    ncclUniqueId u;
    NCCL_TRY(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, PVS_UNIQUE_ID_BYTES);
    return PVS_OK;
}

PVS_EXPORT pvs_status pvs_comm_create(const uint8_t id[PVS_UNIQUE_ID_BYTES], int32_t world, int32_t rank, int32_t device,
                                      pvs_comm **out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return pvs_fail(PVS_ERR_INVALID_ARG, "bad communicator arguments");
    *out = nullptr;
    PVS_TRY(load_rccl());
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= n) return pvs_fail(PVS_ERR_INVALID_ARG, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    // ncclCommInitRank blocks until EVERY rank has called it: on a helper thread, waited for with the deadline (the thread is
    // left behind if it never comes back: there is no way to cancel it, and the host is about to give up on this run anyway)
    struct Init {
        std::mutex mu;
        std::condition_variable cv;
        bool done = false;
        ncclResult_t r = ncclSuccess;
        ncclComm_t comm = nullptr;
    };
    auto st = std::make_shared<Init>();
    ncclUniqueId u;
    memcpy(&u, id, PVS_UNIQUE_ID_BYTES);
    std::thread([st, u, world, rank, device]() {
        (void)hipSetDevice(device);
        ncclComm_t cm = nullptr;
        const ncclResult_t r = g_rccl.CommInitRank(&cm, world, u, rank);
        std::lock_guard<std::mutex> lk(st->mu);
        st->r = r;
        st->comm = cm;
        st->done = true;
        st->cv.notify_all();
    }).detach();
    {
        std::unique_lock<std::mutex> lk(st->mu);
        const double limit = comm_timeout_s();
        if (!st->cv.wait_for(lk, std::chrono::duration<double>(limit), [&] { return st->done; }))
            return pvs_fail(PVS_ERR_COMM, "ncclCommInitRank: rank %d of %d waited %.0f s for the other ranks to join the communicator", rank, world, limit);
        if (st->r != ncclSuccess) return pvs_fail(PVS_ERR_COMM, "ncclCommInitRank (rank %d of %d, device %d): %s", rank, world, device, g_rccl.GetErrorString(st->r));
    }
    pvs_comm *c = new (std::nothrow) pvs_comm();
    if (!c) {
        if (g_rccl.CommAbort) (void)g_rccl.CommAbort(st->comm);
        return pvs_fail(PVS_ERR_OOM, "host allocation failed");
    }
    c->comm = st->comm;
    c->world = world;
    c->rank = rank;
    c->device = device;
    // the communicator's FIRST collective, here: one word, sum over the ranks must be `world` (RCCL sets rings and buffers up lazily
    // — seconds on 8 GPUs — and a peer that died after joining shows up now, under the deadline)
    uint32_t *d_one = nullptr;
    hipStream_t hs = nullptr;
    auto handshake = [&]() -> pvs_status {
        HIP_TRY(hipStreamCreateWithFlags(&hs, hipStreamNonBlocking));
        HIP_TRY(hipMalloc((void **)&d_one, 4));
        const uint32_t one = 1;
        HIP_TRY(hipMemcpyAsync(d_one, &one, 4, hipMemcpyHostToDevice, hs));
        NCCL_TRY(g_rccl.AllReduce(d_one, d_one, 1, ncclUint32, ncclSum, c->comm, hs));
        PVS_TRY(pvs_comm_wait_stream_(c, hs, "communicator handshake (first all-reduce)"));
        uint32_t got = 0;
        HIP_TRY(hipMemcpy(&got, d_one, 4, hipMemcpyDeviceToHost));
        if (got != (uint32_t)world) return pvs_fail(PVS_ERR_COMM, "communicator handshake: %u ranks answered, %d expected", got, world);
        return PVS_OK;
    };
    const pvs_status hst = handshake();
    if (d_one) (void)hipFree(d_one);
    if (hs) (void)hipStreamDestroy(hs);
    if (hst != PVS_OK) {
        const std::string why = pvs_last_error();
        pvs_comm_abort_(c);
        delete c;
        return pvs_fail(hst, "%s", why.c_str());
    }
    *out = c;
    return PVS_OK;
}

PVS_EXPORT void pvs_comm_destroy(pvs_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (!c->aborted.load()) (void)hipDeviceSynchronize();
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    delete c;
}

// ---- internal interface used by pvs_api.hip (stream-ordered sharded search)
int pvs_comm_world_(pvs_comm *c) { return c->world; }
int pvs_comm_device_(pvs_comm *c) { return c->device; }
// ONE all-gather of every rank's packed record (row pages: pvs_page_record_*; per-item pages: groups | values | counts) on `s`
pvs_status pvs_comm_gather_records_(pvs_comm *c, const void *rec, void *all_rec, size_t rec_bytes, hipStream_t s) {
    if (c->aborted.load() || !c->comm) return pvs_fail(PVS_ERR_COMM, "the communicator was aborted after a rank failed to arrive: create a new one");
    NCCL_TRY(g_rccl.AllGather(rec, all_rec, rec_bytes, ncclInt8, c->comm, s));
    return PVS_OK;
}
// max over the ranks of one float per rank, in place on the device (the int8 scale of a sharded space: absmax of the shards)
pvs_status pvs_comm_allreduce_max_(pvs_comm *c, float *d_inout, uint64_t n, hipStream_t s) {
    if (c->aborted.load() || !c->comm) return pvs_fail(PVS_ERR_COMM, "the communicator was aborted after a rank failed to arrive: create a new one");
    NCCL_TRY(g_rccl.AllReduce(d_inout, d_inout, n, ncclFloat32, ncclMax, c->comm, s));
    return PVS_OK;
}

// ---- host-buffer collectives for the small control messages of the sharded fusion (pvs_rrf_sharded.hip): staged through the
// scratch cache, on the null stream, synchronous
pvs_status pvs_comm_allgather_host_(pvs_comm *c, const void *send, void *recv, size_t bytes) {
    if (bytes == 0) return PVS_OK;
    HIP_TRY(hipSetDevice(c->device));
    uint8_t *d = nullptr;
    HIP_TRY(pvs_scratch_alloc((void **)&d, bytes * (size_t)(c->world + 1)));
    pvs_status st = PVS_OK;
    hipError_t e = hipMemcpy(d, send, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        st = pvs_comm_gather_records_(c, d, d + bytes, bytes, nullptr);
        if (st == PVS_OK) st = pvs_comm_wait_stream_(c, nullptr, "all-gather of a control message");
        if (st == PVS_OK) e = hipMemcpy(recv, d + bytes, bytes * (size_t)c->world, hipMemcpyDeviceToHost);
    }
    pvs_scratch_free_on(d, nullptr);
    if (st == PVS_OK && e != hipSuccess) st = pvs_fail(PVS_ERR_DEVICE, "all-gather staging: %s", hipGetErrorString(e));
    return st;
}
pvs_status pvs_comm_allreduce_u64_host_(pvs_comm *c, uint64_t *inout, size_t n, int op) {
    if (n == 0) return PVS_OK;
    HIP_TRY(hipSetDevice(c->device));
    uint64_t *d = nullptr;
    HIP_TRY(pvs_scratch_alloc((void **)&d, n * 8));
    pvs_status st = PVS_OK;
    hipError_t e = hipMemcpy(d, inout, n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        if (c->aborted.load() || !c->comm) {
            st = pvs_fail(PVS_ERR_COMM, "the communicator was aborted after a rank failed to arrive: create a new one");
        } else {
            ncclResult_t r = g_rccl.AllReduce(d, d, n, ncclUint64, op == 0 ? ncclMin : ncclSum, c->comm, nullptr);
            if (r != ncclSuccess) st = pvs_fail(PVS_ERR_COMM, "ncclAllReduce: %s", g_rccl.GetErrorString(r));
        }
        if (st == PVS_OK) st = pvs_comm_wait_stream_(c, nullptr, "all-reduce of a control message");
        if (st == PVS_OK) e = hipMemcpy(inout, d, n * 8, hipMemcpyDeviceToHost);
    }
    pvs_scratch_free_on(d, nullptr);
    if (st == PVS_OK && e != hipSuccess) st = pvs_fail(PVS_ERR_DEVICE, "all-reduce staging: %s", hipGetErrorString(e));
    return st;
}

// ---- C ABI: the max over the ranks of a host float (compute_int8_scale_artifact over a sharded space: every rank passes the
// absmax of its shard, db/vector_quants.rs:1513-1554, and gets the space's)
PVS_EXPORT pvs_status pvs_comm_allreduce_max_f32(pvs_comm *c, float *inout) {
    if (!c || !inout) return pvs_fail(PVS_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    float *d = nullptr;
    HIP_TRY(pvs_scratch_alloc((void **)&d, 4));
    pvs_status st = PVS_OK;
    hipError_t e = hipMemcpy(d, inout, 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        st = pvs_comm_allreduce_max_(c, d, 1, nullptr);
        if (st == PVS_OK) st = pvs_comm_wait_stream_(c, nullptr, "all-reduce(max) of the shard absmax");
        if (st == PVS_OK) e = hipMemcpy(inout, d, 4, hipMemcpyDeviceToHost);
    }
    pvs_scratch_free_on(d, nullptr);
    if (st == PVS_OK && e != hipSuccess) st = pvs_fail(PVS_ERR_DEVICE, "allreduce max: %s", hipGetErrorString(e));
    return st;
}
