// pvs_comm.hip — multi-GPU: one process per GPU, the corpus row-sharded across ranks,
// per-shard pages exchanged by ONE RCCL all-gather (one packed record per rank) over xGMI and merged on every rank
// (SURVEY.md §8e).  The reference has no distributed code at all; this is the build's
// only collective.  The payload is tiny (batch*k*16 B per rank, <= 410 KB at 256x100), so
// the step is latency-bound, not the per-link 153 GB/s ring bound.
//
// RCCL is bound at run time with dlopen so that libpvs.so loads on machines without RCCL
// (single-GPU hosts never touch it).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "pvs_kernels.hpp"

namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
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
};

static_assert(sizeof(ncclUniqueId) == PVS_UNIQUE_ID_BYTES, "ncclUniqueId size");

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
    PVS_TRY(load_rccl());
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= n) return pvs_fail(PVS_ERR_INVALID_ARG, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    pvs_comm *c = new (std::nothrow) pvs_comm();
    if (!c) return pvs_fail(PVS_ERR_OOM, "host allocation failed");
    c->world = world;
    c->rank = rank;
    c->device = device;
    ncclUniqueId u;
    memcpy(&u, id, PVS_UNIQUE_ID_BYTES);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return pvs_fail(PVS_ERR_COMM, "ncclCommInitRank: %s", g_rccl.GetErrorString(r));
    }
    *out = c;
    return PVS_OK;
}

PVS_EXPORT void pvs_comm_destroy(pvs_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    delete c;
}

// ---- internal interface used by pvs_api.hip (stream-ordered sharded search)
int pvs_comm_world_(pvs_comm *c) { return c->world; }
int pvs_comm_device_(pvs_comm *c) { return c->device; }
// ONE all-gather of every rank's packed record (row pages: pvs_page_record_*; per-item pages: groups | values | counts) on `s`
pvs_status pvs_comm_gather_records_(pvs_comm *c, const void *rec, void *all_rec, size_t rec_bytes, hipStream_t s) {
    NCCL_TRY(g_rccl.AllGather(rec, all_rec, rec_bytes, ncclInt8, c->comm, s));
    return PVS_OK;
}
// max over the ranks of one float per rank, in place on the device (the int8 scale of a sharded space: absmax of the shards)
pvs_status pvs_comm_allreduce_max_(pvs_comm *c, float *d_inout, uint64_t n, hipStream_t s) {
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
        ncclResult_t r = g_rccl.AllReduce(d, d, n, ncclUint64, op == 0 ? ncclMin : ncclSum, c->comm, nullptr);
        if (r != ncclSuccess) st = pvs_fail(PVS_ERR_COMM, "ncclAllReduce: %s", g_rccl.GetErrorString(r));
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
        if (st == PVS_OK) e = hipMemcpy(inout, d, 4, hipMemcpyDeviceToHost);  // (synchronises with the null stream's collective)
    }
    pvs_scratch_free_on(d, nullptr);
    if (st == PVS_OK && e != hipSuccess) st = pvs_fail(PVS_ERR_DEVICE, "allreduce max: %s", hipGetErrorString(e));
    return st;
}
