// pvs_comm.hip — multi-GPU: one process per GPU, the corpus row-sharded across ranks,
// per-shard pages exchanged by ONE RCCL all-gather over xGMI and merged on every rank
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
// one grouped all-gather of (ids i64, dist f32, counts u32, flags u32) on `s`
pvs_status pvs_comm_gather_pages_(pvs_comm *c, const int64_t *ids, const float *dist, const uint32_t *cnt, const uint32_t *flags,
                                  int64_t *all_ids, float *all_dist, uint32_t *all_cnt, uint32_t *all_flags, uint64_t elems,
                                  uint32_t batch, hipStream_t s) {
    NCCL_TRY(g_rccl.GroupStart());
    NCCL_TRY(g_rccl.AllGather(ids, all_ids, elems, ncclInt64, c->comm, s));
    NCCL_TRY(g_rccl.AllGather(dist, all_dist, elems, ncclFloat32, c->comm, s));
    NCCL_TRY(g_rccl.AllGather(cnt, all_cnt, batch, ncclUint32, c->comm, s));
    NCCL_TRY(g_rccl.AllGather(flags, all_flags, batch, ncclUint32, c->comm, s));
    NCCL_TRY(g_rccl.GroupEnd());
    return PVS_OK;
}

// (group id i64, value f64, count u32) pages of a per-item search
pvs_status pvs_comm_gather_group_pages_(pvs_comm *c, const int64_t *groups, const double *values, const uint32_t *cnt, int64_t *all_groups,
                                        double *all_values, uint32_t *all_cnt, uint64_t elems, uint32_t batch, hipStream_t s) {
    NCCL_TRY(g_rccl.GroupStart());
    NCCL_TRY(g_rccl.AllGather(groups, all_groups, elems, ncclInt64, c->comm, s));
    NCCL_TRY(g_rccl.AllGather(values, all_values, elems, ncclFloat64, c->comm, s));
    NCCL_TRY(g_rccl.AllGather(cnt, all_cnt, batch, ncclUint32, c->comm, s));
    NCCL_TRY(g_rccl.GroupEnd());
    return PVS_OK;
}
