"""Create / use / destroy large indexes repeatedly: a leak of device memory shows up as an allocation failure."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
N, D = 8_000_000, 768
q = np.random.default_rng(1).standard_normal((8, D)).astype(np.float32)
stage = pvs.DeviceBuffer(1_000_000 * D * 4)
import ctypes as C
def free_bytes():
    f, t = C.c_uint64(), C.c_uint64()
    L.check(lib.pvs_device_synchronize(0)); L.check(lib.pvs_device_mem_info(0, C.byref(f), C.byref(t)))
    return f.value
free0 = None
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    dt = [pvs.I8, pvs.F16, pvs.F32][it % 3]
    n = N if dt != pvs.F32 else N // 2
    ix = pvs.VectorIndex(dt, D, capacity_rows=n)
    if dt == pvs.I8:
        ix.set_scale(1.0 / 127 * 0.2)
    for off in range(0, n, 1_000_000):
        L.check(lib.pvs_synth_rows_f32(0, 1, off, 1_000_000, D, stage.ptr))
        g = (np.arange(off, off + 1_000_000, dtype=np.int64) // 5)
        L.check(lib.pvs_index_add_f32(ix._h, stage.ptr, 1_000_000, None, g.ctypes.data, L.DEVICE))
    ix.search(q, 10, pvs.COSINE)
    ix.set_path(1); ix.search(q[:1], 10, pvs.L2); ix.set_path(0)
    ix.search_groups(q[:2], 5, pvs.COSINE, pvs.AGG_AVG)
    ix.score_all(q[0], pvs.COSINE)
    m = (np.arange(n) % 3 == 0).astype(np.uint8)
    ix.search_filtered(q[:2], 5, m, pvs.COSINE)
    pvs.rrf_search([dict(index=ix, query=q[0], metric=pvs.COSINE), dict(index=ix, query=q[1], metric=pvs.L2)], 5)
    ix.similar_to(ix.read_ids(0, 3), 5)
    ix.close()
    if it == 5:
        free0 = free_bytes()  # after the first rounds (allocator pools, RCCL-free caches warmed up)
    if it % 10 == 9:
        print(f"iteration {it + 1} ok, free HBM {free_bytes() / 2**30:.2f} GiB", flush=True)
drift = free0 - free_bytes()
print(f"free-memory drift since iteration 6: {drift / 2**20:.1f} MiB")
sys.exit(1 if drift > 256 * 2**20 else 0)
