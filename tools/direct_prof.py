"""Phase times of the LAST workgroup of k_direct_topk (build with PVS_FLAGS_pvs_direct=-DPVS_DIR_PROF)."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
for name, dt, N, D, k in (("f32", pvs.F32, 10_000, 512, 10), ("f32", pvs.F32, 10_000, 512, 100), ("i8", pvs.I8, 690_000, 768, 10), ("i8", pvs.I8, 690_000, 768, 100)):
    ix = pvs.VectorIndex(dt, D, capacity_rows=N)
    if dt == pvs.I8:
        ix.set_scale(1.0 / 127 * 0.2)
    st = pvs.DeviceBuffer(N * D * 4)
    L.check(lib.pvs_synth_rows_f32(0, 1, 0, N, D, st.ptr))
    ix.add_f32((st, N))
    st.free()
    q = np.random.default_rng(1).standard_normal((1, D)).astype(np.float32)
    print(name, N, D, k, flush=True)
    for _ in range(4):
        ix.search(q, k, pvs.COSINE)
    ix.close()
