"""Phase times of k_direct_topk (build with PVS_FLAGS_pvs_direct_i8=-DPVS_DIR_PROF ...): the spread of every workgroup's start /
queries in LDS / stream end / publish, and the finaliser's end.  Knobs: direct_static_pct, direct_unit (pvs_debug_set)."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
cases = (("i8", pvs.I8, 690_000, 768, 10, 1), ("i8", pvs.I8, 690_000, 768, 100, 1), ("i8", pvs.I8, 690_000, 768, 10, 4), ("i8", pvs.I8, 690_000, 768, 10, 8),
         ("i8", pvs.I8, 10_000, 768, 10, 1), ("f32", pvs.F32, 690_000, 768, 10, 1), ("f32", pvs.F32, 690_000, 768, 10, 4))
for name, dt, N, D, k, nb in cases:
    ix = pvs.VectorIndex(dt, D, capacity_rows=N)
    if dt == pvs.I8:
        ix.set_scale(1.0 / 127 * 0.2)
    st = pvs.DeviceBuffer(N * D * 4)
    L.check(lib.pvs_synth_rows_f32(0, 1, 0, N, D, st.ptr))
    ix.add_f32((st, N))
    st.free()
    q = np.random.default_rng(1).standard_normal((nb, D)).astype(np.float32)
    for pct in (0,):
        pvs.debug_set("direct_static_pct", pct)
        print(name, N, D, "k", k, "batch", nb, "static_pct", pct, flush=True)
        for _ in range(4):
            ix.search(q, k, pvs.COSINE)
    pvs.debug_set("direct_static_pct", 0)
    ix.close()
