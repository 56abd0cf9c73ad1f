"""pvs_search p50 with the kernel's pinned flag words polled (default) against the stream's completion event (no_flag_poll=1)."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
rng = np.random.default_rng(1)
for name, dt, N, D in (("i8", pvs.I8, 10_000, 768), ("i8", pvs.I8, 690_000, 768), ("f32", pvs.F32, 690_000, 768)):
    ix = pvs.VectorIndex(dt, D, capacity_rows=N)
    if dt == pvs.I8:
        ix.set_scale(1.0 / 127 * 0.2)
    st = pvs.DeviceBuffer(N * D * 4)
    L.check(lib.pvs_synth_rows_f32(0, 1, 0, N, D, st.ptr))
    ix.add_f32((st, N))
    st.free()
    q = rng.standard_normal((64, 4, D)).astype(np.float32)
    for nb in (1, 4):
        for rep in range(2):
            for mode in (1, 0):
                pvs.debug_set("no_flag_poll", mode)
                for i in range(20):
                    ix.search(q[i][:nb], 10, pvs.COSINE)
                ts = []
                for i in range(400):
                    t = time.perf_counter()
                    ix.search(q[i % 64][:nb], 10, pvs.COSINE)
                    ts.append(time.perf_counter() - t)
                ts = np.sort(ts) * 1e3
                print(f"{name} N={N} batch {nb}: {'event wait' if mode else 'flag poll '} p50 {ts[200]:.4f} p99 {ts[395]:.4f}", flush=True)
    pvs.debug_set("no_flag_poll", 0)
    ix.close()
