"""Host-buffer entry point (pvs_search: H2D queries, search, D2H page) latency, one caller thread."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
D = 768
rng = np.random.default_rng(1)
for name, dt, N in (("i8", pvs.I8, 10_000_000), ("i8", pvs.I8, 1_000_000), ("f32", pvs.F32, 1_000_000)):
    ix = pvs.VectorIndex(dt, D, capacity_rows=N)
    if dt == pvs.I8:
        ix.set_scale(1.0 / 127 * 0.2)
    stage = pvs.DeviceBuffer(1_000_000 * D * 4)
    for off in range(0, N, 1_000_000):
        L.check(lib.pvs_synth_rows_f32(0, 1, off, 1_000_000, D, stage.ptr))
        ix.add_f32((stage, 1_000_000))
    stage.free()
    for B in (1, 8, 128) if N == 10_000_000 else (1, 8):
        q = rng.standard_normal((64, B, D)).astype(np.float32)
        q /= np.linalg.norm(q, axis=2, keepdims=True)
        for i in range(5):
            ix.search(q[i], 100, pvs.COSINE)
        ts = []
        for i in range(200):
            t = time.perf_counter()
            ix.search(q[i % 64], 100, pvs.COSINE)
            ts.append(time.perf_counter() - t)
        ts = np.sort(np.array(ts)) * 1e3
        print(f"pvs_search {name} N={N} batch={B} k=100: p50 {ts[100]:.3f} ms  p90 {ts[180]:.3f} ms  p99 {ts[197]:.3f} ms  ({B/ts[100]*1e3:.0f} q/s serial)", flush=True)
    ix.close()
