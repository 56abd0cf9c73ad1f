"""How often does the filter path hand a query back (dense path / flat rescan)?  Random queries against synthetic corpora."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
for N, D, dt in ((10_000_000, 768, pvs.I8), (1_000_000, 768, pvs.F16), (300_000, 512, pvs.F32)):
    ix = pvs.VectorIndex(dt, D, capacity_rows=N)
    if dt == pvs.I8:
        ix.set_scale(1.0 / 127 * 0.2)
    step = min(N, 1_000_000)
    stage = pvs.DeviceBuffer(step * D * 4)
    for off in range(0, N, step):
        L.check(lib.pvs_synth_rows_f32(0, 7, off, step, D, stage.ptr))
        ix.add_f32((stage, step))
    stage.free()
    rng = np.random.default_rng(3)
    for B, k in ((128, 100), (32, 100), (1, 100), (128, 10), (128, 1000)):
        s0 = ix.stats()
        reps = 40
        for _ in range(reps):
            q = rng.standard_normal((B, D)).astype(np.float32)
            ix.search(q, k, pvs.COSINE)
        s1 = ix.stats()
        print(f"N={N} dtype={dt} batch={B} k={k}: {reps * B} queries, dense {s1.dense_queries - s0.dense_queries}, rescanned {s1.rescanned_queries - s0.rescanned_queries}", flush=True)
    ix.close()
