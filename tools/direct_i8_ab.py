"""One-launch search over int8 rows: the register-staged kernel (k_direct_topk_i8r, k <= 128) against the LDS-staged one
(pvs_debug_set("direct_lds_i8", 1)), p50 of pvs_search, alternated."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
rng = np.random.default_rng(1)
for D in (768, 512, 1024):
    for N in (10_000, 100_000, 690_000, 4_000_000, 10_000_000) if D == 768 else (690_000,):
        ix = pvs.VectorIndex(pvs.I8, D, capacity_rows=N)
        ix.set_scale(1.0 / 127 * 0.2)
        ch = min(N, 1_000_000)
        stage = pvs.DeviceBuffer(ch * D * 4)
        for off in range(0, N, ch):
            L.check(lib.pvs_synth_rows_f32(0, 1, off, min(ch, N - off), D, stage.ptr))
            ix.add_f32((stage, min(ch, N - off)))
        stage.free()
        q = rng.standard_normal((64, 1, D)).astype(np.float32)
        pvs.debug_set("direct_max_mb", 1 << 20)
        for k in (10, 100):
            row = {}
            for lds in (1, 0, 1, 0):
                pvs.debug_set("direct_lds_i8", lds)
                for i in range(8):
                    ix.search(q[i], k, pvs.COSINE)
                ts = []
                for i in range(100):
                    t = time.perf_counter()
                    ix.search(q[i % 64], k, pvs.COSINE)
                    ts.append(time.perf_counter() - t)
                row.setdefault(lds, []).append(round(float(np.sort(ts)[50]) * 1e3, 4))
            pvs.debug_set("direct_lds_i8", 0)
            print(f"i8 dim {D} N={N} k={k}: LDS-staged {row[1]}  register-staged {row[0]}", flush=True)
        pvs.debug_set("direct_max_mb", 0)
        ix.close()
