"""similar_to on the device at the reference's measured scale (docs/or-composition-penalty.md:225: 9.5-31 s at ~690k vectors):
690,000 x 768 vectors (~8 per item), the target item's vectors against everything else, AVG per item (the reference's default),
f32 rows (exact mode) and int8 rows (quant mode).  Prints time per call and the fraction of HBM peak on the bytes that have to be read."""
import sys, time
sys.path.insert(0, "/root/repo")
import json
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
N, D, PER = 690_000, 768, 8
out = {}
for name, dt, esz in (("f32", pvs.F32, 4), ("f16", pvs.F16, 2), ("i8", pvs.I8, 1)):
    ix = pvs.VectorIndex(dt, D, capacity_rows=N)
    if dt == pvs.I8:
        ix.set_scale(0.0015)
    stage = pvs.DeviceBuffer(N * D * 4)
    L.check(lib.pvs_synth_rows_f32(0, 7, 0, N, D, stage.ptr))
    g = np.arange(N, dtype=np.int64) // PER
    L.check(lib.pvs_index_add_f32(ix._h, stage.ptr, N, None, g.ctypes.data, L.DEVICE))
    stage.free()
    targets = np.arange(8 * 1000, 8 * 1000 + PER, dtype=np.int64)  # every vector of item 1000
    for metric, mname in ((pvs.L2, "l2"), (pvs.COSINE, "cosine")):
        ix.similar_to(targets, 100, metric, pvs.AGG_AVG)
        ts = []
        for _ in range(10):
            t = time.perf_counter()
            gg, vv = ix.similar_to(targets, 100, metric, pvs.AGG_AVG)
            ts.append(time.perf_counter() - t)
        ms = float(np.median(ts)) * 1e3
        out[f"{name}_{mname}"] = {"ms_per_call": round(ms, 3), "corpus_GB": round(N * D * esz / 1e9, 3),
                                  "frac_of_8TBs": round(N * D * esz / (ms * 1e-3) / 8e12, 4), "first": [int(gg[0]), float(vv[0])]}
    ix.close()
out["reference"] = "9.5-31 s per similar_to at ~690k vectors (docs/or-composition-penalty.md:225), SQLite self-join"
print(json.dumps(out, indent=1))
