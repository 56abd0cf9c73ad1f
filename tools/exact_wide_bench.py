"""k_exact_wide (16 / 32 float queries per pass, query components as scalar operands) against k_dense_exact's 8 per pass
(pvs_debug_set("no_exact_wide", 1)) through pvs_score_batch into device memory.  Usage: python tools/exact_wide_bench.py [rows] [out.json]"""
import ctypes as C, json, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L
lib = pvs.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
D = 768
res = {"rows": N, "dim": D}
for name, dt in (("f16", pvs.F16), ("f32", pvs.F32)):
    ix = pvs.VectorIndex(dt, D, capacity_rows=N)
    stage = pvs.DeviceBuffer(1_000_000 * D * 4)
    for off in range(0, N, 1_000_000):
        m = min(1_000_000, N - off)
        L.check(lib.pvs_synth_rows_f32(0, 1, off, m, D, stage.ptr))
        ix.add_f32((stage, m))
    stage.free()
    q = np.random.default_rng(1).standard_normal((32, D)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    out = pvs.DeviceBuffer(N * 32 * 4)
    for metric, mn in ((pvs.COSINE, "cosine"), (pvs.L2, "l2")):
        for nb in (16, 32):
            row, bits = {}, {}
            for off_ in (1, 0):
                pvs.debug_set("no_exact_wide", off_)
                for rep in range(2):
                    t = time.perf_counter()
                    for _ in range(3):
                        L.check(lib.pvs_score_batch(ix._h, q.ctypes.data, L.F32, nb, metric, C.c_void_p(out.ptr), L.DEVICE))
                    ms = (time.perf_counter() - t) / 3 * 1e3
                row["lds8_ms" if off_ else "wide_ms"] = round(ms, 3)
                bits[off_] = out.to_numpy(np.uint32, (N * nb,))[: 1 << 22].copy()
            pvs.debug_set("no_exact_wide", 0)
            row["same_bits"] = bool(np.array_equal(bits[0], bits[1]))
            ops = 2 if metric == pvs.COSINE else 3
            row["valu_floor_ms"] = round(N * D * nb / 2 * ops / 64 * 4 / (1024 * 2.4e9) * 1e3, 3)  # packed instructions x 4 cycles on 1,024 SIMDs at 2.4 GHz
            if nb == 32 and len(sys.argv) > 3:  # the clock the board holds under this kernel (packed f32 at full rate draws power)
                from bench import sample_clock_and_power
                ul = sample_clock_and_power(lambda i: L.check(lib.pvs_score_batch(ix._h, q.ctypes.data, L.F32, nb, metric, C.c_void_p(out.ptr), L.DEVICE)), lambda: None)
                row["sclk_mhz"], row["socket_power_w"] = ul.get("sclk_mhz"), ul.get("socket_power_w")
                if ul.get("sclk_mhz"):
                    row["valu_floor_at_clock_ms"] = round(row["valu_floor_ms"] * 2400 / ul["sclk_mhz"], 3)
            res[f"{name}_{mn}_b{nb}"] = row
            print(name, mn, nb, row, flush=True)
    out.free()
    ix.close()
line = json.dumps(res)
print(line)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(line + "\n")
