#!/bin/bash
# GPU box: the one-launch search's tests (short timeouts: a hung kernel must not eat the lease), its latency table, then its phase
# profile (a -DPVS_DIR_PROF build made on the box).  Stops at the first failing step.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; out=$R/gpurun_out/r5; mkdir -p $out
cd $R
timeout 120 python - > $out/quick.out 2>&1 <<'PY'
import sys, time; sys.path.insert(0, '.')
import numpy as np, oracle as orc, panoptikon_amd as pvs
for dt, n, dim, nb, k in ((pvs.F32, 10000, 512, 1, 10), (pvs.I8, 100000, 768, 1, 10), (pvs.I8, 100000, 768, 4, 10), (pvs.I8, 690000, 768, 8, 10), (pvs.F32, 200000, 768, 3, 50)):
    rows = orc.synth_rows(1, 0, n, dim); q = orc.synth_rows(2, 0, nb, dim)
    scale = orc.compute_int8_scale(rows)
    ix = pvs.VectorIndex(dt, dim)
    if dt == pvs.I8: ix.set_scale(scale)
    ix.add_f32(rows)
    hc = orc.quantize_int8(rows, scale) if dt == pvs.I8 else rows
    hq = orc.quantize_int8(q, scale) if dt == pvs.I8 else q
    t = time.time(); gi, gd, gc = ix.search(q, k, pvs.COSINE); t = time.time() - t
    ei, ed = orc.search(dt, pvs.COSINE, hc, hq, k, threads=8)
    ok = np.array_equal(gi, ei) and np.array_equal(gd.view(np.uint32), ed.view(np.uint32))
    print(dt, n, dim, nb, k, "ok" if ok else "MISMATCH", f"{t*1e3:.2f} ms", "direct", pvs.debug_get("direct_queries"), flush=True)
    ix.close()
PY
rc=$?; cat $out/quick.out | tail -8; echo "quick rc $rc"
[ $rc -ne 0 ] && exit 1
grep -q MISMATCH $out/quick.out && exit 1
timeout 900 python -m pytest tests/test_gpu_direct.py tests/test_gpu_advice_r5.py -x -q --timeout 120 > $out/t_direct.out 2>&1; rc=$?; echo "tests rc $rc" >> $out/t_direct.out
tail -15 $out/t_direct.out
[ $rc -ne 0 ] && exit 1
timeout 300 python tools/latency_small.py $out/latency_small.json > $out/latency_small.out 2>&1
grep -E "690000|N=10000 " $out/latency_small.out
cp panoptikon_amd/libpvs.so /tmp/libpvs_keep.so
PVS_FLAGS_pvs_direct_i8=-DPVS_DIR_PROF PVS_FLAGS_pvs_direct_f16=-DPVS_DIR_PROF PVS_FLAGS_pvs_direct_f32=-DPVS_DIR_PROF python - <<'PY' > $out/prof_build.out 2>&1
import sys; sys.path.insert(0, '.')
from panoptikon_amd import build as b
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(3) as ex: list(ex.map(lambda s: b._compile(s, force=True), ["pvs_direct_i8.hip", "pvs_direct_f16.hip", "pvs_direct_f32.hip"]))
print(b.build())
PY
tail -2 $out/prof_build.out
timeout 200 python tools/direct_prof.py > $out/direct_prof.out 2>&1
cp /tmp/libpvs_keep.so panoptikon_amd/libpvs.so
grep -v "^$" $out/direct_prof.out | cut -c1-420 | tail -60
