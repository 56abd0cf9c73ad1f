#!/bin/bash
# GPU box: parity suite + the 128-query headline on k_scan_wide / k_scan + the 256-query pass.  Usage: tools/r3_head.sh <tag>
set -u
tag=${1:-r3h}; O=gpurun_out/$tag; mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[1], d["value"], "q/s  step", d["ms_per_step"], "ms  scan", r["avg_launch_ms"], "ms  frac", r["frac"], "passA", r.get("sample_pass_avg_ms"), "passC", r.get("finalize_avg_ms"), "parity", (d.get("parity") or {}).get("ids_and_distances_bit_exact"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
(timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "rc=$?" >> $O/pytest_parity.log); tail -2 $O/pytest_parity.log | head -1
for spec in "b128:" "b128_old:--debug scan_no_wide128=1" "b128_l2:--metric l2" "b256:--batch 256" "b128_again:" "b128_old_again:--debug scan_no_wide128=1"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 400 python bench.py --no-secondary $args --steps 30 --warmup 5 --no-cpu-baseline --no-peaks > $O/$name.json 2> $O/$name.err || tail -3 $O/$name.err
  line $O/$name.json
done
