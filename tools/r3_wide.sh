#!/bin/bash
# GPU box: parity suite + the 256-query pass at k = 100 / 1 (cosine, L2) + the 128-query headline.  Usage: tools/r3_wide.sh <tag>
set -u
tag=${1:-r3a}; O=gpurun_out/$tag; mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[1], d["value"], "q/s  step", d["ms_per_step"], "ms  scan", r["avg_launch_ms"], "ms  frac", r["frac"], "parity", d.get("parity", {}).get("ids_and_distances_bit_exact"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
(timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "rc=$?" >> $O/pytest_parity.log); tail -4 $O/pytest_parity.log
for spec in "b256_k100:--batch 256" "b256_k1:--batch 256 --k 1" "b256_l2:--batch 256 --metric l2" "b128:"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 400 python bench.py $args --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/$name.json 2> $O/$name.err || tail -3 $O/$name.err
  line $O/$name.json
done
