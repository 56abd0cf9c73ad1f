#!/bin/bash
# GPU box: the headline and the 256-query pass, quick (no CPU baseline, 2 oracle queries).  Usage: tools/r3_quick.sh <tag> [env...]
set -u
tag=${1:-r3q}; shift; O=gpurun_out/$tag; mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[1], d["value"], "q/s  step", d["ms_per_step"], "ms  scan", r["avg_launch_ms"], "ms  passA", r.get("sample_pass_avg_ms"), "passC", r.get("finalize_avg_ms"), "parity", (d.get("parity") or {}).get("ids_and_distances_bit_exact"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
for rep in 1 2; do
for b in 128 256; do
  env "$@" timeout 300 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-peaks --check-queries 2 > $O/b${b}_$rep.json 2> $O/b${b}_$rep.err || tail -3 $O/b${b}_$rep.err
  line $O/b${b}_$rep.json
done
done
