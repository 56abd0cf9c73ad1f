#!/bin/bash
# GPU box: pass-A sample fraction sweep (PVS_SAMPLE_DIV) at 128 and 256 queries.  Usage: tools/r3_sample_sweep.sh <tag>
set -u
tag=${1:-r3s}; O=gpurun_out/$tag; mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[1], d["value"], "q/s  step", d["ms_per_step"], "ms  scan", r["avg_launch_ms"], "ms  passA", r.get("sample_pass_avg_ms"), "passC", r.get("finalize_avg_ms"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
for b in 128 256; do
  for div in 8 12 16 24 32 48; do
    timeout 300 python bench.py --debug sample_div=$div --no-secondary --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-peaks --no-verify > $O/b${b}_div$div.json 2> $O/b${b}_div$div.err || tail -3 $O/b${b}_div$div.err
    line $O/b${b}_div$div.json
  done
done
