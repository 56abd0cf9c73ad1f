#!/bin/bash
# GPU box: threshold rank / sample size sweep (PVS_SAMPLE_J_DIV x PVS_SAMPLE_DIV).  Usage: tools/r3_jsweep.sh <tag>
set -u
tag=${1:-r3j}; O=gpurun_out/$tag; mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[1], d["value"], "q/s  step", d["ms_per_step"], "ms  scan", r["avg_launch_ms"], "ms  passA", r.get("sample_pass_avg_ms"), "passC", r.get("finalize_avg_ms"), "parity", (d.get("parity") or {}).get("ids_and_distances_bit_exact"), "dense", (d.get("parity") or {}).get("dense_queries"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
for b in 128 256; do
  for spec in "1:16" "2:16" "4:16" "8:16" "4:12" "4:24" "12:16"; do
    j=${spec%%:*}; div=${spec#*:}
    timeout 300 python bench.py --debug sample_j_div=$j --debug sample_div=$div --no-secondary --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-peaks --check-queries 2 > $O/b${b}_j${j}_div$div.json 2> $O/b${b}_j${j}_div$div.err || tail -3 $O/b${b}_j${j}_div$div.err
    line $O/b${b}_j${j}_div$div.json
  done
done
