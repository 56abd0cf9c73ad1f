#!/bin/bash
# GPU box: what do the register spills of the widest-pitch scan instances cost?  The instances with 12 (int8, f16) / 24 (f32) k-slabs per
# row spill 36-380 B at 512 VGPRs (tools/check_scratch.py); their neighbours (8 / 16 k-slabs) do not.  Same rows x bytes per row, so the
# scan's fraction of the HBM roofline is comparable.  Usage: tools/r4_spill_cost.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4/spill; mkdir -p $O
cd $R
for spec in "i8 3072 1000000" "i8 2048 1500000" "f16 1536 1000000" "f16 1024 1500000" "f32 1536 500000" "f32 1024 750000"; do
  set -- $spec
  for b in 1 32 128; do
    timeout 300 python bench.py --dtype $1 --dim $2 --rows $3 --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-peaks --no-verify > $O/$1_$2_b$b.json 2> $O/$1_$2_b$b.err || { echo "$spec b$b failed"; tail -2 $O/$1_$2_b$b.err; continue; }
    python - $O/$1_$2_b$b.json <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print(r["config"]["workload"][:44].ljust(46), r["roofline"]["kernel"][:46].ljust(48), "scan", r["roofline"]["avg_launch_ms"], "ms  frac", r["roofline"]["frac"], " q/s", r["value"])
PY
  done
done
