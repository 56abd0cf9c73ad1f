#!/bin/bash
# GPU box: rebuilds pvs_scan_i8_wide.hip with extra -D switches per variant and times the 256-query scan with bench.py.
# Usage: tools/sweep_wide.sh <out-tag> "<bench args>" name1:"-DFLAG ..." name2:"..."   (name "base" = no flags)
set -u
tag=$1; bargs=$2; shift 2
O=gpurun_out/$tag; mkdir -p $O
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  touch panoptikon_amd/csrc/pvs_scan_i8_wide.hip
  PVS_FLAGS_pvs_scan_i8_wide="-DPVS_WIDE_ONLY_KS3 $flags" python -m panoptikon_amd.build > $O/build_$name.log 2>&1 || { echo "build $name failed"; grep -m3 error $O/build_$name.log; continue; }
  timeout 300 python bench.py $bargs --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/$name.json 2> $O/$name.err
  python - $O/$name.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[1], d["value"], "q/s  step", d["ms_per_step"], "ms  scan", r["avg_launch_ms"], "ms  frac", r["frac"], "parity", (d.get("parity") or {}).get("ids_and_distances_bit_exact"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
