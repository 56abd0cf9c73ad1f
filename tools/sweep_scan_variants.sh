#!/bin/bash
# Runs on the GPU box (gpurun): builds pvs_scan_i8.hip with extra -D switches per variant and times the scan with bench.py.
# The switches are the tuning experiments kept in pvs_scan_kernel.hpp (PVS_TILE_PROF, PVS_ELASTIC, PVS_QG8_TWO_CHAINS, PVS_PF=n,
# PVS_PRIO_SWAP=n, PVS_ABL_NOEPI / NODMA / NOEMIT / FOLDONLY, PVS_WIDE_GPW2); per-TU flags reach the build as PVS_FLAGS_<stem>.
# Usage: tools/sweep_scan_variants.sh <out-tag> "<bench args>" name1:"-DFLAG ..." name2:"..."   (name "base" = no flags)
set -u
tag=$1; bargs=$2; shift 2
O=gpurun_out/$tag; mkdir -p $O
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  touch panoptikon_amd/csrc/pvs_scan_i8.hip
  PVS_FLAGS_pvs_scan_i8="-DPVS_ONLY_KS3 $flags" python -m panoptikon_amd.build > $O/build_$name.log 2>&1 || { echo "build $name failed"; tail -5 $O/build_$name.log; continue; }
  timeout 300 python bench.py $bargs --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/$name.json 2> $O/$name.err
  python - $O/$name.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[1], d["value"], "q/s  step", d["ms_per_step"], "ms  scan", r["avg_launch_ms"], "ms  frac", r["frac"])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
touch panoptikon_amd/csrc/pvs_scan_i8.hip; python -m panoptikon_amd.build > /dev/null 2>&1   # back to the default build
