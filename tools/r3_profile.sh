#!/bin/bash
# GPU box: the round's profile set — rocprofv3 kernel stats + FETCH/WRITE/SQ PMC passes of the headline command and of the 256-query
# pass, plus one bench line each for configs[0], [1], [3], the single-query f16 shape and the f32 / L2 variants.  -> gpurun_out/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
bash $R/tools/profile_round.sh r03_b128 --steps 20 --warmup 5 --no-peaks
bash $R/tools/profile_round.sh r03_b256 --batch 256 --steps 20 --warmup 5 --no-peaks
cd $R
for spec in "cfg0:--config 0" "cfg1:--config 1" "cfg3_1gpu:--config 3" "f16_b1:--dtype f16 --batch 1" "f32_b128:--dtype f32" "l2_b128:--metric l2" "b256:--batch 256" "headline:"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 900 python bench.py $args --steps 20 --warmup 5 > $O/bench_r03_$name.json 2> $O/bench_r03_$name.err || tail -2 $O/bench_r03_$name.err
  python - $O/bench_r03_$name.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], "q/s step", d["ms_per_step"], "scan", r["avg_launch_ms"], "frac", r["frac"], "parity", (d.get("parity") or {}).get("ids_and_distances_bit_exact"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
