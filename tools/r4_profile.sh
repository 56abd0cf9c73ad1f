#!/bin/bash
# GPU box: the round-4 profile set -> gpurun_out/ (copied to profiles/ by hand).
#  * rocprofv3 kernel stats + FETCH / WRITE / SQ PMC passes (each its own run) of the headline command and of the 256-query pass;
#  * rocprofv3 kernel stats of the single-query 10M x 768 f16 run, of configs[1] and of the 32-query per-item AVG search (4M x 768);
#  * one bench line each for configs[0], [1], [3], [4], the f16 / f32 / L2 variants, and the default run with its secondary lines;
#  * the loader / backfill phases.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O $O/r4
bash $R/tools/profile_round.sh r04_10Mx768_i8_b128 --steps 20 --warmup 5 --no-peaks --no-secondary
bash $R/tools/profile_round.sh r04_10Mx768_i8_b256 --batch 256 --steps 20 --warmup 5 --no-peaks --no-secondary
kstats() { # tag, bench args
  local tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf $O/prof_$tag && timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o p -- python $R/bench.py "$@" --no-verify --no-cpu-baseline --no-peaks --no-secondary > $O/${tag}.bench.json 2> $O/${tag}.err
    db=$(ls $O/prof_$tag/*.db $O/prof_$tag/*/*.db 2>/dev/null | head -1); python $R/profiles/summarize_rocpd.py "$db" $O/${tag}.md > /dev/null; rm -rf $O/prof_$tag )
  grep -E "k_scan|k_finalize|k_kth" $O/${tag}.md | head -5 | cut -c1-200
}
kstats r04_kernel_stats_10Mx768_f16_b1 --dtype f16 --batch 1 --steps 20 --warmup 5
kstats r04_kernel_stats_cfg1_1Mx768_f16_b32 --config 1 --steps 50 --warmup 5
bash $R/tools/r4_prof.sh groups_avg_4Mx768_b32 $R/tools/groups_one.py avg cosine 1 4000000 32 20 2>&1 | tail -12
cd $R
for spec in "cfg0:--config 0" "cfg1:--config 1" "cfg3_1gpu:--config 3" "cfg4_1gpu:--config 4" "f16_b1:--dtype f16 --batch 1" "f32_b128:--dtype f32" "l2_b128:--metric l2" "b256:--batch 256 --no-secondary" "default:"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 1200 python bench.py $args > $O/bench_r04_$name.json 2> $O/bench_r04_$name.err || tail -2 $O/bench_r04_$name.err
  python - $O/bench_r04_$name.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], d["value"], d["unit"], "step", d["ms_per_step"], "kernel", r.get("avg_launch_ms"), "frac", r.get("frac"), "parity", (d.get("parity") or {}).get("ids_and_distances_bit_exact"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
timeout 900 python tools/loader_bench.py > $O/r04_loader_and_backfill_300kx768.json 2> $O/r04_loader.err; tail -c 1500 $O/r04_loader_and_backfill_300kx768.json
