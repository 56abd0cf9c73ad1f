#!/bin/bash
# GPU box, end of round 4: kernel stats + FETCH / WRITE / SQ passes of the headline command (events bound to the dispatches), kernel
# stats of the one-launch single-query search at the reference's scale, the default bench line with its three secondary regions.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
bash $R/tools/profile_round.sh r04b_10Mx768_i8_b128 --steps 20 --warmup 5 --no-peaks --no-secondary
tag=r04b_kernel_stats_direct_690kx768_i8_b1
( cd /tmp && export TMPDIR=/tmp && rm -rf $O/prof_$tag && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o p -- python $R/bench.py --rows 690000 --batch 1 --k 10 --steps 200 --warmup 10 --no-verify --no-cpu-baseline --no-peaks --no-secondary > $O/$tag.bench.json 2> $O/$tag.err
  db=$(ls $O/prof_$tag/*.db $O/prof_$tag/*/*.db 2>/dev/null | head -1); python $R/profiles/summarize_rocpd.py "$db" $O/$tag.md > /dev/null; rm -rf $O/prof_$tag )
grep -E "k_direct|k_prep" $O/$tag.md | cut -c1-220
cd $R
timeout 900 python bench.py > $O/bench_r04b_default.json 2> $O/bench_r04b_default.err || tail -3 $O/bench_r04b_default.err
timeout 600 python bench.py --config 1 > $O/bench_r04b_cfg1.json 2>/dev/null
timeout 600 python bench.py --config 0 > $O/bench_r04b_cfg0.json 2>/dev/null
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/bench_r04b_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
        print(os.path.basename(f), d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"), r.get("kernel"))
    except Exception as e:
        print(f, "ERR", e)
PY
