#!/bin/bash
# A/B on one box: profiling spans as hipEventRecord markers around every kernel (round-3 form), as events bound to the dispatches
# (hipExtLaunchKernelGGL), and no events at all; alternated so that thermal drift hits all three alike.
R=${GRAFT_REPO_ROOT:-/root/repo}
python $R/bench.py --steps 400 --warmup 5 --no-verify --no-cpu-baseline --no-peaks --no-secondary > /dev/null 2>&1  # warm the box
for i in 1 2 3 4 5 6; do for f in "--debug marker_events=1" "" "--no-kernel-events"; do python $R/bench.py --steps 200 --warmup 5 --no-verify --no-cpu-baseline --no-peaks --no-secondary $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.ljust(26), d['value'], d['ms_per_step'], r.get('avg_launch_ms'), r.get('sample_pass_avg_ms'), r.get('finalize_avg_ms'))"; done; done
