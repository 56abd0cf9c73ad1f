#!/bin/bash
# GPU box: kernel timeline of single-query per-item AVG calls at 690k x 768 int8 + latency numbers.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_avg
timeout 300 rocprofv3 --kernel-trace -d $O/prof_avg -o p -- python $R/tools/one_avg.py > /dev/null 2> $O/avg.err
db=$(ls $O/prof_avg/*.db $O/prof_avg/*/*.db 2>/dev/null | head -1)
python $R/tools/timeline_rocpd.py "$db" 16 $O/one_avg_timeline.md | cut -c1-150
rm -rf $O/prof_avg
cd $R && timeout 300 python tools/latency_items.py 2>&1 | tail -12
