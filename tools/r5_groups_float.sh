#!/bin/bash
# GPU box: per-item search over float rows: kernel stats of 32-query AVG calls (f16).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dt in ${1:-f16}; do
rm -rf $O/prof_gf
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_gf -o p -- python $R/tools/one_avg_float.py $dt > /dev/null 2> $O/gf.err
db=$(ls $O/prof_gf/*.db $O/prof_gf/*/*.db 2>/dev/null | head -1)
python $R/profiles/summarize_rocpd.py "$db" $O/groups_float_${dt}_kernel_stats.md > /dev/null
python $R/tools/timeline_rocpd.py "$db" 40 $O/groups_float_${dt}_timeline.md | cut -c1-150
rm -rf $O/prof_gf
head -30 $O/groups_float_${dt}_kernel_stats.md | cut -c1-200
done
