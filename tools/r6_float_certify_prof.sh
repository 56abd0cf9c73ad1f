#!/bin/bash
# GPU box, round 6: kernel stats + timeline of 32-query per-item AVG calls over 4M x 768 float rows through the certified route.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dt in ${1:-f16}; do
rm -rf $O/prof_fc
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_fc -o p -- python $R/tools/one_avg_float.py $dt > /dev/null 2> $O/fc.err
db=$(ls $O/prof_fc/*.db $O/prof_fc/*/*.db 2>/dev/null | head -1)
python $R/profiles/summarize_rocpd.py "$db" $O/r06_items_certified_${dt}_4Mx768_b32_kernel_stats.md > /dev/null
python $R/tools/timeline_rocpd.py "$db" 40 $O/r06_items_certified_${dt}_4Mx768_b32_timeline.md | cut -c1-150
rm -rf $O/prof_fc
head -34 $O/r06_items_certified_${dt}_4Mx768_b32_kernel_stats.md | cut -c1-200
done
