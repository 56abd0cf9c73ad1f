#!/bin/bash
# GPU box: kernel timeline of similar_to calls at 690k x 768 (int8 and f32) + the bench numbers.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dt in i8 f32; do
  rm -rf $O/prof_sim_$dt
  timeout 300 rocprofv3 --kernel-trace -d $O/prof_sim_$dt -o p -- python $R/tools/one_similar.py $dt > /dev/null 2> $O/sim_$dt.err
  db=$(ls $O/prof_sim_$dt/*.db $O/prof_sim_$dt/*/*.db 2>/dev/null | head -1)
  python $R/tools/timeline_rocpd.py "$db" 14 $O/similar_timeline_$dt.md | cut -c1-150
  rm -rf $O/prof_sim_$dt
done
cd $R && timeout 300 python tools/similar_bench.py > $O/similar_bench.json; cat $O/similar_bench.json | head -40
