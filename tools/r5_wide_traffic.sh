#!/bin/bash
# GPU box: HBM traffic of k_exact_wide (FETCH_SIZE / WRITE_SIZE in separate passes, kernel trace only) at 1M x 768 rows x 32 queries.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dt in f16 f32; do
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/prof_w
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/prof_w -o p -- python $R/tools/one_wide.py $dt 32 > /dev/null 2> $O/wide_pmc.err || tail -3 $O/wide_pmc.err
  db=$(ls $O/prof_w/*.db $O/prof_w/*/*.db 2>/dev/null | head -1)
  python $R/profiles/summarize_rocpd.py "$db" $O/wide_${dt}_pmc_$c.md > /dev/null 2>&1
  grep -E "k_exact_wide.*$c" $O/wide_${dt}_pmc_$c.md | head -2 | cut -c1-200
  rm -rf $O/prof_w
done
done
