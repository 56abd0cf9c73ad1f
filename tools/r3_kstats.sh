#!/bin/bash
# GPU box: rocprofv3 kernel stats of the headline command only.  Usage: tools/r3_kstats.sh <tag> [bench args]
set -u
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; out=$R/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats -d $out/prof_$tag -o p -- python $R/bench.py "$@" --steps 20 --warmup 5 --no-peaks --no-verify --no-cpu-baseline > $out/${tag}.bench.json 2> $out/${tag}.err
db=$(ls $out/prof_$tag/*.db $out/prof_$tag/*/*.db 2>/dev/null | head -1)
python $R/profiles/summarize_rocpd.py "$db" $out/${tag}_kernel_stats.md > /dev/null
rm -rf $out/prof_$tag
head -40 $out/${tag}_kernel_stats.md | cut -c1-200
