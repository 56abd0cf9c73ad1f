#!/bin/bash
# GPU box: rocprofv3 kernel stats of an arbitrary python command.  Usage: tools/r4_prof.sh <tag> <script and args...>
set -u
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; out=$R/gpurun_out/r4; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats -d $out/prof_$tag -o p -- python "$@" > $out/${tag}.out 2> $out/${tag}.err
db=$(ls $out/prof_$tag/*.db $out/prof_$tag/*/*.db 2>/dev/null | head -1)
python $R/profiles/summarize_rocpd.py "$db" $out/${tag}_kernel_stats.md > /dev/null
rm -rf $out/prof_$tag
head -45 $out/${tag}_kernel_stats.md | cut -c1-220
