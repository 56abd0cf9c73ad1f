set -x
O=gpurun_out/r2o; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/p4 -o p -- python $R/bench.py --config 4 --steps 5 --warmup 1 --no-verify > $R/$O/cfg4_prof.json 2> $R/$O/cfg4_prof.err
cd $R
db=$(find $O/p4 -name "*.db" | head -1); python profiles/summarize_rocpd.py "$db" $O/cfg4_kernels.md > /dev/null 2>&1; rm -rf $O/p4
head -30 $O/cfg4_kernels.md
