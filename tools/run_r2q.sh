set -x
O=gpurun_out/r2q; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -5 $O/pytest.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/p -o p -- python $R/bench.py --steps 20 --warmup 5 --no-verify --no-cpu-baseline --no-peaks > $R/$O/headline_prof.json 2> $R/$O/headline_prof.err
cd $R
db=$(find $O/p -name "*.db" | head -1); python profiles/summarize_rocpd.py "$db" $O/headline_kernels.md > /dev/null 2>&1; rm -rf $O/p
head -20 $O/headline_kernels.md
