#!/bin/bash
# GPU box, end of round 6: smoke, the whole -m gpu suite, the default bench line, the float per-item bench (f16, f32), the profile set.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6z; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
(timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo bench rc=$?
for dt in f16 f32; do timeout 600 python tools/float_certify_bench.py $dt 4000000 $O/r06_float_certify_${dt}_4Mx768.json > $O/fc_$dt.log 2>&1; tail -1 $O/fc_$dt.log | cut -c1-400; done
timeout 1500 bash tools/r6_profile.sh > $O/profile.log 2>&1; tail -12 $O/profile.log | cut -c1-200
