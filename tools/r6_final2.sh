#!/bin/bash
# GPU box, very end of round 6: the default bench line and the items profile of the last build (the float-certify tail changed after tools/r6_final.sh ran)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6y; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo bench rc=$?
for dt in f16 f32; do timeout 600 python tools/float_certify_bench.py $dt 4000000 $O/r06_float_certify_${dt}_4Mx768.json > $O/fc_$dt.log 2>&1; tail -1 $O/fc_$dt.log | cut -c1-200; done
timeout 500 bash tools/r6_float_certify_prof.sh f16 > $O/prof.log 2>&1
(timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
