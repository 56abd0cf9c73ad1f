set -x
O=gpurun_out/r3v; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
(timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo bench rc=$?
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err
