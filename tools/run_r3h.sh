set -x
O=gpurun_out/r3h; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
timeout 900 python tools/loader_bench.py --rows 300000 > $O/loader_bench.json 2> $O/loader_bench.err
cat $O/loader_bench.json
