set -x
O=gpurun_out/r2p; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q -k "rrf or leak or groups" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
timeout 900 python bench.py --config 4 --steps 10 --warmup 2 > $O/cfg4_full.json 2> $O/cfg4_full.err; tail -3 $O/cfg4_full.err
timeout 300 python tools/leak_check.py > $O/leak.log 2>&1; tail -3 $O/leak.log
