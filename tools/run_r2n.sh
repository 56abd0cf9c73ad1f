set -x
O=gpurun_out/r2n; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -5 $O/pytest.log
timeout 600 python bench.py --config 4 --rows 3000000 --steps 5 --warmup 1 > $O/cfg4_3M.json 2> $O/cfg4_3M.err; tail -3 $O/cfg4_3M.err
timeout 900 python bench.py --config 4 --steps 10 --warmup 2 > $O/cfg4_full.json 2> $O/cfg4_full.err; tail -3 $O/cfg4_full.err
PVS_RRF_FULL=1 timeout 900 python bench.py --config 4 --steps 5 --warmup 1 --no-verify > $O/cfg4_full_oldpath.json 2> $O/cfg4_full_oldpath.err
timeout 600 python bench.py --config 4 --rows 3000000 --gpus 2 --allow-host-gather --steps 5 --warmup 1 > $O/cfg4_3M_n2.json 2> $O/cfg4_3M_n2.err; tail -3 $O/cfg4_3M_n2.err
ls $O
