set -x
O=gpurun_out/r3b; mkdir -p $O
for k in 100 1; do timeout 200 python bench.py --batch 256 --k $k --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/split_k${k}_b256.json 2> $O/split_k${k}_b256.err; done
for k in 100; do PVS_NO_QSPLIT=1 timeout 200 python bench.py --batch 256 --k $k --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/nosplit_k${k}_b256.json 2> $O/nosplit_k${k}_b256.err; done
(timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
timeout 300 python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --no-peaks > $O/split_cfg3.json 2> $O/split_cfg3.err
ls $O
