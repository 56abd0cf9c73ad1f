set -x
O=gpurun_out/r2w; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
for k in 1 10 100 400 1000; do timeout 200 python bench.py --batch 256 --k $k --steps 20 --warmup 5 --no-cpu-baseline --no-peaks --no-verify > $O/k${k}_b256.json 2> $O/k${k}_b256.err; done
for k in 1 10 100 1000; do timeout 200 python bench.py --batch 128 --k $k --steps 20 --warmup 5 --no-cpu-baseline --no-peaks --no-verify > $O/k${k}_b128.json 2> $O/k${k}_b128.err; done
ls $O
