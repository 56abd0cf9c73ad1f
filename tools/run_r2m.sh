set -x
O=gpurun_out/r2m; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/headline.json 2> $O/headline.err
for c in "--batch 256" "--config 1" "--batch 1 --dtype f16" "--batch 128 --metric l2" "--gpus 2 --single-process --devices 0,0"; do n=$(echo $c | tr -d ' -,'); timeout 300 python bench.py $c --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/x_$n.json 2> $O/x_$n.err; done
ls $O
