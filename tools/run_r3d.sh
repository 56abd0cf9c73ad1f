set -x
O=gpurun_out/r3d; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
bash tools/profile_round.sh r3d_b128 --steps 20 --warmup 5 > $O/profile_b128.log 2>&1
bash tools/profile_round.sh r3d_b256 --batch 256 --steps 20 --warmup 5 > $O/profile_b256.log 2>&1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for c in "--batch 256" "--config 1" "--config 3" "--batch 1 --dtype f16" "--batch 128 --dtype f32" "--batch 128 --metric l2" "--batch 32" "--batch 64"; do n=$(echo $c | tr -d ' -'); timeout 300 python bench.py $c --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/x_$n.json 2> $O/x_$n.err; done
ls $O
