set -x
O=gpurun_out/r2t; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
for b in 256 128; do timeout 200 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/base_b$b.json 2> $O/base_b$b.err; done
for c in "--config 1" "--batch 1 --dtype f16" "--batch 128 --dtype f32" "--batch 128 --metric l2" "--batch 32" "--batch 64"; do n=$(echo $c | tr -d ' -'); timeout 200 python bench.py $c --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/x_$n.json 2> $O/x_$n.err; done
touch panoptikon_amd/csrc/pvs_scan_i8.hip
PVS_FLAGS_pvs_scan_i8="-DPVS_ABL_NOEMIT" python -m panoptikon_amd.build > $O/build_noemit.log 2>&1
for b in 256 128; do timeout 200 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/noemit_b$b.json 2> $O/noemit_b$b.err; done
ls $O
