set -x
O=gpurun_out/r2y; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
for k in 100 1; do timeout 200 python bench.py --batch 256 --k $k --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/el_k${k}_b256.json 2> $O/el_k${k}_b256.err; done
timeout 300 python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --no-peaks > $O/el_cfg3.json 2> $O/el_cfg3.err
touch panoptikon_amd/csrc/pvs_scan_i8.hip
PVS_FLAGS_pvs_scan_i8="-DPVS_NO_ELASTIC" python -m panoptikon_amd.build > $O/build_noel.log 2>&1
for k in 100 1; do timeout 200 python bench.py --batch 256 --k $k --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/noel_k${k}_b256.json 2> $O/noel_k${k}_b256.err; done
ls $O
