set -x
O=gpurun_out/r2j; mkdir -p $O
variant() { # name flags...
  name=$1; shift
  touch panoptikon_amd/csrc/pvs_scan_i8.hip
  PVS_FLAGS_pvs_scan_i8="-DPVS_ONLY_KS3 $*" python -m panoptikon_amd.build > $O/build_$name.log 2>&1 || { echo "build $name failed"; tail -5 $O/build_$name.log; return; }
  for b in 256 128; do timeout 200 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/${name}_b$b.json 2> $O/${name}_b$b.err; done
}
(timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -3 $O/pytest.log
variant base
variant w8 -DPVS_WIDE8
variant w8_noemit -DPVS_WIDE8 -DPVS_ABL_NOEMIT
for c in "--config 1" "--batch 1 --dtype f16" "--batch 128 --dtype f32" "--batch 128 --metric l2"; do n=$(echo $c | tr -d ' -'); timeout 200 python bench.py $c --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/x_$n.json 2> $O/x_$n.err; done
ls $O
