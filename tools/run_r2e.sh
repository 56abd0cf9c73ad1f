set -x
O=gpurun_out/r2e; mkdir -p $O
variant() { # name flags...
  name=$1; shift
  touch panoptikon_amd/csrc/pvs_scan_i8.hip
  PVS_FLAGS_pvs_scan_i8="-DPVS_ONLY_KS3 $*" python -m panoptikon_amd.build > $O/build_$name.log 2>&1 || { echo "build $name failed"; tail -5 $O/build_$name.log; return; }
  for b in 256; do timeout 200 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-verify --no-peaks > $O/${name}_b$b.json 2> $O/${name}_b$b.err; done
}
variant w8_foldonly -DPVS_WIDE8 -DPVS_ABL_FOLDONLY
variant w8_noemit -DPVS_WIDE8 -DPVS_ABL_NOEMIT
variant b_foldonly -DPVS_ABL_FOLDONLY
variant b_noemit -DPVS_ABL_NOEMIT
variant w8 -DPVS_WIDE8
PVS_SAMPLE_DIV=4 timeout 200 python bench.py --batch 256 --steps 20 --warmup 5 --no-cpu-baseline --no-verify --no-peaks > $O/w8_div4_b256.json 2> $O/w8_div4.err
PVS_SAMPLE_DIV=64 timeout 200 python bench.py --batch 256 --steps 20 --warmup 5 --no-cpu-baseline --no-verify --no-peaks > $O/w8_div64_b256.json 2> $O/w8_div64.err
ls $O
