set -x
O=gpurun_out/r2u; mkdir -p $O
variant() { # name flags...
  name=$1; shift
  touch panoptikon_amd/csrc/pvs_scan_i8.hip
  PVS_FLAGS_pvs_scan_i8="-DPVS_ONLY_KS3 $*" python -m panoptikon_amd.build > $O/build_$name.log 2>&1 || { echo "build $name failed"; tail -5 $O/build_$name.log; return; }
  for b in 256 128; do timeout 200 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-peaks --no-verify > $O/${name}_b$b.json 2> $O/${name}_b$b.err; done
}
variant base
variant nostore -DPVS_ABL_NOSTORE
variant fixedstore -DPVS_ABL_FIXEDSTORE
ls $O
