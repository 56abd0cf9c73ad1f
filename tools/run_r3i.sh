set -x
O=gpurun_out/r3i; mkdir -p $O
variant() { # name flags...
  name=$1; shift
  touch panoptikon_amd/csrc/pvs_scan_i8.hip
  PVS_FLAGS_pvs_scan_i8="-DPVS_ONLY_KS3 $*" python -m panoptikon_amd.build > $O/build_$name.log 2>&1 || { echo "build $name failed"; tail -5 $O/build_$name.log; return; }
  for k in 100 1; do timeout 200 python bench.py --batch 256 --k $k --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/${name}_k${k}_b256.json 2> $O/${name}_k${k}_b256.err; done
}
variant base
variant swap8 -DPVS_PRIO_SWAP=8
variant swap6 -DPVS_PRIO_SWAP=6
variant swap10 -DPVS_PRIO_SWAP=10
variant swap16 -DPVS_PRIO_SWAP=16
ls $O
