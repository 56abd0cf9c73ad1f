# second sweep: 8 waves x 1 group (2 waves/SIMD) with the v_max3 pre-test, and the on-box peaks
set -x
O=gpurun_out/r2d; mkdir -p $O
python - > $O/peaks.json 2> $O/peaks.err <<'PY'
import json, panoptikon_amd as pvs
print(json.dumps(pvs.microbench(0)))
PY
cat $O/peaks.json
variant() { # name flags...
  name=$1; shift
  touch panoptikon_amd/csrc/pvs_scan_i8.hip
  PVS_FLAGS_pvs_scan_i8="-DPVS_ONLY_KS3 $*" python -m panoptikon_amd.build > $O/build_$name.log 2>&1 || { echo "build $name failed"; tail -5 $O/build_$name.log; return; }
  for b in 256; do timeout 200 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-verify --no-peaks > $O/${name}_b$b.json 2> $O/${name}_b$b.err; done
}
variant wide8 -DPVS_WIDE8
timeout 200 python bench.py --batch 256 --steps 5 --warmup 2 --no-cpu-baseline --no-peaks > $O/wide8_verify.json 2> $O/wide8_verify.err
variant wide8_noepi -DPVS_WIDE8 -DPVS_ABL_NOEPI
variant wide8_noepi_nodma -DPVS_WIDE8 -DPVS_ABL_NOEPI -DPVS_ABL_NODMA
variant wide8_vform -DPVS_WIDE8 -mllvm -amdgpu-mfma-vgpr-form=1
ls $O
