# kernel tuning sweep for the 256-query int8 pass (run on the GPU box through gpurun)
set -x
O=gpurun_out/r2c; mkdir -p $O
variant() { # name flags...
  name=$1; shift
  touch panoptikon_amd/csrc/pvs_scan_i8.hip
  PVS_FLAGS_pvs_scan_i8="-DPVS_ONLY_KS3 $*" python -m panoptikon_amd.build > $O/build_$name.log 2>&1 || { echo "build $name failed"; tail -5 $O/build_$name.log; return; }
  for b in 256 128; do timeout 200 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-verify --no-peaks > $O/${name}_b$b.json 2> $O/${name}_b$b.err; done
}
variant base
variant pf8 -DPVS_PF=8
variant pf6 -DPVS_PF=6
variant pf8_noepi -DPVS_PF=8 -DPVS_ABL_NOEPI
variant pf8_nodma -DPVS_PF=8 -DPVS_ABL_NODMA
variant pf8_noepi_nodma -DPVS_PF=8 -DPVS_ABL_NODMA -DPVS_ABL_NOEPI
variant pf8_vform -DPVS_PF=8 -mllvm -amdgpu-mfma-vgpr-form=1
# counters on pf8
touch panoptikon_amd/csrc/pvs_scan_i8.hip
PVS_FLAGS_pvs_scan_i8="-DPVS_ONLY_KS3 -DPVS_PF=8" python -m panoptikon_amd.build > $O/build_pmc.log 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d $R/$O/pmc1 -o p -- python $R/bench.py --batch 256 --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-peaks > $R/$O/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU -d $R/$O/pmc2 -o p -- python $R/bench.py --batch 256 --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-peaks > $R/$O/pmc2.log 2>&1
cd $R
for d in pmc1 pmc2; do db=$(find $O/$d -name "*.db" | head -1); python profiles/summarize_rocpd.py "$db" $O/$d.md > /dev/null 2>&1 || true; rm -rf $O/$d; done
ls $O
