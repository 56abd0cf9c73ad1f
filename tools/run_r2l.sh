set -x
O=gpurun_out/r2l; mkdir -p $O
R=$GRAFT_REPO_ROOT
prof() { # name flags...
  name=$1; shift
  cd $R
  touch panoptikon_amd/csrc/pvs_scan_i8.hip
  PVS_FLAGS_pvs_scan_i8="-DPVS_ONLY_KS3 $*" python -m panoptikon_amd.build > $O/build_$name.log 2>&1 || { echo "build $name failed"; return; }
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM -d $R/$O/p_$name -o p -- python $R/bench.py --batch 256 --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-peaks > $R/$O/$name.log 2>&1
  cd $R
  db=$(find $O/p_$name -name "*.db" | head -1); python profiles/summarize_rocpd.py "$db" $O/$name.md > /dev/null 2>&1; rm -rf $O/p_$name
  grep "k_scanILi2ELi3ELi8ELi0ELi1" $O/$name.md
}
prof w8 -DPVS_WIDE8
prof w8_noemit -DPVS_WIDE8 -DPVS_ABL_NOEMIT
prof w8_foldonly -DPVS_WIDE8 -DPVS_ABL_FOLDONLY
