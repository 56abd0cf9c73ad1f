#!/bin/bash
# GPU box: cost of the parts of the per-group fold in k_scan MODE 2 (4M x 768 int8, 32 queries, AVG cosine).
# PVS_FOLD_ABL = 1: closed-form distances only; 2: + half exchange + tile record; 3: + fold without the value stores; unset: everything.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O
cd $R
for abl in ${ABLS:-1 2 0}; do
  if [ $abl = 0 ]; then fl="-DPVS_ONLY_KS3"; else fl="-DPVS_ONLY_KS3 -DPVS_FOLD_ABL=$abl"; fi
  rm -f panoptikon_amd/csrc/build/pvs_scan_i8.o; PVS_FLAGS_pvs_scan_i8="$fl" python -m panoptikon_amd.build > $O/abl_build_$abl.log 2>&1 || { echo "build $abl failed"; tail -5 $O/abl_build_$abl.log; continue; }
  echo "== ablation level $abl"
  $R/tools/r4_prof.sh abl_$abl $R/tools/groups_one.py ${1:-avg} ${2:-cosine} 1 | grep "k_scan" | cut -c1-120
done
