#!/bin/bash
# GPU box: SQ counters of k_exact_wide (separate --pmc passes, kernel trace only).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY" "SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf $O/prof_w
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/prof_w -o p -- python $R/tools/one_wide.py ${1:-f16} ${2:-32} > /dev/null 2> $O/wide_pmc.err || tail -3 $O/wide_pmc.err
  db=$(ls $O/prof_w/*.db $O/prof_w/*/*.db 2>/dev/null | head -1)
  python $R/profiles/summarize_rocpd.py "$db" $O/wide_pmc_$tag.md > /dev/null 2>&1
  grep -E "k_exact_wide|counter|^\| kernel" $O/wide_pmc_$tag.md | head -12 | cut -c1-220
  rm -rf $O/prof_w
done
