#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel-trace stats of the default bench.py command and the
# HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE, each in its own run, no other trace domains).
# Usage: tools/profile_round.sh <tag> [bench args...]   -> gpurun_out/<tag>_*.md, gpurun_out/<tag>_traffic.json
set -u
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args="$* --no-verify --no-cpu-baseline"
run() { # name, rocprof flags...
  local name=$1; shift
  rm -rf $out/prof_${tag}_$name
  timeout 900 rocprofv3 "$@" -d $out/prof_${tag}_$name -o p -- python $R/bench.py $args > $out/${tag}_$name.bench.json 2> $out/${tag}_$name.err
  local db=$(ls $out/prof_${tag}_$name/*.db $out/prof_${tag}_$name/*/*.db 2>/dev/null | head -1)
  python $R/profiles/summarize_rocpd.py "$db" $out/${tag}_$name.md > /dev/null
  rm -rf $out/prof_${tag}_$name
}
run kernel_stats --kernel-trace --stats
run pmc_FETCH_SIZE --kernel-trace --pmc FETCH_SIZE
run pmc_WRITE_SIZE --kernel-trace --pmc WRITE_SIZE
run pmc_SQ --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA
python $R/tools/make_traffic.py $out/${tag}_pmc_FETCH_SIZE.md $out/${tag}_pmc_WRITE_SIZE.md $out/${tag}_kernel_stats.bench.json > $out/${tag}_traffic.json
cat $out/${tag}_traffic.json
grep "k_scan" $out/${tag}_kernel_stats.md | head -6
