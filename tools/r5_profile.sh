#!/bin/bash
# GPU box, round 5: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes (separate runs, kernel-trace only) for
#  (1) the headline command (k_scan_wide, 128 queries), (2) the one-launch search at the reference's scale, one query and four,
#  (3) the single-query f16 north-star shape.  Usage: tools/r5_profile.sh   -> gpurun_out/r05_*
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() { # tag, kernel regex, bench args...
  local tag=$1 kre=$2; shift 2
  for pass in kernel_stats pmc_FETCH_SIZE pmc_WRITE_SIZE; do
    local flags="--kernel-trace --stats"
    [ $pass = pmc_FETCH_SIZE ] && flags="--kernel-trace --pmc FETCH_SIZE"
    [ $pass = pmc_WRITE_SIZE ] && flags="--kernel-trace --pmc WRITE_SIZE"
    rm -rf $O/prof_${tag}_$pass
    timeout 600 rocprofv3 $flags -d $O/prof_${tag}_$pass -o p -- python $R/bench.py "$@" --no-verify --no-cpu-baseline --no-peaks --no-secondary > $O/${tag}_$pass.bench.json 2> $O/${tag}_$pass.err
    local db=$(ls $O/prof_${tag}_$pass/*.db $O/prof_${tag}_$pass/*/*.db 2>/dev/null | head -1)
    python $R/profiles/summarize_rocpd.py "$db" $O/${tag}_$pass.md > /dev/null
    rm -rf $O/prof_${tag}_$pass
  done
  python $R/tools/make_traffic.py --kernel "$kre" $O/${tag}_pmc_FETCH_SIZE.md $O/${tag}_pmc_WRITE_SIZE.md $O/${tag}_kernel_stats.bench.json > $O/${tag}_traffic.json
  python - <<PY
import json
t = json.load(open("$O/${tag}_traffic.json"))
print("$tag", t["kernel"][:60], "hbm bytes/launch", t["hbm_bytes_per_launch"], "algorithmic", t["algorithmic_bytes_per_launch"], "ratio", round(t["hbm_bytes_per_launch"] / t["algorithmic_bytes_per_launch"], 3))
PY
  grep -E "k_direct|k_scan|k_prep|k_final|k_kth" $O/${tag}_kernel_stats.md | head -6 | cut -c1-200
}
prof r05_10Mx768_i8_b128 '_Z\d+k_scan\w*?ELi1EEv5ScanK\S*' --steps 20 --warmup 5
prof r05_direct_690kx768_i8_b1 '_ZN10pvs_direct13k_direct_topk\S*' --rows 690000 --batch 1 --k 10 --steps 200 --warmup 10
prof r05_direct_690kx768_i8_b4 '_ZN10pvs_direct13k_direct_topk\S*' --rows 690000 --batch 4 --k 10 --steps 200 --warmup 10
prof r05_10Mx768_f16_b1 '_Z\d+k_scan\w*?ELi1EEv5ScanK\S*' --dtype f16 --batch 1 --steps 30 --warmup 3
