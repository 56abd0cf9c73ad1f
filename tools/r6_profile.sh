#!/bin/bash
# GPU box, round 6: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes (separate runs, kernel-trace only) for
#  (1) the headline command (k_scan_wide, 128 queries), (2) the 256-query pass, (3) the single-query f16 north-star shape,
#  (4) the one-launch search at the reference's scale, (5) the certified per-item search over 4M x 768 f16 rows (k_scan MODE 5).
# Usage: tools/r6_profile.sh   -> gpurun_out/r06_*
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() { # tag, kernel regex, command...
  local tag=$1 kre=$2; shift 2
  for pass in kernel_stats pmc_FETCH_SIZE pmc_WRITE_SIZE; do
    local flags="--kernel-trace --stats"
    [ $pass = pmc_FETCH_SIZE ] && flags="--kernel-trace --pmc FETCH_SIZE"
    [ $pass = pmc_WRITE_SIZE ] && flags="--kernel-trace --pmc WRITE_SIZE"
    rm -rf $O/prof_${tag}_$pass
    timeout 600 rocprofv3 $flags -d $O/prof_${tag}_$pass -o p -- "$@" > $O/${tag}_$pass.bench.json 2> $O/${tag}_$pass.err
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
  grep -E "k_direct|k_scan|k_prep|k_final|k_kth|k_cand|k_spill" $O/${tag}_kernel_stats.md | head -6 | cut -c1-200
}
B="python $R/bench.py --no-verify --no-cpu-baseline --no-peaks --no-secondary"
prof r06_10Mx768_i8_b128 '_Z\d+k_scan\w*?ELi1EEv5ScanK\S*' $B --steps 20 --warmup 5
prof r06_10Mx768_i8_b256 '_Z\d+k_scan\w*?ELi1EEv5ScanK\S*' $B --batch 256 --steps 20 --warmup 5
prof r06_10Mx768_f16_b1 '_Z\d+k_scan\w*?ELi1EEv5ScanK\S*' $B --dtype f16 --batch 1 --steps 30 --warmup 3
prof r06_direct_690kx768_i8_b1 '_ZN10pvs_direct13k_direct_topk\S*' $B --rows 690000 --batch 1 --k 10 --steps 200 --warmup 10
prof r06_items_certified_4Mx768_f16_b32 '_Z\d+k_scan\w*?ELi5EEv5ScanK\S*' python $R/tools/one_avg_float.py f16 32 --json
