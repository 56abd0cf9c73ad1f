#!/bin/bash
# GPU box: the sixth profiled region of round 6 — certified per-item search over 4M x 768 F32 rows (k_scan<f32, MODE 5>): kernel stats + FETCH / WRITE passes
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
tag=r06_items_certified_4Mx768_f32_b32
for pass in kernel_stats pmc_FETCH_SIZE pmc_WRITE_SIZE; do
  flags="--kernel-trace --stats"
  [ $pass = pmc_FETCH_SIZE ] && flags="--kernel-trace --pmc FETCH_SIZE"
  [ $pass = pmc_WRITE_SIZE ] && flags="--kernel-trace --pmc WRITE_SIZE"
  rm -rf $O/prof_${tag}_$pass
  timeout 600 rocprofv3 $flags -d $O/prof_${tag}_$pass -o p -- python $R/tools/one_avg_float.py f32 32 --json > $O/${tag}_$pass.bench.json 2> $O/${tag}_$pass.err
  db=$(ls $O/prof_${tag}_$pass/*.db $O/prof_${tag}_$pass/*/*.db 2>/dev/null | head -1)
  python $R/profiles/summarize_rocpd.py "$db" $O/${tag}_$pass.md > /dev/null
  rm -rf $O/prof_${tag}_$pass
done
python $R/tools/make_traffic.py --kernel '_Z\d+k_scan\w*?ELi5EEv5ScanK\S*' $O/${tag}_pmc_FETCH_SIZE.md $O/${tag}_pmc_WRITE_SIZE.md $O/${tag}_kernel_stats.bench.json > $O/${tag}_traffic.json
cat $O/${tag}_traffic.json | cut -c1-400
grep -E "k_scan|k_spill|k_cand" $O/${tag}_kernel_stats.md | head -4 | cut -c1-160
