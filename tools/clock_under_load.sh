#!/bin/bash
# Runs on the GPU box: samples rocm-smi clocks / power while bench.py loops over the scan (what clock does the part hold under each kernel?)
# Usage: tools/clock_under_load.sh <out-dir> "<bench args>" ...
O=$1; shift; mkdir -p $O
for args in "$@"; do
  n=$(echo $args | tr -d ' -')
  timeout 300 python bench.py $args --steps 20000 --warmup 20 --no-cpu-baseline --no-peaks --no-verify > $O/load_$n.json 2> $O/load_$n.err &
  pid=$!
  sleep 14
  for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | tr '\n' ';' >> $O/smi_$n.txt; echo >> $O/smi_$n.txt; sleep 0.4; done
  wait $pid
  echo "== $args"; cat $O/smi_$n.txt | head -6
done
