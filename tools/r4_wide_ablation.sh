#!/bin/bash
# Round-4 experiments on the 256-query int8 pass (profiles/r04_wide_ablation.md): the probe binaries (tools/probe/wide_probe_*,
# built here with tools/probe/build_wide_probes.sh flags) interleaved on ONE box, three rounds each, 10M x 768, 1,600 candidates per query.
mkdir -p gpurun_out/r4_wide
out=gpurun_out/r4_wide/probe.txt
: > $out
tools/probe/dma_offset_test | tee -a $out
for round in 1 2 3; do
  for v in ${VARIANTS:-base b c bc ahalf base}; do
    echo "== round $round variant $v" >> $out
    timeout 120 tools/probe/wide_probe_$v 10000000 1600 40 2>&1 | grep -E "threshold|k_scan_wide" >> $out
  done
done
cat $out
