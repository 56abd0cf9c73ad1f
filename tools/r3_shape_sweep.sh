#!/bin/bash
# GPU box: scan rate (roofline.frac of 8 TB/s) over dtype x dim x batch, ~4 GB corpora: finds instances that do not stream.
# Usage: tools/r3_shape_sweep.sh > gpurun_out/shape_sweep.txt
set -u
for dt in i8 f16 f32; do
  for dim in 256 384 512 768 1024 1536; do
    esz=1; [ $dt = f16 ] && esz=2; [ $dt = f32 ] && esz=4
    rows=$(( 4000000000 / (dim * esz) )); [ $rows -gt 10000000 ] && rows=10000000
    for b in 1 32 64 128 256; do
      [ $b = 256 ] && [ $dt != i8 ] && continue
      timeout 200 python bench.py --dtype $dt --dim $dim --rows $rows --batch $b --steps 15 --warmup 3 --no-cpu-baseline --no-peaks --no-verify 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$dt dim $dim rows $rows batch $b:', 'frac', r['frac'], 'scan_ms', r['avg_launch_ms'], 'step_ms', d['ms_per_step'], r['kernel'][:40])
"
    done
  done
done
