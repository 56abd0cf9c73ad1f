#!/bin/bash
# GPU box: configs[4] stage-digest race hunt.  Usage: tools/r4_digest.sh [iters]
IT=${1:-2000}
O=gpurun_out/r4_digest
mkdir -p $O
for mode in racing streams2 serial; do
  timeout 900 python tools/rrf_stage_digest.py --iters $IT --mode $mode --out $O/digest_$mode.json > $O/digest_$mode.log 2>&1
  echo "$mode rc=$?"; tail -c 600 $O/digest_$mode.json
done
timeout 600 python tools/rrf_stage_digest.py --iters 200 --mode bypass --out $O/digest_bypass.json > $O/digest_bypass.log 2>&1
echo "bypass rc=$?"; tail -c 400 $O/digest_bypass.json
