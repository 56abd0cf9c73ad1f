#!/bin/bash
# GPU box 2: the stage-digest hunt as the full-size test saw the one event (k alternating 100 / 1000, other per-item operators in between)
O=gpurun_out/r4_digest2; mkdir -p $O
timeout 1200 python tools/rrf_stage_digest.py --iters 2000 --mode racing --testlike --out $O/digest_racing_testlike.json > $O/racing_testlike.log 2>&1; echo "racing+testlike rc=$?"; cut -c1-260 $O/digest_racing_testlike.json
timeout 900 python tools/rrf_stage_digest.py --iters 1000 --mode streams2 --testlike --out $O/digest_streams2_testlike.json > $O/streams2_testlike.log 2>&1; echo "streams2+testlike rc=$?"; cut -c1-260 $O/digest_streams2_testlike.json
