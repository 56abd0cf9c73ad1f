#!/bin/bash
# GPU box: parity tests that exercise k_dense_exact and similar_to, then similar_bench + score_bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q --timeout 600 -k "dense or score or similar or float or bounded or packed" > $O/t_dense.out 2>&1; echo "tests rc $?"; tail -4 $O/t_dense.out | cut -c1-200
timeout 300 python tools/similar_bench.py > $O/similar_bench2.json; python - <<PY
import json
d=json.load(open("$O/similar_bench2.json"))
print({k:v["ms_per_call"] for k,v in d.items() if isinstance(v,dict)})
PY
timeout 300 python tools/score_bench.py 2>&1 | tail -12
