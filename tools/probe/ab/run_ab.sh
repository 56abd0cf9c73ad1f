#!/bin/bash
# same-box A/B of libpvs builds: tools/probe/ab/libpvs_<X>.so copied over the in-tree library, the float per-item bench on each, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cp panoptikon_amd/libpvs.so /tmp/libpvs_orig.so
for round in 1 2 3; do for v in "$@"; do
  cp tools/probe/ab/libpvs_$v.so panoptikon_amd/libpvs.so
  echo -n "$v: "; timeout 300 python tools/float_certify_bench.py f16 4000000 /tmp/x.json --quick 2>&1 | tail -1 | cut -c80-190
done; done
cp /tmp/libpvs_orig.so panoptikon_amd/libpvs.so
