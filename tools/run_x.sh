set -x
O=gpurun_out/r4k; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -5 $O/pytest.log
for c in "--batch 128" "--batch 256" "--config 1" "--batch 1 --dtype f16" "--batch 128 --dtype f32" "--batch 32"; do n=$(echo $c | tr -d ' -'); timeout 300 python bench.py $c --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/x_$n.json 2> $O/x_$n.err; done
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-peaks --no-verify > $R/$O/prof.json 2> $R/$O/prof.err
db=$(ls $R/$O/prof/*.db $R/$O/prof/*/*.db 2>/dev/null | head -1)
python $R/profiles/summarize_rocpd.py "$db" $R/$O/kernel_stats.md > /dev/null; rm -rf $R/$O/prof
grep "ingest\|k_norm2\|k_scan_aux" $R/$O/kernel_stats.md | head
