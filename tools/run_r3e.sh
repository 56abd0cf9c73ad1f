set -x
O=gpurun_out/r3e; mkdir -p $O
for d in 12 16 24 32 48; do PVS_SAMPLE_DIV=$d timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/div${d}_b128.json 2> $O/div${d}_b128.err; done
for d in 16 32; do PVS_SAMPLE_DIV=$d timeout 200 python bench.py --batch 256 --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/div${d}_b256.json 2> $O/div${d}_b256.err; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/$O/prof4
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof4 -o p -- python $R/bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline --no-peaks --no-verify > $R/$O/cfg4.json 2> $R/$O/cfg4.err
db=$(ls $R/$O/prof4/*.db $R/$O/prof4/*/*.db 2>/dev/null | head -1)
python $R/profiles/summarize_rocpd.py "$db" $R/$O/cfg4_kernel_stats.md > /dev/null
rm -rf $R/$O/prof4
cd $R
timeout 600 python bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline --no-peaks > $O/cfg4_plain.json 2> $O/cfg4_plain.err
ls $O
