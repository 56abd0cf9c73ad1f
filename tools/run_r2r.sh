set -x
O=gpurun_out/r2r; mkdir -p $O
bash tools/profile_round.sh r02 --steps 20 --warmup 5 > $O/profile_round.log 2>&1
cp gpurun_out/r02_* $O/ 2>/dev/null
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_headline.json 2> $O/bench_headline.err
timeout 400 python bench.py --batch 256 --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/bench_b256.json 2> $O/bench_b256.err
timeout 400 python bench.py --config 1 --steps 20 --warmup 5 --no-peaks > $O/bench_cfg1.json 2> $O/bench_cfg1.err
timeout 400 python bench.py --config 1 --streams 2 --inflight 4 --steps 40 --warmup 5 --no-cpu-baseline --no-peaks > $O/bench_cfg1_streams2.json 2> $O/bench_cfg1_s2.err
timeout 400 python bench.py --batch 1 --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/bench_f16_b1.json 2> $O/bench_f16_b1.err
timeout 400 python bench.py --batch 128 --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline --no-peaks > $O/bench_f32_b128.json 2> $O/bench_f32_b128.err
timeout 400 python bench.py --rows 10000 --dim 512 --dtype f32 --batch 1 --k 10 --steps 50 --warmup 5 --no-peaks > $O/bench_cfg0.json 2> $O/bench_cfg0.err
timeout 900 python bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline --no-peaks > $O/bench_cfg3_1gpu.json 2> $O/bench_cfg3.err
timeout 900 python bench.py --config 4 --steps 10 --warmup 2 > $O/bench_cfg4_1gpu.json 2> $O/bench_cfg4.err
timeout 600 python bench.py --gpus 2 --single-process --devices 0,0 --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/bench_sp2_one_gpu.json 2> $O/bench_sp2.err
timeout 600 python bench.py --force-comm --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/bench_rccl_1rank.json 2> $O/bench_fc.err
ls $O
