set -x
O=gpurun_out/r2z; mkdir -p $O
for k in 100 1; do timeout 200 python bench.py --batch 256 --k $k --steps 20 --warmup 5 --no-cpu-baseline --no-peaks > $O/pc5_k${k}_b256.json 2> $O/pc5_k${k}_b256.err; done
touch panoptikon_amd/csrc/pvs_scan_i8.hip
PVS_FLAGS_pvs_scan_i8="-DPVS_ONLY_KS3 -DPVS_TILE_PROF" python -m panoptikon_amd.build > $O/build_prof.log 2>&1 || tail -20 $O/build_prof.log
for k in 100 1; do timeout 200 python bench.py --batch 256 --k $k --steps 2 --warmup 1 --no-cpu-baseline --no-peaks --no-verify > $O/prof_k${k}_b256.txt 2> $O/prof_k${k}_b256.err; done
timeout 200 python bench.py --batch 128 --k 100 --steps 2 --warmup 1 --no-cpu-baseline --no-peaks --no-verify > $O/prof_k100_b128.txt 2> $O/prof_k100_b128.err
ls $O
