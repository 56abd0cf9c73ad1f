set -x
mkdir -p gpurun_out/r2b
(timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log); tail -3 gpurun_out/r2b/pytest.log
for b in 128 256; do timeout 300 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2b/b$b.json 2> gpurun_out/r2b/b$b.err; done
timeout 300 python bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2b/cfg1.json 2> gpurun_out/r2b/cfg1.err
timeout 300 python bench.py --batch 1 --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2b/f16b1.json 2> gpurun_out/r2b/f16b1.err
touch panoptikon_amd/csrc/pvs_scan_i8.hip
PVS_FLAGS_pvs_scan_i8="-mllvm -amdgpu-mfma-vgpr-form=1" python -m panoptikon_amd.build > gpurun_out/r2b/build_v.log 2>&1
timeout 300 python bench.py --batch 256 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2b/b256_vform.json 2> gpurun_out/r2b/b256_vform.err
timeout 300 python bench.py --batch 128 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2b/b128_vform.json 2> gpurun_out/r2b/b128_vform.err
