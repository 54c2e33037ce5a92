mkdir -p gpurun_out/r01c
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r01c/pytest_gpu.log 2>&1; tail -15 gpurun_out/r01c/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r01c/bench_anymal.json 2>gpurun_out/r01c/bench_anymal.err; tail -c 900 gpurun_out/r01c/bench_anymal.json
timeout 300 python bench.py --no-cpu-baseline --dtype f32 > gpurun_out/r01c/bench_anymal_f32.json 2>&1; tail -c 500 gpurun_out/r01c/bench_anymal_f32.json
timeout 300 python bench.py --no-cpu-baseline --model atlas --batch 32768 --steps 50 --warmup 5 --dt 2.5e-4 > gpurun_out/r01c/bench_atlas.json 2>&1; tail -c 900 gpurun_out/r01c/bench_atlas.json
JM_KERNEL_VARIANT=lane timeout 300 python bench.py --no-cpu-baseline --model atlas --batch 32768 --steps 20 --warmup 3 --dt 2.5e-4 > gpurun_out/r01c/bench_atlas_lane.json 2>&1; tail -c 500 gpurun_out/r01c/bench_atlas_lane.json
timeout 300 python bench.py --no-cpu-baseline --model atlas --batch 4096 --steps 50 --warmup 5 --dt 2.5e-4 > gpurun_out/r01c/bench_atlas_4096.json 2>&1; tail -c 500 gpurun_out/r01c/bench_atlas_4096.json
