mkdir -p gpurun_out/r01f
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r01f/bench_anymal.json 2>gpurun_out/r01f/bench_anymal.err; tail -c 400 gpurun_out/r01f/bench_anymal.json
timeout 600 python -m pytest tests -m gpu -x -q -k "anymal or quad or dynamics" > gpurun_out/r01f/pytest_gpu.log 2>&1; tail -3 gpurun_out/r01f/pytest_gpu.log
