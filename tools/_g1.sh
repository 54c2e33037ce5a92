mkdir -p gpurun_out/r01p
timeout 900 python -m pytest tests/test_gpu_env.py -m gpu -x -q > gpurun_out/r01p/pytest_env.log 2>&1; tail -15 gpurun_out/r01p/pytest_env.log
timeout 300 python tools/bench_env.py > gpurun_out/r01p/env_hip.json 2> gpurun_out/r01p/env_hip.err; cat gpurun_out/r01p/env_hip.json
JIMINY_AMD_TENSOR_BLOCKS=1 timeout 300 python tools/bench_env.py > gpurun_out/r01p/env_tensor.json 2> gpurun_out/r01p/env_tensor.err; cat gpurun_out/r01p/env_tensor.json
