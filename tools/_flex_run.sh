mkdir -p gpurun_out/flex1
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/flex1/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/flex1/smoke.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/flex1/bench.err | tail -1 > gpurun_out/flex1/bench.json; cut -c1-300 gpurun_out/flex1/bench.json
