mkdir -p gpurun_out/lane1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "small_robots and (arm7 or pendulum or tree_arm-)" 2>&1 | tail -5 | tee gpurun_out/lane1/parity.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "small_robots and (arm7 or tree_arm) and not tree_arm_ff" 2>&1 | tail -5 | tee -a gpurun_out/lane1/parity.txt
for args in "--robot arm7" "--robot arm7 --extra" "--robot arm7 --solver euler_explicit" "--robot arm7 --dtype float32" "--robot tree_arm"; do
  timeout 300 python tools/bench_lane.py $args 2>&1 | tail -1 | tee -a gpurun_out/lane1/bench.jsonl
done
