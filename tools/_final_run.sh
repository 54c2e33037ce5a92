mkdir -p gpurun_out/fin1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/fin1/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/fin1/smoke.txt
timeout 900 python bench.py 2>gpurun_out/fin1/bench.err | tail -1 > gpurun_out/fin1/bench.json; cut -c1-600 gpurun_out/fin1/bench.json
for args in "--robot arm7" "--robot arm7 --extra" "--robot arm7 --solver euler_explicit" "--robot arm7 --dtype float32" "--robot arm7 --model constraint --solver euler_explicit" "--robot arm7 --model constraint" "--robot tree_arm" "--robot tree_arm --model constraint --solver euler_explicit" "--robot double_pendulum" "--robot pendulum"; do
  timeout 300 python tools/bench_lane.py $args 2>&1 | tail -1 | tee -a gpurun_out/fin1/lane_bench.jsonl | cut -c1-200
done
JM_PROFILE_DOMINANT='%k_batch%' bash tools/gpu_profile_lane.sh r05_lane_arm7 --robot arm7 --extra 2>&1 | tail -50
