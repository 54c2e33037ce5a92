mkdir -p gpurun_out/lane0
for args in "--robot arm7" "--robot arm7 --extra" "--robot arm7 --solver euler_explicit" "--robot arm7 --model constraint --solver euler_explicit" "--robot arm7 --dtype float32" "--robot tree_arm"; do
  timeout 300 python tools/bench_lane.py $args 2>&1 | tail -1 | tee -a gpurun_out/lane0/base.jsonl
done
