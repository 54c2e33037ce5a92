mkdir -p gpurun_out/lane3
for v in 0 1; do
for args in "--robot arm7 --model constraint --solver euler_explicit" "--robot tree_arm --model constraint --solver euler_explicit" "--robot arm7"; do
  JIMINY_AMD_BUILD_VARIANT=$v timeout 300 python tools/bench_lane.py $args 2>&1 | tail -1 | sed "s/^/v$v /" | tee -a gpurun_out/lane3/bench.txt
done
done
JIMINY_AMD_BUILD_VARIANT=1 timeout 900 python -m pytest tests/test_constraint_model.py -q -k "tree_arm" 2>&1 | tail -5 | tee gpurun_out/lane3/parity_v1.txt
