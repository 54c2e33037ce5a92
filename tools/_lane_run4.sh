mkdir -p gpurun_out/lane4
timeout 1200 python -m pytest tests/test_constraint_model.py tests/test_user_frame_constraints.py tests/test_gpu_parity.py -m gpu -q -k "tree_arm or arm7 or pendulum or point_mass or two_masses or fix or locks or rolling or tethered or rod or self_test" 2>&1 | tail -8 | tee gpurun_out/lane4/parity.txt
for args in "--robot arm7 --model constraint --solver euler_explicit" "--robot arm7 --model constraint" "--robot tree_arm --model constraint --solver euler_explicit" "--robot arm7" "--robot arm7 --extra"; do
  timeout 300 python tools/bench_lane.py $args 2>&1 | tail -1 | tee -a gpurun_out/lane4/bench.txt
done
