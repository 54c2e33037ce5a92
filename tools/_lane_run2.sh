mkdir -p gpurun_out/lane2
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/lane2/gpu_suite.txt
for args in "--robot arm7" "--robot arm7 --extra" "--robot arm7 --solver euler_explicit" "--robot arm7 --model constraint --solver euler_explicit" "--robot arm7 --dtype float32" "--robot tree_arm" "--robot tree_arm --model constraint --solver euler_explicit"; do
  timeout 300 python tools/bench_lane.py $args 2>&1 | tail -1 | tee -a gpurun_out/lane2/bench.jsonl
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/lane2/prof -o arm7 -- python $GRAFT_REPO_ROOT/tools/bench_lane.py --robot arm7 --extra > $GRAFT_REPO_ROOT/gpurun_out/lane2/prof.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/lane2/prof | head -20
