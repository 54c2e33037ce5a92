mkdir -p gpurun_out/lane5
for tag in "" nolsv; do
for args in "--robot arm7 --model constraint --solver euler_explicit" "--robot tree_arm --model constraint --solver euler_explicit"; do
  JIMINY_AMD_LIB_TAG=$tag timeout 300 python tools/bench_lane.py $args 2>&1 | tail -1 | sed "s/^/[$tag] /" | tee -a gpurun_out/lane5/bench.txt
done
done
