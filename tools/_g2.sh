mkdir -p gpurun_out/r01k
for t in "" w2; do
  JIMINY_AMD_LIB_TAG=$t timeout 300 python bench.py --no-cpu-baseline --steps 100 > gpurun_out/r01k/bench_$t.json 2>gpurun_out/r01k/bench_$t.err
  echo "tag=[$t] $(python -c "import json,sys; d=json.loads(open('gpurun_out/r01k/bench_$t.json').read()); print(d['value'], d['roofline']['avg_launch_ms'])")"
done
