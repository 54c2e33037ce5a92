mkdir -p gpurun_out/r01h
for t in "" ilp clause noref; do
  JIMINY_AMD_LIB_TAG=$t timeout 300 python bench.py --no-cpu-baseline --steps 100 > gpurun_out/r01h/bench_$t.json 2>gpurun_out/r01h/bench_$t.err
  echo "tag=[$t] $(python -c "import json,sys; d=json.loads(open('gpurun_out/r01h/bench_$t.json').read()); print(d['value'], d['roofline']['avg_launch_ms'])")"
done
