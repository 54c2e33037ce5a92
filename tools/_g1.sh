mkdir -p gpurun_out/r01j
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r01j/pytest_gpu.log 2>&1; tail -5 gpurun_out/r01j/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --steps 100 > gpurun_out/r01j/bench_anymal.json 2>gpurun_out/r01j/err.log; python -c "import json; d=json.loads(open('gpurun_out/r01j/bench_anymal.json').read()); print('anymal', d['value'], d['roofline']['avg_launch_ms'])"
timeout 300 python bench.py --no-cpu-baseline --model atlas --batch 32768 --steps 60 --warmup 25 --dt 2.5e-4 > gpurun_out/r01j/bench_atlas.json 2>>gpurun_out/r01j/err.log; python -c "import json; d=json.loads(open('gpurun_out/r01j/bench_atlas.json').read()); print('atlas', d['value'], d['roofline']['avg_launch_ms'])"
