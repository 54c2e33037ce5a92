O=$(pwd)/gpurun_out/r01i; mkdir -p $O; export TMPDIR=/tmp; R=$(pwd)
python bench.py --no-cpu-baseline --steps 100 > $O/bench_anymal.json 2>$O/err.log; python -c "import json; d=json.loads(open('$O/bench_anymal.json').read()); print('anymal', d['value'], d['roofline']['avg_launch_ms'])"
python bench.py --no-cpu-baseline --model atlas --batch 32768 --steps 60 --warmup 25 --dt 2.5e-4 > $O/bench_atlas.json 2>>$O/err.log; python -c "import json; d=json.loads(open('$O/bench_atlas.json').read()); print('atlas', d['value'], d['roofline']['avg_launch_ms'])"
cd /tmp
for M in anymal atlas; do
  EXTRA=""; [ $M = atlas ] && EXTRA="--model atlas --batch 32768 --dt 2.5e-4"
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_$M -o run -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --episode 0 $EXTRA > $O/pmc_$M.log 2>&1
  python - <<PY
import sqlite3,glob,statistics
db=sqlite3.connect(glob.glob('$O/pmc_$M/*.db')[0])
rows=db.execute("select counter_name, value, duration from counters_collection where kernel_name like '%k_quad%'").fetchall()
med=statistics.median(r[2] for r in rows)
out={}
for n,v,d in rows:
    if d>0.6*med: out.setdefault(n,[]).append(v)
print('$M', {k:statistics.median(v) for k,v in out.items()}, 'dur_ns', med)
PY
done
find $O -name '*.db' -delete
