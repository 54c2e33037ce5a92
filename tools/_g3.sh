O=$(pwd)/gpurun_out/r01n; mkdir -p $O; export TMPDIR=/tmp; R=$(pwd)
for t in "" w2 w2lean lean; do
  JIMINY_AMD_LIB_TAG=$t python bench.py --no-cpu-baseline --steps 100 > $O/bench_$t.json 2>$O/err.log; python -c "import json; d=json.loads(open('$O/bench_$t.json').read()); print('tag[$t]', d['value'], d['roofline']['avg_launch_ms'])"
done
cd /tmp
for t in w2lean; do
  JIMINY_AMD_LIB_TAG=$t rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LEVEL_WAVES -d $O/pmc_$t -o run -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --episode 0 > $O/pmc_$t.log 2>&1
  python - <<PY
import sqlite3,glob,statistics
db=sqlite3.connect(glob.glob('$O/pmc_$t/*.db')[0])
rows=db.execute("select counter_name, value, duration from counters_collection where kernel_name like '%k_quad%'").fetchall()
med=statistics.median(r[2] for r in rows)
out={}
for n,v,d in rows:
    if d>0.6*med: out.setdefault(n,[]).append(v)
print('$t', {k:statistics.median(v) for k,v in out.items()}, 'dur_ns', med)
print(db.execute("select distinct vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size from kernels where name like '%k_quad%'").fetchall())
PY
done
find $O -name '*.db' -delete
