#!/bin/bash
# A/B timing of tagged experimental builds (tools/ab_build.py) against the default build, same box, interleaved repeats.
# Usage (through gpurun):  bash tools/measure_ab.sh <outtag> "<tag1> <tag2> ..." [model] [reps]     ("none" = the default build)
set -u
exec < /dev/null
OUTTAG=${1:-ab}; TAGS=${2:-none}; MODEL=${3:-anymal}; REPS=${4:-2}; WHAT=${5:-bc}
REPO=$(pwd); export PYTHONPATH=$REPO TMPDIR=/tmp
OUT=$REPO/gpurun_out/$OUTTAG; mkdir -p $OUT
export JIMINY_AMD_SELF_TEST=${JIMINY_AMD_SELF_TEST:-0}
EXTRA=""
[ "$MODEL" = atlas ] && EXTRA="--model atlas --batch 32768 --dt 2.5e-4"
for rep in $(seq 1 $REPS); do
for tag in $TAGS; do
  if [ $tag = none ]; then unset JIMINY_AMD_LIB_TAG; else export JIMINY_AMD_LIB_TAG=$tag; fi
  if [[ $WHAT == *b* ]]; then
    timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 100 --warmup 10 $EXTRA > $OUT/b_${tag}_$rep.json 2>$OUT/b_${tag}_$rep.err
  fi
  if [[ $WHAT == *c* ]]; then
    CE="--solver euler_explicit"; [ "$MODEL" = atlas ] && CE="--model atlas --batch 32768 --dt 5e-4 --solver euler_explicit"
    timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 40 --warmup 5 --contact-model constraint $CE > $OUT/c_${tag}_$rep.json 2>$OUT/c_${tag}_$rep.err
  fi
  python - <<PY
import json
for k in "$WHAT":
    try:
        b=json.loads(open('$OUT/%s_${tag}_$rep.json'%k).read().strip().splitlines()[-1])
        print('$tag', $rep, k, 'value %.4g'%b['value'], 'ms_per_step %.4f'%b['ms_per_step'], 'launch %.4f'%b['roofline']['avg_launch_ms'], 'ok %.4f'%b['config'].get('lanes_ok_min', -1))
    except Exception as e: print('$tag', k, 'ERR', e)
PY
done
done
