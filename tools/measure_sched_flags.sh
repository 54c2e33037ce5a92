#!/bin/bash
# tagged ANYmal libraries first:  JIMINY_AMD_LIB_TAG=<ilp|nocl|bias|td> python -c "from jiminy_amd import codegen, load_builtin; codegen.build_library(load_builtin('anymal'), force=True, extra_flags=['-mllvm', '<switch>'])"   (switches: DESIGN.md section 4.1)
set -u
exec < /dev/null
REPO=$(pwd); export PYTHONPATH=$REPO TMPDIR=/tmp
OUT=$REPO/gpurun_out/r4l; rm -rf $OUT; mkdir -p $OUT
export JIMINY_AMD_SELF_TEST=0
for rep in 1 2; do
for tag in none ilp nocl bias td; do
  if [ $tag = none ]; then unset JIMINY_AMD_LIB_TAG; else export JIMINY_AMD_LIB_TAG=$tag; fi
  timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 50 --warmup 10 > $OUT/b_${tag}_$rep.json 2>$OUT/b_${tag}_$rep.err
  timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 30 --warmup 5 --contact-model constraint --solver euler_explicit > $OUT/c_${tag}_$rep.json 2>$OUT/c_${tag}_$rep.err
  python - <<PY
import json
for k in ("b","c"):
    try:
        b=json.loads(open('$OUT/%s_${tag}_$rep.json'%k).read().strip().splitlines()[-1])
        print('$tag', $rep, k, '%.4g'%b['value'], 'launch %.4f'%b['roofline']['avg_launch_ms'])
    except Exception as e: print('$tag', k, 'ERR', e)
PY
done
done
