#!/bin/bash
# tagged ANYmal libraries first:  JIMINY_AMD_LIB_TAG=<s1|s3|s7> ... codegen.build_library(load_builtin('anymal'), force=True, extra_flags=['-DJM_QCON_SKIP=<1|3|7>'])   
set -u
exec < /dev/null
REPO=$(pwd); export PYTHONPATH=$REPO TMPDIR=/tmp
OUT=$REPO/gpurun_out/r4m; rm -rf $OUT; mkdir -p $OUT
export JIMINY_AMD_SELF_TEST=0
for tag in none s1 s3 s7; do
  if [ $tag = none ]; then unset JIMINY_AMD_LIB_TAG; else export JIMINY_AMD_LIB_TAG=$tag; fi
  timeout 300 python tools/bench_env.py --contact-model constraint --zero-action > $OUT/env_$tag.log 2>&1; echo "$tag env: $(tail -n 1 $OUT/env_$tag.log | cut -c1-250)"
  timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 30 --warmup 5 --contact-model constraint --solver euler_explicit > $OUT/c_$tag.json 2>$OUT/c_$tag.err
  python - <<PY
import json
try:
    b=json.loads(open('$OUT/c_$tag.json').read().strip().splitlines()[-1])
    print('$tag bench', '%.4g'%b['value'], 'launch %.4f'%b['roofline']['avg_launch_ms'])
except Exception as e: print('$tag', 'ERR', e)
PY
done
