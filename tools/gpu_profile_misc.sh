#!/bin/bash
# kernel-trace statistics of the workloads that are not fixed-step bench lines: the environment benches and the adaptive solver
#   gpurun -- 'bash tools/gpu_profile_misc.sh r04b'   ->  gpurun_out/<tag>_misc/<tag>_{env_*,dopri_*}_kernel_stats.csv  (copy to profiles/)
set -u
exec < /dev/null
TAG=${1:-r05}
REPO=$(pwd); export PYTHONPATH=$REPO TMPDIR=/tmp JIMINY_AMD_SELF_TEST=0
OUT=$REPO/gpurun_out/${TAG}_misc; rm -rf $OUT; mkdir -p $OUT
run() {  # name, script, args...
  local name=$1 script=$2; shift 2
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace -d $OUT/$name -o run -- python $REPO/tools/$script "$@" > $OUT/$name.log 2>&1
  cd $REPO
  grep -o '"value": [0-9.]*' $OUT/$name.log | head -1
  timeout 60 python tools/rocpd_stats.py $OUT/$name 12 > $OUT/${TAG}_${name}_kernel_stats.csv; head -4 $OUT/${TAG}_${name}_kernel_stats.csv | cut -c1-150
  find $OUT/$name -name '*.db' -delete
}
run env_spring bench_env.py --contact-model spring_damper
run env_constraint bench_env.py --contact-model constraint --zero-action
run env_atlas_constraint bench_env.py --model atlas --envs 32768 --contact-model constraint --zero-action --steps 10
run dopri_anymal bench_adaptive.py --model anymal
run dopri_atlas bench_adaptive.py --model atlas --batch 32768
