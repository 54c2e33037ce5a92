#!/bin/bash
# rocprofv3 kernel statistics of one bench command (per-kernel average durations), summarised on the box.
# Usage (through gpurun): bash tools/gpu_kstats.sh <tag> <bench args...>     (env JIMINY_AMD_LIB_TAG etc. are inherited)
set -u
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$REPO JIMINY_AMD_SELF_TEST=0
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- python $REPO/bench.py --no-cpu-baseline --no-secondary "$@" > $OUT/stats.log 2>&1
tail -2 $OUT/stats.log | cut -c1-300
python $REPO/tools/rocpd_stats.py $OUT/stats 14 > $OUT/kernel_stats.csv; cat $OUT/kernel_stats.csv | cut -c1-170
find $OUT -name '*.db' -delete
