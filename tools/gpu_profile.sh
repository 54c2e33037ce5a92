#!/bin/bash
# One profiled workload on the GPU box: the bench line, rocprofv3 kernel stats, then the PMC passes (separate
# runs, --pmc never combined with trace domains other than --kernel-trace), summarised on the box.
# Usage (through gpurun):  bash tools/gpu_profile.sh <tag> <latest-file> [bench.py arguments...]
# Output: gpurun_out/<tag>/summary/{<tag>_kernel_stats.csv,<tag>_pmc.json,<latest-file>} -> copy into profiles/.
set -u
TAG=$1; LATEST=$2; shift 2
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$REPO
timeout 600 python bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; tail -c 400 "$OUT/bench.json"; echo
BENCH="python $REPO/bench.py --steps 60 --warmup 5 --no-cpu-baseline $*"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run -- $BENCH > "$OUT/stats.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" \
         "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_$N" -o run -- $BENCH > "$OUT/pmc_$N.log" 2>&1
  echo "pmc $N rc=$?"
done
cd "$REPO"
python tools/summarise_profiles.py "$OUT" "$TAG" "$OUT/summary" "$LATEST" > "$OUT/summary.log" 2>&1
tail -45 "$OUT/summary.log"
find "$OUT" -name '*.db' -delete
du -sh "$OUT"
