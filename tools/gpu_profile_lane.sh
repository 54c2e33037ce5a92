#!/bin/bash
# The one-robot-per-lane kernels under rocprofv3: kernel stats, then the PMC passes (separate runs), summarised on the box.
# Usage (through gpurun):  bash tools/gpu_profile_lane.sh <tag> [tools/bench_lane.py arguments...]
# Output: gpurun_out/<tag>/summary/{<tag>_kernel_stats.csv,<tag>_pmc.json} -> copy into profiles/.
set -u
TAG=$1; shift 1
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$REPO
timeout 300 python tools/bench_lane.py "$@" > "$OUT/bench_lane.json" 2> "$OUT/bench.err"
tail -c 500 "$OUT/bench_lane.json"; echo
BENCH="python $REPO/tools/bench_lane.py --steps 60 $*"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run -- $BENCH > "$OUT/stats.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" \
         "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_$N" -o run -- $BENCH > "$OUT/pmc_$N.log" 2>&1
  echo "pmc $N rc=$?"
done
cd "$REPO"
python tools/summarise_profiles.py "$OUT" "$TAG" "$OUT/summary" pmc_lane_latest.json > "$OUT/summary.log" 2>&1
tail -40 "$OUT/summary.log"
find "$OUT" -name '*.db' -delete
du -sh "$OUT"
