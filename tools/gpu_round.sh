#!/bin/bash
# One GPU-box pass: parity tests, the bench line, rocprofv3 kernel stats and the PMC passes
# (separate runs, --pmc never combined with trace domains other than --kernel-trace).
# Usage (through gpurun):  bash tools/gpu_round.sh <tag> [skip_tests]
# Everything lands under gpurun_out/<tag>/ ; tools/summarise_profiles.py turns it into profiles/.
set -u
TAG=${1:-r01}
SKIP_TESTS=${2:-0}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export PYTHONPATH=$REPO

if [ "$SKIP_TESTS" != "1" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
  tail -3 "$OUT/pytest_gpu.log"
fi

timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; tail -c 600 "$OUT/bench.json"

# secondary workloads (one JSON line each): adaptive solver, constraint contact model, environment pipelines
timeout 300 python tools/bench_adaptive.py --intervals 5 > "$OUT/adaptive.json" 2>/dev/null
timeout 600 python bench.py --contact-model constraint --steps 40 --warmup 5 > "$OUT/bench_constraint.json" 2>/dev/null
timeout 300 python tools/bench_env.py --envs 65536 --steps 10 --warmup 2 > "$OUT/env_spring.json" 2>/dev/null
timeout 300 python tools/bench_env.py --envs 65536 --steps 5 --warmup 2 --contact-model constraint --zero-action > "$OUT/env_constraint.json" 2>/dev/null
head -c 400 "$OUT/adaptive.json" "$OUT/bench_constraint.json" "$OUT/env_spring.json" "$OUT/env_constraint.json"

BENCH="python $REPO/bench.py --steps 60 --warmup 5 --no-cpu-baseline ${BENCH_EXTRA:-}"
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
# summarise on the box (the rocpd databases are too large to merge back), keep only the summaries
python tools/summarise_profiles.py "$OUT" "$TAG" "$OUT/summary" > "$OUT/summary.log" 2>&1
tail -40 "$OUT/summary.log"
find "$OUT" -name '*.db' -delete
du -sh "$OUT"
