#!/bin/bash
# What the VALU-issue floor of bench.py's `roofline.secondary` is made of, measured on the box (round 5):
#   1. tools/microbench/op_issue: cycles per wave-instruction of every instruction class of the k_quad loop and the
#      clock the chip sustains under each (s_memtime / s_memrealtime inside the kernel);
#   2. the clock under k_quad itself: GRBM_GUI_ACTIVE (gfx-clock cycles the GPU was busy during the dispatch) /
#      dispatch duration, from one rocprofv3 --pmc pass over the bench command (kernel-trace only, no other domain).
# Usage (through gpurun): bash tools/measure_floor.sh <tag>   -> gpurun_out/<tag>/op_issue.txt, clock_k_quad.json
set -u
TAG=${1:-floor}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$REPO
"$REPO/tools/microbench/op_issue" > "$OUT/op_issue.txt" 2>&1
cat "$OUT/op_issue.txt"
cd /tmp
BENCH="python $REPO/bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-secondary"
for C in "GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/clk_$N" -o run -- $BENCH > "$OUT/clk_$N.log" 2>&1
done
cd "$REPO"
python tools/pmc_quick.py "$OUT" k_quad > "$OUT/clock_k_quad.json"
python - "$OUT/clock_k_quad.json" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
ns = sorted(r["median_ns"])[len(r["median_ns"]) // 2]
if "GRBM_GUI_ACTIVE" in r:
    r["sustained_clock_ghz"] = r["GRBM_GUI_ACTIVE"] / ns
    print("k_quad: median launch %.1f us, GRBM_GUI_ACTIVE %.0f cycles -> %.3f GHz sustained" % (ns / 1e3, r["GRBM_GUI_ACTIVE"], r["sustained_clock_ghz"]))
json.dump(r, open(sys.argv[1], "w"), indent=1)
PY
find "$OUT" -name '*.db' -delete
