#!/bin/bash
# The 1 / 2 / 4 / 8-GPU lines of one node in one call (the driver runs the same `bench.py --gpus N`):
#   weak scaling on ANYmal (config 3 per GPU), BASELINE config 4 (`--strong`: Atlas, 32 768 robots sharded, float32
#   observation all-gather every 8th step), and the weak line with the asynchronous gather switched on.
# One rank per GPU under torch.distributed.run, rendezvous on 127.0.0.1, every rank pinned to the cores of its GPU's
# NUMA node (rocm-smi --showtoponuma) so that its launch thread does not migrate.
#   bash tools/gpu_scale.sh [out_dir]     -> one JSON line per run in <out_dir>/scale.jsonl
set -u
OUT=${1:-gpurun_out/scale}
mkdir -p "$OUT"
REPO=$(cd "$(dirname "$0")/.." && pwd)
export PYTHONPATH=$REPO HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c 'import torch; print(torch.cuda.device_count())')
: > "$OUT/scale.jsonl"
run() {   # run <n> <tag> <bench args...>
  local n=$1 tag=$2; shift 2
  local launcher=(python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $((29600 + n)))
  if command -v numactl > /dev/null; then
    # --no-python + a wrapper that binds by LOCAL_RANK: the NUMA node of GPU i as rocm-smi reports it (node 0 when unknown)
    launcher+=(--no-python bash -c 'node=$(rocm-smi --showtoponuma 2>/dev/null | awk -v g="GPU[$LOCAL_RANK]" "\$1==g && /Numa Node/ {print \$NF; exit}"); exec numactl --cpunodebind=${node:-0} --membind=${node:-0} python "$@"' _)
  fi
  "${launcher[@]}" "$REPO/bench.py" --gpus "$n" --steps 40 --warmup 5 --no-cpu-baseline --no-secondary "$@" 2> "$OUT/${tag}_n$n.err" | grep '^{' | tail -1 | \
    python -c "import sys, json; r = json.loads(sys.stdin.read()); r['run'] = '$tag'; print(json.dumps(r))" >> "$OUT/scale.jsonl"
  tail -1 "$OUT/scale.jsonl" | python -c "import sys, json; r = json.loads(sys.stdin.read()); print('$tag', 'n =', r['n_gpus'], 'value =', '%.3e' % r['value'], r['unit'], 'ms/step =', round(r['ms_per_step'], 4))"
}
for N in 1 2 4 8; do
  [ "$N" -le "$NGPU" ] || break
  run "$N" weak
  run "$N" weak_gather --gather-obs --gather-dtype f32
  run "$N" config4_strong --model atlas --batch 32768 --strong --dt 2.5e-4 --gather-obs --gather-dtype f32 --gather-every 8
done
python - "$OUT/scale.jsonl" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
for tag in sorted({r["run"] for r in rows}):
    rs = sorted((r for r in rows if r["run"] == tag), key=lambda r: r["n_gpus"])
    base = rs[0]["value"] / rs[0]["n_gpus"]
    print(tag, " ".join(f"N={r['n_gpus']}: {r['value']:.3e} ({r['value'] / (base * r['n_gpus']):.2f})" for r in rs))
PY
