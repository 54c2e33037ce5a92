#!/bin/bash
set -u
exec < /dev/null
REPO=$(pwd); export PYTHONPATH=$REPO TMPDIR=/tmp
OUT=$REPO/gpurun_out/r4k; rm -rf $OUT; mkdir -p $OUT
cat > $OUT/dopri_con.py <<'PY'
import torch, time, sys
from jiminy_amd import load_builtin
from jiminy_amd.engine import BatchedEngine
from jiminy_amd.synthetic import sample_states
name, B, cm = sys.argv[1], int(sys.argv[2]), sys.argv[3]
m = load_builtin(name)
e = BatchedEngine(m, B, dtype=torch.float64, device="cuda:0")
e.set_options({"stepper": {"odeSolver": "runge_kutta_dopri", "dtMax": 0.02, "controllerUpdatePeriod": 0.01, "sensorsUpdatePeriod": 0.01},
               "contacts": {"model": cm}})
st = sample_states(m, B, seed=3, base_height=(0.5, 0.6) if name == "anymal" else (0.95, 1.05), grounded_fraction=0.5, command_fraction=0.1)
e.set_command(torch.from_numpy(st["command"]))
e.start(torch.from_numpy(st["q"]), torch.from_numpy(st["v"]))
e.step(0.01)
torch.cuda.synchronize(); t0 = time.perf_counter(); att = 0
n = 5
for _ in range(n):
    e.step(0.01); att += e.adaptive_attempts
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
ok = ((e.status.reshape(-1) & 9) == 0).float().mean().item()
print(name, cm, B, "ms/interval %.2f" % (dt * 1e3), "attempts/interval %.1f" % (att / n), "robot-intervals/s %.3g" % (B / dt), "ok %.4f" % ok)
PY
timeout 600 python $OUT/dopri_con.py anymal 65536 spring_damper 2>&1 | tail -n 1
timeout 600 python $OUT/dopri_con.py anymal 65536 constraint 2>&1 | tail -n 1
timeout 600 python $OUT/dopri_con.py anymal 4096 constraint 2>&1 | tail -n 1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o dopri_con -- python $OUT/dopri_con.py anymal 65536 constraint > $OUT/prof.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/prof/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    for r in rows[:12]:
        print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
