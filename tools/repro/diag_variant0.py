#!/usr/bin/env python
"""Where does build variant 0 of crane_walker go wrong?  Per step / field / row / lane differences between the
run that takes a(t+) from the in-loop copy of the evaluation and the one that takes it from the peeled copy
(MI355X box; see repro_crane_walker.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jiminy_amd import engine as E  # noqa: E402
from tests import robots  # noqa: E402

model = robots.crane_walker()
dev = torch.device("cuda", 0)
n, dt = 64, 1e-4
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
solver = sys.argv[2] if len(sys.argv) > 2 else "euler_explicit"
q, v, cmd = (torch.as_tensor(x, dtype=torch.float64, device=dev) for x in E._probe_state(model, n))
np.set_printoptions(linewidth=200, precision=3)
runs = {}
for changed in (False, True):
    for ref in (1, variant):
        p = E.BatchedEngine(model, n, dtype=torch.float64, device=dev, extra_outputs=(), _lib_variant=ref)
        p.set_options({"stepper": {"odeSolver": solver, "dtMax": dt, "controllerUpdatePeriod": 0.0,
                                   "sensorsUpdatePeriod": 0.0}, "contacts": {"model": "spring_damper"}})
        p.set_command(cmd)
        p.start(q, v)
        res = [("start.a", p._fields["a"].cpu().numpy().copy())]
        for i in range(2):
            if changed:
                p.mark_command_changed()
            p.step(dt)
            for f in ("q", "v", "a"):
                res.append((f"step{i}.{f}", p._fields[f].cpu().numpy().copy()))
        runs[(changed, ref)] = res
        p.stop()
for changed in (False, True):
    print(f"==== command_changed={changed}: variant {variant} against variant 1 (known good)")
    for (name, x), (_, y) in zip(runs[(changed, variant)], runs[(changed, 1)]):
        d = np.abs(x - y)
        d[np.isnan(d)] = np.inf
        rows = np.nonzero(d.max(axis=1) > 1e-12)[0]
        lanes = np.nonzero(d.max(axis=0) > 1e-12)[0]
        print(f"{name:10s} max {d.max():.3e} rows {rows.tolist()} lanes({len(lanes)}) {lanes.tolist()[:40]}")
        if len(rows):
            print("           per-row max:", d.max(axis=1)[rows])
