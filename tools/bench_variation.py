#!/usr/bin/env python
"""Cost of the per-environment variation on the benchmarked workload (ANYmal, B = 65 536, RK4 dt = 1e-3):
the plain kernel vs the GEN kernel with per-lane body parameters / a height-map ground / a root wrench bound.
Prints one JSON line (ms per launch, wall clock)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jiminy_amd import load_builtin  # noqa: E402
from jiminy_amd.engine import BatchedEngine  # noqa: E402
from jiminy_amd.synthetic import sample_states  # noqa: E402


def run(model, st, B, what, steps=100):
    dev = torch.device("cuda", 0)
    eng = BatchedEngine(model, B, dtype=torch.float64, device=dev)
    eng.set_options({"stepper": {"odeSolver": "runge_kutta_4", "dtMax": 1e-3, "controllerUpdatePeriod": 1e-3,
                                 "sensorsUpdatePeriod": 1e-3}, "contacts": {"model": "spring_damper"}})
    if "model" in what:
        eng.set_model_options({"dynamics": {"massBodiesBiasStd": 0.05, "inertiaBodiesBiasStd": 0.05,
                                            "centerOfMassPositionBodiesBiasStd": 0.02}})
    if "ground" in what:
        rg = np.random.default_rng(0)
        eng.set_ground_heightmap(0.01 * rg.standard_normal((65, 65)), -2.0, -2.0, 1.0 / 16, 1.0 / 16)
    if "force" in what:
        frame = next(n for n, f in model.frames.items() if f.parent_joint == 1)
        w = torch.tensor([20.0, 0, 0, 0, 0, 0], dtype=torch.float64, device=dev)
        eng.register_profile_force(frame, lambda t, q, v: w, update_period=1.0)
    eng.set_command(torch.from_numpy(st["command"]))
    q, v = torch.from_numpy(st["q"]), torch.from_numpy(st["v"])
    eng.start(q, v)
    mask = torch.ones(B, dtype=torch.uint8, device=dev)
    qd, vd = q.to(dev), v.to(dev)
    for _ in range(10):
        eng.step(1e-3)
    torch.cuda.synchronize()
    eng.enable_timing(True)
    t0 = time.perf_counter()
    for i in range(steps):
        eng.step(1e-3)
        if i % 20 == 19:
            eng.reset_lanes(mask, qd, vd)
    torch.cuda.synchronize()
    wall = 1e3 * (time.perf_counter() - t0) / steps
    n, ms = eng.timing_summary()
    return {"wall_ms": wall, "kernel_ms": ms / max(n, 1), "nan_lanes": int((eng.status & 1).sum())}


def main():
    model = load_builtin("anymal")
    B = 65536
    st = sample_states(model, B, seed=0)
    out = {"metric": "ms per launch, anymal B=65536 RK4 dt=1e-3 (wall clock, re-seed every 20 steps)"}
    for what in ("plain", "model", "ground", "force", "model+ground+force"):
        out[what] = run(model, st, B, what)
    rows = 13 * model.njoints
    out["model_lane_bytes_per_env"] = rows * 8
    print(json.dumps(out))


if __name__ == "__main__":
    main()
