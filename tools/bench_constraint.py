#!/usr/bin/env python
"""env-steps/s of the constraint contact model (`contacts.model = "constraint"`, the option the
reference's shipped ANYmal / Atlas files select) on one GPU: standing robots (several active contact
constraints per lane, a fraction of the lanes with joints beyond their limits), held command.
    python tools/bench_constraint.py [--model anymal] [--batch 65536] [--steps 50] [--solver euler_explicit]
Prints one JSON line (kernel time from the library's per-launch HIP events + wall clock)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jiminy_amd import _abi, load_builtin  # noqa: E402
from jiminy_amd.engine import BatchedEngine  # noqa: E402
from jiminy_amd.synthetic import sample_standing_states  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="anymal")
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--solver", default="euler_explicit")
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--episode", type=int, default=25, help="re-seed every lane every N steps")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = load_builtin(args.model)
    B = args.batch
    st = sample_standing_states(model, B, seed=0)
    q0, v0 = torch.from_numpy(st["q"]).to(dev), torch.from_numpy(st["v"]).to(dev)
    eng = BatchedEngine(model, B, dtype=torch.float64, device=dev)
    eng.set_options({"stepper": {"odeSolver": args.solver, "dtMax": args.dt, "controllerUpdatePeriod": args.dt,
                                 "sensorsUpdatePeriod": args.dt}, "contacts": {"model": "constraint"}})
    eng.set_command(torch.from_numpy(st["command"]))
    eng.start(q0, v0)
    torch.cuda.synchronize()
    active0 = (eng.field("con_flags") & 1).sum(0).double().mean().item()
    mask = torch.ones(B, dtype=torch.uint8, device=dev)
    for i in range(args.warmup):
        eng.step(args.dt)
    eng.reset_lanes(mask, q0, v0)
    torch.cuda.synchronize()
    eng.enable_timing(True)
    t0 = time.perf_counter()
    failed = 0.0
    for i in range(args.steps):
        eng.step(args.dt)
        if (i + 1) % args.episode == 0 and i + 1 < args.steps:
            eng.reset_lanes(mask, q0, v0)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    n, ms = eng.timing_summary()
    status = eng.status
    failed = ((status & _abi.JM_LANE_SOLVER_FAILURE) != 0).double().mean().item()
    nan = ((status & _abi.JM_LANE_NAN) != 0).double().mean().item()
    active1 = (eng.field("con_flags") & 1).sum(0).double().mean().item()
    rows = _abi.constraint_rows(model)
    print(json.dumps({
        "metric": f"env-steps/s {args.model} constraint contact model", "value": B * args.steps / el,
        "kernel_env_steps_per_s": B * n / (ms * 1e-3) if ms > 0 else None,
        "ms_per_launch": ms / max(n, 1), "batch": B, "steps": args.steps, "solver": args.solver, "dt": args.dt,
        "mean_active_constraints_start": active0, "mean_active_constraints_end": active1,
        "constraint_rows_max": rows["n_rows"], "workspace_MB": eng.field("workspace").numel() * 8 / 1e6,
        "lanes_pgs_not_converged_last_eval": failed, "lanes_nan": nan}))


if __name__ == "__main__":
    main()
