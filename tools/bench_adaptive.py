#!/usr/bin/env python
"""Throughput of the adaptive Dormand-Prince solver (`odeSolver = "runge_kutta_dopri"`, the reference's
default; one step size per lane): robot-intervals/s over breakpoint intervals of `--interval` seconds.
    python tools/bench_adaptive.py [--model anymal] [--batch 65536] [--intervals 5] [--tol-abs 1e-5 --tol-rel 1e-4]
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jiminy_amd import load_builtin  # noqa: E402
from jiminy_amd.engine import BatchedEngine  # noqa: E402
from jiminy_amd.synthetic import sample_states  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="anymal")
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--intervals", type=int, default=5)
    ap.add_argument("--interval", type=float, default=0.01)
    ap.add_argument("--tol-abs", type=float, default=1e-5)
    ap.add_argument("--tol-rel", type=float, default=1e-4)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = load_builtin(args.model)
    B = args.batch
    st = sample_states(model, B, seed=0)
    eng = BatchedEngine(model, B, dtype=torch.float64, device=dev)
    eng.set_options({"stepper": {"odeSolver": "runge_kutta_dopri", "tolAbs": args.tol_abs, "tolRel": args.tol_rel,
                                 "controllerUpdatePeriod": args.interval, "sensorsUpdatePeriod": args.interval}, "contacts": {"model": "spring_damper"}})
    eng.set_command(torch.from_numpy(st["command"]))
    eng.start(torch.from_numpy(st["q"]), torch.from_numpy(st["v"]))
    eng.step(args.interval)  # leaves the 1 us initial step size behind
    torch.cuda.synchronize()
    it0 = eng.stepper_state.iter_lanes.double().mean().item()
    if0 = eng.stepper_state.iter_failed_lanes.double().mean().item()
    attempts = 0
    t0 = time.perf_counter()
    for _ in range(args.intervals):
        eng.step(args.interval)
        attempts += eng.adaptive_attempts
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ss = eng.stepper_state
    status = eng.status
    print(json.dumps({
        "metric": f"robot-intervals/s {args.model} runge_kutta_dopri", "value": B * args.intervals / el,
        "ms_per_interval": 1e3 * el / args.intervals, "interval_s": args.interval, "batch": B,
        "tol_abs": args.tol_abs, "tol_rel": args.tol_rel,
        "device_attempts_per_interval": attempts / args.intervals,
        "mean_accepted_steps_per_interval": (ss.iter_lanes.double().mean().item() - it0) / args.intervals,
        "mean_rejected_steps_per_interval": (ss.iter_failed_lanes.double().mean().item() - if0) / args.intervals,
        "lanes_failed": ((status & 9) != 0).double().mean().item()}))


if __name__ == "__main__":
    main()
