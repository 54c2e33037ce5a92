#!/usr/bin/env python
"""Cost of `start` / `reset_lanes` next to a step, per contact model (HIP events around each call):
    python tools/bench_reset.py [--model anymal] [--batch 65536]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from jiminy_amd import load_builtin
    from jiminy_amd.engine import BatchedEngine
    from jiminy_amd.synthetic import sample_standing_states, sample_states
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="anymal")
    ap.add_argument("--batch", type=int, default=65536)
    args = ap.parse_args()
    model, B = load_builtin(args.model), args.batch
    dev = torch.device("cuda", 0)

    def timed(fn, reps=3):
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record(); fn(); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return min(ts)

    for cm, solver, dt in (("spring_damper", "runge_kutta_4", 1e-3), ("constraint", "euler_explicit", 1e-3)):
        st = sample_standing_states(model, B, seed=0, joint_noise=0.01, base_angle_max=0.004, depth_range=(-6e-3, -5e-3),
                                    twist_std=0.02, joint_vel_std=0.05, command_fraction=0.1, out_of_bounds_fraction=0.05) \
            if cm == "constraint" else sample_states(model, B, seed=0)
        eng = BatchedEngine(model, B, dtype=torch.float64, device=dev)
        eng.set_options({"stepper": {"odeSolver": solver, "dtMax": dt, "controllerUpdatePeriod": dt, "sensorsUpdatePeriod": dt},
                         "contacts": {"model": cm}})
        q, v = torch.from_numpy(st["q"]).to(dev), torch.from_numpy(st["v"]).to(dev)
        eng.set_command(torch.from_numpy(st["command"]))
        eng.start(q, v)
        eng.step(dt)
        res = {"model": args.model, "batch": B, "contact_model": cm, "solver": solver}
        res["step_ms"] = timed(lambda: eng.step(dt), 5)
        for frac in (1.0, 0.25, 0.05, 0.01, 0.001):
            mask = (torch.rand(B, device=dev) < frac).to(torch.uint8) if frac < 1 else torch.ones(B, dtype=torch.uint8, device=dev)
            res[f"reset_{frac:g}_ms"] = timed(lambda: eng.reset_lanes(mask, q, v))
        eng.stop()
        res["start_ms"] = timed(lambda: (eng.start(q, v), eng.stop()), 2)
        print(json.dumps(res))


if __name__ == "__main__":
    main()
