#!/usr/bin/env python
"""gym-steps/s of the device-resident ANYmal pipeline (BASELINE.json configs[4] shape on one GPU):
PDAdapter -> PDController (5 ms) -> physics (explicit Euler, dtMax 1 ms, spring-damper ground)
-> Mahony filter, 40 ms per environment step, random actions, B environments.
    python tools/bench_env.py [--envs 65536] [--steps 20]
Prints one JSON line.  JIMINY_AMD_TENSOR_BLOCKS=1 runs the per-tick blocks as tensor programs."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jiminy_amd.envs import make_anymal_env  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="anymal", choices=("anymal", "atlas"))
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--solver", default="euler_explicit")
    ap.add_argument("--contact-model", default="spring_damper", choices=("spring_damper", "constraint"))
    ap.add_argument("--zero-action", action="store_true", help="hold the neutral stance (standing robots)")
    ap.add_argument("--graph", action="store_true", help="replay one environment step as one captured HIP graph")
    ap.add_argument("--whole-step", action="store_true", help="with --graph: capture the whole env.step (episode clock, "
                    "termination, reward, masked auto-reset): no host read-back per step")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    if args.model == "atlas":
        from jiminy_amd.envs import make_atlas_env
        env = make_atlas_env(args.envs, device=dev, ode_solver=args.solver, contact_model=args.contact_model)
    else:
        env = make_anymal_env(args.envs, device=dev, ode_solver=args.solver, contact_model=args.contact_model)
    env.reset(seed=0)
    if args.graph:
        env.enable_graph(whole_step=args.whole_step)
    g = torch.Generator(device="cpu").manual_seed(0)
    n_act = env.engine.model.nmotors
    action = ((torch.rand(args.envs, n_act, generator=g, dtype=torch.float64) - 0.5) * 0.5).to(dev)
    if args.zero_action:
        action = torch.zeros_like(action)
    for _ in range(args.warmup):
        env.step(action)
    # one untimed lane reset: the first use of the reset path loads its kernels (tens of ms, once per process)
    warm = torch.zeros(args.envs, dtype=torch.bool, device=dev)
    warm[:1] = True
    env.reset_lanes(warm)
    env.step(action)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_reset_dev = torch.zeros((), dtype=torch.int64, device=dev)
    for _ in range(args.steps):
        _, _, _, _, info = env.step(action)
        if "reset_mask" in info:
            n_reset_dev += info["reset_mask"].sum()     # (on the device: no read-back inside the timed loop)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    n_reset = int(n_reset_dev)
    extra = {}
    if args.contact_model == "constraint":
        eng = env.engine
        extra = {"mean_active_constraints": (eng.field("con_flags") & 1).sum(0).double().mean().item(),
                 "mean_base_height": eng.field("q")[2].mean().item(),
                 "lanes_pgs_not_converged_last_eval": ((eng.status & 16) != 0).double().mean().item()}
    print(json.dumps({"metric": f"gym-steps/s {args.model} PD + Mahony pipeline", "value": args.envs * args.steps / el,
                      "contact_model": args.contact_model, **extra,
                      "ms_per_env_step": 1e3 * el / args.steps, "envs": args.envs, "steps": args.steps,
                      "integrator_steps_per_env_step": 40, "solver": args.solver, "graph": bool(args.graph), "whole_step": bool(args.whole_step), "lanes_reset": n_reset,
                      "blocks": "tensor" if os.environ.get("JIMINY_AMD_TENSOR_BLOCKS") == "1" else "hip"}))


if __name__ == "__main__":
    main()
