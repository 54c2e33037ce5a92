#!/usr/bin/env python
"""Can ONE process feed its GPU at the per-GPU shard sizes of BASELINE.json's multi-GPU configurations?

For every workload the table gives, per call (`engine.step` or `env.step`):
  * `host_issue_ms`  -- what the process needs to ISSUE one call: Python + ctypes + HIP launches, measured on a batch of
                        64 robots, where the kernels are shorter than their launches and the call never waits for the
                        device (no synchronisation inside the timed loop);
  * `gpu_ms`         -- wall clock per call at the shard size with the queue kept full (one synchronisation at the end);
  * `fed`            -- host_issue_ms < gpu_ms: the GPU never waits for the host; the ratio says by how much.
and the same with the environment step replayed as a captured HIP graph (one `hipGraphLaunch` per call).
    python tools/host_cost.py  > gpurun_out/<tag>/host_cost.json        (prints one JSON document)
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jiminy_amd import load_builtin  # noqa: E402
from jiminy_amd.engine import BatchedEngine  # noqa: E402
from jiminy_amd.envs import make_anymal_env, make_atlas_env  # noqa: E402
from jiminy_amd.synthetic import sample_states  # noqa: E402

DEV = torch.device("cuda", 0)


def timed(call, n, sync_each=False):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        call()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e3 * t_issue / n, 1e3 * (time.perf_counter() - t0) / n


def engine_case(name, B, dt, contact_model, solver):
    model = load_builtin(name)
    out = {}
    for tag, b in (("host", 64), ("shard", B)):
        st = sample_states(model, b, seed=1, **({"base_height": (0.9, 1.1)} if name == "atlas" else {}))
        eng = BatchedEngine(model, b, dtype=torch.float64, device=DEV)
        eng.set_options({"stepper": {"odeSolver": solver, "dtMax": dt, "controllerUpdatePeriod": dt, "sensorsUpdatePeriod": dt},
                         "contacts": {"model": contact_model}})
        eng.set_command(torch.from_numpy(st["command"]))
        eng.start(torch.from_numpy(st["q"]), torch.from_numpy(st["v"]))
        for _ in range(5):
            eng.step(dt)
        issue, wall = timed(lambda: eng.step(dt), 200 if tag == "host" else 60)
        out[tag] = (issue, wall)
        eng.stop()
    return {"call": "engine.step", "model": name, "lanes_per_gpu": B, "contact_model": contact_model, "solver": solver,
            "host_issue_ms": out["host"][0], "gpu_ms": out["shard"][1], "fed": out["host"][0] < out["shard"][1],
            "gpu_over_host": out["shard"][1] / out["host"][0]}


def env_case(name, B, contact_model, graph):
    make = make_atlas_env if name == "atlas" else make_anymal_env
    out = {}
    for tag, b in (("host", 64), ("shard", B)):
        env = make(b, device=DEV, contact_model=contact_model)
        env.reset(seed=0)
        if graph:
            env.enable_graph(whole_step=(graph == "whole"))
        action = torch.zeros((b, env.engine.model.nmotors), dtype=torch.float64, device=DEV)
        for _ in range(4):
            env.step(action)
        out[tag] = timed(lambda: env.step(action), 40 if tag == "host" else 10)
        env.close() if hasattr(env, "close") else None
    return {"call": "env.step" + (f" (HIP graph, {graph})" if graph else ""), "model": name, "lanes_per_gpu": B,
            "contact_model": contact_model, "host_issue_ms": out["host"][0], "gpu_ms": out["shard"][1],
            "fed": out["host"][0] < out["shard"][1], "gpu_over_host": out["shard"][1] / out["host"][0]}


def main():
    rows = []
    # config 3 / 4 of BASELINE.json sharded over 8 GPUs, and the whole batches on one
    for B in (8192, 65536):
        rows.append(engine_case("anymal", B, 1e-3, "spring_damper", "runge_kutta_4"))
    for B in (4096, 32768):
        rows.append(engine_case("atlas", B, 2.5e-4, "spring_damper", "runge_kutta_4"))
    rows.append(engine_case("anymal", 8192, 1e-3, "constraint", "euler_explicit"))
    rows.append(engine_case("atlas", 4096, 1e-3, "constraint", "euler_explicit"))
    # config 5: the PD + Mahony pipeline environments (40 ms gym step = 8 controller ticks x 5 integrator steps)
    for graph in (None, "physics", "whole"):
        rows.append(env_case("anymal", 8192, "spring_damper", graph))
    rows.append(env_case("anymal", 8192, "constraint", None))
    rows.append(env_case("atlas", 4096, "constraint", None))
    rows.append(env_case("atlas", 4096, "constraint", "whole"))
    print(json.dumps({"device": torch.cuda.get_device_name(0), "host_cores": os.cpu_count(), "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
