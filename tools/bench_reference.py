#!/usr/bin/env python
"""Time the REAL reference (jiminy_py, single-threaded C++ core) on the bench workload -- run on a host
where `import jiminy_py` works; prints one JSON object shaped like bench.py's `cpu_baseline` with
`"kind": "reference"`.  bench.py itself reports the oracle port
(`"kind": "port"`) because neither the build container nor the GPU box can import jiminy_py.

    python tools/bench_reference.py --data /path/to/jiminy/data [--seconds 20]

Workload = BASELINE.json configs[2]: ANYmal, spring-damper contacts, `runge_kutta_4`, dt = 1e-3, command
held, sensors refreshed every step; one `jiminy.Engine`, one robot, one thread, lanes of the seeded
bench batch visited one after the other (`engine.start` excluded from the timed region).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", required=True)
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--steps-per-lane", type=int, default=20)
    ap.add_argument("--dt", type=float, default=1e-3)
    args = ap.parse_args()
    try:
        import jiminy_py.core as jiminy
    except ImportError as e:
        raise SystemExit(f"jiminy_py is not importable on this host ({e})")
    from jiminy_amd import load_builtin
    from jiminy_amd.synthetic import sample_states
    from tools.dump_reference import MODELS, build_reference_robot, model_pins, to_reference_order

    name = "anymal"
    urdf = os.path.join(args.data, MODELS[name][0])
    robot = build_reference_robot(jiminy, name, urdf, True)
    pins = model_pins(robot)
    model = load_builtin(name)
    st = sample_states(model, 4096, seed=0)
    motor_names = [str(n) for n in pins["pin_motor_names"]]
    cmd_perm = [motor_names.index(m.name) for m in model.motors]
    command = np.zeros(len(motor_names))

    def compute_command(t, q, v, sensor_measurements, u_command):
        u_command[:] = command
    robot.controller = jiminy.FunctionalController(compute_command, None)
    engine = jiminy.Engine()
    engine.add_robot(robot)
    opts = engine.get_options()
    opts["stepper"].update({"odeSolver": "runge_kutta_4", "dtMax": args.dt, "controllerUpdatePeriod": args.dt,
                            "sensorsUpdatePeriod": args.dt})
    opts["contacts"].update({"model": "spring_damper"})
    opts["telemetry"].update({k: False for k in opts["telemetry"] if k.startswith("enable")})
    engine.set_options(opts)
    done, timed, lane = 0, 0.0, 0
    while timed < args.seconds and lane < st["q"].shape[1]:
        for i_ours, i_ref in enumerate(cmd_perm):
            command[i_ref] = st["command"][i_ours, lane]
        qr, vr = to_reference_order(model, pins, st["q"][:, lane], st["v"][:, lane])
        engine.reset(False)
        try:
            engine.start(qr, vr)
        except Exception:
            lane += 1
            continue
        t0 = time.perf_counter()
        try:
            for _ in range(args.steps_per_lane):
                engine.step(args.dt)
            done += args.steps_per_lane
        except Exception:
            pass
        timed += time.perf_counter() - t0
        engine.stop()
        lane += 1
    print(json.dumps({"value": done / timed, "unit": "env-steps/s", "cores": 1, "kind": "reference",
                      "sample": f"{lane} lanes x {args.steps_per_lane} RK4 steps dt={args.dt} of the seeded ANYmal "
                                f"bench batch through jiminy_py {getattr(jiminy, '__version__', '?')} "
                                f"(one Engine, one thread, {timed:.1f} s timed)"}))


if __name__ == "__main__":
    main()
