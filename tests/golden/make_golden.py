"""Regression fixtures of the CPU oracle (tests/golden/*.npz).

These are NOT reference outputs (the reference cannot run here): they
freeze the oracle's own outputs on seeded inputs, after it passed the known-answer pins of
tests/test_oracle_known_answers.py, so that later edits of oracle.cpp / the model compiler cannot
silently change the numbers every GPU parity test is compared with.
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from jiminy_amd import load_builtin  # noqa: E402
from jiminy_amd.synthetic import sample_standing_states, sample_states  # noqa: E402
from tests.helpers import (ReferenceFixedStepLoop, alloc_constraint_state, alloc_soa, oracle_batch,  # noqa: E402
                           oracle_engine_step)

OUT = os.path.join(ROOT, "tests", "golden")
FIELDS = ("q", "v", "a", "u_motor", "imu", "force", "encoder", "effort", "energy", "contact_forces")
CON_FIELDS = FIELDS + ("u", "con_data", "con_flags")
CON_OPTIONS = dict(tol_abs=1e-11, tol_rel=1e-10)  # PGS run to stagnation: fixtures insensitive to round-off


def main() -> None:
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]     # optional: regenerate the fixtures of these models only
    for name, B, kw in (("cartpole", 16, {}), ("anymal", 16, {"grounded_fraction": 0.5}),
                        ("atlas", 8, {"base_height": (0.9, 1.0), "grounded_fraction": 0.5}),
                        ("arm7", 16, {})):          # (the authored 7-joint arm: the one-robot-per-lane kernels' robot)
        if only and name not in only:
            continue
        model = load_builtin(name)
        st = sample_states(model, B, seed=123, **kw)
        arr = alloc_soa(model, B)
        for k in ("q", "v", "command"):
            arr[k][:] = st[k]
        oracle_batch(model, arr, "start")
        loop = ReferenceFixedStepLoop(5e-4)     # `Engine::step` periods: the first one opens with the reference's 1 us step
        out = {"in_q": st["q"], "in_v": st["v"], "in_command": st["command"]}
        out.update({"start_" + k: arr[k].copy() for k in FIELDS})
        for i in range(10):
            oracle_engine_step(model, arr, loop, 5e-4, "runge_kutta_4", command_changed=(i == 0))
        out.update({"rk4_" + k: arr[k].copy() for k in FIELDS})
        out["status"] = arr["status"].copy()
        np.savez_compressed(os.path.join(OUT, f"{name}_oracle.npz"), **out)
        print(name, "ok", {k: v.shape for k, v in list(out.items())[:3]})
    # contacts.model = "constraint": standing robots, some joints beyond their limits
    for name, B in (("anymal", 16), ("atlas", 4)):
        if only and name not in only:
            continue
        model = load_builtin(name)
        st = sample_standing_states(model, B, seed=321)
        arr = alloc_soa(model, B)
        alloc_constraint_state(model, arr, B)
        for k in ("q", "v", "command"):
            arr[k][:] = st[k]
        oracle_batch(model, arr, "start", constraint_options=CON_OPTIONS)
        loop = ReferenceFixedStepLoop(5e-4)
        out = {"in_q": st["q"], "in_v": st["v"], "in_command": st["command"]}
        out.update({"start_" + k: arr[k].copy() for k in CON_FIELDS})
        for i in range(6):
            oracle_engine_step(model, arr, loop, 5e-4, "euler_explicit", command_changed=True, constraint_options=CON_OPTIONS)
        out.update({"euler_" + k: arr[k].copy() for k in CON_FIELDS})
        out["status"] = arr["status"].copy()
        np.savez_compressed(os.path.join(OUT, f"{name}_constraint_oracle.npz"), **out)
        print(name, "constraint ok, active constraints per lane", (arr["con_flags"] & 1).sum(0))


if __name__ == "__main__":
    main()
