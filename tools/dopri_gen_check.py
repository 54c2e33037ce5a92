"""The variation form of the persistent adaptive kernel (k_quad_dopri_gen, selected here by a per-lane friction field that
holds the nominal coefficient) against the per-stage launches, on the probe batch -- the check `engine._adaptive_self_test`
runs for the plain kernel:    python tools/dopri_gen_check.py [atlas|anymal]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jiminy_amd import _abi, engine as E, load_builtin  # noqa: E402
from jiminy_amd.engine import BatchedEngine  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "atlas"
    m = load_builtin(name)
    dev = torch.device("cuda", 0)
    n = 64
    q, v, cmd = (torch.as_tensor(x, dtype=torch.float64, device=dev) for x in E._probe_state(m, n))
    outs = []
    for form in (1, 0):
        p = BatchedEngine(m, n, dtype=torch.float64, device=dev, extra_outputs=())
        p._adaptive_form_override = form
        p.set_options({"stepper": {"odeSolver": "runge_kutta_dopri", "tolAbs": 1e-8, "tolRel": 1e-7, "dtMax": 1e-3,
                                   "controllerUpdatePeriod": 1e-3, "sensorsUpdatePeriod": 1e-3},
                       "contacts": {"model": "spring_damper"}})
        p.set_lane_friction(torch.full((n,), float(p._options["contacts"]["friction"]), dtype=torch.float64))
        p.set_command(cmd)
        p.start(q, v)
        for _ in range(3):
            p.step(1e-3)
        ss = p.stepper_state
        outs.append((p._fields["q"].clone(), p._fields["v"].clone(), ss.iter_lanes.clone(), ss.iter_failed_lanes.clone(),
                     p.status.reshape(-1).clone()))
        p.stop()
    (q1, v1, it1, if1, st1), (q0, v0, it0, if0, st0) = outs
    fail = _abi.JM_LANE_NAN | _abi.JM_LANE_STEPPER_FAILURE
    same = (it0 == it1) & (if0 == if1) & ((st1 & fail) == 0) & ((st0 & fail) == 0)
    err = max(float(((x - y)[:, same]).abs().max() / torch.clamp(y[:, same].abs().max(), min=1.0)) for x, y in ((q0, q1), (v0, v1))) \
        if bool(same.any()) else float("inf")
    print({"robot": name, "check": "persistent adaptive kernel, variation form", "same_sequences": float(same.double().mean()),
           "failed_lanes_persistent": int(((st0 & fail) != 0).sum()), "error": err})


if __name__ == "__main__":
    main()
