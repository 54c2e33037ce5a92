"""Persistent adaptive kernel against the per-stage launches on the probe batch, verbose:
    JIMINY_AMD_LIB_TAG=<tag> python tools/dopri_debug.py [intervals]"""
import os, sys
import numpy as np, torch
from jiminy_amd import load_builtin, _abi
from jiminy_amd import engine as E
from jiminy_amd.engine import BatchedEngine
m = load_builtin("atlas")
dev = torch.device("cuda", 0)
n = 64
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
q, v, cmd = (torch.as_tensor(x, dtype=torch.float64, device=dev) for x in E._probe_state(m, n))
outs = []
for form in (1, 0):
    p = BatchedEngine(m, n, dtype=torch.float64, device=dev, extra_outputs=(), _lib_variant=0)
    p._adaptive_form_override = form
    p.set_options({"stepper": {"odeSolver": "runge_kutta_dopri", "tolAbs": 1e-8, "tolRel": 1e-7, "dtMax": 1e-3, "controllerUpdatePeriod": 1e-3, "sensorsUpdatePeriod": 1e-3}, "contacts": {"model": "spring_damper"}})
    p.set_command(cmd); p.start(q, v)
    for _ in range(nsteps): p.step(1e-3)
    ss = p.stepper_state
    outs.append(dict(q=p._fields["q"].cpu().numpy(), v=p._fields["v"].cpu().numpy(), it=ss.iter_lanes.cpu().numpy(), itf=ss.iter_failed_lanes.cpu().numpy(),
                     st=p.status.reshape(-1).cpu().numpy(), dt=ss.dt_lanes.cpu().numpy(), att=p.adaptive_attempts))
    p.stop()
a, b = outs
print("attempts per-stage / persistent:", a["att"], b["att"])
print("status per-stage:", np.unique(a["st"], return_counts=True), " persistent:", np.unique(b["st"], return_counts=True))
print("iter   per-stage:", a["it"][:8], " persistent:", b["it"][:8])
print("failed per-stage:", a["itf"][:8], " persistent:", b["itf"][:8])
print("dt     per-stage:", a["dt"][:4], " persistent:", b["dt"][:4])
same = (a["it"] == b["it"]) & (a["itf"] == b["itf"])
print("same sequences:", same.mean())
dq = np.abs(a["q"] - b["q"]); dv = np.abs(a["v"] - b["v"])
print("max |dq| per row:", np.array2string(dq.max(axis=1), precision=1, max_line_width=250))
print("max |dv| per row:", np.array2string(dv.max(axis=1), precision=1, max_line_width=250))
