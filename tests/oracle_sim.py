"""A reference-shaped simulation loop around the CPU oracle for the known-answer tests: one robot, breakpoints at the
controller period and at the start / end of every registered impulse force (engine.cc:1838-1893, 1962-2020), the
adaptive Dormand-Prince stepper (the reference's default) or a fixed-step one between them, a log row per breakpoint.
Test infrastructure, like the oracle itself."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from oracle.oracle_py import OracleEngine, adaptive_state
from tests.helpers import ReferenceFixedStepLoop, alloc_constraint_state, alloc_soa, oracle_io

MIN_DT = 1e-10   # STEPPER_MIN_TIMESTEP (constants.h:18)


class OracleSim:
    def __init__(self, model, options: Optional[dict] = None, constraint_options: Optional[dict] = None) -> None:
        self.model = model
        self.arr = alloc_soa(model, 1)
        self.e = OracleEngine(model, **(options or {}))
        if constraint_options is not None:
            alloc_constraint_state(model, self.arr, 1)
            self.e.set_constraint_options(**constraint_options)
            self.e.bind_constraints(self.arr["con_flags"], self.arr["con_data"])
        self.io = oracle_io(self.arr)
        self.forces: List[dict] = []
        self.frames: List[str] = []
        self.wrench: Optional[np.ndarray] = None
        self.t = 0.0
        self.ad = adaptive_state(1)

    def register_impulse_force(self, frame_name: str, t: float, dt: float, wrench: Sequence[float]) -> None:
        if frame_name not in self.frames:
            self.frames.append(frame_name)
        self.forces.append({"frame": self.frames.index(frame_name), "t": float(t), "dt": float(dt),
                            "F": np.asarray(wrench, dtype=np.float64)})

    def start(self, q, v, command=None) -> None:
        self.arr["q"][:, 0] = q
        self.arr["v"][:, 0] = v
        if command is not None:
            self.arr["command"][:, 0] = command
        if self.frames:
            fr = [self.model.frame(n) for n in self.frames]
            self.wrench = np.zeros((6 * len(fr), 1))
            self.e.bind_applied(self.wrench, np.array([f.p for f in fr]), np.array([f.parent_joint for f in fr]))
            self._set_wrench(0.0)
        self.e.batch_run("start", self.io)
        self.t = 0.0
        self.ad = adaptive_state(1)
        self.loop = ReferenceFixedStepLoop(0.0)      # stepper state of a fresh simulation: dt = 1 us (engine.cc:1176)

    def _set_wrench(self, t: float) -> None:
        if self.wrench is None:
            return
        self.wrench[:] = 0.0
        for f in self.forces:
            if f["t"] - MIN_DT <= t < f["t"] + f["dt"] - MIN_DT:
                self.wrench[6 * f["frame"]:6 * f["frame"] + 6, 0] += f["F"]

    def row(self) -> Dict[str, np.ndarray]:
        r = {k: self.arr[k][:, 0].copy() for k in ("q", "v", "a", "energy", "imu", "force", "contact", "f_external",
                                                   "contact_forces", "u")}
        r["t"] = self.t
        return r

    def run(self, t_end: float, solver: str = "runge_kutta_dopri", period: float = 0.0, dt_max: float = 0.02,
            tol_abs: float = 1e-5, tol_rel: float = 1e-4, log_dt: Optional[float] = None) -> Dict[str, np.ndarray]:
        """Advance to `t_end`.  `period` > 0: controller / sensor update period (a breakpoint each); `log_dt`: extra
        log points in continuous mode (the stepper is stopped there too: the analogue of logging internal steps)."""
        log = [self.row()]
        force_pts = sorted({x for f in self.forces for x in (f["t"], f["t"] + f["dt"])})
        k = 0
        while t_end - self.t > MIN_DT:
            cands = [t_end] + [x for x in force_pts if x > self.t + MIN_DT]
            if period > 0.0:
                k = int(np.floor(self.t / period + 1e-9)) + 1
                cands.append(k * period)
            if log_dt:
                cands.append((int(np.floor(self.t / log_dt + 1e-9)) + 1) * log_dt)
            t_next = min(cands)
            at_force = any(abs(self.t - x) < MIN_DT for x in force_pts)
            at_ctrl = period > 0.0 and abs(self.t / period - round(self.t / period)) < 1e-9
            self._set_wrench(self.t)
            changed = bool(at_force or at_ctrl)
            if solver == "runge_kutta_dopri":
                self.ad["t"][:] = self.t
                self.e.batch_run_dopri(self.arr, self.ad, t_next, tol_rel=tol_rel, tol_abs=tol_abs, dt_max=dt_max,
                                       new_step=True, command_changed=changed, update_sensors=True)
            else:
                # the reference's own sub-step rule, opening microsecond step included (engine.cc:2021-2222)
                self.loop.dt_max = float(dt_max)
                self.loop.advance(lambda dt, first: self.e.batch_run("step", self.io, solver=solver, dt=dt, n_substeps=1,
                                                                     command_changed=first), t_next - self.t, changed)
            self.t = t_next
            log.append(self.row())
        out = {k: np.stack([r[k] for r in log]) for k in log[0] if k != "t"}
        out["t"] = np.array([r["t"] for r in log])
        out["status"] = int(self.arr["status"][0, 0])
        return out
