"""Shared helpers of the test-suite: SoA buffers, oracle batch runs, error metrics."""
from __future__ import annotations

from typing import Dict

import numpy as np

from jiminy_amd import _abi
from jiminy_amd.model import CompiledModel
from oracle.oracle_py import OracleEngine

ORACLE_FIELDS = ("q", "v", "a", "command", "u_motor", "imu", "force", "contact", "encoder",
                 "effort", "energy", "contact_forces", "f_external", "joint_forces", "centroidal",
                 "u")


def alloc_soa(model: CompiledModel, B: int, dtype=np.float64) -> Dict[str, np.ndarray]:
    rows = _abi.field_rows(model)
    arr = {k: np.zeros((max(n, 1), B), dtype=dtype) for k, n in rows.items() if k != "status"}
    arr["status"] = np.zeros((1, B), dtype=np.int32)
    return arr


def oracle_io(arr: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    io = {k: arr[k] for k in ORACLE_FIELDS}
    io["status"] = arr["status"].reshape(-1)
    return io


def oracle_batch(model: CompiledModel, arr: Dict[str, np.ndarray], mode: str, options=None,
                 constraint_options=None, **kw) -> None:
    e = OracleEngine(model, **(options or {}))
    if constraint_options is not None:
        e.set_constraint_options(**constraint_options)
        e.bind_constraints(arr["con_flags"], arr["con_data"])
    if arr.get("friction") is not None:     # per-lane contacts.friction, either contact model
        e.bind_friction(arr["friction"])
    if arr.get("flexibility") is not None:  # per-lane flexibilityConfig (stiffness 3, damping 3 per spherical joint)
        e.bind_flexibility(arr["flexibility"])
    e.batch_run(mode, oracle_io(arr), **kw)


def alloc_constraint_state(model: CompiledModel, arr: Dict[str, np.ndarray], B: int, dtype=np.float64) -> None:
    """Add the per-lane constraint state rows (`contacts.model = "constraint"`) to an SoA dict."""
    rows = _abi.constraint_rows(model)
    arr["con_flags"] = np.zeros((max(rows["con_flags"], 1), B), dtype=np.int32)
    arr["con_data"] = np.zeros((max(rows["con_data"], 1), B), dtype=dtype)


def rel_err(x: np.ndarray, ref: np.ndarray, lanes=None) -> float:
    """max |x - ref|_inf / max(|ref|_inf, 1) per lane, worst lane (SURVEY.md 8d metric)."""
    if lanes is not None:
        x, ref = x[:, lanes], ref[:, lanes]
    if x.size == 0:
        return 0.0
    num = np.abs(x - ref).max(axis=0)
    den = np.maximum(np.abs(ref).max(axis=0), 1.0)
    return float((num / den).max())


SIMULATION_MIN_TIMESTEP, STEPPER_MIN_TIMESTEP = 1e-6, 1e-10     # reference constants.h:18-20


class ReferenceFixedStepLoop:
    """The reference's inner integration loop (engine.cc:2021-2222) for a fixed-step stepper, around the oracle: the
    test-side statement of WHICH sub-steps `Engine::step` takes between two breakpoints, written from the reference and
    independent of `jiminy_amd.engine.substep_sizes`.

    `self.dt` is `stepperState_.dt`: `SIMULATION_MIN_TIMESTEP` after `Engine::start` (engine.cc:1176: every
    simulation opens with one microsecond step), `min(dtLargest = INF, dtMax)` after every try (:2220)."""

    def __init__(self, dt_max: float) -> None:
        self.dt_max = float(dt_max)
        self.dt = SIMULATION_MIN_TIMESTEP

    def sizes(self, interval: float):
        """Sub-step sizes up to the next breakpoint, `interval` seconds away."""
        left = float(interval)
        while left > STEPPER_MIN_TIMESTEP:
            dt = self.dt
            residual_thr = min(max(0.1 * dt, STEPPER_MIN_TIMESTEP), SIMULATION_MIN_TIMESTEP)    # :2063-2068
            if left < dt or left < dt + residual_thr:                                           # :2069-2073
                dt = left
            if dt > SIMULATION_MIN_TIMESTEP:                                                    # :2080-2089
                r = float(np.fmod(dt, SIMULATION_MIN_TIMESTEP))
                if STEPPER_MIN_TIMESTEP < r < SIMULATION_MIN_TIMESTEP - STEPPER_MIN_TIMESTEP and dt - r > STEPPER_MIN_TIMESTEP:
                    dt -= r
            yield dt
            left -= dt
            self.dt = self.dt_max                                                               # :2220

    def advance(self, run_step, interval: float, command_changed: bool = False) -> int:
        """`run_step(dt, command_changed)` once per sub-step of the interval; returns their number."""
        n = 0
        for dt in self.sizes(interval):
            run_step(dt, command_changed and n == 0)
            n += 1
        return n


def oracle_engine_step(model: CompiledModel, arr: Dict[str, np.ndarray], loop: ReferenceFixedStepLoop, interval: float,
                       solver: str, command_changed: bool = False, options=None, constraint_options=None, **kw) -> int:
    """One breakpoint interval of `Engine::step` on the oracle: the sub-steps of `loop` (opening microsecond step
    included), `command_changed` (the a(t+) refresh, engine.cc:2030-2042) on the first one only."""
    return loop.advance(lambda dt, changed: oracle_batch(model, arr, "step", options=options,
                                                         constraint_options=constraint_options, solver=solver, dt=dt,
                                                         n_substeps=1, command_changed=changed, **kw),
                        interval, command_changed)
