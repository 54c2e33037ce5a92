"""BatchedEngine: the `jiminy.Engine` surface for one robot model x B independent lanes.

Mirrors, for the hot path only, the Python-visible API of the reference engine
(python/jiminy_pywrap/src/engine.cc:587-786): `set_options/get_options`, `start`, `step`,
`stop`, `reset`, `compute_robots_dynamics`, `robot_state`, `stepper_state`,
`is_simulation_running`, plus `sensor_measurements` (the `SensorMeasurementTree` seen by
controllers).  Every array has a trailing batch axis in storage (`[rows][B]`, lane contiguous) and
is a `torch.Tensor` on the HIP device, so that learners consume it without a host round trip.

All physics runs in the hand-written HIP kernels behind include/jiminy_hip.h; there is no CPU
fallback: constructing an engine without a HIP device raises.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import warnings
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _abi, codegen
from ._lib import BadControlFlow, HipLibrary, load_for
from .model import (JT_FREEFLYER, JT_RUBU, JT_RUBX, JT_RUBY, JT_RUBZ, CompiledModel)

# constants of reference core/include/jiminy/core/constants.h:18-20
STEPPER_MIN_TIMESTEP = 1e-10
INIT_ITERATIONS = 4   # engine.cc:61
SIMULATION_MIN_TIMESTEP = 1e-6
SIMULATION_MAX_TIMESTEP = 0.02
EPS = float(np.finfo(np.float64).eps)

SOLVER_IDS = {"euler_explicit": _abi.JM_SOLVER_EULER_EXPLICIT,
              "runge_kutta_4": _abi.JM_SOLVER_RUNGE_KUTTA_4,
              "runge_kutta_dopri": _abi.JM_SOLVER_RUNGE_KUTTA_DOPRI}


def default_options(dtype: torch.dtype = torch.float64) -> Dict[str, Dict[str, Any]]:
    """Hot-path subset of `Engine::getDefaultEngineOptions` (reference engine.h:260-481), same names
    and defaults: `odeSolver` = "runge_kutta_dopri" (adaptive, every lane carries its own step size) and
    `contacts.model` = "constraint" (engine.h:273).  The constraint solver is float64 only (the reference
    has no float32 mode to mirror): a float32 engine starts with "spring_damper"."""
    return {
        "world": {"gravity": [0.0, 0.0, -9.81, 0.0, 0.0, 0.0]},
        "stepper": {"odeSolver": "runge_kutta_dopri", "tolAbs": 1.0e-5, "tolRel": 1.0e-4,
                    "dtMax": SIMULATION_MAX_TIMESTEP, "dtRestoreThresholdRel": 0.2,
                    "successiveIterFailedMax": 1000, "iterMax": 0,
                    "controllerUpdatePeriod": 0.0, "sensorsUpdatePeriod": 0.0},
        "contacts": {"model": "constraint" if dtype == torch.float64 else "spring_damper",
                     "stiffness": 1.0e6, "damping": 2.0e3,
                     "friction": 1.0, "torsion": 0.0, "transitionEps": 1.0e-3,
                     "transitionVelocity": 1.0e-2, "stabilizationFreq": 20.0},
        "constraints": {"solver": "PGS", "regularization": 1.0e-3},
    }


def _breakpoint_intervals(t: float, t_error: float, step_size: float, options: Dict[str, Dict[str, Any]],
                          extra_breakpoints: Tuple[float, ...] = ()
                          ) -> Tuple[List[Tuple[float, float, bool, bool]], float, float]:
    """Breakpoint schedule of one `Engine::step(step_size)` call.

    Re-states the breakpoint logic of the reference's discrete branch (engine.cc:1755-1795 end
    time with Kahan compensation; :1920-1940 controller breakpoints; :1985-2016 time to the next
    breakpoint; :2386-2410 sensor breakpoints).  Returns (intervals, t_end, t_error) with one
    `(t_next, dt_next, command_changed, update_sensors)` per interval between two breakpoints.
    """
    st = options["stepper"]
    dt_max = float(st["dtMax"])
    ctrl = float(st["controllerUpdatePeriod"])
    sens = float(st["sensorsUpdatePeriod"])
    if step_size < EPS:
        # default step size: controller period, else sensor period, else dtMax (engine.cc:1758-1778)
        step_size = ctrl if ctrl > EPS else (sens if sens > EPS else dt_max)
    if step_size < SIMULATION_MIN_TIMESTEP:
        raise ValueError("Step size out of bounds.")
    corrected = step_size - t_error
    t_end = t + corrected
    t_error = (t_end - t) - corrected
    periods = [p for p in (ctrl, sens) if p > EPS]
    update_period = min(periods) if periods else math.inf

    def hit(period: float, time: float) -> bool:
        return period < EPS or update_due(period, time)

    intervals: List[Tuple[float, float, bool, bool]] = []
    while t_end - t >= STEPPER_MIN_TIMESTEP:
        command_changed = hit(ctrl, t)
        if math.isfinite(update_period):
            nxt = update_period - math.fmod(t, update_period)
            if nxt < SIMULATION_MIN_TIMESTEP:
                nxt += update_period
            dt_next = nxt
            if t_end - t - STEPPER_MIN_TIMESTEP < dt_next:
                dt_next = t_end - t
        else:
            dt_next = t_end - t
        # start / end of the impulse forces and refresh times of the profile forces (engine.cc:1985-2016)
        for tb in extra_breakpoints:
            if STEPPER_MIN_TIMESTEP <= tb - t < dt_next - STEPPER_MIN_TIMESTEP:
                dt_next = tb - t
        t = t + dt_next
        intervals.append((t, dt_next, command_changed, hit(sens, t)))
    return intervals, t_end, t_error


def plan_breakpoints(t: float, t_error: float, step_size: float, options: Dict[str, Dict[str, Any]],
                     extra_breakpoints: Tuple[float, ...] = ()) -> Tuple[List[Tuple[float, bool, bool]], float, float]:
    """Adaptive solver: `(t_next, command_changed, update_sensors)` per breakpoint interval; the
    step sizes inside an interval are chosen per lane on the device (engine.cc:2021-2222)."""
    intervals, t_end, t_err = _breakpoint_intervals(t, t_error, step_size, options, extra_breakpoints)
    return [(tn, c, s) for tn, _, c, s in intervals], t_end, t_err


def update_due(period: float, t: float) -> bool:
    """Is a periodic update (controller command engine.cc:1923-1927, profile force :1903-1907, sensors :2386-2410) due at time
    `t`: the time to the next multiple of the period is below a microsecond (the update is taken early and the multiple
    skipped) or the last multiple is less than STEPPER_MIN_TIMESTEP behind.  Pinned to the reference's compiled lines
    (`tests/golden/ref_cpp_leaves.npz`, `update_*`)."""
    dt_next = period - math.fmod(t, period)
    return dt_next < SIMULATION_MIN_TIMESTEP or period - dt_next < STEPPER_MIN_TIMESTEP


def impulse_active(t_impulse: float, dt_impulse: float, t: float, was_active: bool = False) -> bool:
    """≙ the active-set update of an impulse force (engine.cc:1857-1869): switched on once `t > t_impulse - 1e-10`, off once
    `t >= t_impulse + dt_impulse - 1e-10` (both tests run, in that order, at every breakpoint)."""
    active = was_active
    if t > t_impulse - STEPPER_MIN_TIMESTEP:
        active = True
    if t >= t_impulse + dt_impulse - STEPPER_MIN_TIMESTEP:
        active = False
    return active


def min_clipped(*values: float) -> float:
    """≙ `minClipped` (utilities/helpers.hxx:59-92): the smallest of the values that exceed EPS, INF when there is none."""
    valid = [float(v) for v in values if float(v) > EPS]
    return min(valid) if valid else math.inf


def is_gcd_included(*values: float) -> Tuple[bool, float]:
    """≙ `isGcdIncluded(values...)` (utilities/helpers.hxx:94-116): `(every value is a multiple of the smallest positive one,
    that smallest one)`, where "multiple" is the reference's test `fmod(value, min) < STEPPER_MIN_TIMESTEP` -- which refuses
    pairs like (0.03, 0.01) or (0.009, 0.003), whose floating-point remainder is the divisor minus one ulp, and accepts a period
    of zero (`fmod(0, min) = 0`: continuous mode).  Kept bit for bit (pinned by `tests/golden/ref_cpp_leaves.npz`, `period_*`):
    the drop-in raises for the option values the reference raises for."""
    value_min = min_clipped(*values)
    if not math.isfinite(value_min):
        return True, math.inf
    return all(math.fmod(float(v), value_min) < STEPPER_MIN_TIMESTEP for v in values), value_min


def substep_sizes(dt_next: float, dt_max: float, dt_first: Optional[float] = None) -> List[float]:
    """Integrator step sizes of a fixed-step solver over one breakpoint interval of length `dt_next`: the inner loop
    of `Engine::step` (engine.cc:2021-2222) for a stepper whose `tryStep` always succeeds and returns `dtLargest = INF`
    (euler_explicit_stepper.cc:19, abstract_runge_kutta_stepper.cc:74-76), i.e. `dt = min(INF, dtMax)` after every try:

      * the step is stretched to land exactly on the breakpoint when what would be left after it is below
        `clamp(0.1 dt, STEPPER_MIN_TIMESTEP, SIMULATION_MIN_TIMESTEP)` -- a residual of less than a microsecond is
        merged into the last step instead of being integrated on its own (:2063-2073);
      * a step longer than a microsecond that is not a whole number of microseconds is shortened to one (:2080-2089),
        so a `dtMax` that is not a multiple of 1 us advances in microsecond multiples and the last step takes the rest.

    `dt_first`: the step size the stepper state carries INTO the interval when it is not `dtMax`.  `Engine::start` resets
    the stepper state with `dt = SIMULATION_MIN_TIMESTEP` (`stepperState_.reset(SIMULATION_MIN_TIMESTEP, ...)`,
    engine.cc:1176), and the loop only settles on `dtMax` AFTER its first try (:2220): every simulation opens with one
    step of a microsecond, so the first interval of a simulation with `dt_next <= dtMax` is integrated as
    (1e-6, dt_next - 1e-6).  The same two rules apply to that first size (1e-6 is neither stretched, unless the whole
    interval is shorter than 1.1 us, nor snapped)."""
    sizes: List[float] = []
    t, t_next = 0.0, float(dt_next)
    dt_state = float(dt_first) if dt_first is not None else float(dt_max)
    while t_next - t > STEPPER_MIN_TIMESTEP:
        dt = dt_state
        thr = min(max(0.1 * dt, STEPPER_MIN_TIMESTEP), SIMULATION_MIN_TIMESTEP)
        if t_next - t < dt + thr:
            dt = t_next - t
        if dt > SIMULATION_MIN_TIMESTEP:
            res = math.fmod(dt, SIMULATION_MIN_TIMESTEP)
            if STEPPER_MIN_TIMESTEP < res < SIMULATION_MIN_TIMESTEP - STEPPER_MIN_TIMESTEP and dt - res > STEPPER_MIN_TIMESTEP:
                dt -= res
        sizes.append(dt)
        t += dt
        dt_state = float(dt_max)      # `dt = min(dtLargest = INF, dtMax)` (engine.cc:2220)
    if not sizes:
        sizes.append(float(dt_next))
    return sizes


def plan_step(t: float, t_error: float, step_size: float, options: Dict[str, Dict[str, Any]],
              extra_breakpoints: Tuple[float, ...] = (), dt_first: Optional[float] = None
              ) -> Tuple[List[Tuple[float, int, bool, bool]], float, float]:
    """Fixed-step schedule of one `Engine::step(step_size)` call: the breakpoint intervals cut in sub-steps of
    `dtMax` by the reference's rule (`substep_sizes`: residual merge and microsecond snapping, engine.cc:2063-2089).
    Returns (launches, t_end, t_error) where each launch is `(dt, n_substeps, command_changed, update_sensors)`:
    consecutive sub-steps of the same size share a launch.  `dt_first` = the step size carried into the FIRST interval
    (`SIMULATION_MIN_TIMESTEP` for the first step after `start`, engine.cc:1176)."""
    dt_max = float(options["stepper"]["dtMax"])
    intervals, t_end, t_error = _breakpoint_intervals(t, t_error, step_size, options, extra_breakpoints)
    launches: List[Tuple[float, int, bool, bool]] = []
    for k, (_, dt_next, command_changed, update_sensors) in enumerate(intervals):
        groups: List[List[Any]] = []
        for dt in substep_sizes(dt_next, dt_max, dt_first if k == 0 else None):
            # (steps that differ by the round-off of the time accumulation -- below 1e-14 s -- share a launch)
            if groups and abs(groups[-1][0] - dt) <= 1.0e-14:
                groups[-1][1] += 1
            else:
                groups.append([dt, 1])
        # sensors with an update period are refreshed at the END of the interval (a breakpoint of theirs); continuous
        # sensors (sensorsUpdatePeriod = 0) after EVERY integrator step (engine.cc:2386-2410): every launch of the
        # interval refreshes them (what a later launch overwrites still draws from the noise streams)
        continuous_sensors = float(options["stepper"]["sensorsUpdatePeriod"]) < EPS
        for i, (dt, n) in enumerate(groups):
            launches.append((dt, n, command_changed and i == 0,
                             update_sensors and (continuous_sensors or i == len(groups) - 1)))
    return launches, t_end, t_error


@dataclass
class RobotState:
    """≙ `struct RobotState` (reference engine.h:134-156), each tensor `[rows][B]`."""
    q: torch.Tensor
    v: torch.Tensor
    a: torch.Tensor
    command: torch.Tensor
    u: torch.Tensor
    u_motor: torch.Tensor
    f_external: Optional[torch.Tensor]


@dataclass
class JointConstraint:
    """≙ `jiminy.JointConstraint(joint_name)` (core/src/constraints/joint_constraint.cc): the joint held at a reference
    configuration (the configuration at `start`, `JointConstraint::reset`) by a bilateral kinematic constraint with the
    Baumgarte stabilisation of the contact model.  Registered with `BatchedEngine.add_constraint`.

    `baumgarte_freq` ≙ `AbstractConstraintBase::setBaumgarteFreq` (abstract_constraint.cc:88-98): a user constraint keeps gains
    of its own -- `Engine::start` only overwrites those of the internal constraints (engine.cc:1276-1285) -- critically
    damped for this frequency; 0.0 = the reference's state of a freshly created constraint (a pure acceleration
    constraint).  `None` (the default HERE, kept from round 3) shares the gains of `contacts.stabilizationFreq`.  The user
    constraints of one batch share one value (`jm_constraint_options::user_stabilization_freq`, ABI 6)."""
    joint_name: str
    baumgarte_freq: Optional[float] = None


@dataclass
class FrameConstraint:
    """≙ `jiminy.FrameConstraint(frame_name, mask_dofs)` (core/src/constraints/frame_constraint.cc): the frame held at a
    reference pose -- its pose at `start` (`FrameConstraint::reset`), or the one given to `set_constraint_reference` --
    along the masked dofs (x, y, z, rot x, rot y, rot z; world aligned), with Baumgarte stabilisation on the position error
    and on `log3(R R_ref^T)`.  The frame and mask must have been declared on the model (`model.add_frame_constraint`: the
    kernels are specialised on them like on the contact points); `BatchedEngine.add_constraint` registers it for (some of)
    the lanes.  `baumgarte_freq` like `JointConstraint`; the default 0.0 is the reference's freshly created constraint
    (an acceleration-level constraint without position feedback)."""
    frame_name: str
    mask_dofs: Tuple[bool, ...] = (True, True, True, True, True, True)
    baumgarte_freq: Optional[float] = 0.0

    @property
    def mask(self) -> int:
        return int(sum(1 << d for d in range(6) if self.mask_dofs[d]))


@dataclass
class SphereConstraint:
    """≙ `jiminy.SphereConstraint(frame_name, sphere_radius, ground_normal)` (core/src/constraints/sphere_constraint.cc):
    declared with `model.add_sphere_constraint`; reference = the frame's pose at `start` (its height along the normal is
    what the Baumgarte term holds)."""
    frame_name: str
    radius: float
    ground_normal: Tuple[float, float, float] = (0.0, 0.0, 1.0)
    baumgarte_freq: Optional[float] = 0.0
    kind = "sphere"


@dataclass
class WheelConstraint:
    """≙ `jiminy.WheelConstraint(frame_name, wheel_radius, ground_normal, wheel_axis)` (core/src/constraints/wheel_constraint.cc);
    declared with `model.add_wheel_constraint`."""
    frame_name: str
    radius: float
    ground_normal: Tuple[float, float, float] = (0.0, 0.0, 1.0)
    wheel_axis: Tuple[float, float, float] = (0.0, 0.0, 1.0)
    baumgarte_freq: Optional[float] = 0.0
    kind = "wheel"


@dataclass
class DistanceConstraint:
    """≙ `jiminy.DistanceConstraint(first_frame_name, second_frame_name)` (core/src/constraints/distance_constraint.cc);
    declared with `model.add_distance_constraint`; reference = the distance at `start` (`reference_distance`)."""
    first_frame_name: str
    second_frame_name: str
    baumgarte_freq: Optional[float] = 0.0
    kind = "distance"


@dataclass
class StepperState:
    """≙ `struct StepperState` (reference engine.h:216-250); time is shared by all lanes."""
    iter: int
    iter_failed: int
    t: float
    t_prev: float
    t_error: float
    dt: float
    q: torch.Tensor
    v: torch.Tensor
    a: torch.Tensor
    # adaptive solver only: per-lane step size / estimated largest step size / counters, shape (B,)
    dt_lanes: Optional[torch.Tensor] = None
    dt_largest: Optional[torch.Tensor] = None
    iter_lanes: Optional[torch.Tensor] = None
    iter_failed_lanes: Optional[torch.Tensor] = None


def _exp3(w: np.ndarray) -> np.ndarray:
    """Rodrigues' formula with pinocchio::exp3's small-angle series (explog.hpp, v2.7.0)."""
    t2 = float(w @ w)
    t = math.sqrt(t2)
    K = np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])
    if t < 1e-8:   # TaylorSeriesExpansion precision
        a, b = 1.0 - t2 / 6.0, 0.5 - t2 / 24.0
    else:
        a, b = math.sin(t) / t, (1.0 - math.cos(t)) / t2
    return np.eye(3) + a * K + b * (K @ K)


_VERIFIED: Dict[Tuple[str, torch.dtype], int] = {}


def _probe_state(model: CompiledModel, n: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Deterministic, generic state for the library self-test: joints spread around the neutral
    configuration inside their bounds, a tilted base some decimetres above the ground (part of the
    contact points end up below it), moderate velocities and commands."""
    rng = np.random.default_rng(20240917)
    q = np.tile(model.neutral()[:, None], (1, n))
    lo, hi = model.position_lower, model.position_upper
    for j in range(1, model.njoints):
        t, iq = int(model.jtypes[j]), int(model.idx_q[j])
        if t == JT_FREEFLYER:
            q[iq:iq + 2] = rng.uniform(-0.2, 0.2, (2, n))
            q[iq + 2] = rng.uniform(0.2, 0.7, n)
            quat = np.concatenate([0.2 * rng.standard_normal((3, n)), np.ones((1, n))])
            q[iq + 3:iq + 7] = quat / np.linalg.norm(quat, axis=0)
        elif t == 14:   # spherical (flexibility) joint: a small deflection (its inertia may be 1e-5 kg m^2: the float32
            # instantiations of the self-tests must stay meaningful on it)
            quat = np.concatenate([5e-4 * rng.standard_normal((3, n)), np.ones((1, n))])
            q[iq:iq + 4] = quat / np.linalg.norm(quat, axis=0)
        elif t in (JT_RUBX, JT_RUBY, JT_RUBZ, JT_RUBU):
            th = rng.uniform(-1.0, 1.0, n)
            q[iq], q[iq + 1] = np.cos(th), np.sin(th)
        else:
            l = lo[iq] if np.isfinite(lo[iq]) else -1.0e9
            h = hi[iq] if np.isfinite(hi[iq]) else 1.0e9
            c = min(max(0.0, l), h)
            q[iq] = np.clip(c + rng.uniform(-0.5, 0.5, n), l, h)
    v = 0.5 * rng.standard_normal((model.nv, n))
    for j in range(1, model.njoints):
        if int(model.jtypes[j]) == 14:
            v[int(model.idx_v[j]):int(model.idx_v[j]) + 3] *= 0.02
    scale = np.array([min(m.effort_limit, 20.0) for m in model.motors]).reshape(-1, 1)
    cmd = rng.uniform(-1.0, 1.0, (model.nmotors, n)) * scale if model.nmotors else np.zeros((0, n))
    return q, v, cmd


def _library_self_test(model: CompiledModel, variant: int, dtype: torch.dtype, device: torch.device) -> float:
    """Consistency check of one compiled library.  The evaluation loop of the step kernel and its
    peeled last evaluation are two separately optimised copies of the same code: an explicit-Euler
    step with and without the `a(t+)` refresh (command unchanged) must agree to round-off -- the
    first takes the acceleration from the in-loop copy, the second from `start`; a second step
    (one more pass through the loop, from a state the kernel itself produced) must agree too.
    Returns the largest relative disagreement over (v, a)."""
    err, _ = _library_self_test_detail(model, variant, dtype, device)
    return err


def _library_self_test_detail(model: CompiledModel, variant: int, dtype: torch.dtype,
                              device: torch.device) -> Tuple[float, Dict[str, float]]:
    """`_library_self_test` with the disagreement of every leg (solver x launch form)."""
    n0, dt = 64, 1e-4
    probe_state = _probe_state(model, n0)
    # the branch-parallel step kernel has a one-wave-per-block instantiation for small batches and a
    # several-waves-per-block one for large batches (jm_lib.cpp `launch_quad`): both are probed, and must
    # agree with each other lane for lane
    sizes = [n0]
    if codegen.quad_structure(model) is not None:
        cus = torch.cuda.get_device_properties(device).multi_processor_count
        sizes.append(2 * cus * 16 * 4)
    err, legs = 0.0, {}

    def rel(x, y, ok):
        scale = torch.clamp(x[:, ok].abs().max(), min=1.0)
        e = float(((x - y)[:, ok]).abs().max() / scale)
        return e if e == e else float("inf")
    # the explicit-Euler and the Runge-Kutta kernels are separate instantiations: both are checked (the
    # Runge-Kutta one takes k1 from the last stage evaluation of the previous step, or from the refresh)
    for solver in ("euler_explicit", "runge_kutta_4"):
        small = None
        for n in sizes:
            q, v, cmd = (torch.as_tensor(np.tile(x, (1, n // n0)), dtype=dtype, device=device) for x in probe_state)
            outs = []
            for changed in (False, True):
                probe = BatchedEngine(model, n, dtype=dtype, device=device, extra_outputs=(), _lib_variant=variant)
                probe.set_options({"stepper": {"odeSolver": solver, "dtMax": dt,
                                               "controllerUpdatePeriod": 0.0, "sensorsUpdatePeriod": 0.0},
                                   "contacts": {"model": "spring_damper"}})
                if model.nmotors:
                    probe.set_command(cmd)
                probe.start(q, v)
                res = []
                for _ in range(2):
                    if changed:
                        probe.mark_command_changed()
                    probe.step(dt)
                    res += [probe._fields["v"].clone(), probe._fields["a"].clone()]
                outs.append((res, probe.status.clone()))
                probe.stop()
            ok = ((outs[0][1] | outs[1][1]) & _abi.JM_LANE_NAN).reshape(-1) == 0
            if not bool(ok.any()):
                legs[f"{solver}/{n}"] = float("inf")   # a state this tame never produces NaN in a sound library
                err = float("inf")
                continue
            e = max(rel(x, y, ok) for x, y in zip(outs[0][0], outs[1][0]))
            if dtype == torch.float32 and solver == "runge_kutta_4":
                # float32: two Runge-Kutta steps through the probe's stiff ground contacts amplify round-off by ~1e6
                # (1.6e-10 in float64), the refresh comparison is meaningless there; the launch-form comparison below
                # (same arithmetic, bit-identical when sound) and the Euler legs remain
                e = 0.0
            if small is None:
                small = (outs[0][0], ok)
            else:   # large-batch launch form against the small-batch one, on the lanes they share
                ok0 = small[1] & ok[:n0]
                e = max([e] + [rel(x, y[:, :n0], ok0) for x, y in zip(small[0], outs[0][0])]) if bool(ok0.any()) else float("inf")
            legs[f"{solver}/{n}"] = e
            err = max(err, e)
    return err, legs


_OUTPUT_ROWS = ("a", "u", "imu", "force", "contact", "encoder", "effort", "contact_forces", "f_external", "energy", "joint_forces",
                "centroidal")


def _output_self_test(model: CompiledModel, variant: int, device: torch.device) -> float:
    """The rows a launch EMITS (sensors, extra terms), which the step self-test does not look at: the float64 and the
    float32 instantiations of the kernels are separate compilations of the same sources, so a mis-compiled output pass
    (third case: a wrong IMU row next to a right `q`, `v`, `a`) shows as a disagreement between
    them.  Probe batch, spring-damper contacts, after `start` and after one explicit-Euler step; returns the largest
    disagreement over the emitted rows, relative to the row's scale + 1e-3 of its field's (sound builds: <= 2e-3, float32 round-off)."""
    n, dt = 64, 1e-4
    q, v, cmd = _probe_state(model, n)
    runs = []
    for dtype in (torch.float64, torch.float32):
        probe = BatchedEngine(model, n, dtype=dtype, device=device, _lib_variant=variant,
                              extra_outputs=("contact_forces", "f_external", "energy", "joint_forces", "centroidal"))
        probe.set_options({"stepper": {"odeSolver": "euler_explicit", "dtMax": dt, "controllerUpdatePeriod": 0.0,
                                       "sensorsUpdatePeriod": 0.0}, "contacts": {"model": "spring_damper"}})
        if model.nmotors:
            probe.set_command(torch.as_tensor(cmd, dtype=dtype, device=device))
        probe.start(torch.as_tensor(q, dtype=dtype, device=device), torch.as_tensor(v, dtype=dtype, device=device))
        rows = [{k: probe._fields[k].double().clone() for k in _OUTPUT_ROWS if k in probe._fields}]
        probe.step(dt)
        rows.append({k: probe._fields[k].double().clone() for k in _OUTPUT_ROWS if k in probe._fields})
        runs.append((rows, probe.status.reshape(-1).clone()))
        probe.stop()
    ok = ((runs[0][1] | runs[1][1]) & _abi.JM_LANE_NAN) == 0
    # (the float64 rows are kept: `_verified_library` compares them ACROSS build variants when float32 cannot serve as the
    # second opinion -- a robot float32 is too coarse for, e.g. a flexibility inertia of 1e-5 next to bodies of 5 kg m^2)
    _OUTPUT_ROWS_F64[(model.topology_hash(), variant)] = (runs[0][0], (runs[0][1] & _abi.JM_LANE_NAN) == 0)
    if not bool(ok.any()):
        return float("inf")
    err = 0.0
    for r64, r32 in zip(runs[0][0], runs[1][0]):
        for k, x in r64.items():
            if x.numel() == 0:
                continue
            y = r32[k]
            # scale of a row + a share of the scale of its field: a row that is zero by cancellation (the joint wrench of a
            # free-flyer: terms of 1e7 N on the probe batch) carries the float32 round-off of the terms, not of the row
            scale = torch.clamp(x[:, ok].abs().amax(dim=1, keepdim=True), min=1.0) + 1e-3 * x[:, ok].abs().max()
            e = float((((x - y)[:, ok]).abs() / scale).max())
            err = max(err, e if e == e else float("inf"))
    return err


_OUTPUT_ROWS_F64: Dict[Tuple[str, int], Any] = {}


def _float64_rows_agree_across_variants(model: CompiledModel, variants: List[int]) -> float:
    """Largest relative disagreement of the float64 emitted rows between separately compiled build variants of one topology
    (rows kept by `_output_self_test`): three compilations with different register allocators / optimisation levels that
    agree to round-off are not three identical mis-compiles."""
    key = model.topology_hash()
    rows0, ok0 = _OUTPUT_ROWS_F64[(key, variants[0])]
    err = 0.0
    for v in variants[1:]:
        rows, ok = _OUTPUT_ROWS_F64[(key, v)]
        lanes = ok0 & ok
        if not bool(lanes.any()):
            return float("inf")
        for a, b in zip(rows0, rows):
            for k, x in a.items():
                if x.numel() == 0:
                    continue
                scale = torch.clamp(x[:, lanes].abs().amax(dim=1, keepdim=True), min=1.0) + 1e-3 * x[:, lanes].abs().max()
                e = float((((x - b[k])[:, lanes]).abs() / scale).max())
                err = max(err, e if e == e else float("inf"))
    return err


_GEN_VERIFIED: Dict[Tuple[str, int, str], float] = {}


def _variation_self_test(model: CompiledModel, variant: int, device: torch.device, contact_model: str) -> float:
    """Consistency check of the per-environment variation kernels (`k_quad_gen`, `k_quad_con_gen`: their own
    instantiations of the evaluation) of one compiled library: with the NOMINAL body
    parameters bound per lane, a flat height map and zero applied wrenches they must reproduce the plain kernels
    to round-off over two Runge-Kutta steps.  Returns the largest relative disagreement over (v, a)."""
    from .randomization import nominal_model_lane
    n, dt = 64, 1e-4
    q, v, cmd = (torch.as_tensor(x, dtype=torch.float64, device=device) for x in _probe_state(model, n))
    err = 0.0
    # (the persistent adaptive stepper has its own variation instantiation, `k_quad_dopri_gen`: spring-damper model)
    for solver in ("runge_kutta_4",) + (("runge_kutta_dopri",) if contact_model == "spring_damper" else ()):
        e = _variation_self_test_leg(model, variant, device, contact_model, solver, n, dt, q, v, cmd)
        # (two adaptive runs whose evaluations round differently agree to the integration tolerance, 1e-7 here, not to
        # round-off: that leg is weighed so that the common bar of 1e-8 means 1e-5 for it; garbage is O(1))
        err = max(err, e * (1e-3 if solver == "runge_kutta_dopri" else 1.0))
    return err


def _variation_self_test_leg(model: CompiledModel, variant: int, device: torch.device, contact_model: str, solver: str,
                             n: int, dt: float, q: torch.Tensor, v: torch.Tensor, cmd: torch.Tensor) -> float:
    from .randomization import nominal_model_lane
    outs = []
    for gen in (False, True):
        probe = BatchedEngine(model, n, dtype=torch.float64, device=device, extra_outputs=(), _lib_variant=variant)
        probe.set_options({"stepper": {"odeSolver": solver, "dtMax": dt if solver != "runge_kutta_dopri" else 1e-3,
                                       "controllerUpdatePeriod": dt, "sensorsUpdatePeriod": dt, "tolAbs": 1e-8, "tolRel": 1e-7},
                           "contacts": {"model": contact_model}},
                          _skip_constraint_check=True)
        if gen:
            probe._gen_checked = True
            probe.set_lane_model(nominal_model_lane(model, n, torch.float64, device))
            if contact_model == "spring_damper":
                probe.set_ground_heightmap(np.zeros((3, 3)), -1.0, -1.0, 1.0, 1.0)
            root = [name for name, f in model.frames.items() if f.parent_joint == 1] if model.has_freeflyer else []
            if root:
                probe.register_impulse_force(root[0], 0.0, 1.0, np.zeros(6))
        if model.nmotors:
            probe.set_command(cmd)
        probe.start(q, v)
        res = []
        for _ in range(2):
            probe.step(dt)
            res += [probe._fields["v"].clone(), probe._fields["a"].clone()]
        outs.append((res, probe.status.clone()))
        probe.stop()
    ok = ((outs[0][1] | outs[1][1]) & _abi.JM_LANE_NAN).reshape(-1) == 0
    if not bool(ok.any()):
        return float("inf")
    err = 0.0
    for x, y in zip(outs[0][0], outs[1][0]):
        scale = torch.clamp(x[:, ok].abs().max(), min=1.0)
        e = float(((x - y)[:, ok]).abs().max() / scale)
        err = max(err, e if e == e else float("inf"))
    return err


_DOPRI_FORM: Dict[Tuple[str, int, bool], int] = {}


def _adaptive_self_test(model: CompiledModel, variant: int, device: torch.device, gen: bool = False) -> float:
    """Persistent adaptive kernel (jm_qdopri.h) against the per-stage launches on the probe batch: three breakpoint
    intervals with tight tolerances.  Returns the largest relative disagreement over (q, v) on the lanes that follow the
    same accept / reject sequence (inf when fewer than 80 % do, or when the persistent kernel flags lanes the per-stage
    path does not).  `gen`: its variation form (`k_quad_dopri_gen`, a separate compilation), selected by a per-lane friction
    field that holds the nominal coefficient."""
    n = 64
    q, v, cmd = (torch.as_tensor(x, dtype=torch.float64, device=device) for x in _probe_state(model, n))
    outs = []
    for form in (1, 0):
        probe = BatchedEngine(model, n, dtype=torch.float64, device=device, extra_outputs=(), _lib_variant=variant)
        probe._adaptive_form_override = form
        probe.set_options({"stepper": {"odeSolver": "runge_kutta_dopri", "tolAbs": 1e-8, "tolRel": 1e-7, "dtMax": 1e-3,
                                       "controllerUpdatePeriod": 1e-3, "sensorsUpdatePeriod": 1e-3},
                           "contacts": {"model": "spring_damper"}})
        if gen:
            probe._gen_checked = True      # (the step kernels' own variation check is not what is probed here)
            probe.set_lane_friction(torch.full((n,), float(probe._options["contacts"]["friction"]), dtype=torch.float64))
        if model.nmotors:
            probe.set_command(cmd)
        probe.start(q, v)
        for _ in range(3):
            probe.step(1e-3)
        ss = probe.stepper_state
        outs.append((probe._fields["q"].clone(), probe._fields["v"].clone(), ss.iter_lanes.clone(), ss.iter_failed_lanes.clone(),
                     probe.status.reshape(-1).clone()))
        probe.stop()
    (q1, v1, it1, if1, st1), (q0, v0, it0, if0, st0) = outs
    bad = ((st0 & (_abi.JM_LANE_NAN | _abi.JM_LANE_STEPPER_FAILURE)) != 0) & ((st1 & (_abi.JM_LANE_NAN | _abi.JM_LANE_STEPPER_FAILURE)) == 0)
    if bool(bad.any()):
        return float("inf")
    same = (it0 == it1) & (if0 == if1) & ((st1 & (_abi.JM_LANE_NAN | _abi.JM_LANE_STEPPER_FAILURE)) == 0)
    if float(same.double().mean()) < 0.8:
        return float("inf")
    err = 0.0
    for x, y in ((q0, q1), (v0, v1)):
        scale = torch.clamp(y[:, same].abs().max(), min=1.0)
        e = float((x - y)[:, same].abs().max() / scale)
        err = max(err, e if e == e else float("inf"))
    return err


_CON_VERIFIED: Dict[Tuple[str, int], float] = {}


def _constraint_self_test(model: CompiledModel, variant: int, device: torch.device) -> float:
    """Consistency check of the constraint-model kernel of one compiled library, from the device's
    own outputs: the joint wrenches of the RNEA-style extra terms (`joint_forces`, computed from
    `a` and `f_external` by sweeps that share nothing with the constrained solve) must reproduce the
    total effort vector, `S^T f_j + rotor a_j == u_j`, on probe lanes with active contact and
    joint-bound constraints.  Returns the largest relative residual over the lanes without NaN."""
    from .model import JT_PU, JT_PX, JT_RU, JT_RX
    n, dt = 64, 1e-4
    q, v, cmd = _probe_state(model, n)
    rng = np.random.default_rng(7)
    bounded = [j for j in range(1, model.njoints) if JT_RX <= int(model.jtypes[j]) <= JT_PU]
    for lane in range(0, n, 2):  # every other lane: one or two joints past a position limit
        for j in rng.permutation(bounded)[:2]:
            iq = int(model.idx_q[j])
            lo, hi = model.position_lower[iq], model.position_upper[iq]
            if np.isfinite(hi) and rng.random() < 0.5:
                q[iq, lane] = hi + 0.01
            elif np.isfinite(lo):
                q[iq, lane] = lo - 0.01
    if model.has_freeflyer and model.ncontacts:
        from .synthetic import lowest_contact_height
        q[2] += -2.0e-3 - lowest_contact_height(model, q)
    probe = BatchedEngine(model, n, dtype=torch.float64, device=device,
                          extra_outputs=("joint_forces",), _lib_variant=variant)
    probe.set_options({"stepper": {"odeSolver": "euler_explicit", "dtMax": dt, "controllerUpdatePeriod": dt,
                                   "sensorsUpdatePeriod": dt}, "contacts": {"model": "constraint"}},
                      _skip_constraint_check=True)
    if model.nmotors:
        probe.set_command(torch.as_tensor(cmd, dtype=torch.float64))
    probe.start(torch.as_tensor(q), torch.as_tensor(v))
    for _ in range(3):
        probe.step(dt)
    torch.cuda.synchronize(device)
    a = probe.field("a").cpu().numpy()
    u = probe.field("u").cpu().numpy().copy()
    jf = probe.field("joint_forces").cpu().numpy().reshape(model.njoints, 6, n)
    flags = probe.field("con_flags").cpu().numpy()
    data = probe.field("con_data").cpu().numpy()
    status = probe.status.cpu().numpy().reshape(-1)
    probe.stop()
    nb = len(bounded)
    for k, j in enumerate(bounded):  # reversed bound: generalised force is -lambda (engine.cc:3786-3790 adds +lambda)
        u[int(model.idx_v[j])] -= np.where(flags[k] & 2, 2.0 * data[nb + k], 0.0)
    tau = np.zeros_like(u)
    for j in range(1, model.njoints):
        t, iv = int(model.jtypes[j]), int(model.idx_v[j])
        if t == JT_FREEFLYER:
            tau[iv:iv + 6] = jf[j]
        elif t == 14:   # spherical joint: S^T f = the angular part
            tau[iv:iv + 3] = jf[j, 3:6]
        else:
            ax = {0: (1, 0, 0), 1: (0, 1, 0), 2: (0, 0, 1)}.get(
                {1: 0, 2: 1, 3: 2, 5: 0, 6: 1, 7: 2, 9: 0, 10: 1, 11: 2}.get(t, -1), tuple(model.axes[j]))
            part = jf[j, 0:3] if JT_PX <= t <= JT_PU else jf[j, 3:6]
            tau[iv] = ax[0] * part[0] + ax[1] * part[1] + ax[2] * part[2]
    tau += model.rotor_inertia[:, None] * a
    ok = (status & _abi.JM_LANE_NAN) == 0
    if not ok.any():
        return float("inf")
    scale = np.maximum(np.abs(u[:, ok]).max(axis=0), 1.0)
    res = np.abs(tau - u)[:, ok].max(axis=0) / scale
    return float(res.max()) if np.isfinite(res).all() else float("inf")


def _constraint_rows_self_test(model: CompiledModel, variant: int, device: torch.device) -> float:
    """The rows the CONSTRAINT-model kernels emit, against the spring-damper kernels of the same library (whose own emitted
    rows `_output_self_test` checks): with no constraint active -- robots in the air, joints inside their bounds -- the
    constrained evaluation is the plain articulated-body solve (`Engine::computeAcceleration`, engine.cc:3861-3865), so every
    output row of `start` and of one Euler step must agree to round-off.  Covers the output pass of the constraint family
    (sensors, extra terms), which the equation-of-motion residual of `_constraint_self_test` does not look at: the class of
    fault of DESIGN.md section 4.7, third case (a wrong IMU row next to right accelerations)."""
    n, dt = 64, 1e-4
    q, v, cmd = _probe_state(model, n)
    if model.has_freeflyer:
        q[2] += 5.0
    elif model.ncontacts:
        return 0.0          # (fixed-base robots cannot be lifted off the ground: nothing to compare without contacts)
    # every bounded joint well inside its range (a joint AT a limit keeps its bound constraint through the hysteresis)
    for j in range(1, model.njoints):
        if 1 <= int(model.jtypes[j]) <= 8:
            iq = int(model.idx_q[j])
            lo, hi = float(model.position_lower[iq]), float(model.position_upper[iq])
            if math.isfinite(lo) and math.isfinite(hi):
                q[iq] = lo + (0.3 + 0.4 * (np.arange(n) % 7) / 6.0) * (hi - lo)
    extras = ("contact_forces", "f_external", "energy", "joint_forces", "centroidal")
    probes = {}
    for contact_model in ("constraint", "spring_damper"):
        probe = BatchedEngine(model, n, dtype=torch.float64, device=device, _lib_variant=variant, extra_outputs=extras)
        probe.set_options({"stepper": {"odeSolver": "euler_explicit", "dtMax": dt, "controllerUpdatePeriod": dt, "sensorsUpdatePeriod": dt},
                           "contacts": {"model": contact_model}}, _skip_constraint_check=True)
        if model.nmotors:
            probe.set_command(torch.as_tensor(cmd, dtype=torch.float64))
        probe.start(torch.as_tensor(q), torch.as_tensor(v))
        probe.step(dt)
        probes[contact_model] = probe
    # `Engine::start` enables every constraint and solves with them (engine.cc:1266-1308): the first evaluations differ by
    # design.  After one step the switching has released them all; from THAT state (q, v, a copied over) one more step of
    # either model is the same computation
    for k in ("q", "v", "a"):
        probes["spring_damper"]._fields[k].copy_(probes["constraint"]._fields[k])
    runs = []
    for contact_model in ("spring_damper", "constraint"):
        probe = probes[contact_model]
        probe.step(dt)
        runs.append(({k: probe._fields[k].clone() for k in _OUTPUT_ROWS if k in probe._fields}, probe.status.reshape(-1).clone()))
    active = probes["constraint"]._fields["con_flags"].bitwise_and(1).sum(0) > 0
    for probe in probes.values():
        probe.stop()
    ok = (((runs[0][1] | runs[1][1]) & (_abi.JM_LANE_NAN | _abi.JM_LANE_OUT_OF_BOUNDS)) == 0) & ~active
    if not bool(ok.any()):
        return float("inf")
    err = 0.0
    for k in runs[0][0]:
        x, y = runs[0][0][k][:, ok], runs[1][0][k][:, ok]
        if x.numel() == 0:
            continue
        e = float((x - y).abs().max() / torch.clamp(x.abs().max(), min=1.0))
        err = max(err, e if e == e else float("inf"))
    return err


def _verified_library(model: CompiledModel, dtype: torch.dtype, device: torch.device) -> HipLibrary:
    try:
        return _verified_library_impl(model, dtype, device)
    finally:
        # the float64 probe rows (device tensors) only serve the cross-variant comparison inside the call above
        for k in [k for k in _OUTPUT_ROWS_F64 if k[0] == model.topology_hash()]:
            del _OUTPUT_ROWS_F64[k]


def _verified_library_impl(model: CompiledModel, dtype: torch.dtype, device: torch.device) -> HipLibrary:
    """The HIP library of `model`, checked once per process, topology and dtype by
    `_library_self_test`.  A build that fails the check is a toolchain mis-compile (DESIGN.md
    section 4.7): the next build variant (codegen.BUILD_VARIANTS) is compiled and checked instead;
    when none passes the engine refuses to run rather than integrate garbage."""
    key = (model.topology_hash(), dtype)
    if key in _VERIFIED:
        return load_for(model, variant=_VERIFIED[key])
    first = codegen.preferred_variant(model)
    if os.environ.get("JIMINY_AMD_SELF_TEST", "1") == "0":
        return load_for(model, variant=first)
    # (garbage from a mis-compiled library is >= 1e-4; sound builds stay below 1e-8: the two copies of the evaluation
    # contract their multiply-adds differently and a stiff ground contact amplifies that over the two steps)
    tol = 1e-7 if dtype == torch.float64 else 1e-3
    tried = []
    rows_only: List[Tuple[int, float]] = []   # variants that pass the step check and only disagree on the emitted rows
    for variant in [first] + [i for i in range(len(codegen.BUILD_VARIANTS)) if i != first]:
        err = _library_self_test(model, variant, dtype, device)
        tried.append(f"variant {variant}: {err:.3e}")
        if err <= tol and dtype == torch.float64:
            # ... and the emitted rows, float64 against float32 (5e-2: gross garbage only -- sound builds of the shipped and
            # the test robots stay below 1.5e-3, the mis-compiled output pass of DESIGN.md section 4.7 measured 1.6)
            err_out = _output_self_test(model, variant, device)
            if not err_out <= 5e-2:
                tried[-1] += f", emitted rows {err_out:.3e}"
                rows_only.append((variant, err_out))
                continue
        if err <= tol:
            if variant != first:
                warnings.warn(f"{model.name}: HIP library build variant {first} failed the kernel self-test "
                              f"({'; '.join(tried)}); using variant {variant} "
                              f"({' '.join(codegen.BUILD_VARIANTS[variant]) or 'default flags'}). Record it in "
                              "jiminy_amd/csrc/build_variants.json to pre-build it.")
            _VERIFIED[key] = variant
            return load_for(model, variant=variant)
    # (fails closed: a non-finite disagreement -- every probe lane NaN -- or a gross one is never taken for round-off)
    if len(rows_only) == len(codegen.BUILD_VARIANTS) and all(math.isfinite(e) for _, e in rows_only) and \
            max(e for _, e in rows_only) <= min(1.0, 2.0 * min(e for _, e in rows_only)):
        # every compilation shows the SAME float64 / float32 disagreement: that is the conditioning of the robot in float32
        # (extreme inertia ratios), not a mis-compile of one of them -- keep the preferred variant and say so
        warnings.warn(f"{model.name}: float64 and float32 kernels disagree on emitted rows alike in every build variant "
                      f"({'; '.join(tried)}): taken as float32 round-off of an ill-conditioned model, not as a mis-compile")
        _VERIFIED[key] = rows_only[0][0]
        return load_for(model, variant=rows_only[0][0])
    if len(rows_only) == len(codegen.BUILD_VARIANTS) and all(math.isfinite(e) for _, e in rows_only):
        # float32 is no second opinion on this robot (every variant disagrees with it, and grossly): let the float64 rows of
        # the separately compiled variants vouch for each other instead
        cross = _float64_rows_agree_across_variants(model, [v for v, _ in rows_only])
        if cross <= 1e-8:
            warnings.warn(f"{model.name}: the float32 kernels cannot check the float64 emitted rows of this model "
                          f"({'; '.join(tried)}); the float64 rows of all {len(rows_only)} build variants agree to {cross:.1e}")
            _VERIFIED[key] = rows_only[0][0]
            return load_for(model, variant=rows_only[0][0])
        tried.append(f"float64 rows across variants: {cross:.3e}")
    raise RuntimeError(
        f"kernel self-test failed for every build variant of topology {model.topology_hash()} "
        f"({model.name}; {'; '.join(tried)}): the in-loop and the peeled evaluation disagree, the "
        "HIP library was mis-compiled for this topology")


class BatchedEngine:
    def __init__(self, model: CompiledModel, batch_size: int, dtype: torch.dtype = torch.float64,
                 device: Optional[torch.device] = None,
                 extra_outputs: Tuple[str, ...] = ("contact_forces",),
                 _lib_variant: Optional[int] = None) -> None:
        if dtype not in (torch.float64, torch.float32):
            raise ValueError("dtype must be torch.float64 or torch.float32")
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device available: the batched engine runs on the GPU only "
                               "(there is no CPU fallback)")
        self.model = model
        self._topology_at_creation = model.topology_hash()    # kernels and the constraint-state rows are sized for THIS model
        self.batch_size = int(batch_size)
        self.dtype = dtype
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None \
            else torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the batched engine needs a HIP ('cuda') device")
        # `_lib_variant` is internal (probe engines of the library self-test)
        self._lib: HipLibrary = (load_for(model, variant=_lib_variant) if _lib_variant is not None
                                 else _verified_library(model, dtype, self.device))
        self._lib_variant_index = _lib_variant if _lib_variant is not None else \
            _VERIFIED.get((model.topology_hash(), dtype), codegen.preferred_variant(model))
        self._L = self._lib.L
        self._desc, self._keep = _abi.make_model_desc(model)
        self._model_h = C.c_void_p()
        self._lib.check(self._L.jm_model_create(C.byref(self._desc), C.byref(self._model_h)))
        self._batch_h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._lib.check(self._L.jm_batch_create(
            self._model_h, self.batch_size,
            _abi.JM_F64 if dtype == torch.float64 else _abi.JM_F32, dev_index,
            C.byref(self._batch_h)))
        self._options = default_options(dtype)
        self._rows = _abi.field_rows(model)
        B = self.batch_size
        mandatory = ("q", "v", "a", "command", "u_motor", "u", "imu", "force", "contact",
                     "encoder", "effort")
        self._fields: Dict[str, torch.Tensor] = {}
        for name in mandatory + tuple(extra_outputs):
            self._alloc(name)
        self._fields["status"] = torch.zeros((1, B), dtype=torch.int32, device=self.device)
        self._bind("status")
        self._running = False
        self._opening_step = False
        self._t = 0.0
        self._t_prev = 0.0
        self._t_error = 0.0
        self._dt = 0.0
        self._iter = 0
        self._command_dirty = True
        self._adaptive: Optional[Dict[str, torch.Tensor]] = None
        self._sensor_noise: Dict[str, Dict[str, Any]] = {}
        self.adaptive_attempts = 0   # device attempts of the last `step` with the adaptive solver
        # model biases per environment, ground profile, impulse / profile forces (section 4.9 of DESIGN.md)
        from .randomization import default_dynamics_options
        self._model_options: Dict[str, Dict[str, float]] = {"dynamics": default_dynamics_options()}
        self._model_rng: Optional[torch.Tensor] = None      # PCG32 state of every lane's engine generator
        self._bias_table: Optional[torch.Tensor] = None     # nominal body parameters + principal axes (device)
        self._ground: Optional[torch.Tensor] = None
        self._force_frames: List[str] = []
        self._user_constraints: Dict[str, Any] = {}
        self._impulse_forces: List[Dict[str, Any]] = []
        self._impulse_active: List[int] = []
        self._profile_forces: List[Dict[str, Any]] = []
        self._apply_options()
        self._apply_hardware_sensor_options()

    # ------------------------------------------------------------------ memory
    def _alloc(self, name: str) -> torch.Tensor:
        rows = max(self._rows[name], 1)
        t = torch.zeros((rows, self.batch_size), dtype=self.dtype, device=self.device)
        self._fields[name] = t
        if self._rows[name] > 0:
            self._bind(name)
        return t

    def _bind(self, name: str) -> None:
        t = self._fields[name]
        self._lib.check(self._L.jm_batch_bind(self._batch_h, _abi.FIELD_NAMES[name],
                                              C.c_void_p(t.data_ptr())))

    def enable_output(self, name: str) -> torch.Tensor:
        """Allocate and bind one of the optional outputs
        ('f_external', 'contact_forces', 'energy', 'joint_forces', 'centroidal')."""
        if name not in ("f_external", "contact_forces", "energy", "joint_forces", "centroidal"):
            raise LookupError(f"unknown optional output '{name}'")
        if name not in self._fields:
            self._alloc(name)
        return self._fields[name]

    def __del__(self) -> None:
        try:
            if getattr(self, "_batch_h", None) and self._batch_h.value:
                self._L.jm_batch_destroy(self._batch_h)
                self._batch_h = C.c_void_p()
            if getattr(self, "_model_h", None) and self._model_h.value:
                self._L.jm_model_destroy(self._model_h)
                self._model_h = C.c_void_p()
        except Exception:  # pragma: no cover
            pass

    # ------------------------------------------------------------------ options
    def get_options(self) -> Dict[str, Dict[str, Any]]:
        return {k: dict(v) for k, v in self._options.items()}

    def set_options(self, options: Dict[str, Dict[str, Any]], _skip_constraint_check: bool = False) -> None:
        """≙ `Engine::setOptions` (reference engine.cc:2654-2795), hot-path subset."""
        if self._running:
            raise BadControlFlow("Please stop the simulation before updating the options.")
        new = self.get_options()
        for section, values in options.items():
            if section not in new:
                continue  # sections outside the hot path (telemetry, constraints, ...) are ignored
            for k, v in values.items():
                if k in new[section]:
                    new[section][k] = v
        st, ct = new["stepper"], new["contacts"]
        if st["odeSolver"] not in SOLVER_IDS:
            raise NotImplementedError(
                f"odeSolver '{st['odeSolver']}' is not available on the batched path "
                f"(available: {sorted(SOLVER_IDS)})")
        if ct["model"] not in _abi.CONTACT_MODELS:
            raise ValueError("The requested contact model is not available.")  # engine.cc:2741-2747
        if ct["model"] == "constraint" and self.dtype != torch.float64:
            raise NotImplementedError("contacts.model='constraint' needs a float64 engine")
        if new["constraints"]["solver"] != "PGS":
            raise ValueError("The requested constraint solver is not available.")  # engine.cc:2720-2728
        if float(new["constraints"]["regularization"]) < 0.0:
            raise ValueError("Constraint option 'regularization' must be positive.")  # engine.cc:2730-2735
        if float(ct["stabilizationFreq"]) < 0.0:
            raise ValueError("Contact option 'stabilizationFreq' must be positive.")  # engine.cc:2758-2763
        dt_max = float(st["dtMax"])
        if not (SIMULATION_MIN_TIMESTEP - EPS <= dt_max <= SIMULATION_MAX_TIMESTEP + EPS):
            raise ValueError("'dtMax' option is out of range.")  # engine.cc:2668-2673
        if int(st["successiveIterFailedMax"]) < 1:
            raise ValueError("'successiveIterFailedMax' must be strictly positive.")  # engine.cc:2677-2684
        if not (float(st["tolAbs"]) > 0.0 and float(st["tolRel"]) > 0.0):
            raise ValueError("'tolAbs' and 'tolRel' must be strictly positive.")
        cp, sp = float(st["controllerUpdatePeriod"]), float(st["sensorsUpdatePeriod"])
        for p in (cp, sp):
            if EPS < p < SIMULATION_MIN_TIMESTEP:
                raise ValueError("Cannot simulate a discrete robot with update period smaller "
                                 "than 1us.")  # engine.cc:2714-2722
        if not is_gcd_included(cp, sp)[0]:
            raise ValueError("In discrete mode, the controller and sensor update periods "
                             "must be multiple of each other.")  # engine.cc:2699-2733 (isGcdIncluded: helpers.hxx:94-116)
        if len(new["world"]["gravity"]) != 6:
            raise ValueError("The size of the gravity force vector must be 6.")
        if self._user_constraints and ct["model"] != "constraint":
            # (they would stay registered and silently inactive: only the constraint contact model solves for them)
            raise ValueError("user constraints are registered: remove them before switching away from "
                             "contacts.model='constraint'")
        self._options = new
        self._apply_options()
        if ct["model"] == "constraint" and not _skip_constraint_check:
            self._check_constraint_kernel()

    def _check_constraint_kernel(self) -> None:
        """First use of the constraint model for this topology in the process: run the kernel
        self-test and refuse to run a library that fails it."""
        key = (self.model.topology_hash(), _VERIFIED.get((self.model.topology_hash(), self.dtype),
                                                         codegen.preferred_variant(self.model)))
        if key in _CON_VERIFIED or os.environ.get("JIMINY_AMD_SELF_TEST", "1") == "0":
            return
        err = _constraint_self_test(self.model, key[1], self.device)
        if not err <= 1e-8:
            raise RuntimeError(
                f"constraint-model kernel self-test failed for topology {self.model.topology_hash()} "
                f"({self.model.name}, build variant {key[1]}): equation-of-motion residual {err:.3e}. "
                "The compiled library is unsound (toolchain mis-compile, DESIGN.md section 4.7); rebuild it "
                "with another variant (JIMINY_AMD_BUILD_VARIANT).")
        err_rows = _constraint_rows_self_test(self.model, key[1], self.device)
        if not err_rows <= 1e-8:
            raise RuntimeError(
                f"constraint-model kernel self-test failed for topology {self.model.topology_hash()} ({self.model.name}, build "
                f"variant {key[1]}): with no constraint active its emitted rows differ from the spring-damper kernels' by "
                f"{err_rows:.3e} (toolchain mis-compile of its output pass, DESIGN.md section 4.7); rebuild the library with "
                "another variant (JIMINY_AMD_BUILD_VARIANT).")
        _CON_VERIFIED[key] = err

    def _check_variation_kernels(self) -> None:
        """First use of per-lane body parameters / a height map / applied forces for this topology and contact
        model in the process: run `_variation_self_test` and refuse to run a library that fails it."""
        if getattr(self, "_gen_checked", False) or self.dtype != torch.float64 or \
                os.environ.get("JIMINY_AMD_SELF_TEST", "1") == "0":
            return
        # (the one-robot-per-lane kernels read the lane's friction in their only contact law: no variation form to check)
        lane_mu = "friction" in self._fields and self._options["contacts"]["model"] != "constraint" and \
            codegen.quad_structure(self.model) is not None and os.environ.get("JM_KERNEL_VARIANT") != "lane"
        # (user constraints: only the branch-parallel family moves to its variation kernels for them; the lane kernel has none)
        locks = bool(self._user_constraints) and codegen.quad_structure(self.model) is not None and \
            os.environ.get("JM_KERNEL_VARIANT") != "lane"
        # (either family: the variation kernels -- `k_quad_gen` ..., or the variation instantiations of the one-robot-per-lane kernels)
        if not ("model_lane" in self._fields or "applied" in self._fields or self._ground is not None or lane_mu or locks):
            return
        self._gen_checked = True
        variant = self._lib_variant_index
        key = (self.model.topology_hash(), variant, self._options["contacts"]["model"])
        if key not in _GEN_VERIFIED:
            err = _variation_self_test(self.model, variant, self.device, key[2])
            if not err <= 1e-8:
                raise RuntimeError(
                    f"variation-kernel self-test failed for topology {key[0]} ({self.model.name}, build variant "
                    f"{variant}, contact model {key[2]}): disagreement with the plain kernel {err:.3e} (toolchain "
                    "mis-compile, DESIGN.md section 4.7); rebuild with another variant (JIMINY_AMD_BUILD_VARIANT).")
            _GEN_VERIFIED[key] = err

    def _apply_options(self) -> None:
        ct = self._options["contacts"]
        o = _abi.make_options(gravity=self._options["world"]["gravity"],
                              stiffness=ct["stiffness"], damping=ct["damping"],
                              friction=ct["friction"], transition_eps=ct["transitionEps"],
                              transition_velocity=ct["transitionVelocity"])
        self._lib.check(self._L.jm_batch_set_options(self._batch_h, C.byref(o)))
        # constraint contact model: PGS tolerances are the stepper tolerances (engine.cc:1366-1374)
        st = self._options["stepper"]
        co = _abi.make_constraint_options(
            model=ct["model"], torsion=ct["torsion"], stabilization_freq=ct["stabilizationFreq"],
            regularization=self._options["constraints"]["regularization"],
            tol_abs=st["tolAbs"], tol_rel=st["tolRel"], user_stabilization_freq=self._user_constraint_freq())
        self._lib.check(self._L.jm_batch_set_constraint_options(self._batch_h, C.byref(co)))
        if ct["model"] == "constraint" and "con_flags" not in self._fields:
            # per-lane constraint state + delassus workspace (caller-owned, like every batch field)
            nf, nd, nw = C.c_int32(0), C.c_int32(0), C.c_int32(0)
            self._lib.check(self._L.jm_batch_constraint_rows(self._batch_h, C.byref(nf), C.byref(nd), C.byref(nw)))
            B = self.batch_size
            self._fields["con_flags"] = torch.zeros((max(nf.value, 1), B), dtype=torch.int32, device=self.device)
            self._fields["con_data"] = torch.zeros((max(nd.value, 1), B), dtype=self.dtype, device=self.device)
            self._fields["workspace"] = torch.zeros((max(nw.value, 1), B), dtype=self.dtype, device=self.device)
            for name in ("con_flags", "con_data", "workspace"):
                self._bind(name)

    # ------------------------------------------------------------------ user-registered constraints
    def add_constraint(self, name: str, constraint: Any, lane_mask: Optional[torch.Tensor] = None) -> None:
        """≙ `Model::addConstraint(name, constraint)` (core/src/robot/model.cc:926-936), user registry; constraint contact
        model, float64 batches.  `lane_mask`: the environments that get it (default: all).

        * `JointConstraint` of a joint with position bounds (both kernel families): the row of the joint's own bound constraint
          becomes bilateral (bit 2 of its flag), is solved first in every Gauss-Seidel sweep without projection
          (constraint_solvers.cc:112-128) and its multiplier is not restored into `RobotState::u` (engine.cc:3771-3790); the
          bound of a locked joint is not switched while the lock holds.
          A joint declared with `model.add_joint_constraint` gets a row OF ITS OWN instead (round 5, one-robot-per-lane
          kernels): any revolute / prismatic joint, next to its bound constraint like in the reference (model.cc:884-905).
        * `FrameConstraint` of a frame declared with `model.add_frame_constraint` (round 5; robots on the one-robot-per-lane
          kernels): rows of their own behind the contact rows, unbounded; when nothing but unbounded constraints is enabled
          the multipliers come from the exact solve of the reference's `isUnbounded` branch (constraint_solvers.cc:362-412).

        * `SphereConstraint`, `WheelConstraint`, `DistanceConstraint` (round 5): declared with `model.add_sphere_constraint` /
          `add_wheel_constraint` / `add_distance_constraint`; three rows at the contact point of the rolling body, one row along
          the line between two frames (either of which may be fixed to the world)."""
        if self._running:
            raise BadControlFlow("Please stop the simulation before adding constraints.")   # model.cc:866-872
        if not isinstance(constraint, (JointConstraint, FrameConstraint, SphereConstraint, WheelConstraint, DistanceConstraint)):
            raise NotImplementedError("JointConstraint, FrameConstraint, SphereConstraint, WheelConstraint and DistanceConstraint "
                                      "can be registered")
        if name in self._user_constraints:
            raise ValueError(f"a constraint named '{name}' is already registered")                  # model.cc:884-890
        if self.model.topology_hash() != self._topology_at_creation:
            raise BadControlFlow("the model was modified after this engine was created (constraints, contact points or sensors "
                                 "declared on it later): the kernels and the con_flags / con_data rows of the batch were sized "
                                 "for the model as it was -- declare everything on the model first, then create the engine")
        freq = constraint.baumgarte_freq
        if freq is not None and freq < 0.0:
            raise ValueError("Natural frequency must be positive.")                                 # abstract_constraint.cc:91-94
        # (the frequency is read HERE: mutating the constraint object afterwards does not move the gains of the batch)
        others = {f for _, _, _, f in self._user_constraints.values()}
        if others and others != {freq}:
            raise NotImplementedError("the user constraints of a batch share one Baumgarte frequency "
                                      f"({next(iter(others))}): give this one the same `baumgarte_freq` (mind the defaults: "
                                      "JointConstraint None = the gains of contacts.stabilizationFreq, the frame-type "
                                      "constraints 0.0 = the reference's freshly created constraint)")
        if self._options["contacts"]["model"] != "constraint" or self.dtype != torch.float64:
            raise NotImplementedError("user constraints need the constraint contact model on a float64 batch")
        if "con_flags" not in self._fields:
            self._apply_options()
        mask = torch.ones(self.batch_size, dtype=torch.bool, device=self.device) if lane_mask is None else \
            lane_mask.to(self.device).bool()
        if not isinstance(constraint, JointConstraint):
            def close(a, b):
                return a is not None and np.allclose(np.asarray(a, dtype=float) / np.linalg.norm(a), np.asarray(b, dtype=float))
            declared = []
            for xd in self.model.constraint_frames:
                kd = xd.get("kind", "frame")
                declared.append(
                    (kd == "frame" and isinstance(constraint, FrameConstraint) and xd["frame"] == constraint.frame_name
                     and int(xd["mask"]) == constraint.mask)
                    or (kd == "sphere" and isinstance(constraint, SphereConstraint) and xd["frame"] == constraint.frame_name
                        and abs(xd["radius"] - constraint.radius) < 1e-12 and close(constraint.ground_normal, xd["normal"]))
                    or (kd == "wheel" and isinstance(constraint, WheelConstraint) and xd["frame"] == constraint.frame_name
                        and abs(xd["radius"] - constraint.radius) < 1e-12 and close(constraint.ground_normal, xd["normal"])
                        and close(constraint.wheel_axis, xd["axis"]))
                    or (kd == "distance" and isinstance(constraint, DistanceConstraint) and xd["frame"] == constraint.first_frame_name
                        and xd["frame2"] == constraint.second_frame_name))
            if True not in declared:
                raise LookupError(f"{constraint!r} is not declared on the model: call jiminy_amd.model.add_frame_constraint / "
                                  "add_sphere_constraint / add_wheel_constraint / add_distance_constraint with the same arguments "
                                  "before creating the engine (the kernels are specialised on the constraint frames)")
            x = declared.index(True)
            if any(k == "frame" and r == x for k, r, _, _ in self._user_constraints.values()):
                raise ValueError(f"{constraint!r} is already registered")
            rows = _abi.constraint_rows(self.model)
            self._fields["con_flags"][rows["n_bounds"] + rows["n_contacts"] + x] = torch.where(mask, 1, 0).to(torch.int32)
            self._user_constraints[name] = ("frame", x, constraint, freq)
            self._apply_options()
            return
        declared_j = [int(x["joint"]) for x in self.model.constraint_joints]
        if self.model.joint_index(constraint.joint_name) in declared_j:
            # a row of its own (declared with `model.add_joint_constraint`; one-robot-per-lane kernels): coexists with the
            # joint's bound constraint like in the reference (model.cc:884-905), any revolute / prismatic joint
            k = declared_j.index(self.model.joint_index(constraint.joint_name))
            if any(kd == "jrow" and r == k for kd, r, _, _ in self._user_constraints.values()):
                raise ValueError(f"joint '{constraint.joint_name}' already carries a user constraint")
            rows = _abi.constraint_rows(self.model)
            self._fields["con_flags"][rows["user_joint_flag"] + k] = torch.where(mask, 1, 0).to(torch.int32)
            self._user_constraints[name] = ("jrow", k, constraint, freq)
            self._apply_options()
            return
        # Deviation from the reference, where a user JointConstraint is a row of its own next to the joint's bound constraint
        # (model.cc:884-905): here it REUSES the bound row of the joint (flag bit 2), so only bounded 1-dof joints can be
        # locked and the bound does not switch while the lock holds -- the lock is the tighter constraint anyway.
        row = self.model.bound_row(constraint.joint_name)
        if any(k == "joint" and r == row for k, r, _, _ in self._user_constraints.values()):
            raise ValueError(f"joint '{constraint.joint_name}' already carries a user constraint")
        self._lib.check(self._L.jm_batch_set_joint_locks(self._batch_h, 1))
        self._fields["con_flags"][row] |= torch.where(mask, 4, 0).to(torch.int32)
        self._user_constraints[name] = ("joint", row, constraint, freq)
        self._apply_options()      # (jm_constraint_options::user_stabilization_freq)

    def _user_constraint_freq(self) -> float:
        """`jm_constraint_options::user_stabilization_freq`: the Baumgarte frequency the registered user constraints share
        (as given when they were registered), -1 = the gains of `contacts.stabilizationFreq` (`baumgarte_freq=None`)."""
        for _, _, _, freq in getattr(self, "_user_constraints", {}).values():
            return -1.0 if freq is None else float(freq)
        return -1.0

    def remove_constraint(self, name: str) -> None:
        """≙ `Model::removeConstraint(name)` (model.cc:1010-1013)."""
        if self._running:
            raise BadControlFlow("Please stop the simulation before removing constraints.")
        kind, row, _, _ = self._user_constraints.pop(name)
        if kind == "frame":
            rows = _abi.constraint_rows(self.model)
            self._fields["con_flags"][rows["n_bounds"] + rows["n_contacts"] + row] = 0
        elif kind == "jrow":
            self._fields["con_flags"][_abi.constraint_rows(self.model)["user_joint_flag"] + row] = 0
        else:
            self._fields["con_flags"][row] &= ~4
        if not any(k == "joint" for k, _, _, _ in self._user_constraints.values()):
            self._lib.check(self._L.jm_batch_set_joint_locks(self._batch_h, 0))
        self._apply_options()

    def set_constraint_reference(self, name: str, reference: Any) -> None:
        """≙ `JointConstraint.reference_configuration = ...` (joint_constraint.cc `setReferenceConfiguration`): where the
        constraint holds its joint, one value per lane or one for all; ≙ `FrameConstraint.reference_transform = ...`
        (frame_constraint.cc:52-60): `reference` = (translation `(3,)` or `(3, B)`, rotation `(3, 3)` or `(3, 3, B)`).
        `start` / `reset_lanes` take the reference from the state again (`reset` of the constraint): call it after them."""
        kind, row, _, _ = self._user_constraints[name]
        if kind == "frame" and isinstance(self._user_constraints[name][2], DistanceConstraint):
            # ≙ `DistanceConstraint.reference_distance = ...` (distance_constraint.cc:44-52): one length per lane or one for all
            d = torch.as_tensor(reference, dtype=self.dtype, device=self.device).reshape(-1)
            if d.numel() not in (1, self.batch_size) or bool((d < 0).any()):
                raise ValueError("one non-negative reference distance per lane (or one for all)")
            self._fields["con_data"][_abi.constraint_rows(self.model)["user_ref"] + 12 * row] = d.expand(self.batch_size)
            return
        if kind == "frame":
            p, R = reference
            p = torch.as_tensor(p, dtype=self.dtype, device=self.device)
            R = torch.as_tensor(R, dtype=self.dtype, device=self.device)
            if p.dim() == 1:
                p = p[:, None].expand(3, self.batch_size)
            if R.dim() == 2:
                R = R[:, :, None].expand(3, 3, self.batch_size)
            if tuple(p.shape) != (3, self.batch_size) or tuple(R.shape) != (3, 3, self.batch_size):
                raise ValueError("reference transform: translation (3,) or (3, B) and rotation (3, 3) or (3, 3, B)")
            r0 = _abi.constraint_rows(self.model)["user_ref"] + 12 * row
            self._fields["con_data"][r0:r0 + 3] = p
            self._fields["con_data"][r0 + 3:r0 + 12] = R.reshape(9, self.batch_size)
            return
        ref = torch.as_tensor(reference, dtype=self.dtype, device=self.device).reshape(-1)
        if ref.numel() not in (1, self.batch_size):
            raise ValueError("one reference per lane (or one for all)")
        if kind == "jrow":
            row = _abi.constraint_rows(self.model)["user_joint_ref"] + row
        self._fields["con_data"][row] = ref.expand(self.batch_size)

    def constraint_reference(self, name: str) -> Any:
        """The reference the constraint holds at the moment: `(B,)` joint positions, or (translation `(3, B)`, rotation
        `(3, 3, B)`) of a `FrameConstraint` (views of the `con_data` field)."""
        kind, row, _, _ = self._user_constraints[name]
        if kind == "frame":
            r0 = _abi.constraint_rows(self.model)["user_ref"] + 12 * row
            d = self._fields["con_data"]
            if isinstance(self._user_constraints[name][2], DistanceConstraint):
                return d[r0]
            return d[r0:r0 + 3], d[r0 + 3:r0 + 12].view(3, 3, self.batch_size)
        if kind == "jrow":
            row = _abi.constraint_rows(self.model)["user_joint_ref"] + row
        return self._fields["con_data"][row]

    @property
    def user_constraints(self) -> Dict[str, Any]:
        return {k: c for k, (_, _, c, _) in self._user_constraints.items()}

    def set_lane_friction(self, friction: Optional[Any]) -> None:
        """Ground friction coefficient of every lane (`contacts.friction` randomised per environment as
        `WalkerJiminyEnv._setup` does per episode, gym_jiminy envs/locomotion.py:257-262).  `(B,)` values, or
        None to go back to the batch-wide option.  Both contact models, both kernel families (the spring-damper law of the
        branch-parallel family reads it in the per-environment variation kernels: float64 batches there)."""
        if friction is None:
            self._fields.pop("friction", None)
            self._lib.check(self._L.jm_batch_bind(self._batch_h, _abi.FIELD_NAMES["friction"], None))
            return
        quad = codegen.quad_structure(self.model) is not None and os.environ.get("JM_KERNEL_VARIANT") != "lane"
        if self._options["contacts"]["model"] != "constraint" and quad and self.dtype != torch.float64:
            raise NotImplementedError("per-lane friction with the spring-damper model on a branch-parallel topology (floating base "
                                      "with limb chains) needs a float64 batch")
        f = torch.as_tensor(friction, dtype=self.dtype, device=self.device).reshape(1, -1)
        if f.shape[1] != self.batch_size or bool((f < 0).any()):
            raise ValueError("friction must hold one non-negative value per lane")
        if "friction" not in self._fields:
            self._fields["friction"] = torch.empty((1, self.batch_size), dtype=self.dtype, device=self.device)
            self._bind("friction")
        self._fields["friction"].copy_(f)

    def set_lane_flexibility(self, stiffness: Optional[Any], damping: Optional[Any] = None) -> None:
        """Stiffness and damping of the flexibility (spherical) joints of every lane: `flexibilityConfig` randomised per
        environment, the way `WalkerJiminyEnv._setup` draws it per episode (gym_jiminy envs/locomotion.py:288-296).
        `stiffness`, `damping`: `(nflex, 3, B)` or `(nflex, 3)` (the same for every lane), the joints in the order of
        `model.flexibility_joint_indices`; None to go back to the model's values.  Read by the flexibility efforts of
        `Engine::computeInternalDynamics` (engine.cc:3365-3391), one-robot-per-lane kernels (the family of every model
        with a spherical joint), both contact models, fixed-step and adaptive solvers."""
        nflex = len(self.model.flexibility_joint_indices)
        if stiffness is None:
            self._fields.pop("flexibility", None)
            self._lib.check(self._L.jm_batch_bind(self._batch_h, _abi.FIELD_NAMES["flexibility"], None))
            return
        if nflex == 0:
            raise LookupError(f"{self.model.name} has no flexibility joint")
        if damping is None:
            raise ValueError("give stiffness and damping together")
        rows = torch.empty((6 * nflex, self.batch_size), dtype=self.dtype, device=self.device)
        for o, val in ((0, stiffness), (3, damping)):
            t = torch.as_tensor(val, dtype=self.dtype, device=self.device)
            if t.dim() == 2:
                t = t.unsqueeze(-1).expand(-1, -1, self.batch_size)
            if tuple(t.shape) != (nflex, 3, self.batch_size) or bool((t < 0).any()):
                raise ValueError(f"expected non-negative values of shape ({nflex}, 3) or ({nflex}, 3, {self.batch_size})")
            for k in range(nflex):
                rows[6 * k + o:6 * k + o + 3] = t[k]
        if "flexibility" not in self._fields:
            self._fields["flexibility"] = rows
            self._bind("flexibility")
        else:
            self._fields["flexibility"].copy_(rows)

    # ------------------------------------------------------------------ state access
    @property
    def is_simulation_running(self) -> bool:
        return self._running

    @property
    def robot_state(self) -> RobotState:
        f = self._fields
        return RobotState(f["q"], f["v"], f["a"], f["command"], f["u"], f["u_motor"],
                          f.get("f_external"))

    @property
    def stepper_state(self) -> StepperState:
        f = self._fields
        ad = self._adaptive
        if ad is None:
            return StepperState(self._iter, 0, self._t, self._t_prev, self._t_error, self._dt,
                                f["q"], f["v"], f["a"])
        # adaptive solver: `iter` / `iter_failed` / `dt` report the worst lane (one host sync)
        return StepperState(int(ad["i32"][0].max()), int(ad["i32"][1].max()), self._t, self._t_prev,
                            self._t_error, float(ad["f64"][1].min()), f["q"], f["v"], f["a"],
                            ad["f64"][1], ad["f64"][2], ad["i32"][0], ad["i32"][1])

    @property
    def status(self) -> torch.Tensor:
        """Per-lane status bits (JM_LANE_*), int32 tensor of shape (B,)."""
        return self._fields["status"][0]

    @property
    def command(self) -> torch.Tensor:
        return self._fields["command"]

    def set_command(self, command: torch.Tensor) -> None:
        """Write the motor commands `[nmotors][B]` (≙ what `computeCommand` leaves in
        `RobotState::command`, reference engine.cc:3240-3251)."""
        self._fields["command"].copy_(command)
        self._command_dirty = True

    def mark_command_changed(self) -> None:
        self._command_dirty = True

    @property
    def sensor_measurements(self) -> Dict[str, torch.Tensor]:
        """`{SensorType: tensor (n_fields, n_sensors, B)}` (reference layout is
        `(n_fields, n_sensors)`, gym_jiminy common/utils/spaces.py:107-152)."""
        f, s = self._fields, self.model.sensors
        B = self.batch_size
        out: Dict[str, torch.Tensor] = {}
        spec = (("ImuSensor", "imu", 6), ("ForceSensor", "force", 6), ("ContactSensor", "contact", 3),
                ("EncoderSensor", "encoder", 2))
        for stype, name, nf in spec:
            n = len(s.get(stype, []))
            if n:
                out[stype] = f[name].view(n, nf, B).permute(1, 0, 2)
        n = len(s.get("EffortSensor", []))
        if n:
            out["EffortSensor"] = f["effort"].view(1, n, B)
        return out

    def field(self, name: str) -> torch.Tensor:
        return self._fields[name]

    # ------------------------------------------------------------------ control flow
    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _to_soa(self, x: Any, rows: int, what: str) -> torch.Tensor:
        t = torch.as_tensor(x, dtype=self.dtype, device=self.device)
        if t.dim() == 1:
            t = t[:, None].expand(rows, self.batch_size)
        if tuple(t.shape) != (rows, self.batch_size):
            raise ValueError(f"{what} must have shape ({rows}, B) or ({rows},)")
        return t

    def reset(self) -> None:
        """≙ `Engine::reset` (reference engine.cc:726-775): stop and clear the state."""
        self.stop()
        for name in ("q", "v", "a", "command", "u", "u_motor"):
            self._fields[name].zero_()
        self._fields["status"].zero_()

    def start(self, q_init: Any, v_init: Any, a_init: Any = None) -> None:
        """≙ `Engine::start(q, v, a)` (reference engine.cc:952-1533)."""
        if self._running:
            raise BadControlFlow("Simulation already running. Please stop it before starting a "
                                 "new one.")  # engine.cc:960-965
        m = self.model
        q = self._to_soa(q_init, m.nq, "q_init")
        v = self._to_soa(v_init, m.nv, "v_init")
        self._fields["q"].copy_(q)
        self._fields["v"].copy_(v)
        if a_init is not None:
            self._fields["a"].copy_(self._to_soa(a_init, m.nv, "a_init"))
        else:
            self._fields["a"].zero_()
        self._t = self._t_prev = self._t_error = 0.0
        self._dt = 0.0
        self._iter = 0
        if any(float(v) > EPS for v in self._model_options["dynamics"].values()):
            self.sample_model_biases()   # ≙ Model::reset -> generateModelBiased at the start of every simulation
        # a new simulation re-evaluates every profile force and forgets the impulses that were active when the last
        # one stopped (the reference evaluates all external forces in Engine::start, engine.cc:1311-1330): a held
        # value with `t_last` from the previous simulation would otherwise survive until t exceeds that time again
        for pf in self._profile_forces:
            pf["value"], pf["t_last"] = None, -math.inf
        self._impulse_active = []
        self._update_applied_forces(0.0)
        self._check_variation_kernels()
        self._lib.check(self._L.jm_batch_start(self._batch_h, self._stream()))
        if self._sensor_noise:
            # `Engine::start` measures the sensors INIT_ITERATIONS times while it solves the initial
            # acceleration / sensor / controller coupling, then once more (engine.cc:61,1400-1441,
            # 1470-1480): same number of draws here, so that the streams stay aligned
            self._reset_sensor_history()
            self._apply_sensor_noise(discard_rounds=INIT_ITERATIONS, t_now=0.0)
        self._setup_adaptive()
        self._running = True
        self._command_dirty = False
        # `stepperState_.reset(SIMULATION_MIN_TIMESTEP, ...)` (engine.cc:1176): the next `step` opens with a 1 us step.
        # (`reset_lanes` re-initialises lanes INSIDE a running simulation whose launches carry one step size for the
        # whole batch: those lanes continue with `dtMax`)
        self._opening_step = True

    def stop(self) -> None:
        """≙ `Engine::stop` (reference engine.cc:2419-2460)."""
        if self._running:
            self._lib.check(self._L.jm_batch_stop(self._batch_h))
        self._running = False

    def _setup_adaptive(self) -> None:
        """Workspace and per-lane stepper state of the adaptive solver (`StepperState::reset`,
        engine.h:219-236: dt = dtLargest = dtLargestPrev = SIMULATION_MIN_TIMESTEP)."""
        if self._options["stepper"]["odeSolver"] != "runge_kutta_dopri":
            self._adaptive = None
            return
        B = self.batch_size
        rows = int(self._L.jm_batch_adaptive_workspace_rows(self._batch_h))  # depends on the contact model
        if self._adaptive is not None and self._adaptive["ws"].shape[0] != rows:
            self._adaptive = None
        if self._adaptive is None:
            self._adaptive = {
                "ws": torch.zeros((rows, B), dtype=self.dtype, device=self.device),
                "f64": torch.zeros((5, B), dtype=torch.float64, device=self.device),
                "i32": torch.zeros((7, B), dtype=torch.int32, device=self.device),
            }
            ad = self._adaptive
            self._lib.check(self._L.jm_batch_bind_adaptive(
                self._batch_h, C.c_void_p(ad["ws"].data_ptr()), C.c_void_p(ad["f64"].data_ptr()),
                C.c_void_p(ad["i32"].data_ptr())))
        ad = self._adaptive
        ad["f64"].zero_()
        ad["f64"][1:4] = SIMULATION_MIN_TIMESTEP
        ad["i32"].zero_()

    def _step_adaptive(self, step_dt: float) -> None:
        st = self._options["stepper"]
        # (impulse / profile forces and per-lane model biases: available where the stepper is the persistent kernel
        # of jm_qdopri.h, whose lanes keep their places; the per-stage path over compacted lanes refuses them)
        intervals, t_end, t_err = plan_breakpoints(self._t, self._t_error, float(step_dt), self._options,
                                                   self._force_breakpoints(self._t))
        o = _abi.AdaptiveOptions(float(st["tolRel"]), float(st["tolAbs"]), float(st["dtMax"]),
                                 float(st["dtRestoreThresholdRel"]), int(st["successiveIterFailedMax"]),
                                 self._adaptive_form())
        stream = self._stream()
        attempts = C.c_int32(0)
        self.adaptive_attempts = 0
        # (only discrete controllers have breakpoints: engine.cc:1919-1940)
        constraint_model = (self._options["contacts"]["model"] == "constraint"
                            and float(st["controllerUpdatePeriod"]) > EPS)
        t_now = self._t
        for i, (t_next, cmd_bp, sens) in enumerate(intervals):
            forces_changed = self._update_applied_forces(t_now)
            t_now = float(t_next)
            changed = (cmd_bp and (self._command_dirty or constraint_model)) or forces_changed
            self._lib.check(self._L.jm_batch_step_adaptive(
                self._batch_h, float(t_next), C.byref(o), int(i == 0), int(changed), int(sens),
                100000, C.byref(attempts), stream))
            if changed and cmd_bp:
                self._command_dirty = False
            self.adaptive_attempts += int(attempts.value)
            if sens and self._sensor_noise:
                if float(st["sensorsUpdatePeriod"]) <= 0.0:
                    raise NotImplementedError(
                        "sensor noise with the adaptive solver needs a positive sensorsUpdatePeriod "
                        "(lanes take different internal steps)")
                self._apply_sensor_noise(t_now=float(t_next))
        self._t_prev = self._t
        self._t = t_end
        self._t_error = t_err

    def _adaptive_form(self) -> int:
        """0 = the persistent kernel where the library has one, 1 = per-stage launches.  The first adaptive step of a
        (topology, build variant) in the process checks the persistent kernel against the per-stage path on a probe
        batch (`_adaptive_self_test`: the per-stage path runs through the step kernels the library self-test covers)
        and falls back to it, with a warning, when they disagree.  JIMINY_AMD_ADAPTIVE_FORM forces a form."""
        env = os.environ.get("JIMINY_AMD_ADAPTIVE_FORM")
        if env is not None:
            return int(env)
        if getattr(self, "_adaptive_form_override", None) is not None:
            return self._adaptive_form_override
        if self.dtype != torch.float64 or codegen.quad_structure(self.model) is None or \
                self._options["contacts"]["model"] != "spring_damper" or os.environ.get("JIMINY_AMD_SELF_TEST", "1") == "0":
            return 0
        # (the variation form of the kernel -- per-lane body parameters / friction, height map, applied forces: jm_lib.cpp
        # `step_adaptive` -- is a compilation of its own and is checked on its own)
        gen = bool("model_lane" in self._fields or self._ground is not None or "applied" in self._fields or "friction" in self._fields)
        key = (self.model.topology_hash(), self._lib_variant_index, gen)
        if key not in _DOPRI_FORM:
            _DOPRI_FORM[key] = 0       # (the probes below are engines of the same topology)
            err = _adaptive_self_test(self.model, self._lib_variant_index, self.device, gen=gen)
            if not err <= 1e-6:
                _DOPRI_FORM[key] = 1
                warnings.warn(f"{self.model.name}: the persistent adaptive kernel{' (variation form)' if gen else ''} of build variant "
                              f"{key[1]} disagrees with the per-stage path on the probe batch ({err:.3e}): using the per-stage "
                              "launches (DESIGN.md section 4.7)")
        return _DOPRI_FORM[key]

    def step(self, step_dt: float = -1.0) -> None:
        """≙ `Engine::step(stepSize)` (reference engine.cc:1724-2417): fixed-step solvers advance
        all lanes with one launch per breakpoint interval; the adaptive solver iterates attempts on
        the device until every lane reached the breakpoint."""
        if not self._running:
            raise BadControlFlow("No simulation running. Please start one before using step "
                                 "method.")
        if self._adaptive is not None:
            self._step_adaptive(step_dt)
            return
        # the first step of a simulation opens with the reference's microsecond step (engine.cc:1176; `substep_sizes`)
        launches, t_end, t_err = plan_step(self._t, self._t_error, float(step_dt), self._options,
                                           self._force_breakpoints(self._t),
                                           dt_first=SIMULATION_MIN_TIMESTEP if self._opening_step else None)
        self._opening_step = False
        solver = SOLVER_IDS[self._options["stepper"]["odeSolver"]]
        stream = self._stream()
        # continuous sensor refresh (sensorsUpdatePeriod = 0) draws noise after every integrator step
        per_step_noise = bool(self._sensor_noise) and float(self._options["stepper"]["sensorsUpdatePeriod"]) <= 0.0
        # continuous profile forces (update_period = 0) are functions of (t, q, v): the reference evaluates them inside
        # every dynamics evaluation (computeExternalForces, engine.cc:3481-3492).  Here they are re-evaluated at the
        # start of every integrator step (launches cut to one step): piecewise constant over dtMax at most.
        per_step_forces = any(p["period"] <= EPS for p in self._profile_forces)
        # (only discrete controllers have breakpoints: engine.cc:1919-1940)
        constraint_model = (self._options["contacts"]["model"] == "constraint"
                            and float(self._options["stepper"]["controllerUpdatePeriod"]) > EPS)
        t_now = self._t
        for dt, n, cmd_bp, sens in launches:
            for k, n_k in enumerate([1] * n if ((per_step_noise and sens) or per_step_forces) else [n]):
                forces_changed = self._update_applied_forces(t_now)
                t_now += dt * n_k
                # a(t+) refresh at a controller breakpoint (engine.cc:2030-2042): skipped when the held
                # command was not rewritten (the evaluation is idempotent) -- except with the constraint
                # contact model, where the reference's refresh re-runs the warm-started PGS solve; and whenever
                # an applied force changed at the start of this launch
                changed = (cmd_bp and (self._command_dirty or constraint_model) and k == 0) or forces_changed
                self._lib.check(self._L.jm_batch_step(self._batch_h, solver, dt, n_k, int(changed),
                                                      int(sens), stream))
                if changed and cmd_bp:
                    self._command_dirty = False
                if sens and self._sensor_noise:
                    # continuous sensors: the reference also measures inside every dynamics evaluation of the step
                    # (engine.cc:3655-3667: the three Runge-Kutta stages + the derivative at the end of the step, or
                    # Euler's one, + the a(t+) refresh) -- those measurements are overwritten by the one after the
                    # step, but they draw from the noise streams: same number of rounds discarded here
                    extra = ((4 if solver == SOLVER_IDS["runge_kutta_4"] else 1) + (1 if changed else 0)) if per_step_noise else 0
                    self._apply_sensor_noise(discard_rounds=extra, t_now=t_now)
            self._iter += n
            self._dt = dt
        self._t_prev = self._t
        self._t = t_end
        self._t_error = t_err

    def simulate(self, t_end: float, q_init: Any, v_init: Any, a_init: Any = None,
                 callback: Optional[Any] = None) -> None:
        """≙ `Engine::simulate(tEnd, qInit, vInit, aInit, callback)` (reference engine.cc:1614-1699):
        reset, start, step by the stepper update period (else `dtMax`) until `t_end`, until
        `callback()` returns False or until `stepper.iterMax` integration steps, then stop."""
        if t_end < 5e-3:
            raise ValueError("Simulation duration cannot be shorter than 5ms.")   # engine.cc:1628-1631
        command = self._fields["command"].clone()
        self.reset()
        self._fields["command"].copy_(command)   # the held command stands for the controller
        self.start(q_init, v_init, a_init)
        st = self._options["stepper"]
        periods = [float(p) for p in (st["controllerUpdatePeriod"], st["sensorsUpdatePeriod"]) if float(p) > EPS]
        period = min(periods) if periods else float(st["dtMax"])
        iter_max = int(st.get("iterMax", 0))
        while True:
            if t_end - self._t < SIMULATION_MIN_TIMESTEP:
                break
            if callback is not None and not callback():
                break
            if 0 < iter_max <= self.stepper_state.iter:
                break
            self.step(min(period, t_end - self._t))
        self.stop()

    def compute_robots_dynamics(self, t: float, q: Any, v: Any) -> torch.Tensor:
        """≙ `Engine.compute_robots_dynamics(t, [q], [v]) -> [a]` (pywrap engine.cc:634-638)."""
        if not self._running:
            raise BadControlFlow("No simulation running. Please start one before calling this "
                                 "method.")
        m = self.model
        qs = self._to_soa(q, m.nq, "q").contiguous()
        vs = self._to_soa(v, m.nv, "v").contiguous()
        a = torch.empty((m.nv, self.batch_size), dtype=self.dtype, device=self.device)
        self._lib.check(self._L.jm_batch_dynamics(self._batch_h, C.c_void_p(qs.data_ptr()),
                                                  C.c_void_p(vs.data_ptr()),
                                                  C.c_void_p(a.data_ptr()), self._stream()))
        return a

    def reset_lanes(self, lane_mask: torch.Tensor, q_init: Any, v_init: Any) -> None:
        """Re-initialise the masked lanes from (q_init, v_init) and recompute their
        acceleration / sensors (per-lane `reset` + `start`)."""
        m = self.model
        mask = lane_mask.to(device=self.device, dtype=torch.uint8).contiguous()
        if tuple(mask.shape) != (self.batch_size,):
            raise ValueError("lane_mask must have shape (B,)")
        q = self._to_soa(q_init, m.nq, "q_init").contiguous()
        v = self._to_soa(v_init, m.nv, "v_init").contiguous()
        self._lib.check(self._L.jm_batch_reset_lanes(
            self._batch_h, C.c_void_p(mask.data_ptr()), C.c_void_p(q.data_ptr()),
            C.c_void_p(v.data_ptr()), self._stream()))
        if self._adaptive is not None:
            m_ = mask.bool()
            ad = self._adaptive
            ad["f64"][1:4, m_] = SIMULATION_MIN_TIMESTEP
            ad["f64"][0, m_] = self._t
            ad["i32"][:, m_] = 0
        # a re-initialised lane starts with an empty sensor history (`AbstractSensorTpl::resetAll`): every
        # stored sample of that lane becomes its fresh raw measurement, which is what the delay lookup returns
        # while "the buffer is not fully initialised" (abstract_sensor.hxx:405-421)
        for e in self._sensor_noise.values():
            if e.get("hist") is not None:
                m_ = mask.bool()
                e["hist"][:, :, m_] = self._fields[e["field"]][:, m_].unsqueeze(0)

    # ------------------------------------------------------------------ per-environment variation
    def get_model_options(self) -> Dict[str, Dict[str, float]]:
        """≙ `robot.get_model_options()`: the `dynamics` bias options (model.h:147-158)."""
        return {k: dict(v) for k, v in self._model_options.items()}

    def set_model_options(self, options: Dict[str, Dict[str, float]]) -> None:
        """≙ `robot.set_model_options({"dynamics": {"massBodiesBiasStd": ..., ...}})`: standard deviations of the
        body mass / centre of mass / inertia / relative position biases.  A biased copy of the model is drawn for
        every lane at each `start` (`Model::reset` -> `addBiasedToExtendedModel`, model.cc:1166-1236)."""
        if self._running:
            raise BadControlFlow("Robot already locked, probably because a simulation is running.")
        for k, v in options.get("dynamics", {}).items():
            if k in self._model_options["dynamics"]:
                if float(v) < 0.0:
                    raise ValueError(f"'{k}' must be non-negative")
                self._model_options["dynamics"][k] = float(v)
        if not any(v > EPS for v in self._model_options["dynamics"].values()):
            self.set_lane_model(None)

    def seed_model(self, seed: Any) -> None:
        """Seed the engine generator of every lane, ≙ `stepper.randomSeedSeq = [seed]` of one reference engine per lane
        (`generator_.seed(std::seed_seq(...))`, engine.cc:756-757): the PCG32 stream the model biases are drawn from
        (`Model::addBiasedToExtendedModel`).  `seed`: `(B,)` uint32 words, or an `int` -- lane l gets `seed + l`."""
        B = self.batch_size
        if isinstance(seed, (int, np.integer)):
            # (Python integers: a negative seed wraps modulo 2^32 instead of overflowing a uint64 array)
            words = np.array([(int(seed) + l) & 0xFFFFFFFF for l in range(B)], dtype=np.uint32)
        else:
            words = np.ascontiguousarray(np.broadcast_to(np.asarray(seed, dtype=np.uint32), (B,)))
        out = np.empty(B, dtype=np.uint64)
        self._lib.check(self._L.jm_engine_rng_seed(words.ctypes.data_as(C.POINTER(C.c_uint32)), B,
                                                   out.ctypes.data_as(C.POINTER(C.c_uint64))))
        self._model_rng = torch.from_numpy(out.view(np.int64)).to(self.device)

    @property
    def model_rng_state(self) -> torch.Tensor:
        """PCG32 state of every lane's engine generator (`(B,)`, int64 view of the uint64 states)."""
        return self._model_rng

    def set_lane_model(self, model_lane: Optional[torch.Tensor]) -> None:
        """Bind body parameters per lane (`[13 * njoints][B]`, layout of `JM_F_MODEL_LANE`), None = the model's."""
        if model_lane is None:
            if "model_lane" in self._fields:
                self._fields.pop("model_lane")
                self._lib.check(self._L.jm_batch_bind(self._batch_h, _abi.FIELD_NAMES["model_lane"], None))
            return
        rows = 13 * self.model.njoints
        t = torch.as_tensor(model_lane, dtype=self.dtype, device=self.device)
        if tuple(t.shape) != (rows, self.batch_size):
            raise ValueError(f"model_lane must have shape ({rows}, B)")
        if "model_lane" not in self._fields:
            self._fields["model_lane"] = torch.empty((rows, self.batch_size), dtype=self.dtype, device=self.device)
            self._bind("model_lane")
        self._fields["model_lane"].copy_(t)

    def sample_model_biases(self, lane_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Draw the biased models (all lanes, or the masked ones: episode-wise re-randomisation) on the device and bind
        them: `jm_block_model_bias` ≙ `Model::addBiasedToExtendedModel` per lane, from the lane's engine generator
        (`seed_model`), whose stream advances exactly like the reference's."""
        from .randomization import DYNAMICS_OPTION_NAMES, nominal_bias_table, nominal_model_lane
        B = self.batch_size
        if "model_lane" not in self._fields:
            self.set_lane_model(nominal_model_lane(self.model, B, self.dtype, self.device))
        if self._model_rng is None:
            self.seed_model(0)
        if self._bias_table is None:
            self._bias_table = torch.from_numpy(nominal_bias_table(self.model)).to(self.device)
        std4 = (C.c_float * 4)(*[float(self._model_options["dynamics"][k]) for k in DYNAMICS_OPTION_NAMES])
        mask = None
        if lane_mask is not None:
            mask = lane_mask.to(device=self.device, dtype=torch.uint8).contiguous()
        first = 2 if self.model.has_freeflyer else 1   # mechanicalJointNames_ excludes the free-flyer (model.cc:337-341)
        self._lib.check(self._L.jm_block_model_bias(
            _abi.JM_F64 if self.dtype == torch.float64 else _abi.JM_F32, B, self.model.njoints, first,
            C.c_void_p(self._bias_table.data_ptr()), std4, C.c_void_p(self._model_rng.data_ptr()),
            C.c_void_p(mask.data_ptr()) if mask is not None else None,
            C.c_void_p(self._fields["model_lane"].data_ptr()), self._stream()))
        return self._fields["model_lane"]

    def set_ground_heightmap(self, heights: Any, x0: float = 0.0, y0: float = 0.0, dx: float = 1.0, dy: float = 1.0) -> None:
        """≙ `engine_options["world"]["groundProfile"]` (engine.h:292-302) as a height map: `heights[iy][ix]` at
        (x0 + ix dx, y0 + iy dy), bilinear patches (height + unit normal), flat continuation outside; None = flat
        ground.  Both contact models: with `contacts.model = "constraint"` the rows of a contact constraint live in the
        local frame of the surface under the contact point (`FrameConstraint::setNormal`, engine.cc:3184-3193)."""
        if self._running:
            raise BadControlFlow("Please stop the simulation before updating the options.")
        self._ground_pool = None
        if heights is None:
            self._ground = None
            self._lib.check(self._L.jm_batch_set_ground(self._batch_h, None, 0, 0, 0.0, 0.0, 1.0, 1.0))
            return
        h = torch.as_tensor(heights, dtype=self.dtype, device=self.device).contiguous()
        if h.dim() != 2 or min(h.shape) < 2:
            raise ValueError("the height map needs at least 2 x 2 samples")
        self._ground = h
        self._ground_grid = (float(x0), float(y0), float(dx), float(dy))
        self._lib.check(self._L.jm_batch_set_ground(self._batch_h, C.c_void_p(h.data_ptr()), int(h.shape[1]), int(h.shape[0]),
                                                    float(x0), float(y0), float(dx), float(dy)))

    def ground_height_around(self, xy: torch.Tensor, radius: float) -> torch.Tensor:
        """Largest height of the bound height map within `radius` of the world points `xy` (`(2, N)`, device tensor):
        where to put a robot down on a patch of terrain without burying its feet (flat continuation outside the map)."""
        if self._ground is None:
            return torch.zeros(xy.shape[1], dtype=self.dtype, device=self.device)
        x0, y0, dx, dy = self._ground_grid
        H = self._ground
        ny, nx = H.shape
        # (+ 1 cell: the query is rounded to the nearest node, i.e. it can sit half a cell away from the point)
        kx, ky = int(math.ceil(radius / dx)) + 1, int(math.ceil(radius / dy)) + 1
        cache = getattr(self, "_ground_pool", None)
        if cache is None or cache[0] != (kx, ky):
            # pooled once per height map and radius (invalidated by set_ground_heightmap): it is asked for at every auto-reset
            pooled = torch.nn.functional.max_pool2d(H[None, None].to(torch.float64), (2 * ky + 1, 2 * kx + 1), stride=1,
                                                    padding=(ky, kx))[0, 0]
            cache = self._ground_pool = ((kx, ky), pooled)
        Hmax = cache[1]
        ix = torch.clamp(torch.round((xy[0] - x0) / dx).long(), 0, nx - 1)
        iy = torch.clamp(torch.round((xy[1] - y0) / dy).long(), 0, ny - 1)
        return Hmax[iy, ix].to(self.dtype)

    def set_ground_offsets(self, offsets: Optional[Any]) -> None:
        """Every lane samples the height map at its own (x, y) offset: `(B, 2)` (or `(2, B)`) values added to the world
        position of the height-map queries, None = no offsets.  One large terrain, every environment its own patch of
        it -- the batched form of one `world.groundProfile` per environment instance.  May be changed while the
        simulation runs (the environments re-draw the offsets of the lanes they reset)."""
        if offsets is None:
            self._fields.pop("ground_offset", None)
            self._lib.check(self._L.jm_batch_bind(self._batch_h, _abi.FIELD_NAMES["ground_offset"], None))
            return
        o = torch.as_tensor(offsets, dtype=self.dtype, device=self.device)
        if o.shape == (self.batch_size, 2):
            o = o.t()
        if o.shape != (2, self.batch_size):
            raise ValueError("ground offsets: one (x, y) pair per lane")
        if "ground_offset" not in self._fields:
            self._fields["ground_offset"] = torch.zeros((2, self.batch_size), dtype=self.dtype, device=self.device)
            self._bind("ground_offset")
        self._fields["ground_offset"].copy_(o)

    def set_ground_profile(self, func: Any, x_range: Tuple[float, float], y_range: Tuple[float, float],
                           resolution: float) -> None:
        """Discretise a `heightmap(x, y) -> height` callable on a regular grid (≙ `discretize_heightmap`,
        core/src/utilities/geometry.cc) and use it as the ground profile.  The callable is tried on device tensors
        first (e.g. `jiminy_amd.terrain.random_tile_ground`: the map is then generated on the GPU), then on numpy arrays."""
        nx = max(int(math.ceil((x_range[1] - x_range[0]) / resolution)) + 1, 2)
        ny = max(int(math.ceil((y_range[1] - y_range[0]) / resolution)) + 1, 2)
        xs = x_range[0] + resolution * torch.arange(nx, dtype=torch.float64, device=self.device)
        ys = y_range[0] + resolution * torch.arange(ny, dtype=torch.float64, device=self.device)
        Y, X = torch.meshgrid(ys, xs, indexing="ij")
        try:
            h = func(X, Y)
            if not isinstance(h, torch.Tensor):
                raise TypeError
        except (TypeError, RuntimeError):
            h = np.asarray(func(X.cpu().numpy(), Y.cpu().numpy()), dtype=np.float64)
        self.set_ground_heightmap(h, x_range[0], y_range[0], resolution, resolution)

    def _force_frame_index(self, frame_name: str) -> int:
        """Slot of the frame among the (at most four) frames that carry applied wrenches.  Any frame of the model: the
        wrench goes to the frame's parent joint, like `Engine::computeExternalForces` (engine.cc:3481-3560)."""
        fr = self.model.frame(frame_name)
        if self.dtype != torch.float64 or (codegen.quad_structure(self.model) is not None and
                                           os.environ.get("JM_KERNEL_VARIANT") == "lane"):
            raise NotImplementedError("external forces need a float64 batch (the kernels that read them are float64 "
                                      "instantiations of their own in either kernel family)")
        if frame_name not in self._force_frames:
            if len(self._force_frames) == 4:
                raise NotImplementedError("at most 4 frames can carry external forces")
            self._force_frames.append(frame_name)
            offs = np.ascontiguousarray([self.model.frame(n).p for n in self._force_frames], dtype=np.float64)
            joints = np.ascontiguousarray([self.model.frame(n).parent_joint for n in self._force_frames], dtype=np.int32)
            self._lib.check(self._L.jm_batch_set_applied_frames(self._batch_h, len(self._force_frames),
                                                                offs.ctypes.data_as(C.POINTER(C.c_double)),
                                                                joints.ctypes.data_as(C.POINTER(C.c_int32))))
            old = self._fields.get("applied")
            new = torch.zeros((6 * len(self._force_frames), self.batch_size), dtype=self.dtype, device=self.device)
            if old is not None:
                new[: old.shape[0]] = old
            self._fields["applied"] = new
            self._bind("applied")
        return self._force_frames.index(frame_name)

    def register_impulse_force(self, frame_name: str, t: float, dt: float, force: Any) -> None:
        """≙ `Engine.register_impulse_force(robot_name, frame_name, t, dt, force)` (engine.cc:1838-1893): the
        world-aligned wrench `force` (6,) -- or one per lane, (6, B) -- acts on the frame during [t, t + dt]."""
        if self._running:
            raise BadControlFlow("Simulation already running. Please stop it before registering new forces.")
        if dt < STEPPER_MIN_TIMESTEP:
            raise ValueError("Force duration cannot be smaller than 1e-10 s.")   # engine.cc:1854-1858
        if t < 0.0:
            raise ValueError("Force application time must be positive.")        # engine.cc:1860-1864
        f = torch.as_tensor(force, dtype=self.dtype, device=self.device)
        if f.dim() == 1:
            f = f[:, None].expand(6, self.batch_size)
        if tuple(f.shape) != (6, self.batch_size):
            raise ValueError("force must have shape (6,) or (6, B)")
        self._impulse_forces.append({"frame": self._force_frame_index(frame_name), "t": float(t), "dt": float(dt),
                                     "force": f.contiguous()})

    def _schedule_impulse_force(self, frame_name: str, t: float, dt: float, force: torch.Tensor) -> None:
        """Internal (environments): append an impulse while the simulation runs -- its start must lie at or after the
        current time, so that the breakpoint schedule of the steps to come sees it -- and drop the impulses that are
        over.  The vectorised environments keep only the NEXT push of their periodic schedule registered instead of
        one (6, B) tensor per push of the whole horizon."""
        if t < self._t - STEPPER_MIN_TIMESTEP:
            raise ValueError("cannot schedule an impulse in the past")
        alive = [i for i, f in enumerate(self._impulse_forces) if f["t"] + f["dt"] > self._t - STEPPER_MIN_TIMESTEP]
        remap = {old: new for new, old in enumerate(alive)}
        self._impulse_forces = [self._impulse_forces[i] for i in alive]
        self._impulse_active = [remap[i] for i in self._impulse_active if i in remap]
        self._impulse_forces.append({"frame": self._force_frames.index(frame_name), "t": float(t), "dt": float(dt),
                                     "force": force.to(dtype=self.dtype, device=self.device).contiguous()})

    def register_profile_force(self, frame_name: str, func: Any, update_period: float = 0.0) -> None:
        """≙ `Engine.register_profile_force(robot_name, frame_name, force_func, update_period)` (engine.cc:1895-1935):
        `func(t, q, v) -> wrench` with `q`, `v` the `[rows][B]` state tensors and the result `(6,)` or `(6, B)`; it is
        evaluated at every launch (update_period = 0) or held between multiples of `update_period`."""
        if self._running:
            raise BadControlFlow("Simulation already running. Please stop it before registering new forces.")
        if EPS < update_period < SIMULATION_MIN_TIMESTEP:
            raise ValueError("Cannot register external force profile with update period smaller than 1us.")
        self._profile_forces.append({"frame": self._force_frame_index(frame_name), "func": func,
                                     "period": float(update_period), "t_last": -math.inf, "value": None})

    def remove_all_forces(self) -> None:
        """≙ `Engine.remove_all_forces` (engine.cc:1937-1960)."""
        if self._running:
            raise BadControlFlow("Simulation already running. Please stop it before removing forces.")
        self._impulse_forces.clear()
        self._impulse_active = []
        self._profile_forces.clear()
        self._force_frames.clear()
        self._fields.pop("applied", None)
        self._lib.check(self._L.jm_batch_set_applied_frames(self._batch_h, 0, None, None))
        self._lib.check(self._L.jm_batch_bind(self._batch_h, _abi.FIELD_NAMES["applied"], None))

    @property
    def impulse_forces(self) -> List[Dict[str, Any]]:
        return [{"frame_name": self._force_frames[f["frame"]], "t": f["t"], "dt": f["dt"], "force": f["force"]}
                for f in self._impulse_forces]

    def _force_breakpoints(self, t: float) -> Tuple[float, ...]:
        """Times after `t` at which an applied force changes: impulse starts / ends, profile refreshes."""
        pts = []
        for f in self._impulse_forces:
            pts += [f["t"], f["t"] + f["dt"]]
        for p in self._profile_forces:
            if p["period"] > EPS:
                pts.append((math.floor(t / p["period"] + 1e-9) + 1) * p["period"])
        return tuple(sorted(x for x in pts if x > t + STEPPER_MIN_TIMESTEP))

    def _update_applied_forces(self, t: float) -> bool:
        """Current value of the registered forces into the `applied` field (start of a launch at time `t`).  Returns
        True when the held wrenches changed -- an impulse started or ended, a profile was re-evaluated -- i.e. when the
        reference sets `hasDynamicsChanged` and recomputes a(t+) before stepping on (engine.cc:1860-1868, 1909-1912,
        1972-1983, 2031-2042)."""
        if "applied" not in self._fields:
            return False
        a = self._fields["applied"]
        a.zero_()
        active = []
        changed = False
        for i, f in enumerate(self._impulse_forces):
            if impulse_active(f["t"], f["dt"], t, i in self._impulse_active):
                a[6 * f["frame"]:6 * f["frame"] + 6] += f["force"]
                active.append(i)
        if active != self._impulse_active:
            self._impulse_active = active
            changed = True
        for p in self._profile_forces:
            if p["value"] is None or p["period"] <= EPS or update_due(p["period"], t):
                w = torch.as_tensor(p["func"](t, self._fields["q"], self._fields["v"]), dtype=self.dtype, device=self.device)
                p["value"] = w[:, None].expand(6, self.batch_size) if w.dim() == 1 else w
                p["t_last"] = t if p["period"] <= EPS else math.floor(t / p["period"] + 1e-9) * p["period"]
                changed = True
            a[6 * p["frame"]:6 * p["frame"] + 6] += p["value"]
        return changed

    # ------------------------------------------------------------------ sensor noise and bias
    _SENSOR_FIELDS = {"ImuSensor": ("imu", 6), "ForceSensor": ("force", 6), "ContactSensor": ("contact", 3),
                      "EncoderSensor": ("encoder", 2), "EffortSensor": ("effort", 1)}

    def set_sensor_options(self, sensor_type: str, noise_std: Any = None, bias: Any = None, delay: Any = None,
                           jitter: Any = None, delay_interpolation_order: int = 0) -> None:
        """≙ `sensor.set_options({"noiseStd": ..., "bias": ..., "delay": ..., "jitter": ...,
        "delayInterpolationOrder": ...})` for every sensor of one type
        (reference abstract_sensor.h:66-100): white noise and bias applied to the raw measurement
        after every sensor refresh (`AbstractSensorBase::measureData`, abstract_sensor.cc:71-85;
        `ImuSensor::measureData`, basic_sensors.cc:166-187).  `noise_std` is `(n_fields,)` or
        `(n_sensors, n_fields)`; `bias` likewise, except for IMUs where it has 9 entries: a rotation
        bias (angle-axis) followed by the gyroscope and accelerometer biases.  `None` leaves the
        option empty.  `delay` / `jitter` (seconds, scalar or `(n_sensors,)`): the measurement is read
        `delay + uniform(0, jitter)` in the past from the history of raw measurements, zero-order hold or
        linear interpolation (`AbstractSensorTpl::interpolateData`, abstract_sensor.hxx:305-429); needs a
        positive `sensorsUpdatePeriod`.  Not allowed while a simulation is running (abstract_sensor.cc:87-96)."""
        if self._running:
            raise BadControlFlow("Robot already locked, probably because a simulation is running. "
                                 "Please stop it before setting sensor options.")
        if sensor_type not in self._SENSOR_FIELDS:
            raise LookupError(f"unknown sensor type '{sensor_type}'")
        field, nf = self._SENSOR_FIELDS[sensor_type]
        n = len(self.model.sensors.get(sensor_type, []))
        if n == 0:
            raise LookupError(f"the robot has no sensor of type '{sensor_type}'")
        nb = 9 if sensor_type == "ImuSensor" else nf

        def table(x, cols, what):
            if x is None:
                return None
            a = np.asarray(x, dtype=np.float64)
            if a.ndim == 1:
                a = np.tile(a[None, :], (n, 1))
            if a.shape != (n, cols):
                raise ValueError(f"{what} must have shape ({cols},) or ({n}, {cols})")
            return np.ascontiguousarray(a)
        std, b = table(noise_std, nf, "noise_std"), table(bias, nb, "bias")

        def per_sensor(x, what):
            a = np.broadcast_to(np.asarray(0.0 if x is None else x, dtype=np.float64), (n,)).copy()
            if np.any(a < 0.0):
                raise ValueError(f"{what} must be non-negative")
            return np.ascontiguousarray(a)
        dly, jit = per_sensor(delay, "delay"), per_sensor(jitter, "jitter")
        has_delay = bool(np.any(dly > EPS) or np.any(jit > EPS))
        if int(delay_interpolation_order) not in (0, 1):
            raise NotImplementedError("`delayInterpolationOrder` must be either 0 or 1.")  # abstract_sensor.hxx:399-403
        if std is None and b is None and not has_delay:
            self._sensor_noise.pop(sensor_type, None)
            return
        if std is not None and np.any(std < 0.0):
            raise ValueError("noise_std must be non-negative")
        entry: Dict[str, Any] = {"field": field, "n": n, "nf": nf, "std": std, "bias": None, "rot": None,
                                 "rng": self._sensor_noise.get(sensor_type, {}).get("rng"),
                                 "delay": dly if has_delay else None, "jitter": jit if has_delay else None,
                                 "order": int(delay_interpolation_order), "hist": None, "samples": []}
        if b is not None:
            if sensor_type == "ImuSensor":
                # sensorRotationBiasInv_ = exp3(-bias.head<3>()) (basic_sensors.cc:121-129)
                entry["rot"] = np.ascontiguousarray(np.stack([_exp3(-b[i, :3]).reshape(9) for i in range(n)]))
                entry["bias"] = np.ascontiguousarray(b[:, 3:])
            else:
                entry["bias"] = b
        self._sensor_noise[sensor_type] = entry

    def _apply_hardware_sensor_options(self) -> None:
        """Measurement options that came with the robot's hardware description file (`noiseStd`, `bias`,
        `delay`, `jitter`, `delayInterpolationOrder` per sensor; model.load_hardware_description_file)."""
        for stype, recs in self.model.sensors.items():
            if stype not in self._SENSOR_FIELDS or not any("options" in r for r in recs):
                continue
            nf = self._SENSOR_FIELDS[stype][1]
            nb = 9 if stype == "ImuSensor" else nf

            def table(key, cols):
                rows = [np.asarray(r.get("options", {}).get(key, np.zeros(cols)), dtype=np.float64) for r in recs]
                rows = [np.broadcast_to(x, (cols,)) if x.size in (1, cols) else x for x in rows]
                t = np.stack(rows)
                return t if np.any(t != 0.0) else None
            scal = lambda key: np.array([float(np.asarray(r.get("options", {}).get(key, 0.0)).reshape(-1)[0])  # noqa: E731
                                         for r in recs])
            orders = {int(r["options"].get("delayInterpolationOrder", 0)) for r in recs if "options" in r}
            self.set_sensor_options(stype, noise_std=table("noiseStd", nf), bias=table("bias", nb),
                                    delay=scal("delay"), jitter=scal("jitter"),
                                    delay_interpolation_order=max(orders) if orders else 0)

    def seed_sensors(self, seeds: Any) -> None:
        """Seed the per-(sensor, lane) PCG32 generators, ≙ `AbstractSensorTpl::resetAll(seed)`
        (abstract_sensor.hxx:213-226) for every lane.  `seeds` maps a sensor type to the `(B,)`
        uint32 group seeds the reference would draw with `g()` in `Robot::reset` (robot.cc:135-141);
        an `int` derives them as `seed + lane * n_types + index(type)` (types in sorted order; the
        reference iterates an unordered_map there, so it defines no order to mirror)."""
        types = sorted(self._sensor_noise)
        B = self.batch_size
        for g, stype in enumerate(types):
            e = self._sensor_noise[stype]
            if isinstance(seeds, dict):
                if stype not in seeds:
                    continue
                gs = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds[stype], dtype=np.uint32), (B,)))
            else:
                gs = ((int(seeds) + np.arange(B, dtype=np.uint64) * len(types) + g) & 0xFFFFFFFF).astype(np.uint32)
            out = np.empty((e["n"], B), dtype=np.uint64)
            self._lib.check(self._L.jm_sensor_rng_seed(
                gs.ctypes.data_as(C.POINTER(C.c_uint32)), B, e["n"], out.ctypes.data_as(C.POINTER(C.c_uint64))))
            e["rng"] = torch.from_numpy(out.view(np.int64)).to(self.device)

    def _reset_sensor_history(self) -> None:
        """`AbstractSensorTpl::resetAll`: empty history (abstract_sensor.hxx:196-199).  The ring holds the raw
        measurements of the last `delayMax + SIMULATION_MAX_TIMESTEP` seconds, like the reference's buffer
        (abstract_sensor.hxx:456-461), at one sample per sensor refresh."""
        period = float(self._options["stepper"]["sensorsUpdatePeriod"])
        for stype, e in self._sensor_noise.items():
            e["samples"] = []
            e["hist"] = None
            if e["delay"] is None:
                continue
            if period <= 0.0:
                raise NotImplementedError(f"{stype}: sensor delay needs a positive sensorsUpdatePeriod")
            delay_max = float((e["delay"] + e["jitter"]).max())
            slots = int(math.ceil((delay_max + SIMULATION_MAX_TIMESTEP) / period)) + 3
            if slots > 64:
                raise NotImplementedError(f"{stype}: the delay spans more than 64 sensor periods")
            rows = max(self._rows[e["field"]], 1)
            e["hist"] = torch.zeros((slots, rows, self.batch_size), dtype=self.dtype, device=self.device)

    def _apply_sensor_noise(self, discard_rounds: int = 0, t_now: float = 0.0) -> None:
        """`AbstractSensorTpl::measureDataAll` (abstract_sensor.hxx:431-443) on the raw measurements the
        physics launch just wrote: delay (history lookup + the jitter draw), then white noise and bias."""
        dp = C.POINTER(C.c_double)
        dtype = _abi.JM_F64 if self.dtype == torch.float64 else _abi.JM_F32
        for stype, e in self._sensor_noise.items():
            if e["std"] is not None and e["rng"] is None:
                raise BadControlFlow(f"{stype}: noise is enabled but the generators were never seeded "
                                     "(call seed_sensors first)")
            if e["delay"] is not None and float(e["jitter"].max()) > EPS and e["rng"] is None:
                raise BadControlFlow(f"{stype}: jitter is enabled but the generators were never seeded "
                                     "(call seed_sensors first)")
            ptr = lambda a: a.ctypes.data_as(dp) if a is not None else None  # noqa: E731
            rng = C.c_void_p(e["rng"].data_ptr()) if e["rng"] is not None else None
            field = self._fields[e["field"]]
            hist, slot_p, time_p, n_hist = None, None, None, 0
            if e["hist"] is not None:
                # store the raw measurement of this refresh (several refreshes at one time, as in `start`,
                # hold the same raw data: one sample)
                samples = e["samples"]
                if samples and abs(samples[-1][1] - t_now) <= EPS:
                    slot = samples[-1][0]
                else:
                    n_slots = e["hist"].shape[0]
                    used = {sl for sl, _ in samples}
                    if len(samples) == n_slots:
                        slot = samples.pop(0)[0]
                    else:
                        slot = next(i for i in range(n_slots) if i not in used)
                    samples.append((slot, float(t_now)))
                e["hist"][slot].copy_(field)
                slots = np.ascontiguousarray([sl for sl, _ in samples], dtype=np.int32)
                times = np.ascontiguousarray([tm for _, tm in samples], dtype=np.float64)
                hist = C.c_void_p(e["hist"].data_ptr())
                slot_p, time_p, n_hist = slots.ctypes.data_as(C.POINTER(C.c_int32)), times.ctypes.data_as(dp), len(samples)
            scratch = torch.zeros_like(field) if discard_rounds else None
            for rnd in range(discard_rounds + 1):
                last = rnd == discard_rounds
                target = C.c_void_p((field if last else scratch).data_ptr())
                # interpolateData: one uniform draw per sensor and round, jitter or not (abstract_sensor.hxx:316-318)
                if hist is not None or rng is not None:
                    self._lib.check(self._L.jm_block_sensor_delay(
                        dtype, self.batch_size, e["n"], e["nf"], target, hist, slot_p, time_p, n_hist, rng,
                        ptr(e["delay"]), ptr(e["jitter"]), e["order"], self._stream()))
                if e["std"] is not None or (last and e["bias"] is not None):
                    self._lib.check(self._L.jm_block_sensor_noise(
                        dtype, self.batch_size, e["n"], e["nf"], target, rng, ptr(e["std"]),
                        ptr(e["bias"]) if last else None, ptr(e["rot"]) if last else None, self._stream()))

    # ------------------------------------------------------------------ measurement helpers
    def enable_timing(self, enable: bool = True) -> None:
        self._lib.check(self._L.jm_batch_enable_timing(self._batch_h, int(enable)))

    def timing_summary(self) -> Tuple[int, float]:
        """(number of kernel launches, summed kernel time in ms) since the last call; HIP events
        recorded on the launch stream around each launch."""
        n, ms = C.c_int32(), C.c_double()
        self._lib.check(self._L.jm_batch_timing_summary(self._batch_h, C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)
