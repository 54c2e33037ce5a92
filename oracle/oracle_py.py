"""ctypes wrapper around liboracle.so (oracle.cpp). TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

from jiminy_amd import _abi
from jiminy_amd.model import CompiledModel

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[C.CDLL] = None


def build(force: bool = False) -> str:
    path = os.path.join(_HERE, "liboracle.so")
    srcs = (os.path.join(_HERE, "oracle.cpp"), os.path.join(_HERE, "oracle_random.cpp"),
            os.path.join(_HERE, "..", "include", "jiminy_hip.h"))
    stale = (not os.path.exists(path)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(path) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return path


class BatchIO(C.Structure):
    _fields_ = [("B", C.c_int64)] + [(n, C.c_void_p) for n in (
        "q", "v", "a", "command", "u_motor", "imu", "force", "contact", "encoder", "effort",
        "energy", "contact_forces", "f_external", "status", "joint_forces", "centroidal", "u")]


class AdaptiveIO(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("t", "dt", "dt_largest", "dt_largest_prev", "iter",
                                          "iter_failed", "succ_too_large", "succ_failed")]


def adaptive_state(B: int) -> Dict[str, np.ndarray]:
    """Per-lane stepper state of the adaptive (Dormand-Prince) solver, as `StepperState::reset`
    leaves it (engine.h:219-236, dtInit = SIMULATION_MIN_TIMESTEP, engine.cc:1176)."""
    st = {k: np.full(B, 1e-6) for k in ("dt", "dt_largest", "dt_largest_prev")}
    st["t"] = np.zeros(B)
    for k in ("iter", "iter_failed", "succ_too_large", "succ_failed"):
        st[k] = np.zeros(B, dtype=np.int64)
    return st


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_engine_create.restype = C.c_void_p
        L.orc_engine_create.argtypes = [C.POINTER(_abi.ModelDesc), C.POINTER(_abi.Options)]
        L.orc_engine_destroy.argtypes = [C.c_void_p]
        L.orc_engine_set_options.argtypes = [C.c_void_p, C.POINTER(_abi.Options)]
        L.orc_engine_set_constraint_options.argtypes = [C.c_void_p, C.POINTER(_abi.ConstraintOptions)]
        L.orc_engine_bind_constraints.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_engine_bind_friction.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_engine_bind_friction.restype = None
        L.orc_engine_bind_flexibility.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_engine_bind_flexibility.restype = None
        L.orc_engine_bind_ground_offset.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_engine_bind_ground_offset.restype = None
        L.orc_engine_bind_model_lane.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_engine_bind_model_lane.restype = None
        L.orc_engine_bind_ground.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]
        L.orc_engine_bind_ground.restype = None
        L.orc_engine_bind_applied.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_engine_bind_applied.restype = None
        L.orc_engine_constraint_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_engine_constraint_counts.restype = C.c_int
        pd = C.POINTER(C.c_double)
        L.orc_engine_set_state.argtypes = [C.c_void_p, pd, pd, pd]
        L.orc_engine_set_command.argtypes = [C.c_void_p, pd]
        L.orc_engine_start.argtypes = [C.c_void_p]
        L.orc_engine_step.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int]
        L.orc_engine_dynamics.argtypes = [C.c_void_p, pd, pd, pd]
        L.orc_engine_status.argtypes = [C.c_void_p]
        L.orc_engine_status.restype = C.c_int
        L.orc_engine_get.argtypes = [C.c_void_p, C.c_int, pd]
        L.orc_engine_get.restype = C.c_int
        L.orc_engine_joint_placement.argtypes = [C.c_void_p, C.c_int, pd]
        L.orc_integrate.argtypes = [C.c_void_p, pd, pd, pd]
        L.orc_batch_run.argtypes = [C.c_void_p, C.POINTER(BatchIO), C.c_int, C.c_int, C.c_double,
                                    C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64]
        u32p, u64p, fp = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_float)
        L.orc_pcg32_stream.argtypes = [u64p, C.c_int64, u32p]
        L.orc_uniform_stream.argtypes = [u64p, C.c_int64, fp]
        L.orc_normal_stream.argtypes = [u64p, C.c_int64, fp]
        L.orc_ziggurat_tables.argtypes = [u32p, fp, fp]
        L.orc_seed_seq.argtypes = [C.c_uint32, C.c_int32, u32p]
        L.orc_sensor_rng_seed.argtypes = [u32p, C.c_int64, C.c_int32, u64p]
        L.orc_engine_rng_seed.argtypes = [u32p, C.c_int64, u64p]
        L.orc_model_bias.argtypes = [C.c_int64, C.c_int32, C.c_int32, pd, C.POINTER(C.c_float), u64p, C.c_void_p, pd]
        L.orc_sensor_noise.argtypes = [C.c_int64, C.c_int32, C.c_int32, pd, u64p, pd, pd, pd]
        L.orc_sensor_delay.argtypes = [C.c_int64, C.c_int32, C.c_int32, pd, pd, C.POINTER(C.c_int32), pd, C.c_int32,
                                       u64p, pd, pd, C.c_int32]
        L.orc_sensor_delay.restype = None
        for f in (L.orc_pcg32_stream, L.orc_uniform_stream, L.orc_normal_stream, L.orc_ziggurat_tables,
                  L.orc_seed_seq, L.orc_sensor_rng_seed, L.orc_sensor_noise):
            f.restype = None
        L.orc_batch_run_dopri.argtypes = [C.c_void_p, C.POINTER(BatchIO), C.POINTER(AdaptiveIO), C.c_double,
                                          C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int64, C.c_int64]
        _LIB = L
    return _LIB


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


SOLVERS = {"euler_explicit": _abi.JM_SOLVER_EULER_EXPLICIT,
           "runge_kutta_4": _abi.JM_SOLVER_RUNGE_KUTTA_4}


class OracleEngine:
    """One robot, one thread: the shape of the reference `jiminy.Engine` for this path."""

    def __init__(self, model: CompiledModel, **options) -> None:
        self.model = model
        self._desc, self._keep = _abi.make_model_desc(model)
        self.options = _abi.make_options(**options)
        self._L = lib()
        self._h = C.c_void_p(self._L.orc_engine_create(C.byref(self._desc), C.byref(self.options)))
        self._rows = _abi.field_rows(model)
        self.t = 0.0

    def __del__(self) -> None:
        if getattr(self, "_h", None):
            self._L.orc_engine_destroy(self._h)
            self._h = None

    def set_options(self, **options) -> None:
        self.options = _abi.make_options(**options)
        self._L.orc_engine_set_options(self._h, C.byref(self.options))

    # ---- `contacts.model = "constraint"`
    def set_constraint_options(self, **options) -> None:
        self.constraint_options = _abi.make_constraint_options(**options)
        self._L.orc_engine_set_constraint_options(self._h, C.byref(self.constraint_options))

    def bind_constraints(self, flags: Optional[np.ndarray], data: Optional[np.ndarray]) -> None:
        """Per-lane constraint state of the batch drivers (`_abi.constraint_rows`): `flags` int32
        `[con_flags][B]`, `data` float64 `[con_data][B]`; with single-robot calls pass `[rows][1]`."""
        if flags is None:
            self._L.orc_engine_bind_constraints(self._h, None, None)
            self._con = None
            return
        assert flags.dtype == np.int32 and data.dtype == np.float64
        assert flags.flags.c_contiguous and data.flags.c_contiguous
        self._con = (flags, data)
        self._L.orc_engine_bind_constraints(self._h, flags.ctypes.data, data.ctypes.data)

    def bind_friction(self, friction: Optional[np.ndarray]) -> None:
        """Per-lane `contacts.friction` of the batch drivers (`[B]` float64), None = the engine option."""
        self._friction = None if friction is None else np.ascontiguousarray(friction, dtype=np.float64)
        self._L.orc_engine_bind_friction(self._h, None if friction is None else self._friction.ctypes.data)

    def bind_flexibility(self, flex: Optional[np.ndarray]) -> None:
        """Per-lane flexibility parameters of the batch drivers (`[6 * nspherical][B]` float64: stiffness 3, damping 3 per
        spherical joint in joint order), None = the model's `flexibilityConfig`."""
        self._flex = None if flex is None else np.ascontiguousarray(flex, dtype=np.float64)
        self._L.orc_engine_bind_flexibility(self._h, None if flex is None else self._flex.ctypes.data)

    def bind_model_lane(self, model_lane: Optional[np.ndarray]) -> None:
        """Per-lane body parameters `[13 * njoints][B]` (mass | com | inertia xx xy xz yy yz zz | placement
        translation per joint), None = the model's own."""
        self._model_lane = None if model_lane is None else np.ascontiguousarray(model_lane, dtype=np.float64)
        self._L.orc_engine_bind_model_lane(self._h, None if model_lane is None else self._model_lane.ctypes.data)

    def bind_ground_offset(self, offsets: Optional[np.ndarray]) -> None:
        """Per-lane (x, y) added to the position at which the ground profile is sampled (`[2][B]` float64, batch drivers)."""
        self._ground_offset = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.float64)
        self._L.orc_engine_bind_ground_offset(self._h, None if offsets is None else self._ground_offset.ctypes.data)

    def bind_ground(self, heights: Optional[np.ndarray], x0: float = 0.0, y0: float = 0.0, dx: float = 1.0,
                    dy: float = 1.0) -> None:
        """Ground height map `[ny][nx]` sampled at (x0 + ix dx, y0 + iy dy), None = flat ground."""
        if heights is None:
            self._ground = None
            self._L.orc_engine_bind_ground(self._h, None, 0, 0, 0.0, 0.0, 1.0, 1.0)
            return
        self._ground = np.ascontiguousarray(heights, dtype=np.float64)
        ny, nx = self._ground.shape
        self._L.orc_engine_bind_ground(self._h, self._ground.ctypes.data, nx, ny, float(x0), float(y0), float(dx), float(dy))

    def bind_applied(self, wrenches: Optional[np.ndarray], offsets=None, joints=None) -> None:
        """World-aligned wrenches `[6 K][B]` applied on K <= 4 frames: `offsets` `[K][3]` in the frame's parent joint,
        `joints` `[K]` the parent joints (default: the root joint)."""
        if wrenches is None:
            self._applied = None
            self._L.orc_engine_bind_applied(self._h, None, 0, None, None)
            return
        self._applied = np.ascontiguousarray(wrenches, dtype=np.float64)
        k = self._applied.shape[0] // 6
        self._applied_p = np.ascontiguousarray(np.zeros((k, 3)) if offsets is None else offsets, dtype=np.float64)
        self._applied_j = np.ascontiguousarray(np.ones(k) if joints is None else joints, dtype=np.int32)
        self._L.orc_engine_bind_applied(self._h, self._applied.ctypes.data, k, self._applied_p.ctypes.data, self._applied_j.ctypes.data)

    @property
    def pgs_iterations(self) -> int:
        nb, nc = C.c_int(0), C.c_int(0)
        return int(self._L.orc_engine_constraint_counts(self._h, C.byref(nb), C.byref(nc)))

    def start(self, q, v, command=None) -> None:
        q = np.ascontiguousarray(q, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        assert q.shape == (self.model.nq,) and v.shape == (self.model.nv,)
        self._L.orc_engine_set_state(self._h, _p(q), _p(v), None)
        self.set_command(np.zeros(self.model.nmotors) if command is None else command)
        self._L.orc_engine_start(self._h)
        self.t = 0.0

    def set_state(self, q, v, a) -> None:
        q = np.ascontiguousarray(q, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        a = np.ascontiguousarray(a, dtype=np.float64)
        self._L.orc_engine_set_state(self._h, _p(q), _p(v), _p(a))

    def set_command(self, command) -> None:
        c = np.ascontiguousarray(command, dtype=np.float64).reshape(-1)
        if c.size == 0:
            c = np.zeros(1)
        self._L.orc_engine_set_command(self._h, _p(c))

    def step(self, dt: float, n_substeps: int = 1, solver: str = "runge_kutta_4",
             command_changed: bool = True, update_sensors: bool = True) -> None:
        self._L.orc_engine_step(self._h, SOLVERS[solver], float(dt), int(n_substeps),
                                int(command_changed), int(update_sensors))
        self.t += dt * n_substeps

    def dynamics(self, q, v) -> np.ndarray:
        q = np.ascontiguousarray(q, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        a = np.zeros(self.model.nv)
        self._L.orc_engine_dynamics(self._h, _p(q), _p(v), _p(a))
        return a

    def get(self, name: str) -> np.ndarray:
        out = np.zeros(max(self._rows[name], 1))
        n = self._L.orc_engine_get(self._h, _abi.FIELD_NAMES[name], _p(out))
        return out[:n]

    @property
    def status(self) -> int:
        return int(self._L.orc_engine_status(self._h))

    def joint_placement(self, joint: int):
        out = np.zeros(12)
        self._L.orc_engine_joint_placement(self._h, joint, _p(out))
        return out[:9].reshape(3, 3), out[9:]

    def integrate(self, q, dv) -> np.ndarray:
        q = np.ascontiguousarray(q, dtype=np.float64)
        dv = np.ascontiguousarray(dv, dtype=np.float64)
        out = np.zeros(self.model.nq)
        self._L.orc_integrate(self._h, _p(q), _p(dv), _p(out))
        return out

    # ---- batch drivers over SoA arrays [rows][B] (numpy float64, C-contiguous)
    def batch_run(self, mode: str, arrays: Dict[str, np.ndarray], solver: str = "runge_kutta_4",
                  dt: float = 1e-3, n_substeps: int = 1, command_changed: bool = True,
                  update_sensors: bool = True, lanes=None) -> None:
        B = arrays["q"].shape[1]
        io = BatchIO()
        io.B = B
        for name, _ in BatchIO._fields_[1:]:
            arr = arrays.get(name)
            if arr is not None:
                assert arr.flags.c_contiguous and arr.shape[-1] == B, name
                want = np.int32 if name == "status" else np.float64
                assert arr.dtype == want, name
                setattr(io, name, arr.ctypes.data)
        lo, hi = (0, B) if lanes is None else lanes
        self._L.orc_batch_run(self._h, C.byref(io), {"start": 0, "step": 1, "dynamics": 2}[mode],
                              SOLVERS[solver], float(dt), int(n_substeps), int(command_changed),
                              int(update_sensors), lo, hi)

    def batch_run_dopri(self, arrays: Dict[str, np.ndarray], adaptive: Dict[str, np.ndarray], t_next: float,
                        tol_rel: float = 1e-4, tol_abs: float = 1e-5, dt_max: float = 0.02,
                        dt_restore_threshold_rel: float = 0.2, successive_iter_failed_max: int = 1000,
                        new_step: bool = True, command_changed: bool = True, update_sensors: bool = True,
                        lanes=None) -> None:
        """Advance every lane to the breakpoint `t_next` with the adaptive Dormand-Prince stepper
        (the reference's default `odeSolver`), per-lane step sizes in `adaptive` (see adaptive_state)."""
        B = arrays["q"].shape[1]
        io = BatchIO()
        io.B = B
        for name, _ in BatchIO._fields_[1:]:
            arr = arrays.get(name)
            if arr is not None:
                assert arr.flags.c_contiguous and arr.shape[-1] == B, name
                setattr(io, name, arr.ctypes.data)
        ad = AdaptiveIO()
        for name, _ in AdaptiveIO._fields_:
            a = adaptive[name]
            assert a.flags.c_contiguous and a.shape == (B,), name
            setattr(ad, name, a.ctypes.data)
        lo, hi = (0, B) if lanes is None else lanes
        self._L.orc_batch_run_dopri(self._h, C.byref(io), C.byref(ad), float(t_next), float(tol_rel),
                                    float(tol_abs), float(dt_max), float(dt_restore_threshold_rel),
                                    int(successive_iter_failed_max), int(new_step), int(command_changed),
                                    int(update_sensors), lo, hi)


# ---- sensor noise path (oracle_random.cpp)
def pcg32_stream(state: int, n: int, kind: str = "bits"):
    """`n` outputs of one PCG32 stream started from `state` (already or-ed with 3 by the
    constructor): raw 32-bit words ("bits"), uniform01 ("uniform") or normal01 ("normal").
    Returns (values, final state)."""
    st = C.c_uint64(state)
    if kind == "bits":
        out = np.empty(n, dtype=np.uint32)
        lib().orc_pcg32_stream(C.byref(st), n, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    else:
        out = np.empty(n, dtype=np.float32)
        fn = lib().orc_uniform_stream if kind == "uniform" else lib().orc_normal_stream
        fn(C.byref(st), n, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out, int(st.value)


def ziggurat_tables():
    kn, fn, wn = np.empty(128, np.uint32), np.empty(128, np.float32), np.empty(128, np.float32)
    lib().orc_ziggurat_tables(kn.ctypes.data_as(C.POINTER(C.c_uint32)), fn.ctypes.data_as(C.POINTER(C.c_float)),
                              wn.ctypes.data_as(C.POINTER(C.c_float)))
    return kn, fn, wn


def seed_seq(seed: int, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.uint32)
    lib().orc_seed_seq(seed, n, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def sensor_rng_seed(group_seed: np.ndarray, n_sensors: int) -> np.ndarray:
    gs = np.ascontiguousarray(group_seed, dtype=np.uint32)
    out = np.empty((n_sensors, gs.shape[0]), dtype=np.uint64)
    lib().orc_sensor_rng_seed(gs.ctypes.data_as(C.POINTER(C.c_uint32)), gs.shape[0], n_sensors,
                              out.ctypes.data_as(C.POINTER(C.c_uint64)))
    return out


def engine_rng_seed(seed: np.ndarray) -> np.ndarray:
    """PCG32 states of `Engine::generator_` seeded with `std::seed_seq{seed[lane]}` (engine.cc:756-757)."""
    sd = np.ascontiguousarray(seed, dtype=np.uint32)
    out = np.empty(sd.shape[0], dtype=np.uint64)
    lib().orc_engine_rng_seed(sd.ctypes.data_as(C.POINTER(C.c_uint32)), sd.shape[0], out.ctypes.data_as(C.POINTER(C.c_uint64)))
    return out


def model_bias(nominal: np.ndarray, first_joint: int, std4, rng: np.ndarray, out: np.ndarray, mask=None) -> None:
    """`Model::addBiasedToExtendedModel` per lane: `nominal` `[njoints][25]`, `std4` (inertia, mass, com, position),
    `rng` `[B]` uint64 in / out, `out` `[13 * njoints][B]` float64 (rows of the masked-out lanes / of the joints
    before `first_joint` are left as they are)."""
    nom = np.ascontiguousarray(nominal, dtype=np.float64)
    s4 = np.ascontiguousarray(std4, dtype=np.float32)
    assert rng.dtype == np.uint64 and rng.flags.c_contiguous and out.dtype == np.float64 and out.flags.c_contiguous
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    lib().orc_model_bias(out.shape[1], nom.shape[0], first_joint, nom.ctypes.data_as(C.POINTER(C.c_double)),
                         s4.ctypes.data_as(C.POINTER(C.c_float)), rng.ctypes.data_as(C.POINTER(C.c_uint64)),
                         None if m is None else m.ctypes.data, out.ctypes.data_as(C.POINTER(C.c_double)))


def sensor_noise(data: np.ndarray, rng: np.ndarray, n_sensors: int, n_fields: int, noise_std=None, bias=None,
                 rot=None) -> None:
    """In place on `data` `[n_sensors * n_fields][B]` float64 and `rng` `[n_sensors][B]` uint64."""
    assert data.dtype == np.float64 and data.flags.c_contiguous and data.shape[0] == n_sensors * n_fields
    assert rng is None or (rng.dtype == np.uint64 and rng.flags.c_contiguous)
    pd = C.POINTER(C.c_double)
    arr = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    std, b, r = arr(noise_std), arr(bias), arr(rot)
    ptr = lambda a: None if a is None else a.ctypes.data_as(pd)  # noqa: E731
    lib().orc_sensor_noise(data.shape[1], n_sensors, n_fields, data.ctypes.data_as(pd),
                           None if rng is None else rng.ctypes.data_as(C.POINTER(C.c_uint64)), ptr(std), ptr(b), ptr(r))


def sensor_delay(data: np.ndarray, hist, slot, times, rng, n_sensors: int, n_fields: int, delay=None, jitter=None,
                 order: int = 0) -> None:
    """`interpolateData` in place on `data` `[n_sensors * n_fields][B]` (float64) from the history ring
    `hist` `[slots][n_sensors * n_fields][B]`; `slot` / `times` describe the samples, oldest first.
    `rng` `[n_sensors][B]` uint64 takes one uniform draw per sensor (None: no generator)."""
    assert data.dtype == np.float64 and data.flags.c_contiguous
    pd = C.POINTER(C.c_double)
    arr = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    ptr = lambda a: None if a is None else a.ctypes.data_as(pd)  # noqa: E731
    dl, jt = arr(delay), arr(jitter)
    if hist is not None:
        assert hist.dtype == np.float64 and hist.flags.c_contiguous
        sl = np.ascontiguousarray(slot, dtype=np.int32)
        tm = np.ascontiguousarray(times, dtype=np.float64)
        n_hist = len(sl)
        hp, sp, tp = hist.ctypes.data_as(pd), sl.ctypes.data_as(C.POINTER(C.c_int32)), tm.ctypes.data_as(pd)
    else:
        n_hist, hp, sp, tp = 0, None, None, None
    lib().orc_sensor_delay(data.shape[1], n_sensors, n_fields, data.ctypes.data_as(pd), hp, sp, tp, n_hist,
                           None if rng is None else rng.ctypes.data_as(C.POINTER(C.c_uint64)), ptr(dl), ptr(jt), order)
