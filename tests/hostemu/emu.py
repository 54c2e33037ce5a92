"""Build + ctypes driver of the host emulation of the kernel code (tests only)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict

import numpy as np

from jiminy_amd import _abi, codegen
from jiminy_amd.model import CompiledModel

_HERE = os.path.dirname(os.path.abspath(__file__))
_CACHE: Dict[str, C.CDLL] = {}

_FIELDS = ("q", "v", "a", "command", "u_motor", "u", "f_external", "contact_forces", "imu", "force",
           "contact", "encoder", "effort", "energy", "joint_forces", "centroidal", "status",
           "q_in", "v_in", "a_out", "mask", "q_init", "v_init")


class EmuIO(C.Structure):
    _fields_ = [("B", C.c_longlong)] + [(n, C.c_void_p) for n in _FIELDS]


def _lib(model: CompiledModel) -> C.CDLL:
    h = model.topology_hash()
    if h in _CACHE:
        return _CACHE[h]
    hdr = codegen.write_header(model)
    # EMU_CXX / EMU_EXTRA_FLAGS / EMU_TAG: alternative host builds of the same kernel code (e.g. amdclang++ with
    # -ftrivial-auto-var-init=pattern: every uninitialised local becomes NaN, the hunt of DESIGN.md section 4.7)
    out = os.path.join(codegen.BUILD, f"libemu_{h}{os.environ.get('EMU_TAG', '')}.so")
    deps = [os.path.join(_HERE, "emu.cpp"), hdr] + codegen._sources()[1:] + \
           []
    if (not os.path.exists(out)) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call([os.environ.get("EMU_CXX", "g++"), "-O1", *os.environ.get("EMU_EXTRA_FLAGS", "").split(), "-std=c++17", "-fPIC", "-shared", "-march=x86-64-v3",
                               "-ffp-contract=off", "-pthread", f"-DJM_TOPO_HEADER=\"{hdr}\"",
                               os.path.join(_HERE, "emu.cpp"), "-o", out])
    L = C.CDLL(out)
    L.emu_run.argtypes = [C.POINTER(_abi.ModelDesc), C.POINTER(_abi.Options), C.POINTER(EmuIO),
                          C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int]
    L.emu_set_constraints.argtypes = [C.POINTER(_abi.ConstraintOptions), C.c_void_p, C.c_void_p]
    L.emu_set_constraints.restype = None
    L.emu_set_friction.argtypes = [C.c_void_p]
    L.emu_set_friction.restype = None
    L.emu_set_ground_offset.argtypes = [C.c_void_p]
    L.emu_set_ground_offset.restype = None
    L.emu_set_flexibility.argtypes = [C.c_void_p]
    L.emu_set_flexibility.restype = None
    L.emu_set_split.argtypes = [C.c_int]
    L.emu_set_split.restype = None
    L.emu_has_split.restype = C.c_int
    L.emu_tip_solves.restype = C.c_longlong
    L.emu_lane_solves.restype = C.c_longlong
    L.emu_set_gen.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                              C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.emu_set_gen.restype = None
    L.emu_run_dopri.argtypes = [C.POINTER(_abi.ModelDesc), C.POINTER(_abi.Options), C.POINTER(EmuIO), C.c_void_p, C.c_void_p,
                                C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p]
    _CACHE[h] = L
    return L


MODES = {"step": 0, "start": 1, "dynamics": 2, "reset": 3}
SOLVERS = {"euler_explicit": 0, "runge_kutta_4": 1}


def run(model: CompiledModel, arrays: Dict[str, np.ndarray], mode: str, options=None,
        solver: str = "runge_kutta_4", dt: float = 1e-3, n_substeps: int = 1,
        command_changed: bool = True, update_sensors: bool = True, dtype=np.float64,
        variant: str = "lane", constraint_options=None, model_lane=None, ground=None, applied=None, split: bool = False) -> None:
    """`split`: step launches of the constraint model in the pre | solve | post form of robots with large solves
    (jm_qcon.h; ignored for topologies that do not have it).  `model_lane` `[13 * njoints][B]`; `ground` = (heights [ny][nx], x0, y0, dx, dy); `applied` = (wrenches
    [6 K][B], offsets [K][3]) -- the optional per-environment variation of the branch-parallel code."""
    L = _lib(model)
    g = ground if ground is not None else (None, 0.0, 0.0, 1.0, 1.0)
    gh = None if g[0] is None else np.ascontiguousarray(g[0], dtype=dtype)
    ap = None if applied is None else np.ascontiguousarray(applied[0], dtype=dtype)
    apo = None if applied is None else np.ascontiguousarray(applied[1], dtype=np.float64)
    # optional third entry: the parent joint of every frame (default: the root joint)
    apj = None if (applied is None or len(applied) < 3) else np.ascontiguousarray(applied[2], dtype=np.int32)
    ml = None if model_lane is None else np.ascontiguousarray(model_lane, dtype=dtype)
    L.emu_set_gen(None if ml is None else ml.ctypes.data, None if gh is None else gh.ctypes.data,
                  0 if gh is None else gh.shape[1], 0 if gh is None else gh.shape[0], float(g[1]), float(g[2]), float(g[3]),
                  float(g[4]), None if ap is None else ap.ctypes.data, 0 if ap is None else ap.shape[0] // 6,
                  None if apo is None else apo.ctypes.data, None if apj is None else apj.ctypes.data)
    if constraint_options is not None:
        co = _abi.make_constraint_options(**constraint_options)
        L.emu_set_constraints(C.byref(co), arrays["con_flags"].ctypes.data, arrays["con_data"].ctypes.data)
    else:
        co = _abi.make_constraint_options(model="spring_damper")
        L.emu_set_constraints(C.byref(co), None, None)
    go = arrays.get("ground_offset")
    L.emu_set_ground_offset(go.ctypes.data if (go is not None and gh is not None) else None)
    fr = arrays.get("friction")
    L.emu_set_friction(fr.ctypes.data if fr is not None else None)
    fx = arrays.get("flexibility")
    L.emu_set_flexibility(fx.ctypes.data if fx is not None else None)
    if variant == "quad" and not L.emu_has_quad():
        raise RuntimeError("this topology has no limb-parallel variant")
    L.emu_set_variant(1 if variant == "quad" else 0)
    L.emu_set_split(1 if split else 0)
    desc, keep = _abi.make_model_desc(model)
    opts = options if options is not None else _abi.make_options()
    io = EmuIO()
    io.B = arrays["q"].shape[-1]
    for n in _FIELDS:
        a = arrays.get(n)
        if a is not None:
            assert a.flags.c_contiguous, n
            setattr(io, n, a.ctypes.data)
    rc = L.emu_run(C.byref(desc), C.byref(opts), C.byref(io),
                   _abi.JM_F64 if dtype == np.float64 else _abi.JM_F32, MODES[mode], SOLVERS[solver],
                   float(dt), int(n_substeps), int(command_changed), int(update_sensors))
    if rc != 0:
        raise RuntimeError(f"emu_run failed with code {rc}")


# rows of the per-lane stepper state (jm_adaptive.h)
_AD_F = ("t", "dt", "dt_largest", "dt_largest_prev", "dt_try")
_AD_I = ("iter", "iter_failed", "succ_too_large", "succ_failed", "active", "bp_reached", "map")


def run_dopri(model: CompiledModel, arrays: Dict[str, np.ndarray], adaptive: Dict[str, np.ndarray], t_next: float,
              tol_rel: float = 1e-4, tol_abs: float = 1e-5, dt_max: float = 0.02, dt_restore_threshold_rel: float = 0.2,
              successive_iter_failed_max: int = 1000, new_step: bool = True, options=None, max_attempts: int = 100000):
    """The persistent adaptive stepper (jm_qdopri.h) on the host: every robot of `arrays` to `t_next`; `adaptive` is
    the per-lane stepper state in the oracle's form (oracle_py.adaptive_state), updated in place.  Returns
    (robots still active, largest attempt count)."""
    L = _lib(model)
    B = arrays["q"].shape[-1]
    fs = np.zeros((len(_AD_F), B))
    isv = np.zeros((len(_AD_I), B), dtype=np.int32)
    for i, n in enumerate(_AD_F[:4]):
        fs[i] = adaptive[n]
    for i, n in enumerate(_AD_I[:4]):
        isv[i] = adaptive[n]
    desc, keep = _abi.make_model_desc(model)
    opts = options if options is not None else _abi.make_options()
    io = EmuIO()
    io.B = B
    for n in _FIELDS:
        a = arrays.get(n)
        if a is not None:
            assert a.flags.c_contiguous, n
            setattr(io, n, a.ctypes.data)
    counters = np.zeros(2, dtype=np.int32)
    rc = L.emu_run_dopri(C.byref(desc), C.byref(opts), C.byref(io), fs.ctypes.data, isv.ctypes.data, float(t_next),
                         float(tol_rel), float(tol_abs), float(dt_max), float(dt_restore_threshold_rel),
                         int(successive_iter_failed_max), int(new_step), int(max_attempts), counters.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"emu_run_dopri failed with code {rc}")
    for i, n in enumerate(_AD_F[:4]):
        adaptive[n][:] = fs[i]
    for i, n in enumerate(_AD_I[:4]):
        adaptive[n][:] = isv[i]
    return int(counters[0]), int(counters[1])


def lane_solves(model: CompiledModel) -> int:
    """Calls of the one-lane-per-robot solve of the split form (jm_qcon.h, qcon_pgs_lane) since the library was loaded."""
    return int(_lib(model).emu_lane_solves())


def tip_solves(model: CompiledModel) -> int:
    """Solves of the split form that took the operational-space form (jm_qtip.h) since the library was loaded."""
    return int(_lib(model).emu_tip_solves())
