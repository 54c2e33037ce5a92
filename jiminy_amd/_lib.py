"""Loader of the per-topology HIP library (C ABI of include/jiminy_hip.h).

There is NO CPU fallback: if the library for a model's topology cannot be loaded (or built with
hipcc), every entry point of the engine raises.  The CPU oracle under oracle/ is test
infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

from . import _abi, codegen
from .model import CompiledModel


class JiminyError(RuntimeError):
    pass


class BadControlFlow(JiminyError):
    """≙ jiminy::bad_control_flow (reference core/include/jiminy/core/fwd.h:173-192)."""


class TopologyMismatch(JiminyError):
    pass


_EXC = {
    _abi.JM_EINVAL: ValueError,
    _abi.JM_ERUNTIME: RuntimeError,
    _abi.JM_ECONTROLFLOW: BadControlFlow,
    _abi.JM_ELOOKUP: LookupError,
    _abi.JM_ENOTIMPL: NotImplementedError,
    _abi.JM_ETOPOLOGY: TopologyMismatch,
}

_LIBS: Dict[Tuple[str, int], "HipLibrary"] = {}

# every symbol include/jiminy_hip.h declares
ABI_SYMBOLS = (
    "jm_topology_signature", "jm_abi_version", "jm_model_create", "jm_model_destroy",
    "jm_batch_create", "jm_batch_destroy", "jm_batch_set_options", "jm_batch_workspace_rows",
    "jm_batch_bind", "jm_batch_start", "jm_batch_stop", "jm_batch_step", "jm_batch_dynamics",
    "jm_batch_reset_lanes", "jm_batch_enable_timing", "jm_batch_timing_summary", "jm_last_error",
    "jm_block_pd_controller", "jm_block_mahony_filter",
    "jm_batch_adaptive_workspace_rows", "jm_batch_bind_adaptive", "jm_batch_step_adaptive",
    "jm_block_sensor_noise", "jm_sensor_rng_seed",
    "jm_batch_set_constraint_options", "jm_batch_constraint_rows", "jm_block_sensor_delay",
    "jm_batch_set_ground", "jm_batch_set_applied_frames", "jm_batch_set_joint_locks", "jm_block_pd_adapter", "jm_block_motor_safety_limit",
    "jm_block_model_bias", "jm_engine_rng_seed",
)


class HipLibrary:
    def __init__(self, path: str) -> None:
        if not os.path.exists(path):
            raise JiminyError(f"HIP library not found: {path}")
        self.path = path
        L = C.CDLL(path)
        self.L = L
        vp = C.c_void_p
        L.jm_topology_signature.restype = C.c_char_p
        L.jm_abi_version.restype = C.c_int32
        L.jm_model_create.argtypes = [C.POINTER(_abi.ModelDesc), C.POINTER(vp)]
        L.jm_model_destroy.argtypes = [vp]
        L.jm_batch_create.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, C.POINTER(vp)]
        L.jm_batch_destroy.argtypes = [vp]
        L.jm_batch_set_options.argtypes = [vp, C.POINTER(_abi.Options)]
        L.jm_batch_workspace_rows.argtypes = [vp]
        L.jm_batch_bind.argtypes = [vp, C.c_int32, vp]
        L.jm_batch_start.argtypes = [vp, vp]
        L.jm_batch_stop.argtypes = [vp]
        L.jm_batch_step.argtypes = [vp, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_int32, vp]
        L.jm_batch_dynamics.argtypes = [vp, vp, vp, vp, vp]
        L.jm_batch_reset_lanes.argtypes = [vp, vp, vp, vp, vp]
        L.jm_batch_enable_timing.argtypes = [vp, C.c_int32]
        L.jm_batch_timing_summary.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        L.jm_last_error.argtypes = [C.c_char_p, C.c_size_t]
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        L.jm_batch_adaptive_workspace_rows.argtypes = [vp]
        L.jm_batch_bind_adaptive.argtypes = [vp, vp, vp, vp]
        L.jm_batch_step_adaptive.argtypes = [vp, C.c_double, C.POINTER(_abi.AdaptiveOptions), C.c_int32,
                                             C.c_int32, C.c_int32, C.c_int32, ip, vp]
        L.jm_block_pd_controller.argtypes = [C.c_int32, C.c_int64, C.c_int32, vp, ip, vp, dp, dp, dp, dp, dp,
                                             C.c_double, vp, vp]
        L.jm_block_mahony_filter.argtypes = [C.c_int32, C.c_int64, C.c_int32, vp, vp, vp, vp, vp,
                                             C.c_double, C.c_double, C.c_double, vp]
        L.jm_block_pd_adapter.argtypes = [C.c_int32, C.c_int64, C.c_int32, vp, C.c_int32, vp, dp, dp, C.c_int32, dp,
                                          C.c_double, vp, vp]
        L.jm_block_motor_safety_limit.argtypes = [C.c_int32, C.c_int64, C.c_int32, vp, ip, vp, dp, dp, dp, dp, dp, dp, vp, vp]
        L.jm_block_sensor_noise.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.c_int32, vp, vp, dp, dp, dp, vp]
        L.jm_sensor_rng_seed.argtypes = [C.POINTER(C.c_uint32), C.c_int64, C.c_int32, C.POINTER(C.c_uint64)]
        L.jm_block_model_bias.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.c_int32, vp, C.POINTER(C.c_float), vp, vp, vp, vp]
        L.jm_engine_rng_seed.argtypes = [C.POINTER(C.c_uint32), C.c_int64, C.POINTER(C.c_uint64)]
        L.jm_block_sensor_delay.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.c_int32, vp, vp, ip, dp, C.c_int32, vp,
                                            dp, dp, C.c_int32, vp]
        L.jm_batch_set_constraint_options.argtypes = [vp, C.POINTER(_abi.ConstraintOptions)]
        L.jm_batch_constraint_rows.argtypes = [vp, ip, ip, ip]
        L.jm_batch_set_ground.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double]
        L.jm_batch_set_applied_frames.argtypes = [vp, C.c_int32, dp, ip]
        L.jm_batch_set_joint_locks.argtypes = [vp, C.c_int32]
        for name in ABI_SYMBOLS:
            getattr(L, name)  # AttributeError if a declared symbol is not exported
            if name not in ("jm_topology_signature",):
                getattr(L, name).restype = getattr(L, name).restype or C.c_int32
        if L.jm_abi_version() != _abi.ABI_VERSION:
            raise JiminyError(f"{path}: ABI version {L.jm_abi_version()} != {_abi.ABI_VERSION}")

    def signature(self) -> str:
        return self.L.jm_topology_signature().decode()

    def check(self, rc: int) -> None:
        if rc == _abi.JM_OK:
            return
        buf = C.create_string_buffer(1024)
        self.L.jm_last_error(buf, 1024)
        raise _EXC.get(rc, JiminyError)(buf.value.decode() or f"jiminy_hip error {rc}")


def load_for(model: CompiledModel, allow_build: bool = True, variant: Optional[int] = None) -> HipLibrary:
    """Return the HIP library specialised for `model`'s topology (build variant `variant`, default:
    the one codegen.preferred_variant names), building it if needed."""
    v = codegen.preferred_variant(model) if variant is None else variant
    key = (model.topology_hash(), v)
    lib = _LIBS.get(key)
    if lib is not None:
        return lib
    path = codegen.lib_path(model, v)
    # A library is (re)built when it is missing (e.g. a user-supplied URDF) or when the digest of the sources it
    # was built from (recorded next to it, codegen.source_digest) differs from the sources in the tree -- contents,
    # not file times, so that a snapshot copied to another machine does not recompile.  JIMINY_AMD_REBUILD=0 turns
    # the rebuild of a stale library into a warning; libraries without a record are only rebuilt with =1.
    stale = os.path.exists(path) and codegen.is_stale(model, v)
    mode = os.environ.get("JIMINY_AMD_REBUILD", "")
    has_record = os.path.exists(path + ".src")
    rebuild = stale and (mode == "1" or (mode != "0" and has_record))
    if stale and not rebuild:
        import warnings
        warnings.warn(f"{path} was built from other kernel sources than the ones in the tree "
                      "(run `python __graft_entry__.py` or set JIMINY_AMD_REBUILD=1)")
    if allow_build and (not os.path.exists(path) or rebuild):
        path = codegen.build_library(model, variant=v)
    lib = HipLibrary(path)
    if lib.signature() != model.topology_signature():
        raise TopologyMismatch(f"{path} was built for another topology")
    _LIBS[key] = lib
    return lib
