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
