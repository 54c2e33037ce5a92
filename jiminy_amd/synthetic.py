"""Seeded synthetic batch states (the recipe of BASELINE.md §3 / SURVEY.md §8d).

Host-side numpy generation (float64), laid out structure-of-arrays `[rows][B]` like every batch
array of the engine.  Lanes stay inside the joint position bounds: the reference enforces bounds
through its constraint solver (engine.cc:3285-3298), which is outside the batched hot path.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from .model import (CompiledModel, JT_FREEFLYER, JT_RUBU, JT_RUBX, JT_RUBY, JT_RUBZ, JT_NV, JT_SPHERICAL)


def _quat_from_axis_angle(axis: np.ndarray, angle: np.ndarray) -> np.ndarray:
    axis = axis / np.linalg.norm(axis, axis=0, keepdims=True)
    s = np.sin(0.5 * angle)
    return np.stack([axis[0] * s, axis[1] * s, axis[2] * s, np.cos(0.5 * angle)])


def _quat_to_matrix(q: np.ndarray) -> np.ndarray:
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def sample_states(model: CompiledModel, batch_size: int, seed: int = 0,
                  base_height=(0.45, 0.65), base_angle_max: float = 0.3,
                  joint_range: float = 1.0, joint_margin: float = 0.05,
                  base_twist_std: float = 0.2, joint_vel_std: float = 0.5,
                  command_fraction: float = 0.5, small: float = 0.05,
                  grounded_fraction: float = 0.25) -> Dict[str, np.ndarray]:
    """Returns {'q': [nq][B], 'v': [nv][B], 'command': [nmotors][B]} (float64).

    Floating-base robots: base height U(base_height), small random base rotation, base twist
    N(0, base_twist_std^2), joints uniform inside their bounds (margin, clipped to +-joint_range
    around neutral), joint velocities N(0, joint_vel_std^2).  A fraction of the lanes is then
    lowered so that its lowest contact point sits within 5 mm of the ground (contact branch).
    Fixed-base toy models: every coordinate / velocity U(+-small) (gym_jiminy cartpole.py:29-35).
    Commands: U(+-command_fraction * effort limit).
    """
    rng = np.random.default_rng(seed)
    B = int(batch_size)
    q = np.repeat(model.neutral()[:, None], B, axis=1)
    v = np.zeros((model.nv, B))
    floating = model.has_freeflyer
    for j in range(1, model.njoints):
        t, iq, iv = int(model.jtypes[j]), int(model.idx_q[j]), int(model.idx_v[j])
        if t == JT_FREEFLYER:
            q[iq + 0] = rng.uniform(-0.5, 0.5, B)
            q[iq + 1] = rng.uniform(-0.5, 0.5, B)
            q[iq + 2] = rng.uniform(base_height[0], base_height[1], B)
            axis = rng.normal(size=(3, B))
            q[iq + 3:iq + 7] = _quat_from_axis_angle(axis, rng.uniform(0.0, base_angle_max, B))
            v[iv:iv + 6] = rng.normal(0.0, base_twist_std, (6, B))
        elif t == JT_SPHERICAL:
            # flexibility joints: a small deflection and a slow rate
            q[iq:iq + 4] = _quat_from_axis_angle(rng.normal(size=(3, B)), rng.uniform(0.0, 0.15, B))
            v[iv:iv + 3] = rng.normal(0.0, 0.3, (3, B))
        elif t in (JT_RUBX, JT_RUBY, JT_RUBZ, JT_RUBU):
            th = rng.uniform(-small, small, B) if not floating else rng.uniform(-joint_range, joint_range, B)
            q[iq], q[iq + 1] = np.cos(th), np.sin(th)
            v[iv] = rng.uniform(-small, small, B) if not floating else rng.normal(0, joint_vel_std, B)
        else:
            if floating:
                lo = max(model.position_lower[iq] + joint_margin, -joint_range)
                hi = min(model.position_upper[iq] - joint_margin, joint_range)
                q[iq] = rng.uniform(lo, hi, B)
                v[iv] = rng.normal(0.0, joint_vel_std, B)
            else:
                q[iq] = rng.uniform(-small, small, B)
                v[iv] = rng.uniform(-small, small, B)
    cmd = np.zeros((model.nmotors, B))
    for i, m in enumerate(model.motors):
        lim = m.effort_limit if np.isfinite(m.effort_limit) else 1.0
        cmd[i] = rng.uniform(-command_fraction * lim, command_fraction * lim, B)
    if floating and model.ncontacts > 0:
        # Make every lane a state the reference would accept at `start` (initial contact force
        # below 1e5 N, engine.cc:1338-1345): lanes whose lowest contact point would sit below the
        # ground are lifted to a clearance U(0, 0.1) m; a fixed fraction of the lanes is then
        # placed with its lowest contact point within 5 mm of the ground (contact branch active).
        zmin = lowest_contact_height(model, q)
        clearance = rng.uniform(0.0, 0.1, B)
        lift = np.where(zmin < clearance, clearance - zmin, 0.0)
        n_g = int(round(grounded_fraction * B))
        target = rng.uniform(-0.005, 0.005, B)
        lift[:n_g] = target[:n_g] - zmin[:n_g]
        q[2] += lift
    return {"q": np.ascontiguousarray(q), "v": np.ascontiguousarray(v),
            "command": np.ascontiguousarray(cmd)}


def _batched_rot(n: np.ndarray, c: np.ndarray, s: np.ndarray) -> np.ndarray:
    """Rodrigues rotation for axis n (3,), cos/sin arrays (B,) -> (B, 3, 3)."""
    K = np.array([[0, -n[2], n[1]], [n[2], 0, -n[0]], [-n[1], n[0], 0.0]])
    K2 = K @ K
    return np.eye(3)[None] + s[:, None, None] * K[None] + (1 - c)[:, None, None] * K2[None]


def joint_world_placements(model: CompiledModel, q: np.ndarray):
    """Plain numpy forward kinematics for q of shape [nq][B] (or [nq]):
    returns lists of rotations (B,3,3) and positions (B,3) per joint."""
    from .model import JT_PU, JT_PX, JT_PY, JT_PZ, JT_RU, JT_RX, JT_RY, JT_RZ
    q = np.asarray(q, dtype=np.float64)
    if q.ndim == 1:
        q = q[:, None]
    B = q.shape[1]
    Rs = [np.repeat(np.eye(3)[None], B, axis=0)] * model.njoints
    ps = [np.zeros((B, 3))] * model.njoints
    for j in range(1, model.njoints):
        t, iq = int(model.jtypes[j]), int(model.idx_q[j])
        ax = {JT_RX: 0, JT_PX: 0, JT_RUBX: 0, JT_RY: 1, JT_PY: 1, JT_RUBY: 1,
              JT_RZ: 2, JT_PZ: 2, JT_RUBZ: 2}.get(t, -1)
        n = np.eye(3)[ax] if ax >= 0 else np.asarray(model.axes[j], dtype=float)
        Rj = np.repeat(np.eye(3)[None], B, axis=0)
        pj = np.zeros((B, 3))
        if t == JT_FREEFLYER:
            x, y, z, w = q[iq + 3], q[iq + 4], q[iq + 5], q[iq + 6]
            Rj = np.stack([
                np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], 1)
            pj = q[iq:iq + 3].T
        elif t == JT_SPHERICAL:
            x, y, z, w = q[iq], q[iq + 1], q[iq + 2], q[iq + 3]
            Rj = np.stack([
                np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], 1)
        elif t in (JT_RX, JT_RY, JT_RZ, JT_RU, JT_RUBX, JT_RUBY, JT_RUBZ, JT_RUBU):
            c, s = (q[iq], q[iq + 1]) if t >= JT_RUBX else (np.cos(q[iq]), np.sin(q[iq]))
            Rj = _batched_rot(n, c, s)
        else:
            pj = q[iq][:, None] * n[None]
        Rl = model.placement_R[j][None] @ Rj
        pl = model.placement_p[j][None] + pj @ model.placement_R[j].T
        p = int(model.parents[j])
        Rs[j] = Rs[p] @ Rl
        ps[j] = ps[p] + np.einsum("bij,bj->bi", Rs[p], pl)
    return Rs, ps


def lowest_contact_height(model: CompiledModel, q: np.ndarray) -> np.ndarray:
    """Lowest world height over the contact points, per lane (array of shape (B,))."""
    Rs, ps = joint_world_placements(model, q)
    z = []
    for c in model.contacts:
        f = model.frames[c]
        z.append((ps[f.parent_joint] + Rs[f.parent_joint] @ f.p)[:, 2])
    return np.min(np.stack(z), axis=0)


def sample_standing_states(model: CompiledModel, batch_size: int, seed: int = 0,
                           joint_noise: float = 0.15, base_angle_max: float = 0.08,
                           depth_range=(-2.0e-3, 5.0e-4), twist_std: float = 0.05,
                           joint_vel_std: float = 0.2, command_fraction: float = 0.3,
                           out_of_bounds_fraction: float = 0.25,
                           out_of_bounds_max: float = 0.02) -> Dict[str, np.ndarray]:
    """Seeded states for `contacts.model = "constraint"`: a floating-base robot standing on the
    ground (neutral stance + joint noise, small base tilt, lowest contact point at a depth
    U(depth_range) so that several contact constraints are active or inside the hysteresis band),
    slow motion, and a fraction of the lanes with one to three bounded joints pushed past a
    position limit by up to `out_of_bounds_max` (joint-bound constraints active, both directions).
    Returns {'q', 'v', 'command'} laid out `[rows][B]` (float64)."""
    rng = np.random.default_rng(seed)
    B = int(batch_size)
    q = np.repeat(model.neutral()[:, None], B, axis=1)
    v = np.zeros((model.nv, B))
    bounded = []
    for j in range(1, model.njoints):
        t, iq, iv = int(model.jtypes[j]), int(model.idx_q[j]), int(model.idx_v[j])
        if t == JT_FREEFLYER:
            q[iq + 0] = rng.uniform(-0.5, 0.5, B)
            q[iq + 1] = rng.uniform(-0.5, 0.5, B)
            q[iq + 2] = 1.0
            axis = rng.normal(size=(3, B))
            q[iq + 3:iq + 7] = _quat_from_axis_angle(axis, rng.uniform(0.0, base_angle_max, B))
            v[iv:iv + 6] = rng.normal(0.0, twist_std, (6, B))
        elif t == JT_SPHERICAL:
            q[iq:iq + 4] = _quat_from_axis_angle(rng.normal(size=(3, B)), rng.uniform(0.0, 0.15, B))
            v[iv:iv + 3] = rng.normal(0.0, 0.3, (3, B))
        elif t in (JT_RUBX, JT_RUBY, JT_RUBZ, JT_RUBU):
            th = rng.uniform(-joint_noise, joint_noise, B)
            q[iq], q[iq + 1] = np.cos(th), np.sin(th)
            v[iv] = rng.normal(0, joint_vel_std, B)
        else:
            lo, hi = model.position_lower[iq], model.position_upper[iq]
            mid = np.clip(q[iq], lo + 0.05, hi - 0.05)
            q[iq] = np.clip(mid + rng.uniform(-joint_noise, joint_noise, B), lo + 0.02, hi - 0.02)
            v[iv] = rng.normal(0.0, joint_vel_std, B)
            bounded.append((iq, lo, hi))
    n_oob = int(round(out_of_bounds_fraction * B)) if bounded else 0
    for lane in range(n_oob):
        for _ in range(int(rng.integers(1, 4))):
            iq, lo, hi = bounded[int(rng.integers(len(bounded)))]
            over = rng.uniform(-0.5e-3, out_of_bounds_max)  # slightly inside = hysteresis band
            q[iq, lane] = hi + over if rng.random() < 0.5 else lo - over
    cmd = np.zeros((model.nmotors, B))
    for i, m in enumerate(model.motors):
        lim = m.effort_limit if np.isfinite(m.effort_limit) else 1.0
        cmd[i] = rng.uniform(-command_fraction * lim, command_fraction * lim, B)
    if model.has_freeflyer and model.ncontacts > 0:
        zmin = lowest_contact_height(model, q)
        q[2] += rng.uniform(depth_range[0], depth_range[1], B) - zmin
    return {"q": np.ascontiguousarray(q), "v": np.ascontiguousarray(v),
            "command": np.ascontiguousarray(cmd)}
