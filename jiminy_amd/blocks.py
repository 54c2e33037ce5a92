"""Batched, device-resident versions of the gym_jiminy pipeline blocks used by the ANYmal/Atlas
environments: PD controller + ZOH command integrator, PD adapter and
Mahony attitude filter.

They restate the numba kernels of the reference
(python/gym_jiminy/common/gym_jiminy/common/blocks/proportional_derivative_controller.py:22-260,
mahony_filter.py:28-101) over `[rows][B]` arrays; every lane is one environment.

Two forms of the per-controller-tick blocks:
* `HipBlocks.pd_controller` / `HipBlocks.mahony_filter`: ONE hand-written HIP kernel launch each
  (csrc/jm_blocks.h behind `jm_block_*` of include/jiminy_hip.h) -- what the environments use;
* `pd_controller` / `mahony_filter` / `integrate_zoh`: the same arithmetic as elementwise tensor
  programs (scalar branches become `torch.where`), ~40 launches per block.  They run wherever their
  tensors live and serve as the readable specification the kernels are tested against.
`HipBlocks.pd_adapter` / `HipBlocks.motor_safety_limit` are the `PDAdapter` / `MotorSafetyLimit` blocks as single
launches; `pd_adapter` below is the tensor-program form.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

EARTH_SURFACE_GRAVITY = 9.81


class HipBlocks:
    """Per-controller-tick pipeline blocks as single HIP kernel launches (`jm_block_*`).

    Bound to one engine: uses its topology library, device, dtype and launch stream.  All tensors
    are `[rows][B]`-shaped, contiguous, on the engine's device."""

    def __init__(self, engine, encoder_index, command_state_lower, command_state_upper, kp, kd,
                 motors_effort_limit) -> None:
        import ctypes as C

        import numpy as np
        from . import _abi
        self._C = C
        self._eng = engine
        self._L = engine._lib.L
        self._check = engine._lib.check
        self._dtype = _abi.JM_F64 if engine.dtype == torch.float64 else _abi.JM_F32

        def host(x, shape):
            a = np.ascontiguousarray(torch.as_tensor(x).detach().cpu().numpy(), dtype=np.float64)
            if a.shape != shape:
                raise ValueError(f"expected an array of shape {shape}, got {a.shape}")
            return a
        M = engine.model.nmotors
        self._M = M
        self._enc = np.ascontiguousarray(torch.as_tensor(encoder_index).cpu().numpy(), dtype=np.int32)
        self._lo, self._hi = host(command_state_lower, (3, M)), host(command_state_upper, (3, M))
        self._kp, self._kd, self._lim = host(kp, (M,)), host(kd, (M,)), host(motors_effort_limit, (M,))

    def _ptr(self, t: torch.Tensor):
        if not t.is_contiguous() or t.device != self._eng.device or t.dtype != self._eng.dtype:
            raise ValueError("pipeline block tensors must be contiguous engine-dtype tensors on the engine's device")
        return self._C.c_void_p(t.data_ptr())

    def pd_controller(self, command_state: torch.Tensor, control_dt: float, out: torch.Tensor) -> None:
        """≙ `pd_controller` (proportional_derivative_controller.py:101-163) on the engine's raw
        encoder field; `command_state` `[3][M][B]` is advanced in place, `out` `[M][B]`."""
        C = self._C
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        self._check(self._L.jm_block_pd_controller(
            self._dtype, self._eng.batch_size, self._M, self._ptr(self._eng.field("encoder")),
            self._enc.ctypes.data_as(ip), self._ptr(command_state), self._lo.ctypes.data_as(dp),
            self._hi.ctypes.data_as(dp), self._kp.ctypes.data_as(dp), self._kd.ctypes.data_as(dp),
            self._lim.ctypes.data_as(dp), float(control_dt), self._ptr(out), self._eng._stream()))

    def mahony_filter(self, q: torch.Tensor, omega: torch.Tensor, cf: torch.Tensor, bias_hat: torch.Tensor,
                      kp: float, ki: float, dt: float) -> None:
        """≙ `mahony_filter` (mahony_filter.py:28-101) on the engine's raw IMU field; `q`
        `[4][n_imu][B]`, the others `[3][n_imu][B]`, all updated in place."""
        n_imu = q.shape[1]
        self._check(self._L.jm_block_mahony_filter(
            self._dtype, self._eng.batch_size, n_imu, self._ptr(self._eng.field("imu")), self._ptr(q),
            self._ptr(omega), self._ptr(cf), self._ptr(bias_hat), float(kp), float(ki), float(dt),
            self._eng._stream()))


    def pd_adapter(self, action: torch.Tensor, order: int, command_state: torch.Tensor, is_instantaneous: bool,
                   velocity_deadband, step_dt: float, out: torch.Tensor) -> None:
        """≙ `pd_adapter` (proportional_derivative_controller.py:166-260): `action` `[M][B]` -> target
        acceleration `out` `[M][B]` (command state `[3][M][B]` moved in place when instantaneous)."""
        import numpy as np
        C = self._C
        dp = C.POINTER(C.c_double)
        db = None
        if velocity_deadband is not None:
            db = np.ascontiguousarray(torch.as_tensor(velocity_deadband).cpu().numpy(), dtype=np.float64)
        self._check(self._L.jm_block_pd_adapter(
            self._dtype, self._eng.batch_size, self._M, self._ptr(action), int(order), self._ptr(command_state),
            self._lo.ctypes.data_as(dp), self._hi.ctypes.data_as(dp), int(bool(is_instantaneous)),
            None if db is None else db.ctypes.data_as(dp), float(step_dt), self._ptr(out), self._eng._stream()))

    def motor_safety_limit(self, command: torch.Tensor, kp, kd, soft_position_lower, soft_position_upper,
                           velocity_limit, out: torch.Tensor) -> None:
        """≙ `apply_safety_limits` (blocks/motor_safety_limit.py:20-77) on the engine's raw encoder field."""
        import numpy as np
        C = self._C
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        arr = lambda x: np.ascontiguousarray(np.broadcast_to(np.asarray(torch.as_tensor(x).cpu().numpy(), dtype=np.float64), (self._M,)))  # noqa: E731
        a = [arr(x) for x in (kp, kd, soft_position_lower, soft_position_upper, velocity_limit)]
        self._check(self._L.jm_block_motor_safety_limit(
            self._dtype, self._eng.batch_size, self._M, self._ptr(self._eng.field("encoder")), self._enc.ctypes.data_as(ip),
            self._ptr(command), *[x.ctypes.data_as(dp) for x in a], self._lim.ctypes.data_as(dp), self._ptr(out),
            self._eng._stream()))


def integrate_zoh(state: torch.Tensor, state_min: torch.Tensor, state_max: torch.Tensor,
                  dt: float) -> None:
    """≙ `integrate_zoh` (proportional_derivative_controller.py:23-98), in place.

    `state` is `[3][M][B]` (position, velocity, acceleration); bounds are `[3][M]` or `[3][M][B]`.
    """
    assert dt >= 0.0, "Integration backward in time is not supported."
    if abs(dt) < 1e-9:
        return
    if state_min.dim() == 2:
        state_min, state_max = state_min[..., None], state_max[..., None]
    position, velocity, acceleration = state[0], state[1], state[2]
    p_min, v_min, a_min = state_min[0], state_min[1], state_min[2]
    p_max, v_max, a_max = state_max[0], state_max[1], state_max[2]
    acc = torch.minimum(torch.maximum(acceleration, a_min), a_max)
    v_prev = velocity.clone()
    vel = velocity + acc * dt
    vel = torch.minimum(torch.maximum(vel, v_min), v_max)
    # slow down early enough not to violate the acceleration limit when hitting position bounds
    horizon = torch.clamp_min(torch.trunc(v_prev.abs() / a_max / dt) * dt, dt)
    d_min = p_min - position
    d_max = p_max - position
    drift = 0.5 * (horizon * (horizon - dt)) * a_max
    far = horizon > dt
    d_min = torch.where(far, d_min - drift, d_min)
    d_max = torch.where(far, d_max + drift, d_max)
    vel = torch.minimum(torch.maximum(vel, d_min / horizon), d_max / horizon)
    # velocity after hitting bounds must be cancellable in a single step
    fast = vel.abs() > dt * a_max
    safe = torch.where(fast, vel, torch.ones_like(vel))
    lo = -torch.clamp_min(d_min / safe, dt) * a_max
    hi = torch.clamp_min(d_max / safe, dt) * a_max
    vel = torch.where(fast, torch.minimum(torch.maximum(vel, lo), hi), vel)
    state[2].copy_((vel - v_prev) / dt)
    state[1].copy_(vel)
    state[0].copy_(position + dt * vel)


def pd_controller(encoder_data: torch.Tensor, command_state: torch.Tensor,
                  command_state_lower: torch.Tensor, command_state_upper: torch.Tensor,
                  kp: torch.Tensor, kd: torch.Tensor, motors_effort_limit: torch.Tensor,
                  control_dt: float, out: torch.Tensor) -> None:
    """≙ `pd_controller` (proportional_derivative_controller.py:101-163).

    `encoder_data` `[2][M][B]` (position, velocity), `command_state` `[3][M][B]` (updated in
    place), gains / limits `[M]` or `[M][B]`, `out` `[M][B]` = clipped motor torques.
    """
    integrate_zoh(command_state, command_state_lower, command_state_upper, control_dt)
    if kp.dim() == 1:
        kp, kd, motors_effort_limit = kp[:, None], kd[:, None], motors_effort_limit[:, None]
    q_error = command_state[0] - encoder_data[0]
    v_error = command_state[1] - encoder_data[1]
    u = kp * (q_error + kd * v_error)
    out.copy_(torch.minimum(torch.maximum(u, -motors_effort_limit), motors_effort_limit))


def pd_adapter(action: torch.Tensor, order: int, command_state: torch.Tensor,
               command_state_lower: torch.Tensor, command_state_upper: torch.Tensor,
               is_instantaneous: bool, motors_velocity_deadband: Optional[torch.Tensor],
               step_dt: float, out: torch.Tensor) -> None:
    """≙ `pd_adapter` (proportional_derivative_controller.py:166-260): target accelerations
    `[M][B]` to hold over `step_dt` so that the `order`-th derivative of the target reaches `action`."""
    if abs(step_dt) < 1e-9:
        return
    lo, hi = command_state_lower, command_state_upper
    if lo.dim() == 2:
        lo, hi = lo[..., None], hi[..., None]
    db = motors_velocity_deadband
    if db is not None and db.dim() == 1:
        db = db[:, None]
    if is_instantaneous:
        if order == 0:
            velocity = (action - command_state[0]) / step_dt
            velocity = torch.minimum(torch.maximum(velocity, lo[1]), hi[1])
            if db is not None:
                velocity = torch.where(velocity.abs() < db, torch.zeros_like(velocity), velocity)
            command_state[0].add_(velocity * step_dt)
            command_state[1].zero_()
        else:
            if db is not None:
                action = action * (action.abs() > db)
            acceleration = (action - command_state[1]) / step_dt
            acceleration = torch.minimum(torch.maximum(acceleration, lo[2]), hi[2])
            command_state[1].add_(acceleration * step_dt)
        out.zero_()
    else:
        velocity = (action - command_state[0]) / step_dt if order == 0 else action
        velocity = torch.minimum(torch.maximum(velocity, lo[1]), hi[1])
        if db is not None:
            velocity = torch.where(velocity.abs() < db, torch.zeros_like(velocity), velocity)
        out.copy_((velocity - command_state[1]) / step_dt)


def compute_tilt_from_quat(q: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """R(q)^T e_z for quaternions `[4][...]` xyzw (reference utils/math.py:1046-1060)."""
    q_x, q_y, q_z, q_w = q[0], q[1], q[2], q[3]
    return (2 * (q_x * q_z - q_y * q_w), 2 * (q_y * q_z + q_w * q_x),
            1 - 2 * (q_x * q_x + q_y * q_y))


def mahony_filter(q: torch.Tensor, omega: torch.Tensor, cf: torch.Tensor, gyro: torch.Tensor,
                  acc: torch.Tensor, bias_hat: torch.Tensor, kp: float, ki: float, dt: float
                  ) -> None:
    """≙ `mahony_filter` (mahony_filter.py:28-101), in place; `q` `[4][K]`, the others `[3][K]`
    with K independent filters (`[...][B]`, or `[...][n_imu][B]` for several IMUs per environment).

    The reference returns early when *no* IMU moves (`(|cf| < 1e-6).all()`); per-lane here: a lane
    whose `cf` is below that threshold keeps its orientation and bias (same values, since its
    update would be the identity up to 1e-6 * dt)."""
    v_x, v_y, v_z = compute_tilt_from_quat(q)
    omega.copy_(gyro - bias_hat)
    a_hat = acc / EARTH_SURFACE_GRAVITY
    omega_mes = torch.stack((a_hat[1] * v_z - a_hat[2] * v_y,
                             a_hat[2] * v_x - a_hat[0] * v_z,
                             a_hat[0] * v_y - a_hat[1] * v_x), 0)
    cf.copy_(omega + kp * omega_mes)
    still = cf.abs() < 1e-6
    if cf.dim() == 3:   # [3][n_imu][B]: the early return of the reference is per environment
        moving = (~still.all(dim=0).all(dim=0))[None, :].expand(cf.shape[1], -1)
    else:
        moving = ~still.all(dim=0)
    theta = torch.sqrt((cf * cf).sum(0))
    safe_theta = torch.where(moving, theta, torch.ones_like(theta))
    axis = cf / safe_theta
    half = safe_theta * (dt / 2)
    p = axis * torch.sin(half)
    p_w = torch.cos(half)
    q_x, q_y, q_z, q_w = q[0].clone(), q[1].clone(), q[2].clone(), q[3].clone()
    n = torch.stack((q_x * p_w + q_w * p[0] - q_z * p[1] + q_y * p[2],
                     q_y * p_w + q_z * p[0] + q_w * p[1] - q_x * p[2],
                     q_z * p_w - q_y * p[0] + q_x * p[1] + q_w * p[2],
                     q_w * p_w - q_x * p[0] - q_y * p[1] - q_z * p[2]), 0)
    n = n * ((3.0 - (n * n).sum(0)) / 2)
    q.copy_(torch.where(moving, n, q))
    bias_hat.copy_(torch.where(moving, bias_hat - ki * dt * omega_mes, bias_hat))
