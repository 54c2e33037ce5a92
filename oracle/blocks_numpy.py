"""Scalar numpy restatement of the gym_jiminy pipeline blocks -- TEST INFRASTRUCTURE ONLY.

Follows, statement by statement and one motor / one IMU at a time, the numba kernels of the
reference: `integrate_zoh`, `pd_controller`, `pd_adapter`
(python/gym_jiminy/common/gym_jiminy/common/blocks/proportional_derivative_controller.py:23-98,
101-163, 166-260) and `mahony_filter` (mahony_filter.py:28-101, with `compute_tilt_from_quat`
of utils/math.py:1046-1060).  Arrays are for ONE environment, as in the reference.
"""
import numpy as np

EARTH_SURFACE_GRAVITY = 9.81


def integrate_zoh(state, state_min, state_max, dt):
    assert dt >= 0.0
    if abs(dt) < 1e-9:
        return
    position, velocity, acceleration = state
    _, dim = state.shape
    for i in range(dim):
        position_min, velocity_min, acceleration_min = state_min[:, i]
        position_max, velocity_max, acceleration_max = state_max[:, i]
        acceleration[i] = min(max(acceleration[i], acceleration_min), acceleration_max)
        velocity_prev = velocity[i]
        velocity[i] += acceleration[i] * dt
        velocity[i] = min(max(velocity[i], velocity_min), velocity_max)
        horizon = max(int(abs(velocity_prev) / acceleration_max / dt) * dt, dt)
        position_min_delta = position_min - position[i]
        position_max_delta = position_max - position[i]
        if horizon > dt:
            drift = 0.5 * (horizon * (horizon - dt)) * acceleration_max
            position_min_delta -= drift
            position_max_delta += drift
        velocity_min = position_min_delta / horizon
        velocity_max = position_max_delta / horizon
        velocity[i] = min(max(velocity[i], velocity_min), velocity_max)
        if np.abs(velocity[i]) > dt * acceleration_max:
            velocity_min = - max(position_min_delta / velocity[i], dt) * acceleration_max
            velocity_max = max(position_max_delta / velocity[i], dt) * acceleration_max
            velocity[i] = min(max(velocity[i], velocity_min), velocity_max)
        acceleration[i] = (velocity[i] - velocity_prev) / dt
        position[i] += dt * velocity[i]


def pd_controller(encoder_data, command_state, command_state_lower, command_state_upper, kp, kd,
                  motors_effort_limit, control_dt, out):
    integrate_zoh(command_state, command_state_lower, command_state_upper, control_dt)
    q_error, v_error = command_state[:2] - encoder_data
    out[:] = kp * (q_error + kd * v_error)
    out[:] = np.minimum(np.maximum(out, -motors_effort_limit), motors_effort_limit)


def pd_adapter(action, order, command_state, command_state_lower, command_state_upper,
               is_instantaneous, motors_velocity_deadband, step_dt, out):
    if abs(step_dt) < 1e-9:
        return
    if is_instantaneous:
        if order == 0:
            velocity = (action - command_state[0]) / step_dt
            velocity = np.minimum(np.maximum(velocity, command_state_lower[1]), command_state_upper[1])
            if motors_velocity_deadband is not None:
                velocity[np.abs(velocity) < motors_velocity_deadband] = 0.0
            command_state[0] += velocity * step_dt
            command_state[1] = 0.0
        else:
            if motors_velocity_deadband is not None:
                action = action * (np.abs(action) > motors_velocity_deadband)
            acceleration = (action - command_state[1]) / step_dt
            acceleration = np.minimum(np.maximum(acceleration, command_state_lower[2]), command_state_upper[2])
            command_state[1] += acceleration * step_dt
        out[:] = 0.0
    else:
        if order == 0:
            velocity = (action - command_state[0]) / step_dt
        else:
            velocity = action
        velocity = np.minimum(np.maximum(velocity, command_state_lower[1]), command_state_upper[1])
        if motors_velocity_deadband is not None:
            velocity[np.abs(velocity) < motors_velocity_deadband] = 0.0
        out[:] = (velocity - command_state[1]) / step_dt


def compute_tilt_from_quat(q):
    q_x, q_y, q_z, q_w = q
    return (2 * (q_x * q_z - q_y * q_w), 2 * (q_y * q_z + q_w * q_x), 1 - 2 * (q_x * q_x + q_y * q_y))


def mahony_filter(q, omega, cf, gyro, acc, bias_hat, kp, ki, dt):
    v_x, v_y, v_z = compute_tilt_from_quat(q)
    omega[:] = gyro - bias_hat
    v_x_hat, v_y_hat, v_z_hat = acc / EARTH_SURFACE_GRAVITY
    omega_mes = np.stack((v_y_hat * v_z - v_z_hat * v_y,
                          v_z_hat * v_x - v_x_hat * v_z,
                          v_x_hat * v_y - v_y_hat * v_x), 0)
    cf[:] = omega + kp * omega_mes
    if (np.abs(cf) < 1e-6).all():
        return
    theta = np.sqrt(np.sum(cf * cf, 0))
    axis = cf / theta
    theta = theta * (dt / 2)
    (p_x, p_y, p_z), p_w = (axis * np.sin(theta)), np.cos(theta)
    q_x, q_y, q_z, q_w = q.copy()
    q[0], q[1], q[2], q[3] = (
        q_x * p_w + q_w * p_x - q_z * p_y + q_y * p_z,
        q_y * p_w + q_z * p_x + q_w * p_y - q_x * p_z,
        q_z * p_w - q_y * p_x + q_x * p_y + q_w * p_z,
        q_w * p_w - q_x * p_x - q_y * p_y - q_z * p_z)
    q *= (3.0 - np.sum(np.square(q), 0)) / 2
    bias_hat -= ki * dt * omega_mes


def apply_safety_limits(command, q_measured, v_measured, kp, kd, motors_soft_position_lower,
                        motors_soft_position_upper, motors_velocity_limit, motors_effort_limit, out):
    """blocks/motor_safety_limit.py:20-77, statement by statement."""
    safe_velocity_lower = motors_velocity_limit * np.minimum(np.maximum(
        -kp * (q_measured - motors_soft_position_lower), -1.0), 1.0)
    safe_velocity_upper = motors_velocity_limit * np.minimum(np.maximum(
        -kp * (q_measured - motors_soft_position_upper), -1.0), 1.0)
    safe_effort_lower = motors_effort_limit * np.minimum(np.maximum(
        -kd * (v_measured - safe_velocity_lower), -1.0), 1.0)
    safe_effort_upper = motors_effort_limit * np.minimum(np.maximum(
        -kd * (v_measured - safe_velocity_upper), -1.0), 1.0)
    out[:] = np.minimum(np.maximum(command, safe_effort_lower), safe_effort_upper)
