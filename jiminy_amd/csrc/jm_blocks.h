// jm_blocks.h -- the gym_jiminy pipeline blocks of the ANYmal / Atlas environments as batched HIP
// kernels: every lane is one environment, arrays are `[rows][B]` like the physics state.
//
// Reference (numba kernels restated here, scalar branches become per-lane selects):
//   integrate_zoh   python/gym_jiminy/common/gym_jiminy/common/blocks/proportional_derivative_controller.py:23-98
//   pd_controller   proportional_derivative_controller.py:101-163
//   mahony_filter   python/gym_jiminy/common/gym_jiminy/common/blocks/mahony_filter.py:28-101
//   compute_tilt_from_quat  python/gym_jiminy/common/gym_jiminy/common/utils/math.py:1046-1060
// One launch replaces the ~40 elementwise tensor kernels the same block costs as a tensor
// program; the blocks are HBM-bound (PD: 96 scalars per lane in, 48 out for 12 motors).
#pragma once
#include <hip/hip_runtime.h>

#include "jm_math.h"
#include "../../include/jiminy_hip.h"

namespace jm
{
struct PdParams
{
    int M, n_enc;
    int enc_index[JM_BLOCK_MAX_MOTORS];
    double lo[3][JM_BLOCK_MAX_MOTORS], hi[3][JM_BLOCK_MAX_MOTORS];
    double kp[JM_BLOCK_MAX_MOTORS], kd[JM_BLOCK_MAX_MOTORS], effort_limit[JM_BLOCK_MAX_MOTORS];
    double dt;
};

// encoder raw field: [n_enc][2][B]; command_state: [3][M][B]; out: [M][B]
template<class T>
__global__ void __launch_bounds__(256) k_pd_controller(const PdParams p, const T * enc, T * cs, T * out, long long B)
{
    const long long lane = (long long)blockIdx.x * 256 + threadIdx.x;
    if (lane >= B) return;
    const T dt = (T)p.dt;
    for (int m = 0; m < p.M; ++m)
    {
        T * ps = cs + (long long)m * B + lane;
        T * vs = ps + (long long)p.M * B;
        T * as = vs + (long long)p.M * B;
        T position = *ps, velocity = *vs;
        if (fabs(p.dt) >= 1e-9)
        {
            const T p_min = (T)p.lo[0][m], v_min = (T)p.lo[1][m], a_min = (T)p.lo[2][m];
            const T p_max = (T)p.hi[0][m], v_max = (T)p.hi[1][m], a_max = (T)p.hi[2][m];
            const T acc = fmin_(fmax_(*as, a_min), a_max);
            const T v_prev = velocity;
            T vel = fmin_(fmax_(velocity + acc * dt, v_min), v_max);
            // slow down early enough not to violate the acceleration limit at the position bounds
            const T absv = v_prev < T(0) ? -v_prev : v_prev;
            const T horizon = fmax_(trunc_(absv / a_max / dt) * dt, dt);
            T d_min = p_min - position, d_max = p_max - position;
            if (horizon > dt)
            {
                const T drift = T(0.5) * (horizon * (horizon - dt)) * a_max;
                d_min -= drift;
                d_max += drift;
            }
            vel = fmin_(fmax_(vel, d_min / horizon), d_max / horizon);
            // velocity after hitting the bounds must be cancellable in a single step
            const T absvel = vel < T(0) ? -vel : vel;
            if (absvel > dt * a_max)
            {
                const T lo = -fmax_(d_min / vel, dt) * a_max;
                const T hi = fmax_(d_max / vel, dt) * a_max;
                vel = fmin_(fmax_(vel, lo), hi);
            }
            *as = (vel - v_prev) / dt;
            velocity = vel;
            position = position + dt * vel;
            *vs = velocity;
            *ps = position;
        }
        const long long e = (long long)p.enc_index[m] * 2 * B + lane;
        const T q_error = position - enc[e];
        const T v_error = velocity - enc[e + B];
        const T u = (T)p.kp[m] * (q_error + (T)p.kd[m] * v_error);
        const T lim = (T)p.effort_limit[m];
        out[(long long)m * B + lane] = fmin_(fmax_(u, -lim), lim);
    }
}

// imu raw field: [n_imu][6][B] (gyro 0-2, accel 3-5); quat: [4][n_imu][B]; omega, cf, bias: [3][n_imu][B]
template<class T>
__global__ void __launch_bounds__(256) k_mahony(int n_imu, const T * imu, T * quat, T * omega, T * cf, T * bias,
                                                 double kp_, double ki_, double dt_, long long B)
{
    const long long lane = (long long)blockIdx.x * 256 + threadIdx.x;
    if (lane >= B) return;
    const T kp = (T)kp_, ki = (T)ki_, dt = (T)dt_;
    const long long nB = (long long)n_imu * B;
    // pass 1: omega, cf; the reference returns early when no IMU of the environment moves
    bool moving = false;
    for (int s = 0; s < n_imu; ++s)
    {
        const long long o = (long long)s * B + lane;
        const T qx = quat[o], qy = quat[nB + o], qz = quat[2 * nB + o], qw = quat[3 * nB + o];
        const T vx = T(2) * (qx * qz - qy * qw), vy = T(2) * (qy * qz + qw * qx), vz = T(1) - T(2) * (qx * qx + qy * qy);
        const T * g = imu + (long long)s * 6 * B + lane;
        const T ax = g[3 * B] / T(9.81), ay = g[4 * B] / T(9.81), az = g[5 * B] / T(9.81);
        const T mx = ay * vz - az * vy, my = az * vx - ax * vz, mz = ax * vy - ay * vx;
        const T ox = g[0] - bias[o], oy = g[B] - bias[nB + o], oz = g[2 * B] - bias[2 * nB + o];
        omega[o] = ox; omega[nB + o] = oy; omega[2 * nB + o] = oz;
        const T cx = ox + kp * mx, cy = oy + kp * my, cz = oz + kp * mz;
        cf[o] = cx; cf[nB + o] = cy; cf[2 * nB + o] = cz;
        const T eps = T(1e-6);
        moving |= !((cx < T(0) ? -cx : cx) < eps && (cy < T(0) ? -cy : cy) < eps && (cz < T(0) ? -cz : cz) < eps);
    }
    if (!moving) return;
    // pass 2: integrate the orientation, update the bias estimate
    for (int s = 0; s < n_imu; ++s)
    {
        const long long o = (long long)s * B + lane;
        const T cx = cf[o], cy = cf[nB + o], cz = cf[2 * nB + o];
        const T theta = sqrt_(cx * cx + cy * cy + cz * cz);
        const T half = theta * (dt / T(2));
        T sh, ch;
        sincos_(half, &sh, &ch);
        const T px = cx / theta * sh, py = cy / theta * sh, pz = cz / theta * sh, pw = ch;
        const T qx = quat[o], qy = quat[nB + o], qz = quat[2 * nB + o], qw = quat[3 * nB + o];
        T nx = qx * pw + qw * px - qz * py + qy * pz;
        T ny = qy * pw + qz * px + qw * py - qx * pz;
        T nz = qz * pw - qy * px + qx * py + qw * pz;
        T nw = qw * pw - qx * px - qy * py - qz * pz;
        const T k = (T(3) - (nx * nx + ny * ny + nz * nz + nw * nw)) / T(2);
        quat[o] = nx * k; quat[nB + o] = ny * k; quat[2 * nB + o] = nz * k; quat[3 * nB + o] = nw * k;
        const T * g = imu + (long long)s * 6 * B + lane;
        const T vx = T(2) * (qx * qz - qy * qw), vy = T(2) * (qy * qz + qw * qx), vz = T(1) - T(2) * (qx * qx + qy * qy);
        const T ax = g[3 * B] / T(9.81), ay = g[4 * B] / T(9.81), az = g[5 * B] / T(9.81);
        const T mx = ay * vz - az * vy, my = az * vx - ax * vz, mz = ax * vy - ay * vx;
        bias[o] -= ki * dt * mx; bias[nB + o] -= ki * dt * my; bias[2 * nB + o] -= ki * dt * mz;
    }
}
// ---- PDAdapter: `pd_adapter` (proportional_derivative_controller.py:166-260), once per environment step.
// action [M][B]; command_state [3][M][B] (position, velocity, acceleration targets); out [M][B] = the target
// acceleration the PD controller holds over the step.  deadband[m] < 0: none.
struct PdAdapterParams
{
    int M, order, instantaneous;
    double lo[3][JM_BLOCK_MAX_MOTORS], hi[3][JM_BLOCK_MAX_MOTORS], deadband[JM_BLOCK_MAX_MOTORS];
    double dt;
};
template<class T>
__global__ void __launch_bounds__(256) k_pd_adapter(const PdAdapterParams p, const T * action, T * cs, T * out, long long B)
{
    const long long lane = (long long)blockIdx.x * 256 + threadIdx.x;
    if (lane >= B || fabs(p.dt) < 1e-9) return;
    const T dt = (T)p.dt;
    for (int m = 0; m < p.M; ++m)
    {
        T * ps = cs + (long long)m * B + lane;
        T * vs = ps + (long long)p.M * B;
        T act = action[(long long)m * B + lane];
        const bool db = p.deadband[m] >= 0.0;
        const T dbv = (T)p.deadband[m];
        T res = T(0);
        if (p.instantaneous)
        {
            if (p.order == 0)
            {
                T vel = (act - *ps) / dt;
                vel = fmin_(fmax_(vel, (T)p.lo[1][m]), (T)p.hi[1][m]);
                if (db && (vel < T(0) ? -vel : vel) < dbv) vel = T(0);
                *ps += vel * dt;
                *vs = T(0);
            }
            else
            {
                if (db) act = ((act < T(0) ? -act : act) > dbv) ? act : act * T(0);
                T acc = (act - *vs) / dt;
                acc = fmin_(fmax_(acc, (T)p.lo[2][m]), (T)p.hi[2][m]);
                *vs += acc * dt;
            }
        }
        else
        {
            T vel = p.order == 0 ? (act - *ps) / dt : act;
            vel = fmin_(fmax_(vel, (T)p.lo[1][m]), (T)p.hi[1][m]);
            if (db && (vel < T(0) ? -vel : vel) < dbv) vel = T(0);
            res = (vel - *vs) / dt;
        }
        out[(long long)m * B + lane] = res;
    }
}

// ---- MotorSafetyLimit: `apply_safety_limits` (blocks/motor_safety_limit.py:20-77).  encoder raw field
// [n_enc][2][B]; command / out [M][B].
struct SafetyParams
{
    int M;
    int enc_index[JM_BLOCK_MAX_MOTORS];
    double kp[JM_BLOCK_MAX_MOTORS], kd[JM_BLOCK_MAX_MOTORS], soft_lo[JM_BLOCK_MAX_MOTORS], soft_hi[JM_BLOCK_MAX_MOTORS],
        vel_lim[JM_BLOCK_MAX_MOTORS], eff_lim[JM_BLOCK_MAX_MOTORS];
};
template<class T>
__global__ void __launch_bounds__(256) k_motor_safety_limit(const SafetyParams p, const T * enc, const T * command, T * out, long long B)
{
    const long long lane = (long long)blockIdx.x * 256 + threadIdx.x;
    if (lane >= B) return;
    for (int m = 0; m < p.M; ++m)
    {
        const long long e = (long long)p.enc_index[m] * 2 * B + lane;
        const T q = enc[e], v = enc[e + B];
        const T kp = (T)p.kp[m], kd = (T)p.kd[m], vl = (T)p.vel_lim[m], el = (T)p.eff_lim[m];
        const T v_lo = vl * fmin_(fmax_(-kp * (q - (T)p.soft_lo[m]), T(-1)), T(1));
        const T v_hi = vl * fmin_(fmax_(-kp * (q - (T)p.soft_hi[m]), T(-1)), T(1));
        const T e_lo = el * fmin_(fmax_(-kd * (v - v_lo), T(-1)), T(1));
        const T e_hi = el * fmin_(fmax_(-kd * (v - v_hi), T(-1)), T(1));
        out[(long long)m * B + lane] = fmin_(fmax_(command[(long long)m * B + lane], e_lo), e_hi);
    }
}
}  // namespace jm
