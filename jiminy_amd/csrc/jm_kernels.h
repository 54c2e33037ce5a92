// jm_kernels.h -- the batched per-step physics, one robot per wavefront lane (gfx950).
//
// The code is specialised at compile time on the robot TOPOLOGY (`Tp`: parents, joint types,
// index maps, which joints carry motors / contact points / sensors -- a generated header, see
// jiminy_amd/codegen.py) and reads every numeric PARAMETER (placements, inertias, limits,
// gains, contact options) from a small block in the constant address space, i.e. through the
// scalar cache into SGPRs: parameters are lane-uniform, state is per lane.
//
// Reference functions restated here (paths relative to the reference tree):
//   eval_dynamics    Engine::computeRobotsDynamics              core/src/engine/engine.cc:3585-3708
//     FK             Engine::computeForwardKinematics           engine.cc:2957-3014
//     contacts       computeContactDynamicsAtFrame / computeContactDynamics   engine.cc:3117-3238
//                    convertForceGlobalFrameToJoint              core/src/utilities/pinocchio.cc:794-809
//     motors         SimpleMotor::computeEffort                 core/src/hardware/basic_motors.cc:83-143
//     ABA            pinocchio_overload::aba / AbaBackwardStep  core/include/jiminy/core/robot/pinocchio_overload_algorithms.h:126-489
//   integrate_q      State::sum -> pinocchio::integrate         core/include/jiminy/core/stepper/lie_group.h:446-455
//   lane_run (step)  AbstractStepper::tryStep + RK4 / Euler     core/src/stepper/abstract_stepper.cc:15-62,
//                    abstract_runge_kutta_stepper.cc:24-77, euler_explicit_stepper.cc:5-21,
//                    tableau core/include/jiminy/core/stepper/runge_kutta4_stepper.h:12-23
//   extra_terms      computeExtraTerms                          engine.cc:800-905
//   write_sensors    Imu/Contact/Force/Encoder/EffortSensor::set core/src/hardware/basic_sensors.cc:142-164,
//                    267-277, 368-387, 509-539, 604-618
#pragma once
#include <cstring>
#include <type_traits>

#include "jm_math.h"
#include "../../include/jiminy_hip.h"

namespace jm
{
template<int I, int N, class F> JM_DEV void static_for(F && f)
{
    if constexpr (I < N)
    {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// I = N-1 ... LO
template<int LO, int N, class F> JM_DEV void static_rfor(F && f)
{
    if constexpr (N > LO)
    {
        f(std::integral_constant<int, N - 1>{});
        static_rfor<LO, N - 1>(f);
    }
}

constexpr bool jt_is_rev(int t) { return (t >= JM_JT_RX && t <= JM_JT_RU) || (t >= JM_JT_RUBX && t <= JM_JT_RUBU); }
constexpr bool jt_is_pri(int t) { return t >= JM_JT_PX && t <= JM_JT_PU; }
constexpr bool jt_is_unb(int t) { return t >= JM_JT_RUBX && t <= JM_JT_RUBU; }
constexpr bool jt_bounded(int t) { return t >= JM_JT_RX && t <= JM_JT_PU; }
constexpr bool jt_is_sph(int t) { return t == JM_JT_SPHERICAL; }
constexpr int jt_nv(int t) { return t == JM_JT_FREEFLYER ? 6 : (t == JM_JT_SPHERICAL ? 3 : (t == JM_JT_NONE ? 0 : 1)); }
// spherical (flexibility) joints of a topology: how many, and the rank of joint j among them
template<class Tp> constexpr int n_spherical()
{
    int n = 0;
    for (int j = 0; j < Tp::NJ; ++j) n += Tp::jtype[j] == JM_JT_SPHERICAL ? 1 : 0;
    return n;
}
template<class Tp> constexpr int spherical_rank(int j)
{
    int n = 0;
    for (int i = 0; i < j; ++i) n += Tp::jtype[i] == JM_JT_SPHERICAL ? 1 : 0;
    return n;
}
constexpr int jt_axis(int t)  // 0,1,2 aligned; -1 unaligned
{
    return (t == JM_JT_RX || t == JM_JT_PX || t == JM_JT_RUBX) ? 0
         : (t == JM_JT_RY || t == JM_JT_PY || t == JM_JT_RUBY) ? 1
         : (t == JM_JT_RZ || t == JM_JT_PZ || t == JM_JT_RUBZ) ? 2 : -1;
}
constexpr int c_max(int a, int b) { return a > b ? a : b; }

// ---------------------------------------------------------------- parameter block layout
// Host packing: jiminy_amd/csrc/jm_lib.cpp pack_params(); same offsets.
template<class Tp> struct Layout
{
    static constexpr int JSTRIDE = 25;  // R9 p3 | mass c3 I6(xx xy xz yy yz zz) | axis3
    static constexpr int JOINT = 0;
    static constexpr int ROTOR = JOINT + Tp::NJ * JSTRIDE;
    static constexpr int QLO = ROTOR + Tp::NV;
    static constexpr int QHI = QLO + Tp::NQ;
    static constexpr int MOTOR = QHI + Tp::NQ;  // NM * 9
    static constexpr int CONTACT = MOTOR + Tp::NM * JM_MOTOR_NPARAMS;  // NC * 12
    static constexpr int IMU = CONTACT + Tp::NC * 12;                  // NIMU * 12
    static constexpr int FREL = IMU + Tp::NIMU * 12;                   // NFORCE * NC * 12
    static constexpr int ENC = FREL + Tp::NFORCE * Tp::NC * 12;        // NENC
    static constexpr int XFRAME = ENC + Tp::NENC;                      // NX * 12: user constraint frames (R9 p3)
    static constexpr int XPAR = XFRAME + Tp::NX * 12;                  // NX * 8: radius, normal 3, axis 3 | -, second frame position 3
    static constexpr int FLEX = XPAR + Tp::NX * 8;                     // 6 per spherical joint: stiffness 3, damping 3
    static constexpr int OPT = FLEX + 6 * n_spherical<Tp>();  // gravity6 k c mu eps vt
    static constexpr int TOTAL = OPT + 11;
};

#ifdef JM_HOST_EMU
template<class T> using CPtr = const T *;
#else
template<class T> using CPtr = const T __attribute__((address_space(4))) *;
#endif

template<class T> JM_DEV M3<T> ld_m3(CPtr<T> P, int o)
{
    return {P[o], P[o + 1], P[o + 2], P[o + 3], P[o + 4], P[o + 5], P[o + 6], P[o + 7], P[o + 8]};
}
template<class T> JM_DEV V3<T> ld_v3(CPtr<T> P, int o) { return {P[o], P[o + 1], P[o + 2]}; }
template<class T> JM_DEV SE3<T> ld_se3(CPtr<T> P, int o) { return {ld_m3<T>(P, o), ld_v3<T>(P, o + 9)}; }
template<class T> JM_DEV RBI<T> ld_rbi(CPtr<T> P, int o)
{
    return {P[o], ld_v3<T>(P, o + 1), S3<T>{P[o + 4], P[o + 5], P[o + 6], P[o + 7], P[o + 8], P[o + 9]}};
}

// ---------------------------------------------------------------- kernel arguments
template<class T> struct BatchArgs
{
    const T * P;           // parameter block (device, Layout<Tp>::TOTAL scalars)
    T * q; T * v; T * a;   // state, [rows][B]
    const T * command;
    T * u_motor; T * u;
    T * f_external; T * contact_forces;
    T * imu; T * force; T * contact; T * encoder; T * effort;
    T * energy; T * joint_forces; T * centroidal;
    int32_t * status;
    T * ws;                // workspace rows (unused by the LDS variant, kept for large models)
    const T * q_in; const T * v_in; T * a_out;                     // MODE_DYNAMICS
    const unsigned char * mask; const T * q_init; const T * v_init;  // MODE_RESET
    long long B;
    int mode, solver, n_sub, command_changed, update_sensors;
    T dt;
    // ---- optional per-environment variation (branch-parallel kernels, GEN instantiation; all null / 0 otherwise)
    // per-lane body parameters, rows per joint: mass | com 3 | inertia xx xy xz yy yz zz | joint placement
    // translation 3 = `[13 * NJ][B]` (Model::addBiasedToExtendedModel, model.cc:1166-1236)
    const T * model_lane;
    // world.groundProfile as a height map `[ny][nx]` at (x0 + ix dx, y0 + iy dy), bilinear patches (engine.h:292-302)
    const T * ground_h;
    int ground_nx, ground_ny;
    T ground_x0, ground_y0, ground_dx, ground_dy;
    // impulse / profile forces on frames of the root joint (engine.cc:1838-2016): world-aligned (force, moment)
    // `[6 K][B]` at the frame offsets `applied_p` (root joint frame), K <= 4
    const T * applied;
    int applied_k;
    T applied_p[12];
    int applied_joint[4];   // parent joint of every frame: the wrench goes to that joint (engine.cc:3481-3560)
    // `[1][B]` ground friction coefficient of every lane (spring-damper model; the constraint model reads its own
    // copy, QConArgs / ConArgs), or null: `contacts.friction` randomised per environment (envs/locomotion.py:257-262)
    const T * friction;
    const T * ground_off;        // [2][B] per-lane (x, y) offset of the height-map queries (JM_F_GROUND_OFFSET), or null
    // compact batches of the per-stage adaptive stepper (jm_adaptive.h): lane c of the launch is lane lane_map[c] of the
    // batch of B_full lanes -- the per-lane OPTIONAL inputs (model_lane, friction, applied, ground_off) stay in batch order
    const int32_t * lane_map;
    long long B_full;
    // `[6 per spherical joint][B]` stiffness 3, damping 3 of the flexibility joints of every lane (JM_F_FLEXIBILITY), or null:
    // `flexibilityConfig` randomised per environment (envs/locomotion.py:288-296); one-robot-per-lane kernels
    const T * flex_lane;
};
// MODE_REFRESH: evaluate at the bound state and emit the outputs (sensors if `update_sensors`), OR-ing
// the lane status into the existing one: the closing launch of an adaptive-step interval
enum { MODE_STEP = 0, MODE_START = 1, MODE_DYNAMICS = 2, MODE_RESET = 3, MODE_REFRESH = 4 };

// stride between the rows of a lane's on-chip buffer (stage rows + stash): the 64 lanes of the block interleave
#ifdef JM_HOST_EMU
constexpr int LANE_STRIDE = 1;
#else
constexpr int LANE_STRIDE = 64;
#endif
// ---------------------------------------------------------------- per-lane working set
template<class T, class Tp> struct Work
{
    SE3<T> liMi[Tp::NJ];
    SE3<T> oMi[Tp::NJ];
    Sp<T> vel[Tp::NJ];
    Sp<T> agf[Tp::NJ];
    Sp<T> f[Tp::NJ];
    Sp<T> fext[Tp::NJ];
    Sp<T> cf[c_max(Tp::NC, 1)];  // contact forces, contact frame (Robot::contactForces_)
    Sp<T> U[Tp::NJ];
    T dinv[Tp::NJ];
    T u[Tp::NV];
    T ueff[Tp::NV];              // total effort vector (RobotState::u)
    T umotor[c_max(Tp::NM, 1)];
    T ddq[Tp::NV];
    // joint coordinate of every 1-dof joint as (cos, sin) or (displacement, -): the sweeps rebuild liMi from
    // these 2 scalars + the constant placement (scalar loads) instead of keeping 12 scalars per joint alive
    T jcs[Tp::NJ][2];
    // spherical joints: what the backward sweep leaves for the forward one (U = [B; D] of the articulated inertia at the
    // visit, the inverse of D + rotor inertia, the reduced effort)
    M3<T> sphB[c_max(n_spherical<Tp>(), 1)];
    S3<T> sphD[c_max(n_spherical<Tp>(), 1)], sphDinv[c_max(n_spherical<Tp>(), 1)];
    V3<T> sphu[c_max(n_spherical<Tp>(), 1)];
    // the lane's column of the sweeps' stash (LDS on the device, see eval_aba): element r at stash[r * LANE_STRIDE]
    T * stash;
    int status;
    // per-environment variation (ground friction, flexibility parameters, applied wrenches of the lane: BatchArgs::friction /
    // flex_lane / applied): read where they are used, through the launch arguments (uniform: scalar registers) and the lane
    // index (alive anyway) -- a pointer per lane kept across the evaluations costs the sweeps their registers (7-joint arm:
    // 0.099 -> 0.119 ms per launch with five such members here)
    const BatchArgs<T> * args;
    long long lane;
    static constexpr bool CONSTRAINED = false;
    // applied wrenches (BatchArgs::applied) and the lane's own body parameters (BatchArgs::model_lane) are read by the VARIATION
    // instantiations (WorkA / WorkCA, `k_batch<..., true>`, `k_constrained<..., true>`): with applied wrenches the external
    // force of EVERY joint is a run-time quantity, without them the joints that carry no contact point have none and the sweeps
    // lose those terms at compile time (7-joint arm: 0.099 against 0.121 ms per launch with the test in the only instantiation)
    static constexpr bool APPLIED = false;
};
template<class T, class Tp> struct WorkA : Work<T, Tp> { static constexpr bool APPLIED = true; };
// working set of the constraint contact model (jm_constraint.h): keeps the factorised root block
template<class T, class Tp> struct WorkC : Work<T, Tp>
{
    T rootA[6][6];   // LDL^T factor of (Ia_root + rotor) left by chol6_solve (unit lower + diagonal)
    T rootdinv[6];
    static constexpr bool CONSTRAINED = true;
};
template<class T, class Tp> struct WorkCA : WorkC<T, Tp> { static constexpr bool APPLIED = true; };
// evaluation policy of lane_run: plain (spring-damper contacts) or constraint contact model
template<class T> struct NoConArgs {};
struct NoCon
{
    static constexpr bool ON = false;
    template<class T, class Tp> using WorkT = Work<T, Tp>;
    template<class T> using ArgsT = NoConArgs<T>;
};
struct NoConA : NoCon
{
    template<class T, class Tp> using WorkT = WorkA<T, Tp>;   // ... with applied wrenches
};

template<class T, class Tp, int J> JM_DEV V3<T> joint_axis(CPtr<T> P)
{
    constexpr int ax = jt_axis(Tp::jtype[J]);
    if constexpr (ax == 0) return {T(1), T(0), T(0)};
    else if constexpr (ax == 1) return {T(0), T(1), T(0)};
    else if constexpr (ax == 2) return {T(0), T(0), T(1)};
    else return ld_v3<T>(P, Layout<Tp>::JOINT + J * Layout<Tp>::JSTRIDE + 22);
}

// joint transform M_j(q) and joint velocity S qd
// `cs`: the joint coordinate as the sweeps cache it -- (cos, sin) of a revolute joint, (displacement, 0) of a
// prismatic one
template<class T, class Tp, int J>
JM_DEV void joint_calc(CPtr<T> P, const T * q, const T * v, SE3<T> & Mj, Sp<T> & vj, T (&cs)[2])
{
    constexpr int t = Tp::jtype[J];
    constexpr int iq = Tp::idx_q[J], iv = Tp::idx_v[J];
    if constexpr (t == JM_JT_FREEFLYER)
    {
        Mj.R = quat_to_matrix(q[iq + 3], q[iq + 4], q[iq + 5], q[iq + 6]);
        Mj.p = {q[iq], q[iq + 1], q[iq + 2]};
        vj = {{v[iv], v[iv + 1], v[iv + 2]}, {v[iv + 3], v[iv + 4], v[iv + 5]}};
        cs[0] = T(0); cs[1] = T(0);
    }
    else if constexpr (jt_is_sph(t))
    {
        Mj.R = quat_to_matrix(q[iq], q[iq + 1], q[iq + 2], q[iq + 3]);
        Mj.p = zero3<T>();
        vj = {zero3<T>(), {v[iv], v[iv + 1], v[iv + 2]}};
        cs[0] = T(0); cs[1] = T(0);
    }
    else if constexpr (jt_is_rev(t))
    {
        T c, s;
        if constexpr (jt_is_unb(t)) { c = q[iq]; s = q[iq + 1]; }
        else sincos_(q[iq], &s, &c);
        constexpr int ax = jt_axis(t);
        const V3<T> n = joint_axis<T, Tp, J>(P);
        if constexpr (ax >= 0) Mj.R = rot_axis<T>(ax, c, s);
        else Mj.R = rot_rodrigues(n, c, s);
        Mj.p = zero3<T>();
        vj = {zero3<T>(), v[iv] * n};
        cs[0] = c; cs[1] = s;
    }
    else
    {
        const V3<T> n = joint_axis<T, Tp, J>(P);
        Mj.R = ident3<T>();
        Mj.p = q[iq] * n;
        vj = {v[iv] * n, zero3<T>()};
        cs[0] = q[iq]; cs[1] = T(0);
    }
}
template<class T, class Tp, int J>
JM_DEV void joint_calc(CPtr<T> P, const T * q, const T * v, SE3<T> & Mj, Sp<T> & vj)
{
    T cs[2];
    joint_calc<T, Tp, J>(P, q, v, Mj, vj, cs);
}
// liMi of a joint for the sweeps.  Trees of JM_LANE_REBUILD_MIN_JOINTS joints and more rebuild it from the cached joint
// coordinate and the constant placement (two scalars alive per joint instead of twelve); smaller trees keep it: with the
// sweeps' hand-over in LDS their registers suffice, and the rebuild costs constant loads and multiply-adds on a kernel that
// is bound by latency (measured at the end of round 5, 65 536 robots, RK4: the 7-joint arm 0.120 -> 0.099 ms per launch,
// 0.147 -> 0.117 with the full extra terms, `tree_arm` 0.083 -> 0.073).  A free-flyer's and a spherical joint's are kept.
#ifndef JM_LANE_REBUILD_MIN_JOINTS
#define JM_LANE_REBUILD_MIN_JOINTS 10
#endif
// Body of joint J (mass, centre of mass, inertia) and placement of joint J in its parent: the model's (parameter block, scalar
// loads) or, in the variation instantiations (W::APPLIED) with BatchArgs::model_lane bound, the lane's own -- one biased model
// per environment (Model::addBiasedToExtendedModel, model.cc:1166-1236); rows `[13 J + k][B]`: mass | com 3 | inertia xx xy xz
// yy yz zz | placement translation 3.
template<class T, class Tp, int J, class W> JM_DEV RBI<T> body_rbi(CPtr<T> P, const W & w)
{
    using L = Layout<Tp>;
    RBI<T> Y = ld_rbi<T>(P, L::JOINT + J * L::JSTRIDE + 12);
    if constexpr (W::APPLIED)
    {
        const BatchArgs<T> & A = *w.args;
        if (A.model_lane)
        {
            const long long st = A.lane_map ? A.B_full : A.B;
            const T * const r = A.model_lane + (long long)(13 * J) * st + (A.lane_map ? (long long)A.lane_map[w.lane] : w.lane);
            Y = {r[0], {r[st], r[2 * st], r[3 * st]}, S3<T>{r[4 * st], r[5 * st], r[6 * st], r[7 * st], r[8 * st], r[9 * st]}};
        }
    }
    return Y;
}
template<class T, class Tp, int J, class W> JM_DEV SE3<T> joint_placement(CPtr<T> P, const W & w)
{
    using L = Layout<Tp>;
    SE3<T> plc = ld_se3<T>(P, L::JOINT + J * L::JSTRIDE);
    if constexpr (W::APPLIED)
    {
        const BatchArgs<T> & A = *w.args;
        if (A.model_lane)
        {
            const long long st = A.lane_map ? A.B_full : A.B;
            const T * const r = A.model_lane + (long long)(13 * J + 10) * st + (A.lane_map ? (long long)A.lane_map[w.lane] : w.lane);
            plc.p = {r[0], r[st], r[2 * st]};
        }
    }
    return plc;
}
template<class Tp> constexpr bool lane_rebuild() { return Tp::NJ - 1 >= JM_LANE_REBUILD_MIN_JOINTS; }
template<class T, class Tp, int J, class W> JM_DEV SE3<T> limi_of(CPtr<T> P, const W & w)
{
    constexpr int t = Tp::jtype[J];
    if constexpr (t == JM_JT_FREEFLYER || jt_is_sph(t) || !lane_rebuild<Tp>()) return w.liMi[J];
    else
    {
        using L = Layout<Tp>;
        const SE3<T> plc = joint_placement<T, Tp, J>(P, w);
        SE3<T> Mj;
        // (opaque copies: common-subexpression elimination would otherwise merge the rebuilt placement with the
        // one the forward kinematics formed and keep all twelve scalars alive in between)
        T c = w.jcs[J][0], sn = w.jcs[J][1];
        JM_OPAQUE(c);
        if constexpr (jt_is_rev(t))
        {
            JM_OPAQUE(sn);
            constexpr int ax = jt_axis(t);
            if constexpr (ax >= 0) Mj.R = rot_axis<T>(ax, c, sn);
            else Mj.R = rot_rodrigues(joint_axis<T, Tp, J>(P), c, sn);
            Mj.p = zero3<T>();
        }
        else
        {
            Mj.R = ident3<T>();
            Mj.p = c * joint_axis<T, Tp, J>(P);
        }
        return plc * Mj;
    }
}
template<class T, class Tp, int J> JM_DEV Sp<T> joint_S_times(CPtr<T> P, const T * x)
{
    constexpr int t = Tp::jtype[J];
    constexpr int iv = Tp::idx_v[J];
    if constexpr (t == JM_JT_FREEFLYER)
        return {{x[iv], x[iv + 1], x[iv + 2]}, {x[iv + 3], x[iv + 4], x[iv + 5]}};
    else if constexpr (jt_is_sph(t))
        return {zero3<T>(), {x[iv], x[iv + 1], x[iv + 2]}};
    else if constexpr (jt_is_rev(t))
        return {zero3<T>(), x[iv] * joint_axis<T, Tp, J>(P)};
    else
        return {x[iv] * joint_axis<T, Tp, J>(P), zero3<T>()};
}
// S^T f for 1-dof joints
template<class T, class Tp, int J> JM_DEV T joint_St_dot(CPtr<T> P, Sp<T> f)
{
    constexpr int t = Tp::jtype[J];
    constexpr int ax = jt_axis(t);
    const V3<T> w = jt_is_rev(t) ? f.a : f.l;
    if constexpr (ax >= 0) return comp(w, ax);
    else return dot(joint_axis<T, Tp, J>(P), w);
}

// Engine::computeContactDynamics (engine.cc:3197-3238), flat ground n = z
// `mu_lane` >= 0: the lane's own friction coefficient (BatchArgs::friction) instead of the option
template<class T, class Tp> JM_DEV V3<T> contact_law(CPtr<T> P, T depth, V3<T> vW, T mu_lane = T(-1))
{
    using L = Layout<Tp>;
    const T k = P[L::OPT + 6], c = P[L::OPT + 7], eps = P[L::OPT + 9], vt = P[L::OPT + 10];
    const T mu = mu_lane >= T(0) ? mu_lane : P[L::OPT + 8];
    const T vDepth = vW.z;
    const T fN = -fmin_(k * depth + c * vDepth, T(0));
    const V3<T> vT = {vW.x, vW.y, vW.z - vDepth};
    const T ratio = fmin_(sqrt_(dot(vT, vT)) * rcp_(vt), T(1));
    const T fT = mu * ratio * fN;
    V3<T> f = {-fT * vT.x, -fT * vT.y, fN - fT * vT.z};
    if (eps > Eps<T>::eps)
    {
        const T blend = tanh_(T(2) * (-depth * rcp_(eps)));
        f = blend * f;
    }
    return f;
}

// the same law on a ground of unit normal n (world.groundProfile, engine.cc:3138-3142)
// `mu_lane` >= 0: the lane's own friction coefficient (BatchArgs::friction) instead of the option
template<class T, class Tp> JM_DEV V3<T> contact_law_n(CPtr<T> P, V3<T> n, T depth, V3<T> vW, T mu_lane = T(-1))
{
    using L = Layout<Tp>;
    const T k = P[L::OPT + 6], c = P[L::OPT + 7], eps = P[L::OPT + 9], vt = P[L::OPT + 10];
    const T mu = mu_lane >= T(0) ? mu_lane : P[L::OPT + 8];
    const T vDepth = dot(vW, n);
    const T fN = -fmin_(k * depth + c * vDepth, T(0));
    const V3<T> vT = vW - vDepth * n;
    const T ratio = fmin_(sqrt_(dot(vT, vT)) * rcp_(vt), T(1));
    const T fT = mu * ratio * fN;
    V3<T> f = fN * n - fT * vT;
    if (eps > Eps<T>::eps)
    {
        const T blend = tanh_(T(2) * (-depth * rcp_(eps)));
        f = blend * f;
    }
    return f;
}
// world.groundProfile(x, y) -> height and unit normal out of the height map of the batch arguments
// (`A`: anything with the height-map fields of BatchArgs -- the batch arguments or the constraint arguments)
template<class T, class G> JM_DEV void ground_profile(const G & A, T x, T y, T & h, V3<T> & n)
{
    const int nx = A.ground_nx, ny = A.ground_ny;
    T u = (x - A.ground_x0) / A.ground_dx, w = (y - A.ground_y0) / A.ground_dy;
    const bool in_x = u >= T(0) && u <= T(nx - 1), in_y = w >= T(0) && w <= T(ny - 1);
    u = fmin_(fmax_(u, T(0)), T(nx - 1));
    w = fmin_(fmax_(w, T(0)), T(ny - 1));
    int ix = (int)u, iy = (int)w;
    ix = ix > nx - 2 ? nx - 2 : ix; iy = iy > ny - 2 ? ny - 2 : iy;
    ix = ix < 0 ? 0 : ix; iy = iy < 0 ? 0 : iy;
    const T fx = u - T(ix), fy = w - T(iy);
    const T h00 = A.ground_h[iy * nx + ix], h10 = A.ground_h[iy * nx + ix + 1];
    const T h01 = A.ground_h[(iy + 1) * nx + ix], h11 = A.ground_h[(iy + 1) * nx + ix + 1];
    h = (T(1) - fy) * ((T(1) - fx) * h00 + fx * h10) + fy * ((T(1) - fx) * h01 + fx * h11);
    // outside the grid the ground continues flat (height of the nearest edge sample, no slope across the edge)
    const T dhdx = in_x ? ((T(1) - fy) * (h10 - h00) + fy * (h11 - h01)) / A.ground_dx : T(0);
    const T dhdy = in_y ? ((T(1) - fx) * (h01 - h00) + fx * (h11 - h10)) / A.ground_dy : T(0);
    const T inv = T(1) / sqrt_(dhdx * dhdx + dhdy * dhdy + T(1));
    n = {-dhdx * inv, -dhdy * inv, inv};
}
// Local frame of a contact constraint on the ground surface, in ROOT coordinates: the reference expresses the rows
// of a FrameConstraint (x, y, z, rotation about z) in `rotationLocal_` = [t0 t1 n] built from the ground normal n under
// the contact point (FrameConstraint::setNormal, frame_constraint.cc:62-68: t1 = normalize(n x e_x), t0 = t1 x n), and
// takes the penetration depth to first order, (z - height) n_z (engine.cc:3133-3141).  Returns M = rotationLocal^T R1:
// `M * x` are the local components of a root-coordinate vector, `M^T * lambda` the root coordinates of a local one;
// the third row of M is the normal.  Flat ground (no height map bound): M = R1, depth = world height.
// GND = false: kernels without the variation code, flat ground at compile time.
template<bool GND, class T, class G> JM_DEV M3<T> contact_frame(const G & g, const M3<T> & R1, V3<T> p1, V3<T> pc, T & depth)
{
    bool flat = true;
    if constexpr (GND) flat = !g.ground_h;
    if (flat)
    {
        depth = p1.z + dot(V3<T>{R1.m20, R1.m21, R1.m22}, pc);
        return R1;
    }
    const V3<T> pW = R1 * pc + p1;
    T hG;
    V3<T> n;
    ground_profile(g, pW.x, pW.y, hG, n);
    depth = (pW.z - hG) * n.z;
    // t1 = normalize(n x e_x) = (0, n.z, -n.y) / |.|, t0 = t1 x n
    const T inv = rsqrt_(n.z * n.z + n.y * n.y);
    const V3<T> t1 = {T(0), n.z * inv, -n.y * inv};
    const V3<T> t0 = cross(t1, n);
    return {t0.x * R1.m00 + t0.y * R1.m10 + t0.z * R1.m20, t0.x * R1.m01 + t0.y * R1.m11 + t0.z * R1.m21, t0.x * R1.m02 + t0.y * R1.m12 + t0.z * R1.m22,
            t1.x * R1.m00 + t1.y * R1.m10 + t1.z * R1.m20, t1.x * R1.m01 + t1.y * R1.m11 + t1.z * R1.m21, t1.x * R1.m02 + t1.y * R1.m12 + t1.z * R1.m22,
            n.x * R1.m00 + n.y * R1.m10 + n.z * R1.m20, n.x * R1.m01 + n.y * R1.m11 + n.z * R1.m21, n.x * R1.m02 + n.y * R1.m12 + n.z * R1.m22};
}

// symmetric positive definite 6x6 solve (Ia + diag(rot)) x = b (calc_aba free-flyer,
// pinocchio_overload_algorithms.h:357-378).  The reference forms the explicit inverse through an
// LL^T factorisation; here one square-root-free LDL^T factorisation + one solve:
// x = Dinv (u - Ia a_gf).  Only the lower triangle of A is read.
template<class T> JM_DEV void chol6_solve(T (&A)[6][6], T (&b)[6])
{
    T dinv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
    {
        T w[6];
        T s = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k)
        {
            w[k] = A[j][k] * A[k][k];  // L_jk * d_k  (diagonal holds d_k)
            s -= A[j][k] * w[k];
        }
        A[j][j] = s;
        dinv[j] = rcp_(s);
#pragma unroll
        for (int i = j + 1; i < 6; ++i)
        {
            T t = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= A[i][k] * w[k];
            A[i][j] = t * dinv[j];
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
    {
        T s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= A[i][k] * b[k];
        b[i] = s;
    }
#pragma unroll
    for (int i = 5; i >= 0; --i)
    {
        T s = b[i] * dinv[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= A[k][i] * b[k];
        b[i] = s;
    }
}

// second solve with the factor left in `A` by chol6_solve (`dinv[j]` = 1 / A[j][j])
template<class T> JM_DEV void chol6_resolve(const T (&A)[6][6], const T (&dinv)[6], T (&b)[6])
{
#pragma unroll
    for (int i = 0; i < 6; ++i)
    {
        T s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= A[i][k] * b[k];
        b[i] = s;
    }
#pragma unroll
    for (int i = 5; i >= 0; --i)
    {
        T s = b[i] * dinv[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= A[k][i] * b[k];
        b[i] = s;
    }
}

// ---------------------------------------------------------------- a = f(q, v) with held command
// W = Work<T, Tp>: spring-damper contact model.  W = WorkC<T, Tp>: the unconstrained part of the
// constraint contact model (no contact forces, no out-of-bounds flag, root factor kept).
// Kinematic half: placements, velocities, bias accelerations, contact forces, motor efforts.  It is all the
// output pass needs of an evaluation, so lane_run calls it on its own after the last evaluation of a launch
// instead of keeping oMi / vel / fext / cf alive across the ABA sweeps of every evaluation.
template<class T, class Tp, class W>
JM_DEV void eval_kinematics(CPtr<T> P, const T * q, const T * v, const T * cmd, W & w)
{
    using L = Layout<Tp>;
    constexpr int NJ = Tp::NJ;
    // universe: frames rigidly attached to the world refer to joint 0
    w.oMi[0] = {ident3<T>(), zero3<T>()};
    w.vel[0] = zero6<T>();
    w.fext[0] = zero6<T>();
    // ---- forward kinematics (+ ABA pass 1 velocity part)
    static_for<1, NJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int p = Tp::parent[j];
        SE3<T> Mj;
        Sp<T> vj;
        joint_calc<T, Tp, j>(P, q, v, Mj, vj, w.jcs[j]);
        const SE3<T> plc = joint_placement<T, Tp, j>(P, w);
        w.liMi[j] = plc * Mj;
        if constexpr (p > 0)
        {
            w.oMi[j] = w.oMi[p] * w.liMi[j];
            w.vel[j] = vj + actinv_motion(w.liMi[j], w.vel[p]);
        }
        else
        {
            w.oMi[j] = w.liMi[j];
            w.vel[j] = vj;
        }
        w.fext[j] = zero6<T>();
        if constexpr (jt_bounded(Tp::jtype[j]) && !W::CONSTRAINED)
        {
            constexpr int iq = Tp::idx_q[j];
            if (P[L::QHI + iq] < q[iq] || q[iq] < P[L::QLO + iq]) w.status |= JM_LANE_OUT_OF_BOUNDS;
        }
    });
    // ---- spring-damper contact forces (engine.cc:3394-3425)
    // (the optional per-lane inputs stay in batch order under the compact launches of the per-stage adaptive stepper)
    const BatchArgs<T> & A_ = *w.args;
    auto lane_g = [&]() -> long long { return A_.lane_map ? (long long)A_.lane_map[w.lane] : w.lane; };
    T mu_lane = T(-1);
    if constexpr (Tp::NC > 0 && !W::CONSTRAINED)
        if (A_.friction) mu_lane = A_.friction[lane_g()];
    static_for<0, Tp::NC>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        constexpr int j = Tp::contact_joint[c];
        const SE3<T> fr = ld_se3<T>(P, L::CONTACT + 12 * c);
        T depth = w.oMi[j].p.z + dot(V3<T>{w.oMi[j].R.m20, w.oMi[j].R.m21, w.oMi[j].R.m22}, fr.p);
        V3<T> nG = {T(0), T(0), T(1)};
        bool on_map = false;
        if constexpr (W::APPLIED && !W::CONSTRAINED)
            if (A_.ground_h)
            {
                // world.groundProfile at the contact point (height map of the variation instantiation, the lane's own patch of
                // it with BatchArgs::ground_off); first-order projection (engine.cc:3138-3145)
                const V3<T> pW = w.oMi[j].p + w.oMi[j].R * fr.p;
                T ox = T(0), oy = T(0);
                if (A_.ground_off)
                {
                    const long long st = A_.lane_map ? A_.B_full : A_.B;
                    ox = A_.ground_off[lane_g()]; oy = A_.ground_off[st + lane_g()];
                }
                T hG;
                ground_profile(A_, pW.x + ox, pW.y + oy, hG, nG);
                depth = (pW.z - hG) * nG.z;
                on_map = true;
            }
        Sp<T> fl = zero6<T>();
        if (!W::CONSTRAINED && depth < T(0))
        {
            // world velocity of the contact point: oMi.R (v_lin + w x p_frame)
            const V3<T> vj = w.vel[j].l + cross(w.vel[j].a, fr.p);
            const V3<T> vW = w.oMi[j].R * vj;
            const V3<T> fW = on_map ? contact_law_n<T, Tp>(P, nG, depth, vW, mu_lane) : contact_law<T, Tp>(P, depth, vW, mu_lane);
            fl.l = tmul(w.oMi[j].R, fW);
            fl.a = cross(fr.p, fl.l);
        }
        w.fext[j] = w.fext[j] + fl;
        w.cf[c] = actinv_force(fr, fl);
    });
    // ---- impulse / profile forces (Engine::computeExternalForces, engine.cc:3481-3560): the world-aligned wrench applied at
    // a frame goes to the frame's parent joint, in the joint frame (convertForceGlobalFrameToJoint, utilities/pinocchio.cc:794-809)
    if constexpr (W::APPLIED)
    if (A_.applied && A_.applied_k > 0)
    {
        const BatchArgs<T> & A = A_;
        const long long st = A.lane_map ? A.B_full : A.B;
        const T * const a0 = A.applied + lane_g();
        static_for<1, NJ>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            for (int i = 0; i < A.applied_k; ++i)
            {
                if (A.applied_joint[i] != j) continue;
                const T * a = a0 + (long long)(6 * i) * st;
                const V3<T> F = {a[0], a[st], a[2 * st]}, M = {a[3 * st], a[4 * st], a[5 * st]};
                const V3<T> p = {A.applied_p[3 * i], A.applied_p[3 * i + 1], A.applied_p[3 * i + 2]};
                Sp<T> f;
                f.l = tmul(w.oMi[j].R, F);
                f.a = tmul(w.oMi[j].R, M) + cross(p, f.l);
                w.fext[j] = w.fext[j] + f;
            }
        });
    }
    // ---- motors (basic_motors.cc:83-143) and total effort
    static_for<0, Tp::NV>([&](auto ic) { w.ueff[decltype(ic)::value] = T(0); });
    static_for<0, Tp::NM>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        constexpr int j = Tp::motor_joint[m];
        constexpr int iv = Tp::idx_v[j];
        constexpr int fl = Tp::motor_flags[m];
        constexpr int o = L::MOTOR + JM_MOTOR_NPARAMS * m;
        const T red = P[o], elim = P[o + 1], vlim = P[o + 2], islope = P[o + 3];
        const T vjnt = v[iv];
        const T vmot = red * vjnt;
        T um = cmd[m];
        if constexpr ((fl & JM_MOTOR_EFFORT_LIMIT) != 0)
        {
            T emin = -elim, emax = elim;
            if constexpr ((fl & JM_MOTOR_VELOCITY_LIMIT) != 0)
            {
                const T vdelta = elim * islope;
                if (vdelta > T(0))
                {
                    const T vthr = fmax_(vlim - vdelta, T(0));
                    const T inv = T(1) / (vlim - vthr);
                    emin *= clamp_((vlim + vmot) * inv, T(0), T(1));
                    emax *= clamp_((vlim - vmot) * inv, T(0), T(1));
                }
            }
            um = clamp_(um, emin, emax);
        }
        T ut = red * um;
        if constexpr ((fl & JM_MOTOR_FRICTION) != 0)
        {
            const T fds = P[o + 8];
            if (vjnt > T(0)) ut += P[o + 4] * vjnt + P[o + 6] * tanh_(fds * vjnt);
            else ut += P[o + 5] * vjnt + P[o + 7] * tanh_(fds * vjnt);
        }
        w.umotor[m] = um;
        w.ueff[iv] += ut;
    });
    // ---- flexibility of the spherical joints (Engine::computeInternalDynamics, engine.cc:3365-3391):
    // u_internal -= Jlog3(q) (stiffness * log3(q)) + damping * w
    static_for<1, NJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (jt_is_sph(Tp::jtype[j]))
        {
            constexpr int iq = Tp::idx_q[j], iv = Tp::idx_v[j];
            constexpr int o = L::FLEX + 6 * spherical_rank<Tp>(j);
            T angle;
            const V3<T> aa = quat_log3(q[iq], q[iq + 1], q[iq + 2], q[iq + 3], angle);
            // stiffness 3, damping 3: the lane's own (JM_F_FLEXIBILITY) or the model's
            constexpr int of = 6 * spherical_rank<Tp>(j);
            T kd[6];
            static_for<0, 6>([&](auto ic) { kd[decltype(ic)::value] = P[o + decltype(ic)::value]; });
            if (A_.flex_lane)
            {
                const long long st = A_.lane_map ? A_.B_full : A_.B;
                const T * const f0 = A_.flex_lane + lane_g();
                static_for<0, 6>([&](auto ic) { kd[decltype(ic)::value] = f0[(long long)(of + decltype(ic)::value) * st]; });
            }
            V3<T> t3 = jlog3_mul(angle, aa, V3<T>{kd[0] * aa.x, kd[1] * aa.y, kd[2] * aa.z});
            // "Flexible joint angle must be smaller than 0.95 * pi": the reference throws (engine.cc:3379-3383) -- a rejected
            // trial of the adaptive stepper, the end of a fixed-step simulation.  The efforts become NaN and so does the
            // acceleration, which takes those very paths (JM_LANE_NAN / a rejected attempt).
            if (angle > T(0.95 * 3.14159265358979323846)) t3.x = T(__builtin_nan(""));
            w.ueff[iv] -= t3.x + kd[3] * v[iv];
            w.ueff[iv + 1] -= t3.y + kd[4] * v[iv + 1];
            w.ueff[iv + 2] -= t3.z + kd[5] * v[iv + 2];
        }
    });
    static_for<0, Tp::NV>([&](auto ic) { w.u[decltype(ic)::value] = w.ueff[decltype(ic)::value]; });
}

// Articulated-body sweeps on what eval_kinematics left in `w`
// The sweeps keep ONE spatial vector per joint between the passes -- its velocity: the bias acceleration
// c_j = v_j x (S qd) and the bias force v x* (I v) - fext are formed where the backward / forward sweeps consume
// them (a dozen multiply-adds each) instead of sitting in registers from the forward kinematics on.
template<class T, class Tp, int J, class W> JM_DEV Sp<T> joint_bias_acc(CPtr<T> P, const T * v, const W & w)
{
    Sp<T> vj = w.vel[J];
    JM_OPAQUE(vj.l.x); JM_OPAQUE(vj.l.y); JM_OPAQUE(vj.l.z); JM_OPAQUE(vj.a.x); JM_OPAQUE(vj.a.y); JM_OPAQUE(vj.a.z);
    return cross_mm(vj, joint_S_times<T, Tp, J>(P, v));  // c_j = 0 for every supported joint
}
// What the backward sweep leaves for the forward sweep -- U_j (6), 1/D_j, the reduced effort u_j of every 1-dof
// joint -- is written once and read once, with both sweeps' whole working set in between (and, under the constraint
// model, every bias-free solve of the delassus columns reads U_j and 1/D_j again): the lane kernels park it in LDS
// (`Work::stash`) instead of leaving it to the register allocator's scratch spills.  Joints are served from the
// leaves (produced first, consumed last) while rows last.
template<class Tp> constexpr int stash_rows_wanted() { return 8 * (Tp::NJ - 1); }
template<class T, class Tp> constexpr int stash_rows()
{
    // one wave per SIMD (512 registers): 4 blocks of 64 lanes share the 160 kB of a CU
    constexpr int budget = (int)(160 * 1024 / 4 / (64 * sizeof(T))) - 3 * Tp::NV;
    constexpr int want = stash_rows_wanted<Tp>();
    if (Tp::NJ - 1 < 4) return 0;   // short chains fit their sweeps in registers
    return budget <= 0 ? 0 : (want < budget ? want : (budget / 8) * 8);
}
template<class Tp> constexpr int stage_rows() { return 3 * Tp::NV; }
// rows of the per-lane buffer of the lane kernels: the Runge-Kutta stage rows, then the sweeps' stash
template<class T, class Tp> constexpr int lane_rows() { return stage_rows<Tp>() + stash_rows<T, Tp>(); }
#ifndef JM_HOST_EMU
// the block's buffer (one object per topology and scalar type, named here so that every access is an LDS instruction
// with a constant base: a pointer kept in the working set decays to a generic one)
template<class T, class Tp> JM_DEV T * lane_lds()
{
    __shared__ T buf[lane_rows<T, Tp>() * 64];
    return buf;
}
#endif
template<class T, class Tp, class W> JM_DEV T * stash_of(const W & w)
{
#ifdef JM_HOST_EMU
    return w.stash;
#else
    (void)w;
    return lane_lds<T, Tp>() + stage_rows<Tp>() * 64 + threadIdx.x;
#endif
}
// joints NJ-1, NJ-2, ... are parked while rows last
template<class T, class Tp, int J> constexpr bool stashed() { return Tp::NJ - J <= stash_rows<T, Tp>() / 8; }
// what the sweeps left for joint J: U_J, 1/D_J (read by the forward sweep and by the bias-free solves)
template<class T, class Tp, int J, class W> JM_DEV Sp<T> sweep_U(const W & w)
{
    if constexpr (stashed<T, Tp, J>())
    {
        constexpr int SS = LANE_STRIDE;
        const T * o = stash_of<T, Tp>(w) + (8 * (Tp::NJ - 1 - J)) * SS;
        return {{o[0], o[SS], o[2 * SS]}, {o[3 * SS], o[4 * SS], o[5 * SS]}};
    }
    else return w.U[J];
}
template<class T, class Tp, int J, class W> JM_DEV T sweep_dinv(const W & w)
{
    if constexpr (stashed<T, Tp, J>()) return stash_of<T, Tp>(w)[(8 * (Tp::NJ - 1 - J) + 6) * LANE_STRIDE];
    else return w.dinv[J];
}
template<class T, class Tp, class W>
JM_DEV void eval_aba(CPtr<T> P, const T * v, W & w)
{
    using L = Layout<Tp>;
    constexpr int NJ = Tp::NJ;
    constexpr int SS = LANE_STRIDE;
    T * const stash = stash_of<T, Tp>(w);
    // ---- ABA pass 2 (AbaBackwardStep), leaves -> root; pass 1's force part f = v x* (I v) - fext at the visit
    AI<T> Yacc[NJ];
    Sp<T> facc[NJ];   // bias forces handed up by the children
#ifdef JM_HOST_EMU
    std::memset(Yacc, 0xFF, sizeof(Yacc));
    std::memset(facc, 0xFF, sizeof(facc));
#endif
    static_rfor<1, NJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int p = Tp::parent[j];
        constexpr int t = Tp::jtype[j];
        constexpr int iv = Tp::idx_v[j];
        const RBI<T> Yj = body_rbi<T, Tp, j>(P, w);
        AI<T> Ia;
        if constexpr (Tp::nchildren[j] > 0) Ia = Yacc[j];
        else Ia = ai_from_rbi(Yj);
        Sp<T> fj = cross_mf(w.vel[j], rbi_mul(Yj, w.vel[j])) - w.fext[j];
        if constexpr (Tp::nchildren[j] > 0) fj = fj + facc[j];
        w.f[j] = fj;
        w.agf[j] = joint_bias_acc<T, Tp, j>(P, v, w);
        if constexpr (t == JM_JT_FREEFLYER)
        {
            static_assert(t != JM_JT_FREEFLYER || p == 0, "free-flyer joints are only supported at the root");
            // u -= S^T f ; (Ia + Im) ddq = u - Ia a_gf   (solved in pass 3 order right here)
            const SE3<T> & M = w.liMi[j];
            const V3<T> g = ld_v3<T>(P, L::OPT), gw = ld_v3<T>(P, L::OPT + 3);
            const Sp<T> a0 = {-g, -gw};
            w.agf[j] = w.agf[j] + actinv_motion(M, a0);
            const Sp<T> Ya = ai_mul(Ia, w.agf[j]);
            T b[6] = {w.u[iv] - w.f[j].l.x - Ya.l.x, w.u[iv + 1] - w.f[j].l.y - Ya.l.y, w.u[iv + 2] - w.f[j].l.z - Ya.l.z,
                      w.u[iv + 3] - w.f[j].a.x - Ya.a.x, w.u[iv + 4] - w.f[j].a.y - Ya.a.y, w.u[iv + 5] - w.f[j].a.z - Ya.a.z};
            T A[6][6];
            A[0][0] = Ia.A.xx; A[1][0] = Ia.A.xy; A[2][0] = Ia.A.xz; A[1][1] = Ia.A.yy; A[2][1] = Ia.A.yz; A[2][2] = Ia.A.zz;
            // lower-left block = B^T : A[3+i][k] = B[k][i]
            A[3][0] = Ia.B.m00; A[3][1] = Ia.B.m10; A[3][2] = Ia.B.m20;
            A[4][0] = Ia.B.m01; A[4][1] = Ia.B.m11; A[4][2] = Ia.B.m21;
            A[5][0] = Ia.B.m02; A[5][1] = Ia.B.m12; A[5][2] = Ia.B.m22;
            A[3][3] = Ia.D.xx; A[4][3] = Ia.D.xy; A[5][3] = Ia.D.xz; A[4][4] = Ia.D.yy; A[5][4] = Ia.D.yz; A[5][5] = Ia.D.zz;
#pragma unroll
            for (int k = 0; k < 6; ++k) A[k][k] += P[L::ROTOR + iv + k];
            chol6_solve(A, b);
            if constexpr (W::CONSTRAINED)
            {
#pragma unroll
                for (int r = 0; r < 6; ++r)
                {
#pragma unroll
                    for (int c = 0; c <= r; ++c) w.rootA[r][c] = A[r][c];
                    w.rootdinv[r] = rcp_(A[r][r]);
                }
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) w.ddq[iv + k] = b[k];
            w.agf[j] = w.agf[j] + Sp<T>{{b[0], b[1], b[2]}, {b[3], b[4], b[5]}};
        }
        else if constexpr (jt_is_sph(t))
        {
            // JointModelSpherical::calc_aba: S = [0; 1]: U = [B; D], Dinv = (D + rotor)^-1, Ia -= U Dinv U^T
            constexpr int k = spherical_rank<Tp>(j);
            const V3<T> u3 = V3<T>{w.u[iv], w.u[iv + 1], w.u[iv + 2]} - w.f[j].a;
            S3<T> Dm = Ia.D;
            Dm.xx += P[L::ROTOR + iv]; Dm.yy += P[L::ROTOR + iv + 1]; Dm.zz += P[L::ROTOR + iv + 2];
            const S3<T> Di = sym_inverse(Dm);
            w.sphB[k] = Ia.B; w.sphD[k] = Ia.D; w.sphDinv[k] = Di; w.sphu[k] = u3;
            if constexpr (p > 0)
            {
                const M3<T> Dif = full(Di), Df = full(Ia.D);
                const M3<T> BDi = Ia.B * Dif, DDi = Df * Dif;
                const M3<T> BDB = mul_bt(BDi, Ia.B), DDD = DDi * Df;
                AI<T> Y;
                Y.A = {Ia.A.xx - BDB.m00, Ia.A.xy - BDB.m01, Ia.A.xz - BDB.m02, Ia.A.yy - BDB.m11, Ia.A.yz - BDB.m12, Ia.A.zz - BDB.m22};
                Y.B = Ia.B - BDi * Df;
                Y.D = {Ia.D.xx - DDD.m00, Ia.D.xy - DDD.m01, Ia.D.xz - DDD.m02, Ia.D.yy - DDD.m11, Ia.D.yz - DDD.m12, Ia.D.zz - DDD.m22};
                const Sp<T> Ya = ai_mul(Y, w.agf[j]);
                const V3<T> t3 = Di * u3;
                const Sp<T> pa = {w.f[j].l + Ya.l + Ia.B * t3, w.f[j].a + Ya.a + Ia.D * t3};
                const SE3<T> M = limi_of<T, Tp, j>(P, w);
                const AI<T> Tr = ai_transform(M, Y);
                if constexpr (Tp::first_child[p] == j)
                {
                    Yacc[p] = ai_from_rbi(body_rbi<T, Tp, p>(P, w)) + Tr;
                    facc[p] = act_force(M, pa);
                }
                else
                {
                    Yacc[p] = Yacc[p] + Tr;
                    facc[p] = facc[p] + act_force(M, pa);
                }
            }
        }
        else
        {
            const V3<T> n = joint_axis<T, Tp, j>(P);
            const T uj = w.u[iv] - joint_St_dot<T, Tp, j>(P, w.f[j]);
            w.u[iv] = uj;
            Sp<T> U;
            if constexpr (jt_is_rev(t)) U = {Ia.B * n, Ia.D * n};
            else U = {Ia.A * n, tmul(Ia.B, n)};
            const T D = joint_St_dot<T, Tp, j>(P, U) + P[L::ROTOR + iv];
            const T dinv = T(1) / D;
            if constexpr (stashed<T, Tp, j>())
            {
                T * o = stash + (8 * (NJ - 1 - j)) * SS;
                o[0] = U.l.x; o[SS] = U.l.y; o[2 * SS] = U.l.z; o[3 * SS] = U.a.x; o[4 * SS] = U.a.y; o[5 * SS] = U.a.z;
                o[6 * SS] = dinv; o[7 * SS] = uj;
            }
            else
            {
                w.U[j] = U;
                w.dinv[j] = dinv;
            }
            if constexpr (p > 0)
            {
                ai_rank1_sub(Ia, U, dinv);
                const Sp<T> Ya = ai_mul(Ia, w.agf[j]);
                const T ud = uj * dinv;
                const Sp<T> pa = {w.f[j].l + Ya.l + ud * U.l, w.f[j].a + Ya.a + ud * U.a};
                const SE3<T> M = limi_of<T, Tp, j>(P, w);
                const AI<T> Tr = ai_transform(M, Ia);
                if constexpr (Tp::first_child[p] == j)
                {
                    Yacc[p] = ai_from_rbi(body_rbi<T, Tp, p>(P, w)) + Tr;
                    facc[p] = act_force(M, pa);
                }
                else
                {
                    Yacc[p] = Yacc[p] + Tr;
                    facc[p] = facc[p] + act_force(M, pa);
                }
            }
        }
    });
    // ---- ABA pass 3 (AbaForwardStep2), root -> leaves
    static_for<1, NJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int p = Tp::parent[j];
        constexpr int t = Tp::jtype[j];
        constexpr int iv = Tp::idx_v[j];
        if constexpr (jt_is_sph(t))
        {
            constexpr int k = spherical_rank<Tp>(j);
            Sp<T> ap;
            if constexpr (p > 0) ap = w.agf[p];
            else
            {
                const V3<T> g = ld_v3<T>(P, L::OPT), gw = ld_v3<T>(P, L::OPT + 3);
                ap = {-g, -gw};
            }
            const Sp<T> ag = joint_bias_acc<T, Tp, j>(P, v, w) + actinv_motion(limi_of<T, Tp, j>(P, w), ap);
            const V3<T> Ua = tmul(w.sphB[k], ag.l) + w.sphD[k] * ag.a;
            const V3<T> dd = w.sphDinv[k] * (w.sphu[k] - Ua);
            w.ddq[iv] = dd.x; w.ddq[iv + 1] = dd.y; w.ddq[iv + 2] = dd.z;
            w.agf[j] = {ag.l, ag.a + dd};
        }
        else if constexpr (t != JM_JT_FREEFLYER)
        {
            Sp<T> ap;
            if constexpr (p > 0) ap = w.agf[p];
            else
            {
                const V3<T> g = ld_v3<T>(P, L::OPT), gw = ld_v3<T>(P, L::OPT + 3);
                ap = {-g, -gw};
            }
            const Sp<T> ag = joint_bias_acc<T, Tp, j>(P, v, w) + actinv_motion(limi_of<T, Tp, j>(P, w), ap);
            Sp<T> U;
            T dinv, uj;
            if constexpr (stashed<T, Tp, j>())
            {
                const T * o = stash + (8 * (NJ - 1 - j)) * SS;
                U = {{o[0], o[SS], o[2 * SS]}, {o[3 * SS], o[4 * SS], o[5 * SS]}};
                dinv = o[6 * SS]; uj = o[7 * SS];
            }
            else { U = w.U[j]; dinv = w.dinv[j]; uj = w.u[iv]; }
            const T Ua = dot(U.l, ag.l) + dot(U.a, ag.a);
            const T dd = dinv * (uj - Ua);
            w.ddq[iv] = dd;
            const V3<T> n = joint_axis<T, Tp, j>(P);
            if constexpr (jt_is_rev(t)) w.agf[j] = {ag.l, ag.a + dd * n};
            else w.agf[j] = {ag.l + dd * n, ag.a};
        }
    });
    static_for<0, Tp::NV>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (w.ddq[i] != w.ddq[i]) w.status |= JM_LANE_NAN;
    });
}

template<class T, class Tp, class W>
JM_DEV void eval_dynamics(CPtr<T> P, const T * q, const T * v, const T * cmd, W & w)
{
    eval_kinematics<T, Tp>(P, q, v, cmd, w);
    eval_aba<T, Tp>(P, v, w);
}

// ---------------------------------------------------------------- q (+) dv on the manifold
template<class T, class Tp> JM_DEV void integrate_q(CPtr<T> P, const T * q, const T * d, T * qo)
{
    static_for<1, Tp::NJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int t = Tp::jtype[j];
        constexpr int iq = Tp::idx_q[j], iv = Tp::idx_v[j];
        if constexpr (t == JM_JT_FREEFLYER)
        {
            SE3<T> M0;
            M0.R = quat_to_matrix(q[iq + 3], q[iq + 4], q[iq + 5], q[iq + 6]);
            M0.p = {q[iq], q[iq + 1], q[iq + 2]};
            const Sp<T> nu = {{d[iv], d[iv + 1], d[iv + 2]}, {d[iv + 3], d[iv + 4], d[iv + 5]}};
            const SE3<T> M1 = M0 * exp6(nu);
            T x, y, z, ww;
            matrix_to_quat(M1.R, x, y, z, ww);
            const T dp = x * q[iq + 3] + y * q[iq + 4] + z * q[iq + 5] + ww * q[iq + 6];
            const T sg = dp < T(0) ? T(-1) : T(1);
            const T n2 = x * x + y * y + z * z + ww * ww;
            const T al = sg * (T(3) - n2) * T(0.5);
            qo[iq] = M1.p.x; qo[iq + 1] = M1.p.y; qo[iq + 2] = M1.p.z;
            qo[iq + 3] = x * al; qo[iq + 4] = y * al; qo[iq + 5] = z * al; qo[iq + 6] = ww * al;
        }
        else if constexpr (jt_is_sph(t))
        {
            // SpecialOrthogonalOperationTpl<3>::integrate_impl: quat * exp3(omega), firstOrderNormalize
            T e4[4], r[4];
            quat_exp3(V3<T>{d[iv], d[iv + 1], d[iv + 2]}, e4);
            quat_mul(q + iq, e4, r);
            const T n2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
            const T al = (T(3) - n2) * T(0.5);
            qo[iq] = r[0] * al; qo[iq + 1] = r[1] * al; qo[iq + 2] = r[2] * al; qo[iq + 3] = r[3] * al;
        }
        else if constexpr (jt_is_unb(t))
        {
            T sw, cw;
            sincos_(d[iv], &sw, &cw);
            const T c = cw * q[iq] - sw * q[iq + 1], s = sw * q[iq] + cw * q[iq + 1];
            const T k = (T(3) - (c * c + s * s)) * T(0.5);
            qo[iq] = c * k; qo[iq + 1] = s * k;
        }
        else
            qo[iq] = q[iq] + d[iv];
    });
    (void)P;
}

// ---------------------------------------------------------------- extra terms + sensors
// Uses the kinematic data (liMi, oMi, vel, fext, cf, umotor) left in `w` by the last dynamics
// evaluation, which is at the new state (engine.cc:2143-2151).
template<class T, class Tp, class W>
JM_DEV void extra_terms_and_outputs(CPtr<T> P, const BatchArgs<T> & A, long long lane, const T * q, const T * v,
                                    const T * acc, W & w, bool sensors)
{
    using L = Layout<Tp>;
    constexpr int NJ = Tp::NJ;
    const long long B = A.B;
    const V3<T> g = ld_v3<T>(P, L::OPT), gw = ld_v3<T>(P, L::OPT + 3);
    if (A.energy)
    {
        T kin = T(0), pot = T(0), rot = T(0);
        static_for<1, NJ>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const RBI<T> Y = body_rbi<T, Tp, j>(P, w);
            kin += rbi_vtiv(Y, w.vel[j]);
            const V3<T> cg = w.oMi[j].p + w.oMi[j].R * Y.c;
            pot -= Y.m * dot(cg, g);
        });
        static_for<0, Tp::NV>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            rot += P[L::ROTOR + i] * v[i] * v[i];
        });
        A.energy[lane] = T(0.5) * kin + T(0.5) * rot;
        A.energy[B + lane] = pot;
    }
    // true spatial accelerations (ForwardKinematicsAccelerationStep, engine.cc:858-868)
    Sp<T> da[NJ], dagf[NJ];
    static_for<1, NJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int p = Tp::parent[j];
        const Sp<T> vj = joint_S_times<T, Tp, j>(P, v);
        const Sp<T> aj = cross_mm(w.vel[j], vj) + joint_S_times<T, Tp, j>(P, acc);
        const SE3<T> M = limi_of<T, Tp, j>(P, w);
        if constexpr (p > 0)
        {
            da[j] = aj + actinv_motion(M, da[p]);
            dagf[j] = aj + actinv_motion(M, dagf[p]);
        }
        else
        {
            da[j] = aj;  // data.a[0] = 0
            dagf[j] = aj + actinv_motion(M, Sp<T>{-g, -gw});
        }
    });
    if (A.joint_forces || A.centroidal)
    {
        // RNEA-like sweeps (engine.cc:870-887)
        // (momenta and wrenches are formed at the visit of the backward sweep and handed up: the frontier of the sweep
        // is all that is alive, not three spatial vectors per joint)
        Sp<T> hup[NJ], fBup[NJ], fjup[NJ];
        Sp<T> h0 = zero6<T>(), fB0 = zero6<T>();
        if (A.joint_forces) static_for<0, 6>([&](auto kc) { A.joint_forces[decltype(kc)::value * B + lane] = T(0); });
        static_rfor<1, NJ>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int p = Tp::parent[j];
            const RBI<T> Y = body_rbi<T, Tp, j>(P, w);
            Sp<T> hj = rbi_mul(Y, w.vel[j]);
            const Sp<T> vxh = cross_mf(w.vel[j], hj);
            Sp<T> fBj = rbi_mul(Y, da[j]) + vxh;
            Sp<T> fjj = vxh + rbi_mul(Y, dagf[j]) - w.fext[j];
            if constexpr (Tp::nchildren[j] > 0)
            {
                hj = hj + hup[j];
                fBj = fBj + fBup[j];
                fjj = fjj + fjup[j];
            }
            if (A.joint_forces)
            {
                T * o = A.joint_forces + (long long)(6 * j) * B + lane;
                o[0] = fjj.l.x; o[B] = fjj.l.y; o[2 * B] = fjj.l.z;
                o[3 * B] = fjj.a.x; o[4 * B] = fjj.a.y; o[5 * B] = fjj.a.z;
            }
            const SE3<T> M = limi_of<T, Tp, j>(P, w);
            if constexpr (p > 0)
            {
                if constexpr (Tp::first_child[p] == j)
                {
                    fBup[p] = act_force(M, fBj);
                    hup[p] = act_force(M, hj);
                    fjup[p] = act_force(M, fjj);
                }
                else
                {
                    fBup[p] = fBup[p] + act_force(M, fBj);
                    hup[p] = hup[p] + act_force(M, hj);
                    fjup[p] = fjup[p] + act_force(M, fjj);
                }
            }
            else
            {
                fB0 = fB0 + act_force(M, fBj);
                h0 = h0 + act_force(M, hj);
            }
        });
        if (A.centroidal)
        {
            // subtree inertia of joint 1 (engine.cc:817-832) -> com (engine.cc:889-904)
            T ms[NJ];
            V3<T> mc[NJ];  // mass * com of each subtree, in the joint frame
            static_for<1, NJ>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const RBI<T> Y = body_rbi<T, Tp, j>(P, w);
                ms[j] = Y.m;
                mc[j] = Y.m * Y.c;
            });
            static_rfor<2, NJ>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int p = Tp::parent[j];
                if constexpr (p > 0)
                {
                    const SE3<T> M = limi_of<T, Tp, j>(P, w);
                    mc[p] = mc[p] + M.R * mc[j] + ms[j] * M.p;
                    ms[p] = ms[p] + ms[j];
                }
            });
            const V3<T> c1 = (T(1) / ms[1]) * mc[1];
            const SE3<T> M1 = limi_of<T, Tp, 1>(P, w);
            const V3<T> com0 = M1.R * c1 + M1.p;
            Sp<T> hg = h0, dhg = fB0;
            hg.a = hg.a + cross(hg.l, com0);
            dhg.a = dhg.a + cross(dhg.l, com0);
            T * o = A.centroidal + lane;
            o[0] = com0.x; o[B] = com0.y; o[2 * B] = com0.z;
            o[3 * B] = hg.l.x; o[4 * B] = hg.l.y; o[5 * B] = hg.l.z; o[6 * B] = hg.a.x; o[7 * B] = hg.a.y; o[8 * B] = hg.a.z;
            o[9 * B] = dhg.l.x; o[10 * B] = dhg.l.y; o[11 * B] = dhg.l.z; o[12 * B] = dhg.a.x; o[13 * B] = dhg.a.y; o[14 * B] = dhg.a.z;
        }
    }
    if (A.f_external)
    {
        static_for<0, 6>([&](auto kc) { A.f_external[decltype(kc)::value * B + lane] = T(0); });
        static_for<1, NJ>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            T * o = A.f_external + (long long)(6 * j) * B + lane;
            o[0] = w.fext[j].l.x; o[B] = w.fext[j].l.y; o[2 * B] = w.fext[j].l.z;
            o[3 * B] = w.fext[j].a.x; o[4 * B] = w.fext[j].a.y; o[5 * B] = w.fext[j].a.z;
        });
    }
    if (A.contact_forces)
        static_for<0, Tp::NC>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            T * o = A.contact_forces + (long long)(6 * c) * B + lane;
            o[0] = w.cf[c].l.x; o[B] = w.cf[c].l.y; o[2 * B] = w.cf[c].l.z;
            o[3 * B] = w.cf[c].a.x; o[4 * B] = w.cf[c].a.y; o[5 * B] = w.cf[c].a.z;
        });
    if (A.u_motor)
        static_for<0, Tp::NM>([&](auto mc) { A.u_motor[decltype(mc)::value * B + lane] = w.umotor[decltype(mc)::value]; });
    if (A.u)
        static_for<0, Tp::NV>([&](auto ic) { A.u[decltype(ic)::value * B + lane] = w.ueff[decltype(ic)::value]; });
    if (!sensors) return;
    // ---- sensors (basic_sensors.cc)
    if (A.imu)
        static_for<0, Tp::NIMU>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int j = Tp::imu_joint[s];
            const SE3<T> fr = ld_se3<T>(P, L::IMU + 12 * s);
            const Sp<T> vf = actinv_motion(fr, w.vel[j]);
            Sp<T> af = actinv_motion(fr, da[j]);
            af.l = af.l + cross(vf.a, vf.l);
            const V3<T> gl = tmul(fr.R, tmul(w.oMi[j].R, g));  // (oMi.R fr.R)^T g
            const V3<T> acc3 = af.l - gl;
            T * o = A.imu + (long long)(6 * s) * B + lane;
            o[0] = vf.a.x; o[B] = vf.a.y; o[2 * B] = vf.a.z; o[3 * B] = acc3.x; o[4 * B] = acc3.y; o[5 * B] = acc3.z;
        });
    if (A.contact)
        static_for<0, Tp::NCS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int c = Tp::cs_contact[s];
            T * o = A.contact + (long long)(3 * s) * B + lane;
            o[0] = w.cf[c].l.x; o[B] = w.cf[c].l.y; o[2 * B] = w.cf[c].l.z;
        });
    if (A.force)
        static_for<0, Tp::NFORCE>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            Sp<T> sum = zero6<T>();
            static_for<0, Tp::NC>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                if constexpr (Tp::contact_joint[c] == Tp::force_joint[s])
                {
                    const SE3<T> rel = ld_se3<T>(P, L::FREL + 12 * (s * Tp::NC + c));
                    sum = sum + act_force(rel, w.cf[c]);
                }
            });
            T * o = A.force + (long long)(6 * s) * B + lane;
            o[0] = sum.l.x; o[B] = sum.l.y; o[2 * B] = sum.l.z; o[3 * B] = sum.a.x; o[4 * B] = sum.a.y; o[5 * B] = sum.a.z;
        });
    if (A.encoder)
        static_for<0, Tp::NENC>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int j = Tp::enc_joint[s];
            constexpr int iq = Tp::idx_q[j], iv = Tp::idx_v[j];
            T pos;
            if constexpr (jt_is_unb(Tp::jtype[j])) pos = atan2_(q[iq + 1], q[iq]);
            else pos = q[iq];
            T vel = v[iv];
            if constexpr (Tp::enc_side[s] == 0)
            {
                const T red = P[L::ENC + s];
                pos *= red;
                vel *= red;
            }
            A.encoder[(long long)(2 * s) * B + lane] = pos;
            A.encoder[(long long)(2 * s + 1) * B + lane] = vel;
        });
    if (A.effort)
        static_for<0, Tp::NEFF>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            A.effort[(long long)s * B + lane] = w.umotor[Tp::eff_motor[s]];
        });
}

// ---------------------------------------------------------------- one lane, all modes
// `sb` is the per-lane stage buffer (LDS on the GPU): element r at sb[r * SBS]; SBS = 0: the stride is
// the run-time `sb_stride` (stage rows kept in an HBM workspace, constraint-model kernel).
// Rows: [0,NV) accumulated velocity increment, [NV,2NV) accumulated acceleration increment,
//       [2NV,3NV) velocity of the previous stage.

// constraint contact model (jm_constraint.h): the free evaluation above + constraint switching +
// the boxed forward dynamics; `start_passes` > 0 runs the Engine::start sequence, < 0 only re-applies the
// stored multipliers (MODE_REFRESH)
template<class T, class Tp, class CA, class WC>
JM_DEV void eval_constrained(CPtr<T> P, const T * q, const T * v, const T * cmd, WC & w, const CA & C,
                             long long lane, long long B, int start_passes);

// efforts and wrenches of the stored multipliers on top of eval_kinematics' (the output pass of the constraint model)
template<class T, class Tp, class CA, class WC>
JM_DEV void constraint_forces_from_multipliers(CPtr<T> P, WC & w, const CA & C, long long lane, long long B);

template<class T, class Tp, class CON, class W>
JM_DEV void eval_any(CPtr<T> P, const T * q, const T * v, const T * cmd, W & w,
                     const typename CON::template ArgsT<T> & C, long long lane, long long B, int start_passes)
{
    if constexpr (CON::ON) eval_constrained<T, Tp>(P, q, v, cmd, w, C, lane, B, start_passes);
    else
    {
        (void)C; (void)lane; (void)B; (void)start_passes;
        eval_dynamics<T, Tp>(P, q, v, cmd, w);
    }
}

template<class T, class Tp, int SBS, class CON = NoCon>
JM_DEV void lane_run(const BatchArgs<T> & A, long long lane, T * sb,
                     const typename CON::template ArgsT<T> & C = typename CON::template ArgsT<T>{},
                     long long sb_stride = 0)
{
    const long long SBSr = SBS > 0 ? (long long)SBS : sb_stride;
    using L = Layout<Tp>;
    constexpr int NQ = Tp::NQ, NV = Tp::NV, NM = Tp::NM;
    const long long B = A.B;
    CPtr<T> P = (CPtr<T>)A.P;
    typename CON::template WorkT<T, Tp> w;
    // the sweeps' stash follows the stage rows
    static_assert(SBS == LANE_STRIDE, "the lane buffer interleaves the lanes of a block");
    T qs[NQ], vs[NV], as[NV], cmd[c_max(NM, 1)];
#ifdef JM_HOST_EMU
    // poison everything a GPU lane would find uninitialised: a read-before-write shows up as NaN
    std::memset(&w, 0xFF, sizeof(w));
    std::memset(qs, 0xFF, sizeof(qs)); std::memset(vs, 0xFF, sizeof(vs)); std::memset(as, 0xFF, sizeof(as));
#endif
    w.status = 0;
    w.stash = sb + (long long)stage_rows<Tp>() * SBS;
    w.args = &A;
    w.lane = lane;
    static_for<0, NM>([&](auto mc) { cmd[decltype(mc)::value] = A.command[decltype(mc)::value * B + lane]; });

    if (A.mode == MODE_RESET)
    {
        if (!A.mask[lane]) return;
        static_for<0, NQ>([&](auto ic) { A.q[decltype(ic)::value * B + lane] = A.q_init[decltype(ic)::value * B + lane]; });
        static_for<0, NV>([&](auto ic) { A.v[decltype(ic)::value * B + lane] = A.v_init[decltype(ic)::value * B + lane]; });
    }
    // The plain kernel runs the other modes through the loop below as ONE evaluation of the "refresh a(t+)" kind
    // (k = -1): a single copy of the evaluation and of the output pass in the kernel instead of two.
    if (CON::ON && A.mode != MODE_STEP)
    {
        // one evaluation at the given state
        const T * qsrc = (A.mode == MODE_DYNAMICS) ? A.q_in : A.q;
        const T * vsrc = (A.mode == MODE_DYNAMICS) ? A.v_in : A.v;
        static_for<0, NQ>([&](auto ic) { qs[decltype(ic)::value] = qsrc[decltype(ic)::value * B + lane]; });
        static_for<0, NV>([&](auto ic) { vs[decltype(ic)::value] = vsrc[decltype(ic)::value * B + lane]; });
        eval_any<T, Tp, CON>(P, qs, vs, cmd, w, C, lane, B,
                             (A.mode == MODE_START || A.mode == MODE_RESET) ? 4 : (A.mode == MODE_REFRESH ? -1 : 0));
        if (A.mode == MODE_DYNAMICS)
        {
            static_for<0, NV>([&](auto ic) { A.a_out[decltype(ic)::value * B + lane] = w.ddq[decltype(ic)::value]; });
            return;
        }
        // Engine::start: refuse huge initial contact forces (engine.cc:1310-1346)
        if (A.mode != MODE_REFRESH && !CON::ON)
        {
            T fmax2 = T(0);
            static_for<0, Tp::NC>([&](auto cc) {
                const Sp<T> & f = w.cf[decltype(cc)::value];
                fmax2 = fmax_(fmax2, dot(f.l, f.l));
            });
            if (fmax2 > T(1e10)) w.status |= JM_LANE_FORCE_OVERFLOW;
        }
        static_for<0, NV>([&](auto ic) { A.a[decltype(ic)::value * B + lane] = w.ddq[decltype(ic)::value]; });
        extra_terms_and_outputs<T, Tp>(P, A, lane, qs, vs, w.ddq, w, A.mode != MODE_REFRESH || A.update_sensors != 0);
        if (A.status) A.status[lane] = (A.mode == MODE_REFRESH) ? (A.status[lane] | w.status) : w.status;
        return;
    }

    // ---- MODE_STEP: n_sub fixed steps of dt with the command held
    const T dt = A.dt;
    const bool stepping = A.mode == MODE_STEP;
    const bool rk4 = A.solver == JM_SOLVER_RUNGE_KUTTA_4;
    const int evals_per_step = rk4 ? 4 : 1;
    const int pre = stepping ? (A.command_changed ? 1 : 0) : 1;
    const int n_evals = stepping ? pre + A.n_sub * evals_per_step : 1;
    const T * const qsrc = (A.mode == MODE_DYNAMICS) ? A.q_in : A.q;
    const T * const vsrc = (A.mode == MODE_DYNAMICS) ? A.v_in : A.v;
    T * const adst = (A.mode == MODE_DYNAMICS) ? A.a_out : A.a;
    // NaN guard on the incoming state (engine.cc:1737-1747)
    if (stepping)
    {
        bool bad = false;
        static_for<0, NQ>([&](auto ic) { const T x = A.q[decltype(ic)::value * B + lane]; bad |= (x != x); });
        static_for<0, NV>([&](auto ic) { const T x = A.v[decltype(ic)::value * B + lane]; bad |= (x != x); });
        static_for<0, NV>([&](auto ic) { const T x = A.a[decltype(ic)::value * B + lane]; bad |= (x != x); });
        if (bad) w.status |= JM_LANE_NAN;
    }
#pragma nounroll
    for (int e = 0; e < n_evals; ++e)
    {
        // stage index within the sub-step: -1 = a(t+) refresh, 0..2 = RK stages 1..3, 3 = final
        const int k = (e < pre) ? -1 : (rk4 ? ((e - pre) & 3) : 3);
        if (k == -1)
        {
            static_for<0, NQ>([&](auto ic) { qs[decltype(ic)::value] = qsrc[decltype(ic)::value * B + lane]; });
            static_for<0, NV>([&](auto ic) { vs[decltype(ic)::value] = vsrc[decltype(ic)::value * B + lane]; });
        }
        else
        {
            // previous stage derivative (kv, ka): for the first stage it is the state (v, a)
            const bool first = rk4 ? (k == 0) : true;
            T q0[NQ], incv[NV];
            static_for<0, NQ>([&](auto ic) { q0[decltype(ic)::value] = A.q[decltype(ic)::value * B + lane]; });
            // weights: b of the previous stage for the accumulators, A(i, i-1) for the stage state
            T bw, aw;
            if (rk4)
            {
                // accumulate derivative k_k with its own weight b_k (b = 1/6, 1/3, 1/3, 1/6)
                bw = (k == 0 || k == 3) ? dt * T(1.0 / 6.0) : dt * T(1.0 / 3.0);
                aw = (k == 2) ? dt : dt * T(0.5);
            }
            else { bw = dt; aw = dt; }
            static_for<0, NV>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const T v0 = A.v[i * B + lane];
                const T kv = first ? v0 : sb[(2 * NV + i) * SBSr];
                const T ka = first ? A.a[i * B + lane] : as[i];
                T accv, acca;
                if (!rk4) { accv = bw * kv; acca = bw * ka; }
                else
                {
                    accv = first ? bw * kv : sb[i * SBSr] + bw * kv;
                    acca = first ? bw * ka : sb[(NV + i) * SBSr] + bw * ka;
                    if (k != 3) { sb[i * SBSr] = accv; sb[(NV + i) * SBSr] = acca; }
                }
                if (k == 3) { incv[i] = accv; vs[i] = v0 + acca; }
                else { incv[i] = aw * kv; vs[i] = v0 + aw * ka; sb[(2 * NV + i) * SBSr] = vs[i]; }
            });
            integrate_q<T, Tp>(P, q0, incv, qs);
            if (k == 3)
            {
                static_for<0, NQ>([&](auto ic) { A.q[decltype(ic)::value * B + lane] = qs[decltype(ic)::value]; });
                static_for<0, NV>([&](auto ic) { A.v[decltype(ic)::value * B + lane] = vs[decltype(ic)::value]; });
            }
        }
        eval_any<T, Tp, CON>(P, qs, vs, cmd, w, C, lane, B, 0);
        static_for<0, NV>([&](auto ic) { as[decltype(ic)::value] = w.ddq[decltype(ic)::value]; });
        if (k == -1 || k == 3)
            static_for<0, NV>([&](auto ic) { adst[decltype(ic)::value * B + lane] = as[decltype(ic)::value]; });
    }
    if constexpr (CON::ON)
    {
        // constraint model: kinematics of its own at the closing state + the forces of the multipliers the closing
        // evaluation stored
        if (n_evals <= 0) return;
        static_for<0, NQ>([&](auto ic) { qs[decltype(ic)::value] = A.q[decltype(ic)::value * B + lane]; });
        static_for<0, NV>([&](auto ic) { vs[decltype(ic)::value] = A.v[decltype(ic)::value * B + lane]; });
        static_for<0, NV>([&](auto ic) { as[decltype(ic)::value] = A.a[decltype(ic)::value * B + lane]; });
        const int status = w.status;
        eval_kinematics<T, Tp>(P, qs, vs, cmd, w);
        w.status |= status;
        constraint_forces_from_multipliers<T, Tp>(P, w, C, lane, B);
        extra_terms_and_outputs<T, Tp>(P, A, lane, qs, vs, as, w, A.update_sensors != 0);
    }
    else
    {
        if (n_evals <= 0 || A.mode == MODE_DYNAMICS) return;
        // the output pass takes its kinematics from an evaluation of its own at the closing state: nothing but
        // the sweeps' own operands stays live inside the evaluations of the loop
        // (state read back from the rows the closing evaluation wrote: qs / vs / as need not survive the loop)
        static_for<0, NQ>([&](auto ic) { qs[decltype(ic)::value] = A.q[decltype(ic)::value * B + lane]; });
        static_for<0, NV>([&](auto ic) { vs[decltype(ic)::value] = A.v[decltype(ic)::value * B + lane]; });
        static_for<0, NV>([&](auto ic) { as[decltype(ic)::value] = A.a[decltype(ic)::value * B + lane]; });
        const int status = w.status;
        eval_kinematics<T, Tp>(P, qs, vs, cmd, w);
        w.status |= status;
        // Engine::start: refuse huge initial contact forces (engine.cc:1310-1346)
        if (A.mode == MODE_START || A.mode == MODE_RESET)
        {
            T fmax2 = T(0);
            static_for<0, Tp::NC>([&](auto cc) {
                const Sp<T> & f = w.cf[decltype(cc)::value];
                fmax2 = fmax_(fmax2, dot(f.l, f.l));
            });
            if (fmax2 > T(1e10)) w.status |= JM_LANE_FORCE_OVERFLOW;
        }
        extra_terms_and_outputs<T, Tp>(P, A, lane, qs, vs, as, w,
                                       A.update_sensors != 0 || (!stepping && A.mode != MODE_REFRESH));
        if (A.status) A.status[lane] = (A.mode == MODE_REFRESH) ? (A.status[lane] | w.status) : w.status;
        return;
    }
    if (A.status) A.status[lane] = w.status;
}

#ifndef JM_HOST_EMU
// VAR: the instantiation that reads the applied wrenches (BatchArgs::applied)
template<class T, class Tp, bool VAR = false>
__global__ void __launch_bounds__(64) k_batch(const BatchArgs<T> A)
{
    const long long lane = (long long)blockIdx.x * 64 + threadIdx.x;
    if (lane >= A.B) return;
    lane_run<T, Tp, 64, typename std::conditional<VAR, NoConA, NoCon>::type>(A, lane, lane_lds<T, Tp>() + threadIdx.x);
}
#endif
}  // namespace jm
