// oracle.cpp -- CPU restatement of the reference's per-step physics. TEST INFRASTRUCTURE ONLY.
//
// This file is the parity oracle and the "port" CPU baseline of the MI355X-native engine.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
// product path (jiminy_amd/, libjm_*.so) never does.
//
// It restates, in scalar float64 and in the reference's order of operations, the single
// threaded algorithm of duburcqa/jiminy v1.8.12 (paths relative to the reference tree):
//
//   dynamics()        Engine::computeRobotsDynamics      core/src/engine/engine.cc:3585-3708
//   forward_kin()     Engine::computeForwardKinematics   core/src/engine/engine.cc:2957-3014
//   contact_*()       computeContactDynamicsAtFrame/computeContactDynamics  engine.cc:3117-3238,
//                     computeCollisionForces engine.cc:3394-3425,
//                     convertForceGlobalFrameToJoint core/src/utilities/pinocchio.cc:794-809
//   motor_efforts()   SimpleMotor::computeEffort         core/src/hardware/basic_motors.cc:83-143
//   aba()             pinocchio_overload::aba + AbaBackwardStep
//                     core/include/jiminy/core/robot/pinocchio_overload_algorithms.h:126-489
//   try_step_*()      AbstractStepper::tryStep, AbstractRungeKuttaStepper::tryStepImpl,
//                     EulerExplicitStepper::tryStepImpl   core/src/stepper/*.cc,
//                     RK4 tableau core/include/jiminy/core/stepper/runge_kutta4_stepper.h:12-23,
//                     State::sum core/include/jiminy/core/stepper/lie_group.h:446-455
//   extra_terms()     computeExtraTerms                  core/src/engine/engine.cc:800-905
//   sensors()         ImuSensor/ContactSensor/ForceSensor/EncoderSensor/EffortSensor::set
//                     core/src/hardware/basic_sensors.cc:142-164,267-277,368-387,509-539,604-618
//   start()           Engine::start                      core/src/engine/engine.cc:952-1533
//
// The parts of the path that live in Pinocchio v2.7.0 (pinned by the reference in
// build_tools/build_install_deps_unix.sh:219-229, NOT vendored under /root/reference) are
// restated from the published algorithm: AbaForwardStep1/2, forwardKinematics, joint calc()
// per type, SE3/Motion/Force/Inertia algebra, integrate() on SE(3) and SO(2), energies.
//
// PARITY PINNING: the reference core cannot be built or imported here (no Eigen/Boost/
// Pinocchio/hpp-fcl) and holds no stored numeric vectors for this path.  The
// oracle is pinned instead against the reference's own known-answer laws re-expressed in
// tests/test_oracle_*.py (pendulum vs analytic/expm/scipy, armature, two-mass spring chain,
// contact equilibrium, friction steady state, energy conservation, IMU closed form,
// diff(v)/dt == a) and against an independently coded CRBA/RNEA (oracle/rbd_numpy.py).
// For ANYmal/Atlas-sized floating-base models against the real binary: "parity unpinned".
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "../include/jiminy_hip.h"

namespace
{
constexpr double INF = std::numeric_limits<double>::infinity();
constexpr double EPS = std::numeric_limits<double>::epsilon();

// ------------------------------------------------------------------ small algebra
struct V3
{
    double x = 0, y = 0, z = 0;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }

struct M3
{
    double m[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    static M3 identity()
    {
        M3 r;
        r.m[0][0] = r.m[1][1] = r.m[2][2] = 1.0;
        return r;
    }
};
inline V3 operator*(const M3 & A, V3 v)
{
    return {A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z,
            A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
            A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z};
}
inline V3 tmul(const M3 & A, V3 v)  // A^T v
{
    return {A.m[0][0] * v.x + A.m[1][0] * v.y + A.m[2][0] * v.z,
            A.m[0][1] * v.x + A.m[1][1] * v.y + A.m[2][1] * v.z,
            A.m[0][2] * v.x + A.m[1][2] * v.y + A.m[2][2] * v.z};
}
inline M3 operator*(const M3 & A, const M3 & B)
{
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
    return r;
}
inline M3 transpose(const M3 & A)
{
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r.m[i][j] = A.m[j][i];
    return r;
}
inline M3 operator+(const M3 & A, const M3 & B)
{
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r.m[i][j] = A.m[i][j] + B.m[i][j];
    return r;
}
inline M3 skew(V3 v)
{
    M3 r;
    r.m[0][1] = -v.z; r.m[0][2] = v.y;
    r.m[1][0] = v.z;  r.m[1][2] = -v.x;
    r.m[2][0] = -v.y; r.m[2][1] = v.x;
    return r;
}

struct SE3
{
    M3 R = M3::identity();
    V3 p;
};
inline SE3 operator*(const SE3 & A, const SE3 & B) { return {A.R * B.R, A.p + A.R * B.p}; }

// Spatial motion / force, [linear; angular]
struct Motion
{
    V3 lin, ang;
};
struct Force
{
    V3 lin, ang;
};
inline Motion operator+(Motion a, Motion b) { return {a.lin + b.lin, a.ang + b.ang}; }
inline Force operator+(Force a, Force b) { return {a.lin + b.lin, a.ang + b.ang}; }
inline Force operator-(Force a, Force b) { return {a.lin - b.lin, a.ang - b.ang}; }
// SE3::act / actInv on motions and forces
inline Motion act(const SE3 & M, Motion m)
{
    V3 Rw = M.R * m.ang;
    return {M.R * m.lin + cross(M.p, Rw), Rw};
}
inline Motion actInv(const SE3 & M, Motion m)
{
    return {tmul(M.R, m.lin - cross(M.p, m.ang)), tmul(M.R, m.ang)};
}
inline Force act(const SE3 & M, Force f)
{
    V3 Rf = M.R * f.lin;
    return {Rf, M.R * f.ang + cross(M.p, Rf)};
}
inline Force actInv(const SE3 & M, Force f)
{
    return {tmul(M.R, f.lin), tmul(M.R, f.ang - cross(M.p, f.lin))};
}
// Motion x Motion and Motion x* Force
inline Motion crossm(Motion a, Motion b)
{
    return {cross(a.ang, b.lin) + cross(a.lin, b.ang), cross(a.ang, b.ang)};
}
inline Force crossf(Motion a, Force f)
{
    return {cross(a.ang, f.lin), cross(a.ang, f.ang) + cross(a.lin, f.lin)};
}

struct Inertia
{
    double mass = 0;
    V3 c;
    M3 I;  // about the COM
};
inline Force mul(const Inertia & Y, Motion v)  // Inertia::__mult__
{
    V3 l = Y.mass * (v.lin - cross(Y.c, v.ang));
    return {l, Y.I * v.ang + cross(Y.c, l)};
}
inline Force vxiv(const Inertia & Y, Motion v)  // v x* (Y v)
{
    V3 mcxw = Y.mass * cross(Y.c, v.ang);
    V3 mv_mcxw = Y.mass * v.lin - mcxw;
    return {cross(v.ang, mv_mcxw),
            cross(v.ang, cross(Y.c, mv_mcxw) + Y.I * v.ang) - cross(v.lin, mcxw)};
}
inline double vtiv(const Inertia & Y, Motion v)  // v^T Y v
{
    V3 cxw = cross(Y.c, v.ang);
    V3 d = v.lin - cxw;
    return Y.mass * dot(d, d) + dot(v.ang, Y.I * v.ang);
}
inline Inertia act(const SE3 & M, const Inertia & Y)
{
    return {Y.mass, M.R * Y.c + M.p, M.R * Y.I * transpose(M.R)};
}
inline Inertia add(const Inertia & A, const Inertia & B)  // Inertia::operator+=
{
    const double mab = A.mass + B.mass;
    const double mab_inv = 1.0 / std::max(mab, EPS);
    const V3 AB = A.c - B.c;
    const M3 S = skew(AB);
    const M3 S2 = S * S;
    Inertia r;
    const double alpha = A.mass * B.mass * mab_inv;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r.I.m[i][j] = A.I.m[i][j] + B.I.m[i][j] - alpha * S2.m[i][j];
    r.c = (A.mass * mab_inv) * A.c + (B.mass * mab_inv) * B.c;
    r.mass = mab;
    return r;
}

struct M6
{
    double m[6][6];
    void zero() { std::memset(m, 0, sizeof(m)); }
};
inline M6 inertia_matrix(const Inertia & Y)  // Inertia::matrix()
{
    M6 r;
    r.zero();
    const M3 cx = skew(Y.c);
    const M3 cx2 = cx * cx;
    for (int i = 0; i < 3; ++i)
    {
        r.m[i][i] = Y.mass;
        for (int j = 0; j < 3; ++j)
        {
            r.m[i][3 + j] = -Y.mass * cx.m[i][j];
            r.m[3 + i][j] = Y.mass * cx.m[i][j];
            r.m[3 + i][3 + j] = Y.I.m[i][j] - Y.mass * cx2.m[i][j];
        }
    }
    return r;
}
// Congruence transform of a force<-motion map from the child frame to the parent frame:
// X_f Ia X_f^T with X_f = [[R, 0], [p^ R, R]]  (internal::SE3actOn)
inline M6 se3_act_on(const SE3 & M, const M6 & Ia)
{
    double X[6][6];
    std::memset(X, 0, sizeof(X));
    const M3 pR = skew(M.p) * M.R;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
        {
            X[i][j] = M.R.m[i][j];
            X[3 + i][3 + j] = M.R.m[i][j];
            X[3 + i][j] = pR.m[i][j];
        }
    double T[6][6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j)
        {
            double s = 0;
            for (int k = 0; k < 6; ++k)
                s += X[i][k] * Ia.m[k][j];
            T[i][j] = s;
        }
    M6 r;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j)
        {
            double s = 0;
            for (int k = 0; k < 6; ++k)
                s += T[i][k] * X[j][k];
            r.m[i][j] = s;
        }
    return r;
}
inline void to6(Motion m, double * o)
{
    o[0] = m.lin.x; o[1] = m.lin.y; o[2] = m.lin.z; o[3] = m.ang.x; o[4] = m.ang.y; o[5] = m.ang.z;
}
inline void to6(Force m, double * o)
{
    o[0] = m.lin.x; o[1] = m.lin.y; o[2] = m.lin.z; o[3] = m.ang.x; o[4] = m.ang.y; o[5] = m.ang.z;
}
inline Motion motion6(const double * o) { return {{o[0], o[1], o[2]}, {o[3], o[4], o[5]}}; }
inline Force force6(const double * o) { return {{o[0], o[1], o[2]}, {o[3], o[4], o[5]}}; }

inline M3 m3_from(const double * r)
{
    M3 A;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            A.m[i][j] = r[3 * i + j];
    return A;
}
inline V3 v3_from(const double * r) { return {r[0], r[1], r[2]}; }

inline M3 quat_to_matrix(double x, double y, double z, double w)  // Eigen toRotationMatrix
{
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    M3 R;
    R.m[0][0] = 1 - (tyy + tzz); R.m[0][1] = txy - twz;       R.m[0][2] = txz + twy;
    R.m[1][0] = txy + twz;       R.m[1][1] = 1 - (txx + tzz); R.m[1][2] = tyz - twx;
    R.m[2][0] = txz - twy;       R.m[2][1] = tyz + twx;       R.m[2][2] = 1 - (txx + tyy);
    return R;
}
inline void matrix_to_quat(const M3 & R, double * q /* xyzw */)  // Eigen quaternion from matrix
{
    double t = R.m[0][0] + R.m[1][1] + R.m[2][2];
    if (t > 0)
    {
        t = std::sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R.m[2][1] - R.m[1][2]) * t;
        q[1] = (R.m[0][2] - R.m[2][0]) * t;
        q[2] = (R.m[1][0] - R.m[0][1]) * t;
    }
    else
    {
        int i = 0;
        if (R.m[1][1] > R.m[0][0]) i = 1;
        if (R.m[2][2] > R.m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R.m[i][i] - R.m[j][j] - R.m[k][k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R.m[k][j] - R.m[j][k]) * t;
        q[j] = (R.m[j][i] + R.m[i][j]) * t;
        q[k] = (R.m[k][i] + R.m[i][k]) * t;
    }
}
inline M3 rot_axis_angle(V3 a, double c, double s)  // Rodrigues
{
    const double oc = 1 - c;
    M3 R;
    R.m[0][0] = c + oc * a.x * a.x;         R.m[0][1] = oc * a.x * a.y - s * a.z;   R.m[0][2] = oc * a.x * a.z + s * a.y;
    R.m[1][0] = oc * a.y * a.x + s * a.z;   R.m[1][1] = c + oc * a.y * a.y;         R.m[1][2] = oc * a.y * a.z - s * a.x;
    R.m[2][0] = oc * a.z * a.x - s * a.y;   R.m[2][1] = oc * a.z * a.y + s * a.x;   R.m[2][2] = c + oc * a.z * a.z;
    return R;
}
// exp6 of a twist [v; w] (upstream explog.hpp), Taylor expansion below eps^(1/4)
inline SE3 exp6(Motion nu)
{
    const V3 v = nu.lin, w = nu.ang;
    const double t2 = dot(w, w);
    const double t = std::sqrt(t2);
    const double prec = std::pow(EPS, 0.25);
    double alpha_wxv, alpha_v, alpha_w, diag;
    if (t < prec)
    {
        alpha_wxv = 0.5 - t2 / 24;
        alpha_v = 1 - t2 / 6;
        alpha_w = 1.0 / 6 - t2 / 120;
        diag = 1 - t2 / 2;
    }
    else
    {
        const double st = std::sin(t), ct = std::cos(t);
        const double inv_t2 = 1.0 / t2;
        alpha_wxv = (1 - ct) * inv_t2;
        alpha_v = st / t;
        alpha_w = (1 - alpha_v) * inv_t2;
        diag = ct;
    }
    SE3 M;
    M.p = alpha_v * v + (alpha_w * dot(w, v)) * w + alpha_wxv * cross(w, v);
    const double ww[3] = {w.x, w.y, w.z};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            M.R.m[i][j] = alpha_wxv * ww[i] * ww[j];
    M.R.m[0][1] -= alpha_v * w.z; M.R.m[1][0] += alpha_v * w.z;
    M.R.m[0][2] += alpha_v * w.y; M.R.m[2][0] -= alpha_v * w.y;
    M.R.m[1][2] -= alpha_v * w.x; M.R.m[2][1] += alpha_v * w.x;
    M.R.m[0][0] += diag; M.R.m[1][1] += diag; M.R.m[2][2] += diag;
    return M;
}

// ------------------------------------------------------------------ model
struct MotorP
{
    int joint, idx_v, flags;
    double red, effort_limit, velocity_limit, inv_slope, fvp, fvn, fdp, fdn, fds;
};
struct FrameP
{
    int joint;
    SE3 M;
};
struct EncoderP
{
    int joint, joint_side;
    double red;
};
struct Model
{
    int njoints = 0, nq = 0, nv = 0;
    std::vector<int> parent, jtype, idx_q, idx_v;
    std::vector<V3> axis;
    std::vector<SE3> placement;
    std::vector<Inertia> inertia;
    std::vector<double> rotor, qlo, qhi;
    // flexibility of the spherical joints (jm_model_desc::flex_stiffness / flex_damping, [3 * njoints]; empty: none)
    std::vector<double> flex_k, flex_d;
    std::vector<MotorP> motors;
    std::vector<FrameP> contacts, imus, forces;
    // frames a user `FrameConstraint(frame, maskDoFs)` may hold (Model::addConstraint; jm_model_desc::cframe_*), bit d of the
    // mask = dof d of (x, y, z, rot x, rot y, rot z) is fixed (frame_constraint.cc:11-35)
    std::vector<FrameP> cframes;
    std::vector<int> cframe_mask;
    // kind (JM_XKIND_*), second parent joint and 8 parameters of every constraint frame (jm_model_desc::cframe_kind ...)
    std::vector<int> cframe_kind, cframe_joint2;
    std::vector<double> cframe_params;
    // 1-dof joints a user `JointConstraint(joint)` may hold on a row OF ITS OWN (jm_model_desc::cjoint_joint), next to the
    // joint's bound constraint like in the reference (model.cc:884-905)
    std::vector<int> cjoints;
    std::vector<int> contact_sensors, effort_sensors;
    std::vector<EncoderP> encoders;
    // force sensor -> (contact index, relative placement)  basic_sensors.cc:326-350
    std::vector<std::vector<std::pair<int, SE3>>> force_pairs;
};

inline int jt_nv(int t) { return t == JM_JT_FREEFLYER ? 6 : (t == JM_JT_SPHERICAL ? 3 : (t == JM_JT_NONE ? 0 : 1)); }
inline bool is_revolute(int t) { return (t >= JM_JT_RX && t <= JM_JT_RU) || (t >= JM_JT_RUBX && t <= JM_JT_RUBU); }
inline bool is_prismatic(int t) { return t >= JM_JT_PX && t <= JM_JT_PU; }
inline bool is_unbounded(int t) { return t >= JM_JT_RUBX && t <= JM_JT_RUBU; }
inline bool has_bounds(int t) { return t >= JM_JT_RX && t <= JM_JT_PU; }

// ------------------------------------------------------------------ engine (one robot)
struct Engine
{
    Model mdl;
    jm_options opt;
    // state (RobotState, engine.h:134-156)
    std::vector<double> q, v, a, command, u, uMotor, uTransmission;
    std::vector<Force> fExternal;       // joint frame
    std::vector<Force> contactFrameForces;  // RobotData::contactFrameForces (joint frame)
    std::vector<Force> contactForces;   // Robot::contactForces_ (contact frame)
    // pinocchio::Data
    std::vector<SE3> liMi, oMi;
    std::vector<Motion> dv, da, da_gf;  // data.v, data.a, data.a_gf
    std::vector<Force> df, dh, fBody;   // data.f, data.h, fPrev buffer
    std::vector<M6> Yaba;
    std::vector<Inertia> Ycrb;
    struct JData
    {
        double U[6][6], Dinv[6][6], UDinv[6][6];
    };
    std::vector<JData> jd;
    std::vector<double> du, ddq;
    double kinetic = 0, potential = 0;
    V3 com0;
    Force hg, dhg;
    // sensors
    std::vector<double> imu, force, contact, encoder, effort;
    int status = 0;
    long iter = 0;
    // stepper buffers (sized on first use: the reference guarantees no allocation inside `step`,
    // core/unit/engine_sanity_check.cc:118-121, and the CPU baseline is timed on this code)
    std::vector<double> st_kv[4], st_ka[4], st_incv, st_inca, st_qs, st_vs, st_as;
    // ---- `contacts.model = "constraint"`: per-robot constraint registry + solver state
    jm_constraint_options copt{JM_CONTACT_SPRING_DAMPER, 100, 0.0, 20.0, 1.0e-3, 1.0e-5, 1.0e-4, -1.0};
    struct BoundCon  // JointConstraint (core/src/constraints/joint_constraint.cc)
    {
        int joint = 0;
        bool enabled = false, reversed = false;
        double ref = 0.0, lambda = 0.0;
        // a user-registered JointConstraint on the same joint (Model::addConstraint, model.cc:926-936; flag bit 2 of the batch
        // state): bilateral, always enabled, no direction; solved first in every sweep (constraint_solvers.cc:112-128); its
        // multiplier is not restored into RobotState::u (engine.cc:3771-3790 restores the bounds only).  While it holds, the
        // joint's own bound constraint is not switched (the device shares the row).
        bool locked = false;
    };
    struct FrameCon  // FrameConstraint with dofsFixed = {x, y, z, rot z} (core/src/robot/model.cc:817-823)
    {
        bool enabled = false;
        double lambda[4] = {0, 0, 0, 0};
        // FrameConstraint::rotationLocal_ (frame_constraint.cc:62-68: columns t0, t1, n from the ground normal) and the
        // penetration depth, both refreshed at every evaluation while the constraint is enabled (engine.cc:3184-3193)
        M3 Rloc = M3::identity();
        double depth = 0.0;
    };
    struct UserFrameCon  // user-registered FrameConstraint (core/src/constraints/frame_constraint.cc), USER registry
    {
        bool enabled = false;          // registered for this robot (batch flag bit 0)
        SE3 ref;                       // transformRef_: the frame's pose at `start` (FrameConstraint::reset) or the caller's
        double lambda[6] = {0, 0, 0, 0, 0, 0};   // per dof of the mask order (x, y, z, rx, ry, rz); unused dofs stay 0
    };
    std::vector<BoundCon> bcon;
    std::vector<FrameCon> fcon;
    struct UserJointCon  // user-registered JointConstraint (joint_constraint.cc), USER registry, own row
    {
        int joint = 0;
        bool enabled = false;
        double ref = 0.0, lambda = 0.0;
    };
    std::vector<UserFrameCon> xcon;
    std::vector<UserJointCon> jcon;
    std::vector<double> uInternal;
    int pgsIterLast = 0;
    // optional per-lane storage of the constraint state for the batch drivers ([rows][B])
    int32_t * con_flags = nullptr;
    double * con_data = nullptr;
    const double * lane_friction = nullptr;  // [B] contacts.friction of every lane, or null
    // [6 per spherical joint][B] stiffness 3, damping 3 of the flexibility joints of every lane (JM_F_FLEXIBILITY), or null
    const double * lane_flex = nullptr;
    // per-lane (x, y) offset of the ground-profile queries ([2][B], batch drivers; JM_F_GROUND_OFFSET): every environment its
    // own patch of the terrain
    const double * lane_ground_offset = nullptr;
    double ground_ox = 0, ground_oy = 0;
    // ---- per-lane model (Model::addBiasedToExtendedModel output, model.cc:1166-1236): rows per joint
    // mass | com 3 | inertia xx xy xz yy yz zz | joint placement translation 3, `[13 * njoints][B]`, or null
    const double * model_lane = nullptr;
    // ---- world.groundProfile as a sampled height map (engine.h:292-302): heights `[ny][nx]` at
    // (x0 + ix dx, y0 + iy dy), bilinear patches, flat continuation outside; null = flat ground at z = 0
    const double * ground_h = nullptr;
    int ground_nx = 0, ground_ny = 0;
    double ground_x0 = 0, ground_y0 = 0, ground_dx = 1, ground_dy = 1;
    // ---- external wrenches on frames of the root joint (impulse / profile forces, engine.cc:1838-2016):
    // `[6 K][B]` world-aligned (force, moment) at frame offsets applied_p (root joint frame), K <= 4
    const double * applied = nullptr;
    int applied_k = 0;
    double applied_p[12] = {0};
    int applied_joint[4] = {1, 1, 1, 1};   // parent joint of every frame (1 = the root joint)
    double applied_now[24] = {0};   // the current lane's wrenches
};
// world.groundProfile(x, y) -> height, unit normal
inline void ground_profile(const Engine & e, double x, double y, double & h, V3 & n)
{
    if (!e.ground_h) { h = 0.0; n = {0, 0, 1}; return; }
    const int nx = e.ground_nx, ny = e.ground_ny;
    x += e.ground_ox; y += e.ground_oy;
    double u = (x - e.ground_x0) / e.ground_dx, w = (y - e.ground_y0) / e.ground_dy;
    const bool in_x = u >= 0.0 && u <= (double)(nx - 1), in_y = w >= 0.0 && w <= (double)(ny - 1);
    u = std::min(std::max(u, 0.0), (double)(nx - 1));
    w = std::min(std::max(w, 0.0), (double)(ny - 1));
    int ix = std::min((int)u, nx - 2), iy = std::min((int)w, ny - 2);
    if (ix < 0) ix = 0;
    if (iy < 0) iy = 0;
    const double fx = u - ix, fy = w - iy;
    const double h00 = e.ground_h[iy * nx + ix], h10 = e.ground_h[iy * nx + ix + 1];
    const double h01 = e.ground_h[(iy + 1) * nx + ix], h11 = e.ground_h[(iy + 1) * nx + ix + 1];
    h = (1.0 - fy) * ((1.0 - fx) * h00 + fx * h10) + fy * ((1.0 - fx) * h01 + fx * h11);
    // outside the grid the ground continues flat (height of the nearest edge sample, no slope across the edge)
    const double dhdx = in_x ? ((1.0 - fy) * (h10 - h00) + fy * (h11 - h01)) / e.ground_dx : 0.0;
    const double dhdy = in_y ? ((1.0 - fx) * (h01 - h00) + fx * (h11 - h10)) / e.ground_dy : 0.0;
    const double inv = 1.0 / std::sqrt(dhdx * dhdx + dhdy * dhdy + 1.0);
    n = {-dhdx * inv, -dhdy * inv, inv};
}
// impulse / profile forces (Engine::computeExternalForces, engine.cc:3481-3560): the world-aligned wrench applied at a
// frame goes to the frame's PARENT JOINT, in the joint frame (convertForceGlobalFrameToJoint, utilities/pinocchio.cc:794-809)
inline void add_applied_wrenches(Engine & e)
{
    for (int k = 0; k < e.applied_k; ++k)
    {
        const int j = e.applied_joint[k];
        const V3 F = {e.applied_now[6 * k], e.applied_now[6 * k + 1], e.applied_now[6 * k + 2]};
        const V3 M = {e.applied_now[6 * k + 3], e.applied_now[6 * k + 4], e.applied_now[6 * k + 5]};
        const V3 p = {e.applied_p[3 * k], e.applied_p[3 * k + 1], e.applied_p[3 * k + 2]};
        Force f;
        f.lin = tmul(e.oMi[j].R, F);
        f.ang = tmul(e.oMi[j].R, M) + cross(p, f.lin);
        e.fExternal[j] = e.fExternal[j] + f;
    }
}

V3 joint_axis(const Model & m, int j)
{
    switch (m.jtype[j])
    {
    case JM_JT_RX: case JM_JT_PX: case JM_JT_RUBX: return {1, 0, 0};
    case JM_JT_RY: case JM_JT_PY: case JM_JT_RUBY: return {0, 1, 0};
    case JM_JT_RZ: case JM_JT_PZ: case JM_JT_RUBZ: return {0, 0, 1};
    default: return m.axis[j];
    }
}

// joint calc(): transform M_j(q) and joint velocity S*qd
void joint_calc(const Model & m, int j, const double * q, const double * v, SE3 & Mj, Motion & vj)
{
    const int t = m.jtype[j];
    const double * qj = q + m.idx_q[j];
    const double * vjv = v + m.idx_v[j];
    Mj = SE3();
    vj = Motion();
    if (t == JM_JT_FREEFLYER)
    {
        Mj.p = {qj[0], qj[1], qj[2]};
        Mj.R = quat_to_matrix(qj[3], qj[4], qj[5], qj[6]);
        vj = motion6(vjv);
        return;
    }
    if (t == JM_JT_SPHERICAL)   // JointModelSphericalTpl::calc: rotation of the unit quaternion, angular joint velocity
    {
        Mj.R = quat_to_matrix(qj[0], qj[1], qj[2], qj[3]);
        vj.ang = {vjv[0], vjv[1], vjv[2]};
        return;
    }
    const V3 ax = joint_axis(m, j);
    if (is_revolute(t))
    {
        double c, s;
        if (is_unbounded(t)) { c = qj[0]; s = qj[1]; }
        else { c = std::cos(qj[0]); s = std::sin(qj[0]); }
        Mj.R = rot_axis_angle(ax, c, s);
        vj.ang = vjv[0] * ax;
    }
    else  // prismatic
    {
        Mj.p = qj[0] * ax;
        vj.lin = vjv[0] * ax;
    }
}
// S * x for a 1-dof joint or the free-flyer
Motion S_times(const Model & m, int j, const double * x)
{
    const int t = m.jtype[j];
    if (t == JM_JT_FREEFLYER) return motion6(x);
    Motion r;
    if (t == JM_JT_SPHERICAL) { r.ang = {x[0], x[1], x[2]}; return r; }
    const V3 ax = joint_axis(m, j);
    if (is_revolute(t)) r.ang = x[0] * ax; else r.lin = x[0] * ax;
    return r;
}

// pinocchio::forwardKinematics(model, data, q, v) + oMi (engine.cc:2957-2969).
// data.a from the `a` argument is not needed by the dynamics and is refreshed by extra_terms().
void forward_kin(Engine & e, const double * q, const double * v)
{
    const Model & m = e.mdl;
    for (int j = 1; j < m.njoints; ++j)
    {
        SE3 Mj; Motion vj;
        joint_calc(m, j, q, v, Mj, vj);
        e.liMi[j] = m.placement[j] * Mj;
        const int p = m.parent[j];
        e.dv[j] = vj;
        if (p > 0)
        {
            e.oMi[j] = e.oMi[p] * e.liMi[j];
            e.dv[j] = e.dv[j] + actInv(e.liMi[j], e.dv[p]);
        }
        else
            e.oMi[j] = e.liMi[j];
    }
}

// Engine::computeContactDynamics (engine.cc:3197-3238)
V3 contact_law(const jm_options & o, V3 n, double depth, V3 vWorld)
{
    V3 f;
    if (depth < 0.0)
    {
        const double vDepth = dot(vWorld, n);
        const double fN = -std::min(o.contact_stiffness * depth + o.contact_damping * vDepth, 0.0);
        f = fN * n;
        const V3 vT = vWorld - vDepth * n;
        const double vRatio = std::min(norm(vT) / o.contact_transition_velocity, 1.0);
        const double fT = o.contact_friction * vRatio * fN;
        f = f - fT * vT;
        if (o.contact_transition_eps > EPS)
        {
            const double blend = -depth / o.contact_transition_eps;
            f = std::tanh(2.0 * blend) * f;
        }
    }
    return f;
}

// computeContactDynamicsAtFrame, spring-damper branch, flat ground h=0 n=z (engine.cc:3117-3195)
Force contact_at_frame(const Engine & e, const FrameP & fr)
{
    const SE3 oMf = e.oMi[fr.joint] * fr.M;
    double hGround;
    V3 n;
    ground_profile(e, oMf.p.x, oMf.p.y, hGround, n);
    const double depth = (oMf.p.z - hGround) * n.z;
    Force fl;
    if (depth < 0.0)
    {
        const V3 vLocal = actInv(fr.M, e.dv[fr.joint]).lin;  // getFrameVelocity(LOCAL).linear
        const V3 vWorld = oMf.R * vLocal;
        const V3 fW = contact_law(e.opt, n, depth, vWorld);
        // convertForceGlobalFrameToJoint (utilities/pinocchio.cc:794-809)
        fl.lin = tmul(e.oMi[fr.joint].R, fW);
        fl.ang = tmul(e.oMi[fr.joint].R, V3{0, 0, 0});
        fl.ang = fl.ang + cross(fr.M.p, fl.lin);
    }
    return fl;
}

// SimpleMotor::computeEffort (basic_motors.cc:83-143), one motor
void motor_law(const MotorP & mp, double vj, double command, double & uMotorOut, double & uTransmissionOut)
{
    const double vMotor = mp.red * vj;
    double effortMin = -INF, effortMax = INF;
    if (mp.flags & JM_MOTOR_EFFORT_LIMIT)
    {
        effortMin = -mp.effort_limit;
        effortMax = mp.effort_limit;
        if (mp.flags & JM_MOTOR_VELOCITY_LIMIT)
        {
            const double velocityDelta = mp.effort_limit * mp.inv_slope;
            if (velocityDelta > 0.0)
            {
                const double velocityThr = std::max(mp.velocity_limit - velocityDelta, 0.0);
                effortMin *= std::clamp((mp.velocity_limit + vMotor) / (mp.velocity_limit - velocityThr), 0.0, 1.0);
                effortMax *= std::clamp((mp.velocity_limit - vMotor) / (mp.velocity_limit - velocityThr), 0.0, 1.0);
            }
        }
    }
    const double uMotor = std::clamp(command, effortMin, effortMax);
    double uT = mp.red * uMotor;
    if (mp.flags & JM_MOTOR_FRICTION)
    {
        if (vj > 0.0) uT += mp.fvp * vj + mp.fdp * std::tanh(mp.fds * vj);
        else uT += mp.fvn * vj + mp.fdn * std::tanh(mp.fds * vj);
    }
    uMotorOut = uMotor;
    uTransmissionOut = uT;
}

// Robot::computeMotorEfforts -> AbstractMotorBase::computeEffortAll (abstract_motor.cc:461-493)
void motor_efforts(Engine & e, const double * v)
{
    for (size_t i = 0; i < e.mdl.motors.size(); ++i)
    {
        const MotorP & mp = e.mdl.motors[i];
        motor_law(mp, v[mp.idx_v], e.command[i], e.uMotor[i], e.uTransmission[i]);
    }
}

// 6x6 SPD inverse through Cholesky (internal::PerformStYSInversion: llt().solveInPlace(I))
void spd_inverse(int n, const double A[6][6], double Ainv[6][6])
{
    double L[6][6] = {};
    for (int j = 0; j < n; ++j)
    {
        double s = A[j][j];
        for (int k = 0; k < j; ++k) s -= L[j][k] * L[j][k];
        L[j][j] = std::sqrt(s);
        for (int i = j + 1; i < n; ++i)
        {
            double t = A[i][j];
            for (int k = 0; k < j; ++k) t -= L[i][k] * L[j][k];
            L[i][j] = t / L[j][j];
        }
    }
    for (int c = 0; c < n; ++c)
    {
        double y[6];
        for (int i = 0; i < n; ++i)
        {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
            y[i] = s / L[i][i];
        }
        for (int i = n - 1; i >= 0; --i)
        {
            double s = y[i];
            for (int k = i + 1; k < n; ++k) s -= L[k][i] * Ainv[k][c];
            Ainv[i][c] = s / L[i][i];
        }
    }
}

// pinocchio_overload::aba (pinocchio_overload_algorithms.h:437-489)
void aba(Engine & e, const double * q, const double * v, const double * tau, const std::vector<Force> & fext)
{
    const Model & m = e.mdl;
    const V3 g = {e.opt.gravity[0], e.opt.gravity[1], e.opt.gravity[2]};
    const V3 gw = {e.opt.gravity[3], e.opt.gravity[4], e.opt.gravity[5]};
    e.dv[0] = Motion();
    e.da_gf[0] = {-g, -gw};
    for (int i = 0; i < m.nv; ++i) e.du[i] = tau[i];
    // Pass 1 (AbaForwardStep1)
    for (int j = 1; j < m.njoints; ++j)
    {
        SE3 Mj; Motion vj;
        joint_calc(m, j, q, v, Mj, vj);
        const int p = m.parent[j];
        e.liMi[j] = m.placement[j] * Mj;
        e.dv[j] = vj;
        if (p > 0) e.dv[j] = e.dv[j] + actInv(e.liMi[j], e.dv[p]);
        e.da_gf[j] = crossm(e.dv[j], vj);  // c = 0 for every supported joint
        e.Yaba[j] = inertia_matrix(m.inertia[j]);
        e.df[j] = vxiv(m.inertia[j], e.dv[j]);
        e.df[j] = e.df[j] - fext[j];
    }
    // Pass 2 (AbaBackwardStep :136-167 and calc_aba specialisations)
    for (int j = m.njoints - 1; j > 0; --j)
    {
        const int p = m.parent[j];
        const int t = m.jtype[j];
        const int iv = m.idx_v[j];
        M6 & Ia = e.Yaba[j];
        Engine::JData & jd = e.jd[j];
        double fv[6];
        to6(e.df[j], fv);
        if (t == JM_JT_FREEFLYER)
        {
            for (int k = 0; k < 6; ++k) e.du[iv + k] -= fv[k];
            double StU[6][6];
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b) { jd.U[a][b] = Ia.m[a][b]; StU[a][b] = Ia.m[a][b]; }
            for (int a = 0; a < 6; ++a) StU[a][a] += m.rotor[iv + a];
            spd_inverse(6, StU, jd.Dinv);
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b)
                {
                    double s = 0;
                    for (int k = 0; k < 6; ++k) s += jd.U[a][k] * jd.Dinv[k][b];
                    jd.UDinv[a][b] = s;
                }
            if (p > 0)
                for (int a = 0; a < 6; ++a)
                    for (int b = 0; b < 6; ++b)
                    {
                        double s = 0;
                        for (int k = 0; k < 6; ++k) s += jd.UDinv[a][k] * jd.U[b][k];
                        Ia.m[a][b] -= s;
                    }
        }
        else if (t == JM_JT_SPHERICAL)
        {
            // JointModelSphericalTpl::calc_aba: S = [0; I3]: U = Ia[:, 3:6], D = U[3:6, :] + armature, Ia -= U D^-1 U^T
            for (int k = 0; k < 3; ++k) e.du[iv + k] -= fv[3 + k];
            double D[6][6] = {};
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 3; ++b) jd.U[a][b] = Ia.m[a][3 + b];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) D[a][b] = jd.U[3 + a][b];
            for (int a = 0; a < 3; ++a) D[a][a] += m.rotor[iv + a];
            spd_inverse(3, D, jd.Dinv);
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 3; ++b)
                {
                    double s = 0;
                    for (int k = 0; k < 3; ++k) s += jd.U[a][k] * jd.Dinv[k][b];
                    jd.UDinv[a][b] = s;
                }
            if (p > 0)
                for (int a = 0; a < 6; ++a)
                    for (int b = 0; b < 6; ++b)
                    {
                        double s = 0;
                        for (int k = 0; k < 3; ++k) s += jd.UDinv[a][k] * jd.U[b][k];
                        Ia.m[a][b] -= s;
                    }
        }
        else
        {
            const V3 ax = joint_axis(m, j);
            const double axv[3] = {ax.x, ax.y, ax.z};
            const int off = is_revolute(t) ? 3 : 0;
            // u -= S^T f
            e.du[iv] -= axv[0] * fv[off] + axv[1] * fv[off + 1] + axv[2] * fv[off + 2];
            // U = Ia S ; D = S^T U + Im
            for (int a = 0; a < 6; ++a)
                jd.U[a][0] = Ia.m[a][off] * axv[0] + Ia.m[a][off + 1] * axv[1] + Ia.m[a][off + 2] * axv[2];
            const double D = axv[0] * jd.U[off][0] + axv[1] * jd.U[off + 1][0] + axv[2] * jd.U[off + 2][0] + m.rotor[iv];
            jd.Dinv[0][0] = 1.0 / D;
            for (int a = 0; a < 6; ++a) jd.UDinv[a][0] = jd.U[a][0] * jd.Dinv[0][0];
            if (p > 0)
                for (int a = 0; a < 6; ++a)
                    for (int b = 0; b < 6; ++b) Ia.m[a][b] -= jd.UDinv[a][0] * jd.U[b][0];
        }
        if (p > 0)
        {
            double pa[6], agf[6];
            to6(e.df[j], pa);
            to6(e.da_gf[j], agf);
            for (int a = 0; a < 6; ++a)
            {
                double s = 0;
                for (int k = 0; k < 6; ++k) s += Ia.m[a][k] * agf[k];
                pa[a] += s;
            }
            const int n = jt_nv(t);
            for (int a = 0; a < 6; ++a)
            {
                double s = 0;
                for (int k = 0; k < n; ++k) s += jd.UDinv[a][k] * e.du[iv + k];
                pa[a] += s;
            }
            e.df[j] = force6(pa);
            const M6 T = se3_act_on(e.liMi[j], Ia);
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b) e.Yaba[p].m[a][b] += T.m[a][b];
            e.df[p] = e.df[p] + act(e.liMi[j], e.df[j]);
        }
    }
    // Pass 3 (AbaForwardStep2)
    for (int j = 1; j < m.njoints; ++j)
    {
        const int p = m.parent[j];
        const int t = m.jtype[j];
        const int iv = m.idx_v[j];
        const int n = jt_nv(t);
        const Engine::JData & jd = e.jd[j];
        e.da_gf[j] = e.da_gf[j] + actInv(e.liMi[j], e.da_gf[p]);
        double agf[6];
        to6(e.da_gf[j], agf);
        for (int a = 0; a < n; ++a)
        {
            double s = 0;
            for (int k = 0; k < n; ++k) s += jd.Dinv[a][k] * e.du[iv + k];
            double r = 0;
            for (int k = 0; k < 6; ++k) r += jd.UDinv[k][a] * agf[k];
            e.ddq[iv + a] = s - r;
        }
        e.da_gf[j] = e.da_gf[j] + S_times(m, j, &e.ddq[iv]);
    }
}

// ------------------------------------------------------------------ constraint contact model
// Restates, in the reference's own formulation (dense joint-space inertia matrix + Cholesky,
// dense constraint Jacobian, J M^-1 J^T, projected Gauss-Seidel):
//   Engine::computeInternalDynamics  bounds hysteresis      engine.cc:3253-3338, 3340-3363
//   computeContactDynamicsAtFrame    contact hysteresis     engine.cc:3117-3195 (CONSTRAINT branch)
//   Model::computeConstraints        crba + drift kinematics core/src/robot/model.cc:1238-1287
//   JointConstraint / FrameConstraint::computeJacobianAndDrift
//                                    core/src/constraints/joint_constraint.cc:139-163, frame_constraint.cc:103-183
//   pinocchio_overload::crba / computeJMinvJt / solveJMinvJtv
//                                    pinocchio_overload_algorithms.h:99-124, 491-551
//   PGSSolver                        core/src/solver/constraint_solvers.cc:107-448
//   Engine::computeAcceleration      engine.cc:3710-3866
// pinocchio::crba, nonLinearEffects, cholesky::decompose/solve (Pinocchio v2.7.0, not in tree) are
// replaced by their dense definitions: M = sum_bodies X^T I X + diag(rotorInertia), nle = RNEA(q, v, 0),
// M = L L^T.  The device path uses a different formulation (articulated-body solves), which is the point.
struct Dense
{
    int r = 0, c = 0;
    std::vector<double> d;
    Dense() {}
    Dense(int r_, int c_) : r(r_), c(c_), d((size_t)r_ * c_, 0.0) {}
    double & operator()(int i, int j) { return d[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return d[(size_t)i * c + j]; }
};

// lower Cholesky factor in place (Eigen::LLT); returns false if not positive definite
bool llt_inplace(Dense & A)
{
    const int n = A.r;
    for (int j = 0; j < n; ++j)
    {
        double s = A(j, j);
        for (int k = 0; k < j; ++k) s -= A(j, k) * A(j, k);
        if (!(s > 0.0)) return false;
        A(j, j) = std::sqrt(s);
        for (int i = j + 1; i < n; ++i)
        {
            double t = A(i, j);
            for (int k = 0; k < j; ++k) t -= A(i, k) * A(j, k);
            A(i, j) = t / A(j, j);
        }
    }
    return true;
}
void llt_solve(const Dense & L, double * x)  // x <- (L L^T)^-1 x
{
    const int n = L.r;
    for (int i = 0; i < n; ++i)
    {
        double s = x[i];
        for (int k = 0; k < i; ++k) s -= L(i, k) * x[k];
        x[i] = s / L(i, i);
    }
    for (int i = n - 1; i >= 0; --i)
    {
        double s = x[i];
        for (int k = i + 1; k < n; ++k) s -= L(k, i) * x[k];
        x[i] = s / L(i, i);
    }
}

void init_constraints(Engine & e)
{
    const Model & m = e.mdl;
    if (e.bcon.empty() && e.fcon.empty())
    {
        for (int j = 1; j < m.njoints; ++j)
            if (has_bounds(m.jtype[j]))
            {
                Engine::BoundCon b;
                b.joint = j;
                e.bcon.push_back(b);
            }
        e.fcon.resize(m.contacts.size());
    }
    if (e.xcon.size() != m.cframes.size()) e.xcon.resize(m.cframes.size());
    if (e.jcon.size() != m.cjoints.size())
    {
        e.jcon.resize(m.cjoints.size());
        for (size_t i = 0; i < e.jcon.size(); ++i) e.jcon[i].joint = m.cjoints[i];
    }
    e.uInternal.assign(m.nv, 0.0);
}

// Engine::start: resetConstraints + "enable constraints by default" (engine.cc:1266-1308)
void reset_constraints(Engine & e)
{
    init_constraints(e);
    for (auto & b : e.bcon)
    {
        b.ref = e.q[e.mdl.idx_q[b.joint]];  // JointConstraint::reset: configurationRef_ = q (bounds and user locks alike)
        b.lambda = 0.0;
        b.reversed = false;
        b.enabled = true;
    }
    for (auto & f : e.fcon)
    {
        f.enabled = true;
        for (double & l : f.lambda) l = 0.0;
    }
    // user constraints keep their enabled state (Model::addConstraints enables USER constraints, model.cc:926-936);
    // FrameConstraint::reset: transformRef_ = oMf at the start configuration, multipliers zeroed (frame_constraint.cc:77-101)
    for (size_t i = 0; i < e.xcon.size(); ++i)
    {
        if (!e.xcon[i].enabled) continue;    // (not registered for this robot: no such constraint object)
        e.xcon[i].ref = e.oMi[e.mdl.cframes[i].joint] * e.mdl.cframes[i].M;
        if (e.mdl.cframe_kind[i] == JM_XKIND_DISTANCE)
        {
            // DistanceConstraint::reset: distanceRef_ = |p_1 - p_2| (distance_constraint.cc:73-77), kept in the first slot
            const double * pr = &e.mdl.cframe_params[8 * i];
            const SE3 & M2 = e.oMi[e.mdl.cframe_joint2[i]];
            const V3 p2 = M2.p + M2.R * V3{pr[1], pr[2], pr[3]};
            e.xcon[i].ref.p = {norm(e.xcon[i].ref.p - p2), 0.0, 0.0};
            e.xcon[i].ref.R = M3::identity();
        }
        for (double & l : e.xcon[i].lambda) l = 0.0;
    }
    for (auto & jc : e.jcon)
    {
        if (!jc.enabled) continue;
        jc.ref = e.q[e.mdl.idx_q[jc.joint]];   // JointConstraint::reset: configurationRef_ = q (joint_constraint.cc:56-83)
        jc.lambda = 0.0;
    }
}

// computePositionLimitsForcesAlgo (engine.cc:3253-3338) for every bounded joint
void toggle_bounds(Engine & e, const double * q)
{
    const Model & m = e.mdl;
    const double eps = e.opt.contact_transition_eps;
    for (auto & b : e.bcon)
    {
        if (b.locked) { b.enabled = true; b.reversed = false; continue; }
        const int iq = m.idx_q[b.joint];
        const double qj = q[iq], lo = m.qlo[iq], hi = m.qhi[iq];
        if (hi < qj || qj < lo)
        {
            b.ref = std::clamp(qj, lo, hi);
            b.reversed = hi < qj;
            b.enabled = true;
        }
        else if (lo + eps < qj && qj < hi - eps)
        {
            b.lambda = 0.0;  // AbstractConstraintBase::disable
            b.enabled = false;
        }
    }
}

// computeContactDynamicsAtFrame, CONSTRAINT branch (engine.cc:3133-3193): height and normal of world.groundProfile
// under the frame, first-order depth, hysteresis, and -- while enabled -- the constraint's local frame from the
// normal (FrameConstraint::setNormal, frame_constraint.cc:62-68); the reference transform moved to the surface shows up
// as deltaPosition = depth * n in the Baumgarte term (compute_acceleration)
void toggle_contacts(Engine & e)
{
    const Model & m = e.mdl;
    for (size_t i = 0; i < m.contacts.size(); ++i)
    {
        const FrameP & fr = m.contacts[i];
        const SE3 oMf = e.oMi[fr.joint] * fr.M;
        double heightGround;
        V3 normalGround;
        ground_profile(e, oMf.p.x, oMf.p.y, heightGround, normalGround);
        const double depth = (oMf.p.z - heightGround) * normalGround.z;
        if (depth < 0.0) e.fcon[i].enabled = true;
        else if (depth > e.opt.contact_transition_eps)
        {
            for (double & l : e.fcon[i].lambda) l = 0.0;
            e.fcon[i].enabled = false;
        }
        if (e.fcon[i].enabled)
        {
            const V3 n = normalGround;
            V3 c1 = cross(n, V3{1.0, 0.0, 0.0});
            const double inv = 1.0 / std::sqrt(dot(c1, c1));
            c1 = {c1.x * inv, c1.y * inv, c1.z * inv};
            const V3 c0 = cross(c1, n);
            M3 R;
            R.m[0][0] = c0.x; R.m[1][0] = c0.y; R.m[2][0] = c0.z;
            R.m[0][1] = c1.x; R.m[1][1] = c1.y; R.m[2][1] = c1.z;
            R.m[0][2] = n.x; R.m[1][2] = n.y; R.m[2][2] = n.z;
            e.fcon[i].Rloc = R;
            e.fcon[i].depth = depth;
        }
        // contactFrameForces stay zero with this model; contactForces_ = actInv(0) (engine.cc:3417-3424)
        e.contactFrameForces[i] = Force();
        e.contactForces[i] = Force();
    }
}

bool has_constraints(const Engine & e)
{
    for (const auto & b : e.bcon) if (b.enabled) return true;
    for (const auto & f : e.fcon) if (f.enabled) return true;
    for (const auto & x : e.xcon) if (x.enabled) return true;
    for (const auto & jc : e.jcon) if (jc.enabled) return true;
    return false;
}

// PGSSolver::ProjectedGaussSeidelSolver over the packed active rows (constraint_solvers.cc:107-333)
struct PgsRowSet
{
    // one entry per active constraint: start index, dim, number of blocks (1 = joint bound, 3 = contact)
    struct C { int start, dim, nblocks; };
    std::vector<C> cons;
};
struct PgsOptions
{
    double friction, torsion, tolAbs, tolRel;
    unsigned iterMax;
};
// PGSSolver::ProjectedGaussSeidelIter (constraint_solvers.cc:107-222): one sweep at relaxation factor w
void pgs_sweep(const PgsOptions & po, const PgsRowSet & rs, const Dense & A, const std::vector<double> & b, double w,
               std::vector<double> & x, std::vector<double> & y)
{
    const int n = (int)b.size();
    const double friction = po.friction, torsion = po.torsion;
    auto col_dot = [&](int i) {
        double s = 0.0;
        for (int k = 0; k < n; ++k) s += A(k, i) * x[k];
        return s;
    };
    // first, the unbounded constraints, coefficient by coefficient (constraint_solvers.cc:112-128)
    for (const auto & c : rs.cons)
    {
        if (c.nblocks != 0) continue;
        for (int i = c.start; i < c.start + c.dim; ++i)
        {
            y[i] = b[i] - col_dot(i);
            x[i] += y[i] / A(i, i);
        }
    }
    for (int blk = 0; blk < 3; ++blk)
        for (const auto & c : rs.cons)
        {
            if (c.nblocks <= blk) continue;
            const int o = c.start;
            // blocks (constraint_solvers.cc:48-87): 0: {2} lo 0 hi inf (joint bound: {0});
            // 1: {3, 2} hi = torsion; 2: {0, 1, 2} hi = friction
            int fIndex[3] = {0, 0, 0}, fSize = 1;
            double lo = 0.0, hi = INF;
            bool isZero = false;
            if (c.nblocks == 3)
            {
                if (blk == 0) { fIndex[0] = 2; fSize = 1; }
                else if (blk == 1) { fIndex[0] = 3; fIndex[1] = 2; fSize = 2; hi = torsion; isZero = torsion < EPS; }
                else { fIndex[0] = 0; fIndex[1] = 1; fIndex[2] = 2; fSize = 3; hi = friction; isZero = friction < EPS; }
            }
            const int i0 = o + fIndex[0];
            double & el = x[i0];
            if (isZero)
            {
                el *= 0;
                for (int j = 1; j < fSize - 1; ++j) x[o + fIndex[j]] *= 0;
                continue;
            }
            double A_max = A(i0, i0);
            y[i0] = b[i0] - col_dot(i0);
            for (int j = 1; j < fSize - 1; ++j)
            {
                const int k = o + fIndex[j];
                y[k] = b[k] - col_dot(k);
                if (A(k, k) > A_max) A_max = A(k, k);
            }
            el += w * y[i0] / A_max;
            for (int j = 1; j < fSize - 1; ++j)
            {
                const int k = o + fIndex[j];
                x[k] += w * y[k] / A_max;
            }
            if (fSize == 1) el = std::clamp(el, lo, hi);
            else
            {
                const double thr = hi * x[o + fIndex[fSize - 1]];
                if (fSize == 2) el = std::clamp(el, -thr, thr);
                else
                {
                    double squaredNorm = el * el;
                    for (int j = 1; j < fSize - 1; ++j) squaredNorm += x[o + fIndex[j]] * x[o + fIndex[j]];
                    if (squaredNorm > thr * thr)
                    {
                        const double scale = thr / std::sqrt(squaredNorm);
                        el *= scale;
                        for (int j = 1; j < fSize - 1; ++j) x[o + fIndex[j]] *= scale;
                    }
                }
            }
        }
}

// PGSSolver::ProjectedGaussSeidelSolver (constraint_solvers.cc:224-326)
bool pgs_solve(const PgsOptions & po, const PgsRowSet & rs, const Dense & A, const std::vector<double> & b,
               std::vector<double> & x, int & iters, std::vector<double> * yOut = nullptr)
{
    const int n = (int)b.size();
    const unsigned iterMax = po.iterMax;
    std::vector<double> y(n, 0.0), yPrev(n, 0.0);
    bool converged = false;
    iters = (int)iterMax;
    for (unsigned iter = 0; iter < iterMax; ++iter)
    {
        yPrev = y;
        const double ratio = (static_cast<double>(iterMax - 20U) - iter) / (iterMax - 20U - 30U);
        double w = 1.0;
        if (ratio < 1.0)
        {
            w = 0.01;
            if (ratio > 0.0) w += (1.0 - 0.01) * std::pow(ratio, 2.0);
        }
        pgs_sweep(po, rs, A, b, w, x, y);
        double ymax = 0.0;
        for (int i = 0; i < n; ++i) ymax = std::max(ymax, std::fabs(y[i]));
        const double tol = po.tolAbs + po.tolRel * ymax + EPS;
        bool ok = true;
        for (int i = 0; i < n; ++i) ok &= std::fabs(y[i] - yPrev[i]) < tol;
        if (ok)
        {
            iters = (int)iter + 1;
            converged = true;
            break;
        }
    }
    if (yOut) *yOut = y;
    return converged;
}
bool pgs_solve(const Engine & e, const PgsRowSet & rs, const Dense & A, const std::vector<double> & b,
               std::vector<double> & x, int & iters)
{
    const PgsOptions po{e.opt.contact_friction, e.copt.torsion, e.copt.tol_abs, e.copt.tol_rel, (unsigned)e.copt.pgs_iter_max};
    return pgs_solve(po, rs, A, b, x, iters);
}

V3 log3(const M3 & R);
// Engine::computeAcceleration (engine.cc:3710-3866). `u` is RobotState::u (in: efforts without the
// constraint forces; out: + joint-bound multipliers), e.fExternal likewise for the contact forces.
void compute_acceleration(Engine & e, const double * q, const double * v, std::vector<double> & u, bool ignoreBounds)
{
    const Model & m = e.mdl;
    const int nv = m.nv, NJ = m.njoints;
    e.status &= ~JM_LANE_SOLVER_FAILURE;  // status of the last evaluation
    e.pgsIterLast = 0;
    if (!has_constraints(e))
    {
        aba(e, q, v, u.data(), e.fExternal);
        return;
    }
    const V3 g = {e.opt.gravity[0], e.opt.gravity[1], e.opt.gravity[2]};
    const V3 gw = {e.opt.gravity[3], e.opt.gravity[4], e.opt.gravity[5]};
    // ---- world-frame joint Jacobian columns (data.J) and supports
    std::vector<Motion> Jw(nv);
    std::vector<int> dof_joint(nv);
    for (int j = 1; j < NJ; ++j)
    {
        const int n = jt_nv(m.jtype[j]), iv = m.idx_v[j];
        for (int k = 0; k < n; ++k)
        {
            double unit[6] = {0, 0, 0, 0, 0, 0};
            unit[k] = 1.0;
            Jw[iv + k] = act(e.oMi[j], S_times(m, j, unit));
            dof_joint[iv + k] = j;
        }
    }
    auto supports = [&](int joint, int dof) {  // dof belongs to an ancestor-or-self of joint
        for (int j = joint; j > 0; j = m.parent[j]) if (dof_joint[dof] == j) return true;
        return false;
    };
    // ---- joint-space inertia matrix with rotor armature (pinocchio_overload::crba :99-124)
    Dense M(nv, nv);
    for (int j = 1; j < NJ; ++j)
    {
        std::vector<int> sup;
        for (int a = 0; a < nv; ++a) if (supports(j, a)) sup.push_back(a);
        std::vector<Motion> X(sup.size());
        std::vector<Force> IX(sup.size());
        for (size_t a = 0; a < sup.size(); ++a)
        {
            X[a] = actInv(e.oMi[j], Jw[sup[a]]);
            IX[a] = mul(m.inertia[j], X[a]);
        }
        for (size_t a = 0; a < sup.size(); ++a)
            for (size_t b2 = 0; b2 < sup.size(); ++b2)
                M(sup[a], sup[b2]) += dot(X[a].lin, IX[b2].lin) + dot(X[a].ang, IX[b2].ang);
    }
    for (int i = 0; i < nv; ++i) M(i, i) += m.rotor[i];
    // ---- non-linear effects: RNEA(q, v, 0) with gravity (pinocchio::nonLinearEffects)
    std::vector<double> nle(nv, 0.0);
    {
        std::vector<Motion> ag(NJ);
        std::vector<Force> f(NJ);
        ag[0] = {-g, -gw};
        for (int j = 1; j < NJ; ++j)
        {
            const Motion vj = S_times(m, j, &v[m.idx_v[j]]);
            ag[j] = crossm(e.dv[j], vj) + actInv(e.liMi[j], ag[m.parent[j]]);
            f[j] = mul(m.inertia[j], ag[j]) + vxiv(m.inertia[j], e.dv[j]);
        }
        for (int j = NJ - 1; j > 0; --j)
        {
            const int n = jt_nv(m.jtype[j]), iv = m.idx_v[j];
            for (int k = 0; k < n; ++k)
            {
                double unit[6] = {0, 0, 0, 0, 0, 0};
                unit[k] = 1.0;
                const Motion S = S_times(m, j, unit);
                nle[iv + k] = dot(S.lin, f[j].lin) + dot(S.ang, f[j].ang);
            }
            if (m.parent[j] > 0) f[m.parent[j]] = f[m.parent[j]] + act(e.liMi[j], f[j]);
        }
    }
    // ---- drift kinematics: accelerations with ddq = 0 and no gravity (model.cc:1252-1268)
    std::vector<Motion> adrift(NJ);
    for (int j = 1; j < NJ; ++j)
    {
        const Motion vj = S_times(m, j, &v[m.idx_v[j]]);
        adrift[j] = crossm(e.dv[j], vj);
        if (m.parent[j] > 0) adrift[j] = adrift[j] + actInv(e.liMi[j], adrift[m.parent[j]]);
    }
    // ---- data.u = u + sum_j J_j^T fext_j (engine.cc:3733-3749)
    std::vector<double> du(u);
    for (int j = 1; j < NJ; ++j)
    {
        double fv[6];
        to6(e.fExternal[j], fv);
        bool any = false;
        for (double x : fv) any |= std::fabs(x) > EPS;
        if (!any) continue;
        for (int a = 0; a < nv; ++a)
            if (supports(j, a))
            {
                const Motion X = actInv(e.oMi[j], Jw[a]);  // LOCAL joint Jacobian column
                du[a] += dot(X.lin, e.fExternal[j].lin) + dot(X.ang, e.fExternal[j].ang);
            }
    }
    // ---- Jacobian, drift, multipliers of the enabled constraints, packed (constraint_solvers.cc:340-360)
    const double omega = 2.0 * M_PI * e.copt.stabilization_freq;  // setBaumgarteFreq (abstract_constraint.cc:88-98)
    const double kp = omega * omega, kd = 2.0 * omega;
    // user-registered constraints keep gains of their own (abstract_constraint.cc:88-98; Engine::start only sets those of the
    // internal ones, engine.cc:1276-1285): jm_constraint_options::user_stabilization_freq, < 0 = share the pair above
    const double omega_u = 2.0 * M_PI * e.copt.user_stabilization_freq;
    const double kp_u = e.copt.user_stabilization_freq < 0.0 ? kp : omega_u * omega_u;
    const double kd_u = e.copt.user_stabilization_freq < 0.0 ? kd : 2.0 * omega_u;
    PgsRowSet rs;
    int rows = 0;
    for (const auto & b : e.bcon) if (b.enabled) { rs.cons.push_back({rows, 1, b.locked ? 0 : 1}); rows += 1; }
    for (const auto & f : e.fcon) if (f.enabled) { rs.cons.push_back({rows, 4, 3}); rows += 4; }
    // USER registry last (constraintNodeTypesAll, model.h:43-46): unbounded constraints, nBlocks = 0
    for (size_t i = 0; i < e.xcon.size(); ++i)
        if (e.xcon[i].enabled)
        {
            const int dim = __builtin_popcount((unsigned)m.cframe_mask[i] & 63u);
            rs.cons.push_back({rows, dim, 0});
            rows += dim;
        }
    for (const auto & jc : e.jcon) if (jc.enabled) { rs.cons.push_back({rows, 1, 0}); rows += 1; }
    Dense J(rows, nv);
    std::vector<double> gamma(rows, 0.0), lambda(rows, 0.0);
    int r = 0;
    for (const auto & b : e.bcon)
    {
        if (!b.enabled) continue;
        const int iq = m.idx_q[b.joint], iv = m.idx_v[b.joint];
        const double sgn = b.reversed ? -1.0 : 1.0;
        J(r, iv) = sgn;
        gamma[r] = sgn * ((b.locked ? kp_u : kp) * (q[iq] - b.ref) + (b.locked ? kd_u : kd) * v[iv]);
        lambda[r] = b.lambda;
        ++r;
    }
    for (size_t i = 0; i < e.fcon.size(); ++i)
    {
        if (!e.fcon[i].enabled) continue;
        const FrameP & fr = m.contacts[i];
        const SE3 oMf = e.oMi[fr.joint] * fr.M;
        const double depth = e.fcon[i].depth;
        const M3 & Rloc = e.fcon[i].Rloc;
        const V3 nG = Rloc * V3{0.0, 0.0, 1.0};
        // frame Jacobian in (rotationLocal, frame translation): transformLocal.actInv(data.J col)
        // (frame_constraint.cc:136-146)
        for (int a = 0; a < nv; ++a)
        {
            if (!supports(fr.joint, a)) continue;
            const V3 lin = tmul(Rloc, Jw[a].lin - cross(oMf.p, Jw[a].ang));
            J(r + 0, a) = lin.x; J(r + 1, a) = lin.y; J(r + 2, a) = lin.z; J(r + 3, a) = dot(nG, Jw[a].ang);
        }
        // velocity and drift acceleration, LOCAL_WORLD_ALIGNED
        const Motion vl = actInv(fr.M, e.dv[fr.joint]);
        const Motion al = actInv(fr.M, adrift[fr.joint]);
        const V3 vlin = oMf.R * vl.lin, vang = oMf.R * vl.ang;
        V3 dlin = oMf.R * al.lin, dang = oMf.R * al.ang;
        dlin = dlin + cross(vang, vlin);
        // Baumgarte: reference transform moved to the ground surface every evaluation
        // (engine.cc:3186-3193) -> deltaPosition = depth * n, deltaRotation = 0
        dlin = dlin + kp * (depth * nG) + kd * vlin;
        dang = dang + kd * vang;
        // drift in the local frame (frame_constraint.cc:172-174)
        dlin = tmul(Rloc, dlin);
        gamma[r] = dlin.x; gamma[r + 1] = dlin.y; gamma[r + 2] = dlin.z; gamma[r + 3] = dot(nG, dang);
        for (int k = 0; k < 4; ++k) lambda[r + k] = e.fcon[i].lambda[k];
        r += 4;
    }
    // user FrameConstraint (frame_constraint.cc:103-183): world-aligned frame Jacobian rows of the fixed dofs; drift =
    // classical frame acceleration (ddq = 0, no gravity) + kp (p - p_ref) | kp log3(R R_ref^T) + kd v, own gains
    for (size_t i = 0; i < e.xcon.size(); ++i)
    {
        if (!e.xcon[i].enabled) continue;
        const FrameP & fr = m.cframes[i];
        const SE3 oMf = e.oMi[fr.joint] * fr.M;
        const Motion vl = actInv(fr.M, e.dv[fr.joint]);
        const Motion al = actInv(fr.M, adrift[fr.joint]);
        const V3 vlin = oMf.R * vl.lin, vang = oMf.R * vl.ang;
        V3 dlin = oMf.R * al.lin, dang = oMf.R * al.ang;
        dlin = dlin + cross(vang, vlin);
        const int kind = m.cframe_kind[i];
        const double * par = &m.cframe_params[8 * i];
        if (kind == JM_XKIND_DISTANCE)
        {
            // DistanceConstraint::computeJacobianAndDrift (distance_constraint.cc:80-150): one row along the direction
            // between the two frame origins
            const int j2 = m.cframe_joint2[i];
            const V3 pl2 = {par[1], par[2], par[3]};
            const V3 p2 = e.oMi[j2].p + e.oMi[j2].R * pl2;
            const V3 delta = oMf.p - p2;
            const double dn = norm(delta);
            const V3 dir = (1.0 / dn) * delta;
            // world-aligned velocity / classical drift acceleration of the second frame origin
            const Motion v2j = e.dv[j2], a2j = j2 > 0 ? adrift[j2] : Motion();
            const V3 v2lin = e.oMi[j2].R * (v2j.lin + cross(v2j.ang, pl2)), v2ang = e.oMi[j2].R * v2j.ang;
            V3 a2lin = e.oMi[j2].R * (a2j.lin + cross(a2j.ang, pl2));
            a2lin = a2lin + cross(v2ang, v2lin);
            const V3 dvel = vlin - v2lin;
            for (int a = 0; a < nv; ++a)
            {
                double jv = 0.0;
                if (supports(fr.joint, a)) jv += dot(dir, Jw[a].lin - cross(oMf.p, Jw[a].ang));
                if (j2 > 0 && supports(j2, a)) jv -= dot(dir, Jw[a].lin - cross(p2, Jw[a].ang));
                J(r, a) = jv;
            }
            const double dvp = dot(dvel, dir);
            gamma[r] = dot(dir, dlin - a2lin) + (dot(dvel, dvel) - dvp * dvp) / dn + kp_u * (dn - e.xcon[i].ref.p.x) + kd_u * dvp;
            lambda[r] = e.xcon[i].lambda[0];
            ++r;
            continue;
        }
        V3 rd = {0.0, 0.0, 0.0};      // radius * (direction from the contact point to the frame origin): skewRadius_ = [rd]x
        if (kind == JM_XKIND_SPHERE || kind == JM_XKIND_WHEEL)
        {
            // SphereConstraint (sphere_constraint.cc:77-140) / WheelConstraint (wheel_constraint.cc:85-153): the three linear
            // rows taken at the contact point, J = J_lin + [rd]x J_ang
            const double radius = par[0];
            const V3 nrm = {par[1], par[2], par[3]};
            const V3 rel = oMf.p - e.xcon[i].ref.p;
            double deltaPosition;
            V3 extra = {0.0, 0.0, 0.0};
            if (kind == JM_XKIND_SPHERE)
            {
                rd = radius * nrm;
                deltaPosition = dot(rel, nrm);
            }
            else
            {
                const V3 axis = oMf.R * V3{par[4], par[5], par[6]};
                const V3 x = cross(cross(axis, nrm), axis);
                const double xn = norm(x);
                const V3 y = (1.0 / xn) * x;
                rd = radius * y;
                deltaPosition = dot(rel + radius * (nrm - y), nrm);
                const V3 daxis = cross(vang, axis);
                const V3 dx = cross(cross(daxis, nrm), axis) + cross(cross(axis, nrm), daxis);
                const V3 z = (1.0 / xn) * dx;
                const V3 dy = z - dot(y, z) * y;
                extra = cross(radius * dy, vang);      // dskewRadius_ * omega
            }
            const V3 velocity = vlin + cross(rd, vang);
            // (dlin / dang still hold the classical drift acceleration here: the Baumgarte terms of this kind differ)
            const V3 drift = dlin + cross(rd, dang) + extra + (kp_u * deltaPosition) * nrm + kd_u * velocity;
            const double drift3[3] = {drift.x, drift.y, drift.z};
            for (int d = 0; d < 3; ++d)
            {
                for (int a = 0; a < nv; ++a)
                {
                    if (!supports(fr.joint, a)) continue;
                    const V3 lin = Jw[a].lin - cross(oMf.p, Jw[a].ang) + cross(rd, Jw[a].ang);
                    const double row3[3] = {lin.x, lin.y, lin.z};
                    J(r, a) = row3[d];
                }
                gamma[r] = drift3[d];
                lambda[r] = e.xcon[i].lambda[d];
                ++r;
            }
            continue;
        }
        const V3 dp = oMf.p - e.xcon[i].ref.p;
        const V3 dr = log3(oMf.R * transpose(e.xcon[i].ref.R));
        dlin = dlin + kp_u * dp + kd_u * vlin;
        dang = dang + kp_u * dr + kd_u * vang;
        const double drift6[6] = {dlin.x, dlin.y, dlin.z, dang.x, dang.y, dang.z};
        for (int d = 0; d < 6; ++d)
        {
            if (!((m.cframe_mask[i] >> d) & 1)) continue;
            for (int a = 0; a < nv; ++a)
            {
                if (!supports(fr.joint, a)) continue;
                const V3 lin = Jw[a].lin - cross(oMf.p, Jw[a].ang);   // transformLocal.actInv(data.J col), rotationLocal = I
                const double row6[6] = {lin.x, lin.y, lin.z, Jw[a].ang.x, Jw[a].ang.y, Jw[a].ang.z};
                J(r, a) = row6[d];
            }
            gamma[r] = drift6[d];
            lambda[r] = e.xcon[i].lambda[d];
            ++r;
        }
    }
    // user JointConstraint on its own row (joint_constraint.cc:139-163): J = selector of the joint's dof, drift =
    // kp (q - q_ref) + kd v with the user gains; never reversed
    for (const auto & jc : e.jcon)
    {
        if (!jc.enabled) continue;
        const int iq = m.idx_q[jc.joint], iv = m.idx_v[jc.joint];
        J(r, iv) = 1.0;
        gamma[r] = kp_u * (q[iq] - jc.ref) + kd_u * v[iv];
        lambda[r] = jc.lambda;
        ++r;
    }
    // ---- JMinvJt (computeJMinvJt :491-533) + regularisation (constraint_solvers.cc:376-387)
    Dense L = M;
    if (!llt_inplace(L)) e.status |= JM_LANE_NAN;
    Dense Y(nv, rows);  // L^-1 J^T
    for (int c = 0; c < rows; ++c)
    {
        for (int i = 0; i < nv; ++i)
        {
            double s = J(c, i);
            for (int k = 0; k < i; ++k) s -= L(i, k) * Y(k, c);
            Y(i, c) = s / L(i, i);
        }
    }
    Dense A(rows, rows);
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < rows; ++j)
        {
            double s = 0.0;
            for (int k = 0; k < nv; ++k) s += Y(k, i) * Y(k, j);
            A(i, j) = s;
        }
    for (int i = 0; i < rows; ++i) A(i, i) += std::max(A(i, i) * e.copt.regularization, 1.0e-11);
    // ---- dynamic drift: torque_residual = M^-1 (u - nle); b = -gamma - J torque_residual
    std::vector<double> tr(nv);
    for (int i = 0; i < nv; ++i) tr[i] = du[i] - nle[i];
    llt_solve(L, tr.data());
    std::vector<double> b(rows);
    for (int i = 0; i < rows; ++i)
    {
        double s = 0.0;
        for (int k = 0; k < nv; ++k) s += J(i, k) * tr[k];
        b[i] = -gamma[i] - s;
    }
    // ---- multipliers
    bool ok = true;
    // `isUnbounded`: every enabled constraint is an unbounded one (constraint_solvers.cc:362-367) -> exact solve like `ignoreBounds`
    bool isUnbounded = true;
    for (const auto & c : rs.cons) isUnbounded &= c.nblocks == 0;
    if (ignoreBounds || isUnbounded)
    {
        Dense LA = A;
        if (!llt_inplace(LA)) e.status |= JM_LANE_NAN;
        lambda = b;
        llt_solve(LA, lambda.data());  // solveJMinvJtv :535-551
        e.pgsIterLast = 0;
    }
    else ok = pgs_solve(e, rs, A, b, lambda, e.pgsIterLast);
    if (std::getenv("ORC_PGS_TRACE")) std::fprintf(stderr, "pgs rows=%d sweeps=%d\n", (int)b.size(), (int)e.pgsIterLast);   // debugging aid
    if (!ok) e.status |= JM_LANE_SOLVER_FAILURE;
    // ---- ddq = M^-1 J^T lambda + torque_residual
    std::vector<double> ddq(nv, 0.0);
    for (int k = 0; k < nv; ++k)
    {
        double s = 0.0;
        for (int i = 0; i < rows; ++i) s += J(i, k) * lambda[i];
        ddq[k] = s;
    }
    llt_solve(L, ddq.data());
    for (int k = 0; k < nv; ++k) e.ddq[k] = ddq[k] + tr[k];
    // ---- write the multipliers back; bounds efforts and contact forces (engine.cc:3770-3857)
    r = 0;
    for (auto & bc : e.bcon)
    {
        if (!bc.enabled) continue;
        bc.lambda = lambda[r++];
        if (bc.locked) continue;   // (user constraints: the multiplier acts through ddq only)
        const int iv = m.idx_v[bc.joint];
        e.uInternal[iv] += bc.lambda;
        u[iv] += bc.lambda;
    }
    for (size_t i = 0; i < e.fcon.size(); ++i)
    {
        if (!e.fcon[i].enabled) continue;
        for (int k = 0; k < 4; ++k) e.fcon[i].lambda[k] = lambda[r + k];
        const FrameP & fr = m.contacts[i];
        const SE3 oMf = e.oMi[fr.joint] * fr.M;
        // multipliers live in the constraint's local frame: fextInGlobal = rotationLocal * fextInLocal (engine.cc:3805-3815)
        const V3 fW = e.fcon[i].Rloc * V3{lambda[r], lambda[r + 1], lambda[r + 2]};
        const V3 tW = e.fcon[i].Rloc * V3{0.0, 0.0, lambda[r + 3]};
        e.contactForces[i].lin = tmul(oMf.R, fW);
        e.contactForces[i].ang = tmul(oMf.R, tW);
        Force fl;  // convertForceGlobalFrameToJoint
        fl.lin = tmul(e.oMi[fr.joint].R, fW);
        fl.ang = tmul(e.oMi[fr.joint].R, tW) + cross(fr.M.p, fl.lin);
        e.fExternal[fr.joint] = e.fExternal[fr.joint] + fl;
        r += 4;
    }
    // user constraints: the multipliers act through ddq only (engine.cc:3770-3857 restores bounds and contacts alone)
    for (size_t i = 0; i < e.xcon.size(); ++i)
    {
        if (!e.xcon[i].enabled) continue;
        for (int d = 0; d < 6; ++d)
            if ((m.cframe_mask[i] >> d) & 1) e.xcon[i].lambda[d] = lambda[r++];
    }
    for (auto & jc : e.jcon) if (jc.enabled) jc.lambda = lambda[r++];
}

// quaternion::exp3 (pinocchio v2.7.0 math/quaternion.hpp): unit quaternion (x y z w) of the rotation vector
inline void quat_exp3(const double * v, double * out)
{
    const double t2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const double ts_prec = std::sqrt(std::sqrt(EPS));   // TaylorSeriesExpansion<double>::precision<3>()
    double k, w;
    if (t2 > ts_prec)
    {
        const double theta = std::sqrt(t2);
        k = std::sin(0.5 * theta) / theta;
        w = std::cos(0.5 * theta);
    }
    else
    {
        k = 0.5 - t2 / 48.0;
        w = 1.0 - t2 / 8.0;
    }
    out[0] = k * v[0]; out[1] = k * v[1]; out[2] = k * v[2]; out[3] = w;
}
// Hamilton product a * b of quaternions stored x y z w (Eigen)
inline void quat_mul(const double * a, const double * b, double * r)
{
    r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
// quaternion::log3 (pinocchio v2.7.0 math/quaternion.hpp): rotation vector of a unit quaternion, theta >= 0 its angle
inline V3 quat_log3(const double * q, double & theta)
{
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
    const double n = std::sqrt(n2);
    const double ts_prec = std::sqrt(std::sqrt(EPS));
    const double sgn = q[3] >= 0.0 ? 1.0 : -1.0;       // the shortest of the two rotations the quaternion stands for
    theta = 2.0 * std::atan2(n, sgn * q[3]);
    const double k = n2 > ts_prec ? sgn * theta / n : sgn * (2.0 / std::fabs(q[3])) * (1.0 - n2 / (3.0 * q[3] * q[3]));
    return {k * q[0], k * q[1], k * q[2]};
}
// Jlog3 (pinocchio v2.7.0 spatial/explog.hpp): derivative of log3 w.r.t. a rotation composed on the right
inline M3 jlog3(double theta, V3 lg)
{
    const double ts_prec = std::sqrt(std::sqrt(EPS));
    double alpha, diag;
    if (theta < ts_prec)
    {
        alpha = 1.0 / 12.0 + theta * theta / 720.0;
        diag = 0.5 * (2.0 - theta * theta / 6.0);
    }
    else
    {
        const double st = std::sin(theta), ct = std::cos(theta), st_1mct = st / (1.0 - ct);
        alpha = 1.0 / (theta * theta) - st_1mct / (2.0 * theta);
        diag = 0.5 * (theta * st_1mct);
    }
    const double l[3] = {lg.x, lg.y, lg.z};
    M3 J;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) J.m[a][b] = alpha * l[a] * l[b];
    for (int a = 0; a < 3; ++a) J.m[a][a] += diag;
    // addSkew(0.5 * log, J)
    J.m[0][1] -= 0.5 * l[2]; J.m[0][2] += 0.5 * l[1];
    J.m[1][0] += 0.5 * l[2]; J.m[1][2] -= 0.5 * l[0];
    J.m[2][0] -= 0.5 * l[1]; J.m[2][1] += 0.5 * l[0];
    return J;
}
// Flexibility efforts of the spherical joints (Engine::computeInternalDynamics, engine.cc:3365-3391):
// u -= Jlog3 (stiffness * log3(q)) + damping * w
void add_flexibility_efforts(Engine & e, const double * q, const double * v, std::vector<double> & u)
{
    const Model & m = e.mdl;
    if (m.flex_k.empty()) return;
    for (int j = 1; j < m.njoints; ++j)
    {
        if (m.jtype[j] != JM_JT_SPHERICAL) continue;
        const int iq = m.idx_q[j], iv = m.idx_v[j];
        double angle;
        const V3 aa = quat_log3(q + iq, angle);
        const M3 J = jlog3(angle, aa);
        const V3 ka = {m.flex_k[3 * j] * aa.x, m.flex_k[3 * j + 1] * aa.y, m.flex_k[3 * j + 2] * aa.z};
        V3 t3 = J * ka;
        // "Flexible joint angle must be smaller than 0.95 * pi" (engine.cc:3379-3383): the reference throws, i.e. the
        // evaluation fails -- a trial of the adaptive stepper is rejected (abstract_stepper.cc:33-54), a fixed step ends the
        // simulation.  Here the efforts become NaN: the acceleration is NaN and takes those very paths.
        if (angle > 0.95 * 3.14159265358979323846) t3.x = std::nan("");
        u[iv] -= t3.x + m.flex_d[3 * j] * v[iv];
        u[iv + 1] -= t3.y + m.flex_d[3 * j + 1] * v[iv + 1];
        u[iv + 2] -= t3.z + m.flex_d[3 * j + 2] * v[iv + 2];
    }
}

// Engine::computeRobotsDynamics with `contacts.model = "constraint"` (engine.cc:3585-3708):
// computeAllTerms (hysteresis) -> motors -> u -> computeAcceleration
void dynamics_constraint(Engine & e, const double * q, const double * v, double * a_out)
{
    const Model & m = e.mdl;
    if (e.uInternal.empty()) init_constraints(e);
    forward_kin(e, q, v);
    for (auto & f : e.fExternal) f = Force();
    if (e.applied_k > 0) add_applied_wrenches(e);
    std::fill(e.uInternal.begin(), e.uInternal.end(), 0.0);
    toggle_bounds(e, q);
    add_flexibility_efforts(e, q, v, e.uInternal);
    toggle_contacts(e);
    motor_efforts(e, v);
    for (int i = 0; i < m.nv; ++i) e.u[i] = e.uInternal[i];
    for (size_t i = 0; i < m.motors.size(); ++i) e.u[m.motors[i].idx_v] += e.uTransmission[i];
    compute_acceleration(e, q, v, e.u, false);
    for (int i = 0; i < m.nv; ++i)
    {
        a_out[i] = e.ddq[i];
        if (e.ddq[i] != e.ddq[i]) e.status |= JM_LANE_NAN;
    }
}

// Engine::computeRobotsDynamics for one robot (engine.cc:3585-3708), spring-damper contacts,
// discrete controller (command held), no user internal dynamics, no flexibility.
void dynamics(Engine & e, const double * q, const double * v, double * a_out)
{
    if (e.copt.contact_model == JM_CONTACT_CONSTRAINT)
    {
        dynamics_constraint(e, q, v, a_out);
        return;
    }
    const Model & m = e.mdl;
    forward_kin(e, q, v);
    // computeAllTerms: reset, internal dynamics (bounds -> status flag), contacts
    for (auto & f : e.fExternal) f = Force();
    for (int j = 1; j < m.njoints; ++j)
        if (has_bounds(m.jtype[j]))
        {
            const double qj = q[m.idx_q[j]];
            if (m.qhi[m.idx_q[j]] < qj || qj < m.qlo[m.idx_q[j]]) e.status |= JM_LANE_OUT_OF_BOUNDS;
        }
    for (size_t i = 0; i < m.contacts.size(); ++i)
    {
        const FrameP & fr = m.contacts[i];
        e.contactFrameForces[i] = contact_at_frame(e, fr);
        e.fExternal[fr.joint] = e.fExternal[fr.joint] + e.contactFrameForces[i];
        e.contactForces[i] = actInv(fr.M, e.contactFrameForces[i]);
    }
    if (e.applied_k > 0) add_applied_wrenches(e);
    motor_efforts(e, v);
    for (int i = 0; i < m.nv; ++i) e.u[i] = 0.0;  // uInternal + uCustom
    add_flexibility_efforts(e, q, v, e.u);
    for (size_t i = 0; i < m.motors.size(); ++i) e.u[m.motors[i].idx_v] += e.uTransmission[i];
    aba(e, q, v, e.u.data(), e.fExternal);
    for (int i = 0; i < m.nv; ++i)
    {
        a_out[i] = e.ddq[i];
        if (e.ddq[i] != e.ddq[i]) e.status |= JM_LANE_NAN;
    }
}

// pinocchio::integrate (lie_group.h:446-455 -> liegroup SE(3), SO(2), R^n)
void integrate(const Model & m, const double * q, const double * dvv, double * qout)
{
    for (int j = 1; j < m.njoints; ++j)
    {
        const int t = m.jtype[j];
        const double * qj = q + m.idx_q[j];
        const double * d = dvv + m.idx_v[j];
        double * o = qout + m.idx_q[j];
        if (t == JM_JT_FREEFLYER)
        {
            SE3 M0;
            M0.R = quat_to_matrix(qj[3], qj[4], qj[5], qj[6]);
            M0.p = {qj[0], qj[1], qj[2]};
            const SE3 M1 = M0 * exp6(motion6(d));
            double quat[4];
            matrix_to_quat(M1.R, quat);
            const double dp = quat[0] * qj[3] + quat[1] * qj[4] + quat[2] * qj[5] + quat[3] * qj[6];
            if (dp < 0) for (int k = 0; k < 4; ++k) quat[k] = -quat[k];
            const double N2 = quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2] + quat[3] * quat[3];
            const double alpha = (3.0 - N2) / 2.0;  // firstOrderNormalize
            o[0] = M1.p.x; o[1] = M1.p.y; o[2] = M1.p.z;
            for (int k = 0; k < 4; ++k) o[3 + k] = quat[k] * alpha;
        }
        else if (t == JM_JT_SPHERICAL)
        {
            // SpecialOrthogonalOperationTpl<3>::integrate_impl (pinocchio v2.7.0 liegroup/special-orthogonal.hpp):
            // quat_out = quat * exp3(omega) as quaternions, then firstOrderNormalize
            double w4[4];
            quat_exp3(d, w4);
            double r[4];
            quat_mul(qj, w4, r);
            const double N2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
            const double alpha = (3.0 - N2) / 2.0;
            for (int k = 0; k < 4; ++k) o[k] = r[k] * alpha;
        }
        else if (is_unbounded(t))
        {
            const double ca = qj[0], sa = qj[1];
            const double cw = std::cos(d[0]), sw = std::sin(d[0]);
            const double c = cw * ca - sw * sa, s = sw * ca + cw * sa;
            const double n2 = c * c + s * s;
            const double k = (3.0 - n2) / 2.0;
            o[0] = c * k; o[1] = s * k;
        }
        else
            o[0] = qj[0] + d[0];
    }
}

// computeExtraTerms (engine.cc:800-905); uses liMi / data.v / oMi of the last dynamics call.
void extra_terms(Engine & e)
{
    const Model & m = e.mdl;
    const V3 g = {e.opt.gravity[0], e.opt.gravity[1], e.opt.gravity[2]};
    const V3 gw = {e.opt.gravity[3], e.opt.gravity[4], e.opt.gravity[5]};
    // energies (pinocchio_overload::computeKineticEnergy :38-57, computePotentialEnergy)
    double kin = 0;
    for (int j = 1; j < m.njoints; ++j) kin += vtiv(m.inertia[j], e.dv[j]);
    kin *= 0.5;
    double rot = 0;
    for (int i = 0; i < m.nv; ++i) rot += m.rotor[i] * e.v[i] * e.v[i];
    kin += 0.5 * rot;
    double pot = 0;
    for (int j = 1; j < m.njoints; ++j)
    {
        const V3 cg = e.oMi[j].p + e.oMi[j].R * m.inertia[j].c;
        pot -= m.inertia[j].mass * dot(cg, g);
    }
    e.kinetic = kin;
    e.potential = pot;
    // subtree inertias
    for (int j = 1; j < m.njoints; ++j) e.Ycrb[j] = m.inertia[j];
    for (int j = m.njoints - 1; j > 0; --j)
    {
        const int p = m.parent[j];
        if (p > 0) e.Ycrb[p] = add(e.Ycrb[p], act(e.liMi[j], e.Ycrb[j]));
    }
    // accelerations, momenta, forces
    e.dh[0] = Force(); e.fBody[0] = Force(); e.df[0] = Force();
    e.da[0] = Motion();
    e.da_gf[0] = {-g, -gw};
    for (int j = 1; j < m.njoints; ++j)
    {
        const int p = m.parent[j];
        const Motion vj = S_times(m, j, &e.v[m.idx_v[j]]);
        Motion aj = crossm(e.dv[j], vj);  // c + v x S qd (ForwardKinematicsAccelerationStep)
        aj = aj + S_times(m, j, &e.a[m.idx_v[j]]);
        e.da_gf[j] = aj;
        e.da[j] = aj + actInv(e.liMi[j], e.da[p]);
        e.da_gf[j] = e.da_gf[j] + actInv(e.liMi[j], e.da_gf[p]);
        e.dh[j] = mul(m.inertia[j], e.dv[j]);
        e.fBody[j] = mul(m.inertia[j], e.da[j]);
        e.df[j] = crossf(e.dv[j], e.dh[j]);
        e.fBody[j] = e.fBody[j] + e.df[j];
        e.df[j] = e.df[j] + mul(m.inertia[j], e.da_gf[j]);
        e.df[j] = e.df[j] - e.fExternal[j];
    }
    for (int j = m.njoints - 1; j > 0; --j)
    {
        const int p = m.parent[j];
        e.fBody[p] = e.fBody[p] + act(e.liMi[j], e.fBody[j]);
        e.dh[p] = e.dh[p] + act(e.liMi[j], e.dh[j]);
        if (p > 0) e.df[p] = e.df[p] + act(e.liMi[j], e.df[j]);
    }
    // centroidal quantities (single root joint assumed by the reference: data.liMi[1])
    if (m.njoints > 1)
    {
        e.com0 = e.liMi[1].R * e.Ycrb[1].c + e.liMi[1].p;
        e.hg = e.dh[0];
        e.hg.ang = e.hg.ang + cross(e.hg.lin, e.com0);
        e.dhg = e.fBody[0];
        e.dhg.ang = e.dhg.ang + cross(e.dhg.lin, e.com0);
    }
}

// Robot::computeSensorMeasurements, noiseless (basic_sensors.cc)
void sensors(Engine & e)
{
    const Model & m = e.mdl;
    const V3 g = {e.opt.gravity[0], e.opt.gravity[1], e.opt.gravity[2]};
    for (size_t i = 0; i < m.imus.size(); ++i)
    {
        const FrameP & fr = m.imus[i];
        const Motion vf = actInv(fr.M, e.dv[fr.joint]);
        Motion af = actInv(fr.M, e.da[fr.joint]);
        af.lin = af.lin + cross(vf.ang, vf.lin);  // classical acceleration
        const M3 Rw = e.oMi[fr.joint].R * fr.M.R;
        const V3 acc = af.lin - tmul(Rw, g);
        double * o = &e.imu[6 * i];
        o[0] = vf.ang.x; o[1] = vf.ang.y; o[2] = vf.ang.z;
        o[3] = acc.x; o[4] = acc.y; o[5] = acc.z;
    }
    for (size_t i = 0; i < m.contact_sensors.size(); ++i)
    {
        const Force & f = e.contactForces[m.contact_sensors[i]];
        e.contact[3 * i] = f.lin.x; e.contact[3 * i + 1] = f.lin.y; e.contact[3 * i + 2] = f.lin.z;
    }
    for (size_t i = 0; i < m.forces.size(); ++i)
    {
        Force s;
        for (const auto & pr : m.force_pairs[i]) s = s + act(pr.second, e.contactForces[pr.first]);
        to6(s, &e.force[6 * i]);
    }
    for (size_t i = 0; i < m.encoders.size(); ++i)
    {
        const EncoderP & en = m.encoders[i];
        const int t = m.jtype[en.joint];
        double pos;
        if (is_unbounded(t)) pos = std::atan2(e.q[m.idx_q[en.joint] + 1], e.q[m.idx_q[en.joint]]);
        else pos = e.q[m.idx_q[en.joint]];
        const double vel = e.v[m.idx_v[en.joint]];
        if (en.joint_side) { e.encoder[2 * i] = pos; e.encoder[2 * i + 1] = vel; }
        else { e.encoder[2 * i] = pos * en.red; e.encoder[2 * i + 1] = vel * en.red; }
    }
    for (size_t i = 0; i < m.effort_sensors.size(); ++i) e.effort[i] = e.uMotor[m.effort_sensors[i]];
}

void check_state_nan(Engine & e)
{
    for (double x : e.q) if (x != x) e.status |= JM_LANE_NAN;
    for (double x : e.v) if (x != x) e.status |= JM_LANE_NAN;
    for (double x : e.a) if (x != x) e.status |= JM_LANE_NAN;
}

// Engine::start with an externally held command: the INIT_ITERATIONS fixed point
// (engine.cc:1399-1467) converges after the first pass since command does not depend on a.
// Engine::start with `contacts.model = "constraint"` (engine.cc:1266-1308, 1380-1467): every
// constraint is enabled first, computeAllTerms applies the hysteresis, then INIT_ITERATIONS = 4
// passes of computeAcceleration -- the first one with `ignoreBounds` (exact unbounded solve) and
// with RobotState::u still zero, the next ones warm-started PGS solves with
// u = uInternal (incl. the bound multipliers of the previous pass) + motor efforts.
void start_constraint(Engine & e)
{
    const Model & m = e.mdl;
    e.status = 0;
    e.iter = 0;
    forward_kin(e, e.q.data(), e.v.data());
    reset_constraints(e);
    for (auto & f : e.fExternal) f = Force();
    toggle_bounds(e, e.q.data());
    toggle_contacts(e);
    std::fill(e.u.begin(), e.u.end(), 0.0);
    for (int it = 0; it < 4; ++it)
    {
        for (auto & f : e.fExternal) f = Force();
        if (e.applied_k > 0) add_applied_wrenches(e);
        // (uInternalConst of engine.cc:1386-1396: what computeAllTerms left, i.e. the flexibility efforts)
        std::fill(e.uInternal.begin(), e.uInternal.end(), 0.0);
        add_flexibility_efforts(e, e.q.data(), e.v.data(), e.uInternal);
        compute_acceleration(e, e.q.data(), e.v.data(), e.u, it == 0);
        for (int i = 0; i < m.nv; ++i)
        {
            e.a[i] = e.ddq[i];
            if (e.ddq[i] != e.ddq[i]) e.status |= JM_LANE_NAN;
        }
        extra_terms(e);
        sensors(e);
        motor_efforts(e, e.v.data());
        for (int i = 0; i < m.nv; ++i) e.u[i] = e.uInternal[i];
        for (size_t i = 0; i < m.motors.size(); ++i) e.u[m.motors[i].idx_v] += e.uTransmission[i];
    }
}

void start(Engine & e)
{
    if (e.copt.contact_model == JM_CONTACT_CONSTRAINT)
    {
        start_constraint(e);
        return;
    }
    const Model & m = e.mdl;
    e.status = 0;
    e.iter = 0;
    forward_kin(e, e.q.data(), e.v.data());
    double forceMax = 0;
    for (const FrameP & fr : m.contacts) forceMax = std::max(forceMax, norm(contact_at_frame(e, fr).lin));
    if (forceMax > 1e5) e.status |= JM_LANE_FORCE_OVERFLOW;
    dynamics(e, e.q.data(), e.v.data(), e.a.data());
    extra_terms(e);
    sensors(e);
}

// One fixed step (AbstractStepper::tryStep + success bookkeeping engine.cc:2132-2187)
namespace rk4   // runge_kutta4_stepper.h:12-23
{
const double A[4][4] = {{0, 0, 0, 0}, {0.5, 0, 0, 0}, {0, 0.5, 0, 0}, {0, 0, 1.0, 0}};
const double c[4] = {0.0, 0.5, 0.5, 1.0};
const double b[4] = {1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0};
}
void try_step(Engine & e, int solver, double dt)
{
    const Model & m = e.mdl;
    const int nq = m.nq, nv = m.nv;
    if ((int)e.st_incv.size() != nv || (int)e.st_qs.size() != nq)
    {
        for (int i = 0; i < 4; ++i) { e.st_kv[i].assign(nv, 0.0); e.st_ka[i].assign(nv, 0.0); }
        e.st_incv.assign(nv, 0.0); e.st_inca.assign(nv, 0.0); e.st_vs.assign(nv, 0.0); e.st_as.assign(nv, 0.0);
        e.st_qs.assign(nq, 0.0);
    }
    std::vector<double> & incv = e.st_incv, & inca = e.st_inca, & qs = e.st_qs, & vs = e.st_vs, & as = e.st_as;
    if (solver == JM_SOLVER_EULER_EXPLICIT)
    {
        // state.sumInPlace(stateDerivative, dt): q = integrate(q, dt*v); v = v + dt*a
        for (int i = 0; i < nv; ++i) incv[i] = dt * e.v[i];
        integrate(m, e.q.data(), incv.data(), qs.data());
        for (int i = 0; i < nv; ++i) e.v[i] = e.v[i] + dt * e.a[i];
        std::copy(qs.begin(), qs.end(), e.q.begin());
        dynamics(e, e.q.data(), e.v.data(), e.a.data());
    }
    else
    {
        using rk4::A;
        using rk4::b;
        std::vector<double> * kv = e.st_kv, * ka = e.st_ka;
        std::copy(e.v.begin(), e.v.end(), kv[0].begin());
        std::copy(e.a.begin(), e.a.end(), ka[0].begin());
        for (int i = 1; i < 4; ++i)
        {
            std::fill(incv.begin(), incv.end(), 0.0);
            std::fill(inca.begin(), inca.end(), 0.0);
            for (int j = 0; j < i; ++j)
            {
                const double s = dt * A[i][j];
                for (int k = 0; k < nv; ++k) { incv[k] += s * kv[j][k]; inca[k] += s * ka[j][k]; }
            }
            integrate(m, e.q.data(), incv.data(), qs.data());
            for (int k = 0; k < nv; ++k) vs[k] = e.v[k] + inca[k];
            dynamics(e, qs.data(), vs.data(), as.data());
            std::copy(vs.begin(), vs.end(), kv[i].begin());
            std::copy(as.begin(), as.end(), ka[i].begin());
        }
        std::fill(incv.begin(), incv.end(), 0.0);
        std::fill(inca.begin(), inca.end(), 0.0);
        for (int i = 0; i < 4; ++i)
        {
            const double s = dt * b[i];
            for (int k = 0; k < nv; ++k) { incv[k] += s * kv[i][k]; inca[k] += s * ka[i][k]; }
        }
        integrate(m, e.q.data(), incv.data(), qs.data());
        for (int k = 0; k < nv; ++k) e.v[k] = e.v[k] + inca[k];
        std::copy(qs.begin(), qs.end(), e.q.begin());
        dynamics(e, e.q.data(), e.v.data(), e.a.data());  // not FSAL
    }
    extra_terms(e);
    ++e.iter;
}

void step(Engine & e, int solver, double dt, int n_sub, int command_changed, int update_sensors)
{
    check_state_nan(e);
    if (command_changed) dynamics(e, e.q.data(), e.v.data(), e.a.data());  // a(t+), engine.cc:2030-2042
    for (int s = 0; s < n_sub; ++s) try_step(e, solver, dt);
    if (update_sensors) sensors(e);
}

// ------------------------------------------------------------------ adaptive Dormand-Prince stepper
// Reference: core/include/jiminy/core/stepper/runge_kutta_dopri_stepper.h:12-58 (tableau, constants),
// core/src/stepper/runge_kutta_dopri_stepper.cc:18-87 (adjustStep / computeError),
// core/src/stepper/abstract_runge_kutta_stepper.cc:24-77 (tryStepImpl, FSAL),
// core/src/stepper/abstract_stepper.cc:15-62 (tryStep, NaN check -> IS_ERROR),
// core/src/engine/engine.cc:2021-2222 (step-size selection between two breakpoints),
// State::difference = pinocchio::difference (lie_group.h:463-471; Pinocchio v2.7.0 explog.hpp log3 / log6).
namespace dopri
{
const double A[7][7] = {
    {0, 0, 0, 0, 0, 0, 0},
    {1.0 / 5.0, 0, 0, 0, 0, 0, 0},
    {3.0 / 40.0, 9.0 / 40.0, 0, 0, 0, 0, 0},
    {44.0 / 45.0, -56.0 / 15.0, 32.0 / 9.0, 0, 0, 0, 0},
    {19372.0 / 6561.0, -25360.0 / 2187.0, 64448.0 / 6561.0, -212.0 / 729.0, 0, 0, 0},
    {9017.0 / 3168.0, -355.0 / 33.0, 46732.0 / 5247.0, 49.0 / 176.0, -5103.0 / 18656.0, 0, 0},
    {35.0 / 384.0, 0.0, 500.0 / 1113.0, 125.0 / 192.0, -2187.0 / 6784.0, 11.0 / 84.0, 0}};
const double b[7] = {35.0 / 384.0, 0.0, 500.0 / 1113.0, 125.0 / 192.0, -2187.0 / 6784.0, 11.0 / 84.0, 0.0};
const double c[7] = {0.0, 2.0 / 10.0, 3.0 / 10.0, 4.0 / 5.0, 8.0 / 9.0, 1.0, 1.0};
const double e[7] = {5179.0 / 57600.0, 0.0, 7571.0 / 16695.0, 393.0 / 640.0, -92097.0 / 339200.0, 187.0 / 2100.0, 1.0 / 40.0};
constexpr double STEPPER_ORDER = 5.0, SAFETY = 0.8, ERROR_THRESHOLD = 0.5, MIN_FACTOR = 0.2, MAX_FACTOR = 5.0;
}
constexpr double STEPPER_MIN_TIMESTEP = 1e-10, SIMULATION_MIN_TIMESTEP = 1e-6;

// Pinocchio v2.7.0 log3 (explog.hpp)
V3 log3(const M3 & R)
{
    const double PI_value = 3.14159265358979323846;
    double tr = R.m[0][0] + R.m[1][1] + R.m[2][2];
    double theta;
    if (tr >= 3.0) { tr = 3.0; theta = 0.0; }
    else if (tr <= -1.0) { tr = -1.0; theta = PI_value; }
    else theta = std::acos((tr - 1.0) / 2.0);
    V3 res;
    if (theta >= PI_value - 1e-2)
    {
        const double cphi = -(tr - 1.0) / 2.0;
        const double beta = theta * theta / (1.0 + cphi);
        const double t0 = (R.m[0][0] + cphi) * beta, t1 = (R.m[1][1] + cphi) * beta, t2 = (R.m[2][2] + cphi) * beta;
        res.x = (R.m[2][1] > R.m[1][2] ? 1.0 : -1.0) * (t0 > 0 ? std::sqrt(t0) : 0.0);
        res.y = (R.m[0][2] > R.m[2][0] ? 1.0 : -1.0) * (t1 > 0 ? std::sqrt(t1) : 0.0);
        res.z = (R.m[1][0] > R.m[0][1] ? 1.0 : -1.0) * (t2 > 0 ? std::sqrt(t2) : 0.0);
    }
    else
    {
        const double prec3 = std::pow(EPS, 1.0 / 4.0);  // TaylorSeriesExpansion<double>::precision<3>()
        const double t = ((theta > prec3) ? theta / std::sin(theta) : 1.0) / 2.0;
        res.x = t * (R.m[2][1] - R.m[1][2]);
        res.y = t * (R.m[0][2] - R.m[2][0]);
        res.z = t * (R.m[1][0] - R.m[0][1]);
    }
    return res;
}
// Pinocchio v2.7.0 log6 (explog.hpp): [linear; angular]
Motion log6(const SE3 & M)
{
    const V3 w = log3(M.R);
    const double t2 = dot(w, w);
    const double t = std::sqrt(t2);
    double alpha, beta;
    const double prec3 = std::pow(EPS, 1.0 / 4.0);
    if (t < prec3)
    {
        alpha = 1.0 - t2 / 12.0 - t2 * t2 / 720.0;
        beta = 1.0 / 12.0 + t2 / 720.0;
    }
    else
    {
        const double st = std::sin(t), ct = std::cos(t);
        alpha = t * st / (2.0 * (1.0 - ct));
        beta = 1.0 / t2 - st / (2.0 * t * (1.0 - ct));
    }
    const V3 lin = alpha * M.p - 0.5 * cross(w, M.p) + (beta * dot(w, M.p)) * w;
    return {lin, w};
}
// pinocchio::difference(model, q0, q1): tangent vector d such that q0 (+) d = q1
void difference(const Model & m, const double * q0, const double * q1, double * out)
{
    for (int j = 1; j < m.njoints; ++j)
    {
        const int t = m.jtype[j];
        const double * a = q0 + m.idx_q[j];
        const double * bq = q1 + m.idx_q[j];
        double * o = out + m.idx_v[j];
        if (t == JM_JT_FREEFLYER)
        {
            SE3 M0, M1;
            M0.R = quat_to_matrix(a[3], a[4], a[5], a[6]); M0.p = {a[0], a[1], a[2]};
            M1.R = quat_to_matrix(bq[3], bq[4], bq[5], bq[6]); M1.p = {bq[0], bq[1], bq[2]};
            SE3 rel;   // M0.actInv(M1)
            rel.R = transpose(M0.R) * M1.R;
            rel.p = tmul(M0.R, M1.p - M0.p);
            to6(log6(rel), o);
        }
        else if (t == JM_JT_SPHERICAL)
        {
            // SpecialOrthogonalOperationTpl<3>::difference_impl: log3(R0^T R1)
            const M3 R0 = quat_to_matrix(a[0], a[1], a[2], a[3]), R1 = quat_to_matrix(bq[0], bq[1], bq[2], bq[3]);
            const V3 d3 = log3(transpose(R0) * R1);
            o[0] = d3.x; o[1] = d3.y; o[2] = d3.z;
        }
        else if (is_unbounded(t))
        {
            // SO(2): R = R0^T R1 from (cos, sin) pairs, log = signed angle (liegroup/special-orthogonal.hpp)
            const double c = a[0] * bq[0] + a[1] * bq[1], sn = a[0] * bq[1] - a[1] * bq[0];
            const double tr = 2.0 * c;
            const double PI_value = 3.14159265358979323846;
            double theta;
            if (tr > 2.0) theta = 0.0;
            else if (tr < -2.0) theta = (sn >= 0.0) ? PI_value : -PI_value;
            else if (tr > 2.0 - 1e-2) theta = std::asin((sn - (-sn)) / 2.0);
            else theta = (sn >= 0.0) ? std::acos(tr / 2.0) : -std::acos(tr / 2.0);
            o[0] = theta;
        }
        else
            o[0] = bq[0] - a[0];
    }
}

struct AdaptiveState   // per robot: StepperState (engine.h:216-250) + the per-step failure counters
{
    double t = 0.0, dt = SIMULATION_MIN_TIMESTEP, dtLargest = SIMULATION_MIN_TIMESTEP, dtLargestPrev = SIMULATION_MIN_TIMESTEP;
    long iter = 0, iterFailed = 0;
    int successiveIterTooLarge = 0, successiveIterFailed = 0;
};
struct AdaptiveOptions
{
    double tolRel = 1e-4, tolAbs = 1e-5, dtMax = 0.02, dtRestoreThresholdRel = 0.2;
    int successiveIterFailedMax = 1000;
};

// RungeKuttaDOPRIStepper::adjustStep after the error estimate (runge_kutta_dopri_stepper.cc:24-56): true = accepted
bool dopri_adjust(double error, double & dt)
{
    if (error < 1.0)
    {
        if (error < std::min(dopri::ERROR_THRESHOLD, std::pow(dopri::SAFETY, dopri::STEPPER_ORDER)))
        {
            const double clipped = std::max(error, std::pow(dopri::MAX_FACTOR / dopri::SAFETY, -dopri::STEPPER_ORDER));
            dt *= dopri::SAFETY * std::pow(clipped, -1.0 / dopri::STEPPER_ORDER);
        }
        return true;
    }
    dt *= std::max(dopri::SAFETY * std::pow(error, -1.0 / (dopri::STEPPER_ORDER - 2.0)), dopri::MIN_FACTOR);
    return false;
}

// RungeKuttaDOPRIStepper::tryStep: returns 0 = success, 1 = failure (error too large), 2 = error (NaN)
int dopri_try_step(Engine & e, const AdaptiveOptions & ao, double & t, double & dt)
{
    const Model & m = e.mdl;
    const int nq = m.nq, nv = m.nv;
    std::vector<double> kv[7], ka[7];
    kv[0] = e.v; ka[0] = e.a;
    std::vector<double> incv(nv), inca(nv), qs(nq), vs(nv), as(nv);
    for (int i = 1; i < 7; ++i)
    {
        std::fill(incv.begin(), incv.end(), 0.0);
        std::fill(inca.begin(), inca.end(), 0.0);
        for (int j = 0; j < i; ++j)
        {
            const double s = dt * dopri::A[i][j];
            for (int k = 0; k < nv; ++k) { incv[k] += s * kv[j][k]; inca[k] += s * ka[j][k]; }
        }
        integrate(m, e.q.data(), incv.data(), qs.data());
        for (int k = 0; k < nv; ++k) vs[k] = e.v[k] + inca[k];
        dynamics(e, qs.data(), vs.data(), as.data());
        kv[i] = vs; ka[i] = as;
    }
    // candidate solution
    std::vector<double> qsol(nq), vsol(nv);
    std::fill(incv.begin(), incv.end(), 0.0);
    std::fill(inca.begin(), inca.end(), 0.0);
    for (int i = 0; i < 7; ++i)
    {
        const double s = dt * dopri::b[i];
        for (int k = 0; k < nv; ++k) { incv[k] += s * kv[i][k]; inca[k] += s * ka[i][k]; }
    }
    integrate(m, e.q.data(), incv.data(), qsol.data());
    for (int k = 0; k < nv; ++k) vsol[k] = e.v[k] + inca[k];
    // computeError: scale = tolAbs + tolRel |x0 (-) 0|
    std::vector<double> qzero(nq, 0.0), scale_q(nv), scale_v(nv);
    difference(m, e.q.data(), qzero.data(), scale_q.data());
    for (int k = 0; k < nv; ++k)
    {
        scale_q[k] = std::fabs(scale_q[k]) * ao.tolRel + ao.tolAbs;
        scale_v[k] = std::fabs(0.0 - e.v[k]) * ao.tolRel + ao.tolAbs;
    }
    std::vector<double> qoth(nq), voth(nv), errq(nv);
    std::fill(incv.begin(), incv.end(), 0.0);
    std::fill(inca.begin(), inca.end(), 0.0);
    for (int i = 0; i < 7; ++i)
    {
        const double s = dt * dopri::e[i];
        for (int k = 0; k < nv; ++k) { incv[k] += s * kv[i][k]; inca[k] += s * ka[i][k]; }
    }
    integrate(m, e.q.data(), incv.data(), qoth.data());
    for (int k = 0; k < nv; ++k) voth[k] = e.v[k] + inca[k];
    difference(m, qsol.data(), qoth.data(), errq.data());
    double error = 0.0;
    bool nan = false;
    for (int k = 0; k < nv; ++k)
    {
        const double eq = std::fabs(errq[k] / scale_q[k]), ev = std::fabs((voth[k] - vsol[k]) / scale_v[k]);
        nan |= (eq != eq) || (ev != ev);
        error = std::max(error, std::max(eq, ev));
    }
    if (nan) return 2;   // "The estimated integration error contains 'nan'."
    // adjustStep (boost odeint controlled stepper rule)
    const double dt_done = dt;
    if (dopri_adjust(error, dt))
    {
        // abstract_stepper.cc:41-48: NaN in the new derivative -> IS_ERROR, state not committed
        for (double x : ka[6]) if (x != x) return 2;
        // success: state <- solution, derivative <- k_last (FSAL)
        e.q = qsol; e.v = vsol; e.a = ka[6];
        t += dt_done;
        return 0;
    }
    return 1;
}

// Size of the next try of Engine::step's inner loop (engine.cc:2063-2089): stretched onto the breakpoint when the rest
// after it would be below clamp(0.1 dt, 1e-10, 1e-6) (not after a try that failed for being too long), then cut to whole
// microseconds.  Pinned to the reference's compiled lines by tests/golden/ref_cpp_leaves.npz (substep_*).
void substep_rule(double & dt, double t, double tNext, uint32_t successiveIterTooLarge)
{
    double dtResidualThr = STEPPER_MIN_TIMESTEP;
    if (successiveIterTooLarge == 0)
        dtResidualThr = std::min(std::max(0.1 * dt, STEPPER_MIN_TIMESTEP), SIMULATION_MIN_TIMESTEP);
    if (tNext - t < dt || (successiveIterTooLarge <= 1 && tNext - t < dt + dtResidualThr)) dt = tNext - t;
    if (dt > SIMULATION_MIN_TIMESTEP)
    {
        const double dtResidual = std::fmod(dt, SIMULATION_MIN_TIMESTEP);
        if (dtResidual > STEPPER_MIN_TIMESTEP && dtResidual < SIMULATION_MIN_TIMESTEP - STEPPER_MIN_TIMESTEP &&
            dt - dtResidual > STEPPER_MIN_TIMESTEP)
            dt -= dtResidual;
    }
}

// Bookkeeping of Engine::step's inner loop after a try (engine.cc:2138-2221): counters, the restoration of the step size after a
// breakpoint cut it (only if the new estimate is above the size just taken and well below the estimate before the
// breakpoint), the recovery from an evaluation error, the size of the next try.  `rc`: 0 success, 1 rejected, 2 error.
// Pinned to the reference's compiled lines by tests/golden/ref_cpp_leaves.npz (after_try_*).
void after_try(int rc, bool isBreakpointReached, double & dt, double & dtLargest, AdaptiveState & S, const AdaptiveOptions & ao)
{
    if (rc == 0)
    {
        S.successiveIterTooLarge = 0;
        S.successiveIterFailed = 0;
        ++S.iter;
        if (isBreakpointReached)
        {
            const double thr = S.dtLargestPrev * ao.dtRestoreThresholdRel;
            if (dt < dtLargest && dtLargest < thr) dtLargest = S.dtLargestPrev;
        }
        S.dtLargestPrev = dtLargest;
    }
    else
    {
        if (rc == 2) dtLargest *= 0.1;
        if (rc == 1) ++S.successiveIterTooLarge;
        ++S.successiveIterFailed;
        ++S.iterFailed;
    }
    dt = std::min(dtLargest, ao.dtMax);
}

// One breakpoint interval [t, tNext] of Engine::step with the adaptive stepper (engine.cc:2021-2222).
void step_dopri(Engine & e, AdaptiveState & S, const AdaptiveOptions & ao, double tNext, int command_changed,
                int update_sensors)
{
    check_state_nan(e);
    bool hasDynamicsChanged = command_changed != 0;
    double & t = S.t; double & dt = S.dt; double & dtLargest = S.dtLargest;
    bool isBreakpointReached = false;
    while (tNext - t > STEPPER_MIN_TIMESTEP)
    {
        if (hasDynamicsChanged)
        {
            dynamics(e, e.q.data(), e.v.data(), e.a.data());   // FSAL fix: a(t+)
            hasDynamicsChanged = false;
        }
        if (dt < STEPPER_MIN_TIMESTEP) { e.status |= JM_LANE_STEPPER_FAILURE; break; }
        substep_rule(dt, t, tNext, static_cast<uint32_t>(S.successiveIterTooLarge));
        if (S.successiveIterFailed > ao.successiveIterFailedMax) { e.status |= JM_LANE_STEPPER_FAILURE; break; }
        isBreakpointReached = (dtLargest > dt);
        dtLargest = dt;
        const int rc = dopri_try_step(e, ao, t, dtLargest);
        if (rc == 0)
        {
            extra_terms(e);
            ++e.iter;
        }
        after_try(rc, isBreakpointReached, dt, dtLargest, S, ao);
    }
    if (update_sensors) sensors(e);
}

Engine * make_engine(const jm_model_desc * d, const jm_options * o)
{
    Engine * e = new Engine();
    Model & m = e->mdl;
    m.njoints = d->njoints; m.nq = d->nq; m.nv = d->nv;
    m.parent.assign(d->parents, d->parents + d->njoints);
    m.jtype.assign(d->jtypes, d->jtypes + d->njoints);
    if (d->flex_stiffness && d->flex_damping)
    {
        m.flex_k.assign(d->flex_stiffness, d->flex_stiffness + 3 * d->njoints);
        m.flex_d.assign(d->flex_damping, d->flex_damping + 3 * d->njoints);
    }
    m.idx_q.assign(d->idx_q, d->idx_q + d->njoints);
    m.idx_v.assign(d->idx_v, d->idx_v + d->njoints);
    m.axis.resize(d->njoints); m.placement.resize(d->njoints); m.inertia.resize(d->njoints);
    for (int j = 0; j < d->njoints; ++j)
    {
        m.axis[j] = v3_from(d->axes + 3 * j);
        m.placement[j].R = m3_from(d->placement_R + 9 * j);
        m.placement[j].p = v3_from(d->placement_p + 3 * j);
        m.inertia[j].mass = d->mass[j];
        m.inertia[j].c = v3_from(d->com + 3 * j);
        m.inertia[j].I = m3_from(d->inertia + 9 * j);
    }
    m.rotor.assign(d->rotor_inertia, d->rotor_inertia + d->nv);
    m.qlo.assign(d->position_lower, d->position_lower + d->nq);
    m.qhi.assign(d->position_upper, d->position_upper + d->nq);
    for (int i = 0; i < d->nmotors; ++i)
    {
        const double * p = d->motor_params + JM_MOTOR_NPARAMS * i;
        m.motors.push_back({d->motor_joint[i], m.idx_v[d->motor_joint[i]], d->motor_flags[i],
                            p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]});
    }
    auto frames = [](int n, const int32_t * j, const double * R, const double * p) {
        std::vector<FrameP> out;
        for (int i = 0; i < n; ++i)
        {
            FrameP f;
            f.joint = j[i];
            f.M.R = m3_from(R + 9 * i);
            f.M.p = v3_from(p + 3 * i);
            out.push_back(f);
        }
        return out;
    };
    m.contacts = frames(d->ncontacts, d->contact_joint, d->contact_R, d->contact_p);
    m.imus = frames(d->nimu, d->imu_joint, d->imu_R, d->imu_p);
    m.forces = frames(d->nforce, d->force_joint, d->force_R, d->force_p);
    m.cframes = frames(d->n_constraint_frames, d->cframe_joint, d->cframe_R, d->cframe_p);
    m.cframe_mask.assign(d->cframe_mask, d->cframe_mask + d->n_constraint_frames);
    m.cframe_kind.assign(d->cframe_kind, d->cframe_kind + d->n_constraint_frames);
    m.cframe_joint2.assign(d->cframe_joint2, d->cframe_joint2 + d->n_constraint_frames);
    m.cframe_params.assign(d->cframe_params, d->cframe_params + 8 * d->n_constraint_frames);
    m.cjoints.assign(d->cjoint_joint, d->cjoint_joint + d->n_constraint_joints);
    m.contact_sensors.assign(d->contact_sensor_contact, d->contact_sensor_contact + d->ncontact_sensors);
    m.effort_sensors.assign(d->effort_motor, d->effort_motor + d->neffort);
    for (int i = 0; i < d->nencoder; ++i)
        m.encoders.push_back({d->encoder_joint[i], d->encoder_joint_side[i], d->encoder_reduction[i]});
    m.force_pairs.resize(m.forces.size());
    for (size_t i = 0; i < m.forces.size(); ++i)
        for (size_t c = 0; c < m.contacts.size(); ++c)
            if (m.contacts[c].joint == m.forces[i].joint)
            {
                // frameRef.placement.actInv(contactFrame.placement) = F^-1 * C
                SE3 Finv;
                Finv.R = transpose(m.forces[i].M.R);
                Finv.p = -(tmul(m.forces[i].M.R, m.forces[i].M.p));
                m.force_pairs[i].push_back({(int)c, Finv * m.contacts[c].M});
            }
    e->opt = *o;
    const int J = m.njoints;
    e->q.assign(m.nq, 0); e->v.assign(m.nv, 0); e->a.assign(m.nv, 0);
    e->command.assign(m.motors.size(), 0); e->uMotor.assign(m.motors.size(), 0);
    e->uTransmission.assign(m.motors.size(), 0); e->u.assign(m.nv, 0);
    e->fExternal.assign(J, Force()); e->contactFrameForces.assign(m.contacts.size(), Force());
    e->contactForces.assign(m.contacts.size(), Force());
    e->liMi.assign(J, SE3()); e->oMi.assign(J, SE3());
    e->dv.assign(J, Motion()); e->da.assign(J, Motion()); e->da_gf.assign(J, Motion());
    e->df.assign(J, Force()); e->dh.assign(J, Force()); e->fBody.assign(J, Force());
    e->Yaba.resize(J); e->Ycrb.resize(J); e->jd.resize(J);
    e->du.assign(m.nv, 0); e->ddq.assign(m.nv, 0);
    e->imu.assign(6 * m.imus.size(), 0); e->force.assign(6 * m.forces.size(), 0);
    e->contact.assign(3 * m.contact_sensors.size(), 0); e->encoder.assign(2 * m.encoders.size(), 0);
    e->effort.assign(m.effort_sensors.size(), 0);
    return e;
}

void copy_out(const std::vector<Force> & f, double * out)
{
    for (size_t i = 0; i < f.size(); ++i) to6(f[i], out + 6 * i);
}
}  // namespace

// ------------------------------------------------------------------ C entry points (ctypes)
extern "C"
{
void * orc_engine_create(const jm_model_desc * d, const jm_options * o) { return make_engine(d, o); }
void orc_engine_destroy(void * h) { delete static_cast<Engine *>(h); }
void orc_engine_set_options(void * h, const jm_options * o) { static_cast<Engine *>(h)->opt = *o; }
void orc_engine_set_constraint_options(void * h, const jm_constraint_options * o) { static_cast<Engine *>(h)->copt = *o; }
// per-lane constraint state of the batch drivers: flags int32 [NB + NC][B], data [2 NB + 4 NC][B]
// (reference configuration and multiplier of every bounded joint, then 4 multipliers per contact)
void orc_engine_bind_constraints(void * h, int32_t * flags, double * data)
{
    Engine & e = *static_cast<Engine *>(h);
    e.con_flags = flags;
    e.con_data = data;
}
void orc_engine_bind_friction(void * h, const double * friction) { static_cast<Engine *>(h)->lane_friction = friction; }
void orc_engine_bind_flexibility(void * h, const double * flex) { static_cast<Engine *>(h)->lane_flex = flex; }
void orc_engine_bind_ground_offset(void * h, const double * offsets) { static_cast<Engine *>(h)->lane_ground_offset = offsets; }
void orc_engine_bind_model_lane(void * h, const double * model_lane) { static_cast<Engine *>(h)->model_lane = model_lane; }
void orc_engine_bind_ground(void * h, const double * heights, int nx, int ny, double x0, double y0, double dx, double dy)
{
    Engine & e = *static_cast<Engine *>(h);
    e.ground_h = heights; e.ground_nx = nx; e.ground_ny = ny; e.ground_x0 = x0; e.ground_y0 = y0; e.ground_dx = dx; e.ground_dy = dy;
}
void orc_engine_bind_applied(void * h, const double * wrenches, int k, const double * offsets, const int * joints)
{
    Engine & e = *static_cast<Engine *>(h);
    e.applied = wrenches;
    e.applied_k = wrenches ? k : 0;
    for (int i = 0; i < 3 * e.applied_k; ++i) e.applied_p[i] = offsets[i];
    for (int i = 0; i < e.applied_k; ++i) e.applied_joint[i] = joints ? joints[i] : 1;
}
int orc_engine_constraint_counts(void * h, int * n_bounds, int * n_contacts)
{
    Engine & e = *static_cast<Engine *>(h);
    if (e.uInternal.empty()) init_constraints(e);
    *n_bounds = (int)e.bcon.size();
    *n_contacts = (int)e.fcon.size();
    return e.pgsIterLast;
}

void orc_engine_set_state(void * h, const double * q, const double * v, const double * a)
{
    Engine & e = *static_cast<Engine *>(h);
    std::copy(q, q + e.mdl.nq, e.q.begin());
    std::copy(v, v + e.mdl.nv, e.v.begin());
    if (a) std::copy(a, a + e.mdl.nv, e.a.begin()); else std::fill(e.a.begin(), e.a.end(), 0.0);
}
void orc_engine_set_command(void * h, const double * c)
{
    Engine & e = *static_cast<Engine *>(h);
    std::copy(c, c + e.command.size(), e.command.begin());
}
void orc_engine_start(void * h) { start(*static_cast<Engine *>(h)); }
void orc_engine_step(void * h, int solver, double dt, int n_sub, int command_changed, int update_sensors)
{
    step(*static_cast<Engine *>(h), solver, dt, n_sub, command_changed, update_sensors);
}
// compute_robots_dynamics: a = f(q, v) with the held command, does not touch the state
void orc_engine_dynamics(void * h, const double * q, const double * v, double * a_out)
{
    Engine & e = *static_cast<Engine *>(h);
    dynamics(e, q, v, a_out);
}
int orc_engine_status(void * h) { return static_cast<Engine *>(h)->status; }
// field getters: same ids as JM_F_*
int orc_engine_get(void * h, int field, double * out)
{
    Engine & e = *static_cast<Engine *>(h);
    auto cp = [&](const std::vector<double> & s) { std::copy(s.begin(), s.end(), out); return (int)s.size(); };
    switch (field)
    {
    case JM_F_Q: return cp(e.q);
    case JM_F_V: return cp(e.v);
    case JM_F_A: return cp(e.a);
    case JM_F_COMMAND: return cp(e.command);
    case JM_F_U_MOTOR: return cp(e.uMotor);
    case JM_F_U: return cp(e.u);
    case JM_F_F_EXTERNAL: copy_out(e.fExternal, out); return 6 * (int)e.fExternal.size();
    case JM_F_CONTACT_FORCES: copy_out(e.contactForces, out); return 6 * (int)e.contactForces.size();
    case JM_F_IMU: return cp(e.imu);
    case JM_F_FORCE: return cp(e.force);
    case JM_F_CONTACT: return cp(e.contact);
    case JM_F_ENCODER: return cp(e.encoder);
    case JM_F_EFFORT: return cp(e.effort);
    case JM_F_ENERGY: out[0] = e.kinetic; out[1] = e.potential; return 2;
    case JM_F_JOINT_FORCES: copy_out(e.df, out); return 6 * (int)e.df.size();
    case JM_F_CENTROIDAL:
        out[0] = e.com0.x; out[1] = e.com0.y; out[2] = e.com0.z;
        to6(e.hg, out + 3); to6(e.dhg, out + 9);
        return 15;
    default: return -1;
    }
}
// extra debug getters for the tests: world placement of a joint (R row-major 9, p 3)
void orc_engine_joint_placement(void * h, int joint, double * out)
{
    Engine & e = *static_cast<Engine *>(h);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) out[3 * i + j] = e.oMi[joint].R.m[i][j];
    out[9] = e.oMi[joint].p.x; out[10] = e.oMi[joint].p.y; out[11] = e.oMi[joint].p.z;
}
void orc_integrate(void * h, const double * q, const double * dv, double * qout)
{
    integrate(static_cast<Engine *>(h)->mdl, q, dv, qout);
}

// ---- batch drivers over structure-of-arrays buffers X[component][B] (same layout as the GPU).
// Lanes [lane_begin, lane_end) are processed sequentially by the calling thread: this is the
// "one engine, one robot, one thread" shape of the reference, used for parity at batch sizes
// and as the CPU baseline (bench.py cpu_baseline, kind "port").
struct orc_batch_io
{
    int64_t B;
    double *q, *v, *a;            // in/out
    const double * command;       // in
    double *u_motor, *imu, *force, *contact, *encoder, *effort, *energy, *contact_forces, *f_external;  // out, may be null
    int32_t * status;             // out, may be null
    double *joint_forces, *centroidal, *u;  // out, may be null
};
static void load_lane(Engine & e, const orc_batch_io & io, int64_t l)
{
    const int64_t B = io.B;
    if (e.lane_friction) e.opt.contact_friction = e.lane_friction[l];
    if (e.lane_flex)
    {
        // one `flexibilityConfig` per environment (gym_jiminy envs/locomotion.py:288-296)
        Model & m = e.mdl;
        m.flex_k.resize(3 * (size_t)m.njoints, 0.0); m.flex_d.resize(3 * (size_t)m.njoints, 0.0);
        int64_t k = 0;
        for (int j = 1; j < m.njoints; ++j)
        {
            if (m.jtype[j] != JM_JT_SPHERICAL) continue;
            for (int i = 0; i < 3; ++i)
            {
                m.flex_k[3 * j + i] = e.lane_flex[(6 * k + i) * B + l];
                m.flex_d[3 * j + i] = e.lane_flex[(6 * k + 3 + i) * B + l];
            }
            ++k;
        }
    }
    if (e.lane_ground_offset) { e.ground_ox = e.lane_ground_offset[l]; e.ground_oy = e.lane_ground_offset[B + l]; }
    else { e.ground_ox = 0; e.ground_oy = 0; }
    if (e.con_flags && e.con_data)
    {
        if (e.uInternal.empty()) init_constraints(e);
        const int64_t nb = (int64_t)e.bcon.size(), nc = (int64_t)e.fcon.size();
        for (int64_t k = 0; k < nb; ++k)
        {
            const int32_t f = e.con_flags[k * B + l];
            e.bcon[k].enabled = f & 1; e.bcon[k].reversed = (f & 2) != 0; e.bcon[k].locked = (f & 4) != 0;
            e.bcon[k].ref = e.con_data[k * B + l];
            e.bcon[k].lambda = e.con_data[(nb + k) * B + l];
        }
        for (int64_t c = 0; c < nc; ++c)
        {
            e.fcon[c].enabled = e.con_flags[(nb + c) * B + l] & 1;
            for (int k = 0; k < 4; ++k) e.fcon[c].lambda[k] = e.con_data[(2 * nb + 4 * c + k) * B + l];
        }
        // user frame constraints: flag rows after the contacts; 6 multiplier rows each after the contact multipliers; then
        // the reference transforms (translation 3, rotation 9 row-major)
        // (... then the user joint constraints: one flag row, one multiplier row, one reference row each)
        const int64_t nx = (int64_t)e.xcon.size(), nxj = (int64_t)e.jcon.size(), lam0 = 2 * nb + 4 * nc, ref0 = lam0 + 6 * nx + nxj;
        for (int64_t k = 0; k < nxj; ++k)
        {
            e.jcon[k].enabled = e.con_flags[(nb + nc + nx + k) * B + l] & 1;
            e.jcon[k].lambda = e.con_data[(lam0 + 6 * nx + k) * B + l];
            e.jcon[k].ref = e.con_data[(ref0 + 12 * nx + k) * B + l];
        }
        for (int64_t x = 0; x < nx; ++x)
        {
            e.xcon[x].enabled = e.con_flags[(nb + nc + x) * B + l] & 1;
            for (int k = 0; k < 6; ++k) e.xcon[x].lambda[k] = e.con_data[(lam0 + 6 * x + k) * B + l];
            double t[12];
            for (int k = 0; k < 12; ++k) t[k] = e.con_data[(ref0 + 12 * x + k) * B + l];
            e.xcon[x].ref.p = v3_from(t);
            e.xcon[x].ref.R = m3_from(t + 3);
        }
    }
    for (int i = 0; i < e.mdl.nq; ++i) e.q[i] = io.q[i * B + l];
    for (int i = 0; i < e.mdl.nv; ++i) e.v[i] = io.v[i * B + l];
    for (int i = 0; i < e.mdl.nv; ++i) e.a[i] = io.a[i * B + l];
    for (size_t i = 0; i < e.command.size(); ++i) e.command[i] = io.command[i * B + l];
    if (e.model_lane)
        for (int j = 1; j < e.mdl.njoints; ++j)
        {
            const double * r = e.model_lane + (int64_t)(13 * j) * B + l;
            Inertia & Y = e.mdl.inertia[j];
            Y.mass = r[0];
            Y.c = {r[B], r[2 * B], r[3 * B]};
            Y.I.m[0][0] = r[4 * B]; Y.I.m[0][1] = Y.I.m[1][0] = r[5 * B]; Y.I.m[0][2] = Y.I.m[2][0] = r[6 * B];
            Y.I.m[1][1] = r[7 * B]; Y.I.m[1][2] = Y.I.m[2][1] = r[8 * B]; Y.I.m[2][2] = r[9 * B];
            e.mdl.placement[j].p = {r[10 * B], r[11 * B], r[12 * B]};
        }
    for (int k = 0; k < 6 * e.applied_k; ++k) e.applied_now[k] = e.applied ? e.applied[(int64_t)k * B + l] : 0.0;
}
static void store_lane(Engine & e, const orc_batch_io & io, int64_t l)
{
    const int64_t B = io.B;
    if (e.con_flags && e.con_data)
    {
        const int64_t nb = (int64_t)e.bcon.size(), nc = (int64_t)e.fcon.size();
        for (int64_t k = 0; k < nb; ++k)
        {
            e.con_flags[k * B + l] = (e.bcon[k].enabled ? 1 : 0) | (e.bcon[k].reversed ? 2 : 0) | (e.bcon[k].locked ? 4 : 0);
            e.con_data[k * B + l] = e.bcon[k].ref;
            e.con_data[(nb + k) * B + l] = e.bcon[k].lambda;
        }
        for (int64_t c = 0; c < nc; ++c)
        {
            e.con_flags[(nb + c) * B + l] = e.fcon[c].enabled ? 1 : 0;
            for (int k = 0; k < 4; ++k) e.con_data[(2 * nb + 4 * c + k) * B + l] = e.fcon[c].lambda[k];
        }
        const int64_t nx = (int64_t)e.xcon.size(), nxj = (int64_t)e.jcon.size(), lam0 = 2 * nb + 4 * nc, ref0 = lam0 + 6 * nx + nxj;
        for (int64_t k = 0; k < nxj; ++k)
        {
            e.con_flags[(nb + nc + nx + k) * B + l] = e.jcon[k].enabled ? 1 : 0;
            e.con_data[(lam0 + 6 * nx + k) * B + l] = e.jcon[k].lambda;
            e.con_data[(ref0 + 12 * nx + k) * B + l] = e.jcon[k].ref;
        }
        for (int64_t x = 0; x < nx; ++x)
        {
            e.con_flags[(nb + nc + x) * B + l] = e.xcon[x].enabled ? 1 : 0;
            for (int k = 0; k < 6; ++k) e.con_data[(lam0 + 6 * x + k) * B + l] = e.xcon[x].lambda[k];
            const V3 & p = e.xcon[x].ref.p;
            const double t[12] = {p.x, p.y, p.z, e.xcon[x].ref.R.m[0][0], e.xcon[x].ref.R.m[0][1], e.xcon[x].ref.R.m[0][2],
                                  e.xcon[x].ref.R.m[1][0], e.xcon[x].ref.R.m[1][1], e.xcon[x].ref.R.m[1][2],
                                  e.xcon[x].ref.R.m[2][0], e.xcon[x].ref.R.m[2][1], e.xcon[x].ref.R.m[2][2]};
            for (int k = 0; k < 12; ++k) e.con_data[(ref0 + 12 * x + k) * B + l] = t[k];
        }
    }
    for (int i = 0; i < e.mdl.nq; ++i) io.q[i * B + l] = e.q[i];
    for (int i = 0; i < e.mdl.nv; ++i) io.v[i * B + l] = e.v[i];
    for (int i = 0; i < e.mdl.nv; ++i) io.a[i * B + l] = e.a[i];
    auto st = [&](double * dst, const std::vector<double> & s) {
        if (dst) for (size_t i = 0; i < s.size(); ++i) dst[i * B + l] = s[i];
    };
    st(io.u_motor, e.uMotor); st(io.imu, e.imu); st(io.force, e.force); st(io.contact, e.contact);
    st(io.encoder, e.encoder); st(io.effort, e.effort);
    if (io.energy) { io.energy[l] = e.kinetic; io.energy[B + l] = e.potential; }
    if (io.contact_forces)
        for (size_t c = 0; c < e.contactForces.size(); ++c)
        {
            double t[6]; to6(e.contactForces[c], t);
            for (int k = 0; k < 6; ++k) io.contact_forces[(6 * c + k) * B + l] = t[k];
        }
    if (io.f_external)
        for (size_t c = 0; c < e.fExternal.size(); ++c)
        {
            double t[6]; to6(e.fExternal[c], t);
            for (int k = 0; k < 6; ++k) io.f_external[(6 * c + k) * B + l] = t[k];
        }
    if (io.status) io.status[l] = e.status;
    if (io.joint_forces)
        for (size_t c = 0; c < e.df.size(); ++c)
        {
            double t[6]; to6(e.df[c], t);
            for (int k = 0; k < 6; ++k) io.joint_forces[(6 * c + k) * B + l] = t[k];
        }
    if (io.centroidal)
    {
        double t[15] = {e.com0.x, e.com0.y, e.com0.z};
        to6(e.hg, t + 3); to6(e.dhg, t + 9);
        for (int k = 0; k < 15; ++k) io.centroidal[k * B + l] = t[k];
    }
    st(io.u, e.u);
}
// adaptive-step batch driver: per-lane stepper state arrays [B] (t, dt, dtLargest, dtLargestPrev as
// doubles; iter, iterFailed, successiveIterTooLarge, successiveIterFailed as int64)
struct orc_adaptive_io
{
    double *t, *dt, *dt_largest, *dt_largest_prev;
    int64_t *iter, *iter_failed, *succ_too_large, *succ_failed;
};
// mode 0 = start, 1 = step, 2 = dynamics only (a = f(q,v), q/v untouched)
void orc_batch_run(void * h, const orc_batch_io * io, int mode, int solver, double dt, int n_sub,
                   int command_changed, int update_sensors, int64_t lane_begin, int64_t lane_end)
{
    Engine & e = *static_cast<Engine *>(h);
    for (int64_t l = lane_begin; l < lane_end; ++l)
    {
        load_lane(e, *io, l);
        e.status = 0;
        if (mode == 0) start(e);
        else if (mode == 1)
        {
            // the per-robot pinocchio::Data of the reference persists between steps; here each
            // lane is reloaded, so refresh the kinematic data the step relies on (fExternal for
            // extra terms comes from the dynamics evaluations inside the step itself).
            step(e, solver, dt, n_sub, command_changed, update_sensors);
        }
        else
            dynamics(e, e.q.data(), e.v.data(), e.a.data());
        store_lane(e, *io, l);
    }
}

void orc_batch_run_dopri(void * h, const orc_batch_io * io, const orc_adaptive_io * ad, double t_next, double tol_rel,
                         double tol_abs, double dt_max, double dt_restore_threshold_rel, int succ_failed_max,
                         int new_step, int command_changed, int update_sensors, int64_t lane_begin, int64_t lane_end)
{
    Engine & e = *static_cast<Engine *>(h);
    AdaptiveOptions ao;
    ao.tolRel = tol_rel; ao.tolAbs = tol_abs; ao.dtMax = dt_max; ao.dtRestoreThresholdRel = dt_restore_threshold_rel;
    ao.successiveIterFailedMax = succ_failed_max;
    for (int64_t l = lane_begin; l < lane_end; ++l)
    {
        load_lane(e, *io, l);
        e.status = io->status ? (io->status[l] & ~JM_LANE_OUT_OF_BOUNDS) : 0;
        AdaptiveState S;
        S.t = ad->t[l]; S.dt = ad->dt[l]; S.dtLargest = ad->dt_largest[l]; S.dtLargestPrev = ad->dt_largest_prev[l];
        S.iter = ad->iter[l]; S.iterFailed = ad->iter_failed[l];
        S.successiveIterTooLarge = new_step ? 0 : (int)ad->succ_too_large[l];
        S.successiveIterFailed = new_step ? 0 : (int)ad->succ_failed[l];
        e.status = 0;
        step_dopri(e, S, ao, t_next, command_changed, update_sensors);
        ad->t[l] = S.t; ad->dt[l] = S.dt; ad->dt_largest[l] = S.dtLargest; ad->dt_largest_prev[l] = S.dtLargestPrev;
        ad->iter[l] = S.iter; ad->iter_failed[l] = S.iterFailed;
        ad->succ_too_large[l] = S.successiveIterTooLarge; ad->succ_failed[l] = S.successiveIterFailed;
        store_lane(e, *io, l);
    }
}

// ---- leaf entry points: the very functions the engine above calls, one application per case.  They exist so that
// tests/test_reference_cpp_leaves.py can hold them against the outputs of the REFERENCE'S OWN TEXT compiled by
// tools/make_ref_cpp_fixtures.py (tests/golden/ref_cpp_leaves.npz); the layouts are that tool's.
// params[n][12] = stiffness damping friction transitionEps transitionVelocity | n(3) depth v(3)
void orc_leaf_contact_law(int64_t n, const double * params, double * out /* [n][3] */)
{
    for (int64_t i = 0; i < n; ++i)
    {
        const double * c = params + i * 12;
        jm_options o{};
        o.contact_stiffness = c[0]; o.contact_damping = c[1]; o.contact_friction = c[2];
        o.contact_transition_eps = c[3]; o.contact_transition_velocity = c[4];
        const V3 f = contact_law(o, V3{c[5], c[6], c[7]}, c[8], V3{c[9], c[10], c[11]});
        out[i * 3 + 0] = f.x; out[i * 3 + 1] = f.y; out[i * 3 + 2] = f.z;
    }
}
// params[n][14] = red effLimOn velLimOn invSlope effortLimit velocityLimit fricOn fvp fvn fdp fdn fds | v command
void orc_leaf_motor_law(int64_t n, const double * params, double * u_motor, double * u_transmission)
{
    for (int64_t i = 0; i < n; ++i)
    {
        const double * c = params + i * 14;
        MotorP mp{};
        mp.red = c[0];
        mp.flags = (c[1] != 0.0 ? JM_MOTOR_EFFORT_LIMIT : 0) | (c[2] != 0.0 ? JM_MOTOR_VELOCITY_LIMIT : 0) |
                   (c[6] != 0.0 ? JM_MOTOR_FRICTION : 0);
        mp.inv_slope = c[3]; mp.effort_limit = c[4]; mp.velocity_limit = c[5];
        mp.fvp = c[7]; mp.fvn = c[8]; mp.fdp = c[9]; mp.fdn = c[10]; mp.fds = c[11];
        motor_law(mp, c[12], c[13], u_motor[i], u_transmission[i]);
    }
}
// in/out per case: dt, dtLargest, dtLargestPrev; counters [tooLarge, failed, iter, iterFailed]
void orc_leaf_after_try(int64_t n, const int32_t * rc, const int32_t * bp, double dtRestoreThresholdRel, double dtMax,
                        double * dt, double * dtLargest, double * dtLargestPrev, int64_t * counters)
{
    AdaptiveOptions ao{};
    ao.dtRestoreThresholdRel = dtRestoreThresholdRel; ao.dtMax = dtMax;
    for (int64_t i = 0; i < n; ++i)
    {
        AdaptiveState S{};
        S.dtLargestPrev = dtLargestPrev[i];
        S.successiveIterTooLarge = (int)counters[4 * i]; S.successiveIterFailed = (int)counters[4 * i + 1];
        S.iter = counters[4 * i + 2]; S.iterFailed = counters[4 * i + 3];
        after_try(rc[i], bp[i] != 0, dt[i], dtLargest[i], S, ao);
        dtLargestPrev[i] = S.dtLargestPrev;
        counters[4 * i] = S.successiveIterTooLarge; counters[4 * i + 1] = S.successiveIterFailed;
        counters[4 * i + 2] = S.iter; counters[4 * i + 3] = S.iterFailed;
    }
}
void orc_leaf_substep(int64_t n, const double * dt, const double * t, const double * tNext, const int32_t * tooLarge,
                      double * dt_out)
{
    for (int64_t i = 0; i < n; ++i)
    {
        double d = dt[i];
        substep_rule(d, t[i], tNext[i], static_cast<uint32_t>(tooLarge[i]));
        dt_out[i] = d;
    }
}
// code: 1 accepted, 0 rejected, 2 NaN error (the reference throws)
void orc_leaf_dopri_adjust(int64_t n, const double * error, const double * dt, int32_t * code, double * dt_out)
{
    for (int64_t i = 0; i < n; ++i)
    {
        double d = dt[i];
        if (error[i] != error[i]) code[i] = 2;
        else code[i] = dopri_adjust(error[i], d) ? 1 : 0;
        dt_out[i] = d;
    }
}
void orc_leaf_dopri_constants(double * out5)
{
    out5[0] = dopri::STEPPER_ORDER; out5[1] = dopri::SAFETY; out5[2] = dopri::ERROR_THRESHOLD;
    out5[3] = dopri::MIN_FACTOR; out5[4] = dopri::MAX_FACTOR;
}
// row major: rk4 A(16) c(4) b(4), dopri A(49) c(7) b(7) e(7)
void orc_leaf_tableaux(double * out94)
{
    double * o = out94;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) *o++ = rk4::A[i][j];
    for (int i = 0; i < 4; ++i) *o++ = rk4::c[i];
    for (int i = 0; i < 4; ++i) *o++ = rk4::b[i];
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) *o++ = dopri::A[i][j];
    for (int i = 0; i < 7; ++i) *o++ = dopri::c[i];
    for (int i = 0; i < 7; ++i) *o++ = dopri::b[i];
    for (int i = 0; i < 7; ++i) *o++ = dopri::e[i];
}
// types[nc]: ConstraintRegistryType (0 contact frame, 2 joint bound, 3 user); A row major (symmetric); prm = friction,
// torsion, tolAbs, tolRel.  sweep_w >= 0: ONE sweep at that relaxation factor; sweep_w < 0: the solver loop.
// Returns the convergence flag of the loop (1 for a sweep); x is updated in place, y receives the residuals.
int orc_leaf_pgs(int nc, const int32_t * types, const int32_t * dims, int n, const double * A, const double * b,
                 const double * prm, int iter_max, double sweep_w, double * x, double * y, int32_t * iters)
{
    PgsRowSet rs;
    int row = 0;
    for (int c = 0; c < nc; ++c)
    {
        // the block table of the PGSSolver constructor (constraint_solvers.cc:46-90)
        const int nblocks = types[c] == 2 ? 1 : (types[c] == 0 || types[c] == 1 ? 3 : 0);
        rs.cons.push_back({row, dims[c], nblocks});
        row += dims[c];
    }
    Dense Ad(n, n);
    std::copy(A, A + (size_t)n * n, Ad.d.begin());
    const std::vector<double> bv(b, b + n);
    std::vector<double> xv(x, x + n), yv(n, 0.0);
    const PgsOptions po{prm[0], prm[1], prm[2], prm[3], (unsigned)iter_max};
    int ok = 1, it = 1;
    if (sweep_w >= 0.0) pgs_sweep(po, rs, Ad, bv, sweep_w, xv, yv);
    else ok = pgs_solve(po, rs, Ad, bv, xv, it, &yv) ? 1 : 0;
    std::copy(xv.begin(), xv.end(), x);
    std::copy(yv.begin(), yv.end(), y);
    if (iters) *iters = it;
    return ok;
}
}
