// jm_adaptive.h -- adaptive Dormand-Prince stepping (the reference's default `odeSolver`), one
// robot per lane with its OWN step size, built around the unchanged dynamics kernels.
//
// One attempt = k_dopri_prepare (all lanes: choose dt, append the lanes that still have to move to the
// ACTIVE LIST) -> 6 x [k_dopri_stage (stage state on the manifold) -> dynamics launch (MODE_DYNAMICS of
// k_quad / k_batch)] -> k_dopri_finish (embedded error estimate, accept / reject, next step size), the
// last three over the n active lanes only: the stage kernels gather lane map[c] into COMPACT workspace
// rows `[rows][n]` (stage configuration, stage derivatives, a copy of the held command), the dynamics
// kernels run unchanged on that dense batch of n robots, the finish kernel scatters the accepted state
// back.  The host (jm_lib.cpp: jm_batch_step_adaptive) repeats attempts until the list is empty; the
// work of an interval is the sum of the lanes' own attempts, not lanes x attempts of the stiffest one.
//
// Reference restated here:
//   tableau / constants   core/include/jiminy/core/stepper/runge_kutta_dopri_stepper.h:12-58
//   tryStepImpl (FSAL)    core/src/stepper/abstract_runge_kutta_stepper.cc:24-77
//   adjustStep / error    core/src/stepper/runge_kutta_dopri_stepper.cc:18-87
//   tryStep NaN check     core/src/stepper/abstract_stepper.cc:15-62
//   step-size selection   core/src/engine/engine.cc:2021-2222
//   State::sum/difference core/include/jiminy/core/stepper/lie_group.h:446-471 (pinocchio::integrate /
//                         difference; Pinocchio v2.7.0 explog.hpp log3 / log6 restated)
#pragma once
#include "jm_kernels.h"
#include "jm_constraint.h"

namespace jm
{
namespace dopri
{
#ifdef JM_HOST_EMU
#define JM_TABLEAU static const
#else
#define JM_TABLEAU __device__ __constant__ const
#endif
JM_TABLEAU double A[7][7] = {
    {0, 0, 0, 0, 0, 0, 0},
    {1.0 / 5.0, 0, 0, 0, 0, 0, 0},
    {3.0 / 40.0, 9.0 / 40.0, 0, 0, 0, 0, 0},
    {44.0 / 45.0, -56.0 / 15.0, 32.0 / 9.0, 0, 0, 0, 0},
    {19372.0 / 6561.0, -25360.0 / 2187.0, 64448.0 / 6561.0, -212.0 / 729.0, 0, 0, 0},
    {9017.0 / 3168.0, -355.0 / 33.0, 46732.0 / 5247.0, 49.0 / 176.0, -5103.0 / 18656.0, 0, 0},
    {35.0 / 384.0, 0.0, 500.0 / 1113.0, 125.0 / 192.0, -2187.0 / 6784.0, 11.0 / 84.0, 0}};
JM_TABLEAU double E[7] = {5179.0 / 57600.0, 0.0, 7571.0 / 16695.0, 393.0 / 640.0, -92097.0 / 339200.0,
                                             187.0 / 2100.0, 1.0 / 40.0};
constexpr double STEPPER_ORDER = 5.0, SAFETY = 0.8, ERROR_THRESHOLD = 0.5, MIN_FACTOR = 0.2, MAX_FACTOR = 5.0;
}
constexpr double STEPPER_MIN_TIMESTEP = 1e-10, SIMULATION_MIN_TIMESTEP = 1e-6;

// per-lane stepper state (always float64, whatever the state dtype): rows of `[B]`
enum { AD_T = 0, AD_DT = 1, AD_DT_LARGEST = 2, AD_DT_LARGEST_PREV = 3, AD_DT_TRY = 4, AD_NROWS_F = 5 };
enum { AD_ITER = 0, AD_ITER_FAILED = 1, AD_SUCC_TOO_LARGE = 2, AD_SUCC_FAILED = 3, AD_ACTIVE = 4, AD_BP_REACHED = 5, AD_MAP = 6, AD_NROWS_I = 7 };

template<class T> struct AdaptiveArgs
{
    const T * P;
    T * q; T * v; T * a;          // state = x0, k_0 = (v, a); committed on success
    T * ws;                       // workspace: kv[6][nv], ka[6][nv], qs[nq], command[nm]: COMPACT rows of [n_act]
    const T * command;            // held command [nm][B]
    long long n_act;              // active lanes of this attempt (host copy of *n_active)
    // constraint contact model: per-lane constraint state (null = spring-damper model) and its compact copy
    // (flags: library-owned int32 [NF][n]; data: workspace rows CDATA), gathered at stage 1, scattered back
    // by the finish kernel whether the step is accepted or not (the reference's constraint objects are
    // mutated by every evaluation, engine.cc:3253-3338, 3145-3193)
    int32_t * con_flags; T * con_data; int32_t * con_flags_c;
    double * fs;                  // [AD_NROWS_F][B]
    int32_t * is;                 // [AD_NROWS_I][B]
    int32_t * status;
    int32_t * n_active;           // device counter
    long long B;
    double t_next, tol_rel, tol_abs, dt_max, dt_restore_threshold_rel;
    int succ_failed_max, new_step, stage;
};
template<class Tp> struct AdaptiveRows
{
    static constexpr int KV = 0, KA = 6 * Tp::NV, QS = 12 * Tp::NV, CMD = 12 * Tp::NV + Tp::NQ, TOTAL = CMD + Tp::NM;
    // constraint contact model: + compact constraint data and delassus workspace
    static constexpr int CDATA = TOTAL, CWS = CDATA + ConRows<Tp>::ND, TOTAL_CON = CWS + ConRows<Tp>::WTOTAL;
};

// (acos_ / asin_ / fabs_ and log3 live in jm_math.h: the user FrameConstraint rows of jm_constraint.h need log3 too)
// Pinocchio v2.7.0 log6, [linear; angular]
template<class T> JM_DEV Sp<T> log6(const SE3<T> & M)
{
    const V3<T> w = log3(M.R);
    const T t2 = dot(w, w);
    const T t = sqrt_(t2);
    T alpha, beta;
    if (t < Eps<T>::taylor)
    {
        alpha = T(1) - t2 / T(12) - t2 * t2 / T(720);
        beta = T(1) / T(12) + t2 / T(720);
    }
    else
    {
        T st, ct;
        sincos_(t, &st, &ct);
        alpha = t * st / (T(2) * (T(1) - ct));
        beta = T(1) / t2 - st / (T(2) * t * (T(1) - ct));
    }
    return {alpha * M.p - T(0.5) * cross(w, M.p) + (beta * dot(w, M.p)) * w, w};
}
// pinocchio::difference: tangent d with q0 (+) d = q1
template<class T, class Tp> JM_DEV void difference_q(const T * q0, const T * q1, T * out)
{
    static_for<1, Tp::NJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int t = Tp::jtype[j];
        constexpr int iq = Tp::idx_q[j], iv = Tp::idx_v[j];
        if constexpr (t == JM_JT_FREEFLYER)
        {
            const M3<T> R0 = quat_to_matrix(q0[iq + 3], q0[iq + 4], q0[iq + 5], q0[iq + 6]);
            const M3<T> R1 = quat_to_matrix(q1[iq + 3], q1[iq + 4], q1[iq + 5], q1[iq + 6]);
            SE3<T> rel;
            rel.R = transpose(R0) * R1;
            rel.p = tmul(R0, V3<T>{q1[iq] - q0[iq], q1[iq + 1] - q0[iq + 1], q1[iq + 2] - q0[iq + 2]});
            const Sp<T> d = log6(rel);
            out[iv] = d.l.x; out[iv + 1] = d.l.y; out[iv + 2] = d.l.z;
            out[iv + 3] = d.a.x; out[iv + 4] = d.a.y; out[iv + 5] = d.a.z;
        }
        else if constexpr (jt_is_sph(t))
        {
            // SpecialOrthogonalOperationTpl<3>::difference_impl: log3(R0^T R1)
            const M3<T> R0 = quat_to_matrix(q0[iq], q0[iq + 1], q0[iq + 2], q0[iq + 3]);
            const M3<T> R1 = quat_to_matrix(q1[iq], q1[iq + 1], q1[iq + 2], q1[iq + 3]);
            const V3<T> d = log3(transpose(R0) * R1);
            out[iv] = d.x; out[iv + 1] = d.y; out[iv + 2] = d.z;
        }
        else if constexpr (jt_is_unb(t))
        {
            const T c = q0[iq] * q1[iq] + q0[iq + 1] * q1[iq + 1], sn = q0[iq] * q1[iq + 1] - q0[iq + 1] * q1[iq];
            const T tr = T(2) * c;
            const T PI_value = T(3.14159265358979323846);
            T theta;
            if (tr > T(2)) theta = T(0);
            else if (tr < T(-2)) theta = (sn >= T(0)) ? PI_value : -PI_value;
            else if (tr > T(2) - T(1e-2)) theta = asin_((sn - (-sn)) / T(2));
            else theta = (sn >= T(0)) ? acos_(tr / T(2)) : -acos_(tr / T(2));
            out[iv] = theta;
        }
        else
            out[iv] = q1[iq] - q0[iq];
    });
}

#ifndef JM_HOST_EMU
// ---- choose the step size of the next attempt (engine.cc:2021-2131, per lane)
template<class T, class Tp>
__global__ void __launch_bounds__(256) k_dopri_prepare(const AdaptiveArgs<T> A)
{
    const long long lane = (long long)blockIdx.x * 256 + threadIdx.x;
    if (lane >= A.B) return;
    const long long B = A.B;
    double * fs = A.fs + lane;
    int32_t * is = A.is + lane;
    if (A.new_step) { is[AD_SUCC_TOO_LARGE * B] = 0; is[AD_SUCC_FAILED * B] = 0; }
    const double t = fs[AD_T * B];
    double dt = fs[AD_DT * B];
    const double dtLargest = fs[AD_DT_LARGEST * B];
    const int st = A.status ? A.status[lane] : 0;
    int active = (A.t_next - t > STEPPER_MIN_TIMESTEP) && !(st & (JM_LANE_STEPPER_FAILURE | JM_LANE_NAN));
    if (active)
    {
        const int tooLarge = is[AD_SUCC_TOO_LARGE * B], failed = is[AD_SUCC_FAILED * B];
        if (dt < STEPPER_MIN_TIMESTEP || failed > A.succ_failed_max)
        {
            if (A.status) A.status[lane] = st | JM_LANE_STEPPER_FAILURE;
            active = 0;
        }
        else
        {
            double thr = STEPPER_MIN_TIMESTEP;
            if (tooLarge == 0) thr = fmin(fmax(0.1 * dt, STEPPER_MIN_TIMESTEP), SIMULATION_MIN_TIMESTEP);
            if (A.t_next - t < dt || (tooLarge <= 1 && A.t_next - t < dt + thr)) dt = A.t_next - t;
            if (dt > SIMULATION_MIN_TIMESTEP)
            {
                const double res = fmod(dt, SIMULATION_MIN_TIMESTEP);
                if (res > STEPPER_MIN_TIMESTEP && res < SIMULATION_MIN_TIMESTEP - STEPPER_MIN_TIMESTEP && dt - res > STEPPER_MIN_TIMESTEP)
                    dt -= res;
            }
            is[AD_BP_REACHED * B] = dtLargest > dt;
            fs[AD_DT * B] = dt;
        }
    }
    fs[AD_DT_TRY * B] = active ? dt : 0.0;
    is[AD_ACTIVE * B] = active;
    // active list (order irrelevant: lanes are independent and every kernel is lane-position agnostic)
    if (active) A.is[(long long)AD_MAP * B + atomicAdd(A.n_active, 1)] = (int32_t)lane;
}

// ---- stage i (1..6): x_i = x0 (+) dt sum_j A_ij k_j ; k_i.v = v_i is stored, k_i.a comes from the dynamics launch
template<class T, class Tp>
__global__ void __launch_bounds__(128) k_dopri_stage(const AdaptiveArgs<T> A)
{
    using R = AdaptiveRows<Tp>;
    constexpr int NQ = Tp::NQ, NV = Tp::NV;
    const long long c = (long long)blockIdx.x * 128 + threadIdx.x;   // position in the active list
    // `n_act` (host copy, sizes the grid and the compact rows) is an upper bound of the current list
    // length when several attempts are issued per synchronisation: the device counter is authoritative
    if (c >= A.n_act || c >= *A.n_active) return;
    const long long B = A.B, N = A.n_act;
    const long long lane = A.is[(long long)AD_MAP * B + c];
    CPtr<T> P = (CPtr<T>)A.P;
    const int i = A.stage;
    if (i == 1)
        static_for<0, Tp::NM>([&](auto mc) {
            A.ws[(long long)(R::CMD + decltype(mc)::value) * N + c] = A.command[decltype(mc)::value * B + lane];
        });
    if (i == 1 && A.con_flags)
    {
        for (int r = 0; r < ConRows<Tp>::NF; ++r) A.con_flags_c[(long long)r * N + c] = A.con_flags[(long long)r * B + lane];
        for (int r = 0; r < ConRows<Tp>::ND; ++r) A.ws[(long long)(R::CDATA + r) * N + c] = A.con_data[(long long)r * B + lane];
    }
    const T dt = (T)A.fs[AD_DT_TRY * B + lane];
    T q0[NQ], incv[NV], qs[NQ];
    static_for<0, NQ>([&](auto ic) { q0[decltype(ic)::value] = A.q[decltype(ic)::value * B + lane]; });
    // k_0 = (v, a)
    {
        const T s = dt * (T)dopri::A[i][0];
        static_for<0, NV>([&](auto ic) { incv[decltype(ic)::value] = s * A.v[decltype(ic)::value * B + lane]; });
    }
    for (int j = 1; j < i; ++j)
    {
        const T s = dt * (T)dopri::A[i][j];
        const T * kv = A.ws + (long long)(R::KV + (j - 1) * NV) * N + c;
        static_for<0, NV>([&](auto ic) { incv[decltype(ic)::value] += s * kv[decltype(ic)::value * N]; });
    }
    integrate_q<T, Tp>(P, q0, incv, qs);
    static_for<0, NQ>([&](auto ic) { A.ws[(long long)(R::QS + decltype(ic)::value) * N + c] = qs[decltype(ic)::value]; });
    // velocity part (reuses incv as the acceleration increment)
    {
        const T s = dt * (T)dopri::A[i][0];
        static_for<0, NV>([&](auto ic) { incv[decltype(ic)::value] = s * A.a[decltype(ic)::value * B + lane]; });
    }
    for (int j = 1; j < i; ++j)
    {
        const T s = dt * (T)dopri::A[i][j];
        const T * ka = A.ws + (long long)(R::KA + (j - 1) * NV) * N + c;
        static_for<0, NV>([&](auto ic) { incv[decltype(ic)::value] += s * ka[decltype(ic)::value * N]; });
    }
    T * kvi = A.ws + (long long)(R::KV + (i - 1) * NV) * N + c;
    static_for<0, NV>([&](auto ic) { kvi[decltype(ic)::value * N] = A.v[decltype(ic)::value * B + lane] + incv[decltype(ic)::value]; });
}

// ---- error estimate, accept / reject, next step size (runge_kutta_dopri_stepper.cc, engine.cc:2132-2221)
template<class T, class Tp>
__global__ void __launch_bounds__(128) k_dopri_finish(const AdaptiveArgs<T> A)
{
    using R = AdaptiveRows<Tp>;
    constexpr int NQ = Tp::NQ, NV = Tp::NV;
    const long long c = (long long)blockIdx.x * 128 + threadIdx.x;   // position in the active list
    if (c >= A.n_act || c >= *A.n_active) return;
    const long long B = A.B, N = A.n_act;
    const long long lane = A.is[(long long)AD_MAP * B + c];
    double * fs = A.fs + lane;
    int32_t * is = A.is + lane;
    if (!is[AD_ACTIVE * B]) return;
    if (A.con_flags)
    {
        for (int r = 0; r < ConRows<Tp>::NF; ++r) A.con_flags[(long long)r * B + lane] = A.con_flags_c[(long long)r * N + c];
        for (int r = 0; r < ConRows<Tp>::ND; ++r) A.con_data[(long long)r * B + lane] = A.ws[(long long)(R::CDATA + r) * N + c];
    }
    CPtr<T> P = (CPtr<T>)A.P;
    const double dt = fs[AD_DT_TRY * B];
    const T dtT = (T)dt;
    T q0[NQ], qa[NQ], qb[NQ], d[NV], sc[NV];
    static_for<0, NQ>([&](auto ic) { q0[decltype(ic)::value] = A.q[decltype(ic)::value * B + lane]; });
    // scale (configuration part): tolAbs + tolRel |x0 (-) 0|
    static_for<0, NQ>([&](auto ic) { qa[decltype(ic)::value] = T(0); });
    difference_q<T, Tp>(q0, qa, sc);
    static_for<0, NV>([&](auto ic) { sc[decltype(ic)::value] = fabs_(sc[decltype(ic)::value]) * (T)A.tol_rel + (T)A.tol_abs; });
    // alternative (4th order) solution x0 (+) dt sum_j e_j k_j
    {
        const T s = dtT * (T)dopri::E[0];
        static_for<0, NV>([&](auto ic) { d[decltype(ic)::value] = s * A.v[decltype(ic)::value * B + lane]; });
    }
    for (int j = 1; j < 7; ++j)
    {
        const T s = dtT * (T)dopri::E[j];
        const T * kv = A.ws + (long long)(R::KV + (j - 1) * NV) * N + c;
        static_for<0, NV>([&](auto ic) { d[decltype(ic)::value] += s * kv[decltype(ic)::value * N]; });
    }
    integrate_q<T, Tp>(P, q0, d, qb);                       // other solution
    static_for<0, NQ>([&](auto ic) { qa[decltype(ic)::value] = A.ws[(long long)(R::QS + decltype(ic)::value) * N + c]; });  // solution
    difference_q<T, Tp>(qa, qb, d);
    double error = 0.0;
    bool nan = false;
    static_for<0, NV>([&](auto ic) {
        const double e = (double)fabs_(d[decltype(ic)::value] / sc[decltype(ic)::value]);
        nan |= (e != e);
        error = fmax(error, e);
    });
    // velocity part
    const T * kv6 = A.ws + (long long)(R::KV + 5 * NV) * N + c;   // k_6.v = solution velocity
    const T * ka6 = A.ws + (long long)(R::KA + 5 * NV) * N + c;   // k_6.a = f(solution): FSAL
    {
        const T s = dtT * (T)dopri::E[0];
        static_for<0, NV>([&](auto ic) { d[decltype(ic)::value] = s * A.a[decltype(ic)::value * B + lane]; });
    }
    for (int j = 1; j < 7; ++j)
    {
        const T s = dtT * (T)dopri::E[j];
        const T * ka = A.ws + (long long)(R::KA + (j - 1) * NV) * N + c;
        static_for<0, NV>([&](auto ic) { d[decltype(ic)::value] += s * ka[decltype(ic)::value * N]; });
    }
    bool a_nan = false;
    static_for<0, NV>([&](auto ic) {
        constexpr int k = decltype(ic)::value;
        const T v0 = A.v[k * B + lane];
        const T scv = fabs_(T(0) - v0) * (T)A.tol_rel + (T)A.tol_abs;
        const double e = (double)fabs_(((v0 + d[k]) - kv6[k * N]) / scv);
        nan |= (e != e);
        error = fmax(error, e);
        const T an = ka6[k * N];
        a_nan |= (an != an);
    });
    double dtLargest = dt;   // tryStep(..., dtLargest) updates it in place
    int rc;
    if (nan) rc = 2;
    else if (error < 1.0)
    {
        if (error < fmin(dopri::ERROR_THRESHOLD, pow(dopri::SAFETY, dopri::STEPPER_ORDER)))
        {
            const double clipped = fmax(error, pow(dopri::MAX_FACTOR / dopri::SAFETY, -dopri::STEPPER_ORDER));
            dtLargest *= dopri::SAFETY * pow(clipped, -1.0 / dopri::STEPPER_ORDER);
        }
        rc = a_nan ? 2 : 0;
    }
    else
    {
        dtLargest *= fmax(dopri::SAFETY * pow(error, -1.0 / (dopri::STEPPER_ORDER - 2.0)), dopri::MIN_FACTOR);
        rc = 1;
    }
    if (rc == 0)
    {
        static_for<0, NQ>([&](auto ic) { A.q[decltype(ic)::value * B + lane] = qa[decltype(ic)::value]; });
        static_for<0, NV>([&](auto ic) {
            constexpr int k = decltype(ic)::value;
            A.v[k * B + lane] = kv6[k * N];
            A.a[k * B + lane] = ka6[k * N];
        });
        fs[AD_T * B] += dt;
        is[AD_SUCC_TOO_LARGE * B] = 0;
        is[AD_SUCC_FAILED * B] = 0;
        is[AD_ITER * B] += 1;
        if (is[AD_BP_REACHED * B])
        {
            const double prev = fs[AD_DT_LARGEST_PREV * B];
            if (dt < dtLargest && dtLargest < prev * A.dt_restore_threshold_rel) dtLargest = prev;
        }
        fs[AD_DT_LARGEST_PREV * B] = dtLargest;
    }
    else
    {
        if (rc == 2) dtLargest *= 0.1;
        if (rc == 1) is[AD_SUCC_TOO_LARGE * B] += 1;
        is[AD_SUCC_FAILED * B] += 1;
        is[AD_ITER_FAILED * B] += 1;
    }
    fs[AD_DT_LARGEST * B] = dtLargest;
    fs[AD_DT * B] = fmin(dtLargest, A.dt_max);
}
#endif  // JM_HOST_EMU
}  // namespace jm
