// jm_qdopri.h -- adaptive Dormand-Prince stepping as ONE persistent launch per breakpoint interval on the
// branch-parallel decomposition (jm_quad.h: four lanes per robot, lane k owns limb k).
//
// Every robot carries its own step size.  Instead of one launch per stage over compacted lists of active lanes
// (jm_adaptive.h: 14 launches and a host synchronisation per attempt), each quad runs the reference's whole
// adaptive loop on the chip until ITS robot reaches the breakpoint:
//     choose dt -> 6 stage evaluations (the limb-parallel `quad_eval`) -> embedded error estimate ->
//     accept / reject -> next dt,
// with the step-start state, k_0 = (v, a), the stage derivatives of the trunk tree and the candidate solution
// in LDS (the stage buffer of k_quad plus 13 NVB + NQB trunk rows), the stage derivatives of the limbs in the
// caller's adaptive workspace (lane-private rows, 12 N values per lane and attempt) and the controller state
// in registers.  A wave (16 robots) iterates until its slowest robot is through; finished quads idle with their
// lanes masked.  The arithmetic (tableau sums, error norm, step-size law) follows the per-stage kernels of
// jm_adaptive.h operation for operation, so both paths follow the same accept / reject sequences.
//
// Reference restated: see the header of jm_adaptive.h (tableau, tryStepImpl with FSAL, adjustStep, the
// step-size selection of Engine::step, engine.cc:2021-2222).
#pragma once
#include "jm_quad.h"
#include "jm_adaptive.h"

namespace jm
{
// extra rows of the trunk stage buffer (after QRows<Tp>::NB): candidate configuration, then k_1..k_6 of the trunk tree
template<class Tp> struct QDopriRows
{
    using R = QRows<Tp>;
    using I = QInfo<Tp>;
    static constexpr int QSB = R::NB, KVB = QSB + I::NQB, KAB = KVB + 6 * I::NVB, NB = KAB + 6 * I::NVB;
    static constexpr int A0B = R::ACCVB;              // k_0.a of the trunk tree (the RK4 accumulator rows are free here)
    static constexpr int QSL = R::ACCVL, A0L = R::ACCAL;
    static constexpr int kvb(int j) { return KVB + (j - 1) * I::NVB; }
    static constexpr int kab(int j) { return KAB + (j - 1) * I::NVB; }
};

// tangent of the root placement: d with q0 (+) d = q1 (pinocchio::difference on SE(3)), [linear; angular]
template<class T> JM_DEV void root_difference(const T * q0, const T * q1, T * out)
{
    const M3<T> R0 = quat_to_matrix(q0[3], q0[4], q0[5], q0[6]);
    const M3<T> R1 = quat_to_matrix(q1[3], q1[4], q1[5], q1[6]);
    SE3<T> rel;
    rel.R = transpose(R0) * R1;
    rel.p = tmul(R0, V3<T>{q1[0] - q0[0], q1[1] - q0[1], q1[2] - q0[2]});
    const Sp<T> d = log6(rel);
    out[0] = d.l.x; out[1] = d.l.y; out[2] = d.l.z;
    out[3] = d.a.x; out[4] = d.a.y; out[5] = d.a.z;
}

// one lane of a quad: robot r, limb k, from its current time to D.t_next (at most `max_attempts` attempts)
template<class T, class Tp, class X, int SL, int SB, bool GEN = false>
JM_DEV void quad_dopri_run(const BatchArgs<T> & A, const AdaptiveArgs<T> & D, long long r, int k, const T * limb_table,
                           const StageBuf<T, SL, SB> & S, int max_attempts)
{
    using Q = QLayout<Tp>;
    using R = QRows<Tp>;
    using I = QInfo<Tp>;
    using DR = QDopriRows<Tp>;
    using AR = AdaptiveRows<Tp>;
    constexpr int N = Tp::QN, NT = Tp::QT, NQB = I::NQB, NVB = I::NVB, NV = Tp::NV;
    const unsigned B32 = (unsigned)A.B, r32 = (unsigned)r;
    CPtr<T> P = (CPtr<T>)A.P;
    const LimbTable<T> LT{limb_table + k * Q::QSTRIDE};
    const QIdx<Tp> ix = quad_indices<Tp>(k);
    const bool lead = (k == 0);
    T qb[NQB], vb[NVB], ql[N], vl[N], cmdl[N], cmdb[NT], ddqb[NVB], ddq[N];
    static_for<0, N>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        cmdl[s] = ix.has[s] ? A.command[(unsigned)ix.rm[s] * B32 + r32] : T(0);
    });
    cmdb[0] = T(0);
    static_for<1, NT>([&](auto tc) { cmdb[decltype(tc)::value] = A.command[(unsigned)Tp::trunk_motor[decltype(tc)::value] * B32 + r32]; });
    if constexpr (R::CMD_LDS)
    {
        static_for<0, N>([&](auto sc) { S.putl(R::CMDL + decltype(sc)::value, cmdl[decltype(sc)::value]); });
        static_for<0, NT>([&](auto tc) { S.putb(R::CMDB + decltype(tc)::value, cmdb[decltype(tc)::value]); });
    }
    // x0 = (q, v), k_0 = (v, a) into the stage buffer
    static_for<0, NQB>([&](auto ic) { S.putb(R::Q0B + decltype(ic)::value, A.q[(unsigned)I::qrow(decltype(ic)::value) * B32 + r32]); });
    static_for<0, NVB>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const unsigned o = (unsigned)I::vrow(i) * B32 + r32;
        S.putb(R::V0B + i, A.v[o]); S.putb(DR::A0B + i, A.a[o]);
    });
    static_for<0, N>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        T x = T(0), y = T(0), z = T(0);
        if (ix.has[s])
        {
            x = A.q[(unsigned)ix.rq[s] * B32 + r32]; y = A.v[(unsigned)ix.rv[s] * B32 + r32]; z = A.a[(unsigned)ix.rv[s] * B32 + r32];
        }
        S.putl(R::Q0L + s, x); S.putl(R::V0L + s, y); S.putl(DR::A0L + s, z);
    });
    // controller state of this robot (identical in its four lanes)
    const unsigned long long Bq = (unsigned long long)D.B;
    double t = D.fs[AD_T * Bq + r32], dt = D.fs[AD_DT * Bq + r32], dtLargest = D.fs[AD_DT_LARGEST * Bq + r32];
    double dtLargestPrev = D.fs[AD_DT_LARGEST_PREV * Bq + r32];
    int iter = D.is[AD_ITER * Bq + r32], iterFailed = D.is[AD_ITER_FAILED * Bq + r32];
    int tooLarge = D.new_step ? 0 : D.is[AD_SUCC_TOO_LARGE * Bq + r32], failed = D.new_step ? 0 : D.is[AD_SUCC_FAILED * Bq + r32];
    int st = D.status ? D.status[r32] : 0;
    int attempts = 0;
    bool moved = false;
    X::sync();
    X::table_ready();
    // limb stage derivatives: rows of the caller's workspace, private to this lane
    T * const kvl = D.ws + (unsigned long long)AR::KV * Bq + r32;
    T * const kal = D.ws + (unsigned long long)AR::KA * Bq + r32;
    auto krow = [&](int j, int s) -> unsigned long long { return (unsigned long long)((j - 1) * NV + ix.rv[s]) * Bq; };
    bool active = true;
    for (;;)
    {
        // ---- step-size selection (engine.cc:2021-2131)
        active = (D.t_next - t > STEPPER_MIN_TIMESTEP) && !(st & (JM_LANE_STEPPER_FAILURE | JM_LANE_NAN));
        if (active && (dt < STEPPER_MIN_TIMESTEP || failed > D.succ_failed_max))
        {
            st |= JM_LANE_STEPPER_FAILURE;
            active = false;
        }
        if (attempts >= max_attempts) break;     // (uniform: every quad counts its own attempts, the bound is shared)
        if (!X::wave_any(active)) break;
        if (active)
        {
            ++attempts;
            {
                double thr = STEPPER_MIN_TIMESTEP;
                if (tooLarge == 0) thr = fmin(fmax(0.1 * dt, STEPPER_MIN_TIMESTEP), SIMULATION_MIN_TIMESTEP);
                if (D.t_next - t < dt || (tooLarge <= 1 && D.t_next - t < dt + thr)) dt = D.t_next - t;
                if (dt > SIMULATION_MIN_TIMESTEP)
                {
                    const double res = fmod(dt, SIMULATION_MIN_TIMESTEP);
                    if (res > STEPPER_MIN_TIMESTEP && res < SIMULATION_MIN_TIMESTEP - STEPPER_MIN_TIMESTEP && dt - res > STEPPER_MIN_TIMESTEP)
                        dt -= res;
                }
            }
            const bool bpReached = dtLargest > dt;
            const T dtT = (T)dt;
            // ---- stages 1..6: x_i = x0 (+) dt sum_j A_ij k_j, k_i = f(x_i); stage 6 is the 5th-order solution (FSAL)
#pragma nounroll
            for (int i = 1; i <= 6; ++i)
            {
                JM_REFRESH();
                const T s0 = dtT * (T)dopri::A[i][0];
                {
                    T q0b[NQB], incb[NVB], acc[NVB];
                    static_for<0, NQB>([&](auto ic) { q0b[decltype(ic)::value] = S.getb(R::Q0B + decltype(ic)::value); });
                    static_for<0, NVB>([&](auto ic) {
                        constexpr int d = decltype(ic)::value;
                        incb[d] = s0 * S.getb(R::V0B + d);
                        acc[d] = s0 * S.getb(DR::A0B + d);
                    });
                    // (all five earlier stages, the ones that do not exist yet with a zero operand: the reads are
                    // issued together instead of one dependent round trip per stage)
                    static_for<1, 6>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        const bool on = j < i;
                        const T sj = dtT * (T)dopri::A[i][j];
                        static_for<0, NVB>([&](auto ic) {
                            constexpr int d = decltype(ic)::value;
                            const T kv = S.getb(DR::KVB + (j - 1) * NVB + d), ka = S.getb(DR::KAB + (j - 1) * NVB + d);
                            incb[d] += sj * (on ? kv : T(0));
                            acc[d] += sj * (on ? ka : T(0));
                        });
                    });
                    integrate_freeflyer<T>(q0b, incb, qb);
                    static_for<1, NT>([&](auto tc) { qb[6 + decltype(tc)::value] = q0b[6 + decltype(tc)::value] + incb[5 + decltype(tc)::value]; });
                    static_for<0, NVB>([&](auto ic) {
                        constexpr int d = decltype(ic)::value;
                        vb[d] = S.getb(R::V0B + d) + acc[d];
                        S.putb(DR::KVB + (i - 1) * NVB + d, vb[d]);
                        if constexpr (R::LONG) S.putb(R::KVB + d, vb[d]);
                    });
                }
                static_for<0, N>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    const T v0 = S.getl(R::V0L + s);
                    T inc = s0 * v0, acc = s0 * S.getl(DR::A0L + s);
                    T kvj[5], kaj[5];
                    static_for<1, 6>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        kvj[j - 1] = kvl[krow(j, s)]; kaj[j - 1] = kal[krow(j, s)];
                    });
                    static_for<1, 6>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        const bool on = ix.has[s] && j < i;
                        const T sj = dtT * (T)dopri::A[i][j];
                        inc += sj * (on ? kvj[j - 1] : T(0));
                        acc += sj * (on ? kaj[j - 1] : T(0));
                    });
                    ql[s] = S.getl(R::Q0L + s) + inc;
                    vl[s] = v0 + acc;
                    if (!ix.has[s]) { ql[s] = T(0); vl[s] = T(0); }   // dummy joints never move
                    else kvl[krow(i, s)] = vl[s];
                    if constexpr (R::LONG) S.putl(R::KVL + s, vl[s]);
                });
                int evst = 0;
                quad_eval<T, Tp, X, false, StageBuf<T, SL, SB>, 0, NoKeep, GEN>(P, LT, A, r32, k, ix, S, qb, vb, ql, vl, cmdb, cmdl, false, ddqb, ddq, evst);
                static_for<0, NVB>([&](auto ic) { S.putb(DR::KAB + (i - 1) * NVB + decltype(ic)::value, ddqb[decltype(ic)::value]); });
                static_for<0, N>([&](auto sc) { if (ix.has[decltype(sc)::value]) kal[krow(i, decltype(sc)::value)] = ddq[decltype(sc)::value]; });
            }
#ifdef JM_DOPRI_BARRIER   // (experiment of DESIGN.md section 4.7: nothing of the stage loop is carried into the estimate in registers)
            JM_REFRESH();
#endif
            // ---- embedded error estimate (runge_kutta_dopri_stepper.cc:18-87): solution = stage 6 = (qb|ql, vb|vl),
            // alternative (4th order) solution = x0 (+) dt sum_j e_j k_j, norm = max |difference / scale|
            double error = 0.0;
            bool nan = false, a_nan = false;
            const T tolRel = (T)D.tol_rel, tolAbs = (T)D.tol_abs;
            const T e0 = dtT * (T)dopri::E[0];
            {
                T q0b[NQB], zero[NQB], sc[NVB], d[NVB], qalt[NQB];
                static_for<0, NQB>([&](auto ic) { q0b[decltype(ic)::value] = S.getb(R::Q0B + decltype(ic)::value); zero[decltype(ic)::value] = T(0); });
                root_difference<T>(q0b, zero, sc);
                static_for<1, NT>([&](auto tc) { sc[5 + decltype(tc)::value] = T(0) - q0b[6 + decltype(tc)::value]; });
                static_for<0, NVB>([&](auto ic) { sc[decltype(ic)::value] = fabs_(sc[decltype(ic)::value]) * tolRel + tolAbs; });
                static_for<0, NVB>([&](auto ic) { d[decltype(ic)::value] = e0 * S.getb(R::V0B + decltype(ic)::value); });
                static_for<1, 7>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const T sj = dtT * (T)dopri::E[j];
                    static_for<0, NVB>([&](auto ic) { d[decltype(ic)::value] += sj * S.getb(DR::KVB + (j - 1) * NVB + decltype(ic)::value); });
                });
                integrate_freeflyer<T>(q0b, d, qalt);
                static_for<1, NT>([&](auto tc) { qalt[6 + decltype(tc)::value] = q0b[6 + decltype(tc)::value] + d[5 + decltype(tc)::value]; });
                root_difference<T>(qb, qalt, d);
                static_for<1, NT>([&](auto tc) { d[5 + decltype(tc)::value] = qalt[6 + decltype(tc)::value] - qb[6 + decltype(tc)::value]; });
                static_for<0, NVB>([&](auto ic) {
                    const double e = (double)fabs_(d[decltype(ic)::value] / sc[decltype(ic)::value]);
                    nan |= (e != e);
                    error = fmax(error, e);
                });
                // velocity part
                static_for<0, NVB>([&](auto ic) { d[decltype(ic)::value] = e0 * S.getb(DR::A0B + decltype(ic)::value); });
                static_for<1, 7>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const T sj = dtT * (T)dopri::E[j];
                    static_for<0, NVB>([&](auto ic) { d[decltype(ic)::value] += sj * S.getb(DR::KAB + (j - 1) * NVB + decltype(ic)::value); });
                });
                static_for<0, NVB>([&](auto ic) {
                    constexpr int c = decltype(ic)::value;
                    const T v0 = S.getb(R::V0B + c);
                    const T scv = fabs_(T(0) - v0) * tolRel + tolAbs;
                    const double e = (double)fabs_(((v0 + d[c]) - vb[c]) / scv);
                    nan |= (e != e);
                    error = fmax(error, e);
                    a_nan |= (ddqb[c] != ddqb[c]);
                });
            }
            static_for<0, N>([&](auto sc_) {
                constexpr int s = decltype(sc_)::value;
                const T q0 = S.getl(R::Q0L + s), v0 = S.getl(R::V0L + s);
                T dq = e0 * v0, dv = e0 * S.getl(DR::A0L + s);
                T kvj[6], kaj[6];
                static_for<1, 7>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    kvj[j - 1] = kvl[krow(j, s)]; kaj[j - 1] = kal[krow(j, s)];
                });
                static_for<1, 7>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const T sj = dtT * (T)dopri::E[j];
                    dq += sj * (ix.has[s] ? kvj[j - 1] : T(0));
                    dv += sj * (ix.has[s] ? kaj[j - 1] : T(0));
                });
                const T scq = fabs_(T(0) - q0) * tolRel + tolAbs;
                const double eq = (double)fabs_(((q0 + dq) - ql[s]) / scq);
                const T scv = fabs_(T(0) - v0) * tolRel + tolAbs;
                const double ev = (double)fabs_(((v0 + dv) - vl[s]) / scv);
                nan |= (eq != eq) || (ev != ev);
                error = fmax(error, fmax(eq, ev));
                a_nan |= (ddq[s] != ddq[s]);
#ifdef JM_DOPRI_DEBUG   // (the terms of the first attempt's estimate, into rows the persistent kernel does not use)
#ifndef JM_DOPRI_DEBUG_ATT
#define JM_DOPRI_DEBUG_ATT 1
#endif
                if (attempts == JM_DOPRI_DEBUG_ATT && ix.has[s])
                {
                    D.ws[(unsigned long long)(AR::QS + ix.rq[s]) * Bq + r32] = JM_DOPRI_DEBUG == 1 ? (q0 + dq) - ql[s] : (JM_DOPRI_DEBUG == 2 ? ql[s] : q0 + dq);
                    D.ws[(unsigned long long)(AR::CMD + ix.rm[s]) * Bq + r32] = JM_DOPRI_DEBUG == 1 ? (v0 + dv) - vl[s] : (JM_DOPRI_DEBUG == 2 ? vl[s] : v0 + dv);
                }
#endif
            });
            {
                // over the four limbs (fmax drops NaN operands: the flags travel separately)
                error = fmax(error, X::template perm_<0xB1>(error));
                error = fmax(error, X::template perm_<0x4E>(error));
                const int fl = X::quad_or((nan ? 1 : 0) | (a_nan ? 2 : 0));
                nan = (fl & 1) != 0; a_nan = (fl & 2) != 0;
            }
            // ---- accept / reject, next step size (adjustStep; engine.cc:2132-2221)
            double dtl = dt;
            int rc;
            if (nan) rc = 2;
            else if (error < 1.0)
            {
                if (error < fmin(dopri::ERROR_THRESHOLD, pow(dopri::SAFETY, dopri::STEPPER_ORDER)))
                {
                    const double clipped = fmax(error, pow(dopri::MAX_FACTOR / dopri::SAFETY, -dopri::STEPPER_ORDER));
                    dtl *= dopri::SAFETY * pow(clipped, -1.0 / dopri::STEPPER_ORDER);
                }
                rc = a_nan ? 2 : 0;
            }
            else
            {
                dtl *= fmax(dopri::SAFETY * pow(error, -1.0 / (dopri::STEPPER_ORDER - 2.0)), dopri::MIN_FACTOR);
                rc = 1;
            }
            if (rc == 0)
            {
                static_for<0, NQB>([&](auto ic) { S.putb(R::Q0B + decltype(ic)::value, qb[decltype(ic)::value]); });
                static_for<0, NVB>([&](auto ic) {
                    S.putb(R::V0B + decltype(ic)::value, vb[decltype(ic)::value]);
                    S.putb(DR::A0B + decltype(ic)::value, ddqb[decltype(ic)::value]);
                });
                static_for<0, N>([&](auto sc_) {
                    constexpr int s = decltype(sc_)::value;
                    S.putl(R::Q0L + s, ql[s]); S.putl(R::V0L + s, vl[s]); S.putl(DR::A0L + s, ddq[s]);
                });
                moved = true;
                t += dt;
                tooLarge = 0; failed = 0; iter += 1;
                if (bpReached && dt < dtl && dtl < dtLargestPrev * D.dt_restore_threshold_rel) dtl = dtLargestPrev;
                dtLargestPrev = dtl;
            }
            else
            {
                if (rc == 2) dtl *= 0.1;
                if (rc == 1) tooLarge += 1;
                failed += 1; iterFailed += 1;
            }
            dtLargest = dtl;
            dt = fmin(dtl, D.dt_max);
        }
    }
    // ---- write back: state (only when it moved), controller state, status
    if (moved)
    {
        JM_REFRESH();
        if (lead)
        {
            static_for<0, NQB>([&](auto ic) { A.q[(unsigned)I::qrow(decltype(ic)::value) * B32 + r32] = S.getb(R::Q0B + decltype(ic)::value); });
            static_for<0, NVB>([&](auto ic) {
                const unsigned o = (unsigned)I::vrow(decltype(ic)::value) * B32 + r32;
                A.v[o] = S.getb(R::V0B + decltype(ic)::value); A.a[o] = S.getb(DR::A0B + decltype(ic)::value);
            });
        }
        static_for<0, N>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if (ix.has[s])
            {
                A.q[(unsigned)ix.rq[s] * B32 + r32] = S.getl(R::Q0L + s);
                A.v[(unsigned)ix.rv[s] * B32 + r32] = S.getl(R::V0L + s);
                A.a[(unsigned)ix.rv[s] * B32 + r32] = S.getl(DR::A0L + s);
            }
        });
    }
    if (lead)
    {
        D.fs[AD_T * Bq + r32] = t; D.fs[AD_DT * Bq + r32] = dt; D.fs[AD_DT_LARGEST * Bq + r32] = dtLargest;
        D.fs[AD_DT_LARGEST_PREV * Bq + r32] = dtLargestPrev; D.fs[AD_DT_TRY * Bq + r32] = 0.0;
        D.is[AD_ITER * Bq + r32] = iter; D.is[AD_ITER_FAILED * Bq + r32] = iterFailed;
        D.is[AD_SUCC_TOO_LARGE * Bq + r32] = tooLarge; D.is[AD_SUCC_FAILED * Bq + r32] = failed;
        D.is[AD_ACTIVE * Bq + r32] = active ? 1 : 0;
        if (D.status) D.status[r32] = st;
#ifndef JM_HOST_EMU
        // robots still on their way (attempt bound reached) and the largest attempt count, for the host
        if (active) atomicAdd(D.n_active, 1);
        atomicMax(D.n_active + 1, attempts);
#else
        if (active) D.n_active[0] += 1;
        if (attempts > D.n_active[1]) D.n_active[1] = attempts;
#endif
    }
}

#ifndef JM_HOST_EMU
template<class T, class Tp> constexpr int qdopri_block_waves()
{
    constexpr long per_wave = (long)sizeof(T) * (QRows<Tp>::NL * 64 + QDopriRows<Tp>::NB * 16);
    constexpr long table = (long)sizeof(T) * QLayout<Tp>::TABLE;
    int best = 1, best_resident = 0;
    for (int w = 1; w <= 4; w *= 2)
    {
        long blocks = (160L * 1024) / (table + w * per_wave);
        if (blocks * w > 8) blocks = 8 / w;
        const int resident = (int)blocks * w;
        if (resident >= best_resident) { best = w; best_resident = resident; }
    }
    return best;
}
template<class T, class Tp>
__global__ void __launch_bounds__((64 * qdopri_block_waves<T, Tp>())) __attribute__((amdgpu_waves_per_eu(1)))
k_quad_dopri(const BatchArgs<T> A, const AdaptiveArgs<T> D, int max_attempts)
{
    using Q = QLayout<Tp>;
    constexpr int NTH = 64 * qdopri_block_waves<T, Tp>();
    __shared__ T table[Q::TABLE];
    __shared__ T stage_l[QRows<Tp>::NL * NTH];
    __shared__ T stage_b[QDopriRows<Tp>::NB * (NTH / 4)];
#pragma nounroll
    for (int i = threadIdx.x; i < Q::TABLE; i += NTH) table[i] = A.P[Q::OFFSET + i];
    const long long r = (long long)blockIdx.x * (NTH / 4) + (threadIdx.x >> 2);
    const int k = threadIdx.x & 3;
    if (r >= A.B) return;
    const StageBuf<T, NTH, NTH / 4> S{stage_l + threadIdx.x, stage_b + (threadIdx.x >> 2), k == 0};
    quad_dopri_run<T, Tp, DppQuad, NTH, NTH / 4>(A, D, r, k, table, S, max_attempts);
}
// the same with the per-environment variation of DESIGN.md section 4.9 compiled in (the lanes keep their places in this
// form of the stepper, so per-lane body parameters, the height map and applied wrenches simply follow)
template<class T, class Tp>
__global__ void __launch_bounds__((64 * qdopri_block_waves<T, Tp>())) __attribute__((amdgpu_waves_per_eu(1)))
k_quad_dopri_gen(const BatchArgs<T> A, const AdaptiveArgs<T> D, int max_attempts)
{
    using Q = QLayout<Tp>;
    constexpr int NTH = 64 * qdopri_block_waves<T, Tp>();
    __shared__ T table[Q::TABLE];
    __shared__ T stage_l[QRows<Tp>::NL * NTH];
    __shared__ T stage_b[QDopriRows<Tp>::NB * (NTH / 4)];
#pragma nounroll
    for (int i = threadIdx.x; i < Q::TABLE; i += NTH) table[i] = A.P[Q::OFFSET + i];
    const long long r = (long long)blockIdx.x * (NTH / 4) + (threadIdx.x >> 2);
    const int k = threadIdx.x & 3;
    if (r >= A.B) return;
    const StageBuf<T, NTH, NTH / 4> S{stage_l + threadIdx.x, stage_b + (threadIdx.x >> 2), k == 0};
    quad_dopri_run<T, Tp, DppQuad, NTH, NTH / 4, true>(A, D, r, k, table, S, max_attempts);
}
#endif
}  // namespace jm
