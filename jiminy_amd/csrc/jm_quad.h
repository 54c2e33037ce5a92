// jm_quad.h -- limb-parallel variant of the per-step physics: FOUR lanes per robot.
//
// Why: in float64 the per-robot live state of the three ABA sweeps of an 18-DoF quadruped
// (~300-400 scalars) does not fit the 512 VGPRs of one lane, and the one-robot-per-lane kernel
// (jm_kernels.h) turns that into ~12 KB/lane of scratch traffic (profiles/r01_v1_*).  Here a robot
// made of a free-flyer trunk and K = 4 kinematic chains ("limbs") of N revolute joints is spread
// over one DPP quad: lane k of the quad owns limb k (its FK, contact point, motors and the ABA
// sweeps along the chain -- all in registers), every lane redundantly carries the trunk, and the
// only cross-lane traffic is the child->parent reduction of the articulated inertia / bias force
// of the 4 limbs into the trunk (27 scalars, two `v_mov_dpp quad_perm` butterflies).  A wave64
// therefore advances 16 robots; batch 65 536 = 4096 waves.
//
// Limb constants differ per lane, so they cannot sit in SGPRs: they are staged once per block in
// LDS as a [4][QSTRIDE] table (odd stride => the four distinct addresses of a `ds_read_b64` fall
// in different banks, identical addresses broadcast).  Trunk constants and options stay in the
// constant address space (scalar loads).
//
// Same reference functions as jm_kernels.h (see its header for file:line citations).
#pragma once
#include "jm_kernels.h"

namespace jm
{
// ---------------------------------------------------------------- limb table layout (doubles)
template<class Tp> struct QLayout
{
    static constexpr int QJ = 38;  // per chain joint: plc 12 | rbi 10 | axis 3 | rotor | qlo qhi | motor 9 | enc red
    static constexpr int J_PLC = 0, J_RBI = 12, J_AXIS = 22, J_ROTOR = 25, J_QLO = 26, J_QHI = 27, J_MOTOR = 28, J_ENC = 37;
    static constexpr int CONTACT = Tp::QN * QJ;        // per contact: frame 12 | force-sensor relative frame 12
    static constexpr int QC = 24;
    static constexpr int RAW = CONTACT + Tp::QCL * QC;
    static constexpr int QSTRIDE = (RAW % 2 == 0) ? RAW + 1 : RAW;  // odd
    static constexpr int TABLE = 4 * QSTRIDE;
    static constexpr int OFFSET = Layout<Tp>::TOTAL;   // appended to the parameter block
    static constexpr int TOTAL = OFFSET + TABLE;
};

// per-lane select among 4 compile-time constants
JM_DEV int sel4(int k, int c0, int c1, int c2, int c3) { return k == 0 ? c0 : (k == 1 ? c1 : (k == 2 ? c2 : c3)); }

// limb table accessor: element `off` of limb k
template<class T> struct LimbTable
{
    const T * base;  // already offset by k * QSTRIDE
    JM_DEV T operator()(int off) const { return base[off]; }
    JM_DEV V3<T> v3(int o) const { return {base[o], base[o + 1], base[o + 2]}; }
    JM_DEV M3<T> m3(int o) const { return {base[o], base[o + 1], base[o + 2], base[o + 3], base[o + 4], base[o + 5], base[o + 6], base[o + 7], base[o + 8]}; }
    JM_DEV SE3<T> se3(int o) const { return {m3(o), v3(o + 9)}; }
    JM_DEV RBI<T> rbi(int o) const { return {base[o], v3(o + 1), S3<T>{base[o + 4], base[o + 5], base[o + 6], base[o + 7], base[o + 8], base[o + 9]}}; }
};

template<class T, class X> JM_DEV Sp<T> quad_sum6(Sp<T> a)
{
    return {{X::quad_sum(a.l.x), X::quad_sum(a.l.y), X::quad_sum(a.l.z)}, {X::quad_sum(a.a.x), X::quad_sum(a.a.y), X::quad_sum(a.a.z)}};
}
template<class T, class X> JM_DEV AI<T> quad_sum_ai(const AI<T> & Y)
{
    AI<T> r;
    r.A = {X::quad_sum(Y.A.xx), X::quad_sum(Y.A.xy), X::quad_sum(Y.A.xz), X::quad_sum(Y.A.yy), X::quad_sum(Y.A.yz), X::quad_sum(Y.A.zz)};
    r.B = {X::quad_sum(Y.B.m00), X::quad_sum(Y.B.m01), X::quad_sum(Y.B.m02), X::quad_sum(Y.B.m10), X::quad_sum(Y.B.m11), X::quad_sum(Y.B.m12),
           X::quad_sum(Y.B.m20), X::quad_sum(Y.B.m21), X::quad_sum(Y.B.m22)};
    r.D = {X::quad_sum(Y.D.xx), X::quad_sum(Y.D.xy), X::quad_sum(Y.D.xz), X::quad_sum(Y.D.yy), X::quad_sum(Y.D.yz), X::quad_sum(Y.D.zz)};
    return r;
}

// stage buffer (LDS on the GPU). Limb rows are per lane (element `row` at sl[row * SL]); trunk rows
// are identical in the 4 lanes of a quad and stored once per robot (sb[row * SB], written by the
// lead lane only, read back by all four: LDS operations of one wave execute in order).
template<class T, int SL, int SB> struct StageBuf
{
    T * sl;
    T * sb;
    bool wb;  // this lane writes the trunk rows
    JM_DEV T getl(int row) const { return sl[row * SL]; }
    JM_DEV void putl(int row, T x) const { sl[row * SL] = x; }
    JM_DEV T getb(int row) const { return sb[row * SB]; }
    JM_DEV void putb(int row, T x) const { if (wb) sb[row * SB] = x; }
};
// rows of the stage buffer
template<class Tp> struct QRows
{
    static constexpr int N = Tp::QN;
    static constexpr int Q0B = 0, V0B = 7, A0B = 13, ACCVB = 19, ACCAB = 25, KVB = 31, NB = 37;  // trunk rows
    static constexpr int Q0L = 0, V0L = N, A0L = 2 * N, ACCVL = 3 * N, ACCAL = 4 * N, KVL = 5 * N, NL = 6 * N;  // limb rows
};

// All global accesses use a uniform base pointer + an unsigned 32-bit per-lane element offset, so
// that they select the `saddr + voffset` addressing form instead of pinning a 64-bit VGPR address
// per access (the batch size is bounded accordingly in jm_batch_create).
template<class T> JM_DEV void put6(T * base, unsigned B, unsigned r, unsigned row0, Sp<T> f)
{
    const unsigned o = row0 * B + r;
    base[o] = f.l.x; base[o + B] = f.l.y; base[o + 2 * B] = f.l.z;
    base[o + 3 * B] = f.a.x; base[o + 4 * B] = f.a.y; base[o + 5 * B] = f.a.z;
}

// a = f(q, v) for one robot spread over a quad; lane k evaluates limb k.
//   qb[7], vb[6] : trunk configuration / velocity (identical in the 4 lanes)
//   ql[N], vl[N], cmd[N] : this limb's joints
// When `emit` (uniform) is set -- last evaluation of a step, `start`, `reset` -- the outputs that
// derive from this evaluation (RobotState::u / uMotor / fExternal, contact forces, the extra terms
// of engine.cc:800-905 and, if `sensors`, the sensor rows) are written right where their inputs
// are live, so that nothing has to stay in registers for a separate output phase.
template<class T, class Tp, class X>
JM_DEV void quad_eval(CPtr<T> P, const LimbTable<T> & LT, const BatchArgs<T> & A, long long r, int k, const int * rv,
                      const int * rm, const T * qb, const T * vb, const T * ql, const T * vl, const T * cmd,
                      bool emit, bool sensors, T * ddq1, T * ddq, int & status)
{
    using L = Layout<Tp>;
    using Q = QLayout<Tp>;
    constexpr int N = Tp::QN;
    const unsigned B32 = (unsigned)A.B;
    unsigned r32 = (unsigned)r;
    JM_OPAQUE(r32);
    const bool lead = (k == 0);
    const bool emit_sens = emit && sensors;
    // ---- encoders read the state itself (basic_sensors.cc:509-539)
    if constexpr (Tp::QHAS_ENC)
        if (emit_sens && A.encoder)
            static_for<0, N>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                const int si = sel4(k, Tp::limb_enc[0][s], Tp::limb_enc[1][s], Tp::limb_enc[2][s], Tp::limb_enc[3][s]);
                T pos = ql[s], vel = vl[s];
                if constexpr (Tp::QENC_SIDE == 0)
                {
                    const T red = LT(s * Q::QJ + Q::J_ENC);
                    pos *= red;
                    vel *= red;
                }
                A.encoder[(unsigned)(2 * si) * B32 + r32] = pos;
                A.encoder[(unsigned)(2 * si + 1) * B32 + r32] = vel;
            });
    // ---- trunk kinematics (free-flyer, joint 1)
    SE3<T> liM1;
    {
        SE3<T> Mj;
        Mj.R = quat_to_matrix(qb[3], qb[4], qb[5], qb[6]);
        Mj.p = {qb[0], qb[1], qb[2]};
        liM1 = Mj;  // the free-flyer root joint sits at the world origin (checked by jm_model_create)
    }
    const Sp<T> v1 = {{vb[0], vb[1], vb[2]}, {vb[3], vb[4], vb[5]}};
    const bool want_energy = emit && A.energy;
    const V3<T> g = ld_v3<T>(P, L::OPT), gw = ld_v3<T>(P, L::OPT + 3);
    // ---- chain kinematics
    SE3<T> liMi[N];
    Sp<T> vel[N];   // spatial velocities; bias acceleration / force are re-derived where they are used

    M3<T> oR = liM1.R;
    V3<T> op = liM1.p;
    T kin = T(0), pot = T(0), rot = T(0);
    {
        Sp<T> vp = v1;
        static_for<0, N>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int o = s * Q::QJ;
            T c, sn;
            sincos_(ql[s], &sn, &c);
            const V3<T> n = LT.v3(o + Q::J_AXIS);
            const SE3<T> plc = LT.se3(o + Q::J_PLC);
            liMi[s] = {plc.R * rot_rodrigues(n, c, sn), plc.p};
            const Sp<T> vj = {zero3<T>(), vl[s] * n};
            const Sp<T> v = vj + actinv_motion(liMi[s], vp);
            vel[s] = v;
            op = op + oR * liMi[s].p;
            oR = oR * liMi[s].R;
            if (want_energy)
            {
                const RBI<T> Y = LT.rbi(o + Q::J_RBI);
                kin += rbi_vtiv(Y, v);
                pot -= Y.m * dot(op + oR * Y.c, g);
                rot += LT(o + Q::J_ROTOR) * vl[s] * vl[s];
            }
            if (LT(o + Q::J_QHI) < ql[s] || ql[s] < LT(o + Q::J_QLO)) status |= JM_LANE_OUT_OF_BOUNDS;
            vp = v;
        });
    }
    const Sp<T> vlast = vel[N - 1];
    const RBI<T> Y1 = ld_rbi<T>(P, L::JOINT + 1 * L::JSTRIDE + 12);
    if (want_energy)
    {
        kin = X::quad_sum(kin);
        pot = X::quad_sum(pot);
        rot = X::quad_sum(rot);
        kin += rbi_vtiv(Y1, v1);
        pot -= Y1.m * dot(liM1.p + liM1.R * Y1.c, g);
#pragma unroll
        for (int i = 0; i < 6; ++i) rot += P[L::ROTOR + i] * vb[i] * vb[i];
        if (lead)
        {
            A.energy[r32] = T(0.5) * kin + T(0.5) * rot;
            A.energy[B32 + r32] = pot;
        }
    }
    // ---- contact points on the last chain joint (engine.cc:3117-3238, 3394-3425)
    Sp<T> fext_last = zero6<T>();
    Sp<T> cf[c_max(Tp::QCL, 1)];
    static_for<0, Tp::QCL>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        const SE3<T> fr = LT.se3(Q::CONTACT + c * Q::QC);
        const T depth = op.z + dot(V3<T>{oR.m20, oR.m21, oR.m22}, fr.p);
        Sp<T> fl = zero6<T>();
        if (depth < T(0))
        {
            const V3<T> vj = vlast.l + cross(vlast.a, fr.p);
            const V3<T> vW = oR * vj;
            const V3<T> fW = contact_law<T, Tp>(P, depth, vW);
            fl.l = tmul(oR, fW);
            fl.a = cross(fr.p, fl.l);
        }
        fext_last = fext_last + fl;
        cf[c] = actinv_force(fr, fl);
    });
    if (A.mode == MODE_START || A.mode == MODE_RESET)
    {
        T fmax2 = T(0);
        static_for<0, Tp::QCL>([&](auto cc) { fmax2 = fmax_(fmax2, dot(cf[decltype(cc)::value].l, cf[decltype(cc)::value].l)); });
        if (fmax2 > T(1e10)) status |= JM_LANE_FORCE_OVERFLOW;
    }
    if (emit)
    {
        if (A.f_external)
        {
            if (lead) { put6(A.f_external, B32, r32, 0, zero6<T>()); put6(A.f_external, B32, r32, 6, zero6<T>()); }
            static_for<0, N>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                const int j = sel4(k, Tp::limb_joint[0][s], Tp::limb_joint[1][s], Tp::limb_joint[2][s], Tp::limb_joint[3][s]);
                put6(A.f_external, B32, r32, 6 * j, (s == N - 1) ? fext_last : zero6<T>());
            });
        }
        static_for<0, Tp::QCL>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if (A.contact_forces)
            {
                const int ci = sel4(k, Tp::limb_contact[0][c], Tp::limb_contact[1][c], Tp::limb_contact[2][c], Tp::limb_contact[3][c]);
                put6(A.contact_forces, B32, r32, 6 * ci, cf[c]);
            }
            if constexpr (Tp::QHAS_CS)
                if (sensors && A.contact)
                {
                    const int si = sel4(k, Tp::limb_cs[0][c], Tp::limb_cs[1][c], Tp::limb_cs[2][c], Tp::limb_cs[3][c]);
                    const unsigned o = (unsigned)(3 * si) * B32 + r32;
                    A.contact[o] = cf[c].l.x; A.contact[o + B32] = cf[c].l.y; A.contact[o + 2 * B32] = cf[c].l.z;
                }
        });
        if constexpr (Tp::QHAS_FORCE)
            if (sensors && A.force)
            {
                Sp<T> sum = zero6<T>();
                static_for<0, Tp::QCL>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    sum = sum + act_force(LT.se3(Q::CONTACT + c * Q::QC + 12), cf[c]);
                });
                const int si = sel4(k, Tp::limb_force[0], Tp::limb_force[1], Tp::limb_force[2], Tp::limb_force[3]);
                put6(A.force, B32, r32, 6 * si, sum);
            }
    }
    // ---- motors (one per chain joint, uniform flags; basic_motors.cc:83-143)
    T u[N];
    static_for<0, N>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int o = s * Q::QJ + Q::J_MOTOR;
        constexpr int fl = Tp::motor_flags[0];
        const T red = LT(o), elim = LT(o + 1), vlim = LT(o + 2), islope = LT(o + 3);
        const T vjnt = vl[s];
        const T vmot = red * vjnt;
        T um = cmd[s];
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
            const T fds = LT(o + 8);
            if (vjnt > T(0)) ut += LT(o + 4) * vjnt + LT(o + 6) * tanh_(fds * vjnt);
            else ut += LT(o + 5) * vjnt + LT(o + 7) * tanh_(fds * vjnt);
        }
        u[s] = ut;
        if (emit)
        {
            if (A.u_motor) A.u_motor[(unsigned)rm[s] * B32 + r32] = um;
            if (A.u) A.u[(unsigned)rv[s] * B32 + r32] = ut;
            if constexpr (Tp::QHAS_EFF)
                if (sensors && A.effort)
                {
                    const int si = sel4(k, Tp::limb_eff[0][s], Tp::limb_eff[1][s], Tp::limb_eff[2][s], Tp::limb_eff[3][s]);
                    A.effort[(unsigned)si * B32 + r32] = um;
                }
        }
    });
    if (emit && A.u && lead)
    {
#pragma unroll
        for (int i = 0; i < 6; ++i) A.u[(unsigned)i * B32 + r32] = T(0);
    }
    // ---- ABA pass 2 along the chain, leaf -> trunk (AbaBackwardStep)
    JM_REFRESH();
    Sp<T> U[N];
    T dinv[N];
    AI<T> Ia;
    Sp<T> pa_up = zero6<T>();
    static_rfor<0, N>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int o = s * Q::QJ;
        const RBI<T> Y = LT.rbi(o + Q::J_RBI);
        const V3<T> n = LT.v3(o + Q::J_AXIS);
        Sp<T> f = cross_mf(vel[s], rbi_mul(Y, vel[s]));  // bias force v x* (I v)
        if constexpr (s == N - 1) { f = f - fext_last; Ia = ai_from_rbi(Y); }
        else { f = f + pa_up; Ia = ai_from_rbi(Y) + Ia; }
        const Sp<T> cb = cross_mm(vel[s], Sp<T>{zero3<T>(), vl[s] * n});  // bias acceleration v x S qd
        const T uj = u[s] - dot(n, f.a);
        u[s] = uj;
        const Sp<T> Us = {Ia.B * n, Ia.D * n};
        const T D = dot(n, Us.a) + LT(o + Q::J_ROTOR);
        const T di = T(1) / D;
        U[s] = Us;
        dinv[s] = di;
        ai_rank1_sub(Ia, Us, di);
        const Sp<T> Ya = ai_mul(Ia, cb);
        const T ud = uj * di;
        const Sp<T> pa = {f.l + Ya.l + ud * Us.l, f.a + Ya.a + ud * Us.a};
        Ia = ai_transform(liMi[s], Ia);
        pa_up = act_force(liMi[s], pa);
    });
    // ---- child -> parent reduction over the 4 limbs (the only cross-lane step of the dynamics)
    const AI<T> Ysum = quad_sum_ai<T, X>(Ia);
    const Sp<T> fsum = quad_sum6<T, X>(pa_up);
    // ---- trunk: u -= S^T f ; (Ia + Im) ddq = u - Ia a_gf (free-flyer calc_aba + pass 3)
    Sp<T> agf1;
    {
        const AI<T> I1 = ai_from_rbi(Y1) + Ysum;
        const Sp<T> f1 = cross_mf(v1, rbi_mul(Y1, v1)) + fsum;
        agf1 = actinv_motion(liM1, Sp<T>{-g, -gw});  // bias v x v = 0 for the free-flyer
        const Sp<T> Ya = ai_mul(I1, agf1);
        T b[6] = {-f1.l.x - Ya.l.x, -f1.l.y - Ya.l.y, -f1.l.z - Ya.l.z, -f1.a.x - Ya.a.x, -f1.a.y - Ya.a.y, -f1.a.z - Ya.a.z};
        T M[6][6];
        M[0][0] = I1.A.xx; M[1][0] = I1.A.xy; M[2][0] = I1.A.xz; M[1][1] = I1.A.yy; M[2][1] = I1.A.yz; M[2][2] = I1.A.zz;
        M[3][0] = I1.B.m00; M[3][1] = I1.B.m10; M[3][2] = I1.B.m20;
        M[4][0] = I1.B.m01; M[4][1] = I1.B.m11; M[4][2] = I1.B.m21;
        M[5][0] = I1.B.m02; M[5][1] = I1.B.m12; M[5][2] = I1.B.m22;
        M[3][3] = I1.D.xx; M[4][3] = I1.D.xy; M[5][3] = I1.D.xz; M[4][4] = I1.D.yy; M[5][4] = I1.D.yz; M[5][5] = I1.D.zz;
#pragma unroll
        for (int i = 0; i < 6; ++i) M[i][i] += P[L::ROTOR + i];
        chol6_solve(M, b);
#pragma unroll
        for (int i = 0; i < 6; ++i) ddq1[i] = b[i];
    }
    const Sp<T> a1 = {{ddq1[0], ddq1[1], ddq1[2]}, {ddq1[3], ddq1[4], ddq1[5]}};
    // ---- IMU on the trunk (basic_sensors.cc:142-164): data.a[1] = S ddq (bias is zero)
    if (emit_sens && A.imu && lead)
        static_for<0, Tp::NIMU>([&](auto ic) {
            constexpr int s = decltype(ic)::value;
            const SE3<T> fr = ld_se3<T>(P, L::IMU + 12 * s);
            const Sp<T> vf = actinv_motion(fr, v1);
            Sp<T> af = actinv_motion(fr, a1);
            af.l = af.l + cross(vf.a, vf.l);
            const V3<T> gl = tmul(fr.R, tmul(liM1.R, g));
            const V3<T> acc3 = af.l - gl;
            put6(A.imu, B32, r32, 6 * s, Sp<T>{vf.a, acc3});
        });
    // ---- ABA pass 3 down the chain (AbaForwardStep2)
    JM_REFRESH();
    {
        Sp<T> ap = agf1 + a1;
        static_for<0, N>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            const V3<T> n = LT.v3(s * Q::QJ + Q::J_AXIS);
            const Sp<T> cb = cross_mm(vel[s], Sp<T>{zero3<T>(), vl[s] * n});
            const Sp<T> ag = cb + actinv_motion(liMi[s], ap);
            const T Ua = dot(U[s].l, ag.l) + dot(U[s].a, ag.a);
            const T dd = dinv[s] * (u[s] - Ua);
            ddq[s] = dd;
            ap = {ag.l, ag.a + dd * n};
        });
    }
    {
        bool bad = false;
#pragma unroll
        for (int i = 0; i < 6; ++i) bad |= (ddq1[i] != ddq1[i]);
        static_for<0, N>([&](auto sc) { bad |= (ddq[decltype(sc)::value] != ddq[decltype(sc)::value]); });
        if (bad) status |= JM_LANE_NAN;
    }
}

// Optional RNEA-like extra terms (engine.cc:858-904): joint internal wrenches (data.f) and
// centroidal quantities.  Off the critical path: when requested, the kinematics are re-derived
// from the state here so that nothing of the dynamics evaluation has to stay live for it.
template<class T, class Tp, class X>
JM_DEV void quad_extra_terms(CPtr<T> P, const LimbTable<T> & LT, const BatchArgs<T> & A, long long r, int k,
                             const T * qb, const T * vb, const T * ql, const T * vl, const T * ddq1, const T * ddq)
{
    using L = Layout<Tp>;
    using Q = QLayout<Tp>;
    constexpr int N = Tp::QN;
    const unsigned B32 = (unsigned)A.B, r32 = (unsigned)r;
    const bool lead = (k == 0);
    const V3<T> g = ld_v3<T>(P, L::OPT), gw = ld_v3<T>(P, L::OPT + 3);
    SE3<T> liM1;
    {
        SE3<T> Mj;
        Mj.R = quat_to_matrix(qb[3], qb[4], qb[5], qb[6]);
        Mj.p = {qb[0], qb[1], qb[2]};
        liM1 = Mj;
    }
    const Sp<T> v1 = {{vb[0], vb[1], vb[2]}, {vb[3], vb[4], vb[5]}};
    const Sp<T> a1 = {{ddq1[0], ddq1[1], ddq1[2]}, {ddq1[3], ddq1[4], ddq1[5]}};
    const Sp<T> agf1f = a1 + actinv_motion(liM1, Sp<T>{-g, -gw});
    const RBI<T> Y1 = ld_rbi<T>(P, L::JOINT + 1 * L::JSTRIDE + 12);
    SE3<T> liMi[N];
    Sp<T> vel[N], da[N], dagf[N];
    M3<T> oR = liM1.R;
    V3<T> op = liM1.p;
    {
        Sp<T> vp = v1, ap = a1, agp = agf1f;
        static_for<0, N>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int o = s * Q::QJ;
            T c, sn;
            sincos_(ql[s], &sn, &c);
            const V3<T> n = LT.v3(o + Q::J_AXIS);
            const SE3<T> plc = LT.se3(o + Q::J_PLC);
            liMi[s] = {plc.R * rot_rodrigues(n, c, sn), plc.p};
            const Sp<T> vj = {zero3<T>(), vl[s] * n};
            vel[s] = vj + actinv_motion(liMi[s], vp);
            const Sp<T> aj = cross_mm(vel[s], vj) + Sp<T>{zero3<T>(), ddq[s] * n};
            da[s] = aj + actinv_motion(liMi[s], ap);
            dagf[s] = aj + actinv_motion(liMi[s], agp);
            op = op + oR * liMi[s].p;
            oR = oR * liMi[s].R;
            vp = vel[s]; ap = da[s]; agp = dagf[s];
        });
    }
    Sp<T> fext_last = zero6<T>();
    static_for<0, Tp::QCL>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        const SE3<T> fr = LT.se3(Q::CONTACT + c * Q::QC);
        const T depth = op.z + dot(V3<T>{oR.m20, oR.m21, oR.m22}, fr.p);
        if (depth < T(0))
        {
            const V3<T> vj = vel[N - 1].l + cross(vel[N - 1].a, fr.p);
            const V3<T> fW = contact_law<T, Tp>(P, depth, oR * vj);
            Sp<T> fl;
            fl.l = tmul(oR, fW);
            fl.a = cross(fr.p, fl.l);
            fext_last = fext_last + fl;
        }
    });
    Sp<T> h[N], fB[N], fj[N];
    static_for<0, N>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const RBI<T> Y = LT.rbi(s * Q::QJ + Q::J_RBI);
        h[s] = rbi_mul(Y, vel[s]);
        const Sp<T> vxh = cross_mf(vel[s], h[s]);
        fB[s] = rbi_mul(Y, da[s]) + vxh;
        fj[s] = vxh + rbi_mul(Y, dagf[s]);
        if constexpr (s == N - 1) fj[s] = fj[s] - fext_last;
    });
    static_rfor<1, N>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        fB[s - 1] = fB[s - 1] + act_force(liMi[s], fB[s]);
        h[s - 1] = h[s - 1] + act_force(liMi[s], h[s]);
        fj[s - 1] = fj[s - 1] + act_force(liMi[s], fj[s]);
    });
    const Sp<T> h1l = rbi_mul(Y1, v1);
    const Sp<T> vxh1 = cross_mf(v1, h1l);
    const Sp<T> h1 = h1l + quad_sum6<T, X>(act_force(liMi[0], h[0]));
    const Sp<T> fB1 = rbi_mul(Y1, a1) + vxh1 + quad_sum6<T, X>(act_force(liMi[0], fB[0]));
    const Sp<T> fj1 = vxh1 + rbi_mul(Y1, agf1f) + quad_sum6<T, X>(act_force(liMi[0], fj[0]));
    if (A.joint_forces)
    {
        if (lead) { put6(A.joint_forces, B32, r32, 0, zero6<T>()); put6(A.joint_forces, B32, r32, 6, fj1); }
        static_for<0, N>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            const int j = sel4(k, Tp::limb_joint[0][s], Tp::limb_joint[1][s], Tp::limb_joint[2][s], Tp::limb_joint[3][s]);
            put6(A.joint_forces, B32, r32, 6 * j, fj[s]);
        });
    }
    if (A.centroidal)
    {
        T ms = T(0);
        V3<T> mc = zero3<T>();
        static_rfor<0, N>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            const RBI<T> Y = LT.rbi(s * Q::QJ + Q::J_RBI);
            ms = ms + Y.m;
            mc = mc + Y.m * Y.c;
            mc = liMi[s].R * mc + ms * liMi[s].p;
        });
        const T mt = Y1.m + X::quad_sum(ms);
        const V3<T> mct = Y1.m * Y1.c + V3<T>{X::quad_sum(mc.x), X::quad_sum(mc.y), X::quad_sum(mc.z)};
        const V3<T> c1 = (T(1) / mt) * mct;
        const V3<T> com0 = liM1.R * c1 + liM1.p;
        Sp<T> hg = act_force(liM1, h1), dhg = act_force(liM1, fB1);
        hg.a = hg.a + cross(hg.l, com0);
        dhg.a = dhg.a + cross(dhg.l, com0);
        if (lead)
        {
            A.centroidal[r32] = com0.x; A.centroidal[B32 + r32] = com0.y; A.centroidal[2 * B32 + r32] = com0.z;
            put6(A.centroidal, B32, r32, 3, hg);
            put6(A.centroidal, B32, r32, 9, dhg);
        }
    }
}

// trunk part of pinocchio::integrate (SE(3)); identical in the 4 lanes
template<class T> JM_DEV void integrate_freeflyer(const T * q, const T * d, T * qo)
{
    SE3<T> M0;
    M0.R = quat_to_matrix(q[3], q[4], q[5], q[6]);
    M0.p = {q[0], q[1], q[2]};
    const Sp<T> nu = {{d[0], d[1], d[2]}, {d[3], d[4], d[5]}};
    const SE3<T> M1 = M0 * exp6(nu);
    T x, y, z, ww;
    matrix_to_quat(M1.R, x, y, z, ww);
    const T dp = x * q[3] + y * q[4] + z * q[5] + ww * q[6];
    const T sg = dp < T(0) ? T(-1) : T(1);
    const T n2 = x * x + y * y + z * z + ww * ww;
    const T al = sg * (T(3) - n2) * T(0.5);
    qo[0] = M1.p.x; qo[1] = M1.p.y; qo[2] = M1.p.z;
    qo[3] = x * al; qo[4] = y * al; qo[5] = z * al; qo[6] = ww * al;
}

// one lane of a quad: robot r, limb k. `S` = stage buffer views of this lane.
template<class T, class Tp, class X, int SL, int SB>
JM_DEV void quad_lane_run(const BatchArgs<T> & A, long long r, int k, const T * limb_table, const StageBuf<T, SL, SB> & S)
{
    using Q = QLayout<Tp>;
    using R = QRows<Tp>;
    constexpr int N = Tp::QN;
    const unsigned B32 = (unsigned)A.B, r32 = (unsigned)r;
    CPtr<T> P = (CPtr<T>)A.P;
    const LimbTable<T> LT{limb_table + k * Q::QSTRIDE};
    // per-lane row indices of this limb's joints
    int rq[N], rv[N], rm[N];
    static_for<0, N>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        rq[s] = sel4(k, Tp::idx_q[Tp::limb_joint[0][s]], Tp::idx_q[Tp::limb_joint[1][s]], Tp::idx_q[Tp::limb_joint[2][s]], Tp::idx_q[Tp::limb_joint[3][s]]);
        rv[s] = sel4(k, Tp::idx_v[Tp::limb_joint[0][s]], Tp::idx_v[Tp::limb_joint[1][s]], Tp::idx_v[Tp::limb_joint[2][s]], Tp::idx_v[Tp::limb_joint[3][s]]);
        rm[s] = sel4(k, Tp::limb_motor[0][s], Tp::limb_motor[1][s], Tp::limb_motor[2][s], Tp::limb_motor[3][s]);
    });
    const bool lead = (k == 0);
    int status = 0;
    T qb[7], vb[6], ql[N], vl[N], cmd[N], ddq1[6], ddq[N];
    static_for<0, N>([&](auto sc) { cmd[decltype(sc)::value] = A.command[(unsigned)rm[decltype(sc)::value] * B32 + r32]; });

    auto load_state = [&](const T * qsrc, const T * vsrc) {
#pragma unroll
        for (int i = 0; i < 7; ++i) qb[i] = qsrc[(unsigned)i * B32 + r32];
#pragma unroll
        for (int i = 0; i < 6; ++i) vb[i] = vsrc[(unsigned)i * B32 + r32];
        static_for<0, N>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            ql[s] = qsrc[(unsigned)rq[s] * B32 + r32];
            vl[s] = vsrc[(unsigned)rv[s] * B32 + r32];
        });
    };
    auto store_a = [&](T * dst) {
        if (lead)
        {
#pragma unroll
            for (int i = 0; i < 6; ++i) dst[(unsigned)i * B32 + r32] = ddq1[i];
        }
        static_for<0, N>([&](auto sc) { dst[(unsigned)rv[decltype(sc)::value] * B32 + r32] = ddq[decltype(sc)::value]; });
    };
    auto store_status = [&]() {
        const int st = X::quad_or(status);
        if (A.status && lead) A.status[r32] = st;
    };

    if (A.mode == MODE_RESET)
    {
        if (!A.mask[r32]) return;  // uniform over the quad
        if (lead)
        {
#pragma unroll
            for (int i = 0; i < 7; ++i) A.q[(unsigned)i * B32 + r32] = A.q_init[(unsigned)i * B32 + r32];
#pragma unroll
            for (int i = 0; i < 6; ++i) A.v[(unsigned)i * B32 + r32] = A.v_init[(unsigned)i * B32 + r32];
        }
        static_for<0, N>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            A.q[(unsigned)rq[s] * B32 + r32] = A.q_init[(unsigned)rq[s] * B32 + r32];
            A.v[(unsigned)rv[s] * B32 + r32] = A.v_init[(unsigned)rv[s] * B32 + r32];
        });
    }
    if (A.mode != MODE_STEP)
    {
        if (A.mode == MODE_DYNAMICS) load_state(A.q_in, A.v_in);
        else if (A.mode == MODE_RESET) load_state(A.q_init, A.v_init);
        else load_state(A.q, A.v);
        const bool emit = A.mode != MODE_DYNAMICS;
        quad_eval<T, Tp, X>(P, LT, A, r, k, rv, rm, qb, vb, ql, vl, cmd, emit, true, ddq1, ddq, status);
        if (A.mode == MODE_DYNAMICS)
        {
            store_a(A.a_out);
            return;
        }
        store_a(A.a);
        if (A.joint_forces || A.centroidal)
        {
            JM_REFRESH();
            if (A.mode == MODE_RESET) load_state(A.q_init, A.v_init);
            else load_state(A.q, A.v);
            quad_extra_terms<T, Tp, X>(P, LT, A, r, k, qb, vb, ql, vl, ddq1, ddq);
        }
        store_status();
        return;
    }

    // ---- MODE_STEP: RK4 / Euler state machine; the step state lives in the stage buffer
    const T dt = A.dt;
    const bool rk4 = A.solver == JM_SOLVER_RUNGE_KUTTA_4;
    const int pre = A.command_changed ? 1 : 0;
    const int n_evals = pre + A.n_sub * (rk4 ? 4 : 1);
    {
        bool bad = false;
#pragma unroll
        for (int i = 0; i < 7; ++i) { const T x = A.q[(unsigned)i * B32 + r32]; S.putb(R::Q0B + i, x); bad |= (x != x); }
#pragma unroll
        for (int i = 0; i < 6; ++i)
        {
            const T x = A.v[(unsigned)i * B32 + r32], y = A.a[(unsigned)i * B32 + r32];
            S.putb(R::V0B + i, x); S.putb(R::A0B + i, y);
            bad |= (x != x) || (y != y);
        }
        static_for<0, N>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            const T x = A.q[(unsigned)rq[s] * B32 + r32], y = A.v[(unsigned)rv[s] * B32 + r32], z = A.a[(unsigned)rv[s] * B32 + r32];
            S.putl(R::Q0L + s, x); S.putl(R::V0L + s, y); S.putl(R::A0L + s, z);
            bad |= (x != x) || (y != y) || (z != z);
        });
        if (bad) status |= JM_LANE_NAN;
    }
    // The 4 lanes of a quad execute in lock-step on the GPU, so every lane has read the trunk
    // rows of q/v/a before the lead lane overwrites them at commit time; the host emulation
    // (one thread per lane) needs an explicit rendez-vous for the same guarantee.
    X::sync();
#pragma nounroll
    for (int e = 0; e < n_evals; ++e)
    {
        const int st = (e < pre) ? -1 : (rk4 ? ((e - pre) & 3) : 3);
        const bool last = (e == n_evals - 1);
        JM_REFRESH();
        if (st == -1)
        {
#pragma unroll
            for (int i = 0; i < 7; ++i) qb[i] = S.getb(R::Q0B + i);
#pragma unroll
            for (int i = 0; i < 6; ++i) vb[i] = S.getb(R::V0B + i);
            static_for<0, N>([&](auto sc) { ql[decltype(sc)::value] = S.getl(R::Q0L + decltype(sc)::value); vl[decltype(sc)::value] = S.getl(R::V0L + decltype(sc)::value); });
        }
        else
        {
            const bool first = rk4 ? (st == 0) : true;
            T bw, aw;
            if (rk4)
            {
                bw = (st == 0 || st == 3) ? dt * T(1.0 / 6.0) : dt * T(1.0 / 3.0);
                aw = (st == 2) ? dt : dt * T(0.5);
            }
            else { bw = dt; aw = dt; }
            T incb[6], q0b[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) q0b[i] = S.getb(R::Q0B + i);
#pragma unroll
            for (int i = 0; i < 6; ++i)
            {
                const T v0 = S.getb(R::V0B + i);
                const T kv = first ? v0 : S.getb(R::KVB + i);
                const T ka = first ? S.getb(R::A0B + i) : ddq1[i];
                const T av = (first || !rk4) ? bw * kv : S.getb(R::ACCVB + i) + bw * kv;
                const T aa = (first || !rk4) ? bw * ka : S.getb(R::ACCAB + i) + bw * ka;
                if (st == 3) { incb[i] = av; vb[i] = v0 + aa; }
                else
                {
                    S.putb(R::ACCVB + i, av); S.putb(R::ACCAB + i, aa);
                    incb[i] = aw * kv; vb[i] = v0 + aw * ka; S.putb(R::KVB + i, vb[i]);
                }
            }
            integrate_freeflyer<T>(q0b, incb, qb);
            static_for<0, N>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                const T q0 = S.getl(R::Q0L + s), v0 = S.getl(R::V0L + s);
                const T kv = first ? v0 : S.getl(R::KVL + s);
                const T ka = first ? S.getl(R::A0L + s) : ddq[s];
                const T av = (first || !rk4) ? bw * kv : S.getl(R::ACCVL + s) + bw * kv;
                const T aa = (first || !rk4) ? bw * ka : S.getl(R::ACCAL + s) + bw * ka;
                if (st == 3) { ql[s] = q0 + av; vl[s] = v0 + aa; }
                else
                {
                    S.putl(R::ACCVL + s, av); S.putl(R::ACCAL + s, aa);
                    ql[s] = q0 + aw * kv; vl[s] = v0 + aw * ka; S.putl(R::KVL + s, vl[s]);
                }
            });
            if (st == 3)
            {
                // commit: the new state becomes the start of the next sub-step
#pragma unroll
                for (int i = 0; i < 7; ++i) S.putb(R::Q0B + i, qb[i]);
#pragma unroll
                for (int i = 0; i < 6; ++i) S.putb(R::V0B + i, vb[i]);
                static_for<0, N>([&](auto sc) { S.putl(R::Q0L + decltype(sc)::value, ql[decltype(sc)::value]); S.putl(R::V0L + decltype(sc)::value, vl[decltype(sc)::value]); });
                if (last)
                {
                    if (lead)
                    {
#pragma unroll
                        for (int i = 0; i < 7; ++i) A.q[(unsigned)i * B32 + r32] = qb[i];
#pragma unroll
                        for (int i = 0; i < 6; ++i) A.v[(unsigned)i * B32 + r32] = vb[i];
                    }
                    static_for<0, N>([&](auto sc) {
                        constexpr int s = decltype(sc)::value;
                        A.q[(unsigned)rq[s] * B32 + r32] = ql[s];
                        A.v[(unsigned)rv[s] * B32 + r32] = vl[s];
                    });
                }
            }
        }
        quad_eval<T, Tp, X>(P, LT, A, r, k, rv, rm, qb, vb, ql, vl, cmd, last, A.update_sensors != 0, ddq1, ddq, status);
        if (st == -1 || st == 3)
        {
#pragma unroll
            for (int i = 0; i < 6; ++i) S.putb(R::A0B + i, ddq1[i]);
            static_for<0, N>([&](auto sc) { S.putl(R::A0L + decltype(sc)::value, ddq[decltype(sc)::value]); });
        }
        if (last)
        {
            store_a(A.a);
            store_status();
            if (A.joint_forces || A.centroidal)
            {
                // the committed state sits in the stage buffer
                JM_REFRESH();
#pragma unroll
                for (int i = 0; i < 7; ++i) qb[i] = S.getb(R::Q0B + i);
#pragma unroll
                for (int i = 0; i < 6; ++i) vb[i] = S.getb(R::V0B + i);
                static_for<0, N>([&](auto sc) { ql[decltype(sc)::value] = S.getl(R::Q0L + decltype(sc)::value); vl[decltype(sc)::value] = S.getl(R::V0L + decltype(sc)::value); });
                quad_extra_terms<T, Tp, X>(P, LT, A, r, k, qb, vb, ql, vl, ddq1, ddq);
            }
        }
    }
}

#ifndef JM_HOST_EMU
// quad butterflies with DPP quad_perm (no LDS traffic)
struct DppQuad
{
    template<int CTRL> static __device__ __forceinline__ int mov(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
    template<int CTRL> static __device__ __forceinline__ double perm(double x)
    {
        const int lo = mov<CTRL>(__double2loint(x)), hi = mov<CTRL>(__double2hiint(x));
        return __hiloint2double(hi, lo);
    }
    template<int CTRL> static __device__ __forceinline__ float perm(float x) { return __int_as_float(mov<CTRL>(__float_as_int(x))); }
    template<class T> static __device__ __forceinline__ T quad_sum(T x)
    {
        x = x + perm<0xB1>(x);  // quad_perm [1,0,3,2]
        x = x + perm<0x4E>(x);  // quad_perm [2,3,0,1]
        return x;
    }
    static __device__ __forceinline__ int quad_or(int x)
    {
        x |= mov<0xB1>(x);
        x |= mov<0x4E>(x);
        return x;
    }
    static __device__ __forceinline__ void sync() {}  // lanes of a wave are already in lock-step
};

#ifndef JM_QUAD_WAVES_PER_EU
#define JM_QUAD_WAVES_PER_EU 1
#endif
template<class T, class Tp>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(JM_QUAD_WAVES_PER_EU)))
k_quad(const BatchArgs<T> A)
{
    using Q = QLayout<Tp>;
    __shared__ T table[Q::TABLE];
    __shared__ T stage_l[QRows<Tp>::NL * 64];
    __shared__ T stage_b[QRows<Tp>::NB * 16];
    for (int i = threadIdx.x; i < Q::TABLE; i += 64) table[i] = A.P[Q::OFFSET + i];
    __syncthreads();
    const long long r = (long long)blockIdx.x * 16 + (threadIdx.x >> 2);
    const int k = threadIdx.x & 3;
    if (r >= A.B) return;  // uniform over the quad
    const StageBuf<T, 64, 16> S{stage_l + threadIdx.x, stage_b + (threadIdx.x >> 2), k == 0};
    quad_lane_run<T, Tp, DppQuad, 64, 16>(A, r, k, table, S);
}
#endif
}  // namespace jm
