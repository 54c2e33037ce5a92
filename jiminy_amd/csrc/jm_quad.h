// jm_quad.h -- branch-parallel variant of the per-step physics: FOUR lanes per robot, ABA in the
// coordinates of the root body.
//
// Why four lanes: in float64 the per-robot live state of the three ABA sweeps of an 18..36-DoF
// tree does not fit the registers of one lane; the one-robot-per-lane kernel (jm_kernels.h) turns
// that into kilobytes of scratch traffic per lane (profiles/r01_v1_*).  Here the tree is split
// (jiminy_amd/codegen.py quad_structure) into
//   * four LIMBS = the four longest leaf chains of revolute joints: lane k of a DPP quad owns limb
//     k -- its kinematics, contact points, motors and its part of the ABA sweeps, all in registers;
//     shorter limbs are padded at the tip with mass-less dummy joints so the lanes stay uniform;
//   * the TRUNK TREE = the free-flyer root and the 1-dof joints the limbs hang from (ANYmal: only
//     the root; Atlas: back_bkz/bky/bkx + neck), carried redundantly by the four lanes.
// The only cross-lane traffic is the child->parent reduction of the limbs' articulated inertia and
// bias force into the trunk joints they attach to (27 scalars, two `v_mov_dpp quad_perm`
// butterflies each).  A wave64 advances 16 robots.
//
// Why root-body coordinates: every spatial quantity (velocities, inertias, forces) is expressed
// in the frame of joint 1 instead of each joint's own frame.  Spatial vectors then propagate
// between parent and child by plain addition -- no 6x6 congruence transform of the articulated
// inertia (`internal::SE3actOn`, pinocchio_overload_algorithms.h:163-164: ~200 flops per joint)
// and no `liMi.actInv` in the forward sweeps -- at the price of rotating each body's constant
// inertia into the root frame (~60 flops).  Spatial algebra is frame invariant, so q_dd and the
// efforts are the reference's up to rounding; local-frame outputs (fExternal, joint wrenches,
// sensor data) are rotated back where they are emitted.  The root frame is also the best
// conditioned choice (all lever arms < 1 m), which matters for the float32 mode.
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
// ---------------------------------------------------------------- limb table layout (scalars)
template<class Tp> struct QLayout
{
    // per chain joint: plc 12 | rbi 10 | axis 3 | axis in the parent joint frame (plc.R axis) 3 |
    //                  rotor | qlo qhi | motor 9 | encoder reduction
    static constexpr int J_PLC = 0, J_RBI = 12, J_AXIS = 22, J_AXP = 25, J_ROTOR = 28, J_QLO = 29, J_QHI = 30, J_MOTOR = 31, J_ENC = 40;
    static constexpr int QJ = 41;
    // per contact point: frame 12 | contact index | contact-sensor index | force-sensor relative frame 12
    static constexpr int C_IDX = 12, C_CS = 13, C_FREL = 14;
    static constexpr int QC = Tp::QHAS_FORCE ? 26 : 14;
    static constexpr int CONTACT = Tp::QN * QJ;
    static constexpr int RAW = CONTACT + Tp::QCL * QC;
    static constexpr int QSTRIDE = (RAW % 2 == 0) ? RAW + 1 : RAW;  // odd
    static constexpr int TABLE = 4 * QSTRIDE;
    static constexpr int OFFSET = Layout<Tp>::TOTAL;   // appended to the parameter block
    static constexpr int TOTAL = OFFSET + TABLE;
};

// per-lane select among 4 compile-time constants (folds when they are all equal)
JM_DEV int sel4(int k, int c0, int c1, int c2, int c3) { return k == 0 ? c0 : (k == 1 ? c1 : (k == 2 ? c2 : c3)); }

// compile-time facts about the decomposition
template<class Tp> struct QInfo
{
    static constexpr int N = Tp::QN, NT = Tp::QT;
    static constexpr int NQB = 7 + (NT - 1), NVB = 6 + (NT - 1);  // trunk-tree state rows
    static constexpr bool limb_at(int t)
    {
        for (int k = 0; k < 4; ++k) if (Tp::limb_attach[k] == t) return true;
        return false;
    }
    static constexpr bool child_after(int t, int c)  // trunk joint t has a trunk child with index > c
    {
        for (int i = c + 1; i < NT; ++i) if (Tp::trunk_parent[i] == t) return true;
        return false;
    }
    static constexpr bool has_child(int t) { return limb_at(t) || child_after(t, t); }
    // is trunk child c the first contribution to the accumulators of its parent?
    static constexpr bool first_contrib(int c) { return !limb_at(Tp::trunk_parent[c]) && !child_after(Tp::trunk_parent[c], c); }
    static constexpr bool uniform_attach = Tp::limb_attach[0] == Tp::limb_attach[1] && Tp::limb_attach[0] == Tp::limb_attach[2] && Tp::limb_attach[0] == Tp::limb_attach[3];
    // global row of trunk-tree state element i
    static constexpr int qrow(int i) { return i < 7 ? Tp::idx_q[1] + i : Tp::idx_q[Tp::trunk_joint[i - 6]]; }
    static constexpr int vrow(int i) { return i < 6 ? Tp::idx_v[1] + i : Tp::idx_v[Tp::trunk_joint[i - 5]]; }
};

// Joint s of the four limbs turns about the same coordinate axis of its own frame, up to the sign (Topo::axis_signed:
// ANYmal's joints all turn about +x or -x; dummy padding joints turn about +x): the joint rotation is then a
// two-column update instead of a Rodrigues matrix + 3x3 product, and the axis in root coordinates is a (signed)
// column of the joint's rotation.  -1: mixed or general axes, general form.
template<class Tp> constexpr int limb_axis_uniform(int s)
{
    int ax = -2;
    for (int k = 0; k < 4; ++k)
    {
        const int c = s < Tp::limb_len[k] ? Tp::axis_signed[Tp::limb_joint[k][s]] : 1;
        if (c == 0) return -1;
        const int a = (c < 0 ? -c : c) - 1;
        if (ax == -2) ax = a;
        else if (ax != a) return -1;
    }
    return ax;
}
// some limb has the NEGATIVE coordinate axis at joint s: the sign is read from the limb table (its axis entry is +-1)
template<class Tp> constexpr bool limb_axis_negative(int s)
{
    for (int k = 0; k < 4; ++k)
        if (s < Tp::limb_len[k] && Tp::axis_signed[Tp::limb_joint[k][s]] < 0) return true;
    return false;
}
template<class T> JM_DEV V3<T> mcol(const M3<T> & R, int ax)
{
    return ax == 0 ? V3<T>{R.m00, R.m10, R.m20} : (ax == 1 ? V3<T>{R.m01, R.m11, R.m21} : V3<T>{R.m02, R.m12, R.m22});
}
// R * rot(axis ax, c, s): the two other columns rotate into each other
template<class T> JM_DEV M3<T> mul_rot_axis(const M3<T> & R, int ax, T c, T s)
{
    if (ax == 0) return {R.m00, c * R.m01 + s * R.m02, c * R.m02 - s * R.m01, R.m10, c * R.m11 + s * R.m12, c * R.m12 - s * R.m11,
                         R.m20, c * R.m21 + s * R.m22, c * R.m22 - s * R.m21};
    if (ax == 1) return {c * R.m00 - s * R.m02, R.m01, c * R.m02 + s * R.m00, c * R.m10 - s * R.m12, R.m11, c * R.m12 + s * R.m10,
                         c * R.m20 - s * R.m22, R.m21, c * R.m22 + s * R.m20};
    return {c * R.m00 + s * R.m01, c * R.m01 - s * R.m00, R.m02, c * R.m10 + s * R.m11, c * R.m11 - s * R.m10, R.m12,
            c * R.m20 + s * R.m21, c * R.m21 - s * R.m20, R.m22};
}
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
// joint axis of limb joint s in root coordinates, from the joint's rotation Rs (root coordinates)
template<class T, class Tp, int s> JM_DEV V3<T> limb_axis_root(const LimbTable<T> & LT, const M3<T> & Rs_)
{
    constexpr int ax = limb_axis_uniform<Tp>(s);
    if constexpr (ax >= 0)
    {
        if constexpr (limb_axis_negative<Tp>(s)) return LT(s * QLayout<Tp>::QJ + QLayout<Tp>::J_AXIS + ax) * mcol(Rs_, ax);
        else return mcol(Rs_, ax);
    }
    else return Rs_ * LT.v3(s * QLayout<Tp>::QJ + QLayout<Tp>::J_AXIS);
}

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
template<class T> JM_DEV Sp<T> operator*(T s, Sp<T> a) { return {s * a.l, s * a.a}; }
template<class T> JM_DEV T dot6(Sp<T> a, Sp<T> b) { return dot(a.l, b.l) + dot(a.a, b.a); }
template<class T> JM_DEV Sp<T> mask6(bool on, Sp<T> a)
{
    const T z = T(0);
    return {{on ? a.l.x : z, on ? a.l.y : z, on ? a.l.z : z}, {on ? a.a.x : z, on ? a.a.y : z, on ? a.a.z : z}};
}
template<class T> JM_DEV AI<T> mask_ai(bool on, const AI<T> & Y)
{
    const T z = T(0);
    AI<T> r;
    r.A = {on ? Y.A.xx : z, on ? Y.A.xy : z, on ? Y.A.xz : z, on ? Y.A.yy : z, on ? Y.A.yz : z, on ? Y.A.zz : z};
    r.B = {on ? Y.B.m00 : z, on ? Y.B.m01 : z, on ? Y.B.m02 : z, on ? Y.B.m10 : z, on ? Y.B.m11 : z, on ? Y.B.m12 : z,
           on ? Y.B.m20 : z, on ? Y.B.m21 : z, on ? Y.B.m22 : z};
    r.D = {on ? Y.D.xx : z, on ? Y.D.xy : z, on ? Y.D.xz : z, on ? Y.D.yy : z, on ? Y.D.yz : z, on ? Y.D.zz : z};
    return r;
}
// rigid-body inertia of a body placed at (R, p), expressed in the coordinates (R, p) refer to
template<class T> JM_DEV RBI<T> rbi_placed(const M3<T> & R, V3<T> p, const RBI<T> & Y)
{
    return {Y.m, R * Y.c + p, rot_sym(R, Y.I)};
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
// rows of the stage buffer: step-start state (q0, v0), RK accumulators (sum_i b_i k_i for the
// velocity and acceleration parts) and the velocity of the previous stage (kv).  The previous
// stage's acceleration is the result of the previous evaluation and stays in registers.
template<class Tp> struct QRows
{
    static constexpr int N = Tp::QN, NQB = QInfo<Tp>::NQB, NVB = QInfo<Tp>::NVB;
    static constexpr int Q0B = 0, V0B = NQB, ACCVB = V0B + NVB, ACCAB = ACCVB + NVB, KVB = ACCAB + NVB;  // trunk rows
    static constexpr int Q0L = 0, V0L = N, ACCVL = 2 * N, ACCAL = 3 * N, KVL = 4 * N;  // limb rows
    // Long limbs (register-bound kernels): the evaluation re-reads the stage velocity (= the kv rows)
    // and the held commands from LDS where it needs them instead of keeping them live in VGPRs.
#ifdef JM_QUAD_LEAN
    static constexpr bool LONG = true;   // tuning: register-lean variant regardless of the limb length
#else
    static constexpr bool LONG = N > 4;
#endif
    // The held commands always live in the stage buffer: in registers they are three to seven loop-invariant
    // pairs per lane that the two-waves-per-SIMD build (256 VGPRs) spills to scratch.
    static constexpr bool CMD_LDS = true;
    static constexpr int CMDL = 5 * N, NL = CMD_LDS ? 6 * N : 5 * N;
    static constexpr int CMDB = KVB + NVB, NB = CMD_LDS ? CMDB + Tp::QT : CMDB;
};

// Split constrained stepping (jm_qcon.h: one evaluation = three launches, pre | solve | post): the stage buffer lives in
// HBM between the launches (one tile per wave: [tile][row][64] limb rows, [tile][row][16] trunk rows) and carries, on top
// of the rows above, the state of the evaluation in flight, the acceleration of the previous one, the lane status and the
// constraint context of the evaluation.
template<class Tp> struct QSplitRows
{
    using R = QRows<Tp>;
    static constexpr int N = Tp::QN, NQB = QInfo<Tp>::NQB, NVB = QInfo<Tp>::NVB;
    static constexpr int CURQL = R::NL, CURVL = CURQL + N, DDQL = CURVL + N, STATUSL = DDQL + N, CXL = STATUSL + 1, NCX = 16;
    static constexpr int NL = CXL + NCX;
    static constexpr int CURQB = R::NB, CURVB = CURQB + NQB, DDQB = CURVB + NVB, NB = DDQB + NVB;
    // scalars per tile of 16 robots
    static constexpr int TILE = NL * 64 + NB * 16;
};

// All global accesses use a uniform base pointer + an unsigned 32-bit per-lane element offset, so
// that they select the `saddr + voffset` addressing form instead of pinning a 64-bit VGPR address
// per access (the batch size is bounded accordingly in jm_batch_create).
template<class T> JM_DEV void add6(T * base, unsigned B, unsigned r, unsigned row0, Sp<T> f)
{
    const unsigned o = row0 * B + r;
    base[o] += f.l.x; base[o + B] += f.l.y; base[o + 2 * B] += f.l.z;
    base[o + 3 * B] += f.a.x; base[o + 4 * B] += f.a.y; base[o + 5 * B] += f.a.z;
}
template<class T> JM_DEV void put6(T * base, unsigned B, unsigned r, unsigned row0, Sp<T> f)
{
    const unsigned o = row0 * B + r;
    base[o] = f.l.x; base[o + B] = f.l.y; base[o + 2 * B] = f.l.z;
    base[o + 3 * B] = f.a.x; base[o + 4 * B] = f.a.y; base[o + 5 * B] = f.a.z;
}

// SimpleMotor::computeEffort (basic_motors.cc:83-143); `mp(i)` reads motor parameter i
template<class T, int FL, class MP> JM_DEV void motor_law(MP && mp, T cmd, T vjnt, T & um, T & ut)
{
    const T red = mp(0), elim = mp(1), vlim = mp(2), islope = mp(3);
    const T vmot = red * vjnt;
    um = cmd;
    if constexpr ((FL & JM_MOTOR_EFFORT_LIMIT) != 0)
    {
        T emin = -elim, emax = elim;
        if constexpr ((FL & JM_MOTOR_VELOCITY_LIMIT) != 0)
        {
            // branch-free form of `if (vdelta > 0) { scale the bounds }` (per-lane constants)
            const T vdelta = elim * islope;
            const bool on = vdelta > T(0);
            const T vthr = fmax_(vlim - vdelta, T(0));
            const T inv = rcp_(on ? vlim - vthr : T(1));
            emin *= on ? clamp_((vlim + vmot) * inv, T(0), T(1)) : T(1);
            emax *= on ? clamp_((vlim - vmot) * inv, T(0), T(1)) : T(1);
        }
        um = clamp_(um, emin, emax);
    }
    ut = red * um;
    if constexpr ((FL & JM_MOTOR_FRICTION) != 0)
    {
        const T fds = mp(8);
        if (vjnt > T(0)) ut += mp(4) * vjnt + mp(6) * tanh_(fds * vjnt);
        else ut += mp(5) * vjnt + mp(7) * tanh_(fds * vjnt);
    }
}

// per-lane row indices of this lane's limb
template<class Tp> struct QIdx
{
    int rq[Tp::QN], rv[Tp::QN], rm[Tp::QN];
    bool has[Tp::QN];  // false on the padded (dummy) joints of a short limb
    int nc;            // contact points of this limb
    int attach;        // trunk-tree index the limb hangs from
};
template<class Tp> JM_DEV QIdx<Tp> quad_indices(int k)
{
    QIdx<Tp> ix;
    static_for<0, Tp::QN>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr auto J = [](int kk) { return Tp::limb_joint[kk][s] < 0 ? 1 : Tp::limb_joint[kk][s]; };
        ix.rq[s] = sel4(k, Tp::idx_q[J(0)], Tp::idx_q[J(1)], Tp::idx_q[J(2)], Tp::idx_q[J(3)]);
        ix.rv[s] = sel4(k, Tp::idx_v[J(0)], Tp::idx_v[J(1)], Tp::idx_v[J(2)], Tp::idx_v[J(3)]);
        ix.rm[s] = sel4(k, Tp::limb_motor[0][s] < 0 ? 0 : Tp::limb_motor[0][s], Tp::limb_motor[1][s] < 0 ? 0 : Tp::limb_motor[1][s],
                        Tp::limb_motor[2][s] < 0 ? 0 : Tp::limb_motor[2][s], Tp::limb_motor[3][s] < 0 ? 0 : Tp::limb_motor[3][s]);
        ix.has[s] = sel4(k, s < Tp::limb_len[0], s < Tp::limb_len[1], s < Tp::limb_len[2], s < Tp::limb_len[3]) != 0;
    });
    ix.nc = sel4(k, Tp::limb_ncontact[0], Tp::limb_ncontact[1], Tp::limb_ncontact[2], Tp::limb_ncontact[3]);
    ix.attach = sel4(k, Tp::limb_attach[0], Tp::limb_attach[1], Tp::limb_attach[2], Tp::limb_attach[3]);
    return ix;
}

// ---- body parameters: the constant block / limb table, or per-lane rows (GEN kernels: model randomisation per
// environment, Model::addBiasedToExtendedModel, model.cc:1166-1236).  `c` is the constant value, returned as is
// by the no-op accessor so that the regular kernels compile to exactly what they were.
struct NoModelLane
{
    template<class V> JM_DEV V rbi(int, const V & c) const { return c; }
    template<class V> JM_DEV V plc_p(int, const V & c) const { return c; }
};
template<class T> struct ModelLane
{
    const T * ml;   // [13 * NJ][B] or null
    unsigned B, r;
    JM_DEV RBI<T> rbi(int j, const RBI<T> & c) const
    {
        if (!ml || j < 1) return c;
        const unsigned o = (unsigned)(13 * j) * B + r;
        return {ml[o], {ml[o + B], ml[o + 2 * B], ml[o + 3 * B]},
                S3<T>{ml[o + 4 * B], ml[o + 5 * B], ml[o + 6 * B], ml[o + 7 * B], ml[o + 8 * B], ml[o + 9 * B]}};
    }
    JM_DEV V3<T> plc_p(int j, const V3<T> & c) const
    {
        if (!ml || j < 1) return c;
        const unsigned o = (unsigned)(13 * j + 10) * B + r;
        return {ml[o], ml[o + B], ml[o + 2 * B]};
    }
};

// kinematics of the trunk tree in root coordinates (placements X_t, velocities v_t, joint motion
// subspaces S_t of the 1-dof joints); element 0 is the root itself (X = identity, S unused)
template<class T, class Tp> struct TrunkKin
{
    SE3<T> X[Tp::QT];
    Sp<T> v[Tp::QT];
};
// motion subspace of trunk joint t in root coordinates, re-derived from its placement (cheaper
// than keeping 6 more scalars per joint live across the sweeps)
template<class T, class Tp, int t> JM_DEV Sp<T> trunk_S(CPtr<T> P, const TrunkKin<T, Tp> & K)
{
    constexpr int j = Tp::trunk_joint[t], jt = Tp::jtype[j];
    constexpr int ax = jt_axis(jt);
    V3<T> a;
    const M3<T> & R = K.X[t].R;
    if constexpr (ax == 0) a = {R.m00, R.m10, R.m20};
    else if constexpr (ax == 1) a = {R.m01, R.m11, R.m21};
    else if constexpr (ax == 2) a = {R.m02, R.m12, R.m22};
    else a = R * joint_axis<T, Tp, j>(P);
    if constexpr (jt_is_rev(jt)) return {cross(K.X[t].p, a), a};
    else return {a, zero3<T>()};
}
template<class T, class Tp, class MA = NoModelLane>
JM_DEV void trunk_fk(CPtr<T> P, const T * qb, const T * vb, TrunkKin<T, Tp> & K, int & status, const MA & ma = MA{})
{
    using L = Layout<Tp>;
    K.X[0] = {ident3<T>(), zero3<T>()};
    K.v[0] = {{vb[0], vb[1], vb[2]}, {vb[3], vb[4], vb[5]}};
    static_for<1, Tp::QT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int j = Tp::trunk_joint[t], tp = Tp::trunk_parent[t], jt = Tp::jtype[j];
        const T qj = qb[6 + t], vj = vb[5 + t];
        SE3<T> plc = ld_se3<T>(P, L::JOINT + j * L::JSTRIDE);
        plc.p = ma.plc_p(j, plc.p);
        const V3<T> n = joint_axis<T, Tp, j>(P);
        SE3<T> li;
        if constexpr (jt_is_rev(jt))
        {
            T c, sn;
            sincos_(qj, &sn, &c);
            constexpr int ax = jt_axis(jt);
            if constexpr (ax >= 0) li.R = plc.R * rot_axis<T>(ax, c, sn);
            else li.R = plc.R * rot_rodrigues(n, c, sn);
            li.p = plc.p;
        }
        else
        {
            li.R = plc.R;
            li.p = plc.p + plc.R * (qj * n);
        }
        if constexpr (tp == 0) K.X[t] = li;
        else K.X[t] = K.X[tp] * li;
        K.v[t] = K.v[tp] + vj * trunk_S<T, Tp, t>(P, K);
        constexpr int iq = Tp::idx_q[j];
        if (P[L::QHI + iq] < qj || qj < P[L::QLO + iq]) status |= JM_LANE_OUT_OF_BOUNDS;
    });
}
// ---- quad-distributed storage of per-trunk-joint data ---------------------------------------
// The trunk tree is evaluated identically by the 4 lanes of a quad; keeping its per-joint data
// (placement, velocity, U, 1/D, u) in every lane would cost 4x the registers for nothing.  Joint t
// is therefore KEPT by lane (t-1) & 3 only (slot (t-1) >> 2) and broadcast over the quad with a
// DPP quad_perm when a sweep needs it: 2 v_mov_dpp per scalar instead of a live register pair
// in all four lanes across the limb sweeps.
template<class T, class X, int LANE> JM_DEV V3<T> qbcast(V3<T> a) { return {X::template bcast<LANE>(a.x), X::template bcast<LANE>(a.y), X::template bcast<LANE>(a.z)}; }
template<class T, class X, int LANE> JM_DEV Sp<T> qbcast(Sp<T> a) { return {qbcast<T, X, LANE>(a.l), qbcast<T, X, LANE>(a.a)}; }
template<class T, class X, int LANE> JM_DEV M3<T> qbcast(const M3<T> & a)
{
    return {X::template bcast<LANE>(a.m00), X::template bcast<LANE>(a.m01), X::template bcast<LANE>(a.m02),
            X::template bcast<LANE>(a.m10), X::template bcast<LANE>(a.m11), X::template bcast<LANE>(a.m12),
            X::template bcast<LANE>(a.m20), X::template bcast<LANE>(a.m21), X::template bcast<LANE>(a.m22)};
}
template<class T, class Tp> struct TrunkStore
{
    static constexpr int SLOTS = (Tp::QT + 2) / 4 > 0 ? (Tp::QT + 2) / 4 : 1;  // ceil((QT-1)/4)
    SE3<T> X[SLOTS];
    Sp<T> v[SLOTS];
    Sp<T> U[SLOTS];
    T dinv[SLOTS], u[SLOTS];
    template<int t> static constexpr int lane() { return (t - 1) & 3; }
    template<int t> static constexpr int slot() { return (t - 1) >> 2; }
    // The first write of a slot in program order goes to ALL lanes: a register assigned on one lane
    // only is `undef` to the compiler on the others, and the quad broadcast reads it on all of them
    // (observed: non-deterministic trunk data).  Forward sweeps fill a slot from its lowest joint,
    // the backward sweep from its highest one.
    template<int t> static constexpr bool first_fwd() { return lane<t>() == 0; }
    template<int t> static constexpr bool first_bwd() { return t == Tp::QT - 1 || lane<t>() == 3; }
    template<int t> JM_DEV void put_kin(int k, const SE3<T> & Xt, Sp<T> vt)
    {
        if (first_fwd<t>() || k == lane<t>()) { X[slot<t>()] = Xt; v[slot<t>()] = vt; }
    }
    template<int t> JM_DEV void put_aba(int k, Sp<T> Ut, T di, T uj)
    {
        if (first_bwd<t>() || k == lane<t>()) { U[slot<t>()] = Ut; dinv[slot<t>()] = di; u[slot<t>()] = uj; }
    }
    template<int t, class Xq> JM_DEV void get_kin(SE3<T> & Xt, Sp<T> & vt) const
    {
        Xt = {qbcast<T, Xq, lane<t>()>(X[slot<t>()].R), qbcast<T, Xq, lane<t>()>(X[slot<t>()].p)};
        vt = qbcast<T, Xq, lane<t>()>(v[slot<t>()]);
    }
    template<int t, class Xq> JM_DEV void get_aba(Sp<T> & Ut, T & di, T & uj) const
    {
        Ut = qbcast<T, Xq, lane<t>()>(U[slot<t>()]);
        di = Xq::template bcast<lane<t>()>(dinv[slot<t>()]);
        uj = Xq::template bcast<lane<t>()>(u[slot<t>()]);
    }
};
// motion subspace of a trunk joint placed at X (root coordinates)
template<class T, class Tp, int t> JM_DEV Sp<T> trunk_S_at(CPtr<T> P, const SE3<T> & Xt)
{
    constexpr int j = Tp::trunk_joint[t], jt = Tp::jtype[j];
    constexpr int ax = jt_axis(jt);
    V3<T> a;
    const M3<T> & R = Xt.R;
    if constexpr (ax == 0) a = {R.m00, R.m10, R.m20};
    else if constexpr (ax == 1) a = {R.m01, R.m11, R.m21};
    else if constexpr (ax == 2) a = {R.m02, R.m12, R.m22};
    else a = R * joint_axis<T, Tp, j>(P);
    if constexpr (jt_is_rev(jt)) return {cross(Xt.p, a), a};
    else return {a, zero3<T>()};
}
// placement of trunk joint t relative to its parent joint: jointPlacement * M_j(q)
template<class T, class Tp, int t, class MA = NoModelLane> JM_DEV SE3<T> trunk_liMi(CPtr<T> P, T qj, const MA & ma = MA{})
{
    using L = Layout<Tp>;
    constexpr int j = Tp::trunk_joint[t], jt = Tp::jtype[j];
    SE3<T> plc = ld_se3<T>(P, L::JOINT + j * L::JSTRIDE);
    plc.p = ma.plc_p(j, plc.p);
    const V3<T> n = joint_axis<T, Tp, j>(P);
    SE3<T> li;
    if constexpr (jt_is_rev(jt))
    {
        T c, sn;
        sincos_(qj, &sn, &c);
        constexpr int ax = jt_axis(jt);
        if constexpr (ax >= 0) li.R = plc.R * rot_axis<T>(ax, c, sn);
        else li.R = plc.R * rot_rodrigues(n, c, sn);
        li.p = plc.p;
    }
    else
    {
        li.R = plc.R;
        li.p = plc.p + plc.R * (qj * n);
    }
    return li;
}
// Forward kinematics of the trunk tree into the distributed store; also hands every lane the
// placement / velocity of the trunk joint its limb hangs from (picked up where it is computed).
template<class T, class Tp, class Xq, class MA = NoModelLane>
JM_DEV void trunk_fk_store(CPtr<T> P, int k, const QIdx<Tp> & ix, const T * qb, const T * vb, TrunkStore<T, Tp> & TS,
                           SE3<T> & Xatt, Sp<T> & vatt, int & status, const MA & ma = MA{})
{
    using L = Layout<Tp>;
    const Sp<T> v1 = {{vb[0], vb[1], vb[2]}, {vb[3], vb[4], vb[5]}};
    Xatt = {ident3<T>(), zero3<T>()};
    vatt = v1;
    SE3<T> Xprev = Xatt;
    Sp<T> vprev = v1;
    static_for<1, Tp::QT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int j = Tp::trunk_joint[t], tp = Tp::trunk_parent[t];
        const T qj = qb[6 + t], vj = vb[5 + t];
        const SE3<T> li = trunk_liMi<T, Tp, t, MA>(P, qj, ma);
        SE3<T> Xt;
        Sp<T> vpar;
        if constexpr (tp == 0) { Xt = li; vpar = v1; }
        else
        {
            SE3<T> Xp;
            if constexpr (tp == t - 1) { Xp = Xprev; vpar = vprev; }
            else TS.template get_kin<tp, Xq>(Xp, vpar);
            Xt = Xp * li;
        }
        const Sp<T> vt = vpar + vj * trunk_S_at<T, Tp, t>(P, Xt);
        TS.template put_kin<t>(k, Xt, vt);
        if constexpr (QInfo<Tp>::limb_at(t))
            if (ix.attach == t) { Xatt = Xt; vatt = vt; }
        Xprev = Xt; vprev = vt;
        constexpr int iq = Tp::idx_q[j];
        if (P[L::QHI + iq] < qj || qj < P[L::QLO + iq]) status |= JM_LANE_OUT_OF_BOUNDS;
    });
}

// per-lane pick of the trunk joint this lane's limb hangs from
// Value-level selects: `on ? a : b` scalar by scalar.  (A select between two array ELEMENTS under a
// lane condition gets rewritten by the optimiser into one load with a per-lane index, which turns the
// whole register array into a private-memory array: scratch traffic, and this hipcc has been seen
// to overlay such arrays with live spill slots.)
template<class T> JM_DEV V3<T> msel(bool on, V3<T> a, V3<T> b) { return {on ? a.x : b.x, on ? a.y : b.y, on ? a.z : b.z}; }
template<class T> JM_DEV Sp<T> msel(bool on, Sp<T> a, Sp<T> b) { return {msel(on, a.l, b.l), msel(on, a.a, b.a)}; }
template<class T> JM_DEV M3<T> msel(bool on, const M3<T> & a, const M3<T> & b)
{
    return {on ? a.m00 : b.m00, on ? a.m01 : b.m01, on ? a.m02 : b.m02, on ? a.m10 : b.m10, on ? a.m11 : b.m11,
            on ? a.m12 : b.m12, on ? a.m20 : b.m20, on ? a.m21 : b.m21, on ? a.m22 : b.m22};
}
template<class T> JM_DEV SE3<T> msel(bool on, const SE3<T> & a, const SE3<T> & b) { return {msel(on, a.R, b.R), msel(on, a.p, b.p)}; }
template<class T, class Tp, class V> JM_DEV V pick_attach(int k, const V * arr)
{
    V r = arr[Tp::limb_attach[0]];
    static_for<1, 4>([&](auto kc) {
        constexpr int kk = decltype(kc)::value;
        if constexpr (Tp::limb_attach[kk] != Tp::limb_attach[0])
        {
            const V alt = arr[Tp::limb_attach[kk]];
            r = msel(k == kk, alt, r);
        }
    });
    return r;
}
// limb kinematics in root coordinates. Only the placements are kept per joint; the velocities are
// unwound from the tip in the backward sweep (v_{s-1} = v_s - S_s qd_s) and re-accumulated in the
// forward sweep, which is cheaper than 6 more live scalars per joint.
template<class T, class Tp, class MA = NoModelLane>
JM_DEV void limb_fk(const LimbTable<T> & LT, const QIdx<Tp> & ix, const SE3<T> & Xp, Sp<T> vp, const T * ql, const T * vl,
                    M3<T> * Rs, V3<T> * ps, Sp<T> & vtip, int & status, const MA & ma = MA{}, int k = 0)
{
    using Q = QLayout<Tp>;
    M3<T> Rp = Xp.R;
    V3<T> pp = Xp.p;
    static_for<0, Tp::QN>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int o = s * Q::QJ;
        T c, sn;
        sincos_(ql[s], &sn, &c);
        SE3<T> plc = LT.se3(o + Q::J_PLC);
        if constexpr (!std::is_same<MA, NoModelLane>::value)
            plc.p = ma.plc_p(sel4(k, Tp::limb_joint[0][s], Tp::limb_joint[1][s], Tp::limb_joint[2][s], Tp::limb_joint[3][s]), plc.p);
        ps[s] = pp + Rp * plc.p;
        V3<T> a;   // joint axis in root coordinates
        constexpr int ax = limb_axis_uniform<Tp>(s);
        if constexpr (ax >= 0)
        {
            if constexpr (limb_axis_negative<Tp>(s))
            {
                const T sg = LT(o + Q::J_AXIS + ax);   // +-1: rotation by q about -e = rotation by -q about e
                Rs[s] = mul_rot_axis(Rp * plc.R, ax, c, sg * sn);
                a = sg * mcol(Rs[s], ax);
            }
            else
            {
                Rs[s] = mul_rot_axis(Rp * plc.R, ax, c, sn);
                a = mcol(Rs[s], ax);
            }
        }
        else
        {
            a = Rp * LT.v3(o + Q::J_AXP);
            Rs[s] = Rp * (plc.R * rot_rodrigues(LT.v3(o + Q::J_AXIS), c, sn));
        }
        vp = {vp.l + vl[s] * cross(ps[s], a), vp.a + vl[s] * a};
        if (LT(o + Q::J_QHI) < ql[s] || ql[s] < LT(o + Q::J_QLO)) status |= JM_LANE_OUT_OF_BOUNDS;
        Rp = Rs[s]; pp = ps[s];
    });
    vtip = vp;
    (void)ix;
}

// Long limbs (register-bound kernels): the forward kinematics keep only cos / sin of every joint angle and the tip
// placement; the backward ABA sweep UNWINDS the placements joint by joint on its way to the trunk
// (R_{s-1} = R_s rot(-q_s) plc_s.R^T, p_{s-1} = p_s - R_{s-1} plc_s.p: 48 more operations per joint) instead of 12
// live scalars per joint across the contact and motor code -- 84 doubles per lane for Atlas' arms, which the
// compiler otherwise shuffles through AGPRs and scratch all along the sweep (profiles/r02_atlas_v1_*).
template<class T, class Tp, class MA = NoModelLane>
JM_DEV void limb_fk_tip(const LimbTable<T> & LT, const SE3<T> & Xp, Sp<T> vp, const T * ql, const T * vl,
                        T * cq, T * sq, M3<T> & Rtip, V3<T> & ptip, Sp<T> & vtip, int & status, const MA & ma = MA{}, int k = 0)
{
    using Q = QLayout<Tp>;
    M3<T> Rp = Xp.R;
    V3<T> pp = Xp.p;
    static_for<0, Tp::QN>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int o = s * Q::QJ;
        T c, sn;
        sincos_(ql[s], &sn, &c);
        SE3<T> plc = LT.se3(o + Q::J_PLC);
        if constexpr (!std::is_same<MA, NoModelLane>::value)
            plc.p = ma.plc_p(sel4(k, Tp::limb_joint[0][s], Tp::limb_joint[1][s], Tp::limb_joint[2][s], Tp::limb_joint[3][s]), plc.p);
        const V3<T> pj = pp + Rp * plc.p;
        V3<T> a;
        M3<T> Rj;
        constexpr int ax = limb_axis_uniform<Tp>(s);
        if constexpr (ax >= 0)
        {
            if constexpr (limb_axis_negative<Tp>(s))
            {
                const T sg = LT(o + Q::J_AXIS + ax);
                sn = sg * sn;   // rotation by q about -e = rotation by -q about e: the signed sine is what is kept
                Rj = mul_rot_axis(Rp * plc.R, ax, c, sn);
                a = sg * mcol(Rj, ax);
            }
            else
            {
                Rj = mul_rot_axis(Rp * plc.R, ax, c, sn);
                a = mcol(Rj, ax);
            }
        }
        else
        {
            a = Rp * LT.v3(o + Q::J_AXP);
            Rj = Rp * (plc.R * rot_rodrigues(LT.v3(o + Q::J_AXIS), c, sn));
        }
        cq[s] = c; sq[s] = sn;
        vp = {vp.l + vl[s] * cross(pj, a), vp.a + vl[s] * a};
        if (LT(o + Q::J_QHI) < ql[s] || ql[s] < LT(o + Q::J_QLO)) status |= JM_LANE_OUT_OF_BOUNDS;
        Rp = Rj; pp = pj;
    });
    Rtip = Rp; ptip = pp; vtip = vp;
}
// placement of limb joint s and its axis (root coordinates) from the parent's placement and the kept cos / signed
// sin of the joint angle: the forward ABA sweep of long limbs walks the kinematics a second time instead of keeping
// origin + axis of every joint (6 scalars each) from the backward sweep
template<class T, class Tp, int s, class MA = NoModelLane>
JM_DEV void limb_rewind(const LimbTable<T> & LT, T c, T sn, M3<T> & R, V3<T> & p, V3<T> & a, const MA & ma = MA{}, int k = 0)
{
    using Q = QLayout<Tp>;
    constexpr int o = s * Q::QJ;
    SE3<T> plc = LT.se3(o + Q::J_PLC);
    if constexpr (!std::is_same<MA, NoModelLane>::value)
        plc.p = ma.plc_p(sel4(k, Tp::limb_joint[0][s], Tp::limb_joint[1][s], Tp::limb_joint[2][s], Tp::limb_joint[3][s]), plc.p);
    p = p + R * plc.p;
    constexpr int ax = limb_axis_uniform<Tp>(s);
    if constexpr (ax >= 0)
    {
        R = mul_rot_axis(R * plc.R, ax, c, sn);
        if constexpr (limb_axis_negative<Tp>(s)) a = LT(o + Q::J_AXIS + ax) * mcol(R, ax);
        else a = mcol(R, ax);
    }
    else
    {
        a = R * LT.v3(o + Q::J_AXP);
        R = R * (plc.R * rot_rodrigues(LT.v3(o + Q::J_AXIS), c, sn));
    }
}
// placement of the parent of limb joint s from the placement (R, p) of joint s itself (see limb_fk_tip)
template<class T, class Tp, int s, class MA = NoModelLane>
JM_DEV void limb_unwind(const LimbTable<T> & LT, T c, T sn, M3<T> & R, V3<T> & p, const MA & ma = MA{}, int k = 0)
{
    using Q = QLayout<Tp>;
    constexpr int o = s * Q::QJ;
    SE3<T> plc = LT.se3(o + Q::J_PLC);
    if constexpr (!std::is_same<MA, NoModelLane>::value)
        plc.p = ma.plc_p(sel4(k, Tp::limb_joint[0][s], Tp::limb_joint[1][s], Tp::limb_joint[2][s], Tp::limb_joint[3][s]), plc.p);
    constexpr int ax = limb_axis_uniform<Tp>(s);
    M3<T> Rq;   // parent rotation times the joint placement
    if constexpr (ax >= 0) Rq = mul_rot_axis(R, ax, c, -sn);
    else Rq = R * rot_rodrigues(LT.v3(o + Q::J_AXIS), c, -sn);
    R = mul_bt(Rq, plc.R);
    p = p - R * plc.p;
}

// impulse / profile forces (BatchArgs::applied: world-aligned wrenches at frames of any joint): the wrenches whose frame
// hangs from joint `jq`, summed, as a wrench about the ROOT origin in root coordinates.  (Rj, pj) = placement of that joint in
// root coordinates.  ≙ convertForceGlobalFrameToJoint (utilities/pinocchio.cc:794-809) followed by the joint -> root
// transform: lin = R1^T F, ang = R1^T M + (pj + Rj p_frame) x lin.
template<class T> JM_DEV Sp<T> applied_wrench_on(const BatchArgs<T> & A, int jq, const M3<T> & R1, const M3<T> & Rj, V3<T> pj,
                                                 unsigned B32, unsigned r32)
{
    Sp<T> f = zero6<T>();
    for (int i = 0; i < A.applied_k; ++i)
    {
        if (A.applied_joint[i] != jq) continue;
        const unsigned o = (unsigned)(6 * i) * B32 + r32;
        const V3<T> F = {A.applied[o], A.applied[o + B32], A.applied[o + 2 * B32]};
        const V3<T> M = {A.applied[o + 3 * B32], A.applied[o + 4 * B32], A.applied[o + 5 * B32]};
        const V3<T> p = pj + Rj * V3<T>{A.applied_p[3 * i], A.applied_p[3 * i + 1], A.applied_p[3 * i + 2]};
        const V3<T> fl = tmul(R1, F);
        f.l = f.l + fl;
        f.a = f.a + tmul(R1, M) + cross(p, fl);
    }
    return f;
}
template<class T> JM_DEV Sp<T> applied_root_wrench(const BatchArgs<T> & A, const M3<T> & R1, unsigned B32, unsigned r32)
{
    return applied_wrench_on(A, 1, R1, ident3<T>(), zero3<T>(), B32, r32);
}
// a wrench about the root origin (root coordinates) expressed in the frame of a joint placed at (Rj, pj)
template<class T> JM_DEV Sp<T> wrench_to_joint(const M3<T> & Rj, V3<T> pj, Sp<T> w)
{
    return {tmul(Rj, w.l), tmul(Rj, w.a - cross(pj, w.l))};
}

// ---- where the forces on the contact points come from, and what else acts on the joints ------------
// CFM = 0: spring-damper contact law (the spring-damper contact model: everything below is unused)
// CFM = 1: no contact forces at all (free evaluation of the constraint contact model)
// CFM = 2: contact forces = stored multipliers of the enabled FrameConstraints (constraint contact model:
//          the evaluation that applies the solved multipliers; jm_qcon.h)
// With CFM != 0 joint position bounds are constraints, not a lane failure, `tau_*` are extra joint efforts
// of the dynamics (bound multipliers, signed) and `uemit_*` what is added to the emitted RobotState::u
// (the reference adds the bound multipliers with a plus sign whatever the direction, engine.cc:3786-3790).
template<class T, class Tp> struct QExtra
{
    T tau_l[Tp::QN], tau_b[Tp::QT], uemit_l[Tp::QN], uemit_b[Tp::QT];
    bool motors_on;              // false: Engine::start's first pass, RobotState::u still zero
    const int32_t * flags;       // [NF][B] constraint flags (bit 0 enabled), rows of the contacts start at `nb`
    const T * lam;               // [NR][B] multipliers, rows of the contacts start at `nb`
    int nb;
};
// what a free evaluation leaves behind for the bias-free solves of the constraint model (per lane)
struct NoKeep { static constexpr bool ON = false; };
template<class T, class Tp> struct TrunkStore;
template<class T, class Tp> struct QKeep
{
    static constexpr bool ON = true;
    V3<T> ps[Tp::QN], as[Tp::QN];    // limb joint origins / axes, root coordinates
    Sp<T> Us[Tp::QN];
    T dinv[Tp::QN];
    M3<T> Rt;                        // rotation of the limb tip body (its origin is ps[N-1])
    Sp<T> vtip, atip;                // tip velocity / spatial acceleration (gravity field included)
    M3<T> R1; V3<T> p1;              // root placement in the world (position: as the height map sees it, JM_F_GROUND_OFFSET)
    Sp<T> agf1;                      // gravity field in root coordinates
    T rootL[6][6], rootdinv[6];      // LDL^T factor of the root block (chol6_solve)
};

// a = f(q, v) for one robot spread over a quad; lane k evaluates limb k.
//   qb[NQB], vb[NVB] : trunk-tree configuration / velocity (identical in the 4 lanes)
//   ql[N], vl[N], cmdl[N] : this limb's joints;  cmdb[NT] : commands of the trunk-tree motors
// When EMIT is set -- last evaluation of a step, `start`, `reset` -- the outputs that
// derive from this evaluation (RobotState::u / uMotor / fExternal, contact forces, energies and,
// if `sensors`, the sensor rows) are written right where their inputs are live, so that nothing
// has to stay in registers for a separate output phase.
// DYN = false (with EMIT): the OUTPUT PASS of the step kernels -- kinematics, contact forces, motors and sensors at
// a state whose accelerations `ddqb` / `ddq` are already known (inputs then), without the three ABA sweeps: the
// dynamics evaluations themselves carry no output code at all, so that their register footprint is the same
// everywhere (two waves per SIMD), and the outputs cost about a third of an evaluation once per launch.
template<class T, class Tp, class X, bool EMIT, class SB, int CFM = 0, class KEEP = NoKeep, bool GEN = false, bool DYN = true>
JM_DEV void quad_eval(CPtr<T> P, const LimbTable<T> & LT, const BatchArgs<T> & A, unsigned r, int k, const QIdx<Tp> & ix,
                      const SB & S_, const T * qb, const T * vb_, const T * ql, const T * vl_, const T * cmdb_, const T * cmdl_,
                      bool sensors, T * ddqb, T * ddq, int & status, const QExtra<T, Tp> * ex = nullptr, KEEP * keep = nullptr,
                      TrunkStore<T, Tp> * ts_out = nullptr)
{
    // velocities / commands: registers for short limbs, re-read from the stage buffer for long ones
    using RW = QRows<Tp>;
    auto vlq = [&](int s) -> T { if constexpr (RW::LONG) return S_.getl(RW::KVL + s); else return vl_[s]; };
    auto vbq = [&](int i) -> T { if constexpr (RW::LONG) return S_.getb(RW::KVB + i); else return vb_[i]; };
    auto cmdlq = [&](int s) -> T { if constexpr (RW::CMD_LDS) return S_.getl(RW::CMDL + s); else return cmdl_[s]; };
    auto cmdbq = [&](int t) -> T { if constexpr (RW::CMD_LDS) return S_.getb(RW::CMDB + t); else return cmdb_[t]; };
    // compile-time: the three non-emitting evaluations of an RK4 step carry no output code at all,
    // which keeps their basic blocks large (LDS reads of the limb table get batched ahead of use)
    constexpr bool emit = EMIT;
    using L = Layout<Tp>;
    using Q = QLayout<Tp>;
    using I = QInfo<Tp>;
    constexpr int N = Tp::QN, NT = Tp::QT;
    constexpr int FL = Tp::motor_flags[0];
    const unsigned B32 = (unsigned)A.B;
    unsigned r32 = r;
    JM_OPAQUE(r32);
    // where the per-lane optional inputs of this lane sit (compact batches of the per-stage adaptive stepper: batch order)
    unsigned Bg = B32, rg = r32;
    if constexpr (GEN)
        if (A.lane_map) { rg = (unsigned)A.lane_map[r32]; Bg = (unsigned)A.B_full; }
    const bool lead = (k == 0);
    const bool emit_sens = emit && sensors;
    const bool want_energy = emit && A.energy;
    // ---- encoders read the state itself (basic_sensors.cc:509-539)
    if constexpr (Tp::QHAS_ENC)
        if (emit_sens && A.encoder)
        {
            static_for<0, N>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                const int si = sel4(k, Tp::limb_enc[0][s], Tp::limb_enc[1][s], Tp::limb_enc[2][s], Tp::limb_enc[3][s]);
                T pos = ql[s], vel = vlq(s);
                if constexpr (Tp::QENC_SIDE == 0)
                {
                    const T red = LT(s * Q::QJ + Q::J_ENC);
                    pos *= red;
                    vel *= red;
                }
                if (ix.has[s])
                {
                    A.encoder[(unsigned)(2 * si) * B32 + r32] = pos;
                    A.encoder[(unsigned)(2 * si + 1) * B32 + r32] = vel;
                }
            });
            if (lead)
                static_for<1, NT>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    constexpr int si = Tp::trunk_enc[t];
                    T pos = qb[6 + t], vel = vbq(5 + t);
                    if constexpr (Tp::QENC_SIDE == 0) { pos *= P[L::ENC + si]; vel *= P[L::ENC + si]; }
                    A.encoder[(unsigned)(2 * si) * B32 + r32] = pos;
                    A.encoder[(unsigned)(2 * si + 1) * B32 + r32] = vel;
                });
        }
    // ---- root (free-flyer at the world origin, checked by jm_model_create) and trunk tree
    const M3<T> R1 = quat_to_matrix(qb[3], qb[4], qb[5], qb[6]);
    const V3<T> p1 = {qb[0], qb[1], qb[2]};
    // root position as the height map sees it: every lane samples the map at its own (x, y) offset (JM_F_GROUND_OFFSET)
    V3<T> p1g = p1;
    if constexpr (GEN)
        if (A.ground_off) { p1g.x += A.ground_off[rg]; p1g.y += A.ground_off[Bg + rg]; }
    const V3<T> g = ld_v3<T>(P, L::OPT), gw = ld_v3<T>(P, L::OPT + 3);
    const Sp<T> v1 = {{vb_[0], vb_[1], vb_[2]}, {vb_[3], vb_[4], vb_[5]}};
    // gravity field in root coordinates (bias v x v = 0 for the free-flyer), taken here so that the root placement
    // is not live across the sweeps (the contact code below is its last reader in the evaluations without outputs)
    const Sp<T> agf1 = actinv_motion(SE3<T>{R1, p1}, Sp<T>{-g, -gw});
    // every lane keeps only "its" trunk joints (see TrunkStore: first write of a slot is unconditional)
    TrunkStore<T, Tp> TS;
#ifdef JM_HOST_EMU
    std::memset(&TS, 0xFF, sizeof(TS));
#endif
    SE3<T> Xatt;
    Sp<T> vatt;
    // body parameters: constants, or (GEN) the per-lane rows when they are bound
    using MA = std::conditional_t<GEN, ModelLane<T>, NoModelLane>;
    MA ma;
    if constexpr (GEN) ma = ModelLane<T>{A.model_lane, Bg, rg};
    auto limb_joint_of = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        return sel4(k, Tp::limb_joint[0][s], Tp::limb_joint[1][s], Tp::limb_joint[2][s], Tp::limb_joint[3][s]);
    };
    (void)limb_joint_of;
    trunk_fk_store<T, Tp, X, MA>(P, k, ix, qb, vb_, TS, Xatt, vatt, status, ma);
    // ---- limb kinematics (long limbs: cos / sin per joint + the tip placement only, see limb_fk_tip)
#ifdef JM_OUTPUT_KEEP_KIN
    constexpr bool UNWIND = QRows<Tp>::LONG;
#else
    // (the output pass always takes the register-lean form: its sweep carries the momentum / wrench accumulators instead)
    constexpr bool UNWIND = QRows<Tp>::LONG || !DYN;
#endif
    constexpr bool REWIND = UNWIND;   // ... and the forward sweep re-derives joint origins / axes (limb_rewind)
    M3<T> Rs[UNWIND ? 1 : N];
    V3<T> ps[N];
    T cq[UNWIND ? N : 1], sq[UNWIND ? N : 1];
    M3<T> Rtip;
    V3<T> ptip;
    Sp<T> vtip;
    if constexpr (UNWIND) limb_fk_tip<T, Tp, MA>(LT, Xatt, vatt, ql, vl_, cq, sq, Rtip, ptip, vtip, status, ma, k);
    else
    {
        limb_fk<T, Tp, MA>(LT, ix, Xatt, vatt, ql, vl_, Rs, ps, vtip, status, ma, k);
        Rtip = Rs[N - 1]; ptip = ps[N - 1];
    }
    if constexpr (CFM != 0) status &= ~JM_LANE_OUT_OF_BOUNDS;  // bounds are constraints there, not failures
    // ---- contact points on the limb tip (engine.cc:3117-3238, 3394-3425)
    Sp<T> fext = zero6<T>();   // total external force on the tip body, root coordinates
    {
        const M3<T> Rt = Rtip;
        const V3<T> pt = ptip;
        const Sp<T> vt = vtip;
        Sp<T> fext_loc = zero6<T>(), fsens = zero6<T>();
        T fmax2 = T(0);
        auto one_contact = [&](int c) {
            const int oc = Q::CONTACT + c * Q::QC;
            const SE3<T> fr = LT.se3(oc);
            const V3<T> pc = Rt * fr.p + pt;
            T depth = p1.z + dot(V3<T>{R1.m20, R1.m21, R1.m22}, pc);
            V3<T> nG = {T(0), T(0), T(1)};
            if constexpr (GEN && CFM == 0)
                if (A.ground_h)
                {
                    // world.groundProfile at the contact point; first-order projection (engine.cc:3138-3145)
                    const V3<T> pW = R1 * pc + p1g;
                    T hG;
                    ground_profile(A, pW.x, pW.y, hG, nG);
                    depth = (pW.z - hG) * nG.z;
                }
            const bool active = c < ix.nc;
            Sp<T> fl = zero6<T>();
            if constexpr (CFM == 0)
            {
                if (active && depth < T(0))
                {
                    const V3<T> vW = R1 * (vt.l + cross(vt.a, pc));
                    V3<T> fW;
                    if constexpr (GEN) fW = contact_law_n<T, Tp>(P, nG, depth, vW, A.friction ? A.friction[rg] : T(-1));
                    else fW = contact_law<T, Tp>(P, depth, vW);
                    const V3<T> fR = tmul(R1, fW);
                    fext.l = fext.l + fR;
                    fext.a = fext.a + cross(pc, fR);
                    fmax2 = fmax_(fmax2, dot(fW, fW));
                    if (emit)
                    {
                        fl.l = tmul(Rt, fR);
                        fl.a = cross(fr.p, fl.l);
                    }
                }
            }
            else if constexpr (CFM == 2)
            {
                // multipliers of the enabled FrameConstraint (x, y, z, torsion about z; world aligned) applied at
                // the contact point: convertForceGlobalFrameToJoint (utilities/pinocchio.cc:794-809)
                (void)depth;
                if (active)
                {
                    const unsigned ci = (unsigned)(int)LT(oc + Q::C_IDX);
                    if (ex->flags[((unsigned)ex->nb + ci) * B32 + r32] & 1)
                    {
                        const unsigned o = ((unsigned)ex->nb + 4u * ci) * B32 + r32;
                        // (multipliers live in the local frame of the ground surface under the point: contact_frame)
                        T dep_;
                        const M3<T> Mc = contact_frame<GEN>(A, R1, p1g, pc, dep_);
                        const V3<T> fR = tmul(Mc, V3<T>{ex->lam[o], ex->lam[o + B32], ex->lam[o + 2 * B32]});
                        const V3<T> tR = ex->lam[o + 3 * B32] * V3<T>{Mc.m20, Mc.m21, Mc.m22};
                        fext.l = fext.l + fR;
                        fext.a = fext.a + cross(pc, fR) + tR;
                        if (emit)
                        {
                            fl.l = tmul(Rt, fR);
                            fl.a = cross(fr.p, fl.l) + tmul(Rt, tR);
                        }
                    }
                }
            }
            else { (void)depth; }
            if (emit && active)
            {
                fext_loc = fext_loc + fl;
                const Sp<T> cf = actinv_force(fr, fl);  // Robot::contactForces_, contact frame
                if (A.contact_forces) put6(A.contact_forces, B32, r32, 6 * (unsigned)(int)LT(oc + Q::C_IDX), cf);
                if constexpr (Tp::QHAS_CS)
                    if (sensors && A.contact)
                    {
                        const unsigned o = 3 * (unsigned)(int)LT(oc + Q::C_CS) * B32 + r32;
                        A.contact[o] = cf.l.x; A.contact[o + B32] = cf.l.y; A.contact[o + 2 * B32] = cf.l.z;
                    }
                if constexpr (Tp::QHAS_FORCE)
                    if (sensors && A.force) fsens = fsens + act_force(LT.se3(oc + Q::C_FREL), cf);
            }
        };
        if constexpr (Tp::QCL <= 2)
            static_for<0, Tp::QCL>([&](auto cc) { one_contact(decltype(cc)::value); });
        else
        {
#pragma nounroll
            for (int c = 0; c < Tp::QCL; ++c) one_contact(c);
        }
        if constexpr (EMIT && CFM == 0)
            if ((A.mode == MODE_START || A.mode == MODE_RESET) && fmax2 > T(1e10)) status |= JM_LANE_FORCE_OVERFLOW;
        if (emit)
        {
            if (A.f_external)
            {
                if (lead)
                {
                    put6(A.f_external, B32, r32, 0, zero6<T>());
                    static_for<0, NT>([&](auto tc) { put6(A.f_external, B32, r32, 6 * Tp::trunk_joint[decltype(tc)::value], zero6<T>()); });
                    if constexpr (GEN)
                        if (A.applied_k > 0) put6(A.f_external, B32, r32, 6 * Tp::trunk_joint[0], applied_root_wrench(A, R1, Bg, rg));
                }
                static_for<0, N>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    const int j = sel4(k, Tp::limb_joint[0][s], Tp::limb_joint[1][s], Tp::limb_joint[2][s], Tp::limb_joint[3][s]);
                    // the tip body of a padded limb is its last real joint (the dummies share its frame)
                    const bool tip = sel4(k, s == Tp::limb_len[0] - 1, s == Tp::limb_len[1] - 1, s == Tp::limb_len[2] - 1, s == Tp::limb_len[3] - 1) != 0;
                    if (ix.has[s]) put6(A.f_external, B32, r32, 6 * (unsigned)j, mask6(tip, fext_loc));
                });
            }
            if constexpr (Tp::QHAS_FORCE)
                if (sensors && A.force)
                {
                    const int si = sel4(k, Tp::limb_force[0], Tp::limb_force[1], Tp::limb_force[2], Tp::limb_force[3]);
                    if (si >= 0) put6(A.force, B32, r32, 6 * (unsigned)si, fsens);
                }
        }
    }
    // ---- motors (one per movable joint, uniform flags; basic_motors.cc:83-143).  Long limbs without output code
    // evaluate each motor where the backward sweep consumes its effort instead of keeping N + NT efforts live across
    // the contact code (MOTORS_IN_SWEEP).
    constexpr bool MOTORS_IN_SWEEP = QRows<Tp>::LONG && !EMIT && DYN;
    T u[N], ut[NT];
    auto limb_motor = [&](auto sc) -> T {
        constexpr int s = decltype(sc)::value;
        constexpr int o = s * Q::QJ + Q::J_MOTOR;
        T um, ue;
        motor_law<T, FL>([&](int i) { return LT(o + i); }, cmdlq(s), vlq(s), um, ue);
        T ueff = ue, ue_out = ue;
        if constexpr (CFM != 0)
        {
            ueff = (ex->motors_on ? ue : T(0)) + ex->tau_l[s];
            ue_out = ue + ex->uemit_l[s];
        }
        if (emit && ix.has[s])
        {
            if (A.u_motor) A.u_motor[(unsigned)ix.rm[s] * B32 + r32] = um;
            if (A.u) A.u[(unsigned)ix.rv[s] * B32 + r32] = ue_out;
            if constexpr (Tp::QHAS_EFF)
                if (sensors && A.effort)
                {
                    const int si = sel4(k, Tp::limb_eff[0][s], Tp::limb_eff[1][s], Tp::limb_eff[2][s], Tp::limb_eff[3][s]);
                    A.effort[(unsigned)si * B32 + r32] = um;
                }
        }
        return ueff;
    };
    auto trunk_motor = [&](auto tc) -> T {
        constexpr int t = decltype(tc)::value;
        constexpr int m = Tp::trunk_motor[t];
        constexpr int o = L::MOTOR + JM_MOTOR_NPARAMS * m;
        T um, ue;
        motor_law<T, FL>([&](int i) { return P[o + i]; }, cmdbq(t), vbq(5 + t), um, ue);
        T ueff = ue, ue_out = ue;
        if constexpr (CFM != 0)
        {
            ueff = (ex->motors_on ? ue : T(0)) + ex->tau_b[t];
            ue_out = ue + ex->uemit_b[t];
        }
        if (emit && lead)
        {
            if (A.u_motor) A.u_motor[(unsigned)m * B32 + r32] = um;
            if (A.u) A.u[(unsigned)Tp::idx_v[Tp::trunk_joint[t]] * B32 + r32] = ue_out;
            if constexpr (Tp::QHAS_EFF)
                if (sensors && A.effort) A.effort[(unsigned)Tp::trunk_eff[t] * B32 + r32] = um;
        }
        return ueff;
    };
    if constexpr (!MOTORS_IN_SWEEP)
    {
        static_for<0, N>([&](auto sc) { u[decltype(sc)::value] = limb_motor(sc); });
        ut[0] = T(0);
        static_for<1, NT>([&](auto tc) { ut[decltype(tc)::value] = trunk_motor(tc); });
    }
    if (emit && A.u && lead)
    {
#pragma unroll
        for (int i = 0; i < 6; ++i) A.u[(unsigned)(Tp::idx_v[1] + i) * B32 + r32] = T(0);
    }
    // ---- ABA pass 2 along the limb, tip -> trunk (AbaBackwardStep), root coordinates
    JM_REFRESH();
    // Forward-sweep inputs per joint: long limbs keep only the joint axis (with ps: 6 scalars) and
    // re-derive S, the parent velocity and the bias acceleration on the way down; short limbs have
    // registers to spare and keep S and c (12 scalars) instead of ~30 extra VALU per joint.
    constexpr bool KEEP_SC = !QRows<Tp>::LONG;
    V3<T> as[N];
    Sp<T> Ss[KEEP_SC ? N : 1], cs[KEEP_SC ? N : 1];
    Sp<T> Us[N];
    T dinv[N];
    AI<T> Ia;
    Sp<T> pa = zero6<T>();
    T kin = T(0), rot = T(0), msum = T(0);
    V3<T> mc = zero3<T>();
    {
        Sp<T> vcur = vtip;
        M3<T> Rcur = Rtip;   // long limbs: placement of the joint being processed, unwound towards the trunk
        V3<T> pcur = ptip;
        static_rfor<0, N>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int o = s * Q::QJ;
            if constexpr (!UNWIND) { Rcur = Rs[s]; pcur = ps[s]; }
            else if constexpr (!REWIND) ps[s] = pcur;
            // (output pass: the bodies are walked once, by the output sweep after the accelerations -- see below)
            if constexpr (!DYN) return;
            const RBI<T> Y = rbi_placed(Rcur, pcur, ma.rbi(GEN ? limb_joint_of(sc) : 0, LT.rbi(o + Q::J_RBI)));
            const V3<T> a = limb_axis_root<T, Tp, s>(LT, Rcur);
            const Sp<T> S = {cross(pcur, a), a};
            Sp<T> f = cross_mf(vcur, rbi_mul(Y, vcur));  // bias force v x* (I v)
            if constexpr (s == N - 1) { f = f - fext; Ia = ai_from_rbi(Y); }
            else { f = f + pa; Ia = ai_from_rbi(Y) + Ia; }
            if constexpr (GEN)
                if (A.applied_k > 0)
                {
                    const Sp<T> w = applied_wrench_on(A, limb_joint_of(sc), R1, Rcur, pcur, Bg, rg);
                    f = f - w;
                    if constexpr (EMIT)   // (evaluations that emit their own outputs: the constraint contact model)
                        if (A.f_external && ix.has[s]) add6(A.f_external, B32, r32, 6u * (unsigned)limb_joint_of(sc), wrench_to_joint(Rcur, pcur, w));
                }
            if (want_energy)
            {
                kin += rbi_vtiv(Y, vcur);
                mc = mc + Y.m * Y.c;
                msum += Y.m;
                rot += LT(o + Q::J_ROTOR) * vlq(s) * vlq(s);
            }
            const Sp<T> vj = vlq(s) * S;
            vcur = vcur - vj;                              // velocity of the parent joint
            const Sp<T> c = cross_mm(vcur, vj);            // bias acceleration v x S qd (S x S = 0)
            if constexpr (MOTORS_IN_SWEEP) u[s] = limb_motor(sc);
            const T uj = u[s] - dot6(S, f);
            const Sp<T> U = ai_mul(Ia, S);
            const T D = dot6(S, U) + LT(o + Q::J_ROTOR);
            const T di = rcp_(D);
            ai_rank1_sub(Ia, U, di);
            const Sp<T> Ya = ai_mul(Ia, c);
            const T ud = uj * di;
            pa = {f.l + Ya.l + ud * U.l, f.a + Ya.a + ud * U.a};
            Us[s] = U; dinv[s] = di; u[s] = uj;
            if constexpr (KEEP_SC) { Ss[s] = S; cs[s] = c; }
            else if constexpr (!REWIND) as[s] = a;
            if constexpr (KEEP::ON) { keep->ps[s] = pcur; keep->as[s] = a; keep->Us[s] = U; keep->dinv[s] = di; }
            if constexpr (UNWIND && s > 0) limb_unwind<T, Tp, s, MA>(LT, cq[s], sq[s], Rcur, pcur, ma, k);
        });
    }
    // ---- child -> parent reduction over the 4 limbs (the only cross-lane step of the dynamics)
    AI<T> accA[NT];
    Sp<T> accF[NT];
#ifdef JM_HOST_EMU
    std::memset(accA, 0xFF, sizeof(accA)); std::memset(accF, 0xFF, sizeof(accF));
#endif
    if constexpr (DYN)
    static_for<0, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (I::limb_at(t))
        {
            if constexpr (I::uniform_attach) { accA[t] = quad_sum_ai<T, X>(Ia); accF[t] = quad_sum6<T, X>(pa); }
            else
            {
                const bool mine = ix.attach == t;
                accA[t] = quad_sum_ai<T, X>(mask_ai(mine, Ia));
                accF[t] = quad_sum6<T, X>(mask6(mine, pa));
            }
        }
    });
    if constexpr (DYN)
    if (want_energy)
    {
        kin = X::quad_sum(kin); rot = X::quad_sum(rot); msum = X::quad_sum(msum);
        mc = {X::quad_sum(mc.x), X::quad_sum(mc.y), X::quad_sum(mc.z)};
    }
    // ---- trunk tree, leaves -> root (identical in the 4 lanes; per-joint data fetched from / kept
    // in the quad-distributed store)
    static_rfor<1, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int j = Tp::trunk_joint[t], tp = Tp::trunk_parent[t];
        SE3<T> Xt;
        Sp<T> vt;
        if constexpr (!DYN) return;
        TS.template get_kin<t, X>(Xt, vt);
        const RBI<T> Y = rbi_placed(Xt.R, Xt.p, ma.rbi(j, ld_rbi<T>(P, L::JOINT + j * L::JSTRIDE + 12)));
        const Sp<T> S = trunk_S_at<T, Tp, t>(P, Xt);
        Sp<T> f = cross_mf(vt, rbi_mul(Y, vt));
        AI<T> It = ai_from_rbi(Y);
        if constexpr (I::has_child(t)) { f = f + accF[t]; It = It + accA[t]; }
        if constexpr (GEN)
            if (A.applied_k > 0)
            {
                const Sp<T> w = applied_wrench_on(A, j, R1, Xt.R, Xt.p, Bg, rg);
                f = f - w;
                if constexpr (EMIT)
                    if (A.f_external && lead) put6(A.f_external, B32, r32, 6 * j, wrench_to_joint(Xt.R, Xt.p, w));
            }
        const Sp<T> vj = vbq(5 + t) * S;
        const Sp<T> c = cross_mm(vt - vj, vj);   // parent velocity x S qd
        if constexpr (MOTORS_IN_SWEEP) ut[t] = trunk_motor(tc);
        const T uj = ut[t] - dot6(S, f);
        const Sp<T> U = ai_mul(It, S);
        const T rotor = P[L::ROTOR + Tp::idx_v[j]];
        const T D = dot6(S, U) + rotor;
        const T di = rcp_(D);
        ai_rank1_sub(It, U, di);
        const Sp<T> Ya = ai_mul(It, c);
        const T ud = uj * di;
        const Sp<T> pt = {f.l + Ya.l + ud * U.l, f.a + Ya.a + ud * U.a};
        if constexpr (I::first_contrib(t)) { accA[tp] = It; accF[tp] = pt; }
        else { accA[tp] = accA[tp] + It; accF[tp] = accF[tp] + pt; }
        TS.template put_aba<t>(k, U, di, uj);
        if (want_energy)
        {
            kin += rbi_vtiv(Y, vt);
            mc = mc + Y.m * Y.c;
            msum += Y.m;
            rot += rotor * vbq(5 + t) * vbq(5 + t);
        }
    });
    // ---- root: u -= S^T f ; (Ia + Im) ddq = u - Ia a_gf (free-flyer calc_aba + pass 3)
    const RBI<T> Y1 = ma.rbi(1, ld_rbi<T>(P, L::JOINT + 1 * L::JSTRIDE + 12));
    // root velocity: long limbs re-read it from the stage buffer (6 scalars less across the sweeps)
    const Sp<T> v1r = {{vbq(0), vbq(1), vbq(2)}, {vbq(3), vbq(4), vbq(5)}};
    if constexpr (DYN)
    {
        AI<T> I1 = ai_from_rbi(Y1);
        Sp<T> f1 = cross_mf(v1r, rbi_mul(Y1, v1r));
        if constexpr (I::has_child(0)) { I1 = I1 + accA[0]; f1 = f1 + accF[0]; }
        if constexpr (GEN)
            if (A.applied_k > 0) f1 = f1 - applied_root_wrench(A, R1, Bg, rg);   // impulse / profile forces on the root body
        const Sp<T> Ya = ai_mul(I1, agf1);
        T b[6] = {-f1.l.x - Ya.l.x, -f1.l.y - Ya.l.y, -f1.l.z - Ya.l.z, -f1.a.x - Ya.a.x, -f1.a.y - Ya.a.y, -f1.a.z - Ya.a.z};
        T M[6][6];
        M[0][0] = I1.A.xx; M[1][0] = I1.A.xy; M[2][0] = I1.A.xz; M[1][1] = I1.A.yy; M[2][1] = I1.A.yz; M[2][2] = I1.A.zz;
        M[3][0] = I1.B.m00; M[3][1] = I1.B.m10; M[3][2] = I1.B.m20;
        M[4][0] = I1.B.m01; M[4][1] = I1.B.m11; M[4][2] = I1.B.m21;
        M[5][0] = I1.B.m02; M[5][1] = I1.B.m12; M[5][2] = I1.B.m22;
        M[3][3] = I1.D.xx; M[4][3] = I1.D.xy; M[5][3] = I1.D.xz; M[4][4] = I1.D.yy; M[5][4] = I1.D.yz; M[5][5] = I1.D.zz;
#pragma unroll
        for (int i = 0; i < 6; ++i) M[i][i] += P[L::ROTOR + Tp::idx_v[1] + i];
        chol6_solve(M, b);
#pragma unroll
        for (int i = 0; i < 6; ++i) ddqb[i] = b[i];
        if constexpr (KEEP::ON)
        {
#pragma unroll
            for (int i = 0; i < 6; ++i)
            {
#pragma unroll
                for (int c = 0; c <= i; ++c) keep->rootL[i][c] = M[i][c];
                keep->rootdinv[i] = rcp_(M[i][i]);
            }
            keep->Rt = Rtip; keep->vtip = vtip; keep->R1 = R1; keep->p1 = p1g; keep->agf1 = agf1;
        }
    }
    if constexpr (DYN)
    if (want_energy)
    {
        // Engine::computeExtraTerms energies (engine.cc:806-815, overload .h:54-55)
        kin += rbi_vtiv(Y1, v1r);
        mc = mc + Y1.m * Y1.c;
        msum += Y1.m;
#pragma unroll
        for (int i = 0; i < 6; ++i) rot += P[L::ROTOR + Tp::idx_v[1] + i] * vb_[i] * vb_[i];
        if (lead)
        {
            A.energy[r32] = T(0.5) * kin + T(0.5) * rot;
            A.energy[B32 + r32] = -(dot(R1 * mc, g) + msum * dot(p1, g));
        }
    }
    // ---- ABA pass 3, root -> leaves (AbaForwardStep2): spatial accelerations add up directly
    JM_REFRESH();
    const Sp<T> at0 = agf1 + Sp<T>{{ddqb[0], ddqb[1], ddqb[2]}, {ddqb[3], ddqb[4], ddqb[5]}};
    // IMU on a trunk-tree joint (basic_sensors.cc:142-164), evaluated where the joint's
    // acceleration is produced
    auto imu_at = [&](auto tcst, const SE3<T> & Xt, Sp<T> vt, Sp<T> atg) {
        constexpr int t = decltype(tcst)::value;
        if constexpr (EMIT)
            if (emit_sens && A.imu && lead)
                static_for<0, Tp::NIMU>([&](auto ic) {
                    constexpr int s = decltype(ic)::value;
                    if constexpr (Tp::imu_trunk[s] == t)
                    {
                        const SE3<T> fr = ld_se3<T>(P, L::IMU + 12 * s);
                        const Sp<T> atrue = atg - agf1;  // data.a: spatial acceleration without the gravity field
                        Sp<T> vj, aj;
                        V3<T> gj = tmul(R1, g);
                        if constexpr (t == 0) { vj = vt; aj = atrue; }
                        else { vj = actinv_motion(Xt, vt); aj = actinv_motion(Xt, atrue); gj = tmul(Xt.R, gj); }
                        const Sp<T> vf = actinv_motion(fr, vj);
                        Sp<T> af = actinv_motion(fr, aj);
                        af.l = af.l + cross(vf.a, vf.l);
                        const V3<T> acc3 = af.l - tmul(fr.R, gj);
                        put6(A.imu, B32, r32, 6 * s, Sp<T>{vf.a, acc3});
                    }
                });
    };
    imu_at(std::integral_constant<int, 0>{}, SE3<T>{ident3<T>(), zero3<T>()}, v1r, at0);
    Sp<T> ap = at0;   // acceleration / velocity (/ placement) of the trunk joint this lane's limb hangs from
    Sp<T> vp = v1r;
    M3<T> Rw = ident3<T>();
    V3<T> pw = zero3<T>();
    // accelerations of trunk joints with a non-adjacent, non-root parent are re-fetched from
    // the store; chains (the common case) carry them in `aprev`
    Sp<T> atst[TrunkStore<T, Tp>::SLOTS];
    {
        Sp<T> aprev = at0;
        static_for<1, NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int tp = Tp::trunk_parent[t];
            SE3<T> Xt;
            Sp<T> vt, Ut;
            T di, uj;
            TS.template get_kin<t, X>(Xt, vt);
            if constexpr (DYN) TS.template get_aba<t, X>(Ut, di, uj);
            const Sp<T> S = trunk_S_at<T, Tp, t>(P, Xt);
            const Sp<T> vj = vbq(5 + t) * S;
            Sp<T> apar;
            if constexpr (tp == 0) apar = at0;
            else if constexpr (tp == t - 1) apar = aprev;
            else apar = qbcast<T, X, TrunkStore<T, Tp>::template lane<tp>()>(atst[TrunkStore<T, Tp>::template slot<tp>()]);
            const Sp<T> ag = apar + cross_mm(vt - vj, vj);
            T dd;
            if constexpr (DYN) { dd = di * (uj - dot6(Ut, ag)); ddqb[5 + t] = dd; }
            else dd = ddqb[5 + t];
            const Sp<T> att = ag + dd * S;
            if (TrunkStore<T, Tp>::template first_fwd<t>() || k == TrunkStore<T, Tp>::template lane<t>())
                atst[TrunkStore<T, Tp>::template slot<t>()] = att;
            aprev = att;
            if constexpr (I::limb_at(t))
                if (ix.attach == t)
                {
                    ap = att; vp = vt;
                    if constexpr (REWIND) { Rw = Xt.R; pw = Xt.p; }
                }
            imu_at(tc, Xt, vt, att);
        });
    }
    if constexpr (DYN)
    static_for<0, N>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (KEEP_SC)
        {
            const Sp<T> ag = ap + cs[s];
            const T dd = ix.has[s] ? dinv[s] * (u[s] - dot6(Us[s], ag)) : T(0);
            ddq[s] = dd;
            ap = ag + dd * Ss[s];
        }
        else
        {
            V3<T> a_s, p_s;
            if constexpr (REWIND)
            {
                limb_rewind<T, Tp, s, MA>(LT, cq[s], sq[s], Rw, pw, a_s, ma, k);
                p_s = pw;
            }
            else { a_s = as[s]; p_s = ps[s]; }
            const Sp<T> S = {cross(p_s, a_s), a_s};
            const Sp<T> vj = vlq(s) * S;
            const Sp<T> ag = ap + cross_mm(vp, vj);
            const T dd = ix.has[s] ? dinv[s] * (u[s] - dot6(Us[s], ag)) : T(0);  // dummy joints never move
            ddq[s] = dd;
            ap = ag + dd * S;
            vp = vp + vj;
        }
    });
    // ---- output pass: Engine::computeExtraTerms (engine.cc:800-905) in ONE walk over the bodies, tip -> root, at the
    // committed state with the accelerations of the last evaluation: energies (:806-815, overload .h:54-55), body momenta
    // and net body forces (:870-877), the RNEA joint wrenches data.f (:878-887: gravity field included, external forces
    // removed), subtree masses / first moments (:817-832) and the centroidal momentum and its derivative (:900-904);
    // RobotState::fExternal of the joints that carry an applied wrench.  Root coordinates: body quantities add up along
    // the tree, a joint's wrench is rotated into its own frame where it is stored.  Only the running velocity and TRUE
    // spatial acceleration are carried: they are wound up to the tip first and unwound joint by joint on the way back
    // (v_{s-1} = v_s - S qd, a_{s-1} = a_s - v_{s-1} x S qd - S qdd), so that nothing per joint stays live.
    if constexpr (!DYN && EMIT)
    {
        bool applied_out = false;
        if constexpr (GEN) applied_out = A.applied_k > 0 && A.f_external;
        const bool want_rnea = A.joint_forces || A.centroidal;
        if (want_energy || want_rnea || applied_out)
        {
            // Accelerations are carried WITH the gravity field (a_gf = a + agf1, one constant spatial vector for every body
            // in root coordinates): data.f needs Y a_gf.  The net body forces, which only enter the centroidal derivative
            // as their total, follow from  sum_bodies (Y a + v x* h) = f_root + sum f_ext - sum_bodies Y agf1,  and the
            // last sum is (m_tot agf1.l, mc_tot x agf1.l) unless the gravity has an angular part (uniform test).
            const bool has_gw = gw.x != T(0) || gw.y != T(0) || gw.z != T(0);
            // limb: wind the acceleration up to the tip
            Sp<T> acur = ap;
            {
                Sp<T> vrun = vp;
                static_for<0, N>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    V3<T> a_s, p_s;
                    if constexpr (REWIND)
                    {
                        limb_rewind<T, Tp, s, MA>(LT, cq[s], sq[s], Rw, pw, a_s, ma, k);
                        p_s = pw;
                    }
                    else { a_s = limb_axis_root<T, Tp, s>(LT, Rs[s]); p_s = ps[s]; }
                    const Sp<T> S = {cross(p_s, a_s), a_s};
                    const Sp<T> vj = vlq(s) * S;
                    acur = acur + cross_mm(vrun, vj) + ddq[s] * S;
                    vrun = vrun + vj;
                });
            }
            // per lane: momentum, joint wrench (external forces removed), what was removed, gravity term (has_gw only)
            Sp<T> hs = zero6<T>(), fjs = zero6<T>(), fxs = fext, Gs = zero6<T>();
            T ms = T(0), ekin = T(0), erot = T(0);
            V3<T> mcs = zero3<T>();
            {
                Sp<T> vcur = vtip;
                M3<T> Rcur = Rtip;
                V3<T> pcur = ptip;
                static_rfor<0, N>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    constexpr int o = s * Q::QJ;
                    if constexpr (!UNWIND) { Rcur = Rs[s]; pcur = ps[s]; }
                    const RBI<T> Y = rbi_placed(Rcur, pcur, ma.rbi(GEN ? limb_joint_of(sc) : 0, LT.rbi(o + Q::J_RBI)));
                    const V3<T> a = limb_axis_root<T, Tp, s>(LT, Rcur);
                    const Sp<T> S = {cross(pcur, a), a};
                    const Sp<T> h = rbi_mul(Y, vcur);
                    hs = hs + h;
                    fjs = fjs + cross_mf(vcur, h) + rbi_mul(Y, acur);
                    if (has_gw) Gs = Gs + rbi_mul(Y, agf1);
                    if constexpr (s == N - 1) fjs = fjs - fext;
                    if constexpr (GEN)
                        if (A.applied_k > 0)
                        {
                            const Sp<T> w = applied_wrench_on(A, limb_joint_of(sc), R1, Rcur, pcur, Bg, rg);
                            fjs = fjs - w;
                            fxs = fxs + w;
                            if (applied_out && ix.has[s]) add6(A.f_external, B32, r32, 6u * (unsigned)limb_joint_of(sc), wrench_to_joint(Rcur, pcur, w));
                        }
                    ms += Y.m;
                    mcs = mcs + Y.m * Y.c;
                    ekin += dot6(vcur, h);
                    erot += LT(o + Q::J_ROTOR) * vlq(s) * vlq(s);
                    if (A.joint_forces && ix.has[s])
                        put6(A.joint_forces, B32, r32, 6 * (unsigned)limb_joint_of(sc), actinv_force(SE3<T>{Rcur, pcur}, fjs));
                    const Sp<T> vj = vlq(s) * S;
                    vcur = vcur - vj;
                    acur = acur - cross_mm(vcur, vj) - ddq[s] * S;
                    if constexpr (UNWIND && s > 0) limb_unwind<T, Tp, s, MA>(LT, cq[s], sq[s], Rcur, pcur, ma, k);
                });
            }
            // totals over the bodies (plain sums: one coordinate frame); only the joint wrenches follow the tree
            Sp<T> htot = quad_sum6<T, X>(hs), fxtot = quad_sum6<T, X>(fxs), Gtot = zero6<T>();
            if (has_gw) Gtot = quad_sum6<T, X>(Gs);
            T mtot = X::quad_sum(ms);
            V3<T> mctot = {X::quad_sum(mcs.x), X::quad_sum(mcs.y), X::quad_sum(mcs.z)};
            ekin = X::quad_sum(ekin); erot = X::quad_sum(erot);
            Sp<T> fjT[NT];
            static_for<0, NT>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                if constexpr (I::limb_at(t))
                {
                    if constexpr (I::uniform_attach) fjT[t] = quad_sum6<T, X>(fjs);
                    else fjT[t] = quad_sum6<T, X>(mask6(ix.attach == t, fjs));
                }
                else fjT[t] = zero6<T>();
            });
            static_rfor<0, NT>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int j = Tp::trunk_joint[t];
                SE3<T> Xt = {ident3<T>(), zero3<T>()};
                Sp<T> vt = v1r, att = at0;
                RBI<T> Y = ma.rbi(j, ld_rbi<T>(P, L::JOINT + j * L::JSTRIDE + 12));
                if constexpr (t > 0)
                {
                    TS.template get_kin<t, X>(Xt, vt);
                    att = qbcast<T, X, TrunkStore<T, Tp>::template lane<t>()>(atst[TrunkStore<T, Tp>::template slot<t>()]);
                    Y = rbi_placed(Xt.R, Xt.p, Y);
                }
                const Sp<T> h = rbi_mul(Y, vt);
                htot = htot + h;
                fjT[t] = fjT[t] + cross_mf(vt, h) + rbi_mul(Y, att);
                if (has_gw) Gtot = Gtot + rbi_mul(Y, agf1);
                if constexpr (GEN)
                    if (A.applied_k > 0)
                    {
                        Sp<T> w;
                        if constexpr (t == 0) w = applied_root_wrench(A, R1, Bg, rg);
                        else
                        {
                            w = applied_wrench_on(A, j, R1, Xt.R, Xt.p, Bg, rg);
                            if (applied_out && lead) put6(A.f_external, B32, r32, 6 * j, wrench_to_joint(Xt.R, Xt.p, w));
                        }
                        fjT[t] = fjT[t] - w;
                        fxtot = fxtot + w;
                    }
                mtot += Y.m;
                mctot = mctot + Y.m * Y.c;
                ekin += dot6(vt, h);
                if constexpr (t > 0) erot += P[L::ROTOR + Tp::idx_v[j]] * vbq(5 + t) * vbq(5 + t);
                if (A.joint_forces && lead)
                {
                    if constexpr (t > 0) put6(A.joint_forces, B32, r32, 6 * j, actinv_force(Xt, fjT[t]));
                    else put6(A.joint_forces, B32, r32, 6 * j, fjT[t]);
                }
                if constexpr (t > 0) fjT[Tp::trunk_parent[t]] = fjT[Tp::trunk_parent[t]] + fjT[t];
            });
            if (lead)
            {
                if (A.joint_forces) put6(A.joint_forces, B32, r32, 0, zero6<T>());
                if (want_energy)
                {
#pragma unroll
                    for (int i = 0; i < 6; ++i) erot += P[L::ROTOR + Tp::idx_v[1] + i] * vb_[i] * vb_[i];
                    A.energy[r32] = T(0.5) * ekin + T(0.5) * erot;
                    A.energy[B32 + r32] = -(dot(R1 * mctot, g) + mtot * dot(p1, g));
                }
                if (A.centroidal)
                {
                    if (!has_gw) Gtot = {mtot * agf1.l, cross(mctot, agf1.l)};
                    const Sp<T> fBtot = fjT[0] + fxtot - Gtot;
                    const SE3<T> M1 = {R1, p1};
                    const V3<T> c1 = rcp_(mtot) * mctot;
                    const V3<T> com0 = R1 * c1 + p1;
                    Sp<T> hg = act_force(M1, htot), dhg = act_force(M1, fBtot);
                    hg.a = hg.a + cross(hg.l, com0);
                    dhg.a = dhg.a + cross(dhg.l, com0);
                    A.centroidal[r32] = com0.x; A.centroidal[B32 + r32] = com0.y; A.centroidal[2 * B32 + r32] = com0.z;
                    put6(A.centroidal, B32, r32, 3, hg);
                    put6(A.centroidal, B32, r32, 9, dhg);
                }
            }
        }
    }
    if constexpr (KEEP::ON)
    {
        keep->atip = ap;
        if (ts_out) *ts_out = TS;
    }
    if constexpr (DYN)
    {
        bool bad = false;
        static_for<0, I::NVB>([&](auto ic) { bad |= (ddqb[decltype(ic)::value] != ddqb[decltype(ic)::value]); });
        static_for<0, N>([&](auto sc) { bad |= ix.has[decltype(sc)::value] && (ddq[decltype(sc)::value] != ddq[decltype(sc)::value]); });
        if (bad) status |= JM_LANE_NAN;
    }
}

// Optional RNEA-like extra terms (engine.cc:858-904): joint internal wrenches (data.f) and
// centroidal quantities.  Off the critical path: when requested, the kinematics are re-derived
// from the state here so that nothing of the dynamics evaluation has to stay live for it.
// Root coordinates again: body forces add up along the tree, each joint's wrench is rotated into
// its own frame only when it is stored.
template<class T, class Tp, class X, int CFM = 0, bool GEN = false>
JM_DEV void quad_extra_terms(CPtr<T> P, const LimbTable<T> & LT, const BatchArgs<T> & A, unsigned r32, int k, const QIdx<Tp> & ix,
                             const T * qb, const T * vb, const T * ql, const T * vl, const T * ddqb, const T * ddq,
                             const QExtra<T, Tp> * ex = nullptr)
{
    using L = Layout<Tp>;
    using Q = QLayout<Tp>;
    using I = QInfo<Tp>;
    constexpr int N = Tp::QN, NT = Tp::QT;
    const unsigned B32 = (unsigned)A.B;
    // where the per-lane optional inputs of this lane sit (compact batches of the per-stage adaptive stepper: batch order)
    unsigned Bg = B32, rg = r32;
    if constexpr (GEN)
        if (A.lane_map) { rg = (unsigned)A.lane_map[r32]; Bg = (unsigned)A.B_full; }
    const bool lead = (k == 0);
    const V3<T> g = ld_v3<T>(P, L::OPT), gw = ld_v3<T>(P, L::OPT + 3);
    const M3<T> R1 = quat_to_matrix(qb[3], qb[4], qb[5], qb[6]);
    const V3<T> p1 = {qb[0], qb[1], qb[2]};
    const SE3<T> M1 = {R1, p1};
    // root position as the height map sees it: every lane samples the map at its own (x, y) offset (JM_F_GROUND_OFFSET)
    V3<T> p1g = p1;
    if constexpr (GEN)
        if (A.ground_off) { p1g.x += A.ground_off[rg]; p1g.y += A.ground_off[Bg + rg]; }
    const Sp<T> agf1 = actinv_motion(M1, Sp<T>{-g, -gw});
    int status = 0;
    using MA = std::conditional_t<GEN, ModelLane<T>, NoModelLane>;
    MA ma;
    if constexpr (GEN) ma = ModelLane<T>{A.model_lane, Bg, rg};
    TrunkKin<T, Tp> K;
    trunk_fk<T, Tp, MA>(P, qb, vb, K, status, ma);
    // true spatial accelerations of the trunk tree (ForwardKinematicsAccelerationStep)
    Sp<T> at[NT];
    at[0] = {{ddqb[0], ddqb[1], ddqb[2]}, {ddqb[3], ddqb[4], ddqb[5]}};
    static_for<1, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const Sp<T> S = trunk_S<T, Tp, t>(P, K);
        at[t] = at[Tp::trunk_parent[t]] + cross_mm(K.v[t], vb[5 + t] * S) + ddqb[5 + t] * S;
    });
    M3<T> Rs[N];
    V3<T> ps[N];
    Sp<T> vs[N];
    {
        Sp<T> vtip;
        limb_fk<T, Tp, MA>(LT, ix, pick_attach<T, Tp>(k, K.X), pick_attach<T, Tp>(k, K.v), ql, vl, Rs, ps, vtip, status, ma, k);
        Sp<T> vp = pick_attach<T, Tp>(k, K.v);
        static_for<0, N>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            const V3<T> a = Rs[s] * LT.v3(s * Q::QJ + Q::J_AXIS);
            vs[s] = {vp.l + vl[s] * cross(ps[s], a), vp.a + vl[s] * a};
            vp = vs[s];
        });
    }
    // contact forces on the tip, root coordinates
    Sp<T> fext = zero6<T>();
    {
        auto one_contact = [&](int c) {
            const V3<T> pc = Rs[N - 1] * LT.v3(Q::CONTACT + c * Q::QC + 9) + ps[N - 1];
            if constexpr (CFM == 0)
            {
                T depth = p1.z + dot(V3<T>{R1.m20, R1.m21, R1.m22}, pc);
                V3<T> nG = {T(0), T(0), T(1)};
                if constexpr (GEN)
                    if (A.ground_h)
                    {
                        const V3<T> pW = R1 * pc + p1g;
                        T hG;
                        ground_profile(A, pW.x, pW.y, hG, nG);
                        depth = (pW.z - hG) * nG.z;
                    }
                if (c < ix.nc && depth < T(0))
                {
                    const V3<T> vWc = R1 * (vs[N - 1].l + cross(vs[N - 1].a, pc));
                    V3<T> fWc;
                    if constexpr (GEN) fWc = contact_law_n<T, Tp>(P, nG, depth, vWc, A.friction ? A.friction[rg] : T(-1));
                    else fWc = contact_law<T, Tp>(P, depth, vWc);
                    const V3<T> fR = tmul(R1, fWc);
                    fext.l = fext.l + fR;
                    fext.a = fext.a + cross(pc, fR);
                }
            }
            else if (c < ix.nc)
            {
                // constraint contact model: the multipliers of the enabled contact constraints
                const unsigned ci = (unsigned)(int)LT(Q::CONTACT + c * Q::QC + Q::C_IDX);
                if (ex->flags[((unsigned)ex->nb + ci) * B32 + r32] & 1)
                {
                    const unsigned o = ((unsigned)ex->nb + 4u * ci) * B32 + r32;
                    T dep_;
                    const M3<T> Mc = contact_frame<GEN>(A, R1, p1g, pc, dep_);
                    const V3<T> fR = tmul(Mc, V3<T>{ex->lam[o], ex->lam[o + B32], ex->lam[o + 2 * B32]});
                    fext.l = fext.l + fR;
                    fext.a = fext.a + cross(pc, fR) + ex->lam[o + 3 * B32] * V3<T>{Mc.m20, Mc.m21, Mc.m22};
                }
            }
        };
        if constexpr (Tp::QCL <= 2)
            static_for<0, Tp::QCL>([&](auto cc) { one_contact(decltype(cc)::value); });
        else
        {
#pragma nounroll
            for (int c = 0; c < Tp::QCL; ++c) one_contact(c);
        }
    }
    // limb: body momenta h, net body forces fB (true accelerations), joint wrenches fj (gravity
    // field included, external forces removed), accumulated tip -> base; subtree mass / first moment
    Sp<T> hs = zero6<T>(), fBs = zero6<T>(), fjs = zero6<T>();
    T ms = T(0);
    V3<T> mcs = zero3<T>();
    {
        Sp<T> al[N];
        Sp<T> ap = pick_attach<T, Tp>(k, at);
        static_for<0, N>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            const V3<T> a = Rs[s] * LT.v3(s * Q::QJ + Q::J_AXIS);
            const Sp<T> S = {cross(ps[s], a), a};
            al[s] = ap + cross_mm(vs[s], vl[s] * S) + ddq[s] * S;
            ap = al[s];
        });
        static_rfor<0, N>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            const RBI<T> Y = rbi_placed(Rs[s], ps[s], ma.rbi(GEN ? sel4(k, Tp::limb_joint[0][s], Tp::limb_joint[1][s], Tp::limb_joint[2][s], Tp::limb_joint[3][s]) : 0,
                                                            LT.rbi(s * Q::QJ + Q::J_RBI)));
            const Sp<T> h = rbi_mul(Y, vs[s]);
            const Sp<T> vxh = cross_mf(vs[s], h);
            hs = hs + h;
            fBs = fBs + rbi_mul(Y, al[s]) + vxh;
            fjs = fjs + vxh + rbi_mul(Y, al[s] + agf1);
            if constexpr (s == N - 1) fjs = fjs - fext;
            if constexpr (GEN)
                if (A.applied_k > 0)
                    fjs = fjs - applied_wrench_on(A, sel4(k, Tp::limb_joint[0][s], Tp::limb_joint[1][s], Tp::limb_joint[2][s], Tp::limb_joint[3][s]),
                                                  R1, Rs[s], ps[s], Bg, rg);
            ms += Y.m;
            mcs = mcs + Y.m * Y.c;
            if (A.joint_forces && ix.has[s])
            {
                const int j = sel4(k, Tp::limb_joint[0][s], Tp::limb_joint[1][s], Tp::limb_joint[2][s], Tp::limb_joint[3][s]);
                put6(A.joint_forces, B32, r32, 6 * (unsigned)j, actinv_force(SE3<T>{Rs[s], ps[s]}, fjs));
            }
        });
    }
    // trunk tree
    Sp<T> hT[NT], fBT[NT], fjT[NT];
    T mT[NT];
    V3<T> mcT[NT];
    static_for<0, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (I::limb_at(t))
        {
            const bool mine = I::uniform_attach || ix.attach == t;
            hT[t] = quad_sum6<T, X>(mask6(mine, hs));
            fBT[t] = quad_sum6<T, X>(mask6(mine, fBs));
            fjT[t] = quad_sum6<T, X>(mask6(mine, fjs));
            mT[t] = X::quad_sum(mine ? ms : T(0));
            mcT[t] = {X::quad_sum(mine ? mcs.x : T(0)), X::quad_sum(mine ? mcs.y : T(0)), X::quad_sum(mine ? mcs.z : T(0))};
        }
        else
        {
            hT[t] = zero6<T>(); fBT[t] = zero6<T>(); fjT[t] = zero6<T>();
            mT[t] = T(0); mcT[t] = zero3<T>();
        }
    });
    static_rfor<0, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int j = Tp::trunk_joint[t];
        RBI<T> Y = ma.rbi(j, ld_rbi<T>(P, L::JOINT + j * L::JSTRIDE + 12));
        if constexpr (t > 0) Y = rbi_placed(K.X[t].R, K.X[t].p, Y);
        const Sp<T> h = rbi_mul(Y, K.v[t]);
        const Sp<T> vxh = cross_mf(K.v[t], h);
        hT[t] = hT[t] + h;
        fBT[t] = fBT[t] + rbi_mul(Y, at[t]) + vxh;
        fjT[t] = fjT[t] + vxh + rbi_mul(Y, at[t] + agf1);
        if constexpr (GEN)
            if (A.applied_k > 0)
            {
                if constexpr (t == 0) fjT[0] = fjT[0] - applied_root_wrench(A, R1, Bg, rg);
                else fjT[t] = fjT[t] - applied_wrench_on(A, j, R1, K.X[t].R, K.X[t].p, Bg, rg);
            }
        mT[t] += Y.m;
        mcT[t] = mcT[t] + Y.m * Y.c;
        if (A.joint_forces && lead)
        {
            if constexpr (t > 0) put6(A.joint_forces, B32, r32, 6 * j, actinv_force(K.X[t], fjT[t]));
            else put6(A.joint_forces, B32, r32, 6 * j, fjT[t]);
        }
        if constexpr (t > 0)
        {
            constexpr int tp = Tp::trunk_parent[t];
            hT[tp] = hT[tp] + hT[t]; fBT[tp] = fBT[tp] + fBT[t]; fjT[tp] = fjT[tp] + fjT[t];
            mT[tp] += mT[t]; mcT[tp] = mcT[tp] + mcT[t];
        }
    });
    if (A.joint_forces && lead) put6(A.joint_forces, B32, r32, 0, zero6<T>());
    if (A.centroidal && lead)
    {
        const V3<T> c1 = (T(1) / mT[0]) * mcT[0];
        const V3<T> com0 = R1 * c1 + p1;
        Sp<T> hg = act_force(M1, hT[0]), dhg = act_force(M1, fBT[0]);
        hg.a = hg.a + cross(hg.l, com0);
        dhg.a = dhg.a + cross(dhg.l, com0);
        A.centroidal[r32] = com0.x; A.centroidal[B32 + r32] = com0.y; A.centroidal[2 * B32 + r32] = com0.z;
        put6(A.centroidal, B32, r32, 3, hg);
        put6(A.centroidal, B32, r32, 9, dhg);
    }
}

// root part of pinocchio::integrate (SE(3)); identical in the 4 lanes
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

// constraint contact model on this decomposition (jm_qcon.h)
template<class Tp> constexpr int qcon_first_contact_row()   // = number of bounded joints (ConRows<Tp>::NB)
{
    int n = 0;
    for (int j = 1; j < Tp::NJ; ++j) n += jt_bounded(Tp::jtype[j]) ? 1 : 0;
    return n;
}
template<class Tp> constexpr int qcon_first_lambda_row() { return qcon_first_contact_row<Tp>(); }   // ConRows<Tp>::LAM
template<class T> struct QConArgs;
template<class T> struct QStore;
template<class Tp> struct QSplitRegion;   // (jm_qcon.h)
template<class T, class Tp, class X, class SB, int CAPC, bool GEN, int PH = 0, int INIT = -1>
JM_DEV void quad_eval_con(CPtr<T> P, const LimbTable<T> & LT, const BatchArgs<T> & A, const QConArgs<T> & C, const QStore<T> & V,
                          unsigned r, int k, const QIdx<Tp> & ix, const SB & S_, const T * qb, const T * vb, const T * ql,
                          const T * vl, const T * cmdb, const T * cmdl, bool emit, bool sensors, T * ddqb, T * ddq, int & status,
                          int start_passes);

// one lane of a quad: robot r, limb k. `S` = stage buffer views of this lane.  QCON: every evaluation is the
// constrained one (`C` / `V`: constraint state and the robot's solver region).
// PH (QCON only): 0 = every evaluation of the launch in this kernel; 1 / 2 = split stepping, the part of evaluation
// `C->split_e` before / after the multipliers are solved (k_quad_con_pre / k_quad_con_post, jm_qcon.h).
// INIT (constraint kernels): -1 = `start` / `reset` and the other modes in one kernel, decided at run time (the split kernels);
// 1 = a kernel that only serves MODE_START / MODE_RESET, 0 = one that never does -- the four-pass initialisation with its exact
// solve is a large piece of code whose register pressure otherwise leaks into the step path (round 6)
template<class T, class Tp, class X, int SL, int SB, bool QCON = false, int CAPC = 0, bool GEN = false, int PH = 0, int INIT = -1>
JM_DEV void quad_lane_run(const BatchArgs<T> & A, long long r, int k, const T * limb_table, const StageBuf<T, SL, SB> & S,
                          const QConArgs<T> * C = nullptr, const QStore<T> * V = nullptr)
{
    using SR = QSplitRows<Tp>;
    using Q = QLayout<Tp>;
    using R = QRows<Tp>;
    using I = QInfo<Tp>;
    constexpr int N = Tp::QN, NT = Tp::QT, NQB = I::NQB, NVB = I::NVB;
    const unsigned B32 = (unsigned)A.B, r32 = (unsigned)r;
    CPtr<T> P = (CPtr<T>)A.P;
    const LimbTable<T> LT{limb_table + k * Q::QSTRIDE};
    const QIdx<Tp> ix = quad_indices<Tp>(k);
    const bool lead = (k == 0);
    int status = 0;
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

    // every mode runs through the same evaluation loop (two inlined copies of the dynamics: with
    // and without the output code): `start`, `reset` and `dynamics` are one evaluation at a given
    // state, `step` is the RK4 / Euler state machine. The step state lives in the stage buffer.
    const bool stepping = INIT == 1 ? false : (A.mode == MODE_STEP);
    const T * qsrc = A.q;
    const T * vsrc = A.v;
    if (A.mode == MODE_DYNAMICS) { qsrc = A.q_in; vsrc = A.v_in; }
    if (A.mode == MODE_RESET)
    {
        if (!A.mask[r32])  // uniform over the quad
        {
            // (split form of a reset: the solve kernels run over every robot of the launch -- nothing to solve for this one)
            if constexpr (PH == 1) { if (k == 0) V->hbm[QSplitRegion<Tp>::HDR] = T(0); }
            return;
        }
        qsrc = A.q_init; vsrc = A.v_init;
    }
    const T dt = A.dt;
    const bool rk4 = A.solver == JM_SOLVER_RUNGE_KUTTA_4;
    const int pre = stepping ? (A.command_changed ? 1 : 0) : 1;
    const int n_evals = stepping ? pre + A.n_sub * (rk4 ? 4 : 1) : 1;
    // Stage-buffer invariant at the start of every integrator step: q0/v0 = state, kv = v0,
    // accumulators = 0, ddq(b) registers = a(state).  Every stage update is then the same
    // straight-line code (no per-element `first stage ?` branches around the LDS reads).
    int split_e = 0;
    if constexpr (PH != 0) split_e = C->split_e;
    if (PH == 0 || (PH == 1 && split_e == 0))
    {
        bool bad = false;
        static_for<0, NQB>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const unsigned o = (unsigned)I::qrow(i) * B32 + r32;
            const T x = qsrc[o];
            S.putb(R::Q0B + i, x); bad |= (x != x);
            if (A.mode == MODE_RESET && lead) A.q[o] = x;
        });
        static_for<0, NVB>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const unsigned o = (unsigned)I::vrow(i) * B32 + r32;
            const T x = vsrc[o], y = stepping ? A.a[o] : T(0);
            S.putb(R::V0B + i, x); S.putb(R::KVB + i, x); S.putb(R::ACCVB + i, T(0)); S.putb(R::ACCAB + i, T(0));
            ddqb[i] = y;
            bad |= (x != x) || (y != y);
            if (A.mode == MODE_RESET && lead) A.v[o] = x;
        });
        static_for<0, N>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            T x = T(0), y = T(0), z = T(0);
            if (ix.has[s])
            {
                const unsigned oq = (unsigned)ix.rq[s] * B32 + r32, ov = (unsigned)ix.rv[s] * B32 + r32;
                x = qsrc[oq]; y = vsrc[ov]; z = stepping ? A.a[ov] : T(0);
                if (A.mode == MODE_RESET) { A.q[oq] = x; A.v[ov] = y; }
            }
            S.putl(R::Q0L + s, x); S.putl(R::V0L + s, y); S.putl(R::KVL + s, y); S.putl(R::ACCVL + s, T(0)); S.putl(R::ACCAL + s, T(0));
            ddq[s] = z;
            bad |= (x != x) || (y != y) || (z != z);
        });
        // NaN guard on the incoming state (engine.cc:1737-1747)
        if (bad && stepping) status |= JM_LANE_NAN;
        // (Engine::start in the split form: the later passes carry the status of the earlier ones)
        if constexpr (PH == 1) { if (C->split_pass > 0) status |= (int)S.getl(SR::STATUSL); }
    }
    else
    {
        // split stepping: what the previous launch left in the stage buffer
        status = (int)S.getl(SR::STATUSL);
        if constexpr (PH == 1)
        {
            static_for<0, NVB>([&](auto ic) { ddqb[decltype(ic)::value] = S.getb(SR::DDQB + decltype(ic)::value); });
            static_for<0, N>([&](auto sc) { ddq[decltype(sc)::value] = S.getl(SR::DDQL + decltype(sc)::value); });
        }
        else
        {
            static_for<0, NQB>([&](auto ic) { qb[decltype(ic)::value] = S.getb(SR::CURQB + decltype(ic)::value); });
            static_for<0, NVB>([&](auto ic) { vb[decltype(ic)::value] = S.getb(SR::CURVB + decltype(ic)::value); });
            static_for<0, N>([&](auto sc) {
                ql[decltype(sc)::value] = S.getl(SR::CURQL + decltype(sc)::value);
                vl[decltype(sc)::value] = S.getl(SR::CURVL + decltype(sc)::value);
            });
        }
    }
    // The 4 lanes of a quad execute in lock-step on the GPU, so every lane has read the trunk
    // rows of q/v/a before the lead lane overwrites them at commit time; the host emulation
    // (one thread per lane) needs an explicit rendez-vous for the same guarantee.
    X::sync();
    // the limb table staged by k_quad is first read below: its block barrier sits here, after the
    // state loads were issued, so that the two memory round trips of a wave's prologue overlap
    X::table_ready();
    // state of evaluation e: a(t+) refresh (-1), RK stages 1..3 (0..2), end of step / Euler (3)
    auto advance = [&](int st, bool last, unsigned rr) {
        if (st == -1)
        {
            static_for<0, NQB>([&](auto ic) { qb[decltype(ic)::value] = S.getb(R::Q0B + decltype(ic)::value); });
            static_for<0, NVB>([&](auto ic) { vb[decltype(ic)::value] = S.getb(R::V0B + decltype(ic)::value); });
            static_for<0, N>([&](auto sc) { ql[decltype(sc)::value] = S.getl(R::Q0L + decltype(sc)::value); vl[decltype(sc)::value] = S.getl(R::V0L + decltype(sc)::value); });
        }
        else
        {
            // RK4 tableau (runge_kutta4_stepper.h:12-23): b = 1/6 1/3 1/3 1/6, A(i, i-1) = 1/2 1/2 1;
            // explicit Euler = a single "final" stage with b = 1.  (kv, ka) = derivative of the
            // previous stage; the increments are summed in the tangent space and applied once
            // from the step-start configuration (abstract_runge_kutta_stepper.cc:51-56).
            // (step size through a per-iteration opaque copy: otherwise dt/6, dt/3 and dt/2 are hoisted out of the
            // evaluation loop as three more live register pairs)
            T dtl = dt;
            JM_OPAQUE(dtl);
            const T bw = dtl * (rk4 ? ((st == 0 || st == 3) ? T(1.0 / 6.0) : T(1.0 / 3.0)) : T(1));
            const T aw = dtl * ((st == 2) ? T(1) : T(0.5));
            T incb[NVB], q0b[NQB], v0b[NVB], incl[N], v0l[N];
            static_for<0, NQB>([&](auto ic) { q0b[decltype(ic)::value] = S.getb(R::Q0B + decltype(ic)::value); });
            static_for<0, NVB>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                v0b[i] = S.getb(R::V0B + i);
                const T kv = S.getb(R::KVB + i);
                incb[i] = S.getb(R::ACCVB + i) + bw * kv;   // sum b_i kv_i so far
                vb[i] = S.getb(R::ACCAB + i) + bw * ddqb[i];  // sum b_i ka_i so far
            });
            static_for<0, N>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                v0l[s] = S.getl(R::V0L + s);
                const T kv = S.getl(R::KVL + s);
                incl[s] = S.getl(R::ACCVL + s) + bw * kv;
                vl[s] = S.getl(R::ACCAL + s) + bw * ddq[s];
                ql[s] = S.getl(R::Q0L + s);
            });
            if (st != 3)
            {
                // intermediate stage: store the accumulators, state = x0 (+) A(i, i-1) dt k_{i-1}
                static_for<0, NVB>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    S.putb(R::ACCVB + i, incb[i]); S.putb(R::ACCAB + i, vb[i]);
                    incb[i] = aw * S.getb(R::KVB + i);
                    vb[i] = v0b[i] + aw * ddqb[i];
                    S.putb(R::KVB + i, vb[i]);
                });
                static_for<0, N>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    S.putl(R::ACCVL + s, incl[s]); S.putl(R::ACCAL + s, vl[s]);
                    incl[s] = aw * S.getl(R::KVL + s);
                    vl[s] = v0l[s] + aw * ddq[s];
                    S.putl(R::KVL + s, vl[s]);
                });
            }
            else
            {
                static_for<0, NVB>([&](auto ic) { vb[decltype(ic)::value] = v0b[decltype(ic)::value] + vb[decltype(ic)::value]; });
                static_for<0, N>([&](auto sc) { vl[decltype(sc)::value] = v0l[decltype(sc)::value] + vl[decltype(sc)::value]; });
            }
            integrate_freeflyer<T>(q0b, incb, qb);
            static_for<1, NT>([&](auto tc) { qb[6 + decltype(tc)::value] = q0b[6 + decltype(tc)::value] + incb[5 + decltype(tc)::value]; });
            static_for<0, N>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                ql[s] = ql[s] + incl[s];
                if (!ix.has[s]) { ql[s] = T(0); vl[s] = T(0); }  // dummy joints never move
            });
            if (st == 3)
            {
                // commit: the new state becomes the start of the next step (invariant restored)
                static_for<0, NQB>([&](auto ic) { S.putb(R::Q0B + decltype(ic)::value, qb[decltype(ic)::value]); });
                static_for<0, NVB>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    S.putb(R::V0B + i, vb[i]); S.putb(R::KVB + i, vb[i]); S.putb(R::ACCVB + i, T(0)); S.putb(R::ACCAB + i, T(0));
                });
                static_for<0, N>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    S.putl(R::Q0L + s, ql[s]); S.putl(R::V0L + s, vl[s]); S.putl(R::KVL + s, vl[s]);
                    S.putl(R::ACCVL + s, T(0)); S.putl(R::ACCAL + s, T(0));
                });
                if (last)
                {
                    if (lead)
                    {
                        static_for<0, NQB>([&](auto ic) { A.q[(unsigned)I::qrow(decltype(ic)::value) * B32 + rr] = qb[decltype(ic)::value]; });
                        static_for<0, NVB>([&](auto ic) { A.v[(unsigned)I::vrow(decltype(ic)::value) * B32 + rr] = vb[decltype(ic)::value]; });
                    }
                    static_for<0, N>([&](auto sc) {
                        constexpr int s = decltype(sc)::value;
                        if (ix.has[s])
                        {
                            A.q[(unsigned)ix.rq[s] * B32 + rr] = ql[s];
                            A.v[(unsigned)ix.rv[s] * B32 + rr] = vl[s];
                        }
                    });
                }
            }
        }
    };
    unsigned rr = r32;
    if constexpr (QCON)
    {
        // constraint contact model: ONE call site of the (large) constrained evaluation, emitting or not at run time
        const int start_passes = INIT == 1 ? 4 : (INIT == 0 ? (A.mode == MODE_REFRESH ? -1 : 0) :
                                 ((A.mode == MODE_START || A.mode == MODE_RESET) ? 4 : (A.mode == MODE_REFRESH ? -1 : 0)));
        if constexpr (PH != 0)
        {
            // one part of one evaluation of a step launch (MODE_STEP only)
            const int e = split_e;
            const int st = (e < pre) ? -1 : (rk4 ? ((e - pre) & 3) : 3);
            const bool last = e == n_evals - 1;
            if constexpr (PH == 1)
            {
                advance(st, last, rr);
                static_for<0, NQB>([&](auto ic) { S.putb(SR::CURQB + decltype(ic)::value, qb[decltype(ic)::value]); });
                static_for<0, NVB>([&](auto ic) { S.putb(SR::CURVB + decltype(ic)::value, vb[decltype(ic)::value]); });
                static_for<0, N>([&](auto sc) {
                    S.putl(SR::CURQL + decltype(sc)::value, ql[decltype(sc)::value]);
                    S.putl(SR::CURVL + decltype(sc)::value, vl[decltype(sc)::value]);
                });
            }
            // (start / reset in the split form: the four passes of Engine::start are launches of their own, C->split_pass)
            quad_eval_con<T, Tp, X, StageBuf<T, SL, SB>, CAPC, GEN, PH, INIT>(P, LT, A, *C, *V, rr, k, ix, S, qb, vb, ql, vl, cmdb, cmdl, last,
                                    !stepping || A.update_sensors != 0, ddqb, ddq, status,
                                    INIT == 1 ? 4 : (INIT == 0 ? 0 : ((A.mode == MODE_START || A.mode == MODE_RESET) ? 4 : 0)));
            if (PH == 1 || !last)
            {
                S.putl(SR::STATUSL, (T)status);
                if constexpr (PH == 2)
                {
                    static_for<0, NVB>([&](auto ic) { S.putb(SR::DDQB + decltype(ic)::value, ddqb[decltype(ic)::value]); });
                    static_for<0, N>([&](auto sc) { S.putl(SR::DDQL + decltype(sc)::value, ddq[decltype(sc)::value]); });
                }
                return;
            }
        }
        else
        {
#pragma nounroll
        for (int e = 0; e < n_evals; ++e)
        {
            const int st = (e < pre) ? -1 : (rk4 ? ((e - pre) & 3) : 3);
            const bool last = e == n_evals - 1;
            JM_REFRESH();
            rr = r32;
            JM_OPAQUE(rr);
            advance(st, last, rr);
            quad_eval_con<T, Tp, X, StageBuf<T, SL, SB>, CAPC, GEN, 0, INIT>(P, LT, A, *C, *V, rr, k, ix, S, qb, vb, ql, vl, cmdb, cmdl, last && A.mode != MODE_DYNAMICS,
                                    (!stepping && A.mode != MODE_REFRESH) || A.update_sensors != 0, ddqb, ddq, status,
                                    start_passes);
        }
        }
    }
    else
    {
    // Dynamics evaluations carry no output code.  Stepping runs all of them in ONE loop (one copy of the evaluation);
    // the single evaluation of `start` / `reset` / `dynamics` / `refresh` is a second, separately optimised copy --
    // the library self-test (engine._library_self_test) compares what the two produce.  The outputs of the launch
    // (RobotState::u / uMotor / fExternal, contact forces, energies, sensors) are then written by the output pass
    // (quad_eval<EMIT, DYN = false>) at the committed state with the accelerations of the last evaluation.
    if (stepping)
    {
#pragma nounroll
        for (int e = 0; e < n_evals; ++e)
        {
            const int st = (e < pre) ? -1 : (rk4 ? ((e - pre) & 3) : 3);
            JM_REFRESH();
            // per-iteration opaque copy of the lane offset: the addresses of the commit stores must
            // not be hoisted out of the loop (each would pin or spill a 64-bit VGPR pair)
            rr = r32;
            JM_OPAQUE(rr);
            // (long limbs / trunk trees only: re-read the parameter block inside the loop; its scalar loads hoisted out
            // of the loop overflow the SGPR file and come back as v_readlane -- 300 per Atlas evaluation)
            CPtr<T> Pl = P;
            if constexpr (R::LONG && !GEN) JM_OPAQUE_S(Pl);
            advance(st, e == n_evals - 1, rr);
            quad_eval<T, Tp, X, false, StageBuf<T, SL, SB>, 0, NoKeep, GEN>(Pl, LT, A, rr, k, ix, S, qb, vb, ql, vl, cmdb, cmdl, false, ddqb, ddq, status);
        }
    }
    else
    {
        JM_REFRESH();
        rr = r32;
        JM_OPAQUE(rr);
        advance(-1, true, rr);
        quad_eval<T, Tp, X, false, StageBuf<T, SL, SB>, 0, NoKeep, GEN>(P, LT, A, rr, k, ix, S, qb, vb, ql, vl, cmdb, cmdl, false, ddqb, ddq, status);
    }
    if (A.mode != MODE_DYNAMICS)
    {
        JM_REFRESH();
        rr = r32;
        JM_OPAQUE(rr);
        // the (committed) state is re-read from the stage buffer: kept in registers it would be live across the
        // whole evaluation loop
        static_for<0, NQB>([&](auto ic) { qb[decltype(ic)::value] = S.getb(R::Q0B + decltype(ic)::value); });
        static_for<0, NVB>([&](auto ic) { vb[decltype(ic)::value] = S.getb(R::V0B + decltype(ic)::value); });
        static_for<0, N>([&](auto sc) { ql[decltype(sc)::value] = S.getl(R::Q0L + decltype(sc)::value); vl[decltype(sc)::value] = S.getl(R::V0L + decltype(sc)::value); });
        quad_eval<T, Tp, X, true, StageBuf<T, SL, SB>, 0, NoKeep, GEN, false>(P, LT, A, rr, k, ix, S, qb, vb, ql, vl, cmdb, cmdl,
                                  (!stepping && A.mode != MODE_REFRESH) || A.update_sensors != 0, ddqb, ddq, status);
    }
    }
    {
        T * adst = (A.mode == MODE_DYNAMICS) ? A.a_out : A.a;
        if (lead) static_for<0, NVB>([&](auto ic) { adst[(unsigned)I::vrow(decltype(ic)::value) * B32 + rr] = ddqb[decltype(ic)::value]; });
        static_for<0, N>([&](auto sc) { if (ix.has[decltype(sc)::value]) adst[(unsigned)ix.rv[decltype(sc)::value] * B32 + rr] = ddq[decltype(sc)::value]; });
        if (A.mode != MODE_DYNAMICS)
        {
            const int stq = X::quad_or(status);
            if (A.status && lead) A.status[rr] = (A.mode == MODE_REFRESH) ? (A.status[rr] | stq) : stq;
            // (spring-damper kernels: the output pass above has written the extra terms already)
            if constexpr (QCON)
            if (A.joint_forces || A.centroidal)
            {
                // the (committed) state sits in the stage buffer
                JM_REFRESH();
                static_for<0, NQB>([&](auto ic) { qb[decltype(ic)::value] = S.getb(R::Q0B + decltype(ic)::value); });
                static_for<0, NVB>([&](auto ic) { vb[decltype(ic)::value] = S.getb(R::V0B + decltype(ic)::value); });
                static_for<0, N>([&](auto sc) { ql[decltype(sc)::value] = S.getl(R::Q0L + decltype(sc)::value); vl[decltype(sc)::value] = S.getl(R::V0L + decltype(sc)::value); });
                QExtra<T, Tp> ex;
                ex.flags = C->flags;
                ex.nb = qcon_first_contact_row<Tp>();
                ex.lam = C->data + (size_t)qcon_first_lambda_row<Tp>() * B32;
                quad_extra_terms<T, Tp, X, 2, GEN>(P, LT, A, rr, k, ix, qb, vb, ql, vl, ddqb, ddq, &ex);
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
    template<int CTRL> static __device__ __forceinline__ int perm(int x) { return mov<CTRL>(x); }
    template<class T> static __device__ __forceinline__ T quad_sum(T x)
    {
        x = x + perm<0xB1>(x);  // quad_perm [1,0,3,2]
        x = x + perm<0x4E>(x);  // quad_perm [2,3,0,1]
        return x;
    }
    // value of lane LANE of the quad, in all four lanes (quad_perm [L,L,L,L])
    template<int LANE, class T> static __device__ __forceinline__ T bcast(T x) { return perm<LANE * 0x55>(x); }
    static __device__ __forceinline__ int quad_or(int x)
    {
        x |= mov<0xB1>(x);
        x |= mov<0x4E>(x);
        return x;
    }
    // value of the quad lane selected by a quad_perm pattern (0xB1 = [1,0,3,2], 0x4E = [2,3,0,1])
    template<int CTRL, class T> static __device__ __forceinline__ T perm_(T x) { return perm<CTRL>(x); }
    // true in every lane of the WAVE when the predicate holds in any of its lanes (scalar result)
    static __device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
    // max(a, |b|) / max(a, b) in one instruction (no NaN canonicalisation: a NaN operand loses)
    static __device__ __forceinline__ double max_abs(double a, double b)
    {
        double r;
        asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    static __device__ __forceinline__ double max_(double a, double b)
    {
        double r;
        asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    static __device__ __forceinline__ float max_abs(float a, float b) { return fmaxf(a, fabsf(b)); }
    static __device__ __forceinline__ float max_(float a, float b) { return fmaxf(a, b); }
#ifdef JM_SYNC_FENCE   // (experiment of DESIGN.md section 4.7: the quad's hand-over points as compiler-visible memory fences)
    static __device__ __forceinline__ void sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
#else
    static __device__ __forceinline__ void sync() {}  // lanes of a wave are already in lock-step
#endif
    // stores of the quad's lanes to the workspace become visible to the other lanes' later loads
    static __device__ __forceinline__ void fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
    static __device__ __forceinline__ void table_ready() { __syncthreads(); }
};

// waves per block: they share one copy of the limb table. Pick the block size that keeps the most
// waves resident per CU under the 160 KiB of LDS (a CU runs at most 2 such waves per SIMD).
template<class T, class Tp> constexpr int quad_block_waves()
{
#ifdef JM_QUAD_BLOCK_WAVES
    return JM_QUAD_BLOCK_WAVES;   // tuning override
#endif
    constexpr long per_wave = (long)sizeof(T) * (QRows<Tp>::NL * 64 + QRows<Tp>::NB * 16);
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
// Resident waves per SIMD the step kernels are compiled for.  Two (256 registers per lane) when a short-limbed
// robot's stage buffer leaves room for eight waves per CU in the 160 KiB of LDS: the evaluation carries no output
// code and keeps neither the held commands nor polynomial coefficients in registers, so that it fits without
// scratch traffic, and the second wave fills the LDS / dependency stalls of the first (ANYmal, MI355X: 0.158 ->
// 0.125 ms per launch at B = 65 536; profiles/r03_quad_*).  Long limbs (Atlas: 137 KiB of LDS per block) stay at one.
// The per-environment variation kernels (GEN) and robots with a trunk tree spill at 256 registers: one wave.
template<class T, class Tp, int W, bool GEN = false> constexpr int quad_waves_per_eu()
{
#ifdef JM_QUAD_WAVES_PER_EU
    return JM_QUAD_WAVES_PER_EU;   // tuning override
#endif
    constexpr long per_wave = (long)sizeof(T) * (QRows<Tp>::NL * 64 + QRows<Tp>::NB * 16);
    constexpr long block = (long)sizeof(T) * QLayout<Tp>::TABLE + W * per_wave;
    return (!GEN && !QRows<Tp>::LONG && Tp::QT == 1 && Tp::QCL <= 2 && (8 / W) * block <= 160L * 1024) ? 2 : 1;
}
// W = waves per block (they share one copy of the limb table).  The default, quad_block_waves(), suits large
// batches; the library also instantiates W = 1, which it launches for SMALL batches (one GPU's share of a
// sharded batch): four times as many blocks, so that e.g. 4096 Atlas robots = 256 waves land on 256 CUs
// instead of 64.
template<class T, class Tp, int W = quad_block_waves<T, Tp>()>
__global__ void __launch_bounds__((64 * W)) __attribute__((amdgpu_waves_per_eu(quad_waves_per_eu<T, Tp, W>())))
k_quad(const BatchArgs<T> A)
{
    using Q = QLayout<Tp>;
    constexpr int NTH = 64 * W;
    __shared__ T table[Q::TABLE];
    __shared__ T stage_l[QRows<Tp>::NL * NTH];
    __shared__ T stage_b[QRows<Tp>::NB * (NTH / 4)];
#pragma nounroll
    for (int i = threadIdx.x; i < Q::TABLE; i += NTH) table[i] = A.P[Q::OFFSET + i];
    const long long r = (long long)blockIdx.x * (NTH / 4) + (threadIdx.x >> 2);
    const int k = threadIdx.x & 3;
    // early exits are whole quads; waves that still run meet at the barrier inside quad_lane_run
    // (DppQuad::table_ready), terminated waves are not waited for
    if (r >= A.B) return;
    const StageBuf<T, NTH, NTH / 4> S{stage_l + threadIdx.x, stage_b + (threadIdx.x >> 2), k == 0};
    quad_lane_run<T, Tp, DppQuad, NTH, NTH / 4>(A, r, k, table, S);
}
// the same kernel with the optional per-environment variation compiled in (per-lane body parameters, height-map
// ground, impulse / profile forces on the root body): launched instead of k_quad when one of them is bound
template<class T, class Tp>
__global__ void __launch_bounds__((64 * quad_block_waves<T, Tp>())) __attribute__((amdgpu_waves_per_eu(quad_waves_per_eu<T, Tp, quad_block_waves<T, Tp>(), true>())))
k_quad_gen(const BatchArgs<T> A)
{
    using Q = QLayout<Tp>;
    constexpr int NTH = 64 * quad_block_waves<T, Tp>();
    __shared__ T table[Q::TABLE];
    __shared__ T stage_l[QRows<Tp>::NL * NTH];
    __shared__ T stage_b[QRows<Tp>::NB * (NTH / 4)];
#pragma nounroll
    for (int i = threadIdx.x; i < Q::TABLE; i += NTH) table[i] = A.P[Q::OFFSET + i];
    const long long r = (long long)blockIdx.x * (NTH / 4) + (threadIdx.x >> 2);
    const int k = threadIdx.x & 3;
    if (r >= A.B) return;
    const StageBuf<T, NTH, NTH / 4> S{stage_l + threadIdx.x, stage_b + (threadIdx.x >> 2), k == 0};
    quad_lane_run<T, Tp, DppQuad, NTH, NTH / 4, false, 0, true>(A, r, k, table, S);
}
#endif
}  // namespace jm
