// jm_qcon.h -- `contacts.model = "constraint"` on the BRANCH-PARALLEL decomposition (4 lanes per robot):
// joint position bounds and contact points as kinematic constraints, multipliers by projected Gauss-Seidel.
//
// Same reference functions as jm_constraint.h (the one-robot-per-lane version, kept for the trees without
// the 4-limb structure; see its header for the file:line citations):
//   switching          computePositionLimitsForcesAlgo / computeContactDynamicsAtFrame (CONSTRAINT)
//                      core/src/engine/engine.cc:3253-3338, 3145-3193
//   rows, drift        Joint/FrameConstraint::computeJacobianAndDrift   core/src/constraints/{joint,frame}_constraint.cc
//   delassus + solve   PGSSolver::SolveBoxedForwardDynamics             core/src/solver/constraint_solvers.cc:335-448
//   pgs                PGSSolver::ProjectedGaussSeidelSolver / Iter     constraint_solvers.cc:107-333
//   outputs            Engine::computeAcceleration                      engine.cc:3710-3866
//   start passes       Engine::start INIT_ITERATIONS loop               engine.cc:1380-1467
//
// How the work is spread over the quad (lane k owns limb k of the robot, jm_quad.h):
//   * FREE EVALUATION = quad_eval without contact forces: the articulated-body solve in root coordinates;
//     it leaves U, 1/D, joint origins / axes of the limb in the lane's registers (QKeep), the LDL^T factor
//     of the root block in all four lanes and the trunk-tree data in the quad-distributed TrunkStore.
//   * SWITCHING: every lane switches the constraints of its own limb (bounds of its joints, contact points
//     of its tip), the bounds of the trunk-tree joints are handled identically by the four lanes; the
//     active-row mask of the robot is OR-ed over the quad.
//   * DELASSUS MATRIX A = J M^-1 J^T, FOUR COLUMNS AT A TIME: in every round each lane picks the next
//     active row that lives on ITS limb as a column, pushes the unit constraint force down its own limb and
//     through the trunk tree (6 scalars per joint, no inertia work: root coordinates make parent <-> child
//     propagation a plain addition), solves the factorised root block, and sweeps the trunk tree back up.
//     The four root / trunk accelerations are then exchanged with `quad_perm` broadcasts and every lane
//     sweeps ITS limb once per column, writing the entries of the rows it owns.  A robot with m active
//     rows needs ceil(max rows per limb) rounds (ANYmal standing on four feet: 3) instead of m sequential
//     solves of the whole tree.
//   * The robot's m-vectors (multipliers x, right-hand side b, residuals y) and the symmetric matrix in
//     packed lower-triangular form sit in ONE on-chip region per robot (LDS, `QStore`), shared by its four
//     lanes; whatever exceeds the region (robots with more than ~16 active rows) overflows into
//     caller-owned workspace rows in HBM, coalesced over the robots.
//   * PROJECTED GAUSS-SEIDEL in the reference's sweep order; one row update = the four lanes of the quad
//     each summing a quarter of `A.col(i).dot(x)` out of LDS + one quad butterfly; the lead lane projects
//     and stores the multiplier.
//   * RESULT: the multipliers are applied by one more quad_eval whose contact forces ARE the multipliers
//     (CFM = 2) and whose joint efforts carry the bound multipliers: the full articulated-body solve with
//     the constraint forces equals a_free + M^-1 J^T lambda and emits every output (efforts, external
//     wrenches, contact / force / IMU sensors) through the one output path of the spring-damper model.
#pragma once
#include "jm_quad.h"
#include "jm_constraint.h"

#ifndef JM_QCON_MAXM
#define JM_QCON_MAXM 64  // most active constraint rows solved per robot (rows beyond it are dropped and flagged)
#endif

namespace jm
{
template<class T> struct QConArgs
{
    int32_t * flags;      // [NF][B]  bit 0 enabled, bit 1 reversed
    T * data;             // [ND][B]  reference configuration per bounded joint, then lambda per row
    T * ws;               // [QConRows::WS][B] overflow of the per-robot solver region
    const T * friction;   // [B] per-lane contacts.friction, or null
    T kp, kd, torsion, reg, tol_abs, tol_rel;
    int iter_max;
};

template<class Tp> struct QConRows
{
    using R = ConRows<Tp>;
    static constexpr int MAXM = R::NR < JM_QCON_MAXM ? R::NR : JM_QCON_MAXM;
    // per-robot solver region: x | b | y | packed lower triangle of A
    static constexpr int VMAX = 3 * MAXM + MAXM * (MAXM + 1) / 2;
    // workspace rows in HBM for a given on-chip capacity (scalars per robot)
    static constexpr int ws_rows(int cap) { return VMAX > cap ? VMAX - cap : 0; }
    // 64-bit words of the row masks
    static constexpr int NWORDS = ((R::NR + 63) / 64 > 0) ? (R::NR + 63) / 64 : 1;
    // trunk-tree joint t is an ancestor-or-self of trunk-tree joint d
    static constexpr bool trunk_anc(int t, int d)
    {
        for (int i = d; i > 0; i = Tp::trunk_parent[i]) if (i == t) return true;
        return t == 0;
    }
    static constexpr int trunk_row(int t) { return t > 0 ? bound_row_of<Tp>(Tp::trunk_joint[t]) : -1; }
    static constexpr int limb_row(int k, int s) { return Tp::limb_joint[k][s] < 0 ? -1 : bound_row_of<Tp>(Tp::limb_joint[k][s]); }
};

// per-robot solver region: the first `cap` scalars on chip, the rest in the workspace rows
template<class T> struct QStore
{
    T * lds;
    T * hbm;       // already offset by the robot index
    unsigned B;
    int cap;
    JM_DEV T get(int e) const { return e < cap ? lds[e] : hbm[(unsigned)(e - cap) * B]; }
    JM_DEV void put(int e, T x) const { if (e < cap) lds[e] = x; else hbm[(unsigned)(e - cap) * B] = x; }
};
JM_DEV int tri_(int i, int c) { return i >= c ? i * (i + 1) / 2 + c : c * (c + 1) / 2 + i; }

template<class X, int NW> JM_DEV void quad_or_mask(RowMaskN<NW> & m)
{
#pragma unroll
    for (int i = 0; i < NW; ++i)
    {
        const int lo = X::quad_or((int)(unsigned)(m.w[i] & 0xffffffffull));
        const int hi = X::quad_or((int)(unsigned)(m.w[i] >> 32));
        m.w[i] = ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
    }
}

// everything one constrained evaluation shares between its phases (per lane)
template<class T, class Tp> struct QConCtx
{
    using QR = QConRows<Tp>;
    using RowMask = RowMaskN<QR::NWORDS>;
    RowMask act, rev, mine;   // active rows of the robot (solved), reversed bounds, active rows this lane owns
    int m, nb;                // packed sizes: rows, of which joint bounds
    int cb;                   // rows per contact block of the solve (4, or 3 when contacts.torsion == 0)
    bool overflow;            // more active rows than JM_QCON_MAXM: the excess was dropped
};

// ---------------------------------------------------------------- switching
// `init`: Engine::start (every constraint enabled first, engine.cc:1266-1308); `readonly`: MODE_REFRESH.
template<class T, class Tp, class X>
JM_DEV void qcon_switch(CPtr<T> P, const LimbTable<T> & LT, const QConArgs<T> & C, unsigned B32, unsigned r32, int k,
                        const QIdx<Tp> & ix, const T * qb, const T * ql, const QKeep<T, Tp> & K, bool init, bool readonly,
                        QConCtx<T, Tp> & cx)
{
    using L = Layout<Tp>;
    using Q = QLayout<Tp>;
    using R = ConRows<Tp>;
    using QR = QConRows<Tp>;
    constexpr int N = Tp::QN, NT = Tp::QT;
    const bool lead = (k == 0);
    const T eps_tr = P[L::OPT + 9];
    // contacts.torsion == 0: the solver forces the torsion multiplier to zero before it is ever read
    // (constraint_solvers.cc:164-170), so that row is left out of the solve -- except in Engine::start, whose first
    // pass is an exact solve of ALL rows (`ignoreBounds`)
    const bool torsion_zero = C.torsion < Eps<T>::eps && !init;
    typename QConCtx<T, Tp>::RowMask en, rv, own;
    en.clear(); rv.clear(); own.clear();
    auto bound = [&](int row, T qj, T lo, T hi, bool writer, bool mine_) {
        const unsigned of = (unsigned)row * B32 + r32, ol = (unsigned)(R::LAM + row) * B32 + r32;
        int32_t f = init ? 1 : C.flags[of];
        if (!readonly)
        {
            T ref = init ? qj : C.data[of];
            bool clear = init;
            if (hi < qj || qj < lo)
            {
                ref = clamp_(qj, lo, hi);
                f = 1 | (hi < qj ? 2 : 0);
            }
            else if (lo + eps_tr < qj && qj < hi - eps_tr)
            {
                f &= ~1;
                clear = true;  // AbstractConstraintBase::disable
            }
            if (writer)
            {
                C.flags[of] = f;
                C.data[of] = ref;
                if (clear) C.data[ol] = T(0);
            }
        }
        if (f & 1) { en.set(row); if (mine_) own.set(row); }
        if (f & 2) rv.set(row);
    };
    // bounds of this lane's limb joints
    static_for<0, N>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const int row = sel4(k, QR::limb_row(0, s), QR::limb_row(1, s), QR::limb_row(2, s), QR::limb_row(3, s));
        if (row >= 0 && ix.has[s]) bound(row, ql[s], LT(s * Q::QJ + Q::J_QLO), LT(s * Q::QJ + Q::J_QHI), true, true);
    });
    // bounds of the trunk-tree joints: identical in the four lanes, written by the lead, owned by lane (t-1)&3
    static_for<1, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int row = QR::trunk_row(t);
        if constexpr (row >= 0)
        {
            constexpr int iq = Tp::idx_q[Tp::trunk_joint[t]];
            bound(row, qb[6 + t], P[L::QLO + iq], P[L::QHI + iq], lead, k == ((t - 1) & 3));
        }
    });
    // contact points of this lane's tip
    auto one_contact = [&](int c) {
        if (c >= ix.nc) return;
        const int oc = Q::CONTACT + c * Q::QC;
        const int ci = (int)LT(oc + Q::C_IDX);
        const V3<T> pc = K.Rt * LT.v3(oc + 9) + K.ps[N - 1];
        const T d = K.p1.z + dot(V3<T>{K.R1.m20, K.R1.m21, K.R1.m22}, pc);
        const unsigned of = (unsigned)(R::NB + ci) * B32 + r32;
        int32_t f = init ? 1 : C.flags[of];
        if (!readonly)
        {
            bool clear = init;
            if (d < T(0)) f = 1;
            else if (d > eps_tr) { f = 0; clear = true; }
            if (clear)
            {
#pragma unroll
                for (int i = 0; i < 4; ++i) C.data[(unsigned)(R::LAM + R::NB + 4 * ci + i) * B32 + r32] = T(0);
            }
            C.flags[of] = f;
        }
        if (f & 1)
        {
            const int r0 = R::NB + 4 * ci;
            const int nrow = torsion_zero ? 3 : 4;
            for (int i = 0; i < nrow; ++i) { en.set(r0 + i); own.set(r0 + i); }
            if (torsion_zero && !readonly) C.data[(unsigned)(R::LAM + r0 + 3) * B32 + r32] = T(0);
        }
    };
    if constexpr (Tp::QCL <= 2) static_for<0, Tp::QCL>([&](auto cc) { one_contact(decltype(cc)::value); });
    else
    {
#pragma nounroll
        for (int c = 0; c < Tp::QCL; ++c) one_contact(c);
    }
    quad_or_mask<X>(en);
    quad_or_mask<X>(rv);
    // row cap: the highest rows are dropped (and the lane flagged) when a robot has more active rows than the
    // solver region was sized for
    cx.overflow = false;
    if (en.count() > QR::MAXM)
    {
        cx.overflow = true;
        int keep = QR::MAXM;
        typename QConCtx<T, Tp>::RowMask lim;
        lim.clear();
        typename QConCtx<T, Tp>::RowMask tmp = en;
        while (keep > 0 && tmp.any()) { lim.set(tmp.pop_lowest()); --keep; }
#pragma unroll
        for (int i = 0; i < QR::NWORDS; ++i) { en.w[i] &= lim.w[i]; own.w[i] &= lim.w[i]; }
    }
    cx.act = en; cx.rev = rv; cx.mine = own;
    cx.m = en.count();
    cx.nb = en.rank(R::NB);
    cx.cb = torsion_zero ? 3 : 4;
}

// ---------------------------------------------------------------- bias-free sweeps, root coordinates
// motion subspace of limb joint s
template<class T, class Tp> JM_DEV Sp<T> limb_S(const QKeep<T, Tp> & K, int s) { return {cross(K.ps[s], K.as[s]), K.as[s]}; }

// tip -> base along the lane's own limb: joint efforts tau[s], force `ftip` applied ON the tip body (root
// coordinates); returns the bias force handed to the attachment joint, leaves u[s] for the way back
template<class T, class Tp>
JM_DEV Sp<T> limb_push(const QKeep<T, Tp> & K, const T * tau, Sp<T> ftip, T * u)
{
    constexpr int N = Tp::QN;
    Sp<T> acc = zero6<T>() - ftip;
    static_rfor<0, N>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const Sp<T> S = limb_S<T, Tp>(K, s);
        const T uj = tau[s] - dot6(S, acc);
        u[s] = uj;
        acc = acc + (uj * K.dinv[s]) * K.Us[s];
    });
    return acc;
}
// base -> tip: acceleration of the attachment joint in, joint accelerations dd[s] and the tip acceleration out
template<class T, class Tp>
JM_DEV Sp<T> limb_pull(const QKeep<T, Tp> & K, const QIdx<Tp> & ix, const T * u, bool with_u, Sp<T> ap, T * dd)
{
    constexpr int N = Tp::QN;
    static_for<0, N>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const T uj = with_u ? u[s] : T(0);
        const T d = ix.has[s] ? K.dinv[s] * (uj - dot6(K.Us[s], ap)) : T(0);  // dummy joints never move
        dd[s] = d;
        ap = ap + d * limb_S<T, Tp>(K, s);
    });
    return ap;
}

// trunk tree for ONE lane's own column: bias force `f_in` enters at trunk joint `t_in` (0 = the root), joint
// efforts tau_b[t]; returns the spatial accelerations of the trunk joints (at[0] = root) and their joint
// accelerations ddb[t].  The TrunkStore broadcasts are executed by the four lanes together (uniform code).
template<class T, class Tp, class X>
JM_DEV void trunk_column(CPtr<T> P, const QKeep<T, Tp> & K, const TrunkStore<T, Tp> & TS, int t_in, Sp<T> f_in, const T * tau_b,
                         Sp<T> * at, T * ddb)
{
    using QR = QConRows<Tp>;
    constexpr int NT = Tp::QT;
    T ub[NT];
    ub[0] = T(0);
    // leaves -> root: the force travels along the chain of ancestors of t_in (plus the efforts of the joints
    // themselves); a single running 6-vector per parent level
    Sp<T> accF[NT];
    static_for<0, NT>([&](auto tc) { accF[decltype(tc)::value] = mask6(t_in == decltype(tc)::value, f_in); });
    static_rfor<1, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int tp = Tp::trunk_parent[t];
        SE3<T> Xt;
        Sp<T> vt, Ut;
        T di, uj_free;
        TS.template get_kin<t, X>(Xt, vt);
        TS.template get_aba<t, X>(Ut, di, uj_free);
        const Sp<T> S = trunk_S_at<T, Tp, t>(P, Xt);
        const T uj = tau_b[t] - dot6(S, accF[t]);
        ub[t] = uj;
        accF[tp] = accF[tp] + accF[t] + (uj * di) * Ut;
        (void)vt; (void)uj_free;
    });
    // root block
    {
        T b[6] = {-accF[0].l.x, -accF[0].l.y, -accF[0].l.z, -accF[0].a.x, -accF[0].a.y, -accF[0].a.z};
        chol6_resolve(K.rootL, K.rootdinv, b);
        at[0] = {{b[0], b[1], b[2]}, {b[3], b[4], b[5]}};
    }
    ddb[0] = T(0);
    static_for<1, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int tp = Tp::trunk_parent[t];
        SE3<T> Xt;
        Sp<T> vt, Ut;
        T di, uj_free;
        TS.template get_kin<t, X>(Xt, vt);
        TS.template get_aba<t, X>(Ut, di, uj_free);
        const Sp<T> S = trunk_S_at<T, Tp, t>(P, Xt);
        const T d = di * (ub[t] - dot6(Ut, at[tp]));
        ddb[t] = d;
        at[t] = at[tp] + d * S;
        (void)vt; (void)uj_free;
    });
    (void)QR::MAXM;
}

// ---------------------------------------------------------------- delassus matrix, four columns per round
template<class T, class Tp, class X>
JM_DEV void qcon_delassus(CPtr<T> P, const LimbTable<T> & LT, const QConArgs<T> & C, int k, const QIdx<Tp> & ix,
                          const QKeep<T, Tp> & K, const TrunkStore<T, Tp> & TS, const QConCtx<T, Tp> & cx, const QStore<T> & V)
{
    using Q = QLayout<Tp>;
    using R = ConRows<Tp>;
    using QR = QConRows<Tp>;
    using I = QInfo<Tp>;
    constexpr int N = Tp::QN, NT = Tp::QT;
    const int m = cx.m;
    const int A0 = 3 * m;
    typename QConCtx<T, Tp>::RowMask rem = cx.mine;
    // acceleration of the joint this lane's limb hangs from, out of the trunk accelerations of column lane `c`
    while (X::quad_or(rem.any() ? 1 : 0))
    {
        const int r = rem.any() ? rem.pop_lowest() : -1;
        const int pcol = r >= 0 ? cx.act.rank(r) : 0;
        const T sgn = (r >= 0 && cx.rev.test(r)) ? T(-1) : T(1);
        // ---- decode the source of this lane's column
        T tau_l[N], tau_b[NT];
        static_for<0, N>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            const int row = sel4(k, QR::limb_row(0, s), QR::limb_row(1, s), QR::limb_row(2, s), QR::limb_row(3, s));
            tau_l[s] = (r >= 0 && row == r) ? sgn : T(0);
        });
        tau_b[0] = T(0);
        bool on_trunk = false;
        static_for<1, NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int row = QR::trunk_row(t);
            const bool hit = row >= 0 && row == r;
            tau_b[t] = hit ? sgn : T(0);
            on_trunk |= hit;
        });
        Sp<T> fu = zero6<T>();
        if (r >= R::NB)
        {
            const int ci = (r - R::NB) >> 2, d = (r - R::NB) & 3;
            // contact point of that index on this tip
            V3<T> pc = zero3<T>();
            auto find = [&](int c) {
                const int oc = Q::CONTACT + c * Q::QC;
                if (c < ix.nc && (int)LT(oc + Q::C_IDX) == ci) pc = K.Rt * LT.v3(oc + 9) + K.ps[N - 1];
            };
            if constexpr (Tp::QCL <= 2) static_for<0, Tp::QCL>([&](auto cc) { find(decltype(cc)::value); });
            else
            {
#pragma nounroll
                for (int c = 0; c < Tp::QCL; ++c) find(c);
            }
            // unit force along world x / y / z at the contact point, or unit torque about world z
            const V3<T> col = d == 0 ? V3<T>{K.R1.m00, K.R1.m01, K.R1.m02}
                            : d == 1 ? V3<T>{K.R1.m10, K.R1.m11, K.R1.m12} : V3<T>{K.R1.m20, K.R1.m21, K.R1.m22};
            if (d < 3) fu = {col, cross(pc, col)};
            else fu = {zero3<T>(), col};
        }
        // ---- own limb down, trunk tree, own column's accelerations
        T ul[N];
        const Sp<T> fbase = limb_push<T, Tp>(K, tau_l, fu, ul);
        Sp<T> at[NT];
        T ddb[NT];
        trunk_column<T, Tp, X>(P, K, TS, on_trunk ? -1 : ix.attach, mask6(!on_trunk, fbase), tau_b, at, ddb);
        // rows of the trunk-tree joints: every lane writes the entries of ITS column
        if (r >= 0)
            static_for<1, NT>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int row = QR::trunk_row(t);
                if constexpr (row >= 0)
                    if (cx.act.test(row))
                    {
                        const int pr = cx.act.rank(row);
                        if (pr >= pcol)
                        {
                            T val = cx.rev.test(row) ? -ddb[t] : ddb[t];
                            if (pr == pcol) val += fmax_(val * C.reg, T(1.0e-11));  // regularisation, constraint_solvers.cc:376-387
                            V.put(A0 + tri_(pr, pcol), val);
                        }
                    }
            });
        // ---- every lane sweeps its limb once per column
        static_for<0, 4>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const int rc = X::template bcast<c>(r);
            const int pc_ = X::template bcast<c>(pcol);
            // acceleration of this lane's attachment joint in column c
            Sp<T> aatt = qbcast<T, X, c>(at[Tp::limb_attach[0]]);
            static_for<1, 4>([&](auto kc) {
                constexpr int kk = decltype(kc)::value;
                if constexpr (Tp::limb_attach[kk] != Tp::limb_attach[0])
                {
                    const Sp<T> alt = qbcast<T, X, c>(at[Tp::limb_attach[kk]]);
                    aatt = msel(k == kk, alt, aatt);
                }
            });
            if (rc >= 0)
            {
                T dd[N];
                const Sp<T> atip = limb_pull<T, Tp>(K, ix, ul, k == c, aatt, dd);
                auto store = [&](int row, T val) {
                    const int pr = cx.act.rank(row);
                    if (pr >= pc_)
                    {
                        if (pr == pc_) val += fmax_(val * C.reg, T(1.0e-11));
                        V.put(A0 + tri_(pr, pc_), val);
                    }
                };
                static_for<0, N>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    const int row = sel4(k, QR::limb_row(0, s), QR::limb_row(1, s), QR::limb_row(2, s), QR::limb_row(3, s));
                    if (row >= 0 && ix.has[s] && cx.act.test(row)) store(row, cx.rev.test(row) ? -dd[s] : dd[s]);
                });
                auto rows_of = [&](int cl) {
                    const int oc = Q::CONTACT + cl * Q::QC;
                    if (cl >= ix.nc) return;
                    const int r0 = R::NB + 4 * (int)LT(oc + Q::C_IDX);
                    if (!cx.act.test(r0)) return;
                    const V3<T> pc = K.Rt * LT.v3(oc + 9) + K.ps[N - 1];
                    const V3<T> lin = K.R1 * (atip.l + cross(atip.a, pc));
                    store(r0, lin.x); store(r0 + 1, lin.y); store(r0 + 2, lin.z);
                    if (cx.cb == 4) store(r0 + 3, dot(V3<T>{K.R1.m20, K.R1.m21, K.R1.m22}, atip.a));
                };
                if constexpr (Tp::QCL <= 2) static_for<0, Tp::QCL>([&](auto c2) { rows_of(decltype(c2)::value); });
                else
                {
#pragma nounroll
                    for (int cl = 0; cl < Tp::QCL; ++cl) rows_of(cl);
                }
            }
        });
    }
    (void)I::NVB;
}

// ---------------------------------------------------------------- right-hand side and warm start
// b = -(drift + J a_free) for the rows this lane owns (Baumgarte terms: abstract_constraint.cc:88-98), x = lambda
template<class T, class Tp>
JM_DEV void qcon_rhs(CPtr<T> P, const LimbTable<T> & LT, const QConArgs<T> & C, unsigned B32, unsigned r32, int k,
                     const QIdx<Tp> & ix, const T * qb, const T * vb, const T * ql, const T * vl, const T * ddqb, const T * ddq,
                     const QKeep<T, Tp> & K, const QConCtx<T, Tp> & cx, const QStore<T> & V)
{
    using Q = QLayout<Tp>;
    using R = ConRows<Tp>;
    using QR = QConRows<Tp>;
    constexpr int N = Tp::QN, NT = Tp::QT;
    const int m = cx.m;
    auto bound = [&](int row, T qj, T vj, T aj) {
        if (!cx.act.test(row)) return;
        const int p = cx.act.rank(row);
        const T s = C.kp * (qj - C.data[(unsigned)row * B32 + r32]) + C.kd * vj + aj;
        V.put(m + p, cx.rev.test(row) ? s : -s);
        V.put(p, C.data[(unsigned)(R::LAM + row) * B32 + r32]);
        V.put(2 * m + p, T(0));
    };
    static_for<0, N>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const int row = sel4(k, QR::limb_row(0, s), QR::limb_row(1, s), QR::limb_row(2, s), QR::limb_row(3, s));
        if (row >= 0 && ix.has[s]) bound(row, ql[s], vl[s], ddq[s]);
    });
    static_for<1, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int row = QR::trunk_row(t);
        if constexpr (row >= 0)
            if (k == ((t - 1) & 3)) bound(row, qb[6 + t], vb[5 + t], ddqb[5 + t]);
    });
    const Sp<T> sa = K.atip - K.agf1;   // true spatial acceleration of the tip under the free motion
    auto contact = [&](int cl) {
        if (cl >= ix.nc) return;
        const int oc = Q::CONTACT + cl * Q::QC;
        const int r0 = R::NB + 4 * (int)LT(oc + Q::C_IDX);
        if (!cx.act.test(r0)) return;
        const V3<T> pc = K.Rt * LT.v3(oc + 9) + K.ps[N - 1];
        const T depth = K.p1.z + dot(V3<T>{K.R1.m20, K.R1.m21, K.R1.m22}, pc);
        const V3<T> vlin = K.R1 * (K.vtip.l + cross(K.vtip.a, pc));
        const V3<T> vang = K.R1 * K.vtip.a;
        V3<T> alin = K.R1 * (sa.l + cross(sa.a, pc));
        const V3<T> aang = K.R1 * sa.a;
        alin = alin + cross(vang, vlin);
        const int p0 = cx.act.rank(r0);
        V.put(m + p0, -(alin.x + C.kd * vlin.x));
        V.put(m + p0 + 1, -(alin.y + C.kd * vlin.y));
        V.put(m + p0 + 2, -(alin.z + C.kp * depth + C.kd * vlin.z));
        if (cx.cb == 4) V.put(m + p0 + 3, -(aang.z + C.kd * vang.z));
        for (int i = 0; i < cx.cb; ++i)
        {
            V.put(p0 + i, C.data[(unsigned)(R::LAM + r0 + i) * B32 + r32]);
            V.put(2 * m + p0 + i, T(0));
        }
    };
    if constexpr (Tp::QCL <= 2) static_for<0, Tp::QCL>([&](auto cc) { contact(decltype(cc)::value); });
    else
    {
#pragma nounroll
        for (int cl = 0; cl < Tp::QCL; ++cl) contact(cl);
    }
    (void)P;
}
// multipliers back to the per-lane constraint state (rows this lane owns)
template<class T, class Tp>
JM_DEV void qcon_scatter(const LimbTable<T> & LT, const QConArgs<T> & C, unsigned B32, unsigned r32, int k, const QIdx<Tp> & ix,
                         const QConCtx<T, Tp> & cx, const QStore<T> & V)
{
    using Q = QLayout<Tp>;
    using R = ConRows<Tp>;
    using QR = QConRows<Tp>;
    constexpr int N = Tp::QN, NT = Tp::QT;
    auto put = [&](int row) { if (cx.act.test(row)) C.data[(unsigned)(R::LAM + row) * B32 + r32] = V.get(cx.act.rank(row)); };
    static_for<0, N>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const int row = sel4(k, QR::limb_row(0, s), QR::limb_row(1, s), QR::limb_row(2, s), QR::limb_row(3, s));
        if (row >= 0 && ix.has[s]) put(row);
    });
    static_for<1, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int row = QR::trunk_row(t);
        if constexpr (row >= 0)
            if (k == ((t - 1) & 3)) put(row);
    });
    auto contact = [&](int cl) {
        if (cl >= ix.nc) return;
        const int r0 = R::NB + 4 * (int)LT(Q::CONTACT + cl * Q::QC + Q::C_IDX);
        for (int i = 0; i < cx.cb; ++i) put(r0 + i);
    };
    if constexpr (Tp::QCL <= 2) static_for<0, Tp::QCL>([&](auto cc) { contact(decltype(cc)::value); });
    else
    {
#pragma nounroll
        for (int cl = 0; cl < Tp::QCL; ++cl) contact(cl);
    }
}

// ---------------------------------------------------------------- solvers (the four lanes of the quad together)
// sum_c A[i][c] x[c]: every lane takes the columns c = k, k+4, ... ; one butterfly adds the quarters
template<class T, class X> JM_DEV T qcon_dot(const QStore<T> & V, int m, int i, int k)
{
    const int A0 = 3 * m;
    T s = T(0);
    for (int c = k; c < m; c += 4) s += V.get(A0 + tri_(i, c)) * V.get(c);
    return X::quad_sum(s);
}
// PGSSolver::ProjectedGaussSeidelSolver (constraint_solvers.cc:107-333) over the m packed rows: `nb` joint
// bounds, then blocks of `cb` rows (x, y, z[, torsion]) per active contact
template<class T, class Tp, class X>
JM_DEV bool qcon_pgs(const QConArgs<T> & C, T friction, int k, const QConCtx<T, Tp> & cx, const QStore<T> & V)
{
    const int m = cx.m, nb = cx.nb, cb = cx.cb, A0 = 3 * m;
    const bool lead = (k == 0);
    const T eps = Eps<T>::eps;
    const bool friction_zero = friction < eps;
    const unsigned iter_max = (unsigned)C.iter_max;
    for (unsigned iter = 0; iter < iter_max; ++iter)
    {
        T dmax = T(0), ymax = T(0);
        // under-relaxation schedule (constraint_solvers.cc:248-258)
        const T ratio = (T(iter_max - 20u) - T(iter)) / T(iter_max - 20u - 30u);
        T w = T(1);
        if (ratio < T(1))
        {
            w = T(0.01);
            if (ratio > T(0)) w += (T(1) - T(0.01)) * (ratio * ratio);
        }
        auto residual = [&](int i) {
            const T y = V.get(m + i) - qcon_dot<T, X>(V, m, i, k);
            dmax = fmax_(dmax, cabs_(y - V.get(2 * m + i)));
            X::sync();
            if (lead) V.put(2 * m + i, y);
            return y;
        };
        // block 0 of every constraint: joint bounds, then the normal force of every contact
        for (int r = 0; r < m; r += (r < nb ? 1 : cb))
        {
            const int i0 = r < nb ? r : r + 2;
            const T y = residual(i0);
            const T e = V.get(i0) + w * y / V.get(A0 + tri_(i0, i0));
            X::sync();
            if (lead) V.put(i0, fmax_(e, T(0)));  // clamp(e, 0, inf)
            X::sync();
        }
        // block 1: torsional friction {3, 2}
        if (cb == 4)
            for (int r = nb; r < m; r += 4)
            {
                const int i0 = r + 3;
                if (C.torsion < eps)
                {
                    X::sync();
                    if (lead) V.put(i0, V.get(i0) * T(0));
                    X::sync();
                    continue;
                }
                const T y = residual(i0);
                const T e = V.get(i0) + w * y / V.get(A0 + tri_(i0, i0));
                const T thr = C.torsion * V.get(r + 2);
                X::sync();
                if (lead) V.put(i0, clamp_(e, -thr, thr));
                X::sync();
            }
        // block 2: friction cone {0, 1, 2}
        for (int r = nb; r < m; r += cb)
        {
            if (friction_zero)
            {
                X::sync();
                if (lead) { V.put(r, V.get(r) * T(0)); V.put(r + 1, V.get(r + 1) * T(0)); }
                X::sync();
                continue;
            }
            const T y0 = residual(r);
            const T y1 = residual(r + 1);
            const T a00 = V.get(A0 + tri_(r, r)), a11 = V.get(A0 + tri_(r + 1, r + 1));
            const T a_max = a11 > a00 ? a11 : a00;
            T e0 = V.get(r) + w * y0 / a_max;
            T e1 = V.get(r + 1) + w * y1 / a_max;
            const T thr = friction * V.get(r + 2);
            const T n2 = e0 * e0 + e1 * e1;
            if (n2 > thr * thr)
            {
                const T scale = thr / sqrt_(n2);
                e0 *= scale;
                e1 *= scale;
            }
            X::sync();
            if (lead) { V.put(r, e0); V.put(r + 1, e1); }
            X::sync();
        }
        // stagnation of the residuals (constraint_solvers.cc:263-278)
        for (int r = 0; r < m; ++r) ymax = fmax_(ymax, cabs_(V.get(2 * m + r)));
        const T tol = C.tol_abs + C.tol_rel * ymax + eps;
        if (dmax < tol) return true;
    }
    return false;
}
// Exact solve A x = b (Engine::start's first pass, `ignoreBounds`: solveJMinvJtv): Cholesky in place in the
// packed triangle -- the caller rebuilds the matrix afterwards.  Serial, lead lane only (start / reset only).
template<class T, class X>
JM_DEV bool qcon_chol(int k, int m, const QStore<T> & V)
{
    const int A0 = 3 * m;
    bool ok = true;
    X::sync();
    if (k == 0)
    {
        for (int j = 0; j < m; ++j)
        {
            T s = V.get(A0 + tri_(j, j));
            for (int c = 0; c < j; ++c) { const T l = V.get(A0 + tri_(j, c)); s -= l * l; }
            ok &= s > T(0);
            const T d = sqrt_(s);
            V.put(A0 + tri_(j, j), d);
            for (int i = j + 1; i < m; ++i)
            {
                T t = V.get(A0 + tri_(i, j));
                for (int c = 0; c < j; ++c) t -= V.get(A0 + tri_(i, c)) * V.get(A0 + tri_(j, c));
                V.put(A0 + tri_(i, j), t / d);
            }
        }
        for (int i = 0; i < m; ++i)
        {
            T s = V.get(m + i);
            for (int c = 0; c < i; ++c) s -= V.get(A0 + tri_(i, c)) * V.get(c);
            V.put(i, s / V.get(A0 + tri_(i, i)));
        }
        for (int i = m - 1; i >= 0; --i)
        {
            T s = V.get(i);
            for (int c = i + 1; c < m; ++c) s -= V.get(A0 + tri_(c, i)) * V.get(c);
            V.put(i, s / V.get(A0 + tri_(i, i)));
        }
    }
    X::sync();
    return X::quad_or(ok ? 0 : 1) == 0;
}

// ---------------------------------------------------------------- one constrained evaluation
// `start_passes` > 0: Engine::start / reset sequence; < 0: MODE_REFRESH (re-apply the stored multipliers);
// 0: a regular evaluation.  Leaves the constrained acceleration in ddqb / ddq.
template<class T, class Tp, class X, bool EMIT, class SB>
JM_DEV void quad_eval_con(CPtr<T> P, const LimbTable<T> & LT, const BatchArgs<T> & A, const QConArgs<T> & C, const QStore<T> & V,
                          unsigned r, int k, const QIdx<Tp> & ix, const SB & S_, const T * qb, const T * vb, const T * ql,
                          const T * vl, const T * cmdb, const T * cmdl, bool sensors, T * ddqb, T * ddq, int & status,
                          int start_passes)
{
    using L = Layout<Tp>;
    using R = ConRows<Tp>;
    using QR = QConRows<Tp>;
    constexpr int N = Tp::QN, NT = Tp::QT;
    const unsigned B32 = (unsigned)A.B;
    unsigned r32 = r;
    JM_OPAQUE(r32);
    QExtra<T, Tp> ex;
    static_for<0, N>([&](auto sc) { ex.tau_l[decltype(sc)::value] = T(0); ex.uemit_l[decltype(sc)::value] = T(0); });
    static_for<0, NT>([&](auto tc) { ex.tau_b[decltype(tc)::value] = T(0); ex.uemit_b[decltype(tc)::value] = T(0); });
    ex.motors_on = true;
    ex.flags = C.flags;
    ex.lam = C.data + (size_t)R::LAM * B32;
    ex.nb = R::NB;
    status &= ~JM_LANE_SOLVER_FAILURE;
    if constexpr (R::NR == 0)
    {
        quad_eval<T, Tp, X, EMIT, SB, 2>(P, LT, A, r32, k, ix, S_, qb, vb, ql, vl, cmdb, cmdl, sensors, ddqb, ddq, status, &ex);
        return;
    }
    const bool refresh = start_passes < 0;
    const bool init = start_passes > 0;
    const int n_pass = init ? start_passes : 1;
    const T friction = C.friction ? C.friction[r32] : P[L::OPT + 8];
    QKeep<T, Tp> K;
    TrunkStore<T, Tp> TS;
#ifdef JM_HOST_EMU
    std::memset(&K, 0xFF, sizeof(K)); std::memset(&TS, 0xFF, sizeof(TS));
#endif
    QConCtx<T, Tp> cx;
    cx.m = 0;
    T uq_l[N], uq_b[NT];   // RobotState::u of the previous start pass minus the motor efforts (bound multipliers, + sign)
    static_for<0, N>([&](auto sc) { uq_l[decltype(sc)::value] = T(0); });
    static_for<0, NT>([&](auto tc) { uq_b[decltype(tc)::value] = T(0); });
    bool any = false;
#pragma nounroll
    for (int pass = 0; pass < n_pass; ++pass)
    {
        // ---- free acceleration of this pass (+ what the bias-free solves need)
        static_for<0, N>([&](auto sc) { ex.tau_l[decltype(sc)::value] = uq_l[decltype(sc)::value]; });
        static_for<0, NT>([&](auto tc) { ex.tau_b[decltype(tc)::value] = uq_b[decltype(tc)::value]; });
        ex.motors_on = !(init && pass == 0);
        quad_eval<T, Tp, X, false, SB, 1, QKeep<T, Tp>>(P, LT, A, r32, k, ix, S_, qb, vb, ql, vl, cmdb, cmdl, false, ddqb, ddq,
                                                       status, &ex, &K, &TS);
        if (pass == 0)
        {
            qcon_switch<T, Tp, X>(P, LT, C, B32, r32, k, ix, qb, ql, K, init, refresh, cx);
            any = cx.act.any();
            if (cx.overflow) status |= JM_LANE_SOLVER_FAILURE;
            if (!any || refresh) break;
            qcon_delassus<T, Tp, X>(P, LT, C, k, ix, K, TS, cx, V);
        }
        X::sync();
        qcon_rhs<T, Tp>(P, LT, C, B32, r32, k, ix, qb, vb, ql, vl, ddqb, ddq, K, cx, V);
        X::sync();
        bool ok;
        if (init && pass == 0)
        {
            ok = qcon_chol<T, X>(k, cx.m, V);
            if (!ok) status |= JM_LANE_NAN;
            X::sync();
            qcon_scatter<T, Tp>(LT, C, B32, r32, k, ix, cx, V);
            X::sync();
            qcon_delassus<T, Tp, X>(P, LT, C, k, ix, K, TS, cx, V);   // the factorisation overwrote the matrix
        }
        else
        {
            ok = qcon_pgs<T, Tp, X>(C, friction, k, cx, V);
            if (ok) status &= ~JM_LANE_SOLVER_FAILURE;
            else status |= JM_LANE_SOLVER_FAILURE;
            if (cx.overflow) status |= JM_LANE_SOLVER_FAILURE;
            X::sync();
            qcon_scatter<T, Tp>(LT, C, B32, r32, k, ix, cx, V);
        }
        X::sync();
        if (pass == n_pass - 1) break;
        // Engine::start: the next pass sees u = uInternal (bound multipliers of this pass, plus sign whatever the
        // direction, engine.cc:3786-3790) + motor efforts (engine.cc:1456-1465)
        static_for<0, N>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            const int row = sel4(k, QR::limb_row(0, s), QR::limb_row(1, s), QR::limb_row(2, s), QR::limb_row(3, s));
            uq_l[s] = (row >= 0 && ix.has[s] && cx.act.test(row)) ? C.data[(unsigned)(R::LAM + row) * B32 + r32] : T(0);
        });
        static_for<1, NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int row = QR::trunk_row(t);
            if constexpr (row >= 0) uq_b[t] = cx.act.test(row) ? C.data[(unsigned)(R::LAM + row) * B32 + r32] : T(0);
        });
    }
    // ---- nothing to enforce and nothing to emit: the free acceleration is the answer (engine.cc:3861-3865)
    if constexpr (!EMIT)
        if (!any) return;
    // ---- apply the multipliers: articulated-body solve with the constraint forces; emits the outputs
    static_for<0, N>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const int row = sel4(k, QR::limb_row(0, s), QR::limb_row(1, s), QR::limb_row(2, s), QR::limb_row(3, s));
        const bool on = any && row >= 0 && ix.has[s] && cx.act.test(row);
        const T lam = on ? C.data[(unsigned)(R::LAM + (on ? row : 0)) * B32 + r32] : T(0);
        ex.tau_l[s] = uq_l[s] + ((on && cx.rev.test(row)) ? -lam : lam);
        ex.uemit_l[s] = lam;
    });
    static_for<1, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int row = QR::trunk_row(t);
        if constexpr (row >= 0)
        {
            const bool on = any && cx.act.test(row);
            const T lam = on ? C.data[(unsigned)(R::LAM + row) * B32 + r32] : T(0);
            ex.tau_b[t] = uq_b[t] + ((on && cx.rev.test(row)) ? -lam : lam);
            ex.uemit_b[t] = lam;
        }
        else { ex.tau_b[t] = uq_b[t]; ex.uemit_b[t] = T(0); }
    });
    ex.motors_on = true;
    quad_eval<T, Tp, X, EMIT, SB, 2>(P, LT, A, r32, k, ix, S_, qb, vb, ql, vl, cmdb, cmdl, sensors, ddqb, ddq, status, &ex);
}
}  // namespace jm
