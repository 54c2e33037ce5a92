// jm_qtip.h -- the constraint solve of robots whose contact points sit on FEW bodies (a humanoid: 16 contact points per
// foot, two feet), in the operational space of those bodies instead of the space of the constraint rows.
//
// All the contact points of a foot act on ONE rigid body.  The delassus matrix of the reference
// (A = J M^-1 J^T + R, PGSSolver::SolveBoxedForwardDynamics, core/src/solver/constraint_solvers.cc:335-448) restricted to
// the contact rows is therefore  X E X^T + R  with
//   E  the inverse operational-space inertia of the contact-bearing tip bodies and the active joint rows: the response
//      (spatial acceleration of every tip body, acceleration of every bounded joint) to a unit wrench on a tip body / a unit
//      effort on a joint -- NE x NE with NE = 6 tips + joint slots, 20 for Atlas whatever the number of contact points;
//   X  one 6-vector per contact row: direction d at point p -> (d, p x d) (a torsion row about n: (0, n)), a unit entry
//      (+-1, the bound's direction) per joint row;
//   R  the diagonal regularisation (constraint_solvers.cc:376-387).
// The matrix is never formed.  The build (`qtip_build`, in the kernel that runs the free evaluation) needs 6 bias-free
// solves per foot + one per active joint row instead of one per constraint ROW (Atlas standing: ~8 rounds of the quad
// instead of ~24), and the projected Gauss-Seidel sweeps (`qtip_pgs`) keep z = E X^T x up to date instead of streaming a
// row of A per update:   y_i = b_i - X_i . z - R_i x_i,   then   z += E[:, tip(i)] (X_i^T dx).
// E lives in registers (each lane of the quad owns the rows j = 4 q + k of E and the entries z_j), z and the multipliers
// in LDS, the per-row records (X_i, b_i, 1 / A_ii, R_i) are read ahead of the dependency chain through a small ring: a
// row update costs ~100 VALU instructions and 88 bytes instead of a round trip to the workspace for a 400-768 byte row.
// Same sweep order, relaxation schedule, projections and stopping rule as `qcon_pgs_lean` (constraint_solvers.cc:107-333):
// the iterates are the reference's up to the rounding of y_i.
//
// Taken by the split stepping (pre | solve | post, jm_qcon.h) when every robot of the wave has at most NBX active joint
// rows; the others keep the streamed form.  Included by jm_qcon.h.
#pragma once

#ifndef JM_QTIP
#define JM_QTIP 1            // 0: never take the operational-space form (A / B runs)
#endif
#ifndef JM_QTIP_WAVES
#define JM_QTIP_WAVES 1      // waves per SIMD of k_qtip_pgs (measured, Atlas B = 32 768: 2.40 ms per solve at 1, 2.66 at 2 -- issue-bound either way)
#endif
#ifndef JM_QTIP_DEPTH
#define JM_QTIP_DEPTH 2      // visits whose records are in flight per robot
#endif

namespace jm
{
JM_DEV void qcon_visit_table(int k, int m, int nb, int cb, unsigned long long lockp, unsigned short * vt);   // (jm_qcon.h)

template<class Tp> struct QTip
{
    static constexpr int count_tips()
    {
        int n = 0;
        for (int k = 0; k < 4; ++k) n += Tp::limb_ncontact[k] > 0 ? 1 : 0;
        return n;
    }
    static constexpr int NCT = count_tips();      // contact-bearing limb tips
    static constexpr int slot_of(int k)           // tip slot of limb k (-1: no contact point on that limb)
    {
        if (Tp::limb_ncontact[k] <= 0) return -1;
        int n = 0;
        for (int i = 0; i < k; ++i) n += Tp::limb_ncontact[i] > 0 ? 1 : 0;
        return n;
    }
    static constexpr int NBX = 8;                 // joint rows (active bounds / user joint constraints) of a solve in this form
                                                  // (Atlas standing in its neutral pose already has five joints at a limit)
    static constexpr int NE = 6 * NCT + NBX;      // extended operational space
    static constexpr int ZL = (NE + 3) / 4;       // entries of z / rows of E per lane (entry e: lane e & 3, slot e >> 2)
    static constexpr bool ON = JM_QTIP != 0 && qcon_split_large<Tp>() && NCT >= 1 && NCT <= 2;
    // the robot's region of the workspace in this form: x | b | y | 1 / diag (4 m, as in every form: what qcon_rhs / qcon_scatter
    // read and write) | E (NE x NE, row-major) | one record of REC scalars per row, everything a row visit reads in one
    // 96-byte block: X[6] | b | 1 / diag | R | y of the previous sweep | z offset (as an integer in the scalar's low word) | -
    static constexpr int e0(int m) { return 4 * m; }
    static constexpr int rec0(int m) { return 4 * m + NE * NE; }
    static constexpr int REC = 12;
    static constexpr int RB = 6, RINVD = 7, RREG = 8, RYP = 9, RMETA = 10;
    static constexpr int ZPAD = NE + 6;           // z on chip: a joint row reads 6 entries from its slot on
    static_assert(!ON || rec0(QConRows<Tp>::MAXM) + REC * QConRows<Tp>::MAXM <= QSplitRegion<Tp>::HDR, "operational-space form fits the region");
};

// ---------------------------------------------------------------- build: E, records, 1 / diag, R  -> the robot's region
// The column schedule of a lane: its active joint rows (own limb; trunk-tree rows by lane (t - 1) & 3), then -- if its tip
// carries an active contact point -- the six unit wrenches on its tip body (root coordinates, about the root origin).
// One round = the quad's four columns through limb_push / trunk_column / limb_pull like qcon_delassus.
template<class T, class Tp, class X, class VS>
JM_DEV void qtip_build(CPtr<T> P, const LimbTable<T> & LT, const QConArgs<T> & C, int k, const QIdx<Tp> & ix,
                       const QKeep<T, Tp> & K, const TrunkStore<T, Tp> & TS, const QConCtx<T, Tp> & cx, const VS & V)
{
    using Q = QLayout<Tp>;
    using R = ConRows<Tp>;
    using QR = QConRows<Tp>;
    using TP = QTip<Tp>;
    constexpr int N = Tp::QN, NT = Tp::QT, NE = TP::NE;
    const int m = cx.m;
    const int E0 = TP::e0(m), REC0 = TP::rec0(m);
#ifdef JM_HOST_EMU_TRACE
    std::fprintf(stderr, "[%d] build m=%d nb=%d cb=%d\n", k, m, cx.nb, cx.cb);
#endif
    const int myslot = sel4(k, TP::slot_of(0), TP::slot_of(1), TP::slot_of(2), TP::slot_of(3));
    // joint rows / contact rows this lane owns
    typename QConCtx<T, Tp>::RowMask remj = cx.mine, minec = cx.mine;
    {
        typename QConCtx<T, Tp>::RowMask lowb;
        lowb.clear();
        for (int r = 0; r < R::NB; ++r) lowb.set(r);
#pragma unroll
        for (int w = 0; w < QR::NWORDS; ++w) { remj.w[w] &= lowb.w[w]; minec.w[w] &= ~lowb.w[w]; }
    }
    const bool tip_on = myslot >= 0 && minec.any();
    // E starts from zero: the slots a robot does not use are read by the solve like the others
    for (int e = k; e < NE * NE; e += 4) V.put(E0 + e, T(0));
    X::sync();
    T Ett[6][6];   // own tip rows x own tip columns
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) Ett[a][b] = T(0);
    int wrench = 0;   // next unit wrench of this lane
    auto put_e = [&](int row, int col, T val) { V.put(E0 + row * NE + col, val); };
    while (X::quad_or((remj.any() || (tip_on && wrench < 6)) ? 1 : 0))
    {
        // ---- this lane's column of the round: a joint row, a unit wrench, or nothing
        int r = -1, cw = -1;
        if (remj.any()) r = remj.pop_lowest();
        else if (tip_on && wrench < 6) cw = wrench++;
        const int ecol = r >= 0 ? 6 * TP::NCT + cx.act.rank(r) : (cw >= 0 ? 6 * myslot + cw : -1);
        T tau_l[N], tau_b[NT];
        static_for<0, N>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            const int row = sel4(k, QR::limb_row(0, s), QR::limb_row(1, s), QR::limb_row(2, s), QR::limb_row(3, s));
            tau_l[s] = (r >= 0 && row == r) ? T(1) : T(0);
        });
        tau_b[0] = T(0);
        bool on_trunk = false;
        static_for<1, NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int row = QR::trunk_row(t);
            const bool hit = row >= 0 && row == r;
            tau_b[t] = hit ? T(1) : T(0);
            on_trunk |= hit;
        });
        Sp<T> fu = zero6<T>();
        if (cw >= 0)
        {
            const T one = T(1), z_ = T(0);
            fu.l = {cw == 0 ? one : z_, cw == 1 ? one : z_, cw == 2 ? one : z_};
            fu.a = {cw == 3 ? one : z_, cw == 4 ? one : z_, cw == 5 ? one : z_};
        }
        T ul[N];
        const Sp<T> fbase = limb_push<T, Tp>(K, tau_l, fu, ul);
        Sp<T> at[NT];
        T ddb[NT];
        {
            Sp<T> accF[NT];
            static_for<0, NT>([&](auto tc) { accF[decltype(tc)::value] = mask6(!on_trunk && ix.attach == decltype(tc)::value, fbase); });
            trunk_column<T, Tp, X>(P, K, TS, accF, tau_b, at, ddb);
        }
        // rows of the trunk-tree joints: every lane writes the entries of ITS column
        if (ecol >= 0)
            static_for<1, NT>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int row = QR::trunk_row(t);
                if constexpr (row >= 0)
                    if (cx.act.test(row)) put_e(6 * TP::NCT + cx.act.rank(row), ecol, ddb[t]);
            });
        // ---- every lane sweeps its limb once per column
        T d_own = T(0);
        static_for<0, 4>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const int ec = X::template bcast<c>(ecol);
            Sp<T> aatt = qbcast<T, X, c>(at[Tp::limb_attach[0]]);
            static_for<1, 4>([&](auto kc) {
                constexpr int kk = decltype(kc)::value;
                if constexpr (Tp::limb_attach[kk] != Tp::limb_attach[0])
                {
                    const Sp<T> alt = qbcast<T, X, c>(at[Tp::limb_attach[kk]]);
                    aatt = msel(k == kk, alt, aatt);
                }
            });
            if (ec >= 0)
            {
                T dd[N];
                const Sp<T> atip = limb_pull<T, Tp>(K, ix, ul, k == c, aatt, dd);
                static_for<0, N>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    const int row = sel4(k, QR::limb_row(0, s), QR::limb_row(1, s), QR::limb_row(2, s), QR::limb_row(3, s));
                    if (row >= 0 && ix.has[s] && cx.act.test(row)) put_e(6 * TP::NCT + cx.act.rank(row), ec, dd[s]);
                    if (k == c && row >= 0 && row == r) d_own = dd[s];   // diagonal entry of this lane's own joint row
                });
                if (tip_on)
                {
                    const T av[6] = {atip.l.x, atip.l.y, atip.l.z, atip.a.x, atip.a.y, atip.a.z};
#pragma unroll
                    for (int a = 0; a < 6; ++a) put_e(6 * myslot + a, ec, av[a]);
                    if (k == c && cw >= 0)
                    {
#pragma unroll
                        for (int a = 0; a < 6; ++a)
#pragma unroll
                            for (int b = 0; b < 6; ++b) Ett[a][b] = (b == cw) ? av[a] : Ett[a][b];
                    }
                }
            }
        });
        // ---- this lane's joint row: diagonal, regularisation (constraint_solvers.cc:376-387), record
        if (r >= 0)
        {
            T d = d_own;
            static_for<1, NT>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int row = QR::trunk_row(t);
                if constexpr (row >= 0)
                    if (row == r) d = ddb[t];
            });
            const int p = cx.act.rank(r);
            const T reg = fmax_(d * C.reg, T(1.0e-11));
            const int rb = REC0 + TP::REC * p;
            V.put(3 * m + p, T(1) / (d + reg));
            V.put(rb, cx.rev.test(r) ? T(-1) : T(1));
            for (int a = 1; a < 6; ++a) V.put(rb + a, T(0));
            V.put(rb + TP::RINVD, T(1) / (d + reg));
            V.put(rb + TP::RREG, reg);
            V.put(rb + TP::RMETA, bits_as<T>((unsigned long long)(unsigned)(6 * TP::NCT + p)));
        }
    }
    // ---- records of the contact rows of this lane's tip: X, diagonal = X^T Ett X, regularisation
    if (tip_on)
    {
        auto rows_of = [&](int cl) {
            const int oc = Q::CONTACT + cl * Q::QC;
            if (cl >= ix.nc) return;
            const int r0 = R::NB + 4 * (int)LT(oc + Q::C_IDX);
            if (!cx.act.test(r0)) return;
            const V3<T> pc = K.Rt * LT.v3(oc + 9) + K.ps[N - 1];
            T dep_;
            const M3<T> Mc = contact_frame<false>(C, K.R1, K.p1, pc, dep_);
            const int p0 = cx.act.rank(r0);
            for (int d = 0; d < cx.cb; ++d)
            {
                const V3<T> dir = d == 0 ? V3<T>{Mc.m00, Mc.m01, Mc.m02} : ((d == 1) ? V3<T>{Mc.m10, Mc.m11, Mc.m12} : V3<T>{Mc.m20, Mc.m21, Mc.m22});
                Sp<T> Xr;
                if (d < 3) Xr = {dir, cross(pc, dir)};
                else Xr = {zero3<T>(), dir};
                const T xv[6] = {Xr.l.x, Xr.l.y, Xr.l.z, Xr.a.x, Xr.a.y, Xr.a.z};
                T diag = T(0);
#pragma unroll
                for (int a = 0; a < 6; ++a)
                {
                    T s_ = T(0);
#pragma unroll
                    for (int b = 0; b < 6; ++b) s_ += Ett[a][b] * xv[b];
                    diag += xv[a] * s_;
                }
                const T reg = fmax_(diag * C.reg, T(1.0e-11));
                const int p = p0 + d, rb = REC0 + TP::REC * p;
                V.put(3 * m + p, T(1) / (diag + reg));
#pragma unroll
                for (int a = 0; a < 6; ++a) V.put(rb + a, xv[a]);
                V.put(rb + TP::RINVD, T(1) / (diag + reg));
                V.put(rb + TP::RREG, reg);
                V.put(rb + TP::RMETA, bits_as<T>((unsigned long long)(unsigned)(6 * myslot)));
            }
        };
#pragma nounroll
        for (int cl = 0; cl < Tp::QCL; ++cl) rows_of(cl);
    }
}

// ---------------------------------------------------------------- exact solve (Engine::start, first pass)
// `ignoreBounds` solve of ALL the active rows, A x = b (PGSSolver::SolveBoxedForwardDynamics with the exact pass of
// Engine::start, engine.cc:1399-1467; the streamed form factorises A: qcon_chol).  With A = X E X^T + R this is the
// Woodbury identity in the NE-dimensional operational space:
//     x = R^-1 (b - X v),   (E^-1 + X^T R^-1 X) v = X^T R^-1 b,
// solved in symmetric form: E = L L^T,  N = I + L^T G L  (G = X^T R^-1 X, block diagonal: one 6 x 6 block per tip, one
// entry per joint slot),  v = L N^-1 L^T h.  Two NE x NE (20 x 20) Cholesky factorisations instead of one of 64-96 rows.  Slots the
// robot does not use (no row maps to them) are decoupled (unit diagonal).  Every lane of the quad computes the same small
// system (private arrays); the rows of x are dealt over the lanes.  Returns false when a factorisation breaks down.
template<class T, class Tp, class X, int LANES = 4>
JM_DEV bool qtip_exact(int k, char * ws, size_t g0)
{
    using RG = QSplitRegion<Tp>;
    using TP = QTip<Tp>;
    constexpr int NE = TP::NE, NCT = TP::NCT, REC = TP::REC;
    auto G = [&](int e) -> T & { return *(T *)(ws + (g0 + (size_t)(unsigned)e * sizeof(T))); };
    const int hdr = (int)G(RG::HDR);
    if (((hdr >> 24) & 1) == 0) return true;   // (a robot of the streamed form: the quad-cooperative factorisation's job)
    const int m = hdr & 0xff;
    if (m == 0) return true;
    const int E0 = TP::e0(m), REC0 = TP::rec0(m);
    T Gm[NE][NE], h[NE], L[NE][NE];
    for (int a = 0; a < NE; ++a) { h[a] = T(0); for (int b = 0; b < NE; ++b) Gm[a][b] = T(0); }
    // G = X^T R^-1 X and h = X^T R^-1 b over the rows
    for (int i = 0; i < m; ++i)
    {
        const int rb = REC0 + REC * i;
        const int zoff = (int)(unsigned)as_bits(G(rb + TP::RMETA));
        const T ri = T(1) / G(rb + TP::RREG), bi = G(m + i);
        T xr[6];
        for (int a = 0; a < 6; ++a) xr[a] = G(rb + a);
        const int n = zoff < 6 * NCT ? 6 : 1;
        for (int a = 0; a < n; ++a)
        {
            h[zoff + a] += xr[a] * ri * bi;
            for (int b = 0; b < n; ++b) Gm[zoff + a][zoff + b] += xr[a] * ri * xr[b];
        }
    }
    // E with the unused slots decoupled, L = chol(E)
    bool ok = true;
    for (int a = 0; a < NE; ++a)
        for (int b = 0; b <= a; ++b)
        {
            const bool used = Gm[a][a] > T(0) && Gm[b][b] > T(0);
            T s_ = used ? G(E0 + a * NE + b) : (a == b ? T(1) : T(0));
            for (int c = 0; c < b; ++c) s_ -= L[a][c] * L[b][c];
            if (a == b) { ok &= s_ > T(0); L[a][a] = sqrt_(s_); }
            else L[a][b] = s_ / L[b][b];
        }
    // N = I + L^T G L (symmetric), factorised in place into the lower triangle of Gm
    {
        T GL[NE][NE];   // G L
        for (int a = 0; a < NE; ++a)
            for (int b = 0; b < NE; ++b)
            {
                T s_ = T(0);
                for (int c = b; c < NE; ++c) s_ += Gm[a][c] * L[c][b];
                GL[a][b] = s_;
            }
        for (int a = 0; a < NE; ++a)
            for (int b = 0; b <= a; ++b)
            {
                T s_ = a == b ? T(1) : T(0);
                for (int c = a; c < NE; ++c) s_ += L[c][a] * GL[c][b];
                Gm[a][b] = s_;
            }
    }
    for (int a = 0; a < NE; ++a)
        for (int b = 0; b <= a; ++b)
        {
            T s_ = Gm[a][b];
            for (int c = 0; c < b; ++c) s_ -= Gm[a][c] * Gm[b][c];
            if (a == b) { ok &= s_ > T(0); Gm[a][a] = sqrt_(s_); }
            else Gm[a][b] = s_ / Gm[b][b];
        }
    // v = L N^-1 L^T h
    T u[NE], v[NE];
    for (int a = 0; a < NE; ++a) { T s_ = T(0); for (int c = a; c < NE; ++c) s_ += L[c][a] * h[c]; u[a] = s_; }
    for (int a = 0; a < NE; ++a) { T s_ = u[a]; for (int c = 0; c < a; ++c) s_ -= Gm[a][c] * u[c]; u[a] = s_ / Gm[a][a]; }
    for (int a = NE - 1; a >= 0; --a) { T s_ = u[a]; for (int c = a + 1; c < NE; ++c) s_ -= Gm[c][a] * u[c]; u[a] = s_ / Gm[a][a]; }
    for (int a = 0; a < NE; ++a) { T s_ = T(0); for (int c = 0; c <= a; ++c) s_ += L[a][c] * u[c]; v[a] = s_; }
    // x = R^-1 (b - X v)
    for (int i = k; i < m; i += LANES)
    {
        const int rb = REC0 + REC * i;
        const int zoff = (int)(unsigned)as_bits(G(rb + TP::RMETA));
        const int n = zoff < 6 * NCT ? 6 : 1;
        T s_ = G(m + i);
        for (int a = 0; a < n; ++a) s_ -= G(rb + a) * v[zoff + a];
        G(i) = s_ / G(rb + TP::RREG);
    }
    if (k == 0) G(RG::OK) = ok ? T(1) : T(0);
    return ok;
}

// ---------------------------------------------------------------- the solve
// Visit table of a sweep in this form: like qcon_visit_table (same order: joint rows -- user-registered JointConstraints
// first --, normal forces, torsion rows, friction cones), but a friction cone is ONE visit (kind 5, row = its first tangential
// row; the second follows it, the normal force is the row after): nb + nc (+ nc) + nc visits.  Returns their number.
JM_DEV int qtip_visit_table(int k, int m, int nb, int cb, unsigned long long lockp, unsigned short * vt)
{
    const int nc = cb > 0 ? (m - nb) / cb : 0;
    const int nv = nb + nc + (cb == 4 ? nc : 0) + nc;
    if (k == 0)
    {
        int t = 0;
        for (int r = 0; r < nb; ++r) if ((lockp >> r) & 1ull) vt[t++] = (unsigned short)(r | (4 << 8));
        for (int r = 0; r < nb; ++r) if (!((lockp >> r) & 1ull)) vt[t++] = (unsigned short)r;
        for (int c = 0; c < nc; ++c) vt[t++] = (unsigned short)(nb + cb * c + 2);
        if (cb == 4) for (int c = 0; c < nc; ++c) vt[t++] = (unsigned short)((nb + 4 * c + 3) | (1 << 8));
        for (int c = 0; c < nc; ++c) vt[t++] = (unsigned short)((nb + cb * c) | (5 << 8));
    }
    return nv;
}

// `x` (MAXM + 2 scalars), `z` (QTip::ZPAD) and `vt` (MAXM + 4 words): the robot's multipliers, z = E X^T x and visit table
// on chip; `ws` + `g0`: the robot's region of the workspace.  Returns false when the robot's wave is not in this form.
// A visit is straight-line code: the projections of the row kinds are computed side by side and selected (the robots of a
// wave are at different places of their sweeps); only the friction cone (two rows) is a branch of its own.
template<class T, class Tp, class X, int D>
JM_DEV bool qtip_pgs(const QConArgs<T> & C, T friction, int k, T * x, T * z, T * ypl, unsigned short * vt, char * ws, unsigned g0)
{
    using RG = QSplitRegion<Tp>;
    using TP = QTip<Tp>;
    constexpr int NE = TP::NE, ZL = TP::ZL, NCT = TP::NCT, NBX = TP::NBX, REC = TP::REC;
    auto G = [&](int e) -> T & { return *(T *)(ws + (g0 + (unsigned)e * (unsigned)sizeof(T))); };
    const int hdr = (int)G(RG::HDR);
    if (!X::wave_any(((hdr >> 24) & 1) != 0)) return false;
    const int m = hdr & 0xff, nb = (hdr >> 8) & 0xff, cb = (hdr >> 16) & 0xff;
    if (m == 0) return true;   // (uniform over the quad)
    const int E0 = TP::e0(m), REC0 = TP::rec0(m);
    const bool lead = (k == 0);
    const T eps = Eps<T>::eps;
    const bool friction_zero = friction < eps, torsion_zero = C.torsion < eps;
    const unsigned iter_max = (unsigned)C.iter_max;
    auto zoff_of = [&](int row) { return (int)(unsigned)as_bits(G(REC0 + REC * row + TP::RMETA)); };
    // (`ypl`: the residuals of the previous sweep, on chip since round 6 -- as a field of the row records they were a global
    // store per visit, and every wait for a prefetched record also waited for the stores issued before it)
    for (int i = k; i < m; i += 4) { x[i] = G(i); G(REC0 + REC * i + TP::RB) = G(m + i); ypl[i] = T(0); }
    if (lead) { x[m] = T(0); x[m + 1] = T(0); ypl[m] = T(0); ypl[m + 1] = T(0); }
    for (int i = k; i < TP::ZPAD; i += 4) z[i] = T(0);
    // this lane's rows of E: the columns of the tips in registers (what every contact row needs), the columns of the joint
    // slots stay in the region (read by the few joint rows of a sweep: 32 registers less)
    constexpr int NTC = 6 * NCT;
    T Er[ZL][NTC];
    static_for<0, ZL>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int e = 4 * j + k;
        static_for<0, NTC>([&](auto cc) { Er[j][decltype(cc)::value] = e < NE ? G(E0 + (e < NE ? e : 0) * NE + decltype(cc)::value) : T(0); });
    });
    auto Eb = [&](int j, int b) -> T { const int e = 4 * j + k; return e < NE ? G(E0 + e * NE + NTC + b) : T(0); };
    const unsigned long long lockp = (unsigned long long)G(RG::LOCK);
    const int nv = qtip_visit_table(k, m, nb, cb, lockp, vt);
    X::fence();
    X::sync();
    // z = E w with w = X^T x (warm start): every lane sums the whole w, then applies its rows of E
    T zo[ZL];
    {
        T w[NE];
        static_for<0, NE>([&](auto cc) { w[decltype(cc)::value] = T(0); });
#pragma nounroll
        for (int i = 0; i < m; ++i)
        {
            const T xi = x[i];
            const int zoff = zoff_of(i);
#ifdef JM_HOST_EMU
            if (!(zoff >= 0 && zoff < NE)) { std::fprintf(stderr, "qtip_pgs: row %d of %d has no record (zoff %d, nb %d cb %d)\n", i, m, zoff, nb, cb); std::abort(); }
#endif
            T xr[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) xr[a] = G(REC0 + REC * i + a);
            static_for<0, NCT>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                const T xm = zoff == 6 * t ? xi : T(0);
#pragma unroll
                for (int a = 0; a < 6; ++a) w[6 * t + a] += xr[a] * xm;
            });
            static_for<0, NBX>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                w[6 * NCT + b] += zoff == 6 * NCT + b ? xr[0] * xi : T(0);
            });
        }
        static_for<0, ZL>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            T s_ = T(0);
            static_for<0, NTC>([&](auto cc) { s_ += Er[j][decltype(cc)::value] * w[decltype(cc)::value]; });
            static_for<0, NBX>([&](auto bc) { s_ += Eb(j, decltype(bc)::value) * w[NTC + decltype(bc)::value]; });
            zo[j] = s_;
            if (4 * j + k < NE) z[4 * j + k] = s_;
        });
    }
    X::sync();
    // what a visit reads of the workspace: the record of its row (and of the next one: a cone's second row)
    struct Rec { T xr[6], b, invd, reg; T xr2[6], b2, invd2, reg2; int i, kind, zoff; };
    auto fetch = [&](int t, Rec & R_) __attribute__((always_inline)) {
        const int v = vt[t];
        R_.i = v & 0xff; R_.kind = v >> 8;
        const T * rp = &G(REC0 + REC * R_.i);
#pragma unroll
        for (int a = 0; a < 6; ++a) R_.xr[a] = rp[a];
        R_.b = rp[TP::RB]; R_.invd = rp[TP::RINVD]; R_.reg = rp[TP::RREG];
        R_.zoff = (int)(unsigned)as_bits(rp[TP::RMETA]);
        // (the row after it: read whatever the kind -- one more 96-byte block at an immediate offset, no branch around loads)
#pragma unroll
        for (int a = 0; a < 6; ++a) R_.xr2[a] = rp[REC + a];
        R_.b2 = rp[REC + TP::RB]; R_.invd2 = rp[REC + TP::RINVD]; R_.reg2 = rp[REC + TP::RREG];
    };
    auto relaxation = [&](unsigned iter) {
        const T ratio = (T(iter_max - 20u) - T(iter)) / T(iter_max - 20u - 30u);
        T w = T(1);
        if (ratio < T(1))
        {
            w = T(0.01);
            if (ratio > T(0)) w += (T(1) - T(0.01)) * (ratio * ratio);
        }
        return w;
    };
    Rec ring[D];
    int tp = 0;   // next visit to fetch
    static_for<0, D - 1>([&](auto dc) { fetch(tp, ring[decltype(dc)::value]); tp = tp + 1 < nv ? tp + 1 : 0; });
    bool ok = false, done = iter_max == 0;
    unsigned iter = 0;
    int tt = 0;   // visit being worked on
    T w = relaxation(0), dmax = T(0), ymax = T(0);
    while (!done)
    {
        static_for<0, D>([&](auto dc) {
            constexpr int d = decltype(dc)::value;
            if (done) return;
            fetch(tp, ring[(d + D - 1) % D]);
            tp = tp + 1 < nv ? tp + 1 : 0;
            const Rec & cur = ring[d];
            const int i = cur.i, kind = cur.kind;
            // multipliers of the row, its neighbours in the contact block (x is zero beyond m), z at the row's slot
            const T xi = x[i], xm1 = x[i > 0 ? i - 1 : 0], xp1 = x[i + 1], xp2 = x[i + 2];
            const T yp_i = ypl[i], yp_i1 = ypl[i + 1];
            T zt[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) zt[a] = z[cur.zoff + a];
            X::sync();   // (host emulation: everything is read before anything moves; lock-step on the device)
            T yz = T(0), yz2 = T(0);
#pragma unroll
            for (int a = 0; a < 6; ++a) { yz += cur.xr[a] * zt[a]; yz2 += cur.xr2[a] * zt[a]; }
            const T y = cur.b - yz - cur.reg * xi;
            T gt[NCT][6], gb;   // X^T dx on each tip (zero on the tips the row does not touch) / on the row's joint slot
            if (kind == 5)
            {
                // friction cone: rows i, i + 1 (multipliers xi, xp1), normal force xp2
                const T y1 = cur.b2 - yz2 - cur.reg2 * xp1;
                T e0 = xi * T(0), e1 = xp1 * T(0);
                if (!friction_zero)
                {
                    dmax = X::max_abs(X::max_abs(dmax, y - yp_i), y1 - yp_i1);
                    ymax = X::max_abs(X::max_abs(ymax, y), y1);
                    const T ia = fmin_(cur.invd, cur.invd2);   // 1 / max(a00, a11)
                    e0 = xi + (w * y) * ia;
                    e1 = xp1 + (w * y1) * ia;
                    const T thr = friction * xp2;
                    const T n2 = e0 * e0 + e1 * e1;
                    // (reciprocal square root + Newton steps like the register-resident solvers, jm_qcon.h)
                    const bool out = n2 > thr * thr;
                    const T scale = out ? thr * rsqrt_(out ? n2 : T(1)) : T(1);
                    e0 *= scale;
                    e1 *= scale;
                }
                const T d0 = e0 - xi, d1 = e1 - xp1;
                static_for<0, NCT>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    const bool on = cur.zoff == 6 * t;
                    const T a0 = on ? d0 : T(0), a1 = on ? d1 : T(0);
#pragma unroll
                    for (int a = 0; a < 6; ++a) gt[t][a] = cur.xr[a] * a0 + cur.xr2[a] * a1;
                });
                gb = T(0);
                // (written by the four lanes alike: same value, same address -- no exec-mask detour for a lead lane)
                x[i] = e0; x[i + 1] = e1;
                if (!friction_zero) { ypl[i] = y; ypl[i + 1] = y1; }
            }
            else
            {
                // joint bound / normal force (0), torsion (1), user joint constraint (4)
                const bool zeroed = kind == 1 && torsion_zero;   // (a row the reference zeroes without looking at its residual)
                const T er = xi + (w * y) * cur.invd;
                const T thr = C.torsion * xm1;
                T xn = fmax_(er, T(0));                            // clamp(e, 0, inf)
                xn = kind == 1 ? clamp_(er, -thr, thr) : xn;
                xn = kind == 4 ? xi + y * cur.invd : xn;
                xn = zeroed ? xi * T(0) : xn;
                if (!zeroed)
                {
                    dmax = X::max_abs(dmax, y - yp_i);
                    ymax = X::max_abs(ymax, y);
                }
                const T dx = xn - xi;
                static_for<0, NCT>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    const T a0 = cur.zoff == 6 * t ? dx : T(0);
#pragma unroll
                    for (int a = 0; a < 6; ++a) gt[t][a] = cur.xr[a] * a0;
                });
                gb = cur.xr[0] * dx;
                x[i] = xn;
                if (!zeroed) ypl[i] = y;
            }
            // z += E[:, slot of the row] X^T dx for this lane's entries
            static_for<0, NCT>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                static_for<0, ZL>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
#pragma unroll
                    for (int a = 0; a < 6; ++a) zo[j] += Er[j][6 * t + a] * gt[t][a];
                });
            });
            if (cur.zoff >= 6 * NCT)
                static_for<0, ZL>([&](auto jc) { zo[decltype(jc)::value] += Eb(decltype(jc)::value, cur.zoff - 6 * NCT) * gb; });
            static_for<0, ZL>([&](auto jc) { if (4 * decltype(jc)::value + k < NE) z[4 * decltype(jc)::value + k] = zo[decltype(jc)::value]; });
            X::sync();   // (multipliers and z in place before the next visit reads them)
            if (++tt == nv)
            {
                // end of the sweep: stagnation of the residuals (constraint_solvers.cc:263-278)
                tt = 0;
                const T tol = C.tol_abs + C.tol_rel * ymax + eps;
                if (dmax < tol) { ok = true; done = true; }
                else if (++iter == iter_max) done = true;
                else { w = relaxation(iter); dmax = T(0); ymax = T(0); }
            }
        });
    }
    X::sync();
    for (int i = k; i < m; i += 4) G(i) = x[i];
    if (lead) G(RG::OK) = ok ? T(1) : T(0);
    return true;
}
}  // namespace jm
