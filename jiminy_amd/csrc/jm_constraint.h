// jm_constraint.h -- `contacts.model = "constraint"` on the batched path: joint position bounds and
// contact points as kinematic constraints, multipliers by projected Gauss-Seidel, one robot per lane.
//
// Reference functions restated here (paths relative to the reference tree):
//   switch_constraints   computePositionLimitsForcesAlgo               core/src/engine/engine.cc:3253-3338
//                        computeContactDynamicsAtFrame (CONSTRAINT)    engine.cc:3145-3193
//   constraint rows      JointConstraint::computeJacobianAndDrift      core/src/constraints/joint_constraint.cc:139-163
//                        FrameConstraint::computeJacobianAndDrift      core/src/constraints/frame_constraint.cc:103-183
//                        Model::computeConstraints (drift kinematics)  core/src/robot/model.cc:1238-1287
//   delassus + solve     PGSSolver::SolveBoxedForwardDynamics          core/src/solver/constraint_solvers.cc:335-448
//                        computeJMinvJt / solveJMinvJtv                pinocchio_overload_algorithms.h:491-551
//   pgs                  PGSSolver::ProjectedGaussSeidelSolver / Iter  constraint_solvers.cc:107-333
//   outputs              Engine::computeAcceleration                   engine.cc:3710-3866
//   start passes         Engine::start INIT_ITERATIONS loop            engine.cc:1380-1467
//
// Formulation.  The reference factorises the dense joint-space inertia matrix (CRBA + Cholesky) and
// forms J M^-1 J^T through triangular solves with a dense Jacobian.  Here nothing of size nv x nv is
// ever formed: the unconstrained acceleration is the articulated-body solve the spring-damper path
// already runs (ABA == M^-1 (u - nle) exactly, armature included), and every column of the delassus
// matrix is one *bias-free* articulated-body solve re-using the articulated inertias of that
// evaluation (U, 1/D, liMi, the factorised root block): a unit constraint force is pushed to the root
// (6 scalars per joint) and the resulting joint accelerations are read back at the constraint rows.
// The final acceleration is the free one plus one more such solve with the multipliers applied.
// Same mathematics (A = J M^-1 J^T, b = -(drift + J a_free)), same PGS sweep order and warm start,
// hence the same iterates up to round-off.  Per lane only the m x m delassus matrix and four m-vectors
// live in HBM (caller-owned workspace, structure-of-arrays), addressed with run-time row indices; the
// tree sweeps stay fully unrolled on compile-time joint indices.
#pragma once
#include "jm_kernels.h"

// tuning switches (A/B measurements)
#ifndef JM_CON_MASKS
#define JM_CON_MASKS 1  // path-restricted sweeps of the bias-free solves
#endif
#ifndef JM_CON_REBUILD
#define JM_CON_REBUILD 1  // rebuild liMi in the bias-free solves instead of reading it back from scratch
#endif
#ifndef JM_CON_PGS_REG
#define JM_CON_PGS_REG 0   // robots with <= 32 constraint rows: PGS vectors in registers, fully unrolled sweeps
                          // (measured slower: every lane pays for all NR x NR predicated slots, + 7 kB of scratch)
#endif
#ifndef JM_CON_XALIAS
#define JM_CON_XALIAS 1  // PGS multipliers in the (unused) RK stage rows of LDS for non-RK4 launches
#endif
#ifndef JM_CON_XLDS
#define JM_CON_XLDS 0   // packed multipliers in LDS during the PGS solve (faster solve, but the extra 14 kB
                        // per block cost one resident wave per CU: measured slower on warm-started workloads)
#endif

namespace jm
{
template<class T> struct ConArgs
{
    int32_t * flags;  // [NF][B]  bit 0 enabled, bit 1 reversed
    T * data;         // [ND][B]  reference configuration per bounded joint, then lambda per row
    T * ws;           // [WTOTAL][B]
    const T * friction;  // [B] per-lane contacts.friction, or null (lane-uniform option)
    T kp, kd;         // Baumgarte gains of contacts.stabilizationFreq (abstract_constraint.cc:88-98)
    T kp_lock, kd_lock;  // ... of the user-registered constraints (jm_constraint_options::user_stabilization_freq)
    T torsion, reg, tol_abs, tol_rel;
    int iter_max;
    // set by the kernel: per-lane on-chip vector (LDS) of the packed multipliers, element p at xl[p * xstride]
    T * xl;
    int xstride;
    // optional on-chip rows for the PGS residuals y (null: workspace rows), usable when m <= yrows
    T * yl;
    int ystride, yrows;
    // runge_kutta_4 launches: the stage rows are live between two evaluations; they are parked in these
    // workspace rows around every PGS solve (null: nothing to park)
    T * park;
    int park_rows;
};
struct WithCon
{
    static constexpr bool ON = true;
    template<class T, class Tp> using WorkT = WorkC<T, Tp>;
    template<class T> using ArgsT = ConArgs<T>;
};
struct WithConA : WithCon
{
    template<class T, class Tp> using WorkT = WorkCA<T, Tp>;   // ... with applied wrenches
};

template<class Tp> struct ConRows
{
    static constexpr int count_bounded()
    {
        int n = 0;
        for (int j = 1; j < Tp::NJ; ++j) n += jt_bounded(Tp::jtype[j]) ? 1 : 0;
        return n;
    }
    static constexpr int NB = count_bounded();  // JointConstraint rows (model joint order)
    static constexpr int NC = Tp::NC;           // FrameConstraint blocks of 4 rows
    // user-registered FrameConstraints (Model::addConstraint, USER registry: last in the reference's row order,
    // model.h:43-46): 6 rows per frame (x, y, z, rot x, rot y, rot z; world aligned), of which the dofs of the compile-time
    // mask `Tp::xframe_mask` can be active; unbounded rows (nBlocks = 0, constraint_solvers.cc:112-128)
    static constexpr int NX = Tp::NX;
    static constexpr int XR0 = NB + 4 * NC;     // first user-frame row
    // user-registered JointConstraints on rows of their own (`Tp::xjoint`: declared 1-dof joints), behind the frames: one
    // unbounded row each, next to the joint's bound constraint like in the reference (model.cc:884-905)
    static constexpr int NXJ = Tp::NXJ;
    static constexpr int XJ0 = XR0 + 6 * NX;    // first user-joint row
    static constexpr int NR = NB + 4 * NC + 6 * NX + NXJ;
    static constexpr int NF = NB + NC + NX + NXJ;
    static constexpr int ND = NB + NR + 12 * NX + NXJ;
    static constexpr int LAM = NB;  // first lambda row in `data`
    static constexpr int XREF = NB + NR;        // FrameConstraint::transformRef_ of every user frame: translation 3, rotation 9
    static constexpr int XJREF = XREF + 12 * NX;  // JointConstraint::configurationRef_ of every user joint constraint
    static constexpr bool USER = NX > 0 || NXJ > 0;
    static_assert(!USER || NR <= 64, "user constraints: the packed row masks of the solve are 64-bit");
    static_assert(NB <= 64, "the packed mask of the locked bound rows is 64-bit");
    // row of the user joint constraint on joint j, or -1
    static constexpr int xjoint_row(int j)
    {
        for (int k = 0; k < NXJ; ++k)
            if (Tp::xjoint[k] == j) return k;
        return -1;
    }
    // lowest dof of the mask of user frame x (its first row that can be active)
    static constexpr int xfirst(int x)
    {
        for (int d = 0; d < 6; ++d)
            if ((Tp::xframe_mask[x] >> d) & 1) return d;
        return 0;
    }
    // workspace rows: delassus matrix over the PACKED active rows (stride NR), then b, y, y_prev, a diagonal
    // backup and the packed multipliers x
    static constexpr int WA = 0, WB = NR * NR, WY = WB + NR, WYP = WY + NR, WD = WYP + NR, WX = WD + NR, WPARK = WX + NR,
                         WTOTAL = WPARK + 3 * Tp::NV;  // + the parked Runge-Kutta stage rows (3 nv)
    // bit mask of the ancestors-or-self of joint j (the joints a force applied on body j travels through)
    static constexpr unsigned long long anc_mask(int j)
    {
        unsigned long long m = 0ull;
        for (int i = j; i > 0; i = Tp::parent[i]) m |= 1ull << i;
        return m;
    }
    static constexpr int bjoint(int k)
    {
        int n = 0;
        for (int j = 1; j < Tp::NJ; ++j)
            if (jt_bounded(Tp::jtype[j]))
            {
                if (n == k) return j;
                ++n;
            }
        return 0;
    }
};

JM_DEV double cabs_(double x) { return __builtin_fabs(x); }
JM_DEV float cabs_(float x) { return __builtin_fabsf(x); }

template<int NW> struct RowMaskN
{
    unsigned long long w[NW];
    JM_DEV void clear()
    {
#pragma unroll
        for (int i = 0; i < NW; ++i) w[i] = 0ull;
    }
    // (value-level selects over the words: a run-time index would turn the mask into a private-memory array)
    JM_DEV bool test(int r) const
    {
        unsigned long long x = w[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) x = (r >> 6) == i ? w[i] : x;
        return (x >> (r & 63)) & 1ull;
    }
    JM_DEV void set(int r)
    {
        const unsigned long long bit = 1ull << (r & 63);
#pragma unroll
        for (int i = 0; i < NW; ++i) w[i] |= (NW == 1 || (r >> 6) == i) ? bit : 0ull;
    }
    JM_DEV bool any() const
    {
        unsigned long long o = 0ull;
#pragma unroll
        for (int i = 0; i < NW; ++i) o |= w[i];
        return o != 0ull;
    }
    JM_DEV int count() const
    {
        int n = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) n += __builtin_popcountll(w[i]);
        return n;
    }
    // number of set bits below row r = packed index of row r among the active rows
    JM_DEV int rank(int r) const
    {
        int n = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i)
        {
            const int lo = r - 64 * i;  // bits of word i below r
            const unsigned long long m = lo >= 64 ? ~0ull : (lo <= 0 ? 0ull : ((1ull << lo) - 1ull));
            n += __builtin_popcountll(w[i] & m);
        }
        return n;
    }
    // index of the lowest set bit, which is cleared (the mask must not be empty)
    JM_DEV int pop_lowest()
    {
        int r = 0;
        bool found = false;
#pragma unroll
        for (int i = 0; i < NW; ++i)
            if (!found && w[i] != 0ull)
            {
                r = 64 * i + __builtin_ctzll(w[i]);
                w[i] &= w[i] - 1ull;
                found = true;
            }
        return r;
    }
};

// visits every contact point: compile-time parent joint `jc`, run-time contact index `c` (robots with many
// contact points per body -- Atlas: 16 per foot -- would otherwise unroll every loop per point)
template<class Tp> constexpr bool joint_has_contact(int j)
{
    for (int c = 0; c < Tp::NC; ++c)
        if (Tp::contact_joint[c] == j) return true;
    return false;
}
template<class Tp> constexpr bool joint_has_xframe(int j)
{
    for (int x = 0; x < Tp::NX; ++x)
        if (Tp::xframe_joint[x] == j) return true;
    return false;
}
template<class Tp> constexpr bool joint_has_xframe2(int j)   // second frame of a DistanceConstraint
{
    for (int x = 0; x < Tp::NX; ++x)
        if (Tp::xframe_kind[x] == JM_XKIND_DISTANCE && Tp::xframe_joint2[x] == j) return true;
    return false;
}
template<class Tp, class F> JM_DEV void for_contacts(F && f)
{
    if constexpr (Tp::NC > 0)
        static_for<1, Tp::NJ>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (joint_has_contact<Tp>(j))
            {
#pragma nounroll
                for (int c = 0; c < Tp::NC; ++c)
                    if (Tp::contact_joint[c] == j) f(jc, c);
            }
        });
}

// liMi of a 1-dof joint rebuilt from the cached joint coordinate (WorkC::jcs) and the constant placement
// (scalar loads): the column solves are bound by the traffic of the spilled working set, not by flops
template<class T, class Tp, int J, class WC> JM_DEV SE3<T> limi_rebuilt(CPtr<T> P, const WC & w)
{
#if JM_CON_REBUILD
    using L = Layout<Tp>;
    constexpr int t = Tp::jtype[J];
    if constexpr (jt_is_sph(t)) return w.liMi[J];
    const SE3<T> plc = joint_placement<T, Tp, J>(P, w);
    SE3<T> Mj;
    if constexpr (jt_is_rev(t))
    {
        constexpr int ax = jt_axis(t);
        if constexpr (ax >= 0) Mj.R = rot_axis<T>(ax, w.jcs[J][0], w.jcs[J][1]);
        else Mj.R = rot_rodrigues(joint_axis<T, Tp, J>(P), w.jcs[J][0], w.jcs[J][1]);
        Mj.p = zero3<T>();
    }
    else
    {
        Mj.R = ident3<T>();
        Mj.p = w.jcs[J][0] * joint_axis<T, Tp, J>(P);
    }
    return plc * Mj;
#else
    return w.liMi[J];
#endif
}

// depth of a joint in the tree (joints hanging from the universe: 1)
template<class Tp> constexpr int joint_depth(int j)
{
    int d = 0;
    for (int i = j; i > 0; i = Tp::parent[i]) ++d;
    return d;
}
template<class Tp> constexpr int max_depth()
{
    int m = 1;
    for (int j = 1; j < Tp::NJ; ++j) m = joint_depth<Tp>(j) > m ? joint_depth<Tp>(j) : m;
    return m;
}

// dd = M^-1 (tau + sum_j J_j^T fb_j): bias-free articulated-body solve with the articulated inertias of the
// last eval_dynamics.  `tau(ic)` joint efforts, `fb(jc)` force applied ON body j (joint frame).  The joint
// accelerations and the spatial acceleration of every joint are handed to `visit(jc, ddj, da_j)` (ddj points to
// the nv_j accelerations of joint j) as the root-to-leaves sweep produces them.
// Joints are numbered depth-first (parents before children, a subtree is contiguous), so neither sweep needs a
// per-joint array: the leaves-to-root sweep keeps ONE bias-force accumulator per tree depth (a joint collects
// its children's contributions in the slot of its own depth and hands its total to the slot above), the
// root-to-leaves sweep ONE spatial acceleration per depth.  That is 2 x depth x 6 scalars of temporaries
// instead of 2 x njoints x 6 -- the temporaries of these solves were what the kernel spilled most.
// `bmask`: joints that carry a non-zero bias force / effort in the leaves-to-root sweep (bit j), `fmask`:
// joints whose acceleration is wanted; the others are skipped.
template<class T, class Tp, class WC, class FT, class FB, class VIS>
JM_DEV void delta_sweeps(CPtr<T> P, const WC & w, FT && tau, FB && fb, VIS && visit,
                         unsigned long long bmask = ~0ull, unsigned long long fmask = ~0ull)
{
    constexpr int NJ = Tp::NJ;
    constexpr int MD = max_depth<Tp>();
    static_assert(NJ <= 64, "joint masks are 64-bit");
    Sp<T> acc[MD + 1];
    T ur[Tp::NV];
    static_for<0, MD + 1>([&](auto dc) { acc[decltype(dc)::value] = zero6<T>(); });
    static_for<0, Tp::NV>([&](auto ic) { ur[decltype(ic)::value] = tau(ic); });
    static_rfor<1, NJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int p = Tp::parent[j];
        constexpr int t = Tp::jtype[j];
        constexpr int iv = Tp::idx_v[j];
        constexpr int d = joint_depth<Tp>(j);
        if constexpr (t == JM_JT_FREEFLYER)
        {
            const Sp<T> pf = acc[d] - fb(jc);
            acc[d] = zero6<T>();
            ur[iv] -= pf.l.x; ur[iv + 1] -= pf.l.y; ur[iv + 2] -= pf.l.z;
            ur[iv + 3] -= pf.a.x; ur[iv + 4] -= pf.a.y; ur[iv + 5] -= pf.a.z;
        }
        else if constexpr (jt_is_sph(t))
        {
            // spherical joint: u -= S^T f (the angular part), pa = pf + U Dinv u with U = [B; D] of the last evaluation
            constexpr int k = spherical_rank<Tp>(j);
            const Sp<T> pf = acc[d] - fb(jc);
            acc[d] = zero6<T>();
            ur[iv] -= pf.a.x; ur[iv + 1] -= pf.a.y; ur[iv + 2] -= pf.a.z;
            if constexpr (p > 0)
            {
                const V3<T> t3 = w.sphDinv[k] * V3<T>{ur[iv], ur[iv + 1], ur[iv + 2]};
                const Sp<T> pa = {pf.l + w.sphB[k] * t3, pf.a + w.sphD[k] * t3};
                acc[d - 1] = acc[d - 1] + act_force(limi_rebuilt<T, Tp, j>(P, w), pa);
            }
        }
        else if (!JM_CON_MASKS || ((bmask >> j) & 1ull))
        {
            const Sp<T> pf = acc[d] - fb(jc);
            acc[d] = zero6<T>();
            const T uj = ur[iv] - joint_St_dot<T, Tp, j>(P, pf);
            ur[iv] = uj;
            if constexpr (p > 0)
            {
                const T ud = uj * sweep_dinv<T, Tp, j>(w);
                const Sp<T> Uj = sweep_U<T, Tp, j>(w);
                const Sp<T> pa = {pf.l + ud * Uj.l, pf.a + ud * Uj.a};
                acc[d - 1] = acc[d - 1] + act_force(limi_rebuilt<T, Tp, j>(P, w), pa);
            }
        }
    });
    Sp<T> lvl[MD + 1];
    static_for<1, NJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int p = Tp::parent[j];
        constexpr int t = Tp::jtype[j];
        constexpr int iv = Tp::idx_v[j];
        constexpr int d = joint_depth<Tp>(j);
        if constexpr (t == JM_JT_FREEFLYER)
        {
            T b[6] = {ur[iv], ur[iv + 1], ur[iv + 2], ur[iv + 3], ur[iv + 4], ur[iv + 5]};
            chol6_resolve(w.rootA, w.rootdinv, b);
            lvl[d] = {{b[0], b[1], b[2]}, {b[3], b[4], b[5]}};
            visit(jc, b, lvl[d]);
        }
        else if constexpr (jt_is_sph(t))
        {
            constexpr int k = spherical_rank<Tp>(j);
            Sp<T> ag;
            if constexpr (p > 0) ag = actinv_motion(limi_rebuilt<T, Tp, j>(P, w), lvl[d - 1]);
            else ag = zero6<T>();
            const V3<T> Ua = tmul(w.sphB[k], ag.l) + w.sphD[k] * ag.a;
            const V3<T> dd = w.sphDinv[k] * (V3<T>{ur[iv], ur[iv + 1], ur[iv + 2]} - Ua);
            const T b[3] = {dd.x, dd.y, dd.z};
            lvl[d] = {ag.l, ag.a + dd};
            visit(jc, b, lvl[d]);
        }
        else if (!JM_CON_MASKS || ((fmask >> j) & 1ull))
        {
            Sp<T> ag;
            if constexpr (p > 0) ag = actinv_motion(limi_rebuilt<T, Tp, j>(P, w), lvl[d - 1]);
            else ag = zero6<T>();
            const Sp<T> Uj = sweep_U<T, Tp, j>(w);
            const T Ua = dot(Uj.l, ag.l) + dot(Uj.a, ag.a);
            const T ddj = sweep_dinv<T, Tp, j>(w) * (ur[iv] - Ua);
            const V3<T> n = joint_axis<T, Tp, j>(P);
            if constexpr (jt_is_rev(t)) lvl[d] = {ag.l, ag.a + ddj * n};
            else lvl[d] = {ag.l + ddj * n, ag.a};
            visit(jc, &ddj, lvl[d]);
        }
    });
}

// index of the joint-bound constraint row of joint j, or -1
template<class Tp> constexpr int bound_row_of(int j)
{
    for (int k = 0; k < ConRows<Tp>::NB; ++k)
        if (ConRows<Tp>::bjoint(k) == j) return k;
    return -1;
}

// Cholesky solve A x = b over the m packed rows (start pass with `ignoreBounds`, solveJMinvJtv).
// The lower triangle and the diagonal of A are used as factor storage and restored afterwards
// (diagonal from a backup, lower triangle mirrored from the upper one like constraint_solvers.cc:421).
template<class T, class Tp, class WS>
JM_DEV bool chol_solve_packed(int m, WS && ws)
{
    using R = ConRows<Tp>;
    constexpr int NR = R::NR;
    bool ok = true;
    for (int j = 0; j < m; ++j)
    {
        ws(R::WD + j) = ws(R::WA + j * NR + j);
        T s = ws(R::WA + j * NR + j);
        for (int k = 0; k < j; ++k) { const T l = ws(R::WA + j * NR + k); s -= l * l; }
        ok &= s > T(0);
        const T d = sqrt_(s);
        ws(R::WA + j * NR + j) = d;
        for (int i = j + 1; i < m; ++i)
        {
            T t = ws(R::WA + i * NR + j);
            for (int k = 0; k < j; ++k) t -= ws(R::WA + i * NR + k) * ws(R::WA + j * NR + k);
            ws(R::WA + i * NR + j) = t / d;
        }
    }
    for (int i = 0; i < m; ++i)
    {
        T s = ws(R::WB + i);
        for (int k = 0; k < i; ++k) s -= ws(R::WA + i * NR + k) * ws(R::WX + k);
        ws(R::WX + i) = s / ws(R::WA + i * NR + i);
    }
    for (int i = m - 1; i >= 0; --i)
    {
        T s = ws(R::WX + i);
        for (int k = i + 1; k < m; ++k) s -= ws(R::WA + k * NR + i) * ws(R::WX + k);
        ws(R::WX + i) = s / ws(R::WA + i * NR + i);
    }
    for (int i = 0; i < m; ++i)
    {
        ws(R::WA + i * NR + i) = ws(R::WD + i);
        for (int k = 0; k < i; ++k) ws(R::WA + i * NR + k) = ws(R::WA + k * NR + i);
    }
    return ok;
}

// PGSSolver::ProjectedGaussSeidelSolver (constraint_solvers.cc:107-333) over the m packed rows:
// the first `nb` rows are joint bounds, then blocks of 4 rows (x, y, z, torsion) per active contact.
// `lockp`: bit p = packed bound row p is a user-registered JointConstraint (Model::addConstraint): unbounded, solved first in
// every sweep, coefficient by coefficient, without relaxation or projection (constraint_solvers.cc:112-128).
// `mc`: packed index of the first row behind the contact blocks (user FrameConstraint rows, all of them unbounded: bits of
// `lockp` too); < 0 = none, the contact blocks run up to m.
template<class T, class Tp, class WS>
JM_DEV bool pgs_solve_packed(const ConArgs<T> & C, T friction, int m, int nb, WS && ws, unsigned long long lockp = 0ull, int mc = -1)
{
    if (mc < 0) mc = m;
    using R = ConRows<Tp>;
    constexpr int NR = R::NR;
    const T eps = Eps<T>::eps;
    // the packed multipliers live on chip for the duration of the solve (they are read m times per row
    // update); the delassus column is fetched 8 entries at a time so that the loads are in flight together
    // (the sum itself keeps the sequential order of `A.col(i).dot(x)`)
    T * const xl = C.xl;
    const int xs = C.xstride;
    for (int r = 0; r < m; ++r) xl[r * xs] = ws(R::WX + r);
    auto col_dot = [&](int i) {
        T s = T(0);
        int k = 0;
        for (; k + 8 <= m; k += 8)
        {
            T av[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) av[u] = ws(R::WA + (k + u) * NR + i);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += av[u] * xl[(k + u) * xs];
        }
        if (k + 4 <= m)
        {
            T av[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) av[u] = ws(R::WA + (k + u) * NR + i);
#pragma unroll
            for (int u = 0; u < 4; ++u) s += av[u] * xl[(k + u) * xs];
            k += 4;
        }
        for (; k < m; ++k) s += ws(R::WA + k * NR + i) * xl[k * xs];
        return s;
    };
    // residuals y: on chip when the kernel lent enough rows, else in the workspace; the reference's test on
    // |y - yPrev| (yPrev = copy taken before the sweep) is evaluated where y is rewritten -- a row that the
    // sweep does not touch contributes 0 either way -- so no copy is kept
    const bool y_chip = C.yl != nullptr && m <= C.yrows;
    auto Y = [&](int r) -> T & { return y_chip ? C.yl[r * C.ystride] : ws(R::WY + r); };
    for (int r = 0; r < m; ++r) Y(r) = T(0);
    const bool torsion_zero = C.torsion < eps, friction_zero = friction < eps;
    const unsigned iter_max = (unsigned)C.iter_max;
    for (unsigned iter = 0; iter < iter_max; ++iter)
    {
        T dmax = T(0);  // max |y - yPrev| of this sweep
        // under-relaxation schedule (constraint_solvers.cc:248-258)
        const T ratio = (T(iter_max - 20u) - T(iter)) / T(iter_max - 20u - 30u);
        T w = T(1);
        if (ratio < T(1))
        {
            w = T(0.01);
            if (ratio > T(0)) w += (T(1) - T(0.01)) * (ratio * ratio);
        }
        if (lockp)
            for (int r = 0; r < m; ++r)
            {
                if (r == nb) r = mc;   // (no unbounded row among the contact blocks)
                if (r >= m) break;
                if (!((lockp >> r) & 1ull)) continue;
                const T y = ws(R::WB + r) - col_dot(r);
                dmax = fmax_(dmax, cabs_(y - Y(r)));
                Y(r) = y;
                xl[r * xs] = xl[r * xs] + y / ws(R::WA + r * NR + r);
            }
        // block 0 of every constraint: joint bounds, then the normal force of every contact
        for (int r = 0; r < mc; r += (r < nb ? 1 : 4))
        {
            if (r < nb && ((lockp >> r) & 1ull)) continue;
            const int i0 = r < nb ? r : r + 2;
            const T y = ws(R::WB + i0) - col_dot(i0);
            dmax = fmax_(dmax, cabs_(y - Y(i0)));
            Y(i0) = y;
            const T e = xl[(i0) * xs] + w * y / ws(R::WA + i0 * NR + i0);
            xl[(i0) * xs] = fmax_(e, T(0));  // clamp(e, 0, inf)
        }
        // block 1: torsional friction {3, 2}
        for (int r = nb; r < mc; r += 4)
        {
            if (torsion_zero) { xl[(r + 3) * xs] = xl[(r + 3) * xs] * T(0); continue; }
            const int i0 = r + 3;
            const T y = ws(R::WB + i0) - col_dot(i0);
            dmax = fmax_(dmax, cabs_(y - Y(i0)));
            Y(i0) = y;
            const T e = xl[(i0) * xs] + w * y / ws(R::WA + i0 * NR + i0);
            const T thr = C.torsion * xl[(r + 2) * xs];
            xl[(i0) * xs] = clamp_(e, -thr, thr);
        }
        // block 2: friction cone {0, 1, 2}
        for (int r = nb; r < mc; r += 4)
        {
            if (friction_zero) { xl[(r) * xs] = xl[(r) * xs] * T(0); xl[(r + 1) * xs] = xl[(r + 1) * xs] * T(0); continue; }
            const T y0 = ws(R::WB + r) - col_dot(r);
            dmax = fmax_(dmax, cabs_(y0 - Y(r)));
            Y(r) = y0;
            const T y1 = ws(R::WB + r + 1) - col_dot(r + 1);
            dmax = fmax_(dmax, cabs_(y1 - Y(r + 1)));
            Y(r + 1) = y1;
            const T a00 = ws(R::WA + r * NR + r), a11 = ws(R::WA + (r + 1) * NR + r + 1);
            const T a_max = a11 > a00 ? a11 : a00;
            T e0 = xl[(r) * xs] + w * y0 / a_max;
            T e1 = xl[(r + 1) * xs] + w * y1 / a_max;
            const T thr = friction * xl[(r + 2) * xs];
            const T n2 = e0 * e0 + e1 * e1;
            if (n2 > thr * thr)
            {
                const T scale = thr / sqrt_(n2);
                e0 *= scale;
                e1 *= scale;
            }
            xl[(r) * xs] = e0;
            xl[(r + 1) * xs] = e1;
        }
        // stagnation of the residuals (constraint_solvers.cc:263-278)
        T ymax = T(0);
        for (int r = 0; r < m; ++r) ymax = fmax_(ymax, cabs_(Y(r)));
        const T tol = C.tol_abs + C.tol_rel * ymax + eps;
        const bool done = dmax < tol;  // all |y - yPrev| < tol
        if (done)
        {
            for (int r = 0; r < m; ++r) ws(R::WX + r) = xl[r * xs];
            return true;
        }
    }
    for (int r = 0; r < m; ++r) ws(R::WX + r) = xl[r * xs];
    return false;
}

// The same solve for robots with at most 32 constraint rows (ANYmal: 28): the sweeps are unrolled on
// compile-time packed indices, so that x, b, y live in registers (no store -> load round trip through memory
// between two row updates of the Gauss-Seidel chain) and every delassus entry sits at a constant offset: the
// loads of the next rows do not depend on the multipliers being updated and are issued ahead of them.
// Rows beyond the lane's m and rows of another block are skipped by run-time predicates; the arithmetic and
// its order are those of pgs_solve_packed.
template<class T, class Tp, class WS>
JM_DEV bool pgs_solve_regs(const ConArgs<T> & C, T friction, int m, int nb, WS && ws)
{
    using R = ConRows<Tp>;
    constexpr int NR = R::NR;
    const T eps = Eps<T>::eps;
    T x[NR], bb[NR], y[NR], yp[NR];
    static_for<0, NR>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        x[p] = p < m ? ws(R::WX + p) : T(0);
        bb[p] = p < m ? ws(R::WB + p) : T(0);
        y[p] = T(0);
    });
    auto col_dot = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        T s = T(0);
        static_for<0, NR>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const T a = k < m ? ws(R::WA + k * NR + i) : T(0);
            s += a * x[k];
        });
        return s;
    };
    const bool torsion_zero = C.torsion < eps, friction_zero = friction < eps;
    const unsigned iter_max = (unsigned)C.iter_max;
    bool converged = false;
#pragma nounroll
    for (unsigned iter = 0; iter < iter_max && !converged; ++iter)
    {
        static_for<0, NR>([&](auto pc) { yp[decltype(pc)::value] = y[decltype(pc)::value]; });
        const T ratio = (T(iter_max - 20u) - T(iter)) / T(iter_max - 20u - 30u);
        T w = T(1);
        if (ratio < T(1))
        {
            w = T(0.01);
            if (ratio > T(0)) w += (T(1) - T(0.01)) * (ratio * ratio);
        }
        // the three block passes of the reference (0: bounds + normals, 1: torsion, 2: friction cones) as a
        // run-time loop around ONE unrolled visit of the rows: a row acts in the pass its kind belongs to.
        // The residual of the first row of a friction pair is parked until its partner has its own (both
        // are taken before either multiplier moves, constraint_solvers.cc:175-195).
#pragma nounroll
        for (int blk = 0; blk < 3; ++blk)
        {
            T y0_pair = T(0);
            static_for<0, NR>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                const int dof = (p - nb) & 3;  // 0..3 = x, y, z (normal), torsion for the rows of a contact
                const bool bound = p < nb;
                const int kind = bound ? 0 : (dof == 2 ? 0 : (dof == 3 ? 1 : 2));
                if (p < m && kind == blk)
                {
                    if (blk == 1 && torsion_zero) x[p] = x[p] * T(0);
                    else if (blk == 2 && friction_zero) x[p] = x[p] * T(0);
                    else
                    {
                        const T yy = bb[p] - col_dot(pc);
                        y[p] = yy;
                        const T app = ws(R::WA + p * NR + p);
                        if (blk == 0) x[p] = fmax_(x[p] + w * yy / app, T(0));
                        else if (blk == 1)
                        {
                            if constexpr (p >= 1)
                            {
                                const T thr = C.torsion * x[p - 1];
                                x[p] = clamp_(x[p] + w * yy / app, -thr, thr);
                            }
                        }
                        else if (dof == 0) y0_pair = yy;
                        else
                        {
                            if constexpr (p >= 1 && p + 1 < NR)
                            {
                                const T a00 = ws(R::WA + (p - 1) * NR + p - 1);
                                const T a_max = app > a00 ? app : a00;
                                T e0 = x[p - 1] + w * y0_pair / a_max;
                                T e1 = x[p] + w * yy / a_max;
                                const T thr = friction * x[p + 1];
                                const T n2 = e0 * e0 + e1 * e1;
                                if (n2 > thr * thr)
                                {
                                    const T scale = thr / sqrt_(n2);
                                    e0 *= scale;
                                    e1 *= scale;
                                }
                                x[p - 1] = e0;
                                x[p] = e1;
                            }
                        }
                    }
                }
            });
        }
        T ymax = T(0);
        static_for<0, NR>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            if (p < m) ymax = fmax_(ymax, cabs_(y[p]));
        });
        const T tol = C.tol_abs + C.tol_rel * ymax + eps;
        bool done = true;
        static_for<0, NR>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            if (p < m) done &= cabs_(y[p] - yp[p]) < tol;
        });
        converged = done;
    }
    static_for<0, NR>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        if (p < m) ws(R::WX + p) = x[p];
    });
    return converged;
}

// ---- kinds of user constraint frames (JM_XKIND_*): what differs from the plain FrameConstraint rows
// SphereConstraint / WheelConstraint: radius * (direction from the contact point to the frame origin), world aligned:
// skewRadius_ = [rd]x (sphere_constraint.cc:96-101: the ground normal; wheel_constraint.cc:96-101: y, which tilts with the wheel)
template<class T, class Tp, int X, class W> JM_DEV V3<T> xframe_rd(CPtr<T> P, const W & w)
{
    using L = Layout<Tp>;
    constexpr int kind = Tp::xframe_kind[X];
    if constexpr (kind == JM_XKIND_SPHERE) return P[L::XPAR + 8 * X] * ld_v3<T>(P, L::XPAR + 8 * X + 1);
    else if constexpr (kind == JM_XKIND_WHEEL)
    {
        constexpr int j = Tp::xframe_joint[X];
        const V3<T> nrm = ld_v3<T>(P, L::XPAR + 8 * X + 1);
        const V3<T> axis = w.oMi[j].R * (ld_m3<T>(P, L::XFRAME + 12 * X) * ld_v3<T>(P, L::XPAR + 8 * X + 4));
        const V3<T> x = cross(cross(axis, nrm), axis);
        return (P[L::XPAR + 8 * X] * rsqrt_(dot(x, x))) * x;
    }
    else return zero3<T>();
}
// DistanceConstraint: world positions of the two frame origins
template<class T, class Tp, int X, class W> JM_DEV void xframe_points(CPtr<T> P, const W & w, V3<T> & p1, V3<T> & p2)
{
    using L = Layout<Tp>;
    constexpr int j1 = Tp::xframe_joint[X], j2 = Tp::xframe_joint2[X];
    p1 = w.oMi[j1].p + w.oMi[j1].R * ld_v3<T>(P, L::XFRAME + 12 * X + 9);
    p2 = w.oMi[j2].p + w.oMi[j2].R * ld_v3<T>(P, L::XPAR + 8 * X + 1);
}

// (WC: WorkC, or WorkCA in the instantiation that reads the applied wrenches)
template<class T, class Tp, class CA, class WC>
JM_DEV void eval_constrained(CPtr<T> P, const T * q, const T * v, const T * cmd, WC & w, const CA & C,
                             long long lane, long long B, int start_passes)
{
    using L = Layout<Tp>;
    using R = ConRows<Tp>;
    constexpr int NJ = Tp::NJ, NV = Tp::NV, NR = R::NR;
    // `start_passes` < 0: refresh only -- no switching, no solve: the stored multipliers (those of the last
    // evaluation, which an adaptive step took at this very state) are applied to the free acceleration, so
    // that the outputs are the ones the reference reads after its last evaluation (engine.cc:2143-2151)
    const bool refresh = start_passes < 0;
    // ---- unconstrained part: FK, motors (the ABA without contact forces follows the switching: lanes without an
    // active constraint -- the common case -- run it in a branch of their own that nothing of the solve lives across)
    eval_kinematics<T, Tp>(P, q, v, cmd, w);
    w.status &= ~JM_LANE_SOLVER_FAILURE;
    if constexpr (NR == 0)
    {
        eval_aba<T, Tp>(P, v, w);
        return;
    }
    auto flag = [&](int r) -> int32_t & { return C.flags[(size_t)r * B + lane]; };
    auto dat = [&](int r) -> T & { return C.data[(size_t)r * B + lane]; };
    auto lam = [&](int r) -> T & { return C.data[(size_t)(R::LAM + r) * B + lane]; };
    auto ws = [&](int r) -> T & { return C.ws[(size_t)r * B + lane]; };
    const T eps_tr = P[L::OPT + 9];
    const T friction = C.friction ? C.friction[lane] : P[L::OPT + 8];
    const V3<T> g = ld_v3<T>(P, L::OPT), gw = ld_v3<T>(P, L::OPT + 3);

    // ---- constraint switching
    constexpr int NWORDS = ((NR + 63) / 64 > 0) ? (NR + 63) / 64 : 1;
    using RowMask = RowMaskN<NWORDS>;
    RowMask act, rev, lck;
    act.clear();
    rev.clear();
    lck.clear();
    static_for<0, R::NB>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int iq = Tp::idx_q[R::bjoint(k)];
        // state of the constraint: loaded once, updated in registers, stored once
        // Engine::start: JointConstraint::reset + enable, not reversed (engine.cc:1266-1308)
        const bool init = start_passes > 0;
        const int32_t f0 = flag(k);
        // bit 2: a user-registered JointConstraint holds the row (Model::addConstraint, model.cc:926-936): always enabled,
        // never reversed, no switching while it holds; its reference is the configuration at `start` (or the caller's)
        const bool locked = (f0 & 4) != 0;
        int32_t f = init ? (1 | (f0 & 4)) : f0;
        if (!refresh)
        {
            const T qj = q[iq], lo = P[L::QLO + iq], hi = P[L::QHI + iq];
            T ref = init ? qj : dat(k);
            bool clear = init;
            if (locked) f = 5;
            else if (hi < qj || qj < lo)
            {
                ref = clamp_(qj, lo, hi);
                f = 1 | (hi < qj ? 2 : 0);
            }
            else if (lo + eps_tr < qj && qj < hi - eps_tr)
            {
                f &= ~1;
                clear = true;  // AbstractConstraintBase::disable
            }
            flag(k) = f;
            dat(k) = ref;
            if (clear) lam(k) = T(0);
        }
        if (f & 1) act.set(k);
        if (f & 2) rev.set(k);
        if (f & 4) lck.set(k);
    });
    for_contacts<Tp>([&](auto jc, int c) {
        constexpr int j = decltype(jc)::value;
        const int r0 = R::NB + 4 * c;
        const V3<T> pc = ld_v3<T>(P, L::CONTACT + 12 * c + 9);
        const T d = w.oMi[j].p.z + dot(V3<T>{w.oMi[j].R.m20, w.oMi[j].R.m21, w.oMi[j].R.m22}, pc);
        const bool init = start_passes > 0;
        int32_t f = init ? 1 : flag(R::NB + c);
        if (!refresh)
        {
            bool clear = init;
            if (d < T(0)) f = 1;
            else if (d > eps_tr)
            {
                f = 0;
                clear = true;
            }
            if (clear)
            {
#pragma unroll
                for (int i = 0; i < 4; ++i) lam(r0 + i) = T(0);
            }
            flag(R::NB + c) = f;
        }
        if (f & 1) { act.set(r0); act.set(r0 + 1); act.set(r0 + 2); act.set(r0 + 3); }
    });
    // user-registered FrameConstraints: never switched (flag bit 0 = this lane's robot holds it, set by the caller);
    // Engine::start -> FrameConstraint::reset: transformRef_ = the frame's pose, multipliers zeroed (frame_constraint.cc:77-101)
    RowMask xact;   // first row of every held user frame
    xact.clear();
    static_for<0, R::NX>([&](auto xc) {
        constexpr int x = decltype(xc)::value;
        constexpr int j = Tp::xframe_joint[x];
        constexpr int mask = Tp::xframe_mask[x];
        if (flag(R::NB + R::NC + x) & 1)
        {
            if (start_passes > 0)
            {
                SE3<T> oMf = w.oMi[j] * ld_se3<T>(P, L::XFRAME + 12 * x);
                if constexpr (Tp::xframe_kind[x] == JM_XKIND_DISTANCE)
                {
                    // DistanceConstraint::reset: distanceRef_ = |p_1 - p_2| in the first slot (distance_constraint.cc:73-77)
                    V3<T> p1, p2;
                    xframe_points<T, Tp, x>(P, w, p1, p2);
                    oMf = {ident3<T>(), {sqrt_(dot(p1 - p2, p1 - p2)), T(0), T(0)}};
                }
                const T t[12] = {oMf.p.x, oMf.p.y, oMf.p.z, oMf.R.m00, oMf.R.m01, oMf.R.m02, oMf.R.m10, oMf.R.m11, oMf.R.m12,
                                 oMf.R.m20, oMf.R.m21, oMf.R.m22};
#pragma unroll
                for (int i = 0; i < 12; ++i) dat(R::XREF + 12 * x + i) = t[i];
#pragma unroll
                for (int d = 0; d < 6; ++d) lam(R::XR0 + 6 * x + d) = T(0);
            }
            xact.set(R::XR0 + 6 * x + R::xfirst(x));
            static_for<0, 6>([&](auto dc) {
                constexpr int d = decltype(dc)::value;
                if constexpr ((mask >> d) & 1) act.set(R::XR0 + 6 * x + d);
            });
        }
    });
    // user-registered JointConstraints on rows of their own: like the frames, held where the caller's flag says so;
    // JointConstraint::reset at `start`: reference = the joint position, multiplier zeroed (joint_constraint.cc:56-83)
    static_for<0, R::NXJ>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if (flag(R::NB + R::NC + R::NX + k) & 1)
        {
            if (start_passes > 0)
            {
                dat(R::XJREF + k) = q[Tp::idx_q[Tp::xjoint[k]]];
                lam(R::XJ0 + k) = T(0);
            }
            act.set(R::XJ0 + k);
        }
    });
    if (!act.any())
    {
        // Engine::computeAcceleration: plain ABA (engine.cc:3861-3865)
#ifndef JM_HOST_EMU
        asm volatile("; free lane: sweeps of its own");   // (keeps this copy from being merged with the one below)
#endif
        eval_aba<T, Tp>(P, v, w);
        return;
    }
    eval_aba<T, Tp>(P, v, w);
    // joints whose acceleration some active row reads (columns of the delassus matrix)
    unsigned long long fmask = 0ull;
    static_for<0, R::NXJ>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if (act.test(R::XJ0 + k)) fmask |= R::anc_mask(Tp::xjoint[k]);
    });
    static_for<0, R::NX>([&](auto xc) {
        constexpr int x = decltype(xc)::value;
        if (xact.test(R::XR0 + 6 * x + R::xfirst(x))) fmask |= R::anc_mask(Tp::xframe_joint[x]) | R::anc_mask(Tp::xframe_joint2[x]);
    });
    static_for<0, R::NB>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if (act.test(k)) fmask |= R::anc_mask(R::bjoint(k));
    });
    for_contacts<Tp>([&](auto jc, int c) {
        if (act.test(R::NB + 4 * c)) fmask |= R::anc_mask(decltype(jc)::value);
    });

    // ---- delassus matrix, one bias-free articulated-body solve per active row
    // (each lane walks ITS OWN active rows, lowest first: a wave runs max-over-lanes solves, not the
    // union of the rows active anywhere in the wave; column index = packed index of the row)
    const int m_act = act.count(), nb_act = act.rank(R::NB), mc_act = act.rank(R::XR0);
    RowMask rem = act;
#pragma nounroll
    for (int pk = 0; pk < (refresh ? 0 : m_act); ++pk)
    {
        const int r = rem.pop_lowest();
        int jr = 0, jr2 = -1, tiv = -1;
        T tsgn = T(0);
        Sp<T> fu = zero6<T>(), fu2 = zero6<T>();
        unsigned long long bmask = 0ull;
        static_for<0, R::NB>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if (r == k)
            {
                tiv = Tp::idx_v[R::bjoint(k)];
                tsgn = rev.test(k) ? T(-1) : T(1);
                bmask = R::anc_mask(R::bjoint(k));
            }
        });
        for_contacts<Tp>([&](auto jc, int c) {
            constexpr int j = decltype(jc)::value;
            const int r0 = R::NB + 4 * c;
            if (r >= r0 && r < r0 + 4)
            {
                const int d = r - r0;
                const V3<T> pc = ld_v3<T>(P, L::CONTACT + 12 * c + 9);
                const M3<T> & Rj = w.oMi[j].R;
                // unit force (x, y, z) or unit torque about z at the contact point, world aligned
                const V3<T> col = d == 0 ? V3<T>{Rj.m00, Rj.m01, Rj.m02}
                                : d == 1 ? V3<T>{Rj.m10, Rj.m11, Rj.m12} : V3<T>{Rj.m20, Rj.m21, Rj.m22};
                if (d < 3) fu = {col, cross(pc, col)};
                else fu = {zero3<T>(), col};
                jr = j;
                bmask = R::anc_mask(j);
            }
        });
        static_for<0, R::NXJ>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if (r == R::XJ0 + k)
            {
                tiv = Tp::idx_v[Tp::xjoint[k]];
                tsgn = T(1);
                bmask = R::anc_mask(Tp::xjoint[k]);
            }
        });
        static_for<0, R::NX>([&](auto xc) {
            constexpr int x = decltype(xc)::value;
            constexpr int j = Tp::xframe_joint[x];
            constexpr int r0 = R::XR0 + 6 * x;
            if (r >= r0 && r < r0 + 6)
            {
                // unit force along / unit torque about a world axis at the frame origin (rows of the world-aligned frame
                // Jacobian, frame_constraint.cc:136-146 with rotationLocal = identity)
                const int d = r - r0, ax = d < 3 ? d : d - 3;
                const V3<T> pc = ld_v3<T>(P, L::XFRAME + 12 * x + 9);
                const M3<T> & Rj = w.oMi[j].R;
                if constexpr (Tp::xframe_kind[x] == JM_XKIND_DISTANCE)
                {
                    // + dir on the first frame origin, - dir on the second one (J = dir^T (J_1 - J_2), distance_constraint.cc:113-121)
                    constexpr int j2 = Tp::xframe_joint2[x];
                    V3<T> p1, p2;
                    xframe_points<T, Tp, x>(P, w, p1, p2);
                    const V3<T> dir = rsqrt_(dot(p1 - p2, p1 - p2)) * (p1 - p2);
                    const V3<T> c1 = tmul(Rj, dir), c2 = tmul(w.oMi[j2].R, dir);
                    fu = {c1, cross(pc, c1)};
                    fu2 = {-c2, -cross(ld_v3<T>(P, L::XPAR + 8 * x + 1), c2)};
                    jr = j; jr2 = j2;
                    bmask = R::anc_mask(j) | R::anc_mask(j2);
                }
                else
                {
                    const V3<T> col = ax == 0 ? V3<T>{Rj.m00, Rj.m01, Rj.m02}
                                    : ax == 1 ? V3<T>{Rj.m10, Rj.m11, Rj.m12} : V3<T>{Rj.m20, Rj.m21, Rj.m22};
                    if (d < 3) fu = {col, cross(pc, col)};
                    else fu = {zero3<T>(), col};
                    if constexpr (Tp::xframe_kind[x] != JM_XKIND_FRAME)
                    {
                        // row taken at the contact point: J = J_lin + [rd]x J_ang  <=>  torque e_d x rd next to the force e_d
                        const V3<T> ed = ax == 0 ? V3<T>{T(1), T(0), T(0)} : ax == 1 ? V3<T>{T(0), T(1), T(0)} : V3<T>{T(0), T(0), T(1)};
                        fu.a = fu.a + tmul(Rj, cross(ed, xframe_rd<T, Tp, x>(P, w)));
                    }
                    jr = j;
                    bmask = R::anc_mask(j);
                }
            }
        });
        // (the entry of a DistanceConstraint row is accumulated by the visits of its two joints: cleared first)
        static_for<0, R::NX>([&](auto xc) {
            constexpr int x = decltype(xc)::value;
            if constexpr (Tp::xframe_kind[x] == JM_XKIND_DISTANCE)
                if (xact.test(R::XR0 + 6 * x)) ws(R::WA + act.rank(R::XR0 + 6 * x) * NR + pk) = T(0);
        });
        // J_row . dd of every active row = column pk of the delassus matrix, written as the sweep reaches the joints
        delta_sweeps<T, Tp>(
            P, w, [&](auto ic) { return decltype(ic)::value == tiv ? tsgn : T(0); },
            [&](auto jc) {
                if constexpr (R::NX > 0) return (decltype(jc)::value == jr ? fu : zero6<T>()) + (decltype(jc)::value == jr2 ? fu2 : zero6<T>());
                else return decltype(jc)::value == jr ? fu : zero6<T>();
            },
            [&](auto jc, const T * ddj, const Sp<T> & daj) {
                constexpr int j = decltype(jc)::value;
                constexpr int k = bound_row_of<Tp>(j);
                if constexpr (k >= 0)
                {
                    if (act.test(k)) ws(R::WA + act.rank(k) * NR + pk) = rev.test(k) ? -ddj[0] : ddj[0];
                }
                if constexpr (joint_has_contact<Tp>(j))
                {
#pragma nounroll
                    for (int c = 0; c < Tp::NC; ++c)
                        if (Tp::contact_joint[c] == j && act.test(R::NB + 4 * c))
                        {
                            const V3<T> pc = ld_v3<T>(P, L::CONTACT + 12 * c + 9);
                            const V3<T> lin = w.oMi[j].R * (daj.l + cross(daj.a, pc));
                            const V3<T> ang = w.oMi[j].R * daj.a;
                            const int p0 = act.rank(R::NB + 4 * c);
                            ws(R::WA + p0 * NR + pk) = lin.x;
                            ws(R::WA + (p0 + 1) * NR + pk) = lin.y;
                            ws(R::WA + (p0 + 2) * NR + pk) = lin.z;
                            ws(R::WA + (p0 + 3) * NR + pk) = ang.z;
                        }
                }
                if constexpr (R::xjoint_row(j) >= 0)
                {
                    constexpr int kj = R::xjoint_row(j);
                    if (act.test(R::XJ0 + kj)) ws(R::WA + act.rank(R::XJ0 + kj) * NR + pk) = ddj[0];
                }
                static_for<0, R::NX>([&](auto xc) {
                    constexpr int x = decltype(xc)::value;
                    if constexpr (Tp::xframe_kind[x] == JM_XKIND_DISTANCE)
                    {
                        if constexpr (Tp::xframe_joint[x] == j || Tp::xframe_joint2[x] == j)
                            if (xact.test(R::XR0 + 6 * x))
                            {
                                V3<T> p1, p2;
                                xframe_points<T, Tp, x>(P, w, p1, p2);
                                const V3<T> dir = rsqrt_(dot(p1 - p2, p1 - p2)) * (p1 - p2);
                                T s = T(0);
                                if constexpr (Tp::xframe_joint[x] == j)
                                    s += dot(dir, w.oMi[j].R * (daj.l + cross(daj.a, ld_v3<T>(P, L::XFRAME + 12 * x + 9))));
                                if constexpr (Tp::xframe_joint2[x] == j)
                                    s -= dot(dir, w.oMi[j].R * (daj.l + cross(daj.a, ld_v3<T>(P, L::XPAR + 8 * x + 1))));
                                ws(R::WA + act.rank(R::XR0 + 6 * x) * NR + pk) += s;
                            }
                    }
                    else if constexpr (Tp::xframe_joint[x] == j)
                    {
                        if (xact.test(R::XR0 + 6 * x + R::xfirst(x)))
                        {
                            const V3<T> pc = ld_v3<T>(P, L::XFRAME + 12 * x + 9);
                            V3<T> lin = w.oMi[j].R * (daj.l + cross(daj.a, pc));
                            const V3<T> ang = w.oMi[j].R * daj.a;
                            if constexpr (Tp::xframe_kind[x] != JM_XKIND_FRAME) lin = lin + cross(xframe_rd<T, Tp, x>(P, w), ang);
                            const T six[6] = {lin.x, lin.y, lin.z, ang.x, ang.y, ang.z};
                            static_for<0, 6>([&](auto dc) {
                                constexpr int d = decltype(dc)::value;
                                if constexpr ((Tp::xframe_mask[x] >> d) & 1) ws(R::WA + act.rank(R::XR0 + 6 * x + d) * NR + pk) = six[d];
                            });
                        }
                    }
                });
            },
            bmask, fmask);
        // regularisation (constraint_solvers.cc:376-387)
        const T arr = ws(R::WA + pk * NR + pk);
        ws(R::WA + pk * NR + pk) = arr + fmax_(arr * C.reg, T(1.0e-11));
    }

    // ---- passes: one in normal operation; Engine::start runs INIT_ITERATIONS with its `u` bookkeeping
    const int n_pass = start_passes > 0 ? start_passes : 1;
    T uq[NV];  // RobotState::u seen by this pass minus the motor efforts already inside the free solve
    static_for<0, NV>([&](auto ic) { uq[decltype(ic)::value] = start_passes > 0 ? -w.ueff[decltype(ic)::value] : T(0); });
    Sp<T> fsum[NJ];
    T ddq_free[NV];
    static_for<0, NV>([&](auto ic) { ddq_free[decltype(ic)::value] = w.ddq[decltype(ic)::value]; });
#pragma nounroll
    for (int pass = 0; pass < n_pass; ++pass)
    {
        // free acceleration of this pass (joint accelerations + spatial accelerations, gravity field removed)
        T af[NV];
        Sp<T> sa[NJ];
        sa[0] = zero6<T>();   // (a DistanceConstraint may anchor one of its frames in the world)
        static_for<0, NV>([&](auto ic) { af[decltype(ic)::value] = ddq_free[decltype(ic)::value]; });
        static_for<1, NJ>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            // (only the joints that carry contact points are read below)
            if constexpr (joint_has_contact<Tp>(j) || joint_has_xframe<Tp>(j) || joint_has_xframe2<Tp>(j)) sa[j] = w.agf[j] + actinv_motion(w.oMi[j], Sp<T>{g, gw});
            else sa[j] = zero6<T>();
        });
        if (start_passes > 0)
        {
            delta_sweeps<T, Tp>(P, w, [&](auto ic) { return uq[decltype(ic)::value]; }, [&](auto) { return zero6<T>(); },
                                [&](auto jc, const T * ddj, const Sp<T> & daj) {
                                    constexpr int j = decltype(jc)::value;
                                    constexpr int iv = Tp::idx_v[j];
                                    constexpr int nvj = jt_nv(Tp::jtype[j]);
                                    static_for<0, nvj>([&](auto kc) { af[iv + decltype(kc)::value] += ddj[decltype(kc)::value]; });
                                    sa[j] = sa[j] + daj;
                                });
        }
        if (!refresh)
        {
            // b = -(drift + J a_free)
            static_for<0, R::NB>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                constexpr int jn = R::bjoint(k);
                constexpr int iq = Tp::idx_q[jn], iv = Tp::idx_v[jn];
                if (act.test(k))
                {
                    const T s = (lck.test(k) ? C.kp_lock : C.kp) * (q[iq] - dat(k)) + (lck.test(k) ? C.kd_lock : C.kd) * v[iv] + af[iv];
                    ws(R::WB + act.rank(k)) = rev.test(k) ? s : -s;
                }
            });
            for_contacts<Tp>([&](auto jc, int c) {
                constexpr int j = decltype(jc)::value;
                const int r0 = R::NB + 4 * c;
                if (act.test(r0))
                {
                    const V3<T> pc = ld_v3<T>(P, L::CONTACT + 12 * c + 9);
                    const M3<T> & Rj = w.oMi[j].R;
                    const T depth = w.oMi[j].p.z + dot(V3<T>{Rj.m20, Rj.m21, Rj.m22}, pc);
                    const V3<T> vlin = Rj * (w.vel[j].l + cross(w.vel[j].a, pc));
                    const V3<T> vang = Rj * w.vel[j].a;
                    V3<T> alin = Rj * (sa[j].l + cross(sa[j].a, pc));
                    const V3<T> aang = Rj * sa[j].a;
                    alin = alin + cross(vang, vlin);
                    const int p0 = act.rank(r0);
                    ws(R::WB + p0) = -(alin.x + C.kd * vlin.x);
                    ws(R::WB + p0 + 1) = -(alin.y + C.kd * vlin.y);
                    ws(R::WB + p0 + 2) = -(alin.z + C.kp * depth + C.kd * vlin.z);
                    ws(R::WB + p0 + 3) = -(aang.z + C.kd * vang.z);
                }
            });
            // user FrameConstraint (frame_constraint.cc:148-182): classical frame acceleration (world aligned) + Baumgarte terms
            // on the position error p - p_ref and the orientation error log3(R R_ref^T), gains of the user constraints
            static_for<0, R::NX>([&](auto xc) {
                constexpr int x = decltype(xc)::value;
                constexpr int j = Tp::xframe_joint[x];
                if (xact.test(R::XR0 + 6 * x + R::xfirst(x)))
                {
                    const SE3<T> fr = ld_se3<T>(P, L::XFRAME + 12 * x);
                    const M3<T> & Rj = w.oMi[j].R;
                    const V3<T> vlin = Rj * (w.vel[j].l + cross(w.vel[j].a, fr.p));
                    const V3<T> vang = Rj * w.vel[j].a;
                    V3<T> alin = Rj * (sa[j].l + cross(sa[j].a, fr.p));
                    V3<T> aang = Rj * sa[j].a;
                    alin = alin + cross(vang, vlin);
                    const V3<T> pos = w.oMi[j].p + Rj * fr.p;
                    const V3<T> pref = {dat(R::XREF + 12 * x), dat(R::XREF + 12 * x + 1), dat(R::XREF + 12 * x + 2)};
                    if constexpr (Tp::xframe_kind[x] == JM_XKIND_DISTANCE)
                    {
                        // distance_constraint.cc:96-150: relative classical acceleration along the direction + the
                        // centripetal term of the turning direction + Baumgarte on (distance - reference) and its rate
                        constexpr int j2 = Tp::xframe_joint2[x];
                        const V3<T> pl2 = ld_v3<T>(P, L::XPAR + 8 * x + 1);
                        const M3<T> & R2 = w.oMi[j2].R;
                        const V3<T> p2 = w.oMi[j2].p + R2 * pl2;
                        Sp<T> v2 = zero6<T>(), a2 = zero6<T>();
                        if constexpr (j2 > 0) { v2 = w.vel[j2]; a2 = sa[j2]; }
                        const V3<T> v2lin = R2 * (v2.l + cross(v2.a, pl2)), v2ang = R2 * v2.a;
                        const V3<T> a2lin = R2 * (a2.l + cross(a2.a, pl2)) + cross(v2ang, v2lin);
                        const V3<T> delta = pos - p2, dvel = vlin - v2lin;
                        const T inv = rsqrt_(dot(delta, delta)), dn = dot(delta, delta) * inv;
                        const V3<T> dir = inv * delta;
                        const T dvp = dot(dvel, dir);
                        const T drift = dot(dir, alin - a2lin) + (dot(dvel, dvel) - dvp * dvp) * inv + C.kp_lock * (dn - pref.x) + C.kd_lock * dvp;
                        const int pr = act.rank(R::XR0 + 6 * x);
                        ws(R::WB + pr) = -drift;
                        ws(R::WX + pr) = lam(R::XR0 + 6 * x);
                        return;
                    }
                    else if constexpr (Tp::xframe_kind[x] != JM_XKIND_FRAME)
                    {
                        // sphere_constraint.cc:103-139 / wheel_constraint.cc:96-152: drift at the contact point + Baumgarte on
                        // the height error along the ground normal and on the contact-point velocity
                        const T radius = P[L::XPAR + 8 * x];
                        const V3<T> nrm = ld_v3<T>(P, L::XPAR + 8 * x + 1);
                        const V3<T> rd = xframe_rd<T, Tp, x>(P, w);
                        T dpos = dot(pos - pref, nrm);
                        V3<T> extra = zero3<T>();
                        if constexpr (Tp::xframe_kind[x] == JM_XKIND_WHEEL)
                        {
                            const V3<T> axis = Rj * (fr.R * ld_v3<T>(P, L::XPAR + 8 * x + 4));
                            const V3<T> xx = cross(cross(axis, nrm), axis);
                            const T xinv = rsqrt_(dot(xx, xx));
                            const V3<T> y = xinv * xx;
                            dpos = dot((pos - pref) + radius * (nrm - y), nrm);
                            const V3<T> daxis = cross(vang, axis);
                            const V3<T> z = xinv * (cross(cross(daxis, nrm), axis) + cross(cross(axis, nrm), daxis));
                            const V3<T> dy = z - dot(y, z) * y;
                            extra = cross(radius * dy, vang);
                        }
                        alin = alin + cross(rd, aang) + extra + (C.kp_lock * dpos) * nrm + C.kd_lock * (vlin + cross(rd, vang));
                    }
                    else
                    {
                    const M3<T> Rf = Rj * fr.R;
                    const M3<T> Rref = {dat(R::XREF + 12 * x + 3), dat(R::XREF + 12 * x + 4), dat(R::XREF + 12 * x + 5),
                                        dat(R::XREF + 12 * x + 6), dat(R::XREF + 12 * x + 7), dat(R::XREF + 12 * x + 8),
                                        dat(R::XREF + 12 * x + 9), dat(R::XREF + 12 * x + 10), dat(R::XREF + 12 * x + 11)};
                    const V3<T> dr = log3(Rf * transpose(Rref));
                    alin = alin + C.kp_lock * (pos - pref) + C.kd_lock * vlin;
                    aang = aang + C.kp_lock * dr + C.kd_lock * vang;
                    }
                    const T six[6] = {alin.x, alin.y, alin.z, aang.x, aang.y, aang.z};
                    static_for<0, 6>([&](auto dc) {
                        constexpr int d = decltype(dc)::value;
                        if constexpr ((Tp::xframe_mask[x] >> d) & 1)
                        {
                            const int pr = act.rank(R::XR0 + 6 * x + d);
                            ws(R::WB + pr) = -six[d];
                            ws(R::WX + pr) = lam(R::XR0 + 6 * x + d);
                        }
                    });
                }
            });
            // user JointConstraint rows (joint_constraint.cc:139-163): drift = kp (q - q_ref) + kd v, gains of the user constraints
            static_for<0, R::NXJ>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                constexpr int iq = Tp::idx_q[Tp::xjoint[k]], iv = Tp::idx_v[Tp::xjoint[k]];
                if (act.test(R::XJ0 + k))
                {
                    const int pr = act.rank(R::XJ0 + k);
                    ws(R::WB + pr) = -(C.kp_lock * (q[iq] - dat(R::XJREF + k)) + C.kd_lock * v[iv] + af[iv]);
                    ws(R::WX + pr) = lam(R::XJ0 + k);
                }
            });
            // multipliers: gather the warm start into the packed vector, solve, scatter back
            static_for<0, R::NB>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if (act.test(k)) ws(R::WX + act.rank(k)) = lam(k);
            });
            for_contacts<Tp>([&](auto, int c) {
                const int r0 = R::NB + 4 * c;
                if (act.test(r0))
                {
                    const int p0 = act.rank(r0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) ws(R::WX + p0 + i) = lam(r0 + i);
                }
            });
            bool ok;
            // packed rows of the unbounded constraints: user-registered JointConstraints (on their joint's bound row) and
            // every row of the user FrameConstraints
            unsigned long long lockp = 0ull;
            static_for<0, R::NB>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if (act.test(k) && lck.test(k)) lockp |= 1ull << act.rank(k);
            });
            if constexpr (R::USER)
                for (int pr = mc_act; pr < m_act; ++pr) lockp |= 1ull << pr;
            // `isUnbounded` (constraint_solvers.cc:362-367, 397-412): no inequality among the enabled constraints -> the
            // exact Cholesky solve of `ignoreBounds`, not the sweeps.  (Robots with user constraint frames only: the
            // kernels of the other topologies keep sweeping their lock-only solves, which converge to the same multipliers.)
            const bool unbounded_only = R::USER && __builtin_popcountll(lockp) == m_act;
            if ((start_passes > 0 && pass == 0) || unbounded_only)
            {
                ok = chol_solve_packed<T, Tp>(m_act, ws);
                if (!ok) w.status |= JM_LANE_NAN;
            }
            else
            {
                if (C.park)
                for (int r = 0; r < C.park_rows; ++r) C.park[(size_t)r * B] = C.xl[r * C.xstride];
            // (the register form of the sweeps knows no unbounded rows: solves with any go through the general form)
            if constexpr (JM_CON_PGS_REG && NR <= 32) ok = lockp ? pgs_solve_packed<T, Tp>(C, friction, m_act, nb_act, ws, lockp, mc_act)
                                                               : pgs_solve_regs<T, Tp>(C, friction, m_act, nb_act, ws);
            else ok = pgs_solve_packed<T, Tp>(C, friction, m_act, nb_act, ws, lockp, mc_act);
            if (C.park)
                for (int r = 0; r < C.park_rows; ++r) C.xl[r * C.xstride] = C.park[(size_t)r * B];
                if (ok) w.status &= ~JM_LANE_SOLVER_FAILURE;
                else w.status |= JM_LANE_SOLVER_FAILURE;
            }
            static_for<0, R::NB>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if (act.test(k)) lam(k) = ws(R::WX + act.rank(k));
            });
            for_contacts<Tp>([&](auto, int c) {
                const int r0 = R::NB + 4 * c;
                if (act.test(r0))
                {
                    const int p0 = act.rank(r0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) lam(r0 + i) = ws(R::WX + p0 + i);
                }
            });
            static_for<0, R::NX>([&](auto xc) {
                constexpr int x = decltype(xc)::value;
                if (xact.test(R::XR0 + 6 * x + R::xfirst(x)))
                    static_for<0, 6>([&](auto dc) {
                        constexpr int d = decltype(dc)::value;
                        if constexpr ((Tp::xframe_mask[x] >> d) & 1) lam(R::XR0 + 6 * x + d) = ws(R::WX + act.rank(R::XR0 + 6 * x + d));
                    });
            });
            static_for<0, R::NXJ>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if (act.test(R::XJ0 + k)) lam(R::XJ0 + k) = ws(R::WX + act.rank(R::XJ0 + k));
            });
        }
        // constraint forces of this pass: joint efforts + wrenches on the contact bodies
        T tl[NV];
        static_for<0, NV>([&](auto ic) { tl[decltype(ic)::value] = T(0); });
        static_for<1, NJ>([&](auto jc) { fsum[decltype(jc)::value] = zero6<T>(); });
        static_for<0, R::NB>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int iv = Tp::idx_v[R::bjoint(k)];
            if (act.test(k)) tl[iv] = rev.test(k) ? -lam(k) : lam(k);
        });
        // (user joint constraints: on top of the bound's effort; through the accelerations only, like the frames)
        static_for<0, R::NXJ>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if (act.test(R::XJ0 + k)) tl[Tp::idx_v[Tp::xjoint[k]]] += lam(R::XJ0 + k);
        });
        for_contacts<Tp>([&](auto jc, int c) {
            constexpr int j = decltype(jc)::value;
            const int r0 = R::NB + 4 * c;
            if (act.test(r0))
            {
                const SE3<T> fr = ld_se3<T>(P, L::CONTACT + 12 * c);
                const V3<T> fW = {lam(r0), lam(r0 + 1), lam(r0 + 2)};
                const V3<T> tW = {T(0), T(0), lam(r0 + 3)};
                // convertForceGlobalFrameToJoint (utilities/pinocchio.cc:794-809)
                Sp<T> fl;
                fl.l = tmul(w.oMi[j].R, fW);
                fl.a = tmul(w.oMi[j].R, tW) + cross(fr.p, fl.l);
                fsum[j] = fsum[j] + fl;
                // Robot::contactForces_ in the contact frame (engine.cc:3806-3817)
                w.cf[c] = {tmul(fr.R, fl.l), tmul(fr.R, tmul(w.oMi[j].R, tW))};
            }
        });
        // user FrameConstraints: force at the frame origin + torque, world aligned, on the frame's parent joint; they act
        // through the accelerations only (engine.cc:3770-3857 writes the bounds and the contacts into u / fExternal)
        Sp<T> fux[R::NX > 0 ? NJ : 1];
        if constexpr (R::NX > 0)
        {
            static_for<1, NJ>([&](auto jc) { fux[decltype(jc)::value] = zero6<T>(); });
            static_for<0, R::NX>([&](auto xc) {
                constexpr int x = decltype(xc)::value;
                constexpr int j = Tp::xframe_joint[x];
                if (xact.test(R::XR0 + 6 * x + R::xfirst(x)))
                {
                    T l6[6];
                    static_for<0, 6>([&](auto dc) {
                        constexpr int d = decltype(dc)::value;
                        if constexpr ((Tp::xframe_mask[x] >> d) & 1) l6[d] = lam(R::XR0 + 6 * x + d);
                        else l6[d] = T(0);
                    });
                    const V3<T> pc = ld_v3<T>(P, L::XFRAME + 12 * x + 9);
                    Sp<T> fl;
                    if constexpr (Tp::xframe_kind[x] == JM_XKIND_DISTANCE)
                    {
                        constexpr int j2 = Tp::xframe_joint2[x];
                        V3<T> p1, p2;
                        xframe_points<T, Tp, x>(P, w, p1, p2);
                        const V3<T> fW = (l6[0] * rsqrt_(dot(p1 - p2, p1 - p2))) * (p1 - p2);
                        fl.l = tmul(w.oMi[j].R, fW);
                        fl.a = cross(pc, fl.l);
                        if constexpr (j2 > 0)
                        {
                            const V3<T> f2 = tmul(w.oMi[j2].R, fW);
                            fux[j2] = fux[j2] - Sp<T>{f2, cross(ld_v3<T>(P, L::XPAR + 8 * x + 1), f2)};
                        }
                        if constexpr (j > 0) fux[j] = fux[j] + fl;
                    }
                    else
                    {
                    const V3<T> fW = {l6[0], l6[1], l6[2]};
                    V3<T> tW = {l6[3], l6[4], l6[5]};
                    if constexpr (Tp::xframe_kind[x] != JM_XKIND_FRAME) tW = cross(fW, xframe_rd<T, Tp, x>(P, w));   // sum_d lambda_d (e_d x rd)
                    fl.l = tmul(w.oMi[j].R, fW);
                    fl.a = tmul(w.oMi[j].R, tW) + cross(pc, fl.l);
                    fux[j] = fux[j] + fl;
                    }
                }
            });
        }
        delta_sweeps<T, Tp>(P, w, [&](auto ic) { return tl[decltype(ic)::value]; },
                            [&](auto jc) {
                                if constexpr (R::NX > 0) return fsum[decltype(jc)::value] + fux[decltype(jc)::value];
                                else return fsum[decltype(jc)::value];
                            },
                            [&](auto jc, const T * ddj, const Sp<T> &) {
                                constexpr int j = decltype(jc)::value;
                                constexpr int iv = Tp::idx_v[j];
                                constexpr int nvj = jt_nv(Tp::jtype[j]);
                                static_for<0, nvj>([&](auto kc) {
                                    w.ddq[iv + decltype(kc)::value] = af[iv + decltype(kc)::value] + ddj[decltype(kc)::value];
                                });
                            });
        // Engine::start: the next pass sees u = uInternal (bounds multipliers of this pass, added with a
        // plus sign whatever the direction, engine.cc:3786-3790) + motor efforts (engine.cc:1456-1465)
        static_for<0, NV>([&](auto ic) { uq[decltype(ic)::value] = T(0); });
        // (the multiplier of a user constraint acts through ddq only: engine.cc:3771-3790 restores the bounds' alone)
        static_for<0, R::NB>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int iv = Tp::idx_v[R::bjoint(k)];
            if (act.test(k) && !lck.test(k)) uq[iv] = lam(k);
        });
    }
    // ---- outputs of the last pass: total efforts, external wrenches
    static_for<0, NV>([&](auto ic) { w.ueff[decltype(ic)::value] += uq[decltype(ic)::value]; });
    static_for<1, NJ>([&](auto jc) { w.fext[decltype(jc)::value] = w.fext[decltype(jc)::value] + fsum[decltype(jc)::value]; });
    static_for<0, NV>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (w.ddq[i] != w.ddq[i]) w.status |= JM_LANE_NAN;
    });
}

// What an evaluation's last pass adds to the efforts / external wrenches / contact forces of the kinematic half, from
// the flags and multipliers it stored (engine.cc:3770-3857: bounds into u, contacts into fExternal; user constraints act
// through the accelerations only)
template<class T, class Tp, class CA, class WC>
JM_DEV void constraint_forces_from_multipliers(CPtr<T> P, WC & w, const CA & C, long long lane, long long B)
{
    using L = Layout<Tp>;
    using R = ConRows<Tp>;
    if constexpr (R::NR == 0) return;
    static_for<0, R::NB>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int iv = Tp::idx_v[R::bjoint(k)];
        const int32_t f = C.flags[(size_t)k * B + lane];
        if ((f & 1) && !(f & 4)) w.ueff[iv] += C.data[(size_t)(R::LAM + k) * B + lane];
    });
    for_contacts<Tp>([&](auto jc, int c) {
        constexpr int j = decltype(jc)::value;
        const int r0 = R::NB + 4 * c;
        if (C.flags[(size_t)(R::NB + c) * B + lane] & 1)
        {
            auto lam = [&](int r) { return C.data[(size_t)(R::LAM + r) * B + lane]; };
            const SE3<T> fr = ld_se3<T>(P, L::CONTACT + 12 * c);
            const V3<T> fW = {lam(r0), lam(r0 + 1), lam(r0 + 2)};
            const V3<T> tW = {T(0), T(0), lam(r0 + 3)};
            Sp<T> fl;
            fl.l = tmul(w.oMi[j].R, fW);
            fl.a = tmul(w.oMi[j].R, tW) + cross(fr.p, fl.l);
            w.fext[j] = w.fext[j] + fl;
            w.cf[c] = {tmul(fr.R, fl.l), tmul(fr.R, tmul(w.oMi[j].R, tW))};
        }
    });
}

#ifndef JM_HOST_EMU
// One wave per SIMD (512 registers): capping the registers for 2-3 resident waves was measured 1.5x
// slower (more spill traffic), see DESIGN.md section 4.8.
template<class T, class Tp, bool VAR = false>
__global__ void __launch_bounds__(64) k_constrained(const BatchArgs<T> A, const ConArgs<T> C)
{
    T * const lds = lane_lds<T, Tp>();
    const long long lane = (long long)blockIdx.x * 64 + threadIdx.x;
    if (lane >= A.B) return;
    ConArgs<T> Cl = C;
#if JM_CON_XLDS
    __shared__ T xs[(ConRows<Tp>::NR > 0 ? ConRows<Tp>::NR : 1) * 64];
    Cl.xl = xs + threadIdx.x;
    Cl.xstride = 64;
    Cl.yl = nullptr; Cl.ystride = 0; Cl.yrows = 0;
    Cl.park = nullptr; Cl.park_rows = 0;
#else
    // The packed multipliers of the PGS solve are read m times per row update: they live in LDS whenever
    // that is free, i.e. in the Runge-Kutta stage rows, which only `runge_kutta_4` steps use (the shipped
    // robots integrate with `euler_explicit`); otherwise in the workspace rows (HBM).
    constexpr bool fits = ConRows<Tp>::NR <= stage_rows<Tp>();
    const bool stage_rows_free = A.mode != MODE_STEP || A.solver != JM_SOLVER_RUNGE_KUTTA_4;
    Cl.yl = nullptr; Cl.ystride = 0; Cl.yrows = 0;
    Cl.park = nullptr; Cl.park_rows = 0;
    if (JM_CON_XALIAS && fits)
    {
        if (!stage_rows_free)
        {
            // runge_kutta_4: the solve borrows the stage rows and gives them back (xl is the first of them)
            Cl.park = C.ws + (size_t)ConRows<Tp>::WPARK * A.B + lane;
            Cl.park_rows = stage_rows<Tp>();
        }
        Cl.xl = lds + threadIdx.x;
        Cl.xstride = 64;
        // the residuals take what is left of the stage rows (used when the lane's active rows fit)
        Cl.yl = lds + (size_t)ConRows<Tp>::NR * 64 + threadIdx.x;
        Cl.ystride = 64;
        Cl.yrows = stage_rows<Tp>() - ConRows<Tp>::NR;
    }
    else
    {
        Cl.xl = C.ws + (size_t)ConRows<Tp>::WX * A.B + lane;
        Cl.xstride = (int)A.B;
    }
#endif
    lane_run<T, Tp, 64, typename std::conditional<VAR, WithConA, WithCon>::type>(A, lane, lds + threadIdx.x, Cl);
}
#endif
}  // namespace jm
