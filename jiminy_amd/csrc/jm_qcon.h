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

#ifndef JM_QCON_SKIP
#define JM_QCON_SKIP 0   // profiling only: bit 0 skip the PGS sweeps, 1 the delassus rounds, 2 the closing evaluation
#endif
#ifndef JM_QCON_PGS_INCR
#define JM_QCON_PGS_INCR 1  // register-resident PGS: 1 = residuals maintained incrementally (y -= A[:, i] dx after every update,
                            // no dot products and no quad reductions inside the sweeps), 0 = residual of a row = b - A x at its turn
#endif
#ifndef JM_QCON_PGS_FIXED
#define JM_QCON_PGS_FIXED 1 // waves whose robots all fit the fixed row layout of qcon_pgs_fixed take it
#endif
#ifndef JM_QCON_DELTA
#define JM_QCON_DELTA 0  // 1: evaluations that emit nothing apply the multipliers with one bias-free solve instead of the closing
                         // full evaluation (cheaper arithmetic, but it keeps the free evaluation's data live across the PGS
                         // sweeps: measured slower on ANYmal because the sweeps then spill)
#endif
#ifndef JM_QCON_INIT_ON_CHIP
#define JM_QCON_INIT_ON_CHIP 1  // Engine::start / reset passes whose solve fits the on-chip part of the region take the branch-free
                                // on-chip store and the register-resident sweeps like every other evaluation (0: round-5 behaviour,
                                // the general store with its per-access LDS / HBM branch and the quad-cooperative sweeps: ANYmal,
                                // 65 536 robots, reset of every lane 11.6 ms -- twelve times a step)
#endif
#ifndef JM_QCON_REGS_NIT
#define JM_QCON_REGS_NIT 4  // on-chip solves of up to 4 * JM_QCON_REGS_NIT rows run out of registers (qcon_pgs_regs / _fixed), larger
                            // ones out of LDS (qcon_pgs with the on-chip store): a quarter of a 40-row matrix is 400 registers
#endif
#ifndef JM_QCON_WS_TILED
#define JM_QCON_WS_TILED 1  // workspace rows tiled per wave (qcon_store)
#endif
#ifndef JM_QCON_PGS_WAVES
#define JM_QCON_PGS_WAVES 2  // waves per SIMD of the split form's solve kernel (k_qcon_pgs)
#endif
#ifndef JM_QCON_PRE_WAVES
#define JM_QCON_PRE_WAVES 1   // waves per SIMD of k_quad_con_split<1> / <2> (tuning)
#endif
#ifndef JM_QCON_POST_WAVES
#define JM_QCON_POST_WAVES 1
#endif
#ifndef JM_QCON_PGS_DEPTH
#define JM_QCON_PGS_DEPTH 4  // rows in flight per robot in k_qcon_pgs (ring of row buffers)
#endif
#ifndef JM_QCON_MAXM
#define JM_QCON_MAXM 96  // most active constraint rows solved per robot (rows beyond it are dropped and flagged); 96 = a humanoid
                         // standing flat on two 8-vertex feet during Engine::start (16 contacts x 4 rows + joint bounds)
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
    T kp_lock, kd_lock;   // Baumgarte gains of the user-registered constraints (jm_constraint_options::user_stabilization_freq)
    int iter_max;
    // world.groundProfile as a height map (variation kernels; fields of BatchArgs): contact rows live in the local
    // frame of the ground surface under every contact point (contact_frame, jm_kernels.h); null = flat ground
    const T * ground_h;
    int ground_nx, ground_ny;
    T ground_x0, ground_y0, ground_dx, ground_dy;
    // split stepping (k_quad_con_pre / k_qcon_pgs / k_quad_con_post): stage buffer in HBM, evaluation of this launch
    T * stage;
    int split_e;
    int split_pass;           // Engine::start / reset in the split form: pass of the initialisation (0 first, 1..3 Gauss-Seidel passes, 4 closing)
    int split_r0, split_r1;   // robots [r0, r1) of this launch (chunks of the batch step on streams of their own, jm_lib.cpp)
};

template<class Tp> struct QConRows
{
    using R = ConRows<Tp>;
    static constexpr int MAXM = R::NR < JM_QCON_MAXM ? R::NR : JM_QCON_MAXM;
    // per-robot solver region: x | b | y | 1 / diag(A) | packed lower triangle of A
    static constexpr int VMAX = 4 * MAXM + MAXM * (MAXM + 1) / 2;
    // largest solve that fits entirely in `cap` scalars
    static constexpr int mfit(int cap)
    {
        int m = 0;
        while (m < MAXM && 4 * (m + 1) + (m + 1) * (m + 2) / 2 <= cap) ++m;
        return m;
    }
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
    static constexpr bool ON_CHIP = false;
    static constexpr int NIT = 1;
    T * lds;
    T * hbm;       // already offset by the robot index
    unsigned B;
    int cap;
    JM_DEV T get(int e) const { return e < cap ? lds[e] : hbm[(unsigned)(e - cap) * B]; }
    JM_DEV void put(int e, T x) const { if (e < cap) lds[e] = x; else hbm[(unsigned)(e - cap) * B] = x; }
    // entry (pr, pc), pr >= pc, of the matrix of an m-row solve (packed lower triangle)
    JM_DEV void put_a(int m, int pr, int pc, T x) const { put(4 * m + pr * (pr + 1) / 2 + pc, x); }
    // the same read without a branch: both homes are read at a valid address and the value is selected, so that a
    // batch of reads is issued together instead of one dependent memory round trip per element
    JM_DEV T get_flat(int e) const
    {
        const bool on = e < cap;
        const T l = lds[on ? e : 0];
        const T h = hbm[(unsigned)(on ? 0 : e - cap) * B];
        return on ? l : h;
    }
};
// the same region when the whole solve of the robot fits on chip (the common case: ANYmal with up to 16
// active rows): no per-access branch, so that the loads of one row update are issued together
template<class T, int NIT_> struct QStoreChip
{
    static constexpr bool ON_CHIP = true;
    static constexpr int NIT = NIT_;   // quarter-sum terms per lane: the store holds solves of up to 4 * NIT rows
    T * lds;
    JM_DEV T get(int e) const { return lds[e]; }
    JM_DEV T get_flat(int e) const { return lds[e]; }
    JM_DEV void put(int e, T x) const { lds[e] = x; }
    JM_DEV void put_a(int m, int pr, int pc, T x) const { put(4 * m + pr * (pr + 1) / 2 + pc, x); }
};
// the region of the split form (k_quad_con_split / k_qcon_pgs): one contiguous block of the workspace per robot ([robot][row]: the
// sweeps of the solve stream the matrix of every robot that has not converged yet, and only those), the matrix SQUARE with
// rows padded to a multiple of four entries (row i at 4 m + i ms: a row update reads consecutive, 32-byte aligned entries,
// no triangle index per element), both halves written
template<class T> struct QStoreSq
{
    static constexpr bool ON_CHIP = false;
    static constexpr int NIT = 1;
    T * hbm;
    static JM_DEV int row_stride(int m) { return (m + 3) & ~3; }
    JM_DEV T get(int e) const { return hbm[e]; }
    JM_DEV T get_flat(int e) const { return hbm[e]; }
    JM_DEV void put(int e, T x) const { hbm[e] = x; }
    JM_DEV void put_a(int m, int pr, int pc, T x) const
    {
        const int ms = row_stride(m);
        put(4 * m + pr * ms + pc, x);
        if (pr != pc) put(4 * m + pc * ms + pr, x);
    }
};
JM_DEV int tri_(int i, int c) { return i >= c ? i * (i + 1) / 2 + c : c * (c + 1) / 2 + i; }
template<class T> JM_DEV T bits_as(unsigned long long u) { static_assert(sizeof(T) == 8, "64-bit scalars"); return __builtin_bit_cast(T, u); }
JM_DEV unsigned long long as_bits(double x) { return __builtin_bit_cast(unsigned long long, x); }

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

// scalars of the robot's region of the split form: x | b | y | 1 / diag | square matrix (rows padded to a multiple of 4) | 32
// entries of slack (a row read runs up to 31 entries past the end of the matrix) | header (rows, bounds, rows per contact
// block) | verdict of the solve
template<class Tp> struct QSplitRegion
{
    static constexpr int MAXM = QConRows<Tp>::MAXM;
    // (LOCK: the packed joint rows that are user-registered JointConstraints, as an integer-valued scalar < 2^53)
    static constexpr int HDR = 4 * MAXM + MAXM * MAXM + 32, OK = HDR + 1, LOCK = HDR + 2, ROWS = (LOCK + 2) & ~1;   // (even: 16-byte aligned regions)
};

// which topologies step in the split form (jm_qcon.h, bottom): solves of more than 32 rows
// (JM_QCON_SPLIT_MIN: experimental builds -- codegen.qcon_split_min -- move the threshold, e.g. to put ANYmal's 28-row solves
// through the split form: measured in round 5)
#ifndef JM_QCON_SPLIT_MIN
#define JM_QCON_SPLIT_MIN 32
#endif
#ifndef JM_QCON_PGS_LANE
#define JM_QCON_PGS_LANE 1   // round 6: robots with few contact points step in the split form too, their solve one lane per robot (qcon_pgs_lane)
#endif
// robots whose solves are LARGE (the streamed / operational-space forms; `start` / `reset` through the split kernels as well)
template<class Tp> constexpr bool qcon_split_large() { return Tp::QUAD && QConRows<Tp>::MAXM > JM_QCON_SPLIT_MIN; }
// robots whose whole solve fits the fixed 16-row layout of qcon_pgs_lane (up to five contact points): pre | solve | post as
// well -- the solve kernel holds a robot per lane, which the single kernel cannot (ANYmal: 0.87 -> 0.72-0.84 ms per launch)
template<class Tp> constexpr bool qcon_split_lane() { return JM_QCON_PGS_LANE != 0 && Tp::QUAD && ConRows<Tp>::NC >= 1 && 3 * ConRows<Tp>::NC <= 16; }
template<class Tp> constexpr bool qcon_split() { return qcon_split_large<Tp>() || qcon_split_lane<Tp>(); }
// which kernels know user-registered JointConstraints (bit 2 of a joint row's flag): the variation kernels, and every
// constraint kernel of the topologies that step in the split form (their solves run out of the workspace anyway); the plain
// kernels of robots with register-resident solves (ANYmal) stay free of it -- the host launches the variation kernel for a
// batch that carries locks (jm_batch_set_joint_locks)
template<class Tp, bool GEN> constexpr bool qcon_locks() { return GEN || qcon_split<Tp>(); }
#ifdef JM_TOPO_QCON_SPLIT
static_assert(qcon_split<Topo>() == (JM_TOPO_QCON_SPLIT != 0), "codegen.qcon_split and jm::qcon_split disagree");
#endif

// everything one constrained evaluation shares between its phases (per lane)
template<class T, class Tp> struct QConCtx
{
    using QR = QConRows<Tp>;
    using RowMask = RowMaskN<QR::NWORDS>;
    RowMask act, rev, mine;   // active rows of the robot (solved), reversed bounds, active rows this lane owns
    RowMask lock;             // joint rows that are user-registered JointConstraints (bit 2 of the flag: bilateral, solved first)
    unsigned long long lockp; // the same over the PACKED joint rows of the solve (bit p: packed row p)
    int m, nb;                // packed sizes: rows, of which joint bounds
    int cb;                   // rows per contact block of the solve (4, or 3 when contacts.torsion == 0)
    bool overflow;            // more active rows than JM_QCON_MAXM: the excess was dropped
};

// ---------------------------------------------------------------- switching
// `init`: Engine::start (every constraint enabled first, engine.cc:1266-1308); `readonly`: MODE_REFRESH.
template<class T, class Tp, class X, bool GND = false>
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
    typename QConCtx<T, Tp>::RowMask en, rv, own, lk;
    en.clear(); rv.clear(); own.clear(); lk.clear();
    auto bound = [&](int row, T qj, T lo, T hi, bool writer, bool mine_) {
        const unsigned of = (unsigned)row * B32 + r32, ol = (unsigned)(R::LAM + row) * B32 + r32;
        const int32_t f0 = C.flags[of];
        int32_t f = init ? (1 | (f0 & 4)) : f0;
        if (qcon_locks<Tp, GND>() && (f & 4))
        {
            // user-registered JointConstraint on this joint (Model::addConstraint, model.cc:926-936): always enabled, never
            // reversed, reference configuration = the configuration at Engine::start (JointConstraint::reset) or what the
            // caller stored; the joint's own bound constraint is not switched while the lock holds
            f = 5;
            if (!readonly && init && writer) { C.flags[of] = f; C.data[of] = qj; C.data[ol] = T(0); }
            lk.set(row);
        }
        else if (!readonly)
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
        T d;
        (void)contact_frame<GND>(C, K.R1, K.p1, pc, d);
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
    if constexpr (qcon_locks<Tp, GND>()) quad_or_mask<X>(lk);
    cx.lock = lk;
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
        int last = -1;
        while (keep > 0 && tmp.any()) { last = tmp.pop_lowest(); lim.set(last); --keep; }
        // never keep a part of a contact block: its rows are addressed as r, r + 1, r + 2 (, r + 3) from its first
        // packed row, a truncated block would index past the m kept rows (and past the solver region)
        if (last >= R::NB)
        {
            const int b0 = R::NB + 4 * ((last - R::NB) / 4);
            bool cut = false;
            for (int i = 0; i < 4; ++i) cut |= en.test(b0 + i) && !lim.test(b0 + i);
            if (cut)
                for (int i = 0; i < 4; ++i)
                {
                    typename QConCtx<T, Tp>::RowMask one;
                    one.clear();
                    one.set(b0 + i);
#pragma unroll
                    for (int w = 0; w < QR::NWORDS; ++w) lim.w[w] &= ~one.w[w];
                }
        }
#pragma unroll
        for (int i = 0; i < QR::NWORDS; ++i) { en.w[i] &= lim.w[i]; own.w[i] &= lim.w[i]; }
    }
    cx.act = en; cx.rev = rv; cx.mine = own;
    cx.m = en.count();
    cx.nb = en.rank(R::NB);
    cx.cb = torsion_zero ? 3 : 4;
    static_assert(R::NB <= 64, "packed joint rows of a solve as one 64-bit mask");
    cx.lockp = 0ull;
    if constexpr (qcon_locks<Tp, GND>())
    {
        typename QConCtx<T, Tp>::RowMask tmp = lk;
        while (tmp.any())
        {
            const int row = tmp.pop_lowest();
            if (en.test(row)) cx.lockp |= 1ull << en.rank(row);
        }
    }
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

// trunk tree, bias-free: bias forces `accF[t]` enter at the trunk joints (0 = the root), joint efforts tau_b[t];
// (a lane's own delassus column: one non-zero entry; the closing solve: the quad sums of the four limbs) returns the spatial accelerations of the trunk joints (at[0] = root) and their joint
// accelerations ddb[t].  The TrunkStore broadcasts are executed by the four lanes together (uniform code).
template<class T, class Tp, class X>
JM_DEV void trunk_column(CPtr<T> P, const QKeep<T, Tp> & K, const TrunkStore<T, Tp> & TS, Sp<T> * accF, const T * tau_b,
                         Sp<T> * at, T * ddb)
{
    using QR = QConRows<Tp>;
    constexpr int NT = Tp::QT;
    T ub[NT];
    ub[0] = T(0);
    // leaves -> root: `accF[t]` = bias force entering trunk joint t from the limbs (plus, on the way, from its
    // trunk children and the efforts of the joints themselves)
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
template<class T, class Tp, class X, class VS, bool GND = false>
JM_DEV void qcon_delassus(CPtr<T> P, const LimbTable<T> & LT, const QConArgs<T> & C, int k, const QIdx<Tp> & ix,
                          const QKeep<T, Tp> & K, const TrunkStore<T, Tp> & TS, const QConCtx<T, Tp> & cx, const VS & V)
{
    using Q = QLayout<Tp>;
    using R = ConRows<Tp>;
    using QR = QConRows<Tp>;
    using I = QInfo<Tp>;
    constexpr int N = Tp::QN, NT = Tp::QT;
    const int m = cx.m;
    const int A0 = 4 * m;
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
            // unit force along the local x / y / normal of the ground surface at the contact point, or unit torque about
            // the normal (world x / y / z on a flat ground)
            T dep_;
            const M3<T> Mc = contact_frame<GND>(C, K.R1, K.p1, pc, dep_);
            const V3<T> col = d == 0 ? V3<T>{Mc.m00, Mc.m01, Mc.m02}
                            : d == 1 ? V3<T>{Mc.m10, Mc.m11, Mc.m12} : V3<T>{Mc.m20, Mc.m21, Mc.m22};
            if (d < 3) fu = {col, cross(pc, col)};
            else fu = {zero3<T>(), col};
        }
        // ---- own limb down, trunk tree, own column's accelerations
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
                            V.put_a(cx.m, pr, pcol, val);
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
                        V.put_a(cx.m, pr, pc_, val);
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
                    T dep_;
                    const M3<T> Mc = contact_frame<GND>(C, K.R1, K.p1, pc, dep_);
                    const V3<T> lin = Mc * (atip.l + cross(atip.a, pc));
                    store(r0, lin.x); store(r0 + 1, lin.y); store(r0 + 2, lin.z);
                    if (cx.cb == 4) store(r0 + 3, dot(V3<T>{Mc.m20, Mc.m21, Mc.m22}, atip.a));
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
template<class T, class Tp, class VS, bool GND = false>
JM_DEV void qcon_rhs(CPtr<T> P, const LimbTable<T> & LT, const QConArgs<T> & C, unsigned B32, unsigned r32, int k,
                     const QIdx<Tp> & ix, const T * qb, const T * vb, const T * ql, const T * vl, const T * ddqb, const T * ddq,
                     const QKeep<T, Tp> & K, const QConCtx<T, Tp> & cx, const VS & V)
{
    using Q = QLayout<Tp>;
    using R = ConRows<Tp>;
    using QR = QConRows<Tp>;
    constexpr int N = Tp::QN, NT = Tp::QT;
    const int m = cx.m;
    auto bound = [&](int row, T qj, T vj, T aj) {
        if (!cx.act.test(row)) return;
        const int p = cx.act.rank(row);
        T kp = C.kp, kd = C.kd;
        if constexpr (qcon_locks<Tp, GND>())
            if (cx.lock.test(row)) { kp = C.kp_lock; kd = C.kd_lock; }
        const T s = kp * (qj - C.data[(unsigned)row * B32 + r32]) + kd * vj + aj;
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
        // velocity / drift acceleration in the local frame of the ground surface (LOCAL_WORLD_ALIGNED rotated by
        // rotationLocal^T, frame_constraint.cc:151-174; a rotation commutes with the cross product); the Baumgarte
        // position term is depth * n, i.e. (0, 0, depth) there
        T depth;
        const M3<T> Mc = contact_frame<GND>(C, K.R1, K.p1, pc, depth);
        const V3<T> vlin = Mc * (K.vtip.l + cross(K.vtip.a, pc));
        const V3<T> vang = Mc * K.vtip.a;
        V3<T> alin = Mc * (sa.l + cross(sa.a, pc));
        const V3<T> aang = Mc * sa.a;
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
template<class T, class Tp, class VS>
JM_DEV void qcon_scatter(const LimbTable<T> & LT, const QConArgs<T> & C, unsigned B32, unsigned r32, int k, const QIdx<Tp> & ix,
                         const QConCtx<T, Tp> & cx, const VS & V)
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
// PGSSolver::ProjectedGaussSeidelSolver (constraint_solvers.cc:107-333) over the m packed rows: `nb` joint
// bounds, then blocks of `cb` rows (x, y, z[, torsion]) per active contact.  One row update: every lane of the
// quad sums the columns c = k, k+4, ... of `A.col(i).dot(x)`, one butterfly adds the quarters, the lead lane
// projects and stores the multiplier.  With the on-chip store (VS::ON_CHIP: the whole solve fits in LDS, at most
// 4 * VS::NIT rows) the quarter sums are unrolled and branch-free, so that all the LDS reads of a row update are
// in flight together; the update uses the reciprocal of the diagonal, computed once per solve.
template<class T, class Tp, class X, class VS>
JM_DEV bool qcon_pgs(const QConArgs<T> & C, T friction, int k, const QConCtx<T, Tp> & cx, const VS & V)
{
    const int m = cx.m, nb = cx.nb, cb = cx.cb, A0 = 4 * m;
    const bool lead = (k == 0);
    const T eps = Eps<T>::eps;
    const bool friction_zero = friction < eps;
    const unsigned iter_max = (unsigned)C.iter_max;
    // 1 / diag(A), once per solve
    X::sync();
    for (int i = k; i < m; i += 4) V.put(3 * m + i, T(1) / V.get(A0 + tri_(i, i)));
    X::sync();
    constexpr int NIT = VS::NIT;
    int tric[NIT];   // c (c + 1) / 2 of this lane's columns
    static_for<0, NIT>([&](auto jc) { const int c = k + 4 * decltype(jc)::value; tric[decltype(jc)::value] = c * (c + 1) / 2; });
    auto col_dot = [&](int i) {
        T s = T(0);
        if constexpr (VS::ON_CHIP)
        {
            const int rowbase = A0 + i * (i + 1) / 2;
            T a[NIT], xv[NIT];
            static_for<0, NIT>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int c = k + 4 * j;
                const bool valid = c < m;
                a[j] = V.get(valid ? (c <= i ? rowbase + c : A0 + tric[j] + i) : 0);
                xv[j] = V.get(valid ? c : 0);
            });
            static_for<0, NIT>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                s = (k + 4 * j < m) ? s + a[j] * xv[j] : s;
            });
        }
        else if constexpr (QConRows<Tp>::MAXM <= 32)
            // (robots whose solves nearly always fit the chip: the plain loop keeps the kernel's register budget)
            for (int c = k; c < m; c += 4) s += V.get(A0 + tri_(i, c)) * V.get(c);
        else
        {
            // this lane's quarter of row i, all reads in flight at once (the overflow part of the region sits in HBM:
            // one round trip per row instead of one per element); same summation order as the plain loop
            constexpr int NG = (QConRows<Tp>::MAXM + 3) / 4, NG0 = NG < 16 ? NG : 16;
            auto batch = [&](auto lo_, auto hi_) __attribute__((always_inline)) {
                constexpr int LO = decltype(lo_)::value, HI = decltype(hi_)::value;
                T a[HI - LO > 0 ? HI - LO : 1], xv[HI - LO > 0 ? HI - LO : 1];
                static_for<LO, HI>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const int c = k + 4 * j;
                    const bool valid = c < m;
                    a[j - LO] = V.get_flat(valid ? A0 + tri_(i, c) : 0);
                    xv[j - LO] = V.get_flat(valid ? c : 0);
                });
                static_for<LO, HI>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    s = (k + 4 * j < m) ? s + a[j - LO] * xv[j - LO] : s;
                });
            };
            batch(std::integral_constant<int, 0>{}, std::integral_constant<int, NG0>{});
            // (rows 64 and above exist during Engine::start of a robot standing on many contact points: one scalar test)
            if constexpr (NG > NG0)
                if (X::wave_any(m > 4 * NG0)) batch(std::integral_constant<int, NG0>{}, std::integral_constant<int, NG>{});
        }
        return X::quad_sum(s);
    };
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
            const T y = V.get(m + i) - col_dot(i);
            dmax = fmax_(dmax, cabs_(y - V.get(2 * m + i)));
            ymax = fmax_(ymax, cabs_(y));
            X::sync();
            if (lead) V.put(2 * m + i, y);
            return y;
        };
        // unbounded constraints first (user-registered JointConstraints: constraint_solvers.cc:112-128, no relaxation
        // factor, no projection)
        if (cx.lockp)
            for (int r = 0; r < nb; ++r)
            {
                if (!((cx.lockp >> r) & 1ull)) continue;
                const T y = residual(r);
                const T e = V.get(r) + y * V.get(3 * m + r);
                X::sync();
                if (lead) V.put(r, e);
                X::sync();
            }
        // block 0 of every constraint: joint bounds, then the normal force of every contact
        for (int r = 0; r < m; r += (r < nb ? 1 : cb))
        {
            if (r < nb && ((cx.lockp >> r) & 1ull)) continue;
            const int i0 = r < nb ? r : r + 2;
            const T y = residual(i0);
            const T e = V.get(i0) + (w * y) * V.get(3 * m + i0);
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
                const T e = V.get(i0) + (w * y) * V.get(3 * m + i0);
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
            // 1 / max(a00, a11)
            const T ia = fmin_(V.get(3 * m + r), V.get(3 * m + r + 1));
            T e0 = V.get(r) + (w * y0) * ia;
            T e1 = V.get(r + 1) + (w * y1) * ia;
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
        // stagnation of the residuals (constraint_solvers.cc:263-278); a row that the sweep did not touch kept
        // its residual of zero, so the running maxima equal the reference's sweeps over the whole vector
        const T tol = C.tol_abs + C.tol_rel * ymax + eps;
        if (dmax < tol) return true;
    }
    return false;
}
// The same solver for a robot whose whole solve is on chip (at most MR = 4 * NIT rows: ANYmal 16), run out of
// REGISTERS: every lane loads its quarter of the columns of A (A[i][k + 4j], i < MR, j < NIT), and a full copy
// of x, b, y and 1 / diag(A), once; the sweeps are unrolled on the PACKED row index, so that every operand is
// a register with a compile-time index and the only cross-lane traffic of a row update is the quad butterfly
// of its quarter sums -- no LDS access, no address arithmetic inside the sweeps.  Which packed rows are joint
// bounds / normal forces, torsion rows or the first row of a friction pair is decided once per solve (three
// bit masks); a row that is none of these for any robot of the wave is skipped by a uniform branch.
template<class T, class Tp, class X, int NIT>
JM_DEV bool qcon_pgs_regs(const QConArgs<T> & C, T friction, int k, const QConCtx<T, Tp> & cx, T * Lr)
{
    constexpr int MR = 4 * NIT;
    const int m = cx.m, nb = cx.nb, cb = cx.cb, A0 = 4 * m;
    const bool lead = (k == 0);
    const T eps = Eps<T>::eps;
    const bool friction_zero = friction < eps, torsion_zero = C.torsion < eps;
    const unsigned iter_max = (unsigned)C.iter_max;
    // registers: this lane's quarter of the columns of A; x and 1 / diag(A) replicated in the four lanes; b and the
    // residuals of the rows i = k (mod 4) only (their owner lane computes the residual and broadcasts it)
    T Aq[MR][NIT], x[MR], invd[MR], xq[NIT], bq[NIT], yq[NIT];
    X::sync();
    // 1 / diag(A): every lane divides for its quarter of the rows, the quad shares them through the region
    static_for<0, NIT>([&](auto jc) {
        const int i = k + 4 * decltype(jc)::value;
        if (i < m) Lr[3 * m + i] = T(1) / Lr[A0 + i * (i + 1) / 2 + i];
    });
    X::sync();
    // rows that exist for some robot of the wave (scalar mask): the others cost one scalar test here, none later
    unsigned rows_any = 0u;
    static_for<0, MR>([&](auto ic) { rows_any |= X::wave_any(decltype(ic)::value < m) ? (1u << decltype(ic)::value) : 0u; });
    static_for<0, MR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        x[i] = T(0); invd[i] = T(0);
        static_for<0, NIT>([&](auto jc) { Aq[i][decltype(jc)::value] = T(0); });
        if ((rows_any >> i) & 1u)
        {
            const bool vi = i < m;
            const T xi = Lr[vi ? i : 0], di = Lr[vi ? 3 * m + i : 0];
            x[i] = vi ? xi : T(0);
            invd[i] = vi ? di : T(0);
            const int rowbase = A0 + i * (i + 1) / 2;
            static_for<0, NIT>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int c = k + 4 * j;
                const bool v = vi && c < m;
                const T a = Lr[v ? (c <= i ? rowbase + c : A0 + c * (c + 1) / 2 + i) : 0];
                Aq[i][j] = v ? a : T(0);
            });
        }
    });
    static_for<0, NIT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        xq[j] = k == 0 ? x[4 * j] : (k == 1 ? x[4 * j + 1] : (k == 2 ? x[4 * j + 2] : x[4 * j + 3]));
        const int i = k + 4 * j;
        const T bi = Lr[i < m ? m + i : 0];
        bq[j] = i < m ? bi : T(0);
        yq[j] = T(0);
    });
    // row kinds (bit i = packed row i): block 0 = bounds + normals, block 1 = torsion, block 2 = first row of a cone
    unsigned mask0 = 0u, mask1 = 0u, mask2 = 0u;
    for (int r = 0; r < m; r += (r < nb ? 1 : cb)) mask0 |= 1u << (r < nb ? r : r + 2);
    if (cb == 4)
        for (int r = nb; r < m; r += 4) mask1 |= 1u << (r + 3);
    for (int r = nb; r < m; r += cb) mask2 |= 1u << r;
    // the same three masks for the whole wave (scalars): a row position that no robot of the wave uses in a block
    // costs one scalar bit test per sweep
    unsigned any0 = 0u, any1 = 0u, any2 = 0u;
    static_for<0, MR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        any0 |= X::wave_any((mask0 >> i) & 1u) ? (1u << i) : 0u;
        any1 |= X::wave_any((mask1 >> i) & 1u) ? (1u << i) : 0u;
        any2 |= X::wave_any((mask2 >> i) & 1u) ? (1u << i) : 0u;
    });
    const bool contacts_only = !X::wave_any(!(nb == 0 && cb == 3));
    bool converged = false;
#if JM_QCON_PGS_INCR
    // Residual-maintaining form: y = b - A x is kept up to date for this lane's rows (i = k mod 4): a row update
    // x_i += dx costs the lane NIT multiply-adds on its rows (column i of A = row i, the matrix is symmetric) instead
    // of a dot product + quad butterfly per row.  At its turn the owner lane broadcasts y_i; the projection and the
    // stagnation bookkeeping (|y_i - y_i of the previous sweep|, |y_i|: constraint_solvers.cc:263-278) then run
    // replicated in the four lanes.  Same iterates as the dot-product form up to the rounding of the running sums.
    T yturn[MR];   // residual of every row at its turn in the previous sweep (zero until a row is touched)
    static_for<0, MR>([&](auto ic) { yturn[decltype(ic)::value] = T(0); });
    static_for<0, MR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if ((rows_any >> i) & 1u)
        {
            T s = T(0);
            static_for<0, NIT>([&](auto jc) { s += Aq[i][decltype(jc)::value] * xq[decltype(jc)::value]; });
            const T tot = X::quad_sum(s);
            if (k == (i & 3)) yq[i >> 2] = bq[i >> 2] - tot;
        }
    });
#pragma nounroll
    for (unsigned iter = 0; iter < iter_max && !converged; ++iter)
    {
        T dmax = T(0), ymax = T(0);
        const T ratio = (T(iter_max - 20u) - T(iter)) / T(iter_max - 20u - 30u);
        T w = T(1);
        if (ratio < T(1))
        {
            w = T(0.01);
            if (ratio > T(0)) w += (T(1) - T(0.01)) * (ratio * ratio);
        }
        auto residual = [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            const T yy = X::template bcast<(i & 3)>(yq[i >> 2]);
            dmax = X::max_abs(dmax, yy - yturn[i]);
            ymax = X::max_abs(ymax, yy);
            yturn[i] = yy;
            return yy;
        };
        auto set_x = [&](auto ic, T val) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            const T dx = val - x[i];
            x[i] = val;
            static_for<0, NIT>([&](auto jc) { yq[decltype(jc)::value] -= Aq[i][decltype(jc)::value] * dx; });
        };
        auto cone = [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            if (friction_zero)
            {
                set_x(ic, x[i] * T(0));
                set_x(std::integral_constant<int, i + 1>{}, x[i + 1] * T(0));
            }
            else
            {
                const T y0 = residual(ic);
                const T y1 = residual(std::integral_constant<int, i + 1>{});
                const T ia = fmin_(invd[i], invd[i + 1]);   // 1 / max(a00, a11)
                T e0 = x[i] + (w * y0) * ia;
                T e1 = x[i + 1] + (w * y1) * ia;
                const T thr = friction * x[i + 2];
                const T n2 = e0 * e0 + e1 * e1;
                if (n2 > thr * thr)
                {
                    const T scale = thr / sqrt_(n2);
                    e0 *= scale;
                    e1 *= scale;
                }
                set_x(ic, e0);
                set_x(std::integral_constant<int, i + 1>{}, e1);
            }
        };
        if (contacts_only)
        {
            static_for<0, MR / 3>([&](auto jc) {
                constexpr int i = 3 * decltype(jc)::value + 2;
                if (i < m)
                {
                    const T yy = residual(std::integral_constant<int, i>{});
                    set_x(std::integral_constant<int, i>{}, X::max_(x[i] + (w * yy) * invd[i], T(0)));
                }
            });
            static_for<0, MR / 3>([&](auto jc) {
                constexpr int i = 3 * decltype(jc)::value;
                if (i + 2 < m) cone(std::integral_constant<int, i>{});
            });
        }
        else
        {
        static_for<0, MR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if ((any0 >> i) & 1u)
                if ((mask0 >> i) & 1u)
                {
                    const T yy = residual(ic);
                    set_x(ic, X::max_(x[i] + (w * yy) * invd[i], T(0)));
                }
        });
        static_for<1, MR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if ((any1 >> i) & 1u)
            if ((mask1 >> i) & 1u)
            {
                if (torsion_zero) set_x(ic, x[i] * T(0));
                else
                {
                    const T yy = residual(ic);
                    const T thr = C.torsion * x[i - 1];
                    set_x(ic, clamp_(x[i] + (w * yy) * invd[i], -thr, thr));
                }
            }
        });
        static_for<0, MR - 2>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if ((any2 >> i) & 1u)
                if ((mask2 >> i) & 1u) cone(ic);
        });
        }
        // (every lane followed every row: dmax / ymax are already those of the robot)
        const T tol = C.tol_abs + C.tol_rel * ymax + eps;
        converged = dmax < tol;
    }
#else
#pragma nounroll
    for (unsigned iter = 0; iter < iter_max && !converged; ++iter)
    {
        T dmax = T(0), ymax = T(0);
        const T ratio = (T(iter_max - 20u) - T(iter)) / T(iter_max - 20u - 30u);
        T w = T(1);
        if (ratio < T(1))
        {
            w = T(0.01);
            if (ratio > T(0)) w += (T(1) - T(0.01)) * (ratio * ratio);
        }
        auto residual = [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            T s = T(0);
            static_for<0, NIT>([&](auto jc) { s += Aq[i][decltype(jc)::value] * xq[decltype(jc)::value]; });
            // the owner lane of row i (i mod 4) holds b and the previous residual; its result goes to the quad
            const T mine = bq[i >> 2] - X::quad_sum(s);
            const bool own = k == (i & 3);
            dmax = X::max_abs(dmax, own ? mine - yq[i >> 2] : T(0));
            ymax = X::max_abs(ymax, own ? mine : T(0));
            yq[i >> 2] = own ? mine : yq[i >> 2];
            return X::template bcast<(i & 3)>(mine);
        };
        auto set_x = [&](auto ic, T val) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            x[i] = val;
            if (k == (i & 3)) xq[i >> 2] = val;
        };
        auto cone = [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            if (friction_zero)
            {
                set_x(ic, x[i] * T(0));
                set_x(std::integral_constant<int, i + 1>{}, x[i + 1] * T(0));
            }
            else
            {
                const T y0 = residual(ic);
                const T y1 = residual(std::integral_constant<int, i + 1>{});
                const T ia = fmin_(invd[i], invd[i + 1]);   // 1 / max(a00, a11)
                T e0 = x[i] + (w * y0) * ia;
                T e1 = x[i + 1] + (w * y1) * ia;
                const T thr = friction * x[i + 2];
                const T n2 = e0 * e0 + e1 * e1;
                if (n2 > thr * thr)
                {
                    const T scale = thr / sqrt_(n2);
                    e0 *= scale;
                    e1 *= scale;
                }
                set_x(ic, e0);
                set_x(std::integral_constant<int, i + 1>{}, e1);
            }
        };
        if (contacts_only)
        {
            // every robot of the wave: no joint bound, 3-row contact blocks (robots standing / walking inside their
            // joint ranges, contacts.torsion = 0): the row kinds are compile-time, no per-row mask tests
            static_for<0, MR / 3>([&](auto jc) {
                constexpr int i = 3 * decltype(jc)::value + 2;
                if (i < m)
                {
                    const T yy = residual(std::integral_constant<int, i>{});
                    set_x(std::integral_constant<int, i>{}, X::max_(x[i] + (w * yy) * invd[i], T(0)));
                }
            });
            static_for<0, MR / 3>([&](auto jc) {
                constexpr int i = 3 * decltype(jc)::value;
                if (i + 2 < m) cone(std::integral_constant<int, i>{});
            });
        }
        else
        {
        // block 0
        static_for<0, MR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if ((any0 >> i) & 1u)
                if ((mask0 >> i) & 1u)
                {
                    const T yy = residual(ic);
                    set_x(ic, X::max_(x[i] + (w * yy) * invd[i], T(0)));
                }
        });
        // block 1: torsional friction, bounded by the normal force of the same contact (the row before)
        static_for<1, MR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if ((any1 >> i) & 1u)
            if ((mask1 >> i) & 1u)
            {
                if (torsion_zero) set_x(ic, x[i] * T(0));
                else
                {
                    const T yy = residual(ic);
                    const T thr = C.torsion * x[i - 1];
                    set_x(ic, clamp_(x[i] + (w * yy) * invd[i], -thr, thr));
                }
            }
        });
        // block 2: friction cone (rows i, i + 1; normal force = row i + 2)
        static_for<0, MR - 2>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if ((any2 >> i) & 1u)
                if ((mask2 >> i) & 1u) cone(ic);
        });
        }
        // (dmax / ymax: every lane saw its own rows)
        dmax = X::max_(dmax, X::template perm_<0xB1>(dmax)); dmax = X::max_(dmax, X::template perm_<0x4E>(dmax));
        ymax = X::max_(ymax, X::template perm_<0xB1>(ymax)); ymax = X::max_(ymax, X::template perm_<0x4E>(ymax));
        const T tol = C.tol_abs + C.tol_rel * ymax + eps;
        converged = dmax < tol;
    }
#endif
    // multipliers back to the region (qcon_scatter reads them from there)
    X::sync();
    if (lead)
        static_for<0, MR>([&](auto ic) { if (decltype(ic)::value < m) Lr[decltype(ic)::value] = x[decltype(ic)::value]; });
    X::sync();
    return converged;
}

// The register-resident solver on a FIXED row layout (robots with few contact points: ANYmal's four feet): contact
// point c always sits at positions 3c (tangential x, y) and 3c + 2 (normal), the active joint bounds at the MR - 3 NC
// positions behind them, whatever the packed order of the robot's solver region.  The kind of every position is
// then known at compile time and identical for every robot of the wave: a sweep is straight-line code -- bounds,
// normals, friction cones in the reference's order -- with no per-robot branch and no per-row mask test; a position
// that a robot does not use holds zeros (x, its row and column of A, b, 1 / diag), whose update is the identity.
// Residuals are maintained incrementally like in qcon_pgs_regs.  Taken when every robot of the wave has 3-row
// contact blocks (contacts.torsion = 0), a positive friction coefficient and at most MR - 3 NC active bounds.
template<class T, class Tp, class X, int NIT>
JM_DEV bool qcon_pgs_fixed(const QConArgs<T> & C, T friction, int k, const QConCtx<T, Tp> & cx, T * Lr)
{
    using R = ConRows<Tp>;
    constexpr int MR = 4 * NIT, NC = R::NC, NBF = MR - 3 * NC;
    static_assert(NBF >= 0, "fixed layout needs 3 rows per contact point on chip");
    const int m = cx.m, nb = cx.nb, A0 = 4 * m;
    const bool lead = (k == 0);
    const T eps = Eps<T>::eps;
    const unsigned iter_max = (unsigned)C.iter_max;
    // position -> packed row of this robot's region (-1: unused)
    int pk[MR];
    unsigned used = 0u;
    static_for<0, NC>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        const bool on = cx.act.test(R::NB + 4 * c);
        const int base = cx.act.rank(R::NB + 4 * c);
        pk[3 * c] = on ? base : -1; pk[3 * c + 1] = on ? base + 1 : -1; pk[3 * c + 2] = on ? base + 2 : -1;
        used |= on ? (7u << (3 * c)) : 0u;
    });
    static_for<0, NBF>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        pk[3 * NC + q] = q < nb ? q : -1;
        used |= q < nb ? (1u << (3 * NC + q)) : 0u;
    });
    T Aq[MR][NIT], x[MR], invd[MR], yturn[MR], bq[NIT], yq[NIT], xq[NIT];
    int pkq[NIT];
    static_for<0, NIT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        pkq[j] = k == 0 ? pk[4 * j] : (k == 1 ? pk[4 * j + 1] : (k == 2 ? pk[4 * j + 2] : pk[4 * j + 3]));
    });
    X::sync();
    static_for<0, NIT>([&](auto jc) {
        const int i = k + 4 * decltype(jc)::value;
        if (i < m) Lr[3 * m + i] = rcp_(Lr[A0 + i * (i + 1) / 2 + i]);
    });
    X::sync();
    unsigned used_any = 0u;
    static_for<0, MR>([&](auto ic) { used_any |= X::wave_any((used >> decltype(ic)::value) & 1u) ? (1u << decltype(ic)::value) : 0u; });
    static_for<0, MR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        x[i] = T(0); invd[i] = T(0); yturn[i] = T(0);
        static_for<0, NIT>([&](auto jc) { Aq[i][decltype(jc)::value] = T(0); });
        if ((used_any >> i) & 1u)
        {
            const int pi = pk[i];
            const bool vi = pi >= 0;
            const T xi = Lr[vi ? pi : 0], di = Lr[vi ? 3 * m + pi : 0];
            x[i] = vi ? xi : T(0);
            invd[i] = vi ? di : T(0);
            static_for<0, NIT>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int pc = pkq[j];
                const bool v = vi && pc >= 0;
                const T a = Lr[v ? A0 + tri_(pi, pc) : 0];
                Aq[i][j] = v ? a : T(0);
            });
        }
    });
    static_for<0, NIT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        xq[j] = k == 0 ? x[4 * j] : (k == 1 ? x[4 * j + 1] : (k == 2 ? x[4 * j + 2] : x[4 * j + 3]));
        const T bi = Lr[pkq[j] >= 0 ? m + pkq[j] : 0];
        bq[j] = pkq[j] >= 0 ? bi : T(0);
        yq[j] = T(0);
    });
    // y = b - A x for this lane's positions
    static_for<0, MR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if ((used_any >> i) & 1u)
        {
            T sacc = T(0);
            static_for<0, NIT>([&](auto jc) { sacc += Aq[i][decltype(jc)::value] * xq[decltype(jc)::value]; });
            const T tot = X::quad_sum(sacc);
            if (k == (i & 3)) yq[i >> 2] = bq[i >> 2] - tot;
        }
    });
    bool converged = false;
    const T ratio_den = T(1) / T(iter_max - 20u - 30u);
#pragma nounroll
    for (unsigned iter = 0; iter < iter_max && !converged; ++iter)
    {
        T dmax = T(0), ymax = T(0);
        const T ratio = (T(iter_max - 20u) - T(iter)) * ratio_den;
        T w = T(1);
        if (ratio < T(1))
        {
            w = T(0.01);
            if (ratio > T(0)) w += (T(1) - T(0.01)) * (ratio * ratio);
        }
        auto residual = [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            const T yy = X::template bcast<(i & 3)>(yq[i >> 2]);
            dmax = X::max_abs(dmax, yy - yturn[i]);
            ymax = X::max_abs(ymax, yy);
            yturn[i] = yy;
            return yy;
        };
        auto set_x = [&](auto ic, T val) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            const T dx = val - x[i];
            x[i] = val;
            static_for<0, NIT>([&](auto jc) { yq[decltype(jc)::value] -= Aq[i][decltype(jc)::value] * dx; });
        };
        // block 0: joint bounds, then the normal forces (unilateral)
        static_for<0, NBF>([&](auto qc) {
            constexpr int i = 3 * NC + decltype(qc)::value;
            if ((used_any >> i) & 1u)
            {
                const T yy = residual(std::integral_constant<int, i>{});
                set_x(std::integral_constant<int, i>{}, X::max_(x[i] + (w * yy) * invd[i], T(0)));
            }
        });
        static_for<0, NC>([&](auto cc) {
            constexpr int i = 3 * decltype(cc)::value + 2;
            if ((used_any >> i) & 1u)
            {
                const T yy = residual(std::integral_constant<int, i>{});
                set_x(std::integral_constant<int, i>{}, X::max_(x[i] + (w * yy) * invd[i], T(0)));
            }
        });
        // block 2: friction cones (rows i, i + 1; normal force = row i + 2)
        static_for<0, NC>([&](auto cc) {
            constexpr int i = 3 * decltype(cc)::value;
            if ((used_any >> i) & 1u)
            {
                const T y0 = residual(std::integral_constant<int, i>{});
                const T y1 = residual(std::integral_constant<int, i + 1>{});
                const T ia = fmin_(invd[i], invd[i + 1]);   // 1 / max(a00, a11)
                T e0 = x[i] + (w * y0) * ia;
                T e1 = x[i + 1] + (w * y1) * ia;
                const T thr = friction * x[i + 2];
                const T n2 = e0 * e0 + e1 * e1;
                const bool out = n2 > thr * thr;
                const T scale = out ? thr * rsqrt_(out ? n2 : T(1)) : T(1);
                set_x(std::integral_constant<int, i>{}, e0 * scale);
                set_x(std::integral_constant<int, i + 1>{}, e1 * scale);
            }
        });
        const T tol = C.tol_abs + C.tol_rel * ymax + eps;
        converged = dmax < tol;
    }
    X::sync();
    if (lead)
        static_for<0, MR>([&](auto ic) { if (pk[decltype(ic)::value] >= 0) Lr[pk[decltype(ic)::value]] = x[decltype(ic)::value]; });
    X::sync();
    return converged;
}

// Exact solve A x = b (Engine::start's first pass, `ignoreBounds`: solveJMinvJtv): Cholesky in place in the
// packed triangle -- the caller rebuilds the matrix afterwards (start / reset only).  Left-looking, by the four lanes of the
// quad: the diagonal entry of column j is computed by every lane (no exchange), the rows below it are dealt round-robin; the
// dot products of two rows of the factor read their entries in batches of eight pairs, branch-free (`get_flat`), so that a
// batch costs one round trip to the workspace instead of sixteen.  (Atlas, `reset` of 32 768 robots standing on sixteen
// contact points, 94 rows: 82 -> 68 ms per launch against the serial lead-lane form; the rest is the three start passes
// of the general Gauss-Seidel form at 94 rows.)
// (SQ: the matrix is stored square, rows padded to a multiple of four entries -- the region of the split form, QStoreSq)
template<class T, class X, class VS, bool SQ = false>
JM_DEV bool qcon_chol(int k, int m, const VS & V)
{
    const int A0 = 4 * m, ms = (m + 3) & ~3;
    auto rb = [&](int i) { return SQ ? A0 + i * ms : A0 + i * (i + 1) / 2; };   // first entry of row i of the lower triangle
    bool ok = true;
    X::sync();
    // sum over c in [c0, n) step `step` of F(i, c) F(j, c), rows i, j of the packed factor
    auto rowdot = [&](int i, int j, int c0, int n, int step) {
        const int bi = rb(i), bj = rb(j);
        T s = T(0);
        int c = c0;
        for (; c + 7 * step < n; c += 8 * step)
        {
            T a[8], b[8];
            static_for<0, 8>([&](auto uc) { constexpr int u = decltype(uc)::value; a[u] = V.get_flat(bi + c + u * step); b[u] = V.get_flat(bj + c + u * step); });
            static_for<0, 8>([&](auto uc) { constexpr int u = decltype(uc)::value; s += a[u] * b[u]; });
        }
        for (; c < n; c += step) s += V.get_flat(bi + c) * V.get_flat(bj + c);
        return s;
    };
    for (int j = 0; j < m; ++j)
    {
        const T sj = V.get_flat(rb(j) + j) - rowdot(j, j, 0, j, 1);
        ok &= sj > T(0);
        const T d = sqrt_(sj);
        X::sync();   // (every lane has read the diagonal entry before the lead lane overwrites it)
        if (k == 0) V.put(rb(j) + j, d);
        for (int i = j + 1 + k; i < m; i += 4)
        {
            const T t = V.get_flat(rb(i) + j) - rowdot(i, j, 0, j, 1);
            V.put(rb(i) + j, t / d);
        }
        X::fence();   // the next column reads rows written by the other lanes of the quad
    }
    // forward and backward substitution: the lanes share the sum of a row, every lane holds the result
    for (int i = 0; i < m; ++i)
    {
        const int bi = rb(i);
        T s = T(0);
        for (int c = k; c < i; c += 4) s += V.get_flat(bi + c) * V.get_flat(c);
        s = V.get_flat(m + i) - X::quad_sum(s);
        const T xi = s / V.get_flat(bi + i);
        X::sync();
        if (k == 0) V.put(i, xi);
        X::fence();
    }
    for (int i = m - 1; i >= 0; --i)
    {
        T s = T(0);
        for (int c = i + 1 + k; c < m; c += 4) s += V.get_flat(rb(c) + i) * V.get_flat(c);
        s = V.get_flat(i) - X::quad_sum(s);
        const T xi = s / V.get_flat(rb(i) + i);
        X::sync();
        if (k == 0) V.put(i, xi);
        X::fence();
    }
    X::sync();
    return X::quad_or(ok ? 0 : 1) == 0;
}

// ---------------------------------------------------------------- multipliers applied: a += M^-1 J^T lambda
// One bias-free solve of the whole robot by the quad (evaluations that emit nothing: the closing full
// evaluation with the constraint forces would cost four times as much): every lane pushes the constraint forces
// of ITS limb (bound multipliers as joint efforts, contact multipliers as a wrench on the tip) down to the
// attachment joint, the quad sums enter the trunk tree, and the accelerations come back up.
template<class T, class Tp, class X, bool GND = false>
JM_DEV void qcon_apply_delta(CPtr<T> P, const LimbTable<T> & LT, const QConArgs<T> & C, unsigned B32, unsigned r32, int k,
                             const QIdx<Tp> & ix, const QKeep<T, Tp> & K, const TrunkStore<T, Tp> & TS, const QConCtx<T, Tp> & cx,
                             T * ddqb, T * ddq, int & status)
{
    using Q = QLayout<Tp>;
    using R = ConRows<Tp>;
    using QR = QConRows<Tp>;
    using I = QInfo<Tp>;
    constexpr int N = Tp::QN, NT = Tp::QT;
    auto lam = [&](int row) { return C.data[(unsigned)(R::LAM + row) * B32 + r32]; };
    T tau_l[N], tau_b[NT];
    static_for<0, N>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const int row = sel4(k, QR::limb_row(0, s), QR::limb_row(1, s), QR::limb_row(2, s), QR::limb_row(3, s));
        const bool on = row >= 0 && ix.has[s] && cx.act.test(row);
        const T l = on ? lam(on ? row : 0) : T(0);
        tau_l[s] = (on && cx.rev.test(row)) ? -l : l;
    });
    tau_b[0] = T(0);
    static_for<1, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int row = QR::trunk_row(t);
        tau_b[t] = T(0);
        if constexpr (row >= 0)
        {
            const T l = cx.act.test(row) ? lam(row) : T(0);
            tau_b[t] = cx.rev.test(row) ? -l : l;
        }
    });
    Sp<T> ftip = zero6<T>();
    auto contact = [&](int cl) {
        if (cl >= ix.nc) return;
        const int oc = Q::CONTACT + cl * Q::QC;
        const int r0 = R::NB + 4 * (int)LT(oc + Q::C_IDX);
        if (!cx.act.test(r0)) return;
        const V3<T> pc = K.Rt * LT.v3(oc + 9) + K.ps[N - 1];
        T dep_;
        const M3<T> Mc = contact_frame<GND>(C, K.R1, K.p1, pc, dep_);
        const V3<T> fR = tmul(Mc, V3<T>{lam(r0), lam(r0 + 1), lam(r0 + 2)});
        ftip.l = ftip.l + fR;
        ftip.a = ftip.a + cross(pc, fR);
        if (cx.cb == 4) ftip.a = ftip.a + lam(r0 + 3) * V3<T>{Mc.m20, Mc.m21, Mc.m22};
    };
    if constexpr (Tp::QCL <= 2) static_for<0, Tp::QCL>([&](auto cc) { contact(decltype(cc)::value); });
    else
    {
#pragma nounroll
        for (int cl = 0; cl < Tp::QCL; ++cl) contact(cl);
    }
    T ul[N];
    const Sp<T> fbase = limb_push<T, Tp>(K, tau_l, ftip, ul);
    Sp<T> accF[NT];
    static_for<0, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (I::limb_at(t))
        {
            if constexpr (I::uniform_attach) accF[t] = quad_sum6<T, X>(fbase);
            else accF[t] = quad_sum6<T, X>(mask6(ix.attach == t, fbase));
        }
        else accF[t] = zero6<T>();
    });
    Sp<T> at[NT];
    T ddb[NT];
    trunk_column<T, Tp, X>(P, K, TS, accF, tau_b, at, ddb);
    T dd[N];
    (void)limb_pull<T, Tp>(K, ix, ul, true, pick_attach<T, Tp>(k, at), dd);
    bool bad = false;
    ddqb[0] += at[0].l.x; ddqb[1] += at[0].l.y; ddqb[2] += at[0].l.z;
    ddqb[3] += at[0].a.x; ddqb[4] += at[0].a.y; ddqb[5] += at[0].a.z;
    static_for<1, NT>([&](auto tc) { ddqb[5 + decltype(tc)::value] += ddb[decltype(tc)::value]; });
    static_for<0, N>([&](auto sc) { ddq[decltype(sc)::value] += dd[decltype(sc)::value]; });
    static_for<0, I::NVB>([&](auto ic) { bad |= (ddqb[decltype(ic)::value] != ddqb[decltype(ic)::value]); });
    static_for<0, N>([&](auto sc) { bad |= ix.has[decltype(sc)::value] && (ddq[decltype(sc)::value] != ddq[decltype(sc)::value]); });
    if (bad) status |= JM_LANE_NAN;
}

}  // namespace jm
#include "jm_qtip.h"
namespace jm
{
// ---------------------------------------------------------------- one constrained evaluation
// `start_passes` > 0: Engine::start / reset sequence; < 0: MODE_REFRESH (re-apply the stored multipliers);
// 0: a regular evaluation.  Leaves the constrained acceleration in ddqb / ddq.
// PH = 1 / 2 (split stepping, regular evaluations only): the part before the solve (free acceleration, switching, delassus
// matrix, right-hand side and warm start into the workspace; constraint context into the stage buffer) / after it (multipliers
// back to the lane state, evaluation that applies them).
template<class T, class Tp, class X, class SB, int CAPC, bool GEN, int PH, int INIT>
JM_DEV void quad_eval_con(CPtr<T> P, const LimbTable<T> & LT, const BatchArgs<T> & A, const QConArgs<T> & C, const QStore<T> & V,
                          unsigned r, int k, const QIdx<Tp> & ix, const SB & S_, const T * qb, const T * vb, const T * ql,
                          const T * vl, const T * cmdb, const T * cmdl, bool emit, bool sensors, T * ddqb, T * ddq, int & status,
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
    auto apply = [&]() __attribute__((always_inline)) {
        if (emit) quad_eval<T, Tp, X, true, SB, 2, NoKeep, GEN>(P, LT, A, r32, k, ix, S_, qb, vb, ql, vl, cmdb, cmdl, sensors, ddqb, ddq, status, &ex);
        else quad_eval<T, Tp, X, false, SB, 2, NoKeep, GEN>(P, LT, A, r32, k, ix, S_, qb, vb, ql, vl, cmdb, cmdl, false, ddqb, ddq, status, &ex);
    };
    if constexpr (R::NR == 0)
    {
        apply();
        return;
    }
    const bool refresh = INIT == 1 ? false : start_passes < 0;
    const bool init = INIT < 0 ? start_passes > 0 : INIT == 1;
    const int n_pass = init ? start_passes : 1;
    // (compact batches of the per-stage adaptive stepper: the per-lane friction stays in batch order, BatchArgs::lane_map)
    const T friction = C.friction ? C.friction[A.lane_map ? (unsigned)A.lane_map[r32] : r32] : P[L::OPT + 8];
    QKeep<T, Tp> K;
    TrunkStore<T, Tp> TS;
#ifdef JM_HOST_EMU
    std::memset(&K, 0xFF, sizeof(K)); std::memset(&TS, 0xFF, sizeof(TS));
#endif
    QConCtx<T, Tp> cx;
    cx.m = 0;
    cx.lockp = 0ull;
    cx.lock.clear();
    T uq_l[N], uq_b[NT];   // RobotState::u of the previous start pass minus the motor efforts (bound multipliers, + sign)
    static_for<0, N>([&](auto sc) { uq_l[decltype(sc)::value] = T(0); });
    static_for<0, NT>([&](auto tc) { uq_b[decltype(tc)::value] = T(0); });
    bool any = false;
    if constexpr (PH != 0)
    {
        using SR = QSplitRows<Tp>;
        static_assert(4 * QR::NWORDS + 2 <= SR::NCX, "context rows of the split stage buffer");
        auto mask_io = [&](typename QConCtx<T, Tp>::RowMask & mk, int row0, bool save) {
            static_for<0, QR::NWORDS>([&](auto wc) {
                constexpr int w = decltype(wc)::value;
                if (save) S_.putl(row0 + w, bits_as<T>(mk.w[w]));
                else mk.w[w] = as_bits(S_.getl(row0 + w));
            });
        };
        if constexpr (PH == 1)
        {
            const QStoreSq<T> W{V.hbm};
            if (init && C.split_pass > 0)
            {
                // ---- Engine::start in the split form, Gauss-Seidel pass 1..3 (engine.cc:1399-1467): the multipliers of the
                // previous pass -> lane state; the bound multipliers enter RobotState::u (plus sign whatever the direction,
                // engine.cc:3786-3790) next to the motor efforts; free evaluation with that u; right-hand side and warm start
                mask_io(cx.act, SR::CXL, false);
                mask_io(cx.rev, SR::CXL + QR::NWORDS, false);
                mask_io(cx.mine, SR::CXL + 2 * QR::NWORDS, false);
                mask_io(cx.lock, SR::CXL + 3 * QR::NWORDS, false);
                const int hdr = (int)S_.getl(SR::CXL + 4 * QR::NWORDS);
                cx.m = hdr & 0xff; cx.nb = (hdr >> 8) & 0xff; cx.cb = (hdr >> 16) & 0xff; cx.overflow = (hdr >> 24) & 1;
                any = cx.act.any();
                const bool tipform = any && ((((int)W.get(QSplitRegion<Tp>::HDR)) >> 24) & 1) != 0;
                if (any)
                {
                    const bool solved = W.get(QSplitRegion<Tp>::OK) != T(0);
                    if (C.split_pass == 1) { if (!solved) status |= JM_LANE_NAN; }   // (the exact solve broke down)
                    else if (solved) status &= ~JM_LANE_SOLVER_FAILURE;
                    else status |= JM_LANE_SOLVER_FAILURE;
                    qcon_scatter<T, Tp, QStoreSq<T>>(LT, C, B32, r32, k, ix, cx, W);
                    X::sync();
                    X::fence();
                }
                static_for<0, N>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    const int row = sel4(k, QR::limb_row(0, s), QR::limb_row(1, s), QR::limb_row(2, s), QR::limb_row(3, s));
                    uq_l[s] = (any && row >= 0 && ix.has[s] && cx.act.test(row) && !cx.lock.test(row)) ? C.data[(unsigned)(R::LAM + row) * B32 + r32] : T(0);
                    S_.putl(SR::DDQL + s, uq_l[s]);
                });
                static_for<1, NT>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    constexpr int row = QR::trunk_row(t);
                    if constexpr (row >= 0) uq_b[t] = (any && cx.act.test(row) && !cx.lock.test(row)) ? C.data[(unsigned)(R::LAM + row) * B32 + r32] : T(0);
                    S_.putb(SR::DDQB + t, uq_b[t]);
                });
                static_for<0, N>([&](auto sc) { ex.tau_l[decltype(sc)::value] = uq_l[decltype(sc)::value]; });
                static_for<0, NT>([&](auto tc) { ex.tau_b[decltype(tc)::value] = uq_b[decltype(tc)::value]; });
                ex.motors_on = true;
                quad_eval<T, Tp, X, false, SB, 1, QKeep<T, Tp>, GEN>(P, LT, A, r32, k, ix, S_, qb, vb, ql, vl, cmdb, cmdl, false, ddqb, ddq,
                                                               status, &ex, &K, &TS);
                if (any)
                {
                    // (the factorisation of the streamed form's exact solve overwrote the matrix)
                    if (C.split_pass == 1 && !tipform) qcon_delassus<T, Tp, X, QStoreSq<T>, GEN>(P, LT, C, k, ix, K, TS, cx, W);
                    X::sync();
                    qcon_rhs<T, Tp, QStoreSq<T>, GEN>(P, LT, C, B32, r32, k, ix, qb, vb, ql, vl, ddqb, ddq, K, cx, W);
                }
                return;
            }
            ex.motors_on = !init;   // (first pass of Engine::start: RobotState::u is still zero)
            quad_eval<T, Tp, X, false, SB, 1, QKeep<T, Tp>, GEN>(P, LT, A, r32, k, ix, S_, qb, vb, ql, vl, cmdb, cmdl, false, ddqb, ddq,
                                                           status, &ex, &K, &TS);
            qcon_switch<T, Tp, X, GEN>(P, LT, C, B32, r32, k, ix, qb, ql, K, init, false, cx);
            any = cx.act.any();
            if (cx.overflow) status |= JM_LANE_SOLVER_FAILURE;
            // which form the solve of this WAVE takes: the operational space of the contact-bearing tip bodies (jm_qtip.h)
            // when every robot of the wave has few active joint rows, else the delassus matrix row by row
            bool tipform = false;
            if constexpr (QTip<Tp>::ON && !GEN) tipform = !X::wave_any(cx.nb > QTip<Tp>::NBX);
            if (any)
            {
                if constexpr (QTip<Tp>::ON && !GEN)
                {
                    if (tipform) qtip_build<T, Tp, X, QStoreSq<T>>(P, LT, C, k, ix, K, TS, cx, W);
                    else qcon_delassus<T, Tp, X, QStoreSq<T>, GEN>(P, LT, C, k, ix, K, TS, cx, W);
                }
                else qcon_delassus<T, Tp, X, QStoreSq<T>, GEN>(P, LT, C, k, ix, K, TS, cx, W);
                X::sync();
                qcon_rhs<T, Tp, QStoreSq<T>, GEN>(P, LT, C, B32, r32, k, ix, qb, vb, ql, vl, ddqb, ddq, K, cx, W);
            }
            // header of the solve: rows | joint bounds | rows per contact block | form (0 rows: nothing to solve)
            if (k == 0)
            {
                W.put(QSplitRegion<Tp>::HDR, (T)((any ? (cx.m | (cx.nb << 8) | (cx.cb << 16)) : 0) | (tipform ? (1 << 24) : 0)));
                W.put(QSplitRegion<Tp>::LOCK, (T)cx.lockp);
            }
            mask_io(cx.act, SR::CXL, true);
            mask_io(cx.rev, SR::CXL + QR::NWORDS, true);
            mask_io(cx.mine, SR::CXL + 2 * QR::NWORDS, true);
            mask_io(cx.lock, SR::CXL + 3 * QR::NWORDS, true);
            S_.putl(SR::CXL + 4 * QR::NWORDS, (T)(cx.m | (cx.nb << 8) | (cx.cb << 16) | (cx.overflow ? (1 << 24) : 0)));
            return;
        }
        else
        {
            mask_io(cx.act, SR::CXL, false);
            mask_io(cx.rev, SR::CXL + QR::NWORDS, false);
            mask_io(cx.mine, SR::CXL + 2 * QR::NWORDS, false);
            mask_io(cx.lock, SR::CXL + 3 * QR::NWORDS, false);
            cx.lockp = 0ull;   // (only the solve needs the packed form)
            const int hdr = (int)S_.getl(SR::CXL + 4 * QR::NWORDS);
            cx.m = hdr & 0xff; cx.nb = (hdr >> 8) & 0xff; cx.cb = (hdr >> 16) & 0xff; cx.overflow = (hdr >> 24) & 1;
            any = cx.act.any();
            if (any)
            {
                const QStoreSq<T> W{V.hbm};
                if (W.get(QSplitRegion<Tp>::OK) != T(0)) status &= ~JM_LANE_SOLVER_FAILURE;
                else status |= JM_LANE_SOLVER_FAILURE;
                if (cx.overflow) status |= JM_LANE_SOLVER_FAILURE;
                qcon_scatter<T, Tp, QStoreSq<T>>(LT, C, B32, r32, k, ix, cx, W);
                X::sync();
            }
            if (init)
            {
                // (closing evaluation of Engine::start: u still carries the bound multipliers of the pass before the last)
                static_for<0, N>([&](auto sc) { uq_l[decltype(sc)::value] = S_.getl(SR::DDQL + decltype(sc)::value); });
                static_for<1, NT>([&](auto tc) { uq_b[decltype(tc)::value] = S_.getb(SR::DDQB + decltype(tc)::value); });
            }
        }
    }
    else
    {
#pragma nounroll
    for (int pass = 0; pass < n_pass; ++pass)
    {
        // ---- free acceleration of this pass (+ what the bias-free solves need)
        static_for<0, N>([&](auto sc) { ex.tau_l[decltype(sc)::value] = uq_l[decltype(sc)::value]; });
        static_for<0, NT>([&](auto tc) { ex.tau_b[decltype(tc)::value] = uq_b[decltype(tc)::value]; });
        ex.motors_on = !(init && pass == 0);
        quad_eval<T, Tp, X, false, SB, 1, QKeep<T, Tp>, GEN>(P, LT, A, r32, k, ix, S_, qb, vb, ql, vl, cmdb, cmdl, false, ddqb, ddq,
                                                       status, &ex, &K, &TS);
        if (pass == 0)
        {
            qcon_switch<T, Tp, X, GEN>(P, LT, C, B32, r32, k, ix, qb, ql, K, init, refresh, cx);
            any = cx.act.any();
            if (cx.overflow) status |= JM_LANE_SOLVER_FAILURE;
            if (!any || refresh) break;
        }
        // delassus matrix (first pass), right-hand side and warm start, solve, multipliers back to the lane state
        auto phases = [&](const auto & W) __attribute__((always_inline)) {
            using VS = std::decay_t<decltype(W)>;
            if (pass == 0 && !(JM_QCON_SKIP & 2)) qcon_delassus<T, Tp, X, VS, GEN>(P, LT, C, k, ix, K, TS, cx, W);
            X::sync();
            qcon_rhs<T, Tp, VS, GEN>(P, LT, C, B32, r32, k, ix, qb, vb, ql, vl, ddqb, ddq, K, cx, W);
            X::sync();
            if (init && pass == 0)
            {
                const bool ok = qcon_chol<T, X, VS>(k, cx.m, W);
                if (!ok) status |= JM_LANE_NAN;
                X::sync();
                qcon_scatter<T, Tp, VS>(LT, C, B32, r32, k, ix, cx, W);
                X::sync();
                qcon_delassus<T, Tp, X, VS, GEN>(P, LT, C, k, ix, K, TS, cx, W);   // the factorisation overwrote the matrix
            }
            else
            {
                bool ok = true;
                if constexpr (VS::ON_CHIP && VS::NIT <= JM_QCON_REGS_NIT)
                {
                    // (robots with user-registered JointConstraints: the general form knows the unbounded rows)
                    bool general = false;
                    if constexpr (qcon_locks<Tp, GEN>()) general = X::wave_any(cx.lockp != 0ull);
                    if (general) { if (!(JM_QCON_SKIP & 1)) ok = qcon_pgs<T, Tp, X, VS>(C, friction, k, cx, W); }
                    else if (!(JM_QCON_SKIP & 1))
                    {
                        bool fixed = false;
                        if constexpr (JM_QCON_PGS_FIXED && 3 * ConRows<Tp>::NC <= 4 * VS::NIT)
                        {
                            fixed = !X::wave_any(cx.cb != 3 || cx.nb > 4 * VS::NIT - 3 * ConRows<Tp>::NC || friction < Eps<T>::eps);
                            if (fixed) ok = qcon_pgs_fixed<T, Tp, X, VS::NIT>(C, friction, k, cx, W.lds);
                        }
                        if (!fixed) ok = qcon_pgs_regs<T, Tp, X, VS::NIT>(C, friction, k, cx, W.lds);
                    }
                }
                else if (!(JM_QCON_SKIP & 1)) ok = qcon_pgs<T, Tp, X, VS>(C, friction, k, cx, W);
                if (ok) status &= ~JM_LANE_SOLVER_FAILURE;
                else status |= JM_LANE_SOLVER_FAILURE;
                if (cx.overflow) status |= JM_LANE_SOLVER_FAILURE;
                X::sync();
                qcon_scatter<T, Tp, VS>(LT, C, B32, r32, k, ix, cx, W);
            }
        };
        constexpr int MFIT = QR::mfit(CAPC);
        if constexpr (MFIT >= 4)
        {
            // the whole solve of this robot fits the on-chip part of its region (quad-uniform decision)
            if ((JM_QCON_INIT_ON_CHIP || !init) && cx.m <= MFIT) phases(QStoreChip<T, (MFIT + 3) / 4>{V.lds});
            else phases(V);
        }
        else phases(V);
        X::sync();
        if (pass == n_pass - 1) break;
        // Engine::start: the next pass sees u = uInternal (bound multipliers of this pass, plus sign whatever the
        // direction, engine.cc:3786-3790) + motor efforts (engine.cc:1456-1465)
        static_for<0, N>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            const int row = sel4(k, QR::limb_row(0, s), QR::limb_row(1, s), QR::limb_row(2, s), QR::limb_row(3, s));
            uq_l[s] = (row >= 0 && ix.has[s] && cx.act.test(row) && !cx.lock.test(row)) ? C.data[(unsigned)(R::LAM + row) * B32 + r32] : T(0);
        });
        static_for<1, NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int row = QR::trunk_row(t);
            if constexpr (row >= 0) uq_b[t] = (cx.act.test(row) && !cx.lock.test(row)) ? C.data[(unsigned)(R::LAM + row) * B32 + r32] : T(0);
        });
    }
    }
    // ---- nothing to enforce and nothing to emit: the free acceleration is the answer (engine.cc:3861-3865)
    if (PH == 0 && !emit && !any && !init) return;
    if (JM_QCON_DELTA && !emit && !init && !refresh && !(JM_QCON_SKIP & 4))
    {
        // nothing to emit: the free acceleration plus one bias-free solve with the multipliers
        qcon_apply_delta<T, Tp, X, GEN>(P, LT, C, B32, r32, k, ix, K, TS, cx, ddqb, ddq, status);
        return;
    }
    // ---- apply the multipliers: articulated-body solve with the constraint forces; emits the outputs
    static_for<0, N>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const int row = sel4(k, QR::limb_row(0, s), QR::limb_row(1, s), QR::limb_row(2, s), QR::limb_row(3, s));
        const bool on = any && row >= 0 && ix.has[s] && cx.act.test(row);
        const T lam = on ? C.data[(unsigned)(R::LAM + (on ? row : 0)) * B32 + r32] : T(0);
        ex.tau_l[s] = uq_l[s] + ((on && cx.rev.test(row)) ? -lam : lam);
        ex.uemit_l[s] = (on && cx.lock.test(row)) ? T(0) : lam;   // (a user constraint's multiplier is not restored into RobotState::u)
    });
    static_for<1, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int row = QR::trunk_row(t);
        if constexpr (row >= 0)
        {
            const bool on = any && cx.act.test(row);
            const T lam = on ? C.data[(unsigned)(R::LAM + row) * B32 + r32] : T(0);
            ex.tau_b[t] = uq_b[t] + ((on && cx.rev.test(row)) ? -lam : lam);
            ex.uemit_b[t] = (on && cx.lock.test(row)) ? T(0) : lam;
        }
        else { ex.tau_b[t] = uq_b[t]; ex.uemit_b[t] = T(0); }
    });
    ex.motors_on = true;
    if (!(JM_QCON_SKIP & 4)) apply();
}

// Visit table of one Gauss-Seidel sweep over the m packed rows of a solve (`nb` joint rows, then blocks of `cb` rows per
// contact), in the reference's order (constraint_solvers.cc:107-333): block 0 of every constraint -- joint bounds, then the
// normal force of every contact --, block 1: torsion rows, block 2: friction cones, the two tangential rows one after the
// other (both updated at the second).  Entry = row | kind << 8 with kind 0 clamp at zero, 1 torsion, 2 / 3 first / second
// tangential row, 4 unbounded row (a user-registered JointConstraint: visited FIRST, no relaxation, no projection,
// constraint_solvers.cc:112-128; `lockp` = those among the packed joint rows).  Filled by the four lanes of the quad.
JM_DEV void qcon_visit_table(int k, int m, int nb, int cb, unsigned long long lockp, unsigned short * vt)
{
    const int nc = cb > 0 ? (m - nb) / cb : 0;
    auto visit = [&](int t, int & kind) __attribute__((always_inline)) {
        kind = 0;
        if (t < nb) return t;
        int u = t - nb;
        if (u < nc) return nb + cb * u + 2;
        u -= nc;
        if (cb == 4)
        {
            kind = 1;
            if (u < nc) return nb + 4 * u + 3;
            u -= nc;
        }
        kind = 2 + (u & 1);
        return nb + cb * (u >> 1) + (u & 1);
    };
    if (lockp == 0ull)
        for (int t = k; t < m; t += 4)
        {
            int kind;
            const int row = visit(t, kind);
            vt[t] = (unsigned short)(row | (kind << 8));
        }
    else if (k == 0)
    {
        int t = 0;
        for (int r = 0; r < nb; ++r) if ((lockp >> r) & 1ull) vt[t++] = (unsigned short)(r | (4 << 8));
        for (int r = 0; r < nb; ++r) if (!((lockp >> r) & 1ull)) vt[t++] = (unsigned short)r;
        for (; t < m; ++t)
        {
            int kind;
            const int row = visit(t, kind);
            vt[t] = (unsigned short)(row | (kind << 8));
        }
    }
}

// The solve.  Four lanes per robot as everywhere, 16 robots per wave; the multipliers `x` of the robot on chip (zero beyond
// its m rows), everything else (b, residuals of the previous sweep, 1 / diag, the square matrix) read from the robot's block
// of the workspace with the row: lane k of the quad reads entries 8 j + 2 k, 8 j + 2 k + 1 of the row (16-byte loads, 64
// contiguous bytes per quad), all the loads of a row issued together at immediate offsets from one address.
// The sweep is ONE sequence of m row visits in the reference's order (block 0 of every constraint: joint bounds, then the
// normal force of every contact; block 1: torsion rows {3, 2}; block 2: friction cones {0, 1, 2}, the two tangential rows
// one after the other, both updated at the second), and the rows of the next D - 1 visits -- of the next sweep after the
// last one -- are in flight while the current one is worked on (a ring of D row buffers, the loop unrolled D times so that
// every buffer is a fixed set of registers): the Gauss-Seidel dependency chain does not wait for the workspace.
// NJ = 16-byte loads per lane and row: the kernel is built for solves of up to 8 NJ rows; a wave whose largest solve needs
// another instantiation leaves at once (LO < largest m <= 8 NJ is this one's job).
// PGSSolver::ProjectedGaussSeidelSolver (constraint_solvers.cc:107-333), statement by statement the sweep of `qcon_pgs`.
template<class T> struct alignas(16) QPair { T a, b; };
// `x` / `vt`: the robot's multipliers (8 NJ + 2 scalars, 16-byte aligned) and visit table (8 NJ + 4 words) on chip; `ws` + `g0`:
// the robot's region of the workspace as uniform base + byte offset.  Returns false when the wave belongs to another
// instantiation (nothing touched).
template<class T, class Tp, class X, int NJ, int LO, int D>
JM_DEV bool qcon_pgs_lean(const QConArgs<T> & C, T friction, int k, T * x, unsigned short * vt, char * ws, unsigned g0)
{
    using RG = QSplitRegion<Tp>;
    using T2 = QPair<T>;
    auto G = [&](int e) -> T & { return *(T *)(ws + (g0 + (unsigned)e * (unsigned)sizeof(T))); };
    const int hdr = (int)G(RG::HDR);
    if (X::wave_any(((hdr >> 24) & 1) != 0)) return false;   // (the wave solves in the operational-space form, jm_qtip.h)
    // (bit 25: solved by the one-lane-per-robot form, qcon_pgs_lane -- nothing left to do for this robot)
    const int m = ((hdr >> 25) & 1) ? 0 : (hdr & 0xff), nb = (hdr >> 8) & 0xff, cb = (hdr >> 16) & 0xff, A0 = 4 * m, ms = QStoreSq<T>::row_stride(m);
    {
        const bool big = X::wave_any(m > 8 * NJ), mine = X::wave_any(m > LO);
        if (big || !mine) return false;   // (uniform over the wave)
    }
    if (m == 0) return true;   // (uniform over the quad)
    const bool lead = (k == 0);
    const T eps = Eps<T>::eps;
    const bool friction_zero = friction < eps, torsion_zero = C.torsion < eps;
    const unsigned iter_max = (unsigned)C.iter_max;
    for (int i = k; i < 8 * NJ; i += 4) x[i] = i < m ? G(i) : T(0);
    for (int i = k; i < m; i += 4)
    {
        G(3 * m + i) = T(1) / G(A0 + i * ms + i);
        for (int c = m; c < ms; ++c) G(A0 + i * ms + c) = T(0);   // padding of the row
    }
    for (int i = k; i < 32; i += 4) G(A0 + m * ms + i) = T(0);    // what the reads of the last rows find past the matrix
    // how many groups of 8 * GJ columns exist in this wave (scalar tests around the loads of a row)
    constexpr int GJ = 4, NGR = (NJ + GJ - 1) / GJ;
    bool wide[NGR];
    static_for<0, NGR>([&](auto gc) { wide[decltype(gc)::value] = X::wave_any(m > 8 * GJ * decltype(gc)::value); });
    X::fence();   // (1 / diag, the padding and the visit table were written by one lane of the quad, every lane reads them)
    const unsigned long long lockp = (unsigned long long)G(RG::LOCK);
    qcon_visit_table(k, m, nb, cb, lockp, vt);
    X::sync();
    struct Row { T2 a[NJ]; T b, yp, invd; int i, kind; };
    // row of visit t: this lane's quarter (entries (i, 8 j + 2 k), (i, 8 j + 2 k + 1)), right-hand side, previous residual
    // (used by the lead lane, which alone writes and reads those), 1 / diag; loads only
    auto fetch = [&](int t, Row & R_) __attribute__((always_inline)) {
        const int v = vt[t];
        R_.i = v & 0xff; R_.kind = v >> 8;
        const char * row = ws + (g0 + (unsigned)(A0 + R_.i * ms + 2 * k) * (unsigned)sizeof(T));
        static_for<0, NGR>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (wide[g])
                static_for<GJ * g, (GJ * g + GJ < NJ ? GJ * g + GJ : NJ)>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    R_.a[j] = *(const T2 *)(row + (unsigned)(8 * j) * (unsigned)sizeof(T));
                });
        });
        R_.b = G(m + R_.i); R_.yp = G(2 * m + R_.i); R_.invd = G(3 * m + R_.i);
    };
    // A.col(i).dot(x) from the fetched quarter (x is zero beyond m: the entries read past the end of a row drop out);
    // four partial sums, then the quad butterfly
    auto dot_row = [&](const T2 * a) __attribute__((always_inline)) {
        T s[4] = {T(0), T(0), T(0), T(0)};
        static_for<0, NGR>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (wide[g])
                static_for<GJ * g, (GJ * g + GJ < NJ ? GJ * g + GJ : NJ)>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const T2 xv = *(const T2 *)(x + 8 * j + 2 * k);
                    s[(2 * j) & 3] += a[j].a * xv.a;
                    s[(2 * j + 1) & 3] += a[j].b * xv.b;
                });
        });
        return X::quad_sum((s[0] + s[1]) + (s[2] + s[3]));
    };
    // under-relaxation schedule (constraint_solvers.cc:248-258)
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
    Row ring[D];
    int tp = 0;   // next visit to fetch
    static_for<0, D - 1>([&](auto dc) { fetch(tp, ring[decltype(dc)::value]); tp = tp + 1 < m ? tp + 1 : 0; });
    bool ok = false, done = iter_max == 0;
    unsigned iter = 0;
    int tt = 0;   // visit being worked on
    T w = relaxation(0), dmax = T(0), ymax = T(0);
    T y0 = T(0), id0 = T(0);   // first tangential row of the cone being worked on
    while (!done)
    {
        static_for<0, D>([&](auto dc) {
            constexpr int d = decltype(dc)::value;
            if (done) return;
            fetch(tp, ring[(d + D - 1) % D]);
            tp = tp + 1 < m ? tp + 1 : 0;
            const Row & cur = ring[d];
            const int i = cur.i, kind = cur.kind;
            if ((kind == 1 && torsion_zero) || ((kind == 2 || kind == 3) && friction_zero))
            {
                // (rows the reference zeroes without looking at their residual)
                if (lead) x[i] = x[i] * T(0);
            }
            else
            {
                // residual of the row (kept by the lead lane for the stagnation test of the next sweep), stagnation maxima
                const T y = cur.b - dot_row(cur.a);
                const T yp = X::template bcast<0>(cur.yp);
                dmax = fmax_(dmax, cabs_(y - yp));
                ymax = fmax_(ymax, cabs_(y));
                if (lead) G(2 * m + i) = y;
                // (solves of fewer rows than the ring: the row is already in flight again, with its residual of the sweep before)
                static_for<0, D>([&](auto ec) { if (decltype(ec)::value != d && ring[decltype(ec)::value].i == i) ring[decltype(ec)::value].yp = y; });
                if (kind == 0)
                {
                    const T e = x[i] + (w * y) * cur.invd;
                    if (lead) x[i] = fmax_(e, T(0));  // clamp(e, 0, inf)
                }
                else if (kind == 4)
                {
                    if (lead) x[i] = x[i] + y * cur.invd;
                }
                else if (kind == 1)
                {
                    const T e = x[i] + (w * y) * cur.invd;
                    const T thr = C.torsion * x[i - 1];
                    if (lead) x[i] = clamp_(e, -thr, thr);
                }
                else if (kind == 2) { y0 = y; id0 = cur.invd; }
                else
                {
                    const T ia = fmin_(id0, cur.invd);   // 1 / max(a00, a11)
                    T e0 = x[i - 1] + (w * y0) * ia;
                    T e1 = x[i] + (w * y) * ia;
                    const T thr = friction * x[i + 1];
                    const T n2 = e0 * e0 + e1 * e1;
                    if (n2 > thr * thr)
                    {
                        const T scale = thr / sqrt_(n2);
                        e0 *= scale;
                        e1 *= scale;
                    }
                    if (lead) { x[i - 1] = e0; x[i] = e1; }
                }
            }
            X::sync();   // (the lead lane's multiplier is in place before the next row's dot product reads it)
            if (++tt == m)
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
    for (int i = k; i < m; i += 4) G(i) = x[i];
    if (lead) G(RG::OK) = ok ? T(1) : T(0);
    return true;
}

// ---------------------------------------------------------------- the solve of small systems, ONE LANE per robot (round 6)
// The register-resident Gauss-Seidel of the single kernel (qcon_pgs_fixed) spreads a 16-row solve over the four lanes of the
// robot's quad: the projection logic of a row runs replicated in the four lanes, only the four multiply-adds of the residual
// update are shared -- ~26 wave instructions per robot and sweep, issue-bound at one wave per SIMD (the sweeps were 0.31 of
// the 0.44 ms of an ANYmal evaluation).  In the split form (pre | solve | post) the solve is a kernel of its own, and a
// kernel of its own can choose its own layout: here a lane owns a whole robot -- the packed lower triangle of the matrix
// (136 scalars), x, 1 / diag, the maintained residuals and the residuals of the previous sweep in its registers, a
// residual update of 16 independent multiply-adds -- ~7 wave instructions per robot and sweep, 64 robots per wave, no
// cross-lane traffic at all.  Same fixed row layout as qcon_pgs_fixed (contact block c at positions 3 c .. 3 c + 2, the
// active joint bounds behind), same sweep order, relaxation schedule, projections and stopping rule
// (PGSSolver::ProjectedGaussSeidelSolver, constraint_solvers.cc:107-326).  Taken by the robots whose solve fits the layout
// (<= 16 rows, 3-row contact blocks i.e. contacts.torsion = 0, at most 16 - 3 NC active bounds, positive friction, no
// user-registered joint lock); it marks the region header (bit 25) so that the streamed form leaves the robot alone.
template<class Tp> struct QLanePgs
{
    static constexpr int MR = 16, NC = ConRows<Tp>::NC, NBF = MR - 3 * NC;
    static constexpr bool FITS = qcon_split_lane<Tp>();
    static constexpr int DONE_BIT = 25;
    static constexpr int tri(int r, int c) { return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r; }
};
// NBS = joint-bound positions of this instantiation (0 .. 16 - 3 NC).  A lane holds the whole triangle of a 3 NC + NBS row
// system; the VALU addresses 256 registers per lane, which hold the 3 NC x 3 NC contact block (78 scalars for four feet) next
// to the vectors of the solve; what does not fit sits in accumulation registers behind a v_accvgpr_read per operand half
// (843 VALU instructions per sweep at NBS = 4 against 405 at NBS = 0).  The kernel picks the smallest instantiation that
// serves every robot of the wave (most waves of standing robots have no active bound).
// (Measured and dropped in round 6: the entries of the bound rows and the residuals of the previous sweep in LDS instead,
// entry-major over the lanes -- 843 -> 562 VALU instructions per sweep, but a wave that is alone on its SIMD waits out every
// one of those reads: 187 -> 278 us per solve of 65 536 ANYmal systems.)
template<class T, class Tp, class X, int NBS>
JM_DEV int qcon_pgs_lane(const QConArgs<T> & C, T friction, T * reg)
{
    using RG = QSplitRegion<Tp>;
    using LP = QLanePgs<Tp>;
    constexpr int NC = LP::NC, NBF = NBS, MR = 3 * NC + NBS, NTRI = MR * (MR + 1) / 2;
    static_assert(MR <= LP::MR, "fixed layout of at most 16 rows");
    const T eps = Eps<T>::eps;
    const int hdr = (int)reg[RG::HDR];
    const int m = hdr & 0xff, nb = (hdr >> 8) & 0xff, cb = (hdr >> 16) & 0xff;
    const int nca = cb == 3 ? (m - nb) / 3 : 0;
    const bool mine = m > 0 && m <= MR && ((hdr >> 24) & 1) == 0 && cb == 3 && nb <= NBF && nca <= NC && nb + 3 * nca == m &&
                      reg[RG::LOCK] == T(0) && !(friction < eps);
    if (!X::wave_any(mine)) return 0;
    const int A0 = 4 * m, ms = QStoreSq<T>::row_stride(m);
    const unsigned iter_max = (unsigned)C.iter_max;
    // position -> packed row of this robot (-1: unused)
    int pk[MR];
    unsigned used = 0u;
    static_for<0, NC>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        const bool on = mine && c < nca;
        const int base = nb + 3 * c;
        pk[3 * c] = on ? base : -1; pk[3 * c + 1] = on ? base + 1 : -1; pk[3 * c + 2] = on ? base + 2 : -1;
        used |= on ? (7u << (3 * c)) : 0u;
    });
    static_for<0, NBF>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const bool on = mine && q < nb;
        pk[3 * NC + q] = on ? q : -1;
        used |= on ? (1u << (3 * NC + q)) : 0u;
    });
    unsigned used_any = 0u;
    static_for<0, MR>([&](auto ic) { used_any |= X::wave_any(((used >> decltype(ic)::value) & 1u) != 0u) ? (1u << decltype(ic)::value) : 0u; });
    T At[NTRI], x[MR], invd[MR], y[MR], yturn[MR];
    // matrix entry (r, c) of the packed triangle
    auto A_ = [&](auto rc, auto cc) __attribute__((always_inline)) -> T { return At[LP::tri(decltype(rc)::value, decltype(cc)::value)]; };
    static_for<0, MR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int pi = pk[i];
        const bool vi = pi >= 0;
        const T xi = reg[vi ? pi : 0], bi = reg[vi ? m + pi : 0], aii = reg[vi ? A0 + pi * ms + pi : 0];
        x[i] = vi ? xi : T(0);
        y[i] = vi ? bi : T(0);
        invd[i] = vi ? rcp_(vi ? aii : T(1)) : T(0);
        yturn[i] = T(0);
        static_for<0, i + 1>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const int pc = pk[c];
            const bool v = vi && pc >= 0;
            const T a = reg[v ? A0 + pi * ms + pc : 0];
            At[LP::tri(i, c)] = v ? a : T(0);
        });
    });
    // y = b - A x
    static_for<0, MR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if ((used_any >> i) & 1u)
        {
            T s0 = T(0), s1 = T(0);
            static_for<0, MR>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                if constexpr (c & 1) s1 += A_(ic, cc) * x[c];
                else s0 += A_(ic, cc) * x[c];
            });
            y[i] -= s0 + s1;
        }
    });
    bool converged = !mine;
    int sweeps = 0;
    const T ratio_den = T(1) / T(iter_max - 20u - 30u);
#pragma nounroll
    for (unsigned iter = 0; iter < iter_max && !converged; ++iter)
    {
        ++sweeps;
        T dmax = T(0), ymax = T(0);
        const T ratio = (T(iter_max - 20u) - T(iter)) * ratio_den;
        T w = T(1);
        if (ratio < T(1))
        {
            w = T(0.01);
            if (ratio > T(0)) w += (T(1) - T(0.01)) * (ratio * ratio);
        }
        auto residual = [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            const T yy = y[i];
            dmax = X::max_abs(dmax, yy - yturn[i]);
            ymax = X::max_abs(ymax, yy);
            yturn[i] = yy;
            return yy;
        };
        auto set_x = [&](auto ic, T val) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            const T dx = val - x[i];
            x[i] = val;
            static_for<0, MR>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                // (a bound position no robot of the wave uses: its residual is never read)
                if (r < 3 * NC || ((used_any >> r) & 1u)) y[r] -= A_(rc, ic) * dx;
            });
        };
        // block 0: joint bounds, then the normal forces (unilateral)
        static_for<0, NBF>([&](auto qc) {
            constexpr int i = 3 * NC + decltype(qc)::value;
            if ((used_any >> i) & 1u)
            {
                const T yy = residual(std::integral_constant<int, i>{});
                set_x(std::integral_constant<int, i>{}, X::max_(x[i] + (w * yy) * invd[i], T(0)));
            }
        });
        static_for<0, NC>([&](auto cc) {
            constexpr int i = 3 * decltype(cc)::value + 2;
            if ((used_any >> i) & 1u)
            {
                const T yy = residual(std::integral_constant<int, i>{});
                set_x(std::integral_constant<int, i>{}, X::max_(x[i] + (w * yy) * invd[i], T(0)));
            }
        });
        // block 2: friction cones (rows i, i + 1; normal force = row i + 2)
        static_for<0, NC>([&](auto cc) {
            constexpr int i = 3 * decltype(cc)::value;
            if ((used_any >> i) & 1u)
            {
                const T y0 = residual(std::integral_constant<int, i>{});
                const T y1 = residual(std::integral_constant<int, i + 1>{});
                const T ia = fmin_(invd[i], invd[i + 1]);   // 1 / max(a00, a11)
                const T e0 = x[i] + (w * y0) * ia;
                const T e1 = x[i + 1] + (w * y1) * ia;
                const T thr = friction * x[i + 2];
                const T n2 = e0 * e0 + e1 * e1;
                const bool out = n2 > thr * thr;
                const T scale = out ? thr * rsqrt_(out ? n2 : T(1)) : T(1);
                set_x(std::integral_constant<int, i>{}, e0 * scale);
                set_x(std::integral_constant<int, i + 1>{}, e1 * scale);
            }
        });
        const T tol = C.tol_abs + C.tol_rel * ymax + eps;
        converged = dmax < tol;
    }
    if (mine)
    {
        static_for<0, MR>([&](auto ic) { if (pk[decltype(ic)::value] >= 0) reg[pk[decltype(ic)::value]] = x[decltype(ic)::value]; });
        reg[RG::OK] = converged ? T(1) : T(0);
        reg[RG::HDR] = (T)(hdr | (1 << LP::DONE_BIT));
    }
    return sweeps;
}

// the smallest instantiation that serves every robot of the wave.  `stat` (device counters or null) -- what the host decides
// the form of the next steps on (jm_lib.cpp): [0] robots with a solve that this form cannot take (they fall to the streamed
// form, an order of magnitude slower per robot), [1] sweeps of the waves (the longest solve of each), [2] waves, [3] the longest
// solve of the launch.  The split form pays ~100 us per evaluation for its kernel boundaries and lasts as long as its longest
// solve (~1.5 us per sweep); the single kernel pays ~5 us per AVERAGE sweep: robots dropped on the ground (redundant contacts
// that only the relaxation schedule ends: ~54 sweeps on average, 100 at most) are the split form's, robots standing under
// control (~10 on average) the single kernel's
template<class T, class Tp, class X>
JM_DEV void qcon_pgs_lane_any(const QConArgs<T> & C, T friction, T * reg, int32_t * stat, bool lead)
{
    using RG = QSplitRegion<Tp>;
    constexpr int NBF = QLanePgs<Tp>::NBF;
    const int hdr = (int)reg[RG::HDR];
    const int m = hdr & 0xff, nb = (hdr >> 8) & 0xff, cb = (hdr >> 16) & 0xff;
    const bool tip = ((hdr >> 24) & 1) != 0;
    const bool fits = m > 0 && m <= 16 && cb == 3 && nb <= NBF && !tip && nb + 3 * ((m - nb) / 3) == m && (m - nb) / 3 <= QLanePgs<Tp>::NC &&
                      reg[RG::LOCK] == T(0) && !(friction < Eps<T>::eps);
    if (stat && m > 0 && !tip && !fits)
    {
#ifndef JM_HOST_EMU
        atomicAdd(stat, 1);
#else
        stat[0] += 1;
#endif
    }
    const int nbq = fits ? nb : 0;
    int sweeps;
    if (!X::wave_any(nbq > 0)) sweeps = qcon_pgs_lane<T, Tp, X, 0>(C, friction, reg);
    else if (NBF >= 1 && !X::wave_any(nbq > 1)) sweeps = qcon_pgs_lane<T, Tp, X, (NBF >= 1 ? 1 : NBF)>(C, friction, reg);
    else if (NBF >= 2 && !X::wave_any(nbq > 2)) sweeps = qcon_pgs_lane<T, Tp, X, (NBF >= 2 ? 2 : NBF)>(C, friction, reg);
    else if (NBF >= 3 && !X::wave_any(nbq > 3)) sweeps = qcon_pgs_lane<T, Tp, X, (NBF >= 3 ? 3 : NBF)>(C, friction, reg);
    else sweeps = qcon_pgs_lane<T, Tp, X, NBF>(C, friction, reg);
    if (stat)
    {
        // longest solve of the wave (<= 127 sweeps: seven uniform tests), reported by its lead lane
        int mx = 0;
#pragma unroll
        for (int bit = 6; bit >= 0; --bit)
            if (X::wave_any(sweeps >= (mx | (1 << bit)))) mx |= 1 << bit;
        if (lead && X::wave_any(fits))
        {
#ifndef JM_HOST_EMU
            atomicAdd(stat + 1, mx);
            atomicAdd(stat + 2, 1);
            atomicMax(stat + 3, mx);
#else
            stat[1] += mx; stat[2] += 1; stat[3] = stat[3] > mx ? stat[3] : mx;
#endif
        }
    }
}

#ifndef JM_HOST_EMU
// ---------------------------------------------------------------- kernel configuration (host-visible)
// waves per block and on-chip scalars per LANE of the per-robot solver region (a robot owns 4 lanes' worth):
// what is left of the 160 KiB of LDS next to the limb table and the stage buffer at 4 resident waves per CU
// (the kernel needs the whole register file: one wave per SIMD), capped by what the largest solve can use.
// LDS plan of the constraint kernels.  The per-robot solver region wants to be on chip at least with its four vectors
// (x | b | y | 1 / diag: they are read and written row by row inside the Gauss-Seidel dependency chain; the matrix is only
// read).  `WR` = waves resident per CU: 4 (one per SIMD) when the stage buffer and the limb table leave >= 3 MAXM scalars
// per robot (ANYmal: 212, whole 16-row solves on chip), else 2 or 1 -- for Atlas the stage rows of four waves take 115 kB
// and leave 12 scalars per robot, which puts every row update of the solver behind HBM round trips; two resident waves
// leave 236 (Atlas, B = 32 768: 33.4 -> 28.4 ms per launch together with the batched row reads of `qcon_pgs`).
template<class T, class Tp> constexpr long qcon_free_lds(int wr, int wb)
{
    const long per_wave = (long)sizeof(T) * (QRows<Tp>::NL * 64 + QRows<Tp>::NB * 16);
    const long table = (long)sizeof(T) * QLayout<Tp>::TABLE;
    return 160L * 1024 - 2048 - (long)(wr / wb) * table - (long)wr * per_wave;   // 2 KiB of slack (alignment, odd strides)
}
template<class T, class Tp> constexpr int qcon_resident_waves()
{
#ifdef JM_QCON_RESIDENT_WAVES
    return JM_QCON_RESIDENT_WAVES;   // tuning override
#endif
    for (int wr = 4; wr > 1; wr /= 2)
    {
        const int wb = wr < quad_block_waves<T, Tp>() ? wr : quad_block_waves<T, Tp>();
        const long per_robot = qcon_free_lds<T, Tp>(wr, wb) / ((long)wr * 16 * (long)sizeof(T));
        if (per_robot >= 3L * (QConRows<Tp>::MAXM < 64 ? QConRows<Tp>::MAXM : 64)) return wr;
    }
    return 1;
}
template<class T, class Tp> constexpr int qcon_block_waves()
{
    return qcon_resident_waves<T, Tp>() < quad_block_waves<T, Tp>() ? qcon_resident_waves<T, Tp>() : quad_block_waves<T, Tp>();
}
template<class T, class Tp> constexpr int qcon_lane_scalars()
{
    constexpr long W = qcon_resident_waves<T, Tp>();
    constexpr long left = qcon_free_lds<T, Tp>((int)W, qcon_block_waves<T, Tp>());
    constexpr long per_lane = left > 0 ? left / (W * 64 * (long)sizeof(T)) : 0;
    constexpr long want = (QConRows<Tp>::VMAX + 3) / 4;
    return (int)(per_lane < want ? per_lane : want);
}
// on-chip scalars per robot / HBM workspace rows per robot
template<class T, class Tp> constexpr int qcon_capacity() { return 4 * qcon_lane_scalars<T, Tp>(); }
template<class T, class Tp> constexpr int qcon_ws_rows() { return QConRows<Tp>::ws_rows(qcon_capacity<T, Tp>()); }

// The robot's solver region.  Workspace rows of the 16 robots of a wave are one contiguous tile ([B / 16][rows][16]:
// a row update reads `m` consecutive 128-byte lines) whenever the batch is a multiple of 16, robot-minor rows
// ([rows][B]: every entry of a robot's matrix 8 B bytes apart, i.e. on its own page) otherwise.
template<class T, class Tp> JM_DEV QStore<T> qcon_store(T * lds, T * ws, long long r, unsigned B)
{
    constexpr int CAP = qcon_capacity<T, Tp>();
#if JM_QCON_WS_TILED
    if ((B & 15u) == 0)
        return {lds, ws + (size_t)(r >> 4) * (size_t)(qcon_ws_rows<T, Tp>() * 16) + (size_t)(r & 15), 16u, CAP};
#endif
    return {lds, ws + r, B, CAP};
}

// INIT = 0: every mode but `start` / `reset`; INIT = 1: those two (jm_lib.cpp picks the kernel by mode)
template<class T, class Tp, int INIT>
__global__ void __launch_bounds__((64 * qcon_block_waves<T, Tp>())) __attribute__((amdgpu_waves_per_eu(1)))
k_quad_con(const BatchArgs<T> A, const QConArgs<T> C)
{
    using Q = QLayout<Tp>;
    constexpr int NTH = 64 * qcon_block_waves<T, Tp>();
    constexpr int CAP = qcon_capacity<T, Tp>();
    constexpr int RSTRIDE = (CAP % 2 == 0) ? CAP + 1 : CAP;   // odd: the 16 robots of a wave start in different banks
    __shared__ T table[Q::TABLE];
    __shared__ T stage_l[QRows<Tp>::NL * NTH];
    __shared__ T stage_b[QRows<Tp>::NB * (NTH / 4)];
    __shared__ T con[(CAP > 0 ? RSTRIDE : 1) * (NTH / 4)];
#pragma nounroll
    for (int i = threadIdx.x; i < Q::TABLE; i += NTH) table[i] = A.P[Q::OFFSET + i];
    const long long r = (long long)blockIdx.x * (NTH / 4) + (threadIdx.x >> 2);
    const int k = threadIdx.x & 3;
    if (r >= A.B) return;
    const StageBuf<T, NTH, NTH / 4> S{stage_l + threadIdx.x, stage_b + (threadIdx.x >> 2), k == 0};
    const QStore<T> V = qcon_store<T, Tp>(con + (threadIdx.x >> 2) * RSTRIDE, C.ws, r, (unsigned)A.B);
    quad_lane_run<T, Tp, DppQuad, NTH, NTH / 4, true, CAP, false, 0, INIT>(A, r, k, table, S, &C, &V);
}
// ---------------------------------------------------------------- split stepping: pre | solve | post
// Robots whose solves do not fit the chip (Atlas: 27-52 rows standing, up to JM_QCON_MAXM) step through THREE launches per
// evaluation instead of one kernel that holds the whole evaluation in 512 registers while it waits on workspace rows:
//   k_quad_con_pre   RK stage update, free acceleration, switching, delassus matrix, right-hand side -> workspace
//   k_qcon_pgs       the projected Gauss-Seidel sweeps alone: ~100 registers, x on chip, 8-12 waves per CU, so that the
//                    round trips of the row reads of one robot overlap with the sweeps of the others
//   k_quad_con_post  multipliers -> lane state, evaluation that applies them (and emits the outputs of the launch)
// The RK stage buffer, the state of the evaluation in flight and the constraint context live in HBM between the launches
// (QSplitRows, one tile per wave).  Same functions, same arithmetic, same order as the single kernel; `start` / `reset` /
// `refresh` / `dynamics` launches and the variation kernels keep the single kernel.
// workspace rows ([rows][B] scalars) of the split form: solver region + header + verdict, then the stage tiles
template<class T, class Tp> constexpr int qcon_split_region_rows() { return QSplitRegion<Tp>::ROWS; }
template<class T, class Tp> constexpr int qcon_split_ws_rows()
{
    return qcon_split_region_rows<T, Tp>() + (QSplitRows<Tp>::TILE + 15) / 16;
}
template<class T, class Tp> JM_DEV QStore<T> qcon_split_store(T * ws, long long r)
{
    return {nullptr, ws + (size_t)r * (size_t)qcon_split_region_rows<T, Tp>(), 1u, 0};
}

// INIT = 0: the parts of a step launch; INIT = 1: the passes of `start` / `reset` (jm_lib.cpp picks by mode)
template<class T, class Tp, int PH, int INIT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PH == 1 ? JM_QCON_PRE_WAVES : JM_QCON_POST_WAVES)))
k_quad_con_split(const BatchArgs<T> A, const QConArgs<T> C)
{
    using Q = QLayout<Tp>;
    using SR = QSplitRows<Tp>;
    __shared__ T table[Q::TABLE];
#pragma nounroll
    for (int i = threadIdx.x; i < Q::TABLE; i += 256) table[i] = A.P[Q::OFFSET + i];
    const long long r = (long long)C.split_r0 + (long long)blockIdx.x * 64 + (threadIdx.x >> 2);   // (r0: a multiple of 64)
    const int k = threadIdx.x & 3;
    if (r >= C.split_r1) return;
    T * tile = C.stage + (size_t)(r >> 4) * (size_t)SR::TILE;
    const StageBuf<T, 64, 16> S{tile + (threadIdx.x & 63), tile + SR::NL * 64 + ((threadIdx.x >> 2) & 15), k == 0};
    const QStore<T> V = qcon_split_store<T, Tp>(C.ws, r);
    quad_lane_run<T, Tp, DppQuad, 64, 16, true, 0, false, PH, INIT>(A, r, k, table, S, &C, &V);
}

template<class T, class Tp, int NJ, int LO, int D>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(JM_QCON_PGS_WAVES)))
k_qcon_pgs(const QConArgs<T> C, const T * P, unsigned)
{
    using L = Layout<Tp>;
    using RG = QSplitRegion<Tp>;
    constexpr int XS = 8 * NJ + 2;               // (stride in scalars: even, so that the pairs stay 16-byte aligned)
    constexpr int VS_ = 8 * NJ + 4;              // visit table of a robot: one 16-bit word per row visit (+ 4: quads in different banks)
    __shared__ QPair<T> xs2[XS / 2 * 64];
    __shared__ unsigned short vis[VS_ * 64];
    const unsigned r = (unsigned)C.split_r0 + blockIdx.x * 64u + (threadIdx.x >> 2);
    if (r >= (unsigned)C.split_r1) return;
    // uniform base (the 64 robots of the block) + unsigned 32-bit BYTE offset per lane
    char * const ws = (char *)(C.ws + ((size_t)C.split_r0 + (size_t)blockIdx.x * 64) * (size_t)RG::ROWS);
    const unsigned g0 = (threadIdx.x >> 2) * (unsigned)(RG::ROWS * sizeof(T));
    qcon_pgs_lean<T, Tp, DppQuad, NJ, LO, D>(C, C.friction ? C.friction[r] : P[L::OPT + 8], (int)(threadIdx.x & 3),
                                             (T *)xs2 + (threadIdx.x >> 2) * XS, vis + (threadIdx.x >> 2) * VS_, ws, g0);
}

// the solve of the robots whose system fits the fixed 16-row layout, one lane per robot (qcon_pgs_lane): launched BEFORE
// k_qcon_pgs, which then finds those robots marked done.  One wave per block: 64 robots, the whole register file.
template<class T, class Tp>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1)))
k_qcon_pgs_lane(const QConArgs<T> C, const T * P, int32_t * stat)
{
    using L = Layout<Tp>;
    using RG = QSplitRegion<Tp>;
    const unsigned r = (unsigned)C.split_r0 + blockIdx.x * 64u + threadIdx.x;
    if (r >= (unsigned)C.split_r1) return;
    if constexpr (QLanePgs<Tp>::FITS)
        qcon_pgs_lane_any<T, Tp, DppQuad>(C, C.friction ? C.friction[r] : P[L::OPT + 8], C.ws + (size_t)r * (size_t)RG::ROWS,
                                          stat, threadIdx.x == 0);
}

// Engine::start / reset in the split form: the exact solve of the first pass (`ignoreBounds`), one quad per robot --
// Woodbury in the operational space for the waves of that form (jm_qtip.h), an in-place Cholesky factorisation of the
// square matrix for the others (`k_quad_con_split<1>` rebuilds the matrix in the next pass, like the single kernel)
template<class T, class Tp>
__global__ void __launch_bounds__(256)
k_qcon_exact(const QConArgs<T> C)
{
    // (streamed form: one quad per robot; the robots of the operational-space form are k_qtip_exact's)
    using RG = QSplitRegion<Tp>;
    const unsigned r = (unsigned)C.split_r0 + blockIdx.x * 64u + (threadIdx.x >> 2);
    if (r >= (unsigned)C.split_r1) return;
    T * const reg = C.ws + (size_t)r * (size_t)RG::ROWS;
    const int k = (int)(threadIdx.x & 3);
    const int hdr = (int)reg[RG::HDR];
    const int m = hdr & 0xff;
    if (m == 0 || ((hdr >> 24) & 1) != 0) return;   // (uniform over the quad)
    const QStoreSq<T> W{reg};
    const bool ok = qcon_chol<T, DppQuad, QStoreSq<T>, true>(k, m, W);
    if (k == 0) reg[RG::OK] = ok ? T(1) : T(0);
}
// ... and of the operational-space form: ONE LANE per robot (the 20-dimensional system is the same small serial computation
// for every lane: 64 robots per wave instead of 16)
template<class T, class Tp>
__global__ void __launch_bounds__(256)
k_qtip_exact(const QConArgs<T> C)
{
    using RG = QSplitRegion<Tp>;
    const unsigned r = (unsigned)C.split_r0 + blockIdx.x * 256u + threadIdx.x;
    if (r >= (unsigned)C.split_r1) return;
    if constexpr (QTip<Tp>::ON)
        qtip_exact<T, Tp, DppQuad, 1>(0, (char *)C.ws, (size_t)r * (size_t)RG::ROWS * sizeof(T));
}

// the solve in the operational-space form (jm_qtip.h): multipliers, z and the visit table of the block's 64 robots on chip
template<class T, class Tp>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(JM_QTIP_WAVES)))
k_qtip_pgs(const QConArgs<T> C, const T * P, unsigned)
{
    using L = Layout<Tp>;
    using RG = QSplitRegion<Tp>;
    using TP = QTip<Tp>;
    constexpr int XS = QConRows<Tp>::MAXM + 3;    // (odd strides: the 16 robots of a wave start in different banks)
    constexpr int ZS = TP::ZPAD + 1 - (TP::ZPAD & 1) + 2;
    constexpr int VS_ = QConRows<Tp>::MAXM + 4;
    __shared__ T xs[XS * 64];
    __shared__ T zs[ZS * 64];
    __shared__ T yps[XS * 64];                   // residuals of the previous sweep (same stride as the multipliers)
    __shared__ unsigned short vis[VS_ * 64];
    const unsigned r = (unsigned)C.split_r0 + blockIdx.x * 64u + (threadIdx.x >> 2);
    if (r >= (unsigned)C.split_r1) return;
    char * const ws = (char *)(C.ws + ((size_t)C.split_r0 + (size_t)blockIdx.x * 64) * (size_t)RG::ROWS);
    const unsigned g0 = (threadIdx.x >> 2) * (unsigned)(RG::ROWS * sizeof(T));
    if constexpr (TP::ON)
        qtip_pgs<T, Tp, DppQuad, JM_QTIP_DEPTH>(C, C.friction ? C.friction[r] : P[L::OPT + 8], (int)(threadIdx.x & 3),
                                                xs + (threadIdx.x >> 2) * XS, zs + (threadIdx.x >> 2) * ZS, yps + (threadIdx.x >> 2) * XS,
                                                vis + (threadIdx.x >> 2) * VS_, ws, g0);
}

template<class T, class Tp, int INIT>
__global__ void __launch_bounds__((64 * qcon_block_waves<T, Tp>())) __attribute__((amdgpu_waves_per_eu(1)))
k_quad_con_gen(const BatchArgs<T> A, const QConArgs<T> C)
{
    using Q = QLayout<Tp>;
    constexpr int NTH = 64 * qcon_block_waves<T, Tp>();
    constexpr int CAP = qcon_capacity<T, Tp>();
    constexpr int RSTRIDE = (CAP % 2 == 0) ? CAP + 1 : CAP;
    __shared__ T table[Q::TABLE];
    __shared__ T stage_l[QRows<Tp>::NL * NTH];
    __shared__ T stage_b[QRows<Tp>::NB * (NTH / 4)];
    __shared__ T con[(CAP > 0 ? RSTRIDE : 1) * (NTH / 4)];
#pragma nounroll
    for (int i = threadIdx.x; i < Q::TABLE; i += NTH) table[i] = A.P[Q::OFFSET + i];
    const long long r = (long long)blockIdx.x * (NTH / 4) + (threadIdx.x >> 2);
    const int k = threadIdx.x & 3;
    if (r >= A.B) return;
    const StageBuf<T, NTH, NTH / 4> S{stage_l + threadIdx.x, stage_b + (threadIdx.x >> 2), k == 0};
    const QStore<T> V = qcon_store<T, Tp>(con + (threadIdx.x >> 2) * RSTRIDE, C.ws, r, (unsigned)A.B);
    quad_lane_run<T, Tp, DppQuad, NTH, NTH / 4, true, CAP, true, 0, INIT>(A, r, k, table, S, &C, &V);
}
#endif
}  // namespace jm
