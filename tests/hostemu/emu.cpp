// Host emulation of the per-lane kernel code (tests only, never part of the product path):
// compiles jiminy_amd/csrc/jm_kernels.h with g++ (-DJM_HOST_EMU) so that the kernel logic can be
// compared with the oracle on machines without a GPU.
#define JM_HOST_EMU 1
#include <pthread.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <string>
#include <thread>
#include <vector>
#include JM_TOPO_HEADER
#include "../../jiminy_amd/csrc/jm_kernels.h"
#include "../../jiminy_amd/csrc/jm_constraint.h"
#include "../../jiminy_amd/csrc/jm_qcon.h"
#include "../../jiminy_amd/csrc/jm_pack.h"
#include "../../jiminy_amd/csrc/jm_qdopri.h"

extern "C"
{
struct emu_io
{
    long long B;
    void *q, *v, *a, *command, *u_motor, *u, *f_external, *contact_forces, *imu, *force, *contact,
        *encoder, *effort, *energy, *joint_forces, *centroidal, *status, *q_in, *v_in, *a_out, *mask, *q_init, *v_init;
};
const char * emu_signature() { return Topo::signature; }
}

// ---- emulation of one DPP quad: 4 host threads in lock-step through a barrier
struct QuadShared
{
    pthread_barrier_t bar;
    double buf[4];
    int ibuf[4];
};
// EMU_UNDEF_CHECK (with amdclang++ -ftrivial-auto-var-init=pattern, which turns every uninitialised floating
// point local into the all-ones NaN): a quad exchange whose operand is uninitialised on ANY of the four lanes
// is reported.  On the device such an operand is `undef` on those lanes to the compiler even when only the
// source lane's value is read, and a DPP move may then read an arbitrary register.
#ifdef EMU_UNDEF_CHECK
#include <execinfo.h>
#include <atomic>
static std::atomic<int> g_undef_reports{0};
template<class T> static inline void undef_check(T x)
{
    // clang's pattern: all-ones for floating point (a NaN), 0xAA bytes for integers
    bool bad;
    if constexpr (std::is_integral<T>::value) { unsigned b32; std::memcpy(&b32, &x, 4); bad = b32 == 0xAAAAAAAAu; }
    else if constexpr (sizeof(T) == 4) { unsigned b32; std::memcpy(&b32, &x, 4); bad = b32 == 0xFFFFFFFFu; }
    else { unsigned long long bits; std::memcpy(&bits, &x, 8); bad = bits == 0xFFFFFFFFFFFFFFFFull; }
    if (bad && g_undef_reports.fetch_add(1) < 40)
    {
        void * bt[6];
        const int n = backtrace(bt, 6);
        backtrace_symbols_fd(bt, n, 2);
        fprintf(stderr, "---- uninitialised quad operand\n");
    }
}
extern "C" int emu_undef_reports() { return g_undef_reports.load(); }
#else
template<class T> static inline void undef_check(T) {}
#endif
#ifdef JM_HOST_EMU_TRACE
static thread_local long long g_bar_count = 0;
#define pthread_barrier_wait(b) (++g_bar_count, (pthread_barrier_wait)(b))
#endif
struct HostQuad
{
    static thread_local QuadShared * sh;
    static thread_local int k;
    template<class T> static T quad_sum(T x)
    {
        undef_check(x);
        sh->buf[k] = (double)x;
        pthread_barrier_wait(&sh->bar);
        // same association as the DPP butterflies: (x_k + x_{k^1}) + (x_{k^2} + x_{k^3})
        const T a = (T)sh->buf[k] + (T)sh->buf[k ^ 1];
        const T b = (T)sh->buf[k ^ 2] + (T)sh->buf[k ^ 3];
        pthread_barrier_wait(&sh->bar);
        return a + b;
    }
    template<int LANE, class T> static T bcast(T x)
    {
        undef_check(x);
        sh->buf[k] = (double)x;
        pthread_barrier_wait(&sh->bar);
        const T r = (T)sh->buf[LANE];
        pthread_barrier_wait(&sh->bar);
        return r;
    }
    template<int CTRL, class T> static T perm_(T x)
    {
        undef_check(x);
        sh->buf[k] = (double)x;
        pthread_barrier_wait(&sh->bar);
        const int src = (CTRL >> (2 * k)) & 3;   // quad_perm: 2 bits per destination lane
        const T r = (T)sh->buf[src];
        pthread_barrier_wait(&sh->bar);
        return r;
    }
    static bool wave_any(bool p) { return p; }   // one robot at a time: its four lanes agree on robot-level predicates
    template<class T> static T max_abs(T a, T b) { return std::fmax(a, std::fabs(b)); }
    template<class T> static T max_(T a, T b) { return std::fmax(a, b); }
    static void sync() { pthread_barrier_wait(&sh->bar); }
    static void fence() { pthread_barrier_wait(&sh->bar); }   // device: memory fence within the wave; here the lanes are threads
    static void table_ready() {}
    static int quad_or(int x)
    {
        sh->ibuf[k] = x;
        pthread_barrier_wait(&sh->bar);
        const int r = sh->ibuf[0] | sh->ibuf[1] | sh->ibuf[2] | sh->ibuf[3];
        pthread_barrier_wait(&sh->bar);
        return r;
    }
};
thread_local QuadShared * HostQuad::sh = nullptr;
thread_local int HostQuad::k = 0;

template<class T, class Tp, bool GEN = false> static void run_quad(const jm::BatchArgs<T> & A, const std::vector<T> & P)
{
    if constexpr (Tp::QUAD)
    {
        QuadShared sh;
        pthread_barrier_init(&sh.bar, nullptr, 4);
        const T * table = P.data() + jm::QLayout<Tp>::OFFSET;
        std::vector<std::thread> th;
        for (int k = 0; k < 4; ++k)
            th.emplace_back([&, k]() {
                HostQuad::sh = &sh;
                HostQuad::k = k;
                // poisoned: the device LDS is not zero-initialised either
                std::vector<T> sl(jm::QRows<Tp>::NL + 1, (T)std::nan("")), sb(jm::QRows<Tp>::NB + 1, (T)std::nan(""));
                const jm::StageBuf<T, 1, 1> S{sl.data(), sb.data(), true};  // private trunk rows per thread
                for (long long r = 0; r < A.B; ++r) jm::quad_lane_run<T, Tp, HostQuad, 1, 1, false, 0, GEN>(A, r, k, table, S);
            });
        for (auto & t : th) t.join();
        pthread_barrier_destroy(&sh.bar);
    }
    else { (void)A; (void)P; }
}

static int g_guard_violations = 0;
extern "C" int emu_guard_violations() { return g_guard_violations; }
// branch-parallel kernel with the constraint contact model (jm_qcon.h): the robot's solver region is split
// between a small "on-chip" array and overflow rows, so that both homes of QStore are exercised
template<class T, class Tp, bool GEN = false> static void run_quad_con(const jm::BatchArgs<T> & A, const std::vector<T> & P, const jm::QConArgs<T> & C0)
{
    if constexpr (Tp::QUAD)
    {
        QuadShared sh;
        pthread_barrier_init(&sh.bar, nullptr, 4);
        const T * table = P.data() + jm::QLayout<Tp>::OFFSET;
        constexpr int CAP = 150;   // solves of up to 13 rows take the on-chip path, larger ones overflow
        const int rows = jm::QConRows<Tp>::ws_rows(CAP);
        // (guard rows behind the workspace: a solver that indexes past its region is caught below)
        constexpr int GUARD = 512;
        const T sentinel = (T)-12345.678;
        std::vector<T> lds((size_t)(CAP + 1) * A.B, (T)std::nan("")), hbm((size_t)(rows + 1 + GUARD) * A.B, (T)std::nan(""));
        for (size_t i = (size_t)(rows + 1) * A.B; i < hbm.size(); ++i) hbm[i] = sentinel;
        std::vector<std::thread> th;
        for (int k = 0; k < 4; ++k)
            th.emplace_back([&, k]() {
                HostQuad::sh = &sh;
                HostQuad::k = k;
                // poisoned: the device LDS is not zero-initialised either
                std::vector<T> sl(jm::QRows<Tp>::NL + 1, (T)std::nan("")), sb(jm::QRows<Tp>::NB + 1, (T)std::nan(""));
                const jm::StageBuf<T, 1, 1> S{sl.data(), sb.data(), true};
                for (long long r = 0; r < A.B; ++r)
                {
                    jm::QConArgs<T> C = C0;
                    C.ws = hbm.data();
                    const jm::QStore<T> V{lds.data() + (size_t)r * CAP, hbm.data() + r, (unsigned)A.B, CAP};
                    jm::quad_lane_run<T, Tp, HostQuad, 1, 1, true, CAP, GEN>(A, r, k, table, S, &C, &V);
                }
            });
        for (auto & t : th) t.join();
        pthread_barrier_destroy(&sh.bar);
        for (size_t i = (size_t)(rows + 1) * A.B; i < hbm.size(); ++i)
            if (hbm[i] != sentinel) { g_guard_violations += 1; break; }
    }
    else { (void)A; (void)P; (void)C0; }
}

// the same launch in the split form of robots with large solves (jm_qcon.h: k_quad_con_split<1> | k_qcon_pgs | k_quad_con_split<2>
// per evaluation, stage buffer and solver region persistent between the parts): one robot at a time, its four lanes as threads
static int g_split = 0;
static long long g_lane_solves = 0;
extern "C" long long emu_lane_solves() { return g_lane_solves; }
static long long g_tip_solves = 0;   // solves that took the operational-space form (jm_qtip.h)
extern "C" long long emu_tip_solves() { return g_tip_solves; }
extern "C" void emu_set_split(int on) { g_split = on; }
extern "C" int emu_has_split() { return jm::qcon_split<Topo>() ? 1 : 0; }
template<class T, class Tp> static void run_quad_con_split(const jm::BatchArgs<T> & A, const std::vector<T> & P, const jm::QConArgs<T> & C0)
{
    if constexpr (jm::qcon_split<Tp>())
    {
        using SR = jm::QSplitRows<Tp>;
        using RG = jm::QSplitRegion<Tp>;
        QuadShared sh;
        pthread_barrier_init(&sh.bar, nullptr, 4);
        const T * table = P.data() + jm::QLayout<Tp>::OFFSET;
        const int pre = A.command_changed ? 1 : 0;
        const int n_evals = pre + A.n_sub * (A.solver == JM_SOLVER_RUNGE_KUTTA_4 ? 4 : 1);
        constexpr int GUARD = 512;
        const T sentinel = (T)-12345.678;
        std::vector<T> region((size_t)RG::ROWS * A.B + GUARD, (T)std::nan(""));
        for (size_t i = (size_t)RG::ROWS * A.B; i < region.size(); ++i) region[i] = sentinel;
        // on-chip arrays of the solve, shared by the four lanes of the robot
        std::vector<jm::QPair<T>> xs(8 * 12 / 2 + 2);
        std::vector<T> zs(jm::QTip<Tp>::ZPAD + 4, (T)std::nan(""));
        std::vector<T> yps(jm::QConRows<Tp>::MAXM + 4, (T)std::nan(""));   // residuals of the previous sweep (on chip on the device)
        std::vector<unsigned short> vis(8 * 12 + 8);
        std::vector<std::thread> th;
        for (int k = 0; k < 4; ++k)
            th.emplace_back([&, k]() {
                HostQuad::sh = &sh;
                HostQuad::k = k;
                std::vector<T> sl(SR::NL + 1, (T)std::nan("")), sb(SR::NB + 1, (T)std::nan(""));
                const jm::StageBuf<T, 1, 1> S{sl.data(), sb.data(), true};   // private trunk rows per thread
                for (long long r = 0; r < A.B; ++r)
                {
                    jm::QConArgs<T> C = C0;
                    C.ws = region.data();
                    C.stage = nullptr; C.split_pass = 0; C.split_r0 = 0; C.split_r1 = (int)A.B;
                    const jm::QStore<T> V{nullptr, region.data() + (size_t)r * RG::ROWS, 1u, 0};
                    const T friction = C.friction ? C.friction[r] : P[jm::Layout<Tp>::OPT + 8];
                    char * ws = (char *)region.data();
                    const unsigned g0 = (unsigned)((size_t)r * RG::ROWS * sizeof(T));
                    auto solve = [&]() {
                        // (robots whose solve fits the fixed 16-row layout: one lane per robot, jm_lib.cpp launches k_qcon_pgs_lane
                        // ahead of the streamed form, which then finds them marked done)
                        if constexpr (jm::QLanePgs<Tp>::FITS)
                        {
                            if (k == 0 && A.mode == jm::MODE_STEP)
                            {
                                jm::qcon_pgs_lane_any<T, Tp, HostQuad>(C, friction, region.data() + (size_t)r * RG::ROWS, (int32_t *)nullptr, true);
                                ++g_lane_solves;
                            }
                            HostQuad::sync();
                        }
                        bool tip = false;
                        if constexpr (jm::QTip<Tp>::ON)
                            tip = jm::qtip_pgs<T, Tp, HostQuad, JM_QTIP_DEPTH>(C, friction, k, (T *)xs.data(), zs.data(), yps.data(), vis.data(), ws, g0);
                        if (tip) { if (k == 0) ++g_tip_solves; }
                        else if (!jm::qcon_pgs_lean<T, Tp, HostQuad, 8, 0, JM_QCON_PGS_DEPTH>(C, friction, k, (T *)xs.data(), vis.data(), ws, g0))
                            jm::qcon_pgs_lean<T, Tp, HostQuad, 12, 64, JM_QCON_PGS_DEPTH - 1>(C, friction, k, (T *)xs.data(), vis.data(), ws, g0);
                    };
                    if (A.mode == jm::MODE_START || A.mode == jm::MODE_RESET)
                    {
                        // Engine::start / reset in the split form (jm_lib.cpp launch_quad_con): first pass | exact solve |
                        // 3 x (pass | Gauss-Seidel) | closing evaluation
                        if (A.mode == jm::MODE_RESET && !A.mask[r]) continue;
                        C.split_e = 0;
                        jm::quad_lane_run<T, Tp, HostQuad, 1, 1, true, 0, false, 1>(A, r, k, table, S, &C, &V);
                        HostQuad::sync();
                        {
                            T * reg = region.data() + (size_t)r * RG::ROWS;
                            const int hdr = (int)reg[RG::HDR];
                            const int m_ = hdr & 0xff;
                            bool ok = true;
                            if ((hdr >> 24) & 1) { if constexpr (jm::QTip<Tp>::ON) { if (k == 0) { ok = jm::qtip_exact<T, Tp, HostQuad, 1>(0, ws, (size_t)g0); ++g_tip_solves; } } }
                            else if (m_ > 0)
                            {
                                const jm::QStoreSq<T> W{reg};
                                ok = jm::qcon_chol<T, HostQuad, jm::QStoreSq<T>, true>(k, m_, W);
                                HostQuad::sync();
                                if (k == 0) reg[RG::OK] = ok ? T(1) : T(0);
                            }
                        }
                        HostQuad::sync();
                        for (int pass = 1; pass <= 3; ++pass)
                        {
                            C.split_pass = pass;
                            jm::quad_lane_run<T, Tp, HostQuad, 1, 1, true, 0, false, 1>(A, r, k, table, S, &C, &V);
                            HostQuad::sync();
                            solve();
                            HostQuad::sync();
                        }
                        C.split_pass = 4;
                        jm::quad_lane_run<T, Tp, HostQuad, 1, 1, true, 0, false, 2>(A, r, k, table, S, &C, &V);
                        HostQuad::sync();
                        continue;
                    }
                    for (int e = 0; e < n_evals; ++e)
                    {
                        C.split_e = e;
                        jm::quad_lane_run<T, Tp, HostQuad, 1, 1, true, 0, false, 1>(A, r, k, table, S, &C, &V);
#ifdef JM_HOST_EMU_TRACE
                        std::fprintf(stderr, "[%d] r=%lld e=%d after pre: %lld barriers\n", k, r, e, g_bar_count);
#endif
                        HostQuad::sync();
                        solve();
                        HostQuad::sync();
                        jm::quad_lane_run<T, Tp, HostQuad, 1, 1, true, 0, false, 2>(A, r, k, table, S, &C, &V);
                        HostQuad::sync();
                    }
                }
            });
        for (auto & t : th) t.join();
        pthread_barrier_destroy(&sh.bar);
        for (size_t i = (size_t)RG::ROWS * A.B; i < region.size(); ++i)
            if (region[i] != sentinel) { g_guard_violations += 1; break; }
    }
    else { (void)A; (void)P; (void)C0; }
}

// constraint contact model: options + per-lane state rows (flags int32 [NF][B], data [ND][B]); the
// delassus workspace is allocated here
static jm_constraint_options g_copt = {JM_CONTACT_SPRING_DAMPER, 100, 0.0, 20.0, 1.0e-3, 1.0e-5, 1.0e-4, -1.0};
static void * g_con_flags = nullptr;
static void * g_con_data = nullptr;
static void * g_friction = nullptr;
extern "C" void emu_set_friction(void * friction) { g_friction = friction; }
static const void * g_flex_lane = nullptr;
extern "C" void emu_set_flexibility(const void * flex) { g_flex_lane = flex; }
extern "C" void emu_set_constraints(const jm_constraint_options * o, void * flags, void * data)
{
    g_copt = *o;
    g_con_flags = flags;
    g_con_data = data;
}
extern "C" void emu_constraint_rows(int * nf, int * nd, int * nw)
{
    *nf = jm::ConRows<Topo>::NF; *nd = jm::ConRows<Topo>::ND; *nw = jm::ConRows<Topo>::WTOTAL;
}
// optional per-environment variation (GEN instantiation of the branch-parallel code)
static const void * g_model_lane = nullptr;
static const void * g_ground = nullptr;
static const void * g_ground_off = nullptr;
extern "C" void emu_set_ground_offset(const void * off) { g_ground_off = off; }
static int g_gnx = 0, g_gny = 0;
static double g_gx0 = 0, g_gy0 = 0, g_gdx = 1, g_gdy = 1;
static const void * g_applied = nullptr;
static int g_applied_k = 0;
static double g_applied_p[12] = {0};
static int g_applied_joint[4] = {1, 1, 1, 1};
extern "C" void emu_set_gen(const void * model_lane, const void * ground, int nx, int ny, double x0, double y0, double dx, double dy,
                            const void * applied, int k, const double * offsets, const int * joints)
{
    g_model_lane = model_lane; g_ground = ground; g_gnx = nx; g_gny = ny; g_gx0 = x0; g_gy0 = y0; g_gdx = dx; g_gdy = dy;
    g_applied = applied; g_applied_k = applied ? k : 0;
    for (int i = 0; i < 3 * g_applied_k; ++i) g_applied_p[i] = offsets[i];
    for (int i = 0; i < 4; ++i) g_applied_joint[i] = (joints && i < g_applied_k) ? joints[i] : 1;
}
static int g_variant = 0;  // 0 = one robot per lane, 1 = limb-parallel (4 lanes per robot)
extern "C" void emu_set_variant(int v) { g_variant = v; }
extern "C" int emu_has_quad() { return Topo::QUAD ? 1 : 0; }

template<class T>
static int run(const jm_model_desc * d, const jm_options * o, const emu_io * io, int mode, int solver, double dt,
               int n_sub, int command_changed, int update_sensors)
{
    std::string why;
    if (!jm::check_topology<Topo>(*d, why)) return JM_ETOPOLOGY;
    std::vector<double> Pd = jm::pack_model<Topo>(*d);
    jm::pack_options<Topo>(Pd, *o);
    jm::pack_quad<Topo>(Pd, *d);
    std::vector<T> P(Pd.begin(), Pd.end());
    jm::BatchArgs<T> A;
    std::memset(&A, 0, sizeof(A));
    A.P = P.data();
    A.q = (T *)io->q; A.v = (T *)io->v; A.a = (T *)io->a; A.command = (const T *)io->command;
    A.u_motor = (T *)io->u_motor; A.u = (T *)io->u; A.f_external = (T *)io->f_external;
    A.contact_forces = (T *)io->contact_forces; A.imu = (T *)io->imu; A.force = (T *)io->force;
    A.contact = (T *)io->contact; A.encoder = (T *)io->encoder; A.effort = (T *)io->effort;
    A.energy = (T *)io->energy; A.joint_forces = (T *)io->joint_forces; A.centroidal = (T *)io->centroidal;
    A.status = (int32_t *)io->status;
    A.q_in = (const T *)io->q_in; A.v_in = (const T *)io->v_in; A.a_out = (T *)io->a_out;
    A.mask = (const unsigned char *)io->mask; A.q_init = (const T *)io->q_init; A.v_init = (const T *)io->v_init;
    A.B = io->B; A.mode = mode; A.solver = solver; A.n_sub = n_sub; A.command_changed = command_changed;
    A.update_sensors = update_sensors; A.dt = (T)dt;
    A.friction = g_copt.contact_model == JM_CONTACT_CONSTRAINT ? nullptr : (const T *)g_friction;
    A.flex_lane = (const T *)g_flex_lane;
    // (per-lane friction alone: the variation form of the branch-parallel code; the one-robot-per-lane code reads it as it is)
    const bool gen = g_model_lane || g_ground || g_applied || (A.friction && g_variant == 1 && Topo::QUAD);
    A.model_lane = (const T *)g_model_lane;
    A.ground_h = (const T *)g_ground; A.ground_nx = g_gnx; A.ground_ny = g_gny;
    A.ground_off = g_ground ? (const T *)g_ground_off : nullptr;
    A.ground_x0 = (T)g_gx0; A.ground_y0 = (T)g_gy0; A.ground_dx = (T)g_gdx; A.ground_dy = (T)g_gdy;
    A.applied = (const T *)g_applied; A.applied_k = g_applied_k;
    for (int i = 0; i < 12; ++i) A.applied_p[i] = (T)g_applied_p[i];
    for (int i = 0; i < 4; ++i) A.applied_joint[i] = g_applied_joint[i];
    // (the one-robot-per-lane code reads the lane's friction and flexibility as they are, applied wrenches and body parameters in
    // its variation instantiation; height maps exist in the variation form of the branch-parallel code only)
    if (g_ground && !(g_variant == 1 && Topo::QUAD) && g_copt.contact_model == JM_CONTACT_CONSTRAINT) return JM_ENOTIMPL;
    if (g_variant == 1 && Topo::QUAD)
    {
        if (g_copt.contact_model == JM_CONTACT_CONSTRAINT)
        {
            jm::QConArgs<T> C;
            C.flags = (int32_t *)g_con_flags; C.data = (T *)g_con_data; C.ws = nullptr;
            C.friction = (const T *)g_friction;
            const double omega = 2.0 * 3.14159265358979323846 * g_copt.stabilization_freq;
            C.kp = (T)(omega * omega); C.kd = (T)(2.0 * omega);
            {
                const double omega_u = 2.0 * 3.14159265358979323846 * g_copt.user_stabilization_freq;
                C.kp_lock = g_copt.user_stabilization_freq < 0.0 ? C.kp : (T)(omega_u * omega_u);
                C.kd_lock = g_copt.user_stabilization_freq < 0.0 ? C.kd : (T)(2.0 * omega_u);
            }
            C.torsion = (T)g_copt.torsion; C.reg = (T)g_copt.regularization;
            C.tol_abs = (T)g_copt.tol_abs; C.tol_rel = (T)g_copt.tol_rel; C.iter_max = g_copt.pgs_iter_max;
            C.ground_h = A.ground_h; C.ground_nx = A.ground_nx; C.ground_ny = A.ground_ny;
            C.ground_x0 = A.ground_x0; C.ground_y0 = A.ground_y0; C.ground_dx = A.ground_dx; C.ground_dy = A.ground_dy;
            C.stage = nullptr; C.split_e = 0; C.split_r0 = 0; C.split_r1 = (int)A.B;
            // user-registered JointConstraints (bit 2 of a joint row): kernels built with them (jm::qcon_locks)
            bool locks = false;
            if (!jm::qcon_split<Topo>() && g_con_flags)
                for (long long i = 0; i < (long long)jm::ConRows<Topo>::NB * A.B && !locks; ++i) locks = (((const int32_t *)g_con_flags)[i] & 4) != 0;
            const bool gen_ = gen || locks;
            bool split = false;
            if constexpr (std::is_same<T, double>::value)
                if (!gen && g_split && jm::qcon_split<Topo>() && (mode == jm::MODE_STEP || mode == jm::MODE_START || mode == jm::MODE_RESET))
                {
                    run_quad_con_split<T, Topo>(A, P, C);
                    split = true;
                }
            if (split) {}
            else if (gen_) run_quad_con<T, Topo, true>(A, P, C);
            else run_quad_con<T, Topo>(A, P, C);
        }
        else if (gen) run_quad<T, Topo, true>(A, P);
        else run_quad<T, Topo>(A, P);
        return 0;
    }
    std::vector<T> sb(jm::lane_rows<T, Topo>() + 1, (T)std::nan(""));
    if (g_copt.contact_model == JM_CONTACT_CONSTRAINT)
    {
        std::vector<T> wsp((size_t)(jm::ConRows<Topo>::WTOTAL + 1) * io->B, (T)std::nan(""));
        jm::ConArgs<T> C;
        C.flags = (int32_t *)g_con_flags; C.data = (T *)g_con_data; C.ws = wsp.data();
        C.friction = (const T *)g_friction;
        const double omega = 2.0 * 3.14159265358979323846 * g_copt.stabilization_freq;
        C.kp = (T)(omega * omega); C.kd = (T)(2.0 * omega);
            {
                const double omega_u = 2.0 * 3.14159265358979323846 * g_copt.user_stabilization_freq;
                C.kp_lock = g_copt.user_stabilization_freq < 0.0 ? C.kp : (T)(omega_u * omega_u);
                C.kd_lock = g_copt.user_stabilization_freq < 0.0 ? C.kd : (T)(2.0 * omega_u);
            }
        C.torsion = (T)g_copt.torsion; C.reg = (T)g_copt.regularization;
        C.tol_abs = (T)g_copt.tol_abs; C.tol_rel = (T)g_copt.tol_rel; C.iter_max = g_copt.pgs_iter_max;
        std::vector<T> xvec(jm::ConRows<Topo>::NR + 1, (T)std::nan(""));
        C.xl = xvec.data(); C.xstride = 1;
        std::vector<T> yvec(8, (T)std::nan(""));
        C.yl = yvec.data(); C.ystride = 1; C.yrows = 7;   // exercises both homes of the residuals
        C.park = nullptr; C.park_rows = 0;
        // (applied wrenches: the instantiation that reads them, like the library's launch)
        for (long long lane = 0; lane < io->B; ++lane)
        {
            if (A.applied || A.model_lane) jm::lane_run<T, Topo, 1, jm::WithConA>(A, lane, sb.data(), C);
            else jm::lane_run<T, Topo, 1, jm::WithCon>(A, lane, sb.data(), C);
        }
        return 0;
    }
    for (long long lane = 0; lane < io->B; ++lane)
    {
        if (A.applied || A.model_lane || A.ground_h) jm::lane_run<T, Topo, 1, jm::NoConA>(A, lane, sb.data());
        else jm::lane_run<T, Topo, 1>(A, lane, sb.data());
    }
    return 0;
}
extern "C" int emu_run(const jm_model_desc * d, const jm_options * o, const emu_io * io, int dtype, int mode, int solver,
            double dt, int n_sub, int command_changed, int update_sensors)
{
    if (dtype == JM_F64) return run<double>(d, o, io, mode, solver, dt, n_sub, command_changed, update_sensors);
    return run<float>(d, o, io, mode, solver, dt, n_sub, command_changed, update_sensors);
}

// persistent adaptive stepper of the branch-parallel decomposition (jm_qdopri.h), float64, spring-damper contacts
template<class Topo>
static int run_dopri_impl(const jm_model_desc * d, const jm_options * o, const emu_io * io, double * fs, int32_t * is,
                          double t_next, double tol_rel, double tol_abs, double dt_max, double dt_restore, int succ_failed_max,
                          int new_step, int max_attempts, int32_t * counters)
{
    using T = double;
    if constexpr (Topo::QUAD)
    {
        std::string why;
        if (!jm::check_topology<Topo>(*d, why)) return JM_ETOPOLOGY;
        std::vector<double> P = jm::pack_model<Topo>(*d);
        jm::pack_options<Topo>(P, *o);
        jm::pack_quad<Topo>(P, *d);
        jm::BatchArgs<T> A;
        std::memset(&A, 0, sizeof(A));
        A.P = P.data();
        A.q = (T *)io->q; A.v = (T *)io->v; A.a = (T *)io->a; A.command = (const T *)io->command;
        A.status = (int32_t *)io->status;
        A.B = io->B; A.mode = jm::MODE_DYNAMICS;
        std::vector<T> ws((size_t)(jm::AdaptiveRows<Topo>::TOTAL + 1) * io->B, std::nan(""));
        jm::AdaptiveArgs<T> D;
        std::memset(&D, 0, sizeof(D));
        D.P = P.data(); D.q = A.q; D.v = A.v; D.a = A.a; D.ws = ws.data(); D.command = A.command;
        D.fs = fs; D.is = is; D.status = A.status; D.n_active = counters; D.B = io->B;
        D.t_next = t_next; D.tol_rel = tol_rel; D.tol_abs = tol_abs; D.dt_max = dt_max; D.dt_restore_threshold_rel = dt_restore;
        D.succ_failed_max = succ_failed_max; D.new_step = new_step;
        counters[0] = counters[1] = 0;
        QuadShared sh;
        pthread_barrier_init(&sh.bar, nullptr, 4);
        const T * table = P.data() + jm::QLayout<Topo>::OFFSET;
        std::vector<std::thread> th;
        for (int k = 0; k < 4; ++k)
            th.emplace_back([&, k]() {
                HostQuad::sh = &sh;
                HostQuad::k = k;
                std::vector<T> sl(jm::QRows<Topo>::NL + 1, std::nan("")), sb(jm::QDopriRows<Topo>::NB + 1, std::nan(""));
                const jm::StageBuf<T, 1, 1> S{sl.data(), sb.data(), true};
                for (long long r = 0; r < A.B; ++r) jm::quad_dopri_run<T, Topo, HostQuad, 1, 1>(A, D, r, k, table, S, max_attempts);
            });
        for (auto & t : th) t.join();
        pthread_barrier_destroy(&sh.bar);
        return 0;
    }
    else
    {
        (void)d; (void)o; (void)io; (void)fs; (void)is; (void)t_next; (void)tol_rel; (void)tol_abs; (void)dt_max; (void)dt_restore;
        (void)succ_failed_max; (void)new_step; (void)max_attempts; (void)counters;
        return JM_ENOTIMPL;
    }
}
extern "C" int emu_run_dopri(const jm_model_desc * d, const jm_options * o, const emu_io * io, double * fs, int32_t * is,
                             double t_next, double tol_rel, double tol_abs, double dt_max, double dt_restore, int succ_failed_max,
                             int new_step, int max_attempts, int32_t * counters)
{
    return run_dopri_impl<::Topo>(d, o, io, fs, is, t_next, tol_rel, tol_abs, dt_max, dt_restore, succ_failed_max, new_step,
                                  max_attempts, counters);
}
