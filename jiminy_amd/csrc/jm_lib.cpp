// jm_lib.cpp -- C ABI (include/jiminy_hip.h) of the per-topology HIP library, compiled with
// `hipcc --offload-arch=gfx950 -x hip -DJM_TOPO_HEADER="topo_<hash>.h"` (jiminy_amd/codegen.py).
//
// The library owns only the model constants (host copy + one small device block per batch).
// All batch state is borrowed from the caller as raw device pointers; nothing is allocated
// inside start/step/dynamics (mirrors the reference's "no malloc during step" guarantee,
// core/unit/engine_sanity_check.cc:118-121).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <random>
#include <string>
#include <vector>

#ifndef JM_TOPO_HEADER
#error "JM_TOPO_HEADER must name the generated topology header"
#endif
#include JM_TOPO_HEADER

#include "jm_kernels.h"
#include "jm_constraint.h"
#include "jm_qcon.h"
#include "jm_pack.h"
#include "jm_blocks.h"
#include "jm_adaptive.h"
#include "jm_qdopri.h"
#include "jm_random.h"

#define JM_ABI_VERSION 9

#ifdef JM_SPLIT_CONSTRAINT
// the constraint-model kernel is instantiated by jm_lib_constraint.cpp (compiled in parallel)
namespace jm
{
extern template __global__ void k_constrained<double, Topo, false>(const BatchArgs<double>, const ConArgs<double>);
#if !JM_TOPO_QUAD
extern template __global__ void k_batch<double, Topo, true>(const BatchArgs<double>);
extern template __global__ void k_constrained<double, Topo, true>(const BatchArgs<double>, const ConArgs<double>);
#endif
#if JM_TOPO_QUAD
extern template __global__ void k_quad_con<double, Topo, 0>(const BatchArgs<double>, const QConArgs<double>);
extern template __global__ void k_quad_con<double, Topo, 1>(const BatchArgs<double>, const QConArgs<double>);
extern template __global__ void k_quad_gen<double, Topo>(const BatchArgs<double>);
extern template __global__ void k_quad_con_gen<double, Topo, 0>(const BatchArgs<double>, const QConArgs<double>);
extern template __global__ void k_quad_con_gen<double, Topo, 1>(const BatchArgs<double>, const QConArgs<double>);
extern template __global__ void k_quad_dopri<double, Topo>(const BatchArgs<double>, const AdaptiveArgs<double>, int);
extern template __global__ void k_quad_dopri_gen<double, Topo>(const BatchArgs<double>, const AdaptiveArgs<double>, int);
#endif
#if JM_TOPO_QCON_SPLIT
extern template __global__ void k_quad_con_split<double, Topo, 1, 0>(const BatchArgs<double>, const QConArgs<double>);
extern template __global__ void k_quad_con_split<double, Topo, 2, 0>(const BatchArgs<double>, const QConArgs<double>);
extern template __global__ void k_quad_con_split<double, Topo, 1, 1>(const BatchArgs<double>, const QConArgs<double>);
extern template __global__ void k_quad_con_split<double, Topo, 2, 1>(const BatchArgs<double>, const QConArgs<double>);
extern template __global__ void k_qcon_pgs<double, Topo, 8, 0, JM_QCON_PGS_DEPTH>(const QConArgs<double>, const double *, unsigned);
extern template __global__ void k_qcon_pgs<double, Topo, 12, 64, JM_QCON_PGS_DEPTH - 1>(const QConArgs<double>, const double *, unsigned);
extern template __global__ void k_qtip_pgs<double, Topo>(const QConArgs<double>, const double *, unsigned);
extern template __global__ void k_qcon_pgs_lane<double, Topo>(const QConArgs<double>, const double *, int32_t *);
extern template __global__ void k_qcon_exact<double, Topo>(const QConArgs<double>);
extern template __global__ void k_qtip_exact<double, Topo>(const QConArgs<double>);
#endif
}
#endif

namespace
{
thread_local std::string g_last_error;

int32_t fail(int32_t code, const std::string & msg)
{
    g_last_error = msg;
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do                                                                                             \
    {                                                                                              \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(JM_ERUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)
}  // namespace

struct jm_model
{
    std::vector<double> params;  // Layout<Topo>, options tail = defaults
    bool root_at_origin = true;  // placement of joint 1 is the identity (limb-parallel kernel)
};

enum { VARIANT_LANE = 0, VARIANT_QUAD = 1 };

struct jm_batch
{
    int variant = VARIANT_LANE;
    const jm_model * model = nullptr;
    long long B = 0;
    int dtype = JM_F64;
    int device = 0;
    std::vector<double> params;
    void * d_params = nullptr;
    void * field[JM_F_COUNT] = {};
    bool started = false;
    bool qcon_split = true;   // constraint model, large solves: split step launches (JIMINY_AMD_QCON_SPLIT=0 at creation: single kernel)
    bool qcon_split_start = true;   // ... and split start / reset launches (JIMINY_AMD_QCON_SPLIT_START=0: single kernel)
    bool joint_locks = false; // the batch carries user-registered JointConstraints (jm_batch_set_joint_locks)
    // split stepping of robots whose solve runs one lane per robot (jm_qcon.h, qcon_pgs_lane): the solve kernel counts the
    // robots it cannot take, its sweeps, its waves and its longest solve on the device (four counters per slot).  A batch
    // with misses, or whose solves are short (robots standing under control), steps with the single kernel for a while, then one
    // step in the split form probes again.  The decision for step n only reads the counters of steps <= n - 2 and WAITS for
    // them (they have long arrived: no stall), and `start` resets the state: the sequence of forms is a function of the
    // simulated data, not of host timing -- two runs from the same state are bit-identical (reference pin 11)
    static constexpr int LANE_SLOTS = 4;
    int32_t * lane_stat = nullptr;        // device, [LANE_SLOTS][4]
    int32_t * lane_stat_host = nullptr;   // pinned, [LANE_SLOTS][4]
    hipEvent_t lane_ev[LANE_SLOTS] = {};
    long long lane_step_of[LANE_SLOTS] = {-1, -1, -1, -1};   // step whose counters the slot is waiting for (-1: free)
    long long lane_step = 0;              // split-capable step launches since `start`
    int split_cooldown = 0;
    int split_chunks = 1;     // ... as this many independent chunks on streams of their own (JIMINY_AMD_QCON_SPLIT_CHUNKS; measured: no gain)
    hipStream_t split_stream[8] = {};
    hipEvent_t split_fork = nullptr, split_join[8] = {};
    bool split_streams_made = false;
    // adaptive stepper: caller-owned workspace / per-lane state, library-owned active-lane counter
    void * ad_ws = nullptr;
    double * ad_fs = nullptr;
    int32_t * ad_is = nullptr;
    int32_t * ad_count = nullptr;       // device
    int32_t * ad_flags = nullptr;       // device, compact constraint flags [NF][B] (constraint model + adaptive)
    // compact-batch overrides of the constraint state pointers (adaptive stepper), null = bound fields
    int32_t * ov_flags = nullptr; void * ov_data = nullptr; void * ov_ws = nullptr;
    int32_t * ad_count_host = nullptr;  // pinned host
    // constraint contact model (jm_constraint.h)
    jm_constraint_options copt = {JM_CONTACT_SPRING_DAMPER, 100, 0.0, 20.0, 1.0e-3, 1.0e-5, 1.0e-4, -1.0};
    // optional per-environment variation (GEN kernels): height map, frames of the applied wrenches
    const void * ground_h = nullptr;
    int ground_nx = 0, ground_ny = 0;
    double ground_x0 = 0, ground_y0 = 0, ground_dx = 1, ground_dy = 1;
    int applied_k = 0;
    double applied_p[12] = {0};
    int applied_joint[4] = {1, 1, 1, 1};
    int n_cus = 256;   // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    // per-launch timing with HIP events recorded on the launch stream (bench.py roofline leg)
    bool timing = false;
    std::vector<hipEvent_t> ev;  // pairs (begin, end), ring of JM_TIMING_RING launches
    size_t n_timed = 0;
};
#define JM_TIMING_RING 2048

namespace
{
int32_t upload_params(jm_batch * b)
{
    HIP_TRY(hipSetDevice(b->device));
    const size_t n = b->params.size();
    if (b->dtype == JM_F64)
    {
        HIP_TRY(hipMemcpy(b->d_params, b->params.data(), n * sizeof(double), hipMemcpyHostToDevice));
    }
    else
    {
        std::vector<float> pf(n);
        for (size_t i = 0; i < n; ++i) pf[i] = (float)b->params[i];
        HIP_TRY(hipMemcpy(b->d_params, pf.data(), n * sizeof(float), hipMemcpyHostToDevice));
    }
    return JM_OK;
}

// rows of the caller-owned constraint workspace: overflow of the per-robot solver region (branch-parallel
// kernel) or the dense per-lane delassus workspace (one-robot-per-lane kernel)
template<class Tp> int32_t constraint_ws_rows_of(const jm_batch * b)
{
    if constexpr (Tp::QUAD)
        if (b->variant == VARIANT_QUAD)
        {
            int rows = jm::qcon_ws_rows<double, Tp>();
            if constexpr (jm::qcon_split<Tp>())
                if (jm::qcon_split_ws_rows<double, Tp>() > rows) rows = jm::qcon_split_ws_rows<double, Tp>();
            return rows > 0 ? rows : 1;
        }
    return jm::ConRows<Tp>::WTOTAL;
}
int32_t constraint_ws_rows(const jm_batch * b) { return constraint_ws_rows_of<Topo>(b); }

template<class T> jm::BatchArgs<T> make_args(const jm_batch * b)
{
    jm::BatchArgs<T> A;
    std::memset(&A, 0, sizeof(A));
    A.P = (const T *)b->d_params;
    A.q = (T *)b->field[JM_F_Q];
    A.v = (T *)b->field[JM_F_V];
    A.a = (T *)b->field[JM_F_A];
    A.command = (const T *)b->field[JM_F_COMMAND];
    A.u_motor = (T *)b->field[JM_F_U_MOTOR];
    A.u = (T *)b->field[JM_F_U];
    A.f_external = (T *)b->field[JM_F_F_EXTERNAL];
    A.contact_forces = (T *)b->field[JM_F_CONTACT_FORCES];
    A.imu = (T *)b->field[JM_F_IMU];
    A.force = (T *)b->field[JM_F_FORCE];
    A.contact = (T *)b->field[JM_F_CONTACT];
    A.encoder = (T *)b->field[JM_F_ENCODER];
    A.effort = (T *)b->field[JM_F_EFFORT];
    A.energy = (T *)b->field[JM_F_ENERGY];
    A.joint_forces = (T *)b->field[JM_F_JOINT_FORCES];
    A.centroidal = (T *)b->field[JM_F_CENTROIDAL];
    A.status = (int32_t *)b->field[JM_F_STATUS];
    A.ws = (T *)b->field[JM_F_WORKSPACE];
    A.B = b->B;
    A.model_lane = (const T *)b->field[JM_F_MODEL_LANE];
    A.ground_h = (const T *)b->ground_h;
    A.ground_nx = b->ground_nx; A.ground_ny = b->ground_ny;
    A.ground_x0 = (T)b->ground_x0; A.ground_y0 = (T)b->ground_y0; A.ground_dx = (T)b->ground_dx; A.ground_dy = (T)b->ground_dy;
    A.ground_off = b->ground_h ? (const T *)b->field[JM_F_GROUND_OFFSET] : nullptr;
    A.applied = b->applied_k > 0 ? (const T *)b->field[JM_F_APPLIED] : nullptr;
    A.applied_k = A.applied ? b->applied_k : 0;
    for (int i = 0; i < 12; ++i) A.applied_p[i] = (T)b->applied_p[i];
    for (int i = 0; i < 4; ++i) A.applied_joint[i] = b->applied_joint[i];
    // spring-damper model: the lane's own friction coefficient when the field is bound (variation kernels)
    A.friction = b->copt.contact_model == JM_CONTACT_CONSTRAINT ? nullptr : (const T *)b->field[JM_F_FRICTION];
    A.flex_lane = (const T *)b->field[JM_F_FLEXIBILITY];
    return A;
}

// limb-parallel kernel (4 lanes per robot): only instantiated for topologies that have the structure
template<class T, class Tp> void launch_quad(jm_batch * b, jm::BatchArgs<T> & A, hipStream_t s)
{
    if constexpr (Tp::QUAD)
    {
        constexpr int nth = 64 * jm::quad_block_waves<T, Tp>();  // 4 lanes per robot
        const unsigned grid = (unsigned)((A.B + nth / 4 - 1) / (nth / 4));  // A.B <= b->B (compact adaptive batches)
        if constexpr (std::is_same<T, double>::value)
        {
            if (A.model_lane || A.ground_h || A.applied || A.friction)
            {
                hipLaunchKernelGGL((jm::k_quad_gen<T, Tp>), dim3(grid), dim3(nth), 0, s, A);
                return;
            }
        }
        // small batch: one wave per block so that the waves spread over all the CUs (the per-block limb table is
        // a few kB, staging it four times as often is noise next to idle CUs)
        if constexpr (jm::quad_block_waves<T, Tp>() > 1)
        {
            if ((long long)grid < 2LL * b->n_cus)
            {
                const unsigned grid1 = (unsigned)((A.B + 15) / 16);
                hipLaunchKernelGGL((jm::k_quad<T, Tp, 1>), dim3(grid1), dim3(64), 0, s, A);
                return;
            }
        }
        hipLaunchKernelGGL((jm::k_quad<T, Tp>), dim3(grid), dim3(nth), 0, s, A);
    }
    else { (void)b; (void)A; (void)s; }
}

// streams / events of the chunked split stepping, made on first use (non-blocking streams: no implicit ordering with the
// legacy default stream; the fork / join events order them with the caller's stream)
bool split_streams(jm_batch * b)
{
    if (b->split_streams_made) return true;
    if (hipEventCreateWithFlags(&b->split_fork, hipEventDisableTiming) != hipSuccess) return false;
    for (int c = 0; c < b->split_chunks; ++c)
        if (hipStreamCreateWithFlags(&b->split_stream[c], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&b->split_join[c], hipEventDisableTiming) != hipSuccess)
            return false;
    b->split_streams_made = true;
    return true;
}

// constraint contact model on the branch-parallel decomposition (jm_qcon.h)
template<class Tp> void launch_quad_con(jm_batch * b, jm::BatchArgs<double> & A, const jm::ConArgs<double> & C0, hipStream_t s)
{
    if constexpr (Tp::QUAD)
    {
        jm::QConArgs<double> C;
        C.flags = C0.flags; C.data = C0.data; C.ws = C0.ws; C.friction = C0.friction;
        C.kp = C0.kp; C.kd = C0.kd; C.kp_lock = C0.kp_lock; C.kd_lock = C0.kd_lock; C.torsion = C0.torsion; C.reg = C0.reg; C.tol_abs = C0.tol_abs; C.tol_rel = C0.tol_rel;
        C.iter_max = C0.iter_max;
        C.ground_h = A.ground_h; C.ground_nx = A.ground_nx; C.ground_ny = A.ground_ny;
        C.ground_x0 = A.ground_x0; C.ground_y0 = A.ground_y0; C.ground_dx = A.ground_dx; C.ground_dy = A.ground_dy;
        C.stage = nullptr; C.split_e = 0; C.split_pass = 0; C.split_r0 = 0; C.split_r1 = (int)A.B;
        constexpr int nth = 64 * jm::qcon_block_waves<double, Tp>();
        const unsigned grid = (unsigned)((A.B + nth / 4 - 1) / (nth / 4));
        if constexpr (jm::qcon_split<Tp>())
        {
            // Engine::start / reset of such robots: the four passes of the initialisation (engine.cc:1399-1467) as launches
            // of the split kernels -- first pass (every constraint enabled, hysteresis, free acceleration with u = 0, matrix
            // and right-hand side) | exact solve | three times (multipliers -> u, free acceleration, right-hand side |
            // Gauss-Seidel) | closing evaluation with the outputs.  The single kernel (k_quad_con: 2064 spilled VGPRs, 137 kB
            // of LDS for stage rows a start does not need, the general Gauss-Seidel form at one row per memory round trip)
            // took 52-67 ms per launch at B = 32 768 whatever the number of lanes that restart.
            if (b->qcon_split && (A.mode == jm::MODE_START || A.mode == jm::MODE_RESET) && !(A.model_lane || A.applied || A.ground_h) &&
                (A.B & 15) == 0 && !b->ov_flags && b->qcon_split_start)
            {
                C.stage = C.ws + (size_t)jm::qcon_split_region_rows<double, Tp>() * (size_t)A.B;
                const unsigned g64 = (unsigned)((A.B + 63) / 64);
                auto solve = [&]() {
                    hipLaunchKernelGGL((jm::k_qcon_pgs<double, Tp, 8, 0, JM_QCON_PGS_DEPTH>), dim3(g64), dim3(256), 0, s, C, A.P, (unsigned)A.B);
                    if constexpr (jm::QConRows<Tp>::MAXM > 64)
                        hipLaunchKernelGGL((jm::k_qcon_pgs<double, Tp, 12, 64, JM_QCON_PGS_DEPTH - 1>), dim3(g64), dim3(256), 0, s, C, A.P, (unsigned)A.B);
                    if constexpr (jm::QTip<Tp>::ON)
                        hipLaunchKernelGGL((jm::k_qtip_pgs<double, Tp>), dim3(g64), dim3(256), 0, s, C, A.P, (unsigned)A.B);
                };
                C.split_pass = 0;
                hipLaunchKernelGGL((jm::k_quad_con_split<double, Tp, 1, 1>), dim3(g64), dim3(256), 0, s, A, C);
                hipLaunchKernelGGL((jm::k_qcon_exact<double, Tp>), dim3(g64), dim3(256), 0, s, C);
                if constexpr (jm::QTip<Tp>::ON)
                    hipLaunchKernelGGL((jm::k_qtip_exact<double, Tp>), dim3((unsigned)((A.B + 255) / 256)), dim3(256), 0, s, C);
                for (int pass = 1; pass <= 3; ++pass)
                {
                    C.split_pass = pass;
                    hipLaunchKernelGGL((jm::k_quad_con_split<double, Tp, 1, 1>), dim3(g64), dim3(256), 0, s, A, C);
                    solve();
                }
                C.split_pass = 4;
                hipLaunchKernelGGL((jm::k_quad_con_split<double, Tp, 2, 1>), dim3(g64), dim3(256), 0, s, A, C);
                return;
            }
            // robots whose solves live in the workspace: step launches go through pre | solve | post per evaluation (jm_qcon.h)
            // robots with small solves (one lane per robot): only while every solve of the batch fits that form -- a robot
            // that does not (more active bounds than the layout holds, torsion rows, a joint lock) falls to the streamed form,
            // an order of magnitude slower per robot, and the split form only pays off for long solves: the counters of the solve
            // kernel decide (see jm_batch::lane_stat).
            bool lane_ok = true;
            hipStreamCaptureStatus cap0 = hipStreamCaptureStatusNone;
            const bool capturing = hipStreamIsCapturing(s, &cap0) == hipSuccess && cap0 != hipStreamCaptureStatusNone;
            if constexpr (!jm::qcon_split_large<Tp>())
            {
                if (A.mode == jm::MODE_START)
                {
                    // a simulation starts in the split form with a clean history: the forms of its steps depend on its data only
                    b->lane_step = 0;
                    b->split_cooldown = 0;
                    for (int i = 0; i < jm_batch::LANE_SLOTS; ++i) b->lane_step_of[i] = -1;
                }
                if (A.mode == jm::MODE_STEP && !capturing)
                {
                    for (int pass = 0; pass < jm_batch::LANE_SLOTS; ++pass)
                    {
                        // oldest outstanding slot of a step <= n - 2
                        int k = -1;
                        for (int i = 0; i < jm_batch::LANE_SLOTS; ++i)
                            if (b->lane_step_of[i] >= 0 && b->lane_step_of[i] <= b->lane_step - 2 && (k < 0 || b->lane_step_of[i] < b->lane_step_of[k])) k = i;
                        if (k < 0) break;
                        (void)hipEventSynchronize(b->lane_ev[k]);
                        b->lane_step_of[k] = -1;
                        const int32_t * st = b->lane_stat_host + 4 * k;   // [0] misfits, [1] sweeps, [2] waves, [3] longest solve
                        if (std::getenv("JM_DEBUG_SPLIT")) std::fprintf(stderr, "[split] miss %d sweeps %d waves %d longest %d\n", st[0], st[1], st[2], st[3]);
                        int cool = 0;
                        if (st[0] > 0) cool = 64;
                        // measured on ANYmal, 65 536 robots, per evaluation: single kernel ~147 us + 5.3 us per average sweep (its
                        // waves queue four deep on a SIMD); split form ~247 us + 1.55 us per sweep of the LONGEST solve of the launch
                        // (every wave of the solve kernel is resident at once)
                        else if (st[2] > 0 && 5.3 * (double)st[1] / (double)st[2] - 1.55 * (double)st[3] < 100.0) cool = 256;
                        if (cool > b->split_cooldown) b->split_cooldown = cool;
                    }
                    if (b->split_cooldown > 0) { --b->split_cooldown; lane_ok = false; }
                    ++b->lane_step;
                }
                // (a captured step keeps one form for all its replays: the single kernel, which is never far off)
                if (capturing && !std::getenv("JIMINY_AMD_QCON_SPLIT_CAPTURE")) lane_ok = false;
                if (C0.torsion >= 2.220446049250313e-16) lane_ok = false;   // (four-row contact blocks: never the fixed layout)
            }
            if (lane_ok && b->qcon_split && A.mode == jm::MODE_STEP && !(A.model_lane || A.applied || A.ground_h) && (A.B & 15) == 0 && !b->ov_flags)
            {
                int32_t * miss = nullptr;
                int lane_slot = -1;
                if constexpr (!jm::qcon_split_large<Tp>())
                    if (!capturing)
                    {
                        if (!b->lane_stat)
                        {
                            bool ok = hipMalloc((void **)&b->lane_stat, 4 * jm_batch::LANE_SLOTS * sizeof(int32_t)) == hipSuccess &&
                                      hipHostMalloc((void **)&b->lane_stat_host, 4 * jm_batch::LANE_SLOTS * sizeof(int32_t), hipHostMallocDefault) == hipSuccess;
                            for (int i = 0; ok && i < jm_batch::LANE_SLOTS; ++i) ok = hipEventCreateWithFlags(&b->lane_ev[i], hipEventDisableTiming) == hipSuccess;
                            if (!ok) b->lane_stat = nullptr;
                            else std::memset(b->lane_stat_host, 0, 4 * jm_batch::LANE_SLOTS * sizeof(int32_t));
                        }
                        if (b->lane_stat)
                        {
                            lane_slot = (int)((b->lane_step - 1) % jm_batch::LANE_SLOTS);   // (drained above: its step is <= n - 4)
                            miss = b->lane_stat + 4 * lane_slot;
                            (void)hipMemsetAsync(miss, 0, 4 * sizeof(int32_t), s);
                        }
                    }
                C.stage = C.ws + (size_t)jm::qcon_split_region_rows<double, Tp>() * (size_t)A.B;
                const int pre = A.command_changed ? 1 : 0;
                const int n_evals = pre + A.n_sub * (A.solver == JM_SOLVER_RUNGE_KUTTA_4 ? 4 : 1);
                // The solve launch lasts as long as its slowest robot (a few of 32 768 run into the iteration cap) while
                // most of the chip idles.  Option (JIMINY_AMD_QCON_SPLIT_CHUNKS = n > 1, off by default): the batch steps as n
                // independent chunks, each through its own chain of launches on a stream of its own, so that the tail of one
                // chunk could overlap the work of the others.  Measured on the MI355X (Atlas, B = 32 768): 9.2 ms per launch
                // with one chain, 8.8 with two chunks, 12.2 with four, 15.9 with eight -- the chains mostly serialise.
                // (Never while the caller's stream is being captured into a graph.)
                int n_chunks = 1;
                hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
                if (b->split_chunks > 1 && A.B >= 64LL * 8 * b->split_chunks && hipStreamIsCapturing(s, &cap) == hipSuccess &&
                    cap == hipStreamCaptureStatusNone && split_streams(b))
                    n_chunks = b->split_chunks;
                const long long per = (((A.B + n_chunks - 1) / n_chunks) + 63) / 64 * 64;
                if (n_chunks > 1)
                {
                    hipEventRecord(b->split_fork, s);
                    for (int c = 0; c < n_chunks; ++c) hipStreamWaitEvent(b->split_stream[c], b->split_fork, 0);
                }
                for (int c = 0; c < n_chunks; ++c)
                {
                    const hipStream_t sc = n_chunks > 1 ? b->split_stream[c] : s;
                    C.split_r0 = (int)(c * per);
                    C.split_r1 = (int)((c + 1) * per < A.B ? (c + 1) * per : A.B);
                    if (C.split_r1 <= C.split_r0) continue;
                    const unsigned g64 = (unsigned)((C.split_r1 - C.split_r0 + 63) / 64);
                    for (int e = 0; e < n_evals; ++e)
                    {
                        C.split_e = e;
                        hipLaunchKernelGGL((jm::k_quad_con_split<double, Tp, 1, 0>), dim3(g64), dim3(256), 0, sc, A, C);
                        // (robots whose system fits the fixed 16-row layout: one lane per robot, out of registers -- jm_qcon.h,
                        // qcon_pgs_lane; they are marked done for the streamed form that follows)
                        if constexpr (jm::QLanePgs<Tp>::FITS && JM_QCON_PGS_LANE)
                            hipLaunchKernelGGL((jm::k_qcon_pgs_lane<double, Tp>), dim3((unsigned)((C.split_r1 - C.split_r0 + 63) / 64)), dim3(64), 0, sc, C, A.P, miss);
                        // (solves of up to 64 rows, then the waves that hold a larger one)
                        hipLaunchKernelGGL((jm::k_qcon_pgs<double, Tp, 8, 0, JM_QCON_PGS_DEPTH>), dim3(g64), dim3(256), 0, sc, C, A.P, (unsigned)A.B);
                        if constexpr (jm::QConRows<Tp>::MAXM > 64)
                            hipLaunchKernelGGL((jm::k_qcon_pgs<double, Tp, 12, 64, JM_QCON_PGS_DEPTH - 1>), dim3(g64), dim3(256), 0, sc, C, A.P, (unsigned)A.B);
                        // (the waves whose robots all have few active joint rows: operational-space form, jm_qtip.h)
                        if constexpr (jm::QTip<Tp>::ON)
                            hipLaunchKernelGGL((jm::k_qtip_pgs<double, Tp>), dim3(g64), dim3(256), 0, sc, C, A.P, (unsigned)A.B);
                        hipLaunchKernelGGL((jm::k_quad_con_split<double, Tp, 2, 0>), dim3(g64), dim3(256), 0, sc, A, C);
                    }
                }
                if (n_chunks > 1)
                    for (int c = 0; c < n_chunks; ++c)
                    {
                        hipEventRecord(b->split_join[c], b->split_stream[c]);
                        hipStreamWaitEvent(s, b->split_join[c], 0);
                    }
                if (miss)
                {
                    (void)hipMemcpyAsync(b->lane_stat_host + 4 * lane_slot, miss, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s);
                    (void)hipEventRecord(b->lane_ev[lane_slot], s);
                    b->lane_step_of[lane_slot] = b->lane_step - 1;
                }
                return;
            }
        }
        // (user-registered JointConstraints: kernels built with them -- the variation kernel, or any kernel of a split topology)
        // (`start` / `reset` -- Engine::start's four passes with the exact solve -- are instantiations of their own)
        const bool init = A.mode == jm::MODE_START || A.mode == jm::MODE_RESET;
        if (A.model_lane || A.applied || A.ground_h || (b->joint_locks && !jm::qcon_split<Tp>()))
        {
            if (init) hipLaunchKernelGGL((jm::k_quad_con_gen<double, Tp, 1>), dim3(grid), dim3(nth), 0, s, A, C);
            else hipLaunchKernelGGL((jm::k_quad_con_gen<double, Tp, 0>), dim3(grid), dim3(nth), 0, s, A, C);
        }
        else if (init) hipLaunchKernelGGL((jm::k_quad_con<double, Tp, 1>), dim3(grid), dim3(nth), 0, s, A, C);
        else hipLaunchKernelGGL((jm::k_quad_con<double, Tp, 0>), dim3(grid), dim3(nth), 0, s, A, C);
    }
    else { (void)b; (void)A; (void)C0; (void)s; }
}

template<class T> int32_t launch(jm_batch * b, jm::BatchArgs<T> & A, void * stream)
{
    HIP_TRY(hipSetDevice(b->device));
    const hipStream_t s = (hipStream_t)stream;
    const unsigned grid = (unsigned)((A.B + 63) / 64);
    // only the step launches are timed: the roofline leg prices one pass of the hot path, not the
    // (cheaper, single-evaluation) start / reset / dynamics launches
    {
        // body parameters per lane and height maps: the variation form of the branch-parallel kernels only; friction per lane and
        // applied wrenches: that form (float64) or the one-robot-per-lane kernels, which read them as they are (ABI 9)
        const bool quad = Topo::QUAD && b->variant == VARIANT_QUAD;
        // (height-map ground on the one-robot-per-lane kernels: the spring-damper law of their variation instantiation)
        if (A.ground_h && !(std::is_same<T, double>::value &&
                            (quad || (!Topo::QUAD && b->copt.contact_model != JM_CONTACT_CONSTRAINT))))
            return fail(JM_ENOTIMPL, "a height-map ground needs a float64 batch; with contacts.model = 'constraint' also a "
                                     "branch-parallel topology (floating base with limb chains)");
        if (A.model_lane && !(std::is_same<T, double>::value && (quad || !Topo::QUAD)))
            return fail(JM_ENOTIMPL, "per-lane body parameters need a float64 batch (and, on a branch-parallel topology, its own kernels)");
        if (A.friction && quad && !std::is_same<T, double>::value)
            return fail(JM_ENOTIMPL, "per-lane friction on a branch-parallel topology needs a float64 batch");
        // (applied wrenches: an instantiation of their own in either family; the one-robot-per-lane one exists for the
        // topologies that have no branch-parallel kernels)
        if (A.applied && !(std::is_same<T, double>::value && (quad || !Topo::QUAD)))
            return fail(JM_ENOTIMPL, "applied wrenches need a float64 batch (and, on a branch-parallel topology, its own kernels)");
    }
    const bool timed = b->timing && A.mode == jm::MODE_STEP && b->n_timed < JM_TIMING_RING;
    if (timed) HIP_TRY(hipEventRecord(b->ev[2 * b->n_timed], s));
    if (b->copt.contact_model == JM_CONTACT_CONSTRAINT)
    {
        // float64 only: the reference's precision; its PGS tolerances (1e-5 absolute on residual
        // differences) are below float32 round-off of the delassus products
        if constexpr (std::is_same<T, double>::value)
        {
            using R = jm::ConRows<Topo>;
            if (R::NR > 0 && (!b->field[JM_F_CON_FLAGS] || !b->field[JM_F_CON_DATA] || !b->field[JM_F_WORKSPACE]))
                return fail(JM_ECONTROLFLOW, "contacts.model = 'constraint': the con_flags, con_data and workspace fields must be bound");
            jm::ConArgs<T> C;
            C.flags = b->ov_flags ? b->ov_flags : (int32_t *)b->field[JM_F_CON_FLAGS];
            C.data = b->ov_data ? (T *)b->ov_data : (T *)b->field[JM_F_CON_DATA];
            C.ws = b->ov_ws ? (T *)b->ov_ws : (T *)b->field[JM_F_WORKSPACE];
            // per-lane friction: bound field (compact batches of the adaptive stepper read it through BatchArgs::lane_map)
            C.friction = (const T *)b->field[JM_F_FRICTION];
            const double omega = 2.0 * 3.14159265358979323846 * b->copt.stabilization_freq;  // abstract_constraint.cc:88-98
            C.kp = (T)(omega * omega);
            C.kd = (T)(2.0 * omega);
            // user-registered constraints: gains of their own when the option says so (jm_constraint_options, ABI 6)
            const double omega_u = 2.0 * 3.14159265358979323846 * b->copt.user_stabilization_freq;
            C.kp_lock = b->copt.user_stabilization_freq < 0.0 ? C.kp : (T)(omega_u * omega_u);
            C.kd_lock = b->copt.user_stabilization_freq < 0.0 ? C.kd : (T)(2.0 * omega_u);
            C.torsion = (T)b->copt.torsion; C.reg = (T)b->copt.regularization;
            C.tol_abs = (T)b->copt.tol_abs; C.tol_rel = (T)b->copt.tol_rel;
            C.iter_max = b->copt.pgs_iter_max;
            C.xl = nullptr; C.xstride = 0;  // set by the kernel (LDS)
            C.yl = nullptr; C.ystride = 0; C.yrows = 0;
            C.park = nullptr; C.park_rows = 0;
            if (Topo::QUAD && b->variant == VARIANT_QUAD && R::NR > 0) launch_quad_con<Topo>(b, A, C, s);
            else
            {
                bool done = false;
                if constexpr (!Topo::QUAD)
                    if (A.applied || A.model_lane) { hipLaunchKernelGGL((jm::k_constrained<T, Topo, true>), dim3(grid), dim3(64), 0, s, A, C); done = true; }
                if (!done) hipLaunchKernelGGL((jm::k_constrained<T, Topo, false>), dim3(grid), dim3(64), 0, s, A, C);
            }
        }
        else return fail(JM_ENOTIMPL, "contacts.model = 'constraint' needs a float64 batch");
    }
    else if (b->variant == VARIANT_QUAD) launch_quad<T, Topo>(b, A, s);
    else
    {
        bool done = false;
        if constexpr (!Topo::QUAD && std::is_same<T, double>::value)
            if (A.applied || A.model_lane || A.ground_h) { hipLaunchKernelGGL((jm::k_batch<T, Topo, true>), dim3(grid), dim3(64), 0, s, A); done = true; }
        if (!done) hipLaunchKernelGGL((jm::k_batch<T, Topo, false>), dim3(grid), dim3(64), 0, s, A);
    }
    HIP_TRY(hipGetLastError());
    if (timed)
    {
        HIP_TRY(hipEventRecord(b->ev[2 * b->n_timed + 1], s));
        ++b->n_timed;
    }
    return JM_OK;
}

int32_t check_bound(const jm_batch * b, bool need_command)
{
    if (!b->field[JM_F_Q] || !b->field[JM_F_V] || !b->field[JM_F_A])
        return fail(JM_ECONTROLFLOW, "state fields q, v, a must be bound before this call");
    if (need_command && Topo::NM > 0 && !b->field[JM_F_COMMAND])
        return fail(JM_ECONTROLFLOW, "the command field must be bound before this call");
    return JM_OK;
}
}  // namespace

namespace
{
template<class T> int32_t step_adaptive(jm_batch * b, double t_next, const jm_adaptive_options * o, int32_t new_step,
                                               int32_t command_changed, int32_t update_sensors, int32_t max_attempts,
                                               int32_t * attempts_out, void * stream)
{
    const hipStream_t s = (hipStream_t)stream;
    using R = jm::AdaptiveRows<Topo>;
    T * ws = (T *)b->ad_ws;
    const long long B = b->B;
    jm::AdaptiveArgs<T> D;
    D.P = (const T *)b->d_params;
    D.q = (T *)b->field[JM_F_Q]; D.v = (T *)b->field[JM_F_V]; D.a = (T *)b->field[JM_F_A];
    D.ws = ws; D.fs = b->ad_fs; D.is = b->ad_is; D.status = (int32_t *)b->field[JM_F_STATUS];
    D.n_active = b->ad_count; D.B = B;
    D.t_next = t_next; D.tol_rel = o->tol_rel; D.tol_abs = o->tol_abs; D.dt_max = o->dt_max;
    D.dt_restore_threshold_rel = o->dt_restore_threshold_rel; D.succ_failed_max = o->successive_iter_failed_max;
    D.new_step = new_step; D.stage = 0;
    if (D.status && new_step) HIP_TRY(hipMemsetAsync(D.status, 0, sizeof(int32_t) * B, s));
    // FSAL fix when the command changed at the breakpoint: a(t+) (engine.cc:2030-2042)
    if (command_changed)
    {
        auto A = make_args<T>(b);
        A.mode = jm::MODE_DYNAMICS; A.q_in = D.q; A.v_in = D.v; A.a_out = D.a;
        const int32_t rc = launch<T>(b, A, stream);
        if (rc != JM_OK) return rc;
    }
    D.command = (const T *)b->field[JM_F_COMMAND];
    D.n_act = B;  // upper bound of the active-list length: the list only shrinks within an interval
    const bool constrained = b->copt.contact_model == JM_CONTACT_CONSTRAINT && jm::ConRows<Topo>::NR > 0;
    D.con_flags = constrained ? (int32_t *)b->field[JM_F_CON_FLAGS] : nullptr;
    D.con_data = constrained ? (T *)b->field[JM_F_CON_DATA] : nullptr;
    D.con_flags_c = b->ad_flags;
    if (constrained && (!D.con_flags || !D.con_data || !b->ad_flags))
        return fail(JM_ECONTROLFLOW, "contacts.model = 'constraint': bind con_flags / con_data, then jm_batch_bind_adaptive");
    // branch-parallel topologies, spring-damper contacts, float64: ONE persistent launch per interval, every quad
    // runs its robot's whole adaptive loop on the chip (jm_qdopri.h); the host only learns whether a robot ran
    // into the attempt bound of a launch (then it launches again) and the largest attempt count
    if constexpr (Topo::QUAD && std::is_same<T, double>::value)
    {
        if (b->variant == VARIANT_QUAD && !constrained && o->form != 1)
        {
            // (per-lane friction alone also needs the variation kernel: only its contact law reads A.friction)
            const bool gen = b->field[JM_F_MODEL_LANE] || b->ground_h || (b->applied_k > 0 && b->field[JM_F_APPLIED]) ||
                             b->field[JM_F_FRICTION];
            auto A = make_args<T>(b);
            A.mode = jm::MODE_DYNAMICS;
            constexpr int nth = 64 * jm::qdopri_block_waves<T, Topo>();
            const unsigned grid = (unsigned)((B + nth / 4 - 1) / (nth / 4));
            const int per_launch = 4096;
            int total = 0;
            for (;;)
            {
                HIP_TRY(hipMemsetAsync(b->ad_count, 0, 2 * sizeof(int32_t), s));
                if (gen) hipLaunchKernelGGL((jm::k_quad_dopri_gen<T, Topo>), dim3(grid), dim3(nth), 0, s, A, D, per_launch);
                else hipLaunchKernelGGL((jm::k_quad_dopri<T, Topo>), dim3(grid), dim3(nth), 0, s, A, D, per_launch);
                HIP_TRY(hipGetLastError());
                D.new_step = 0;
                HIP_TRY(hipMemcpyAsync(b->ad_count_host, b->ad_count, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                total += b->ad_count_host[1];
                if (b->ad_count_host[0] == 0) break;
                if (total >= max_attempts) return fail(JM_ERUNTIME, "adaptive stepper: too many attempts for one breakpoint interval");
            }
            if (attempts_out) *attempts_out = total;
            auto Ar = make_args<T>(b);
            Ar.mode = jm::MODE_REFRESH; Ar.update_sensors = update_sensors;
            return launch<T>(b, Ar, stream);
        }
    }
    const unsigned g256 = (unsigned)((B + 255) / 256);
    int attempts = 0;
    for (;;)
    {
        // all lanes: step-size selection + the list of the lanes that still have to move
        HIP_TRY(hipMemsetAsync(b->ad_count, 0, sizeof(int32_t), s));
        hipLaunchKernelGGL((jm::k_dopri_prepare<T, Topo>), dim3(g256), dim3(256), 0, s, D);
        D.new_step = 0;
        HIP_TRY(hipMemcpyAsync(b->ad_count_host, b->ad_count, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        const long long n = *b->ad_count_host;
        if (n == 0) break;
        if (attempts >= max_attempts) return fail(JM_ERUNTIME, "adaptive stepper: too many attempts for one breakpoint interval");
        // The n active lanes only: compact workspace rows [rows][n], dense batch of n robots for the dynamics
        // kernels.  Short lists are launch-bound: several attempts are then issued per synchronisation, the
        // grids and row strides stay sized for n while the kernels read the current length on the device.
        D.n_act = n;
        const int burst = n > 16384 ? 1 : (n > 2048 ? 3 : 8);
        const unsigned g128 = (unsigned)((n + 127) / 128);
        for (int k = 0; k < burst; ++k, ++attempts)
        {
            if (k > 0)
            {
                HIP_TRY(hipMemsetAsync(b->ad_count, 0, sizeof(int32_t), s));
                hipLaunchKernelGGL((jm::k_dopri_prepare<T, Topo>), dim3(g256), dim3(256), 0, s, D);
            }
            for (int i = 1; i <= 6; ++i)
            {
                D.stage = i;
                hipLaunchKernelGGL((jm::k_dopri_stage<T, Topo>), dim3(g128), dim3(128), 0, s, D);
                auto A = make_args<T>(b);
                A.mode = jm::MODE_DYNAMICS;
                A.B = n;
                A.command = ws + (long long)R::CMD * n;
                A.q_in = ws + (long long)R::QS * n;
                A.v_in = ws + (long long)(R::KV + (i - 1) * Topo::NV) * n;
                A.a_out = ws + (long long)(R::KA + (i - 1) * Topo::NV) * n;
                // (per-lane optional inputs -- body parameters, friction, applied wrenches, terrain patch -- stay in batch order)
                A.lane_map = b->ad_is + (long long)jm::AD_MAP * b->B;
                A.B_full = b->B;
                if (constrained)
                {
                    b->ov_flags = b->ad_flags;
                    b->ov_data = ws + (long long)R::CDATA * n;
                    b->ov_ws = ws + (long long)R::CWS * n;
                }
                const int32_t rc = launch<T>(b, A, stream);
                b->ov_flags = nullptr; b->ov_data = nullptr; b->ov_ws = nullptr;
                if (rc != JM_OK) return rc;
            }
            hipLaunchKernelGGL((jm::k_dopri_finish<T, Topo>), dim3(g128), dim3(128), 0, s, D);
            HIP_TRY(hipGetLastError());
        }
    }
    if (attempts_out) *attempts_out = attempts;
    // extra terms + sensors at the breakpoint (engine.cc:2148, 2386-2410)
    auto A = make_args<T>(b);
    A.mode = jm::MODE_REFRESH; A.update_sensors = update_sensors;
    return launch<T>(b, A, stream);
}
}  // namespace

extern "C"
{
const char * jm_topology_signature(void) { return Topo::signature; }
int32_t jm_abi_version(void) { return JM_ABI_VERSION; }

int32_t jm_model_create(const jm_model_desc * desc, jm_model ** out)
{
    if (!desc || !out) return fail(JM_EINVAL, "jm_model_create: null argument");
    std::string why;
    if (!jm::check_topology<Topo>(*desc, why)) return fail(JM_ETOPOLOGY, why);
    jm_model * m = new (std::nothrow) jm_model();
    if (!m) return fail(JM_ERUNTIME, "out of host memory");
    m->params = jm::pack_model<Topo>(*desc);
    jm::pack_options<Topo>(m->params, jm::default_options());
    jm::pack_quad<Topo>(m->params, *desc);
    if (desc->njoints > 1)
    {
        static const double I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int k = 0; k < 9; ++k) m->root_at_origin &= (desc->placement_R[9 + k] == I9[k]);
        for (int k = 0; k < 3; ++k) m->root_at_origin &= (desc->placement_p[3 + k] == 0.0);
    }
    *out = m;
    return JM_OK;
}
int32_t jm_model_destroy(jm_model * model)
{
    delete model;
    return JM_OK;
}

int32_t jm_batch_create(const jm_model * model, int64_t batch_size, int32_t dtype, int32_t device, jm_batch ** out)
{
    if (!model || !out) return fail(JM_EINVAL, "jm_batch_create: null argument");
    if (batch_size <= 0) return fail(JM_EINVAL, "batch size must be positive");
    // kernels address every field with unsigned 32-bit element offsets (row * B + lane)
    {
        const long long max_rows = 6LL * (Topo::NJ > Topo::NC ? Topo::NJ : Topo::NC) + Topo::NQ + 16;
        if (batch_size * max_rows >= (1LL << 29))
            return fail(JM_EINVAL, "batch too large for one jm_batch (32-bit field offsets): shard it over several batches");
    }
    if (dtype != JM_F64 && dtype != JM_F32) return fail(JM_EINVAL, "dtype must be JM_F64 or JM_F32");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(JM_EINVAL, "invalid HIP device ordinal");
    jm_batch * b = new (std::nothrow) jm_batch();
    if (!b) return fail(JM_ERUNTIME, "out of host memory");
    b->model = model;
    b->B = batch_size;
    b->dtype = dtype;
    b->device = device;
    b->params = model->params;
    // kernel variant: limb-parallel when the topology allows it; JM_KERNEL_VARIANT=lane forces the
    // generic one-robot-per-lane kernel (A/B measurements)
    b->variant = (Topo::QUAD && model->root_at_origin) ? VARIANT_QUAD : VARIANT_LANE;
    // (`start` / `reset` of robots with small solves: the single kernel -- its passes run on chip; jm_qcon.h, k_quad_con<1>)
    if constexpr (Topo::QUAD) b->qcon_split_start = jm::qcon_split_large<Topo>();
    if (const char * e = std::getenv("JIMINY_AMD_QCON_SPLIT")) b->qcon_split = e[0] != '0';
    if (const char * e = std::getenv("JIMINY_AMD_QCON_SPLIT_START")) b->qcon_split_start = e[0] != '0';
    if (const char * e = std::getenv("JIMINY_AMD_QCON_SPLIT_CHUNKS"))
    {
        const int n = std::atoi(e);
        b->split_chunks = n < 1 ? 1 : (n > 8 ? 8 : n);
    }
    if (const char * v = std::getenv("JM_KERNEL_VARIANT"))
        if (std::string(v) == "lane") b->variant = VARIANT_LANE;
    hipError_t e = hipSetDevice(device);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) b->n_cus = cus;
    }
    if (e == hipSuccess) e = hipMalloc(&b->d_params, b->params.size() * sizeof(double));
    if (e != hipSuccess)
    {
        delete b;
        return fail(JM_ERUNTIME, std::string("jm_batch_create: ") + hipGetErrorString(e));
    }
    const int32_t rc = upload_params(b);
    if (rc != JM_OK)
    {
        jm_batch_destroy(b);
        return rc;
    }
    *out = b;
    return JM_OK;
}
int32_t jm_batch_destroy(jm_batch * b)
{
    if (!b) return JM_OK;
    (void)hipSetDevice(b->device);
    if (b->d_params) (void)hipFree(b->d_params);
    if (b->ad_count) (void)hipFree(b->ad_count);
    if (b->lane_stat) (void)hipFree(b->lane_stat);
    if (b->lane_stat_host) (void)hipHostFree(b->lane_stat_host);
    for (int i = 0; i < jm_batch::LANE_SLOTS; ++i) if (b->lane_ev[i]) (void)hipEventDestroy(b->lane_ev[i]);
    if (b->ad_flags) (void)hipFree(b->ad_flags);
    if (b->ad_count_host) (void)hipHostFree(b->ad_count_host);
    for (hipEvent_t e : b->ev) (void)hipEventDestroy(e);
    if (b->split_fork) (void)hipEventDestroy(b->split_fork);
    for (hipEvent_t e : b->split_join) if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : b->split_stream) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    delete b;
    return JM_OK;
}
int32_t jm_batch_set_joint_locks(jm_batch * b, int32_t on)
{
    if (!b) return fail(JM_EINVAL, "jm_batch_set_joint_locks: null batch");
    // (both kernel families read bit 2 of the constraint flags; the branch-parallel one also selects its general sweep form by this switch)
    b->joint_locks = on != 0;
    return JM_OK;
}
int32_t jm_batch_set_options(jm_batch * b, const jm_options * o)
{
    if (!b || !o) return fail(JM_EINVAL, "jm_batch_set_options: null argument");
    if (b->started)
        return fail(JM_ECONTROLFLOW, "options cannot be changed while a simulation is running");  // engine.cc:2656-2662
    if (!(o->contact_stiffness >= 0.0) || !(o->contact_damping >= 0.0) || !(o->contact_friction >= 0.0))
        return fail(JM_EINVAL, "contact stiffness, damping and friction must be non-negative");
    if (!(o->contact_transition_velocity > 0.0)) return fail(JM_EINVAL, "contacts.transitionVelocity must be positive");
    if (o->contact_transition_eps < 0.0) return fail(JM_EINVAL, "contacts.transitionEps must be non-negative");  // engine.cc:2697-2702
    jm::pack_options<Topo>(b->params, *o);
    return upload_params(b);
}
int32_t jm_batch_workspace_rows(const jm_batch * b)
{
    return (b && b->copt.contact_model == JM_CONTACT_CONSTRAINT) ? constraint_ws_rows(b) : 0;
}
int32_t jm_batch_set_constraint_options(jm_batch * b, const jm_constraint_options * o)
{
    if (!b || !o) return fail(JM_EINVAL, "jm_batch_set_constraint_options: null argument");
    if (b->started)
        return fail(JM_ECONTROLFLOW, "options cannot be changed while a simulation is running");  // engine.cc:2656-2662
    if (o->contact_model != JM_CONTACT_SPRING_DAMPER && o->contact_model != JM_CONTACT_CONSTRAINT)
        return fail(JM_EINVAL, "The requested contact model is not available.");  // engine.cc:2741-2747
    if (!(o->regularization >= 0.0)) return fail(JM_EINVAL, "Constraint option 'regularization' must be positive.");
    if (!(o->stabilization_freq >= 0.0)) return fail(JM_EINVAL, "Contact option 'stabilizationFreq' must be positive.");
    if (!(o->torsion >= 0.0)) return fail(JM_EINVAL, "contacts.torsion must be non-negative");
    if (!(o->tol_abs > 0.0) || !(o->tol_rel > 0.0)) return fail(JM_EINVAL, "tolAbs and tolRel must be positive");
    if (o->pgs_iter_max <= 50) return fail(JM_EINVAL, "pgs_iter_max must exceed 50 (relaxation schedule of the PGS solver)");
    // one-robot-per-lane kernel with 64-bit row offsets into the workspace; the batch must still fit
    b->copt = *o;
    return JM_OK;
}
int32_t jm_batch_constraint_rows(const jm_batch * b, int32_t * n_flag_rows, int32_t * n_data_rows, int32_t * n_workspace_rows)
{
    if (!b) return fail(JM_EINVAL, "jm_batch_constraint_rows: null batch");
    using R = jm::ConRows<Topo>;
    if (n_flag_rows) *n_flag_rows = R::NF;
    if (n_data_rows) *n_data_rows = R::ND;
    if (n_workspace_rows) *n_workspace_rows = constraint_ws_rows(b);
    return JM_OK;
}
int32_t jm_batch_set_ground(jm_batch * b, const void * heights, int32_t nx, int32_t ny, double x0, double y0, double dx, double dy)
{
    if (!b) return fail(JM_EINVAL, "jm_batch_set_ground: null batch");
    if (heights && (nx < 2 || ny < 2 || !(dx > 0.0) || !(dy > 0.0)))
        return fail(JM_EINVAL, "jm_batch_set_ground: the height map needs at least 2 x 2 samples and positive spacings");
    b->ground_h = heights; b->ground_nx = nx; b->ground_ny = ny;
    b->ground_x0 = x0; b->ground_y0 = y0; b->ground_dx = dx; b->ground_dy = dy;
    return JM_OK;
}
int32_t jm_batch_set_applied_frames(jm_batch * b, int32_t k, const double * offsets, const int32_t * joints)
{
    if (!b) return fail(JM_EINVAL, "jm_batch_set_applied_frames: null batch");
    if (k < 0 || k > 4 || (k > 0 && !offsets)) return fail(JM_EINVAL, "jm_batch_set_applied_frames: 0 <= K <= 4 frames with their offsets");
    for (int i = 0; i < k; ++i)
        if (joints && (joints[i] < 1 || joints[i] >= Topo::NJ)) return fail(JM_EINVAL, "jm_batch_set_applied_frames: parent joint out of range");
    b->applied_k = k;
    for (int i = 0; i < 3 * k; ++i) b->applied_p[i] = offsets[i];
    for (int i = 0; i < 4; ++i) b->applied_joint[i] = (joints && i < k) ? joints[i] : 1;
    return JM_OK;
}
int32_t jm_batch_bind(jm_batch * b, int32_t field, void * ptr)
{
    if (!b) return fail(JM_EINVAL, "jm_batch_bind: null batch");
    if (field < 0 || field >= JM_F_COUNT) return fail(JM_ELOOKUP, "jm_batch_bind: unknown field id");
    b->field[field] = ptr;
    return JM_OK;
}

int32_t jm_batch_start(jm_batch * b, void * stream)
{
    if (!b) return fail(JM_EINVAL, "jm_batch_start: null batch");
    int32_t rc = check_bound(b, true);
    if (rc != JM_OK) return rc;
    if (b->dtype == JM_F64)
    {
        auto A = make_args<double>(b);
        A.mode = jm::MODE_START;
        rc = launch<double>(b, A, stream);
    }
    else
    {
        auto A = make_args<float>(b);
        A.mode = jm::MODE_START;
        rc = launch<float>(b, A, stream);
    }
    if (rc == JM_OK) b->started = true;
    return rc;
}

int32_t jm_batch_stop(jm_batch * b)
{
    if (!b) return fail(JM_EINVAL, "jm_batch_stop: null batch");
    b->started = false;
    return JM_OK;
}

int32_t jm_batch_step(jm_batch * b, int32_t solver, double dt, int32_t n_substeps, int32_t command_changed,
                      int32_t update_sensors, void * stream)
{
    if (!b) return fail(JM_EINVAL, "jm_batch_step: null batch");
    if (!b->started)
        return fail(JM_ECONTROLFLOW, "No simulation running. Please start one before using step method.");  // engine.cc:1727-1731
    if (solver != JM_SOLVER_EULER_EXPLICIT && solver != JM_SOLVER_RUNGE_KUTTA_4)
        return fail(JM_ENOTIMPL, "only 'euler_explicit' and 'runge_kutta_4' are available on the batched path");
    // (`dt` is the step of the integrator -- the inner loop of Engine::step, which goes down to STEPPER_MIN_TIMESTEP when what
    // is left of a breakpoint interval is that short, engine.cc:2047-2089 --, not the user-level step size, whose bound
    // SIMULATION_MIN_TIMESTEP <= step (engine.cc:1750-1753) is checked where the schedule is planned)
    if (!(dt >= 1e-10) || !(dt <= 0.02 + 1e-12))
        return fail(JM_EINVAL, "Step size out of bounds.");  // constants.h:18-20
    if (n_substeps < 1) return fail(JM_EINVAL, "n_substeps must be >= 1");
    int32_t rc = check_bound(b, true);
    if (rc != JM_OK) return rc;
    if (b->dtype == JM_F64)
    {
        auto A = make_args<double>(b);
        A.mode = jm::MODE_STEP; A.solver = solver; A.dt = dt; A.n_sub = n_substeps;
        A.command_changed = command_changed; A.update_sensors = update_sensors;
        return launch<double>(b, A, stream);
    }
    auto A = make_args<float>(b);
    A.mode = jm::MODE_STEP; A.solver = solver; A.dt = (float)dt; A.n_sub = n_substeps;
    A.command_changed = command_changed; A.update_sensors = update_sensors;
    return launch<float>(b, A, stream);
}

// ---- adaptive Dormand-Prince stepping (jm_adaptive.h)
int32_t jm_batch_adaptive_workspace_rows(const jm_batch * b)
{
    return (b && b->copt.contact_model == JM_CONTACT_CONSTRAINT) ? jm::AdaptiveRows<Topo>::CWS + constraint_ws_rows(b)
                                                                 : jm::AdaptiveRows<Topo>::TOTAL;
}
int32_t jm_batch_bind_adaptive(jm_batch * b, void * workspace, double * state_f64, int32_t * state_i32)
{
    if (!b) return fail(JM_EINVAL, "jm_batch_bind_adaptive: null batch");
    b->ad_ws = workspace; b->ad_fs = state_f64; b->ad_is = state_i32;
    if (!b->ad_count)
    {
        HIP_TRY(hipSetDevice(b->device));
        HIP_TRY(hipMalloc((void **)&b->ad_count, 2 * sizeof(int32_t)));
        if (jm::ConRows<Topo>::NF > 0)
            HIP_TRY(hipMalloc((void **)&b->ad_flags, sizeof(int32_t) * (size_t)jm::ConRows<Topo>::NF * (size_t)b->B));
        HIP_TRY(hipHostMalloc((void **)&b->ad_count_host, 2 * sizeof(int32_t), hipHostMallocDefault));
    }
    return JM_OK;
}
int32_t jm_batch_step_adaptive(jm_batch * b, double t_next, const jm_adaptive_options * options, int32_t new_step,
                               int32_t command_changed, int32_t update_sensors, int32_t max_attempts,
                               int32_t * attempts_out, void * stream)
{
    if (!b || !options) return fail(JM_EINVAL, "jm_batch_step_adaptive: null argument");
    if (!b->started)
        return fail(JM_ECONTROLFLOW, "No simulation running. Please start one before using step method.");
    if (!b->ad_ws || !b->ad_fs || !b->ad_is)
        return fail(JM_ECONTROLFLOW, "jm_batch_bind_adaptive must be called before the adaptive stepper is used");
    if (b->copt.contact_model == JM_CONTACT_CONSTRAINT && b->dtype != JM_F64)
        return fail(JM_ENOTIMPL, "contacts.model = 'constraint' needs a float64 batch");
    if (!(options->tol_rel > 0.0) || !(options->tol_abs > 0.0)) return fail(JM_EINVAL, "tolRel and tolAbs must be positive");
    if (!(options->dt_max >= 1e-6) || !(options->dt_max <= 0.02 + 1e-12)) return fail(JM_EINVAL, "'dtMax' option is out of range.");
    int32_t rc = check_bound(b, true);
    if (rc != JM_OK) return rc;
    HIP_TRY(hipSetDevice(b->device));
    if (b->dtype == JM_F64)
        return step_adaptive<double>(b, t_next, options, new_step, command_changed, update_sensors, max_attempts, attempts_out, stream);
    return step_adaptive<float>(b, t_next, options, new_step, command_changed, update_sensors, max_attempts, attempts_out, stream);
}

int32_t jm_batch_dynamics(jm_batch * b, const void * q_in, const void * v_in, void * a_out, void * stream)
{
    if (!b || !q_in || !v_in || !a_out) return fail(JM_EINVAL, "jm_batch_dynamics: null argument");
    if (!b->started)
        return fail(JM_ECONTROLFLOW, "No simulation running. Please start one before calling this method.");  // engine.cc:3594-3599
    if (Topo::NM > 0 && !b->field[JM_F_COMMAND]) return fail(JM_ECONTROLFLOW, "the command field must be bound");
    if (b->dtype == JM_F64)
    {
        auto A = make_args<double>(b);
        A.mode = jm::MODE_DYNAMICS; A.q_in = (const double *)q_in; A.v_in = (const double *)v_in; A.a_out = (double *)a_out;
        return launch<double>(b, A, stream);
    }
    auto A = make_args<float>(b);
    A.mode = jm::MODE_DYNAMICS; A.q_in = (const float *)q_in; A.v_in = (const float *)v_in; A.a_out = (float *)a_out;
    return launch<float>(b, A, stream);
}

int32_t jm_batch_reset_lanes(jm_batch * b, const uint8_t * lane_mask, const void * q_init, const void * v_init, void * stream)
{
    if (!b || !lane_mask || !q_init || !v_init) return fail(JM_EINVAL, "jm_batch_reset_lanes: null argument");
    if (!b->started) return fail(JM_ECONTROLFLOW, "No simulation running. Please start one before resetting lanes.");
    int32_t rc = check_bound(b, true);
    if (rc != JM_OK) return rc;
    if (b->dtype == JM_F64)
    {
        auto A = make_args<double>(b);
        A.mode = jm::MODE_RESET; A.mask = lane_mask; A.q_init = (const double *)q_init; A.v_init = (const double *)v_init;
        return launch<double>(b, A, stream);
    }
    auto A = make_args<float>(b);
    A.mode = jm::MODE_RESET; A.mask = lane_mask; A.q_init = (const float *)q_init; A.v_init = (const float *)v_init;
    return launch<float>(b, A, stream);
}

int32_t jm_batch_enable_timing(jm_batch * b, int32_t enable)
{
    if (!b) return fail(JM_EINVAL, "jm_batch_enable_timing: null batch");
    HIP_TRY(hipSetDevice(b->device));
    if (enable && b->ev.empty())
    {
        b->ev.resize(2 * JM_TIMING_RING, nullptr);
        for (hipEvent_t & e : b->ev) HIP_TRY(hipEventCreate(&e));
    }
    b->timing = enable != 0;
    b->n_timed = 0;
    return JM_OK;
}
int32_t jm_batch_timing_summary(jm_batch * b, int32_t * n_launches, double * total_ms)
{
    if (!b || !n_launches || !total_ms) return fail(JM_EINVAL, "jm_batch_timing_summary: null argument");
    double sum = 0.0;
    for (size_t i = 0; i < b->n_timed; ++i)
    {
        HIP_TRY(hipEventSynchronize(b->ev[2 * i + 1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, b->ev[2 * i], b->ev[2 * i + 1]));
        sum += ms;
    }
    *n_launches = (int32_t)b->n_timed;
    *total_ms = sum;
    b->n_timed = 0;
    return JM_OK;
}

int32_t jm_block_pd_controller(int32_t dtype, int64_t B, int32_t M, const void * encoder,
                               const int32_t * encoder_index, void * command_state, const double * lower,
                               const double * upper, const double * kp, const double * kd,
                               const double * effort_limit, double control_dt, void * out_torque, void * stream)
{
    if (!encoder || !encoder_index || !command_state || !lower || !upper || !kp || !kd || !effort_limit || !out_torque)
        return fail(JM_EINVAL, "jm_block_pd_controller: null argument");
    if (B <= 0 || M <= 0 || M > JM_BLOCK_MAX_MOTORS) return fail(JM_EINVAL, "jm_block_pd_controller: bad sizes");
    if (control_dt < 0.0) return fail(JM_EINVAL, "Integration backward in time is not supported.");
    if (dtype != JM_F64 && dtype != JM_F32) return fail(JM_EINVAL, "jm_block_pd_controller: bad dtype");
    jm::PdParams p;
    std::memset(&p, 0, sizeof(p));
    p.M = M;
    p.dt = control_dt;
    for (int m = 0; m < M; ++m)
    {
        p.enc_index[m] = encoder_index[m];
        for (int k = 0; k < 3; ++k) { p.lo[k][m] = lower[k * M + m]; p.hi[k][m] = upper[k * M + m]; }
        p.kp[m] = kp[m]; p.kd[m] = kd[m]; p.effort_limit[m] = effort_limit[m];
    }
    const unsigned grid = (unsigned)((B + 255) / 256);
    const hipStream_t s = (hipStream_t)stream;
    if (dtype == JM_F64)
        hipLaunchKernelGGL((jm::k_pd_controller<double>), dim3(grid), dim3(256), 0, s, p, (const double *)encoder,
                           (double *)command_state, (double *)out_torque, (long long)B);
    else
        hipLaunchKernelGGL((jm::k_pd_controller<float>), dim3(grid), dim3(256), 0, s, p, (const float *)encoder,
                           (float *)command_state, (float *)out_torque, (long long)B);
    HIP_TRY(hipGetLastError());
    return JM_OK;
}

int32_t jm_block_mahony_filter(int32_t dtype, int64_t B, int32_t n_imu, const void * imu, void * quat, void * omega,
                               void * cf, void * bias, double kp, double ki, double dt, void * stream)
{
    if (!imu || !quat || !omega || !cf || !bias) return fail(JM_EINVAL, "jm_block_mahony_filter: null argument");
    if (B <= 0 || n_imu <= 0) return fail(JM_EINVAL, "jm_block_mahony_filter: bad sizes");
    if (dtype != JM_F64 && dtype != JM_F32) return fail(JM_EINVAL, "jm_block_mahony_filter: bad dtype");
    const unsigned grid = (unsigned)((B + 255) / 256);
    const hipStream_t s = (hipStream_t)stream;
    if (dtype == JM_F64)
        hipLaunchKernelGGL((jm::k_mahony<double>), dim3(grid), dim3(256), 0, s, n_imu, (const double *)imu, (double *)quat,
                           (double *)omega, (double *)cf, (double *)bias, kp, ki, dt, (long long)B);
    else
        hipLaunchKernelGGL((jm::k_mahony<float>), dim3(grid), dim3(256), 0, s, n_imu, (const float *)imu, (float *)quat,
                           (float *)omega, (float *)cf, (float *)bias, kp, ki, dt, (long long)B);
    HIP_TRY(hipGetLastError());
    return JM_OK;
}

int32_t jm_block_pd_adapter(int32_t dtype, int64_t B, int32_t M, const void * action, int32_t order, void * command_state,
                            const double * lower, const double * upper, int32_t is_instantaneous, const double * velocity_deadband,
                            double step_dt, void * out, void * stream)
{
    if (!action || !command_state || !lower || !upper || !out) return fail(JM_EINVAL, "jm_block_pd_adapter: null argument");
    if (B <= 0 || M <= 0 || M > JM_BLOCK_MAX_MOTORS) return fail(JM_EINVAL, "jm_block_pd_adapter: bad sizes");
    if (order != 0 && order != 1) return fail(JM_EINVAL, "Derivative order of the target must be either 0 or 1.");
    if (dtype != JM_F64 && dtype != JM_F32) return fail(JM_EINVAL, "jm_block_pd_adapter: bad dtype");
    jm::PdAdapterParams p;
    std::memset(&p, 0, sizeof(p));
    p.M = M; p.order = order; p.instantaneous = is_instantaneous != 0; p.dt = step_dt;
    for (int m = 0; m < M; ++m)
    {
        for (int k = 0; k < 3; ++k) { p.lo[k][m] = lower[k * M + m]; p.hi[k][m] = upper[k * M + m]; }
        p.deadband[m] = velocity_deadband ? velocity_deadband[m] : -1.0;
    }
    const unsigned grid = (unsigned)((B + 255) / 256);
    const hipStream_t s = (hipStream_t)stream;
    if (dtype == JM_F64)
        hipLaunchKernelGGL((jm::k_pd_adapter<double>), dim3(grid), dim3(256), 0, s, p, (const double *)action, (double *)command_state,
                           (double *)out, (long long)B);
    else
        hipLaunchKernelGGL((jm::k_pd_adapter<float>), dim3(grid), dim3(256), 0, s, p, (const float *)action, (float *)command_state,
                           (float *)out, (long long)B);
    HIP_TRY(hipGetLastError());
    return JM_OK;
}

int32_t jm_block_motor_safety_limit(int32_t dtype, int64_t B, int32_t M, const void * encoder, const int32_t * encoder_index,
                                    const void * command, const double * kp, const double * kd, const double * soft_lo,
                                    const double * soft_hi, const double * vel_lim, const double * eff_lim, void * out, void * stream)
{
    if (!encoder || !encoder_index || !command || !kp || !kd || !soft_lo || !soft_hi || !vel_lim || !eff_lim || !out)
        return fail(JM_EINVAL, "jm_block_motor_safety_limit: null argument");
    if (B <= 0 || M <= 0 || M > JM_BLOCK_MAX_MOTORS) return fail(JM_EINVAL, "jm_block_motor_safety_limit: bad sizes");
    if (dtype != JM_F64 && dtype != JM_F32) return fail(JM_EINVAL, "jm_block_motor_safety_limit: bad dtype");
    jm::SafetyParams p;
    std::memset(&p, 0, sizeof(p));
    p.M = M;
    for (int m = 0; m < M; ++m)
    {
        p.enc_index[m] = encoder_index[m];
        p.kp[m] = kp[m]; p.kd[m] = kd[m]; p.soft_lo[m] = soft_lo[m]; p.soft_hi[m] = soft_hi[m];
        p.vel_lim[m] = vel_lim[m]; p.eff_lim[m] = eff_lim[m];
    }
    const unsigned grid = (unsigned)((B + 255) / 256);
    const hipStream_t s = (hipStream_t)stream;
    if (dtype == JM_F64)
        hipLaunchKernelGGL((jm::k_motor_safety_limit<double>), dim3(grid), dim3(256), 0, s, p, (const double *)encoder,
                           (const double *)command, (double *)out, (long long)B);
    else
        hipLaunchKernelGGL((jm::k_motor_safety_limit<float>), dim3(grid), dim3(256), 0, s, p, (const float *)encoder,
                           (const float *)command, (float *)out, (long long)B);
    HIP_TRY(hipGetLastError());
    return JM_OK;
}

// ziggurat tables: computed once on the host (random.cc:66-96), one copy per device
static int32_t ziggurat_tables_on_device(jm::rnd::ZigguratTables ** out)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    static std::mutex mtx;
    static std::map<int, jm::rnd::ZigguratTables *> tables;
    std::lock_guard<std::mutex> lock(mtx);
    auto it = tables.find(dev);
    if (it == tables.end())
    {
        jm::rnd::ZigguratTables host;
        jm::rnd::ziggurat_tables(host);
        jm::rnd::ZigguratTables * tab = nullptr;
        HIP_TRY(hipMalloc((void **)&tab, sizeof(host)));
        HIP_TRY(hipMemcpy(tab, &host, sizeof(host), hipMemcpyHostToDevice));
        tables[dev] = tab;
        *out = tab;
    }
    else *out = it->second;
    return JM_OK;
}

int32_t jm_block_sensor_noise(int32_t dtype, int64_t B, int32_t n_sensors, int32_t n_fields, void * data,
                              uint64_t * rng_state, const double * noise_std, const double * bias,
                              const double * rot_bias_inv, void * stream)
{
    if (!data) return fail(JM_EINVAL, "jm_block_sensor_noise: null data");
    if (B <= 0 || n_sensors <= 0 || n_fields <= 0 || n_fields > JM_NOISE_MAX_FIELDS ||
        n_sensors * n_fields > JM_NOISE_MAX_ROWS)
        return fail(JM_EINVAL, "jm_block_sensor_noise: bad sizes");
    if (dtype != JM_F64 && dtype != JM_F32) return fail(JM_EINVAL, "jm_block_sensor_noise: bad dtype");
    if (noise_std && !rng_state) return fail(JM_EINVAL, "jm_block_sensor_noise: noise needs the generator states");
    if (rot_bias_inv && (n_fields != 6 || n_sensors > JM_NOISE_MAX_ROT || !bias))
        return fail(JM_EINVAL, "jm_block_sensor_noise: the rotation bias applies to IMU fields (6 rows) with a bias");
    if (!noise_std && !bias) return JM_OK;
    jm::NoiseParams p{};
    p.n_sensors = n_sensors; p.n_fields = n_fields;
    p.has_noise = noise_std != nullptr; p.has_bias = bias != nullptr; p.has_rot = rot_bias_inv != nullptr;
    for (int i = 0; i < n_sensors * n_fields; ++i)
    {
        if (noise_std)
        {
            if (!(noise_std[i] >= 0.0)) return fail(JM_EINVAL, "jm_block_sensor_noise: negative noise standard deviation");
            p.noise_std[i] = (float)noise_std[i];
        }
        if (bias) p.bias[i] = bias[i];
    }
    if (rot_bias_inv)
        for (int s = 0; s < n_sensors; ++s)
            for (int k = 0; k < 9; ++k) p.rot[s][k] = rot_bias_inv[9 * s + k];
    jm::rnd::ZigguratTables * tab = nullptr;
    {
        const int32_t rc = ziggurat_tables_on_device(&tab);
        if (rc != JM_OK) return rc;
    }
    const dim3 grid((unsigned)((B + 255) / 256), (unsigned)n_sensors);
    const hipStream_t s = (hipStream_t)stream;
    if (dtype == JM_F64)
        hipLaunchKernelGGL((jm::k_sensor_noise<double>), grid, dim3(256), 0, s, p, tab, (double *)data, rng_state, (long long)B);
    else
        hipLaunchKernelGGL((jm::k_sensor_noise<float>), grid, dim3(256), 0, s, p, tab, (float *)data, rng_state, (long long)B);
    HIP_TRY(hipGetLastError());
    return JM_OK;
}

int32_t jm_block_sensor_delay(int32_t dtype, int64_t B, int32_t n_sensors, int32_t n_fields, void * data,
                              const void * history, const int32_t * slot, const double * times, int32_t n_history,
                              uint64_t * rng_state, const double * delay, const double * jitter, int32_t order,
                              void * stream)
{
    if (!data) return fail(JM_EINVAL, "jm_block_sensor_delay: null data");
    if (B <= 0 || n_sensors <= 0 || n_fields <= 0 || n_fields > JM_NOISE_MAX_FIELDS || n_sensors > JM_NOISE_MAX_ROWS)
        return fail(JM_EINVAL, "jm_block_sensor_delay: bad sizes");
    if (dtype != JM_F64 && dtype != JM_F32) return fail(JM_EINVAL, "jm_block_sensor_delay: bad dtype");
    if (order != 0 && order != 1)
        return fail(JM_ENOTIMPL, "`delayInterpolationOrder` must be either 0 or 1.");  // abstract_sensor.hxx:399-403
    if (history && (!slot || !times || n_history < 1 || n_history > JM_DELAY_MAX_HISTORY))
        return fail(JM_EINVAL, "jm_block_sensor_delay: the history needs 1..64 samples with their slots and times");
    if (!history && !rng_state) return JM_OK;
    jm::DelayParams p{};
    p.n_sensors = n_sensors; p.n_fields = n_fields; p.order = order;
    p.has_history = history != nullptr; p.n_hist = history ? n_history : 0;
    for (int i = 0; i < p.n_hist; ++i)
    {
        if (i > 0 && !(times[i] >= times[i - 1])) return fail(JM_EINVAL, "jm_block_sensor_delay: sample times must ascend");
        p.slot[i] = slot[i];
        p.times[i] = times[i];
    }
    for (int s = 0; s < n_sensors; ++s)
    {
        p.delay[s] = delay ? delay[s] : 0.0;
        p.jitter[s] = jitter ? (float)jitter[s] : 0.0f;
        if (!(p.delay[s] >= 0.0) || !(p.jitter[s] >= 0.0f)) return fail(JM_EINVAL, "jm_block_sensor_delay: negative delay or jitter");
    }
    const dim3 grid((unsigned)((B + 255) / 256), (unsigned)n_sensors);
    const hipStream_t s = (hipStream_t)stream;
    if (dtype == JM_F64)
        hipLaunchKernelGGL((jm::k_sensor_delay<double>), grid, dim3(256), 0, s, p, (double *)data, (const double *)history, rng_state, (long long)B);
    else
        hipLaunchKernelGGL((jm::k_sensor_delay<float>), grid, dim3(256), 0, s, p, (float *)data, (const float *)history, rng_state, (long long)B);
    HIP_TRY(hipGetLastError());
    return JM_OK;
}

int32_t jm_block_model_bias(int32_t dtype, int64_t B, int32_t njoints, int32_t first_joint, const double * nominal,
                            const float * std4, uint64_t * rng_state, const uint8_t * mask, void * model_lane, void * stream)
{
    if (!nominal || !std4 || !rng_state || !model_lane) return fail(JM_EINVAL, "jm_block_model_bias: null argument");
    if (B <= 0 || njoints < 1 || first_joint < 1 || first_joint > njoints) return fail(JM_EINVAL, "jm_block_model_bias: bad sizes");
    if (dtype != JM_F64 && dtype != JM_F32) return fail(JM_EINVAL, "jm_block_model_bias: bad dtype");
    for (int i = 0; i < 4; ++i)
        if (!(std4[i] >= 0.0f)) return fail(JM_EINVAL, "jm_block_model_bias: negative standard deviation");
    jm::BiasParams p{};
    p.njoints = njoints; p.first = first_joint;
    p.inertia_std = std4[0]; p.mass_std = std4[1]; p.com_std = std4[2]; p.pos_std = std4[3];
    jm::rnd::ZigguratTables * tab = nullptr;
    const int32_t rc = ziggurat_tables_on_device(&tab);
    if (rc != JM_OK) return rc;
    const dim3 grid((unsigned)((B + 255) / 256));
    const hipStream_t s = (hipStream_t)stream;
    if (dtype == JM_F64)
        hipLaunchKernelGGL((jm::k_model_bias<double>), grid, dim3(256), 0, s, p, tab, nominal, rng_state, mask, (double *)model_lane, (long long)B);
    else
        hipLaunchKernelGGL((jm::k_model_bias<float>), grid, dim3(256), 0, s, p, tab, nominal, rng_state, mask, (float *)model_lane, (long long)B);
    HIP_TRY(hipGetLastError());
    return JM_OK;
}

int32_t jm_engine_rng_seed(const uint32_t * seed, int64_t B, uint64_t * state_out)
{
    if (!seed || !state_out || B <= 0) return fail(JM_EINVAL, "jm_engine_rng_seed: bad arguments");
    for (int64_t lane = 0; lane < B; ++lane)
    {
        // internal::generateState (random.hxx:20-44): two 32-bit words of the sequence, low word first
        std::seed_seq seq{seed[lane]};
        uint32_t w[2];
        seq.generate(w, w + 2);
        state_out[lane] = jm::rnd::pcg32_init((uint64_t)w[0] | ((uint64_t)w[1] << 32));
    }
    return JM_OK;
}

int32_t jm_sensor_rng_seed(const uint32_t * group_seed, int64_t B, int32_t n_sensors, uint64_t * state_out)
{
    if (!group_seed || !state_out || B <= 0 || n_sensors <= 0) return fail(JM_EINVAL, "jm_sensor_rng_seed: bad arguments");
    std::vector<uint32_t> words((size_t)n_sensors);
    for (int64_t lane = 0; lane < B; ++lane)
    {
        std::seed_seq seq{group_seed[lane]};
        seq.generate(words.begin(), words.end());
        for (int32_t s = 0; s < n_sensors; ++s) state_out[(size_t)s * B + lane] = jm::rnd::pcg32_init(words[s]);
    }
    return JM_OK;
}

int32_t jm_last_error(char * buffer, size_t size)
{
    if (!buffer || size == 0) return JM_EINVAL;
    std::snprintf(buffer, size, "%s", g_last_error.c_str());
    return JM_OK;
}
}  // extern "C"
