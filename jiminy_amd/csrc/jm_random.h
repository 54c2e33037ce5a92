// jm_random.h -- sensor white noise and bias as a batched HIP kernel: one PCG32 stream per
// (sensor, lane), consumed exactly like the reference consumes the per-sensor generator.
//
// Reference:
//   PCG32 (pcg32_fast: 64-bit MCG, XSH-RS output)      core/src/utilities/random.cc:10-37
//   uniform = std::generate_canonical<float, 24>        core/src/utilities/random.cc:41-44
//   normal  = ziggurat, 128 strips, float arithmetic    core/src/utilities/random.cc:52-167
//   AbstractSensorBase::measureData (noise, then bias)  core/src/hardware/abstract_sensor.cc:71-85
//   ImuSensor::measureData (+ rotation bias)            core/src/hardware/basic_sensors.cc:166-187
//   generator seeding, one std::seed_seq per sensor group  core/include/jiminy/core/hardware/abstract_sensor.hxx:213-226
// The integer stream (PCG32 outputs, strip index, fast-path samples = 98.8 % of the draws) is
// bit-exact; the wedge / tail samples go through logf / expf, whose device implementation may
// differ from glibc's in the last float ulp.
#pragma once
#include <cmath>
#include <cstdint>

#ifndef JM_HOST_EMU
#include <hip/hip_runtime.h>
#define JM_RDEV __host__ __device__ inline
#else
#define JM_RDEV inline
#endif

#include "../../include/jiminy_hip.h"

namespace jm
{
namespace rnd
{
// random.cc:10-13: the constructor forces the two low bits (an MCG needs an odd state)
JM_RDEV uint64_t pcg32_init(uint64_t seed) { return seed | 3ULL; }
// random.cc:15-37 with the constants folded: opBits = 3, xShift = 22, bottomSpare - randShiftMax = 22
JM_RDEV uint32_t pcg32_next(uint64_t & state)
{
    state *= 6364136223846793005ULL;
    uint64_t s = state;
    const unsigned rshift = (unsigned)(s >> 61);
    s ^= s >> 22;
    return (uint32_t)(s >> (22u + rshift));
}
// std::generate_canonical<float, 24> over a 32-bit generator (libstdc++ bits/random.tcc): one
// draw, float(g()) / 2^32, results that round up to 1 are replaced by nextafter(1, 0)
JM_RDEV float uniform01(uint64_t & state)
{
    const float r = (float)pcg32_next(state) / 4294967296.0f;
    return r >= 1.0f ? 0.99999994f : r;
}

struct ZigguratTables
{
    uint32_t kn[128];
    float fn[128], wn[128];
};
// random.cc:66-96 (host, double precision, evaluated once)
inline void ziggurat_tables(ZigguratTables & z)
{
    const double m1 = 2147483648.0;
    const double vn = 9.91256303526217e-03;
    double dn = 3.442619855899, tn = dn;
    const double q = vn / std::exp(-0.5 * dn * dn);
    z.kn[0] = (uint32_t)((dn / q) * m1);
    z.kn[1] = 0;
    z.wn[0] = (float)(q / m1);
    z.wn[127] = (float)(dn / m1);
    z.fn[0] = 1.0f;
    z.fn[127] = (float)std::exp(-0.5 * dn * dn);
    for (int i = 126; i >= 1; --i)
    {
        dn = std::sqrt(-2.0 * std::log(vn / dn + std::exp(-0.5 * dn * dn)));
        z.kn[i + 1] = (uint32_t)((dn / tn) * m1);
        tn = dn;
        z.fn[i] = (float)std::exp(-0.5 * dn * dn);
        z.wn[i] = (float)(dn / m1);
    }
}
// random.cc:103-160. `std::fabs(hz)` of an int32 is a double; the comparison is done in double.
template<class Tab> JM_RDEV float normal01(uint64_t & state, const Tab & kn, const float * fn, const float * wn)
{
    // the reference is built without FMA contraction: keep `a + b * c` as two roundings
#pragma clang fp contract(off)
    const float r = 3.442620f;
    int32_t hz = (int32_t)pcg32_next(state);
    uint32_t iz = (uint32_t)hz & 127u;
    if (fabs((double)hz) < (double)kn[iz]) return (float)hz * wn[iz];
    for (;;)
    {
        float x, y;
        if (iz == 0)
        {
            for (;;)
            {
                x = -0.2904764f * logf(uniform01(state));
                y = -logf(uniform01(state));
                if (x * x <= y + y) break;
            }
            return hz <= 0 ? -r - x : r + x;
        }
        x = (float)hz * wn[iz];
        if (fn[iz] + uniform01(state) * (fn[iz - 1] - fn[iz]) < expf(-0.5f * x * x)) return x;
        hz = (int32_t)pcg32_next(state);
        iz = (uint32_t)hz & 127u;
        if (fabs((double)hz) < (double)kn[iz]) return (float)hz * wn[iz];
    }
}
}  // namespace rnd

struct NoiseParams
{
    int n_sensors, n_fields;
    int has_noise, has_bias, has_rot;
    float noise_std[JM_NOISE_MAX_ROWS];      // [sensor][field]
    double bias[JM_NOISE_MAX_ROWS];          // [sensor][field] (IMU: the last 6 of its 9 bias entries)
    double rot[JM_NOISE_MAX_ROT][9];         // IMU: exp3(-bias.head<3>()), row-major
};

// ---- sensor delay and jitter: AbstractSensorTpl<T>::interpolateData (abstract_sensor.hxx:305-429)
#define JM_DELAY_MAX_HISTORY 64
struct DelayParams
{
    int n_sensors, n_fields, n_hist, order, has_history;
    int slot[JM_DELAY_MAX_HISTORY];      // physical ring slot of the i-th oldest sample
    double times[JM_DELAY_MAX_HISTORY];  // ascending sample times, the last one is the current time
    double delay[JM_NOISE_MAX_ROWS];     // [sensor]
    float jitter[JM_NOISE_MAX_ROWS];     // [sensor]
};
// Which stored sample a lane reads for `t_read = t_now - delay` (and the interpolation weight towards the
// next one), same answers as the reference's history lookup (abstract_sensor.hxx:305-429) in a form that
// suits the device: the sample times are lane-uniform kernel parameters (at most 64 of them, scalar
// registers), so the position of `t_read` among them is a branch-free COUNT of the samples not younger than
// it -- no per-lane bisection, no divergent loop.  Cases, in the reference's order:
//   * zero-order hold adds 1e-10 s to `t_read`, so that a delay equal to a whole number of periods always
//     lands on the same sample;
//   * `t_read` inside the recorded history: the newest sample with time <= t_read (linear interpolation
//     towards its successor when order == 1); older than everything recorded -> the oldest sample (the
//     engine sizes the ring so that this does not occur; the reference raises there);
//   * `t_read` < 0 or ahead of the newest sample while a delay is configured (the ring is still filling up
//     after a start / reset): the newest sample taken at t <= 0, i.e. the initial measurement;
//   * no delay at all: the current sample.
JM_RDEV void delay_lookup(const DelayParams & p, int s, double delay, int & idx, double & ratio)
{
    const int n = p.n_hist;
    const double t_read = p.times[n - 1] - delay + (p.order == 0 ? 1.0e-10 : 0.0);
    int not_younger = 0, not_positive = 0;
    for (int i = 0; i < n; ++i)
    {
        not_younger += (p.times[i] <= t_read) ? 1 : 0;
        not_positive += (p.times[i] <= 0.0) ? 1 : 0;
    }
    const int newest_before = not_younger - 1;                 // -1: nothing that old was recorded
    const bool inside = t_read >= 0.0 && newest_before + 1 < n;
    const bool delayed = p.delay[s] > 2.220446049250313e-16 || (double)p.jitter[s] > 2.220446049250313e-16;
    const int initial = not_positive > 0 ? not_positive - 1 : 0;
    idx = inside ? (newest_before < 0 ? 0 : newest_before) : (delayed ? initial : n - 1);
    ratio = 0.0;
    if (inside && p.order == 1 && newest_before >= 0)
    {
        const double t0 = p.times[newest_before], t1 = p.times[newest_before + 1];
        ratio = (t_read - t0) / (t1 - t0);
    }
}

#ifndef JM_HOST_EMU
// data [n_sensors * n_fields][B]: overwritten with the delayed measurement read from the history ring
// `hist` [slots][n_sensors * n_fields][B]; rng [n_sensors][B] (one uniform draw per sensor and call, taken
// whether or not a jitter is configured, exactly like the reference)
template<class T>
__global__ void __launch_bounds__(256) k_sensor_delay(const DelayParams p, T * data, const T * hist, uint64_t * rng, long long B)
{
    const long long lane = (long long)blockIdx.x * 256 + threadIdx.x;
    const int s = blockIdx.y;
    if (lane >= B) return;
    double delay = p.delay[s];
    if (rng)
    {
        uint64_t st = rng[(long long)s * B + lane];
        // uniform(generator_, 0.0F, jitter) = std::uniform_real_distribution<float>(0, jitter)(g)
        const float u = rnd::uniform01(st);
        delay += (double)(u * (p.jitter[s] - 0.0f) + 0.0f);
        rng[(long long)s * B + lane] = st;
    }
    if (!p.has_history) return;
    int idx;
    double ratio;
    delay_lookup(p, s, delay, idx, ratio);
    const long long rows = (long long)p.n_sensors * p.n_fields;
    const T * h0 = hist + ((long long)p.slot[idx] * rows + (long long)s * p.n_fields) * B + lane;
    T * d = data + (long long)s * p.n_fields * B + lane;
    if (ratio != 0.0)
    {
        const T * h1 = hist + ((long long)p.slot[idx + 1] * rows + (long long)s * p.n_fields) * B + lane;
        for (int f = 0; f < p.n_fields; ++f)
        {
            const double a = (double)h0[(long long)f * B], b = (double)h1[(long long)f * B];
            d[(long long)f * B] = (T)(a + ratio * (b - a));
        }
    }
    else
        for (int f = 0; f < p.n_fields; ++f) d[(long long)f * B] = h0[(long long)f * B];
}
#endif

// ---- model biases: Model::addBiasedToExtendedModel (model.cc:1166-1236) for one robot, generator state in / out.
// `nom`: the 25 nominal scalars of the joint (jiminy_hip.h jm_block_model_bias), `out`: its 13 biased scalars.
struct BiasParams
{
    int njoints, first;
    float inertia_std, mass_std, com_std, pos_std;
};
namespace rnd
{
// pinocchio::exp3 (explog.hpp, v2.7.0): Rodrigues formula, Taylor expansion below eps^(1/4)
JM_RDEV void exp3_rowmajor(const double * v, double * R)
{
    const double t2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const double t = sqrt(t2);
    double a_vxvx, a_vx, dg;
    if (t > 1.220703125e-4)
    {
        const double ct = cos(t), st = sin(t);
        a_vxvx = (1.0 - ct) / t2; a_vx = st / t; dg = ct;
    }
    else
    {
        a_vxvx = 0.5 - t2 / 24.0; a_vx = 1.0 - t2 / 6.0; dg = 1.0 - t2 / 2.0;
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[3 * i + j] = a_vxvx * v[i] * v[j];
    R[1] -= a_vx * v[2]; R[3] += a_vx * v[2];
    R[2] += a_vx * v[1]; R[6] -= a_vx * v[1];
    R[5] -= a_vx * v[0]; R[7] += a_vx * v[0];
    R[0] += dg; R[4] += dg; R[8] += dg;
}
template<class Tab> JM_RDEV void bias_one_joint(const BiasParams & p, const double * nom, uint64_t & st, const Tab & kn,
                                                const float * fn, const float * wn, double * out)
{
    // `normal(g, mean, std)` = normal01(g) * std + mean in float, two roundings (random.cc:162-167)
#pragma clang fp contract(off)
    const double eps = 2.220446049250313e-16;
    for (int i = 0; i < 13; ++i) out[i] = nom[i];
    if ((double)p.com_std > eps)
        for (int i = 0; i < 3; ++i) out[1 + i] *= (double)(normal01(st, kn, fn, wn) * p.com_std + 1.0f);
    if ((double)p.mass_std > eps)
    {
        const double m = out[0];
        const double mb = m * (double)(normal01(st, kn, fn, wn) * p.mass_std + 1.0f);
        const double lo = m < 1.0e-3 ? m : 1.0e-3;
        out[0] = mb > lo ? mb : lo;
    }
    if ((double)p.inertia_std > eps)
    {
        // principal axes rotated by exp3(N(0, std)^3), principal moments scaled by N(1, std)^3, I = A diag(M) A^T
        // (the reference goes through Eigen::Quaterniond(exp3(.)): the same rotation up to rounding)
        double ra[3], R[9], A[9], M[3];
        for (int i = 0; i < 3; ++i) ra[i] = (double)(normal01(st, kn, fn, wn) * p.inertia_std + 0.0f);
        exp3_rowmajor(ra, R);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                A[3 * i + j] = nom[16 + 3 * i] * R[j] + nom[16 + 3 * i + 1] * R[3 + j] + nom[16 + 3 * i + 2] * R[6 + j];
        for (int i = 0; i < 3; ++i) M[i] = nom[13 + i] * (double)(normal01(st, kn, fn, wn) * p.inertia_std + 1.0f);
        int o = 4;
        for (int i = 0; i < 3; ++i)
            for (int j = i; j < 3; ++j)
                out[o++] = A[3 * i] * M[0] * A[3 * j] + A[3 * i + 1] * M[1] * A[3 * j + 1] + A[3 * i + 2] * M[2] * A[3 * j + 2];
    }
    if ((double)p.pos_std > eps)
        for (int i = 0; i < 3; ++i) out[10 + i] *= (double)(normal01(st, kn, fn, wn) * p.pos_std + 1.0f);
}
}  // namespace rnd

#ifndef JM_HOST_EMU
// one thread per lane: the mechanical joints in index order, the lane's engine generator advanced like the reference's
template<class T>
__global__ void __launch_bounds__(256) k_model_bias(const BiasParams p, const rnd::ZigguratTables * tables, const double * nominal,
                                                    uint64_t * rng, const uint8_t * mask, T * model_lane, long long B)
{
    __shared__ uint32_t kn[128];
    __shared__ float fn[128], wn[128];
    if (threadIdx.x < 128)
    {
        kn[threadIdx.x] = tables->kn[threadIdx.x];
        fn[threadIdx.x] = tables->fn[threadIdx.x];
        wn[threadIdx.x] = tables->wn[threadIdx.x];
    }
    __syncthreads();
    const long long lane = (long long)blockIdx.x * 256 + threadIdx.x;
    if (lane >= B) return;
    if (mask && !mask[lane]) return;
    uint64_t st = rng[lane];
    for (int j = p.first; j < p.njoints; ++j)
    {
        double nom[25], out[13];
        for (int i = 0; i < 25; ++i) nom[i] = nominal[25 * j + i];
        rnd::bias_one_joint(p, nom, st, kn, fn, wn, out);
        for (int i = 0; i < 13; ++i) model_lane[(long long)(13 * j + i) * B + lane] = (T)out[i];
    }
    rng[lane] = st;
}
#endif

#ifndef JM_HOST_EMU
// data: [n_sensors * n_fields][B] (row = sensor * n_fields + field), rng: [n_sensors][B]
template<class T>
__global__ void __launch_bounds__(256) k_sensor_noise(const NoiseParams p, const rnd::ZigguratTables * tables, T * data,
                                                      uint64_t * rng, long long B)
{
    __shared__ uint32_t kn[128];
    __shared__ float fn[128], wn[128];
    if (threadIdx.x < 128)
    {
        kn[threadIdx.x] = tables->kn[threadIdx.x];
        fn[threadIdx.x] = tables->fn[threadIdx.x];
        wn[threadIdx.x] = tables->wn[threadIdx.x];
    }
    __syncthreads();
    const long long lane = (long long)blockIdx.x * 256 + threadIdx.x;
    const int s = blockIdx.y;
    if (lane >= B) return;
    T * d = data + (long long)s * p.n_fields * B + lane;
    double x[JM_NOISE_MAX_FIELDS];
    for (int f = 0; f < p.n_fields; ++f) x[f] = (double)d[(long long)f * B];
    if (p.has_noise)
    {
        uint64_t st = rng[(long long)s * B + lane];
        // get() += normal(generator_, 0.0F, noiseStd.cast<float>()).cast<double>()
        for (int f = 0; f < p.n_fields; ++f)
            x[f] += (double)(rnd::normal01(st, kn, fn, wn) * p.noise_std[s * p.n_fields + f] + 0.0f);
        rng[(long long)s * B + lane] = st;
    }
    if (p.has_bias)
    {
        for (int f = 0; f < p.n_fields; ++f) x[f] += p.bias[s * p.n_fields + f];
        if (p.has_rot)
        {
            const double * R = p.rot[s];
            for (int h = 0; h < 6; h += 3)
            {
                const double a = x[h], b = x[h + 1], c = x[h + 2];
                x[h] = R[0] * a + R[1] * b + R[2] * c;
                x[h + 1] = R[3] * a + R[4] * b + R[5] * c;
                x[h + 2] = R[6] * a + R[7] * b + R[8] * c;
            }
        }
    }
    for (int f = 0; f < p.n_fields; ++f) d[(long long)f * B] = (T)x[f];
}
#endif
}  // namespace jm
