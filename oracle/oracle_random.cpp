// oracle_random.cpp -- CPU restatement of the reference's sensor noise path. TEST INFRASTRUCTURE ONLY
// (same rules as oracle.cpp: only tests/, smoke() and bench.py's cpu_baseline leg may load it).
//
// Follows, in the reference's order of operations (paths relative to the reference tree):
//   Pcg32                PCG32::PCG32 / operator()          core/src/utilities/random.cc:10-37
//   uniform01()          uniform(g) = std::generate_canonical<float, 24>   random.cc:41-44
//                        -- the real libstdc++ template, as in the reference build
//   Ziggurat / normal01  ZigguratNormalData, internal::normal   random.cc:52-160
//   normal()             normal(g, mean, stddev)               random.cc:164-167
//   orc_sensor_noise     AbstractSensorBase::measureData       core/src/hardware/abstract_sensor.cc:71-85
//                        ImuSensor::measureData                core/src/hardware/basic_sensors.cc:166-187
//   orc_model_bias       Model::addBiasedToExtendedModel       core/src/robot/model.cc:1166-1236
//   orc_engine_rng_seed  Engine::generator_ seeding            core/src/engine/engine.cc:756-757, utilities/random.hxx:20-51
//   orc_sensor_rng_seed  AbstractSensorTpl<T>::resetAll        core/include/jiminy/core/hardware/abstract_sensor.hxx:213-226
//                        -- std::seed_seq is the real libstdc++ class, as in the reference build
// Parity status: the reference holds no golden vectors for its generators (core/unit/random_test.cc
// checks interfaces and gradients only) and cannot be compiled here (Eigen absent): PARITY UNPINNED
// against reference outputs; pinned instead by tests/test_sensor_noise.py against an independent
// numpy restatement of PCG-XSH-RS 64/32 (MCG), of the std::seed_seq algorithm ([rand.util.seedseq])
// and by distribution tests of the ziggurat (moments, Kolmogorov-Smirnov against scipy).
#include <array>
#include <cmath>
#include <cstdint>
#include <limits>
#include <random>
#include <vector>

namespace
{
class Pcg32
{
public:
    using result_type = uint32_t;
    explicit Pcg32(uint64_t state) noexcept : state_{state | 3ULL} {}
    static constexpr result_type min() noexcept { return std::numeric_limits<result_type>::min(); }
    static constexpr result_type max() noexcept { return std::numeric_limits<result_type>::max(); }
    result_type operator()() noexcept
    {
        constexpr uint8_t bits = 64, uint32Bits = 32, spareBits = bits - uint32Bits;
        constexpr uint8_t opBits = spareBits - 5 >= 64 ? 5 : spareBits - 4 >= 32 ? 4 : spareBits - 3 >= 16 ? 3 :
                                   spareBits - 2 >= 4 ? 2 : spareBits - 1 >= 1 ? 1 : 0;
        constexpr uint8_t mask = (1 << opBits) - 1;
        constexpr uint8_t randShiftMax = mask;
        constexpr uint8_t topSpare = opBits;
        constexpr uint8_t bottomSpare = spareBits - topSpare;
        constexpr uint8_t xShift = topSpare + (uint32Bits + randShiftMax) / 2;
        state_ *= 6364136223846793005ULL;
        uint64_t state = state_;
        uint8_t rshift = opBits ? static_cast<uint8_t>(state >> (bits - opBits)) & mask : 0U;
        state ^= state >> xShift;
        return static_cast<uint32_t>(state >> (bottomSpare - randShiftMax + rshift));
    }
    uint64_t state() const noexcept { return state_; }

private:
    uint64_t state_;
};

float uniform01(Pcg32 & g) { return std::generate_canonical<float, std::numeric_limits<float>::digits>(g); }

struct Ziggurat
{
    Ziggurat()
    {
        constexpr double m1 = 2147483648.0;
        constexpr double vn = 9.91256303526217e-03;
        double dn = 3.442619855899;
        double tn = dn;
        const double q = vn / std::exp(-0.5 * dn * dn);
        kn[0] = static_cast<uint32_t>((dn / q) * m1);
        kn[1] = 0;
        wn[0] = static_cast<float>(q / m1);
        wn[127] = static_cast<float>(dn / m1);
        fn[0] = 1.0F;
        fn[127] = static_cast<float>(std::exp(-0.5 * dn * dn));
        for (uint8_t i = 126; 1 <= i; i--)
        {
            dn = std::sqrt(-2.0 * std::log(vn / dn + std::exp(-0.5 * dn * dn)));
            kn[i + 1] = static_cast<uint32_t>((dn / tn) * m1);
            tn = dn;
            fn[i] = static_cast<float>(std::exp(-0.5 * dn * dn));
            wn[i] = static_cast<float>(dn / m1);
        }
    }
    std::array<uint32_t, 128> kn{};
    std::array<float, 128> fn{};
    std::array<float, 128> wn{};
};
const Ziggurat ZIG{};

float normal01(Pcg32 & g)
{
    const auto & kn = ZIG.kn; const auto & fn = ZIG.fn; const auto & wn = ZIG.wn;
    constexpr float r = 3.442620F;
    int32_t hz;
    uint32_t iz;
    float x, y;
    hz = static_cast<int32_t>(g());
    iz = (static_cast<uint32_t>(hz) & 127UL);
    if (std::fabs(hz) < kn[iz]) return static_cast<float>(hz) * wn[iz];
    while (true)
    {
        if (iz == 0)
        {
            while (true)
            {
                x = -0.2904764F * std::log(uniform01(g));
                y = -std::log(uniform01(g));
                if (x * x <= y + y) break;
            }
            if (hz <= 0) return -r - x;
            return r + x;
        }
        x = static_cast<float>(hz) * wn[iz];
        if (fn[iz] + uniform01(g) * (fn[iz - 1] - fn[iz]) < std::exp(-0.5F * x * x)) return x;
        hz = static_cast<int32_t>(g());
        iz = (hz & 127);
        if (std::fabs(hz) < kn[iz]) return static_cast<float>(hz) * wn[iz];
    }
}

float normal(Pcg32 & g, float mean, float stddev) { return normal01(g) * stddev + mean; }
}  // namespace

extern "C"
{
// raw generator outputs / normal samples of one stream (state in/out)
void orc_pcg32_stream(uint64_t * state, int64_t n, uint32_t * out)
{
    Pcg32 g(*state);
    for (int64_t i = 0; i < n; ++i) out[i] = g();
    *state = g.state();
}
void orc_uniform_stream(uint64_t * state, int64_t n, float * out)
{
    Pcg32 g(*state);
    for (int64_t i = 0; i < n; ++i) out[i] = uniform01(g);
    *state = g.state();
}
void orc_normal_stream(uint64_t * state, int64_t n, float * out)
{
    Pcg32 g(*state);
    for (int64_t i = 0; i < n; ++i) out[i] = normal01(g);
    *state = g.state();
}
void orc_ziggurat_tables(uint32_t * kn, float * fn, float * wn)
{
    for (int i = 0; i < 128; ++i) { kn[i] = ZIG.kn[i]; fn[i] = ZIG.fn[i]; wn[i] = ZIG.wn[i]; }
}
void orc_seed_seq(uint32_t seed, int32_t n, uint32_t * out)
{
    std::seed_seq seq{seed};
    std::vector<uint32_t> w((size_t)n);
    seq.generate(w.begin(), w.end());
    for (int32_t i = 0; i < n; ++i) out[i] = w[i];
}
// ---- model biases: Model::addBiasedToExtendedModel (core/src/robot/model.cc:1166-1236), one robot per lane.
// The engine generator of a lane: PCG32(std::seed_seq{seed}) through internal::generateState
// (core/include/jiminy/core/utilities/random.hxx:20-51: two 32-bit words, low word first; engine.cc:756-757)
void orc_engine_rng_seed(const uint32_t * seed, int64_t B, uint64_t * state_out)
{
    for (int64_t l = 0; l < B; ++l)
    {
        std::seed_seq seq{seed[l]};
        std::array<uint32_t, 2> buffer;
        seq.generate(buffer.begin(), buffer.end());
        uint64_t value = 0;
        uint32_t shift = 0;
        for (std::size_t j = 0; j < 2; ++j)
        {
            value |= static_cast<uint64_t>(buffer[j]) << shift;
            shift += 32;
        }
        state_out[l] = Pcg32(value).state();
    }
}
// nominal [njoints][25]: mass | com 3 | inertia xx xy xz yy yz zz | placement translation 3 | principal moments 3 |
// principal axes 9 (row-major, columns = eigenvectors) -- the eigen-decomposition is an input because its sign /
// ordering conventions are the solver's (Eigen::SelfAdjointEigenSolver in the reference), not part of the law.
// std4: inertia, mass, com, relative position.  out [13 * njoints][B]; rng [B] in / out; mask [B] or null.
void orc_model_bias(int64_t B, int32_t njoints, int32_t first_joint, const double * nominal, const float * std4,
                    uint64_t * rng, const uint8_t * mask, double * out)
{
    const double EPS = std::numeric_limits<double>::epsilon();
    const float inertiaBiasStd = std4[0], massBiasStd = std4[1], comBiasStd = std4[2], relativeBodyPosBiasStd = std4[3];
    for (int64_t l = 0; l < B; ++l)
    {
        if (mask && !mask[l]) continue;
        Pcg32 g(rng[l]);
        for (int32_t j = first_joint; j < njoints; ++j)   // mechanicalJointNames_: every joint but the free-flyer root
        {
            const double * nom = nominal + 25 * j;
            double mass = nom[0], com[3] = {nom[1], nom[2], nom[3]}, I[6] = {nom[4], nom[5], nom[6], nom[7], nom[8], nom[9]};
            double pos[3] = {nom[10], nom[11], nom[12]};
            if (comBiasStd > EPS)
                for (int i = 0; i < 3; ++i) com[i] *= static_cast<double>(normal(g, 1.0F, comBiasStd));
            if (massBiasStd > EPS)
                mass = std::max(mass * normal(g, 1.0F, massBiasStd), std::min(mass, 1.0e-3));
            if (inertiaBiasStd > EPS)
            {
                double randAxis[3];
                for (int i = 0; i < 3; ++i) randAxis[i] = static_cast<double>(normal(g, 0.0F, inertiaBiasStd));
                // pinocchio::exp3 (explog.hpp)
                const double t2 = randAxis[0] * randAxis[0] + randAxis[1] * randAxis[1] + randAxis[2] * randAxis[2];
                const double t = std::sqrt(t2);
                double alpha_vxvx, alpha_vx, diag;
                if (t > 1.220703125e-4) { alpha_vxvx = (1.0 - std::cos(t)) / t2; alpha_vx = std::sin(t) / t; diag = std::cos(t); }
                else { alpha_vxvx = 0.5 - t2 / 24.0; alpha_vx = 1.0 - t2 / 6.0; diag = 1.0 - t2 / 2.0; }
                double R[3][3];
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) R[a][b] = alpha_vxvx * randAxis[a] * randAxis[b];
                R[0][1] -= alpha_vx * randAxis[2]; R[1][0] += alpha_vx * randAxis[2];
                R[0][2] += alpha_vx * randAxis[1]; R[2][0] -= alpha_vx * randAxis[1];
                R[1][2] -= alpha_vx * randAxis[0]; R[2][1] += alpha_vx * randAxis[0];
                for (int a = 0; a < 3; ++a) R[a][a] += diag;
                // inertiaBodyAxes = inertiaBodyAxes * Quaterniond(exp3(randAxis)); moments *= normal(3, 1, g, 1, std)
                double A[3][3], M[3];
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b)
                        A[a][b] = nom[16 + 3 * a] * R[0][b] + nom[16 + 3 * a + 1] * R[1][b] + nom[16 + 3 * a + 2] * R[2][b];
                for (int i = 0; i < 3; ++i) M[i] = nom[13 + i] * static_cast<double>(normal(g, 1.0F, inertiaBiasStd));
                int o = 0;
                for (int a = 0; a < 3; ++a)
                    for (int b = a; b < 3; ++b) I[o++] = A[a][0] * M[0] * A[b][0] + A[a][1] * M[1] * A[b][1] + A[a][2] * M[2] * A[b][2];
            }
            if (relativeBodyPosBiasStd > EPS)
                for (int i = 0; i < 3; ++i) pos[i] *= static_cast<double>(normal(g, 1.0F, relativeBodyPosBiasStd));
            double * o = out + (size_t)(13 * j) * B + l;
            o[0] = mass;
            for (int i = 0; i < 3; ++i) o[(size_t)(1 + i) * B] = com[i];
            for (int i = 0; i < 6; ++i) o[(size_t)(4 + i) * B] = I[i];
            for (int i = 0; i < 3; ++i) o[(size_t)(10 + i) * B] = pos[i];
        }
        rng[l] = g.state();
    }
}
// states [n_sensors][B] of AbstractSensorTpl::resetAll(group_seed[lane]) for every lane
void orc_sensor_rng_seed(const uint32_t * group_seed, int64_t B, int32_t n_sensors, uint64_t * state_out)
{
    std::vector<uint32_t> w((size_t)n_sensors);
    for (int64_t l = 0; l < B; ++l)
    {
        std::seed_seq seq{group_seed[l]};
        seq.generate(w.begin(), w.end());
        for (int32_t s = 0; s < n_sensors; ++s) state_out[(size_t)s * B + l] = Pcg32(w[s]).state();
    }
}
// measureData of every (sensor, lane): data [n_sensors][n_fields][B] float64, rng [n_sensors][B];
// noise_std / bias [n_sensors][n_fields] or null; rot [n_sensors][9] (IMU only) or null
void orc_sensor_noise(int64_t B, int32_t n_sensors, int32_t n_fields, double * data, uint64_t * rng,
                      const double * noise_std, const double * bias, const double * rot)
{
    for (int32_t s = 0; s < n_sensors; ++s)
        for (int64_t l = 0; l < B; ++l)
        {
            double x[16];
            for (int32_t f = 0; f < n_fields; ++f) x[f] = data[((size_t)s * n_fields + f) * B + l];
            if (noise_std)
            {
                Pcg32 g(rng[(size_t)s * B + l]);
                for (int32_t f = 0; f < n_fields; ++f)
                    x[f] += static_cast<double>(normal(g, 0.0F, static_cast<float>(noise_std[s * n_fields + f])));
                rng[(size_t)s * B + l] = g.state();
            }
            if (bias)
            {
                for (int32_t f = 0; f < n_fields; ++f) x[f] += bias[s * n_fields + f];
                if (rot)
                {
                    const double * R = rot + 9 * s;
                    for (int h = 0; h < 6; h += 3)
                    {
                        const double a = x[h], b = x[h + 1], c = x[h + 2];
                        x[h] = R[0] * a + R[1] * b + R[2] * c;
                        x[h + 1] = R[3] * a + R[4] * b + R[5] * c;
                        x[h + 2] = R[6] * a + R[7] * b + R[8] * c;
                    }
                }
            }
            for (int32_t f = 0; f < n_fields; ++f) data[((size_t)s * n_fields + f) * B + l] = x[f];
        }
}

// interpolateData of every (sensor, lane) (abstract_sensor.hxx:305-429): data [n_sensors][n_fields][B] is
// overwritten with the delayed measurement read from hist [slots][n_sensors * n_fields][B]; slot / times
// [n_hist] name the ring slot and the time of the i-th oldest sample (the last one is the current time).
// One `uniform(generator_, 0.0F, jitter)` draw per sensor and call; hist null: only that draw.
void orc_sensor_delay(int64_t B, int32_t n_sensors, int32_t n_fields, double * data, const double * hist,
                      const int32_t * slot, const double * times, int32_t n_hist, uint64_t * rng,
                      const double * delay, const double * jitter, int32_t order)
{
    const int64_t rows = (int64_t)n_sensors * n_fields;
    const double EPS = std::numeric_limits<double>::epsilon();
    for (int32_t s = 0; s < n_sensors; ++s)
        for (int64_t l = 0; l < B; ++l)
        {
            double d = delay ? delay[s] : 0.0;
            const float jit = jitter ? static_cast<float>(jitter[s]) : 0.0F;
            if (rng)
            {
                Pcg32 g(rng[(size_t)s * B + l]);
                d += std::uniform_real_distribution<float>(0.0F, jit)(g);
                rng[(size_t)s * B + l] = g.state();
            }
            if (!hist) continue;
            double timeDesired = times[n_hist - 1] - d;
            if (order == 0) timeDesired += 1e-10;  // STEPPER_MIN_TIMESTEP
            std::ptrdiff_t idxLeft;
            {
                std::ptrdiff_t left = 0, right = n_hist - 1, mid = 0;
                if (timeDesired >= times[n_hist - 1]) idxLeft = right;
                else if (timeDesired < times[0]) idxLeft = -1;
                else
                {
                    bool found = false;
                    idxLeft = 0;
                    while (left < right)
                    {
                        mid = (left + right) / 2;
                        if (timeDesired < times[mid]) right = mid;
                        else if (timeDesired > times[mid]) left = mid + 1;
                        else { idxLeft = mid; found = true; break; }
                    }
                    if (!found) idxLeft = (timeDesired < times[mid]) ? mid - 1 : mid;
                }
            }
            auto sample = [&](std::ptrdiff_t i, int32_t f) {
                return hist[((size_t)slot[i] * rows + (size_t)s * n_fields + f) * B + l];
            };
            for (int32_t f = 0; f < n_fields; ++f)
            {
                double out;
                if (timeDesired >= 0.0 && idxLeft + 1 < n_hist)
                {
                    if (idxLeft < 0) out = sample(0, f);  // the reference throws "No data old enough is available."
                    else if (order == 0) out = sample(idxLeft, f);
                    else
                    {
                        const double ratio = (timeDesired - times[idxLeft]) / (times[idxLeft + 1] - times[idxLeft]);
                        const double prev = sample(idxLeft, f), next = sample(idxLeft + 1, f);
                        out = prev + ratio * (next - prev);
                    }
                }
                else if ((delay && delay[s] > EPS) || jit > EPS)
                {
                    std::ptrdiff_t index = n_hist - 1;
                    for (std::ptrdiff_t i = 0; i < n_hist; ++i)
                        if (times[i] > 0) { index = std::max<std::ptrdiff_t>(0, i - 1); break; }
                    out = sample(index, f);
                }
                else out = sample(n_hist - 1, f);
                data[((size_t)s * n_fields + f) * B + l] = out;
            }
        }
}
}
