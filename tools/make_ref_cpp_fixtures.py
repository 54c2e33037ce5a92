"""Reference-pinned fixtures of the C++ leaf functions of the hot path: COMPILES AND RUNS THE REFERENCE'S OWN TEXT.

The reference core cannot be built here (Eigen / Boost / Pinocchio / hpp-fcl are absent), but its LEAF
functions need little or nothing of those.  This script cuts them, by line range, out of the files where they lie
under /root/reference, pastes them into a generated translation unit under oracle/_ref/ (git-ignored: nothing of the
reference is committed), compiles it with g++ and runs it on seeded inputs.  Inputs and the outputs the reference's
code produced for them go to tests/golden/ref_cpp_leaves.npz.  Every cut is anchored: the first and last line of a
range must start with the expected text, otherwise the script stops (a different reference revision would move them).

Two tiers, labelled per array group in the .npz (`tier__<group>`), in DESIGN.md section 5 and in the tests:

  tier A  "reference-compiled": the TU contains reference text + standard headers only.
          PCG32 (+ its seed_seq constructor), uniform, the ziggurat normal and its tables, xxHash, MurmurHash3
          (core/src/utilities/random.cc:8-318, utilities/random.h, random.hxx, fwd.h `function_ref`),
          the sub-step selection of Engine::step (engine.cc:2063-2089: stretch onto the breakpoint, snap to microseconds),
          the update-period arithmetic minClipped / isGcdIncluded (utilities/helpers.hxx:59-116),
          the end time of Engine::step with its Kahan compensation and the time to the next breakpoint (engine.cc:1793-1795,
          1991-2018), the bookkeeping of the adaptive loop after a try (engine.cc:2166-2172, 2197-2208, 2221), the refresh rule of profile forces / the controller and the activity of impulse forces (engine.cc:1857-1869,
          1903-1907, 1923-1927),
          the step-size controller of RungeKuttaDOPRIStepper::adjustStep (runge_kutta_dopri_stepper.cc:24-56 with
          the constants of runge_kutta_dopri_stepper.h:34-47) and the body of SimpleMotor::computeEffort
          (basic_motors.cc:89-142; the option struct around it is a plain data holder with the reference's member
          names -- the real one is built from a boost::variant map).
  tier B  "reference text on a stand-in vector type": Engine::computeContactDynamics (engine.cc:3197-3238), the PGS
          block table, sweep and solver loop (constraint_solvers.cc:45-91, 107-222, 224-326) and the RK4 / DOPRI
          Butcher tableaux (runge_kutta4_stepper.h:12-23, runge_kutta_dopri_stepper.h:12-29) are written against
          Eigen; they compile here against tools/ref_cpp/mini_linalg.h, ~150 lines giving the same spelling.  By
          the rules of this build that is NOT a reference build (a stand-in for a header the image lacks): these
          arrays are a second reading of those lines by the reference's own text, not a pin.

Run in the build container:   python tools/make_ref_cpp_fixtures.py
Consumers: tests/test_reference_cpp_leaves.py (oracle on the CPU; HIP kernels through the C ABI with -m gpu).
"""
from __future__ import annotations

import os
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("JIMINY_REFERENCE", "/root/reference")
WORK = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(ROOT, "tests", "golden", "ref_cpp_leaves.npz")
CORE = "core/include/jiminy/core"
MOTOR_GROUP, CONTACT_GROUP, CONTACT_FLAT_GROUPS = 64, 64, 6


def grab(rel: str, first: int, last: int, starts: str, ends: str) -> str:
    """Lines first..last (1-based, inclusive) of a reference file; both ends anchored on their expected text."""
    with open(os.path.join(REF, rel)) as f:
        lines = f.read().split("\n")
    a, b = lines[first - 1].strip(), lines[last - 1].strip()
    if not a.startswith(starts) or not b.startswith(ends):
        raise RuntimeError(f"{rel}:{first}-{last}: anchors moved ({a!r} / {b!r}); expected {starts!r} / {ends!r}")
    return f"// ---- {rel}:{first}-{last}\n" + "\n".join(lines[first - 1:last]) + "\n"


STD_HEADERS = """
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <numeric>
#include <random>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>
#define JIMINY_DLLAPI
#define JIMINY_THROW(exc, ...) throw exc("jiminy")
"""

IO_HELPERS = r"""
namespace io
{
static FILE * fin = nullptr;
static FILE * fout = nullptr;
template<typename T> std::vector<T> get(size_t n)
{
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, fin) != n) { fprintf(stderr, "short read\n"); exit(3); }
    return v;
}
inline int64_t geti() { return get<int64_t>(1)[0]; }
template<typename T> void put(const std::vector<T> & v) { if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), fout); }
}
"""


def tu_tier_a() -> str:
    """Reference text + standard headers only."""
    parts = [STD_HEADERS, "namespace jiminy\n{\n"]
    parts.append(grab(f"{CORE}/constants.h", 24, 26, "inline constexpr double INF", "inline constexpr double qNAN"))
    parts.append(grab(f"{CORE}/traits.h", 87, 94, "template<class T>", "using remove_cvref_t"))
    parts.append(grab(f"{CORE}/fwd.h", 100, 166, "template<typename F>", "function_ref(F *) -> function_ref<F>;"))
    parts.append(grab(f"{CORE}/fwd.h", 192, 195, "template<typename ResultType,", "class uniform_random_bit_generator_ref;"))
    parts.append(grab(f"{CORE}/utilities/random.h", 37, 71, "class JIMINY_DLLAPI PCG32", "};"))
    parts.append(grab(f"{CORE}/utilities/random.h", 87, 106, "template<typename ResultType, ResultType min_, ResultType max_>", "};"))
    parts.append(grab(f"{CORE}/utilities/random.h", 213, 216, "float JIMINY_DLLAPI uniform(", "const uniform_random_bit_generator_ref<uint32_t> & g, float lo, float hi);"))
    parts.append(grab(f"{CORE}/utilities/random.h", 262, 264, "float JIMINY_DLLAPI normal(", "float stddev = 1.0F);"))
    parts.append(grab(f"{CORE}/utilities/random.hxx", 16, 57, "namespace internal", "}"))
    # PCG32, uniform, ziggurat normal, xxHash, MurmurHash3: the whole first part of random.cc
    parts.append(grab("core/src/utilities/random.cc", 8, 318, "// ***************************** Uniform random bit generators", "}"))
    # sub-step selection of Engine::step (engine.cc:2063-2089): stretch onto the breakpoint, snap to whole microseconds
    parts.append(grab(f"{CORE}/constants.h", 18, 20, "inline constexpr double STEPPER_MIN_TIMESTEP", "inline constexpr double SIMULATION_MAX_TIMESTEP"))
    parts.append("void substep_body(double & dt, const double t, const double tNext, const uint32_t successiveIterTooLarge)\n{\n")
    parts.append(grab("core/src/engine/engine.cc", 2063, 2089, "double dtResidualThr = STEPPER_MIN_TIMESTEP;", "}"))
    parts.append("}\n")
    # breakpoints of Engine::step: the end time with its Kahan compensation (engine.cc:1793-1795) and the time to the next
    # breakpoint of the discrete branch (engine.cc:1991-2018)
    parts.append("struct StepperStateTimes { double t; double tError; };\n"
                 "double end_time_body(StepperStateTimes & stepperState_, const double stepSize)\n{\n")
    parts.append(grab("core/src/engine/engine.cc", 1793, 1795, "const double stepSizeCorrected = stepSize - stepperState_.tError;",
                      "stepperState_.tError = (tEnd - stepperState_.t) - stepSizeCorrected;"))
    parts.append("return tEnd;\n}\n"
                 "void next_breakpoint_body(const double stepperUpdatePeriod_, const double t, const double tImpulseForceNext, "
                 "const double tEnd, double & tNext)\n{\n")
    parts.append(grab("core/src/engine/engine.cc", 1991, 2018, "double dtNextGlobal;  // dt to apply for the next stepper step", "tNext += dtNextGlobal;"))
    parts.append("}\n")
    # bookkeeping of the adaptive loop after a try (engine.cc:2166-2172 restore of the step size after a breakpoint, 2197-2208
    # failure counters and error recovery, 2221 size of the next try); data holders with the member names those lines read
    parts.append("namespace stepper\n{\n")
    parts.append(grab(f"{CORE}/stepper/abstract_stepper.h", 15, 20, "enum class ReturnCode : uint8_t", "};"))
    parts.append("struct StatusInfoHolder { ReturnCode returnCode; };\n}\n"
                 "struct StepperStateHolder { double dtLargestPrev; int64_t iterFailed; };\n"
                 "struct StepperOptsHolder { double dtRestoreThresholdRel; double dtMax; };\nstruct EngineOptsHolder { StepperOptsHolder stepper; };\n"
                 "void restore_body(const StepperStateHolder & stepperState_, const EngineOptsHolder * engineOptions_, const double dt, double & dtLargest)\n{\n")
    parts.append(grab("core/src/engine/engine.cc", 2166, 2172, "double dtRestoreThresholdAbs =", "}"))
    parts.append("}\nvoid failure_body(const stepper::StatusInfoHolder & status, StepperStateHolder & stepperState_, double & dtLargest, "
                 "uint32_t & successiveIterTooLarge, uint32_t & successiveIterFailed)\n{\n")
    parts.append(grab("core/src/engine/engine.cc", 2197, 2208, "if (status.returnCode == stepper::ReturnCode::IS_ERROR)", "++stepperState_.iterFailed;"))
    parts.append("}\nvoid next_dt_body(const EngineOptsHolder * engineOptions_, const double dtLargest, double & dt)\n{\n")
    parts.append(grab("core/src/engine/engine.cc", 2221, 2221, "dt = std::min(dtLargest, engineOptions_->stepper.dtMax);", "dt = std::min(dtLargest, engineOptions_->stepper.dtMax);"))
    parts.append("}\n")
    # when a profile force / the controller is refreshed (engine.cc:1903-1907, 1923-1927) and when an impulse force is active
    # (engine.cc:1857-1869); the structs are data holders with the member names those lines read
    parts.append("struct ProfileForceHolder { double updatePeriod; };\n"
                 "bool force_update_body(const ProfileForceHolder & profileForce, const double t)\n{\n")
    parts.append(grab("core/src/engine/engine.cc", 1903, 1907, "double forceUpdatePeriod = profileForce.updatePeriod;",
                      "forceUpdatePeriod - dtNextForceUpdatePeriod < STEPPER_MIN_TIMESTEP)"))
    parts.append("{ return true; }\nreturn false;\n}\n"
                 "struct StepperOptionsHolder { double controllerUpdatePeriod; };\nstruct EngineOptionsHolder { StepperOptionsHolder stepper; };\n"
                 "bool controller_update_body(const EngineOptionsHolder * engineOptions_, const double t)\n{\n")
    parts.append(grab("core/src/engine/engine.cc", 1923, 1927, "double controllerUpdatePeriod = engineOptions_->stepper.controllerUpdatePeriod;",
                      "controllerUpdatePeriod - dtNextControllerUpdatePeriod < STEPPER_MIN_TIMESTEP)"))
    parts.append("{ return true; }\nreturn false;\n}\n"
                 "struct ImpulseForceHolder { double t; double dt; };\n"
                 "void impulse_active_body(const ImpulseForceHolder * impulseForceIt, bool * isImpulseForceActiveIt, bool & hasDynamicsChanged, const double t)\n{\n")
    parts.append(grab("core/src/engine/engine.cc", 1857, 1869, "double tImpulseForce = impulseForceIt->t;", "}"))
    parts.append("}\n")
    # update-period arithmetic of Engine::setOptions / reset (helpers.hxx:59-116: minClipped, isGcdIncluded over doubles)
    parts.append(grab(f"{CORE}/utilities/helpers.hxx", 59, 116, "inline const double & minClipped()", "}"))
    # step-size controller
    parts.append("namespace DOPRI\n{\n")
    parts.append(grab(f"{CORE}/stepper/runge_kutta_dopri_stepper.h", 34, 47, "/// \\brief Stepper order", "inline constexpr double MAX_FACTOR"))
    parts.append("}\n")
    parts.append("bool adjustStep_body(const double error, double & dt)\n{\n")
    parts.append(grab("core/src/stepper/runge_kutta_dopri_stepper.cc", 24, 56, "// Make sure the error is well defined", "}"))
    # motor law: a data holder with the member names the body reads (SimpleMotorOptions is built from a
    # boost::variant map in the reference, basic_motors.h:14-60; AbstractMotorBase holds the two limits and `data()`)
    parts.append(r"""
struct bad_control_flow : std::logic_error { using std::logic_error::logic_error; };
struct SimpleMotorOptions
{
    double mechanicalReduction;
    bool enableEffortLimit, enableVelocityLimit;
    double velocityEffortInvSlope;
    bool enableFriction;
    double frictionViscousPositive, frictionViscousNegative, frictionDryPositive, frictionDryNegative, frictionDrySlope;
};
struct SimpleMotor
{
    bool isInitialized_ = true;
    std::unique_ptr<const SimpleMotorOptions> motorOptions_;
    double effortLimit_ = 0.0, velocityLimit_ = 0.0;
    double uMotor_ = 0.0, uTransmission_ = 0.0;
    std::tuple<double &, double &> data() { return {uMotor_, uTransmission_}; }
    void computeEffort(double /* t */, const double * /* q */, double v, double /* a */, double command);
};
void SimpleMotor::computeEffort(double /* t */, const double * /* q */, double v, double /* a */, double command)
""")
    parts.append(grab("core/src/hardware/basic_motors.cc", 88, 143, "{", "}"))
    parts.append("}  // namespace jiminy\n")
    parts.append(IO_HELPERS)
    parts.append(r"""
using namespace jiminy;
int main(int argc, char ** argv)
{
    if (argc != 3) return 2;
    io::fin = fopen(argv[1], "rb");
    io::fout = fopen(argv[2], "wb");
    if (!io::fin || !io::fout) return 2;
    // ---- raw draws, uniform, normal: one generator per seed state
    {
        const int64_t ns = io::geti(), nd = io::geti();
        const auto states = io::get<uint64_t>(ns);
        const auto lo = io::get<float>(ns), hi = io::get<float>(ns), mean = io::get<float>(ns), sd = io::get<float>(ns);
        std::vector<uint32_t> raw(ns * nd), after(ns * 3);
        std::vector<float> u01(ns * nd), ulh(ns * nd), nrm(ns * nd), nrm01(ns * nd);
        std::vector<uint32_t> after01(ns);
        for (int64_t s = 0; s < ns; ++s)
        {
            PCG32 g0(states[s]), g1(states[s]), g2(states[s]), g3(states[s]), g4(states[s]);
            for (int64_t k = 0; k < nd; ++k) nrm01[s * nd + k] = normal(g4);
            after01[s] = g4();
            for (int64_t k = 0; k < nd; ++k) raw[s * nd + k] = g0();
            for (int64_t k = 0; k < nd; ++k) u01[s * nd + k] = uniform(g1);
            for (int64_t k = 0; k < nd; ++k) ulh[s * nd + k] = uniform(g2, lo[s], hi[s]);
            for (int64_t k = 0; k < nd; ++k) nrm[s * nd + k] = normal(g3, mean[s], sd[s]);
            // the next raw draw pins how many draws each distribution consumed
            after[s * 3 + 0] = g1(); after[s * 3 + 1] = g2(); after[s * 3 + 2] = g3();
        }
        io::put(raw); io::put(u01); io::put(ulh); io::put(nrm); io::put(after); io::put(nrm01); io::put(after01);
    }
    // ---- PCG32(std::seed_seq): the state built by internal::generateState (two sets: 3-word and 1-word sequences)
    for (int rep = 0; rep < 2; ++rep)
    {
        const int64_t n = io::geti(), len = io::geti(), nd = io::geti();
        const auto words = io::get<uint32_t>(n * len);
        std::vector<uint32_t> raw(n * nd);
        for (int64_t s = 0; s < n; ++s)
        {
            // seeded the way Engine::reset does (engine.cc:757): `generator_.seed(std::seed_seq(first, last))`
            PCG32 g;
            g.seed(std::seed_seq(words.begin() + s * len, words.begin() + (s + 1) * len));
            for (int64_t k = 0; k < nd; ++k) raw[s * nd + k] = g();
        }
        io::put(raw);
    }
    // ---- ziggurat tables
    {
        using namespace jiminy::internal::ziggurat;
        io::put(std::vector<uint32_t>(kn.begin(), kn.end()));
        io::put(std::vector<float>(fn.begin(), fn.end()));
        io::put(std::vector<float>(wn.begin(), wn.end()));
    }
    // ---- hashes of byte strings
    {
        const int64_t n = io::geti(), stride = io::geti();
        const auto lens = io::get<int32_t>(n);
        const auto seeds = io::get<uint32_t>(n);
        const auto bytes = io::get<uint8_t>(n * stride);
        std::vector<uint32_t> xx(n), mm(n);
        for (int64_t i = 0; i < n; ++i)
        {
            // 4-byte aligned copy: the reference reads the key through uint32_t pointers
            alignas(8) uint8_t buf[256];
            std::memcpy(buf, bytes.data() + i * stride, static_cast<size_t>(stride));
            xx[i] = xxHash(buf, lens[i], seeds[i]);
            mm[i] = MurmurHash3(buf, lens[i], seeds[i]);
        }
        io::put(xx); io::put(mm);
    }
    // ---- DOPRI step-size controller
    {
        const int64_t n = io::geti();
        const auto err = io::get<double>(n), dt = io::get<double>(n);
        std::vector<double> dtOut(n);
        std::vector<int32_t> code(n);       // 1 accepted, 0 rejected, 2 threw (NaN error)
        for (int64_t i = 0; i < n; ++i)
        {
            double d = dt[i];
            try { code[i] = adjustStep_body(err[i], d) ? 1 : 0; }
            catch (const std::runtime_error &) { code[i] = 2; }
            dtOut[i] = d;
        }
        io::put(code); io::put(dtOut);
        io::put(std::vector<double>{DOPRI::STEPPER_ORDER, DOPRI::SAFETY, DOPRI::ERROR_THRESHOLD, DOPRI::MIN_FACTOR, DOPRI::MAX_FACTOR});
    }
    // ---- sub-step rule: one application per case, then whole intervals of a fixed-step solver (the loop around the rule --
    //      t += dt, dt = min(dtLargest = INF, dtMax) after every try, engine.cc:2136-2222 -- is this driver's)
    {
        const int64_t n = io::geti();
        const auto dt = io::get<double>(n), t = io::get<double>(n), tn = io::get<double>(n);
        const auto tl = io::get<int32_t>(n);
        std::vector<double> out(n);
        for (int64_t i = 0; i < n; ++i) { double d = dt[i]; substep_body(d, t[i], tn[i], static_cast<uint32_t>(tl[i])); out[i] = d; }
        io::put(out);
        const int64_t ni = io::geti();
        const auto iv = io::get<double>(ni), dmax = io::get<double>(ni), dfirst = io::get<double>(ni);
        std::vector<double> seq(ni * 64, 0.0);
        std::vector<int32_t> cnt(ni);
        for (int64_t i = 0; i < ni; ++i)
        {
            double tt = 0.0, d = dfirst[i];
            int32_t k = 0;
            while (iv[i] - tt > STEPPER_MIN_TIMESTEP && k < 64)
            {
                substep_body(d, tt, iv[i], 0U);
                seq[i * 64 + k++] = d;
                tt += d;
                d = dmax[i];
            }
            cnt[i] = k;
        }
        io::put(cnt); io::put(seq);
    }
    // ---- breakpoints of consecutive Engine::step calls.  The loop around the two reference bodies is this driver's: the next
    //      impulse breakpoint is the first one at least STEPPER_MIN_TIMESTEP ahead (engine.cc:1872-1889), the integration is
    //      taken to land on tNext exactly (t = tNext), the outer loop runs while tEnd - t >= STEPPER_MIN_TIMESTEP (:1838)
    {
        const int64_t n = io::geti();
        const auto period = io::get<double>(n), step = io::get<double>(n);
        const auto nsteps = io::get<int32_t>(n);
        const auto imp = io::get<double>(n * 4);     // up to four impulse breakpoints per case (INF: none)
        std::vector<double> tend(n * 64, 0.0), terr(n * 64, 0.0), bps(n * 512, 0.0);
        std::vector<int32_t> nbp(n, 0);
        for (int64_t i = 0; i < n; ++i)
        {
            StepperStateTimes st{0.0, 0.0};
            double tNext = 0.0;
            int32_t k = 0;
            for (int32_t s = 0; s < nsteps[i] && s < 64; ++s)
            {
                const double tEnd = end_time_body(st, step[i]);
                tend[i * 64 + s] = tEnd; terr[i * 64 + s] = st.tError;
                while (tEnd - st.t >= STEPPER_MIN_TIMESTEP && k < 512)
                {
                    double tImpulseForceNext = INF;
                    for (int j = 0; j < 4; ++j)
                        if (!(imp[i * 4 + j] - st.t < STEPPER_MIN_TIMESTEP)) tImpulseForceNext = std::min(tImpulseForceNext, imp[i * 4 + j]);
                    next_breakpoint_body(period[i], st.t, tImpulseForceNext, tEnd, tNext);
                    st.t = tNext;
                    bps[i * 512 + k++] = tNext;
                }
            }
            nbp[i] = k;
        }
        io::put(tend); io::put(terr); io::put(nbp); io::put(bps);
    }
    // ---- after a try of the adaptive loop.  Around the three reference bodies the driver restates the success branch's plain
    //      assignments (engine.cc:2141-2142 counters to zero, 2157 ++iter, 2160 `if (isBreakpointReached)`, 2184 dtLargestPrev)
    {
        const int64_t n = io::geti();
        const auto relmax = io::get<double>(2);
        const double rel = relmax[0], dtMax = relmax[1];
        const auto rc = io::get<int32_t>(n), bp = io::get<int32_t>(n);
        auto dt = io::get<double>(n), dtl = io::get<double>(n), dtlp = io::get<double>(n);
        auto cnt = io::get<int64_t>(n * 4);
        const EngineOptsHolder eo{{rel, dtMax}};
        for (int64_t i = 0; i < n; ++i)
        {
            StepperStateHolder st{dtlp[i], cnt[4 * i + 3]};
            uint32_t tooLarge = static_cast<uint32_t>(cnt[4 * i]), failed = static_cast<uint32_t>(cnt[4 * i + 1]);
            if (rc[i] == 0)
            {
                tooLarge = 0; failed = 0;
                ++cnt[4 * i + 2];
                if (bp[i]) restore_body(st, &eo, dt[i], dtl[i]);
                st.dtLargestPrev = dtl[i];
            }
            else
            {
                const stepper::StatusInfoHolder status{static_cast<stepper::ReturnCode>(rc[i])};
                failure_body(status, st, dtl[i], tooLarge, failed);
            }
            next_dt_body(&eo, dtl[i], dt[i]);
            dtlp[i] = st.dtLargestPrev;
            cnt[4 * i] = tooLarge; cnt[4 * i + 1] = failed; cnt[4 * i + 3] = st.iterFailed;
        }
        io::put(dt); io::put(dtl); io::put(dtlp); io::put(cnt);
    }
    // ---- refresh of profile forces / the controller at time t, activity of an impulse force carried over a time sequence
    {
        const int64_t n = io::geti();
        const auto period = io::get<double>(n), t = io::get<double>(n);
        std::vector<int32_t> fu(n), cu(n);
        for (int64_t i = 0; i < n; ++i)
        {
            fu[i] = force_update_body(ProfileForceHolder{period[i]}, t[i]) ? 1 : 0;
            const EngineOptionsHolder eo{{period[i]}};
            cu[i] = controller_update_body(&eo, t[i]) ? 1 : 0;
        }
        io::put(fu); io::put(cu);
        const int64_t m = io::geti(), nt = io::geti();
        const auto it = io::get<double>(m), idt = io::get<double>(m);
        const auto ts = io::get<double>(nt);
        std::vector<int32_t> act(m * nt), chg(m * nt);
        for (int64_t i = 0; i < m; ++i)
        {
            bool active = false;
            const ImpulseForceHolder f{it[i], idt[i]};
            for (int64_t k = 0; k < nt; ++k)
            {
                bool changed = false;
                impulse_active_body(&f, &active, changed, ts[k]);
                act[i * nt + k] = active ? 1 : 0; chg[i * nt + k] = changed ? 1 : 0;
            }
        }
        io::put(act); io::put(chg);
    }
    // ---- isGcdIncluded(sensorsUpdatePeriod, controllerUpdatePeriod) (engine.cc:749-750, 2699-2700)
    {
        const int64_t n = io::geti();
        const auto a = io::get<double>(n), b = io::get<double>(n);
        std::vector<int32_t> inc(n);
        std::vector<double> vmin(n);
        for (int64_t i = 0; i < n; ++i)
        {
            auto [isIncluded, valueMin] = isGcdIncluded(a[i], b[i]);
            inc[i] = isIncluded ? 1 : 0;
            vmin[i] = valueMin;
        }
        io::put(inc); io::put(vmin);
    }
    // ---- SimpleMotor::computeEffort
    {
        const int64_t n = io::geti();
        const auto p = io::get<double>(n * 14);
        std::vector<double> uMotor(n), uTrans(n);
        for (int64_t i = 0; i < n; ++i)
        {
            const double * c = p.data() + i * 14;
            SimpleMotor m;
            m.motorOptions_ = std::make_unique<const SimpleMotorOptions>(SimpleMotorOptions{
                c[0], c[1] != 0.0, c[2] != 0.0, c[3], c[6] != 0.0, c[7], c[8], c[9], c[10], c[11]});
            m.effortLimit_ = c[4];
            m.velocityLimit_ = c[5];
            m.computeEffort(0.0, nullptr, c[12], 0.0, c[13]);
            uMotor[i] = m.uMotor_;
            uTrans[i] = m.uTransmission_;
        }
        io::put(uMotor); io::put(uTrans);
    }
    fclose(io::fout);
    return 0;
}
""")
    return "".join(parts)


def tu_tier_b() -> str:
    """Reference text against tools/ref_cpp/mini_linalg.h (a stand-in for Eigen: NOT a reference build)."""
    parts = [STD_HEADERS, '#include "mini_linalg.h"\n', "namespace jiminy\n{\n"]
    parts.append(grab(f"{CORE}/constants.h", 24, 26, "inline constexpr double INF", "inline constexpr double qNAN"))
    # ---- contact law
    parts.append(r"""
struct ContactOptions { double stiffness, damping, friction, torsion, transitionEps, transitionVelocity; };
struct EngineOptions { ContactOptions contacts; };
struct Engine
{
    std::unique_ptr<const EngineOptions> engineOptions_;
    pinocchio::Force computeContactDynamics(const Eigen::Vector3d & normalGround, double depth,
                                            const Eigen::Vector3d & vContactInWorld) const;
};
""")
    parts.append(grab("core/src/engine/engine.cc", 3197, 3238, "pinocchio::Force Engine::computeContactDynamics(", "}"))
    # ---- Butcher tableaux
    parts.append("namespace RK4\n{\n")
    parts.append(grab(f"{CORE}/stepper/runge_kutta4_stepper.h", 12, 23, "const Eigen::MatrixXd A(", ").finished());"))
    parts.append("}\nnamespace DOPRI\n{\n")
    parts.append(grab(f"{CORE}/stepper/runge_kutta_dopri_stepper.h", 12, 29, "const Eigen::MatrixXd A(", ").finished());"))
    parts.append("}\n")
    # ---- PGS: block table (constructor switch), sweep, solver loop
    parts.append(grab("core/include/jiminy/core/robot/model.h", 32, 38, "enum class ConstraintRegistryType", "};"))
    parts.append("class AbstractConstraintBase;\n")
    parts.append(grab(f"{CORE}/solver/constraint_solvers.h", 12, 29, "struct ConstraintBlock", "};"))
    parts.append(grab("core/src/solver/constraint_solvers.cc", 15, 21, "inline constexpr double MIN_REGULARIZER", "inline constexpr double RELAX_SLOPE_ORDER"))
    parts.append(r"""
struct PGSSolver
{
    uint32_t iterMax_;
    double tolAbs_, tolRel_;
    std::vector<ConstraintData> constraintsData_{};
    Eigen::VectorXd y_{}, yPrev_{};
    void addConstraint(ConstraintRegistryType type, Eigen::Index constraintSize, double friction, double torsion);
    void ProjectedGaussSeidelIter(const Eigen::MatrixXd & A, const Eigen::VectorXd::SegmentReturnType & b,
                                  const double w, Eigen::VectorXd::SegmentReturnType & x);
    bool ProjectedGaussSeidelSolver(const Eigen::MatrixXd & A, const Eigen::VectorXd::SegmentReturnType & b,
                                    Eigen::VectorXd::SegmentReturnType & x);
};
void PGSSolver::addConstraint(ConstraintRegistryType type, Eigen::Index constraintSize, double friction, double torsion)
{
""")
    parts.append(grab("core/src/solver/constraint_solvers.cc", 45, 91, "ConstraintBlock block{};", "}"))
    parts.append(r"""
    constraintData.dim = constraintSize;
    constraintsData_.emplace_back(std::move(constraintData));
}
""")
    parts.append(grab("core/src/solver/constraint_solvers.cc", 107, 222, "void PGSSolver::ProjectedGaussSeidelIter(", "}"))
    parts.append(grab("core/src/solver/constraint_solvers.cc", 224, 326, "bool PGSSolver::ProjectedGaussSeidelSolver(", "}"))
    parts.append("}  // namespace jiminy\n")
    parts.append(IO_HELPERS)
    parts.append(r"""
using namespace jiminy;
int main(int argc, char ** argv)
{
    if (argc != 3) return 2;
    io::fin = fopen(argv[1], "rb");
    io::fout = fopen(argv[2], "wb");
    if (!io::fin || !io::fout) return 2;
    // ---- contact law: per case [stiffness damping friction transitionEps transitionVelocity | n(3) depth v(3)]
    {
        const int64_t n = io::geti();
        const auto p = io::get<double>(n * 12);
        std::vector<double> f(n * 6);
        for (int64_t i = 0; i < n; ++i)
        {
            const double * c = p.data() + i * 12;
            Engine e;
            e.engineOptions_ = std::make_unique<const EngineOptions>(EngineOptions{ContactOptions{c[0], c[1], c[2], 0.0, c[3], c[4]}});
            const pinocchio::Force w = e.computeContactDynamics({c[5], c[6], c[7]}, c[8], {c[9], c[10], c[11]});
            for (int k = 0; k < 3; ++k) { f[i * 6 + k] = w.linear()[k]; f[i * 6 + 3 + k] = w.angular()[k]; }
        }
        io::put(f);
    }
    // ---- tableaux, row major
    {
        std::vector<double> t;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) t.push_back(RK4::A(i, j));
        for (int i = 0; i < 4; ++i) t.push_back(RK4::c[i]);
        for (int i = 0; i < 4; ++i) t.push_back(RK4::b[i]);
        for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) t.push_back(DOPRI::A(i, j));
        for (int i = 0; i < 7; ++i) t.push_back(DOPRI::c[i]);
        for (int i = 0; i < 7; ++i) t.push_back(DOPRI::b[i]);
        for (int i = 0; i < 7; ++i) t.push_back(DOPRI::e[i]);
        io::put(t);
    }
    // ---- PGS: per problem a list of constraints (type, dim), a symmetric A (column major), b, x0;
    //      out: x after ONE sweep at each of three relaxation factors, then the solver's x, flag and residuals
    {
        const int64_t np = io::geti();
        for (int64_t pb = 0; pb < np; ++pb)
        {
            const int64_t nc = io::geti(), n = io::geti(), iterMax = io::geti();
            const auto types = io::get<int32_t>(nc), dims = io::get<int32_t>(nc);
            const auto prm = io::get<double>(4);    // friction, torsion, tolAbs, tolRel
            const auto Av = io::get<double>(n * n), bv = io::get<double>(n), x0 = io::get<double>(n);
            const auto ws = io::get<double>(3);
            PGSSolver s;
            s.iterMax_ = static_cast<uint32_t>(iterMax);
            s.tolAbs_ = prm[2];
            s.tolRel_ = prm[3];
            Eigen::Index row = 0;
            for (int64_t c = 0; c < nc; ++c)
            {
                s.addConstraint(static_cast<ConstraintRegistryType>(types[c]), dims[c], prm[0], prm[1]);
                s.constraintsData_.back().startIndex = row;     // SolveBoxedForwardDynamics, constraint_solvers.cc:347-368
                row += dims[c];
            }
            s.y_.resize(n);
            s.yPrev_.resize(n);
            Eigen::MatrixXd A(n, n);
            A.v = Av;
            Eigen::VectorXd b(n), x(n);
            b.v = bv;
            auto bs = b.head(n);
            for (int k = 0; k < 3; ++k)
            {
                x.v = x0;
                s.y_.setZero();
                auto xs = x.head(n);
                s.ProjectedGaussSeidelIter(A, bs, ws[k], xs);
                io::put(x.v);
                io::put(s.y_.v);
            }
            x.v = x0;
            auto xs = x.head(n);
            const bool ok = s.ProjectedGaussSeidelSolver(A, bs, xs);
            io::put(x.v);
            io::put(s.y_.v);
            io::put(std::vector<int32_t>{ok ? 1 : 0});
        }
    }
    fclose(io::fout);
    return 0;
}
""")
    return "".join(parts)


class Blob:
    """Little-endian input stream for the drivers."""

    def __init__(self) -> None:
        self.parts: list[bytes] = []

    def i(self, *vals: int) -> None:
        self.parts.append(struct.pack(f"<{len(vals)}q", *vals))

    def a(self, arr: np.ndarray, dtype) -> None:
        self.parts.append(np.ascontiguousarray(arr, dtype=dtype).tobytes())

    def write(self, path: str) -> None:
        with open(path, "wb") as f:
            f.write(b"".join(self.parts))


class Reader:
    def __init__(self, path: str) -> None:
        with open(path, "rb") as f:
            self.buf = f.read()
        self.off = 0

    def take(self, dtype, *shape: int) -> np.ndarray:
        n = int(np.prod(shape)) if shape else 1
        dt = np.dtype(dtype)
        out = np.frombuffer(self.buf, dtype=dt, count=n, offset=self.off).reshape(shape).copy()
        self.off += n * dt.itemsize
        return out

    def done(self) -> None:
        if self.off != len(self.buf):
            raise RuntimeError(f"driver wrote {len(self.buf)} bytes, {self.off} consumed")


def compile_and_run(name: str, source: str, blob: Blob) -> Reader:
    os.makedirs(WORK, exist_ok=True)
    src, exe = os.path.join(WORK, f"{name}.cpp"), os.path.join(WORK, name)
    fin, fout = os.path.join(WORK, f"{name}.in"), os.path.join(WORK, f"{name}.out")
    with open(src, "w") as f:
        f.write(source)
    # no fast-math, no contraction: the arithmetic of the text as written
    cmd = ["g++", "-std=c++17", "-O2", "-fno-fast-math", "-ffp-contract=off", "-Wno-unused-parameter",
           "-I", os.path.join(HERE, "ref_cpp"), "-o", exe, src]
    subprocess.run(cmd, check=True)
    blob.write(fin)
    subprocess.run([exe, fin, fout], check=True)
    return Reader(fout)


def spd_delassus(rg: np.random.Generator, n: int, nv: int) -> np.ndarray:
    """A = J M^-1 J^T + damping: the shape of the matrix the reference hands to its PGS."""
    J = rg.standard_normal((n, nv))
    L = rg.standard_normal((nv, nv)) * 0.3 + np.eye(nv) * 1.5
    Minv = np.linalg.inv(L @ L.T)
    A = J @ Minv @ J.T
    A = 0.5 * (A + A.T)
    A[np.diag_indices(n)] += 1e-3 * np.diag(A).max()
    return A


def main(out_path: str = OUT) -> None:
    if not os.path.isdir(REF):
        sys.exit(f"{REF} not found: this generator runs in the build container only")
    rg = np.random.default_rng(20261001)
    out: dict = {}

    # ============================================================ tier A
    blob = Blob()
    ns, nd = 8, 4096
    states = np.array([0xcafef00dd15ea5e5, 0, 1, 3, 0xffffffffffffffff, 0x0123456789abcdef,
                       *rg.integers(0, 2**63, 2, dtype=np.uint64)], dtype=np.uint64)
    lo = np.array([0, -1, 2, -5, 0, 1e-3, -1e6, 3], dtype=np.float32)
    hi = np.array([1, 1, 2.5, 5, 1e-30, 1e3, 1e6, 3.0000002], dtype=np.float32)
    mean = np.array([0, 1, -2, 0, 0, 10, 0, 1e-3], dtype=np.float32)
    sd = np.array([1, 0.5, 3, 1e-3, 0, 100, 1, 1e-6], dtype=np.float32)
    blob.i(ns, nd)
    blob.a(states, np.uint64)
    for arr in (lo, hi, mean, sd):
        blob.a(arr, np.float32)
    nss, sslen, ssd = 6, 3, 16
    sswords = rg.integers(0, 2**32, (nss, sslen), dtype=np.uint64).astype(np.uint32)
    sswords[0] = (0, 0, 0)
    sswords[1] = (1, 0, 0)
    blob.i(nss, sslen, ssd)
    blob.a(sswords, np.uint32)
    ss1 = np.array([0, 1, 2, 3, 42, 0xDEADBEEF, 0xFFFFFFFF, *rg.integers(0, 2**32, 9, dtype=np.uint64)], dtype=np.uint32)[:, None]
    blob.i(len(ss1), 1, ssd)
    blob.a(ss1, np.uint32)
    nk, stride = 160, 64
    lens = np.concatenate([np.arange(0, 49), rg.integers(0, stride + 1, nk - 49)]).astype(np.int32)
    hseeds = rg.integers(0, 2**32, nk, dtype=np.uint64).astype(np.uint32)
    hseeds[:8] = 0
    keys = rg.integers(0, 256, (nk, stride), dtype=np.uint64).astype(np.uint8)
    blob.i(nk, stride)
    blob.a(lens, np.int32)
    blob.a(hseeds, np.uint32)
    blob.a(keys, np.uint8)
    # controller: errors across every branch (grow, hold, reject, the clipping of both factors, NaN, 0, inf)
    err = np.concatenate([[0.0, 1e-300, 1e-12, 1e-6, 3.2e-4, 0.1, 0.32, 0.32768, 0.33, 0.4999, 0.5, 0.51, 0.999999, 1.0,
                           1.0000001, 1.5, 10.0, 1e3, 1e9, 1e300, np.inf, np.nan],
                          10.0 ** rg.uniform(-8, 4, 106)])
    dts = np.concatenate([[1e-3] * 22, 10.0 ** rg.uniform(-9, -1.5, 106)])
    blob.i(len(err))
    blob.a(err, np.float64)
    blob.a(dts, np.float64)
    # sub-step rule: (dt, t, tNext, successiveIterTooLarge) across every branch, then whole intervals (interval, dtMax, first dt)
    nsr = 400
    sr_dt = 10.0 ** rg.uniform(-7.5, -1.7, nsr)
    sr_t = rg.uniform(0.0, 2.0, nsr)
    gap = np.where(rg.random(nsr) < 0.5, sr_dt * rg.uniform(0.3, 1.3, nsr), sr_dt * rg.uniform(1.0, 40.0, nsr))
    gap[:40] = sr_dt[:40] + 10.0 ** rg.uniform(-11, -6.2, 40)          # just beyond the step: the residual-merge band
    sr_dt[40:60] = np.round(sr_dt[40:60], 6) + rg.uniform(0, 1e-6, 20)  # around whole microseconds
    sr_tn = sr_t + gap
    sr_tl = rg.integers(0, 3, nsr).astype(np.int32)
    blob.i(nsr)
    for arr in (sr_dt, sr_t, sr_tn):
        blob.a(arr, np.float64)
    blob.a(sr_tl, np.int32)
    iv = np.array([1e-3, 1e-3, 1e-3, 5e-3, 5e-3, 1e-3, 1.05e-6, 1.2e-6, 4e-2, 1e-3, 2.5e-4, 1e-3, 7.77e-4, 1e-2], dtype=np.float64)
    ivmax = np.array([1e-3, 1e-3, 1 / 3e3, 1e-3, 7e-4, 2e-2, 1e-3, 1e-3, 5e-3, 1e-4 + 3e-8, 1e-3, 9.999e-4, 1e-4, 2e-2])
    ivfirst = np.array([1e-6, 1e-3, 1 / 3e3, 1e-6, 7e-4, 1e-6, 1e-6, 1e-6, 5e-3, 1e-6, 1e-6, 9.999e-4, 1e-4, 1e-6])
    blob.i(len(iv))
    for arr in (iv, ivmax, ivfirst):
        blob.a(arr, np.float64)
    # breakpoints: (update period, step size, number of consecutive steps, impulse breakpoints)
    bk = [(1e-3, 1e-3, 60, ()), (1e-3, 5e-3, 40, ()), (5e-3, 1e-3, 60, ()), (1e-3, 1e-3, 40, (2.5e-3, 7.2e-3)), (2e-3, 0.04, 10, (0.013, 0.0131)),
          (1e-3, 2.5e-3, 40, ()), (1 / 3e3, 1e-3, 60, ()), (1e-3, 1 / 3e2, 30, ()), (7e-4, 2e-3, 50, (3e-3 + 5e-11,)), (1e-2, 1e-2, 64, ()),
          (1e-3, 1.0000005e-3, 64, ()), (5e-3, 0.04, 12, (0.02, 0.06, 0.1, 0.1000000001)), (4e-3, 0.02, 25, ()), (1e-3, 3.7e-3, 40, (1.85e-3,))]
    blob.i(len(bk))
    blob.a(np.array([b[0] for b in bk]), np.float64); blob.a(np.array([b[1] for b in bk]), np.float64)
    blob.a(np.array([b[2] for b in bk]), np.int32)
    bk_imp = np.full((len(bk), 4), np.inf)
    for i, b in enumerate(bk):
        bk_imp[i, :len(b[3])] = b[3]
    blob.a(bk_imp, np.float64)
    # after a try: (return code, breakpoint reached, dt tried, dtLargest the stepper returned, dtLargestPrev, counters)
    nat = 300
    at_rc = rg.choice([0, 0, 0, 1, 2], nat).astype(np.int32)
    at_bp = (rg.random(nat) < 0.5).astype(np.int32)
    at_dt = 10.0 ** rg.uniform(-6, -2, nat)
    at_dtl = at_dt * 10.0 ** rg.uniform(-0.5, 1.5, nat)
    at_dtl[::9] = np.inf                                               # (fixed-step steppers return INF)
    at_dtlp = at_dtl * 10.0 ** rg.uniform(-0.3, 1.7, nat)
    at_dtlp[::9] = 10.0 ** rg.uniform(-4, -1, len(at_dtlp[::9]))
    at_cnt = np.stack([rg.integers(0, 3, nat), rg.integers(0, 5, nat), rg.integers(0, 1000, nat), rg.integers(0, 50, nat)], axis=1).astype(np.int64)
    blob.i(nat)
    blob.a(np.array([0.2, 0.02]), np.float64)     # dtRestoreThresholdRel, dtMax (engine.h defaults)
    blob.a(at_rc, np.int32); blob.a(at_bp, np.int32)
    for arr in (at_dt, at_dtl, at_dtlp):
        blob.a(arr, np.float64)
    blob.a(at_cnt, np.int64)
    # refresh rule: times on, just before, just after and between multiples of the period
    up_p = np.repeat(np.array([1e-3, 5e-3, 1e-2, 1 / 3e2, 7e-4, 2.5e-3]), 60)
    kk = np.tile(np.repeat(np.arange(1, 13), 5), 6).astype(np.float64)
    off = np.tile(np.array([0.0, -5e-7, -2e-6, 5e-11, 3e-4]), 72)
    up_t = kk * up_p + np.minimum(off, 0.4 * up_p)
    up_t[::7] = (kk * up_p)[::7]
    blob.i(len(up_p))
    blob.a(up_p, np.float64); blob.a(up_t, np.float64)
    imp_t = np.array([0.0, 2.0, 2.0, 1.9, 1e-3, 0.5])
    imp_dt = np.array([1e-2, 1e-2, 1e-10, 0.2, 1e-3, 5e-11])
    imp_ts = np.concatenate([np.arange(0, 40) * 1e-3, 2.0 + np.array([-1e-3, -1e-10, -5e-11, 0.0, 5e-11, 5e-3, 1e-2 - 1e-10, 1e-2 - 5e-11, 1e-2, 2e-2]),
                             [0.5 - 5e-11, 0.5, 0.5 + 1e-10, 1.9, 2.1 - 1e-10, 2.1, 3.0]])
    imp_ts = np.sort(imp_ts)
    blob.i(len(imp_t)); blob.i(len(imp_ts))
    blob.a(imp_t, np.float64); blob.a(imp_dt, np.float64); blob.a(imp_ts, np.float64)
    # update periods: multiples that are exact in binary, multiples that are not (0.009 / 0.003: fmod leaves 0.003 - 1 ulp and
    # the reference refuses the pair), non-multiples, zeros (continuous mode), values around EPS
    base = np.array([1e-3, 5e-4, 2.5e-3, 3e-3, 4e-3, 1e-2, 7e-4, 1.1e-3, 1e-6, 2e-2])
    mult = np.arange(1, 13, dtype=np.float64)
    gp_a = np.concatenate([np.repeat(base, len(mult)) * np.tile(mult, len(base)), [0.0, 0.0, 1e-3, 5e-3, 1e-17, 1e-3, 2.5e-3, 1.5e-3]])
    gp_b = np.concatenate([np.repeat(base, len(mult)), [0.0, 1e-3, 0.0, 2e-3, 1e-3, 1e-3 + 1e-11, 1e-3, 1e-3]])
    swap = rg.random(len(gp_a)) < 0.5
    gp_a, gp_b = np.where(swap, gp_b, gp_a), np.where(swap, gp_a, gp_b)
    blob.i(len(gp_a))
    blob.a(gp_a, np.float64); blob.a(gp_b, np.float64)
    # motors: [red, effLimOn, velLimOn, invSlope, effortLimit, velocityLimit, fricOn, fvp, fvn, fdp, fdn, fds, v, command]
    # MOTOR_GROUP rows share one parameter set (a motor's options are model constants: the device test builds one
    # model per group); group 0 = ANYmal's shipped motor (anymal_hardware.toml:7-11, URDF effort 80 / velocity 7.5)
    ng, gsz = 10, MOTOR_GROUP
    nm = ng * gsz
    gp = np.zeros((ng, 12))
    gp[:, 0] = np.where(rg.random(ng) < 0.5, 1.0, rg.uniform(0.5, 120.0, ng))
    gp[:, 1] = 1
    gp[:, 2] = 1
    gp[:, 3] = rg.uniform(0.0, 0.2, ng)
    gp[:, 4] = rg.uniform(1.0, 200.0, ng)
    gp[:, 5] = rg.uniform(0.5, 30.0, ng)
    gp[:, 6] = rg.random(ng) < 0.5
    gp[:, 7:9] = -rg.uniform(0.0, 2.0, (ng, 2))
    gp[:, 9:11] = -rg.uniform(0.0, 5.0, (ng, 2))
    gp[:, 11] = rg.uniform(0.5, 50.0, ng)
    gp[0] = (1.0, 1, 1, 0.02, 80.0, 7.5, 0, 0, 0, 0, 0, 1)
    gp[1, 1:3] = (0, 0)                 # no limit at all
    gp[2, 1:3] = (1, 0)                 # effort limit only
    gp[3, 1:3] = (0, 1)                 # a velocity limit without the effort limit is ignored (basic_motors.cc:104-119)
    gp[4, 3] = 0.0                      # zero slope: velocityDelta = 0, the velocity branch is skipped (:111)
    gp[5, 3], gp[5, 4], gp[5, 5] = 0.2, 150.0, 5.0     # velocityThr clipped at 0 (:113)
    gp[6, 6], gp[7, 6] = 1, 1           # friction on
    mp = np.zeros((nm, 14))
    mp[:, :12] = np.repeat(gp, gsz, axis=0)
    mp[:, 12] = rg.standard_normal(nm) * np.repeat(gp[:, 5] / gp[:, 0], gsz) * 0.8
    mp[:, 13] = rg.standard_normal(nm) * np.repeat(gp[:, 4], gsz) * 1.5
    mp[::gsz, 12], mp[1::gsz, 12], mp[2::gsz, 12], mp[3::gsz, 12] = 0.0, -0.0, 1e-12, -1e-12
    blob.i(nm)
    blob.a(mp, np.float64)

    rd = compile_and_run("ref_leaves_a", tu_tier_a(), blob)
    out.update(pcg_state=states, uniform_lo=lo, uniform_hi=hi, normal_mean=mean, normal_std=sd,
               pcg_raw=rd.take(np.uint32, ns, nd), uniform01=rd.take(np.float32, ns, nd),
               uniform_lohi=rd.take(np.float32, ns, nd), normal=rd.take(np.float32, ns, nd),
               next_raw_after=rd.take(np.uint32, ns, 3), normal01=rd.take(np.float32, ns, nd),
               next_raw_after_normal01=rd.take(np.uint32, ns))
    out.update(seedseq_words=sswords, seedseq_raw=rd.take(np.uint32, nss, ssd))
    out.update(seedseq1_words=ss1, seedseq1_raw=rd.take(np.uint32, len(ss1), ssd))
    out.update(zig_kn=rd.take(np.uint32, 128), zig_fn=rd.take(np.float32, 128), zig_wn=rd.take(np.float32, 128))
    out.update(hash_len=lens, hash_seed=hseeds, hash_key=keys, xxhash=rd.take(np.uint32, nk), murmur3=rd.take(np.uint32, nk))
    out.update(dopri_err=err, dopri_dt=dts, dopri_code=rd.take(np.int32, len(err)), dopri_dt_out=rd.take(np.float64, len(err)),
               dopri_constants=rd.take(np.float64, 5))
    out.update(substep_dt=sr_dt, substep_t=sr_t, substep_tnext=sr_tn, substep_too_large=sr_tl, substep_dt_out=rd.take(np.float64, nsr))
    out.update(interval=iv, interval_dt_max=ivmax, interval_dt_first=ivfirst, interval_count=rd.take(np.int32, len(iv)),
               interval_sizes=rd.take(np.float64, len(iv), 64))
    out.update(bp_period=np.array([b[0] for b in bk]), bp_step=np.array([b[1] for b in bk]), bp_nsteps=np.array([b[2] for b in bk], dtype=np.int32),
               bp_impulse=bk_imp, bp_t_end=rd.take(np.float64, len(bk), 64), bp_t_error=rd.take(np.float64, len(bk), 64),
               bp_count=rd.take(np.int32, len(bk)), bp_times=rd.take(np.float64, len(bk), 512))
    out.update(after_try_rc=at_rc, after_try_bp=at_bp, after_try_dt=at_dt, after_try_dt_largest=at_dtl, after_try_dt_largest_prev=at_dtlp,
               after_try_counters=at_cnt, after_try_dt_out=rd.take(np.float64, nat), after_try_dt_largest_out=rd.take(np.float64, nat),
               after_try_dt_largest_prev_out=rd.take(np.float64, nat), after_try_counters_out=rd.take(np.int64, nat, 4))
    out.update(update_period=up_p, update_t=up_t, update_force=rd.take(np.int32, len(up_p)), update_controller=rd.take(np.int32, len(up_p)))
    out.update(impulse_t=imp_t, impulse_dt=imp_dt, impulse_times=imp_ts, impulse_active=rd.take(np.int32, len(imp_t), len(imp_ts)),
               impulse_changed=rd.take(np.int32, len(imp_t), len(imp_ts)))
    out.update(period_a=gp_a, period_b=gp_b, period_included=rd.take(np.int32, len(gp_a)), period_min=rd.take(np.float64, len(gp_a)))
    out.update(motor_group=np.array(MOTOR_GROUP), motor_params=mp, motor_u=rd.take(np.float64, nm), motor_u_transmission=rd.take(np.float64, nm))
    rd.done()
    for group in ("pcg", "uniform", "normal", "seedseq", "zig", "hash", "xxhash", "murmur3", "dopri", "motor", "substep", "interval", "period", "bp", "update", "impulse", "after"):
        out[f"tier__{group}"] = np.array("A")

    # ============================================================ tier B
    blob = Blob()
    ncl = 768
    cp = np.zeros((ncl, 12))
    cp[:, 0] = 10.0 ** rg.uniform(4, 7, ncl)
    cp[:, 1] = 10.0 ** rg.uniform(1, 4, ncl)
    cp[:, 2] = rg.uniform(0.0, 2.0, ncl)
    cp[:, 3] = np.where(rg.random(ncl) < 0.15, 0.0, 10.0 ** rg.uniform(-4, -1, ncl))
    cp[:, 4] = 10.0 ** rg.uniform(-3, -1, ncl)
    nrm = rg.standard_normal((ncl, 3)) * (0.3, 0.3, 1.0) + (0, 0, 1.5)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    cp[:, 5:8] = nrm
    cp[:, 8] = np.where(rg.random(ncl) < 0.12, rg.uniform(0, 0.01, ncl), -(10.0 ** rg.uniform(-7, -1.5, ncl)))
    vel = rg.standard_normal((ncl, 3)) * (10.0 ** rg.uniform(-4, 0.5, (ncl, 1)))
    cp[:, 9:12] = vel
    # the first CONTACT_FLAT_GROUPS * CONTACT_GROUP rows: flat ground (n = z) and ONE option set per group of
    # CONTACT_GROUP rows (contact options belong to the engine, not to the lane: the device test runs one batch per
    # group).  Group 0 = the engine's defaults (engine.h:276-283); group 1 = no blending (transitionEps = 0);
    # group 2 = frictionless.  Rows 0-3 of every group: zero depth, minus zero depth, zero velocity, pure sliding.
    nflat = CONTACT_FLAT_GROUPS * CONTACT_GROUP
    cp[:nflat, 5:8] = (0.0, 0.0, 1.0)
    gopt = cp[:nflat:CONTACT_GROUP, :5].copy()
    gopt[0] = (1e6, 2e3, 1.0, 1e-3, 1e-2)
    gopt[1, 3] = 0.0
    gopt[2, 2] = 0.0
    gopt[3:, 3] = np.maximum(gopt[3:, 3], 1e-4)
    cp[:nflat, :5] = np.repeat(gopt, CONTACT_GROUP, axis=0)
    cp[0:nflat:CONTACT_GROUP, 8], cp[1:nflat:CONTACT_GROUP, 8] = 0.0, -0.0
    cp[2:nflat:CONTACT_GROUP, 9:12], cp[3:nflat:CONTACT_GROUP, 9:12] = (0, 0, 0), (0.3, -0.2, 0.0)
    cp[2:nflat:CONTACT_GROUP, 8] = cp[3:nflat:CONTACT_GROUP, 8] = -1e-3
    blob.i(ncl)
    blob.a(cp, np.float64)
    # PGS problems.  types: 0 contact frame (dim 4: x, y, z, torsion), 2 joint bound (dim 1), 3 user (unbounded, dim 1..6)
    problems = []
    layouts = [
        ([0], [4]), ([2], [1]), ([3], [3]), ([0, 0, 0, 0], [4] * 4), ([2, 2, 0, 0], [1, 1, 4, 4]),
        ([3, 0, 2, 0, 3], [6, 4, 1, 4, 2]), ([0] * 8, [4] * 8), ([2] * 5 + [0] * 4 + [3], [1] * 5 + [4] * 4 + [3]),
    ]
    for k, (types, dims) in enumerate(layouts * 2):
        n = int(sum(dims))
        friction = (1.0, 0.5, 0.0, 1.0)[k % 4]
        torsion = (0.0, 0.1, 0.0, 0.0)[k % 4]
        A = spd_delassus(rg, n, max(6, n // 2 + 3))
        b = rg.standard_normal(n) * 5.0
        x0 = rg.standard_normal(n) * (k >= len(layouts))     # second half warm-started like the engine's lambda
        problems.append(dict(types=np.array(types, np.int32), dims=np.array(dims, np.int32), iter_max=100,
                             prm=np.array([friction, torsion, 1e-8, 1e-6] if k % 2 else [friction, torsion, 1e-6, 1e-4]),
                             A=A, b=b, x0=x0, w=np.array([1.0, 0.3, 0.01])))
    blob.i(len(problems))
    for pb in problems:
        blob.i(len(pb["types"]), len(pb["b"]), pb["iter_max"])
        blob.a(pb["types"], np.int32)
        blob.a(pb["dims"], np.int32)
        blob.a(pb["prm"], np.float64)
        blob.a(pb["A"].T, np.float64)            # column major
        blob.a(pb["b"], np.float64)
        blob.a(pb["x0"], np.float64)
        blob.a(pb["w"], np.float64)

    rd = compile_and_run("ref_leaves_b", tu_tier_b(), blob)
    out.update(contact_group=np.array(CONTACT_GROUP), contact_flat_groups=np.array(CONTACT_FLAT_GROUPS), contact_params=cp, contact_force=rd.take(np.float64, ncl, 6))
    tab = rd.take(np.float64, 16 + 8 + 49 + 21)
    out.update(rk4_A=tab[:16].reshape(4, 4), rk4_c=tab[16:20], rk4_b=tab[20:24], dopri_A=tab[24:73].reshape(7, 7),
               dopri_c=tab[73:80], dopri_b=tab[80:87], dopri_e=tab[87:94])
    out["pgs_count"] = np.array(len(problems))
    for k, pb in enumerate(problems):
        n = len(pb["b"])
        for key in ("types", "dims", "prm", "A", "b", "x0", "w"):
            out[f"pgs{k}_{key}"] = pb[key]
        out[f"pgs{k}_iter_max"] = np.array(pb["iter_max"])
        sweeps = np.zeros((3, 2, n))
        for j in range(3):
            sweeps[j, 0] = rd.take(np.float64, n)
            sweeps[j, 1] = rd.take(np.float64, n)
        out[f"pgs{k}_sweep_x_y"] = sweeps
        out[f"pgs{k}_solve_x"] = rd.take(np.float64, n)
        out[f"pgs{k}_solve_y"] = rd.take(np.float64, n)
        out[f"pgs{k}_solve_ok"] = rd.take(np.int32, 1)
    rd.done()
    for group in ("contact", "rk4", "dopri_tableau", "pgs"):
        out[f"tier__{group}"] = np.array("B")

    np.savez_compressed(out_path, **out)
    print(f"wrote {os.path.relpath(out_path, ROOT)}: {len(out)} arrays, {os.path.getsize(out_path)} bytes")


if __name__ == "__main__":
    main()
