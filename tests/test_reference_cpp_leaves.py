"""The oracle and the HIP kernels against outputs of the REFERENCE'S OWN C++ TEXT (tests/golden/ref_cpp_leaves.npz).

The fixtures are made by tools/make_ref_cpp_fixtures.py, which cuts the leaf functions of the hot path out of
/root/reference by line range, compiles them with g++ and runs them on seeded inputs (nothing of the reference is
committed).  Two tiers (the `tier__*` entries of the file):

  A  reference text + standard headers only ("reference-compiled"): PCG32 and its seeding, uniform, the ziggurat
     normal and its tables, xxHash, the DOPRI step-size controller, SimpleMotor::computeEffort.
  B  reference text on tools/ref_cpp/mini_linalg.h, a stand-in for the Eigen members those bodies use -- by the rules of
     this build NOT a reference build, a second reading of the same lines: the contact law, the PGS block table /
     sweep / solver loop, the RK4 and DOPRI tableaux.

CPU tests hold the oracle (oracle.cpp, oracle_random.cpp, oracle/terrain_numpy.py) and the host-side tensor programs
against them; `-m gpu` tests hold the HIP kernels against them through the C ABI.  Integer streams bit-exact; float
laws at the tolerance written next to each assertion.
"""
from __future__ import annotations

import ctypes as C
import os
import struct

import numpy as np
import pytest

from oracle import oracle_py
from oracle import terrain_numpy

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = np.load(os.path.join(HERE, "golden", "ref_cpp_leaves.npz"))
REF = os.environ.get("JIMINY_REFERENCE", "/root/reference")

pd = C.POINTER(C.c_double)


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def _lib():
    return oracle_py.lib()


# ------------------------------------------------------------------------------------------ the fixtures themselves
def test_tiers_are_labelled():
    tiers = {k[6:]: str(FIX[k]) for k in FIX.files if k.startswith("tier__")}
    assert tiers["pcg"] == tiers["normal"] == tiers["hash"] == tiers["dopri"] == tiers["motor"] == "A"
    assert tiers["contact"] == tiers["pgs"] == tiers["rk4"] == tiers["dopri_tableau"] == "B"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_fixtures_regenerate_bit_identically_from_the_reference(tmp_path):
    """The committed fixtures ARE what the reference's text computes here: regenerate and compare."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_ref_cpp_fixtures", os.path.join(HERE, "..", "tools", "make_ref_cpp_fixtures.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    out = str(tmp_path / "again.npz")
    gen.main(out)
    again = np.load(out)
    assert set(again.files) == set(FIX.files)
    for k in FIX.files:
        a, b = FIX[k], again[k]
        assert a.dtype == b.dtype and a.shape == b.shape, k
        assert a.tobytes() == b.tobytes(), k


# ------------------------------------------------------------------------------------------ tier A: generator streams
def test_pcg32_streams_are_bit_exact():
    L = _lib()
    L.orc_pcg32_stream.argtypes = [C.POINTER(C.c_uint64), C.c_int64, C.POINTER(C.c_uint32)]
    raw = FIX["pcg_raw"]
    for s, state in enumerate(FIX["pcg_state"]):
        st = C.c_uint64(int(state))
        out = np.zeros(raw.shape[1], dtype=np.uint32)
        L.orc_pcg32_stream(C.byref(st), out.size, _p(out, C.c_uint32))
        assert np.array_equal(out, raw[s]), s
        g = terrain_numpy.Pcg32(int(state))                      # the numpy restatement used by the terrain oracle
        assert [g() for _ in range(256)] == raw[s, :256].tolist(), s


def test_uniform_streams_are_bit_exact():
    L = _lib()
    L.orc_uniform_stream.argtypes = [C.POINTER(C.c_uint64), C.c_int64, C.POINTER(C.c_float)]
    u01, ulh = FIX["uniform01"], FIX["uniform_lohi"]
    for s, state in enumerate(FIX["pcg_state"]):
        st = C.c_uint64(int(state))
        out = np.zeros(u01.shape[1], dtype=np.float32)
        L.orc_uniform_stream(C.byref(st), out.size, _p(out, C.c_float))
        assert np.array_equal(out, u01[s]), s
        assert np.all((out >= 0) & (out < 1))
        g = terrain_numpy.Pcg32(int(state))
        assert np.array_equal(np.array([g.uniform() for _ in range(256)], dtype=np.float32), u01[s, :256]), s
        # uniform(g, lo, hi) = std::uniform_real_distribution<float>: canonical * (hi - lo) + lo in float, ONE draw each
        # (what jm_random.h and oracle_random.cpp compute for the sensor jitter, oracle/terrain_numpy.py for the gradients)
        lo, hi = FIX["uniform_lo"][s], FIX["uniform_hi"][s]
        assert np.array_equal(u01[s] * np.float32(hi - lo) + lo, ulh[s]), s
        nxt = np.zeros(1, dtype=np.uint32)
        L.orc_pcg32_stream(C.byref(st), 1, _p(nxt, C.c_uint32))
        assert nxt[0] == FIX["next_raw_after"][s, 0] == FIX["next_raw_after"][s, 1], s


def test_ziggurat_normal_streams_and_tables_are_bit_exact():
    L = _lib()
    L.orc_normal_stream.argtypes = [C.POINTER(C.c_uint64), C.c_int64, C.POINTER(C.c_float)]
    kn, fn, wn = np.zeros(128, np.uint32), np.zeros(128, np.float32), np.zeros(128, np.float32)
    L.orc_ziggurat_tables(_p(kn, C.c_uint32), _p(fn, C.c_float), _p(wn, C.c_float))
    assert np.array_equal(kn, FIX["zig_kn"]) and np.array_equal(fn, FIX["zig_fn"]) and np.array_equal(wn, FIX["zig_wn"])
    n01 = FIX["normal01"]
    for s, state in enumerate(FIX["pcg_state"]):
        st = C.c_uint64(int(state))
        out = np.zeros(n01.shape[1], dtype=np.float32)
        L.orc_normal_stream(C.byref(st), out.size, _p(out, C.c_float))
        assert np.array_equal(out, n01[s]), s
        # normal(g, mean, std) = normal(g) * std + mean in float (random.cc:163-166)
        assert np.array_equal(out * FIX["normal_std"][s] + FIX["normal_mean"][s], FIX["normal"][s]), s
        # ... and every accept / reject decision consumed the same number of draws
        nxt = np.zeros(1, dtype=np.uint32)
        L.orc_pcg32_stream(C.byref(st), 1, _p(nxt, C.c_uint32))
        assert nxt[0] == FIX["next_raw_after_normal01"][s] == FIX["next_raw_after"][s, 2], s
    # the fixture exercises the slow paths: wedge rejections and the tail (|x| > 3.44)
    assert np.abs(n01).max() > 3.442620


def test_engine_generator_seeding_is_bit_exact():
    """`generator_.seed(std::seed_seq(seedSeq))` (engine.cc:757) through internal::generateState (random.hxx:17-45)."""
    L = _lib()
    L.orc_pcg32_stream.argtypes = [C.POINTER(C.c_uint64), C.c_int64, C.POINTER(C.c_uint32)]
    seeds = np.ascontiguousarray(FIX["seedseq1_words"][:, 0])
    states = np.zeros(len(seeds), dtype=np.uint64)
    L.orc_engine_rng_seed(_p(seeds, C.c_uint32), C.c_int64(len(seeds)), _p(states, C.c_uint64))
    for i, state in enumerate(states):
        st = C.c_uint64(int(state))
        out = np.zeros(FIX["seedseq1_raw"].shape[1], dtype=np.uint32)
        L.orc_pcg32_stream(C.byref(st), out.size, _p(out, C.c_uint32))
        assert np.array_equal(out, FIX["seedseq1_raw"][i]), i


def test_xxhash_is_bit_exact():
    import torch
    from jiminy_amd.terrain import xxh32_words
    for ln, seed, key, want in zip(FIX["hash_len"], FIX["hash_seed"], FIX["hash_key"], FIX["xxhash"]):
        data = key[:ln].tobytes()
        assert terrain_numpy.xx_hash(data, int(seed)) == int(want), ln
        if ln in (4, 8, 12):        # the tensor program of the product hashes 1-3 int32 words (tiles, Perlin knots)
            words = [torch.tensor([w], dtype=torch.int64) for w in struct.unpack(f"<{ln // 4}I", data)]
            assert int(xxh32_words(words, int(seed))[0]) == int(want), ln
    assert (FIX["hash_len"] >= 16).sum() > 40       # the `len &= 15` branch (random.cc:228) is covered


# ------------------------------------------------------------------------------------------ tier A: scalar laws
def test_dopri_step_size_controller():
    L = _lib()
    L.orc_leaf_dopri_adjust.argtypes = [C.c_int64, pd, pd, C.POINTER(C.c_int32), pd]
    err, dt = FIX["dopri_err"], FIX["dopri_dt"]
    code, out = np.zeros(len(err), np.int32), np.zeros(len(err))
    L.orc_leaf_dopri_adjust(len(err), _p(err), _p(dt), _p(code, C.c_int32), _p(out))
    assert np.array_equal(code, FIX["dopri_code"])
    assert set(code.tolist()) == {0, 1, 2}
    assert np.array_equal(out, FIX["dopri_dt_out"])            # same libm `pow`, same operation order: bit for bit
    k = np.zeros(5)
    L.orc_leaf_dopri_constants(_p(k))
    assert np.array_equal(k, FIX["dopri_constants"])


def test_substep_selection_rule():
    """`Engine::step`'s choice of the next try (engine.cc:2063-2089) on the reference's compiled lines: the oracle's
    `substep_rule` (which `step_dopri` calls), the fixed-step planner of the product and the loop of tests/helpers.py."""
    from jiminy_amd.engine import substep_sizes
    from tests.helpers import ReferenceFixedStepLoop
    L = _lib()
    L.orc_leaf_substep.argtypes = [C.c_int64, pd, pd, pd, C.POINTER(C.c_int32), pd]
    dt, t, tn, tl = (np.ascontiguousarray(FIX[k]) for k in ("substep_dt", "substep_t", "substep_tnext", "substep_too_large"))
    out = np.zeros(len(dt))
    L.orc_leaf_substep(len(dt), _p(dt), _p(t), _p(tn), _p(tl, C.c_int32), _p(out))
    want = FIX["substep_dt_out"]
    assert np.array_equal(out, want)                             # comparisons, one subtraction, one libm fmod: bit for bit
    # every branch is in the fixture: stretched onto the breakpoint, refused after a too-long try, snapped, untouched
    gap = tn - t
    thr = np.where(tl == 0, np.clip(0.1 * dt, 1e-10, 1e-6), 1e-10)
    stretched = (gap < dt) | ((tl <= 1) & (gap < dt + thr))
    assert stretched.sum() > 50 and (~stretched).sum() > 50
    assert ((gap >= dt) & stretched).sum() > 10                  # the residual merge proper
    assert ((tl == 2) & (gap >= dt) & (gap < dt + 1e-6)).sum() > 0
    base = np.where(stretched, gap, dt)
    assert (want != base).sum() > 50 and (want == base).sum() > 10     # snapped to whole microseconds / left alone
    for i, (iv, dmax, dfirst, n) in enumerate(zip(FIX["interval"], FIX["interval_dt_max"], FIX["interval_dt_first"],
                                                  FIX["interval_count"])):
        ref = FIX["interval_sizes"][i, :n]
        assert n < 64
        got = substep_sizes(float(iv), float(dmax), float(dfirst))
        assert len(got) == n and np.array_equal(np.array(got), ref), (i, got, ref)
        loop = ReferenceFixedStepLoop(float(dmax))
        loop.dt = float(dfirst)
        mine = np.array(list(loop.sizes(float(iv))))             # counts the time left down instead of t up: last bits only
        assert len(mine) == n and np.abs(mine - ref).max() <= 1e-18 + 4e-16 * iv, i


def test_breakpoints_of_consecutive_steps():
    """The end time of `Engine::step` with its Kahan compensation (engine.cc:1793-1795) and the time to the next breakpoint of
    the discrete branch (engine.cc:1991-2018: update period, impulse breakpoints, end of the step; a remainder below a
    microsecond skips one update) on the reference's compiled lines, driven over up to 64 consecutive steps: the host planner
    `engine._breakpoint_intervals` gives the same end times, compensation terms and breakpoints, bit for bit."""
    from jiminy_amd.engine import _breakpoint_intervals
    n_bp = 0
    for i in range(len(FIX["bp_period"])):
        per, step, ns = float(FIX["bp_period"][i]), float(FIX["bp_step"][i]), int(FIX["bp_nsteps"][i])
        imp = tuple(float(x) for x in FIX["bp_impulse"][i] if np.isfinite(x))
        opts = {"stepper": {"dtMax": 1e-3, "controllerUpdatePeriod": per, "sensorsUpdatePeriod": per}}
        t, t_err = 0.0, 0.0
        times, t_ends, t_errs = [], [], []
        for _ in range(ns):
            intervals, t_end, t_err = _breakpoint_intervals(t, t_err, step, opts, imp)
            times += [iv[0] for iv in intervals]
            t_ends.append(t_end)
            t_errs.append(t_err)
            t = intervals[-1][0] if intervals else t_end
        n = int(FIX["bp_count"][i])
        assert n < 512 and len(times) == n and np.array_equal(np.array(times), FIX["bp_times"][i, :n]), i
        assert np.array_equal(np.array(t_ends), FIX["bp_t_end"][i, :ns]) and np.array_equal(np.array(t_errs), FIX["bp_t_error"][i, :ns]), i
        n_bp += n
    assert n_bp > 1500 and np.abs(FIX["bp_t_error"]).max() > 0.0      # (the compensation term is exercised)


def test_adaptive_loop_bookkeeping_after_a_try():
    """What `Engine::step`'s inner loop does with the outcome of a try (engine.cc:2138-2221): counters, the restoration of the
    step size after a breakpoint cut it (`dtRestoreThresholdRel`, :2166-2172), the recovery from an evaluation error
    (`dtLargest *= 0.1`, :2197-2200), the next try `min(dtLargest, dtMax)` (:2221) -- on the reference's compiled lines against
    the oracle's `after_try`, which `step_dopri` calls after every try."""
    L = _lib()
    pi64 = C.POINTER(C.c_int64)
    L.orc_leaf_after_try.argtypes = [C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_double, C.c_double, pd, pd, pd, pi64]
    rc, bp = np.ascontiguousarray(FIX["after_try_rc"]), np.ascontiguousarray(FIX["after_try_bp"])
    dt, dtl, dtlp = (FIX[k].copy() for k in ("after_try_dt", "after_try_dt_largest", "after_try_dt_largest_prev"))
    cnt = np.ascontiguousarray(FIX["after_try_counters"].copy())
    L.orc_leaf_after_try(len(rc), _p(rc, C.c_int32), _p(bp, C.c_int32), 0.2, 0.02, _p(dt), _p(dtl), _p(dtlp), _p(cnt, C.c_int64))
    assert np.array_equal(dt, FIX["after_try_dt_out"]) and np.array_equal(dtl, FIX["after_try_dt_largest_out"])
    assert np.array_equal(dtlp, FIX["after_try_dt_largest_prev_out"]) and np.array_equal(cnt, FIX["after_try_counters_out"])
    # every branch: restored step sizes, untouched ones, errors, rejections, the dtMax clip, the INF of fixed-step steppers
    ok = rc == 0
    restored = ok & (FIX["after_try_dt_largest_out"] != FIX["after_try_dt_largest"])
    assert restored.sum() > 5 and (ok & (bp == 1) & ~restored).sum() > 5 and (rc == 1).sum() > 20 and (rc == 2).sum() > 20
    assert (FIX["after_try_dt_out"] == 0.02).sum() > 10 and np.isinf(FIX["after_try_dt_largest"]).sum() > 10


def test_periodic_update_and_impulse_activity_rules():
    """When the controller command / a profile force is refreshed (engine.cc:1923-1927, 1903-1907: the same expression) and when
    an impulse force is active (engine.cc:1857-1869) on the reference's compiled lines: `engine.update_due` and
    `engine.impulse_active`, the two predicates the host side plans its launches with."""
    from jiminy_amd.engine import impulse_active, update_due
    per, t = FIX["update_period"], FIX["update_t"]
    got = np.array([update_due(float(p), float(x)) for p, x in zip(per, t)], dtype=np.int32)
    assert np.array_equal(got, FIX["update_force"]) and np.array_equal(got, FIX["update_controller"])
    assert 0.2 < got.mean() < 0.9
    ts = FIX["impulse_times"]
    for i, (ti, dti) in enumerate(zip(FIX["impulse_t"], FIX["impulse_dt"])):
        active = False
        for k, x in enumerate(ts):
            before = active
            active = impulse_active(float(ti), float(dti), float(x), active)
            assert int(active) == FIX["impulse_active"][i, k], (i, k)
            # (`hasDynamicsChanged` is raised whenever one of the two tests fires, also when the flag keeps its value)
            fired = (x > ti - 1e-10) or (x >= ti + dti - 1e-10)
            assert int(fired) == FIX["impulse_changed"][i, k], (i, k, before)
    assert FIX["impulse_active"].sum() > 10 and (FIX["impulse_active"].sum(axis=1) == 0).any()   # (a 1e-10 s impulse never acts)


def test_update_period_arithmetic():
    """`isGcdIncluded(sensorsUpdatePeriod, controllerUpdatePeriod)` (utilities/helpers.hxx:59-116; engine.cc:749-750 the stepper
    update period, engine.cc:2699-2733 the refusal of periods that are not multiples of each other) on the reference's compiled
    text: the host side's `is_gcd_included`, including the pairs the reference refuses although they are multiples on paper."""
    from jiminy_amd.engine import is_gcd_included
    a, b = FIX["period_a"], FIX["period_b"]
    got = [is_gcd_included(float(x), float(y)) for x, y in zip(a, b)]
    assert np.array_equal(np.array([g[0] for g in got], dtype=np.int32), FIX["period_included"])
    assert np.array_equal(np.array([g[1] for g in got]), FIX["period_min"])
    inc = FIX["period_included"].astype(bool)
    ratio = np.maximum(a, b) / np.where(np.minimum(a, b) > 0, np.minimum(a, b), 1.0)
    on_paper = (np.minimum(a, b) > 1e-15) & (np.abs(ratio - np.round(ratio)) < 1e-9)
    assert (on_paper & ~inc).sum() > 10 and (on_paper & inc).sum() > 50         # (0.03, 0.01) is refused, (0.005, 0.001) is not
    assert np.isinf(FIX["period_min"]).sum() == 1 and (~on_paper & ~inc).sum() >= 2


def test_simple_motor_law():
    L = _lib()
    L.orc_leaf_motor_law.argtypes = [C.c_int64, pd, pd, pd]
    p = np.ascontiguousarray(FIX["motor_params"])
    um, ut = np.zeros(len(p)), np.zeros(len(p))
    L.orc_leaf_motor_law(len(p), _p(p), _p(um), _p(ut))
    assert np.array_equal(um, FIX["motor_u"])                   # clamps and one product: bit for bit
    assert np.array_equal(ut, FIX["motor_u_transmission"])      # + the friction branch through the same libm tanh
    # the fixture reaches every branch: saturated both ways, velocity-scaled bounds, friction of both signs
    lim = p[:, 4]
    assert (np.abs(um) == lim).any() and ((np.abs(um) < lim) & (um != p[:, 13])).any() and (um == p[:, 13]).any()


# ------------------------------------------------------------------------------------------ tier B
def test_contact_law():
    L = _lib()
    L.orc_leaf_contact_law.argtypes = [C.c_int64, pd, pd]
    p = np.ascontiguousarray(FIX["contact_params"])
    f = np.zeros((len(p), 3))
    L.orc_leaf_contact_law(len(p), _p(p), _p(f))
    want = FIX["contact_force"]
    assert np.all(want[:, 3:] == 0.0)                            # the law returns no moment (engine.cc:3237)
    scale = np.maximum(np.abs(want[:, :3]).max(axis=1, keepdims=True), 1e-300)
    assert np.max(np.abs(f - want[:, :3]) / scale) <= 1e-15      # same operations in the same order
    assert (want[:, :3] != 0).any(axis=1).sum() > 500 and (p[:, 8] >= 0).sum() > 20   # in and out of contact


def test_butcher_tableaux():
    L = _lib()
    t = np.zeros(94)
    L.orc_leaf_tableaux(_p(t))
    want = np.concatenate([FIX["rk4_A"].ravel(), FIX["rk4_c"], FIX["rk4_b"], FIX["dopri_A"].ravel(), FIX["dopri_c"],
                           FIX["dopri_b"], FIX["dopri_e"]])
    assert np.array_equal(t, want)
    # the product's host-side copy (the step schedule of the per-stage adaptive path)
    from jiminy_amd import engine
    for name, key in (("DOPRI_C", "dopri_c"),):
        if hasattr(engine, name):
            assert np.array_equal(np.asarray(getattr(engine, name)), FIX[key])


def _pgs_call(L, k, sweep_w):
    types, dims = np.ascontiguousarray(FIX[f"pgs{k}_types"]), np.ascontiguousarray(FIX[f"pgs{k}_dims"])
    A, b = np.ascontiguousarray(FIX[f"pgs{k}_A"]), np.ascontiguousarray(FIX[f"pgs{k}_b"])
    prm = np.ascontiguousarray(FIX[f"pgs{k}_prm"])
    x, y = FIX[f"pgs{k}_x0"].copy(), np.zeros(len(b))
    it = C.c_int32(0)
    ok = L.orc_leaf_pgs(len(types), _p(types, C.c_int32), _p(dims, C.c_int32), len(b), _p(A), _p(b), _p(prm),
                        int(FIX[f"pgs{k}_iter_max"]), C.c_double(sweep_w), _p(x), _p(y), C.byref(it))
    return ok, x, y, it.value


def test_projected_gauss_seidel_sweep_and_solver():
    L = _lib()
    L.orc_leaf_pgs.argtypes = [C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, pd, pd, pd, C.c_int, C.c_double,
                               pd, pd, C.POINTER(C.c_int32)]
    L.orc_leaf_pgs.restype = C.c_int
    n_conv = 0
    for k in range(int(FIX["pgs_count"])):
        for j, w in enumerate(FIX[f"pgs{k}_w"]):
            _, x, y, _ = _pgs_call(L, k, float(w))
            wx, wy = FIX[f"pgs{k}_sweep_x_y"][j]
            # one sweep: identical operations; the only freedom is the summation inside A.col(i).dot(x)
            assert np.max(np.abs(x - wx)) <= 1e-14 * max(1.0, np.abs(wx).max()), (k, j)
            assert np.max(np.abs(y - wy)) <= 1e-14 * max(1.0, np.abs(wy).max()), (k, j)
        ok, x, y, _ = _pgs_call(L, k, -1.0)
        assert ok == int(FIX[f"pgs{k}_solve_ok"][0]), k
        n_conv += ok
        wx, wy = FIX[f"pgs{k}_solve_x"], FIX[f"pgs{k}_solve_y"]
        # up to 100 sweeps of a contraction: round-off does not grow
        assert np.max(np.abs(x - wx)) <= 1e-12 * max(1.0, np.abs(wx).max()), k
        assert np.max(np.abs(y - wy)) <= 1e-12 * max(1.0, np.abs(wy).max()), k
    assert 0 < n_conv < int(FIX["pgs_count"])        # both exits of the solver loop are pinned


# ------------------------------------------------------------------------------------------ the HIP kernels (C ABI)
def _engine(model, B, device, options):
    import torch
    from jiminy_amd.engine import BatchedEngine
    eng = BatchedEngine(model, B, dtype=torch.float64, device=device)
    eng.set_options(options)
    return eng


@pytest.mark.gpu
def test_hip_contact_law_matches_the_reference_text(gpu_device):
    """Spring-damper law of the device kernels (k_batch, point mass with one contact point at its origin) on the
    flat-ground groups of the fixture: the base sits at z = depth with identity orientation and moves with the
    fixture's velocity, so the contact frame IS the world frame and `contact_forces` is the law's output."""
    import torch
    from tests import robots
    model = robots.point_mass()
    G, ng = int(FIX["contact_group"]), int(FIX["contact_flat_groups"])
    p, want = FIX["contact_params"], FIX["contact_force"]
    for g in range(ng):
        rows = slice(g * G, (g + 1) * G)
        k, c, mu, eps, vt = p[g * G, :5]
        eng = _engine(model, G, gpu_device, {
            "stepper": {"odeSolver": "runge_kutta_4", "dtMax": 1e-3, "controllerUpdatePeriod": 1e-3, "sensorsUpdatePeriod": 1e-3},
            "contacts": {"model": "spring_damper", "stiffness": float(k), "damping": float(c), "friction": float(mu),
                         "transitionEps": float(eps), "transitionVelocity": float(vt)}})
        q = np.zeros((7, G))
        q[2], q[6] = p[rows, 8], 1.0
        v = np.zeros((6, G))
        v[:3] = p[rows, 9:12].T
        eng.start(torch.from_numpy(q), torch.from_numpy(v))
        torch.cuda.synchronize()
        f = eng.field("contact_forces").cpu().numpy()[:3].T
        w = want[rows, :3]
        scale = np.maximum(np.abs(w).max(axis=1, keepdims=True), 1e-300)
        err = np.abs(f - w) / scale
        # the device's own tanh / sqrt (a few ulp) against libm's: 1e-13
        assert err[np.abs(w).max(axis=1) > 0].max() <= 1e-13, (g, err.max())
        assert np.all(f[np.abs(w).max(axis=1) == 0] == 0.0), g
        eng.stop()


@pytest.mark.gpu
def test_hip_motor_law_matches_the_reference_text(gpu_device):
    """SimpleMotor::computeEffort on the device: one pendulum per parameter group of the fixture, lanes = its (v, command)."""
    import torch
    from jiminy_amd.model import add_motor, build_model_from_urdf
    G = int(FIX["motor_group"])
    p = FIX["motor_params"]
    for g in range(len(p) // G):
        rows = slice(g * G, (g + 1) * G)
        red, eff_on, vel_on, slope, eff, vel, fr_on, fvp, fvn, fdp, fdn, fds = p[g * G, :12]
        model = build_model_from_urdf(os.path.join(HERE, "data", "pendulum.urdf"), name="pendulum")
        add_motor(model, "pivot", "pivot", mechanicalReduction=float(red), enableEffortLimit=bool(eff_on),
                  enableVelocityLimit=bool(vel_on), velocityEffortInvSlope=float(slope), effortLimitFromUrdf=False,
                  effortLimit=float(eff), velocityLimitFromUrdf=False, velocityLimit=float(vel), enableFriction=bool(fr_on),
                  frictionViscousPositive=float(fvp), frictionViscousNegative=float(fvn), frictionDryPositive=float(fdp),
                  frictionDryNegative=float(fdn), frictionDrySlope=float(fds))
        eng = _engine(model, G, gpu_device, {
            "stepper": {"odeSolver": "runge_kutta_4", "dtMax": 1e-3, "controllerUpdatePeriod": 1e-3, "sensorsUpdatePeriod": 1e-3},
            "contacts": {"model": "spring_damper"}})
        eng.set_command(torch.from_numpy(np.ascontiguousarray(p[rows, 13][None, :])))
        eng.start(torch.zeros((1, G), dtype=torch.float64), torch.from_numpy(np.ascontiguousarray(p[rows, 12][None, :])))
        torch.cuda.synchronize()
        um = eng.field("u_motor").cpu().numpy()[0]
        u = eng.field("u").cpu().numpy()[0]
        wm, wt = FIX["motor_u"][rows], FIX["motor_u_transmission"][rows]
        assert np.max(np.abs(um - wm)) <= 1e-14 * max(1.0, np.abs(wm).max()), g     # clamps: exact up to the division
        # u = uTransmission on this robot (no bound, no flexibility); friction goes through the device tanh
        assert np.max(np.abs(u - wt)) <= 1e-13 * max(1.0, np.abs(wt).max()), g
        eng.stop()


@pytest.mark.gpu
def test_hip_normal_stream_matches_the_reference_text(gpu_device):
    """jm_block_sensor_noise with unit standard deviation on zero data IS the stream `normal(g)`: 512 draws of each of
    the fixture's generators, generator state bit-exact afterwards (every accept / reject decision agreed)."""
    import torch
    from jiminy_amd import _abi, _lib, load_builtin
    lib = _lib.load_for(load_builtin("cartpole"))
    states = FIX["pcg_state"] | np.uint64(3)                     # PCG32's constructor (random.cc:10-13)
    B, nd = len(states), 512
    rng = torch.from_numpy(states.view(np.int64).copy()[None, :]).cuda()
    std = np.ones((1, 1))
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    got = np.zeros((B, nd), dtype=np.float32)
    for k in range(nd):
        data = torch.zeros((1, B), dtype=torch.float32, device="cuda")
        lib.check(lib.L.jm_block_sensor_noise(_abi.JM_F32, B, 1, 1, C.c_void_p(data.data_ptr()), C.c_void_p(rng.data_ptr()),
                                              _p(std), None, None, stream))
        got[:, k] = data.cpu().numpy()[0]
    want = FIX["normal01"][:, :nd]
    same = got == want
    assert same.mean() > 0.98                                    # the fast path is exact
    ulp = np.abs(got[~same].view(np.int32).astype(np.int64) - want[~same].view(np.int32).astype(np.int64))
    assert ulp.size == 0 or ulp.max() <= 1                       # wedge / tail through the device's logf / expf
    # state after 512 normals == the reference's: continue both streams on the host and compare the next raw draws
    L = _lib_o = oracle_py.lib()
    L.orc_pcg32_stream.argtypes = [C.POINTER(C.c_uint64), C.c_int64, C.POINTER(C.c_uint32)]
    L.orc_normal_stream.argtypes = [C.POINTER(C.c_uint64), C.c_int64, C.POINTER(C.c_float)]
    dev_state = rng.cpu().numpy().view(np.uint64)[0]
    for s in range(B):
        st = C.c_uint64(int(dev_state[s]))
        rest = np.zeros(FIX["normal01"].shape[1] - nd, dtype=np.float32)
        L.orc_normal_stream(C.byref(st), rest.size, _p(rest, C.c_float))      # oracle == reference (CPU test above)
        assert np.array_equal(rest, FIX["normal01"][s, nd:]), s
