"""The per-lane kernel code (jiminy_amd/csrc/*.h) compiled for the host (tests/hostemu) against
the CPU oracle: same seeded inputs, every mode, both kernel variants.  This runs without a GPU and
pins the kernel *logic*; the `-m gpu` tests pin the device build of the same sources."""
import numpy as np
import pytest

from jiminy_amd import load_builtin
from jiminy_amd.synthetic import sample_states
from tests import robots
from tests.helpers import alloc_soa, oracle_batch, rel_err
from tests.hostemu import emu

OUTS = ("q", "v", "a", "u_motor", "u", "imu", "force", "contact", "encoder", "effort", "energy",
        "contact_forces", "f_external", "joint_forces", "centroidal")


def _models():
    return {
        "cartpole": lambda: load_builtin("cartpole"),
        "double_pendulum": lambda: load_builtin("double_pendulum"),
        "anymal": lambda: load_builtin("anymal"),
        "pendulum": robots.pendulum,
        "point_mass": robots.point_mass,
        "two_masses": robots.two_masses,
        "tree_arm": lambda: robots.tree_arm(False),
        "tree_arm_ff": lambda: robots.tree_arm(True),
        "arm7": robots.arm7,
        "pendulum_flexible": robots.pendulum_flexible,
        "tree_arm_flex": lambda: robots.tree_arm_flexible(False),
        "tree_arm_flex_ff": lambda: robots.tree_arm_flexible(True),
        "crane_walker": robots.crane_walker,
        "biped": robots.biped,
        "biped_torso": lambda: robots.biped(True),
    }


def _states(model, B, seed):
    return sample_states(model, B, seed=seed, base_height=(0.3, 0.6), grounded_fraction=0.6)


def _pair(model, B, seed):
    st = _states(model, B, seed)
    ref, got = alloc_soa(model, B), alloc_soa(model, B)
    for k in ("q", "v", "command"):
        if st[k].shape[0]:
            ref[k][:] = st[k]
            got[k][:] = st[k]
    return ref, got


def _check(got, ref, tol, lanes=None, what=""):
    for k in OUTS:
        e = rel_err(got[k], ref[k], lanes)
        assert e < tol, (what, k, e)


@pytest.mark.parametrize("name", list(_models()))
@pytest.mark.parametrize("solver", ["runge_kutta_4", "euler_explicit"])
def test_lane_kernel_matches_oracle(name, solver):
    model = _models()[name]()
    B = 24
    ref, got = _pair(model, B, seed=2)
    oracle_batch(model, ref, "start")
    emu.run(model, got, "start")
    # (flexibility joints: rotor inertias of 1e-5 .. 1e-3 kg m^2 under random deflections give accelerations of 1e5 rad/s^2,
    # the momentum derivative cancels them against each other)
    _check(got, ref, 1e-10 if "flex" in name else 1e-12, what="start")
    assert np.array_equal(got["status"], ref["status"])
    for i in range(6):
        # (the flexible pendulum's yaw mode decays at damping / flexibility inertia = 1e4 1/s: explicit steps of 5e-5 s)
        kw = dict(solver=solver, dt=5e-5 if name == "pendulum_flexible" else 5e-4, n_substeps=2 if i % 2 else 1,
                  command_changed=(i % 3 == 0))
        oracle_batch(model, ref, "step", **kw)
        emu.run(model, got, "step", **kw)
    ok = (ref["status"][0] & 1) == 0
    assert ok.any()
    _check(got, ref, 1e-8 if "flex" in name else 1e-9, ok, what="steps")
    assert np.array_equal(got["status"][0][ok], ref["status"][0][ok])


@pytest.mark.parametrize("name", ["tree_arm_flex_ff", "pendulum_flexible", "anymal"])
def test_lane_kernel_per_environment_friction_and_flexibility(name):
    """The per-environment rows of the one-robot-per-lane kernels: ground friction of every lane under the spring-damper law
    (JM_F_FRICTION, envs/locomotion.py:257-262) and stiffness / damping of the flexibility joints of every lane
    (JM_F_FLEXIBILITY, envs/locomotion.py:288-296) -- different values per lane, against the oracle's one-robot engine."""
    model = _models()[name]()
    B = 16
    rng = np.random.default_rng(5)
    ref, got = _pair(model, B, seed=3)
    nflex = len(model.flexibility_joint_indices)
    extra = {}
    if model.ncontacts:
        extra["friction"] = np.ascontiguousarray(10.0 ** rng.uniform(-0.7, 0.3, (1, B)))
    if nflex:
        fx = np.zeros((6 * nflex, B))
        for k, j in enumerate(model.flexibility_joint_indices):
            fx[6 * k:6 * k + 3] = model.flex_stiffness[j][:, None] * rng.uniform(0.5, 1.5, (1, B))
            fx[6 * k + 3:6 * k + 6] = model.flex_damping[j][:, None] * rng.uniform(0.5, 1.5, (1, B))
        extra["flexibility"] = fx
    assert extra
    for arr in (ref, got):
        arr.update(extra)
    plain = {k: v.copy() for k, v in got.items() if k not in extra}
    oracle_batch(model, ref, "start")
    emu.run(model, got, "start")
    emu.run(model, plain, "start")
    _check(got, ref, 1e-10, what="start")
    dt = 5e-5 if name == "pendulum_flexible" else 5e-4
    for i in range(6):
        kw = dict(solver="runge_kutta_4", dt=dt, n_substeps=2 if i % 2 else 1, command_changed=(i % 3 == 0))
        oracle_batch(model, ref, "step", **kw)
        emu.run(model, got, "step", **kw)
        emu.run(model, plain, "step", **kw)
    ok = (ref["status"][0] & 1) == 0
    assert ok.any()
    _check(got, ref, 1e-8, ok, what="steps")
    # ... and the rows are read: the batch-wide parameters give another trajectory
    assert rel_err(plain["v"], ref["v"], ok) > 1e-6


QUAD_CASES = {
    # name: (states kwargs, dt)
    "anymal": (dict(base_height=(0.3, 0.6), grounded_fraction=0.6), 5e-4),
    "atlas": (dict(base_height=(0.85, 1.0), grounded_fraction=0.6), 2.5e-4),
    "crane_walker": (dict(base_height=(0.45, 0.65), grounded_fraction=0.6), 2.5e-4),
    # two / three leaf chains only: the decomposition is completed with empty limbs (codegen.quad_structure)
    "biped": (dict(base_height=(0.55, 0.75), grounded_fraction=0.6), 2.5e-4),
    "biped_torso": (dict(base_height=(0.55, 0.75), grounded_fraction=0.6), 2.5e-4),
}


@pytest.mark.parametrize("name", list(QUAD_CASES))
@pytest.mark.parametrize("solver", ["runge_kutta_4", "euler_explicit"])
def test_quad_kernel_matches_oracle(name, solver):
    """Branch-parallel kernel (4 lanes per robot, ABA in root coordinates): ANYmal (root + 4 equal
    limbs) and Atlas (5-joint trunk tree, padded limbs attached at two different trunk joints,
    16 contact points per foot).  The formulation differs from the oracle's joint-local one, so
    agreement is at accumulated round-off (1e-11), not bitwise."""
    model = _models()[name]() if name in _models() else load_builtin(name)
    kw_states, dt = QUAD_CASES[name]
    B = 24 if name == "anymal" else 12
    st = sample_states(model, B, seed=4, **kw_states)
    ref, got = alloc_soa(model, B), alloc_soa(model, B)
    for k in ("q", "v", "command"):
        ref[k][:] = st[k]
        got[k][:] = st[k]
    oracle_batch(model, ref, "start")
    emu.run(model, got, "start", variant="quad")
    _check(got, ref, 2e-11, what="start")
    assert np.array_equal(got["status"], ref["status"])
    assert (np.abs(ref["contact_forces"]).sum(axis=0) > 0).sum() >= 3  # contact branch exercised
    for i in range(6):
        kw = dict(solver=solver, dt=dt, n_substeps=2 if i % 2 else 1, command_changed=(i % 3 == 0))
        oracle_batch(model, ref, "step", **kw)
        emu.run(model, got, "step", variant="quad", **kw)
    ok = (ref["status"][0] & 1) == 0
    assert ok.sum() >= B // 2
    _check(got, ref, 1e-8, ok, what="steps")
    assert np.array_equal(got["status"][0][ok], ref["status"][0][ok])


@pytest.mark.parametrize("name", ["anymal", "atlas"])
def test_quad_extra_terms_with_a_general_gravity_field(name):
    """`world.gravity` is a 6-vector (engine.h:291); the output sweep of the branch-parallel kernel takes the total of
    Y * a_gravity over the bodies from the subtree mass and first moment when the angular part is zero and sums it body
    by body otherwise: both branches against the oracle (energy, joint wrenches, centroidal momentum derivative)."""
    from jiminy_amd import _abi
    model = load_builtin(name)
    kw_states, dt = QUAD_CASES[name]
    B = 8
    st = sample_states(model, B, seed=11, **kw_states)
    for grav in ((0.4, -0.3, -9.81, 0.0, 0.0, 0.0), (0.4, -0.3, -9.81, 0.05, -0.02, 0.03)):
        ref, got = alloc_soa(model, B), alloc_soa(model, B)
        for k in ("q", "v", "command"):
            ref[k][:] = st[k]
            got[k][:] = st[k]
        oracle_batch(model, ref, "start", options=dict(gravity=grav))
        emu.run(model, got, "start", variant="quad", options=_abi.make_options(gravity=grav))
        _check(got, ref, 2e-11, what=f"start {grav}")
        for _ in range(2):
            oracle_batch(model, ref, "step", options=dict(gravity=grav), solver="runge_kutta_4", dt=dt, n_substeps=1, command_changed=False)
            emu.run(model, got, "step", variant="quad", options=_abi.make_options(gravity=grav), solver="runge_kutta_4", dt=dt,
                    n_substeps=1, command_changed=False)
        ok = (ref["status"][0] & 1) == 0
        _check(got, ref, 1e-8, ok, what=f"steps {grav}")
        assert np.abs(ref["centroidal"][9:]).max() > 1e-3


@pytest.mark.parametrize("variant", ["lane", "quad"])
def test_sensors_are_only_refreshed_on_request(variant):
    model = load_builtin("anymal")
    ref, got = _pair(model, 8, seed=5)
    emu.run(model, got, "start", variant=variant)
    before = {k: got[k].copy() for k in ("imu", "force", "encoder", "effort")}
    emu.run(model, got, "step", dt=1e-3, update_sensors=False, variant=variant)
    for k, v in before.items():
        assert np.array_equal(got[k], v), k
    emu.run(model, got, "step", dt=1e-3, update_sensors=True, variant=variant)
    assert not np.array_equal(got["encoder"], before["encoder"])


@pytest.mark.parametrize("variant", ["lane", "quad"])
def test_dynamics_and_reset_modes(variant):
    model = load_builtin("anymal")
    B = 16
    st = _states(model, B, 6)
    ref, got = _pair(model, B, 6)
    oracle_batch(model, ref, "start")
    emu.run(model, got, "start", variant=variant)
    # compute_robots_dynamics at another state leaves the bound state untouched
    st2 = _states(model, B, 7)
    got["q_in"], got["v_in"] = st2["q"].copy(), st2["v"].copy()
    got["a_out"] = np.zeros_like(got["a"])
    keep = {k: got[k].copy() for k in ("q", "v", "a", "imu")}
    emu.run(model, got, "dynamics", variant=variant)
    for k, v in keep.items():
        assert np.array_equal(got[k], v), k
    ref2 = alloc_soa(model, B)
    ref2["q"][:], ref2["v"][:], ref2["command"][:] = st2["q"], st2["v"], st["command"]
    oracle_batch(model, ref2, "dynamics")
    # (round-off: the kernel works in root-body coordinates, the oracle in joint-local ones; observed 1.2e-12)
    assert rel_err(got["a_out"], ref2["a"]) < 5e-12
    # reset of a subset of lanes == start of those lanes, the others untouched
    for _ in range(3):
        emu.run(model, got, "step", dt=1e-3, variant=variant)
    mask = np.zeros(B, dtype=np.uint8)
    mask[::3] = 1
    got["mask"], got["q_init"], got["v_init"] = mask, st2["q"].copy(), st2["v"].copy()
    before = {k: got[k].copy() for k in OUTS}
    emu.run(model, got, "reset", variant=variant)
    sel = mask.astype(bool)
    ref3 = alloc_soa(model, B)
    ref3["q"][:], ref3["v"][:], ref3["command"][:] = st2["q"], st2["v"], st["command"]
    oracle_batch(model, ref3, "start")
    for k in OUTS:
        assert rel_err(got[k], ref3[k], sel) < 5e-12, k
        assert np.array_equal(got[k][:, ~sel], before[k][:, ~sel]), k


@pytest.mark.parametrize("variant", ["lane", "quad"])
def test_fp32_build_is_within_single_precision_of_fp64(variant):
    model = load_builtin("anymal")
    B = 16
    st = sample_states(model, B, seed=8, grounded_fraction=0.0, base_height=(1.0, 1.2))
    a64, a32 = alloc_soa(model, B), alloc_soa(model, B, np.float32)
    for k in ("q", "v", "command"):
        a64[k][:] = st[k]
        a32[k][:] = st[k]
    emu.run(model, a64, "start", variant=variant)
    emu.run(model, a32, "start", variant=variant, dtype=np.float32)
    assert rel_err(a32["a"].astype(np.float64), a64["a"]) < 2e-4
    for _ in range(10):
        emu.run(model, a64, "step", dt=1e-3, variant=variant)
        emu.run(model, a32, "step", dt=1e-3, variant=variant, dtype=np.float32)
    assert rel_err(a32["q"].astype(np.float64), a64["q"]) < 1e-4
    assert rel_err(a32["a"].astype(np.float64), a64["a"]) < 5e-3


@pytest.mark.slow
def test_long_horizon_sensitivity():
    """1000 RK4 steps with ground contacts: the kernel code and the oracle only differ by
    summation order, yet the worst lane drifts by many orders of magnitude (chaotic contact
    events) while the median stays near round-off. This is the yardstick for the statistical bar
    of the GPU contact parity test."""
    model = load_builtin("anymal")
    mask = model.bounded_position_mask()
    model.position_lower[mask] = -np.inf
    model.position_upper[mask] = np.inf
    B, dt = 48, 5e-4
    st = sample_states(model, B, seed=0, command_fraction=0.05, joint_vel_std=0.1,
                       base_twist_std=0.05, joint_range=0.4)
    ref, got = alloc_soa(model, B), alloc_soa(model, B)
    for k in ("q", "v", "command"):
        ref[k][:] = st[k]
        got[k][:] = st[k]
    from oracle.oracle_py import OracleEngine
    from tests.helpers import oracle_io
    eng = OracleEngine(model)
    io = oracle_io(ref)
    eng.batch_run("start", io)
    emu.run(model, got, "start", variant="lane")
    bad = np.zeros(B, dtype=bool)
    for _ in range(1000):
        eng.batch_run("step", io, solver="runge_kutta_4", dt=dt, command_changed=False)
        emu.run(model, got, "step", solver="runge_kutta_4", dt=dt, command_changed=False, variant="lane")
        bad |= ref["status"][0] != 0
    ok = ~bad
    num = np.abs(got["a"] - ref["a"]).max(axis=0)[ok]
    den = np.maximum(np.abs(ref["a"]).max(axis=0), 1.0)[ok]
    err = num / den
    assert ok.sum() >= 32
    assert np.median(err) < 1e-9
    assert (err <= 1e-5).mean() >= 0.9


@pytest.mark.parametrize("name", ["anymal", "crane_walker", "atlas"])
def test_persistent_adaptive_stepper_matches_oracle(name):
    """jm_qdopri.h (every quad runs its robot's whole Dormand-Prince loop to the breakpoint) against the oracle's
    restatement of the reference's adaptive loop: robots in free flight and robots landing on the ground, three
    breakpoint intervals.  Accept / reject decisions are discontinuous in the error estimate, so the robots that
    follow the oracle's step sequence (all but a few) agree to round-off and the others within the tolerance."""
    from oracle.oracle_py import OracleEngine, adaptive_state
    from tests.helpers import oracle_io
    model = robots.crane_walker() if name == "crane_walker" else load_builtin(name)
    B = 16 if name != "atlas" else 8       # (Atlas: long limbs, the stage velocities / commands travel through the stage buffer)
    heights = {"anymal": (0.5, 0.7), "crane_walker": (0.5, 0.8), "atlas": (0.9, 1.1)}[name]
    st = sample_states(model, B, seed=21, base_height=heights, grounded_fraction=0.4)
    ref, got = alloc_soa(model, B), alloc_soa(model, B)
    for k in ("q", "v", "command"):
        ref[k][:] = st[k]
        got[k][:] = st[k]
    orc = OracleEngine(model)
    io = oracle_io(ref)
    orc.batch_run("start", io)
    emu.run(model, got, "start", variant="quad")
    ad_ref, ad_got = adaptive_state(B), adaptive_state(B)
    for k in ("iter", "iter_failed", "succ_too_large", "succ_failed"):
        ad_got[k] = ad_got[k].astype(np.int64)
    tol = dict(tol_rel=1e-6, tol_abs=1e-7)
    for i, t_next in enumerate((2e-3, 4e-3, 7e-3)):
        orc.batch_run_dopri(io, ad_ref, t_next, new_step=True, command_changed=False, update_sensors=False, **tol)
        left, attempts = emu.run_dopri(model, got, ad_got, t_next, new_step=True, **tol)
        assert left == 0 and attempts >= 1
    ok = ((ref["status"][0] | got["status"][0]) & 9) == 0
    assert ok.sum() >= B - 2
    same = ok & (ad_got["iter"] == ad_ref["iter"]) & (ad_got["iter_failed"] == ad_ref["iter_failed"])
    assert same.mean() > 0.8, (ad_got["iter"], ad_ref["iter"], ad_got["iter_failed"], ad_ref["iter_failed"])
    assert np.allclose(ad_got["t"][ok], 7e-3, rtol=0, atol=1e-12)
    for k in ("q", "v", "a"):
        scale = max(np.abs(ref[k][:, same]).max(), 1.0)
        assert np.abs(got[k] - ref[k])[:, same].max() / scale < 1e-7, k
        assert np.abs(got[k] - ref[k])[:, ok].max() / scale < 1e-2, k
    assert np.allclose(ad_got["dt_largest"][same], ad_ref["dt_largest"][same], rtol=1e-5)
