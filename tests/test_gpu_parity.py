"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch

from jiminy_amd import load_builtin
from jiminy_amd.engine import BatchedEngine
from jiminy_amd.synthetic import sample_states
from tests.helpers import ReferenceFixedStepLoop, alloc_soa, oracle_batch, oracle_engine_step, rel_err

pytestmark = pytest.mark.gpu

OUTS = ("q", "v", "a", "u_motor", "imu", "force", "encoder", "effort", "contact_forces")


EXTRA = ("contact_forces", "energy", "f_external", "joint_forces", "centroidal")


def _engine(model, B, dtype, solver, dt):
    eng = BatchedEngine(model, B, dtype=dtype, extra_outputs=EXTRA)
    eng.set_options({"stepper": {"odeSolver": solver, "dtMax": dt, "controllerUpdatePeriod": dt,
                                 "sensorsUpdatePeriod": dt}, "contacts": {"model": "spring_damper"}})
    return eng


@pytest.mark.parametrize("name,B", [("cartpole", 4096), ("double_pendulum", 256), ("anymal", 256),
                                    ("atlas", 128)])
@pytest.mark.parametrize("solver", ["runge_kutta_4", "euler_explicit"])
def test_start_and_steps_match_oracle_fp64(gpu_device, name, B, solver):
    model = load_builtin(name)
    st = sample_states(model, B, seed=3, base_height=(0.9, 1.1) if name == "atlas" else (0.45, 0.65))
    dt = 1e-3 if name != "atlas" else 2.5e-4
    ref = alloc_soa(model, B)
    for k in ("q", "v", "command"):
        ref[k][:] = st[k]
    oracle_batch(model, ref, "start")
    loop = ReferenceFixedStepLoop(dt)   # the reference's sub-step rule: opens with a 1 us step (engine.cc:1176)
    eng = _engine(model, B, torch.float64, solver, dt)
    eng.set_command(torch.from_numpy(st["command"]))
    eng.start(torch.from_numpy(st["q"]), torch.from_numpy(st["v"]))
    torch.cuda.synchronize()
    for k in OUTS + ("u", "energy", "f_external", "joint_forces", "centroidal"):
        got = eng.field(k).cpu().numpy()
        assert rel_err(got, ref[k]) < 1e-11, (k, rel_err(got, ref[k]))
    assert np.array_equal(eng.status.cpu().numpy(), ref["status"][0])
    nsteps = 20
    for i in range(nsteps):
        oracle_engine_step(model, ref, loop, dt, solver, command_changed=(i == 0))
        if i == 0:
            eng.mark_command_changed()
        eng.step(dt)
    torch.cuda.synchronize()
    ok = (ref["status"][0] & 1) == 0
    assert ok.sum() > 0.5 * B
    for k in OUTS + ("u", "energy", "f_external", "joint_forces", "centroidal"):
        got = eng.field(k).cpu().numpy()
        # fp64 tolerance: only operation order / FMA contraction differ from the oracle
        assert rel_err(got, ref[k], ok) < 1e-8, (k, rel_err(got, ref[k], ok))
    assert np.array_equal(eng.status.cpu().numpy()[ok], ref["status"][0][ok])


@pytest.mark.parametrize("robot", ["pendulum", "point_mass", "two_masses", "tree_arm", "tree_arm_ff", "arm7",
                                   "tree_arm_flex", "tree_arm_flex_ff", "crane_walker", "biped", "biped_torso"])
def test_small_robots_cover_every_joint_type(gpu_device, robot):
    """Authored test robots: aligned / unaligned revolute and prismatic joints, unbounded joints,
    fixed and floating base, friction motors, world-fixed contact frames (lane kernel) and, with
    crane_walker, the branch-parallel kernel on a trunk tree with a prismatic + an unaligned joint,
    padded limbs on two attachment joints and a ragged batch (B % 16 != 0); with the bipeds, the same kernel on robots
    with two / three leaf chains only (empty limbs)."""
    from tests import robots
    model = {"pendulum": robots.pendulum, "point_mass": robots.point_mass, "two_masses": robots.two_masses,
             "tree_arm": lambda: robots.tree_arm(False), "tree_arm_ff": lambda: robots.tree_arm(True), "arm7": robots.arm7,
             # (spherical flexibility joints: in place of a fixed joint and in front of a mechanical one)
             "tree_arm_flex": lambda: robots.tree_arm_flexible(False), "tree_arm_flex_ff": lambda: robots.tree_arm_flexible(True),
             "crane_walker": robots.crane_walker, "biped": robots.biped, "biped_torso": lambda: robots.biped(True)}[robot]()
    quad = robot in ("crane_walker", "biped", "biped_torso")
    if quad:
        from jiminy_amd import codegen
        assert codegen.quad_structure(model) is not None
    B, dt = (203, 2.5e-4) if quad else (192, 5e-4)
    st = sample_states(model, B, seed=21, base_height=(0.55, 0.75) if robot.startswith("biped") else (0.3, 0.6),
                       grounded_fraction=0.5)
    ref = alloc_soa(model, B)
    for k in ("q", "v", "command"):
        if st[k].shape[0]:
            ref[k][:] = st[k]
    oracle_batch(model, ref, "start")
    loop = ReferenceFixedStepLoop(dt)   # the reference's sub-step rule: opens with a 1 us step (engine.cc:1176)
    eng = _engine(model, B, torch.float64, "runge_kutta_4", dt)
    if model.nmotors:
        eng.set_command(torch.from_numpy(st["command"]))
    eng.start(torch.from_numpy(st["q"]), torch.from_numpy(st["v"]))
    for i in range(8):
        oracle_engine_step(model, ref, loop, dt, "runge_kutta_4", command_changed=False)
        eng.step(dt)
    ok = (ref["status"][0] & 1) == 0
    assert ok.sum() > B // 2
    for k in OUTS + ("contact", "energy", "joint_forces", "centroidal", "f_external"):
        assert rel_err(eng.field(k).cpu().numpy(), ref[k], ok) < 1e-9, k
    assert np.array_equal(eng.status.cpu().numpy()[ok], ref["status"][0][ok])


@pytest.mark.parametrize("name", ["anymal", "atlas", "tree_arm_ff", "cartpole"])
def test_emitted_rows_of_float64_and_float32_kernels_agree(gpu_device, name):
    """engine._output_self_test: the rows a launch emits (sensors, extra terms), float64 against the separately compiled
    float32 kernels of the same library -- the check that would have caught the mis-compiled output pass of DESIGN.md section
    4.7 (third case; measured 1.6 on that build) without an oracle.  Sound builds: float32 round-off."""
    from jiminy_amd import codegen, engine as engine_mod
    from tests import robots
    model = robots.tree_arm(True) if name == "tree_arm_ff" else load_builtin(name)
    assert engine_mod._output_self_test(model, codegen.preferred_variant(model), gpu_device) < 5e-3


def test_a_library_with_garbage_in_its_emitted_rows_is_replaced(gpu_device, monkeypatch):
    """`_verified_library` treats a variant whose emitted rows disagree like one that fails the step self-test: the next
    build variant is taken, with a warning (the disagreement is injected here: no build of today shows one)."""
    import warnings
    from jiminy_amd import codegen, engine as engine_mod
    model = load_builtin("cartpole")
    real = engine_mod._output_self_test
    monkeypatch.setattr(engine_mod, "_VERIFIED", {})
    monkeypatch.setattr(engine_mod, "_output_self_test", lambda m, v, d: 1.6 if v == 0 else real(m, v, d))
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        eng = _engine(model, 32, torch.float64, "runge_kutta_4", 1e-3)
    chosen = engine_mod._VERIFIED[(model.topology_hash(), torch.float64)]
    assert chosen == 1 and eng._lib.path == codegen.lib_path(model, 1)
    assert any("failed the kernel self-test" in str(w.message) and "emitted rows" in str(w.message) for w in caught)


@pytest.mark.parametrize("name", ["crane_walker", "tree_arm"])
def test_library_self_test_guards_against_miscompiled_builds(gpu_device, monkeypatch, name):
    """Every HIP library is checked on first use (engine._verified_library): a build whose in-loop
    evaluation disagrees with its peeled copy is replaced by the next build variant.  The engine is started from
    variant 1 here -- the compiler's default register allocators, under which `tree_arm`'s and (up to round 3)
    `crane_walker`'s step kernels have been mis-compiled by hipcc 7.2 (variant 0, the default
    of every topology, is the basic SGPR allocator that repairs them); whichever variant ends up selected, the engine
    must match the oracle."""
    from jiminy_amd import codegen, engine as engine_mod
    from tests import robots
    model = robots.crane_walker() if name == "crane_walker" else robots.tree_arm(False)
    first = 1
    monkeypatch.setenv("JIMINY_AMD_BUILD_VARIANT", str(first))
    monkeypatch.setattr(engine_mod, "_VERIFIED", {})
    B, dt = 64, 2.5e-4
    st = sample_states(model, B, seed=5, base_height=(0.3, 0.6), grounded_fraction=0.5)
    ref = alloc_soa(model, B)
    for k in ("q", "v", "command"):
        ref[k][:] = st[k]
    oracle_batch(model, ref, "start")
    loop = ReferenceFixedStepLoop(dt)   # the reference's sub-step rule: opens with a 1 us step (engine.cc:1176)
    import warnings
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        eng = _engine(model, B, torch.float64, "runge_kutta_4", dt)
    chosen = engine_mod._VERIFIED[(model.topology_hash(), torch.float64)]
    assert 0 <= chosen < len(codegen.BUILD_VARIANTS)
    if chosen != first:
        assert any("failed the kernel self-test" in str(w.message) for w in caught)
    assert eng._lib.path == codegen.lib_path(model, chosen)
    eng.set_command(torch.from_numpy(st["command"]))
    eng.start(torch.from_numpy(st["q"]), torch.from_numpy(st["v"]))
    for i in range(4):
        oracle_engine_step(model, ref, loop, dt, "runge_kutta_4", command_changed=False)
        eng.step(dt)
    ok = (ref["status"][0] & 1) == 0
    for k in OUTS:
        assert rel_err(eng.field(k).cpu().numpy(), ref[k], ok) < 1e-9, k
    # the self-test itself: a sound library agrees with itself to round-off
    assert engine_mod._library_self_test(model, chosen, torch.float64, eng.device) < 1e-9


def test_anymal_generic_lane_kernel_matches_oracle(gpu_device, monkeypatch):
    """The one-robot-per-lane kernel (used for topologies without the 4-limb structure) on ANYmal."""
    monkeypatch.setenv("JM_KERNEL_VARIANT", "lane")
    model = load_builtin("anymal")
    B, dt = 128, 1e-3
    st = sample_states(model, B, seed=11)
    ref = alloc_soa(model, B)
    for k in ("q", "v", "command"):
        ref[k][:] = st[k]
    oracle_batch(model, ref, "start")
    loop = ReferenceFixedStepLoop(dt)   # the reference's sub-step rule: opens with a 1 us step (engine.cc:1176)
    eng = _engine(model, B, torch.float64, "runge_kutta_4", dt)
    eng.set_command(torch.from_numpy(st["command"]))
    eng.start(torch.from_numpy(st["q"]), torch.from_numpy(st["v"]))
    for i in range(10):
        oracle_engine_step(model, ref, loop, dt, "runge_kutta_4", command_changed=False)
        eng.step(dt)
    ok = (ref["status"][0] & 1) == 0
    for k in OUTS + ("energy", "joint_forces", "centroidal"):
        assert rel_err(eng.field(k).cpu().numpy(), ref[k], ok) < 1e-9, k


def _run_1000(model, st, dt, gpu_device):
    B = st["q"].shape[1]
    ref = alloc_soa(model, B)
    for k in ("q", "v", "command"):
        ref[k][:] = st[k]
    oracle_batch(model, ref, "start")
    loop = ReferenceFixedStepLoop(dt)   # the reference's sub-step rule: opens with a 1 us step (engine.cc:1176)
    eng = _engine(model, B, torch.float64, "runge_kutta_4", dt)
    eng.set_command(torch.from_numpy(st["command"]))
    eng.start(torch.from_numpy(st["q"]), torch.from_numpy(st["v"]))
    valid = np.ones(B, dtype=bool)
    contact = np.zeros(B, dtype=bool)
    err = np.zeros(B)
    for i in range(1000):
        oracle_engine_step(model, ref, loop, dt, "runge_kutta_4", command_changed=False)
        eng.step(dt)
        contact |= np.abs(ref["contact_forces"]).sum(axis=0) > 0
        if (i + 1) % 50 == 0:
            valid &= ref["status"][0] == 0
            got = eng.field("a").cpu().numpy()
            num = np.abs(got - ref["a"]).max(axis=0)
            den = np.maximum(np.abs(ref["a"]).max(axis=0), 1.0)
            err = np.maximum(err, np.where(valid, num / den, 0.0))
    return err, valid, contact


def _open_bounds(model):
    # Position bounds are numeric parameters of the model (same topology, same library): they are
    # opened because the reference enforces them through its constraint solver, which is outside
    # this path; lanes are then valid as long as they stay finite.
    mask = model.bounded_position_mask()
    model.position_lower[mask] = -np.inf
    model.position_upper[mask] = np.inf


def test_anymal_1000_steps_parity_free_motion(gpu_device):
    """north_star bar, strict form: <= 1e-5 relative on generalised accelerations over 1000 RK4
    steps of dt = 1e-3 (fp64), first 256 lanes, every lane.  The robots are dropped from 6-7 m with
    the full random joint states / held commands, so the whole second is articulated free motion
    (ABA + motors + integrator on SE(3), no ground contact)."""
    model = load_builtin("anymal")
    _open_bounds(model)
    st = sample_states(model, 256, seed=0, base_height=(6.0, 7.0), grounded_fraction=0.0)
    err, valid, contact = _run_1000(model, st, 1e-3, gpu_device)
    assert not contact.any()
    assert valid.sum() >= 250, valid.sum()
    assert err[valid].max() <= 1e-5, err[valid].max()


def test_anymal_1000_steps_parity_with_contacts(gpu_device):
    """Contact-rich trajectories (robots landing on their feet with the reference's default
    k = 1e6 N/m, c = 2e3 N.s/m ground) are chaotic: the same oracle evaluated with a different
    summation order already spreads to ~1e-4 on its worst lane after 1000 steps
    (tests/test_hostemu_vs_oracle.py::test_long_horizon_sensitivity).  The bar is therefore
    statistical here: median <= 1e-8 and >= 90 % of the lanes within 1e-5.
    dt = 5e-4 because explicit RK4 at dt = 1e-3 is outside its stability region for ANYmal's
    0.59 kg shank on this ground (oracle and GPU blow up identically within ~150 steps)."""
    model = load_builtin("anymal")
    _open_bounds(model)
    st = sample_states(model, 256, seed=0, command_fraction=0.05, joint_vel_std=0.1,
                       base_twist_std=0.05, joint_range=0.4)
    err, valid, contact = _run_1000(model, st, 5e-4, gpu_device)
    assert (valid & contact).sum() >= 128, (valid & contact).sum()
    e = err[valid]
    assert np.median(e) <= 1e-8, np.median(e)
    assert (e <= 1e-5).mean() >= 0.9, (e <= 1e-5).mean()


def test_anymal_free_running_window_at_the_benchmarked_step(gpu_device):
    """The BENCHMARKED configuration free-running (no teacher forcing): first 256 lanes of the bench batch
    (`sample_states(seed=0)`, a quarter start in ground contact), RK4 at dt = 1e-3, default ground.  Oracle and device
    integrate independently from the same initial state through the engine's own `step` (opening microsecond step
    included); every lane is compared at EVERY step for as long as the oracle's lane stays in the region where the
    comparison means something (|v| <= 1e3, |a| <= 1e9, status 0: explicit RK4 is outside its stability region on this
    ground, lanes that touch down leave it within ~150 steps).  Bars: the median window is >= 100
    steps; over its window a lane's worst relative error on the generalised accelerations has median <= 1e-8 and
    >= 90 % of the lanes stay within the north-star 1e-5 (an unstable integration amplifies the round-off of two
    summation orders at the rate it amplifies the state: the host emulation of the same kernels against the oracle
    gives 97 % on 128 lanes)."""
    model = load_builtin("anymal")
    _open_bounds(model)
    B, dt, steps = 256, 1e-3, 300
    st = sample_states(model, 65536, seed=0)
    ref = alloc_soa(model, B)
    for k in ("q", "v", "command"):
        ref[k][:] = st[k][:, :B]
    oracle_batch(model, ref, "start")
    loop = ReferenceFixedStepLoop(dt)
    eng = _engine(model, B, torch.float64, "runge_kutta_4", dt)
    eng.set_command(torch.from_numpy(np.ascontiguousarray(st["command"][:, :B])))
    eng.start(torch.from_numpy(np.ascontiguousarray(st["q"][:, :B])), torch.from_numpy(np.ascontiguousarray(st["v"][:, :B])))
    alive = np.ones(B, dtype=bool)
    window = np.zeros(B, dtype=int)
    worst = np.zeros(B)
    contact = np.zeros(B, dtype=bool)
    for i in range(steps):
        oracle_engine_step(model, ref, loop, dt, "runge_kutta_4", command_changed=False)
        eng.step(dt)
        alive &= (ref["status"][0] == 0) & np.isfinite(ref["a"]).all(axis=0) & (np.abs(ref["v"]).max(axis=0) <= 1e3) \
            & (np.abs(ref["a"]).max(axis=0) <= 1e9)
        got = eng.field("a").cpu().numpy()
        e = np.abs(got - ref["a"]).max(axis=0) / np.maximum(np.abs(ref["a"]).max(axis=0), 1.0)
        e = np.where(np.isfinite(e), e, np.inf)
        worst = np.where(alive, np.maximum(worst, e), worst)
        window += alive
        contact |= alive & (np.abs(ref["contact_forces"]).sum(axis=0) > 0)
    print(f"free-running dt=1e-3: median window {np.median(window):.0f} steps (min {window.min()}), {int(contact.sum())} lanes touched "
          f"the ground inside their window, {int(alive.sum())} still inside after {steps} steps; worst-lane error median "
          f"{np.median(worst):.1e}, p90 {np.quantile(worst, 0.9):.1e}, within 1e-5: {(worst <= 1e-5).mean():.3f}")
    assert np.median(window) >= 100, np.median(window)
    assert contact.sum() >= B // 4, contact.sum()
    assert np.median(worst) <= 1e-8, np.median(worst)
    assert (worst <= 1e-5).mean() >= 0.9, (worst <= 1e-5).mean()


def _teacher_forced_run(model, st, B, dt, steps, step_and_compare):
    """Oracle trajectory of the first `B` lanes of the seeded batch `st` (RK4, `dt`), re-seeding a lane
    from the next unused lane of the batch once it has numerically blown up (the bench's auto-reset).
    `step_and_compare(ref, ok0, advance)` is called once per step with the oracle's SoA arrays at
    the step start, the mask of sane lanes, and a callable that advances the oracle by one step."""
    from oracle.oracle_py import OracleEngine
    from tests.helpers import oracle_io
    ref = alloc_soa(model, B)
    for k in ("q", "v", "command"):
        ref[k][:] = st[k][:, :B]
    orc = OracleEngine(model)
    io = oracle_io(ref)
    orc.batch_run("start", io)
    pool = B

    def sane():
        return (ref["status"][0] == 0) & np.isfinite(ref["a"]).all(axis=0) & np.isfinite(ref["q"]).all(axis=0) \
            & (np.abs(ref["v"]).max(axis=0) <= 1e3) & (np.abs(ref["a"]).max(axis=0) <= 1e9)

    n_reseeded = 0
    for _ in range(steps):
        ok0 = sane()
        for lane in np.where(~ok0)[0]:
            for k in ("q", "v", "command"):
                ref[k][:, lane] = st[k][:, pool]
            pool += 1
            n_reseeded += 1
            orc.batch_run("start", io, lanes=(int(lane), int(lane) + 1))
        ok0 = sane()
        step_and_compare(ref, ok0, lambda: orc.batch_run("step", io, solver="runge_kutta_4", dt=dt, n_substeps=1,
                                                         command_changed=False), sane)
    return n_reseeded


def test_anymal_1000_steps_teacher_forced_at_the_benchmarked_step(gpu_device):
    """north_star bar on the BENCHMARKED configuration, without chaos amplification: the oracle
    integrates the first 256 lanes of the bench batch (`sample_states(seed=0)`: a quarter of them
    start in ground contact, default ground k = 1e6, c = 2e3) for 1000 RK4 steps at dt = 1e-3, and at
    EVERY step the device restarts from the oracle's state (q, v, a, command): one `jm_batch_dynamics`
    evaluation and one `jm_batch_step` are compared with the oracle's.  Every lane whose oracle state
    is sane must agree within 1e-5 relative on the generalised accelerations (observed ~1e-11).
    "Sane" = the lane has not numerically blown up: explicit RK4 at dt = 1e-3 is outside its stability
    region on this ground, lanes that touch down diverge to 1e70 within ~200
    steps in the oracle and on the device alike, and the device `sincos` is specified for |x| < 1e5
    (jm_math.h): lanes are compared while |v| <= 1e3 and |a| <= 1e9, then re-seeded from the next
    unused lane of the bench batch, like the bench's own auto-reset."""
    _teacher_forced_test(gpu_device, "anymal", 256, 1e-3, 1000, 65536, 50000)


def test_atlas_teacher_forced_at_the_benchmarked_step(gpu_device):
    """The same per-step comparison on BASELINE's Atlas configuration (`bench.py --model atlas --dt 2.5e-4`: 30 motors,
    32 contact points, trunk tree + padded limbs on the branch-parallel kernel), first 64 lanes of its bench batch,
    400 steps."""
    _teacher_forced_test(gpu_device, "atlas", 64, 2.5e-4, 400, 4096, 2000)


def _teacher_forced_test(gpu_device, name, B, dt, steps, pool, min_contact_steps):
    model = load_builtin(name)
    _open_bounds(model)
    st = sample_states(model, pool, seed=0)
    eng = _engine(model, B, torch.float64, "runge_kutta_4", dt)
    eng.set_command(torch.from_numpy(np.ascontiguousarray(st["command"][:, :B])))
    eng.start(torch.from_numpy(np.ascontiguousarray(st["q"][:, :B])), torch.from_numpy(np.ascontiguousarray(st["v"][:, :B])))
    eng.step(dt)     # (the opening 1 us + 999 us interval of the simulation: every step below restarts from the oracle's state)
    worst = {"dynamics": np.zeros(B), "a": np.zeros(B), "q": np.zeros(B), "v": np.zeros(B)}
    count = {"lane_steps": 0, "contact_steps": 0}

    def step_and_compare(ref, ok0, advance, sane):
        q0, v0, a0 = (torch.from_numpy(ref[k]).to(gpu_device) for k in ("q", "v", "a"))
        eng.set_command(torch.from_numpy(ref["command"]))
        # (a) one evaluation at the oracle's state
        a_dev = eng.compute_robots_dynamics(0.0, q0, v0).cpu().numpy()
        e = np.abs(a_dev - ref["a"]).max(axis=0) / np.maximum(np.abs(ref["a"]).max(axis=0), 1.0)
        assert np.isfinite(e[ok0]).all()
        worst["dynamics"] = np.maximum(worst["dynamics"], np.where(ok0, e, 0.0))
        # (b) one integrator step from the oracle's state
        for k, x in (("q", q0), ("v", v0), ("a", a0)):
            eng.field(k).copy_(x)
        eng.step(dt)
        advance()
        ok = ok0 & sane()
        for k in ("a", "q", "v"):
            got = eng.field(k).cpu().numpy()
            e = np.abs(got - ref[k]).max(axis=0) / np.maximum(np.abs(ref[k]).max(axis=0), 1.0)
            assert np.isfinite(e[ok]).all(), k
            worst[k] = np.maximum(worst[k], np.where(ok, e, 0.0))
        count["lane_steps"] += int(ok.sum())
        count["contact_steps"] += int((ok & (np.abs(ref["contact_forces"]).sum(axis=0) > 0)).sum())

    n_reseeded = _teacher_forced_run(model, st, B, dt, steps, step_and_compare)
    print(f"teacher-forced {name} dt={dt:g}: {count['lane_steps']} lane-steps compared ({count['contact_steps']} in ground "
          f"contact, {n_reseeded} lanes re-seeded); worst lane " + ", ".join(f"{k} {v.max():.1e}" for k, v in worst.items()))
    assert count["lane_steps"] >= 0.95 * B * steps, count
    assert count["contact_steps"] >= min_contact_steps, count
    for k, v in worst.items():
        assert v.max() <= 1e-5, (k, v.max())
    assert np.median(worst["a"]) <= 1e-9, np.median(worst["a"])


@pytest.mark.parametrize("name", ["anymal", "atlas"])
def test_fp32_engines_of_the_branch_parallel_topologies(gpu_device, name):
    """float32 batches of the big robots (`bench.py --dtype f32`): the library self-test accepts the float32 kernels (its
    Runge-Kutta leg is conditioned for float64 only) and ten RK4 steps in free flight stay within float32 accuracy of the
    float64 engine."""
    model = load_builtin(name)
    B, dt = 64, 2.5e-4
    st = sample_states(model, B, seed=3, base_height=(2.0, 2.5), grounded_fraction=0.0, command_fraction=0.3)
    out = {}
    for dtype in (torch.float64, torch.float32):
        eng = _engine(model, B, dtype, "runge_kutta_4", dt)
        eng.set_command(torch.from_numpy(st["command"]).to(dtype))
        eng.start(torch.from_numpy(st["q"]).to(dtype), torch.from_numpy(st["v"]).to(dtype))
        for _ in range(10):
            eng.step(dt)
        assert int((eng.status & 1).sum()) == 0
        out[dtype] = (eng.field("q").double().cpu().numpy(), eng.field("v").double().cpu().numpy())
    assert rel_err(out[torch.float32][0], out[torch.float64][0]) < 1e-4
    assert rel_err(out[torch.float32][1], out[torch.float64][1]) < 5e-3


def test_fp32_tolerance_study_cartpole(gpu_device):
    """Config 2 of BASELINE.json: cartpole batch 4096, ABA + RK4, fp64 vs fp32."""
    model = load_builtin("cartpole")
    B, dt = 4096, 1e-3
    st = sample_states(model, B, seed=5)
    outs = {}
    for dtype in (torch.float64, torch.float32):
        eng = _engine(model, B, dtype, "runge_kutta_4", dt)
        eng.set_command(torch.from_numpy(st["command"]).to(dtype))
        eng.start(torch.from_numpy(st["q"]).to(dtype), torch.from_numpy(st["v"]).to(dtype))
        for _ in range(100):
            eng.step(dt)
        outs[dtype] = eng.field("a").double().cpu().numpy()
    err = rel_err(outs[torch.float32], outs[torch.float64])
    assert err < 5e-3, err


def test_compute_robots_dynamics_and_reset_lanes(gpu_device):
    model = load_builtin("anymal")
    B = 128
    st = sample_states(model, B, seed=7)
    eng = _engine(model, B, torch.float64, "runge_kutta_4", 1e-3)
    eng.set_command(torch.from_numpy(st["command"]))
    q, v = torch.from_numpy(st["q"]), torch.from_numpy(st["v"])
    eng.start(q, v)
    a0 = eng.field("a").clone()
    a = eng.compute_robots_dynamics(0.0, q, v)
    # the output-free copy of the evaluation is compiled separately (other FMA contractions):
    # agreement at round-off; repeatability of one entry point is bitwise (reset below)
    assert rel_err(a.cpu().numpy(), a0.cpu().numpy()) < 1e-12
    for _ in range(5):
        eng.step(1e-3)
    mask = torch.zeros(B, dtype=torch.bool)
    mask[::2] = True
    a_before = eng.field("a").clone()
    eng.reset_lanes(mask, q, v)
    a_after = eng.field("a")
    assert torch.equal(a_after[:, mask.cuda()], a0[:, mask.cuda()])
    assert torch.equal(a_after[:, ~mask.cuda()], a_before[:, ~mask.cuda()])


def test_control_flow_errors(gpu_device):
    from jiminy_amd._lib import BadControlFlow
    model = load_builtin("cartpole")
    eng = BatchedEngine(model, 64)
    with pytest.raises(BadControlFlow):
        eng.step(1e-3)
    eng.start(np.array([0.0, 1.0, 0.0]), np.zeros(2))
    with pytest.raises(BadControlFlow):
        eng.start(np.array([0.0, 1.0, 0.0]), np.zeros(2))
    with pytest.raises(BadControlFlow):
        eng.set_options({"stepper": {"dtMax": 1e-3}})
    eng.stop()
    eng.set_options({"stepper": {"odeSolver": "runge_kutta_dopri"}})   # the reference's default solver
    with pytest.raises(NotImplementedError):
        eng.set_options({"stepper": {"odeSolver": "runge_kutta_fehlberg"}})
    with pytest.raises(ValueError):
        eng.set_options({"contacts": {"model": "impulse"}})   # engine.cc:2741-2747
    eng.set_options({"contacts": {"model": "constraint"}})     # the reference's default contact model
    eng.set_options({"contacts": {"model": "spring_damper"}})
    with pytest.raises(ValueError):
        eng.set_options({"stepper": {"tolRel": 0.0}})


@pytest.mark.parametrize("name", ["anymal", "atlas"])
@pytest.mark.parametrize("solver", ["runge_kutta_4", "euler_explicit"])
def test_multi_substep_launches_match_oracle(gpu_device, name, solver):
    """One launch = several integrator steps (the controller period of the environments:
    `dtMax` < `controllerUpdatePeriod`), with and without the a(t+) refresh after a command change,
    on a ragged batch (B % 64 != 0, B % 16 != 0)."""
    model = load_builtin(name)
    B = 77
    dt = 5e-4 if name == "anymal" else 2.5e-4
    n_sub = 5
    st = sample_states(model, B, seed=11, base_height=(0.9, 1.1) if name == "atlas" else (0.45, 0.65))
    ref = alloc_soa(model, B)
    for k in ("q", "v", "command"):
        ref[k][:] = st[k]
    oracle_batch(model, ref, "start")
    loop = ReferenceFixedStepLoop(dt)
    eng = BatchedEngine(model, B, dtype=torch.float64, extra_outputs=EXTRA)
    eng.set_options({"stepper": {"odeSolver": solver, "dtMax": dt, "controllerUpdatePeriod": n_sub * dt,
                                 "sensorsUpdatePeriod": n_sub * dt}, "contacts": {"model": "spring_damper"}})
    eng.set_command(torch.from_numpy(st["command"]))
    eng.start(torch.from_numpy(st["q"]), torch.from_numpy(st["v"]))
    rng = np.random.default_rng(0)
    for i in range(4):
        changed = i % 2 == 0
        if changed:
            cmd = st["command"] * rng.uniform(0.5, 1.0)
            ref["command"][:] = cmd
            eng.set_command(torch.from_numpy(cmd))
        n_done = oracle_engine_step(model, ref, loop, n_sub * dt, solver, command_changed=changed)
        assert n_done == (n_sub + 1 if i == 0 else n_sub)       # the opening microsecond step of the simulation
        eng.step(n_sub * dt)
    torch.cuda.synchronize()
    ok = (ref["status"][0] & 1) == 0
    assert ok.sum() > 0.5 * B
    for k in OUTS + ("u", "energy", "f_external"):
        assert rel_err(eng.field(k).cpu().numpy(), ref[k], ok) < 1e-8, k
    assert abs(eng.stepper_state.t - 4 * n_sub * dt) < 1e-12 and eng.stepper_state.iter == 4 * n_sub + 1


def _dopri_pair(model, B, st, step_dt, n_steps, tol_rel, tol_abs, dt_max=0.02, ctrl=0.0):
    """Engine with the adaptive solver vs the oracle's restatement of the reference's adaptive loop,
    same breakpoints; returns (engine, ref arrays, oracle adaptive state)."""
    from jiminy_amd.engine import plan_breakpoints
    from oracle.oracle_py import OracleEngine, adaptive_state
    from tests.helpers import oracle_io
    ref = alloc_soa(model, B)
    for k in ("q", "v", "command"):
        if st[k].shape[0]:
            ref[k][:] = st[k]
    orc = OracleEngine(model)
    io = oracle_io(ref)
    orc.batch_run("start", io)
    ad = adaptive_state(B)
    eng = BatchedEngine(model, B, dtype=torch.float64, extra_outputs=EXTRA)
    eng.set_options({"stepper": {"odeSolver": "runge_kutta_dopri", "tolRel": tol_rel, "tolAbs": tol_abs,
                                 "dtMax": dt_max, "controllerUpdatePeriod": ctrl, "sensorsUpdatePeriod": ctrl}, "contacts": {"model": "spring_damper"}})
    if model.nmotors:
        eng.set_command(torch.from_numpy(st["command"]))
    eng.start(torch.from_numpy(st["q"]), torch.from_numpy(st["v"]))
    t, t_err = 0.0, 0.0
    for _ in range(n_steps):
        intervals, t_end, t_err = plan_breakpoints(t, t_err, step_dt, eng.get_options())
        for i, (t_next, cmd, sens) in enumerate(intervals):
            orc.batch_run_dopri(io, ad, t_next, tol_rel=tol_rel, tol_abs=tol_abs, dt_max=dt_max,
                                new_step=(i == 0), command_changed=False, update_sensors=sens)
        t = t_end
        eng.step(step_dt)
    torch.cuda.synchronize()
    return eng, ref, ad


@pytest.mark.parametrize("robot", ["pendulum", "double_pendulum", "cartpole", "pendulum_flexible", "tree_arm_flex"])
def test_adaptive_dopri_matches_oracle_small_robots(gpu_device, robot):
    """`odeSolver = "runge_kutta_dopri"` (reference default) on the one-robot-per-lane kernel: every
    lane carries its own step size.  The accept / reject decisions are discontinuous in the error
    estimate, so agreement is statistical: nearly all lanes follow the oracle's step sequence
    (round-off agreement), the rest stay within the integration tolerance."""
    from tests import robots
    model = {"pendulum": robots.pendulum, "double_pendulum": robots.double_pendulum,
             "cartpole": lambda: load_builtin("cartpole"), "pendulum_flexible": robots.pendulum_flexible,
             "tree_arm_flex": lambda: robots.tree_arm_flexible(False)}[robot]()
    B = 192
    st = sample_states(model, B, seed=31)
    eng, ref, ad = _dopri_pair(model, B, st, 0.01, 30, 1e-6, 1e-7)
    err = np.abs(eng.field("q").cpu().numpy() - ref["q"]).max(axis=0)
    assert np.median(err) < 1e-10 and err.max() < 1e-4, (np.median(err), err.max())
    ss = eng.stepper_state
    same = (ss.iter_lanes.cpu().numpy() == ad["iter"]) & (ss.iter_failed_lanes.cpu().numpy() == ad["iter_failed"])
    assert same.mean() > 0.9
    assert np.allclose(ss.dt_largest.cpu().numpy()[same], ad["dt_largest"][same], rtol=1e-6)
    if "flex" in robot:
        # (`tree_arm`'s seeded states leave a joint outside its bounds -- flagged alike on both sides --, the flexible
        # pendulum's stiffest lanes may give up alike)
        assert np.array_equal(eng.status.cpu().numpy().reshape(-1)[same], ref["status"].reshape(-1)[same])
    else:
        assert int(eng.status.abs().sum()) == 0


def test_adaptive_dopri_anymal_free_flight_and_energy(gpu_device):
    """Branch-parallel kernel under the adaptive solver: ANYmal in free flight (no contact, zero
    command, unbounded joints so that no lane is flagged), tight tolerances: matches the oracle and
    conserves the total energy."""
    model = load_builtin("anymal")
    mask = model.bounded_position_mask()
    model.position_lower[mask] = -np.inf
    model.position_upper[mask] = np.inf
    B = 64
    st = sample_states(model, B, seed=12, base_height=(5.0, 6.0), grounded_fraction=0.0, command_fraction=0.0)
    eng, ref, ad = _dopri_pair(model, B, st, 5e-3, 20, 1e-8, 1e-9)
    for k in ("q", "v"):
        assert rel_err(eng.field(k).cpu().numpy(), ref[k]) < 1e-7, k
    e_gpu = eng.field("energy").cpu().numpy().sum(axis=0)
    e_ref = ref["energy"].sum(axis=0)
    assert np.abs(e_gpu - e_ref).max() < 1e-6 * np.abs(e_ref).max()
    assert int(eng.stepper_state.iter) >= 20 and eng.adaptive_attempts >= 1


@pytest.mark.parametrize("B", [1, 3, 65])
def test_adaptive_dopri_ragged_and_tiny_batches(gpu_device, B):
    """Batches that fill neither a quad-wave (16 robots) nor a block under the persistent adaptive stepper: the tail
    robots are integrated, the padding lanes stay out of memory (the engine's fields carry no slack: a stray write would
    land in a neighbouring tensor and show up in the comparison with the oracle)."""
    model = load_builtin("anymal")
    st = sample_states(model, B, seed=23, base_height=(0.5, 0.7), grounded_fraction=0.5, command_fraction=0.2)
    eng, ref, ad = _dopri_pair(model, B, st, 5e-3, 4, 1e-7, 1e-8)
    dev_status = eng.status.cpu().numpy().reshape(-1)
    ok = ((ref["status"][0] | dev_status) & 9) == 0
    ss = eng.stepper_state
    same = ok & (ss.iter_lanes.cpu().numpy() == ad["iter"]) & (ss.iter_failed_lanes.cpu().numpy() == ad["iter_failed"])
    assert same.sum() >= max(1, int(0.7 * B)), (ss.iter_lanes.cpu().numpy(), ad["iter"])
    for k in ("q", "v", "a"):
        assert rel_err(eng.field(k).cpu().numpy(), ref[k], same) < 1e-6, k
    assert abs(ss.t - 0.02) < 1e-12


@pytest.mark.parametrize("name", ["crane_walker", "biped_torso"])
def test_adaptive_dopri_crane_walker_trunk_tree_and_ragged_limbs(gpu_device, name):
    """The persistent stepper on the authored robot with a prismatic / unaligned trunk tree and limbs of 3/3/2/2 joints,
    and on the biped with limbs of 3/3/1/0 joints: landings on the spring-damper ground, against the oracle."""
    from tests import robots
    model = robots.crane_walker() if name == "crane_walker" else robots.biped(True)
    B = 48
    st = sample_states(model, B, seed=19, base_height=(0.5, 0.8), grounded_fraction=0.4, command_fraction=0.2)
    eng, ref, ad = _dopri_pair(model, B, st, 5e-3, 6, 1e-7, 1e-8)
    dev_status = eng.status.cpu().numpy().reshape(-1)
    ok = ((ref["status"][0] | dev_status) & 9) == 0
    assert ok.mean() > 0.9, (np.unique(ref["status"][0], return_counts=True), np.unique(dev_status, return_counts=True))
    ss = eng.stepper_state
    same = ok & (ss.iter_lanes.cpu().numpy() == ad["iter"]) & (ss.iter_failed_lanes.cpu().numpy() == ad["iter_failed"])
    assert same.mean() > 0.8
    for k in ("q", "v"):
        assert rel_err(eng.field(k).cpu().numpy(), ref[k], same) < 1e-7, k


def test_adaptive_dopri_atlas_long_limbs(gpu_device):
    """The persistent stepper on a topology with long limbs and a trunk tree (Atlas: the stage velocities and commands of
    the evaluation travel through the stage buffer): free flight and landings, tight tolerances, against the oracle."""
    model = load_builtin("atlas")
    B = 32
    st = sample_states(model, B, seed=14, base_height=(1.0, 1.3), grounded_fraction=0.3, command_fraction=0.1)
    eng, ref, ad = _dopri_pair(model, B, st, 5e-3, 6, 1e-7, 1e-8)
    dev_status = eng.status.cpu().numpy().reshape(-1)
    ok = ((ref["status"][0] | dev_status) & 9) == 0
    assert ok.mean() > 0.9, (np.unique(ref["status"][0], return_counts=True), np.unique(dev_status, return_counts=True))
    ss = eng.stepper_state
    same = ok & (ss.iter_lanes.cpu().numpy() == ad["iter"]) & (ss.iter_failed_lanes.cpu().numpy() == ad["iter_failed"])
    assert same.mean() > 0.8
    for k in ("q", "v"):
        assert rel_err(eng.field(k).cpu().numpy(), ref[k], same) < 1e-7, k
    assert abs(ss.t - 0.03) < 1e-12


def test_adaptive_dopri_with_controller_breakpoints_and_contacts(gpu_device):
    """Default tolerances (tolRel 1e-4, tolAbs 1e-5), 5 ms controller / sensor breakpoints, ANYmal
    landing on the spring-damper ground: lanes reach every breakpoint, none is lost, the state stays
    within the integration tolerance of the oracle for most lanes."""
    model = load_builtin("anymal")
    B = 128
    st = sample_states(model, B, seed=13, base_height=(0.5, 0.6), grounded_fraction=0.5, command_fraction=0.1)
    eng, ref, ad = _dopri_pair(model, B, st, 5e-3, 8, 1e-4, 1e-5, ctrl=5e-3)
    ok = (ref["status"][0] & 9) == 0
    stt = eng.status.cpu().numpy()
    assert ((stt & 9) == 0)[ok].mean() > 0.95
    err = np.abs(eng.field("q").cpu().numpy() - ref["q"]).max(axis=0)[ok & ((stt & 9) == 0)]
    assert np.median(err) < 1e-6 and (err < 1e-2).mean() > 0.95, (np.median(err), err.max())
    assert abs(eng.stepper_state.t - 0.04) < 1e-12


@pytest.mark.parametrize("name,B", [("anymal", 1), ("anymal", 3), ("anymal", 65), ("cartpole", 1), ("cartpole", 67)])
def test_ragged_and_tiny_batches(gpu_device, name, B):
    """Batch sizes that fill neither a quad-wave (16 robots) nor a block: the tail lanes must be
    computed, the padding lanes must not touch memory (guard rows around every field)."""
    model = load_builtin(name)
    dt = 1e-3
    st = sample_states(model, B, seed=17)
    ref = alloc_soa(model, B)
    for k in ("q", "v", "command"):
        ref[k][:] = st[k]
    oracle_batch(model, ref, "start")
    loop = ReferenceFixedStepLoop(dt)   # the reference's sub-step rule: opens with a 1 us step (engine.cc:1176)
    eng = _engine(model, B, torch.float64, "runge_kutta_4", dt)
    eng.set_command(torch.from_numpy(st["command"]))
    eng.start(torch.from_numpy(st["q"]), torch.from_numpy(st["v"]))
    for i in range(5):
        oracle_engine_step(model, ref, loop, dt, "runge_kutta_4", command_changed=False)
        eng.step(dt)
    ok = (ref["status"][0] & 1) == 0
    for k in OUTS:
        assert rel_err(eng.field(k).cpu().numpy(), ref[k], ok) < 1e-9, k
    assert np.array_equal(eng.status.cpu().numpy()[ok], ref["status"][0][ok])


def test_empty_batch_is_rejected(gpu_device):
    with pytest.raises(ValueError, match="batch size must be positive"):
        BatchedEngine(load_builtin("cartpole"), 0)


@pytest.mark.parametrize("name,blk,reps,dt", [("anymal", 256, 256, 1e-3),      # BASELINE config 3: B = 65 536
                                              ("atlas", 64, 512, 2.5e-4),       # config 4, whole batch: B = 32 768
                                              ("atlas", 64, 64, 2.5e-4)])       # config 4, one GPU's share: B = 4 096
def test_full_size_batch_replica_invariance_and_repeatability(gpu_device, name, blk, reps, dt):
    """BASELINE sizes (ANYmal B = 65 536; Atlas B = 32 768 and its per-GPU share 4 096: the large-batch and the
    one-wave-per-block launch forms), checked through size-independent properties: the batch is `reps` replicas of one
    seeded block, so (i) every replica must equal the first one bit for bit wherever it sits in the grid, (ii) the
    first block must match the oracle, (iii) re-running from the same state after `stop()` reproduces the result bit
    for bit (the reference's own pin: gym_jiminy unit_py/test_pipeline_control.py:315-330)."""
    model = load_builtin(name)
    steps = 10
    B = blk * reps
    st = sample_states(model, blk, seed=23, **({"base_height": (0.9, 1.1)} if name == "atlas" else {}))
    q = torch.from_numpy(np.tile(st["q"], (1, reps)))
    v = torch.from_numpy(np.tile(st["v"], (1, reps)))
    cmd = torch.from_numpy(np.tile(st["command"], (1, reps)))
    ref = alloc_soa(model, blk)
    for k in ("q", "v", "command"):
        ref[k][:] = st[k]
    oracle_batch(model, ref, "start")
    loop = ReferenceFixedStepLoop(dt)   # the reference's sub-step rule: opens with a 1 us step (engine.cc:1176)
    for i in range(steps):
        oracle_engine_step(model, ref, loop, dt, "runge_kutta_4", command_changed=False)
    eng = _engine(model, B, torch.float64, "runge_kutta_4", dt)
    runs = []
    for rep in range(2):
        eng.set_command(cmd)
        eng.start(q, v)
        for i in range(steps):
            eng.step(dt)
        runs.append({k: eng.field(k).clone() for k in OUTS})
        eng.stop()
    ok = (ref["status"][0] & 1) == 0
    for k in OUTS:
        x = runs[0][k]
        first = x[:, :blk]
        assert torch.equal(x.view(x.shape[0], reps, blk), first[:, None, :].expand(-1, reps, -1)) or \
            bool(((x.view(x.shape[0], reps, blk) == first[:, None, :]) | torch.isnan(first[:, None, :])).all()), k
        assert rel_err(first.cpu().numpy(), ref[k], ok) < 1e-9, k
        same = (runs[0][k] == runs[1][k]) | (torch.isnan(runs[0][k]) & torch.isnan(runs[1][k]))
        assert bool(same.all()), k


def test_simulate_runs_to_t_end_or_until_the_callback_stops_it(gpu_device):
    """≙ `Engine::simulate` (engine.cc:1614-1699): end time, abort callback, `iterMax`, 5 ms floor."""
    model = load_builtin("double_pendulum")
    B, dt = 8, 1e-3
    st = sample_states(model, B, seed=2)
    eng = _engine(model, B, torch.float64, "runge_kutta_4", dt)
    eng.set_command(torch.from_numpy(st["command"]))
    q0, v0 = torch.from_numpy(st["q"]), torch.from_numpy(st["v"])
    with pytest.raises(ValueError, match="shorter than 5ms"):
        eng.simulate(1e-3, q0, v0)
    eng.simulate(0.05, q0, v0)
    assert not eng.is_simulation_running and abs(eng.stepper_state.t - 0.05) < 1e-12 and eng.stepper_state.iter == 51   # (50 periods + the opening 1 us step)
    ref = alloc_soa(model, B)
    for k in ("q", "v", "command"):
        ref[k][:] = st[k]
    oracle_batch(model, ref, "start")
    loop = ReferenceFixedStepLoop(dt)   # the reference's sub-step rule: opens with a 1 us step (engine.cc:1176)
    for i in range(50):
        oracle_engine_step(model, ref, loop, dt, "runge_kutta_4", command_changed=False)
    assert rel_err(eng.field("q").cpu().numpy(), ref["q"], np.ones(B, bool)) < 1e-10
    calls = []
    eng.simulate(0.05, q0, v0, callback=lambda: len(calls) < 7 and not calls.append(0))
    assert eng.stepper_state.iter == 8
    eng.set_options({"stepper": {"iterMax": 12}})
    eng.simulate(0.05, q0, v0)
    assert eng.stepper_state.iter == 12


def test_bench_line_runs_the_full_extra_terms(gpu_device):
    """The driver's command on a small batch: ONE JSON line whose launch ran the whole `computeExtraTerms` (the three
    optional outputs bound and written), with the roofline object priced on the algorithmic bytes of SURVEY.md 8d and the
    secondary workloads declared (ANYmal euler + constraint model = the reference's shipped options, Atlas both ways)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2", "--batch", "4096",
                          "--episode", "3", "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["metric"].startswith("env-steps/s") and rec["unit"] == "env-steps/s" and rec["dtype"] == "f64"
    assert rec["steps"] == 6 and rec["warmup"] == 2 and rec["n_gpus"] == 1 and rec["vs_baseline"] is None
    cfg = rec["config"]
    assert cfg["extra_terms"] == "full" and cfg["extra_terms_written"] is True and "computeExtraTerms" in cfg["workload"]
    rf = rec["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["launches_timed"] == 6
    assert rf["algorithmic_bytes_per_launch"] == 188 * 8 * 4096           # 188 scalars per ANYmal env-step
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.0 < rf["avg_launch_ms"] < rec["ms_per_step"] * 1.5
    sys.path.insert(0, root)
    import bench
    assert [(c["model_name"], c["contact_model"], c["solver"]) for c in bench.SECONDARY] == [
        ("anymal", "spring_damper", "runge_kutta_4"),     # (round 6: the headline robot at dt = 2.5e-4, every lane finite)
        ("anymal", "constraint", "euler_explicit"), ("atlas", "spring_damper", "runge_kutta_4"), ("atlas", "constraint", "euler_explicit"),
        ("arm7", "spring_damper", "runge_kutta_4")]       # (round 5: the one-robot-per-lane kernels' robot)
    assert bench.SECONDARY[0]["dt"] == 2.5e-4
    assert set(bench.FULL_EXTRA_OUTPUTS) >= {"energy", "joint_forces", "centroidal"}


@pytest.mark.gpu
def test_flexibility_with_armature_is_a_series_elastic_actuator_on_the_device(gpu_device):
    """unit_py/test_simple_pendulum.py:662-750 through the engine: the flexible pendulum with a rotor inertia and a PD law on
    its motor against the linear SEA system (exact discretisation of plant + held command), on every lane of a small batch
    with lane-dependent gains; 1 s of 1e-4 s steps."""
    from scipy.linalg import expm
    from tests import robots
    k, nu, J, I = 20.0, 0.1, 0.1, 5.0
    model = robots.pendulum_flexible(k, nu, J)
    B, dt, n = 8, 1e-4, 10000
    kc = torch.linspace(60.0, 130.0, B, dtype=torch.float64, device=gpu_device)
    nc = torch.linspace(0.5, 1.5, B, dtype=torch.float64, device=gpu_device)
    eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device)
    eng.set_options({"world": {"gravity": [0.0] * 6},
                     "stepper": {"odeSolver": "runge_kutta_4", "dtMax": dt, "controllerUpdatePeriod": dt, "sensorsUpdatePeriod": dt},
                     "contacts": {"model": "spring_damper"}})
    q0 = torch.zeros((5, B), dtype=torch.float64)
    q0[3] = 1.0
    v0 = torch.zeros((4, B), dtype=torch.float64)
    v0[1] = 0.1
    eng.set_command(torch.zeros((1, B), dtype=torch.float64))
    eng.start(q0, v0)
    Ap = np.array([[0, 0, 1, 0], [0, 0, 0, 1], [-k * (1 / I + 1 / J), 0, -nu * (1 / I + 1 / J), 0], [k / J, 0, nu / J, 0]])
    Bp = np.array([0, 0, -1 / J, 1 / J])
    aug = np.zeros((5, 5))
    aug[:4, :4], aug[:4, 4] = Ap, Bp
    Ed = expm(aug * dt)
    x = np.tile(np.array([0.0, 0.0, 0.1, 0.0])[:, None], (1, B))
    kc_h, nc_h = kc.cpu().numpy(), nc.cpu().numpy()
    q, v = eng.field("q"), eng.field("v")
    err = 0.0
    for i in range(n):
        u = -kc * q[4] - nc * v[3]
        eng.set_command(u[None, :])
        eng.step(dt)
        x = Ed[:4, :4] @ x + np.outer(Ed[:4, 4], -kc_h * x[1] - nc_h * x[3])
        if i % 500 == 499 or i == n - 1:
            qh, vh = q.cpu().numpy(), v.cpu().numpy()
            err = max(err, np.abs(2 * np.arctan2(qh[1], qh[3]) - x[0]).max(), np.abs(qh[4] - x[1]).max(),
                      np.abs(vh[1] - x[2]).max(), np.abs(vh[3] - x[3]).max())
            assert np.abs(qh[[0, 2]]).max() == 0.0 and np.abs(vh[[0, 2]]).max() == 0.0
    assert err < 1e-4, err
    assert int(eng.status.abs().sum()) == 0


@pytest.mark.gpu
def test_backlash_two_phases_on_the_device(gpu_device):
    """unit_py/test_simple_pendulum.py:269-332 through the engine (constraint contact model, whose bound rows are the backlash):
    inside the backlash the rotor and the pendulum move independently, 0.4 s after the impact they move as one body of inertia
    m l^2 + J -- both phases against an independent integration to the reference's tolerance, on lanes with different motor
    torques (so that they reach the limit at different times)."""
    from scipy.integrate import solve_ivp
    from tests import robots
    G, J, BACK = 9.81, 1.0, 1.1
    model = robots.pendulum_backlash(2 * BACK, J)
    B, dt = 6, 1e-4
    taus = np.linspace(3.0, 8.0, B)
    eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device)
    eng.set_options({"stepper": {"odeSolver": "runge_kutta_4", "dtMax": dt, "controllerUpdatePeriod": dt, "sensorsUpdatePeriod": dt},
                     "contacts": {"model": "constraint"}, "constraints": {"regularization": 0.0}})
    x0 = [0.0, 0.1, 0.0, 0.0]
    q0 = torch.tensor(x0[:2], dtype=torch.float64)[:, None].repeat(1, B)
    eng.set_command(torch.from_numpy(-taus)[None, :])
    eng.start(q0, torch.zeros((2, B), dtype=torch.float64))
    t_imp = []
    for tau in taus:
        free = lambda t, x, tau=tau: [x[2], x[3], -tau / J, G * np.sin(x[0] + x[1]) + tau / J]   # noqa: E731
        hit = lambda t, x: x[1] - BACK                                                            # noqa: E731
        hit.terminal = True
        t_imp.append(solve_ivp(free, (0, 5), x0, events=hit, method="DOP853", rtol=1e-12, atol=1e-12).t_events[0][0])
    n = int(round((max(t_imp) + 0.9) / dt))
    every = 20
    X, T = [], []
    q, v = eng.field("q"), eng.field("v")
    for i in range(n):
        eng.step(dt)
        if (i + 1) % every == 0:
            X.append(torch.cat([q, v]).cpu().numpy().copy())
            T.append((i + 1) * dt)
    X, T = np.array(X), np.array(T)          # [time][4][lane]
    for lane, tau in enumerate(taus):
        t1, t2 = np.searchsorted(T, [t_imp[lane] - 0.02, t_imp[lane] + 0.4])
        free = lambda t, x: [x[2], x[3], -tau / J, G * np.sin(x[0] + x[1]) + tau / J]            # noqa: E731
        sol = solve_ivp(free, (0, T[t1 - 1]), x0, t_eval=T[:t1], method="DOP853", rtol=1e-12, atol=1e-12)
        assert np.abs(sol.y.T - X[:t1, :, lane]).max() < 1e-7, lane
        I_total = 5.0 + J
        joined = lambda t, x: [x[2], x[3], 5.0 * G / I_total * np.sin(x[0] + x[1]) - tau / I_total, 0.0]   # noqa: E731
        sol = solve_ivp(joined, (0, T[-1] - T[t2]), X[t2, :, lane], t_eval=T[t2:] - T[t2], method="DOP853", rtol=1e-12, atol=1e-12)
        assert np.abs(sol.y.T - X[t2:, :, lane]).max() < 1e-7, lane
    assert int(eng.status.abs().sum()) == 0
