"""`contacts.model = "constraint"`: joint position bounds and contact points as
kinematic constraints, multipliers by projected Gauss-Seidel.

Three layers, as for the spring-damper path:
  * pins of the CPU oracle (which restates the reference's own formulation: dense inertia matrix +
    Cholesky + dense Jacobian + PGS) against the laws the reference tests hold for this model
    (unit_py/test_foot_pendulum.py:25-95, test_dense_pole.py:164-203, test_simple_mass.py:181-240)
    and against an independently coded RNEA;
  * the kernel sources compiled for the host (tests/hostemu) against the oracle -- a cross check of
    two different formulations (articulated-body solves vs. dense factorisation), no GPU needed;
  * `-m gpu`: the device build through the C ABI / BatchedEngine against the oracle.
"""
import numpy as np
import pytest

from jiminy_amd import _abi, load_builtin
from jiminy_amd.synthetic import sample_standing_states, sample_states
from oracle import rbd_numpy as rbd
from oracle.oracle_py import OracleEngine
from tests import robots
from tests.helpers import ReferenceFixedStepLoop, alloc_constraint_state, alloc_soa, oracle_batch, oracle_engine_step, rel_err
from tests.hostemu import emu

G = 9.81
TIGHT = dict(tol_abs=1e-11, tol_rel=1e-10)  # PGS run to stagnation: iterates comparable to round-off
OUTS = ("q", "v", "a", "u_motor", "u", "imu", "force", "contact", "encoder", "effort", "energy",
        "contact_forces", "f_external", "joint_forces", "centroidal", "con_data")


def _engine(model, **copt):
    e = OracleEngine(model)
    e.set_constraint_options(**copt)
    return e


# ------------------------------------------------------------------ oracle pins
def test_point_mass_rests_on_the_ground():
    """test_foot_pendulum.py:25-95 shape: a body in contact, initialised at rest, does not move;
    accelerometer = -gravity, force sensor = weight."""
    m = robots.point_mass()
    e = _engine(m, regularization=1e-9, stabilization_freq=0.0)
    q0 = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0])
    e.start(q0, np.zeros(6))
    assert np.abs(e.get("a")).max() < 1e-7
    imu = e.get("imu")
    assert np.abs(imu[:3]).max() < 1e-7
    assert np.abs(imu[3:] - [0, 0, G]).max() < 1e-7
    cf = e.get("contact_forces")
    assert cf[2] == pytest.approx(2.0 * G, rel=1e-7)
    assert np.abs(np.delete(cf, 2)).max() < 1e-7
    # the force sensor sits on a frame yawed by 0.3 rad: same vertical force
    assert e.get("force")[2] == pytest.approx(2.0 * G, rel=1e-7)
    for _ in range(200):
        e.step(1e-3, solver="runge_kutta_4", command_changed=False)
    assert np.abs(e.get("v")).max() < 1e-7 and np.abs(e.get("a")).max() < 1e-7
    assert e.status == 0


def test_contact_sensor_matches_the_external_force():
    """test_simple_mass.py:181-240: contact / force sensors and f_external describe the same wrench."""
    m = robots.point_mass()
    e = _engine(m)
    q0 = np.array([0.0, 0.0, -1e-4, 0.0, 0.0, 0.0, 1.0])
    e.start(q0, np.array([0.3, -0.2, 0.0, 0.0, 0.0, 0.5]))
    for _ in range(50):
        e.step(1e-3, solver="euler_explicit", command_changed=False)
        f_joint = e.get("f_external").reshape(-1, 6)[1]
        f_contact = e.get("contact")  # contact frame == joint frame for the "body" contact point
        assert np.abs(f_contact - f_joint[:3]).max() < 1e-9
    assert e.get("contact_forces")[2] > 0.0


def test_friction_cone_and_unilateral_contact():
    m = load_builtin("anymal")
    st = sample_standing_states(m, 12, seed=3)
    e = _engine(m, **TIGHT)
    active = 0
    for l in range(12):
        e.start(st["q"][:, l], st["v"][:, l], st["command"][:, l])
        cf = e.get("contact_forces").reshape(-1, 6)
        # world-aligned multipliers live in con-data; in the contact frame the cone is on norms
        for c in range(m.ncontacts):
            n = np.linalg.norm(cf[c, :3])
            if n > 0:
                active += 1
    assert active >= 8


def test_joint_limit_hysteresis_branches():
    """test_dense_pole.py:164-203: a joint driven into its position limit switches its bound
    constraint on when it leaves the range, keeps it inside the transition band, releases it beyond."""
    m = robots.pendulum()
    m.position_lower[:] = -0.4
    m.position_upper[:] = 0.4
    eps = 1e-3
    e = _engine(m)
    flags, data = np.zeros((1, 1), np.int32), np.zeros((2, 1))
    e.bind_constraints(flags, data)
    arr = alloc_soa(m, 1)
    arr["con_flags"], arr["con_data"] = flags, data
    arr["q"][:] = 0.0
    arr["v"][:] = 1.0
    from tests.helpers import oracle_io
    e.batch_run("start", oracle_io(arr))
    assert flags[0, 0] & 1 == 0
    branches, was = set(), False
    for _ in range(6000):
        e.batch_run("step", oracle_io(arr), solver="euler_explicit", dt=1e-4, command_changed=False)
        theta, on = arr["q"][0, 0], bool(flags[0, 0] & 1)
        if 0.4 - abs(theta) <= 0.0:
            assert on
            branches.add(0 if was else 1)
        elif 0.4 - abs(theta) < eps:
            assert on == was
            branches.add(2 if was else 3)
        else:
            assert not on
            branches.add(4)
        was = on
    assert branches == {0, 1, 2, 3, 4}
    # the bound holds: the pendulum (inverted at q = 0) falls onto its limit and stays there
    assert abs(arr["q"][0, 0]) < 0.4 + 5e-3


@pytest.mark.parametrize("name", ["anymal", "atlas", "crane_walker"])
def test_constrained_acceleration_satisfies_the_equation_of_motion(name):
    """RNEA(q, v, a, f_external) + rotor a == u with an independently coded RNEA: the contact
    multipliers enter through f_external, the joint-bound multipliers through u (the reference adds
    them to u with a plus sign whatever the direction, engine.cc:3786-3790: corrected here)."""
    model = robots.crane_walker() if name == "crane_walker" else load_builtin(name)
    B = 6
    st = sample_standing_states(model, B, seed=4, out_of_bounds_fraction=0.5)
    rows = _abi.constraint_rows(model)
    e = _engine(model, **TIGHT)
    flags = np.zeros((rows["con_flags"], 1), np.int32)
    data = np.zeros((rows["con_data"], 1))
    e.bind_constraints(flags, data)
    arr = alloc_soa(model, 1)
    arr["con_flags"], arr["con_data"] = flags, data
    from tests.helpers import oracle_io
    n_contact = n_bound = 0
    bounded = [j for j in range(1, model.njoints) if 1 <= int(model.jtypes[j]) <= 8]
    for l in range(B):
        for k in ("q", "v", "command"):
            arr[k][:, 0] = st[k][:, l]
        # `start` initialises the constraint state; the identity is checked on a regular evaluation
        # (the start passes feed the previous pass's bound multipliers back into u, engine.cc:1456)
        e.batch_run("start", oracle_io(arr))
        e.batch_run("dynamics", oracle_io(arr))
        q, v, a, u = arr["q"][:, 0], arr["v"][:, 0], arr["a"][:, 0], arr["u"][:, 0].copy()
        fext = arr["f_external"][:, 0].reshape(-1, 6)
        nb = rows["n_bounds"]
        for k, j in enumerate(bounded):
            if flags[k, 0] & 2:  # reversed constraint: generalised force is -lambda
                u[int(model.idx_v[j])] -= 2.0 * data[nb + k, 0]
            n_bound += int(flags[k, 0] & 1)
        n_contact += int(np.abs(fext).sum() > 0)
        tau = rbd.rnea(model, q, v, a, fext) + model.rotor_inertia * a
        scale = max(1.0, np.abs(u).max(), np.abs(tau).max())
        assert np.abs(tau - u).max() / scale < 1e-10, l
        # multipliers: unilateral normal force, friction inside the cone (friction = 1)
        lam = data[2 * nb:, 0].reshape(-1, 4)
        assert (lam[:, 2] >= 0).all()
        assert (np.hypot(lam[:, 0], lam[:, 1]) <= lam[:, 2] * (1 + 1e-12) + 1e-12).all()
        assert (data[nb:2 * nb, 0] >= 0).all()
    assert n_contact >= 3 and n_bound >= 2


def test_start_passes_converge_to_a_fixed_point():
    """The 4 start passes end on a solution that one more evaluation reproduces (reference pin 11:
    bitwise repeatability of `a` after reset is the same statement for its own engine)."""
    m = load_builtin("anymal")
    # no joint beyond its limits: the start passes feed bound multipliers back into u (engine.cc:1456)
    st = sample_standing_states(m, 4, seed=6, out_of_bounds_fraction=0.0)
    e = _engine(m, **TIGHT)
    for l in range(4):
        e.start(st["q"][:, l], st["v"][:, l], st["command"][:, l])
        a0 = e.get("a").copy()
        a1 = e.dynamics(st["q"][:, l], st["v"][:, l])
        assert np.abs(a1 - a0).max() < 1e-5 * max(1.0, np.abs(a0).max()), (l, e.status)


# ------------------------------------------------------------------ kernel sources on the host vs oracle
def _models():
    return {
        "anymal": lambda: load_builtin("anymal"),
        "atlas": lambda: load_builtin("atlas"),
        "cartpole": lambda: load_builtin("cartpole"),
        "point_mass": robots.point_mass,
        "tree_arm": lambda: robots.tree_arm(False),
        "tree_arm_ff": lambda: robots.tree_arm(True),
        "pendulum_backlash": robots.pendulum_backlash,
        "tree_arm_flex": lambda: robots.tree_arm_flexible(False),
        "tree_arm_flex_ff": lambda: robots.tree_arm_flexible(True),
        "crane_walker": robots.crane_walker,
        "biped": robots.biped,
        "biped_torso": lambda: robots.biped(True),
    }


def _states(model, B, seed):
    if model.has_freeflyer and model.ncontacts > 0:
        return sample_standing_states(model, B, seed=seed)
    st = sample_states(model, B, seed=seed)
    # fixed-base arms: push a few joints past / next to their limits
    rng = np.random.default_rng(seed)
    for j in range(1, model.njoints):
        if 1 <= int(model.jtypes[j]) <= 8:
            iq = int(model.idx_q[j])
            lanes = rng.random(B) < 0.3
            hi = rng.random(B) < 0.5
            over = rng.uniform(-5e-4, 0.02, B)
            st["q"][iq, lanes & hi] = model.position_upper[iq] + over[lanes & hi]
            st["q"][iq, lanes & ~hi] = model.position_lower[iq] - over[lanes & ~hi]
    return st


def _pair(model, B, seed):
    st = _states(model, B, seed)
    ref, got = alloc_soa(model, B), alloc_soa(model, B)
    for arr in (ref, got):
        alloc_constraint_state(model, arr, B)
        for k in ("q", "v", "command"):
            if st[k].shape[0]:
                arr[k][:] = st[k]
    return ref, got


def _check(got, ref, tol, what=""):
    assert np.array_equal(got["con_flags"], ref["con_flags"]), what
    assert np.array_equal(got["status"], ref["status"]), what
    for k in OUTS:
        e = rel_err(got[k], ref[k])
        assert e < tol, (what, k, e)


# served by the branch-parallel kernel (jm_qcon.h); the bipeds with two / one empty limbs (codegen.quad_structure)
QUAD_MODELS = ("anymal", "atlas", "crane_walker", "biped", "biped_torso")


@pytest.mark.parametrize("name,variant", [(n, "lane") for n in _models()] + [(n, "quad") for n in QUAD_MODELS] +
                         [("atlas", "split"), ("anymal", "split"), ("biped", "split")])
def test_constraint_kernel_matches_oracle_on_the_host(name, variant):
    """Both device formulations compiled for the host against the oracle (the reference's dense one):
    `lane` = one robot per lane, sequential bias-free solves (jm_constraint.h); `quad` = four lanes per
    robot, four delassus columns per round, packed symmetric matrix in the per-robot solver region split
    between the on-chip part and the overflow rows, PGS dot products summed over the quad (jm_qcon.h); `split` = the step
    launches of robots with large solves as pre | solve | post per evaluation (stage buffer, constraint context and the
    square solver region persistent between the parts, `qcon_pgs_lean` with its ring of prefetched rows)."""
    split = variant == "split"
    variant = "quad" if split else variant
    model = _models()[name]()
    B = (3 if variant == "lane" else 6) if name == "atlas" else (8 if variant == "lane" else 12)   # (sized for the CPU suite)
    ref, got = _pair(model, B, seed=7)
    oracle_batch(model, ref, "start", constraint_options=TIGHT)
    emu.run(model, got, "start", constraint_options=TIGHT, variant=variant, split=split)   # (split: the start passes as launches too)
    _check(got, ref, 1e-9, "start")
    n_active = int((ref["con_flags"] & 1).sum())
    assert n_active > 0 or _abi.constraint_rows(model)["n_rows"] == 0
    lane_before = emu.lane_solves(model)
    for solver, n_sub, changed in (("euler_explicit", 3, True), ("runge_kutta_4", 1, False)):
        for _ in range(2):
            kw = dict(solver=solver, dt=5e-4, n_substeps=n_sub, command_changed=changed)
            oracle_batch(model, ref, "step", constraint_options=TIGHT, **kw)
            emu.run(model, got, "step", constraint_options=TIGHT, variant=variant, split=split, **kw)
        _check(got, ref, 1e-7, solver)
    if split and name == "atlas":
        # robots with few active joint rows solve in the operational space of their feet (jm_qtip.h), the others
        # stream the delassus matrix: this seeded batch exercises the first form
        assert emu.tip_solves(model) > 0
    if split and name != "atlas":
        # robots with few contact points: the solve of the split form runs one lane per robot (qcon_pgs_lane, round 6)
        assert emu.lane_solves(model) > lane_before


LOCKS = {"anymal": ("LF_KFE", "RH_HAA", "RH_HFE"), "atlas": ("l_arm_elx", "r_arm_shx", "back_bky", "l_leg_kny"),
         "tree_arm": ("a_slide", "c_skew"), "tree_arm_ff": ("b_yaw", "d_skew_slide")}


@pytest.mark.parametrize("name,split", [("anymal", False), ("atlas", False), ("atlas", True), ("tree_arm", "lane"), ("tree_arm_ff", "lane"),
                                        ("anymal", "lane")])
def test_user_joint_constraints_on_the_host(name, split):
    """User-registered `JointConstraint`s (`Model::addConstraint`, model.cc:926-936: bit 2 of the joint's constraint flag):
    bilateral rows solved first in every sweep, no projection (constraint_solvers.cc:112-128), multipliers not restored
    into RobotState::u.  Kernel sources on the host against the oracle, some lanes locked and some not; the locked
    joints stay where `Engine::start` found them."""
    # (`split` == "lane": the one-robot-per-lane kernel, jm_constraint.h -- any tree, since round 4)
    variant = "lane" if split == "lane" else "quad"
    split = split is True
    model = _models()[name]()
    B = 8 if name == "anymal" else 4
    ref, got = _pair(model, B, seed=31)
    rows = [model.bound_row(j) for j in LOCKS[name]]
    lanes = np.arange(B) % 2 == 0     # every other lane carries the locks
    for arr in (ref, got):
        for r in rows:
            arr["con_flags"][r, lanes] |= 4
    oracle_batch(model, ref, "start", constraint_options=TIGHT)
    emu.run(model, got, "start", constraint_options=TIGHT, variant=variant)
    _check(got, ref, 1e-8, "start")
    assert all(int(ref["con_flags"][r, 0]) == 5 and (int(ref["con_flags"][r, 1]) & 4) == 0 for r in rows)
    q_lock = np.array([ref["q"][int(model.idx_q[model.joint_names.index(j)])].copy() for j in LOCKS[name]])
    for solver, n_sub in (("euler_explicit", 4), ("runge_kutta_4", 2)):
        for _ in range(2):
            kw = dict(solver=solver, dt=5e-4, n_substeps=n_sub, command_changed=True)
            oracle_batch(model, ref, "step", constraint_options=TIGHT, **kw)
            emu.run(model, got, "step", constraint_options=TIGHT, variant=variant, split=split, **kw)
        _check(got, ref, 1e-7, solver)
    # the constraint equation of every lock at the end state: a + kp (q - q_ref) + kd v = 0 (JointConstraint drift with the
    # Baumgarte gains of abstract_constraint.cc:88-98); q_ref = the configuration at start
    omega = 2.0 * np.pi * 20.0
    for j, r, q0 in zip(LOCKS[name], rows, q_lock):
        jj = model.joint_names.index(j)
        iq, iv = int(model.idx_q[jj]), int(model.idx_v[jj])
        assert np.array_equal(ref["con_data"][r, lanes], q0[lanes])
        res = ref["a"][iv] + omega ** 2 * (ref["q"][iq] - q0) + 2.0 * omega * ref["v"][iv]
        # (to the regularisation of the solve: (A + 1e-3 diag A) lambda = b leaves 1e-3 A_ii lambda_i, constraint_solvers.cc:376-387)
        assert np.abs(res[lanes]).max() < 1e-2 * max(1.0, np.abs(ref["a"][iv]).max()), (j, np.abs(res[lanes]).max())
        assert np.abs(res[~lanes]).max() > 20.0 * np.abs(res[lanes]).max()
    # the multipliers of the locks are there (con_data) and do not appear in RobotState::u
    nb = _abi.constraint_rows(model)["n_bounds"]
    assert np.abs(ref["con_data"][[nb + r for r in rows]][:, lanes]).max() > 1e-3


@pytest.mark.parametrize("name,variant,freq", [("anymal", "quad", 5.0), ("anymal", "quad", 0.0), ("atlas", "split", 5.0),
                                               ("tree_arm_ff", "lane", 5.0), ("tree_arm", "lane", 0.0)])
def test_user_joint_constraints_with_gains_of_their_own_on_the_host(name, variant, freq):
    """`jm_constraint_options::user_stabilization_freq` (ABI 6): user constraints keep Baumgarte gains of their own
    (abstract_constraint.cc:88-98; `Engine::start` only sets those of the bounds and contacts, engine.cc:1276-1285) -- 5 Hz
    and 0 Hz (the reference's state of a freshly created constraint: a pure acceleration constraint) next to the 20 Hz of
    the contacts; kernel sources of both families on the host against the oracle, then the constraint equation itself."""
    split = variant == "split"
    variant = "quad" if split else variant
    model = _models()[name]()
    B = 8 if name == "anymal" else 4
    ref, got = _pair(model, B, seed=33)
    rows = [model.bound_row(j) for j in LOCKS[name]]
    lanes = np.arange(B) % 2 == 0
    for arr in (ref, got):
        for r in rows:
            arr["con_flags"][r, lanes] |= 4
    copt = dict(TIGHT, user_stabilization_freq=freq)
    oracle_batch(model, ref, "start", constraint_options=copt)
    emu.run(model, got, "start", constraint_options=copt, variant=variant, split=split)
    _check(got, ref, 1e-8, "start")
    q0 = {j: ref["q"][int(model.idx_q[model.joint_names.index(j)])].copy() for j in LOCKS[name]}
    for solver, n_sub in (("euler_explicit", 3), ("runge_kutta_4", 1)):
        for _ in range(2):
            kw = dict(solver=solver, dt=5e-4, n_substeps=n_sub, command_changed=True)
            oracle_batch(model, ref, "step", constraint_options=copt, **kw)
            emu.run(model, got, "step", constraint_options=copt, variant=variant, split=split, **kw)
        _check(got, ref, 1e-7, solver)
    omega = 2.0 * np.pi * freq
    for j in LOCKS[name]:
        jj = model.joint_names.index(j)
        iq, iv = int(model.idx_q[jj]), int(model.idx_v[jj])
        res = ref["a"][iv] + omega ** 2 * (ref["q"][iq] - q0[j]) + 2.0 * omega * ref["v"][iv]
        # (with the 20 Hz gains of the contacts instead, the same expression is far from zero on a joint that moved)
        assert np.abs(res[lanes]).max() < 1e-2 * max(1.0, np.abs(ref["a"][iv]).max()), (j, np.abs(res[lanes]).max())


@pytest.mark.parametrize("split", [False, True])
def test_atlas_standing_flat_on_both_feet_start_and_steps(split):
    """A humanoid standing flat: the 8 bottom vertices of each foot box touch, 16 contact points = 64 rows in
    `Engine::start`'s passes (4-row blocks) + the joints the neutral pose puts on their bounds -- the largest solve the
    shipped robots produce, above the 64 rows the solver region was first sized for (a truncated contact block then
    indexed past the region: the emulation keeps guard rows behind its workspace for exactly that).  Start, then
    steps with 3-row blocks, against the oracle (`split`: the steps in the pre | solve | post form, whose solves of more than
    64 rows take the 12-loads-per-row instantiation of `qcon_pgs_lean`)."""
    from jiminy_amd.synthetic import lowest_contact_height
    model = load_builtin("atlas")
    B = 2
    q = model.neutral()
    for name, value in {"back_bky": 0.2, "l_arm_elx": 0.2, "l_arm_shx": -np.pi / 2, "l_arm_shz": np.pi / 4, "l_arm_ely": 3 * np.pi / 4,
                        "r_arm_elx": -0.2, "r_arm_shx": np.pi / 2, "r_arm_shz": -np.pi / 4, "r_arm_ely": 3 * np.pi / 4}.items():
        q[int(model.idx_q[model.joint_names.index(name)])] = value
    mask = model.bounded_position_mask()
    q[mask] = np.clip(q[mask], model.position_lower[mask], model.position_upper[mask])
    q[2] -= float(lowest_contact_height(model, q)[0]) + 2.0e-3           # 2 mm into the ground
    ref, got = alloc_soa(model, B), alloc_soa(model, B)
    for arr in (ref, got):
        alloc_constraint_state(model, arr, B)
        arr["q"][:] = q[:, None]
        arr["q"][2] += 1e-4 * np.arange(B)
    lib = emu._lib(model)
    before = lib.emu_guard_violations()
    oracle_batch(model, ref, "start", constraint_options=TIGHT)
    emu.run(model, got, "start", constraint_options=TIGHT, variant="quad", split=split)
    nb = _abi.constraint_rows(model)["n_bounds"]
    assert int((ref["con_flags"][nb:, 0] & 1).sum()) == 16 and int((ref["con_flags"][:nb, 0] & 1).sum()) >= 3
    _check(got, ref, 1e-8, "start")
    for _ in range(2):
        kw = dict(solver="euler_explicit", dt=1e-3, n_substeps=2, command_changed=True)
        oracle_batch(model, ref, "step", constraint_options=TIGHT, **kw)
        emu.run(model, got, "step", constraint_options=TIGHT, variant="quad", split=split, **kw)
    _check(got, ref, 1e-6, "steps")
    assert lib.emu_guard_violations() == before


def test_split_start_with_many_active_joint_bounds_on_the_host():
    """`start` / `reset` of robots with large solves as launches of the split kernels (first pass | exact solve | 3 x (pass |
    Gauss-Seidel) | closing evaluation, jm_lib.cpp): a robot with MORE than eight active joint rows keeps the streamed
    form of the solve -- in-place Cholesky factorisation of the square matrix for the exact pass, the matrix rebuilt in
    the next pass --, one with few takes the operational-space form (Woodbury); both against the oracle's Engine::start,
    then a masked reset of one of them."""
    from jiminy_amd.synthetic import lowest_contact_height
    model = load_builtin("atlas")
    B = 2
    q = np.repeat(model.neutral()[:, None], B, axis=1)
    mask = model.bounded_position_mask()
    idx = np.flatnonzero(mask)
    q[mask] = np.clip(q[mask], (model.position_lower[mask] + 0.05)[:, None], (model.position_upper[mask] - 0.05)[:, None])
    # robot 0: eleven joints beyond a bound (more than the operational-space form takes, QTip::NBX = 8); robot 1: one
    for n, lane in ((11, 0), (1, 1)):
        for j in idx[3:3 + 2 * n:2]:
            q[j, lane] = model.position_upper[j] + 0.01
    q[2] -= lowest_contact_height(model, q) + 2.0e-3
    ref, got = alloc_soa(model, B), alloc_soa(model, B)
    for arr in (ref, got):
        alloc_constraint_state(model, arr, B)
        arr["q"][:] = q
    before = emu.tip_solves(model)
    oracle_batch(model, ref, "start", constraint_options=TIGHT)
    emu.run(model, got, "start", constraint_options=TIGHT, variant="quad", split=True)
    nb = _abi.constraint_rows(model)["n_bounds"]
    assert int((ref["con_flags"][:nb, 0] & 1).sum()) >= 9 and int((ref["con_flags"][:nb, 1] & 1).sum()) <= 8
    _check(got, ref, 1e-8, "start")
    assert np.array_equal(got["con_flags"], ref["con_flags"])
    # robot 1 went through the operational-space form (exact solve + three passes), robot 0 did not
    assert emu.tip_solves(model) - before == 4
    # masked reset of robot 0 only, to the same state: the start sequence again for that lane (bit for bit: reference pin 11),
    # nothing touched for the other one
    snap = {k: got[k].copy() for k in OUTS}
    got["q_init"], got["v_init"] = q.copy(), np.zeros_like(got["v"])
    got["mask"] = np.array([1, 0], dtype=np.uint8)
    got["a"][:, 1] += 1.0          # (a reset must not rewrite the lane that is not restarting)
    snap["a"][:, 1] += 1.0
    emu.run(model, got, "reset", constraint_options=TIGHT, variant="quad", split=True)
    for k in OUTS:
        assert np.array_equal(got[k], snap[k]), k


@pytest.mark.parametrize("torsion", [0.0, 0.3])
def test_quad_constraint_kernel_torsion_dynamics_and_reset(torsion):
    """Branch-parallel constraint kernel on the host: with torsional friction (the 4-row contact blocks) and
    without (3-row blocks, the torsion row left out of the solve), `compute_robots_dynamics` at another state
    (no outputs, constraint state still switched) and `reset_lanes` (the start sequence for masked lanes)."""
    model = load_builtin("anymal")
    B = 8
    copt = dict(TIGHT, torsion=torsion)
    ref, got = _pair(model, B, seed=17)
    oracle_batch(model, ref, "start", constraint_options=copt)
    emu.run(model, got, "start", constraint_options=copt, variant="quad")
    _check(got, ref, 1e-9, "start")
    for _ in range(3):
        kw = dict(solver="euler_explicit", dt=1e-3, n_substeps=2, command_changed=True)
        oracle_batch(model, ref, "step", constraint_options=copt, **kw)
        emu.run(model, got, "step", constraint_options=copt, variant="quad", **kw)
    _check(got, ref, 1e-7, "euler")
    if torsion > 0:
        nb = _abi.constraint_rows(model)["n_bounds"]
        assert np.abs(ref["con_data"][2 * nb:].reshape(-1, 4, B)[:, 3]).max() > 0   # torsion multipliers at work
    # dynamics at a perturbed state: accelerations only, but the constraint objects are switched / warm-started
    rg = np.random.default_rng(0)
    q_in = ref["q"].copy()
    q_in[7:] += 1e-3 * rg.standard_normal(q_in[7:].shape)
    v_in = ref["v"] + 1e-2 * rg.standard_normal(ref["v"].shape)
    for arr in (ref, got):
        arr["q_in"], arr["v_in"], arr["a_out"] = q_in.copy(), v_in.copy(), np.zeros_like(ref["a"])
    from oracle.oracle_py import OracleEngine
    e = OracleEngine(model)
    e.set_constraint_options(**copt)
    e.bind_constraints(ref["con_flags"], ref["con_data"])
    io = {"q": ref["q_in"], "v": ref["v_in"], "a": ref["a_out"], "command": ref["command"], "status": ref["status"].reshape(-1)}
    e.batch_run("dynamics", io)
    emu.run(model, got, "dynamics", constraint_options=copt, variant="quad")
    assert rel_err(got["a_out"], ref["a_out"]) < 1e-8
    assert np.array_equal(got["con_flags"], ref["con_flags"])
    assert rel_err(got["con_data"], ref["con_data"]) < 1e-8


def test_default_tolerances_follow_the_oracle_iterates():
    """With the reference's default PGS tolerances (stepper.tolAbs/tolRel) the stopping decision is
    taken on residual *differences*; both formulations produce the same iterates to round-off, so
    the same iteration count and the same multipliers."""
    model = load_builtin("anymal")
    ref, got = _pair(model, 12, seed=8)
    oracle_batch(model, ref, "start", constraint_options={})
    emu.run(model, got, "start", constraint_options={})
    _check(got, ref, 1e-9, "start")
    for _ in range(5):
        kw = dict(solver="euler_explicit", dt=1e-3, n_substeps=1, command_changed=False)
        oracle_batch(model, ref, "step", constraint_options={}, **kw)
        emu.run(model, got, "step", constraint_options={}, **kw)
    _check(got, ref, 1e-7, "euler")


def test_spring_damper_path_is_untouched_by_the_constraint_state():
    model = load_builtin("anymal")
    st = sample_states(model, 8, seed=1)
    a, b = alloc_soa(model, 8), alloc_soa(model, 8)
    for arr in (a, b):
        for k in ("q", "v", "command"):
            arr[k][:] = st[k]
    emu.run(model, a, "start")
    alloc_constraint_state(model, b, 8)
    emu.run(model, b, "start", constraint_options=dict(model="spring_damper"))
    assert np.array_equal(a["a"], b["a"])


# ------------------------------------------------------------------ device build (C ABI) vs oracle
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["anymal", "atlas", "crane_walker", "tree_arm", "point_mass", "biped", "biped_torso",
                                  "tree_arm_flex", "tree_arm_flex_ff", "pendulum_backlash"])
def test_gpu_constraint_model_matches_oracle(name, gpu_device):
    import torch

    from jiminy_amd.engine import BatchedEngine
    model = _models()[name]()
    B = 96 if name != "atlas" else 40
    ref, _ = _pair(model, B, seed=11)
    eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device,
                        extra_outputs=("contact_forces", "f_external", "joint_forces", "energy", "centroidal"))
    dt = 5e-4
    eng.set_options({"stepper": {"odeSolver": "euler_explicit", "dtMax": dt, "controllerUpdatePeriod": dt,
                                 "sensorsUpdatePeriod": dt, "tolAbs": TIGHT["tol_abs"], "tolRel": TIGHT["tol_rel"]},
                     "contacts": {"model": "constraint"}})
    if model.nmotors:
        eng.set_command(torch.from_numpy(ref["command"]))
    eng.start(torch.from_numpy(ref["q"]), torch.from_numpy(ref["v"]))
    oracle_batch(model, ref, "start", constraint_options=TIGHT)
    loop = ReferenceFixedStepLoop(dt)   # the reference's sub-step rule: opens with a 1 us step (engine.cc:1176)

    def check(tol, what, tol_forces=None):
        torch.cuda.synchronize()
        assert np.array_equal(eng.field("con_flags").cpu().numpy(), ref["con_flags"]), what
        assert np.array_equal(eng.status.cpu().numpy().reshape(-1), ref["status"].reshape(-1)), what
        worst = {}
        for k in OUTS:
            if k in eng._fields and ref[k].size and eng._rows.get(k, 1) > 0:
                worst[k] = rel_err(eng.field(k).cpu().numpy(), ref[k])
        print(f"[{name}] {what}: " + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()))
        for k, e in worst.items():
            # (multipliers and what is derived from them: see the note at the Euler leg below)
            assert e < (tol_forces if (tol_forces is not None and k not in ("q", "v", "a", "imu", "encoder", "energy")) else tol), (what, k, e)

    # device build: FMA contraction + another summation order than the oracle; the PGS fixed point
    # amplifies round-off by 1 / (1 - contraction rate)
    check(1e-7, "start")
    for i in range(4):
        eng.step(dt)
        oracle_engine_step(model, ref, loop, dt, "euler_explicit", command_changed=True, constraint_options=TIGHT)
    # after steps (four periods = five integrator steps: the simulation opens with the reference's 1 us step): the
    # north-star bar (1e-5 relative on accelerations); observed 1e-13 (Atlas) ... 8e-6 (ANYmal, whose PGS solves run
    # into the iteration cap at these tolerances: the iterate reached after 100 sweeps moves with round-off, and the
    # multipliers of its redundant contact rows -- `force`, `contact_forces`, `f_external`, `con_data` -- move more
    # than the accelerations they produce: 1.4e-5 observed, same bar as the Runge-Kutta leg below)
    check(1e-5, "euler", tol_forces=1e-4 if name == "anymal" else None)
    eng.stop()
    eng.set_options({"stepper": {"odeSolver": "runge_kutta_4"}})
    # both sides restart from the SAME state (the oracle's): the constrained acceleration is stiff in v
    # (Baumgarte damping 2 * omega = 250 /s on the contact rows), 1e-8 of state difference would show as 1e-5
    eng.start(torch.from_numpy(ref["q"]), torch.from_numpy(ref["v"]))
    oracle_batch(model, ref, "start", constraint_options=TIGHT)
    loop = ReferenceFixedStepLoop(dt)   # the reference's sub-step rule: opens with a 1 us step (engine.cc:1176)
    check(1e-5 if name != "anymal" else 1e-4, "restart")    # (ANYmal at the iteration cap: 1.1e-5 on `a` observed)
    for i in range(2):
        eng.step(dt)
        oracle_engine_step(model, ref, loop, dt, "runge_kutta_4", command_changed=True, constraint_options=TIGHT)
    # 10 more evaluations: ANYmal's solves sit at the iteration cap with these tolerances (status bit 16 on
    # both sides), the iterate reached after 100 sweeps moves with round-off: 6e-6 on `a` (inside the
    # north-star bar), 1.3e-5 on the contact forces; every other robot stays below 1e-10
    check(1e-4, "rk4")


@pytest.mark.gpu
@pytest.mark.parametrize("name,solver,n_sub,B", [("anymal", "euler_explicit", 2, 96), ("anymal", "runge_kutta_4", 1, 96),
                                                 ("biped", "euler_explicit", 1, 64), ("anymal", "euler_explicit", 1, 4096)])
def test_gpu_split_stepping_with_the_one_lane_per_robot_solve(gpu_device, monkeypatch, name, solver, n_sub, B):
    """Robots with few contact points (ANYmal, bipeds) step through k_quad_con_split<1> | k_qcon_pgs_lane | k_quad_con_split<2>
    when the batch is a multiple of 16 (round 6, jm_qcon.h: the solve holds a robot per lane): same flags and status, states /
    multipliers / outputs at round-off of the single kernel (JIMINY_AMD_QCON_SPLIT=0) and, for the small batches, of the oracle.
    Twelve steps: long enough for the library to read its sweep counters and switch the form of the following steps (robots
    resting on the ground converge in a few sweeps: the single kernel takes over after the first steps; jm_lib.cpp)."""
    import torch

    from jiminy_amd.engine import BatchedEngine
    model = _models()[name]()
    with_oracle = B <= 128
    ref, _ = _pair(model, B, seed=29)
    dt = 5e-4
    engines = []
    for split in ("1", "0"):
        monkeypatch.setenv("JIMINY_AMD_QCON_SPLIT", split)
        eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device,
                            extra_outputs=("contact_forces", "f_external", "joint_forces", "energy", "centroidal"))
        eng.set_options({"stepper": {"odeSolver": solver, "dtMax": dt, "controllerUpdatePeriod": n_sub * dt,
                                     "sensorsUpdatePeriod": n_sub * dt, "tolAbs": TIGHT["tol_abs"], "tolRel": TIGHT["tol_rel"]},
                         "contacts": {"model": "constraint"}})
        eng.set_command(torch.from_numpy(ref["command"]))
        eng.start(torch.from_numpy(ref["q"]), torch.from_numpy(ref["v"]))
        engines.append(eng)
    if with_oracle:
        oracle_batch(model, ref, "start", constraint_options=TIGHT)
        loop = ReferenceFixedStepLoop(dt)
    split, single = engines
    for i in range(12):
        for eng in engines:
            eng.step(n_sub * dt)
        torch.cuda.synchronize()     # (the counters of this step are in: the next one may change form)
        if with_oracle:
            oracle_engine_step(model, ref, loop, n_sub * dt, solver, command_changed=True, constraint_options=TIGHT)
        if i in (0, 2, 11):
            fa, fb = split.field("con_flags").cpu().numpy(), single.field("con_flags").cpu().numpy()
            sa, sb = split.status.cpu().numpy(), single.status.cpu().numpy()
            if B <= 128:
                assert np.array_equal(fa, fb), i
                assert np.array_equal(sa, sb), i
            else:
                assert (fa != fb).any(axis=0).mean() < 2e-3 and (sa != sb).mean() < 2e-3, i     # (a hysteresis tie on a drifting lane)
            for k in OUTS:
                if k in split._fields and split._rows.get(k, 1) > 0:
                    a, b = split.field(k).cpu().numpy(), single.field(k).cpu().numpy()
                    if B <= 128:
                        assert rel_err(a, b) < 1e-8, (i, k, rel_err(a, b))
                    else:
                        # the large batch holds robots whose four feet load redundant contact rows: their solve ends with the
                        # relaxation schedule, not by convergence, and its iterates drift along the null space with the
                        # summation order (the two forms sum differently) -- per-lane bar with a tail
                        per_lane = np.abs(a - b).max(axis=0) / np.maximum(np.abs(b).max(axis=0), 1.0)
                        assert np.quantile(per_lane, 0.99) < 1e-8 and per_lane.max() < 1e-3, (i, k, np.quantile(per_lane, 0.99), per_lane.max())
                    if with_oracle and k in ref and ref[k].size:
                        assert rel_err(a, ref[k]) < 1e-6, (i, k, rel_err(a, ref[k]))
    if with_oracle:
        assert np.array_equal(split.field("con_flags").cpu().numpy(), ref["con_flags"])


@pytest.mark.gpu
@pytest.mark.parametrize("solver,n_sub,B", [("runge_kutta_4", 1, 48), ("runge_kutta_4", 3, 48), ("euler_explicit", 2, 48),
                                            ("runge_kutta_4", 2, 2048 + 80)])
def test_gpu_split_stepping_of_large_solves(gpu_device, monkeypatch, solver, n_sub, B):
    """Atlas-sized solves step through three launches per evaluation (k_quad_con_split<1> | k_qcon_pgs |
    k_quad_con_split<2>, jm_qcon.h) when the batch is a multiple of 16: same states, multipliers, flags and outputs
    as the single kernel (JIMINY_AMD_QCON_SPLIT=0) and as the oracle.  The large batch steps as four chunks on streams of
    their own (the JIMINY_AMD_QCON_SPLIT_CHUNKS option of jm_lib.cpp, launch_quad_con; last chunk ragged): compared with the
    single kernel."""
    import torch

    from jiminy_amd.engine import BatchedEngine
    model = _models()["atlas"]()
    with_oracle = B <= 64
    ref, _ = _pair(model, B, seed=23)
    dt = 2.5e-4
    engines = []
    for split in ("1", "0"):
        monkeypatch.setenv("JIMINY_AMD_QCON_SPLIT", split)
        monkeypatch.setenv("JIMINY_AMD_QCON_SPLIT_CHUNKS", "4" if B > 64 else "1")
        eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device,
                            extra_outputs=("contact_forces", "f_external", "joint_forces", "energy", "centroidal"))
        eng.set_options({"stepper": {"odeSolver": solver, "dtMax": dt, "controllerUpdatePeriod": n_sub * dt,
                                     "sensorsUpdatePeriod": n_sub * dt, "tolAbs": TIGHT["tol_abs"], "tolRel": TIGHT["tol_rel"]},
                         "contacts": {"model": "constraint"}})
        eng.set_command(torch.from_numpy(ref["command"]))
        eng.start(torch.from_numpy(ref["q"]), torch.from_numpy(ref["v"]))
        engines.append(eng)
    if with_oracle:
        oracle_batch(model, ref, "start", constraint_options=TIGHT)
        loop = ReferenceFixedStepLoop(dt)   # the reference's sub-step rule: opens with a 1 us step (engine.cc:1176)
    for i in range(3):
        for eng in engines:
            eng.step(n_sub * dt)
        if with_oracle:
            oracle_engine_step(model, ref, loop, n_sub * dt, solver, command_changed=True, constraint_options=TIGHT)
    torch.cuda.synchronize()
    split, single = engines
    assert np.array_equal(split.field("con_flags").cpu().numpy(), single.field("con_flags").cpu().numpy())
    assert np.array_equal(split.status.cpu().numpy(), single.status.cpu().numpy())
    if with_oracle:
        assert np.array_equal(split.field("con_flags").cpu().numpy(), ref["con_flags"])
    worst = {}
    for k in OUTS:
        if k in split._fields and split._rows.get(k, 1) > 0:
            a, b = split.field(k).cpu().numpy(), single.field(k).cpu().numpy()
            worst[k] = rel_err(a, b)
            assert worst[k] < 1e-9, (k, worst[k])
            if with_oracle and k in ref and ref[k].size:
                assert rel_err(a, ref[k]) < 1e-7, (k, rel_err(a, ref[k]))
    print("split vs single kernel:", ", ".join(f"{k} {v:.1e}" for k, v in worst.items()))


@pytest.mark.gpu
@pytest.mark.parametrize("name,B,freq", [("anymal", 96, None), ("atlas", 48, None), ("atlas", 40, None), ("tree_arm", 64, None),
                                         ("tree_arm_ff", 72, None), ("anymal", 64, 5.0), ("atlas", 48, 0.0), ("tree_arm_ff", 40, 5.0)])
def test_gpu_user_joint_constraints(gpu_device, name, B, freq):
    """`BatchedEngine.add_constraint(name, JointConstraint(joint))` on the device against the oracle: every other lane
    locked (ANYmal: the general Gauss-Seidel form out of LDS; Atlas, 48 lanes: the split form with the unbounded rows first
    in its visit table; 40 lanes: the single kernel; `tree_arm` / `tree_arm_ff`: the one-robot-per-lane kernel), RK4 and Euler steps;
    `remove_constraint` gives the joints back."""
    import torch

    from jiminy_amd.engine import BadControlFlow, BatchedEngine, JointConstraint
    model = _models()[name]()
    # (`freq`: Baumgarte frequency of the locks -- None = the gains of the contacts, else gains of their own, ABI 6)
    COPT = dict(TIGHT, user_stabilization_freq=-1.0 if freq is None else freq)
    ref, _ = _pair(model, B, seed=37)
    lanes = np.arange(B) % 2 == 0
    rows = [model.bound_row(j) for j in LOCKS[name]]
    for r in rows:
        ref["con_flags"][r, lanes] |= 4
    dt = 5e-4 if name == "anymal" else 2.5e-4
    eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device, extra_outputs=("contact_forces", "f_external", "energy"))
    eng.set_options({"stepper": {"odeSolver": "runge_kutta_4", "dtMax": dt, "controllerUpdatePeriod": 2 * dt, "sensorsUpdatePeriod": 2 * dt,
                                 "tolAbs": TIGHT["tol_abs"], "tolRel": TIGHT["tol_rel"]}, "contacts": {"model": "constraint"}})
    mask = torch.from_numpy(lanes).to(gpu_device)
    for j in LOCKS[name]:
        eng.add_constraint("lock_" + j, JointConstraint(j, baumgarte_freq=freq), lane_mask=mask)
    with pytest.raises(ValueError):
        eng.add_constraint("lock_" + LOCKS[name][0], JointConstraint(LOCKS[name][1], baumgarte_freq=freq))
    eng.set_command(torch.from_numpy(ref["command"]))
    eng.start(torch.from_numpy(ref["q"]), torch.from_numpy(ref["v"]))
    with pytest.raises(BadControlFlow):
        eng.remove_constraint("lock_" + LOCKS[name][0])
    oracle_batch(model, ref, "start", constraint_options=COPT)
    loop = ReferenceFixedStepLoop(dt)   # the reference's sub-step rule: opens with a 1 us step (engine.cc:1176)
    for _ in range(3):
        eng.step(2 * dt)
        oracle_engine_step(model, ref, loop, 2 * dt, "runge_kutta_4", command_changed=True, constraint_options=COPT)
    torch.cuda.synchronize()
    assert np.array_equal(eng.field("con_flags").cpu().numpy(), ref["con_flags"])
    for k in OUTS:
        if k in eng._fields and eng._rows.get(k, 1) > 0 and ref[k].size:
            # (ANYmal's solves run close to the iteration cap at these tolerances, see test_gpu_constraint_model_matches_oracle)
            assert rel_err(eng.field(k).cpu().numpy(), ref[k]) < (1e-4 if name == "anymal" else 1e-7), k
    nb = _abi.constraint_rows(model)["n_bounds"]
    assert np.abs(ref["con_data"][[nb + r for r in rows]][:, lanes]).max() > 1e-3
    eng.stop()
    for j in LOCKS[name]:
        eng.remove_constraint("lock_" + j)
    assert not eng.user_constraints and int((eng.field("con_flags") & 4).sum()) == 0


@pytest.mark.gpu
def test_gpu_joint_constraint_follows_its_reference(gpu_device):
    """`set_constraint_reference`: the Baumgarte-stabilised lock (20 Hz, critically damped) pulls its joint to a new
    reference within a few periods; robots in free fall, so that nothing else acts on the joint."""
    import torch

    from jiminy_amd.engine import BatchedEngine, JointConstraint
    from jiminy_amd.synthetic import sample_states
    model = _models()["anymal"]()
    B = 32
    st = sample_states(model, B, seed=3, base_height=(3.0, 4.0), grounded_fraction=0.0)
    eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device)
    eng.set_options({"stepper": {"odeSolver": "euler_explicit", "dtMax": 5e-4, "controllerUpdatePeriod": 5e-3, "sensorsUpdatePeriod": 5e-3},
                     "contacts": {"model": "constraint"}})
    eng.add_constraint("knee", JointConstraint("LF_KFE"))
    eng.set_command(torch.zeros(model.nmotors, B, dtype=torch.float64))
    eng.start(torch.from_numpy(st["q"]), torch.zeros_like(torch.from_numpy(st["v"])))
    iq = int(model.idx_q[model.joint_names.index("LF_KFE")])
    q0 = eng.field("q")[iq].clone()
    target = q0 + torch.linspace(-0.1, 0.1, B, dtype=torch.float64, device=gpu_device)
    eng.set_constraint_reference("knee", target)
    for _ in range(20):
        eng.step(5e-3)
    err = (eng.field("q")[iq] - target).abs()
    assert float(err.max()) < 2e-3, float(err.max())
    assert float((eng.field("q")[iq] - q0).abs().max()) > 0.09


@pytest.mark.gpu
def test_gpu_anymal_stands_still_under_the_constraint_model(gpu_device):
    """Reference acceptance for its quadrupeds / bipeds (gym_jiminy unit_py/test_pipeline_control.py:46-133):
    the robot keeps standing.  Here: ANYmal at its neutral stance with the shipped options
    (euler_explicit, constraint contacts), zero command: after 0.5 s the base has not fallen."""
    import torch

    from jiminy_amd.engine import BatchedEngine
    model = load_builtin("anymal")
    B = 64
    st = sample_standing_states(model, B, seed=0, joint_noise=0.0, base_angle_max=0.0, twist_std=0.0,
                                joint_vel_std=0.0, command_fraction=0.0, out_of_bounds_fraction=0.0,
                                depth_range=(-1e-4, 0.0))
    eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device)
    eng.set_options({"stepper": {"odeSolver": "euler_explicit", "dtMax": 1e-3, "controllerUpdatePeriod": 1e-3,
                                 "sensorsUpdatePeriod": 1e-3}, "contacts": {"model": "constraint"}})
    eng.set_command(torch.zeros((model.nmotors, B), dtype=torch.float64))
    eng.start(torch.from_numpy(st["q"]), torch.from_numpy(st["v"]))
    z0 = eng.field("q")[2].clone()
    for _ in range(100):
        eng.step(1e-3)
    torch.cuda.synchronize()
    assert not (eng.status & _abi.JM_LANE_NAN).any()
    assert torch.isfinite(eng.field("q")).all()
    # unactuated legs fold slowly; the feet stay on the ground and the base does not free-fall
    # (free fall over 0.1 s would be 4.9 cm)
    assert (z0 - eng.field("q")[2]).max() < 0.03
    fz = eng.field("contact_forces").reshape(model.ncontacts, 6, B)[:, 2].sum(0)
    assert (fz > 0.5 * 30 * G).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["anymal", "atlas"])
def test_gpu_constrained_solution_satisfies_the_equation_of_motion(name, gpu_device):
    """Independent of the oracle: the device outputs (a, u, f_external, multipliers) of the constraint
    model satisfy RNEA(q, v, a, f_ext) + rotor a = u with the numpy RNEA, after several steps with the
    reference's default PGS tolerances."""
    import torch

    from jiminy_amd.engine import BatchedEngine
    model = load_builtin(name)
    B = 32
    st = sample_standing_states(model, B, seed=13, out_of_bounds_fraction=0.5)
    eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device,
                        extra_outputs=("contact_forces", "f_external"))
    dt = 1e-3
    eng.set_options({"stepper": {"odeSolver": "euler_explicit", "dtMax": dt, "controllerUpdatePeriod": dt,
                                 "sensorsUpdatePeriod": dt}, "contacts": {"model": "constraint"}})
    eng.set_command(torch.from_numpy(st["command"]))
    eng.start(torch.from_numpy(st["q"]), torch.from_numpy(st["v"]))
    for _ in range(5):
        eng.step(dt)
    torch.cuda.synchronize()
    f = {k: eng.field(k).cpu().numpy() for k in ("q", "v", "a", "u", "f_external", "con_flags", "con_data")}
    rows = _abi.constraint_rows(model)
    nb = rows["n_bounds"]
    bounded = [j for j in range(1, model.njoints) if 1 <= int(model.jtypes[j]) <= 8]
    n_contact = n_bound = 0
    for l in range(B):
        u = f["u"][:, l].copy()
        for k, j in enumerate(bounded):
            if f["con_flags"][k, l] & 2:
                u[int(model.idx_v[j])] -= 2.0 * f["con_data"][nb + k, l]
            n_bound += int(f["con_flags"][k, l] & 1)
        fext = f["f_external"][:, l].reshape(-1, 6)
        n_contact += int(np.abs(fext).sum() > 0)
        tau = rbd.rnea(model, f["q"][:, l], f["v"][:, l], f["a"][:, l], fext) + model.rotor_inertia * f["a"][:, l]
        scale = max(1.0, np.abs(u).max(), np.abs(tau).max())
        assert np.abs(tau - u).max() / scale < 1e-9, (l, np.abs(tau - u).max())
        lam = f["con_data"][2 * nb:, l].reshape(-1, 4)
        assert (lam[:, 2] >= 0).all() and (f["con_data"][nb:2 * nb, l] >= 0).all()
        assert (np.hypot(lam[:, 0], lam[:, 1]) <= lam[:, 2] * (1 + 1e-9) + 1e-9).all()
    assert n_contact >= B // 2 and n_bound >= 4


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["anymal", "atlas", "crane_walker", "tree_arm", "cartpole", "biped"])
def test_gpu_constraint_kernel_self_test(name, gpu_device):
    """The engine's own guard against an unsound build of the constraint kernel:
    equation-of-motion residual from the device's RNEA outputs, run once per topology."""
    from jiminy_amd import codegen
    from jiminy_amd.engine import _constraint_self_test
    model = _models()[name]()
    err = _constraint_self_test(model, codegen.preferred_variant(model), gpu_device)
    assert err < 1e-8, err
    # ... and the rows its output pass emits: with no constraint active they are the spring-damper kernels' rows (round 5)
    from jiminy_amd.engine import _constraint_rows_self_test
    err_rows = _constraint_rows_self_test(model, codegen.preferred_variant(model), gpu_device)
    assert err_rows < 1e-8, err_rows


@pytest.mark.gpu
def test_gpu_constraint_model_under_the_adaptive_stepper(gpu_device):
    """The reference's two defaults together: `contacts.model = "constraint"` + `runge_kutta_dopri`.
    The constraint state of the active lanes travels with them through the compact batches.  As for the
    spring-damper model the accept / reject decisions are discontinuous, so the agreement with the
    oracle's restatement of the reference loop is statistical."""
    import torch

    from jiminy_amd.engine import BatchedEngine, plan_breakpoints
    from oracle.oracle_py import adaptive_state
    from tests.helpers import oracle_io
    model = load_builtin("anymal")
    B = 96
    st = sample_standing_states(model, B, seed=21, out_of_bounds_fraction=0.2, command_fraction=0.1)
    ref = alloc_soa(model, B)
    alloc_constraint_state(model, ref, B)
    for k in ("q", "v", "command"):
        ref[k][:] = st[k]
    tol_rel, tol_abs, ctrl = 1e-4, 1e-5, 5e-3
    copt = dict(tol_abs=tol_abs, tol_rel=tol_rel)
    orc = OracleEngine(model)
    orc.set_constraint_options(**copt)
    orc.bind_constraints(ref["con_flags"], ref["con_data"])
    io = oracle_io(ref)
    orc.batch_run("start", io)
    ad = adaptive_state(B)
    eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device)
    eng.set_options({"stepper": {"odeSolver": "runge_kutta_dopri", "tolRel": tol_rel, "tolAbs": tol_abs,
                                 "controllerUpdatePeriod": ctrl, "sensorsUpdatePeriod": ctrl},
                     "contacts": {"model": "constraint"}})
    eng.set_command(torch.from_numpy(st["command"]))
    eng.start(torch.from_numpy(st["q"]), torch.from_numpy(st["v"]))
    torch.cuda.synchronize()
    assert rel_err(eng.field("a").cpu().numpy(), ref["a"]) < 1e-7
    t, t_err = 0.0, 0.0
    for _ in range(4):
        intervals, t_end, t_err = plan_breakpoints(t, t_err, ctrl, eng.get_options())
        for i, (t_next, cmd, sens) in enumerate(intervals):
            orc.batch_run_dopri(io, ad, t_next, tol_rel=tol_rel, tol_abs=tol_abs, new_step=(i == 0),
                                command_changed=bool(cmd), update_sensors=sens)
        t = t_end
        eng.step(ctrl)
    torch.cuda.synchronize()
    assert abs(eng.stepper_state.t - 0.02) < 1e-12
    stt = eng.status.cpu().numpy().reshape(-1)
    ok = ((ref["status"][0] & 9) == 0) & ((stt & 9) == 0)
    assert ok.mean() > 0.9
    err = np.abs(eng.field("q").cpu().numpy() - ref["q"]).max(axis=0)[ok]
    assert np.median(err) < 1e-6 and (err < 1e-3).mean() > 0.9, (np.median(err), err.max())
    same_flags = (eng.field("con_flags").cpu().numpy() == ref["con_flags"]).all(axis=0)[ok]
    assert same_flags.mean() > 0.9
    # the robots are still standing on the ground
    fz = eng.field("contact_forces").reshape(model.ncontacts, 6, B)[:, 2].sum(0).cpu().numpy()
    assert (fz[ok] > 0).mean() > 0.9


def test_per_lane_friction_on_the_host():
    """`JM_F_FRICTION`: every lane solves its friction cones with its own coefficient (ground-friction
    randomisation of the reference's locomotion envs): kernel sources on the host vs the oracle, and the
    cone bound |lambda_t| <= mu_lane * lambda_n."""
    model = load_builtin("anymal")
    B = 12
    ref, got = _pair(model, B, seed=9)
    mu = np.linspace(0.05, 1.5, B)
    ref["friction"] = mu.copy()
    got["friction"] = mu.copy()
    oracle_batch(model, ref, "start", constraint_options=TIGHT)
    emu.run(model, got, "start", constraint_options=TIGHT)
    for _ in range(3):
        kw = dict(solver="euler_explicit", dt=5e-4, n_substeps=1, command_changed=True)
        oracle_batch(model, ref, "step", constraint_options=TIGHT, **kw)
        emu.run(model, got, "step", constraint_options=TIGHT, **kw)
    ref.pop("friction"), got.pop("friction")
    _check(got, ref, 1e-7, "per-lane friction")
    nb = _abi.constraint_rows(model)["n_bounds"]
    lam = got["con_data"][2 * nb:].reshape(-1, 4, B)
    assert (np.hypot(lam[:, 0], lam[:, 1]) <= mu[None, :] * lam[:, 2] * (1 + 1e-9) + 1e-9).all()
    assert (lam[:, 2] > 0).any()


@pytest.mark.gpu
def test_gpu_per_lane_friction_and_env_ground_randomisation(gpu_device):
    """Device build: per-lane friction against the oracle, then the ANYmal environment with the reference's
    shipped contact model and `std_ratio={'ground': ...}`: every environment gets its own friction
    coefficient at reset, the multipliers stay inside each lane's own cone."""
    import torch

    from jiminy_amd.engine import BatchedEngine
    from jiminy_amd.envs import make_anymal_env
    model = load_builtin("anymal")
    B = 64
    ref, _ = _pair(model, B, seed=15)
    mu = np.linspace(0.05, 1.8, B)
    ref["friction"] = mu.copy()
    eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device)
    dt = 5e-4
    eng.set_options({"stepper": {"odeSolver": "euler_explicit", "dtMax": dt, "controllerUpdatePeriod": dt,
                                 "sensorsUpdatePeriod": dt, "tolAbs": TIGHT["tol_abs"], "tolRel": TIGHT["tol_rel"]},
                     "contacts": {"model": "constraint"}})
    eng.set_lane_friction(mu)
    eng.set_command(torch.from_numpy(ref["command"]))
    eng.start(torch.from_numpy(ref["q"]), torch.from_numpy(ref["v"]))
    oracle_batch(model, ref, "start", constraint_options=TIGHT)
    loop = ReferenceFixedStepLoop(dt)   # the reference's sub-step rule: opens with a 1 us step (engine.cc:1176)
    for _ in range(3):
        eng.step(dt)
        oracle_engine_step(model, ref, loop, dt, "euler_explicit", command_changed=True, constraint_options=TIGHT)
    torch.cuda.synchronize()
    assert np.array_equal(eng.field("con_flags").cpu().numpy(), ref["con_flags"])
    for k in ("q", "v", "a", "con_data"):
        assert rel_err(eng.field(k).cpu().numpy(), ref[k]) < 1e-5, k

    env = make_anymal_env(256, device=gpu_device, contact_model="constraint", std_ratio={"ground": 0.3})
    env.reset(seed=5)
    fr = env.engine.field("friction")[0].clone()
    lo, hi = 10 ** (1.1 - 0.27), 10 ** (1.1 + 0.27)
    assert float(fr.min()) >= lo - 1e-9 and float(fr.max()) <= hi + 1e-9 and float(fr.std()) > 0.5
    action = torch.zeros((256, model.nmotors), dtype=torch.float64, device=gpu_device)
    for _ in range(2):
        env.step(action)
    nb = _abi.constraint_rows(model)["n_bounds"]
    lam = env.engine.field("con_data")[2 * nb:].reshape(-1, 4, 256)
    assert bool((torch.hypot(lam[:, 0], lam[:, 1]) <= fr[None, :] * lam[:, 2] * (1 + 1e-9) + 1e-9).all())
    assert bool((lam[:, 2] > 0).any())


@pytest.mark.gpu
def test_gpu_constraint_model_full_size_replicas_and_repeatability(gpu_device):
    """BASELINE size (ANYmal, B = 65 536) with the constraint model, through size-independent properties: the
    batch is 256 replicas of one seeded 256-lane block, so every replica must equal the first one bit for bit
    wherever it sits in the grid (state, multipliers, flags), the first block must match the oracle, and a second
    run from the same state reproduces the first bit for bit (reference pin: test_pipeline_control.py:315-330)."""
    import torch

    from jiminy_amd.engine import BatchedEngine
    model = load_builtin("anymal")
    blk, reps, dt, steps = 256, 256, 1e-3, 3
    B = blk * reps
    st = sample_standing_states(model, blk, seed=17)
    q = torch.from_numpy(np.tile(st["q"], (1, reps)))
    v = torch.from_numpy(np.tile(st["v"], (1, reps)))
    cmd = torch.from_numpy(np.tile(st["command"], (1, reps)))
    ref = alloc_soa(model, blk)
    alloc_constraint_state(model, ref, blk)
    for k in ("q", "v", "command"):
        ref[k][:] = st[k]
    oracle_batch(model, ref, "start", constraint_options={})
    loop = ReferenceFixedStepLoop(dt)   # the reference's sub-step rule: opens with a 1 us step (engine.cc:1176)
    for _ in range(steps):
        oracle_engine_step(model, ref, loop, dt, "euler_explicit", command_changed=True, constraint_options={})
    eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device)
    eng.set_options({"stepper": {"odeSolver": "euler_explicit", "dtMax": dt, "controllerUpdatePeriod": dt,
                                 "sensorsUpdatePeriod": dt}, "contacts": {"model": "constraint"}})
    names = ("q", "v", "a", "imu", "contact_forces", "con_data", "con_flags")
    runs = []
    for _ in range(2):
        eng.set_command(cmd)
        eng.start(q, v)
        for _ in range(steps):
            eng.step(dt)
        torch.cuda.synchronize()
        runs.append({k: eng.field(k).clone() for k in names})
        eng.stop()
    for k in names:
        x = runs[0][k]
        first = x[:, :blk]
        assert torch.equal(x.view(x.shape[0], reps, blk), first[:, None, :].expand(-1, reps, -1)), k
        assert torch.equal(runs[0][k], runs[1][k]), k
    assert np.array_equal(runs[0]["con_flags"][:, :blk].cpu().numpy(), ref["con_flags"])
    # default PGS tolerances: same iterates, same stopping sweep on both sides up to rare ties
    for k in ("q", "v", "a", "con_data"):
        assert rel_err(runs[0][k][:, :blk].cpu().numpy(), ref[k]) < 1e-5, k


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 3, 65])
def test_gpu_constraint_model_ragged_and_tiny_batches(gpu_device, B):
    """Batch sizes that do not fill a wave: tail lanes computed, padding lanes silent."""
    import torch

    from jiminy_amd.engine import BatchedEngine
    model = load_builtin("anymal")
    ref, _ = _pair(model, B, seed=23)
    dt = 1e-3
    eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device)
    eng.set_options({"stepper": {"odeSolver": "runge_kutta_4", "dtMax": dt, "controllerUpdatePeriod": dt,
                                 "sensorsUpdatePeriod": dt, "tolAbs": TIGHT["tol_abs"], "tolRel": TIGHT["tol_rel"]},
                     "contacts": {"model": "constraint"}})
    eng.set_command(torch.from_numpy(ref["command"]))
    eng.start(torch.from_numpy(ref["q"]), torch.from_numpy(ref["v"]))
    oracle_batch(model, ref, "start", constraint_options=TIGHT)
    loop = ReferenceFixedStepLoop(dt)   # the reference's sub-step rule: opens with a 1 us step (engine.cc:1176)
    for _ in range(2):
        eng.step(dt)
        oracle_engine_step(model, ref, loop, dt, "runge_kutta_4", command_changed=True, constraint_options=TIGHT)
    torch.cuda.synchronize()
    assert np.array_equal(eng.field("con_flags").cpu().numpy(), ref["con_flags"])
    for k in ("q", "v", "a", "con_data", "imu"):
        assert rel_err(eng.field(k).cpu().numpy(), ref[k]) < 1e-5, k


@pytest.mark.gpu
def test_gpu_constraint_model_long_horizon(gpu_device):
    """The north-star criterion on this path: 1000 steps (Euler 1 ms, the shipped ANYmal solver, default PGS
    tolerances), robots standing under zero command and slowly folding onto their joint limits: contacts make
    and break, some forty joint-bound constraints switch on.  Every lane must stay within 1e-5 relative on the
    generalised accelerations at every check point (observed: median 8e-13, worst lane 3e-11)."""
    import torch

    from jiminy_amd.engine import BatchedEngine
    model = load_builtin("anymal")
    B, dt, steps = 64, 1e-3, 1000
    st = sample_standing_states(model, B, seed=29, out_of_bounds_fraction=0.0, command_fraction=0.0,
                                joint_noise=0.05, twist_std=0.02, joint_vel_std=0.05)
    ref = alloc_soa(model, B)
    alloc_constraint_state(model, ref, B)
    for k in ("q", "v"):
        ref[k][:] = st[k]
    eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device)
    eng.set_options({"stepper": {"odeSolver": "euler_explicit", "dtMax": dt, "controllerUpdatePeriod": 0.0,
                                 "sensorsUpdatePeriod": 0.0}, "contacts": {"model": "constraint"}})
    eng.set_command(torch.zeros((model.nmotors, B), dtype=torch.float64))
    eng.start(torch.from_numpy(ref["q"]), torch.from_numpy(ref["v"]))
    oracle_batch(model, ref, "start", constraint_options={})
    loop = ReferenceFixedStepLoop(dt)   # the reference's sub-step rule: opens with a 1 us step (engine.cc:1176)
    worst = np.zeros(B)
    for i in range(steps // 50):
        eng.step(50 * dt)
        oracle_engine_step(model, ref, loop, 50 * dt, "euler_explicit", command_changed=False, constraint_options={})
        a = eng.field("a").cpu().numpy()
        err = np.abs(a - ref["a"]).max(axis=0) / np.maximum(np.abs(ref["a"]).max(axis=0), 1.0)
        worst = np.maximum(worst, err)
    stt = eng.status.cpu().numpy().reshape(-1)
    assert ((stt & 1) == 0).all() and ((ref["status"][0] & 1) == 0).all()
    n_bounds = int((ref["con_flags"][: _abi.constraint_rows(model)["n_bounds"]] & 1).sum())
    print(f"long horizon: median {np.median(worst):.1e}, 90th pct {np.quantile(worst, 0.9):.1e}, max {worst.max():.1e}, "
          f"active bound constraints at the end {n_bounds}")
    assert n_bounds >= 10
    assert np.median(worst) <= 1e-9 and worst.max() <= 1e-5, (np.median(worst), worst.max())


@pytest.mark.gpu
def test_gpu_solver_region_stays_inside_its_workspace(gpu_device):
    """Device twin of the guard-row check of the host emulation: Atlas standing flat on both feet (the largest solve of
    the shipped robots: 16 contact points x 4 rows in the start passes + its joint bounds) with a workspace that carries
    256 sentinel rows behind the rows the library asked for -- they must come back untouched, and the lanes unflagged."""
    import torch

    from jiminy_amd.engine import BatchedEngine
    from jiminy_amd.synthetic import lowest_contact_height
    model = load_builtin("atlas")
    B = 48
    q = model.neutral()
    for name, value in {"back_bky": 0.2, "l_arm_elx": 0.2, "l_arm_shx": -np.pi / 2, "l_arm_shz": np.pi / 4, "l_arm_ely": 3 * np.pi / 4,
                        "r_arm_elx": -0.2, "r_arm_shx": np.pi / 2, "r_arm_shz": -np.pi / 4, "r_arm_ely": 3 * np.pi / 4}.items():
        q[int(model.idx_q[model.joint_names.index(name)])] = value
    mask = model.bounded_position_mask()
    q[mask] = np.clip(q[mask], model.position_lower[mask], model.position_upper[mask])
    q[2] -= float(lowest_contact_height(model, q)[0]) + 2.0e-3
    eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device)
    eng.set_options({"stepper": {"odeSolver": "euler_explicit", "dtMax": 1e-3, "controllerUpdatePeriod": 5e-3, "sensorsUpdatePeriod": 5e-3},
                     "contacts": {"model": "constraint"}})
    rows = eng._fields["workspace"].shape[0]
    guarded = torch.zeros((rows + 256, B), dtype=torch.float64, device=gpu_device)
    guarded[rows:] = -12345.678
    eng._fields["workspace"] = guarded
    eng._bind("workspace")
    eng.set_command(torch.zeros(model.nmotors, B, dtype=torch.float64))
    eng.start(torch.from_numpy(np.tile(q[:, None], (1, B))), torch.zeros(model.nv, B, dtype=torch.float64))
    for _ in range(4):
        eng.mark_command_changed()
        eng.step(5e-3)
    torch.cuda.synchronize()
    assert bool((guarded[rows:] == -12345.678).all())
    nb = _abi.constraint_rows(model)["n_bounds"]
    assert int((eng.field("con_flags")[nb:, 0] & 1).sum()) == 16
    assert int((eng.status & ~16).abs().sum()) == 0


@pytest.mark.gpu
def test_gpu_masked_reset_through_the_split_start_path(gpu_device, monkeypatch):
    """A reset of a FEW lanes of a robot whose `start` / `reset` launches go through the split pipeline (Atlas, 96 lanes: four
    pre passes, the exact solves, the sweep kernels, the post pass over the whole batch, every robot that does not restart
    leaving through its header): the untouched lanes keep q / v / a / multipliers / flags bit for bit, and the restarted lanes
    end where the single kernel (`JIMINY_AMD_QCON_SPLIT_START=0`) puts them."""
    import torch

    from jiminy_amd.engine import BatchedEngine
    model = load_builtin("atlas")
    B, dt = 96, 5e-4
    ref, _ = _pair(model, B, seed=41)
    other = sample_standing_states(model, B, seed=43)
    lanes = np.zeros(B, dtype=bool)
    lanes[[3, 17, 18, 64, 95]] = True
    results = {}
    for split in ("1", "0"):
        monkeypatch.setenv("JIMINY_AMD_QCON_SPLIT_START", split)
        eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device, extra_outputs=("contact_forces",))
        eng.set_options({"stepper": {"odeSolver": "euler_explicit", "dtMax": dt, "controllerUpdatePeriod": dt, "sensorsUpdatePeriod": dt,
                                     "tolAbs": TIGHT["tol_abs"], "tolRel": TIGHT["tol_rel"]}, "contacts": {"model": "constraint"}})
        eng.set_command(torch.from_numpy(ref["command"]))
        eng.start(torch.from_numpy(ref["q"]), torch.from_numpy(ref["v"]))
        for _ in range(3):
            eng.step(dt)
        torch.cuda.synchronize()
        keys = ("q", "v", "a", "con_data", "con_flags", "contact_forces")
        before = {k: eng.field(k).clone() for k in keys}
        eng.reset_lanes(torch.from_numpy(lanes.astype(np.uint8)).to(gpu_device), torch.from_numpy(other["q"]), torch.from_numpy(other["v"]))
        torch.cuda.synchronize()
        after = {k: eng.field(k).clone() for k in keys}
        keep = torch.from_numpy(~lanes).to(gpu_device)
        for k in keys:
            assert torch.equal(after[k][..., keep], before[k][..., keep]), (split, k)
        assert torch.equal(after["q"][:, ~keep], torch.from_numpy(other["q"][:, lanes]).to(gpu_device))
        assert not torch.equal(after["a"][:, ~keep], before["a"][:, ~keep])
        eng.step(dt)
        torch.cuda.synchronize()
        results[split] = {k: eng.field(k).cpu().numpy().astype(np.float64) for k in ("q", "v", "a", "con_data")}
        eng.stop()
    for k, x in results["1"].items():
        assert rel_err(x, results["0"][k]) < 1e-7, (k, rel_err(x, results["0"][k]))
