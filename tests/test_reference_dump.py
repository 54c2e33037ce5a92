"""Pins against the REAL reference binary: consumes `tests/golden/ref_<model>.npz` written by
`tools/dump_reference.py` on a host where jiminy_py is installed.  Neither this container nor the GPU
box can import jiminy_py, so the files may be absent: the tests are then SKIPPED (and DESIGN.md keeps
saying "parity unpinned against the binary").  The first host that runs the dump script turns them on.

Checks, per model: (1) the model compiler reproduces `Robot::pinocchioModel_` (joint order, indices,
placements, lumped inertias, rotor inertias, limits, motor / contact order); (2) the oracle's
`compute_robots_dynamics` and `start` accelerations; (3) 20 RK4 steps; (4) on the GPU, the HIP path
against the same vectors."""
import os

import numpy as np
import pytest

from jiminy_amd import load_builtin
from tests.helpers import ReferenceFixedStepLoop, alloc_soa, oracle_batch, oracle_engine_step, rel_err

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["double_pendulum", "cartpole", "anymal", "atlas"]


def _load(name):
    path = os.path.join(GOLDEN, f"ref_{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} absent: run tools/dump_reference.py where jiminy_py is installed")
    return np.load(path, allow_pickle=False)


@pytest.mark.parametrize("name", NAMES)
def test_model_compiler_reproduces_the_pinocchio_model(name):
    g = _load(name)
    m = load_builtin(name)
    assert [str(x) for x in g["pin_joint_names"]] == list(m.joint_names)
    assert np.array_equal(g["pin_parents"], m.parents)
    assert np.array_equal(g["pin_idx_q"][1:], m.idx_q[1:]) and np.array_equal(g["pin_idx_v"][1:], m.idx_v[1:])
    assert np.abs(g["pin_placement_R"] - m.placement_R).max() < 1e-12
    assert np.abs(g["pin_placement_p"] - m.placement_p).max() < 1e-12
    assert np.abs(g["pin_mass"] - m.mass).max() < 1e-12 * max(m.mass.max(), 1.0)
    assert np.abs(g["pin_com"] - m.com).max() < 1e-12
    assert np.abs(g["pin_inertia"] - m.inertia).max() < 1e-12 * max(np.abs(m.inertia).max(), 1.0)
    assert np.abs(g["pin_rotor_inertia"] - m.rotor_inertia).max() < 1e-12
    fin = np.isfinite(m.position_lower)
    assert np.abs(g["pin_position_lower"][fin] - m.position_lower[fin]).max() < 1e-12
    assert [str(x) for x in g["pin_motor_names"]] == [mo.name for mo in m.motors]
    assert [str(x) for x in g["pin_contact_frame_names"]] == list(m.contacts)


def _oracle_arrays(model, g):
    B = g["in_q"].shape[1]
    arr = alloc_soa(model, B)
    arr["q"][:], arr["v"][:] = g["in_q"], g["in_v"]
    if model.nmotors:
        arr["command"][:] = g["in_command"]
    return arr, B


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_the_reference_binary(name):
    g = _load(name)
    model = load_builtin(name)
    arr, B = _oracle_arrays(model, g)
    started = np.isfinite(g["start_a"]).all(axis=0)
    assert started.any()
    oracle_batch(model, arr, "start")
    assert rel_err(arr["a"], g["start_a"], started) < 1e-9
    assert rel_err(arr["a"], g["dynamics_a"], started) < 1e-9
    dt = float(g["dt"])
    worst = 0.0
    loop = ReferenceFixedStepLoop(dt)     # the binary's first `step(dt)` is 1 us + the rest (engine.cc:1176)
    for i in range(g["traj_a"].shape[0]):
        oracle_engine_step(model, arr, loop, dt, "runge_kutta_4", command_changed=False)
        ok = started & (arr["status"][0] == 0) & np.isfinite(g["traj_a"][i]).all(axis=0)
        for k in ("q", "v", "a"):
            worst = max(worst, rel_err(arr[k], g["traj_" + k][i], ok))
    # north_star: 1e-5 relative on the generalised accelerations
    assert worst < 1e-5, worst


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_path_matches_the_reference_binary(gpu_device, name):
    import torch

    from jiminy_amd.engine import BatchedEngine
    g = _load(name)
    model = load_builtin(name)
    B = g["in_q"].shape[1]
    dt = float(g["dt"])
    eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device)
    eng.set_options({"stepper": {"odeSolver": "runge_kutta_4", "dtMax": dt, "controllerUpdatePeriod": dt,
                                 "sensorsUpdatePeriod": dt}, "contacts": {"model": "spring_damper"}})
    if model.nmotors:
        eng.set_command(torch.from_numpy(g["in_command"]))
    eng.start(torch.from_numpy(g["in_q"]), torch.from_numpy(g["in_v"]))
    started = np.isfinite(g["start_a"]).all(axis=0)
    assert rel_err(eng.field("a").cpu().numpy(), g["start_a"], started) < 1e-9
    a = eng.compute_robots_dynamics(0.0, torch.from_numpy(g["in_q"]), torch.from_numpy(g["in_v"]))
    assert rel_err(a.cpu().numpy(), g["dynamics_a"], started) < 1e-9
    worst = 0.0
    for i in range(g["traj_a"].shape[0]):
        eng.step(dt)
        ok = started & (eng.status.cpu().numpy() == 0) & np.isfinite(g["traj_a"][i]).all(axis=0)
        for k in ("q", "v", "a"):
            worst = max(worst, rel_err(eng.field(k).cpu().numpy(), g["traj_" + k][i], ok))
    assert worst < 1e-5, worst


# ---- the contact model the reference's shipped options select (`contacts.model = "constraint"`), standing robots
CON_NAMES = ["anymal", "atlas"]
CON_TIGHT = dict(tol_abs=1e-11, tol_rel=1e-10)


def _load_con(name):
    g = _load(name)
    if "con_traj_a" not in g.files:
        pytest.skip("the dump predates the constraint-model trajectory: re-run tools/dump_reference.py")
    return g


@pytest.mark.parametrize("name", CON_NAMES)
def test_oracle_constraint_model_matches_the_reference_binary(name):
    from oracle.oracle_py import OracleEngine
    from tests.helpers import alloc_constraint_state, oracle_io
    g = _load_con(name)
    model = load_builtin(name)
    B = g["con_in_q"].shape[1]
    arr = alloc_soa(model, B)
    alloc_constraint_state(model, arr, B)
    arr["q"][:], arr["v"][:], arr["command"][:] = g["con_in_q"], g["con_in_v"], g["con_in_command"]
    e = OracleEngine(model)
    e.set_constraint_options(**CON_TIGHT)
    e.bind_constraints(arr["con_flags"], arr["con_data"])
    io = oracle_io(arr)
    started = np.isfinite(g["con_start_a"]).all(axis=0)
    assert started.any()
    e.batch_run("start", io)
    assert rel_err(arr["a"], g["con_start_a"], started) < 1e-7
    dt = float(g["con_dt"])
    worst = 0.0
    loop = ReferenceFixedStepLoop(dt)
    for i in range(g["con_traj_a"].shape[0]):
        loop.advance(lambda h, first: e.batch_run("step", io, solver="euler_explicit", dt=h, n_substeps=1, command_changed=first), dt, True)
        ok = started & ((arr["status"][0] & 1) == 0) & np.isfinite(g["con_traj_a"][i]).all(axis=0)
        for k in ("q", "v", "a"):
            worst = max(worst, rel_err(arr[k], g["con_traj_" + k][i], ok))
    assert worst < 1e-5, worst


@pytest.mark.gpu
@pytest.mark.parametrize("name", CON_NAMES)
def test_hip_constraint_model_matches_the_reference_binary(gpu_device, name):
    import torch

    from jiminy_amd.engine import BatchedEngine
    g = _load_con(name)
    model = load_builtin(name)
    B = g["con_in_q"].shape[1]
    dt = float(g["con_dt"])
    eng = BatchedEngine(model, B, dtype=torch.float64, device=gpu_device)
    eng.set_options({"stepper": {"odeSolver": "euler_explicit", "dtMax": dt, "controllerUpdatePeriod": dt,
                                 "sensorsUpdatePeriod": dt, "tolAbs": CON_TIGHT["tol_abs"], "tolRel": CON_TIGHT["tol_rel"]},
                     "contacts": {"model": "constraint"}})
    eng.set_command(torch.from_numpy(g["con_in_command"]))
    eng.start(torch.from_numpy(g["con_in_q"]), torch.from_numpy(g["con_in_v"]))
    started = np.isfinite(g["con_start_a"]).all(axis=0)
    assert rel_err(eng.field("a").cpu().numpy(), g["con_start_a"], started) < 1e-7
    worst = 0.0
    for i in range(g["con_traj_a"].shape[0]):
        eng.mark_command_changed()
        eng.step(dt)
        ok = started & ((eng.status.cpu().numpy() & 1) == 0) & np.isfinite(g["con_traj_a"][i]).all(axis=0)
        for k in ("q", "v", "a"):
            worst = max(worst, rel_err(eng.field(k).cpu().numpy(), g["con_traj_" + k][i], ok))
    assert worst < 1e-5, worst
