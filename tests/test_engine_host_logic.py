"""Host-side logic of the engine facade that does not need a GPU."""
import math

import numpy as np
import pytest

from jiminy_amd import engine as E


def _opts(**stepper):
    o = E.default_options()
    o["stepper"].update(stepper)
    return o


def test_default_options_follow_the_reference_names():
    o = E.default_options()
    assert o["world"]["gravity"] == [0.0, 0.0, -9.81, 0.0, 0.0, 0.0]
    assert o["contacts"]["stiffness"] == 1.0e6 and o["contacts"]["damping"] == 2.0e3
    assert o["contacts"]["transitionEps"] == 1.0e-3 and o["contacts"]["transitionVelocity"] == 1.0e-2
    assert o["stepper"]["dtMax"] == 0.02
    # reference defaults (engine.h:273, :307): the constraint contact model and the adaptive stepper; float32
    # engines have no constraint solver and start with the spring-damper model
    import torch
    assert o["contacts"]["model"] == "constraint" and o["stepper"]["odeSolver"] == "runge_kutta_dopri"
    assert E.default_options(torch.float32)["contacts"]["model"] == "spring_damper"


def test_plan_gym_style_step():
    # shipped ANYmal options: 5 ms control/sensor period, 40 ms env step (anymal.py:20)
    o = _opts(dtMax=1e-3, controllerUpdatePeriod=5e-3, sensorsUpdatePeriod=5e-3)
    launches, t_end, t_err = E.plan_step(0.0, 0.0, 0.04, o)
    assert len(launches) == 8
    for dt, n, cmd, sens in launches:
        assert dt == pytest.approx(1e-3) and n == 5 and cmd and sens
    assert t_end == pytest.approx(0.04)


def test_plan_single_fixed_step_and_default_step_size():
    o = _opts(dtMax=1e-3, controllerUpdatePeriod=1e-3, sensorsUpdatePeriod=1e-3)
    launches, t_end, _ = E.plan_step(0.0, 0.0, -1.0, o)       # default = controller period
    assert launches == [(pytest.approx(1e-3), 1, True, True)]
    # continuous controller, dtMax sub-steps, last one shortened to land on t_end
    o = _opts(dtMax=1e-3, controllerUpdatePeriod=0.0, sensorsUpdatePeriod=0.0)
    launches, t_end, _ = E.plan_step(0.0, 0.0, 2.5e-3, o)
    assert [(round(dt, 9), n) for dt, n, _, _ in launches] == [(1e-3, 2), (5e-4, 1)]
    assert launches[0][2] and not launches[1][2] and launches[-1][3]


def test_plan_sensor_period_multiple_of_controller_period():
    o = _opts(dtMax=2e-3, controllerUpdatePeriod=2e-3, sensorsUpdatePeriod=4e-3)
    launches, _, _ = E.plan_step(0.0, 0.0, 8e-3, o)
    assert [l[3] for l in launches] == [False, True, False, True]
    assert all(l[2] for l in launches)


def test_time_accumulation_is_compensated():
    o = _opts(dtMax=1e-3, controllerUpdatePeriod=1e-3, sensorsUpdatePeriod=1e-3)
    t, err = 0.0, 0.0
    for _ in range(10000):
        launches, t, err = E.plan_step(t, err, 1e-3, o)
        assert len(launches) == 1 and launches[0][1] == 1
    assert abs(t - 10.0) < 1e-12


def test_substeps_follow_the_reference_rule_for_a_non_dividing_dtmax():
    """engine.cc:2063-2089 for a fixed-step solver: (i) what is left of an interval after a `dtMax` step is merged
    into that step when it is below clamp(0.1 dt, 1e-10, 1e-6); (ii) steps that are not a whole number of
    microseconds are shortened to one, the last step of the interval takes the rest."""
    # (i) the stretched last step is itself snapped to whole microseconds, so a 0.4 us residual IS integrated on its own
    # (0.5 ms, 0.5 ms, 0.4 us: what the reference does) ...
    sizes = E.substep_sizes(1.0004e-3, 5e-4)
    assert len(sizes) == 3 and sizes[0] == 5e-4 and sizes[1] == pytest.approx(5e-4, rel=1e-12) and sizes[2] == pytest.approx(4e-7, rel=1e-6)
    # ... while a residual below STEPPER_MIN_TIMESTEP (round-off of the breakpoint arithmetic) is merged
    sizes = E.substep_sizes(1e-3 + 5e-11, 5e-4)
    assert len(sizes) == 2 and sizes[1] == pytest.approx(5e-4 + 5e-11, rel=1e-12)
    # steps of half a microsecond: a residual below 0.1 dt is merged (nothing is snapped below 1 us)
    sizes = E.substep_sizes(1.52e-6, 5e-7)
    assert len(sizes) == 3 and sizes[2] == pytest.approx(5.2e-7, rel=1e-9)
    # (ii) dtMax = 1/3 ms does not divide the 1 ms period: 333 us, 333 us, then the rest (334 us)
    sizes = E.substep_sizes(1e-3, 1e-3 / 3.0)
    assert [round(x * 1e6, 6) for x in sizes] == [333.0, 333.0, 334.0]
    assert abs(sum(sizes) - 1e-3) < 1e-18
    # dtMax = 0.3 ms: 0.3, 0.3, 0.3 and a last step of 0.1 ms
    sizes = E.substep_sizes(1e-3, 3e-4)
    assert [round(x * 1e6, 6) for x in sizes] == [300.0, 300.0, 300.0, 100.0]
    # sub-microsecond step sizes are left alone
    sizes = E.substep_sizes(2.5e-6, 1e-6)
    assert len(sizes) == 3 and sizes[0] == 1e-6 and sizes[2] == pytest.approx(5e-7)
    # the launch plan groups equal sub-steps, flags on the first / last launch of the interval
    o = _opts(dtMax=1e-3 / 3.0, controllerUpdatePeriod=1e-3, sensorsUpdatePeriod=1e-3, odeSolver="runge_kutta_4")
    launches, t_end, _ = E.plan_step(0.0, 0.0, 1e-3, o)
    assert [(round(dt * 1e6, 6), n, c, u) for dt, n, c, u in launches] == [(333.0, 2, True, False), (334.0, 1, False, True)]
    assert t_end == pytest.approx(1e-3)
    # a dividing dtMax is one launch per interval, whatever the round-off of the accumulated time
    o = _opts(dtMax=1e-3, controllerUpdatePeriod=5e-3, sensorsUpdatePeriod=5e-3)
    t, err = 0.0, 0.0
    for _ in range(2000):
        launches, t, err = E.plan_step(t, err, 5e-3, o)
        assert len(launches) == 1 and launches[0][1] == 5


def test_step_size_out_of_bounds():
    with pytest.raises(ValueError):
        E.plan_step(0.0, 0.0, 1e-8, _opts(dtMax=1e-3))


def test_engine_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    from jiminy_amd import load_builtin
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        E.BatchedEngine(load_builtin("cartpole"), 8)


def test_a_simulation_opens_with_the_reference_microsecond_step():
    """`Engine::start` resets the stepper state with dt = SIMULATION_MIN_TIMESTEP (engine.cc:1176) and the loop only
    settles on dtMax after its first try (:2220): the first interval is (1 us, rest)."""
    assert E.substep_sizes(1e-3, 1e-3, E.SIMULATION_MIN_TIMESTEP) == [1e-6, pytest.approx(9.99e-4, rel=1e-12)]
    sizes = E.substep_sizes(5e-3, 1e-3, 1e-6)       # 1 us, four dtMax steps, then what is left (0.999 ms)
    assert len(sizes) == 6 and sizes[0] == 1e-6 and sizes[1:5] == [1e-3] * 4 and sizes[5] == pytest.approx(9.99e-4, rel=1e-9)
    assert abs(sum(sizes) - 5e-3) < 1e-18
    assert E.substep_sizes(1e-3, 2e-2, 1e-6) == [1e-6, pytest.approx(9.99e-4, rel=1e-12)]     # dtMax above the period
    # the launch plan: the a(t+) refresh belongs to the microsecond step, the sensor update to the end of the interval
    o = _opts(dtMax=1e-3, controllerUpdatePeriod=5e-3, sensorsUpdatePeriod=5e-3, odeSolver="euler_explicit")
    launches, t_end, _ = E.plan_step(0.0, 0.0, 5e-3, o, dt_first=1e-6)
    assert [(round(dt * 1e9), n, c, s) for dt, n, c, s in launches] == [(1000, 1, True, False), (1000000, 4, False, False),
                                                                        (999000, 1, False, True)]
    assert t_end == pytest.approx(5e-3)
    # only the first interval of the call carries it
    launches, _, _ = E.plan_step(0.0, 0.0, 1e-2, o, dt_first=1e-6)
    assert [n for _, n, _, _ in launches] == [1, 4, 1, 5]
    # and the test-side statement of the reference's loop (tests/helpers.py) agrees with the engine's on random cases
    from tests.helpers import ReferenceFixedStepLoop
    rg = np.random.default_rng(3)
    for _ in range(2000):
        dt_max = float(rg.choice([1e-3, 5e-4, 1.2345e-3, 2e-2, 3.3e-6, 1e-6, 7.77e-4]))
        interval = float(rg.choice([1e-3, 5e-3, 4e-2, 1.05e-6, 1e-6, 3.14159e-3, 2e-6]))
        loop = ReferenceFixedStepLoop(dt_max)
        assert np.allclose(list(loop.sizes(interval)), E.substep_sizes(interval, dt_max, 1e-6), rtol=0, atol=1e-15)
        assert np.allclose(list(loop.sizes(interval)), E.substep_sizes(interval, dt_max), rtol=0, atol=1e-15)   # (settled)


def test_first_interval_with_euler_equals_the_hand_computed_two_step_result():
    """A point mass in free fall (a = -g exactly), explicit Euler, dt = 1 ms: the reference integrates the first
    interval as 1 us + 999 us, i.e. z = z0 + v0 dt1 + (v0 - g dt1) dt2 and v = v0 - g (dt1 + dt2) -- 9.8e-9 m below the
    single-step result.  The launches of `plan_step` run through the host emulation of the kernels, the oracle
    through the test-side loop; both must give the hand-computed numbers."""
    from tests.hostemu import emu
    from tests.helpers import ReferenceFixedStepLoop, alloc_soa, oracle_engine_step, oracle_batch
    from tests.robots import point_mass
    model = point_mass()
    g, z0, vz0, dt = 9.81, 5.0, 0.3, 1e-3
    o = _opts(dtMax=dt, controllerUpdatePeriod=dt, sensorsUpdatePeriod=dt, odeSolver="euler_explicit")

    def fresh():
        arr = alloc_soa(model, 1)
        arr["q"][:, 0] = [0.1, -0.2, z0, 0.0, 0.0, 0.0, 1.0]
        arr["v"][2, 0] = vz0
        return arr
    got, ref = fresh(), fresh()
    emu.run(model, got, "start")
    oracle_batch(model, ref, "start")
    launches, _, _ = E.plan_step(0.0, 0.0, dt, o, dt_first=E.SIMULATION_MIN_TIMESTEP)
    assert len(launches) == 2
    for h, n, changed, sens in launches:
        emu.run(model, got, "step", solver="euler_explicit", dt=h, n_substeps=n, command_changed=changed, update_sensors=sens)
    loop = ReferenceFixedStepLoop(dt)
    assert oracle_engine_step(model, ref, loop, dt, "euler_explicit", command_changed=True) == 2
    dt1, dt2 = 1e-6, dt - 1e-6
    z_hand = z0 + vz0 * dt1 + (vz0 - g * dt1) * dt2
    v_hand = vz0 - g * dt1 - g * dt2
    for arr in (got, ref):
        assert arr["q"][2, 0] == pytest.approx(z_hand, abs=1e-15) and arr["v"][2, 0] == pytest.approx(v_hand, abs=1e-15)
    assert abs(z_hand - (z0 + vz0 * dt)) > 9e-9          # the single-step result is measurably different
    # second interval: one step of dtMax
    launches, _, _ = E.plan_step(dt, 0.0, dt, o)
    assert [(n) for _, n, _, _ in launches] == [1]
    assert oracle_engine_step(model, ref, loop, dt, "euler_explicit", command_changed=True) == 1


def test_cross_variant_agreement_of_the_float64_rows_is_a_strict_second_opinion():
    """`_verified_library`'s last resort when float32 cannot vouch for the float64 emitted rows (a flexibility inertia of 1e-5
    next to 5 kg m^2): the rows of the separately compiled build variants must agree with each other.  Round-off passes, a
    wrong row in ONE variant does not, lanes that are NaN in any variant are left out, nothing left to compare is a failure."""
    import torch

    from tests import robots
    model = robots.pendulum()
    key = model.topology_hash()
    rng = np.random.default_rng(0)
    base = [{"a": torch.from_numpy(rng.standard_normal((1, 8))), "imu": torch.from_numpy(rng.standard_normal((6, 8)))}
            for _ in range(2)]
    ok = torch.ones(8, dtype=torch.bool)

    def put(variant, rows, lanes=ok):
        E._OUTPUT_ROWS_F64[(key, variant)] = (rows, lanes)

    try:
        for v in (0, 1, 2):
            put(v, [{k: x * (1.0 + 1e-13 * v) for k, x in r.items()} for r in base])
        assert E._float64_rows_agree_across_variants(model, [0, 1, 2]) < 1e-11
        bad = [{k: x.clone() for k, x in r.items()} for r in base]
        bad[1]["imu"][3, 5] += 0.5
        put(2, bad)
        assert E._float64_rows_agree_across_variants(model, [0, 1, 2]) > 1e-2
        lanes = ok.clone()
        lanes[5] = False                         # ... unless that lane is NaN there: it is not compared
        put(2, bad, lanes)
        assert E._float64_rows_agree_across_variants(model, [0, 1, 2]) < 1e-11
        put(2, bad, torch.zeros(8, dtype=torch.bool))
        assert E._float64_rows_agree_across_variants(model, [0, 1, 2]) == float("inf")
    finally:
        for v in (0, 1, 2):
            E._OUTPUT_ROWS_F64.pop((key, v), None)
