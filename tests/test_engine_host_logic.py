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
