"""Small robots authored for the test-suite (URDFs under tests/data/, written for this repo)."""
from __future__ import annotations

import os
from typing import List

from jiminy_amd.model import (CompiledModel, add_contact_points, add_motor, add_sensor,
                              build_model_from_urdf, build_robot)

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def pendulum(armature: float = 0.0) -> CompiledModel:
    m = build_model_from_urdf(os.path.join(DATA, "pendulum.urdf"), name="pendulum")
    add_motor(m, "pivot", "pivot", enableVelocityLimit=False, enableEffortLimit=False,
              enableArmature=armature > 0, armature=armature)
    add_sensor(m, "EncoderSensor", "pivot", joint_name="pivot")
    add_sensor(m, "ImuSensor", "tip", frame_name="tip")
    return m


def double_pendulum() -> CompiledModel:
    m = build_model_from_urdf(os.path.join(DATA, "double_pendulum.urdf"), name="double_pendulum_test")
    for j in ("shoulder", "elbow"):
        add_motor(m, j, j, enableVelocityLimit=False, enableEffortLimit=False)
    return m


def point_mass() -> CompiledModel:
    m = build_model_from_urdf(os.path.join(DATA, "point_mass.urdf"), has_freeflyer=True,
                              name="point_mass")
    add_contact_points(m, ["body"])
    add_sensor(m, "ContactSensor", "body", frame_name="body")
    add_sensor(m, "ForceSensor", "sole", frame_name="sole")
    add_sensor(m, "ImuSensor", "sole", frame_name="sole")
    return m


def two_masses() -> CompiledModel:
    m = build_model_from_urdf(os.path.join(DATA, "two_masses.urdf"), name="two_masses")
    for j in ("slide_a", "slide_b"):
        add_motor(m, j, j, enableVelocityLimit=False, enableEffortLimit=False)
        add_sensor(m, "EncoderSensor", j, joint_name=j)
    return m


def tree_arm(has_freeflyer: bool) -> CompiledModel:
    return build_robot(os.path.join(DATA, "tree_arm.urdf"),
                       os.path.join(DATA, "tree_arm_hardware.toml"),
                       has_freeflyer=has_freeflyer,
                       name="tree_arm_ff" if has_freeflyer else "tree_arm")


def crane_walker() -> CompiledModel:
    """Free-flyer whose tree decomposes (codegen.quad_structure) into a trunk tree with a prismatic
    and an unaligned revolute joint, two 3-joint arms on the turret and two 2-joint legs on the
    base: uneven (padded) limbs attached at two different trunk joints, 1 or 2 contact points per
    limb, contact / force / IMU sensors, friction-enabled motors."""
    return build_robot(os.path.join(DATA, "crane_walker.urdf"),
                       os.path.join(DATA, "crane_walker_hardware.toml"),
                       has_freeflyer=True, name="crane_walker")


def biped(torso: bool = False) -> CompiledModel:
    """Free-flying biped without arms: two 3-joint legs with toe / heel contact points (two leaf chains), optionally a
    1-joint torso on a waist joint (three uneven leaf chains). `codegen.quad_structure` completes the decomposition with
    empty limbs, so the branch-parallel kernels serve it."""
    name = "biped_torso" if torso else "biped"
    return build_robot(os.path.join(DATA, name + ".urdf"), os.path.join(DATA, name + "_hardware.toml"),
                       has_freeflyer=True, name=name)


def hanging_pendulum() -> CompiledModel:
    """The reference's `simple_pendulum` fixture restated (tests/data/hanging_pendulum.urdf): bob welded 1 m below the
    pivot, motor without limits or armature (unit_py/utilities.py:18-59 `load_urdf_default`), IMU on the bob."""
    m = build_model_from_urdf(os.path.join(DATA, "hanging_pendulum.urdf"), name="hanging_pendulum")
    add_motor(m, "pivot", "pivot", enableVelocityLimit=False, enableEffortLimit=False, enableArmature=False)
    add_sensor(m, "ImuSensor", "bob", frame_name="bob")
    m.position_lower[:] = -1e9   # (the reference tests run it without position limits)
    m.position_upper[:] = 1e9
    return m


def foot_pendulum() -> CompiledModel:
    """The reference's `foot_pendulum` fixture restated (tests/data/foot_pendulum.urdf + hardware file): free-flying
    inverted pendulum on a square foot whose 8 box vertices are contact points."""
    return build_robot(os.path.join(DATA, "foot_pendulum.urdf"), os.path.join(DATA, "foot_pendulum_hardware.toml"),
                       has_freeflyer=True, name="foot_pendulum")


def all_test_models() -> List[CompiledModel]:
    return [pendulum(), double_pendulum(), point_mass(), two_masses(), tree_arm(False),
            tree_arm(True), crane_walker(), biped(False), biped(True), arm7(),
            pendulum_flexible(), tree_arm_flexible(False), tree_arm_flexible(True), pendulum_backlash()]


# ---- robots of the reference's user-FrameConstraint tests, restated (the constraint frames are part of the topology)
def two_masses_fixed_second() -> CompiledModel:
    """unit_py/test_double_spring_mass.py:225-252: `FrameConstraint("SecondMass")`, all six dofs, on the second mass."""
    from jiminy_amd.model import add_frame_constraint
    m = two_masses()
    m.name = "two_masses_fix"
    add_frame_constraint(m, "fixMass", "mass_b")
    return m


def sphere_fixed_frame() -> CompiledModel:
    """unit_py/test_simple_mass.py:335-377: free-flying body, `FrameConstraint("MassBody", [True] * 6)`."""
    from jiminy_amd.model import add_frame_constraint
    m = build_model_from_urdf(os.path.join(DATA, "point_mass.urdf"), has_freeflyer=True, name="sphere_fix")
    add_frame_constraint(m, "MassBody", "body")
    return m


def pendulum_ff_fixed_world(armature: float = 0.1) -> CompiledModel:
    """unit_py/test_simple_pendulum.py:752-813: pendulum on a free-flyer whose root body is held by
    `FrameConstraint("world")`, rotor inertia on the pendulum joint."""
    from jiminy_amd.model import add_frame_constraint
    m = build_model_from_urdf(os.path.join(DATA, "pendulum_base.urdf"), has_freeflyer=True, name="pendulum_ff_fix")
    add_motor(m, "pivot", "pivot", enableVelocityLimit=False, enableEffortLimit=False, enableArmature=armature > 0,
              armature=armature)
    add_frame_constraint(m, "world", "world_link")
    return m


def tree_arm_own_locks(has_freeflyer: bool) -> CompiledModel:
    """`tree_arm` whose `b_yaw` (revolute) and `d_skew_slide` (prismatic, unaligned) joints are declared for user
    JointConstraints on rows of their own (`add_joint_constraint`), plus a FrameConstraint on the position of the tool:
    user rows next to bounds and contact points in one solve."""
    from jiminy_amd.model import add_frame_constraint, add_joint_constraint
    m = tree_arm(has_freeflyer)
    m.name = "tree_arm_ff_locks" if has_freeflyer else "tree_arm_locks"
    add_joint_constraint(m, "hold_yaw", "b_yaw")
    add_joint_constraint(m, "hold_slide", "d_skew_slide")
    add_frame_constraint(m, "pin_tool", "tool", (True, True, True, False, False, False))
    return m


def rolling_ball() -> CompiledModel:
    """Free body (2 kg, inertia 0.5 kg.m^2) that a `SphereConstraint` of radius 0.5 m makes roll on the plane z = z_start - r."""
    from jiminy_amd.model import add_sphere_constraint
    m = build_model_from_urdf(os.path.join(DATA, "point_mass.urdf"), has_freeflyer=True, name="rolling_ball")
    add_sphere_constraint(m, "roll", "body", 0.5)
    return m


def rolling_wheel() -> CompiledModel:
    """The same body as a thin wheel about its y axis (`WheelConstraint`, radius 0.5 m)."""
    from jiminy_amd.model import add_wheel_constraint
    m = build_model_from_urdf(os.path.join(DATA, "point_mass.urdf"), has_freeflyer=True, name="rolling_wheel")
    add_wheel_constraint(m, "roll", "body", 0.5, (0.0, 0.0, 1.0), (0.0, 1.0, 0.0))
    return m


def tethered_mass() -> CompiledModel:
    """Free body held at a fixed distance from a world-fixed anchor (`DistanceConstraint` with one frame on the universe)."""
    import numpy as np

    from jiminy_amd.model import Frame, add_distance_constraint
    m = build_model_from_urdf(os.path.join(DATA, "point_mass.urdf"), has_freeflyer=True, name="tethered_mass")
    m.frames["anchor"] = Frame("anchor", 0, np.eye(3), np.array([0.0, 0.0, 1.0]), "op")
    add_distance_constraint(m, "tether", "body", "anchor")
    return m


def two_masses_rod() -> CompiledModel:
    """The two sliding masses joined by a rigid rod (`DistanceConstraint` between two moving frames)."""
    from jiminy_amd.model import add_distance_constraint
    m = two_masses()
    m.name = "two_masses_rod"
    add_distance_constraint(m, "rod", "mass_b", "mass_a")
    return m


def anymal_held() -> CompiledModel:
    """BASELINE's ANYmal with user constraints declared: its base may be held in the world (`FrameConstraint("base")`) and a
    foot tied to the base by a rod (`DistanceConstraint`).  A robot that declares user constraint frames runs on the
    one-robot-per-lane kernels."""
    from jiminy_amd import load_builtin
    from jiminy_amd.model import add_distance_constraint, add_frame_constraint
    m = load_builtin("anymal")
    m.name = "anymal_held"
    add_frame_constraint(m, "hold_base", "base")
    add_distance_constraint(m, "rod", m.contacts[0], "base")
    return m


def frame_constraint_models() -> List[CompiledModel]:
    return [two_masses_fixed_second(), sphere_fixed_frame(), pendulum_ff_fixed_world(), tree_arm_own_locks(False),
            tree_arm_own_locks(True), rolling_ball(), rolling_wheel(), tethered_mass(), two_masses_rod(), anymal_held()]


def arm7() -> CompiledModel:
    """Seven-joint fixed-base arm: a serial chain, i.e. the kind of robot the branch-parallel kernels
    cannot take (no leaf chains on a free-flyer) and the one-robot-per-lane kernels get."""
    return build_robot(os.path.join(DATA, "arm7.urdf"), os.path.join(DATA, "arm7_hardware.toml"),
                       has_freeflyer=False, name="arm7")


def pendulum_flexible(k: float = 20.0, nu: float = 0.1, armature: float = 0.1, flex_inertia: float = 1e-5) -> CompiledModel:
    """unit_py/test_simple_pendulum.py:662-691: the pendulum with a flexibility (a spherical joint with a spring-damper on its
    rotation) in front of its joint and a rotor inertia on its motor -- a series-elastic actuator."""
    import numpy as np
    m = build_model_from_urdf(os.path.join(DATA, "pendulum.urdf"), name="pendulum_flexible",
                              flexibility=[{"frameName": "pivot", "stiffness": k * np.ones(3), "damping": nu * np.ones(3),
                                            "inertia": flex_inertia * np.ones(3)}])
    add_motor(m, "pivot", "pivot", enableVelocityLimit=False, enableEffortLimit=False,
              enableArmature=armature > 0, armature=armature)
    add_sensor(m, "EncoderSensor", "pivot", joint_name="pivot")
    add_sensor(m, "ImuSensor", "tip", frame_name="tip")
    return m


def tree_arm_flexible(has_freeflyer: bool = False) -> CompiledModel:
    """`tree_arm` with two flexibilities: in place of the fixed joint that carries the plate (the plate then hangs on a
    spherical joint of its own) and in front of the unaligned revolute joint `c_skew`."""
    import numpy as np
    flex = [{"frameName": "z_fixed_plate", "stiffness": np.array([60.0, 80.0, 50.0]), "damping": np.array([0.4, 0.5, 0.3]),
             "inertia": np.array([2e-3, 1e-3, 3e-3])},
            {"frameName": "c_skew", "stiffness": np.array([150.0, 120.0, 90.0]), "damping": np.array([0.8, 0.6, 0.7]),
             "inertia": np.array([1e-3, 2e-3, 1e-3])}]
    return build_robot(os.path.join(DATA, "tree_arm.urdf"), os.path.join(DATA, "tree_arm_hardware.toml"),
                       has_freeflyer=has_freeflyer, name="tree_arm_flex_ff" if has_freeflyer else "tree_arm_flex",
                       flexibility=flex)


def pendulum_backlash(backlash: float = 2.2, armature: float = 1.0) -> CompiledModel:
    """unit_py/test_simple_pendulum.py:269-285: the pendulum with a rotor inertia on its motor and a backlash between the
    motor and the mass (`pivotBacklash`, bounded to +- backlash / 2)."""
    m = build_model_from_urdf(os.path.join(DATA, "pendulum.urdf"), name="pendulum_backlash", backlash={"pivot": backlash})
    add_motor(m, "pivot", "pivot", enableVelocityLimit=False, enableEffortLimit=False,
              enableArmature=armature > 0, armature=armature)
    add_sensor(m, "EncoderSensor", "pivot", joint_name="pivot")
    add_sensor(m, "ImuSensor", "tip", frame_name="tip")
    return m
