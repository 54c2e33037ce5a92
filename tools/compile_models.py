"""Compile the reference's robot descriptions into portable flat models.

Run in the build container (where /root/reference exists):
    python tools/compile_models.py
Outputs jiminy_amd/data/models/<name>.json, consumed by `jiminy_amd.load_builtin` on machines
that do not have the URDF files (the GPU box).  Only derived numeric arrays are stored.

Sources (reference tree):
  data/toys_models/double_pendulum/double_pendulum.urdf
  data/toys_models/cartpole/cartpole.urdf      + motor/encoders as gym_jiminy/envs/cartpole.py:108-133
  data/quadrupedal_robots/anymal/anymal.urdf   + anymal_hardware.toml
  data/bipedal_robots/atlas/atlas.urdf         + atlas_hardware.toml
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from jiminy_amd.model import add_motor, add_sensor, build_model_from_urdf, build_robot  # noqa: E402

REF = os.environ.get("JIMINY_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "jiminy_amd", "data", "models")


def main() -> None:
    os.makedirs(OUT, exist_ok=True)
    data = os.path.join(REF, "data")

    # the authored seven-joint fixed-base arm of the test-suite (tests/data/arm7.urdf: no reference file involved): the
    # secondary workload of bench.py on the one-robot-per-lane kernels
    tdata = os.path.join(ROOT, "tests", "data")
    m = build_robot(os.path.join(tdata, "arm7.urdf"), os.path.join(tdata, "arm7_hardware.toml"), has_freeflyer=False, name="arm7")
    m.save(os.path.join(OUT, "arm7.json"))
    print("arm7", m.joint_names, m.nq, m.nv)

    # double pendulum: motors on both joints like the reference example
    # (python/jiminy_py/examples/double_pendulum.py uses "PendulumJoint" / "SecondPendulumJoint")
    m = build_model_from_urdf(os.path.join(data, "toys_models/double_pendulum/double_pendulum.urdf"),
                              has_freeflyer=False, name="double_pendulum")
    for jn in m.joint_names[1:]:
        add_motor(m, jn, jn, enableVelocityLimit=False, enableEffortLimit=False)
        add_sensor(m, "EncoderSensor", jn, joint_name=jn)
    m.save(os.path.join(OUT, "double_pendulum.json"))
    print("double_pendulum", m.joint_names, m.nq, m.nv)

    # cartpole (gym_jiminy/envs/cartpole.py:108-133): one motor without velocity limit, 2 encoders
    m = build_model_from_urdf(os.path.join(data, "toys_models/cartpole/cartpole.urdf"),
                              has_freeflyer=False, name="cartpole")
    add_motor(m, "slider_to_cart", "slider_to_cart", enableVelocityLimit=False)
    add_sensor(m, "EncoderSensor", "slider", joint_name="slider_to_cart")
    add_sensor(m, "EncoderSensor", "pole", joint_name="cart_to_pole")
    m.save(os.path.join(OUT, "cartpole.json"))
    print("cartpole", m.joint_names, m.nq, m.nv)

    m = build_robot(os.path.join(data, "quadrupedal_robots/anymal/anymal.urdf"),
                    has_freeflyer=True, name="anymal")
    m.save(os.path.join(OUT, "anymal.json"))
    print("anymal", m.nq, m.nv, m.nmotors, m.ncontacts, m.topology_hash())

    m = build_robot(os.path.join(data, "bipedal_robots/atlas/atlas.urdf"),
                    has_freeflyer=True, name="atlas")
    m.save(os.path.join(OUT, "atlas.json"))
    print("atlas", m.nq, m.nv, m.nmotors, m.ncontacts, m.topology_hash())


if __name__ == "__main__":
    main()
