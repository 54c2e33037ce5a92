"""Model compiler: URDF + hardware TOML -> flat arrays, conventions of the reference's model."""
import math
import os

import numpy as np
import pytest

from jiminy_amd import load_builtin
from jiminy_amd.model import (CompiledModel, JT_FREEFLYER, JT_PU, JT_PY, JT_RU, JT_RUBU, JT_RUBX,
                              JT_RY, JT_RZ, SE3, Inertia, add_motor, add_sensor, rpy_to_matrix)
from tests import robots


def test_joint_order_is_depth_first_alphabetical_by_joint_name():
    m = robots.tree_arm(True)
    assert m.joint_names == ["universe", "root_joint", "a_slide", "a_spin", "b_yaw", "c_skew",
                             "d_skew_slide", "d_skew_spin"]
    assert list(m.parents) == [0, 0, 1, 2, 1, 4, 1, 6]
    assert list(m.idx_q) == [0, 0, 7, 8, 10, 11, 12, 13]
    assert list(m.idx_v) == [0, 0, 6, 7, 8, 9, 10, 11]
    assert m.nq == 15 and m.nv == 12


def test_joint_type_classification():
    m = robots.tree_arm(False)
    t = {n: int(x) for n, x in zip(m.joint_names, m.jtypes)}
    assert t["a_slide"] == JT_PY          # exact +y axis -> aligned
    assert t["a_spin"] == JT_RUBX         # continuous, aligned
    assert t["b_yaw"] == JT_RZ
    assert t["c_skew"] == JT_RU           # (0.6, 0, 0.8) -> unaligned, already unit
    assert t["d_skew_slide"] == JT_PU
    assert t["d_skew_spin"] == JT_RUBU    # (0, 1, 1) normalised
    j = m.joint_index("d_skew_spin")
    assert np.allclose(m.axes[j], [0, 1 / math.sqrt(2), 1 / math.sqrt(2)])
    # -x is NOT axis aligned (ANYmal RF/RH/LH joints)
    a = load_builtin("anymal")
    assert int(a.jtypes[a.joint_index("RF_HFE")]) == JT_RU
    assert np.allclose(a.axes[a.joint_index("RF_HFE")], [-1, 0, 0])
    assert int(a.jtypes[1]) == JT_FREEFLYER


def test_fixed_joint_inertia_lumping_matches_hand_computation():
    m = robots.tree_arm(False)
    # root link chain pedestal -> trunk -> plate are all rigidly attached to the universe here;
    # with a free-flyer they are lumped into the root joint body
    mf = robots.tree_arm(True)
    Mt = SE3(np.eye(3), np.array([0, 0, 0.75]))
    trunk = Inertia(4.0, np.array([0.01, -0.02, 0.03]), None)
    R = rpy_to_matrix([0.1, 0.2, -0.3])
    I_trunk = R @ np.array([[0.08, 0.004, -0.003], [0.004, 0.09, 0.002], [-0.003, 0.002, 0.05]]) @ R.T
    trunk = Inertia(4.0, np.array([0.01, -0.02, 0.03]), I_trunk).transformed(Mt)
    Mp = Mt * SE3(rpy_to_matrix([0, 0.5, 0]), np.array([0.1, 0, 0.2]))
    Rp = rpy_to_matrix([0.3, 0, 0])
    I_plate = Rp @ np.array([[0.004, 0, 0.001], [0, 0.006, 0], [0.001, 0, 0.005]]) @ Rp.T
    plate = Inertia(0.7, np.array([0, 0.05, 0]), I_plate).transformed(Mp)
    tot = trunk.add(plate)
    assert mf.mass[1] == pytest.approx(4.7)
    assert np.allclose(mf.com[1], tot.com, atol=1e-14)
    assert np.allclose(mf.inertia[1], tot.I, atol=1e-14)
    # parallel axis check against the definition: I about the common COM
    def about(I, m_, c, pt):
        d = c - pt
        return I + m_ * (d @ d * np.eye(3) - np.outer(d, d))
    ref = about(trunk.I, trunk.mass, trunk.com, tot.com) + about(plate.I, plate.mass, plate.com, tot.com)
    assert np.allclose(tot.I, ref, atol=1e-14)
    # joint placements accumulate the fixed transforms
    j = m.joint_index("b_yaw")
    assert np.allclose(m.placement_p[j], [0, 0.15, 0.85])
    f = m.frame("tool")
    assert f.parent_joint == m.joint_index("c_skew") and np.allclose(f.p, [0, 0, 0.3])


def test_hardware_file_semantics():
    m = robots.tree_arm(True)
    assert m.contacts == ["fan", "plate", "tool"]           # registered in sorted order (robot.py:717)
    mot = {x.name: x for x in m.motors}
    assert mot["m_yaw"].reduction == 2.0
    assert mot["m_yaw"].effort_limit == pytest.approx(30 / 2.0)   # URDF effort / reduction
    assert mot["m_yaw"].velocity_limit == pytest.approx(20 * 2.0)
    assert mot["m_yaw"].armature == pytest.approx(0.01 * 4.0)     # armature * reduction^2, forced on
    assert not mot["m_slide"].enable_effort_limit
    iv = int(m.idx_v[m.joint_index("b_yaw")])
    assert m.rotor_inertia[iv] == pytest.approx(0.04)
    assert [s["name"] for s in m.sensors["ImuSensor"]] == ["imu_tool", "imu_new"]
    fr = m.frame("imu_new_frame")
    assert fr.parent_joint == m.joint_index("b_yaw")
    enc = {s["name"]: s for s in m.sensors["EncoderSensor"]}
    assert enc["enc_yaw"]["joint_side"] is False and enc["enc_yaw"]["reduction"] == 2.0
    assert enc["enc_spin"]["joint_side"] is True
    with pytest.raises(ValueError):
        add_motor(m, "m_yaw", "b_yaw")                     # duplicate motor name
    with pytest.raises(LookupError):
        add_motor(m, "other", "not_a_joint")
    with pytest.raises(ValueError):
        add_motor(m, "ff", "root_joint")                   # motors need a 1-dof joint
    with pytest.raises(ValueError):
        add_sensor(m, "ContactSensor", "x", frame_name="tool_nope")


def test_neutral_and_bounds():
    m = robots.tree_arm(True)
    q = m.neutral()
    assert q[6] == 1.0 and np.allclose(q[:6], 0)
    iq = int(m.idx_q[m.joint_index("a_spin")])
    assert q[iq] == 1.0 and q[iq + 1] == 0.0
    mask = m.bounded_position_mask()
    assert mask.sum() == 4 and not mask[:7].any()
    assert m.position_lower[int(m.idx_q[m.joint_index("b_yaw")])] == -2.5


def test_json_roundtrip_and_topology_hash():
    m = robots.tree_arm(True)
    m2 = CompiledModel.from_json(m.to_json())
    assert m2.topology_signature() == m.topology_signature()
    for k in ("placement_R", "placement_p", "mass", "com", "inertia", "rotor_inertia", "position_lower",
              "effort_limit"):
        assert np.array_equal(getattr(m, k), getattr(m2, k)), k
    assert m2.motors[0] == m.motors[0]
    # parameters do not enter the hash, structure does
    m2.mass[1] += 1.0
    assert m2.topology_hash() == m.topology_hash()
    assert robots.tree_arm(False).topology_hash() != m.topology_hash()


@pytest.mark.parametrize("name,nq,nv,nm,nc", [("double_pendulum", 2, 2, 2, 0), ("cartpole", 3, 2, 1, 0),
                                              ("anymal", 19, 18, 12, 4), ("atlas", 37, 36, 30, 32)])
def test_builtin_models(name, nq, nv, nm, nc):
    m = load_builtin(name)
    assert (m.nq, m.nv, m.nmotors, m.ncontacts) == (nq, nv, nm, nc)
    if name == "anymal":
        # SURVEY.md section 7 "hard parts": model order is LF, LH, RF, RH (alphabetical), the
        # hardware file lists motors as LF, RF, LH, RH; contacts are sorted by name
        assert m.joint_names[2:5] == ["LF_HAA", "LF_HFE", "LF_KFE"] and m.joint_names[5] == "LH_HAA"
        assert [x.name for x in m.motors][3] == "RF_HAA"
        assert m.contacts == ["LF_FOOT", "LH_FOOT", "RF_FOOT", "RH_FOOT"]
        assert np.allclose(m.rotor_inertia[6:], 0.1)
        assert m.mass.sum() == pytest.approx(52.13485)
    if name == "cartpole":
        assert int(m.jtypes[2]) == 10  # continuous about y -> [cos, sin]
        assert m.mass[2] == pytest.approx(0.1) and np.allclose(m.com[2], [0, 0, 1.0])


def test_builtin_models_match_the_reference_urdfs_when_present():
    ref = "/root/reference/data/quadrupedal_robots/anymal/anymal.urdf"
    if not os.path.exists(ref):
        pytest.skip("reference tree not available on this machine")
    from jiminy_amd.model import build_robot
    m = build_robot(ref, has_freeflyer=True, name="anymal")
    b = load_builtin("anymal")
    assert m.topology_signature() == b.topology_signature()
    assert np.array_equal(m.inertia, b.inertia) and np.array_equal(m.placement_R, b.placement_R)


# ---- pins that do not go through CompiledModel's own construction: the URDF XML walked directly --------------
def _urdf_composite(urdf_path):
    """Total mass, centre of mass and inertia about the root-link origin of a URDF at its zero
    configuration, from the XML alone: own element walk, own roll-pitch-yaw convention (URDF: fixed-axis
    X-Y-Z = Rz Ry Rx), nothing imported from jiminy_amd.model."""
    import xml.etree.ElementTree as ET
    root = ET.parse(urdf_path).getroot()

    def floats(s, n):
        return np.array([float(x) for x in s.split()]) if s else np.zeros(n)

    def rot(rpy):
        r, p, y = rpy
        Rx = np.array([[1, 0, 0], [0, math.cos(r), -math.sin(r)], [0, math.sin(r), math.cos(r)]])
        Ry = np.array([[math.cos(p), 0, math.sin(p)], [0, 1, 0], [-math.sin(p), 0, math.cos(p)]])
        Rz = np.array([[math.cos(y), -math.sin(y), 0], [math.sin(y), math.cos(y), 0], [0, 0, 1]])
        return Rz @ Ry @ Rx

    def origin(e):
        o = e.find("origin") if e is not None else None
        if o is None:
            return np.eye(3), np.zeros(3)
        return rot(floats(o.get("rpy"), 3)), floats(o.get("xyz"), 3)
    links = {l.get("name"): l for l in root.findall("link")}
    children, child_names = {}, set()
    for j in root.findall("joint"):
        children.setdefault(j.find("parent").get("link"), []).append(j)
        child_names.add(j.find("child").get("link"))
    (root_link,) = [n for n in links if n not in child_names]
    tot_m, tot_mc, tot_I = 0.0, np.zeros(3), np.zeros((3, 3))
    stack = [(root_link, np.eye(3), np.zeros(3))]
    while stack:
        name, R, p = stack.pop()
        inert = links[name].find("inertial")
        if inert is not None and inert.find("mass") is not None:
            m = float(inert.find("mass").get("value"))
            Ri, pi = origin(inert)
            i = inert.find("inertia")
            I = np.zeros((3, 3))
            if i is not None:
                ixx, ixy, ixz, iyy, iyz, izz = (float(i.get(k, 0.0)) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz"))
                I = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
            c = p + R @ pi                      # COM in the root-link frame
            Iw = (R @ Ri) @ I @ (R @ Ri).T      # rotational inertia about the COM, root-link axes
            tot_m += m
            tot_mc += m * c
            tot_I += Iw + m * (c @ c * np.eye(3) - np.outer(c, c))   # parallel axis to the origin
        for j in children.get(name, []):
            Rj, pj = origin(j)
            stack.append((j.find("child").get("link"), R @ Rj, p + R @ pj))
    return tot_m, tot_mc / tot_m, tot_I


def _model_composite(m):
    """The same three quantities from the compiled model at its neutral configuration, root joint at the
    origin (free-flyer) -- plain sums over the joints' bodies."""
    from jiminy_amd.synthetic import joint_world_placements
    Rs, ps = joint_world_placements(m, m.neutral())
    tot_m, tot_mc, tot_I = 0.0, np.zeros(3), np.zeros((3, 3))
    for j in range(1, m.njoints):
        R, p = Rs[j][0], ps[j][0]
        c = p + R @ m.com[j]
        tot_m += m.mass[j]
        tot_mc += m.mass[j] * c
        tot_I += R @ m.inertia[j] @ R.T + m.mass[j] * (c @ c * np.eye(3) - np.outer(c, c))
    return tot_m, tot_mc / tot_m, tot_I


# frozen from the reference's URDF files by `_urdf_composite` (total mass [kg], COM [m], diagonal of the
# composite inertia about the root-link origin [kg m^2]); the XML itself is re-walked when the reference
# tree is present
URDF_COMPOSITES = {
    "anymal": ("quadrupedal_robots/anymal/anymal.urdf", None),
    "atlas": ("bipedal_robots/atlas/atlas.urdf", None),
}
FROZEN_COMPOSITES = {
    "anymal": (52.134849999999986, [-0.009001324210294755, -9.012968292868901e-05, -0.07019512926579306],
               [2.396821450925352, 6.452496555268907, 6.267118106699941]),
    "atlas": (174.05030000000002, [0.00012255752916805809, 0.0010554102245931125, 0.28773675163260287],
              [56.410983630038714, 46.106879692483204, 13.53299071151775]),
}


@pytest.mark.parametrize("name", ["anymal", "atlas"])
def test_composite_inertia_matches_the_urdf_xml(name):
    m = load_builtin(name)
    got = _model_composite(m)
    frozen = FROZEN_COMPOSITES[name]
    assert got[0] == pytest.approx(frozen[0], rel=1e-12)
    assert np.allclose(got[1], frozen[1], rtol=0, atol=1e-12)
    assert np.allclose(np.diag(got[2]), frozen[2], rtol=1e-12)
    urdf = os.path.join(os.environ.get("JIMINY_REFERENCE", "/root/reference"), "data", URDF_COMPOSITES[name][0])
    if not os.path.exists(urdf):
        pytest.skip("reference URDF not available on this machine: checked against the frozen values only")
    want = _urdf_composite(urdf)
    assert got[0] == pytest.approx(want[0], rel=1e-13)
    assert np.allclose(got[1], want[1], rtol=0, atol=1e-13)
    assert np.allclose(got[2], want[2], rtol=1e-12, atol=1e-12)


def test_branch_parallel_decomposition_of_the_shipped_and_the_authored_robots():
    """`codegen.quad_structure`: the four longest leaf chains are the limbs, everything else the trunk tree; trees with
    two or three leaf chains get empty limbs (round 4); fixed-base arms, single chains and robots whose contact points or
    IMUs sit where the 4-lane kernels cannot serve them fall back to the one-robot-per-lane kernels (None)."""
    from jiminy_amd import codegen
    from tests import robots
    q = codegen.quad_structure(load_builtin("anymal"))
    assert q["n"] == 3 and q["limb_len"] == [3, 3, 3, 3] and q["limb_attach"] == [0, 0, 0, 0] and len(q["trunk"]) == 1
    q = codegen.quad_structure(load_builtin("atlas"))
    assert q["n"] == 7 and sorted(q["limb_len"]) == [6, 6, 7, 7] and len(q["trunk"]) == 5 and sum(q["limb_ncontact"]) == 32
    q = codegen.quad_structure(robots.crane_walker())
    assert sorted(q["limb_len"]) == [2, 2, 3, 3] and sorted(q["limb_attach"]) == [0, 0, 2, 2]
    q = codegen.quad_structure(robots.biped(False))
    assert q["limb_len"] == [3, 3, 0, 0] and q["limb_attach"] == [0, 0, 0, 0] and q["limb_ncontact"] == [2, 2, 0, 0]
    assert all(j == -1 for row in q["limb_joint"][2:] for j in row) if "limb_joint" in q else True
    q = codegen.quad_structure(robots.biped(True))
    assert sorted(q["limb_len"]) == [0, 1, 3, 3] and q["limb_len"][3] == 0
    for m in (load_builtin("cartpole"), load_builtin("double_pendulum"), robots.tree_arm(False), robots.tree_arm(True),
              robots.point_mass()):
        assert codegen.quad_structure(m) is None


def test_constraint_names_are_unique_across_frame_and_joint_constraints():
    """`Model::addConstraint` refuses any duplicate name (model.cc:884-890), whatever the constraint types."""
    from jiminy_amd.model import add_frame_constraint, add_joint_constraint
    from tests import robots
    m = robots.tree_arm(False)
    joint = m.joint_names[1]
    add_joint_constraint(m, "hold", joint)
    with pytest.raises(ValueError, match="already declared"):
        add_frame_constraint(m, "hold", next(iter(m.frames)) if isinstance(m.frames, dict) else m.frames[0].name)
