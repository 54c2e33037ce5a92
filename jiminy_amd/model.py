"""Model compiler: URDF (+ Jiminy hardware TOML) -> flat, topologically ordered arrays.

This is the host-side "compile the robot once" step of the batched engine.  It
reproduces the conventions of the model the reference engine integrates
(`Robot::pinocchioModel_`, built by `pinocchio::urdf::buildModel` through
reference core/src/utilities/pinocchio.cc:887-933 and completed by
python/jiminy_py/src/jiminy_py/robot.py:518-860 for motors / sensors / contact
points):

* joints are numbered depth-first from the root link, children visited in
  alphabetical order of the *joint* name (urdfdom stores joints in a name-keyed
  map), joint 0 is the universe and, with a free-flyer, joint 1 is `root_joint`
  (reference core/src/robot/model.cc:336-342);
* fixed joints do not create a model joint: the child link inertia is lumped
  into the parent joint body and a frame is recorded with the accumulated
  placement;
* an axis exactly equal to +x/+y/+z gives an axis-aligned joint, anything else
  (including -x) an "unaligned" joint; URDF `continuous` joints are unbounded
  revolute joints whose configuration is `[cos(theta), sin(theta)]`
  (reference core/include/jiminy/core/fwd.h:84-96);
* spatial quantities are `[linear; angular]`, quaternions `xyzw`;
* motor armature enters the dynamics as `rotorInertia[idx_v] += armature *
  reduction**2` (reference core/src/hardware/abstract_motor.cc:337-344,
  core/src/robot/robot.cc:240-247).

The result (`CompiledModel`) is plain numpy + python containers; `to_json` /
`from_json` make it portable so that machines without the URDF (the GPU box)
can still instantiate the engine.
"""
from __future__ import annotations

import hashlib
import json
import math
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

try:  # python >= 3.11
    import tomllib as _toml
except ModuleNotFoundError:  # pragma: no cover
    import tomli as _toml

# Joint type codes (shared with include/jiminy_hip.h and oracle/oracle.cpp)
JT_NONE = 0
JT_RX, JT_RY, JT_RZ = 1, 2, 3
JT_RU = 4
JT_PX, JT_PY, JT_PZ = 5, 6, 7
JT_PU = 8
JT_RUBX, JT_RUBY, JT_RUBZ = 9, 10, 11
JT_RUBU = 12
JT_FREEFLYER = 13
JT_SPHERICAL = 14     # flexibility joints (unit quaternion x y z w, angular velocity in the joint frame)

JT_NQ = {JT_NONE: 0, JT_RX: 1, JT_RY: 1, JT_RZ: 1, JT_RU: 1,
         JT_PX: 1, JT_PY: 1, JT_PZ: 1, JT_PU: 1,
         JT_RUBX: 2, JT_RUBY: 2, JT_RUBZ: 2, JT_RUBU: 2, JT_FREEFLYER: 7, JT_SPHERICAL: 4}
JT_NV = {JT_NONE: 0, JT_RX: 1, JT_RY: 1, JT_RZ: 1, JT_RU: 1,
         JT_PX: 1, JT_PY: 1, JT_PZ: 1, JT_PU: 1,
         JT_RUBX: 1, JT_RUBY: 1, JT_RUBZ: 1, JT_RUBU: 1, JT_FREEFLYER: 6, JT_SPHERICAL: 3}
JT_NAME = {JT_NONE: "universe", JT_RX: "RX", JT_RY: "RY", JT_RZ: "RZ",
           JT_RU: "RU", JT_PX: "PX", JT_PY: "PY", JT_PZ: "PZ", JT_PU: "PU",
           JT_RUBX: "RUBX", JT_RUBY: "RUBY", JT_RUBZ: "RUBZ",
           JT_RUBU: "RUBU", JT_FREEFLYER: "FF", JT_SPHERICAL: "S"}
BACKLASH_JOINT_SUFFIX = "Backlash"      # `<name>Backlash`: the backlash joint behind the motorised joint `<name>`
FLEXIBLE_JOINT_SUFFIX = "Flexibility"   # core/include/jiminy/core/robot/model.h (name of a flexibility joint inserted
                                        # in front of the mechanical joint `<name>`: `<name>Flexibility`)

# Sensor type names follow the reference (core/src/hardware/basic_sensors.cc)
SENSOR_TYPES = ("ImuSensor", "ContactSensor", "ForceSensor",
                "EncoderSensor", "EffortSensor")


# ----------------------------------------------------------------------------
# small SE3 / inertia helpers (numpy, float64)
# ----------------------------------------------------------------------------

def rpy_to_matrix(rpy: Sequence[float]) -> np.ndarray:
    """URDF fixed-axis roll/pitch/yaw -> rotation matrix Rz(y) Ry(p) Rx(r)."""
    r, p, y = (float(x) for x in rpy)
    cr, sr = math.cos(r), math.sin(r)
    cp, sp = math.cos(p), math.sin(p)
    cy, sy = math.cos(y), math.sin(y)
    return np.array([
        [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
        [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
        [-sp, cp * sr, cp * cr]], dtype=np.float64)


def skew(v: np.ndarray) -> np.ndarray:
    return np.array([[0.0, -v[2], v[1]],
                     [v[2], 0.0, -v[0]],
                     [-v[1], v[0], 0.0]])


class SE3:
    """Rigid placement (R, p): maps coordinates of the child frame to the parent."""
    __slots__ = ("R", "p")

    def __init__(self, R: Optional[np.ndarray] = None,
                 p: Optional[np.ndarray] = None) -> None:
        self.R = np.eye(3) if R is None else np.asarray(R, dtype=np.float64)
        self.p = np.zeros(3) if p is None else np.asarray(p, dtype=np.float64)

    def __mul__(self, other: "SE3") -> "SE3":
        return SE3(self.R @ other.R, self.p + self.R @ other.p)

    def inverse(self) -> "SE3":
        return SE3(self.R.T, -self.R.T @ self.p)

    def copy(self) -> "SE3":
        return SE3(self.R.copy(), self.p.copy())


@dataclass
class Inertia:
    """Spatial inertia as (mass, centre of mass, rotational inertia about the COM)."""
    mass: float = 0.0
    com: np.ndarray = field(default_factory=lambda: np.zeros(3))
    I: np.ndarray = field(default_factory=lambda: np.zeros((3, 3)))

    def transformed(self, M: SE3) -> "Inertia":
        return Inertia(self.mass, M.R @ self.com + M.p, M.R @ self.I @ M.R.T)

    def add(self, other: "Inertia") -> "Inertia":
        """Parallel-axis sum, same formula as the upstream `Inertia::operator+=`."""
        eps = np.finfo(np.float64).eps
        mab = self.mass + other.mass
        mab_inv = 1.0 / max(mab, eps)
        AB = self.com - other.com
        S = skew(AB)
        I = self.I + other.I - (self.mass * other.mass * mab_inv) * (S @ S)
        com = self.com * (self.mass * mab_inv) + (other.mass * mab_inv) * other.com
        return Inertia(mab, com, I)


# ----------------------------------------------------------------------------
# URDF parsing
# ----------------------------------------------------------------------------

def _floats(text: Optional[str], n: int, default: float = 0.0) -> np.ndarray:
    if text is None:
        return np.full(n, default, dtype=np.float64)
    vals = [float(x) for x in text.split()]
    if len(vals) != n:
        raise ValueError(f"expected {n} floats, got '{text}'")
    return np.array(vals, dtype=np.float64)


def _origin(elem: Optional[ET.Element]) -> SE3:
    if elem is None:
        return SE3()
    o = elem.find("origin")
    if o is None:
        return SE3()
    return SE3(rpy_to_matrix(_floats(o.get("rpy"), 3)), _floats(o.get("xyz"), 3))


@dataclass
class _UrdfLink:
    name: str
    inertia: Inertia
    collision_boxes: List[Tuple[np.ndarray, SE3]]  # (size, placement in link)


@dataclass
class _UrdfJoint:
    name: str
    jtype: str
    parent: str
    child: str
    origin: SE3
    axis: np.ndarray
    lower: float
    upper: float
    effort: float
    velocity: float


def _parse_urdf(urdf_path: str) -> Tuple[str, Dict[str, _UrdfLink], Dict[str, _UrdfJoint]]:
    root = ET.parse(urdf_path).getroot()
    if root.tag != "robot":
        raise ValueError("not a URDF file: missing <robot> root element")
    links: Dict[str, _UrdfLink] = {}
    for le in root.findall("link"):
        name = le.get("name")
        inert = Inertia()
        ie = le.find("inertial")
        if ie is not None:
            M = _origin(ie)
            mass = float(ie.find("mass").get("value")) if ie.find("mass") is not None else 0.0
            it = ie.find("inertia")
            I = np.zeros((3, 3))
            if it is not None:
                ixx, ixy, ixz = (float(it.get(k, 0.0)) for k in ("ixx", "ixy", "ixz"))
                iyy, iyz, izz = (float(it.get(k, 0.0)) for k in ("iyy", "iyz", "izz"))
                I = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
            inert = Inertia(mass, M.p.copy(), M.R @ I @ M.R.T)
        boxes = []
        for ce in le.findall("collision"):
            ge = ce.find("geometry")
            if ge is not None and ge.find("box") is not None:
                boxes.append((_floats(ge.find("box").get("size"), 3), _origin(ce)))
        links[name] = _UrdfLink(name, inert, boxes)
    joints: Dict[str, _UrdfJoint] = {}
    for je in root.findall("joint"):
        name = je.get("name")
        jtype = je.get("type")
        ax = je.find("axis")
        axis = _floats(ax.get("xyz"), 3) if ax is not None else np.array([1.0, 0.0, 0.0])
        lim = je.find("limit")
        lower = upper = 0.0
        effort = velocity = math.inf
        if lim is not None:
            lower = float(lim.get("lower", 0.0))
            upper = float(lim.get("upper", 0.0))
            effort = float(lim.get("effort", math.inf))
            velocity = float(lim.get("velocity", math.inf))
        if je.find("mimic") is not None:
            raise NotImplementedError(f"mimic joint '{name}' is not supported")
        joints[name] = _UrdfJoint(
            name, jtype, je.find("parent").get("link"), je.find("child").get("link"),
            _origin(je), axis, lower, upper, effort, velocity)
    return root.get("name", ""), links, joints


def _classify_joint(uj: _UrdfJoint) -> Tuple[int, np.ndarray]:
    """URDF joint -> (type code, unit axis); exact-axis test as the upstream parser."""
    axis = uj.axis
    aligned = None
    for k in range(3):
        e = np.zeros(3)
        e[k] = 1.0
        if np.array_equal(axis, e):
            aligned = k
    if uj.jtype == "revolute":
        base, unaligned = JT_RX, JT_RU
    elif uj.jtype == "continuous":
        base, unaligned = JT_RUBX, JT_RUBU
    elif uj.jtype == "prismatic":
        base, unaligned = JT_PX, JT_PU
    else:
        raise NotImplementedError(f"URDF joint type '{uj.jtype}' is not supported")
    if aligned is not None:
        e = np.zeros(3)
        e[aligned] = 1.0
        return base + aligned, e
    n = float(np.linalg.norm(axis))
    if n == 0.0:
        raise ValueError(f"joint '{uj.name}' has a null axis")
    return unaligned, axis / n


# ----------------------------------------------------------------------------
# Compiled model
# ----------------------------------------------------------------------------

@dataclass
class Frame:
    name: str
    parent_joint: int
    R: np.ndarray
    p: np.ndarray
    kind: str = "body"  # body | fixed_joint | joint | op


@dataclass
class Motor:
    name: str
    joint_name: str
    joint: int
    idx_q: int
    idx_v: int
    reduction: float = 1.0
    effort_limit: float = math.inf      # motor side
    velocity_limit: float = math.inf    # motor side
    enable_effort_limit: bool = True
    enable_velocity_limit: bool = False
    velocity_effort_inv_slope: float = 0.0
    armature: float = 0.0               # joint side (already x reduction^2)
    enable_friction: bool = False
    friction_viscous_pos: float = 0.0
    friction_viscous_neg: float = 0.0
    friction_dry_pos: float = 0.0
    friction_dry_neg: float = 0.0
    friction_dry_slope: float = 0.0


@dataclass
class CompiledModel:
    name: str
    has_freeflyer: bool
    joint_names: List[str]            # index 0 = "universe"
    parents: np.ndarray               # (njoints,) int32, parents[0] = 0
    jtypes: np.ndarray                # (njoints,) int32
    axes: np.ndarray                  # (njoints, 3)
    idx_q: np.ndarray                 # (njoints,) int32
    idx_v: np.ndarray                 # (njoints,) int32
    placement_R: np.ndarray           # (njoints, 3, 3) joint placement wrt parent joint
    placement_p: np.ndarray           # (njoints, 3)
    mass: np.ndarray                  # (njoints,)
    com: np.ndarray                   # (njoints, 3)
    inertia: np.ndarray               # (njoints, 3, 3) about COM
    rotor_inertia: np.ndarray         # (nv,)
    position_lower: np.ndarray        # (nq,)
    position_upper: np.ndarray        # (nq,)
    effort_limit: np.ndarray          # (nv,) URDF values (theoretical model)
    velocity_limit: np.ndarray        # (nv,)
    gravity: np.ndarray               # (6,) world frame [linear; angular]
    frames: Dict[str, Frame]
    motors: List[Motor]
    contacts: List[str]               # contact frame names, in engine order
    sensors: Dict[str, List[Dict[str, Any]]]
    # frames a user `FrameConstraint(frame, mask)` may hold (`add_frame_constraint`): {"name", "frame", "mask"} with bit d
    # of the mask = dof d of (x, y, z, rot x, rot y, rot z) fixed.  Part of the topology, like the contact points.
    constraint_frames: List[Dict[str, Any]] = field(default_factory=list)
    # 1-dof joints a user `JointConstraint(joint)` may hold on a row of its own (`add_joint_constraint`): {"name", "joint"}
    constraint_joints: List[Dict[str, Any]] = field(default_factory=list)
    # flexibility of the spherical joints (`flexibilityConfig`: stiffness / damping per axis), (njoints, 3); None: no such joint
    flex_stiffness: Optional[np.ndarray] = None
    flex_damping: Optional[np.ndarray] = None

    # ---- sizes
    @property
    def njoints(self) -> int:
        return len(self.joint_names)

    @property
    def nq(self) -> int:
        return int(sum(JT_NQ[int(t)] for t in self.jtypes))

    @property
    def nv(self) -> int:
        return int(sum(JT_NV[int(t)] for t in self.jtypes))

    @property
    def nmotors(self) -> int:
        return len(self.motors)

    @property
    def ncontacts(self) -> int:
        return len(self.contacts)

    # ---- lookups (same spirit as reference utilities/pinocchio.cc index helpers)
    def joint_index(self, name: str) -> int:
        try:
            return self.joint_names.index(name)
        except ValueError:
            raise LookupError(f"joint '{name}' not found in model") from None

    def frame(self, name: str) -> Frame:
        try:
            return self.frames[name]
        except KeyError:
            raise LookupError(f"frame '{name}' not found in model") from None

    def neutral(self) -> np.ndarray:
        """Neutral configuration (`pinocchio::neutral`): zeros, [1,0], quat (0,0,0,1)."""
        q = np.zeros(self.nq)
        for j in range(1, self.njoints):
            t, iq = int(self.jtypes[j]), int(self.idx_q[j])
            if t in (JT_RUBX, JT_RUBY, JT_RUBZ, JT_RUBU):
                q[iq] = 1.0
            elif t == JT_FREEFLYER:
                q[iq + 6] = 1.0
            elif t == JT_SPHERICAL:
                q[iq + 3] = 1.0
        return q

    @property
    def flexibility_joint_indices(self) -> List[int]:
        """≙ `robot.flexibility_joint_indices`."""
        return [j for j in range(1, self.njoints) if int(self.jtypes[j]) == JT_SPHERICAL]

    def bounded_position_mask(self) -> np.ndarray:
        """nq mask of coordinates subject to position bounds (1-dof R/P joints only,
        reference engine.cc:3253-3338)."""
        m = np.zeros(self.nq, dtype=bool)
        for j in range(1, self.njoints):
            if int(self.jtypes[j]) in (JT_RX, JT_RY, JT_RZ, JT_RU, JT_PX, JT_PY, JT_PZ, JT_PU):
                m[int(self.idx_q[j])] = True
        return m

    def bound_row(self, joint_name: str) -> int:
        """Row of the joint's `JointConstraint` in the per-lane constraint state (`con_flags` / `con_data`: one row per
        bounded 1-dof joint, model joint order).  LookupError for joints without position bounds."""
        j = self.joint_names.index(joint_name)
        if not 1 <= int(self.jtypes[j]) <= 8:
            raise LookupError(f"joint '{joint_name}' has no position bounds: no constraint row")
        return int(sum(1 for i in range(1, j) if 1 <= int(self.jtypes[i]) <= 8))

    # ---- sensors bookkeeping: fixed layout of the observation vector
    def sensor_names(self, sensor_type: str) -> List[str]:
        return [s["name"] for s in self.sensors.get(sensor_type, [])]

    def topology_signature(self) -> str:
        """Everything the kernels are specialised on at compile time (no parameters)."""
        parts = [
            "J", ",".join(str(int(x)) for x in self.parents),
            "T", ",".join(str(int(x)) for x in self.jtypes),
            "M", ",".join(f"{m.joint}:{int(m.enable_effort_limit)}{int(m.enable_velocity_limit)}"
                          f"{int(m.enable_friction)}" for m in self.motors),
            "C", ",".join(str(self.frames[c].parent_joint) for c in self.contacts),
            "IMU", ",".join(str(self.frames[s["frame"]].parent_joint)
                            for s in self.sensors.get("ImuSensor", [])),
            "F", ",".join(str(self.frames[s["frame"]].parent_joint)
                          for s in self.sensors.get("ForceSensor", [])),
            "CS", ",".join(str(self.contacts.index(s["frame"]))
                           for s in self.sensors.get("ContactSensor", [])),
            "E", ",".join(f"{s['joint']}:{int(s['joint_side'])}"
                          for s in self.sensors.get("EncoderSensor", [])),
            "U", ",".join(str(s["motor_index"]) for s in self.sensors.get("EffortSensor", [])),
        ]
        # "unaligned" 1-dof joints whose axis is the NEGATIVE of a coordinate axis (ANYmal: every other joint turns
        # about -x): structural like RX / RY / RZ, and the branch-parallel kernel specialises on it (jm_quad.h
        # limb_axis_uniform).  Appended only when there is such a joint, so that other topologies keep their hash.
        sa = self.signed_axes()
        if any(int(self.jtypes[j]) in (JT_RU, JT_PU) and sa[j] != 0 for j in range(self.njoints)):
            parts += ["AX", ",".join(str(int(x)) for x in sa)]
        # user constraint frames (parent joint : mask), appended only when there are any: other topologies keep their hash
        if self.constraint_frames:
            parts += ["X", ",".join(f"{self.frames[x['frame']].parent_joint}:{int(x['mask'])}"
                                    + (f":{x['kind']}" if x.get("kind", "frame") != "frame" else "")
                                    + (f":{self.frames[x['frame2']].parent_joint}" if x.get("frame2") else "")
                                    for x in self.constraint_frames)]
        if self.constraint_joints:
            parts += ["XJ", ",".join(str(int(x["joint"])) for x in self.constraint_joints)]
        return "|".join(parts)

    def signed_axes(self) -> np.ndarray:
        """Per joint: +-(i + 1) when the axis of a 1-dof joint is exactly +-e_i, else 0."""
        out = np.zeros(self.njoints, dtype=np.int32)
        for j in range(1, self.njoints):
            if JT_NV.get(int(self.jtypes[j]), 0) != 1:
                continue
            a = np.asarray(self.axes[j], dtype=np.float64)
            for i in range(3):
                e = np.zeros(3)
                e[i] = 1.0
                if np.array_equal(a, e):
                    out[j] = i + 1
                elif np.array_equal(a, -e):
                    out[j] = -(i + 1)
        return out

    def topology_hash(self) -> str:
        return hashlib.sha1(self.topology_signature().encode()).hexdigest()[:12]

    # ---- (de)serialisation
    def to_json(self) -> str:
        def enc(o: Any) -> Any:
            if isinstance(o, np.ndarray):
                return {"__nd__": o.tolist(), "dtype": str(o.dtype)}
            if isinstance(o, (Frame, Motor)):
                return {k: enc(v) for k, v in o.__dict__.items()}
            if isinstance(o, dict):
                return {k: enc(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return [enc(v) for v in o]
            if isinstance(o, float) and math.isinf(o):
                return {"__inf__": 1 if o > 0 else -1}
            if isinstance(o, (np.floating, np.integer, np.bool_)):
                return o.item()
            return o
        return json.dumps({k: enc(v) for k, v in self.__dict__.items()
                           if not k.startswith("_")}, indent=1)

    @staticmethod
    def from_json(text: str) -> "CompiledModel":
        def dec(o: Any) -> Any:
            if isinstance(o, dict):
                if "__nd__" in o:
                    arr = np.array(dec(o["__nd__"]), dtype=o["dtype"])
                    return arr
                if "__inf__" in o:
                    return math.inf * o["__inf__"]
                return {k: dec(v) for k, v in o.items()}
            if isinstance(o, list):
                return [dec(v) for v in o]
            return o

        # (arrays holding infinities go through tolist(): python's json round-trips Infinity)
        raw = json.loads(text)
        d = {k: dec(v) for k, v in raw.items()}
        d["frames"] = {k: Frame(**v) for k, v in d["frames"].items()}
        d["motors"] = [Motor(**m) for m in d["motors"]]
        return CompiledModel(**d)

    def save(self, path: str) -> None:
        with open(path, "w") as f:
            f.write(self.to_json())

    @staticmethod
    def load(path: str) -> "CompiledModel":
        with open(path, "r") as f:
            return CompiledModel.from_json(f.read())


# ----------------------------------------------------------------------------
# build from URDF
# ----------------------------------------------------------------------------

def build_model_from_urdf(urdf_path: str,
                          has_freeflyer: bool = False,
                          name: Optional[str] = None,
                          gravity: Sequence[float] = (0.0, 0.0, -9.81, 0.0, 0.0, 0.0),
                          flexibility: Optional[Sequence[Dict[str, Any]]] = None,
                          backlash: Optional[Dict[str, float]] = None
                          ) -> CompiledModel:
    """URDF -> CompiledModel without hardware (≙ `jiminy.Robot.initialize(urdf, has_freeflyer)`).

    `flexibility`: the reference's `model_options["dynamics"]["flexibilityConfig"]` with `enableFlexibility` -- a list of
    `{"frameName", "stiffness" (3), "damping" (3), "inertia" (3)}`.  A spherical joint is inserted at every named frame
    (Model::addFlexibilityJointsToExtendedModel, core/src/robot/model.cc:1087-1165): in FRONT of a mechanical joint of that name
    (`<name>Flexibility`, at the joint's placement, the mechanical joint then sits at its origin; weightless body --
    addFlexibilityJointBeforeMechanicalJoint, utilities/pinocchio.cc:460-503) or IN PLACE of a fixed joint of that name (the
    links behind it hang on the new joint -- addFlexibilityJointAtFixedFrame :578-700); `inertia` is the joint's rotor
    inertia.  Joints stay numbered depth-first (the reference re-sorts its joints after the insertion).

    `backlash`: `{joint name: backlash}` -- the reference's motor options `enableBacklash` / `backlash`
    (Robot::initializeExtendedModel, core/src/robot/robot.cc:580-629): a second joint of the same kind `<name>Backlash` BEHIND
    the joint, at its origin, that takes over the joint's body (the motorised joint is left weightless, its rotor inertia is
    the motor's armature) and is bounded to +- backlash / 2 (addBacklashJointAfterMechanicalJoint, utilities/pinocchio.cc:505-576).
    Its bound is a `JointConstraint` like any position limit: use `contacts.model = "constraint"`."""
    robot_name, links, joints = _parse_urdf(urdf_path)
    flex = {str(f["frameName"]): f for f in (flexibility or [])}
    backlash = {str(k): float(v) for k, v in (backlash or {}).items() if float(v) >= 2.220446049250313e-16}
    for jn in backlash:
        if jn not in joints or joints[jn].jtype == "fixed":
            raise LookupError(f"joint '{jn}' not found in model: no backlash joint can be inserted behind it")
    for fname in flex:
        if fname not in joints:
            raise LookupError(f"Frame '{fname}' does not exists. Impossible to insert flexibility joint on it.")
    children = {j.child for j in joints.values()}
    roots = [l for l in links if l not in children]
    if len(roots) != 1:
        raise ValueError(f"URDF must have exactly one root link, found {roots}")
    root_link = roots[0]
    # child joints per link, alphabetical by joint name (urdfdom name-keyed map)
    child_joints: Dict[str, List[_UrdfJoint]] = {l: [] for l in links}
    for jn in sorted(joints):
        child_joints[joints[jn].parent].append(joints[jn])

    joint_names = ["universe"]
    parents = [0]
    jtypes = [JT_NONE]
    axes = [np.zeros(3)]
    placements = [SE3()]
    inertias = [Inertia()]
    lower: List[float] = []
    upper: List[float] = []
    eff: List[float] = []
    vel: List[float] = []
    frames: Dict[str, Frame] = {}
    flex_of_joint: Dict[int, Dict[str, Any]] = {}

    def add_spherical(jname: str, parent: int, M: SE3, cfg: Dict[str, Any]) -> int:
        joint_names.append(jname)
        parents.append(parent)
        jtypes.append(JT_SPHERICAL)
        axes.append(np.zeros(3))
        placements.append(M)
        inertias.append(Inertia())
        lower.extend([-1.01] * 4)
        upper.extend([1.01] * 4)
        eff.extend([math.inf] * 3)
        vel.extend([math.inf] * 3)
        new = len(joint_names) - 1
        flex_of_joint[new] = cfg
        add_frame(jname, new, SE3(), "joint")
        return new

    def add_frame(fname: str, jidx: int, M: SE3, kind: str) -> None:
        if fname in frames:
            # joint and link may share a name in some URDFs (e.g. ANYmal `LF_HAA`):
            # body frames take precedence for sensors/contacts, as `getFrameIndex`
            # returns the first match and BODY frames of links are what users name.
            if kind != "body":
                return
        frames[fname] = Frame(fname, jidx, M.R.copy(), M.p.copy(), kind)

    if has_freeflyer:
        joint_names.append("root_joint")
        parents.append(0)
        jtypes.append(JT_FREEFLYER)
        axes.append(np.zeros(3))
        placements.append(SE3())
        inertias.append(Inertia())
        big = math.inf
        lower += [-big] * 3 + [-1.01] * 4
        upper += [big] * 3 + [1.01] * 4
        eff += [math.inf] * 6
        vel += [math.inf] * 6
        add_frame("root_joint", 1, SE3(), "joint")
        root_joint_idx = 1
    else:
        root_joint_idx = 0

    def visit(link_name: str, jidx: int, M_link: SE3) -> None:
        """`link_name` is rigidly attached to joint `jidx` with placement `M_link`."""
        link = links[link_name]
        add_frame(link_name, jidx, M_link, "body")
        if link.inertia.mass != 0.0 or np.any(link.inertia.I != 0.0):
            inertias[jidx] = inertias[jidx].add(link.inertia.transformed(M_link))
        for uj in child_joints[link_name]:
            M_joint = M_link * uj.origin
            if uj.jtype == "fixed" and uj.name in flex:
                # flexibility in place of the fixed joint: what hangs behind it hangs on the spherical joint
                new = add_spherical(uj.name, jidx, M_joint, flex[uj.name])
                visit(uj.child, new, SE3())
            elif uj.jtype == "fixed":
                add_frame(uj.name, jidx, M_joint, "fixed_joint")
                visit(uj.child, jidx, M_joint)
            else:
                t, ax = _classify_joint(uj)
                jparent = jidx
                if uj.name in flex:
                    # flexibility in front of the mechanical joint: at the joint's placement, the joint at its origin
                    jparent = add_spherical(uj.name + FLEXIBLE_JOINT_SUFFIX, jidx, M_joint, flex[uj.name])
                    M_joint = SE3()
                joint_names.append(uj.name)
                parents.append(jparent)
                jtypes.append(t)
                axes.append(ax)
                placements.append(M_joint)
                inertias.append(Inertia())
                new = len(joint_names) - 1
                if t in (JT_RUBX, JT_RUBY, JT_RUBZ, JT_RUBU):
                    lower.extend([-1.01, -1.01])
                    upper.extend([1.01, 1.01])
                else:
                    lower.append(uj.lower)
                    upper.append(uj.upper)
                eff.append(uj.effort)
                vel.append(uj.velocity)
                add_frame(uj.name, new, SE3(), "joint")
                if uj.name in backlash:
                    if t not in (JT_RX, JT_RY, JT_RZ, JT_RU, JT_PX, JT_PY, JT_PZ, JT_PU):
                        raise NotImplementedError("Backlash can only be associated with a bounded 1-dof linear or rotary joint "
                                                  "here (the reference also takes unbounded rotary ones).")
                    joint_names.append(uj.name + BACKLASH_JOINT_SUFFIX)
                    parents.append(new)
                    jtypes.append(t)
                    axes.append(ax)
                    placements.append(SE3())
                    inertias.append(Inertia())
                    lower.append(-0.5 * backlash[uj.name])
                    upper.append(0.5 * backlash[uj.name])
                    eff.append(math.inf)
                    vel.append(math.inf)
                    new = len(joint_names) - 1
                    add_frame(uj.name + BACKLASH_JOINT_SUFFIX, new, SE3(), "joint")
                visit(uj.child, new, SE3())

    visit(root_link, root_joint_idx, SE3())

    n = len(joint_names)
    idx_q = np.zeros(n, dtype=np.int32)
    idx_v = np.zeros(n, dtype=np.int32)
    iq = iv = 0
    for j in range(n):
        idx_q[j], idx_v[j] = iq, iv
        iq += JT_NQ[jtypes[j]]
        iv += JT_NV[jtypes[j]]
    rotor = np.zeros(iv)
    fk = fd = None
    if flex_of_joint:
        fk, fd = np.zeros((n, 3)), np.zeros((n, 3))
        for j, cfg in flex_of_joint.items():
            fk[j] = np.asarray(cfg["stiffness"], dtype=np.float64)
            fd[j] = np.asarray(cfg["damping"], dtype=np.float64)
            rotor[idx_v[j]:idx_v[j] + 3] = np.asarray(cfg["inertia"], dtype=np.float64)   # model.cc:1133-1143
    model = CompiledModel(
        name=name or robot_name,
        has_freeflyer=has_freeflyer,
        joint_names=joint_names,
        parents=np.array(parents, dtype=np.int32),
        jtypes=np.array(jtypes, dtype=np.int32),
        axes=np.array(axes, dtype=np.float64),
        idx_q=idx_q, idx_v=idx_v,
        placement_R=np.array([M.R for M in placements]),
        placement_p=np.array([M.p for M in placements]),
        mass=np.array([I.mass for I in inertias]),
        com=np.array([I.com for I in inertias]),
        inertia=np.array([I.I for I in inertias]),
        rotor_inertia=rotor,
        position_lower=np.array(lower, dtype=np.float64),
        position_upper=np.array(upper, dtype=np.float64),
        effort_limit=np.array(eff, dtype=np.float64),
        velocity_limit=np.array(vel, dtype=np.float64),
        gravity=np.array(gravity, dtype=np.float64),
        frames=frames, motors=[], contacts=[],
        sensors={k: [] for k in SENSOR_TYPES}, flex_stiffness=fk, flex_damping=fd)
    if flex_of_joint:
        # model.cc:1146-1163: the diagonal inertia seen by a flexibility joint must not vanish
        for j in flex_of_joint:
            diag = rotor[idx_v[j]:idx_v[j] + 3] + np.diag(model.inertia[j])
            if np.any(diag < 1e-5):
                raise ValueError(f"The subtree diagonal inertia for flexibility joint {j} must be larger than 1e-5 for "
                                 f"numerical stability: {diag}")
    model._urdf_links = links  # type: ignore[attr-defined]  (collision boxes for hardware loading)
    return model


# ----------------------------------------------------------------------------
# hardware (motors / sensors / contact points) -- mirrors robot.py semantics
# ----------------------------------------------------------------------------

def add_frame(model: CompiledModel, frame_name: str, body_name: str,
              R: np.ndarray, p: np.ndarray) -> None:
    """≙ `Robot.add_frame(name, parent_body, placement)`."""
    parent = model.frame(body_name)
    M = SE3(parent.R, parent.p) * SE3(R, p)
    if frame_name in model.frames:
        raise ValueError(f"frame '{frame_name}' already exists")
    model.frames[frame_name] = Frame(frame_name, parent.parent_joint, M.R, M.p, "op")


def add_contact_points(model: CompiledModel, frame_names: Sequence[str]) -> None:
    """≙ `Robot.add_contact_points` (reference core/src/robot/model.cc addContactPoints)."""
    for fn in frame_names:
        model.frame(fn)
        if fn in model.contacts:
            raise ValueError(f"contact point '{fn}' already registered")
        model.contacts.append(fn)


def add_frame_constraint(model: CompiledModel, name: str, frame_name: str,
                         mask_dofs: Sequence[bool] = (True, True, True, True, True, True)) -> None:
    """≙ `robot.add_constraint(name, jiminy.FrameConstraint(frame_name, mask_dofs))` at MODEL level (reference
    core/src/constraints/frame_constraint.cc:27-35, Model::addConstraint model.cc:926-936): declares that robots of this
    model may hold the frame at a reference pose along the masked dofs (x, y, z, rot x, rot y, rot z; world aligned).  The
    kernels are specialised on it like on the contact points, so it is declared before the engine is created; which
    lanes actually hold it (and with which gains / reference) is run-time state: `BatchedEngine.add_constraint(name,
    FrameConstraint(frame_name, mask_dofs))`."""
    model.frame(frame_name)
    if len(mask_dofs) != 6 or not any(mask_dofs):
        raise ValueError("mask_dofs must hold six booleans, at least one of them set")
    if any(x["name"] == name for x in model.constraint_frames + model.constraint_joints):
        raise ValueError(f"constraint '{name}' already declared")     # model.cc: "A constraint with name ... already exists"
    model.constraint_frames.append({"name": name, "frame": frame_name,
                                    "mask": int(sum(1 << d for d in range(6) if mask_dofs[d]))})


def add_sphere_constraint(model: CompiledModel, name: str, frame_name: str, radius: float,
                          ground_normal: Sequence[float] = (0.0, 0.0, 1.0)) -> None:
    """≙ `jiminy.SphereConstraint(frame_name, radius, ground_normal)` (core/src/constraints/sphere_constraint.cc): a sphere
    of that radius centred on the frame rolls on the plane through its lowest point without slipping or lifting off -- the
    three velocity components of the contact point (frame origin - radius * normal) are held at zero."""
    model.frame(frame_name)
    if any(x["name"] == name for x in model.constraint_frames + model.constraint_joints):
        raise ValueError(f"constraint '{name}' already declared")
    n = np.asarray(ground_normal, dtype=np.float64)
    model.constraint_frames.append({"name": name, "frame": frame_name, "mask": 7, "kind": "sphere", "radius": float(radius),
                                    "normal": [float(v) for v in n / np.linalg.norm(n)]})


def add_wheel_constraint(model: CompiledModel, name: str, frame_name: str, radius: float,
                         ground_normal: Sequence[float] = (0.0, 0.0, 1.0), wheel_axis: Sequence[float] = (0.0, 0.0, 1.0)) -> None:
    """≙ `jiminy.WheelConstraint(frame_name, radius, ground_normal, wheel_axis)` (core/src/constraints/wheel_constraint.cc): a
    thin wheel whose axis is `wheel_axis` in the frame rolls on the plane without slipping; the contact point follows the
    wheel's tilt (frame origin - radius * y with y the direction from the contact point to the centre)."""
    model.frame(frame_name)
    if any(x["name"] == name for x in model.constraint_frames + model.constraint_joints):
        raise ValueError(f"constraint '{name}' already declared")
    n = np.asarray(ground_normal, dtype=np.float64)
    a = np.asarray(wheel_axis, dtype=np.float64)
    model.constraint_frames.append({"name": name, "frame": frame_name, "mask": 7, "kind": "wheel", "radius": float(radius),
                                    "normal": [float(v) for v in n / np.linalg.norm(n)],
                                    "axis": [float(v) for v in a / np.linalg.norm(a)]})


def add_distance_constraint(model: CompiledModel, name: str, first_frame_name: str, second_frame_name: str) -> None:
    """≙ `jiminy.DistanceConstraint(first_frame_name, second_frame_name)` (core/src/constraints/distance_constraint.cc): the
    distance between the origins of the two frames is held at its value at `start` (Cassie's and Digit's push-rods,
    gym_jiminy/envs/cassie.py:135-158).  Either frame may be fixed to the world (parent joint 0)."""
    model.frame(first_frame_name)
    model.frame(second_frame_name)
    if any(x["name"] == name for x in model.constraint_frames + model.constraint_joints):
        raise ValueError(f"constraint '{name}' already declared")
    model.constraint_frames.append({"name": name, "frame": first_frame_name, "frame2": second_frame_name, "mask": 1,
                                    "kind": "distance"})


def add_joint_constraint(model: CompiledModel, name: str, joint_name: str) -> None:
    """≙ `robot.add_constraint(name, jiminy.JointConstraint(joint_name))` at MODEL level (joint_constraint.cc, Model::addConstraint
    model.cc:926-936): robots of this model may hold the 1-dof joint at a reference position through a constraint row OF ITS
    OWN, which coexists with the joint's bound constraint like in the reference and also exists for joints without position
    bounds.  Declared before the engine is created (the kernels are specialised on it; one-robot-per-lane kernel family);
    registered per lane with `BatchedEngine.add_constraint(name, JointConstraint(joint_name))`."""
    j = model.joint_index(joint_name)
    if not 1 <= int(model.jtypes[j]) <= 8:
        raise NotImplementedError("JointConstraint rows of their own: revolute / prismatic 1-dof joints")
    if any(x["name"] == name for x in model.constraint_frames + model.constraint_joints):
        raise ValueError(f"constraint '{name}' already declared")
    if any(x["joint"] == j for x in model.constraint_joints):
        raise ValueError(f"joint '{joint_name}' already carries a declared user constraint")
    model.constraint_joints.append({"name": name, "joint": int(j)})


def add_motor(model: CompiledModel, name: str, joint_name: str, **options: Any) -> Motor:
    """≙ `SimpleMotor(name)` + `attach_motor` + `initialize(joint_name)` + `set_options`.

    Option names and defaults are the reference's
    (core/include/jiminy/core/hardware/abstract_motor.h:41-57, basic_motors.h:15-31).
    """
    opts = dict(mechanicalReduction=1.0, velocityLimitFromUrdf=True, velocityLimit=0.0,
                effortLimitFromUrdf=True, effortLimit=0.0, enableArmature=False, armature=0.0,
                enableBacklash=False, backlash=0.0,
                enableVelocityLimit=False, velocityEffortInvSlope=0.0, enableEffortLimit=True,
                enableFriction=False, frictionViscousPositive=0.0, frictionViscousNegative=0.0,
                frictionDryPositive=0.0, frictionDryNegative=0.0, frictionDrySlope=0.0)
    for k, v in options.items():
        if k not in opts:
            raise ValueError(f"'{k}' is not a valid option for motor '{name}'")
        opts[k] = v
    if opts["enableBacklash"]:
        raise NotImplementedError("motor backlash is outside the batched hot path")
    if any(m.name == name for m in model.motors):
        raise ValueError(f"another motor with name '{name}' is already attached")
    j = model.joint_index(joint_name)
    t = int(model.jtypes[j])
    if JT_NV[t] != 1:
        raise ValueError("a motor can only be associated with a 1-dof joint")
    iv = int(model.idx_v[j])
    red = float(opts["mechanicalReduction"])
    # abstract_motor.cc:311-335
    eff = model.effort_limit[iv] / red if opts["effortLimitFromUrdf"] else float(opts["effortLimit"])
    vel = model.velocity_limit[iv] * red if opts["velocityLimitFromUrdf"] else float(opts["velocityLimit"])
    arm = float(opts["armature"]) * red ** 2 if opts["enableArmature"] else 0.0
    m = Motor(name=name, joint_name=joint_name, joint=j, idx_q=int(model.idx_q[j]), idx_v=iv,
              reduction=red, effort_limit=float(eff), velocity_limit=float(vel),
              enable_effort_limit=bool(opts["enableEffortLimit"]),
              enable_velocity_limit=bool(opts["enableVelocityLimit"]),
              velocity_effort_inv_slope=float(opts["velocityEffortInvSlope"]),
              armature=arm, enable_friction=bool(opts["enableFriction"]),
              friction_viscous_pos=float(opts["frictionViscousPositive"]),
              friction_viscous_neg=float(opts["frictionViscousNegative"]),
              friction_dry_pos=float(opts["frictionDryPositive"]),
              friction_dry_neg=float(opts["frictionDryNegative"]),
              friction_dry_slope=float(opts["frictionDrySlope"]))
    model.motors.append(m)
    model.rotor_inertia[iv] += arm  # robot.cc:243-246
    return m


def add_sensor(model: CompiledModel, sensor_type: str, name: str, **kw: Any) -> None:
    """≙ `<SensorType>(name)` + `attach_sensor` + `initialize(**kw)`."""
    if sensor_type not in SENSOR_TYPES:
        raise NotImplementedError(f"sensor type '{sensor_type}' is not supported")
    if any(s["name"] == name for s in model.sensors[sensor_type]):
        raise ValueError(f"a {sensor_type} named '{name}' is already attached")
    if sensor_type in ("ImuSensor", "ForceSensor"):
        model.frame(kw["frame_name"])
        model.sensors[sensor_type].append({"name": name, "frame": kw["frame_name"]})
    elif sensor_type == "ContactSensor":
        if kw["frame_name"] not in model.contacts:
            raise ValueError("sensor frame not associated with any contact point of the robot")
        model.sensors[sensor_type].append({"name": name, "frame": kw["frame_name"]})
    elif sensor_type == "EncoderSensor":
        if "joint_name" in kw and kw["joint_name"] is not None:
            j = model.joint_index(kw["joint_name"])
            rec = {"name": name, "joint": j, "joint_side": True, "reduction": 1.0,
                   "motor_index": -1}
        else:
            mi = [m.name for m in model.motors].index(kw["motor_name"])
            mot = model.motors[mi]
            rec = {"name": name, "joint": mot.joint, "joint_side": False,
                   "reduction": mot.reduction, "motor_index": mi}
        if JT_NV[int(model.jtypes[rec["joint"]])] != 1:
            raise ValueError("encoder sensors can only be associated with a 1-dof joint")
        model.sensors[sensor_type].append(rec)
    elif sensor_type == "EffortSensor":
        names = [m.name for m in model.motors]
        if kw["motor_name"] not in names:
            raise ValueError(f"'{kw['motor_name']}' is not a valid motor name")
        model.sensors[sensor_type].append(
            {"name": name, "motor_index": names.index(kw["motor_name"])})


def load_hardware_description_file(model: CompiledModel, hardware_path: str,
                                   avoid_instable_collisions: bool = True) -> Dict[str, Any]:
    """Apply a Jiminy `*_hardware.toml` to the model.

    Same rules as reference python/jiminy_py/src/jiminy_py/robot.py:518-860:
    collision bodies with box primitives are replaced by contact points at the 8
    vertices of each box (`:628-647`, names `<body>_CollisionBox_<i>_<j>`),
    contact points are registered in **sorted name order** (`:717`), every motor
    gets `enableArmature=True` (`:753`).  Mesh-based collision bodies need
    trimesh bounding boxes and are not supported here.
    """
    with open(hardware_path, "rb") as f:
        info = _toml.load(f)
    extra = info.pop("Global", {})
    motors_info = info.pop("Motor", {})
    sensors_info = info.pop("Sensor", {})
    collision_body_names = list(extra.pop("collisionBodyNames", []))
    contact_frame_names = list(extra.pop("contactFrameNames", []))
    links = getattr(model, "_urdf_links", {})
    for body_name in collision_body_names:
        boxes = links[body_name].collision_boxes if body_name in links else []
        if not boxes:
            raise NotImplementedError(
                f"collision body '{body_name}' has no box primitive; mesh bounding boxes "
                "(trimesh) are outside this compiler")
        if not avoid_instable_collisions:
            raise NotImplementedError("true collision bodies (hpp-fcl) are out of scope")
        for i, (size, origin) in enumerate(boxes):
            verts = [e.flatten() for e in np.meshgrid(
                *[0.5 * v * np.array([-1.0, 1.0]) for v in size])]
            for j, (x, y, z) in enumerate(zip(*verts)):
                fname = "_".join((body_name, "CollisionBox", str(i), str(j)))
                M = origin * SE3(np.eye(3), np.array([x, y, z]))
                add_frame(model, fname, body_name, M.R, M.p)
                contact_frame_names.append(fname)
    add_contact_points(model, sorted(set(contact_frame_names)))

    for motor_type, descr in motors_info.items():
        if motor_type != "SimpleMotor":
            raise NotImplementedError(f"motor type '{motor_type}' is not supported")
        for motor_name, d in descr.items():
            d = dict(d)
            joint_name = d.pop("joint_name")
            if joint_name not in model.joint_names:
                continue
            d["enableArmature"] = True
            add_motor(model, motor_name, joint_name, **d)

    for sensor_type, descr in sensors_info.items():
        for sensor_name, d in descr.items():
            d = dict(d)
            kw = {k: d.pop(k) for k in ("joint_name", "motor_name", "frame_name",
                                        "body_name", "frame_pose") if k in d}
            fname = kw.get("frame_name")
            if fname is not None and fname not in model.frames:
                pose = np.array(kw.pop("frame_pose"), dtype=np.float64)
                add_frame(model, fname, kw.pop("body_name"), rpy_to_matrix(pose[3:]), pose[:3])
            add_sensor(model, sensor_type, sensor_name, **kw)
            # measurement options of the hardware file (abstract_sensor.h:66-100): kept with the sensor record
            # and applied by BatchedEngine at construction (`set_sensor_options`)
            opt = {k: np.asarray(d[k], dtype=np.float64).reshape(-1).tolist()
                   for k in ("noiseStd", "bias", "delay", "jitter") if k in d and np.any(np.asarray(d[k]) != 0.0)}
            if opt:
                if "delayInterpolationOrder" in d:
                    opt["delayInterpolationOrder"] = int(d["delayInterpolationOrder"])
                model.sensors[sensor_type][-1]["options"] = opt
    return extra


def build_robot(urdf_path: str, hardware_path: Optional[str] = None,
                has_freeflyer: bool = False, name: Optional[str] = None,
                flexibility: Optional[Sequence[Dict[str, Any]]] = None,
                backlash: Optional[Dict[str, float]] = None) -> CompiledModel:
    """≙ `BaseJiminyRobot.initialize(urdf_path, hardware_path, has_freeflyer=...)`.

    As in the reference, a `<urdf>_hardware.toml` next to the URDF is picked up
    automatically when `hardware_path` is None.
    """
    model = build_model_from_urdf(urdf_path, has_freeflyer, name, flexibility=flexibility, backlash=backlash)
    if hardware_path is None:
        cand = os.path.splitext(urdf_path)[0] + "_hardware.toml"
        if os.path.exists(cand):
            hardware_path = cand
    if hardware_path is not None:
        load_hardware_description_file(model, hardware_path)
    return model


_MODELS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "models")


def load_builtin(name: str) -> CompiledModel:
    """Load one of the pre-compiled models shipped with the package (no URDF needed)."""
    path = os.path.join(_MODELS_DIR, name + ".json")
    if not os.path.exists(path):
        raise LookupError(f"no built-in model '{name}' (looked for {path})")
    return CompiledModel.load(path)
