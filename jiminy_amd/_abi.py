"""ctypes mirror of include/jiminy_hip.h (structs, enums) and CompiledModel -> jm_model_desc."""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, List, Tuple

import numpy as np

from .model import CompiledModel

ABI_VERSION = 9

# error codes
JM_OK, JM_EINVAL, JM_ERUNTIME, JM_ECONTROLFLOW = 0, -1, -2, -3
JM_ELOOKUP, JM_ENOTIMPL, JM_ETOPOLOGY = -4, -5, -6

JM_F64, JM_F32 = 0, 1
JM_SOLVER_EULER_EXPLICIT, JM_SOLVER_RUNGE_KUTTA_4, JM_SOLVER_RUNGE_KUTTA_DOPRI = 0, 1, 2
JM_MOTOR_EFFORT_LIMIT, JM_MOTOR_VELOCITY_LIMIT, JM_MOTOR_FRICTION = 1, 2, 4
JM_MOTOR_NPARAMS = 9
JM_LANE_OK, JM_LANE_NAN, JM_LANE_OUT_OF_BOUNDS, JM_LANE_FORCE_OVERFLOW, JM_LANE_STEPPER_FAILURE = 0, 1, 2, 4, 8
JM_LANE_SOLVER_FAILURE = 16
JM_CONTACT_SPRING_DAMPER, JM_CONTACT_CONSTRAINT = 0, 1
CONTACT_MODELS = {"spring_damper": JM_CONTACT_SPRING_DAMPER, "constraint": JM_CONTACT_CONSTRAINT}

(JM_F_Q, JM_F_V, JM_F_A, JM_F_COMMAND, JM_F_U_MOTOR, JM_F_U, JM_F_F_EXTERNAL,
 JM_F_CONTACT_FORCES, JM_F_IMU, JM_F_FORCE, JM_F_CONTACT, JM_F_ENCODER, JM_F_EFFORT,
 JM_F_ENERGY, JM_F_JOINT_FORCES, JM_F_CENTROIDAL, JM_F_STATUS, JM_F_WORKSPACE,
 JM_F_CON_FLAGS, JM_F_CON_DATA, JM_F_FRICTION, JM_F_MODEL_LANE, JM_F_APPLIED, JM_F_GROUND_OFFSET, JM_F_FLEXIBILITY,
 JM_F_COUNT) = range(26)

FIELD_NAMES = {
    "q": JM_F_Q, "v": JM_F_V, "a": JM_F_A, "command": JM_F_COMMAND, "u_motor": JM_F_U_MOTOR,
    "u": JM_F_U, "f_external": JM_F_F_EXTERNAL, "contact_forces": JM_F_CONTACT_FORCES,
    "imu": JM_F_IMU, "force": JM_F_FORCE, "contact": JM_F_CONTACT, "encoder": JM_F_ENCODER,
    "effort": JM_F_EFFORT, "energy": JM_F_ENERGY, "joint_forces": JM_F_JOINT_FORCES,
    "centroidal": JM_F_CENTROIDAL, "status": JM_F_STATUS, "workspace": JM_F_WORKSPACE,
    "con_flags": JM_F_CON_FLAGS, "con_data": JM_F_CON_DATA, "friction": JM_F_FRICTION,
    "model_lane": JM_F_MODEL_LANE, "applied": JM_F_APPLIED, "ground_offset": JM_F_GROUND_OFFSET,
    "flexibility": JM_F_FLEXIBILITY,
}

_pi = C.POINTER(C.c_int32)
_pd = C.POINTER(C.c_double)


class ModelDesc(C.Structure):
    _fields_ = [
        ("njoints", C.c_int32), ("nq", C.c_int32), ("nv", C.c_int32),
        ("nmotors", C.c_int32), ("ncontacts", C.c_int32),
        ("nimu", C.c_int32), ("nforce", C.c_int32), ("ncontact_sensors", C.c_int32),
        ("nencoder", C.c_int32), ("neffort", C.c_int32),
        ("parents", _pi), ("jtypes", _pi), ("idx_q", _pi), ("idx_v", _pi),
        ("axes", _pd), ("placement_R", _pd), ("placement_p", _pd),
        ("mass", _pd), ("com", _pd), ("inertia", _pd), ("rotor_inertia", _pd),
        ("position_lower", _pd), ("position_upper", _pd),
        ("motor_joint", _pi), ("motor_flags", _pi), ("motor_params", _pd),
        ("contact_joint", _pi), ("contact_R", _pd), ("contact_p", _pd),
        ("imu_joint", _pi), ("imu_R", _pd), ("imu_p", _pd),
        ("force_joint", _pi), ("force_R", _pd), ("force_p", _pd),
        ("contact_sensor_contact", _pi),
        ("encoder_joint", _pi), ("encoder_joint_side", _pi), ("encoder_reduction", _pd),
        ("effort_motor", _pi),
        ("n_constraint_frames", C.c_int32),
        ("cframe_joint", _pi), ("cframe_mask", _pi), ("cframe_R", _pd), ("cframe_p", _pd),
        ("cframe_kind", _pi), ("cframe_joint2", _pi), ("cframe_params", _pd),
        ("n_constraint_joints", C.c_int32), ("cjoint_joint", _pi),
        ("flex_stiffness", _pd), ("flex_damping", _pd),
    ]


class Options(C.Structure):
    _fields_ = [
        ("gravity", C.c_double * 6),
        ("contact_stiffness", C.c_double),
        ("contact_damping", C.c_double),
        ("contact_friction", C.c_double),
        ("contact_transition_eps", C.c_double),
        ("contact_transition_velocity", C.c_double),
    ]


class ConstraintOptions(C.Structure):
    """struct jm_constraint_options (include/jiminy_hip.h)."""
    _fields_ = [
        ("contact_model", C.c_int32),
        ("pgs_iter_max", C.c_int32),
        ("torsion", C.c_double),
        ("stabilization_freq", C.c_double),
        ("regularization", C.c_double),
        ("tol_abs", C.c_double),
        ("tol_rel", C.c_double),
        ("user_stabilization_freq", C.c_double),
    ]


def make_constraint_options(model="constraint", torsion=0.0, stabilization_freq=20.0,
                            regularization=1.0e-3, tol_abs=1.0e-5, tol_rel=1.0e-4,
                            pgs_iter_max=100, user_stabilization_freq=-1.0) -> ConstraintOptions:
    """Defaults = reference engine.h:262-286, 307-325 (`contacts`, `constraints`, `stepper.tol*`),
    PGS_MAX_ITERATIONS engine.cc:62."""
    o = ConstraintOptions()
    o.contact_model = CONTACT_MODELS[model]
    o.pgs_iter_max = int(pgs_iter_max)
    o.torsion = float(torsion)
    o.stabilization_freq = float(stabilization_freq)
    o.regularization = float(regularization)
    o.tol_abs = float(tol_abs)
    o.tol_rel = float(tol_rel)
    o.user_stabilization_freq = float(user_stabilization_freq)   # (< 0: user constraints share the gains of `stabilization_freq`)
    return o


def constraint_rows(model: CompiledModel) -> Dict[str, int]:
    """Rows of the per-lane constraint state: one constraint per bounded 1-dof joint (model joint
    order), one per contact point with 4 rows (x, y, z, torsion), one per user constraint frame with 6 rows
    (x, y, z, rot x, rot y, rot z: the dofs outside its mask are never active).  `con_data`: reference
    configuration of every bound, the multipliers of all rows, then 12 rows of reference transform per user frame."""
    nb = int(sum(1 for t in model.jtypes if 1 <= int(t) <= 8))
    nc = model.ncontacts
    nx = len(model.constraint_frames)
    nxj = len(model.constraint_joints)      # user JointConstraints on rows of their own (one-robot-per-lane kernels)
    nr = nb + 4 * nc + 6 * nx + nxj
    return {"n_bounds": nb, "n_contacts": nc, "n_user_frames": nx, "n_user_joints": nxj, "n_rows": nr,
            "con_flags": nb + nc + nx + nxj, "con_data": nb + nr + 12 * nx + nxj,
            "user_lambda": 2 * nb + 4 * nc, "user_ref": nb + nr,
            "user_joint_flag": nb + nc + nx, "user_joint_lambda": 2 * nb + 4 * nc + 6 * nx, "user_joint_ref": nb + nr + 12 * nx,
            "workspace": nr * nr + 5 * nr + 3 * model.nv}


class AdaptiveOptions(C.Structure):
    """struct jm_adaptive_options (include/jiminy_hip.h)."""
    _fields_ = [
        ("tol_rel", C.c_double),
        ("tol_abs", C.c_double),
        ("dt_max", C.c_double),
        ("dt_restore_threshold_rel", C.c_double),
        ("successive_iter_failed_max", C.c_int32),
        ("form", C.c_int32),
    ]


def make_options(gravity=(0.0, 0.0, -9.81, 0.0, 0.0, 0.0), stiffness=1.0e6, damping=2.0e3,
                 friction=1.0, transition_eps=1.0e-3, transition_velocity=1.0e-2) -> Options:
    """Defaults = reference engine.h:273-291."""
    o = Options()
    for i in range(6):
        o.gravity[i] = float(gravity[i])
    o.contact_stiffness = float(stiffness)
    o.contact_damping = float(damping)
    o.contact_friction = float(friction)
    o.contact_transition_eps = float(transition_eps)
    o.contact_transition_velocity = float(transition_velocity)
    return o


def _i32(x: Any) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(x, dtype=np.int32).reshape(-1))
    return a if a.size else np.zeros(1, dtype=np.int32)


def _f64(x: Any) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
    return a if a.size else np.zeros(1, dtype=np.float64)


XKINDS = {"frame": 0, "sphere": 1, "wheel": 2, "distance": 3}


def constraint_frame_params(model: CompiledModel, x: Dict[str, Any]) -> List[float]:
    """The 8 parameters of a user constraint frame (jm_model_desc::cframe_params)."""
    kind = x.get("kind", "frame")
    if kind == "distance":
        return [0.0, *[float(v) for v in model.frames[x["frame2"]].p], 0.0, 0.0, 0.0, 0.0]
    n = [float(v) for v in x.get("normal", (0.0, 0.0, 1.0))]
    a = [float(v) for v in x.get("axis", (0.0, 0.0, 0.0))]
    return [float(x.get("radius", 0.0)), *n, *a, 0.0]


def make_model_desc(model: CompiledModel) -> Tuple[ModelDesc, List[np.ndarray]]:
    """Flatten a CompiledModel into the C description. Returns (desc, keepalive arrays)."""
    keep: List[np.ndarray] = []

    def pi(x: Any) -> Any:
        a = _i32(x)
        keep.append(a)
        return a.ctypes.data_as(_pi)

    def pd(x: Any) -> Any:
        a = _f64(x)
        keep.append(a)
        return a.ctypes.data_as(_pd)

    d = ModelDesc()
    d.njoints, d.nq, d.nv = model.njoints, model.nq, model.nv
    d.nmotors, d.ncontacts = model.nmotors, model.ncontacts
    s = model.sensors
    d.nimu = len(s.get("ImuSensor", []))
    d.nforce = len(s.get("ForceSensor", []))
    d.ncontact_sensors = len(s.get("ContactSensor", []))
    d.nencoder = len(s.get("EncoderSensor", []))
    d.neffort = len(s.get("EffortSensor", []))
    d.parents, d.jtypes = pi(model.parents), pi(model.jtypes)
    d.idx_q, d.idx_v = pi(model.idx_q), pi(model.idx_v)
    d.axes = pd(model.axes)
    d.placement_R, d.placement_p = pd(model.placement_R), pd(model.placement_p)
    d.mass, d.com, d.inertia = pd(model.mass), pd(model.com), pd(model.inertia)
    d.rotor_inertia = pd(model.rotor_inertia)
    d.position_lower, d.position_upper = pd(model.position_lower), pd(model.position_upper)
    d.motor_joint = pi([m.joint for m in model.motors])
    d.motor_flags = pi([
        (JM_MOTOR_EFFORT_LIMIT if m.enable_effort_limit else 0)
        | (JM_MOTOR_VELOCITY_LIMIT if m.enable_velocity_limit else 0)
        | (JM_MOTOR_FRICTION if m.enable_friction else 0) for m in model.motors])
    d.motor_params = pd([[m.reduction, m.effort_limit, m.velocity_limit,
                          m.velocity_effort_inv_slope, m.friction_viscous_pos,
                          m.friction_viscous_neg, m.friction_dry_pos, m.friction_dry_neg,
                          m.friction_dry_slope] for m in model.motors])
    cf = [model.frames[c] for c in model.contacts]
    d.contact_joint = pi([f.parent_joint for f in cf])
    d.contact_R, d.contact_p = pd([f.R for f in cf]), pd([f.p for f in cf])
    imu = [model.frames[x["frame"]] for x in s.get("ImuSensor", [])]
    d.imu_joint = pi([f.parent_joint for f in imu])
    d.imu_R, d.imu_p = pd([f.R for f in imu]), pd([f.p for f in imu])
    frc = [model.frames[x["frame"]] for x in s.get("ForceSensor", [])]
    d.force_joint = pi([f.parent_joint for f in frc])
    d.force_R, d.force_p = pd([f.R for f in frc]), pd([f.p for f in frc])
    d.contact_sensor_contact = pi([model.contacts.index(x["frame"])
                                   for x in s.get("ContactSensor", [])])
    enc = s.get("EncoderSensor", [])
    d.encoder_joint = pi([x["joint"] for x in enc])
    d.encoder_joint_side = pi([1 if x["joint_side"] else 0 for x in enc])
    d.encoder_reduction = pd([x["reduction"] for x in enc])
    d.effort_motor = pi([x["motor_index"] for x in s.get("EffortSensor", [])])
    xf = [model.frames[x["frame"]] for x in model.constraint_frames]
    d.n_constraint_frames = len(xf)
    d.cframe_joint = pi([f.parent_joint for f in xf])
    d.cframe_mask = pi([x["mask"] for x in model.constraint_frames])
    d.cframe_R, d.cframe_p = pd([f.R for f in xf]), pd([f.p for f in xf])
    d.cframe_kind = pi([XKINDS[x.get("kind", "frame")] for x in model.constraint_frames])
    d.cframe_joint2 = pi([model.frames[x["frame2"]].parent_joint if x.get("frame2") else 0 for x in model.constraint_frames])
    d.cframe_params = pd([constraint_frame_params(model, x) for x in model.constraint_frames])
    d.n_constraint_joints = len(model.constraint_joints)
    d.cjoint_joint = pi([x["joint"] for x in model.constraint_joints])
    if getattr(model, "flex_stiffness", None) is not None:
        d.flex_stiffness, d.flex_damping = pd(model.flex_stiffness), pd(model.flex_damping)
    return d, keep


def field_rows(model: CompiledModel) -> Dict[str, int]:
    """Number of [B]-rows of each bindable field."""
    s = model.sensors
    return {
        "q": model.nq, "v": model.nv, "a": model.nv, "command": model.nmotors,
        "u_motor": model.nmotors, "u": model.nv, "f_external": 6 * model.njoints,
        "contact_forces": 6 * model.ncontacts,
        "imu": 6 * len(s.get("ImuSensor", [])), "force": 6 * len(s.get("ForceSensor", [])),
        "contact": 3 * len(s.get("ContactSensor", [])),
        "encoder": 2 * len(s.get("EncoderSensor", [])),
        "effort": len(s.get("EffortSensor", [])), "energy": 2,
        "joint_forces": 6 * model.njoints, "centroidal": 15, "status": 1,
    }
