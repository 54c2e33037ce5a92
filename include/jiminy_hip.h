/* jiminy_hip.h -- C ABI of the MI355X-native batched rigid-body dynamics library.
 *
 * This is the drop-in boundary for ONE hot path of duburcqa/jiminy: the per-step
 * physics (ABA forward dynamics with rotor armature + spring-damper contact
 * forces + SimpleMotor law + explicit Euler / RK4 step on the configuration
 * manifold + the RNEA-style "extra terms" + noiseless sensors), batched one robot
 * per wavefront lane on gfx950.
 *
 * What each entry point replaces in the reference (paths relative to the
 * reference tree, duburcqa/jiminy @ v1.8.12):
 *
 *   jm_model_create     Robot::pinocchioModel_ + motors/sensors/contact registries as seen
 *                       by the engine (core/src/robot/model.cc, robot.cc; built on the Python
 *                       side by jiminy_py/robot.py:518-860).  Binding replaced:
 *                       python/jiminy_pywrap/src/robot.cc (Robot.initialize, attach_motor, ...)
 *   jm_batch_create     Engine::add_robot + RobotState/StepperState allocation
 *                       (core/include/jiminy/core/engine/engine.h:134-156, 216-250;
 *                        python/jiminy_pywrap/src/engine.cc:600-606)
 *   jm_batch_bind       the numpy views on Eigen buffers (`DEF_READONLY` on RobotState::q...,
 *                       python/jiminy_pywrap/src/engine.cc:173-188): here the caller owns
 *                       the memory and lends device pointers
 *   jm_batch_set_options Engine::setOptions, hot-path subset (core/src/engine/engine.cc:2654-2795)
 *   jm_batch_start      Engine::start (core/src/engine/engine.cc:952-1533)
 *   jm_batch_stop       Engine::stop (core/src/engine/engine.cc:2419-2460)
 *   jm_batch_step       Engine::step fixed-step branch (core/src/engine/engine.cc:1724-2417)
 *                       = AbstractStepper::tryStep (core/src/stepper/abstract_stepper.cc:15-62)
 *                       + computeAllExtraTerms (engine.cc:800-915) + computeSensorMeasurements
 *   jm_batch_dynamics   Engine::computeRobotsDynamics (core/src/engine/engine.cc:3585-3708;
 *                       python binding `compute_robots_dynamics`, pywrap engine.cc:634-638)
 *   jm_batch_reset_lanes BaseJiminyEnv.reset state injection for a subset of the batch
 *                       (python/gym_jiminy/common/gym_jiminy/common/envs/generic.py:521-760)
 *   jm_last_error       JIMINY_THROW message text (core/include/jiminy/core/macros.h:82-86)
 *
 * Conventions: all batch arrays are structure-of-arrays `X[component][B]` (component-major,
 * lane-contiguous) of the batch scalar type (float64 or float32), living in device memory
 * owned by the caller.  Spatial vectors are [linear; angular], quaternions xyzw, free-flyer
 * velocity is expressed in the body frame (engine.cc:3591-3592).
 * All functions return 0 on success and a negative JM_E* code otherwise; they never throw.
 * Numerical failures are per lane: see the `status` field.
 */
#ifndef JIMINY_HIP_H
#define JIMINY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (mapped by the Python facade to the exception classes of
 *      python/jiminy_pywrap/src/module.cc:96-103) */
#define JM_OK 0
#define JM_EINVAL (-1)      /* std::invalid_argument  -> ValueError   */
#define JM_ERUNTIME (-2)    /* std::runtime_error     -> RuntimeError */
#define JM_ECONTROLFLOW (-3)/* jiminy::bad_control_flow -> BadControlFlow */
#define JM_ELOOKUP (-4)     /* jiminy::lookup_error   -> LookupError  */
#define JM_ENOTIMPL (-5)    /* jiminy::not_implemented_error -> NotImplementedError */
#define JM_ETOPOLOGY (-6)   /* model topology does not match the compiled-in specialisation */

/* ---- joint type codes (core/include/jiminy/core/fwd.h:84-96 groups them as
 *      LINEAR / ROTARY / ROTARY_UNBOUNDED / FREE; the axis variant is explicit here) */
enum {
    JM_JT_NONE = 0,
    JM_JT_RX = 1, JM_JT_RY = 2, JM_JT_RZ = 3, JM_JT_RU = 4,
    JM_JT_PX = 5, JM_JT_PY = 6, JM_JT_PZ = 7, JM_JT_PU = 8,
    JM_JT_RUBX = 9, JM_JT_RUBY = 10, JM_JT_RUBZ = 11, JM_JT_RUBU = 12,
    JM_JT_FREEFLYER = 13,
    /* spherical joint (nq = 4: unit quaternion x y z w, nv = 3: angular velocity in the joint frame): the reference inserts
     * them as FLEXIBILITY joints (Model::addFlexibilityJointsToExtendedModel, core/src/robot/model.cc:1087-1165) and
     * applies a spring-damper on their rotation (Engine::computeInternalDynamics, core/src/engine/engine.cc:3365-3391).
     * One-robot-per-lane kernels. */
    JM_JT_SPHERICAL = 14
};

/* ---- scalar type of a batch */
enum { JM_F64 = 0, JM_F32 = 1 };

/* ---- ODE solvers (core/include/jiminy/core/engine/engine.h:30-38 `odeSolver`) */
enum { JM_SOLVER_EULER_EXPLICIT = 0, JM_SOLVER_RUNGE_KUTTA_4 = 1, JM_SOLVER_RUNGE_KUTTA_DOPRI = 2 };

/* ---- motor flags */
enum { JM_MOTOR_EFFORT_LIMIT = 1, JM_MOTOR_VELOCITY_LIMIT = 2, JM_MOTOR_FRICTION = 4 };

/* number of doubles per motor in `motor_params`:
 * reduction, effortLimit, velocityLimit, velocityEffortInvSlope,
 * frictionViscousPositive, frictionViscousNegative, frictionDryPositive,
 * frictionDryNegative, frictionDrySlope  (core/src/hardware/basic_motors.cc:83-143) */
#define JM_MOTOR_NPARAMS 9

/* ---- per-lane status bits written by the kernels */
enum {
    JM_LANE_OK = 0,
    JM_LANE_NAN = 1,            /* NaN in q/v/a (engine.cc:1737-1747, abstract_stepper.cc:41-48) */
    JM_LANE_OUT_OF_BOUNDS = 2,  /* a bounded joint left [lower, upper]: the reference would switch
                                   to its constraint solver (engine.cc:3285-3298, 3722) */
    JM_LANE_FORCE_OVERFLOW = 4, /* initial contact force > 1e5 N (engine.cc:1338-1345) */
    JM_LANE_STEPPER_FAILURE = 8,/* adaptive stepper: step size below 1e-10 s or too many successive
                                   failed iterations (engine.cc:2340-2384 raises for the robot) */
    JM_LANE_SOLVER_FAILURE = 16 /* constraint model: the PGS solver hit its iteration cap on the last
                                   evaluation (engine.cc:3755-3768 counts `successiveSolveFailed`) */
};

/* ---- model description: plain arrays, all host memory, copied by jm_model_create.
 * Joint 0 is the universe. Matrices are row-major 3x3. */
/* kinds of user constraint frames (jm_model_desc::cframe_kind) */
enum { JM_XKIND_FRAME = 0, JM_XKIND_SPHERE = 1, JM_XKIND_WHEEL = 2, JM_XKIND_DISTANCE = 3 };

typedef struct jm_model_desc {
    int32_t njoints, nq, nv;
    int32_t nmotors, ncontacts;
    int32_t nimu, nforce, ncontact_sensors, nencoder, neffort;
    const int32_t * parents;       /* [njoints] */
    const int32_t * jtypes;        /* [njoints] JM_JT_* */
    const int32_t * idx_q;         /* [njoints] */
    const int32_t * idx_v;         /* [njoints] */
    const double * axes;           /* [njoints*3] unit axis (unaligned joints) */
    const double * placement_R;    /* [njoints*9] joint placement wrt parent joint */
    const double * placement_p;    /* [njoints*3] */
    const double * mass;           /* [njoints] */
    const double * com;            /* [njoints*3] */
    const double * inertia;        /* [njoints*9] rotational inertia about the COM */
    const double * rotor_inertia;  /* [nv] armature on joint side */
    const double * position_lower; /* [nq] */
    const double * position_upper; /* [nq] */
    const int32_t * motor_joint;   /* [nmotors] */
    const int32_t * motor_flags;   /* [nmotors] JM_MOTOR_* */
    const double * motor_params;   /* [nmotors*JM_MOTOR_NPARAMS] */
    const int32_t * contact_joint; /* [ncontacts] parent joint of the contact frame */
    const double * contact_R;      /* [ncontacts*9] frame placement in the joint frame */
    const double * contact_p;      /* [ncontacts*3] */
    const int32_t * imu_joint;     /* [nimu] */
    const double * imu_R;          /* [nimu*9] */
    const double * imu_p;          /* [nimu*3] */
    const int32_t * force_joint;   /* [nforce] */
    const double * force_R;        /* [nforce*9] */
    const double * force_p;        /* [nforce*3] */
    const int32_t * contact_sensor_contact; /* [ncontact_sensors] index into contacts */
    const int32_t * encoder_joint;          /* [nencoder] */
    const int32_t * encoder_joint_side;     /* [nencoder] 1 = joint side */
    const double * encoder_reduction;       /* [nencoder] */
    const int32_t * effort_motor;           /* [neffort] motor index */
    /* ABI 7 -- frames a user-registered FrameConstraint may hold (`robot.add_constraint(name, FrameConstraint(frame,
     * maskDoFs))`, core/src/constraints/frame_constraint.cc:27-35, python/jiminy_pywrap/src/constraints.cc): part of the
     * TOPOLOGY like the contact points (the kernels are specialised on parent joint and mask).  Bit d of the mask = dof d
     * of (x, y, z, rot x, rot y, rot z), world aligned, is fixed.  One-robot-per-lane kernels. */
    int32_t n_constraint_frames;
    const int32_t * cframe_joint;  /* [n_constraint_frames] parent joint of the frame */
    const int32_t * cframe_mask;   /* [n_constraint_frames] 6-bit mask of the fixed dofs */
    const double * cframe_R;       /* [n_constraint_frames*9] frame placement in the joint frame */
    const double * cframe_p;       /* [n_constraint_frames*3] */
    /* kind of every constraint frame (JM_XKIND_*), second parent joint (DistanceConstraint, else 0) and 8 parameters:
     * SphereConstraint (sphere_constraint.cc) radius, normal[3]; WheelConstraint (wheel_constraint.cc) radius, normal[3],
     * axis[3] (wheel axis in the frame); DistanceConstraint (distance_constraint.cc) -, position[3] of the second frame
     * in ITS parent joint.  Masks: frame = the user's, sphere / wheel = 0b000111 (three rows at the contact point),
     * distance = 0b000001 (one row). */
    const int32_t * cframe_kind;   /* [n_constraint_frames] */
    const int32_t * cframe_joint2; /* [n_constraint_frames] */
    const double * cframe_params;  /* [n_constraint_frames*8] */
    /* ... and the 1-dof joints (revolute / prismatic) a user-registered JointConstraint may hold on a row of its own, next
     * to the joint's bound constraint like in the reference (`robot.add_constraint(name, JointConstraint(joint))`,
     * core/src/constraints/joint_constraint.cc, model.cc:884-905).  One-robot-per-lane kernels; the branch-parallel kernels
     * keep the flag-bit-2 form on the joint's bound row (jm_batch_set_joint_locks). */
    int32_t n_constraint_joints;
    const int32_t * cjoint_joint;  /* [n_constraint_joints] */
    /* ABI 8.  Flexibility of the spherical joints (`model_options["dynamics"]["flexibilityConfig"]`: stiffness, damping per
     * axis; the `inertia` entry is the joint's rotor inertia): u_internal -= Jlog3(q) (stiffness * log3(q)) + damping * w,
     * engine.cc:3377-3390.  [3 * njoints], rows of the other joints ignored; NULL when the model has no spherical joint. */
    const double * flex_stiffness;
    const double * flex_damping;
} jm_model_desc;

/* ---- hot-path subset of the engine options, same names/defaults as the reference
 * (core/include/jiminy/core/engine/engine.h:273-325) */
typedef struct jm_options {
    double gravity[6];                 /* world.gravity, default (0,0,-9.81,0,0,0) */
    double contact_stiffness;          /* contacts.stiffness          1e6  */
    double contact_damping;            /* contacts.damping            2e3  */
    double contact_friction;           /* contacts.friction           1.0  */
    double contact_transition_eps;     /* contacts.transitionEps      1e-3 */
    double contact_transition_velocity;/* contacts.transitionVelocity 1e-2 */
} jm_options;

/* ---- bindable batch fields (jm_batch_bind).  Shapes are [rows][B]. */
enum {
    JM_F_Q = 0,            /* [nq]      in/out  RobotState::q            */
    JM_F_V = 1,            /* [nv]      in/out  RobotState::v            */
    JM_F_A = 2,            /* [nv]      in/out  RobotState::a            */
    JM_F_COMMAND = 3,      /* [nmotors] in      RobotState::command      */
    JM_F_U_MOTOR = 4,      /* [nmotors] out     RobotState::uMotor       */
    JM_F_U = 5,            /* [nv]      out     RobotState::u            */
    JM_F_F_EXTERNAL = 6,   /* [njoints*6] out   RobotState::fExternal (joint frame)   optional */
    JM_F_CONTACT_FORCES = 7,/* [ncontacts*6] out Robot::contactForces_ (contact frame) optional */
    JM_F_IMU = 8,          /* [nimu*6]   out  gyro(3), accel(3)  (basic_sensors.cc:142-164) */
    JM_F_FORCE = 9,        /* [nforce*6] out  (basic_sensors.cc:368-387) */
    JM_F_CONTACT = 10,     /* [ncontact_sensors*3] out (basic_sensors.cc:267-277) */
    JM_F_ENCODER = 11,     /* [nencoder*2] out Q,V (basic_sensors.cc:509-539) */
    JM_F_EFFORT = 12,      /* [neffort]  out  (basic_sensors.cc:604-618) */
    JM_F_ENERGY = 13,      /* [2] out kinetic, potential (engine.cc:808-810)       optional */
    JM_F_JOINT_FORCES = 14,/* [njoints*6] out data.f joint internal wrenches (engine.cc:878-887) optional */
    JM_F_CENTROIDAL = 15,  /* [15] out com(3), hg(6), dhg(6) (engine.cc:889-904)    optional */
    JM_F_STATUS = 16,      /* [1] int32 per lane, JM_LANE_* bits */
    JM_F_WORKSPACE = 17,   /* scratch, jm_batch_workspace_rows() rows */
    JM_F_CON_FLAGS = 18,   /* [n_flag_rows] int32 in/out: per constraint, bit 0 = enabled, bit 1 = reversed
                              (AbstractConstraintBase::isEnabled_, JointConstraint::isReversed_); joint rows, bit 2
                              (set by the caller, kept by the library): a user-registered JointConstraint holds the
                              joint (Model::addConstraint, core/src/robot/model.cc:926-936) -- the row is bilateral,
                              always enabled, its reference configuration taken at jm_batch_start; branch-parallel
                              topologies */
    JM_F_CON_DATA = 19,    /* [n_data_rows] in/out: JointConstraint::configurationRef_ per bounded joint, then
                              the Lagrange multipliers `lambda_` of every constraint row (PGS warm start: bounds, 4 per
                              contact point, 6 per user constraint frame), then FrameConstraint::transformRef_ of every
                              user constraint frame (translation 3, rotation 9 row-major; taken at jm_batch_start, the
                              caller may overwrite it: `constraint.reference_transform`).  JM_F_CON_FLAGS carries one
                              more row per user constraint frame after the contacts (bit 0, set by the caller: the
                              lane's robot holds that FrameConstraint), then one per user constraint joint; the latter
                              add one multiplier row each behind the frames' and one reference-configuration row each
                              behind the frames' reference transforms */
    JM_F_FRICTION = 20,    /* [1] in, optional: contacts.friction of every lane (domain randomisation of the ground
                              friction, gym_jiminy envs/locomotion.py:257-262); both contact models, every topology
                              (spring-damper on the one-robot-per-lane kernels: ABI 9), unbound = the batch-wide
                              contacts.friction option */
    JM_F_MODEL_LANE = 21,  /* [13 * njoints] in, optional: body parameters of every lane, rows per joint
                              mass | com 3 | inertia xx xy xz yy yz zz | joint placement translation 3 -- the
                              output of Model::addBiasedToExtendedModel (core/src/robot/model.cc:1166-1236: mass,
                              centre of mass, inertia and relative body position biases), one model per
                              environment; float64, every topology (the one-robot-per-lane kernels: ABI 9);
                              unbound = the model's own */
    JM_F_APPLIED = 22,     /* [6 * K] in, optional: world-aligned (force, moment) applied at K <= 4 frames of any
                              joint (jm_batch_set_applied_frames), i.e. the current value of the impulse /
                              profile forces of core/src/engine/engine.cc:1838-2016 (the caller owns their time
                              schedule and cuts the launches at their breakpoints); every topology (the
                              one-robot-per-lane kernels: ABI 9) */
    JM_F_GROUND_OFFSET = 23, /* [2] in, optional: (x, y) added to the world position at which every lane samples the height map
                              of jm_batch_set_ground -- every environment its own patch of one large terrain, the batched
                              form of one `world.groundProfile` per environment instance (gym_jiminy: a new random
                              ground per episode); unbound = (0, 0) */
    JM_F_FLEXIBILITY = 24, /* [6 * nspherical] in, optional (ABI 9): stiffness 3, damping 3 of every spherical (flexibility) joint, in
                              joint order, per lane -- `flexibilityConfig` randomised per environment the way
                              WalkerJiminyEnv._setup does per episode (gym_jiminy envs/locomotion.py:288-296); read by
                              Engine::computeInternalDynamics' flexibility efforts (core/src/engine/engine.cc:3365-3391);
                              unbound = jm_model_desc::flex_stiffness / flex_damping */
    JM_F_COUNT = 25
};

/* ---- `contacts.model = "constraint"` (the reference's default contact model, engine.h:273) and the
 * joint position bounds it enforces: options of the boxed forward dynamics
 * (core/src/solver/constraint_solvers.cc:335-448, core/src/engine/engine.cc:3253-3338, 3710-3866).
 * Constraint rows of one robot, in the reference's registry order (robot/model.h:40-46):
 *   one row per bounded 1-dof joint (`JointConstraint`, model joint order; continuous joints never
 *   activate and own no row), then 4 rows per contact point (`FrameConstraint` with the translation
 *   and the rotation about the ground normal fixed: x, y, z, torsion; core/src/robot/model.cc:817-823).
 * Float64 batches only.  With the adaptive stepper the constraint state of the active lanes travels with
 * them (gathered / scattered around every attempt); call jm_batch_set_constraint_options before
 * jm_batch_adaptive_workspace_rows, whose result depends on the contact model. */
enum { JM_CONTACT_SPRING_DAMPER = 0, JM_CONTACT_CONSTRAINT = 1 };
typedef struct jm_constraint_options {
    int32_t contact_model;      /* contacts.model: JM_CONTACT_*                      */
    int32_t pgs_iter_max;       /* PGS_MAX_ITERATIONS = 100 (engine.cc:62)            */
    double torsion;             /* contacts.torsion            0.0                    */
    double stabilization_freq;  /* contacts.stabilizationFreq  20.0 (Baumgarte)       */
    double regularization;      /* constraints.regularization  1e-3                   */
    double tol_abs;             /* stepper.tolAbs 1e-5: PGS tolerances (engine.cc:1372-1373) */
    double tol_rel;             /* stepper.tolRel 1e-4                                */
    /* ABI 6: Baumgarte frequency of the user-registered constraints (jm_batch_set_joint_locks).  In the reference they keep
     * gains of their own -- zero until `setBaumgarteFreq` is called on them -- and Engine::start only overwrites those of the
     * internal constraints (abstract_constraint.cc:88-98, engine.cc:1276-1285).  < 0: the gains of `stabilization_freq`
     * (what ABI 5 did); >= 0: critically damped gains of this frequency, 0 = a pure acceleration constraint. */
    double user_stabilization_freq;
} jm_constraint_options;

typedef struct jm_model jm_model;
typedef struct jm_batch jm_batch;

/* Topology signature the library was specialised for (see DESIGN.md "model compilation"). */
const char * jm_topology_signature(void);
/* ABI version of this header. */
int32_t jm_abi_version(void);

int32_t jm_model_create(const jm_model_desc * desc, jm_model ** out);
int32_t jm_model_destroy(jm_model * model);

/* `device` is the HIP device ordinal the batch lives on. */
int32_t jm_batch_create(const jm_model * model, int64_t batch_size, int32_t dtype,
                        int32_t device, jm_batch ** out);
int32_t jm_batch_destroy(jm_batch * batch);
int32_t jm_batch_set_options(jm_batch * batch, const jm_options * options);
/* Number of [B]-rows of scratch the caller must provide through JM_F_WORKSPACE. */
int32_t jm_batch_workspace_rows(const jm_batch * batch);
/* Constraint contact model: options (not while a simulation is running) and the row counts of the
 * per-lane constraint state the caller lends through JM_F_CON_FLAGS (int32), JM_F_CON_DATA and
 * JM_F_WORKSPACE (batch dtype; the delassus matrix J M^-1 J^T of every lane and the PGS vectors).
 * With JM_CONTACT_CONSTRAINT, start / step / dynamics / reset_lanes run the constrained evaluation:
 * joint-bound and contact constraints are switched with the reference's hysteresis
 * (engine.cc:3285-3298, 3145-3193) and the multipliers solved by projected Gauss-Seidel
 * (constraint_solvers.cc:107-333) on every dynamics evaluation. */
int32_t jm_batch_set_constraint_options(jm_batch * batch, const jm_constraint_options * options);
int32_t jm_batch_constraint_rows(const jm_batch * batch, int32_t * n_flag_rows, int32_t * n_data_rows,
                                 int32_t * n_workspace_rows);
/* world.groundProfile (core/include/jiminy/core/engine/engine.h:292-302, used by
 * computeContactDynamicsAtFrame, core/src/engine/engine.cc:3138-3145) as a height map: `heights` is a device
 * array `[ny][nx]` in the batch dtype, sampled at (x0 + ix dx, y0 + iy dy) with bilinear patches (height and
 * unit normal), flat continuation outside the grid; NULL = flat ground at z = 0.  Both contact models (the rows of
 * a contact constraint live in the local frame of the surface, FrameConstraint::setNormal), branch-parallel topologies. */
int32_t jm_batch_set_ground(jm_batch * batch, const void * heights, int32_t nx, int32_t ny, double x0, double y0,
                            double dx, double dy);
/* User-registered JointConstraints (`Model::addConstraint(name, JointConstraint)`, core/src/robot/model.cc:926-936;
 * python/jiminy_pywrap/src/robot.cc:215 `add_constraint`): tell the batch that joint rows of JM_F_CON_FLAGS may carry
 * bit 2, so that its launches take kernels built with the unbounded rows (the plain kernels of robots with
 * register-resident solves ignore the bit).  Branch-parallel topologies, constraint contact model; 0 = none any more. */
int32_t jm_batch_set_joint_locks(jm_batch * batch, int32_t on);

/* Frames that carry the JM_F_APPLIED wrenches (`Engine::registerImpulseForce` / `registerProfileForce`,
 * core/src/engine/engine.cc:1838-1935, accept any frame of the model): `offsets` = K x 3 frame positions in the frame of
 * their parent joint, `joints` = the K parent joint indices (NULL = all on the root joint); K <= 4, K = 0 disables.
 * Every dynamics evaluation adds the wrench to `fext[parent joint]` in the joint frame, like
 * `Engine::computeExternalForces` (engine.cc:3481-3560) through `convertForceGlobalFrameToJoint`. */
int32_t jm_batch_set_applied_frames(jm_batch * batch, int32_t k, const double * offsets, const int32_t * joints);
/* Lend a device pointer for one field; NULL unbinds an optional output. */
int32_t jm_batch_bind(jm_batch * batch, int32_t field, void * device_ptr);

/* Engine::start: from bound (q, v, command) compute a, extra terms and sensors;
 * checks the initial contact forces. `stream` is a hipStream_t (NULL = default). */
int32_t jm_batch_start(jm_batch * batch, void * stream);
/* Engine::stop (core/src/engine/engine.cc:2419-2460): the model may be re-configured again. */
int32_t jm_batch_stop(jm_batch * batch);
/* Engine::step fixed-step branch: `n_substeps` integrator steps of `dt` with the command
 * held. `command_changed` != 0 re-evaluates a(t+) with the current command before the
 * first sub-step (engine.cc:2030-2042); `update_sensors` != 0 refreshes the sensor outputs
 * at the end (engine.cc:2386-2410). */
int32_t jm_batch_step(jm_batch * batch, int32_t solver, double dt, int32_t n_substeps,
                      int32_t command_changed, int32_t update_sensors, void * stream);
/* Engine::computeRobotsDynamics: a = f(q, v) with the bound command; q_in/v_in are device
 * arrays [nq][B] / [nv][B]; a_out [nv][B]. Does not modify the bound state. */
int32_t jm_batch_dynamics(jm_batch * batch, const void * q_in, const void * v_in,
                          void * a_out, void * stream);
/* Re-initialise the lanes whose mask byte is non-zero from (q_init, v_init) ([nq][B], [nv][B]):
 * copies the state, zeroes a, then performs the `start` computation for those lanes only. */
int32_t jm_batch_reset_lanes(jm_batch * batch, const uint8_t * lane_mask,
                             const void * q_init, const void * v_init, void * stream);

/* Per-launch kernel timing with HIP events recorded on the launch stream around every kernel
 * `jm_batch_step` launch of this batch (start / reset / dynamics launches are not timed; up to 2048 launches between two summaries).  `jm_batch_timing_summary`
 * blocks until the recorded launches completed, returns their count and summed duration (ms)
 * and restarts the recording. */
int32_t jm_batch_enable_timing(jm_batch * batch, int32_t enable);
int32_t jm_batch_timing_summary(jm_batch * batch, int32_t * n_launches, double * total_ms);

/* ---- adaptive stepping: `odeSolver = "runge_kutta_dopri"`, the reference's default
 * (core/include/jiminy/core/stepper/runge_kutta_dopri_stepper.h, core/src/stepper/runge_kutta_dopri_stepper.cc,
 *  step-size selection of core/src/engine/engine.cc:2021-2222).  Every lane carries its own step size.
 * The caller lends (jm_batch_bind_adaptive):
 *   workspace  [jm_batch_adaptive_workspace_rows()][B] scalars of the batch dtype (stage derivatives),
 *   state_f64  [5][B] float64: t, dt, dtLargest, dtLargestPrev, (scratch) -- initialise t = 0 and the
 *              three step sizes to 1e-6 (StepperState::reset, engine.h:219-236, engine.cc:1176),
 *   state_i32  [7][B] int32: iter, iterFailed, successiveIterTooLarge, successiveIterFailed, (2 scratch rows,
 *              then the list of the lanes still active in the current attempt) = 0.
 * jm_batch_step_adaptive advances every lane from its `t` to the breakpoint `t_next` (the caller splits
 * Engine::step at controller / sensor breakpoints exactly as for the fixed-step solvers), then refreshes
 * the extra terms and, if asked, the sensors.  `new_step` != 0 on the first interval of an Engine::step
 * call (resets the successive-failure counters, clears the status row).  Lanes whose step size falls
 * below 1e-10 s or that fail more than `successive_iter_failed_max` times in a row get
 * JM_LANE_STEPPER_FAILURE (the reference raises for its single robot, engine.cc:2340-2384).
 * The call synchronises the stream once per attempt (active-lane count); `attempts_out` (optional)
 * receives the number of attempts, `max_attempts` bounds it. */
typedef struct jm_adaptive_options
{
    double tol_rel;                     /* stepper.tolRel  (1e-4) */
    double tol_abs;                     /* stepper.tolAbs  (1e-5) */
    double dt_max;                      /* stepper.dtMax */
    double dt_restore_threshold_rel;    /* stepper.dtRestoreThresholdRel (0.2) */
    int32_t successive_iter_failed_max; /* stepper.successiveIterFailedMax (1000) */
    int32_t form;                       /* 0: one persistent launch per interval where the topology has that kernel
                                           (jm_qdopri.h), 1: always the per-stage launches over compacted lanes */
} jm_adaptive_options;
int32_t jm_batch_adaptive_workspace_rows(const jm_batch * batch);
int32_t jm_batch_bind_adaptive(jm_batch * batch, void * workspace, double * state_f64, int32_t * state_i32);
int32_t jm_batch_step_adaptive(jm_batch * batch, double t_next, const jm_adaptive_options * options,
                               int32_t new_step, int32_t command_changed, int32_t update_sensors,
                               int32_t max_attempts, int32_t * attempts_out, void * stream);

/* ---- gym_jiminy pipeline blocks, batched: one lane = one environment.
 * Topology independent; arrays are device pointers in the dtype of the call, `[rows][B]`.
 *
 * jm_block_pd_controller ≙ `pd_controller` + `integrate_zoh`
 *   (python/gym_jiminy/common/gym_jiminy/common/blocks/proportional_derivative_controller.py:23-163):
 *   advances the command state (position, velocity, acceleration targets, `[3][M][B]`, in place) by
 *   one controller period under its bounds and writes the clipped PD torques `[M][B]`.
 *   `encoder` is the raw JM_F_ENCODER field (`[n_enc][2][B]`), `encoder_index[m]` the encoder of
 *   motor m; `lower` / `upper` are `[3][M]`, `kp`, `kd`, `effort_limit` `[M]` host arrays.
 * jm_block_mahony_filter ≙ `mahony_filter` (blocks/mahony_filter.py:28-101): `imu` is the raw
 *   JM_F_IMU field (`[n_imu][6][B]`), `quat` `[4][n_imu][B]` (xyzw) and `bias` `[3][n_imu][B]`
 *   are updated in place, `omega` / `cf` `[3][n_imu][B]` receive the de-biased rates. */
#define JM_BLOCK_MAX_MOTORS 40
int32_t jm_block_pd_controller(int32_t dtype, int64_t batch_size, int32_t nmotors, const void * encoder,
                               const int32_t * encoder_index, void * command_state,
                               const double * lower, const double * upper, const double * kp,
                               const double * kd, const double * effort_limit, double control_dt,
                               void * out_torque, void * stream);
int32_t jm_block_mahony_filter(int32_t dtype, int64_t batch_size, int32_t n_imu, const void * imu,
                               void * quat, void * omega, void * cf, void * bias, double kp, double ki,
                               double dt, void * stream);
/* jm_block_pd_adapter ≙ `pd_adapter` (proportional_derivative_controller.py:166-260), the `PDAdapter` block: from
 *   the action `[M][B]` (target motor position, `order` 0, or velocity, `order` 1) to the target acceleration
 *   `out` `[M][B]` held over `step_dt` (or, `is_instantaneous`, the command state `[3][M][B]` moved at once and a
 *   zero acceleration); `lower` / `upper` `[3][M]`, `velocity_deadband` `[M]` or NULL (host arrays).
 * jm_block_motor_safety_limit ≙ `apply_safety_limits` (blocks/motor_safety_limit.py:20-77), the `MotorSafetyLimit`
 *   block: clips the command torques `[M][B]` so that they act against soft position bounds / the velocity limit;
 *   `encoder` is the raw JM_F_ENCODER field, `encoder_index[m]` the encoder of motor m, the gains and limits are
 *   host arrays `[M]`. */
int32_t jm_block_pd_adapter(int32_t dtype, int64_t batch_size, int32_t nmotors, const void * action, int32_t order,
                            void * command_state, const double * lower, const double * upper,
                            int32_t is_instantaneous, const double * velocity_deadband, double step_dt,
                            void * out_acceleration, void * stream);
int32_t jm_block_motor_safety_limit(int32_t dtype, int64_t batch_size, int32_t nmotors, const void * encoder,
                                    const int32_t * encoder_index, const void * command, const double * kp,
                                    const double * kd, const double * soft_position_lower,
                                    const double * soft_position_upper, const double * velocity_limit,
                                    const double * effort_limit, void * out_command, void * stream);

/* ---- Sensor white noise and bias (sensor part), batched.
 *
 * jm_block_sensor_noise ≙ `AbstractSensorBase::measureData` (core/src/hardware/abstract_sensor.cc:71-85)
 *   and, with `rot_bias_inv`, `ImuSensor::measureData` (core/src/hardware/basic_sensors.cc:166-187),
 *   applied in place to one raw measurement field (`data`, device, `[n_sensors][n_fields][B]`, e.g.
 *   JM_F_IMU) right after the step that refreshed it: white noise `normal(generator, 0, noiseStd)`
 *   first (float ziggurat over a PCG32 stream, core/src/utilities/random.cc:10-167), then the additive
 *   bias, then (IMU) the rotation bias applied to both 3-vectors. `rng_state` (device, uint64
 *   `[n_sensors][B]`) is the PCG32 state of every (sensor, lane), advanced exactly as the reference
 *   advances `AbstractSensorBase::generator_`. `noise_std`, `bias` are host arrays
 *   `[n_sensors][n_fields]` (NULL = option left empty), `rot_bias_inv` a host array `[n_sensors][9]`
 *   (row-major `exp3(-bias.head<3>())`, basic_sensors.cc:121-129) or NULL.
 * jm_sensor_rng_seed ≙ the generator seeding of `AbstractSensorTpl<T>::resetAll(seed)`
 *   (core/include/jiminy/core/hardware/abstract_sensor.hxx:213-226): for every lane,
 *   `std::seed_seq{group_seed[lane]}.generate` of `n_sensors` words, sensor s gets the PCG32 state
 *   `word_s | 3`. Host arrays in, host array `[n_sensors][B]` out (upload it as `rng_state`). */
#define JM_NOISE_MAX_ROWS 128
#define JM_NOISE_MAX_FIELDS 6
#define JM_NOISE_MAX_ROT 8
int32_t jm_block_sensor_noise(int32_t dtype, int64_t batch_size, int32_t n_sensors, int32_t n_fields,
                              void * data, uint64_t * rng_state, const double * noise_std,
                              const double * bias, const double * rot_bias_inv, void * stream);
int32_t jm_sensor_rng_seed(const uint32_t * group_seed, int64_t batch_size, int32_t n_sensors,
                           uint64_t * state_out);

/* ---- Model biases per environment (model part), batched, drawn on the device.
 *
 * jm_block_model_bias ≙ `Model::addBiasedToExtendedModel(g)` (core/src/robot/model.cc:1166-1236) for every
 *   lane: for the mechanical joints in index order (`first_joint`..njoints-1: the free-flyer root is not one of
 *   them, model.cc:337-341) and in the reference's field order -- centre of mass (3 normals), mass (1), inertia
 *   (3 for the rotation vector of the principal axes, 3 for the principal moments), joint placement translation
 *   (3), each group only when its standard deviation is > EPS -- the lane's engine generator `rng_state[lane]`
 *   (device, uint64 `[B]`: the PCG32 stream of `Engine::generator_`) is advanced exactly as the reference advances
 *   it, the float normals are `normal(g, mean, std)` of core/src/utilities/random.cc:52-167, and the biased body
 *   parameters are written to the rows of `model_lane` (device, `[13 * njoints][B]`, JM_F_MODEL_LANE layout).
 *   `nominal` (device, float64 `[njoints][25]`): mass | com 3 | inertia xx xy xz yy yz zz | placement translation 3
 *   | principal moments 3 (ascending) | principal axes 9 (row-major, columns = axes) of the unbiased model.
 *   `std4` (host): inertia, mass, centre of mass, relative position standard deviations.  `mask` (device, uint8
 *   `[B]`) or NULL: only the masked lanes draw (episode-wise re-randomisation of the lanes being reset).
 * jm_engine_rng_seed ≙ `generator_.seed(std::seed_seq(randomSeedSeq))` (core/src/engine/engine.cc:756-757,
 *   random.hxx:20-51) with one seed word per lane: PCG32 state `(w0 | w1 << 32) | 3` of the two words
 *   `std::seed_seq{seed[lane]}` generates.  Host arrays. */
int32_t jm_block_model_bias(int32_t dtype, int64_t batch_size, int32_t njoints, int32_t first_joint,
                            const double * nominal, const float * std4, uint64_t * rng_state,
                            const uint8_t * mask, void * model_lane, void * stream);
int32_t jm_engine_rng_seed(const uint32_t * seed, int64_t batch_size, uint64_t * state_out);

/* ---- Sensor delay and jitter, batched.
 * jm_block_sensor_delay ≙ `AbstractSensorTpl<T>::interpolateData`
 *   (core/include/jiminy/core/hardware/abstract_sensor.hxx:305-429), the first half of `measureDataAll`: call it
 *   after every sensor refresh, before jm_block_sensor_noise.  `data` (device, `[n_sensors][n_fields][B]`) is
 *   overwritten with the measurement delayed by `delay[s] + uniform(0, jitter[s])` read from the history ring
 *   `history` (device, `[slots][n_sensors * n_fields][B]`, raw measurements the caller stored after each
 *   refresh, the current one included): zero-order hold (`order` 0) or linear interpolation (1), the oldest
 *   sample while the ring does not reach back far enough.  `slot[i]` / `times[i]` (host, i < n_history <= 64,
 *   ascending times, the last one the current time) name the ring slot and the time of the i-th oldest sample.
 *   The uniform number is drawn from `rng_state` (the generators of jm_block_sensor_noise) on every call,
 *   whether or not a jitter is configured, as the reference does; `history` NULL only takes that draw. */
#define JM_DELAY_MAX_HISTORY 64
int32_t jm_block_sensor_delay(int32_t dtype, int64_t batch_size, int32_t n_sensors, int32_t n_fields, void * data,
                              const void * history, const int32_t * slot, const double * times, int32_t n_history,
                              uint64_t * rng_state, const double * delay, const double * jitter,
                              int32_t interpolation_order, void * stream);

/* Copy the message of the last error raised on the calling thread. */
int32_t jm_last_error(char * buffer, size_t size);

#ifdef __cplusplus
}
#endif
#endif /* JIMINY_HIP_H */
