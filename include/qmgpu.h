/*
 * qmgpu.h -- C ABI of the MI355X-native MPC + whole-body-control hot path for the AlienGo+Z1
 *            quadruped manipulator (drop-in for the numeric path of danisotelo/qm_door).
 *
 * Nothing like this interface exists in the reference (it has no FFI); each entry point below
 * names the reference interface it replaces (paths relative to the reference tree):
 *
 *   qmgpu_load_problem      <-> qm::QMInterface ctor + setupOptimalControlProblem
 *                               (qm_interface/src/QMInterface.cpp:37-142) and
 *                               WbcBase::loadTasksSetting (qm_wbc/src/WbcBase.cpp:597-627)
 *   qmgpu_create/destroy    <-> QMController::setupMpc / setupWbc
 *                               (qm_controllers/src/QMController.cpp:273-277, 287-307)
 *   qmgpu_mpc_solve_batch   <-> ocs2::MPC_BASE::run -> SqpSolver::runImpl, one SQP iteration
 *                               (object built at qm_controllers/src/QMController.cpp:288-289)
 *   qmgpu_policy_eval_batch <-> MPC_MRT_Interface::evaluatePolicy (QMController.cpp:134-142)
 *   qmgpu_warm_start_batch  <-> upstream SqpSolver::runImpl's initial guess from the previous PrimalSolution
 *                               (the solver object built at QMController.cpp:288-289 keeps it between runs)
 *   qmgpu_wbc_solve_batch   <-> qm::WbcBase::update / HierarchicalWbc::update
 *                               (qm_wbc/include/qm_wbc/WbcBase.h:31-34, qm_wbc/src/HierarchicalWbc.cpp:18-44)
 *   qmgpu_cycle_batch       <-> one QMController::update tick fed by one advanceMpc
 *                               (QMController.cpp:129-176, 316-327) for a batch of robots
 *
 * Conventions
 *   - all floating point data is IEEE fp64, all matrices row-major unless stated otherwise
 *   - "dev" pointers are HIP device pointers (HBM resident); the *_host variants take host
 *     pointers and stage through pinned buffers
 *   - the caller owns every buffer; the library owns device scratch behind the opaque handle
 *   - no C++ exception crosses this boundary; every function returns a qmgpu_status
 *   - one handle == one HIP stream; calls on one handle must be serialised by the caller
 *
 * Vector layouts (reference: SURVEY.md Appendix A)
 *   state x[30]   = [ h_lin/m (3), h_ang/m (3) ; p_base (3), yaw, pitch, roll ; q_joint (18) ]
 *   input u[30]   = [ f_LF, f_RF, f_LH, f_RH (12) ; v_joint (18) ]
 *   joints (18)   = LF(HAA,HFE,KFE), LH, RF, RH, z1_joint_1..6          (Pinocchio order)
 *   target[37]    = [ x_ref (30) ; p_ee (3) ; quat_ee (x,y,z,w) ]
 *   rbd[55]       = [ zyx (3), p (3), q_j (18) ; w_world (3), v_lin (3), dq_j (18) ; p_ee (3), quat_ee (4) ]
 *   wbc out[54]   = [ ddq (24), F (12) ; tau (18) ]
 *   mode          = 8*LF + 4*RF + 2*LH + 1*RH   (1 = stance)
 */
#ifndef QMGPU_H
#define QMGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QMGPU_NX 30
#define QMGPU_NU 30
#define QMGPU_NV 24
#define QMGPU_NJ 18
#define QMGPU_NB 19 /* moving bodies: floating base + 18 joint bodies */
#define QMGPU_NC 4  /* 3-DoF contacts, order LF RF LH RH */
#define QMGPU_NTARGET 37
#define QMGPU_NRBD 55
#define QMGPU_NWBC_DEC 36
#define QMGPU_NWBC_OUT 54
#define QMGPU_MAX_EVENTS 40 /* per-instance mode-schedule capacity */
#define QMGPU_NSTATS 10
#define QMGPU_WBC_STATE_WORDS 48 /* uint64 words per instance of qmgpu_wbc_args::working_set */
#define QMGPU_F32_MAX_TARGET_KNOTS 64 /* target knots per instance an fp32 handle (qmgpu_create_ex, QMGPU_F32) can stage; more -> QMGPU_ERR_CAPACITY */

typedef enum qmgpu_status {
  QMGPU_OK = 0,
  QMGPU_ERR_INVALID_ARGUMENT = 1,
  QMGPU_ERR_FILE_NOT_FOUND = 2, /* reference: std::invalid_argument, QMInterface.cpp:45,53,61 */
  QMGPU_ERR_PARSE = 3,
  QMGPU_ERR_UNSUPPORTED_MODEL = 4,
  QMGPU_ERR_NO_DEVICE = 5, /* no HIP device / kernels missing: the library never falls back to a CPU path */
  QMGPU_ERR_HIP = 6,
  QMGPU_ERR_CAPACITY = 7,
  QMGPU_ERR_NUMERICAL = 8
} qmgpu_status;

/* Flat rigid-body model.  Every joint origin of the reference URDF has rpy = 0 and every axis is a
 * positive unit coordinate axis; the loader rejects anything else. */
typedef struct qmgpu_model {
  int32_t parent[QMGPU_NB];          /* parent body, -1 for the base */
  int32_t axis[QMGPU_NB];            /* joint axis 0/1/2 = x/y/z (unused for body 0) */
  double joint_offset[QMGPU_NB][3];  /* joint origin in the parent body frame */
  double mass[QMGPU_NB];             /* fixed children (feet, imu, z1_link_0, gripper) merged in */
  double com[QMGPU_NB][3];           /* body frame */
  double inertia[QMGPU_NB][6];       /* about com, body axes: xx xy xz yy yz zz */
  int32_t foot_body[QMGPU_NC];
  double foot_offset[QMGPU_NC][3];
  int32_t ee_body;
  double ee_offset[3];
  double q_lower[QMGPU_NJ], q_upper[QMGPU_NJ], effort_limit[QMGPU_NJ], velocity_limit[QMGPU_NJ];
  double total_mass;
} qmgpu_model;

typedef struct qmgpu_settings {
  /* model_settings + swing_trajectory_config (task.info:8-31) */
  double position_error_gain, phase_transition_stance_time;
  double liftoff_velocity, touchdown_velocity, swing_height, touchdown_after_horizon, swing_time_scale;
  /* sqp + mpc (task.info:76-93,139-149); alpha_decay/alpha_min/gamma_c/armijo are OCS2 defaults */
  double dt, time_horizon, delta_tol, g_max, g_min, alpha_decay, alpha_min, gamma_c, armijo_factor;
  double cost_tol;                     /* sqp.costTol (upstream default 1e-4): convergence test between SQP iterations */
  int32_t sqp_iterations, reserved0;
  /* cost (task.info:193-288) */
  double initial_state[QMGPU_NX];
  double Q[QMGPU_NX * QMGPU_NX];
  double R_task[QMGPU_NU * QMGPU_NU]; /* as loaded; the leg-velocity block is mapped by J^T R J at create time */
  double ee_mu_position, ee_mu_orientation, ee_final_mu_position, ee_final_mu_orientation;
  /* soft constraints (task.info:291-344); cone regularization / hessian shift are OCS2 defaults */
  double friction_coefficient, friction_barrier_mu, friction_barrier_delta, friction_regularization, friction_hessian_shift;
  double joint_pos_barrier_mu, joint_pos_barrier_delta;
  double joint_vel_barrier_mu, joint_vel_barrier_delta;
  double arm_vel_lower[6], arm_vel_upper[6];
  /* reference.info */
  double com_height, default_joint_state[QMGPU_NJ], target_displacement_velocity, target_rotation_velocity;
  /* whole body control (task.info:347-350, qm_wbc/cfg/wbcWigeht.cfg:7-47) */
  double wbc_friction_coefficient;
  double kp_swing, kd_swing, kp_base_height, kd_base_height, kp_base_linear, kd_base_linear, kp_base_angular, kd_base_angular;
  double kp_arm_joint[6], kd_arm_joint[6];
  double kp_ee_linear[3], kd_ee_linear[3], kp_ee_angular[3], kd_ee_angular[3];
  double gravity; /* 9.81 */
  /* Force tracking (BASELINE.json configs[3]; OWN FORMULATION -- the reference's force-tracking branch is not in the mounted tree,
   * README.md:15,161,180 is all it says).  The arm end-effector touches a compliant environment (the door) anchored at p_env(t):
   *     f_e(x, t) = -K_e (p_ee(x) - p_env(t))                              force ON the end-effector, world axes
   * which (1) acts on the centroidal dynamics, d(h_lin)/dt += f_e, d(h_ang)/dt += (p_ee - p_com) x f_e, and (2) is held to a
   * reference by the soft constraint  1/2 mu_f |f_e - f_ref(t)|^2  on the intermediate nodes.  K_e = 0 or a NULL
   * qmgpu_mpc_args::ee_contact_ref switches both off (task.info keys forceTracking.stiffness / forceTracking.muForce). */
  double ee_contact_stiffness, ee_force_mu;
  /* DDP variant (qmgpu_mpc_args::algorithm = QMGPU_ALG_DDP): task.info ddp{} block (:34-72) -- lineSearch.minStepLength / maxStepLength,
   * constraintPenaltyInitialValue */
  double ddp_min_step, ddp_max_step, ddp_constraint_penalty;
} qmgpu_settings;

typedef struct qmgpu_problem {
  qmgpu_model model;
  qmgpu_settings settings;
} qmgpu_problem;

/* A gait template (gait.info) */
typedef struct qmgpu_gait {
  int32_t num_modes;
  int32_t modes[QMGPU_MAX_EVENTS];
  double switching_times[QMGPU_MAX_EVENTS + 1];
} qmgpu_gait;

typedef struct qmgpu_context* qmgpu_handle;

const char* qmgpu_strerror(int status);
/* Last error detail (thread local, valid until the next failing call on this thread). */
const char* qmgpu_last_error(void);

/* ---- host-side configuration (no GPU needed) -------------------------------------------------- */
int qmgpu_load_problem(const char* task_file, const char* urdf_file, const char* reference_file,
                       const char* wbc_gains_file /* may be NULL: compiled-in defaults */, qmgpu_problem* out);
int qmgpu_load_gait(const char* gait_file, const char* gait_name, qmgpu_gait* out);
int qmgpu_mode_from_string(const char* name); /* "LF_RH" -> 9, unknown -> -1 */
/* Tile a gait template over [t_begin, t_end] starting the first cycle at t_phase0, the way
 * ocs2::legged_robot::GaitSchedule extends its template; writes a mode schedule
 * (num_events event times, num_events + 1 modes). Returns QMGPU_ERR_CAPACITY if it does not fit. */
int qmgpu_tile_gait(const qmgpu_gait* gait, double t_phase0, double t_begin, double t_end,
                    int32_t* num_events, double* event_times /*[MAX_EVENTS]*/, int32_t* modes /*[MAX_EVENTS+1]*/);
/* The same for a gait command that arrives while mode `prev_mode` is running (upstream GaitSchedule::insertModeSequenceTemplate behind
 * GaitReceiver, fed by GaitTopicPublisher.cpp:31-44): prev_mode until t_switch; unless prev_mode is STANCE or the template's first mode, a
 * STANCE phase of transition_stance_time (model_settings.phaseTransitionStanceTime, task.info:11) is inserted before the first cycle. */
int qmgpu_switch_gait(const qmgpu_gait* gait, int32_t prev_mode, double transition_stance_time, double t_switch, double t_begin, double t_end,
                      int32_t* num_events, double* event_times /*[MAX_EVENTS]*/, int32_t* modes /*[MAX_EVENTS+1]*/);

/* Shooting grid over [t0, tf] the way upstream ocs2::timeDiscretizationWithEvents lays it out for the SQP solver built at
 * qm_controllers/src/QMController.cpp:288-289 (dt = task.info:79): steps of dt, every event time inside (t0, tf) becomes a
 * node and the dt stepping restarts from it; a remainder shorter than 1e-3 dt is merged into its neighbour.  Upstream keeps a
 * (pre-event, post-event) node pair at the same time; with the identity jump map of this robot the zero-length stage is a
 * no-op and only one node is emitted (a node takes the mode that starts at its time).
 * Writes num_nodes_minus_1 (= N to pass as qmgpu_mpc_args::num_nodes) and grid[0..N].  QMGPU_ERR_CAPACITY if N > max_nodes. */
int qmgpu_time_grid_with_events(double t0, double tf, double dt, int32_t num_events, const double* event_times,
                                int32_t max_nodes, int32_t* num_nodes_minus_1, double* grid);

/* ---- device context ---------------------------------------------------------------------------- */
int qmgpu_create(const qmgpu_problem* problem, int device, int max_batch, int max_nodes, qmgpu_handle* out);
/* Arithmetic type of the MPC kernels behind a handle.  QMGPU_F64 is the reference's (ocs2::scalar_t = double) and what qmgpu_create
 * gives.  QMGPU_F32 runs the whole MPC chain (derivatives, projection, Riccati, line search) in IEEE fp32 on
 * v_mfma_f32_16x16x4_f32 with fp32 scratch; every array of this interface stays fp64 and is converted on the device.  The WBC, the
 * front end and the policy evaluation always run in fp64.  Exists for BASELINE.json configs[4] (fp32-vs-fp64 tolerance sweep,
 * DESIGN.md section 5): fp32 results are NOT within the 1e-6 parity bar of the reference. */
typedef enum qmgpu_dtype { QMGPU_F64 = 0, QMGPU_F32 = 1 } qmgpu_dtype;
int qmgpu_create_ex(const qmgpu_problem* problem, int device, int max_batch, int max_nodes, int dtype, qmgpu_handle* out);
int qmgpu_destroy(qmgpu_handle h);
/* Use an externally owned HIP stream (hipStream_t passed as void*); NULL restores the handle's own stream. */
int qmgpu_set_stream(qmgpu_handle h, void* hip_stream);
int qmgpu_synchronize(qmgpu_handle h);
/* Optional (default off): the WBC launch of qmgpu_cycle_batch goes to a second stream owned by the handle, ordered behind the cycle's policy evaluation.  A WBC launch lasts as
 * long as its slowest instance while most CUs have finished theirs; the node kernels of the NEXT qmgpu_cycle_batch (which do not depend on it -- the reference runs its MPC and
 * its WBC in different threads, QMController.cpp:116-157 / 316-327) then fill those CUs.  With the option on, the WBC outputs of a cycle (out, out_status, input_last, working_set)
 * are complete after qmgpu_synchronize, or on the handle's stream after qmgpu_join_wbc (a device-side wait, no host wait) -- NOT after the caller synchronises its own stream.
 * Which calls join by themselves: qmgpu_cycle_batch (before its policy evaluation), qmgpu_wbc_solve_batch, qmgpu_set_stream (the NEW stream waits), qmgpu_set_overlap,
 * qmgpu_update_settings, qmgpu_debug_poison, qmgpu_synchronize, qmgpu_destroy.  Which do NOT: qmgpu_mpc_solve_batch, qmgpu_policy_eval_batch, qmgpu_frontend_batch,
 * qmgpu_warm_start_batch, qmgpu_gait_schedule_batch, qmgpu_get_input_weight, qmgpu_pack_results -- they touch nothing the library owns that a pending WBC reads or writes, and run next to it.
 * The pending WBC still READS the caller's rbd_measured / period / time (and ee_force) of that cycle and reads and writes input_last / working_set: those buffers must not
 * be modified -- by the caller's own kernels or copies on any stream, or through a non-joining call above -- until qmgpu_join_wbc, qmgpu_synchronize, or the next
 * qmgpu_cycle_batch / qmgpu_wbc_solve_batch has been issued on the handle. */
int qmgpu_set_overlap(qmgpu_handle h, int enable);
int qmgpu_join_wbc(qmgpu_handle h);
/* Replace the settings behind a live handle (gains, weights, limits, barrier parameters; the model is fixed at create time):
 * what the reference's dynamic_reconfigure callbacks do at run time (WbcBase::dynamicCallback, qm_wbc/src/WbcBase.cpp:74-121;
 * QMController::dynamicCallback, qm_controllers/src/QMController.cpp:358-363).  Ordered on the handle's stream: calls enqueued
 * afterwards see the new values; R' is recomputed. */
int qmgpu_update_settings(qmgpu_handle h, const qmgpu_settings* settings);
/* Leg-velocity-mapped input weight R' (30x30, row major) computed at create time (QMInterface.cpp:274-299). */
int qmgpu_get_input_weight(qmgpu_handle h, double* R_host);

/* Per-batch problem data, all device pointers. */
typedef struct qmgpu_mpc_args {
  int32_t batch, num_nodes;            /* N shooting intervals -> N+1 nodes */
  int32_t num_target_knots;            /* K >= 1 */
  int32_t line_search;                 /* 0: take the full step, 1: OCS2 filter line search */
  const double* t0;                    /* [batch] */
  const double* x0;                    /* [batch][30] */
  const double* time_grid;             /* [batch][N+1] or NULL -> t0 + k*dt */
  const double* target_times;          /* [batch][K] */
  const double* target_states;         /* [batch][K][37] */
  const int32_t* sched_num_events;     /* [batch] */
  const double* sched_event_times;     /* [batch][MAX_EVENTS] */
  const int32_t* sched_modes;          /* [batch][MAX_EVENTS+1] */
  const double* warm_x;                /* [batch][N+1][30] or NULL -> initializer (QMInitializer.cpp:33-41) */
  const double* warm_u;                /* [batch][N][30]   or NULL */
  double* out_t;                       /* [batch][N+1] */
  double* out_x;                       /* [batch][N+1][30] */
  double* out_u;                       /* [batch][N][30] */
  int32_t* out_mode;                   /* [batch][N+1] */
  double* out_stats;                   /* [batch][NSTATS]: merit0, violation0, merit1, violation1, alpha, step_type, armijo, status,
                                          SQP iterations performed, convergence (1 iteration limit, 2 step size, 3 metrics, 4 primal step) */
  const double* ee_contact_ref;        /* [batch][K][6] or NULL: per target knot the end-effector force reference f_ref (3) and the anchor
                                          p_env (3) of the compliant environment, interpolated linearly like the other references
                                          (force tracking, see qmgpu_settings::ee_contact_stiffness) */
  int32_t algorithm;                   /* QMGPU_ALG_SQP (0, the solver the reference instantiates) or QMGPU_ALG_DDP */
  int32_t reserved1;
} qmgpu_mpc_args;

/* Solver variants of qmgpu_mpc_solve_batch.
 *   QMGPU_ALG_SQP  multiple-shooting SQP: ocs2::SqpMpc, the object the reference builds (QMController.cpp:288-289, task.info:76-93).
 *   QMGPU_ALG_DDP  single-shooting DDP of the family the task file's ddp{} block configures (task.info:34-72, parsed at QMInterface.cpp:70 but never
 *                  instantiated by the reference): forward rollout of the nonlinear dynamics, LQ approximation along it,
 *                  the same projected Riccati recursion, and a line search that rolls the feedback policy u = u_nom + alpha du_ff + K (x - x_nom) out
 *                  for alpha = maxStepLength * 2^-i >= minStepLength, accepting the first whose merit (cost + constraintPenalty * dt |eq|^2) passes the
 *                  Armijo test.  OWN RESTATEMENT with stated deviations from upstream's SLQ: the rollout uses the shooting grid's RK2 steps (not
 *                  ODE45, task.info:129-137) and the backward pass is the discrete-time recursion (upstream's ILQR form, not the continuous-time
 *                  Riccati ODE).  One iteration per call (ddp.maxNumIterations 1).  The nominal trajectory the LQ approximation is formed along is
 *                  the warm start (warm_x, warm_u) as it is -- a previous solution's defects are part of the linearisation, as in Gauss-Newton
 *                  multiple shooting -- or, without warm_x, the open-loop rollout of warm_u / the initializer's inputs from x0.
 *                  out_stats: merit0, eq-violation0, merit1, eq-violation1, alpha (0 = no step accepted), trials evaluated, armijo, status. */
enum { QMGPU_ALG_SQP = 0, QMGPU_ALG_DDP = 1 };

int qmgpu_mpc_solve_batch(qmgpu_handle h, const qmgpu_mpc_args* args);

/* Linear interpolation of a solved trajectory at t_eval (MRT evaluatePolicy). All device pointers. */
int qmgpu_policy_eval_batch(qmgpu_handle h, int batch, int num_nodes, const double* t_grid, const double* X,
                            const double* U, const int32_t* modes, const double* t_eval, double* x_out /*[batch][30]*/,
                            double* u_out /*[batch][30]*/, int32_t* mode_out /*[batch]*/);

/* Initial guess of the next MPC call from the previous solution: (X, U) of the previous grid resampled (linear interpolation, end values held, as
 * upstream's LinearInterpolation) on new_grid [batch][new_nodes + 1]; x[0] of each instance is overwritten with x0 when x0 != NULL.  The results
 * are what qmgpu_mpc_args::warm_x / warm_u expect.  All device pointers; prev_* and warm_* must not alias. */
int qmgpu_warm_start_batch(qmgpu_handle h, int batch, int prev_nodes, const double* prev_grid, const double* prev_X, const double* prev_U,
                           int new_nodes, const double* new_grid, const double* x0 /*[batch][30] or NULL*/, double* warm_x /*[batch][new_nodes+1][30]*/,
                           double* warm_u /*[batch][new_nodes][30]*/);

typedef struct qmgpu_wbc_args {
  int32_t batch;
  int32_t variant;                     /* 0: HierarchicalWbc, 1: HierarchicalMpcWbc */
  const double* state_desired;         /* [batch][30] */
  const double* input_desired;         /* [batch][30] */
  const double* rbd_measured;          /* [batch][55] */
  const int32_t* mode;                 /* [batch] */
  const double* period;                /* [batch] */
  const double* time;                  /* [batch]  (t < 10 s selects the start-up task set, HierarchicalWbc.cpp:32-37) */
  double* input_last;                  /* [batch][30] in/out: WbcBase::inputLast_ (WbcBase.cpp:224-225) */
  double* out;                         /* [batch][54] */
  int32_t* out_status;                 /* [batch] 0 = all three QPs converged; bit l set = level l hit the iteration cap */
  const double* ee_force;              /* [batch][3] or NULL: external force on the arm end-effector (world axes, measured or from the contact
                                          model); enters the equations of motion, the torque limits and the torque recovery as J_ee^T f_e
                                          (force tracking, own formulation) */
  uint64_t* working_set;               /* [batch][QMGPU_WBC_STATE_WORDS] in/out or NULL: solver state carried from tick to tick, next to input_last.
                                          The reference cold-starts qpOASES on every level of every tick (HoQp.cpp:136-149); consecutive ticks of one
                                          robot end almost always with the same rows on their bounds, so each level's active-set method starts from the
                                          rows the previous tick pinned (same vertex whatever the path; a guess the first step refutes falls back to the
                                          cold path).  Zero-initialise it; zero it again to force a cold tick.
                                          word 0: bit 63 valid | contact mode | variant << 8 | start-up task set << 9  (a tick with another key starts cold)
                                          words 1..12: bit 63 valid | bits 0..55 rows pinned, one word per solve [1 + 6 pass + 2 level + completion]
                                          words 13, 14: passes of every solve of the last tick, pass 0 / canonical pass (one byte per [2 level + completion]:
                                          bits 0..6 interior-point + active-set iterations, bit 7 the carried guess was refuted); word 15 reserved */
} qmgpu_wbc_args;

int qmgpu_wbc_solve_batch(qmgpu_handle h, const qmgpu_wbc_args* args);

/* One control cycle per robot: MPC solve, policy evaluation at t_eval, WBC.  */
int qmgpu_cycle_batch(qmgpu_handle h, const qmgpu_mpc_args* mpc, const double* t_eval /*[batch]*/, qmgpu_wbc_args* wbc);

/* The solved batch as ONE record per instance, [batch][(N+1)*30 + N*30 + 54 + (N+1)] doubles = X | U | WBC output | contact modes (as doubles: exact): what the ranks of a
 * sharded batch all-gather (the path's only exchange; the collective itself is issued by the host through RCCL).  All device pointers; enqueued
 * on the handle's stream.  Does NOT join a WBC pending on the overlap stream (qmgpu_set_overlap): pack the buffers of a cycle whose WBC has been joined -- behind the
 * next qmgpu_cycle_batch for alternating output buffers, or behind qmgpu_join_wbc. */
int qmgpu_pack_results(qmgpu_handle h, int batch, int num_nodes, const double* X, const double* U, const double* wbc_out, const int32_t* modes, double* packed);

/* ---- front end of a control cycle: what sits immediately before the MPC call ----------
 * (1) state estimate -> MPC observation: rbdState[55] -> centroidal state x[30] with the yaw unwrapped against the previous
 *     observation (QMController::updateStateEstimation, qm_controllers/src/QMController.cpp:239-244; upstream
 *     CentroidalModelRbdConversions::computeCentroidalStateFromRbdModel);
 * (2) command -> TargetTrajectories (two knots of 37-dim states), the free functions of
 *     qm_controllers/src/QmTargetTrajectoriesPublisher_node.cpp:
 *       kind 0  hold           the initial target of QMController::starting (QMController.cpp:107-113)
 *       kind 1  base cmd_vel   cmdVelToTargetTrajectories (:89-129), command = (vx, vy, vz, yaw rate) in the base frame
 *       kind 2  EE cmd_vel     EeCmdVelToTargetTrajectories (:134-188)
 *       kind 3  EE goal pose   EEgoalPoseToTargetTrajectories (:195-238) + positionCommandCallback (:240-254),
 *                              command = position (3) + quaternion x y z w (4)
 * The outputs plug straight into the MPC arguments: x0, target_times, target_states with num_target_knots = 2. */
typedef struct qmgpu_frontend_args {
  int32_t batch;
  const double* rbd_measured;   /* [batch][55] */
  const double* time;           /* [batch] observation time */
  const double* yaw_last;       /* [batch] yaw of the previous observation, or NULL (no unwrapping) */
  const int32_t* command_kind;  /* [batch] 0..3 */
  const double* command;        /* [batch][7] */
  double* last_ee_target;       /* [batch][7] in/out: lastEeTarget_ of the publisher (position + quaternion xyzw) */
  const double* feet_height;    /* [batch] mean z of the contact feet (FEET_HEIGHT, :27-35), or NULL -> 0 */
  double arm_dist;              /* StartingPosition.h:13 (0.6) */
  double start_x, start_y, start_psi; /* StartingPosition.h:9-12 (-2, 0, 0): only kind 0 uses them */
  double* x0;                   /* [batch][30] out */
  double* target_times;         /* [batch][2] out */
  double* target_states;        /* [batch][2][37] out */
} qmgpu_frontend_args;

int qmgpu_frontend_batch(qmgpu_handle h, const qmgpu_frontend_args* args);

/* Gait front end on the device: the per-instance mode schedules of a batch from gait templates, instead of
 * tiling them on the host (qmgpu_tile_gait) and shipping MAX_EVENTS times + modes per instance every cycle.  Replaces, for a batch of
 * robots, what GaitTopicPublisher::gaitCommandCallback (qm_controllers/src/GaitTopicPublisher.cpp:31-44) + upstream GaitReceiver /
 * GaitSchedule::tileModeSequenceTemplate do for one: instance i runs template gait_index[i] of `templates` (host array, copied), first
 * cycle at t_phase0[i] (prev_mode[i] before it -- STANCE when prev_mode is NULL -- with the phase-transition stance of qmgpu_switch_gait,
 * settings.phase_transition_stance_time, where upstream inserts one), tiled over [t_begin[i], t_end[i]], default STANCE after the last tiled
 * cycle.  The outputs are bit-identical to qmgpu_tile_gait / qmgpu_switch_gait and plug straight into qmgpu_mpc_args::sched_*.  status[i] = QMGPU_ERR_CAPACITY (and a pure STANCE
 * schedule) when the schedule needs more than QMGPU_MAX_EVENTS events.  All pointers except `templates` are device pointers. */
int qmgpu_gait_schedule_batch(qmgpu_handle h, int batch, const qmgpu_gait* templates, int num_templates, const int32_t* gait_index,
                              const int32_t* prev_mode /*[batch] or NULL*/, const double* t_phase0 /*[batch]: t_switch of qmgpu_switch_gait*/,
                              const double* t_begin, const double* t_end, int32_t* sched_num_events /*[batch]*/, double* sched_event_times /*[batch][MAX_EVENTS]*/,
                              int32_t* sched_modes /*[batch][MAX_EVENTS+1]*/, int32_t* status /*[batch] or NULL*/);

/* Diagnostics used by the parity tests: per-node LQ blocks of the last qmgpu_mpc_solve_batch
 * (before projection: A B b | Q R q r | C D e ; nc rows valid).  Host pointers, any may be NULL. */
int qmgpu_debug_get_lq(qmgpu_handle h, int instance, int node, double* A, double* B, double* b, double* Q, double* R,
                       double* q, double* r, double* C, double* D, double* e, int32_t* nc);

/* Device time (ms) of each kernel of the last timed call, measured with HIP events on the handle's stream:
 * [0] ad_node  [1] lq_node (projection)  [2] riccati  [3] line search + update  [4] wbc  [5] whole call */
int qmgpu_last_kernel_ms(qmgpu_handle h, double* ms6);
/* Same, averaged over the last `last_calls` timed calls (event ring of 256 calls; no per-call host sync is needed). */
int qmgpu_kernel_ms_mean(qmgpu_handle h, int last_calls, double* ms6);
/* The same six durations of EACH of the last `last_calls` timed calls, oldest first: ms6_per_call [last_calls][6] (a launch of the sequential kernels lasts as long as
 * its slowest instance: the spread from call to call is what a real-time caller has to budget for). */
int qmgpu_kernel_ms_history(qmgpu_handle h, int last_calls, double* ms6_per_call);
int qmgpu_enable_timing(qmgpu_handle h, int enable);
/* Test aid: fills every scratch buffer behind the handle and the LDS of every CU with NaN, so that a kernel reading memory that
 * nothing wrote in this call shows up as NaN in the results instead of passing on left-over values. */
int qmgpu_debug_poison(qmgpu_handle h);
/* Allocate / enable the per-node dump read by qmgpu_debug_get_lq (off by default: 37 KiB per node). */
int qmgpu_enable_debug(qmgpu_handle h, int enable);

#ifdef __cplusplus
}
#endif
#endif /* QMGPU_H */
