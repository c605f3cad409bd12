/*
 * hsqp.h — C ABI of the MI355X-native multiple-shooting SQP iteration for the
 * ocs2-based humanoid NMPC of manumerous/wb_humanoid_mpc (Unitree G1): the whole-body
 * acceleration-level formulation and the centroidal formulation (hsqp_model_desc::formulation).
 *
 * This is the drop-in boundary.  It replaces what the reference
 * reaches through `ocs2::SqpMpc` / `ocs2::SolverBase`:
 *   - humanoid_nmpc/humanoid_wb_mpc_ros2/src/WBMpcSqpNode.cpp:64      (SqpMpc construction)
 *   - humanoid_nmpc/humanoid_wb_mpc_ros2/src/WBMpcSqpNode.cpp:85-89   (solver ptr handed to MPC_ROS_Interface)
 *   - humanoid_nmpc/humanoid_wb_mpc/src/mrt/WBMpcMrtJointController.cpp:200-213 (advanceMpc -> SolverBase::run)
 *   - humanoid_nmpc/humanoid_centroidal_mpc_ros2/src/CentroidalMpcSqpNode.cpp:63  (the same seam of the centroidal MPC)
 * The per-node callbacks the reference's solver makes into the problem
 * definition (SystemDynamicsBase / StateInputCost / StateInputConstraint /
 * PreComputation virtuals) do not cross this boundary: their bodies are device
 * code, their time-varying inputs arrive as the per-node parameter table
 * `hsqp_problem::node_params` that the adaptor samples from the reference's own
 * ReferenceManager / SwingTrajectoryPlanner.
 *
 * Plain C, POD only, no exceptions, no torch types.  All floating point is IEEE
 * double (ocs2::scalar_t).  All matrices row-major.  The caller owns every
 * host buffer for the duration of a call; nothing is retained afterwards.
 * A handle is used by one thread at a time (the reference's MPC thread).
 */
#ifndef HSQP_H
#define HSQP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- problem dimensions (G1 whole-body: WBAccelMpcRobotModel.h:47-94) -------------------- */
#define HSQP_NJ 23              /* actuated joints kept in the MPC model            */
#define HSQP_NV 29              /* generalized velocities: 6 base + NJ               */
#define HSQP_NX 58              /* x = [p_b(3) eulerZYX(3) q_j | v_b(3) eulerZYXdot(3) qd_j] */
#define HSQP_NU 35              /* u = [W_left(6)=f,m  W_right(6)  qdd_j]            */
#define HSQP_NB 24              /* rigid bodies: base + one per joint                */
#define HSQP_NCONTACT 2
#define HSQP_NODE_PARAMS 72     /* doubles per shooting node in hsqp_problem::node_params */

/* per-node parameter table layout (what the reference samples per node through
 * SwitchedModelReferenceManager::getContactFlags / getDesiredState,
 * SwingTrajectoryPlanner::getZ*Constraint / getImpactProximityFactor) */
#define HSQP_P_XDES 0           /* [58] TargetTrajectories::getDesiredState(t)                        */
#define HSQP_P_ARMSWING 58      /* sin(2*pi*(phase(t)-0.15)); 0 disables the arm-swing reference      */
#define HSQP_P_CONTACT 59       /* [2]  contact flags {left,right} as 0.0/1.0                         */
#define HSQP_P_SWING 61         /* [2][3] per foot z*, zdot*, zddot* of the swing spline               */
#define HSQP_P_IMPACT 67        /* [2]  impact proximity factor per foot                               */

/* ---- centroidal formulation (hsqp_model_desc::formulation = HSQP_FORM_CENTROIDAL) ----------------
 * x = [h/m (6: v_com, L/m) | p_b(3) eulerZYX(3) | q_j(23)], u = [W_left(6) W_right(6) | qd_j(23)]
 * (humanoid_nmpc/humanoid_centroidal_mpc/include/humanoid_centroidal_mpc/common/CentroidalMpcRobotModel.h:49-71,89-95).
 * Every array of this ABI keeps the whole-body row strides (58 per state, 35 per input, 72 per node-parameter row): the
 * centroidal state occupies the first HSQP_CNX doubles of a state row, the remaining 23 are padding and must be zero on
 * input (they come back zero).  Node-parameter rows: [0..34] desired state, [HSQP_PC_TORSO..+12] the torso task-space
 * reference {position(3), quaternion x y z w, linear velocity(3), angular velocity(3)} — EndEffectorKinematicsQuadraticCost::
 * getParameters (humanoid_common_mpc/src/cost/EndEffectorKinematicsQuadraticCost.cpp:80-104) —, the slots from
 * HSQP_P_ARMSWING on as above with HSQP_P_SWING holding {z*, zdot*, unused} per foot. */
#define HSQP_FORM_WB 0
#define HSQP_FORM_CENTROIDAL 1
#define HSQP_CNX 35
#define HSQP_PC_TORSO 35

/* ---- error codes ------------------------------------------------------------------------- */
#define HSQP_OK 0
#define HSQP_ERR_BAD_ARG (-1)
#define HSQP_ERR_NO_DEVICE (-2)
#define HSQP_ERR_OOM (-3)
#define HSQP_ERR_NUMERIC (-4)   /* reduced Hessian not positive definite / rank-deficient D            */
#define HSQP_ERR_HIP (-5)       /* HIP runtime failure, see hsqp_last_error                            */
#define HSQP_ERR_NOT_CONVERGED (-6)

typedef struct hsqp_body {
  int32_t parent;               /* parent body index, -1 for the base (pelvis)                         */
  int32_t reserved;
  double R[9];                  /* joint placement in the parent body frame (rotation, row-major)      */
  double p[3];                  /* joint placement translation                                          */
  double axis[3];               /* revolute axis in the joint (= child body) frame                      */
  double mass;
  double com[3];                /* centre of mass in the body frame                                     */
  double inertia[9];            /* rotational inertia about the com, body axes                          */
  double q_lo, q_hi;            /* URDF position limits (unused for the base)                           */
} hsqp_body;

typedef struct hsqp_frame {
  int32_t body;
  int32_t reserved;
  double p[3];                  /* translation in the body frame; rotation is identity (createPinocchioModel.cpp:77-83) */
} hsqp_frame;

/* Relaxed / piece-wise polynomial barrier settings (mu, delta). */
typedef struct hsqp_barrier { double mu, delta; } hsqp_barrier;

typedef struct hsqp_model_desc {
  int32_t formulation;          /* HSQP_FORM_WB (whole-body acceleration-level) or HSQP_FORM_CENTROIDAL  */
  int32_t n_joints;             /* must equal HSQP_NJ                                                    */
  hsqp_body bodies[HSQP_NB];    /* bodies[0] = base; bodies[1+j] moved by joint j; parents before children */
  hsqp_frame contact[2];        /* foot_{l,r}_contact                                                    */
  hsqp_frame collision_p1[2];   /* foot_*_contact_collision_p_1                                          */
  hsqp_frame collision_p2[2];
  hsqp_frame ankle[2];
  hsqp_frame knee[2];
  double gravity;               /* 9.81 */
  double Q[HSQP_NX];            /* diagonal state weights  (task.info Q, scaling applied)                */
  double R[HSQP_NU];            /* diagonal input weights                                                */
  double Qf[HSQP_NX];           /* diagonal terminal weights (Q_final * terminalCostScaling)             */
  double foot_sqrt_w[18];       /* sqrt of the effective EndEffectorDynamicsWeights vector               */
  /* stance / swing foot constraint gains (ModelSettings::FootConstraintConfig) */
  double gain_pos_z, gain_ori, gain_linvel_z, gain_linvel_xy, gain_angvel;
  double gain_linacc_z, gain_linacc_xy, gain_angacc;
  /* friction cone (FrictionForceConeConstraint::Config) + its relaxed barrier */
  double friction_mu, friction_reg, friction_grip, friction_hess_shift;
  hsqp_barrier friction_barrier;
  /* contact moment XY: rectangle bounds in the contact frame + relaxed barrier */
  double rect_x_min, rect_x_max, rect_y_min, rect_y_max;
  hsqp_barrier moment_barrier;
  hsqp_barrier joint_limit_barrier;   /* PieceWisePolynomialBarrierPenalty */
  double r_foot, r_knee;
  hsqp_barrier collision_barrier;     /* PieceWisePolynomialBarrierPenalty */
  int32_t arm_swing_joint[4];   /* joint indices {l_shoulder_y, r_shoulder_y, l_elbow_y, r_elbow_y}      */
  /* ---- centroidal formulation only (ignored for HSQP_FORM_WB).  Q / R / Qf hold the 35 centroidal weights in their first
   * entries (rest zero), gain_pos_z / gain_ori the foot_constraint gains of the centroidal task file. */
  hsqp_frame torso;             /* task_space_costs.torso.link_name (mid360_link): body + translation     */
  double torso_R[9];            /* ... and the frame's rotation in the body frame (row-major)             */
  double torso_sqrt_w[12];      /* sqrt(EndEffectorKinematicsWeights) of the torso cost: pos, ori, lin vel, ang vel */
  double cent_foot_sqrt_w[12];  /* sqrt(task_space_foot_cost_weights) of CentroidalMpcEndEffectorFootCost  */
  double ext_torque_sqrt_w[2][6];   /* sqrt({left,right}_leg_torque_cost.weights)                          */
  int32_t ext_torque_joint[2][6];   /* their activeJointNames as joint indices                              */
} hsqp_model_desc;

#define HSQP_FLAG_LINESEARCH 1   /* hsqp_solve runs the filter line search (as the reference's SqpSolver does) instead of alpha = 1 */
/* Backward sweep of the stage QP.  Default: the serial Riccati recursion, one workgroup per instance.  With at most
 * HSQP_SCAN_AUTO_BATCH instances and at least HSQP_SCAN_AUTO_MIN_NODES shooting intervals (both formulations) the parallel-in-time
 * sweep (associative scan over the stages, ceil(log2(N+1)) levels; csrc/hsqp_scan.h) is used instead, because one or two serial
 * chains leave the device idle (N = 100, sweep + roll-out, round 6: whole-body 0.52 vs 1.11 ms; centroidal 0.30 vs 0.8 ms).  Its result agrees with the serial
 * recursion's to ~1e-11 of the step's scale on the QPs of a cold start or of a tracking MPC; it degrades on far-from-feasible
 * line-search iterates and on badly scaled QPs (cond(I + C1 J2) up to 1e9), so every scan result is GATED by the KKT residual of the
 * QP (min(1e-9 max(1, |g|_inf), 2e-8)) and the iteration is redone with the serial recursion when it fails: hsqp_scan_fallbacks()
 * counts those. */
#define HSQP_FLAG_SERIAL_RICCATI 2     /* always the serial recursion                                              */
#define HSQP_FLAG_PARALLEL_RICCATI 4   /* the scan for every batch size and horizon (still gated); excludes HSQP_FLAG_SERIAL_RICCATI */
/* Cost of the gate: it is all-or-nothing per call — one rejected instance redoes the serial sweep, the step, the KKT report and the
 * performance indices for the whole batch.  Up to 256 instances nothing would be saved by redoing only the rejected ones (the serial
 * sweep is one workgroup per instance on 256 CUs: 1.2 - 1.3 ms for 1 or 256 of them); beyond that, and on far-from-feasible line-search
 * iterates that fail the gate regularly, a forced scan roughly doubles the iteration time — and it holds two value-function buffers of
 * max_batch * (max_nodes + 1) * 3 422 doubles (1.4 GB at 256 x 100).  The flag is for measurements; the automatic choice is
 * max_batch <= HSQP_SCAN_AUTO_BATCH.  After a rejection the handle backs off: the next 1, 3, 7, .. 63 iterations go straight through the
 * serial recursion before the scan is tried again (an accepted scan resets the count) — the iterates that fail the gate come in runs. */
#define HSQP_SCAN_AUTO_BATCH 2
#define HSQP_SCAN_AUTO_MIN_NODES 48
/* Two-level (segmented) sweep for the batches in between (csrc/hsqp_segment.h) — OPT-IN.  One workgroup per instance leaves 256 - B CUs idle
 * during the N dependent stages (BASELINE config 4 as written puts 32 instances on each of 8 GPUs: 1.22 of the 1.83 ms of a step, round 6).  With this
 * flag the horizon of every instance is cut into P = 7 (or 3) segments: segment elements (Riccati recursion from J = 0 + prepended closed
 * loops) on B P workgroups, a suffix scan over the P + 1 elements, ordinary recursions per segment from the boundary value functions, the
 * roll-out: 0.93 ms per sweep at 32 instances (step 1.57 against 1.83 ms).  Gated like the scan — by the KKT residual of the segments' last stages — with the serial
 * recursion as fallback (hsqp_scan_fallbacks counts; after a rejection the handle backs off for 1, 3, 7, .. iterations).  DECLARED RELAXATION
 * of BASELINE.md §6: its step differs from the serial recursion's by 4e-11 .. 4e-10 of the step's scale (measured on perturbed config-4
 * batches: up to 7e-8 absolute where the bound on trajectories is 1e-8), which is why no default path takes it. */
#define HSQP_FLAG_SEGMENTED_RICCATI 8  /* excludes HSQP_FLAG_SERIAL_RICCATI and HSQP_FLAG_PARALLEL_RICCATI */
typedef struct hsqp_settings {
  int32_t max_nodes;            /* N_max: shooting intervals per instance                                */
  int32_t max_batch;            /* independent MPC instances per call on this device                     */
  int32_t device;               /* HIP device ordinal                                                    */
  int32_t flags;                /* HSQP_FLAG_* bits                                                      */
} hsqp_settings;

typedef struct hsqp_problem {
  int32_t batch;                /* B independent instances                                               */
  int32_t n_nodes;              /* N shooting intervals -> N+1 nodes                                     */
  double dt;                    /* node spacing of a uniform grid (used when dt_nodes is NULL)             */
  const double* x_init;         /* [B][58]        measured state                                          */
  const double* x_traj;         /* [B][N+1][58]   linearisation trajectory (warm start)                   */
  const double* u_traj;         /* [B][N][35]                                                            */
  const double* node_params;    /* [B][N+1][HSQP_NODE_PARAMS]                                            */
  /* Optional non-uniform grid with event nodes (ocs2 multiple-shooting transcription, SURVEY.md A.5: the grid is split at the
   * mode-switch times, the last interval is shortened to the final time, and at an event time a pre- and a post-event node share
   * the time stamp, joined by the identity jump map).  dt_nodes[b][k] >= 0 is the length of interval k of instance b;
   * dt_nodes[b][k] == 0 marks interval k as an EVENT: x_{k+1} = x_k (A = I, B = 0, defect x_k - x_{k+1} counted unscaled in the
   * dynamics SSE), no cost, no constraints, du_k = 0.  NULL: every interval has length dt. */
  const double* dt_nodes;       /* optional [B][N]                                                       */
} hsqp_problem;

/* ---- device-side parameter generation ------------------------------------
 * Instead of sampling the reference's ReferenceManager / SwingTrajectoryPlanner on the host for every node and
 * uploading the table, the adaptor hands over the compact per-instance reference and the table is built on the device. */
typedef struct hsqp_swing_config {      /* SwingTrajectoryPlanner::Config (task.info swing_trajectory_config)       */
  double lift_off_velocity, touch_down_velocity, swing_height, touch_down_height_offset, swing_time_scale;
  double impact_mid, impact_lift_velocity, impact_touch_velocity;   /* impactProximityFactor{MidPointValue,LiftOffVelocity,TouchDownVelocity} */
} hsqp_swing_config;

typedef struct hsqp_reference {
  int32_t batch, n_nodes;
  double t0, dt;                        /* node k sits at t0 + k dt                                               */
  int32_t max_events;                   /* row length of event_times; mode_sequence rows have max_events + 1      */
  const int32_t* n_events;              /* [B]        events of each instance's ocs2::ModeSchedule                */
  const double* event_times;            /* [B][max_events]                                                        */
  const int32_t* mode_sequence;         /* [B][max_events + 1]   0 FLY, 1 RF, 2 LF, 3 STANCE (MotionPhaseDefinition.h:47-56) */
  int32_t n_knots;                      /* knots of the TargetTrajectories (>= 1)                                  */
  const double* target_times;           /* [B][n_knots]                                                           */
  const double* target_states;          /* [B][n_knots][58]  (centroidal: the first 35 entries of a row, rest zero)     */
  hsqp_swing_config swing;
  double terrain_height;
  int32_t arm_swing;                    /* 0 disables the arm-swing reference                                     */
  int32_t reserved;
  const double* node_times;             /* optional [B][N+1]: time stamp of every node (non-uniform grids / event nodes: a post-event
                                           node carries t_event + epsilon, as ocs2's getIntervalStart does); NULL: t0 + k dt  */
} hsqp_reference;

typedef struct hsqp_perf {      /* ocs2::PerformanceIndex subset, per instance                           */
  double merit, cost, dynamics_sse, equality_sse;
} hsqp_perf;

typedef struct hsqp_timings {   /* SqpSolver::getBenchmarks() buckets (SqpBenchmarksPublisher.cpp:54-57), seconds */
  double lq_approximation, solve_qp, linesearch, compute_controller, total;
} hsqp_timings;

/* Filter line search of the SQP step.  Replaces ocs2::FilterLinesearch + the back-tracking loop of
 * ocs2::SqpSolver::takeStep (upstream ocs2_sqp; the reference sets g_max / g_min / deltaTol in
 * robot_models/unitree_g1/g1_wb_mpc/config/mpc/task.info:81-90, the rest are the upstream defaults). */
typedef struct hsqp_linesearch_settings {
  double g_max, g_min;          /* constraint-violation thresholds of the filter                         */
  double gamma_c;               /* required relative decrease of the violation                           */
  double armijo_factor;         /* Armijo sufficient-decrease factor                                     */
  double alpha_decay, alpha_min;/* back-tracking factor and smallest step length tried                   */
  double delta_tol;             /* escape when alpha*|dx| and alpha*|du| fall below it                   */
  double cost_tol;              /* HSQP_ITER_UNTIL_CONVERGED: |merit after - merit before| below it (with the violation below g_min) ends the loop (ocs2 costTol) */
} hsqp_linesearch_settings;

#define HSQP_STEP_COST 0        /* accepted by the Armijo condition on the cost                          */
#define HSQP_STEP_DUAL 1        /* accepted by cost or constraint decrease                               */
#define HSQP_STEP_CONSTRAINT 2  /* accepted by constraint decrease only                                  */
#define HSQP_STEP_ZERO 3        /* no step length accepted: the trajectory is kept                       */
#define HSQP_STEP_FULL 4        /* plain full step, no line search requested                             */

typedef struct hsqp_solution {
  double* x;                    /* [B][N+1][58]  x + alpha dx                                            */
  double* u;                    /* [B][N][35]    u + alpha du                                            */
  double* dx;                   /* optional [B][N+1][58] QP step (may be NULL)                           */
  double* du;                   /* optional [B][N][35]                                                   */
  hsqp_perf* perf_before;       /* optional [B] performance index of the linearisation trajectory       */
  hsqp_perf* perf_after;        /* optional [B] performance index after the full step                    */
  double* kkt;                  /* optional [B][2] {stationarity, primal} inf-norm residual of the projected QP */
  double* alpha;                /* optional [B]  accepted step length (1 without line search)            */
  int32_t* step_type;           /* optional [B]  HSQP_STEP_*                                              */
  double* armijo;               /* optional [B]  descent metric  sum_k q~.dx + r~.ut  of the projected QP */
  double* grad_inf;             /* optional [B]  |g|_inf of the projected QP (q~_k, r~_k of every node): the scale the KKT residuals are
                                   judged against (r <= 1e-9 max(1, |g|_inf)); filled when kkt is                   */
  hsqp_timings timings;
} hsqp_solution;

typedef struct hsqp_handle hsqp_handle;

/* Build the device problem image (model constants, weights) and allocate all device
 * workspaces for (max_batch, max_nodes).  Replaces the SqpMpc/SqpSolver constructor. */
int hsqp_create(const hsqp_model_desc* model, const hsqp_settings* settings, hsqp_handle** out);
void hsqp_destroy(hsqp_handle* h);

/* One SQP iteration for every instance: LQ approximation at all nodes, equality
 * projection, Riccati QP, full step, performance indices.  Blocking. Replaces one
 * pass of SqpSolver::runImpl with sqpIteration = 1. */
int hsqp_solve(hsqp_handle* h, const hsqp_problem* problem, hsqp_solution* solution);

/* Same, but the problem is already resident on the device from the previous
 * hsqp_solve/hsqp_upload and the result is left there (bench / multi-iteration use). */
int hsqp_upload(hsqp_handle* h, const hsqp_problem* problem);
/* Like hsqp_upload, but hsqp_problem::node_params may be NULL: the per-node table is generated on the device from `ref`
 * (batch, n_nodes, dt must agree).  Fails with HSQP_ERR_BAD_ARG if a swing phase has no lift-off / touch-down inside
 * the schedule (the reference's SwingTrajectoryPlanner throws there).  Centroidal formulation: the torso task-space reference
 * of every node is computed on the device as well (kinematics of the torso link at the interpolated target state). */
int hsqp_upload_reference(hsqp_handle* h, const hsqp_problem* problem, const hsqp_reference* ref);
#define HSQP_ITER_TAKE_STEP 1    /* after every iteration but the last: x <- x + dx, u <- u + du            */
#define HSQP_ITER_KKT 2          /* also evaluate the KKT residual of the projected QP (not part of a step)  */
#define HSQP_ITER_LINESEARCH 4   /* filter line search on the step length instead of the plain full step     */
/* n_iterations is an upper bound: stop as soon as EVERY instance has converged the way ocs2's SqpSolver::checkConvergence does, in its
 * order — no step length accepted (STEPSIZE), |merit after - merit before| < cost_tol with the constraint violation below g_min (METRICS),
 * alpha |dx| and alpha |du| below delta_tol (PRIMAL).  One small read-back per iteration
 * (the line-search state); the trajectories never leave the device.  hsqp_last_iterations() tells how many were run, hsqp_iteration_log()
 * returns the performance index / step length / step type each of them ended with (sqpIteration > 1 of the reference in ONE call). */
#define HSQP_ITER_UNTIL_CONVERGED 8
int hsqp_iterate_device(hsqp_handle* h, int n_iterations, int flags);
int hsqp_last_iterations(const hsqp_handle* h);     /* iterations the last hsqp_iterate_device call ran (-1: h is NULL) */
/* Per-iteration record of the last hsqp_iterate_device call with HSQP_ITER_UNTIL_CONVERGED: iteration in [0, hsqp_last_iterations),
 * any output may be NULL; perf = performance index after that iteration's (line-searched) step. */
int hsqp_iteration_log(const hsqp_handle* h, int iteration, hsqp_perf* perf /*[B]*/, double* alpha /*[B]*/, int32_t* step_type /*[B]*/);
/* Change the quadratic cost weights of a live handle (the reference's centroidal node retunes Q / R at run time through its gains
 * receiver: humanoid_centroidal_mpc_ros2 GainsReceiver): diagonal Q[58], R[35], Qf[58] in the layout of hsqp_model_desc; NULL keeps
 * the current values.  Takes effect with the next iteration. */
int hsqp_update_weights(hsqp_handle* h, const double* Q, const double* R, const double* Qf);
/* The weights and gains of the other task terms, as the reference's gains updaters retune them at run time
 * (humanoid_centroidal_mpc_ros2/include/humanoid_centroidal_mpc_ros2/gains/: EndEffectorFootGainsUpdater — foot task-space cost weights —,
 * StateInputConstraintGainsUpdater — stance / swing foot constraint gains —, StateInputSoftConstraintGainsUpdater, JointLimitsGainsUpdater,
 * FootCollisionGainsUpdater — barrier (mu, delta) —, EndEffectorKinematicsGainsUpdater — torso cost of the centroidal task).  Same fields and
 * meaning as in hsqp_model_desc.  hsqp_get_term_weights fills the struct with the handle's current values; hsqp_update_term_weights
 * replaces ALL of them (get, change, update) after re-running the model validation of hsqp_create on the result (HSQP_ERR_BAD_ARG and no
 * change if it fails).  Takes effect with the next iteration. */
typedef struct hsqp_term_weights {
  double foot_sqrt_w[18];
  double gain_pos_z, gain_ori, gain_linvel_z, gain_linvel_xy, gain_angvel, gain_linacc_z, gain_linacc_xy, gain_angacc;
  hsqp_barrier friction_barrier, moment_barrier, joint_limit_barrier, collision_barrier;
  double torso_sqrt_w[12], cent_foot_sqrt_w[12], ext_torque_sqrt_w[2][6];   /* centroidal formulation only */
} hsqp_term_weights;
int hsqp_get_term_weights(const hsqp_handle* h, hsqp_term_weights* out);
int hsqp_update_term_weights(hsqp_handle* h, const hsqp_term_weights* w);

/* Page-lock a caller-owned host buffer (hipHostRegister / hipHostUnregister behind the C ABI, so that a caller need not link the HIP
 * runtime): hsqp_solve / hsqp_upload / hsqp_download move 36 + 39 MB per iteration at 256 instances x 100 nodes; from pageable memory the
 * runtime stages them through its own bounce buffers, from registered buffers they are one DMA each.  The library does not copy into a staging area of its own: that would add a host memcpy of the same size.  Register the
 * trajectory / parameter / solution arrays once, reuse them every MPC cycle, unregister before freeing them.
 * HSQP_OK, HSQP_ERR_NO_DEVICE, HSQP_ERR_BAD_ARG (null / zero bytes) or HSQP_ERR_HIP (the runtime refused, e.g. the memlock limit).
 * Measured (bench.py `pcie_inclusive`, buffers reused every call, round 6): 7.0 ms per hsqp_solve from pageable memory, 6.7 ms page-locked, of which
 * the iteration with its KKT report is 5.0 ms — the runtime's pageable path already reaches ~40 GB/s on this host. */
int hsqp_host_register(void* buffer, size_t bytes);
int hsqp_host_unregister(void* buffer);
/* Line-search settings of the handle (defaults: task.info g_max 1e-2, g_min 1e-6, deltaTol 1e-4; upstream gamma_c 1e-6,
 * armijoFactor 1e-4, alpha_decay 0.5, alpha_min 1e-4). */
void hsqp_linesearch_defaults(hsqp_linesearch_settings* s);
int hsqp_set_linesearch(hsqp_handle* h, const hsqp_linesearch_settings* s);
int hsqp_download(hsqp_handle* h, hsqp_solution* solution);
/* The same two with every array pointer of hsqp_problem / hsqp_solution being DEVICE memory of the handle's GPU (multi-GPU data
 * path, SURVEY.md §8e: the shards RCCL scatters arrive in HBM and the solutions are gathered from HBM — no host staging).
 * hsqp_upload_device: centroidal padding is the caller's responsibility.  hsqp_download_device: alpha / step_type / armijo must be
 * NULL; the numeric status is still checked on the host and reported through the return code. */
int hsqp_upload_device(hsqp_handle* h, const hsqp_problem* problem);
int hsqp_download_device(hsqp_handle* h, hsqp_solution* solution);

/* Debug/parity access to intermediate device blocks of the LAST iteration.
 * `what` is one of the HSQP_BLK_* ids; copies min(bytes, block size) and
 * returns the block size in bytes (negative error code on failure). */
#define HSQP_BLK_AB 1           /* [B][N][58][93]   [A|B] of the RK4 sensitivity discretisation          */
#define HSQP_BLK_BVEC 2         /* [B][N][58]       defect b_k                                            */
#define HSQP_BLK_H 3            /* [B][N][93][93]   dt * Hessian of the stage cost wrt [x;u]              */
#define HSQP_BLK_G 4            /* [B][N][93]       dt * gradient                                         */
#define HSQP_BLK_CDE 5          /* [B][N][14][94]   [C|D|e] active equality rows (zero padded)            */
#define HSQP_BLK_NE 6           /* [B][N] int32     number of active equality rows                        */
#define HSQP_BLK_COST 7         /* [B][N+1]         dt*l_k (terminal: l_N)                                 */
#define HSQP_BLK_DX 8           /* [B][N+1][58]                                                            */
#define HSQP_BLK_DU 9           /* [B][N][35]                                                              */
#define HSQP_BLK_FLOW 10        /* [B][N][58]       xdot at (x_k,u_k)                                      */
#define HSQP_BLK_PARAMS 11      /* [B][N+1][72]     the per-node parameter table resident on the device    */
#define HSQP_BLK_FORMS 12       /* int32[5]         which kernel forms the handle runs (decided at hsqp_create from its size): [0] whole-body LQ approximation
                                                    on limb lanes (k_lq_limb + k_lq_rows; see [4]) instead of k_lq<true>, [1] value pass on quads of lanes
                                                    (k_value_quad) instead of k_step_value; [2] node ranges the limb-lane LQ kernels are launched in, each on a
                                                    stream of its own, once a launch exceeds one round of the chip (0: phase form; HSQP_LQ_SPLIT in the
                                                    environment at hsqp_create overrides the default 2); [3] the whole-body serial sweep runs on the factors of [A~ | B~]
                                                    (k_riccati_fact; HSQP_RICCATI_DENSE in the environment at hsqp_create: the dense stage k_riccati<58>); [4] limb-lane form: the RK4
                                                    chain of the columns of [A|B] runs inside k_project, the defect on the lanes of k_lq_rows, and k_lq_chain is not launched
                                                    (HSQP_LQ_CHAIN_SEPARATE in the environment at hsqp_create: chain and defect in k_lq_chain, P6 / V6 through the LQ record).
                                                    Available at any time */
long long hsqp_debug_read(hsqp_handle* h, int what, void* dst, long long bytes);

/* Elapsed device time (ms) of the kernels of the last hsqp_iterate_device call,
 * measured with HIP events on the handle's stream: {lq, project, riccati, step, total}. */
int hsqp_last_kernel_ms(hsqp_handle* h, double out_ms[5]);

/* ---- the step after the solve -------------------------------------------------------------
 * Joint torques as the reference's computeJointTorques (humanoid_common_mpc/src/pinocchio_model/DynamicsHelperFunctions.cpp:
 * 233-270): tau_j = M_j [a_b; qdd_j] + nle_j - (sum J^T W)_j with a_b from the flow map's base solve, for n (x, u) pairs. */
int hsqp_joint_torques(hsqp_handle* h, int n, const double* x /*[n][58]*/, const double* u /*[n][35]*/, double* tau /*[n][23]*/);
/* Centroidal handle (CentroidalMpcMrtJointController::computeJointControlAction, humanoid_centroidal_mpc/src/mrt/
 * CentroidalMpcMrtJointController.cpp:155-175): the same computeJointTorques with q = the state's generalized coordinates, the
 * generalized velocities from the centroidal momentum (v_b = A_b^-1 (m h - A_j qd_j)), the input's contact wrenches, and the DESIRED
 * joint accelerations the controller forms from its PD law, passed in entries 35..57 of the state row (zeros: pure feed-forward);
 * hsqp_evaluate_policy on a centroidal handle uses zeros there. */
/* Feed-forward policy of the solution resident on the device (MPC_MRT_Interface::evaluatePolicy as used in
 * humanoid_wb_mpc/src/mrt/WBMpcMrtJointController.cpp:136-147): clamped linear interpolation of the optimal state / input
 * trajectories at s[b] seconds after the first node, and the joint torques there.  Any output may be NULL. */
int hsqp_evaluate_policy(hsqp_handle* h, const double* s /*[B]*/, double* x /*[B][58]*/, double* u /*[B][35]*/, double* tau /*[B][23]*/);

const char* hsqp_last_error(const hsqp_handle* h);   /* h may be NULL: last creation error */
/* Iterations since hsqp_create whose parallel-in-time sweep failed the KKT gate and were redone with the serial recursion (-1: h is NULL). */
long long hsqp_scan_fallbacks(const hsqp_handle* h);
/* Iterations that ran the serial recursion because the AUTOMATIC sweep choice was backing off after a rejected gate (a sweep forced by
 * HSQP_FLAG_PARALLEL_RICCATI / HSQP_FLAG_SEGMENTED_RICCATI is attempted every iteration; every upload resets the back-off unless
 * hsqp_set_scan_backoff_persistent is on). */
long long hsqp_scan_backoffs(const hsqp_handle* h);
/* The gate's back-off across uploads.  Default (0): every hsqp_upload* starts a new problem and resets it — a handle's history never decides
 * which sweep a problem takes.  1 (what a receding-horizon caller wants: ocs2's MPC_BASE::run uploads the shifted problem every cycle and
 * runs sqpIteration = 1): the back-off survives uploads of the same (batch, n_nodes), so that a regime whose iterates keep failing the gate
 * pays the rejected parallel-in-time sweep once in a while (after 1, 3, 7, .. 63 cycles) and not in every cycle; a change of shape resets it.
 * host/HipSqpSolver.h switches it on. */
int hsqp_set_scan_backoff_persistent(hsqp_handle* h, int on);
const char* hsqp_version(void);
/* Binary interface revision: bumped whenever a public struct or an entry point's meaning changes (5: hsqp_linesearch_settings::cost_tol,
 * hsqp_set_scan_backoff_persistent; 6: hsqp_comm_*, HSQP_BLK_FORMS[2]).  A caller compares hsqp_abi_version() with the HSQP_ABI_VERSION it was compiled against before it
 * passes structs (host/HipSqpSolver.h and the Python binding do). */
#define HSQP_ABI_VERSION 6
int hsqp_abi_version(void);
int hsqp_device_count(void);

/* ---- the batch axis over the GPUs of one node: one process per GPU, contiguous blocks of ceil(B / world) instances.
 * The reference has no counterpart (its solver runs ONE instance on the host: /root/reference/humanoid_nmpc/humanoid_wb_mpc/src/
 * WBMpcInterface.cpp:113-121 constructs one ocs2::SqpMpc); BASELINE.json's north_star asks for "a batch axis over independent MPC
 * instances ... sharded across the 8 GPUs of one node with RCCL over xGMI only for the batch broadcast/gather".  These entry points are
 * that exchange for a host without torch.distributed: every pointer named d_* is DEVICE memory of the communicator's GPU, so a shard goes
 * HBM -> xGMI -> HBM -> hsqp_upload_device without host staging.  Nothing here runs inside a solve.  RCCL is loaded at the first call
 * (librccl.so.1; HSQP_RCCL_LIB in the environment names another copy): a single-GPU host never needs it.
 *   rank 0:  hsqp_comm_unique_id(id)  -> ship the 128 bytes to the other processes (the host's own means: a file, a socket, MPI)
 *   all:     hsqp_comm_create(&c, id, rank, world, device)
 *            hsqp_comm_broadcast(c, d_shared, bytes, 0)                       shared problem image (model constants, shared node parameters)
 *            hsqp_comm_scatter_rows(c, d_global_x, d_x, (N + 1) * 58, B, 0)   one call per array of hsqp_problem; then hsqp_upload_device
 *            ... hsqp_iterate_device ... hsqp_download_device ...
 *            hsqp_comm_gather_rows(c, d_x, d_global_x, (N + 1) * 58, B, 0)    the solutions back to rank 0
 * All calls are collective (every rank of the communicator makes the same call with the same sizes) and return once the transfer is
 * complete on this rank.  Errors: the HSQP_ERR_* codes; text through hsqp_comm_last_error (hsqp_comm_create_error for the two creators). */
typedef struct hsqp_comm hsqp_comm;
#define HSQP_COMM_ID_BYTES 128
int hsqp_comm_unique_id(void* id /* HSQP_COMM_ID_BYTES */);
int hsqp_comm_create(hsqp_comm** out, const void* id, int rank, int world, int device);
void hsqp_comm_destroy(hsqp_comm* c);
const char* hsqp_comm_create_error(void);
const char* hsqp_comm_last_error(const hsqp_comm* c);
int hsqp_comm_rank(const hsqp_comm* c);
int hsqp_comm_world(const hsqp_comm* c);
/* the block [lo, hi) of a global batch this rank owns (the same split the scatter / gather use; the last ranks may own nothing) */
int hsqp_comm_shard(const hsqp_comm* c, int global_batch, int* lo, int* hi);
int hsqp_comm_shard_of(int global_batch, int world, int rank, int* lo, int* hi);   /* the same split without a communicator */
int hsqp_comm_broadcast(hsqp_comm* c, void* d_buf, long long bytes, int root);
/* d_global (root only; may be NULL elsewhere): [global_batch][row_doubles]; d_local: [hi - lo][row_doubles] */
int hsqp_comm_scatter_rows(hsqp_comm* c, const double* d_global, double* d_local, long long row_doubles, int global_batch, int root);
int hsqp_comm_gather_rows(hsqp_comm* c, const double* d_local, double* d_global, long long row_doubles, int global_batch, int root);
/* element-wise maximum over the ranks of n HOST values, in place (max-over-ranks timing); hsqp_comm_barrier is the same with one value */
int hsqp_comm_max(hsqp_comm* c, double* values, int n);
int hsqp_comm_barrier(hsqp_comm* c);

#ifdef __cplusplus
}
#endif
#endif /* HSQP_H */
