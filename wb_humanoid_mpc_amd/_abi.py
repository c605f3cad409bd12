"""ctypes mirror of include/hsqp.h (the C ABI of the HIP SQP library).

Field order and sizes must match the header exactly; tests/test_abi.py checks
sizeof() of every struct against the values the C library reports.
"""
import ctypes as C

NJ, NV, NX, NU, NB = 23, 29, 58, 35, 24
NZ = NX + NU
NE_MAX = 14
NODE_PARAMS = 72
P_XDES, P_ARMSWING, P_CONTACT, P_SWING, P_IMPACT = 0, 58, 59, 61, 67
FORM_WB, FORM_CENTROIDAL = 0, 1
CNX, PC_TORSO = 35, 35

ABI_VERSION = 6   # HSQP_ABI_VERSION of include/hsqp.h (tests/test_abi.py compares them)

OK, ERR_BAD_ARG, ERR_NO_DEVICE, ERR_OOM, ERR_NUMERIC, ERR_HIP, ERR_NOT_CONVERGED = 0, -1, -2, -3, -4, -5, -6

BLK_AB, BLK_BVEC, BLK_H, BLK_G, BLK_CDE, BLK_NE, BLK_COST, BLK_DX, BLK_DU, BLK_FLOW = range(1, 11)
BLK_PARAMS, BLK_FORMS = 11, 12
COMM_ID_BYTES = 128   # HSQP_COMM_ID_BYTES


class Body(C.Structure):
    _fields_ = [("parent", C.c_int32), ("reserved", C.c_int32), ("R", C.c_double * 9), ("p", C.c_double * 3),
                ("axis", C.c_double * 3), ("mass", C.c_double), ("com", C.c_double * 3),
                ("inertia", C.c_double * 9), ("q_lo", C.c_double), ("q_hi", C.c_double)]


class Frame(C.Structure):
    _fields_ = [("body", C.c_int32), ("reserved", C.c_int32), ("p", C.c_double * 3)]


class Barrier(C.Structure):
    _fields_ = [("mu", C.c_double), ("delta", C.c_double)]


class ModelDesc(C.Structure):
    _fields_ = [("formulation", C.c_int32), ("n_joints", C.c_int32), ("bodies", Body * NB),
                ("contact", Frame * 2), ("collision_p1", Frame * 2), ("collision_p2", Frame * 2),
                ("ankle", Frame * 2), ("knee", Frame * 2), ("gravity", C.c_double),
                ("Q", C.c_double * NX), ("R", C.c_double * NU), ("Qf", C.c_double * NX),
                ("foot_sqrt_w", C.c_double * 18),
                ("gain_pos_z", C.c_double), ("gain_ori", C.c_double), ("gain_linvel_z", C.c_double),
                ("gain_linvel_xy", C.c_double), ("gain_angvel", C.c_double), ("gain_linacc_z", C.c_double),
                ("gain_linacc_xy", C.c_double), ("gain_angacc", C.c_double),
                ("friction_mu", C.c_double), ("friction_reg", C.c_double), ("friction_grip", C.c_double),
                ("friction_hess_shift", C.c_double), ("friction_barrier", Barrier),
                ("rect_x_min", C.c_double), ("rect_x_max", C.c_double), ("rect_y_min", C.c_double),
                ("rect_y_max", C.c_double), ("moment_barrier", Barrier), ("joint_limit_barrier", Barrier),
                ("r_foot", C.c_double), ("r_knee", C.c_double), ("collision_barrier", Barrier),
                ("arm_swing_joint", C.c_int32 * 4),
                ("torso", Frame), ("torso_R", C.c_double * 9), ("torso_sqrt_w", C.c_double * 12),
                ("cent_foot_sqrt_w", C.c_double * 12), ("ext_torque_sqrt_w", (C.c_double * 6) * 2),
                ("ext_torque_joint", (C.c_int32 * 6) * 2)]


class Settings(C.Structure):
    _fields_ = [("max_nodes", C.c_int32), ("max_batch", C.c_int32), ("device", C.c_int32), ("flags", C.c_int32)]


class Problem(C.Structure):
    _fields_ = [("batch", C.c_int32), ("n_nodes", C.c_int32), ("dt", C.c_double),
                ("x_init", C.POINTER(C.c_double)), ("x_traj", C.POINTER(C.c_double)),
                ("u_traj", C.POINTER(C.c_double)), ("node_params", C.POINTER(C.c_double)), ("dt_nodes", C.POINTER(C.c_double))]


class Perf(C.Structure):
    _fields_ = [("merit", C.c_double), ("cost", C.c_double), ("dynamics_sse", C.c_double),
                ("equality_sse", C.c_double)]


class Timings(C.Structure):
    _fields_ = [("lq_approximation", C.c_double), ("solve_qp", C.c_double), ("linesearch", C.c_double),
                ("compute_controller", C.c_double), ("total", C.c_double)]


class Solution(C.Structure):
    _fields_ = [("x", C.POINTER(C.c_double)), ("u", C.POINTER(C.c_double)), ("dx", C.POINTER(C.c_double)),
                ("du", C.POINTER(C.c_double)), ("perf_before", C.POINTER(Perf)), ("perf_after", C.POINTER(Perf)),
                ("kkt", C.POINTER(C.c_double)), ("alpha", C.POINTER(C.c_double)), ("step_type", C.POINTER(C.c_int32)),
                ("armijo", C.POINTER(C.c_double)), ("grad_inf", C.POINTER(C.c_double)), ("timings", Timings)]


class SwingConfig(C.Structure):
    _fields_ = [("lift_off_velocity", C.c_double), ("touch_down_velocity", C.c_double), ("swing_height", C.c_double),
                ("touch_down_height_offset", C.c_double), ("swing_time_scale", C.c_double), ("impact_mid", C.c_double),
                ("impact_lift_velocity", C.c_double), ("impact_touch_velocity", C.c_double)]


class Reference(C.Structure):
    _fields_ = [("batch", C.c_int32), ("n_nodes", C.c_int32), ("t0", C.c_double), ("dt", C.c_double), ("max_events", C.c_int32),
                ("n_events", C.POINTER(C.c_int32)), ("event_times", C.POINTER(C.c_double)), ("mode_sequence", C.POINTER(C.c_int32)),
                ("n_knots", C.c_int32), ("target_times", C.POINTER(C.c_double)), ("target_states", C.POINTER(C.c_double)),
                ("swing", SwingConfig), ("terrain_height", C.c_double), ("arm_swing", C.c_int32), ("reserved", C.c_int32),
                ("node_times", C.POINTER(C.c_double))]


class LinesearchSettings(C.Structure):
    _fields_ = [("g_max", C.c_double), ("g_min", C.c_double), ("gamma_c", C.c_double), ("armijo_factor", C.c_double),
                ("alpha_decay", C.c_double), ("alpha_min", C.c_double), ("delta_tol", C.c_double), ("cost_tol", C.c_double)]


class TermWeights(C.Structure):
    _fields_ = [("foot_sqrt_w", C.c_double * 18)] + [(k, C.c_double) for k in ("gain_pos_z", "gain_ori", "gain_linvel_z", "gain_linvel_xy", "gain_angvel", "gain_linacc_z",
                                                                               "gain_linacc_xy", "gain_angacc")] + \
               [(k, Barrier) for k in ("friction_barrier", "moment_barrier", "joint_limit_barrier", "collision_barrier")] + \
               [("torso_sqrt_w", C.c_double * 12), ("cent_foot_sqrt_w", C.c_double * 12), ("ext_torque_sqrt_w", (C.c_double * 6) * 2)]


STEP_COST, STEP_DUAL, STEP_CONSTRAINT, STEP_ZERO, STEP_FULL = 0, 1, 2, 3, 4
FLAG_LINESEARCH = 1
FLAG_SERIAL_RICCATI, FLAG_PARALLEL_RICCATI, SCAN_AUTO_BATCH, SCAN_AUTO_MIN_NODES = 2, 4, 2, 48
FLAG_SEGMENTED_RICCATI = 8   # the two-level (segmented) sweep (csrc/hsqp_segment.h): opt-in, declared relaxation of the trajectory tolerance
BLK_PARAMS = 11
