"""TEST INFRASTRUCTURE: ctypes front-end of oracle/_ref/libref_terms.so — more of the REFERENCE'S OWN code compiled in place from
/root/reference (oracle/Makefile target `ref`, oracle/ref_terms_driver.cpp): the whole-body robot model's index maps, the friction-cone
and zero-wrench constraints, the switched-model reference manager (contact flags, gait phase, arm-swing reference), the end-effector
weights loader.  Only tests/ and tests/golden/make_ref_terms_golden.py import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_ref", "libref_terms.so")
_dp, _ip = C.POINTER(C.c_double), C.POINTER(C.c_int)


def available():
    if os.path.exists(LIB):
        return True
    if os.path.isdir("/root/reference/humanoid_nmpc/humanoid_common_mpc/src/constraint"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
        return os.path.exists(LIB)
    return False


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class RefTerms:
    def __init__(self, nj=23):
        if not available():
            raise RuntimeError("oracle/_ref/libref_terms.so is missing and /root/reference is not mounted")
        self.lib = C.CDLL(LIB)
        self.nj = nj
        self.nx, self.nu = 2 * (6 + nj), 12 + nj

    def layout(self):
        out = (C.c_int * 12)()
        self.lib.ref_wb_layout(self.nj, out)
        keys = ["state_dim", "input_dim", "base_start", "joint_start", "joint_velocity_start", "gen_coordinates_dim", "wrench_start_0", "wrench_start_1",
                "force_start_0", "force_start_1", "moment_start_0", "moment_start_1"]
        return dict(zip(keys, list(out)))

    def accessors(self, x, u):
        nj = self.nj
        sizes = [("base_pose", 6), ("joint_angles", nj), ("base_lin_vel", 3), ("base_vel", 6), ("joint_velocities", nj), ("gen_coordinates", 6 + nj),
                 ("gen_velocities", 6 + nj), ("wrench_0", 6), ("wrench_1", 6), ("force_0", 3), ("moment_1", 3)]
        out = np.zeros(sum(n for _, n in sizes))
        self.lib.ref_wb_accessors(nj, _d(x).ctypes.data_as(_dp), _d(u).ctypes.data_as(_dp), out.ctypes.data_as(_dp))
        res, o = {}, 0
        for k, n in sizes:
            res[k] = out[o:o + n].copy(); o += n
        return res

    def friction_cone(self, cfg, contact, event_times, mode_sequence, time, x, u):
        """cfg = (frictionCoefficient, regularization, gripperForce, hessianDiagonalShift) -> dict(f, dfdu, dfduu, dfdxx_diag, active)."""
        ev, seq = _d(event_times), _i(mode_sequence)
        f, dfdu, dfduu, dxx = np.zeros(1), np.zeros(self.nu), np.zeros((self.nu, self.nu)), np.zeros(self.nx)
        act = C.c_int(0)
        rc = self.lib.ref_friction_cone(self.nj, _d(cfg).ctypes.data_as(_dp), int(contact), len(ev), ev.ctypes.data_as(_dp), seq.ctypes.data_as(_ip), C.c_double(time),
                                        _d(x).ctypes.data_as(_dp), _d(u).ctypes.data_as(_dp), f.ctypes.data_as(_dp), dfdu.ctypes.data_as(_dp), dfduu.ctypes.data_as(_dp),
                                        dxx.ctypes.data_as(_dp), C.byref(act))
        assert rc == 0, rc
        return dict(f=f[0], dfdu=dfdu, dfduu=dfduu, dfdxx_diag=dxx, active=bool(act.value))

    def zero_wrench(self, contact, event_times, mode_sequence, time, x, u):
        ev, seq = _d(event_times), _i(mode_sequence)
        f, dfdu = np.zeros(6), np.zeros((6, self.nu))
        act = C.c_int(0)
        rc = self.lib.ref_zero_wrench(self.nj, int(contact), len(ev), ev.ctypes.data_as(_dp), seq.ctypes.data_as(_ip), C.c_double(time), _d(x).ctypes.data_as(_dp),
                                      _d(u).ctypes.data_as(_dp), f.ctypes.data_as(_dp), dfdu.ctypes.data_as(_dp), C.byref(act))
        assert rc == 0, rc
        return dict(f=f, dfdu=dfdu, active=bool(act.value))

    def desired_state(self, arm_joints, event_times, mode_sequence, target_times, target_states, arm_swing, t0, tf, state, time):
        """(xnom[nx], phase variable, contact flags) after preSolverRun(t0, tf, state)."""
        ev, seq, tt, ts = _d(event_times), _i(mode_sequence), _d(target_times), _d(target_states)
        xn, ph, fl = np.zeros(self.nx), C.c_double(0.0), (C.c_int * 2)()
        rc = self.lib.ref_desired_state(self.nj, _i(arm_joints).ctypes.data_as(_ip), len(ev), ev.ctypes.data_as(_dp), seq.ctypes.data_as(_ip), len(tt), tt.ctypes.data_as(_dp),
                                        ts.ctypes.data_as(_dp), int(bool(arm_swing)), C.c_double(t0), C.c_double(tf), _d(state).ctypes.data_as(_dp), C.c_double(time),
                                        xn.ctypes.data_as(_dp), C.byref(ph), fl)
        assert rc == 0, rc
        return xn, ph.value, (bool(fl[0]), bool(fl[1]))

    def foot_weights(self, task_file, prefix):
        w = np.zeros(18)
        rc = self.lib.ref_foot_weights(task_file.encode(), prefix.encode(), w.ctypes.data_as(_dp))
        assert rc == 0, rc
        return w

    def velocity_targets(self, reference_info, horizon, cmd, t0, x0, calls=200):
        """WBMpcTargetTrajectoriesCalculator::commandedVelocityToTargetTrajectories called `calls` times with the same arguments (its command
        filter is a function-local static: 0.8^200 ~ 4e-20 leaves the steady state) -> (times[3], states[3, nx])."""
        t, s = np.zeros(3), np.zeros((3, self.nx))
        rc = self.lib.ref_wb_velocity_targets(self.nj, reference_info.encode(), C.c_double(horizon), _d(cmd).ctypes.data_as(_dp), C.c_double(t0), _d(x0).ctypes.data_as(_dp),
                                              int(calls), t.ctypes.data_as(_dp), s.ctypes.data_as(_dp))
        assert rc == 0, rc
        return t, s

    # ---- round 4: the assembly files (ref_terms_driver.cpp, second half) -------------------------------------------------------------
    GAIN_KEYS = ("gain_pos_z", "gain_ori", "gain_linvel_z", "gain_linvel_xy", "gain_angvel", "gain_linacc_z", "gain_linacc_xy", "gain_angacc")   # FootConstraintConfig order

    def stance_foot_constraint(self, gains, contact, event_times, mode_sequence, time, kin18, jac):
        """ZeroAccelerationConstraintCppAd / EndEffectorDynamicsAccelerationsConstraint over handed-in kinematics -> dict(value, f, dfdx, dfdu, active)."""
        ev, seq = _d(event_times), _i(mode_sequence)
        value, f, dfdx, dfdu = np.zeros(6), np.zeros(6), np.zeros((6, self.nx)), np.zeros((6, self.nu))
        act = C.c_int(0)
        rc = self.lib.ref_stance_foot_constraint(self.nj, _d(gains).ctypes.data_as(_dp), int(contact), len(ev), ev.ctypes.data_as(_dp), seq.ctypes.data_as(_ip), C.c_double(time),
                                                 _d(kin18).ctypes.data_as(_dp), _d(jac).ctypes.data_as(_dp), value.ctypes.data_as(_dp), f.ctypes.data_as(_dp),
                                                 dfdx.ctypes.data_as(_dp), dfdu.ctypes.data_as(_dp), C.byref(act))
        assert rc == 0, rc
        return dict(value=value, f=f, dfdx=dfdx, dfdu=dfdu, active=bool(act.value))

    def swing_foot_constraint(self, gains, zref, kin18, jac):
        """EndEffectorDynamicsLinearAccConstraint (one row) with WBMpcPreComputation's per-node configuration -> dict(value, f, dfdx, dfdu)."""
        value, f, dfdx, dfdu = np.zeros(1), np.zeros(1), np.zeros((1, self.nx)), np.zeros((1, self.nu))
        rc = self.lib.ref_swing_foot_constraint(self.nj, _d(gains).ctypes.data_as(_dp), _d(zref).ctypes.data_as(_dp), _d(kin18).ctypes.data_as(_dp), _d(jac).ctypes.data_as(_dp),
                                                value.ctypes.data_as(_dp), f.ctypes.data_as(_dp), dfdx.ctypes.data_as(_dp), dfdu.ctypes.data_as(_dp))
        assert rc == 0, rc
        return dict(value=value, f=f, dfdx=dfdx, dfdu=dfdu)

    def state_input_quadratic_cost(self, arm_joints, total_mass, event_times, mode_sequence, target_times, target_states, arm_swing, t0, tf, Q, R, state, input, time):
        """StateInputQuadraticCost -> (dx, du, value): the deviation its quadratic form is taken of, and 1/2 dx'Q dx + 1/2 du'R du."""
        ev, seq, tt, ts = _d(event_times), _i(mode_sequence), _d(target_times), _d(target_states)
        dx, du, val = np.zeros(self.nx), np.zeros(self.nu), C.c_double(0.0)
        rc = self.lib.ref_state_input_quadratic_cost(self.nj, _i(arm_joints).ctypes.data_as(_ip), C.c_double(total_mass), len(ev), ev.ctypes.data_as(_dp), seq.ctypes.data_as(_ip),
                                                     len(tt), tt.ctypes.data_as(_dp), ts.ctypes.data_as(_dp), int(bool(arm_swing)), C.c_double(t0), C.c_double(tf),
                                                     _d(Q).ctypes.data_as(_dp), _d(R).ctypes.data_as(_dp), _d(state).ctypes.data_as(_dp), _d(input).ctypes.data_as(_dp),
                                                     C.c_double(time), dx.ctypes.data_as(_dp), du.ctypes.data_as(_dp), C.byref(val))
        assert rc == 0, rc
        return dx, du, val.value

    def joint_limits(self, q_lo, q_hi, mu, delta, state):
        f, dfdx, dxx = C.c_double(0.0), np.zeros(self.nx), np.zeros(self.nx)
        rc = self.lib.ref_joint_limits(self.nj, _d(q_lo).ctypes.data_as(_dp), _d(q_hi).ctypes.data_as(_dp), C.c_double(mu), C.c_double(delta), _d(state).ctypes.data_as(_dp),
                                       C.byref(f), dfdx.ctypes.data_as(_dp), dxx.ctypes.data_as(_dp))
        assert rc == 0, rc
        return f.value, dfdx, dxx

    def weight_comp_initializer(self, total_mass, event_times, mode_sequence, time, next_time, state):
        ev, seq = _d(event_times), _i(mode_sequence)
        u, xn = np.zeros(self.nu), np.zeros(self.nx)
        rc = self.lib.ref_weight_comp_initializer(self.nj, C.c_double(total_mass), len(ev), ev.ctypes.data_as(_dp), seq.ctypes.data_as(_ip), C.c_double(time), C.c_double(next_time),
                                                  _d(state).ctypes.data_as(_dp), u.ctypes.data_as(_dp), xn.ctypes.data_as(_dp))
        assert rc == 0, rc
        return u, xn
