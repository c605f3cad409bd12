"""TEST INFRASTRUCTURE: ctypes front-end of oracle/_ref/libref_swing.so — the REFERENCE'S OWN swing-foot planner and gait
schedule compiled in place from /root/reference (oracle/Makefile target `ref`, oracle/ref_driver.cpp).  Only tests/ and
tests/golden/make_ref_swing_golden.py import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_ref", "libref_swing.so")
_dp, _ip = C.POINTER(C.c_double), C.POINTER(C.c_int)


def available():
    """True when the library is there, or can be built because /root/reference exists (the build container)."""
    if os.path.exists(LIB):
        return True
    if os.path.isdir("/root/reference/humanoid_nmpc/humanoid_common_mpc/src/swing_foot_planner"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
        return os.path.exists(LIB)
    return False


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class RefSwing:
    def __init__(self):
        if not available():
            raise RuntimeError("oracle/_ref/libref_swing.so is missing and /root/reference is not mounted")
        self.lib = C.CDLL(LIB)

    @staticmethod
    def config_vector(cfg):
        """hsqp_swing_config order from task.info's swing_trajectory_config dict."""
        return _d([cfg["liftOffVelocity"], cfg["touchDownVelocity"], cfg["swingHeight"], cfg["touchDownHeightOffset"], cfg["swingTimeScale"],
                   cfg["impactProximityFactorMidPointValue"], cfg["impactProximityFactorLiftOffVelocity"], cfg["impactProximityFactorTouchDownVelocity"]])

    def cubic_spline(self, start, end, t):
        t = _d(t); out = np.zeros((len(t), 3))
        self.lib.ref_cubic_spline(_d(start).ctypes.data_as(_dp), _d(end).ctypes.data_as(_dp), len(t), t.ctypes.data_as(_dp), out.ctypes.data_as(_dp))
        return out

    def spline_cpg(self, lift, mid, touch, t):
        t = _d(t); out = np.zeros((len(t), 3))
        self.lib.ref_spline_cpg(_d(lift).ctypes.data_as(_dp), C.c_double(mid), _d(touch).ctypes.data_as(_dp), len(t), t.ctypes.data_as(_dp), out.ctypes.data_as(_dp))
        return out

    def stance_legs(self, mode):
        f = (C.c_int * 2)()
        self.lib.ref_mode_to_stance_legs(int(mode), f)
        return bool(f[0]), bool(f[1])

    def swing_planner(self, cfg, event_times, mode_sequence, t, terrain_height=0.0):
        """(ok, out[n][2][4] = {z, zdot, zddot, impact proximity} per leg, modes[n])."""
        ev, t = _d(event_times), _d(t)
        seq = np.ascontiguousarray(mode_sequence, dtype=np.int32)
        assert len(seq) == len(ev) + 1
        out, modes = np.zeros((len(t), 2, 4)), np.zeros(len(t), dtype=np.int32)
        rc = self.lib.ref_swing_planner(self.config_vector(cfg).ctypes.data_as(_dp), len(ev), ev.ctypes.data_as(_dp), seq.ctypes.data_as(_ip),
                                        C.c_double(terrain_height), len(t), t.ctypes.data_as(_dp), out.ctypes.data_as(_dp), modes.ctypes.data_as(_ip))
        return rc == 0, out, modes

    def gait_schedule(self, template_times, template_modes, phase_transition_stance_time, insert_start, insert_final, lower, upper, cap=512):
        tt = _d(template_times)
        tm = np.ascontiguousarray(template_modes, dtype=np.int32)
        ev, seq = np.zeros(cap), np.zeros(cap + 1, dtype=np.int32)
        n = self.lib.ref_gait_schedule(len(tm), tt.ctypes.data_as(_dp), tm.ctypes.data_as(_ip), C.c_double(phase_transition_stance_time),
                                       C.c_double(insert_start), C.c_double(insert_final), C.c_double(lower), C.c_double(upper), cap,
                                       ev.ctypes.data_as(_dp), seq.ctypes.data_as(_ip))
        if n < 0:
            raise RuntimeError("reference GaitSchedule threw")
        return ev[:n].copy(), seq[:n + 1].copy()
