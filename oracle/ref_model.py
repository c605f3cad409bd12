"""TEST INFRASTRUCTURE: ctypes front-end of oracle/_ref/libref_model.so — the reference's ASSEMBLY of rigid-body quantities (flow map from CRBA's M,
nle and the foot Jacobians; contact-moment rows; foot-collision pairs; foot-cost residual) compiled in place from /root/reference against a mock of
Pinocchio that returns what the caller hands in (oracle/Makefile target `ref`, oracle/ref_model_driver.cpp, oracle/ref_stubs_model/).  Only tests/ and
tests/golden/make_ref_model_golden.py import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_ref", "libref_model.so")
_dp, _ip = C.POINTER(C.c_double), C.POINTER(C.c_int)


def available():
    if os.path.exists(LIB):
        return True
    if os.path.isdir("/root/reference/humanoid_nmpc/humanoid_common_mpc/src/constraint"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "_ref/libref_model.so"])
        return os.path.exists(LIB)
    return False


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class RefModel:
    def __init__(self, nj=23):
        if not available():
            raise RuntimeError("oracle/_ref/libref_model.so is missing and /root/reference is not mounted")
        self.lib = C.CDLL(LIB)
        self.nj, self.nv = nj, 6 + nj

    def state_derivative(self, M, nle, Jl, Jr, x, u):
        """The reference's computeStateDerivative / computeBaseAcceleration with CRBA's M (nv x nv), nle (nv) and the LOCAL_WORLD_ALIGNED Jacobians of
        the two contact frames (6 x nv, linear rows first) handed in -> (xdot[2 nv], ab[6])."""
        xdot, ab = np.zeros(2 * self.nv), np.zeros(6)
        rc = self.lib.refm_state_derivative(self.nj, _d(M).ctypes.data_as(_dp), _d(nle).ctypes.data_as(_dp), _d(Jl).ctypes.data_as(_dp), _d(Jr).ctypes.data_as(_dp),
                                            _d(x).ctypes.data_as(_dp), _d(u).ctypes.data_as(_dp), xdot.ctypes.data_as(_dp), ab.ctypes.data_as(_dp))
        assert rc == 0, rc
        return xdot, ab

    def foot_collision(self, pos, radii, event_times, mode_sequence, time, x):
        """FootCollisionConstraint on frame positions pos[10][3] (ankle_l, ankle_r, foot_l, foot_r, l_p1, r_p1, l_p2, r_p2, knee_l, knee_r) -> (h[16], active)."""
        h, act, n = np.zeros(16), C.c_int(0), C.c_int(0)
        ev, ms = _d(event_times), _i(mode_sequence)
        rc = self.lib.refm_foot_collision(self.nj, _d(pos).ctypes.data_as(_dp), _d(radii).ctypes.data_as(_dp), len(ev), ev.ctypes.data_as(_dp), ms.ctypes.data_as(_ip),
                                          C.c_double(time), _d(x).ctypes.data_as(_dp), h.ctypes.data_as(_dp), C.byref(act), C.byref(n))
        assert rc == 0 and n.value == 16, (rc, n.value)
        return h, bool(act.value)

    def contact_moment(self, contact, Rf, rect, event_times, mode_sequence, time, x, u):
        """ContactMomentXYConstraintCppAd of a contact on its frame's rotation Rf (local -> world), rect = (x_min, x_max, y_min, y_max) -> (h[4], active)."""
        h, act = np.zeros(4), C.c_int(0)
        ev, ms = _d(event_times), _i(mode_sequence)
        rc = self.lib.refm_contact_moment(self.nj, contact, _d(Rf).ctypes.data_as(_dp), _d(rect).ctypes.data_as(_dp), len(ev), ev.ctypes.data_as(_dp), ms.ctypes.data_as(_ip),
                                          C.c_double(time), _d(x).ctypes.data_as(_dp), _d(u).ctypes.data_as(_dp), h.ctypes.data_as(_dp), C.byref(act))
        assert rc == 0, rc
        return h, bool(act.value)

    def foot_cost(self, contact, R, vlin, vang, alin, aang, params37, x, u):
        """Residual r[18] of EndEffectorDynamicsFootCost::costVectorFunction on the handed-in quantities of the contact frame."""
        r = np.zeros(18)
        rc = self.lib.refm_foot_cost(self.nj, contact, _d(R).ctypes.data_as(_dp), _d(vlin).ctypes.data_as(_dp), _d(vang).ctypes.data_as(_dp), _d(alin).ctypes.data_as(_dp),
                                     _d(aang).ctypes.data_as(_dp), _d(params37).ctypes.data_as(_dp), _d(x).ctypes.data_as(_dp), _d(u).ctypes.data_as(_dp), r.ctypes.data_as(_dp))
        assert rc == 0, rc
        return r
