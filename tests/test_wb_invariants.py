"""Physical invariants of the whole-body flow map (oracle and kernel sources compiled for the host): world translation invariance and
equivariance under a rotation of the world about the vertical.  Independent of either restatement's algebra, they pin the frame
conventions (world-frame base velocities and wrenches, ZYX euler rates) that no reference test pins."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import random_state_input
from test_centroidal_invariants import rotz

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
P = lambda a: a.ctypes.data_as(_dp)  # noqa: E731
NV = 29


def rotate_world(x, u, a):
    R = rotz(a)
    x2, u2 = x.copy(), u.copy()
    x2[0:3] = R @ x[0:3]
    x2[3] = x[3] + a
    x2[NV:NV + 3] = R @ x[NV:NV + 3]
    for f in range(2):
        u2[6 * f:6 * f + 3], u2[6 * f + 3:6 * f + 6] = R @ u[6 * f:6 * f + 3], R @ u[6 * f + 3:6 * f + 6]
    return x2, u2


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_flow_map_invariants(model, oracle, seed):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "hostemu"), "all"])
    lib = C.CDLL(os.path.join(HERE, "hostemu", "libhsqp_hostemu.so"))
    lib.emu_create.restype = C.c_void_p
    err = C.create_string_buffer(256)
    h = C.c_void_p(lib.emu_create(C.byref(model.desc), err, 256))
    assert h.value, err.value

    def emu_ab(x, u):
        ab = np.zeros(6)
        lib.emu_stage_eval(h, P(np.ascontiguousarray(x)), P(np.ascontiguousarray(u)), 0, P(ab), None)
        return ab

    rng = np.random.default_rng(seed)
    x, u = random_state_input(model, rng)
    for name, f in (("oracle", lambda a, b: oracle.flow_map(a, b)[NV:NV + 6]), ("kernel sources", emu_ab)):
        a0 = f(x, u)
        xt = x.copy()
        xt[:3] += np.array([-4.0, 1.5, 0.3])
        assert np.abs(f(xt, u) - a0).max() <= 1e-10 * max(1.0, np.abs(a0).max()), name
        ang = -1.1
        xr, ur = rotate_world(x, u, ang)
        ar = f(xr, ur)
        want = a0.copy()
        want[:3] = rotz(ang) @ a0[:3]           # linear base acceleration is a world vector; euler accelerations are not
        assert np.abs(ar - want).max() <= 1e-9 * max(1.0, np.abs(a0).max()), name
    lib.emu_destroy(h)
