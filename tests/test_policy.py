"""Policy evaluation and joint torques (hsqp_policy.h): the device code path against the oracle's full
joint-space dynamics — tau_j = M_j [a_b; qdd_j] + nle_j - (J_l^T W_l + J_r^T W_r)_j, the reference's computeJointTorques
(humanoid_common_mpc/src/pinocchio_model/DynamicsHelperFunctions.cpp:233-270)."""
import ctypes as C

import numpy as np
import pytest

from conftest import random_state_input
from test_hostemu import emu  # noqa: F401  (fixture)
from wb_humanoid_mpc_amd import _abi
from wb_humanoid_mpc_amd.reference import make_problem

NX, NU, NV, NJ = _abi.NX, _abi.NU, _abi.NV, 23
_dp = C.POINTER(C.c_double)
P = lambda a: a.ctypes.data_as(_dp)  # noqa: E731


def oracle_torques(oracle, x, u):
    M, nle = oracle.full_dynamics(x)
    ab, _, _ = oracle.base_dynamics(x, u)
    _, _, J = oracle.foot_kinematics(x, u, jac=True)
    # frame Jacobians (LOCAL_WORLD_ALIGNED, rows [v_lin; v_ang]) = d(vlin, vang)/d(generalized velocities)
    ext = np.zeros(NV)
    for f in range(2):
        Jf = J[f][6:12, NV:NX]
        ext += Jf.T @ u[6 * f:6 * f + 6]
    qdd = np.concatenate([ab, u[12:]])
    return M[6:] @ qdd + nle[6:] - ext[6:]


def test_joint_torques_equal_the_full_inverse_dynamics(model, oracle, emu, rng):  # noqa: F811
    lib, h = emu
    for _ in range(6):
        x, u = random_state_input(model, rng)
        tau = np.zeros(NJ)
        lib.emu_joint_torques(h, P(x), P(u), P(tau))
        want = oracle_torques(oracle, x, u)
        assert np.abs(tau - want).max() <= 1e-9 * max(1.0, np.abs(want).max())


def test_standing_torques_balance_gravity(model, oracle, emu):  # noqa: F811
    """At rest with weight-compensating contact forces the torques are the static gravity/contact balance (finite, leg joints loaded)."""
    lib, h = emu
    x0, x, u, par, dt = make_problem(model, n_nodes=2, batch=1, gait="stance", v_cmd=(0, 0, 0.7925, 0))
    tau = np.zeros(NJ)
    lib.emu_joint_torques(h, P(np.ascontiguousarray(x[0, 0])), P(np.ascontiguousarray(u[0, 0])), P(tau))
    want = oracle_torques(oracle, x[0, 0], u[0, 0])
    assert np.abs(tau - want).max() <= 1e-9 * max(1.0, np.abs(want).max())
    assert np.abs(tau[:12]).max() > 1.0


def test_feed_forward_policy_interpolation(emu, rng):  # noqa: F811
    lib, _ = emu
    N, dt = 5, 0.035
    xt, ut = rng.standard_normal((N + 1, NX)), rng.standard_normal((N, NU))
    for s in (-0.01, 0.0, 0.005, 0.5 * dt, 2.3 * dt, (N - 1) * dt + 0.01, N * dt, N * dt + 1.0):
        x, u = np.zeros(NX), np.zeros(NU)
        lib.emu_policy_interpolate(P(xt), P(ut), N, C.c_double(dt), C.c_double(s), P(x), P(u))
        t = np.arange(N + 1) * dt
        sc = min(max(s, 0.0), N * dt)
        xw = np.array([np.interp(sc, t, xt[:, i]) for i in range(NX)])
        ue = np.vstack([ut, ut[-1:]])                 # ocs2 repeats the last input at the final time stamp
        uw = np.array([np.interp(sc, t, ue[:, i]) for i in range(NU)])
        np.testing.assert_allclose(x, xw, rtol=0, atol=1e-13)
        np.testing.assert_allclose(u, uw, rtol=0, atol=1e-13)


@pytest.mark.gpu
def test_policy_and_torques_on_the_device(model, oracle):
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    B, N = 3, 8
    x0, x, u, par, dt = make_problem(model, n_nodes=N, batch=B, perturb=True, seed=3)
    s = HipSqpSolver(model, max_nodes=N, max_batch=B)
    try:
        out = s.run(x0, x, u, par, dt)
        tq = np.array([0.005, 1.7 * dt, N * dt])
        xp, up, tau = s.evaluate_policy(tq)
        t = np.arange(N + 1) * dt
        for b in range(B):
            xw = np.array([np.interp(tq[b], t, out["x"][b][:, i]) for i in range(NX)])
            ue = np.vstack([out["u"][b], out["u"][b][-1:]])
            uw = np.array([np.interp(tq[b], t, ue[:, i]) for i in range(NU)])
            np.testing.assert_allclose(xp[b], xw, rtol=0, atol=1e-12)
            np.testing.assert_allclose(up[b], uw, rtol=0, atol=1e-9)
            want = oracle_torques(oracle, xp[b], up[b])
            assert np.abs(tau[b] - want).max() <= 1e-9 * max(1.0, np.abs(want).max())
        tau2 = s.joint_torques(xp, up)
        assert np.array_equal(tau2, tau)
    finally:
        s.close()
