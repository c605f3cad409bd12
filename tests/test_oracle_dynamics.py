"""Self-consistency of the oracle's rigid-body restatement (SURVEY.md §8c 'oracle self-consistency checks')."""
import numpy as np
import pytest

from conftest import random_state_input


def fd_jac(fun, z, eps=1e-6):
    f0 = fun(z)
    J = np.zeros((f0.size, z.size))
    for i in range(z.size):
        zp, zm = z.copy(), z.copy()
        zp[i] += eps
        zm[i] -= eps
        J[:, i] = (fun(zp) - fun(zm)) / (2 * eps)
    return J


def test_mass_matrix_rows_and_symmetry(model, oracle, rng):
    x, u = random_state_input(model, rng)
    M, nle = oracle.full_dynamics(x)
    ab, M6, nle6 = oracle.base_dynamics(x, u)
    assert np.allclose(M, M.T, atol=1e-12) and np.linalg.eigvalsh(M).min() > 0
    assert np.allclose(M[:6], M6, atol=1e-12) and np.allclose(nle[:6], nle6, atol=1e-12)
    # translation block is m*I because the Translation joint velocity is expressed in the world frame
    assert np.allclose(M[:3, :3], model.total_mass * np.eye(3), atol=1e-10)


def test_base_acceleration_is_the_block_diagonal_solve(model, oracle, rng):
    # computeBaseAcceleration (common DynamicsHelperFunctions.cpp:197-218): blkdiag(M_lin, M_ang) a_b = -nle_b - M_bj qdd_j + J_b^T W
    x, u = random_state_input(model, rng)
    nj = model.nj
    ab, M6, nle6 = oracle.base_dynamics(x, u)
    out, R = oracle.foot_kinematics(x, u)
    # external base wrench from the two contact frames, built independently: force + moment about the base origin
    # projected on the euler-rate axes
    ez, ey, ex = x[3:6]
    cz, sz, cy, sy = np.cos(ez), np.sin(ez), np.cos(ey), np.sin(ey)
    E = np.array([[0, -sz, cz * cy], [0, cz, sz * cy], [1, 0, -sy]])  # columns: world axes of eulerZ, eulerY, eulerX rates
    tau = np.zeros(6)
    for f in range(2):
        force, moment = u[6 * f:6 * f + 3], u[6 * f + 3:6 * f + 6]
        r = out[f, :3] - x[:3]
        tau[:3] += force
        tau[3:] += E.T @ (moment + np.cross(r, force))
    rhs = -nle6 - M6[:, 6:] @ u[12:] + tau
    assert np.allclose(M6[:3, :3] @ ab[:3], rhs[:3], atol=1e-9)
    assert np.allclose(M6[3:6, 3:6] @ ab[3:], rhs[3:], atol=1e-9)
    # the exact 6x6 solve differs (the reference ignores the lin/ang coupling) — report, do not "fix"
    exact = np.linalg.solve(M6[:, :6], rhs)
    assert np.linalg.norm(exact - ab) > 1e-6


def test_weight_compensation_gives_zero_linear_momentum_rate(model, oracle):
    # Z/test/testDynamicsHelperFunctions.cpp:95-127: weightCompensatingInput = m*9.81 split over stance feet
    x = model.initial_state.copy()
    for flags in ((1, 1), (1, 0), (0, 1)):
        u = np.zeros(model.nu)
        ns = sum(flags)
        for f in range(2):
            if flags[f]:
                u[6 * f + 2] = model.total_mass * 9.81 / ns
        xd = oracle.flow_map(x, u)
        assert np.allclose(xd[29:32], 0.0, atol=1e-10)       # base linear acceleration
        assert np.allclose(xd[:29], x[29:], atol=0) and np.allclose(xd[35:], u[12:], atol=0)


def test_flow_map_jacobian_matches_central_differences(model, oracle, rng):
    x, u = random_state_input(model, rng)
    f, J = oracle.flow_map_jac(x, u)
    assert np.allclose(f, oracle.flow_map(x, u), atol=1e-13)
    Jfd = fd_jac(lambda z: oracle.flow_map(z[:58], z[58:]), np.concatenate([x, u]))
    assert np.abs(J - Jfd).max() <= 1e-6 * max(1.0, np.abs(J).max())
    # structure: d(qdot)/dv = I, d(qdd_j)/du = I, the flow is independent of the base position
    assert np.allclose(J[:29, 29:58], np.eye(29)) and np.allclose(J[35:, 58 + 12:], np.eye(23))
    assert np.allclose(J[:, :3], 0.0)


def test_foot_kinematics_are_time_derivatives_along_the_flow(model, oracle, rng):
    # twist = d/dt position, classical acceleration = d/dt twist along xdot = f(x,u)
    x, u = random_state_input(model, rng)
    xd = oracle.flow_map(x, u)
    eps = 1e-6
    (op, _), (om, _) = oracle.foot_kinematics(x + eps * xd, u), oracle.foot_kinematics(x - eps * xd, u)
    o0, R0 = oracle.foot_kinematics(x, u)
    d = (op - om) / (2 * eps)
    for f in range(2):
        assert np.allclose(d[f, 0:3], o0[f, 6:9], atol=1e-6)       # d pos/dt = linear velocity
        assert np.allclose(d[f, 6:9], o0[f, 12:15], atol=1e-5)     # d vlin/dt = classical linear acceleration
        assert np.allclose(d[f, 9:12], o0[f, 15:18], atol=1e-5)    # d omega/dt = angular acceleration
    # angular velocity is the rotation rate of the frame: Rdot = [omega]x R
    (_, Rp), (_, Rm) = oracle.foot_kinematics(x + eps * xd, u), oracle.foot_kinematics(x - eps * xd, u)
    for f in range(2):
        W = ((Rp[f] - Rm[f]) / (2 * eps)) @ R0[f].T
        assert np.allclose([W[2, 1], W[0, 2], W[1, 0]], o0[f, 9:12], atol=1e-6)


def test_foot_kinematics_jacobians(model, oracle, rng):
    x, u = random_state_input(model, rng)
    out, R, J = oracle.foot_kinematics(x, u, jac=True)
    Jfd = fd_jac(lambda z: oracle.foot_kinematics(z[:58], z[58:])[0].reshape(-1), np.concatenate([x, u]))
    assert np.abs(J.reshape(36, 93) - Jfd).max() <= 2e-6 * max(1.0, np.abs(J).max())
    # position and orientation error depend on q only
    for f in range(2):
        assert np.allclose(J[f, :6, 29:], 0.0)


def test_orientation_error_small_angle_limit(model, oracle):
    # ASSUMPTION A2: e ~ 0.5 * (n x R ez): a foot rolled by +a about world x has error (+a/2, 0, 0)
    x = np.zeros(model.nx)
    x[2] = 0.8
    a = 1e-3
    x[5] = a  # euler X (roll) of the base, legs straight -> foot rolled by a
    out, R = oracle.foot_kinematics(x, np.zeros(model.nu))
    assert np.allclose(out[0, 3:6], [np.sin(a / 2), 0.0, 0.0], atol=1e-9)


def test_collision_distances(model, oracle):
    h = oracle.collision(model.initial_state)
    assert h.shape == (16,) and np.all(h > 0.0)   # nominal stance is collision free
    # knee-knee distance = lateral knee separation - 2 r_knee
    R, p = oracle.body_placements(model.initial_state[:29])
    kl, kr = model.raw["frames"]["knee"][0]["body"], model.raw["frames"]["knee"][1]["body"]
    assert abs(h[9] - (np.linalg.norm(p[kl] - p[kr]) - 2 * model.raw["collision"]["r_knee"])) < 1e-12


@pytest.mark.parametrize("kind,mu,delta", [(0, 0.2, 5.0), (0, 0.6, 0.03), (1, 1200.0, 0.1), (1, 1500.0, 0.04)])
def test_penalties_are_c1_and_match_fd(oracle, kind, mu, delta):
    for h in (delta * 0.3, delta * 0.999, delta * 1.001, delta * 3.0, -0.5 * delta):
        p, d1, d2 = oracle.penalty(kind, mu, delta, h)
        e = 1e-6 * delta
        pp, d1p, _ = oracle.penalty(kind, mu, delta, h + e)
        pm, d1m, _ = oracle.penalty(kind, mu, delta, h - e)
        assert abs((pp - pm) / (2 * e) - d1) <= 1e-4 * max(1.0, abs(d1))
        if abs(h - delta) > 0.01 * delta:
            assert abs((d1p - d1m) / (2 * e) - d2) <= 1e-4 * max(1.0, abs(d2))
    # relaxed barrier: -mu ln h above delta; PWP barrier: zero above delta, mu at h = 0 (ASSUMPTION A1)
    if kind == 0:
        assert abs(oracle.penalty(0, mu, delta, 2 * delta)[0] + mu * np.log(2 * delta)) < 1e-12
    else:
        assert oracle.penalty(1, mu, delta, 1.5 * delta)[0] == 0.0 and abs(oracle.penalty(1, mu, delta, 0.0)[0] - mu) < 1e-9
