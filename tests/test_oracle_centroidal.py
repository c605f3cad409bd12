"""Oracle groundwork for the centroidal formulation: flow map of x = [h/m, q_b, q_j], u = [W_l, W_r, qd_j].

No HIP path exists for this formulation yet; these tests pin the restatement (oracle/centroidal.hpp, ASSUMPTION A7) against
the two usable known answers of the reference and against independent identities of the whole-body oracle.
"""
import numpy as np

from conftest import random_state_input
from test_oracle_dynamics import fd_jac


def euler_rate_axes(e):
    ez, ey = e[0], e[1]
    cz, sz, cy, sy = np.cos(ez), np.sin(ez), np.cos(ey), np.sin(ey)
    return np.array([[0, -sz, cz * cy], [0, cz, sz * cy], [1, 0, -sy]])   # columns: world axes of the Z, Y, X rates


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def cent_state_input(model, rng):
    xw, uw = random_state_input(model, rng)
    nj = model.nj
    q = xw[:6 + nj]
    x = np.concatenate([0.3 * rng.standard_normal(6), q])
    u = np.concatenate([uw[:12], 0.5 * rng.standard_normal(nj)])
    return x, u


def test_layout_matches_the_reference_robot_model(model, oracle):
    # humanoid_centroidal_mpc/common/CentroidalMpcRobotModel.h:73 (dims), :89-95 (start indices);
    # humanoid_centroidal_mpc_test/src/testCentroidalMpcRobotModel.cpp:64-87 checks the same facts
    assert oracle.CENT_NX == 12 + model.nj == 35 and oracle.CENT_NU == 6 * 2 + model.nj == 35
    assert (oracle.CENT_BASE_START, oracle.CENT_JOINT_START, oracle.CENT_JOINT_VEL_START) == (6, 12, 12)
    x = np.concatenate([np.zeros(6), model.initial_state[:29]])   # [h/m, q_b, q_j]
    u = 100.0 + np.arange(35.0)
    xd = oracle.cent_flow_map(x, u)
    assert np.array_equal(xd[12:], u[12:])            # joint angles integrate the joint-velocity inputs


def test_weight_compensation_gives_zero_normalized_momentum_rate(model, oracle, rng):
    # humanoid_centroidal_mpc/test/testDynamicsHelperFunctions.cpp:95-127 (last expectation); the force part holds for
    # any stance set, the moment part for the double-support split only when the centre of mass is midway (not asserted)
    x, _ = cent_state_input(model, rng)
    for flags in ((1, 1), (1, 0), (0, 1)):
        u = np.zeros(35)
        for f in range(2):
            if flags[f]:
                u[6 * f + 2] = model.total_mass * 9.81 / sum(flags)
        assert np.allclose(oracle.cent_momentum_rate(x[6:], u)[:3], 0.0, atol=1e-12)
    # nominal symmetric stance: the whole 6-vector vanishes up to the lateral asymmetry of the model's mass distribution
    q0 = model.initial_state[:29]
    u = np.zeros(35)
    u[2] = u[8] = model.total_mass * 9.81 / 2
    rate = oracle.cent_momentum_rate(q0, u)
    out, _ = oracle.foot_kinematics(model.initial_state, np.zeros(model.nu))
    _, com = oracle.cent_momentum_matrix(q0)
    mid = 0.5 * (out[0, :3] + out[1, :3])
    expect = np.cross(mid - com, [0, 0, 9.81])          # (sum_c (p_c - com) x f_c) / m
    assert np.allclose(rate[:3], 0.0, atol=1e-12) and np.allclose(rate[3:], expect, atol=1e-10)


def test_momentum_matrix_equals_the_base_rows_of_the_mass_matrix(model, oracle, rng):
    # d(kinetic energy)/d(pdot) = linear momentum, d/d(euler rates) = E^T (angular momentum about the base origin):
    # A_lin = M[0:3], A_ang = E^-T M[3:6] - (com - p_b) x A_lin — M from the independent projected Newton-Euler restatement
    for _ in range(3):
        xw, _u = random_state_input(model, rng)
        q = xw[:29]
        M, _nle = oracle.full_dynamics(xw)
        A, com = oracle.cent_momentum_matrix(q)
        E = euler_rate_axes(q[3:6])
        assert np.allclose(A[:3], M[:3], atol=1e-10)
        assert np.allclose(A[3:], np.linalg.solve(E.T, M[3:6]) - skew(com - q[:3]) @ A[:3], atol=1e-9)
        # centre of mass from the body placements
        R, p = oracle.body_placements(q)
        masses = np.array([b.mass for b in model.desc.bodies])
        coms = np.array([p[i] + R[i] @ np.array(model.desc.bodies[i].com) for i in range(len(masses))])
        assert np.allclose(com, (masses[:, None] * coms).sum(0) / masses.sum(), atol=1e-12)
        # the block structure computeFloatingBaseCentroidalMomentumMatrixInverse relies on
        assert np.allclose(A[:3, :3], model.total_mass * np.eye(3), atol=1e-10) and np.allclose(A[3:, :3], 0.0, atol=1e-10)


def test_linear_momentum_is_mass_times_com_velocity(model, oracle, rng):
    xw, _u = random_state_input(model, rng)
    q, v = xw[:29], xw[29:]
    A, _ = oracle.cent_momentum_matrix(q)
    eps = 1e-6
    cp = oracle.cent_momentum_matrix(q + eps * v)[1]
    cm = oracle.cent_momentum_matrix(q - eps * v)[1]
    assert np.allclose(A[:3] @ v, model.total_mass * (cp - cm) / (2 * eps), atol=1e-6)


def test_flow_map_reproduces_the_momentum(model, oracle, rng):
    x, u = cent_state_input(model, rng)
    xd = oracle.cent_flow_map(x, u)
    A, _ = oracle.cent_momentum_matrix(x[6:])
    assert np.allclose(A @ xd[6:], model.total_mass * x[:6], atol=1e-9)     # A(q) qdot = m h
    assert np.allclose(xd[:6], oracle.cent_momentum_rate(x[6:], u), atol=0)
    # free fall: no wrenches -> v_com accelerates with g, angular momentum is conserved
    u0 = u.copy()
    u0[:12] = 0.0
    assert np.allclose(oracle.cent_flow_map(x, u0)[:6], [0, 0, -9.81, 0, 0, 0], atol=1e-12)


def test_flow_map_jacobian_matches_central_differences(model, oracle, rng):
    x, u = cent_state_input(model, rng)
    f, J = oracle.cent_flow_map_jac(x, u)
    assert np.allclose(f, oracle.cent_flow_map(x, u), atol=1e-13)
    Jfd = fd_jac(lambda z: oracle.cent_flow_map(z[:35], z[35:]), np.concatenate([x, u]))
    assert np.abs(J - Jfd).max() <= 1e-6 * max(1.0, np.abs(J).max())
    # structure: joint rows = identity on qd_j; the momentum rate does not depend on h or qd_j; nothing depends on the
    # absolute base position except through (p_c - com), which is translation invariant
    assert np.allclose(J[12:, 35 + 12:], np.eye(23)) and np.allclose(J[12:, :35 + 12], 0.0)
    assert np.allclose(J[:6, :6], 0.0) and np.allclose(J[:6, 35 + 12:], 0.0)
    assert np.allclose(J[:, 6:9], 0.0, atol=1e-12)


# ---- velocity-level foot kinematics and the equality constraints
G1C_GAIN_POS_Z, G1C_GAIN_ORI = 5.0, 20.0    # robot_models/unitree_g1/g1_centroidal_mpc/config/mpc/task.info:16-17


def test_foot_twist_is_the_time_derivative_of_the_foot_placement_along_the_flow(model, oracle, rng):
    x, u = cent_state_input(model, rng)
    xd = oracle.cent_flow_map(x, u)
    fk = oracle.cent_foot_kinematics(x, u)
    eps = 1e-6
    xp, xm = x.copy(), x.copy()
    xp[6:] += eps * xd[6:]
    xm[6:] -= eps * xd[6:]
    fp, fm = oracle.cent_foot_kinematics(xp, u), oracle.cent_foot_kinematics(xm, u)
    for f in range(2):
        assert np.allclose((fp[f, :3] - fm[f, :3]) / (2 * eps), fk[f, 6:9], atol=1e-6)
    # positions and orientation errors agree with the whole-body restatement at the same configuration
    xw = np.concatenate([x[6:], np.zeros(29)])
    out, _R = oracle.foot_kinematics(xw, np.zeros(model.nu))
    assert np.allclose(out[:, :6], fk[:, :6], atol=1e-13)
    # and the twist with the whole-body one at the generalized velocity the centroidal mapping yields
    xw[29:] = xd[6:]
    out, _R = oracle.foot_kinematics(xw, np.zeros(model.nu))
    assert np.allclose(out[:, 6:12], fk[:, 6:12], atol=1e-12)


def test_equality_rows_follow_the_contact_pattern(model, oracle, rng):
    x, u = cent_state_input(model, rng)
    fk = oracle.cent_foot_kinematics(x, u)
    zpos, zvel = (0.01, 0.05), (0.2, -0.3)
    for contact, ne in (((1, 1), 12), ((1, 0), 13), ((0, 1), 13), ((0, 0), 14)):
        eq = oracle.cent_equalities(x, u, contact, zpos, zvel, G1C_GAIN_POS_Z, G1C_GAIN_ORI)
        assert eq.size == ne
        row = 0
        for f in range(2):
            if contact[f]:
                want = np.concatenate([fk[f, 6:9], fk[f, 9:12] + G1C_GAIN_ORI * fk[f, 3:6]])
                want[2] += G1C_GAIN_POS_Z * (fk[f, 2] - zpos[f])
                assert np.allclose(eq[row:row + 6], want, atol=1e-13)
                row += 6
            else:
                assert np.array_equal(eq[row:row + 6], u[6 * f:6 * f + 6])
                assert np.isclose(eq[row + 6], fk[f, 8] - zvel[f] + G1C_GAIN_POS_Z * (fk[f, 2] - zpos[f]), atol=1e-13)
                row += 7
        assert row == ne


def test_stance_constraint_vanishes_at_rest_on_the_ground(model, oracle):
    # nominal stance, zero momentum, zero joint velocity, reference height = the foot's own height: nothing to correct
    x = np.concatenate([np.zeros(6), model.initial_state[:29]])
    u = np.zeros(35)
    fk = oracle.cent_foot_kinematics(x, u)
    eq = oracle.cent_equalities(x, u, (1, 1), fk[:, 2], (0.0, 0.0), G1C_GAIN_POS_Z, G1C_GAIN_ORI)
    assert np.allclose(fk[:, 6:], 0.0, atol=1e-13)
    assert np.allclose(eq.reshape(2, 6)[:, :3], 0.0, atol=1e-13)
    assert np.allclose(eq.reshape(2, 6)[:, 3:], G1C_GAIN_ORI * fk[:, 3:6], atol=1e-13)   # feet of the nominal pose are flat
    assert np.abs(fk[:, 3:6]).max() < 1e-6


def test_equality_jacobian_matches_central_differences(model, oracle, rng):
    x, u = cent_state_input(model, rng)
    args = ((1, 0), (0.0, 0.04), (0.0, 0.3), G1C_GAIN_POS_Z, G1C_GAIN_ORI)
    eq, J = oracle.cent_equalities(x, u, *args, jac=True)
    assert np.allclose(eq, oracle.cent_equalities(x, u, *args), atol=1e-13)
    Jfd = fd_jac(lambda z: oracle.cent_equalities(z[:35], z[35:], *args), np.concatenate([x, u]))
    assert np.abs(J - Jfd).max() <= 1e-6 * max(1.0, np.abs(J).max())
    # the swing foot's zero-wrench rows are unit rows of D (what the projection kernel deflates in the whole-body problem)
    assert np.array_equal(J[6:12, 35 + 6:35 + 12], np.eye(6)) and np.count_nonzero(J[6:12]) == 6
    # D = d eq / du has full row rank
    assert np.linalg.matrix_rank(J[:, 35:]) == eq.size
