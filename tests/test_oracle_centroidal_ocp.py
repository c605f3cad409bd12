"""The centroidal OCP of the oracle (oracle/centroidal.hpp) checked by independent means: finite differences for
the first-order data, a dense KKT solve of the UNPADDED and UNPROJECTED 35-state QP for the step, the reference's known answers
for the individual terms."""
import numpy as np
import pytest

from test_oracle_lq import dense_qp_solution
from wb_humanoid_mpc_amd import _abi
from wb_humanoid_mpc_amd.reference import (body_placements, centroidal_base_velocity, make_centroidal_problem, torso_reference,
                                           weight_compensating_input)

NX, NU, NZ, CNX = _abi.NX, _abi.NU, _abi.NZ, _abi.CNX
COLS = np.concatenate([np.arange(CNX), NX + np.arange(NU)])   # the 70 live columns of the padded z = [x; u]


def perturbed_centroidal_problem(model, n_nodes, gait, seed=7, arm_swing=True):
    x0, x, u, par, dt = make_centroidal_problem(model, n_nodes=n_nodes, batch=1, gait=gait, perturb=True, seed=seed)
    rng = np.random.default_rng(seed)
    x, u = x[0].copy(), u[0].copy()
    x[:, :CNX] += 0.02 * rng.standard_normal((n_nodes + 1, CNX))
    u[:, :12] += 3.0 * rng.standard_normal((n_nodes, 12))
    u[:, 12:] += 0.5 * rng.standard_normal((n_nodes, model.nj))
    x[:, 12:CNX] = np.clip(x[:, 12:CNX], model.q_lo + 0.02, model.q_hi - 0.02)
    par = par[0].copy()
    if not arm_swing:
        par[:, _abi.P_ARMSWING] = 0.0
    return x0[0], x, u, par, dt


@pytest.mark.parametrize("gait", ["stance", "walk", "run"])
def test_first_order_data_matches_finite_differences(cmodel, coracle, gait):
    n = 12 if gait == "run" else 5
    x0, x, u, par, dt = perturbed_centroidal_problem(cmodel, n, gait, arm_swing=False)
    lq = coracle.cent_lq(dt, x, u, par)
    eps = 1e-6
    modes = set()
    for k in range(0, n, 2 if gait == "run" else 1):
        modes.add(tuple(par[k, _abi.P_CONTACT:_abi.P_CONTACT + 2] > 0.5))
        z = np.concatenate([x[k], u[k]])
        ne = lq["ne"][k]
        gfd, ABfd, CDfd = np.zeros(NZ), np.zeros((CNX, NZ)), np.zeros((ne, NZ))
        for i in COLS:
            zp, zm = z.copy(), z.copy()
            zp[i] += eps
            zm[i] -= eps
            cp, ep = coracle.cent_stage_cost(zp[:NX], zp[NX:], par[k])
            cm, em = coracle.cent_stage_cost(zm[:NX], zm[NX:], par[k])
            gfd[i] = dt * (cp - cm) / (2 * eps)
            CDfd[:, i] = (ep - em) / (2 * eps)
            ABfd[:, i] = (coracle.cent_rk4(zp[:NX], zp[NX:], dt) - coracle.cent_rk4(zm[:NX], zm[NX:], dt)) / (2 * eps)
        c0, e0 = coracle.cent_stage_cost(x[k], u[k], par[k])
        assert abs(dt * c0 - lq["cost"][k]) <= 1e-12 * max(1.0, abs(lq["cost"][k]))
        assert np.allclose(lq["CDe"][k, :ne, NZ], e0, atol=1e-12)
        assert np.abs(lq["g"][k] - gfd).max() <= 2e-5 * max(1.0, np.abs(gfd).max())
        assert np.abs(lq["CDe"][k, :ne, :NZ] - CDfd).max() <= 1e-5 * max(1.0, np.abs(CDfd).max())
        assert np.abs(lq["AB"][k, :CNX] - ABfd).max() <= 1e-6 * max(1.0, np.abs(ABfd).max())
        # padding: decoupled identity states without cost
        assert np.array_equal(lq["AB"][k, CNX:, :], np.eye(NX, NZ)[CNX:]) and not lq["AB"][k, :CNX, CNX:NX].any()
        assert not lq["H"][k][CNX:NX].any() and not lq["g"][k][CNX:NX].any() and not lq["b"][k][CNX:].any()
        assert np.allclose(lq["b"][k][:CNX], coracle.cent_rk4(x[k], u[k], dt) - x[k + 1, :CNX], atol=1e-13)
        assert np.allclose(lq["flow"][k][:CNX], coracle.cent_flow_map(x[k, :CNX], u[k]), atol=1e-13)
        H = lq["H"][k]
        assert np.allclose(H, H.T, atol=1e-9 * np.abs(H).max())
        assert np.linalg.eigvalsh(H[NX:, NX:]).min() > 0.0
    if gait == "run":
        assert (False, False) in modes and len(modes) >= 2


def test_equality_rows_follow_the_contact_mode(cmodel, coracle):
    x0, x, u, par, dt = perturbed_centroidal_problem(cmodel, 2, "stance")
    for flags, ne in (((1, 1), 12), ((1, 0), 13), ((0, 1), 13), ((0, 0), 14)):
        par[0, _abi.P_CONTACT:_abi.P_CONTACT + 2] = flags
        c, e = coracle.cent_stage_cost(x[0], u[0], par[0])
        assert e.size == ne
        zp = par[0, [_abi.P_SWING, _abi.P_SWING + 3]]
        zv = par[0, [_abi.P_SWING + 1, _abi.P_SWING + 4]]
        fc = cmodel.raw["foot_constraint"]
        want = coracle.cent_equalities(x[0, :CNX], u[0], flags, zp, zv, fc["positionErrorGain_z"], fc["orientationErrorGain"])
        assert np.allclose(e, want, atol=1e-14)


def test_torso_error_vanishes_at_its_own_reference(cmodel, coracle):
    """EndEffectorKinematicsQuadraticCost compares the torso link with the SAME kinematics evaluated at (xRef, uRef = 0):
    at x = xRef, u = 0 every error is zero; the host-side generator (reference.py) and the oracle kinematics agree."""
    x0, x, u, par, dt = perturbed_centroidal_problem(cmodel, 2, "walk")
    xr = np.zeros(NX)
    xr[:CNX] = x[1, :CNX]
    p = par[0].copy()
    p[_abi.PC_TORSO:_abi.PC_TORSO + 13] = torso_reference(cmodel, xr[:CNX])
    t = coracle.cent_terms(xr, np.zeros(NU), p)
    assert np.abs(t["torso"]).max() <= 1e-12
    # a pure yaw of the base by a small angle gives an orientation error about z of magnitude ~ angle / 2 (quaternion vector part)
    xy = xr.copy()
    xy[9] += 0.02
    t2 = coracle.cent_terms(xy, np.zeros(NU), p)
    assert abs(abs(t2["torso"][0, 2]) - np.sin(0.01)) <= 2e-4 and np.abs(t2["torso"][0, :2]).max() <= 2e-3


def test_external_torque_is_the_contact_jacobian_transpose(cmodel, coracle):
    """(J_ee^T W)[6 + j] = w_j . (m + (p_c - p_j) x f) for the joints above the foot; checked with the python placements."""
    x0, x, u, par, dt = perturbed_centroidal_problem(cmodel, 2, "stance")
    q = x[0, 6:CNX]
    R, p = body_placements(cmodel, q)
    t = coracle.cent_terms(x[0], u[0], par[0])
    bodies = cmodel.raw["bodies"]
    for f in range(2):
        fr = cmodel.raw["frames"]["contact"][f]
        pc = p[fr["body"]] + R[fr["body"]] @ np.array(fr["p"])
        force, moment = u[0, 6 * f:6 * f + 3], u[0, 6 * f + 3:6 * f + 6]
        for a, j in enumerate(cmodel.raw["ext_torque"]["joints"][f]):
            b = bodies[1 + j]
            w = R[b["parent"]] @ np.array(b["R"]).reshape(3, 3) @ np.array(b["axis"])
            want = w @ (moment + np.cross(pc - p[1 + j], force))
            assert abs(t["tau"][f, a] - want) <= 1e-11 * max(1.0, abs(want))


def test_weight_compensation_is_a_momentum_equilibrium(cmodel, coracle):
    # testDynamicsHelperFunctions.cpp:95-127 at the OCP level: zero momentum rate, and v_b from h = 0, qd = 0 is zero
    x = np.zeros(NX)
    x[:CNX] = cmodel.initial_state
    u = weight_compensating_input(cmodel, (True, True))
    f = coracle.cent_flow_map(x[:CNX], u)
    assert np.abs(f[:3]).max() <= 1e-12 and np.abs(f[6:12]).max() <= 1e-14
    assert np.abs(centroidal_base_velocity(cmodel, x[6:CNX], x[:6])).max() == 0.0


@pytest.mark.parametrize("gait", ["stance", "walk", "run"])
def test_padded_projection_plus_riccati_equals_the_unpadded_dense_kkt_solution(cmodel, coracle, gait):
    N = 5
    x0, x, u, par, dt = perturbed_centroidal_problem(cmodel, N, gait, seed=11)
    lq = coracle.cent_lq(dt, x, u, par)
    res = coracle.cent_sqp_iteration(dt, x0, x, u, par)
    # strip the padding: a 35-state / 35-input QP
    keep = COLS
    small = dict(AB=lq["AB"][:, :CNX][:, :, keep], b=lq["b"][:, :CNX], H=lq["H"][:, keep][:, :, keep], g=lq["g"][:, keep],
                 CDe=lq["CDe"][:, :, np.concatenate([keep, [NZ]])], ne=lq["ne"])
    Hf = np.array(cmodel.raw["Qf"])
    gf = Hf * (x[N, :CNX] - par[N, :CNX])
    import test_oracle_lq as T
    old = (T.NX, T.NZ)
    T.NX, T.NZ = CNX, CNX + NU
    try:
        dx, du = dense_qp_solution(small, dt, (x0 - x[0])[:CNX], Hf, gf)
    finally:
        T.NX, T.NZ = old
    scale = max(1.0, np.abs(dx).max(), np.abs(du).max())
    assert np.abs(res["dx"][:, :CNX] - dx).max() <= 1e-7 * scale
    assert np.abs(res["du"] - du).max() <= 1e-7 * scale
    assert not res["dx"][:, CNX:].any()
    assert res["kkt"][0] <= 1e-9 * max(1.0, np.abs(lq["g"]).max()) and res["kkt"][1] <= 1e-10


def test_sqp_converges_on_stance(cmodel, coracle):
    x0, x, u, par, dt = make_centroidal_problem(cmodel, n_nodes=12, batch=1, gait="stance", v_cmd=(0, 0, 0.7925, 0))
    x0, x, u, par = x0[0], x[0], u[0], par[0]
    steps = []
    for _ in range(5):
        r = coracle.cent_sqp_iteration(dt, x0, x, u, par)
        x, u = r["x"], r["u"]
        steps.append(np.abs(r["dx"]).max())
    assert steps[-1] < 0.05 * steps[0]
    assert r["perf_after"]["dynamics_sse"] < 1e-10 and r["perf_after"]["equality_sse"] < 1e-6
