"""The ASSEMBLY rows of the hot path pinned against REFERENCE-COMPILED code: the reference's own
EndEffectorDynamicsAccelerationsConstraint.cpp (behind ZeroAccelerationConstraintCppAd.cpp), EndEffectorDynamicsLinearAccConstraint.cpp,
StateInputQuadraticCost.cpp, JointLimitsSoftConstraint.cpp and WeightCompInitializer.cpp, compiled in place from /root/reference
(oracle/Makefile `ref`, oracle/ref_terms_driver.cpp) and run on kinematics handed in through the EndEffectorDynamics interface — the
reference's only implementation of it is CppAD-generated.  The committed fixture tests/golden/ref_assembly.npz
(tests/golden/make_ref_assembly_golden.py) travels to the GPU box.

What this pins: HOW position / orientation error / twist / acceleration and their Jacobians, the task file's gains and the planner's z
references are combined into the rows [C | D | e] (a9, a10: which Jacobian blocks enter, the sign the references enter with — this is
where assumption A2's sign meets the velocity / acceleration feedback), what the quadratic cost is taken of (a5), the structure of the
joint-limit soft constraint around the assumed penalty (a14) and the initializer (a18).  What it does NOT pin: the kinematics themselves
(oracle-defined).  The CPU tests check the oracle, the `gpu` tests check the HIP kernels THROUGH THE C ABI against the same
reference-compiled numbers."""
import copy
import os

import numpy as np
import pytest

from wb_humanoid_mpc_amd import _abi
from wb_humanoid_mpc_amd.reference import ModeSchedule, TargetTrajectories, cold_start, mode_to_contact_flags, phase_variable

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "ref_assembly.npz"))
GT = np.load(os.path.join(HERE, "golden", "ref_terms.npz"))
NX, NU, NZ, NJ = _abi.NX, _abi.NU, _abi.NZ, _abi.NJ
GAIN_KEYS = ("gain_pos_z", "gain_ori", "gain_linvel_z", "gain_linvel_xy", "gain_angvel", "gain_linacc_z", "gain_linacc_xy", "gain_angacc")


def reference_rows(k):
    """[C | D | e] of fixture node k as the reference's constraints return them, in the order the terms are added per foot
    (WBMpcInterface.cpp:175-177: zero wrench, stance-foot acceleration, swing-foot vertical) — the zero-wrench rows from the formula its
    own fixture pins (tests/test_ref_terms.py)."""
    flags = G["node.par"][k, _abi.P_CONTACT:_abi.P_CONTACT + 2] > 0.5
    rows = []
    for f in range(2):
        if not flags[f]:
            for r in range(6):
                row = np.zeros(NZ + 1)
                row[NX + 6 * f + r] = 1.0
                row[NZ] = G["node.u"][k, 6 * f + r]
                rows.append(row)
        else:
            assert G["stance.active"][k, f]
            for r in range(6):
                rows.append(np.concatenate([G["stance.dfdx"][k, f, r], G["stance.dfdu"][k, f, r], [G["stance.f"][k, f, r]]]))
        if not flags[f]:
            assert not G["stance.active"][k, f]
            rows.append(np.concatenate([G["swing.dfdx"][k, f, 0], G["swing.dfdu"][k, f, 0], [G["swing.f"][k, f, 0]]]))
    return np.array(rows)


def model_with(model, **changes):
    m2 = copy.copy(model)
    m2.desc = type(model.desc).from_buffer_copy(model.desc)
    for key, v in changes.items():
        obj, parts = m2.desc, key.split(".")
        for p in parts[:-1]:
            obj = getattr(obj, p)
        if isinstance(v, (list, tuple, np.ndarray)):
            getattr(obj, parts[-1])[:] = list(v)
        else:
            setattr(obj, parts[-1], v)
    return m2


def test_fixture_is_built_on_the_task_files_gains_and_the_oracles_kinematics(model, oracle):
    assert np.array_equal(G["gains"], [getattr(model.desc, k) for k in GAIN_KEYS])
    modes = set()
    for k in range(len(G["node.x"])):
        kin, _, jac = oracle.foot_kinematics(G["node.x"][k], G["node.u"][k], jac=True)
        np.testing.assert_allclose(kin, G["node.kin"][k], rtol=0, atol=1e-12 * max(1.0, np.abs(kin).max()))
        np.testing.assert_allclose(jac, G["node.jac"][k], rtol=0, atol=1e-11 * max(1.0, np.abs(jac).max()))
        modes.add(tuple(G["node.par"][k, _abi.P_CONTACT:_abi.P_CONTACT + 2] > 0.5))
        # the reference's getValue and the value part of its linear approximation are the same numbers
        assert np.array_equal(G["stance.value"][k], G["stance.f"][k]) and np.array_equal(G["swing.value"][k], G["swing.f"][k])
    assert modes == {(True, True), (True, False), (False, True)}


def test_equality_rows_of_the_oracle_equal_the_reference_compiled_assembly(oracle):
    """a9 / a10: the oracle's [C | D | e] of every fixture node (double support, left swing, right swing) against the rows the reference's
    constraint classes form from the same kinematics."""
    dt = float(G["node.dt"])
    worst = 0.0
    for k in range(len(G["node.x"])):
        x, u, par = G["node.x"][k], G["node.u"][k], G["node.par"][k]
        lq = oracle.lq(dt, np.stack([x, x]), u[None], np.stack([par, par]))
        want = reference_rows(k)
        assert lq["ne"][0] == len(want)
        got = lq["CDe"][0][:len(want)]
        sc = max(1.0, np.abs(want).max())
        worst = max(worst, np.abs(got - want).max() / sc)
        assert np.abs(got - want).max() <= 1e-12 * sc, k
        assert not lq["CDe"][0][len(want):].any()
    print(f"oracle vs reference-compiled stance / swing rows: worst {worst:.1e} of the rows' scale")


def _cost_params(t):
    sched = ModeSchedule(G["cost.event_times"].tolist(), G["cost.mode_sequence"].tolist())
    targets = TargetTrajectories(G["cost.tt"], G["cost.ts"])
    par = np.zeros(_abi.NODE_PARAMS)
    par[_abi.P_XDES:_abi.P_XDES + NX] = targets.desired_state(t)
    par[_abi.P_ARMSWING] = np.sin(2.0 * np.pi * (phase_variable(sched, t) - 0.15))
    par[_abi.P_CONTACT:_abi.P_CONTACT + 2] = mode_to_contact_flags(sched.mode_at(t))
    return par


def test_quadratic_cost_deviation_equals_the_reference_compiled_cost(model, oracle):
    """a5: StateInputQuadraticCost::getStateInputDeviation (state - x_nom(t) with the arm-swing reference on the current yaw, input - weight
    compensation on the contact flags at t) against host mirror -> node parameters -> the oracle's nominal state / input."""
    assert abs(float(G["cost.total_mass"]) - oracle.total_mass()) <= 1e-12
    Q, R = np.array(model.desc.Q), np.array(model.desc.R)
    seen = set()
    for t, x, u, dx, du, val in zip(G["cost.times"], G["cost.x"], G["cost.u"], G["cost.dx"], G["cost.du"], G["cost.value"]):
        par = _cost_params(t)
        xn, un = oracle.nominal(x, par)
        np.testing.assert_allclose(x - xn, dx, rtol=0, atol=1e-13)
        np.testing.assert_allclose(u - un, du, rtol=0, atol=1e-11)
        assert abs(0.5 * (dx * Q) @ dx + 0.5 * (du * R) @ du - val) <= 1e-12 * max(1.0, abs(val))
        seen.add(tuple(par[_abi.P_CONTACT:_abi.P_CONTACT + 2]))
    assert len(seen) >= 2


def test_initializer_equals_the_reference_compiled_one(model):
    """a18: WeightCompInitializer::compute (input = weight compensation on the contact flags at `time`, next state = state) against the
    host-side cold start."""
    for t, x, u, xn in zip(G["cost.times"], G["cost.x"], G["init.u"], G["init.next_state"]):
        par = np.stack([_cost_params(t), _cost_params(t)])
        xs, us = cold_start(model, x, par)
        np.testing.assert_allclose(us[0], u, rtol=1e-14, atol=0)
        assert np.array_equal(xs[1], xn) and np.array_equal(xn, x)


def test_joint_limit_terms_of_the_oracle_equal_the_reference_compiled_constraint(model, oracle):
    """a14: JointLimitsSoftConstraint.cpp:64-100 around the stand-in penalty: which offsets are penalised, the gradient's sign, the diagonal
    Hessian, no offset.  The oracle's share of it = its LQ model with the barrier minus the one without."""
    from hsqp_oracle import Oracle
    mu, delta = G["jl.mu_delta"]
    assert (mu, delta) == (model.desc.joint_limit_barrier.mu, model.desc.joint_limit_barrier.delta)
    off = Oracle(model_with(model, **{"joint_limit_barrier.mu": 0.0}))
    dt = 0.035
    u = np.zeros(NU)
    par = np.zeros((2, _abi.NODE_PARAMS))
    par[:, _abi.P_CONTACT:_abi.P_CONTACT + 2] = 1.0
    active = 0
    for x, f, g, h in zip(G["jl.x"], G["jl.f"], G["jl.dfdx"], G["jl.dfdxx_diag"]):
        a, b = oracle.lq(dt, np.stack([x, x]), u[None], par), off.lq(dt, np.stack([x, x]), u[None], par)
        dH, dg, dc = (a["H"][0] - b["H"][0]) / dt, (a["g"][0] - b["g"][0]) / dt, (a["cost"][0] - b["cost"][0]) / dt
        sc = max(1.0, np.abs(h).max())
        assert np.abs(np.diag(dH)[:NX] - h).max() <= 1e-9 * sc and np.abs(dH - np.diag(np.diag(dH))).max() <= 1e-9 * sc
        assert np.abs(dg[:NX] - g).max() <= 1e-9 * max(1.0, np.abs(g).max()) and not dg[NX:].any()
        assert abs(dc - f) <= 1e-9 * max(1.0, abs(f))
        active += int(f > 0)
    assert active >= 4


# ---- the same reference-compiled numbers against the HIP kernels, through the C ABI ----------------------------------------------------------

@pytest.fixture(scope="module")
def gpu(model):
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    s = HipSqpSolver(model, max_nodes=16, max_batch=1)
    yield s
    s.close()


def _lq_blocks(solver, x, u, par, dt):
    """One node through hsqp_upload + one iteration; its LQ blocks from the device (hsqp_debug_read)."""
    n = len(u)
    solver.upload(x[0], x, u, par, dt)
    solver.iterate(1)
    return {k: solver.debug_read(b)[0] for k, b in (("H", _abi.BLK_H), ("g", _abi.BLK_G), ("CDe", _abi.BLK_CDE), ("ne", _abi.BLK_NE), ("cost", _abi.BLK_COST))}, n


@pytest.mark.gpu
def test_gpu_equality_rows_equal_the_reference_compiled_assembly(gpu):
    """BLK_CDE of the device (zero-wrench, stance-foot and swing-foot rows) against the reference-compiled rows of the fixture: all nine
    nodes as ONE horizon (the rows of a node depend on that node's (x, u, parameters) only)."""
    dt = float(G["node.dt"])
    n = len(G["node.x"])
    x = np.concatenate([G["node.x"], G["node.x"][-1:]])
    par = np.concatenate([G["node.par"], G["node.par"][-1:]])
    blk, _ = _lq_blocks(gpu, x, G["node.u"], par, dt)
    worst = 0.0
    for k in range(n):
        want = reference_rows(k)
        assert blk["ne"][k] == len(want)
        got = blk["CDe"][k][:len(want)]
        sc = max(1.0, np.abs(want).max())
        worst = max(worst, np.abs(got - want).max() / sc)
        assert np.abs(got - want).max() <= 1e-10 * sc, k
    print(f"device vs reference-compiled equality rows: worst {worst:.1e} of the rows' scale")


@pytest.mark.gpu
def test_gpu_nominal_state_and_input_equal_the_reference_compiled_cost(model):
    """x_nom and u_nom of the device: the gradient of the stage cost is linear in the weights, so g(2Q, 2R) - g(Q, R) = dt [Q (x - x_nom);
    R (u - u_nom)] isolates the quadratic cost's deviation — compared with StateInputQuadraticCost's (reference-compiled)."""
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    Q, R = np.array(model.desc.Q), np.array(model.desc.R)
    dt = 0.035
    n = len(G["cost.times"])
    x = np.concatenate([G["cost.x"], G["cost.x"][-1:]])
    par = np.stack([_cost_params(t) for t in G["cost.times"]] + [_cost_params(G["cost.times"][-1])])
    s = HipSqpSolver(model, max_nodes=n, max_batch=1)
    try:
        g1 = _lq_blocks(s, x, G["cost.u"], par, dt)[0]["g"]
        s.update_weights(2.0 * Q, 2.0 * R, None)
        g2 = _lq_blocks(s, x, G["cost.u"], par, dt)[0]["g"]
    finally:
        s.close()
    d = (g2 - g1) / dt
    for k in range(n):
        np.testing.assert_allclose(d[k, :NX], Q * G["cost.dx"][k], rtol=0, atol=1e-9 * max(1.0, np.abs(Q * G["cost.dx"][k]).max()))
        np.testing.assert_allclose(d[k, NX:], R * G["cost.du"][k], rtol=0, atol=1e-9 * max(1.0, np.abs(R * G["cost.du"][k]).max()))


@pytest.mark.gpu
def test_gpu_joint_limit_terms_equal_the_reference_compiled_constraint(model):
    """The device's joint-limit share of (H, g, cost) = handle with the barrier minus handle without, against JointLimitsSoftConstraint
    (reference-compiled around the stand-in penalty)."""
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    dt = 0.035
    n = len(G["jl.x"])
    x = np.concatenate([G["jl.x"], G["jl.x"][-1:]])
    u = np.zeros((n, NU))
    par = np.zeros((n + 1, _abi.NODE_PARAMS))
    par[:, _abi.P_CONTACT:_abi.P_CONTACT + 2] = 1.0
    blks = []
    for m in (model, model_with(model, **{"joint_limit_barrier.mu": 0.0})):
        s = HipSqpSolver(m, max_nodes=n, max_batch=1)
        try:
            blks.append(_lq_blocks(s, x, u, par, dt)[0])
        finally:
            s.close()
    a, b = blks
    for k in range(n):
        dH, dg, dc = (a["H"][k] - b["H"][k]) / dt, (a["g"][k] - b["g"][k]) / dt, (a["cost"][k] - b["cost"][k]) / dt
        h, g, f = G["jl.dfdxx_diag"][k], G["jl.dfdx"][k], G["jl.f"][k]
        sc = max(1.0, np.abs(h).max())
        assert np.abs(np.diag(dH)[:NX] - h).max() <= 1e-8 * sc and np.abs(dH - np.diag(np.diag(dH))).max() <= 1e-8 * sc
        assert np.abs(dg[:NX] - g).max() <= 1e-8 * max(1.0, np.abs(g).max())
        assert abs(dc - f) <= 1e-8 * max(1.0, abs(f))


@pytest.mark.gpu
def test_gpu_friction_cone_block_equals_the_reference_compiled_constraint(model):
    """The friction cone's share of the device's (H, g) = twice (handle with the relaxed barrier minus handle with half its weight), against
    FrictionForceConeConstraint's value / gradient / Hessian (reference-compiled, tests/golden/ref_terms.npz) pushed through ocs2's soft-
    constraint wrapper as SURVEY.md A.2 restates it: p'' grad h grad h' + p' Hess h, p' grad h, with the relaxed barrier (mu, delta)."""
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    sched = ModeSchedule(GT["con.event_times"].tolist(), GT["con.mode_sequence"].tolist())
    times, U, xr = GT["con.times"], GT["con.u"], GT["acc.x"]
    n = len(times)
    dt = 0.035
    x = np.tile(xr, (n + 1, 1))
    par = np.zeros((n + 1, _abi.NODE_PARAMS))
    for k, t in enumerate(list(times) + [times[-1]]):
        par[k, _abi.P_CONTACT:_abi.P_CONTACT + 2] = mode_to_contact_flags(sched.mode_at(t))
    mu, delta = model.desc.friction_barrier.mu, model.desc.friction_barrier.delta
    blks = []
    # (half the barrier weight, not zero: the kernels carry a penalty row as sqrt(p'') grad h with rho = p' / sqrt(p''))
    for m in (model, model_with(model, **{"friction_barrier.mu": 0.5 * mu})):
        s = HipSqpSolver(m, max_nodes=n, max_batch=1)
        try:
            blks.append(_lq_blocks(s, x, U, par, dt)[0])
        finally:
            s.close()
    a, b = blks
    checked = 0
    for k in range(n):
        dH, dg = 2.0 * (a["H"][k] - b["H"][k]) / dt, 2.0 * (a["g"][k] - b["g"][k]) / dt      # the share is linear in mu
        wantH, wantg = np.zeros((NZ, NZ)), np.zeros(NZ)
        for c in range(2):
            if not GT["fric.active"][k, c]:
                continue
            h, dh, d2 = GT["fric.f"][k, c], GT["fric.dfdu"][k, c], GT["fric.dfduu"][k, c]
            if h > delta:
                p1, p2 = -mu / h, mu / (h * h)
            else:
                p1, p2 = mu * (h - 2.0 * delta) / (delta * delta), mu / (delta * delta)
            wantH[NX:, NX:] += p2 * np.outer(dh, dh) + p1 * d2
            wantH[np.arange(NX), np.arange(NX)] += p1 * GT["fric.dfdxx_diag"][k, c]
            wantg[NX:] += p1 * dh
            checked += 1
        sc = max(1.0, np.abs(a["H"][k]).max() / dt)
        assert np.abs(dH - wantH).max() <= 1e-9 * sc, k
        assert np.abs(dg - wantg).max() <= 1e-9 * max(1.0, np.abs(a["g"][k]).max() / dt), k
    assert checked >= 6


def test_fixture_is_what_the_reference_compiled_library_returns(model, oracle):
    """Where /root/reference is mounted (the build container): the library itself, freshly compiled, reproduces the committed fixture."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    import ref_terms
    if not ref_terms.available():
        pytest.skip("oracle/_ref/libref_terms.so is missing and /root/reference is not mounted")
    ref = ref_terms.RefTerms(model.nj)
    from wb_humanoid_mpc_amd.reference import STANCE
    k = 0
    flags = tuple(G["node.par"][k, _abi.P_CONTACT:_abi.P_CONTACT + 2] > 0.5)
    mode = {(True, True): 3, (True, False): 2, (False, True): 1}[flags]
    for f in range(2):
        r = ref.stance_foot_constraint(G["gains"], f, [1.0], [mode, STANCE], 0.5, G["node.kin"][k, f], G["node.jac"][k, f])
        assert np.array_equal(r["f"], G["stance.f"][k, f]) and np.array_equal(r["dfdx"], G["stance.dfdx"][k, f]) and np.array_equal(r["dfdu"], G["stance.dfdu"][k, f])
        r = ref.swing_foot_constraint(G["gains"], G["node.par"][k, _abi.P_SWING + 3 * f:_abi.P_SWING + 3 * f + 3], G["node.kin"][k, f], G["node.jac"][k, f])
        assert np.array_equal(r["f"], G["swing.f"][k, f]) and np.array_equal(r["dfdx"], G["swing.dfdx"][k, f])
    f0, g0, h0 = ref.joint_limits(model.q_lo, model.q_hi, *G["jl.mu_delta"], G["jl.x"][2])
    assert f0 == G["jl.f"][2] and np.array_equal(g0, G["jl.dfdx"][2]) and np.array_equal(h0, G["jl.dfdxx_diag"][2])
