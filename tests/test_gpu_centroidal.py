"""GPU parity tests of the centroidal formulation (BASELINE configs 1-2): the HIP path through the C ABI against
the oracle and the committed golden fixtures.  Same tolerances as the whole-body tests (f64 end to end)."""
import os

import numpy as np
import pytest

from test_oracle_centroidal_ocp import perturbed_centroidal_problem
from wb_humanoid_mpc_amd import _abi
from wb_humanoid_mpc_amd.reference import make_centroidal_problem

from tolerances import TRAJ_ABS, assert_kkt, assert_perf, assert_perf_arrays, assert_step, assert_linear_residual

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NX, NU, NZ, CNX = _abi.NX, _abi.NU, _abi.NZ, _abi.CNX


@pytest.fixture(scope="module")
def cgpu(cmodel):
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    s = HipSqpSolver(cmodel, max_nodes=100, max_batch=8)
    yield s
    s.close()


def rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


@pytest.mark.parametrize("name", ["cent_stance_n4", "cent_walk_n8", "cent_run_n14"])
def test_against_golden_fixtures(cgpu, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = cgpu.run(g["x_init"], g["x"], g["u"], g["par"], float(g["dt"]))
    assert_step(out, dict(dx=g["dx"], du=g["du"]), 0, name)
    assert not out["dx"][0][:, CNX:].any() and not out["x"][0][:, CNX:].any()
    for got, want in ((out["perf_before"][0], g["perf_before"]), (out["perf_after"][0], g["perf_after"])):
        assert_perf_arrays([got["cost"], got["dynamics_sse"], got["equality_sse"]], want, name)
    assert_kkt(out["kkt"][0], np.abs(g["g"]).max(), name)
    assert rel(cgpu.debug_read(_abi.BLK_BVEC)[0], g["b"]) <= 1e-12
    assert rel(cgpu.debug_read(_abi.BLK_G)[0], g["g"]) <= 1e-11
    assert rel(cgpu.debug_read(_abi.BLK_COST)[0][:-1], g["cost"]) <= 1e-11
    assert rel(cgpu.debug_read(_abi.BLK_FLOW)[0], g["flow"]) <= 1e-12
    assert np.array_equal(cgpu.debug_read(_abi.BLK_NE)[0], g["ne"])
    assert rel(cgpu.debug_read(_abi.BLK_CDE)[0][:, :, -1], g["e"]) <= 1e-11
    assert rel(cgpu.debug_read(_abi.BLK_AB)[0][:, 7, :], g["AB_row7"]) <= 1e-11
    assert rel(np.einsum("kii->ki", cgpu.debug_read(_abi.BLK_H)[0]), g["H_diag"]) <= 1e-11


@pytest.mark.parametrize("gait,n", [("stance", 5), ("walk", 10), ("run", 14)])
def test_lq_blocks_against_oracle(cgpu, cmodel, coracle, gait, n):
    x0, x, u, par, dt = perturbed_centroidal_problem(cmodel, n, gait, seed=41)
    if gait == "run":   # make sure a flight node is covered
        par = par.copy()
        par[3:5, _abi.P_CONTACT:_abi.P_CONTACT + 2] = 0.0
    cgpu.run(x0, x, u, par, dt)
    lq = coracle.cent_lq(dt, x, u, par, threads=4)
    for blk, key in ((_abi.BLK_AB, "AB"), (_abi.BLK_BVEC, "b"), (_abi.BLK_H, "H"), (_abi.BLK_G, "g"), (_abi.BLK_CDE, "CDe"), (_abi.BLK_FLOW, "flow")):
        assert rel(cgpu.debug_read(blk)[0], lq[key]) <= 1e-11, key
    assert rel(cgpu.debug_read(_abi.BLK_COST)[0][:-1], lq["cost"][:-1]) <= 1e-11
    assert np.array_equal(cgpu.debug_read(_abi.BLK_NE)[0], lq["ne"])


def test_batch_of_perturbed_instances_against_oracle(cgpu, cmodel, coracle):
    x0, x, u, par, dt = make_centroidal_problem(cmodel, n_nodes=16, batch=6, perturb=True)
    out = cgpu.run(x0, x, u, par, dt)
    for b in range(6):
        r = coracle.cent_sqp_iteration(dt, x0[b], x[b], u[b], par[b], threads=4)
        # DECLARED RELAXATION (not a BASELINE configuration: configs 1-2 are single unperturbed instances).  Perturbing the centroidal
        # state by the whole-body sigmas throws some instances far outside the barriers (instance 1: Armijo metric -2.9e4, |du| 191);
        # the oracle's own KKT residual on that QP is 3.7e-9 and the two solutions differ by 1.2e-6 = 6.5e-9 of the step.
        assert_step(out, r, b, rel=1e-8)
        assert_perf(out["perf_before"][b], r["perf_before"], f"instance {b} before")
        assert_perf(out["perf_after"][b], r["perf_after"], f"instance {b} after", rel=1e-8)   # same instance: cost 3.4e4 after the step, 8.6e-10
    solo = cgpu.run(x0[3], x[3], u[3], par[3], dt)
    assert np.array_equal(solo["dx"][0], out["dx"][3]) and np.array_equal(solo["du"][0], out["du"][3])


@pytest.mark.parametrize("cfg,n,gait,v_cmd", [(1, 20, "stance", (0.0, 0.0, 0.7925, 0.0)), (2, 100, "walk", (0.3, 0.0, 0.7925, 0.0))])
def test_baseline_configs_1_and_2_against_the_oracle(cgpu, cmodel, coracle, cfg, n, gait, v_cmd):
    """BASELINE.md §4 configs 1 (N = 20, stance, v_cmd = 0) and 2 (N = 100, walk) exactly as specified: cold start from task.info's
    initialState, one SQP iteration, full size against the oracle."""
    x0, x, u, par, dt = make_centroidal_problem(cmodel, n_nodes=n, batch=1, gait=gait, v_cmd=v_cmd)
    out = cgpu.run(x0, x, u, par, dt)
    r = coracle.cent_sqp_iteration(dt, x0[0], x[0], u[0], par[0], threads=4)
    assert_step(out, r, 0, f"config {cfg}")
    assert_perf(out["perf_before"][0], r["perf_before"], "before")
    assert_perf(out["perf_after"][0], r["perf_after"], "after")
    assert_kkt(out["kkt"][0], np.abs(cgpu.debug_read(_abi.BLK_G)[0]).max(), f"config {cfg}")
    assert r["kkt"][1] <= 1e-10


def test_config2_size_properties(cgpu, cmodel):
    """BASELINE config 2 (centroidal, N = 100, one instance): size-independent properties of the QP step."""
    x0, x, u, par, dt = make_centroidal_problem(cmodel, n_nodes=100, batch=1, gait="walk")
    x0 = x0.copy()
    x0[:, :CNX] += 1e-3
    out = cgpu.run(x0, x, u, par, dt)
    dx, du = out["dx"], out["du"]
    AB, b = cgpu.debug_read(_abi.BLK_AB), cgpu.debug_read(_abi.BLK_BVEC)
    CDe, ne = cgpu.debug_read(_abi.BLK_CDE), cgpu.debug_read(_abi.BLK_NE)
    assert np.abs(dx[:, 0] - (x0 - x[:, 0])).max() <= 1e-12
    z = np.concatenate([dx[:, :-1], du], axis=2)
    assert_linear_residual(AB, z, np.abs(b) + np.abs(dx[:, 1:]), dx[:, 1:] - np.einsum("bkij,bkj->bki", AB, z) - b, 1e-10, "linearised dynamics")
    assert_linear_residual(CDe[..., :NZ], z, CDe[..., NZ], np.einsum("bkrj,bkj->bkr", CDe[..., :NZ], z) + CDe[..., NZ], 1e-10,
                           "linearised equality constraints")
    assert set(np.unique(ne)) <= {12, 13, 14} and 13 in ne
    assert_kkt(out["kkt"][0], np.abs(cgpu.debug_read(_abi.BLK_G)[0]).max(), "config 2")
    assert np.all(np.isfinite(out["x"])) and not out["x"][..., CNX:].any()


def test_multi_iteration_with_linesearch(cmodel):
    """sqpIteration > 1 with the filter line search, device resident: equal to repeated hsqp_solve calls of a line-search solver;
    above g_max the filter only accepts steps that lower the constraint violation, so it never grows."""
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    x0, x, u, par, dt = make_centroidal_problem(cmodel, n_nodes=10, batch=3, perturb=True, seed=9)
    s = HipSqpSolver(cmodel, max_nodes=10, max_batch=3, linesearch=True)
    try:
        xs, us, viol = x, u, []
        for _ in range(3):
            out = s.run(x0, xs, us, par, dt)
            assert np.all(out["step_type"] != _abi.STEP_FULL) and np.all((out["alpha"] >= 0.0) & (out["alpha"] <= 1.0))
            if not viol:
                viol.append([np.sqrt(p["dynamics_sse"] + p["equality_sse"]) for p in out["perf_before"]])
            viol.append([np.sqrt(p["dynamics_sse"] + p["equality_sse"]) for p in out["perf_after"]])
            xs, us = out["x"], out["u"]
        s.upload(x0, x, u, par, dt)
        s.iterate(3, take_step=True, linesearch=True)
        res = s.download()
        assert np.array_equal(res["x"], xs) and np.array_equal(res["u"], us)
        viol = np.array(viol)
        g_max = s.linesearch_settings().g_max
        grow = (viol[1:] > viol[:-1] * (1 + 1e-12)) & (viol[:-1] > g_max)
        assert not grow.any()
        assert np.all(np.isfinite(res["x"])) and not res["x"][..., CNX:].any()
    finally:
        s.close()


def test_invalid_centroidal_inputs_fail_cleanly(cgpu, cmodel):
    from wb_humanoid_mpc_amd.solver import HsqpError
    x0, x, u, par, dt = make_centroidal_problem(cmodel, n_nodes=4, batch=1, gait="stance")
    cgpu.run(x0, x, u, par, dt)
    bad = x.copy()
    bad[0, 1, 40] = 1.0
    with pytest.raises(HsqpError) as e:
        cgpu.run(x0, bad, u, par, dt)
    assert e.value.code == _abi.ERR_BAD_ARG


def test_centroidal_policy_and_joint_torques(cgpu, cmodel, model, oracle):
    """hsqp_evaluate_policy / hsqp_joint_torques on a centroidal handle (CentroidalMpcMrtJointController.cpp:155-175): the policy is
    the interpolated trajectory; the torques are the reference's computeJointTorques at q = the state's generalized coordinates,
    qd = the generalized velocities of the centroidal momentum (independent numpy restatement: reference.centroidal_base_velocity),
    the policy's wrenches and the desired joint accelerations handed over in entries 35..57 — checked against the whole-body
    oracle's full inverse dynamics."""
    from test_policy import oracle_torques
    from wb_humanoid_mpc_amd.reference import centroidal_base_velocity
    B, N = 3, 8
    x0, x, u, par, dt = make_centroidal_problem(cmodel, n_nodes=N, batch=B, perturb=True, seed=3)
    out = cgpu.run(x0, x, u, par, dt)
    tq = np.array([0.004, 1.7 * dt, N * dt])
    xp, up, tau = cgpu.evaluate_policy(tq)
    t = np.arange(N + 1) * dt
    nj = cmodel.nj
    rng = np.random.default_rng(1)
    for b in range(B):
        xw = np.array([np.interp(tq[b], t, out["x"][b][:, i]) for i in range(NX)])
        ue = np.vstack([out["u"][b], out["u"][b][-1:]])
        uw = np.array([np.interp(tq[b], t, ue[:, i]) for i in range(NU)])
        np.testing.assert_allclose(xp[b], xw, rtol=0, atol=1e-12)
        np.testing.assert_allclose(up[b], uw, rtol=0, atol=1e-9)
        assert not xp[b][CNX:].any()

        def want_torques(xc, uc, qdd):
            q = xc[6:CNX]
            vb = centroidal_base_velocity(cmodel, q, xc[:6], uc[12:])
            x_wb = np.concatenate([q, vb, uc[12:]])
            u_wb = np.concatenate([uc[:12], qdd])
            return oracle_torques(oracle, x_wb, u_wb)

        want = want_torques(xp[b], up[b], np.zeros(nj))
        assert np.abs(tau[b] - want).max() <= 1e-8 * max(1.0, np.abs(want).max())
        # with a PD-shaped desired joint acceleration in the padding entries of the state row
        qdd = rng.standard_normal(nj)
        xr = xp[b].copy()
        xr[CNX:] = qdd
        got = cgpu.joint_torques(xr, up[b])[0]
        want = want_torques(xp[b], up[b], qdd)
        assert np.abs(got - want).max() <= 1e-8 * max(1.0, np.abs(want).max())
    assert np.abs(tau).max() > 1.0


@pytest.mark.parametrize("n,batch,gait", [(100, 1, "walk"), (20, 1, "stance"), (37, 3, "run"), (64, 12, "walk")])
def test_parallel_in_time_backward_sweep_equals_the_serial_recursion(cmodel, coracle, n, batch, gait):
    """hsqp_scan.h on the device (k_scan_init / k_scan_combine / k_scan_gains / k_scan_forward) against k_riccati<35> on the same
    QP, and against the oracle; batch 12 forces the scan beyond its automatic range."""
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    x0, x, u, par, dt = make_centroidal_problem(cmodel, n_nodes=n, batch=batch, gait=gait, perturb=batch > 1, seed=13)
    outs = {}
    for mode in ("serial", "parallel"):
        s = HipSqpSolver(cmodel, max_nodes=n, max_batch=batch, riccati=mode)
        try:
            outs[mode] = s.run(x0, x, u, par, dt)
        finally:
            s.close()
    a, b = outs["serial"], outs["parallel"]
    assert np.abs(a["dx"] - b["dx"]).max() <= TRAJ_ABS and np.abs(a["du"] - b["du"]).max() <= TRAJ_ABS
    for i in range(batch):
        assert_perf(b["perf_after"][i], a["perf_after"][i], f"scan vs serial, instance {i}")
    r = coracle.cent_sqp_iteration(dt, x0[0], x[0], u[0], par[0], threads=4)
    assert_step(b, r, 0, "scan vs oracle")


@pytest.mark.parametrize("riccati", ["serial", "parallel"])
def test_no_kernel_reads_lds_it_did_not_write(cmodel, riccati):
    """The centroidal kernels with NaN bit patterns in every LDS word in front of every launch (HSQP_POISON_LDS, csrc/hsqp_capi.hip): the same bits
    as without (see the whole-body test of the same name)."""
    from test_gpu_parity import _poisoned_and_clean

    def make():
        return [make_centroidal_problem(cmodel, n_nodes=n, batch=b, gait=g, perturb=True, seed=7 + n) for n, b, g in ((100, 2, "walk"), (14, 3, "run"))]
    poisoned, clean = _poisoned_and_clean(cmodel, make, max_nodes=100, max_batch=4, riccati=riccati)
    for a, b in zip(poisoned, clean):
        for key in ("dx", "du", "x", "u", "kkt"):
            assert np.isfinite(a[key]).all(), key
            assert np.array_equal(a[key], b[key]), key
