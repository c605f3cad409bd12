"""Health of the benchmarked problems on the device: the convergence behaviour tests/test_oracle_convergence.py
establishes for the oracle, reproduced by the HIP path through the C ABI — at the BASELINE configurations themselves, which the GPU
iterates in milliseconds.  DESIGN.md §2b explains the numbers."""
import os

import numpy as np
import pytest

from test_oracle_convergence import violation
from tolerances import TRAJ_ABS, assert_perf
from wb_humanoid_mpc_amd import _abi
from wb_humanoid_mpc_amd.reference import (make_centroidal_problem, make_problem, pack_reference, swing_config, tile_gait,
                                           velocity_command_targets, weight_compensating_input, cold_start, build_node_params)

pytestmark = pytest.mark.gpu


def run_sqp(solver, problem, iterations, stop=None):
    """`iterations` SQP iterations with the filter line search (a solver created with linesearch=True: hsqp_solve does what the
    reference's SqpSolver::runImpl does per iteration), the accepted trajectory fed back as the next linearisation point."""
    x0, x, u, par, dt = problem
    hist, out = [], None
    for it in range(iterations):
        out = solver.run(x0, x, u, par, dt)
        hist.append(dict(viol=np.array([violation(p) for p in out["perf_before"]]), cost=np.array([p["cost"] for p in out["perf_before"]]),
                         dx=np.abs(out["dx"]).reshape(len(x0), -1).max(1), alpha=out["alpha"].copy(), step_type=out["step_type"].copy(),
                         viol_after=np.array([violation(p) for p in out["perf_after"]])))
        if stop and stop(hist[-1]):
            break
        x, u = out["x"], out["u"]
    return hist, out


def converged(h):
    return bool(np.all(h["dx"] < 1e-3) and np.all(h["viol"] < 1e-4))


@pytest.mark.parametrize("formulation,limit", [("wb", 14), ("centroidal", 2)])
def test_stance_converges_on_the_device(model, cmodel, formulation, limit):
    """|dx|_inf < 1e-3 and violation < 1e-4 within the oracle's iteration count (tests/test_oracle_convergence.py): whole-body stance
    (N = 20) in 14 iterations, BASELINE config 1 (centroidal, N = 20, stance) in 2."""
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    cent = formulation == "centroidal"
    m = cmodel if cent else model
    p = (make_centroidal_problem if cent else make_problem)(m, n_nodes=20, batch=1, gait="stance", v_cmd=(0, 0, 0.7925, 0))
    s = HipSqpSolver(m, max_nodes=20, max_batch=1, linesearch=True)
    try:
        hist, _ = run_sqp(s, p, limit + 2, stop=converged)
    finally:
        s.close()
    assert converged(hist[-1]) and len(hist) <= limit, (len(hist), hist[-1])


CASES = {  # name: (formulation, N, batch, perturbed, iterations, violation reached)
    "config2": ("centroidal", 100, 1, False, 120, 1e-2),
    "config3": ("wb", 100, 1, False, 200, 1e-2),
    # perturbed starts (joint rates off by 0.1 rad/s, poses by 0.05 rad) are slower: measured 24-63 -> 0.010, 0.008, 0.009, 0.023 after
    # 500 iterations (instance 3 creeps along the band's edge with step lengths of 2e-3)
    "config4_slice": ("wb", 100, 4, True, 500, 3e-2),
}


@pytest.mark.parametrize("name", list(CASES))
def test_cold_start_walk_configs_settle_under_the_filter(model, cmodel, name):
    """BASELINE configs 2, 3 and the first four instances of 4, cold-started into the walk gait: the full step is unusable there
    (DESIGN.md §2b), the filter line search brings every instance inside the filter's g_max band within the stated iteration count,
    never accepts a step that raises the violation above that band, and ends among sane trajectories (finite cost near the stance
    optimum's, no numeric failure on the way).  Below g_max the filter trades violation for cost (ocs2::FilterLinesearch's dual
    branch), so |dx| < 1e-3 is NOT reached on these problems by the oracle either: the measured behaviour is stated, not hidden."""
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    formulation, N, B, perturb, iterations, target = CASES[name]
    cent = formulation == "centroidal"
    m = cmodel if cent else model
    p = (make_centroidal_problem if cent else make_problem)(m, n_nodes=N, batch=B, gait="walk", perturb=perturb)
    s = HipSqpSolver(m, max_nodes=N, max_batch=B, linesearch=True)
    try:
        hist, out = run_sqp(s, p, iterations)
        g_max = s.linesearch_settings().g_max
    finally:
        s.close()
    v = np.array([h["viol"] for h in hist])                   # [iteration][instance]
    grow = (v[1:] > v[:-1] * (1 + 1e-12)) & (v[:-1] > g_max) & (v[1:] > g_max)
    assert not grow.any(), np.argwhere(grow)[:5]
    if not perturb:
        assert np.all(hist[0]["alpha"] < 1.0)                 # the full step is rejected at the unperturbed cold start
    assert np.all(v[-1] <= target), v[-1]
    assert np.all(np.isfinite(out["x"])) and np.all(np.isfinite(out["u"]))
    assert np.all(hist[-1]["cost"] < 50.0), hist[-1]["cost"]
    print(f"{name}: violation {v[0].max():.2e} -> {v[-1].max():.2e}, cost {hist[-1]['cost']}, last |dx| {hist[-1]['dx']}")


@pytest.mark.parametrize("name", ["config2", "config3"])
def test_first_linesearch_iterations_at_full_size_equal_the_oracle(model, cmodel, oracle, coracle, name):
    """Three line-search iterations of configs 2 and 3 at full size: step length, step type and the accepted trajectory after each
    iteration equal the oracle's (BASELINE.md §6 tolerance on the trajectories)."""
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    formulation, N, B, perturb, _, _ = CASES[name]
    cent = formulation == "centroidal"
    m, o = (cmodel, coracle) if cent else (model, oracle)
    x0, x, u, par, dt = (make_centroidal_problem if cent else make_problem)(m, n_nodes=N, batch=1, gait="walk")
    threads = os.cpu_count() or 4
    s = HipSqpSolver(m, max_nodes=N, max_batch=1, linesearch=True)
    try:
        xs, us = x, u
        xo, uo = x[0], u[0]
        for it in range(3):
            out = s.run(x0, xs, us, par, dt)
            r = (o.cent_sqp_iteration if cent else o.sqp_iteration)(dt, x0[0], xo, uo, par[0], threads=threads)
            ls = (o.cent_linesearch if cent else o.linesearch)(dt, xo, uo, r["dx"], r["du"], par[0], r["armijo"], threads=threads)
            assert out["alpha"][0] == ls["alpha"] and out["step_type"][0] == ls["step_type"], (it, out["alpha"][0], ls["alpha"])
            # iteration 1 from the cold start: BASELINE.md §6 as is.  DECLARED RELAXATION for iterations 2 and 3: they linearise about the
            # first accepted trial point, which is far from feasible (|du| 600 - 1000 N on these QPs); measured error 1.6e-7 (config 2)
            # and 5.4e-8 (config 3) = 3e-10 / 7e-11 of the step, allowed: 1e-8 + 1e-9 |step|_inf.  On these iterates the parallel-in-time
            # sweep fails its KKT gate (stationarity 1e-5 .. 1e-4) and the iteration is redone with the serial recursion: asserted below.
            lim = TRAJ_ABS + (0.0 if it == 0 else 1e-9 * max(np.abs(r["dx"]).max(), np.abs(r["du"]).max()))
            assert np.abs(out["x"][0] - ls["x"]).max() <= lim and np.abs(out["u"][0] - ls["u"]).max() <= lim, it
            assert_perf(out["perf_after"][0], ls["perf"], f"{name} iteration {it}", rel=1e-10 if it == 0 else 1e-9)   # same relaxation (measured 2.4e-10)
            xs, us = out["x"], out["u"]
            xo, uo = out["x"][0], out["u"][0]      # the oracle follows the device's trajectory: errors are per iteration, not compounded
        # one instance on 100 nodes takes the parallel-in-time sweep: accepted on the cold-start QP, rejected by the gate on the two
        # ill-conditioned iterates that follow.  Every s.run() uploads its problem, and an upload resets the gate's back-off (ADVICE r3:
        # what a handle returns must not depend on the problems it solved before), so both are attempted and both fall back; inside ONE
        # hsqp_iterate_device call of several iterations the back-off still spares the iterations that follow a rejection (next test)
        assert s.scan_fallbacks() == 2 and s.scan_backoffs() == 0, (s.scan_fallbacks(), s.scan_backoffs())
    finally:
        s.close()


def test_receding_horizon_from_stance_into_walk_on_the_device(model, oracle):
    """The reference's operating regime (one iteration with line search per MPC call, shifted warm start, gait entering from the end
    of the horizon; native horizon N = 30) with the node parameters generated on the device for every call: full steps, a stable
    configuration plan, the robot walks — and the device's closed loop equals the oracle's over the first calls."""
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    N, calls, t_switch, B = 30, 60, 1.2, 2
    dt = model.sqp["dt"]
    horizon = N * dt
    schedule = tile_gait(model.gaits["walk"], t_switch, t_switch + calls * dt + 2 * horizon + 3.0)
    rng = np.random.default_rng(5)
    x0 = np.stack([model.initial_state, model.initial_state + np.concatenate([np.zeros(6), 0.02 * rng.standard_normal(model.nj), np.zeros(6 + model.nj)])])
    s = HipSqpSolver(model, max_nodes=N, max_batch=B)
    t, x, u = 0.0, None, None
    alphas, near, vx = [], [], []
    try:
        for call in range(calls):
            v_cmd = (0.3, 0.0, 0.7925, 0.0) if t + horizon > t_switch else (0.0, 0.0, 0.7925, 0.0)
            targets = [velocity_command_targets(model, v_cmd, t, x0[b], horizon) for b in range(B)]
            par_last = [build_node_params(model, schedule, targets[b], t, dt, N)[N - 1] for b in range(B)] if x is not None else None
            if x is None:
                par = np.stack([build_node_params(model, schedule, targets[b], t, dt, N) for b in range(B)])
                x, u = map(np.stack, zip(*[cold_start(model, x0[b], par[b]) for b in range(B)]))
            else:
                tail = np.stack([weight_compensating_input(model, par_last[b][_abi.P_CONTACT:_abi.P_CONTACT + 2] > 0.5) for b in range(B)])
                x = np.concatenate([x[:, 1:], x[:, -1:]], axis=1)
                u = np.concatenate([u[:, 1:], tail[:, None]], axis=1)
            s.upload_reference(x0, x, u, dt, t, *pack_reference([schedule] * B, targets), swing_config(model))
            s.iterate(1, linesearch=True)
            out = s.download()
            if call < 3:   # the same call in the oracle, from the same inputs
                par0 = build_node_params(model, schedule, targets[0], t, dt, N)
                r = oracle.sqp_iteration(dt, x0[0], x[0], u[0], par0, threads=os.cpu_count() or 4)
                ls = oracle.linesearch(dt, x[0], u[0], r["dx"], r["du"], par0, r["armijo"], threads=os.cpu_count() or 4)
                assert out["alpha"][0] == ls["alpha"]
                assert np.abs(out["x"][0] - ls["x"]).max() <= TRAJ_ABS and np.abs(out["u"][0] - ls["u"]).max() <= TRAJ_ABS
            alphas.append(out["alpha"].copy())
            near.append(np.abs(out["dx"][:, :10, :6 + model.nj]).max())
            x, u = out["x"], out["u"]
            assert np.all(np.isfinite(x)) and np.all(np.isfinite(u))
            x0 = x[:, 1].copy()
            vx.append(x0[:, 6 + model.nj].copy())
            assert np.all((x0[:, 2] > 0.74) & (x0[:, 2] < 0.82))
            t += dt
    finally:
        s.close()
    alphas = np.array(alphas)
    assert (alphas == 1.0).mean() >= 0.9 and np.all(alphas > 0.0)
    assert max(near[1:]) < 0.25, max(near[1:])        # 0.15 for the nominal instance (oracle test); the perturbed one reaches 0.18
    assert np.all(np.max(vx[-10:], axis=0) > 0.1)
