"""Non-uniform time grid with event nodes (hsqp_problem::dt_nodes / hsqp_reference::node_times):
the ocs2 multiple-shooting grid splits at the mode-switch times, with a pre- and a post-event node joined by the identity jump map.
Oracle properties, kernel sources on the host against the oracle, the HIP path through the C ABI against the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tolerances import TRAJ_ABS, assert_kkt, assert_perf, assert_step
from wb_humanoid_mpc_amd import _abi
from wb_humanoid_mpc_amd.reference import (EVENT_EPS, build_node_params_at, cold_start, event_grid, pack_reference, swing_config, tile_gait,
                                           time_discretization_with_events, velocity_command_targets)

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)


def walk_problem_with_events(model, horizon=0.6, perturb_seed=None, gait="walk", t_start=-0.45, t0=0.0):
    """A walk horizon whose mode switches fall INSIDE intervals of the nominal grid: (x0, x, u, par, dts, node_times, schedule, targets)."""
    dt = model.sqp["dt"]
    schedule = tile_gait(model.gaits[gait], t_start, 3.0 + t0)
    x0 = model.initial_state.copy()
    if perturb_seed is not None:
        rng = np.random.default_rng(perturb_seed)
        x0[6:6 + model.nj] += 0.03 * rng.standard_normal(model.nj)
        x0[6 + model.nj:] += 0.05 * rng.standard_normal(6 + model.nj)
    targets = velocity_command_targets(model, (0.3, 0.0, 0.7925, 0.0), t0, x0, horizon)
    dts, node_times = event_grid(t0, t0 + horizon, dt, schedule.event_times)
    par = build_node_params_at(model, schedule, targets, node_times)
    x, u = cold_start(model, x0, par)
    if perturb_seed is not None:      # leave the cold start so that the jump defects are non-zero
        x = x + 0.01 * rng.standard_normal(x.shape)
        u = u + 0.5 * rng.standard_normal(u.shape)
    return x0, x, u, par, dts, node_times, schedule, targets


def test_time_discretization_follows_the_ocs2_rules():
    t, post = time_discretization_with_events(0.0, 0.35, 0.035, [0.1, 0.2505, 0.33, 0.9])
    assert t[0] == 0.0 and t[-1] == 0.35 and np.all(np.diff(t) >= 0.0)
    for e in (0.1, 0.2505, 0.33):                                    # every event inside the horizon: a pre and a post node
        idx = np.flatnonzero(t == e)
        assert len(idx) == 2 and not post[idx[0]] and post[idx[1]]
    assert 0.9 not in t and post.sum() == 3
    d = np.diff(t)
    assert np.all(d[~post[1:]] > 1e-4) and np.all(d[post[1:]] == 0.0) and d.max() <= 0.035 + 1e-15
    # an event closer than dt_min to the node before it replaces that node
    t2, _ = time_discretization_with_events(0.0, 0.2, 0.035, [0.07 + 5e-5])
    assert 0.07 not in t2 and (t2 == 0.07 + 5e-5).sum() == 2


def test_oracle_event_interval_is_an_identity_jump(model, oracle):
    x0, x, u, par, dts, node_times, _, _ = walk_problem_with_events(model, perturb_seed=3)
    ev = np.flatnonzero(dts == 0.0)
    assert len(ev) >= 2 and np.all(par[ev, _abi.P_CONTACT:_abi.P_CONTACT + 2] != par[ev + 1, _abi.P_CONTACT:_abi.P_CONTACT + 2]) is not None
    assert any((par[k, _abi.P_CONTACT:_abi.P_CONTACT + 2] != par[k + 1, _abi.P_CONTACT:_abi.P_CONTACT + 2]).any() for k in ev)   # modes change across events
    oracle.set_grid(dts)
    try:
        lq = oracle.lq(0.0, x, u, par, threads=4)
        r = oracle.sqp_iteration(0.0, x0, x, u, par, threads=4)
        perf = oracle.performance(0.0, x, u, par, threads=4)
    finally:
        oracle.set_grid(None)
    for k in ev:
        assert np.array_equal(lq["AB"][k][:, :_abi.NX], np.eye(_abi.NX)) and not lq["AB"][k][:, _abi.NX:].any()
        np.testing.assert_array_equal(lq["b"][k], x[k] - x[k + 1])
        assert lq["ne"][k] == 0 and lq["cost"][k] == 0.0
        assert not r["du"][k].any()                                                   # the inputs of a pre-event node stay
        np.testing.assert_allclose(r["dx"][k + 1], r["dx"][k] + lq["b"][k], atol=1e-12)   # dx+ = dx + defect
    assert r["kkt"][0] <= 1e-9 * max(1.0, np.abs(lq["g"]).max()) and r["kkt"][1] <= 1e-10
    # performance index: event defects enter the dynamics SSE unscaled
    jump = sum(((x[k] - x[k + 1]) ** 2).sum() for k in ev)
    assert perf["dynamics_sse"] > jump > 0.0
    assert_perf(perf, r["perf_before"], "performance vs sqp_iteration")      # (OpenMP reduction order: equal to rounding only)


def test_uniform_grid_given_as_dt_nodes_changes_nothing(model, oracle):
    from wb_humanoid_mpc_amd.reference import make_problem
    x0, x, u, par, dt = make_problem(model, n_nodes=8, batch=1, perturb=True, seed=2)
    a = oracle.sqp_iteration(dt, x0[0], x[0], u[0], par[0], threads=2)
    oracle.set_grid(np.full(8, dt))
    try:
        b = oracle.sqp_iteration(0.0, x0[0], x[0], u[0], par[0], threads=2)
    finally:
        oracle.set_grid(None)
    assert np.array_equal(a["dx"], b["dx"]) and np.array_equal(a["du"], b["du"])
    assert_perf(a["perf_after"], b["perf_after"])


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "hostemu")])
    lib = C.CDLL(os.path.join(HERE, "hostemu", "libhsqp_hostemu.so"))
    lib.emu_create.restype = C.c_void_p
    return lib


@pytest.mark.parametrize("seed", [None, 7])
def test_kernel_sources_on_the_event_grid_equal_the_oracle(model, oracle, emu, seed):
    x0, x, u, par, dts, _, _, _ = walk_problem_with_events(model, perturb_seed=seed)
    n = len(dts)
    err = C.create_string_buffer(256)
    h = C.c_void_p(emu.emu_create(C.byref(model.desc), err, 256))
    assert h.value, err.value
    P = lambda a: a.ctypes.data_as(_dp)  # noqa: E731
    xn, un, dx, du = np.zeros_like(x), np.zeros_like(u), np.zeros_like(x), np.zeros_like(u)
    kkt, pb, pa = np.zeros(2), np.zeros(3), np.zeros(3)
    d = np.ascontiguousarray(dts)
    rc = emu.emu_sqp_iteration(h, n, C.c_double(0.0), P(x0), P(x), P(u), P(par), P(xn), P(un), P(dx), P(du), P(kkt), P(pb), P(pa), None, P(d))
    assert rc == 0
    oracle.set_grid(dts)
    try:
        r = oracle.sqp_iteration(0.0, x0, x, u, par, threads=4)
    finally:
        oracle.set_grid(None)
    scale = max(np.abs(r["dx"]).max(), np.abs(r["du"]).max())
    assert np.abs(dx - r["dx"]).max() <= TRAJ_ABS + 1e-10 * scale and np.abs(du - r["du"]).max() <= TRAJ_ABS + 1e-10 * scale
    for got, want in ((pb, r["perf_before"]), (pa, r["perf_after"])):
        assert np.allclose(got, [want["cost"], want["dynamics_sse"], want["equality_sse"]], rtol=1e-9, atol=1e-12)
    assert kkt[1] <= 1e-9
    emu.emu_destroy(h)


def first_interval_event_problem(model, perturb_seed=7):
    """A horizon that starts 5e-5 s BEFORE a mode switch: the switch lies within dt_min of the initial time, so ocs2's grid replaces
    the initial node by the pre-event node and the FIRST interval is the event (dt_nodes[0] = 0).  In a receding-horizon run this
    happens about once per hundred switches."""
    sched = tile_gait(model.gaits["walk"], -0.45, 4.0)
    e = [t for t in sched.event_times if t > 0.1][0]
    prob = walk_problem_with_events(model, perturb_seed=perturb_seed, t0=e - 5e-5)
    assert prob[4][0] == 0.0 and (prob[4] == 0.0).sum() >= 2
    return prob


def test_kernel_sources_with_an_event_as_the_first_interval(model, oracle, emu):
    x0, x, u, par, dts, _, _, _ = first_interval_event_problem(model)
    n = len(dts)
    err = C.create_string_buffer(256)
    h = C.c_void_p(emu.emu_create(C.byref(model.desc), err, 256))
    assert h.value, err.value
    P = lambda a: a.ctypes.data_as(_dp)  # noqa: E731
    xn, un, dx, du = np.zeros_like(x), np.zeros_like(u), np.zeros_like(x), np.zeros_like(u)
    kkt, pb, pa = np.zeros(2), np.zeros(3), np.zeros(3)
    d = np.ascontiguousarray(dts)
    assert emu.emu_sqp_iteration(h, n, C.c_double(0.0), P(x0), P(x), P(u), P(par), P(xn), P(un), P(dx), P(du), P(kkt), P(pb), P(pa), None, P(d)) == 0
    oracle.set_grid(dts)
    try:
        r = oracle.sqp_iteration(0.0, x0, x, u, par, threads=4)
    finally:
        oracle.set_grid(None)
    scale = max(np.abs(r["dx"]).max(), np.abs(r["du"]).max())
    assert np.abs(dx - r["dx"]).max() <= TRAJ_ABS + 1e-10 * scale and np.abs(du - r["du"]).max() <= TRAJ_ABS + 1e-10 * scale
    assert not du[0].any() and np.allclose(dx[1], dx[0] + (x[0] - x[1]), atol=1e-12)     # node 0 is the identity jump
    assert np.allclose(dx[0], x0 - x[0], atol=0)
    assert kkt[1] <= 1e-9
    emu.emu_destroy(h)


def test_event_grid_phases_are_race_free(model):
    """The reverse-order host build (every phase executes its items backwards) reproduces the forward build bit for bit on the
    event grid too (jump_node_qp and the dt = 0 paths of the LQ / value phases)."""
    x0, x, u, par, dts, _, _, _ = walk_problem_with_events(model, perturb_seed=5)
    n = len(dts)
    outs = []
    for lib_name in ("libhsqp_hostemu.so", "libhsqp_hostemu_rev.so"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "hostemu")])
        lib = C.CDLL(os.path.join(HERE, "hostemu", lib_name))
        lib.emu_create.restype = C.c_void_p
        err = C.create_string_buffer(256)
        h = C.c_void_p(lib.emu_create(C.byref(model.desc), err, 256))
        P = lambda a: a.ctypes.data_as(_dp)  # noqa: E731
        xn, un, dx, du = np.zeros_like(x), np.zeros_like(u), np.zeros_like(x), np.zeros_like(u)
        kkt, pb, pa = np.zeros(2), np.zeros(3), np.zeros(3)
        d = np.ascontiguousarray(dts)
        assert lib.emu_sqp_iteration(h, n, C.c_double(0.0), P(x0), P(x), P(u), P(par), P(xn), P(un), P(dx), P(du), P(kkt), P(pb), P(pa), None, P(d)) == 0
        outs.append((dx.copy(), du.copy(), pa.copy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])


def test_policy_interpolation_on_a_grid_with_events(emu, rng):
    N = 7
    dts = np.array([0.03, 0.02, 0.0, 0.035, 0.035, 0.0, 0.01])
    t = np.concatenate([[0.0], np.cumsum(dts)])
    xt, ut = rng.standard_normal((N + 1, _abi.NX)), rng.standard_normal((N, _abi.NU))
    P = lambda a: a.ctypes.data_as(_dp)  # noqa: E731
    for s in (0.0, 0.01, 0.03, 0.0499, 0.05, 0.0501, 0.07, 0.12, 0.13, 0.5, -1.0):
        x, u = np.zeros(_abi.NX), np.zeros(_abi.NU)
        emu.emu_policy_interpolate_grid(P(xt), P(ut), N, P(dts), C.c_double(s), P(x), P(u))
        sc = min(max(s, 0.0), t[-1])
        # reference: piece-wise linear in time on the stamps, the POST-event node at an event time
        k = max(i for i in range(N + 1) if t[i] <= sc)
        k = min(k, N - 1)
        while k < N - 1 and dts[k] == 0.0:
            k += 1
        a = (sc - t[k]) / dts[k] if dts[k] > 0 else 1.0
        np.testing.assert_allclose(x, (1 - a) * xt[k] + a * xt[k + 1], atol=1e-12)
        ku, au = (k, a) if k <= N - 2 else (N - 2, 1.0)
        if dts[ku] == 0.0:
            au = 1.0
        elif ku + 1 <= N - 1 and dts[ku + 1] == 0.0:
            au = 0.0          # node ku + 1 is a pre-event node: the input before it is held up to the switch
        np.testing.assert_allclose(u, (1 - au) * ut[ku] + au * ut[ku + 1], atol=1e-12)


# ---------------------------------------------------------------------------------------------- the HIP path
@pytest.mark.gpu
def test_device_on_the_event_grid_equals_the_oracle(model, oracle):
    """Three instances with different perturbations on a 0.6 s walk horizon with its mode switches as event nodes: step, performance
    index, KKT residual against the oracle at the BASELINE.md §6 tolerances; the same problem with the node table generated on the
    device from the mode schedule (hsqp_upload_reference with node_times); policy evaluation across an event."""
    from wb_humanoid_mpc_amd.solver import HipSqpSolver, HsqpError
    probs = [walk_problem_with_events(model, perturb_seed=s) for s in (None, 7, 11)]
    dts = probs[0][4]
    N, B = len(dts), len(probs)
    assert (dts == 0.0).sum() >= 2
    x0, x, u, par = (np.stack([p[i] for p in probs]) for i in range(4))
    s = HipSqpSolver(model, max_nodes=N, max_batch=B)
    try:
        out = s.run(x0, x, u, par, dts)
        oracle.set_grid(dts)
        try:
            for b in range(B):
                r = oracle.sqp_iteration(0.0, x0[b], x[b], u[b], par[b], threads=os.cpu_count() or 4)
                assert_step(out, r, b, "event grid")
                assert_perf(out["perf_before"][b], r["perf_before"], f"instance {b} before")
                assert_perf(out["perf_after"][b], r["perf_after"], f"instance {b} after")
                assert_kkt(out["kkt"][b], out["grad_inf"][b], f"instance {b}")
        finally:
            oracle.set_grid(None)
        ev = np.flatnonzero(dts == 0.0)
        assert not out["du"][:, ev].any()
        # device-generated node table on the same grid
        node_times = probs[0][5]
        scheds, targets = [p[6] for p in probs], [p[7] for p in probs]
        s.upload_reference(x0, x, u, dts, 0.0, *pack_reference(scheds, targets), swing_config(model), node_times=node_times)
        np.testing.assert_allclose(s.device_params(), par, rtol=0, atol=1e-12)
        s.iterate(1, take_step=True, kkt=True)
        out2 = s.download()
        assert np.abs(out2["x"] - out["x"]).max() <= 1e-9 and np.abs(out2["u"] - out["u"]).max() <= 1e-8
        # policy evaluation just before / at / after the first event
        te = node_times[ev[0]]
        xs, us, tau = s.evaluate_policy(np.array([te - 1e-3, te, te + 1e-3]))
        assert np.all(np.isfinite(tau))
        np.testing.assert_allclose(xs[1], out2["x"][1, ev[0] + 1], atol=1e-12)       # at the event time: the post-event node
        # an event as the last interval is rejected — before anything is copied: the handle holds no half-uploaded problem afterwards
        bad = dts.copy(); bad[-1] = 0.0
        with pytest.raises(HsqpError) as e:
            s.run(x0, x, u, par, bad)
        assert e.value.code == _abi.ERR_BAD_ARG
        with pytest.raises(HsqpError) as e:
            s.iterate(1)
        assert e.value.code == _abi.ERR_BAD_ARG
    finally:
        s.close()


@pytest.mark.gpu
def test_device_with_an_event_as_the_first_interval(model, oracle):
    """dt_nodes[0] = 0 (a mode switch within dt_min after the initial time) is a legal ocs2 grid: accepted and solved like any other
    event stage (it used to be rejected, which threw out of the MPC loop about once per hundred switches)."""
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    x0, x, u, par, dts, node_times, sched, targets = first_interval_event_problem(model)
    N = len(dts)
    s = HipSqpSolver(model, max_nodes=N, max_batch=1)
    try:
        out = s.run(x0[None], x[None], u[None], par[None], dts)
        oracle.set_grid(dts)
        try:
            r = oracle.sqp_iteration(0.0, x0, x, u, par, threads=os.cpu_count() or 4)
        finally:
            oracle.set_grid(None)
        assert_step(out, r, 0, "event as the first interval")
        assert_perf(out["perf_after"][0], r["perf_after"], "after")
        assert_kkt(out["kkt"][0], out["grad_inf"][0], "first-interval event")
        assert not out["du"][0, 0].any()
        # the device-generated table on the same grid, and the policy at the very start (the post-event node)
        t0 = node_times[0]
        s.upload_reference(x0[None], x[None], u[None], dts, t0, *pack_reference([sched], [targets]), swing_config(model), node_times=node_times)
        np.testing.assert_allclose(s.device_params()[0], par, rtol=0, atol=1e-12)
        s.iterate(1, take_step=True)
        out2 = s.download()
        xs, us, tau = s.evaluate_policy(np.array([0.0]))
        np.testing.assert_allclose(xs[0], out2["x"][0, 1], atol=1e-12)
        assert np.all(np.isfinite(tau))
    finally:
        s.close()


@pytest.mark.gpu
def test_parallel_in_time_sweep_on_a_long_event_grid(model):
    """One instance on a 1.8 s walk horizon (>= 48 intervals, several of them events): the default path takes the parallel-in-time
    sweep, whose stage elements treat an event interval like any other stage (A~ = I, B~ = 0, R~ = I): against the serial recursion,
    with the KKT gate's verdict — the scan's result if accepted (<= 2e-10 of the step's scale from the serial one), else the serial
    recursion's bit for bit."""
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    x0, x, u, par, dts, node_times, _, _ = walk_problem_with_events(model, horizon=1.8)
    N = len(dts)
    assert N >= 48 and (dts == 0.0).sum() >= 4
    outs = {}
    for mode in ("auto", "serial"):
        s = HipSqpSolver(model, max_nodes=N, max_batch=1, riccati=mode)
        try:
            outs[mode] = s.run(x0[None], x[None], u[None], par[None], dts)
            outs[mode]["fallbacks"] = s.scan_fallbacks()
        finally:
            s.close()
    a, b = outs["serial"], outs["auto"]
    ev = np.flatnonzero(dts == 0.0)
    assert not b["du"][:, ev].any()
    if b["fallbacks"]:
        assert np.array_equal(a["dx"], b["dx"]) and np.array_equal(a["du"], b["du"])
    else:
        sc = max(1.0, np.abs(a["dx"]).max(), np.abs(a["du"]).max())
        err = max(np.abs(a["dx"] - b["dx"]).max(), np.abs(a["du"] - b["du"]).max())
        assert err <= 2e-10 * sc, (err, sc)
        assert_perf(b["perf_after"][0], a["perf_after"][0], "scan vs serial on the event grid", rel=1e-9)
    print(f"event grid N={N}: fallbacks {b['fallbacks']}")


@pytest.mark.gpu
def test_device_centroidal_event_grid_equals_the_oracle(cmodel, coracle):
    from wb_humanoid_mpc_amd.reference import build_centroidal_node_params, centroidal_velocity_command_targets
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    dt = cmodel.sqp["dt"]
    horizon = 0.5
    schedule = tile_gait(cmodel.gaits["walk"], -0.45, 3.0)
    x0 = cmodel.initial_state.copy()
    targets = centroidal_velocity_command_targets(cmodel, (0.3, 0.0, 0.7925, 0.0), 0.0, x0, horizon)
    dts, node_times = event_grid(0.0, horizon, dt, schedule.event_times)
    N = len(dts)
    # the centroidal table builder samples t0 + k dt: build it per node on the event grid
    par = np.stack([build_centroidal_node_params(cmodel, schedule, targets, t, dt, 0)[0] for t in node_times])
    x0p = np.zeros(_abi.NX); x0p[:_abi.CNX] = x0
    x, u = cold_start(cmodel, x0p, par)
    rng = np.random.default_rng(4)
    x[:, :_abi.CNX] += 0.01 * rng.standard_normal((N + 1, _abi.CNX))
    s = HipSqpSolver(cmodel, max_nodes=N, max_batch=1, riccati="serial")
    try:
        out = s.run(x0p, x, u, par, dts)
    finally:
        s.close()
    coracle.set_grid(dts)
    try:
        r = coracle.cent_sqp_iteration(0.0, x0p, x, u, par, threads=os.cpu_count() or 4)
    finally:
        coracle.set_grid(None)
    assert (dts == 0.0).sum() >= 1
    assert_step(out, r, 0, "centroidal event grid")
    assert_perf(out["perf_after"][0], r["perf_after"], "after")
    assert not out["du"][0, np.flatnonzero(dts == 0.0)].any()
    # the parallel-in-time sweep on the same grid (an event stage is the scan element (I, defect, 0, 0, 0))
    s2 = HipSqpSolver(cmodel, max_nodes=N, max_batch=1, riccati="parallel")
    try:
        out2 = s2.run(x0p, x, u, par, dts)
    finally:
        s2.close()
    assert_step(out2, r, 0, "centroidal event grid, scan")
