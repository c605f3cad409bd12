"""The kernel SOURCES (wb_humanoid_mpc_amd/csrc/*.h) compiled for the host with a one-thread context
(tests/hostemu) against the oracle: checks the arithmetic of the HIP kernels in the GPU-less container.
The GPU tests (-m gpu) check the same thing through the real device build and the C ABI."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import random_state_input
from test_oracle_lq import perturbed_problem
from wb_humanoid_mpc_amd import _abi

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
P = lambda a: a.ctypes.data_as(_dp)  # noqa: E731


@pytest.fixture(scope="module")
def emu(model):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "hostemu"), "all"])
    lib = C.CDLL(os.path.join(HERE, "hostemu", "libhsqp_hostemu.so"))
    lib.emu_create.restype = C.c_void_p
    err = C.create_string_buffer(256)
    h = C.c_void_p(lib.emu_create(C.byref(model.desc), err, 256))
    assert h.value, err.value
    return lib, h


def test_analytic_flow_jacobian(model, oracle, emu, rng):
    lib, h = emu
    for _ in range(5):
        x, u = random_state_input(model, rng)
        ab, G = np.zeros(6), np.zeros((6, 93))
        lib.emu_stage_eval(h, P(x), P(u), 1, P(ab), P(G))
        f, J = oracle.flow_map_jac(x, u)
        assert np.abs(ab - f[29:35]).max() <= 1e-11 * max(1.0, np.abs(f).max())
        assert np.abs(G - J[29:35]).max() <= 1e-10 * max(1.0, np.abs(J).max())
        ab2 = np.zeros(6)
        lib.emu_stage_eval(h, P(x), P(u), 0, P(ab2), None)
        assert np.allclose(ab, ab2, rtol=1e-13, atol=1e-13)   # the value-only path sums the body forces in a different order


@pytest.mark.parametrize("gait,n", [("stance", 3), ("walk", 6), ("run", 12)])
def test_lq_record_expands_to_the_oracle_blocks(model, oracle, emu, gait, n):
    lib, h = emu
    x0, x, u, par, dt = perturbed_problem(model, n, gait, seed=5)
    lq = oracle.lq(dt, x, u, par)
    RS = lib.emu_rec_size()
    for k in range(n):
        rec = np.zeros(RS)
        lib.emu_lq_node(h, P(x[k]), P(u[k]), P(x[k + 1]), P(par[k]), C.c_double(dt), 1, P(rec))
        AB, H, g, CDe = np.zeros((58, 93)), np.zeros((93, 93)), np.zeros(93), np.zeros((14, 94))
        lib.emu_expand(P(rec), C.c_double(dt), P(AB), P(H), P(g), P(CDe))
        for a, b in ((AB, lq["AB"][k]), (H, lq["H"][k]), (g, lq["g"][k]), (CDe, lq["CDe"][k])):
            assert np.abs(a - b).max() <= 1e-11 * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("gait,n", [("stance", 3), ("walk", 10), ("run", 14)])
def test_limb_lane_form_of_the_lq_approximation_equals_the_phase_form(model, emu, gait, n):
    """hsqp_lql.h (a lane per limb: forward walk, unwinding walk back with the composites in registers, rows by the column owners, stored transposed;
    then the column-lane RK4 chain) against lq_node<true> (one workgroup per node, ~70 phases over an LDS workspace, row-major rows): the records
    expand to the same LQ blocks — the two forms share formulas, not code paths or summation orders — far inside the 1e-11 either holds against the
    oracle.  `run` has active collision rows (64 row slots in use); node 1 is also taken as an event interval (dt = 0)."""
    lib, h = emu
    assert lib.emu_ql_ok(h) == 1
    x0, x, u, par, dt = perturbed_problem(model, n, gait, seed=5)
    RS, off, fo = lib.emu_rec_size(), lib.emu_rec_misc_offset(), lib.emu_rec_flow_offset()

    def expand(rec, dtk):
        AB, H, g, CDe = np.zeros((58, 93)), np.zeros((93, 93)), np.zeros(93), np.zeros((14, 94))
        lib.emu_expand(P(rec), C.c_double(dtk), P(AB), P(H), P(g), P(CDe))
        return AB, H, g, CDe

    nrows = set()
    try:
        for k in range(n):
            for dtk in (dt, 0.0) if k == 1 else (dt,):
                recs = []
                for limb in (1, 0):
                    lib.emu_set_lq_limb(limb)
                    rec = np.zeros(RS)
                    lib.emu_lq_node(h, P(x[k]), P(u[k]), P(x[k + 1]), P(par[k]), C.c_double(dtk), 1, P(rec))
                    recs.append(rec)
                for a, b in zip(expand(recs[0], dtk), expand(recs[1], dtk)):
                    assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max())
                ma, mb = recs[0][off:off + 8], recs[1][off:off + 8]          # ne, cost, eq / dyn SSE, contact flags, row offsets
                assert np.abs(ma - mb).max() <= 1e-12 * max(1.0, np.abs(mb).max())
                assert np.abs(recs[0][fo:fo + 64] - recs[1][fo:fo + 64]).max() <= 1e-12 * max(1.0, np.abs(recs[1][fo:fo + 64]).max())
                assert recs[0][off + 9] == 1.0 and recs[1][off + 9] == 0.0      # REC_LAYOUT: transposed / row-major
                nrows.add((recs[0][off + 8], recs[1][off + 8]))
    finally:
        lib.emu_set_lq_limb(1)
    assert nrows <= {(48.0, 46.0), (48.0, 38.0), (48.0, 30.0), (64.0, 54.0), (64.0, 62.0), (64.0, 46.0)}
    if gait == "run":
        assert any(a == 64.0 for a, _ in nrows)


def test_limb_tables_of_the_quad_value_pass(model, emu):
    """build_dev_model's limbs (hsqp_host.h): the G1 tree has four root-to-leaf paths (two legs, waist + arm twice), the feet sit on different limbs,
    and every moving body is owned — counted in the sums over bodies — by exactly one limb (the waist bodies, walked by both arm lanes, by the first)."""
    lib, h = emu
    out = np.zeros(8 + _abi.NB - 1, dtype=np.int32)
    lib.emu_limbs(h, out.ctypes.data_as(C.POINTER(C.c_int)))
    assert out[0] == 4 and out[2] != out[3] and min(out[2], out[3]) >= 0
    assert sorted(out[4:8]) == [6, 6, 7, 7] and out[1] == 7
    assert np.array_equal(out[8:], np.ones(_abi.NB - 1, dtype=np.int32))


@pytest.mark.parametrize("gait,n", [("stance", 3), ("walk", 10), ("run", 14)])
def test_value_pass_on_a_quad_of_lanes_equals_the_phase_form(model, emu, gait, n):
    """hsqp_lqv.h (one lane per limb, four lanes per node) against lq_node<false> (one wave per node, phases over an LDS workspace): the same
    numbers with the sums over bodies / cost terms taken in another order."""
    lib, h = emu
    x0, x, u, par, dt = perturbed_problem(model, n, gait, seed=11)
    off = lib.emu_rec_misc_offset()
    for k in range(n):
        for dtk in (dt, 0.0) if k == 1 else (dt,):
            rec, misc = np.zeros(lib.emu_rec_size()), np.zeros(8)
            lib.emu_lq_node(h, P(x[k]), P(u[k]), P(x[k + 1]), P(par[k]), C.c_double(dtk), 0, P(rec))
            assert lib.emu_value_quad(h, P(x[k]), P(u[k]), P(x[k + 1]), P(par[k]), C.c_double(dtk), P(misc)) == 4
            want = rec[off:off + 8]
            assert np.array_equal(misc[[0, 4, 5, 6, 7]], want[[0, 4, 5, 6, 7]])
            assert np.allclose(misc[1:4], want[1:4], rtol=1e-12, atol=1e-13 * max(1.0, np.abs(want[1:4]).max())), (k, misc, want)


@pytest.mark.parametrize("gait,n", [("stance", 4), ("walk", 8), ("run", 14)])
def test_full_iteration_matches_oracle(model, oracle, emu, gait, n):
    lib, h = emu
    x0, x, u, par, dt = perturbed_problem(model, n, gait, seed=5)
    r = oracle.sqp_iteration(dt, x0, x, u, par, want_proj=True)
    xn, un, dx, du = np.zeros_like(x), np.zeros_like(u), np.zeros_like(x), np.zeros_like(u)
    kkt, pb, pa = np.zeros(2), np.zeros(3), np.zeros(3)
    qp = np.zeros((n, lib.emu_qp_size()))
    rc = lib.emu_sqp_iteration(h, n, C.c_double(dt), P(x0), P(x), P(u), P(par), P(xn), P(un), P(dx), P(du), P(kkt), P(pb), P(pa), P(qp), None)
    assert rc == 0
    sc = max(1.0, np.abs(r["dx"]).max(), np.abs(r["du"]).max())
    assert np.abs(dx - r["dx"]).max() <= 1e-9 * sc and np.abs(du - r["du"]).max() <= 1e-9 * sc
    assert kkt[0] <= 1e-9 * sc and kkt[1] <= 1e-10 * sc
    for got, want in ((pb, r["perf_before"]), (pa, r["perf_after"])):
        assert np.allclose(got, [want["cost"], want["dynamics_sse"], want["equality_sse"]], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("gait,n", [("stance", 4), ("walk", 8), ("run", 14)])
def test_phases_are_free_of_intra_phase_dependencies(model, emu, gait, n):
    """Race check: the same kernel sources built with every phase executing its work items in REVERSE order
    (-DHSQP_EMU_REVERSE) must give bit-identical results — the items of a barrier-separated phase may not depend on each
    other, whatever order the hardware runs them in."""
    lib, h = emu
    rev = C.CDLL(os.path.join(HERE, "hostemu", "libhsqp_hostemu_rev.so"))
    rev.emu_create.restype = C.c_void_p
    err = C.create_string_buffer(256)
    hr = C.c_void_p(rev.emu_create(C.byref(model.desc), err, 256))
    x0, x, u, par, dt = perturbed_problem(model, n, gait, seed=5)
    res = []
    for L, hh in ((lib, h), (rev, hr)):
        xn, un, dx, du = np.zeros_like(x), np.zeros_like(u), np.zeros_like(x), np.zeros_like(u)
        kkt, pb, pa = np.zeros(2), np.zeros(3), np.zeros(3)
        qp = np.zeros((n, L.emu_qp_size()))
        assert L.emu_sqp_iteration(hh, n, C.c_double(dt), P(x0), P(x), P(u), P(par), P(xn), P(un), P(dx), P(du), P(kkt), P(pb), P(pa), P(qp), None) == 0
        res.append((dx, du, qp, pb, pa))
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("gait,n", [("stance", 5), ("walk", 24), ("run", 40)])
def test_factored_serial_sweep_equals_the_dense_stage(model, emu, gait, n):
    """hsqp_riccati_fact.h (the whole-body serial sweep on the factors [A~ | B~] = [E_J | 0] + F [Vx | Vu]: 35-deep contractions with row
    combinations formed while the operands are fetched, S A~ and W under the elimination, roll-out on the factors — what k_riccati_fact
    runs) against the dense stage of hsqp_riccati.h on the same kernel sources: same minimiser to rounding, same KKT residual.  The
    factored run also proves that the joint rows of A~ / B~ are never read: the emulation leaves them NaN until its KKT check."""
    lib, h = emu
    x0, x, u, par, dt = perturbed_problem(model, n, gait, seed=11)
    res = []
    for fact in (0, 1):
        lib.emu_set_ric_fact(fact)
        xn, un, dx, du = np.zeros_like(x), np.zeros_like(u), np.zeros_like(x), np.zeros_like(u)
        kkt, pb, pa = np.zeros(2), np.zeros(3), np.zeros(3)
        rc = lib.emu_sqp_iteration(h, n, C.c_double(dt), P(x0), P(x), P(u), P(par), P(xn), P(un), P(dx), P(du), P(kkt), P(pb), P(pa), None, None)
        lib.emu_set_ric_fact(1)
        assert rc == 0
        res.append((dx, du, kkt, pa))
    (dx0, du0, kkt0, pa0), (dx1, du1, kkt1, pa1) = res
    sc = max(1.0, np.abs(dx0).max(), np.abs(du0).max())
    assert np.isfinite(dx1).all() and np.isfinite(du1).all()
    assert max(np.abs(dx1 - dx0).max(), np.abs(du1 - du0).max()) <= 2e-11 * sc, (np.abs(dx1 - dx0).max(), np.abs(du1 - du0).max(), sc)
    assert kkt1[0] <= max(3.0 * kkt0[0], 1e-10 * sc) and kkt1[1] <= max(3.0 * kkt0[1], 1e-12 * sc), (kkt0, kkt1)
    assert np.allclose(pa1, pa0, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("gait,n,accurate", [("walk", 16, True), ("walk", 37, True), ("run", 37, False)])
def test_whole_body_parallel_scan_backward_sweep_equals_the_serial_recursion(model, emu, gait, n, accurate):
    """hsqp_scan.h at n = 58 (elements of all stages, ceil(log2(N+1)) levels of combinations, single-stage gains, closed-loop roll-out)
    against the serial recursion of the same kernel sources.  The scan inverts I + C1 J2 of partial horizons, cond up to 1e9 on the
    whole-body problem: on the walk QPs it agrees to ~1e-10 of the step's scale and its stationarity stays below the 2e-8 of the
    device's KKT gate (hsqp_capi.hip); on the randomly perturbed run-gait QP (|du| = 625) it loses digits (3e-8 of the scale) AND its
    stationarity exceeds the gate — the device would redo that iteration with the serial recursion (tests/test_gpu_parity.py)."""
    lib, h = emu
    x0, x, u, par, dt = perturbed_problem(model, n, gait, seed=5)
    res = []
    for scan in (0, 1):
        lib.emu_set_scan(scan)
        xn, un, dx, du = np.zeros_like(x), np.zeros_like(u), np.zeros_like(x), np.zeros_like(u)
        kkt, pb, pa = np.zeros(2), np.zeros(3), np.zeros(3)
        qp = np.zeros((n, lib.emu_qp_size()))
        rc = lib.emu_sqp_iteration(h, n, C.c_double(dt), P(x0), P(x), P(u), P(par), P(xn), P(un), P(dx), P(du), P(kkt), P(pb), P(pa), P(qp), None)
        lib.emu_set_scan(0)
        assert rc == 0
        res.append((dx, du, kkt, pa))
    (dx0, du0, kkt0, pa0), (dx1, du1, kkt1, pa1) = res
    sc = max(1.0, np.abs(dx0).max(), np.abs(du0).max())
    err = max(np.abs(dx1 - dx0).max(), np.abs(du1 - du0).max())
    lib.emu_scan_gate_accepts.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int]
    accepted = bool(lib.emu_scan_gate_accepts(kkt1[0], kkt1[1], 1e5, 0))   # |g|_inf of these QPs is 1e4 .. 1e5: the absolute bound decides
    if accurate:
        assert err <= 1e-9 * sc, (err, sc)
        assert accepted and kkt1[1] <= 1e-12 * sc
        assert np.allclose(pa1, pa0, rtol=1e-8, atol=1e-12)
    else:
        assert err <= 1e-6 * sc, (err, sc)          # still a usable SQP direction ...
        assert not accepted, kkt1                   # ... but flagged: the gate's criterion sees it


def test_scan_elements_keep_their_digits_on_a_long_horizon(model, emu):
    """Round 5: the scan's stage elements are formed with the Cholesky factor of R~ applied to both factors of every product (hsqp_scan.h::
    scan_init_node) instead of products with the explicit R~^-1 (cond 1e7).  On this perturbed walk QP over 100 stages the emulated scan was 6.5e-7 of
    the step's scale from the serial recursion (stationarity 6.7e-6) with the explicit inverse; it is 1.1e-9 (6e-8) now."""
    lib, h = emu
    n = 100
    x0, x, u, par, dt = perturbed_problem(model, n, "walk", seed=5)
    res = []
    for scan in (0, 1):
        lib.emu_set_scan(scan)
        xn, un, dx, du = np.zeros_like(x), np.zeros_like(u), np.zeros_like(x), np.zeros_like(u)
        kkt, pb, pa = np.zeros(2), np.zeros(3), np.zeros(3)
        qp = np.zeros((n, lib.emu_qp_size()))
        rc = lib.emu_sqp_iteration(h, n, C.c_double(dt), P(x0), P(x), P(u), P(par), P(xn), P(un), P(dx), P(du), P(kkt), P(pb), P(pa), P(qp), None)
        lib.emu_set_scan(0)
        assert rc == 0
        res.append((dx, du, kkt))
    sc = max(1.0, np.abs(res[0][0]).max(), np.abs(res[0][1]).max())
    err = max(np.abs(res[1][0] - res[0][0]).max(), np.abs(res[1][1] - res[0][1]).max())
    assert err <= 1e-8 * sc and res[1][2][0] <= 5e-7, (err / sc, res[1][2])


def test_scan_gate_decisions(emu):
    """scan_gate_accepts (hsqp_scan.h) on the measured populations: accurate scans pass, inaccurate ones, flagged ones and NaN do not."""
    lib, _ = emu
    lib.emu_scan_gate_accepts.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int]
    gate = lambda *a: bool(lib.emu_scan_gate_accepts(*a))  # noqa: E731
    assert gate(3.07e-11, 1.4e-15, 122.0, 0)            # config 2, cold start
    assert gate(1.07e-9, 7.2e-14, 38.1, 0)              # config 3, cold start
    assert gate(3.4e-9, 3.1e-14, 5e3, 0)                # perturbed walk instance
    assert not gate(7.8e-6, 2.1e-14, 2.71e4, 0)         # config 2, far-from-feasible line-search iterate (3e-10 of |g|_inf!)
    assert not gate(1.85e-4, 1.1e-13, 3.38e5, 0)        # config 3, the same
    assert not gate(1.6e-7, 4.4e-14, 1e5, 0)            # randomly perturbed run-gait QP
    assert not gate(2e-9, 1e-14, 0.1, 0)                # small gradient scale: BASELINE.md's 1e-9 max(1, |g|_inf) binds
    assert gate(5e-10, 1e-14, 0.1, 0)
    assert not gate(1e-12, 3e-8, 50.0, 0)               # primal residual
    assert not gate(1e-12, 1e-14, 50.0, 2)              # a bad pivot inside the scan
    assert not gate(float("nan"), 1e-14, 50.0, 0) and not gate(1e-12, 1e-14, float("nan"), 0)


@pytest.mark.parametrize("form,gait,n,segments", [("wb", "walk", 24, 3), ("wb", "run", 33, 7), ("wb", "walk", 60, 7), ("centroidal", "walk", 40, 7)])
def test_two_level_sweep_of_the_kernel_sources_equals_the_serial_recursion(model, cmodel, emu, form, gait, n, segments):
    """hsqp_segment.h through the host build, pass by pass as launch_segmented runs it (segment elements = Riccati recursion from J = 0 +
    prepended closed loops, suffix scan over the segment elements with the scan's combination, exact recursions per segment from the
    boundary value functions): the step of the serial recursion to 1e-10 of its scale, and the forward / reverse item orders agree bit for bit."""
    from test_oracle_centroidal_ocp import perturbed_centroidal_problem
    cent = form == "centroidal"
    m = cmodel if cent else model
    x0, x, u, par, dt = (perturbed_centroidal_problem if cent else perturbed_problem)(m, n, gait, seed=5)
    outs = {}
    for lib_name in ("libhsqp_hostemu.so", "libhsqp_hostemu_rev.so"):
        L = C.CDLL(os.path.join(HERE, "hostemu", lib_name))
        L.emu_create.restype = C.c_void_p
        err = C.create_string_buffer(256)
        hh = C.c_void_p(L.emu_create(C.byref(m.desc), err, 256))
        for P_ in ((0, segments) if lib_name.endswith("emu.so") else (segments,)):
            L.emu_set_segments(P_)
            xn, un, dx, du = np.zeros_like(x), np.zeros_like(u), np.zeros_like(x), np.zeros_like(u)
            kkt, pb, pa = np.zeros(2), np.zeros(3), np.zeros(3)
            assert L.emu_sqp_iteration(hh, n, C.c_double(dt), P(x0), P(x), P(u), P(par), P(xn), P(un), P(dx), P(du), P(kkt), P(pb), P(pa), None, None) == 0
            outs[(lib_name, P_)] = (dx.copy(), du.copy(), kkt.copy())
        L.emu_set_segments(0)
    ser, seg, rev = outs[("libhsqp_hostemu.so", 0)], outs[("libhsqp_hostemu.so", segments)], outs[("libhsqp_hostemu_rev.so", segments)]
    sc = max(1.0, np.abs(ser[0]).max(), np.abs(ser[1]).max())
    err = max(np.abs(seg[0] - ser[0]).max(), np.abs(seg[1] - ser[1]).max())
    # the device's gate: the KKT residual of the segments' last stages (= the maximum over all stages up to rounding, which is what the host
    # build evaluates) against the scan's bounds; an accepted sweep must be accurate, a rejected one is redone with the serial recursion
    L.emu_scan_gate_accepts.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int]
    accepted = L.emu_scan_gate_accepts(seg[2][0], seg[2][1], 1e9, 0) == 1
    print(f"{form} {gait} N={n} P={segments}: |step - serial| = {err:.2e} ({err / sc:.1e} of the scale), KKT {seg[2][0]:.1e} / {seg[2][1]:.1e} -> {'accepted' if accepted else 'rejected'}")
    # (The run-gait QP with |du| ~ 1e3 is the hard case: its segment elements equal the numpy prototype's to 1e-12, but the scan's Gauss-Jordan
    # combination of two of them — cond(I + C1 J2) ~ 1e9 — loses digits that numpy's LU keeps: 2.5e-9 of the scale instead of 2e-12.)
    if accepted:
        assert err <= 2e-10 * sc
    else:
        assert gait == "run"
    assert err <= 1e-8 * sc
    assert np.array_equal(seg[0], rev[0]) and np.array_equal(seg[1], rev[1])   # race check


@pytest.mark.parametrize("nxe", [58, 35])
def test_blocked_matrix_core_factorisation_on_the_wave_emulation(emu, nxe):
    """hsqp_elim.h (six panels of four rows in the accumulator layout of v_mfma_f64_16x16x4; what k_riccati runs on the device)
    executed lane by lane on the host's 64-lane wave emulation: L^-1, Z = L^-1 G, z = L^-1 g against numpy's Cholesky."""
    lib, _ = emu
    rng = np.random.default_rng(11 + nxe)
    for trial in range(6):
        Bm = rng.standard_normal((23, 40)) * 10.0 ** rng.uniform(-2, 2, size=(23, 1))   # badly scaled rows: cond up to ~1e8
        lam = Bm @ Bm.T + np.diag(10.0 ** rng.uniform(-5, 0, 23))
        lam = 0.5 * (lam + lam.T)
        G = rng.standard_normal((23, nxe)) * 10.0 ** rng.uniform(-1, 3)
        g = rng.standard_normal(23)
        linv, linvT, Z, z = np.zeros((23, 23)), np.zeros((23, 23)), np.zeros((23, nxe)), np.zeros(23)
        ok = lib.emu_eliminate_blocked(nxe, P(lam), P(G), P(g), P(linv), P(linvT), P(Z), P(z))
        assert ok == 1
        L = np.linalg.cholesky(lam)
        want_inv = np.linalg.inv(L)
        cond = np.linalg.cond(lam)
        tol = 1e-15 * cond ** 0.5 * 50
        assert np.array_equal(linv, linvT.T) and np.all(np.triu(linv, 1) == 0.0)
        assert np.abs(linv - want_inv).max() <= tol * np.abs(want_inv).max()
        assert np.abs(Z - want_inv @ G).max() <= tol * np.abs(want_inv @ G).max()
        assert np.abs(z - want_inv @ g).max() <= tol * np.abs(want_inv @ g).max()
        # what the stage uses them for: Z^T Z = G^T Lam^-1 G, K = -L^-T Z = -Lam^-1 G
        K = -linv.T @ Z
        assert np.abs(lam @ K + G).max() <= 1e-13 * cond ** 0.5 * np.abs(G).max()
    # an indefinite matrix is reported, not silently factorised
    lam = np.eye(23); lam[7, 7] = -1.0
    assert lib.emu_eliminate_blocked(nxe, P(lam), P(G), P(g), P(linv), P(linvT), P(Z), P(z)) == 0
