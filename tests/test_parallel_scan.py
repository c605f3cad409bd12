"""Groundwork for the parallel-in-time Riccati (oracle/parallel_scan.py): the associative-scan formulation reproduces the serial
Riccati solution of the projected stage QPs that the kernel sources produce, for both formulations, in ceil(log2(N+1)) levels."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

import parallel_scan
from test_oracle_centroidal_ocp import perturbed_centroidal_problem
from test_oracle_lq import perturbed_problem
from wb_humanoid_mpc_amd import _abi

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
P = lambda a: a.ctypes.data_as(_dp)  # noqa: E731
NX, NU, NUT = _abi.NX, _abi.NU, 23
# QP record layout (wb_humanoid_mpc_amd/csrc/hsqp_project.h)
QP_A = 0
QP_B = QP_A + NX * NX
QP_BV = QP_B + NX * NUT
QP_Q = QP_BV + NX
QP_P = QP_Q + NX * NX
QP_R = QP_P + NUT * NX
QP_QV = QP_R + NUT * NUT
QP_RV = QP_QV + NX
QP_PX = QP_RV + NUT
QP_PU = QP_PX + NU * NX
QP_PE = QP_PU + NU * NUT


def stages_from_records(qp, nxe):
    out = []
    for rec in qp:
        m = lambda o, r, c: rec[o:o + r * c].reshape(r, c)  # noqa: E731
        out.append(dict(A=m(QP_A, NX, NX)[:nxe, :nxe], B=m(QP_B, NX, NUT)[:nxe], b=rec[QP_BV:QP_BV + nxe], Q=m(QP_Q, NX, NX)[:nxe, :nxe],
                        P=m(QP_P, NUT, NX)[:, :nxe], R=m(QP_R, NUT, NUT), q=rec[QP_QV:QP_QV + nxe], r=rec[QP_RV:QP_RV + NUT]))
    return out


# Accuracy found: the combination solves with M = I + C1 J2, whose condition number reaches ~1e9 on the whole-body problem (input
# weights 1e-3 dt against orientation / barrier Hessians 1e4..1e6) and ~1e5 on the centroidal one; the scan then agrees with the
# serial recursion to ~5e-8 (states) and ~1.4e-6 (inputs) resp. ~1e-11 of the step's scale.  A device version for the whole-body problem needs a remedy
# (scaling of the state coordinates or one step of iterative refinement on the serial recursion's residual).
TOL = {"wb": 1e-5, "centroidal": 1e-8}


@pytest.mark.parametrize("form,gait,n", [("wb", "walk", 20), ("wb", "run", 33), ("wb", "walk", 100), ("centroidal", "walk", 20),
                                         ("centroidal", "run", 64), ("centroidal", "walk", 100)])
def test_scan_reproduces_the_serial_riccati_solution(model, cmodel, form, gait, n):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "hostemu"), "all"])
    lib = C.CDLL(os.path.join(HERE, "hostemu", "libhsqp_hostemu.so"))
    lib.emu_create.restype = C.c_void_p
    cent = form == "centroidal"
    m = cmodel if cent else model
    err = C.create_string_buffer(256)
    h = C.c_void_p(lib.emu_create(C.byref(m.desc), err, 256))
    assert h.value, err.value
    x0, x, u, par, dt = (perturbed_centroidal_problem if cent else perturbed_problem)(m, n, gait, seed=5)
    xn, un, dx, du = np.zeros_like(x), np.zeros_like(u), np.zeros_like(x), np.zeros_like(u)
    kkt, pb, pa = np.zeros(2), np.zeros(3), np.zeros(3)
    qp = np.zeros((n, lib.emu_qp_size()))
    assert lib.emu_sqp_iteration(h, n, C.c_double(dt), P(x0), P(x), P(u), P(par), P(xn), P(un), P(dx), P(du), P(kkt), P(pb), P(pa), P(qp), None) == 0
    nxe = _abi.CNX if cent else NX
    Qf = np.array(m.raw["Qf"])
    qN = Qf * (x[n, :nxe] - par[n, :nxe])
    sdx, sut, S, s, levels = parallel_scan.solve_qp(stages_from_records(qp, nxe), np.diag(Qf), qN, (x0 - x[0])[:nxe])
    assert levels == math.ceil(math.log2(n + 1))
    sc = max(1.0, np.abs(dx).max(), np.abs(du).max())
    tol = TOL[form]
    assert np.abs(sdx - dx[:, :nxe]).max() <= tol * sc
    # the stage elements as the device forms them since round 5 (Cholesky factor of R applied to both factors of every product instead of
    # products with the explicit inverse): where the explicit inverse loses digits (whole-body walk: 6e-8 / 8e-9 of the scale) this form keeps
    # them (2e-11 / 7e-10), elsewhere the two agree
    pdx, put, _, _, _ = parallel_scan.solve_qp(stages_from_records(qp, nxe), np.diag(Qf), qN, (x0 - x[0])[:nxe], prepended=True)
    e_inv, e_pre = np.abs(sdx - dx[:, :nxe]).max() / sc, np.abs(pdx - dx[:, :nxe]).max() / sc
    print(f"{form} {gait} N={n}: scan vs serial, explicit inverse {e_inv:.1e}, prepended form {e_pre:.1e}")
    assert e_pre <= (2e-9 if form == "wb" else 1e-9)
    if form == "wb" and gait == "walk":
        assert e_pre <= 0.1 * e_inv
    # inputs: du = Px dx + Pu ut + Pe with the scan's ut
    for k in range(n):
        rec = qp[k]
        Px = rec[QP_PX:QP_PX + NU * NX].reshape(NU, NX)[:, :nxe]
        Pu = rec[QP_PU:QP_PU + NU * NUT].reshape(NU, NUT)
        assert np.abs(Px @ sdx[k] + Pu @ sut[k] + rec[QP_PE:QP_PE + NU] - du[k]).max() <= tol * sc
    # the value functions of the scan are symmetric positive semi-definite like the Riccati ones
    assert all(np.linalg.eigvalsh(Sk).min() >= -1e-9 * max(1.0, np.abs(Sk).max()) for Sk in S)
    lib.emu_destroy(h)


@pytest.mark.parametrize("form,gait,n,segments", [("wb", "walk", 40, 4), ("wb", "run", 33, 8), ("wb", "walk", 100, 8), ("centroidal", "walk", 100, 5)])
def test_segmented_sweep_reproduces_the_serial_riccati_solution(model, cmodel, form, gait, n, segments):
    """The two-level sweep of DESIGN.md §6 (segment elements by prepending stages — rank-nu Woodbury = the Riccati gain solve —, a suffix
    scan over the segment elements, ordinary recursions per segment from the boundary value functions) against the serial recursion."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "hostemu"), "all"])
    lib = C.CDLL(os.path.join(HERE, "hostemu", "libhsqp_hostemu.so"))
    lib.emu_create.restype = C.c_void_p
    cent = form == "centroidal"
    m = cmodel if cent else model
    err = C.create_string_buffer(256)
    h = C.c_void_p(lib.emu_create(C.byref(m.desc), err, 256))
    x0, x, u, par, dt = (perturbed_centroidal_problem if cent else perturbed_problem)(m, n, gait, seed=5)
    xn, un, dx, du = np.zeros_like(x), np.zeros_like(u), np.zeros_like(x), np.zeros_like(u)
    kkt, pb, pa = np.zeros(2), np.zeros(3), np.zeros(3)
    qp = np.zeros((n, lib.emu_qp_size()))
    assert lib.emu_sqp_iteration(h, n, C.c_double(dt), P(x0), P(x), P(u), P(par), P(xn), P(un), P(dx), P(du), P(kkt), P(pb), P(pa), P(qp), None) == 0
    nxe = _abi.CNX if cent else NX
    Qf = np.array(m.raw["Qf"])
    stages = stages_from_records(qp, nxe)
    qN = Qf * (x[n, :nxe] - par[n, :nxe])
    sdx, sut, bounds, vf, levels = parallel_scan.solve_qp_segmented(stages, np.diag(Qf), qN, (x0 - x[0])[:nxe], segments)
    assert levels == math.ceil(math.log2(segments + 1)) and bounds[0] == 0 and bounds[-1] == n
    sc = max(1.0, np.abs(dx).max(), np.abs(du).max())
    err_x = np.abs(sdx - dx[:, :nxe]).max() / sc
    # the prepend form equals the generic combination of the stage's element with the accumulated one
    e2 = parallel_scan.identity_element(nxe)
    for k in (n - 1, n - 2, n - 3):
        a = parallel_scan.prepend_stage(stages[k], e2)
        b = parallel_scan.combine(parallel_scan.stage_element(**stages[k]), e2)
        for u_, v_ in zip(a, b):
            assert np.abs(u_ - v_).max() <= 1e-9 * max(1.0, np.abs(v_).max())
        e2 = a
    print(f"{form} {gait} N={n} P={segments}: |dx - serial| / scale = {err_x:.2e}")
    assert err_x <= TOL[form]
    # guess shift (oracle/parallel_scan.py::shifted_segment_element): any symmetric guesses of the boundary value functions leave the result
    # unchanged in exact arithmetic — here: the boundary value functions of the sweep above, 5 % off
    rng = np.random.default_rng(3)
    guesses = []
    for Sb, _ in vf:
        F = np.eye(nxe) + 0.05 * rng.standard_normal((nxe, nxe)) / np.sqrt(nxe)
        guesses.append(F @ Sb @ F.T)
    gdx, gut, _, _, _ = parallel_scan.solve_qp_segmented(stages, np.diag(Qf), qN, (x0 - x[0])[:nxe], segments, guesses)
    assert np.abs(gdx - dx[:, :nxe]).max() / sc <= 10 * TOL[form] * 1e-3 + 1e-9


def test_device_stage_elements_equal_the_prepended_form(model):
    """hsqp_scan.h::scan_init_node (the kernel source, host build: the blocked elimination of [R~ | I | P~ | r~] on the emulated wave, then the
    Gram-type products) against the numpy prototype's element of the same stage — the stage prepended to the empty interval, Cholesky factor on
    both sides of every product — entry by entry, on the QP records of a perturbed walk; and C, J symmetric to the bit."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "hostemu"), "all"])
    lib = C.CDLL(os.path.join(HERE, "hostemu", "libhsqp_hostemu.so"))
    lib.emu_create.restype = C.c_void_p
    err = C.create_string_buffer(256)
    h = C.c_void_p(lib.emu_create(C.byref(model.desc), err, 256))
    n = 12
    x0, x, u, par, dt = perturbed_problem(model, n, "walk", seed=5)
    xn, un, dx, du = np.zeros_like(x), np.zeros_like(u), np.zeros_like(x), np.zeros_like(u)
    kkt, pb, pa = np.zeros(2), np.zeros(3), np.zeros(3)
    qp = np.zeros((n, lib.emu_qp_size()))
    assert lib.emu_sqp_iteration(h, n, C.c_double(dt), P(x0), P(x), P(u), P(par), P(xn), P(un), P(dx), P(du), P(kkt), P(pb), P(pa), P(qp), None) == 0
    stages = stages_from_records(qp, NX)
    SZ = lib.emu_el_size58()
    oA, oC, oJ = 0, NX * NX, 2 * NX * NX
    oB, oE = 3 * NX * NX, 3 * NX * NX + NX
    for k in (0, 5, n - 1):
        el = np.zeros(SZ)
        assert lib.emu_scan_stage_element58(h, P(np.ascontiguousarray(qp[k])), P(el)) == 1
        A, b, Cm, eta, J = parallel_scan.prepend_stage(stages[k], parallel_scan.identity_element(NX))
        for name, got, want in (("A", el[oA:oA + NX * NX].reshape(NX, NX), A), ("C", el[oC:oC + NX * NX].reshape(NX, NX), Cm), ("J", el[oJ:oJ + NX * NX].reshape(NX, NX), J),
                                ("b", el[oB:oB + NX], b), ("eta", el[oE:oE + NX], eta)):
            assert np.abs(got - want).max() <= 1e-10 * max(1.0, np.abs(want).max()), (k, name, np.abs(got - want).max(), np.abs(want).max())
        Cd, Jd = el[oC:oC + NX * NX].reshape(NX, NX), el[oJ:oJ + NX * NX].reshape(NX, NX)
        assert np.array_equal(Cd, Cd.T) and np.array_equal(Jd, Jd.T)
    lib.emu_destroy(h)
