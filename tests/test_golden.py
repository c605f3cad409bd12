"""The committed golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the oracle) on the CPU:
the oracle still reproduces them (they pin the oracle between rounds), and the kernel sources compiled for the host reproduce
them without running the oracle (the same comparison the GPU tests make through the C ABI)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
_dp = C.POINTER(C.c_double)
P = lambda a: np.ascontiguousarray(a).ctypes.data_as(_dp)  # noqa: E731
WB = ["wb_stance_n4", "wb_walk_n8", "wb_run_n14"]
CENT = ["cent_stance_n4", "cent_walk_n8", "cent_run_n14"]


@pytest.mark.parametrize("name", WB + CENT)
def test_oracle_reproduces_the_fixture(model, oracle, cmodel, coracle, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    dt = float(g["dt"])
    if name.startswith("cent"):
        r = coracle.cent_sqp_iteration(dt, g["x_init"], g["x"], g["u"], g["par"])
        lq = coracle.cent_lq(dt, g["x"], g["u"], g["par"])
    else:
        r = oracle.sqp_iteration(dt, g["x_init"], g["x"], g["u"], g["par"])
        lq = oracle.lq(dt, g["x"], g["u"], g["par"])
    sc = max(1.0, np.abs(g["dx"]).max(), np.abs(g["du"]).max())
    assert np.abs(r["dx"] - g["dx"]).max() <= 1e-10 * sc and np.abs(r["du"] - g["du"]).max() <= 1e-10 * sc
    for got, want in ((r["perf_before"], g["perf_before"]), (r["perf_after"], g["perf_after"])):
        assert np.allclose([got["cost"], got["dynamics_sse"], got["equality_sse"]], want, rtol=1e-10, atol=1e-13)
    assert np.allclose(lq["b"], g["b"], atol=1e-13) and np.allclose(lq["g"], g["g"], rtol=1e-11, atol=1e-12)
    assert np.array_equal(lq["ne"], g["ne"])


@pytest.mark.parametrize("name", WB + CENT)
def test_kernel_sources_on_the_host_reproduce_the_fixture(model, cmodel, name):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "hostemu"), "all"])
    lib = C.CDLL(os.path.join(HERE, "hostemu", "libhsqp_hostemu.so"))
    lib.emu_create.restype = C.c_void_p
    err = C.create_string_buffer(256)
    m = cmodel if name.startswith("cent") else model
    h = C.c_void_p(lib.emu_create(C.byref(m.desc), err, 256))
    assert h.value, err.value
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    x, u = np.ascontiguousarray(g["x"]), np.ascontiguousarray(g["u"])
    n = u.shape[0]
    xn, un, dx, du = np.zeros_like(x), np.zeros_like(u), np.zeros_like(x), np.zeros_like(u)
    kkt, pb, pa = np.zeros(2), np.zeros(3), np.zeros(3)
    rc = lib.emu_sqp_iteration(h, n, C.c_double(float(g["dt"])), P(g["x_init"]), P(x), P(u), P(g["par"]), P(xn), P(un), P(dx), P(du), P(kkt), P(pb),
                               P(pa), None, None)
    assert rc == 0
    sc = max(1.0, np.abs(g["dx"]).max(), np.abs(g["du"]).max())
    assert np.abs(dx - g["dx"]).max() <= 1e-8 * sc and np.abs(du - g["du"]).max() <= 1e-8 * sc
    assert np.allclose(pb, g["perf_before"], rtol=1e-9, atol=1e-12) and np.allclose(pa, g["perf_after"], rtol=1e-9, atol=1e-12)
    lib.emu_destroy(h)
