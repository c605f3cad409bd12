"""Batch-consistency probe (GPU box): every instance of a large batch must get bit-identical results to the same instance
solved alone, and the batch must be repeatable.  Uses only the oldest part of the C ABI so that it can be pointed at older
builds of the library (HSQP_LIB) when bisecting."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa: F401  (its HIP runtime first)
from wb_humanoid_mpc_amd import load_model, _abi
from wb_humanoid_mpc_amd.reference import make_problem, BENCH_SEED

lib = C.CDLL(os.environ.get("HSQP_LIB") or os.path.join(os.getcwd(), "wb_humanoid_mpc_amd", "libhsqp_hip.so"))
lib.hsqp_last_error.restype = C.c_char_p
dp = C.POINTER(C.c_double)
m = load_model()
N, gait, B = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
x0, x, u, par, dt = make_problem(m, n_nodes=N, batch=B, gait=gait, perturb=True, seed=BENCH_SEED)
st = _abi.Settings(max_nodes=N, max_batch=B, device=0, flags=0)
h = C.c_void_p()
assert lib.hsqp_create(C.byref(m.desc), C.byref(st), C.byref(h)) == 0


def solve(x0_, x_, u_, par_):
    x0_, x_, u_, par_ = (np.ascontiguousarray(a) for a in (x0_, x_, u_, par_))
    b = x_.shape[0]
    p = _abi.Problem(batch=b, n_nodes=N, dt=dt, x_init=x0_.ctypes.data_as(dp), x_traj=x_.ctypes.data_as(dp), u_traj=u_.ctypes.data_as(dp),
                     node_params=par_.ctypes.data_as(dp))
    dx, du = np.zeros((b, N + 1, _abi.NX)), np.zeros((b, N, _abi.NU))
    s = _abi.Solution(dx=dx.ctypes.data_as(dp), du=du.ctypes.data_as(dp))
    rc = lib.hsqp_solve(h, C.byref(p), C.byref(s))
    assert rc == 0, lib.hsqp_last_error(h)
    return dx, du


dx, du = solve(x0, x, u, par)
dx2, _ = solve(x0, x, u, par)
nbad, worst = 0, 0.0
for b in range(0, B, max(1, B // 64)):
    d1, _ = solve(x0[b:b + 1], x[b:b + 1], u[b:b + 1], par[b:b + 1])
    d = np.abs(d1[0] - dx[b]).max()
    worst = max(worst, d)
    nbad += d > 0
print(os.path.basename(os.environ.get("HSQP_LIB", "libhsqp_hip.so")), "N", N, "B", B, "repeatable:", np.array_equal(dx, dx2), "| instances differing from the solo solve:", nbad,
      "worst |ddx| %.3e" % worst)
