#!/usr/bin/env python3
"""Debug tool (GPU box): gains K, k and the roll-out dx of the factored serial sweep (k_riccati_fact) against the dense stage
(HSQP_RICCATI_DENSE=1) on the same small whole-body problem, stage by stage."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from wb_humanoid_mpc_amd import load_model  # noqa: E402
from wb_humanoid_mpc_amd.reference import make_problem  # noqa: E402
from wb_humanoid_mpc_amd.solver import HipSqpSolver  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
m = load_model()
x0, x, u, par, dt = make_problem(m, n_nodes=N, batch=B, perturb=True)
NX, NUT = 58, 23
RIC = ((NUT * NX + NUT + 7) // 8) * 8
res = {}
for name, env in (("dense", "1"), ("fact", None)):
    if env: os.environ["HSQP_RICCATI_DENSE"] = env
    else: os.environ.pop("HSQP_RICCATI_DENSE", None)
    s = HipSqpSolver(m, max_nodes=N, max_batch=B, riccati="serial")
    s.upload(x0, x, u, par, dt)
    s.iterate(1, kkt=(len(sys.argv) > 3))
    ric = np.zeros((B, N, RIC))
    s.lib.hsqp_debug_read(s.h, 101, ric.ctypes.data_as(C.c_void_p), ric.nbytes)
    dx = np.zeros((B, N + 1, NX))
    s.lib.hsqp_debug_read(s.h, 8, dx.ctypes.data_as(C.c_void_p), dx.nbytes)
    res[name] = (ric.copy(), dx.copy())
    s.close()
rd, dd = res["dense"]
rf, df = res["fact"]
for b in range(min(B, 2)):
    for k in range(N - 1, -1, -1):
        Kd, Kf = rd[b, k, :NUT * NX].reshape(NUT, NX), rf[b, k, :NUT * NX].reshape(NUT, NX)
        kd, kf = rd[b, k, NUT * NX:NUT * NX + NUT], rf[b, k, NUT * NX:NUT * NX + NUT]
        e = np.abs(Kd - Kf)
        i, j = np.unravel_index(np.nanargmax(e), e.shape)
        print(f"b {b} stage {k}: |K| {np.abs(Kd).max():.3e}  K err {np.nanmax(e):.3e} at ({i},{j}) nan {np.isnan(Kf).sum()}  k err {np.abs(kd - kf).max():.3e} of {np.abs(kd).max():.3e}")
        if np.nanmax(e) > 1e-6 * np.abs(Kd).max():
            bad_cols = np.flatnonzero(e.max(axis=0) > 1e-6 * np.abs(Kd).max())
            bad_rows = np.flatnonzero(e.max(axis=1) > 1e-6 * np.abs(Kd).max())
            print("   bad cols", bad_cols.tolist(), "bad rows", bad_rows.tolist())
    print(f"b {b} dx err per node", [float(f"{v:.2e}") for v in np.abs(dd[b] - df[b]).max(axis=1)], "of", float(np.abs(dd[b]).max()))
