#!/usr/bin/env python3
"""Per-iteration error of the line-search iterations of configs 2 / 3 against the oracle (what tests/test_gpu_convergence.py asserts)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from wb_humanoid_mpc_amd import load_model
from wb_humanoid_mpc_amd.reference import make_centroidal_problem, make_problem
from wb_humanoid_mpc_amd.solver import HipSqpSolver
import hsqp_oracle
for name, cent, mode in (("config2", True, "auto"), ("config2-serial", True, "serial"), ("config3", False, "serial"), ("config3-parallel", False, "parallel")):
    m = load_model(formulation="centroidal" if cent else "wb")
    o = hsqp_oracle.Oracle(m)
    x0, x, u, par, dt = (make_centroidal_problem if cent else make_problem)(m, n_nodes=100, batch=1, gait="walk")
    s = HipSqpSolver(m, max_nodes=100, max_batch=1, linesearch=True, riccati=mode)
    xs, us, xo, uo = x, u, x[0], u[0]
    for it in range(3):
        out = s.run(x0, xs, us, par, dt)
        r = (o.cent_sqp_iteration if cent else o.sqp_iteration)(dt, x0[0], xo, uo, par[0], threads=16)
        ls = (o.cent_linesearch if cent else o.linesearch)(dt, xo, uo, r["dx"], r["du"], par[0], r["armijo"], threads=16)
        sc = max(np.abs(r["dx"]).max(), np.abs(r["du"]).max())
        print(f"{name} it {it}: alpha {out['alpha'][0]} / {ls['alpha']}  |step| {sc:.3g}  x err {np.abs(out['x'][0] - ls['x']).max():.2e}  u err {np.abs(out['u'][0] - ls['u']).max():.2e}  kkt {out['kkt'][0]} ginf {out['grad_inf'][0]:.3g} fallbacks {s.scan_fallbacks()}")
        xs, us, xo, uo = out["x"], out["u"], out["x"][0], out["u"][0]
    s.close()
