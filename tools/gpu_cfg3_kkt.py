#!/usr/bin/env python3
"""config 3 through the default (scan) path: absolute KKT residuals and the two gradient yardsticks (projected QP / stage gradients)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wb_humanoid_mpc_amd import _abi, load_model
from wb_humanoid_mpc_amd.reference import make_problem
from wb_humanoid_mpc_amd.solver import HipSqpSolver
m = load_model()
x0, x, u, par, dt = make_problem(m, n_nodes=100, batch=1, gait="walk")
for mode in ("auto", "serial"):
    s = HipSqpSolver(m, max_nodes=100, max_batch=1, riccati=mode)
    out = s.run(x0, x, u, par, dt)
    g = s.debug_read(_abi.BLK_G)
    print(mode, "kkt", out["kkt"][0], "|g| projected", out["grad_inf"][0], "|g| stage", np.abs(g[0]).max(), "fallbacks", s.scan_fallbacks(),
          "-> stat / max(1, |g|_stage) =", out["kkt"][0][0] / max(1.0, np.abs(g[0]).max()))
    s.close()
