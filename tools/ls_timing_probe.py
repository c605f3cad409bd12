#!/usr/bin/env python3
"""Wall time of line-search iterations (device-resident, one instance): what a trial costs (tools, not a test)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wb_humanoid_mpc_amd import load_model
from wb_humanoid_mpc_amd.reference import make_centroidal_problem, make_problem
from wb_humanoid_mpc_amd.solver import HipSqpSolver
for name, cent in (("config3", False), ("config2", True)):
    m = load_model(formulation="centroidal" if cent else "wb")
    x0, x, u, par, dt = (make_centroidal_problem if cent else make_problem)(m, n_nodes=100, batch=1, gait="walk")
    for ls in (False, True):
        s = HipSqpSolver(m, max_nodes=100, max_batch=1, linesearch=ls)
        s.upload(x0, x, u, par, dt)
        for rep in range(3):
            s.upload(x0, x, u, par, dt)
            t0 = time.perf_counter()
            s.iterate(1, take_step=True, linesearch=ls)
            t1 = time.perf_counter()
            out = s.download()
        print(f"{name} linesearch={ls}: first iteration {1e3 * (t1 - t0):.3f} ms, alpha {out['alpha'][0]}, kernels {s.kernel_ms()}")
        s.close()
