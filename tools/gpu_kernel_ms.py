#!/usr/bin/env python3
"""Kernel times of one SQP iteration of BASELINE config 4 through the library named by HSQP_LIB — for timing-experiment builds whose results are WRONG
(the solve's error code is ignored; hsqp_last_kernel_ms still holds the HIP-event times of the launches that ran).  Usage: gpurun -- 'HSQP_LIB=... python tools/gpu_kernel_ms.py [B] [N]'"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wb_humanoid_mpc_amd import load_model
from wb_humanoid_mpc_amd.reference import make_problem
from wb_humanoid_mpc_amd.solver import HipSqpSolver, HsqpError

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
m = load_model()
x0, x, u, par, dt = make_problem(m, n_nodes=N, batch=B, perturb=True)
s = HipSqpSolver(m, max_nodes=N, max_batch=B, riccati="serial")
s.upload(x0, x, u, par, dt)
ms = []
for it in range(8):
    try:
        s.iterate(1, take_step=False, kkt=False)
    except HsqpError as e:
        if it == 0:
            print("(solver error ignored:", str(e)[:80], ")")
    ms.append(s.kernel_ms())
ms = ms[3:]
print({k: round(float(np.mean([d[k] for d in ms])), 4) for k in ms[0]})
