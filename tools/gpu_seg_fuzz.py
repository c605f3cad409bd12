#!/usr/bin/env python3
"""Two-level sweep on perturbed config-4 batches of 32 (several seeds), ungated view: per-instance KKT residual of the QP and distance of the step
from the serial recursion's — the populations the gate has to separate (GPU box)."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
from wb_humanoid_mpc_amd import load_model
from wb_humanoid_mpc_amd.reference import make_problem
from wb_humanoid_mpc_amd.solver import HipSqpSolver
m = load_model()
B, N = 32, 100
for seed in (77, 1, 2, 3, 20250808):
    x0, x, u, par, dt = make_problem(m, n_nodes=N, batch=B, perturb=True, seed=seed)
    outs = {}
    for md in ("serial", "segmented"):
        s = HipSqpSolver(m, max_nodes=N, max_batch=B, riccati=md)
        s.upload(x0, x, u, par, dt); s.iterate(1, kkt=True); outs[md] = s.download(); outs[md]["fb"] = s.scan_fallbacks(); s.close()
    a, b = outs["serial"], outs["segmented"]
    sc = np.maximum(1.0, np.maximum(np.abs(a["dx"]).max((1, 2)), np.abs(a["du"]).max((1, 2))))
    err = np.maximum(np.abs(a["dx"] - b["dx"]).max((1, 2)), np.abs(a["du"] - b["du"]).max((1, 2)))
    k = b["kkt"].max(1)
    print(f"seed {seed}: fallbacks {b['fb']}  KKT max {k.max():.2e} median {np.median(k):.2e}  |step - serial| max {err.max():.2e} ({(err / sc).max():.1e} of scale)  worst instances:",
          [(int(i), f"{k[i]:.1e}", f"{err[i]:.1e}") for i in np.argsort(-k)[:3]])
