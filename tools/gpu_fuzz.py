#!/usr/bin/env python3
"""Randomised GPU-vs-oracle sweep (tool, not a test): many seeds / gaits / horizon lengths of randomly perturbed problems, both formulations,
default solver path (incl. the gated parallel-in-time sweep where it applies).  Prints the worst errors relative to the step's scale."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hsqp_oracle
from test_oracle_lq import perturbed_problem
from test_oracle_centroidal_ocp import perturbed_centroidal_problem
from wb_humanoid_mpc_amd import load_model
from wb_humanoid_mpc_amd.solver import HipSqpSolver

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(2024)
worst = {}
t0 = time.time()
for form in ("wb", "centroidal"):
    m = load_model(formulation=form)
    o = hsqp_oracle.Oracle(m)
    cent = form == "centroidal"
    for seed in range(nseeds):
        gait = ["stance", "walk", "run", "slow_walk"][int(rng.integers(4))]
        n = int(rng.integers(6, 64))
        prob = (perturbed_centroidal_problem if cent else perturbed_problem)(m, n, gait, seed=int(seed))
        x0, x, u, par, dt = prob
        s = HipSqpSolver(m, max_nodes=n, max_batch=1)
        out = s.run(x0[None], x[None], u[None], par[None], dt)
        fb = s.scan_fallbacks()
        s.close()
        r = (o.cent_sqp_iteration if cent else o.sqp_iteration)(dt, x0, x, u, par, threads=16)
        sc = max(1.0, np.abs(r["dx"]).max(), np.abs(r["du"]).max())
        err = max(np.abs(out["dx"][0] - r["dx"]).max(), np.abs(out["du"][0] - r["du"]).max())
        perf = max(abs(out["perf_after"][0][k] - r["perf_after"][k]) / max(abs(r["perf_after"][k]), 1e-300) for k in ("cost",))
        key = (form, "scan" if (n >= 48 and fb == 0) else ("fallback" if fb else "serial"))
        w = worst.setdefault(key, [0, 0.0, 0.0, 0.0])
        w[0] += 1; w[1] = max(w[1], err / sc); w[2] = max(w[2], err); w[3] = max(w[3], perf)
        print(f"{form:10s} seed {seed:2d} {gait:9s} N={n:2d} |step| {sc:8.3g} err {err:.2e} ({err / sc:.1e} of scale) perf rel {perf:.1e} kkt {out['kkt'][0]} fallbacks {fb}", flush=True)
print("worst per path: (count, err/scale, abs err, perf rel)")
for k, v in worst.items(): print(k, v)
print(f"{time.time() - t0:.0f} s")
