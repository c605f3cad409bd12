"""Measured parity errors of the HIP path against the CPU oracle, per BASELINE config (GPU box; writes JSON).

For every case: max-abs error of dx / du / x+dx / u+du (absolute, and relative to the step's scale), relative error of the
performance-index terms, the KKT residuals with the gradient scale they are judged against (BASELINE.md §6:
r <= 1e-9 * max(1, |g|_inf)).  The tolerances asserted in tests/test_gpu_*.py are declared from this table.

    python tools/parity_report.py [--out gpurun_out/parity_report.json] [--quick]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

from hsqp_oracle import Oracle  # noqa: E402  (the checker)
from wb_humanoid_mpc_amd import _abi, load_model  # noqa: E402
from wb_humanoid_mpc_amd.reference import make_centroidal_problem, make_problem  # noqa: E402
from wb_humanoid_mpc_amd.solver import HipSqpSolver  # noqa: E402


def case(name, model, oracle, problem, instances, threads, cent=False, riccati="auto"):
    x0, x, u, par, dt = problem
    B, N = u.shape[0], u.shape[1]
    s = HipSqpSolver(model, max_nodes=N, max_batch=B, riccati=riccati)
    try:
        out = s.run(x0, x, u, par, dt)
        g = s.debug_read(_abi.BLK_G)
    finally:
        s.close()
    rows = []
    for b in instances:
        t = time.time()
        r = (oracle.cent_sqp_iteration if cent else oracle.sqp_iteration)(dt, x0[b], x[b], u[b], par[b], threads=threads)
        sc = max(1.0, np.abs(r["dx"]).max(), np.abs(r["du"]).max())
        perf = {}
        for which in ("perf_before", "perf_after"):
            for key in ("cost", "dynamics_sse", "equality_sse"):
                a, w = out[which][b][key], r[which][key]
                perf[f"{which}.{key}"] = abs(a - w) / max(abs(w), 1e-300)
        gs = max(1.0, float(np.abs(g[b]).max()))
        rows.append(dict(instance=int(b), dx_scale=float(np.abs(r["dx"]).max()), du_scale=float(np.abs(r["du"]).max()),
                         dx_abs=float(np.abs(out["dx"][b] - r["dx"]).max()), du_abs=float(np.abs(out["du"][b] - r["du"]).max()),
                         x_abs=float(np.abs(out["x"][b] - r["x"]).max()), u_abs=float(np.abs(out["u"][b] - r["u"]).max()),
                         step_rel=float(max(np.abs(out["dx"][b] - r["dx"]).max(), np.abs(out["du"][b] - r["du"]).max()) / sc),
                         perf_rel=perf, perf_rel_max=float(max(perf.values())),
                         kkt_stat=float(out["kkt"][b, 0]), kkt_prim=float(out["kkt"][b, 1]), g_inf=gs,
                         kkt_stat_normalised=float(out["kkt"][b, 0] / gs), oracle_kkt=[float(v) for v in r["kkt"]],
                         oracle_seconds=round(time.time() - t, 2)))
    worst = dict(dx_abs=max(r["dx_abs"] for r in rows), du_abs=max(r["du_abs"] for r in rows), step_rel=max(r["step_rel"] for r in rows),
                 perf_rel=max(r["perf_rel_max"] for r in rows), kkt_stat_normalised=max(r["kkt_stat_normalised"] for r in rows),
                 kkt_prim=max(r["kkt_prim"] for r in rows))
    print(f"{name:44s} B={B:4d} N={N:3d}  dx {worst['dx_abs']:.2e}  du {worst['du_abs']:.2e}  rel {worst['step_rel']:.2e}  perf {worst['perf_rel']:.2e}  "
          f"kkt/|g| {worst['kkt_stat_normalised']:.2e}  prim {worst['kkt_prim']:.2e}", flush=True)
    return dict(case=name, batch=B, n_nodes=N, worst=worst, instances=rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_report.json"))
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    threads = os.cpu_count() or 4
    wb = load_model()
    cm = load_model(formulation="centroidal")
    o_wb, o_c = Oracle(wb), Oracle(cm)
    res = []
    res.append(case("config 1: centroidal N=20 stance", cm, o_c, make_centroidal_problem(cm, n_nodes=20, batch=1, gait="stance", v_cmd=(0, 0, 0.7925, 0)), [0], threads, cent=True))
    res.append(case("config 2: centroidal N=100 walk (scan)", cm, o_c, make_centroidal_problem(cm, n_nodes=100, batch=1, gait="walk"), [0], threads, cent=True))
    res.append(case("config 2: centroidal N=100 walk (serial)", cm, o_c, make_centroidal_problem(cm, n_nodes=100, batch=1, gait="walk"), [0], threads, cent=True, riccati="serial"))
    res.append(case("config 3: WB N=100 walk, one instance", wb, o_wb, make_problem(wb, n_nodes=100, batch=1, gait="walk"), [0], threads))
    res.append(case("WB N=16 walk, 6 perturbed instances", wb, o_wb, make_problem(wb, n_nodes=16, batch=6, perturb=True), range(6), threads))
    if not a.quick:
        res.append(case("config 4: WB N=100 walk, 256 perturbed", wb, o_wb, make_problem(wb, n_nodes=100, batch=256, perturb=True), [0, 37, 128, 255], threads))
        res.append(case("config 5 slice: WB N=200 slow_walk, 8 perturbed", wb, o_wb, make_problem(wb, n_nodes=200, batch=8, gait="slow_walk", perturb=True), range(8), threads))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(dict(tolerances_of_BASELINE_md_6=dict(trajectory_abs=1e-8, kkt_normalised=1e-9, perf_rel=1e-10), cases=res), f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
