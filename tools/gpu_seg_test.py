import sys, numpy as np
sys.path.insert(0,'/root/repo')
from wb_humanoid_mpc_amd import load_model
from wb_humanoid_mpc_amd.reference import make_problem, make_centroidal_problem
from wb_humanoid_mpc_amd.solver import HipSqpSolver
for form, B, N, mode in (("wb", 2, 40, "segmented"), ("wb", 32, 100, "auto"), ("wb", 64, 100, "auto"), ("wb", 16, 100, "auto"), ("centroidal", 8, 100, "auto")):
    m = load_model(formulation=form)
    mk = make_centroidal_problem if form == "centroidal" else make_problem
    x0, x, u, par, dt = mk(m, n_nodes=N, batch=B, perturb=True)
    out = {}
    for md in ("serial", mode):
        s = HipSqpSolver(m, max_nodes=N, max_batch=B, riccati=md)
        s.upload(x0, x, u, par, dt)
        for _ in range(3): s.iterate(1, kkt=False)
        s.iterate(1, kkt=True)
        o = s.download(); o["ms"] = s.kernel_ms(); o["fb"] = s.scan_fallbacks()
        s.iterate(1); o["ms_nokkt"] = s.kernel_ms()
        out[md] = o; s.close()
    a, b = out["serial"], out[mode]
    sc = max(1.0, np.abs(a["dx"]).max(), np.abs(a["du"]).max())
    print(form, "B", B, "N", N, mode, "err/scale", max(np.abs(a["dx"]-b["dx"]).max(), np.abs(a["du"]-b["du"]).max())/sc, "kkt", a["kkt"].max(), b["kkt"].max(),
          "fallbacks", b["fb"], "riccati ms serial", round(a["ms_nokkt"]["riccati"],3), mode, round(b["ms_nokkt"]["riccati"],3), "total", round(a["ms_nokkt"]["total"],3), round(b["ms_nokkt"]["total"],3))
