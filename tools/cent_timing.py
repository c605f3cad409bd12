#!/usr/bin/env python3
"""Kernel times of the centroidal formulation on the GPU (no torch needed): BASELINE config 1 (N=20), config 2 (N=100) and a
64-instance batch; HIP-event durations from the library (hsqp_last_kernel_ms) + wall clock around hsqp_iterate_device."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np  # noqa: E402

from wb_humanoid_mpc_amd import load_model  # noqa: E402
from wb_humanoid_mpc_amd.reference import make_centroidal_problem  # noqa: E402
from wb_humanoid_mpc_amd.solver import HipSqpSolver  # noqa: E402


def main():
    model = load_model(formulation="centroidal")
    res = {}
    cases = [("config1_N20_B1", 20, 1), ("config2_N100_B1", 100, 1), ("N100_B64", 100, 64)]
    if len(sys.argv) > 1:
        cases = cases[:int(sys.argv[1])]
    for name, n, b in cases:
        x0, x, u, par, dt = make_centroidal_problem(model, n_nodes=n, batch=b, perturb=b > 1)
        s = HipSqpSolver(model, max_nodes=n, max_batch=b, riccati=os.environ.get("HSQP_RICCATI", "auto"))
        s.upload(x0, x, u, par, dt)
        for _ in range(2):
            s.iterate(1)
        steps = 10
        kms = np.zeros(5)
        t0 = time.perf_counter()
        for _ in range(steps):
            s.iterate(1)
            k = s.kernel_ms()
            kms += [k["lq"], k["project"], k["riccati"], k["step_perf"], k["total"]]
        wall = (time.perf_counter() - t0) / steps
        s.iterate(1, kkt=True)
        out = s.download()
        res[name] = dict(ms_per_iteration=1e3 * wall, sqp_iters_per_s=b / wall,
                         kernel_ms=dict(zip(("lq", "project", "riccati", "step_perf", "sum"), (kms / steps).tolist())),
                         kkt=out["kkt"].max(), perf_before=out["perf_before"][0], perf_after=out["perf_after"][0])
        s.close()
        print(name, json.dumps(res[name]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "cent_timing.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
