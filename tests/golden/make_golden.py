#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (run in the build container).

There are no reference-produced vectors for this path (the solver lives in an absent
submodule and no reference test pins a solver result), so these fixtures pin the ORACLE: inputs built by
wb_humanoid_mpc_amd.reference.make_problem (seeded) and the oracle's outputs for them.  The GPU parity
tests compare the HIP path against these files without needing to run the oracle.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(HERE)]

from hsqp_oracle import Oracle  # noqa: E402
from test_oracle_lq import perturbed_problem  # noqa: E402
from wb_humanoid_mpc_amd import load_model  # noqa: E402


def main():
    model = load_model()
    oracle = Oracle(model)
    for name, gait, n, seed in (("wb_walk_n8", "walk", 8, 21), ("wb_run_n14", "run", 14, 22), ("wb_stance_n4", "stance", 4, 23)):
        x0, x, u, par, dt = perturbed_problem(model, n, gait, seed=seed)
        lq = oracle.lq(dt, x, u, par)
        r = oracle.sqp_iteration(dt, x0, x, u, par)
        pb, pa = r["perf_before"], r["perf_after"]
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), x_init=x0, x=x, u=u, par=par, dt=dt, dx=r["dx"], du=r["du"], kkt=r["kkt"],
            perf_before=np.array([pb["cost"], pb["dynamics_sse"], pb["equality_sse"]]),
            perf_after=np.array([pa["cost"], pa["dynamics_sse"], pa["equality_sse"]]),
            b=lq["b"], g=lq["g"], cost=lq["cost"], ne=lq["ne"], flow=lq["flow"], e=lq["CDe"][:, :, -1],
            AB_row29=lq["AB"][:, 29, :], H_diag=np.einsum("kii->ki", lq["H"]))
        print(name, "dx max", np.abs(r["dx"]).max(), "du max", np.abs(r["du"]).max())
    # centroidal formulation (padded 58 / 35 layout; BASELINE configs 1-2 at a size the fixture stays small)
    from test_oracle_centroidal_ocp import perturbed_centroidal_problem
    cmodel = load_model(formulation="centroidal")
    coracle = Oracle(cmodel)
    for name, gait, n, seed in (("cent_walk_n8", "walk", 8, 31), ("cent_run_n14", "run", 14, 32), ("cent_stance_n4", "stance", 4, 33)):
        x0, x, u, par, dt = perturbed_centroidal_problem(cmodel, n, gait, seed=seed)
        lq = coracle.cent_lq(dt, x, u, par)
        r = coracle.cent_sqp_iteration(dt, x0, x, u, par)
        pb, pa = r["perf_before"], r["perf_after"]
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), x_init=x0, x=x, u=u, par=par, dt=dt, dx=r["dx"], du=r["du"], kkt=r["kkt"],
            perf_before=np.array([pb["cost"], pb["dynamics_sse"], pb["equality_sse"]]),
            perf_after=np.array([pa["cost"], pa["dynamics_sse"], pa["equality_sse"]]),
            b=lq["b"], g=lq["g"], cost=lq["cost"][:n], ne=lq["ne"], flow=lq["flow"], e=lq["CDe"][:, :, -1],
            AB_row7=lq["AB"][:, 7, :], H_diag=np.einsum("kii->ki", lq["H"]))
        print(name, "dx max", np.abs(r["dx"]).max(), "du max", np.abs(r["du"]).max())


if __name__ == "__main__":
    main()
