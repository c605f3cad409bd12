"""Probe (GPU box): which stage differs between an instance solved inside a large batch and the same instance solved alone."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa: F401
from wb_humanoid_mpc_amd import load_model, _abi
from wb_humanoid_mpc_amd.reference import make_problem, BENCH_SEED
from wb_humanoid_mpc_amd.solver import HipSqpSolver
m = load_model()
N, gait, B = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
x0, x, u, par, dt = make_problem(m, n_nodes=N, batch=B, gait=gait, perturb=True, seed=BENCH_SEED)
s = HipSqpSolver(m, max_nodes=N, max_batch=B)
out = s.run(x0, x, u, par, dt)
blocks = {k: s.debug_read(getattr(_abi, "BLK_" + k)) for k in ("AB", "BVEC", "H", "G", "CDE", "COST", "FLOW")}
out2 = s.run(x0, x, u, par, dt)
print("batch run repeatable:", np.array_equal(out["dx"], out2["dx"]), "max diff", np.abs(out["dx"] - out2["dx"]).max())
nbad = 0
for b in range(B):
    o1 = s.run(x0[b], x[b], u[b], par[b], dt)
    d = np.abs(o1["dx"][0] - out["dx"][b]).max()
    if d > 1e-9 * max(1, np.abs(o1["dx"]).max()):
        nbad += 1
        if nbad <= 4:
            diffs = {k: float(np.abs(s.debug_read(getattr(_abi, "BLK_" + k))[0] - blocks[k][b]).max()) for k in blocks}
            node = np.abs(o1["dx"][0] - out["dx"][b]).max(axis=1)
            print("instance", b, "dx diff %.3e" % d, "first node with diff > 1e-9:", int(np.argmax(node > 1e-9)), "LQ block diffs", diffs)
print("bad instances:", nbad, "of", B)
