#!/usr/bin/env python3
"""Device LQ blocks against the oracle, block by block and entry by entry (GPU box; HSQP_LIB selects the build): where a kernel change broke parity."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from hsqp_oracle import Oracle  # noqa: E402
from test_oracle_lq import perturbed_problem  # noqa: E402
from wb_humanoid_mpc_amd import _abi, load_model  # noqa: E402
from wb_humanoid_mpc_amd.solver import HipSqpSolver  # noqa: E402

m = load_model()
o = Oracle(m)
gait, n = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("walk", 6)
x0, x, u, par, dt = perturbed_problem(m, n, gait, seed=31)
s = HipSqpSolver(m, max_nodes=n, max_batch=1)
s.run(x0, x, u, par, dt)
lq = o.lq(dt, x, u, par, threads=4)
for blk, key in ((_abi.BLK_FLOW, "flow"), (_abi.BLK_BVEC, "b"), (_abi.BLK_AB, "AB"), (_abi.BLK_H, "H"), (_abi.BLK_G, "g"), (_abi.BLK_CDE, "CDe"), (_abi.BLK_COST, "cost")):
    a, b = s.debug_read(blk)[0], np.asarray(lq[key])
    d = np.abs(a - b)
    print(f"{key:5s} rel {d.max() / max(1.0, np.abs(b).max()):.3e}", end="")
    if d.max() > 1e-9 * max(1.0, np.abs(b).max()):
        bad = np.argwhere(d > 1e-9 * max(1.0, np.abs(b).max()))
        print("  bad entries", len(bad), "of", d.size, "; first", bad[:6].tolist(), end="")
        if key == "AB":
            k0 = bad[0][0]
            rows = sorted(set(bad[bad[:, 0] == k0][:, 1].tolist())); cols = sorted(set(bad[bad[:, 0] == k0][:, 2].tolist()))
            print("\n      node", k0, "rows", rows, "\n      cols", cols, end="")
    print()
