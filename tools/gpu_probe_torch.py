"""GPU-box probe: can torch (bundled ROCm runtime) and libhsqp_hip.so (system ROCm) share one process, and in which import order?"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, time, os
sys.path.insert(0, %r)
t0 = time.time()
order = sys.argv[1]
def load_ours():
    from wb_humanoid_mpc_amd.solver import load_library
    lib = load_library(); print("  ours loaded, devices:", lib.hsqp_device_count(), "t=%%.1f" %% (time.time()-t0), flush=True)
def load_torch():
    import torch
    print("  torch", torch.__version__, "cuda avail", torch.cuda.is_available(), "t=%%.1f" %% (time.time()-t0), flush=True)
    if torch.cuda.is_available():
        x = torch.ones(4, device="cuda"); torch.cuda.synchronize(); print("  torch tensor ok", float(x.sum()), flush=True)
if order == "ours_first": load_ours(); load_torch()
else: load_torch(); load_ours()
import numpy as np
from wb_humanoid_mpc_amd import load_model
from wb_humanoid_mpc_amd.reference import make_problem
from wb_humanoid_mpc_amd.solver import HipSqpSolver
m = load_model()
x0, x, u, par, dt = make_problem(m, n_nodes=8, batch=2, perturb=True)
s = HipSqpSolver(m, max_nodes=8, max_batch=2)
out = s.run(x0, x, u, par, dt)
print("  solver ok kkt", out["kkt"].max(), s.kernel_ms(), "t=%%.1f" %% (time.time()-t0), flush=True)
os.system("grep -E 'libamdhip64|libhsa-runtime' /proc/%%d/maps | awk '{print $6}' | sort -u" %% os.getpid())
''' % ROOT
for order in ("torch_first", "ours_first"):
    print("==", order, flush=True)
    t = time.time()
    r = subprocess.run([sys.executable, "-c", CODE, order], capture_output=True, text=True, timeout=600)
    print(r.stdout[-1500:], r.stderr[-800:], "rc", r.returncode, "wall %.1f" % (time.time() - t), flush=True)
