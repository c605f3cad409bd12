#!/usr/bin/env python3
"""Per-phase shader-clock profile of the kernels (GPU box): builds/uses libhsqp_hip_prof.so (-DHSQP_PHASE_PROFILE),
runs a few iterations of the bench workload and prints ticks per phase for workgroup 0 of each kernel."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HSQP_LIB"] = os.environ.get("HSQP_PROF_LIB") or os.path.join(ROOT, "wb_humanoid_mpc_amd", "libhsqp_hip_prof.so")
os.environ.setdefault("HSQP_LQ_SPLIT", "1")   # one launch per LQ kernel: workgroup 0 of every range would tick into the same slots
import numpy as np  # noqa: E402

from wb_humanoid_mpc_amd import load_model  # noqa: E402
from wb_humanoid_mpc_amd.reference import make_centroidal_problem, make_problem  # noqa: E402
from wb_humanoid_mpc_amd.solver import HipSqpSolver  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
iters = 3
CENT = len(sys.argv) > 3 and sys.argv[3] == "centroidal"   # phase_profile.py B N centroidal: k_riccati<35> etc.
m = load_model(formulation="centroidal" if CENT else "wb")
x0, x, u, par, dt = make_centroidal_problem(m, n_nodes=N, batch=B, perturb=B > 1) if CENT else make_problem(m, n_nodes=N, batch=B, perturb=True)
s = HipSqpSolver(m, max_nodes=N, max_batch=B)
s.upload(x0, x, u, par, dt)
for _ in range(iters):
    s.iterate(1, kkt=True)
print("kernel ms (last iteration):", {k: round(v, 3) for k, v in s.kernel_ms().items()})
t = np.zeros((4, 128), dtype=np.int64)
s.lib.hsqp_debug_read(s.h, 100, t.ctypes.data_as(C.c_void_p), t.nbytes)
names = ["k_lq<true>", "k_project", "k_riccati", "k_lq<false> | k_lq_limb (ids 0-4: base kin + forward + solve, -, back walk, base columns; x4 stages) + k_lq_rows (10: loads, 11: kinematics, 12: terms, 13: joint rows, 14: base rows)"]
for k in range(4):
    # slots 0..39: phase ticks (PH_TICK); 40..111: per-wave barrier arrivals (PH_ARRIVE); 90..94: tile-call split; 112..119: PH_MARK stamps
    tot = t[k, :20].sum()
    print(f"== {names[k]}: {tot / iters:.0f} ticks per launch (workgroup 0)  [~{tot / iters / 2.4e3:.1f} us at 2.4 GHz]")
    for i in range(20):
        if t[k, i]:
            print(f"   phase {i:3d}: {t[k, i] / iters:12.0f} ticks  {100.0 * t[k, i] / tot:5.1f}%")
    if k == 2 and t[k, 20:39].any():
        print("   laps of the chosen wave (slots 20..38):", [int(v / iters) for v in t[k, 20:39] if v])
    if k == 2 and t[k, 40:112].any():   # k_riccati: per-wave arrival at the barriers ending Ph1, Ph2, Ph4, Ph3 (ticks per launch)
        for slot, name in enumerate(("Ph1", "Ph2", "Ph4", "Ph3")):
            print(f"   {name} arrival per wave:", [int(v / iters) for v in t[k, 40 + 8 * slot:48 + 8 * slot]])
    if t[k, 93]:   # wave 0: tile-job calls split into prologue / matrix loop / epilogue
        n, nm = t[k, 93] / iters, t[k, 94] / iters
        print(f"   tile jobs of wave 0: {n:.0f} calls, {nm:.0f} MFMA; prologue {t[k, 90] / iters:.0f}, loop {t[k, 91] / iters:.0f} "
              f"({t[k, 91] / max(t[k, 94], 1):.0f} per MFMA), epilogue {t[k, 92] / iters:.0f} ticks")
