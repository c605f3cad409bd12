#!/usr/bin/env python3
"""Tuning builds of the HIP library (launch shapes etc.) -> wb_humanoid_mpc_amd/variants/libhsqp_<name>.so.
Run on the build host; tools/gpu_ab_quick.sh benches every variant found on the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wb_humanoid_mpc_amd import build as B  # noqa: E402

VARIANTS = {}
KEEP = "--keep" in sys.argv   # keep the variants already built (e.g. the previous commit's library as the A/B baseline)
for arg in [a for a in sys.argv[1:] if a != "--keep"]:
    name, flags = arg.split("=", 1)
    VARIANTS[name] = tuple(f for f in flags.split(",") if f)
out = os.path.join(ROOT, "wb_humanoid_mpc_amd", "variants")
os.makedirs(out, exist_ok=True)
for f in ([] if KEEP else os.listdir(out)):
    os.remove(os.path.join(out, f))
for name, flags in VARIANTS.items():
    lib = os.path.join(out, f"libhsqp_{name}.so")
    B._build(lib, True, False, flags)
    print("built", lib, flags)
