#!/usr/bin/env python3
"""Per-kernel HBM traffic from the rocprofv3 --pmc passes of tools/gpu_profiles.sh.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch (summed over XCDs).  On gfx950 FETCH_SIZE
counts 128-B fabric requests at 64 B (MI355X_MICROARCH.md, HBM section): it is doubled here; WRITE_SIZE is taken as is
(uncalibrated per the guide)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
res = {"unit": "bytes per launch", "correction": {"FETCH_SIZE": "x2 (gfx950 128-B requests tallied at 64 B)", "WRITE_SIZE": "none (uncalibrated)"},
       "kernels": defaultdict(dict)}
for counter, scale in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
    files = glob.glob(os.path.join(out, f"pmc_{counter}", "**", "*counter_collection.csv"), recursive=True)
    per = defaultdict(list)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row.get("Kernel_Name", "?")
                m = re.search(r"(k_\w+(?:<\w+>)?)", name)
                short = m.group(1) if m else name.split("(")[0][:40]
                per[short].append(float(row["Counter_Value"]) * 1024.0 * scale)
    for k, v in per.items():
        res["kernels"][k][counter] = {"mean": sum(v) / len(v), "dispatches": len(v)}
print(json.dumps(res, indent=1))
