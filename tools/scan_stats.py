#!/usr/bin/env python3
"""Per-kernel averages of gpurun_out/scan_{wb,cent}_kernel_stats.csv (tools/gpu_scan_profile.sh)."""
import csv, os, re, sys
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
for d in ("wb", "cent"):
    print("==", d)
    for row in csv.DictReader(open(os.path.join(root, f"scan_{d}_kernel_stats.csv"))):
        m = re.search(r"(k_\w+(?:<\w+>)?)", row["Name"])
        if m:
            print(f"{m.group(1):24s} calls {row['Calls']:>5s} avg {float(row['AverageNs']) / 1e3:8.1f} us")
