#!/usr/bin/env python3
"""SQ counters of the bench kernels (tools/gpu_profiles.sh): per-launch means, the matrix-pipe busy share
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x 2.4 GHz) with the duration from the rocprofv3 kernel stats of the same
command, the issue share ACTIVE_INST_ANY / WAVE_CYCLES, and the dynamic instruction mix per wave.  SQ_INSTS_VALU_MFMA_MOPS_F64 counts
in units of 512 flop: a v_mfma_f64_16x16x4 (2048 flop) is 4 of them."""
import collections
import csv
import glob
import json
import os
import re
import sys

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"


def per_kernel(sub):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(k_\w+(?:<\w+>)?)", row["Kernel_Name"])
            if m:
                per[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in per.items()}


dur = {}
try:
    for row in csv.DictReader(open(os.path.join(out, "kernel_stats.csv"))):
        m = re.search(r"(k_\w+(?:<\w+>)?)", row["Name"])
        if m:
            dur[m.group(1)] = float(row["AverageNs"]) * 1e-9
except Exception as e:  # noqa: BLE001
    print("no kernel stats:", e, file=sys.stderr)
sq, insts = per_kernel("pmc_sq"), per_kernel("pmc_insts")
summary = {}
with open(os.path.join(out, "pmc_sq_summary.txt"), "w") as fh:
    fh.write("# tools/gpu_profiles.sh: SQ counters per launch (M), bench.py --steps 2 --warmup 1, config 4\n")
    for k, v in sq.items():
        e = {"counters_M_per_launch": {c: round(x / 1e6, 2) for c, x in v.items()}}
        if k in dur and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            e["mfma_busy"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * dur[k] * 2.4e9)
            e["duration_ms"] = dur[k] * 1e3
        if v.get("SQ_WAVE_CYCLES"):
            e["issue_share_of_wave_cycles"] = v.get("SQ_ACTIVE_INST_ANY", 0.0) / v["SQ_WAVE_CYCLES"]
        summary[k] = e
        fh.write(f"{k} {e}\n")
with open(os.path.join(out, "pmc_insts_summary.txt"), "w") as fh:
    fh.write("# dynamic instructions per wave (SQ_INSTS_* / SQ_WAVES); MFMA column = SQ_INSTS_VALU_MFMA_MOPS_F64 / 4 = v_mfma_f64_16x16x4 instructions\n")
    for k, v in insts.items():
        w = v.get("SQ_WAVES", 0.0) or 1.0
        e = {c.replace("SQ_INSTS_", ""): round(x / w, 1) for c, x in v.items() if c != "SQ_WAVES"}
        if "VALU_MFMA_MOPS_F64" in e:
            e["MFMA_f64_16x16x4"] = round(e.pop("VALU_MFMA_MOPS_F64") / 4.0, 1)
        e["waves"] = int(w)
        summary.setdefault(k, {})["insts_per_wave"] = e
        fh.write(f"{k} {e}\n")
json.dump(summary, open(os.path.join(out, "pmc_sq_summary.json"), "w"), indent=1)
