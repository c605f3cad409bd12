#!/usr/bin/env python3
"""profiles/r06_strong_prediction.json in ONE GPU call: the step time of every shard size of BASELINE config 4 (256 instances over 1 / 2 / 4 / 8 GPUs =
256 / 128 / 64 / 32 instances per GPU, N = 100) on ONE GPU, through the same code path `bench.py --gpus N` runs per rank (the data-path leg with the
collectives degenerate to copies: --force-strong), with the opt-in two-level sweep and its measured distance from the exact path beside it.  `bench.py
--gpus N` prints `strong_prediction.predicted_ms_per_step` from this file next to what it measures, so the first real multi-GPU run confirms or
refutes it.  No multi-GPU node was available to this build in any round: this is a prediction, never a measurement of scaling.

    gpurun -- 'python tools/strong_prediction.py > gpurun_out/strong_prediction.json'
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "5", "--no-cpu-baseline", *extra],
                         capture_output=True, text=True, timeout=900)
    for line in out.stdout.splitlines():
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"bench.py {extra}: no JSON line\n{out.stdout[-2000:]}\n{out.stderr[-2000:]}")


def main():
    shards, nodes = {}, 100
    full = bench("--sustained", "300")
    shards["256"] = {"gpus": 1, "exact": {"ms_per_step": full["ms_per_step"], "kernel_ms": full["kernel_ms"], "sustained_ms_per_step": full["sustained"]["ms_per_step"]}}
    clocks = full.get("gpu_clocks")
    for b, g in ((128, 2), (64, 4), (32, 8)):
        r = bench("--force-strong", "--global-batch", str(b), "--batch", str(b), "--sustained", "0")
        s = r["strong_scaling"]
        row = {"gpus": g, "exact": {"ms_per_step": s["ms_per_step"], "kernel_ms": s["kernel_ms"], "gathered_solution_equals_single_gpu_solve": s["gathered_solution_equals_single_gpu_solve"],
                                    "collective_inclusive_ms_per_step": s["collective_inclusive"]["ms_per_step"]}}
        if "two_level_sweep" in s:
            t = s["two_level_sweep"]
            row["two_level_sweep"] = {"ms_per_step": t["ms_per_step"], "kernel_ms": t["kernel_ms"], "max_abs_difference_from_the_default_path": t["max_abs_difference_from_the_default_path"],
                                      "gate_fallbacks": t["gate_fallbacks_max_over_ranks"], "segments_per_instance": t["segments_per_instance"]}
        shards[str(b)] = row
    curve = {}
    for b, row in shards.items():
        g = row["gpus"]
        curve[str(g)] = {"batch_per_gpu": int(b), "exact_iters_per_s": 256 / (row["exact"]["ms_per_step"] * 1e-3),
                         "two_level_iters_per_s": 256 / (row["two_level_sweep"]["ms_per_step"] * 1e-3) if "two_level_sweep" in row else None}
    one = curve["1"]["exact_iters_per_s"]
    for g in curve:
        curve[g]["exact_speedup_vs_1_gpu"] = curve[g]["exact_iters_per_s"] / one
    print(json.dumps({"what": "BASELINE config 4 (256 instances x 100 nodes) by shard size, each timed on ONE MI355X", "nodes": nodes, "measured_on": "one MI355X (gpurun box), one call",
                      "gpu_clocks": clocks, "config4_shards": shards, "predicted_curve_by_gpus": curve,
                      "note": "prediction for `bench.py --gpus N` (strong scaling, shards resident): per-step time of the shard one GPU holds; the exact (default) sweep is the headline, "
                              "the two-level sweep an opt-in relaxation with the stated distance"}, indent=1))


if __name__ == "__main__":
    main()
