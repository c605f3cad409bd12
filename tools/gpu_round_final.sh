#!/bin/bash
# a round's evidence in one GPU call: the bench lines of every BASELINE config on one GPU, the data-path leg at 32 / 64 / 128 instances (what one
# GPU of an 8- / 4- / 2-GPU run of config 4 holds) and at 128 x 200 nodes (one GPU of an 8-GPU run of config 5), the parity report, the phase profile, then tools/gpu_profiles.sh (rocprofv3 kernel
# stats + the PMC passes).  tools/collect_profiles.sh <round> copies the results to profiles/<round>_*.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
run() { name=$1; shift; timeout 600 python bench.py "$@" > "$OUT/$name.log" 2> "$OUT/$name.err"; echo "$name rc=$?"; python - "$OUT/$name.log" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["kernel_ms"].items()}, "kkt/|g|", d["kkt_over_max_1_g_inf"])
    if "strong_scaling" in d: print("   strong:", {k: d["strong_scaling"][k] for k in ("batch_per_gpu", "value", "ms_per_step", "kernel_ms", "gathered_solution_equals_single_gpu_solve")})
except Exception as e:
    print("   failed", e)
PY
}
run bench_wb --steps 20 --warmup 3
run bench_cfg3 --batch 1 --nodes 100 --no-perturb --steps 20 --warmup 3 --no-cpu-baseline
run bench_cfg3_serial --batch 1 --nodes 100 --no-perturb --steps 20 --warmup 3 --no-cpu-baseline --riccati serial
run bench_cfg5 --batch 1024 --nodes 200 --gait slow_walk --steps 3 --warmup 1 --no-cpu-baseline
run bench_cent_cfg1 --formulation centroidal --nodes 20 --steps 20 --warmup 3 --no-cpu-baseline
run bench_cent_cfg2 --formulation centroidal --nodes 100 --steps 20 --warmup 3 --no-cpu-baseline
run bench_strong32 --steps 10 --warmup 3 --no-cpu-baseline --force-strong --global-batch 32 --batch 32
run bench_strong64 --steps 10 --warmup 3 --no-cpu-baseline --force-strong --global-batch 64 --batch 64
# the shards nobody had timed: 128 x 100 = one GPU of a 2-GPU run of config 4; 128 x 200 slow_walk = one GPU of an 8-GPU run of config 5
run bench_strong128 --steps 10 --warmup 3 --no-cpu-baseline --force-strong --global-batch 128 --batch 128
run bench_strong128_n200 --steps 5 --warmup 2 --no-cpu-baseline --force-strong --global-batch 128 --batch 128 --nodes 200 --gait slow_walk
timeout 900 python tools/strong_prediction.py > "$OUT/strong_prediction.json" 2> "$OUT/strong_prediction.err"; echo "strong prediction rc=$?"
timeout 900 python tools/parity_report.py > "$OUT/parity_report.log" 2>&1; echo "parity rc=$?"; tail -12 "$OUT/parity_report.log"
timeout 300 python tools/phase_profile.py > "$OUT/phase.log" 2>&1; echo "phase rc=$?"
bash tools/gpu_profiles.sh > "$OUT/profiles.log" 2>&1; echo "profiles rc=$?"; tail -40 "$OUT/profiles.log"
