#!/bin/bash
# end-of-round evidence that is not in gpu_profiles_r02.sh / gpu_bench.sh: parity report, phase profiles, config 5 on one GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
timeout 900 python tools/parity_report.py > "$OUT/parity_report.log" 2>&1; echo "parity rc=$?"; tail -12 "$OUT/parity_report.log"
timeout 300 python tools/phase_profile.py > "$OUT/phase.log" 2>&1; echo "phase rc=$?"
timeout 300 python tools/phase_profile.py 1 100 > "$OUT/phase_b1.log" 2>&1; echo "phase b1 rc=$?"
timeout 600 python bench.py --batch 1024 --nodes 200 --gait slow_walk --no-cpu-baseline --steps 5 --warmup 1 > "$OUT/bench_cfg5.log" 2> "$OUT/bench_cfg5.err"; echo "cfg5 rc=$?"; tail -c 800 "$OUT/bench_cfg5.log"
