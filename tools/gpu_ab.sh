#!/bin/bash
# A/B of tuning builds (tools/build_variants.py) on the GPU box, PARITY FIRST: every variant must reproduce the oracle's LQ blocks and
# pass the whole-body parity tests before its timing counts (round 3: a variant that benched 14 % faster computed wrong placements —
# the host emulation cannot see device-only miscompiles).  Usage: gpurun -- 'bash tools/gpu_ab.sh [variant ...]'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/ab.log
libs=("$@")
[ ${#libs[@]} -eq 0 ] && libs=($(ls wb_humanoid_mpc_amd/variants/libhsqp_*.so | sed 's/.*libhsqp_//; s/\.so//'))
for v in "${libs[@]}"; do
  lib=$PWD/wb_humanoid_mpc_amd/variants/libhsqp_$v.so
  echo "== $v" >> gpurun_out/ab.log
  HSQP_LIB=$lib timeout 300 python tools/gpu_blocks.py walk 6 2>&1 | cut -c1-80 | awk '{printf "%s | ", $0} END {print ""}' >> gpurun_out/ab.log
  HSQP_LIB=$lib timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_convergence.py -m gpu -x -q 2>&1 | tail -1 >> gpurun_out/ab.log
  HSQP_LIB=$lib timeout 300 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline 2>&1 | python -c '
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line); print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["kernel_ms"].items()})
    elif "rror" in line: print(line[:300])
' >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
