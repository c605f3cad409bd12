#!/bin/bash
# Bench every tuning build under wb_humanoid_mpc_amd/variants/ (GPU box, via gpurun).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/variants.log
for lib in wb_humanoid_mpc_amd/variants/libhsqp_*.so; do
  echo "== $lib" >> gpurun_out/variants.log
  HSQP_LIB=$PWD/$lib timeout 300 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline 2>&1 | python -c '
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line); print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["kernel_ms"].items()})
    elif "rror" in line: print(line[:300])
' >> gpurun_out/variants.log
done
cat gpurun_out/variants.log
