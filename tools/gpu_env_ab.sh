#!/bin/bash
# (GPU) A/B of environment switches read at hsqp_create: one bench line pair per setting.
# Usage: gpurun -- 'VAR=HSQP_LQ_SPLIT VALUES="1 2 3 4" bash tools/gpu_env_ab.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
line() { python -c '
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line); print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["kernel_ms"].items()}, "kkt", d.get("kkt_over_max_1_g_inf"))
    elif "rror" in line: print(line[:300])
'; }
{
[ -n "$TESTS" ] && timeout 900 python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -5
for v in $VALUES; do
  echo "== $VAR=$v"
  for rep in 1 2; do env $VAR=$v timeout 300 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline ${BENCH_ARGS} 2>&1 | line; done
done
} > gpurun_out/env_ab.log 2>&1
cat gpurun_out/env_ab.log
