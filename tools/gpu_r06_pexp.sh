#!/bin/bash
# Round 6: kernel times of the timing-experiment builds of k_project (wb_humanoid_mpc_amd/variants/libhsqp_p*.so, -DHSQP_PEXP=<mask>: WRONG results, timings only).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r06_pexp.log
: > $L
for lib in wb_humanoid_mpc_amd/variants/libhsqp_*.so; do
  v=$(basename $lib .so | sed 's/libhsqp_//')
  HSQP_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sustained 0 2>&1 | python -c '
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1], "ms/step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["kernel_ms"].items()})
    elif "rror" in line: print(sys.argv[1], line[:200])
' "$v" >> $L
done
cat $L
