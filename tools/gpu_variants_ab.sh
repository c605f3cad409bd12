#!/bin/bash
# (GPU) interleaved bench lines of the product library and every library under wb_humanoid_mpc_amd/variants/ (REPS rounds, default 4): kernel buckets only
cd "${GRAFT_REPO_ROOT:-/root/repo}"
libs=("" $(ls wb_humanoid_mpc_amd/variants/libhsqp_*.so 2>/dev/null | sed 's/.*libhsqp_//; s/\.so//'))
for rep in $(seq 1 ${REPS:-4}); do for v in "${libs[@]}"; do
 lib=$PWD/wb_humanoid_mpc_amd/libhsqp_hip.so; [ -n "$v" ] && lib=$PWD/wb_humanoid_mpc_amd/variants/libhsqp_$v.so
 HSQP_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sustained 0 ${BENCH_ARGS} 2>&1 | python -c '
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1], round(d["ms_per_step"], 3), {k: round(v, 4) for k, v in d["kernel_ms"].items()})
' "${v:-product}"
done; done
