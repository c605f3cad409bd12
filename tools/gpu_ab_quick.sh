#!/bin/bash
# Quick A/B on the GPU box: the product library and every library under wb_humanoid_mpc_amd/variants/: a small parity check first (config-3 style
# instance against the oracle + serial-vs-batch), then the bench line's kernel times; optionally the Riccati phase profile of the product library.
# Usage: gpurun -- 'PROFILE=1 TESTS="tests/test_gpu_parity.py -k config3" bash tools/gpu_ab_quick.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
libs=("" $(ls wb_humanoid_mpc_amd/variants/libhsqp_*.so 2>/dev/null | sed 's/.*libhsqp_//; s/\.so//'))
for v in "${libs[@]}"; do
  lib=$PWD/wb_humanoid_mpc_amd/libhsqp_hip.so; [ -n "$v" ] && lib=$PWD/wb_humanoid_mpc_amd/variants/libhsqp_$v.so
  echo "== ${v:-product}"
  [ -n "$TESTS" ] && HSQP_LIB=$lib timeout 900 python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -3
  for rep in 1 2; do
  HSQP_LIB=$lib timeout 300 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline ${BENCH_ARGS} 2>&1 | python -c '
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line); print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["kernel_ms"].items()}, "kkt", d.get("kkt_over_max_1_g_inf"))
    elif "rror" in line: print(line[:300])
'
  done
done
if [ -n "$PROFILE" ]; then
  echo "== phase profile (product library, -DHSQP_PHASE_PROFILE build)"
  timeout 300 python tools/phase_profile.py ${PB:-256} ${PN:-100} 2>&1 | sed -n '/k_riccati/,/k_lq<false>/p'
fi
} > gpurun_out/ab_quick.log 2>&1
cat gpurun_out/ab_quick.log
