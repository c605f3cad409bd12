#!/bin/bash
# Round 4, first GPU pass of the blocked factorisation: the probe (both forms alone), the parity tests on the new library, A/B bench against the
# column-by-column build (variants/libhsqp_colwise.so), the phase profile.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
echo "== probe"; timeout 120 tools/microbench/elim_blocked_probe.bin
echo "== tests (new library)"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_convergence.py tests/test_gpu_centroidal.py -m gpu -x -q 2>&1 | tail -5
for v in "" colwise; do
  lib=$PWD/wb_humanoid_mpc_amd/libhsqp_hip.so; [ -n "$v" ] && lib=$PWD/wb_humanoid_mpc_amd/variants/libhsqp_$v.so
  echo "== bench ${v:-new}"
  HSQP_LIB=$lib timeout 300 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline 2>&1 | python -c '
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line); print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["kernel_ms"].items()}, "kkt", d.get("kkt_over_max_1_g_inf"))
    elif "rror" in line: print(line[:300])
'
done
echo "== phase profile"
timeout 300 python tools/phase_profile.py 256 100 2>&1 | sed -n '/k_riccati/,/k_lq<false>/p'
} > gpurun_out/r4_elim.log 2>&1
cat gpurun_out/r4_elim.log
