#!/bin/bash
# (GPU) round 6: the RK4 chain fused into k_project against the chain in k_lq_chain (HSQP_LQ_CHAIN_SEPARATE=1): parity first, then interleaved bench lines.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
line() { python -c '
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line); print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), {k: round(v, 4) for k, v in d["kernel_ms"].items()}, "kkt", d.get("kkt_over_max_1_g_inf"))
    elif "rror" in line: print(line[:300])
'; }
{
timeout 1200 python -m pytest ${TESTS:-tests/test_gpu_parity.py} -m gpu -x -q 2>&1 | tail -8
for rep in 1 2 3 ${REPS}; do
  echo "== fused"; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_ARGS} 2>&1 | line
  echo "== separate"; HSQP_LQ_CHAIN_SEPARATE=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_ARGS} 2>&1 | line
done
} > gpurun_out/chain_ab.log 2>&1
cat gpurun_out/chain_ab.log
