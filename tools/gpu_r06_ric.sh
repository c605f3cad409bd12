#!/bin/bash
# Round 6: the factored serial sweep (k_riccati_fact) against the dense stage (HSQP_RICCATI_DENSE=1) — PARITY FIRST, then timings, then the
# phase profile.  Usage: gpurun -- 'bash tools/gpu_r06_ric.sh [quick]'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r06_ric.log
: > $L
if [ "$1" != "quick" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 >> $L
else
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_convergence.py -m gpu -x -q 2>&1 | tail -15 >> $L
fi
line() { python -c '
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1], "value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["kernel_ms"].items()})
    elif "rror" in line: print(line[:300])
' "$1"; }
for rep in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | line "fact256" >> $L
  HSQP_RICCATI_DENSE=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | line "dense256" >> $L
done
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batch 32 --no-strong 2>&1 | line "fact32" >> $L
HSQP_RICCATI_DENSE=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batch 32 --no-strong 2>&1 | line "dense32" >> $L
timeout 300 python tools/phase_profile.py 256 100 > gpurun_out/r06_phase_fact.log 2>&1
sed -n '/k_project/,/k_lq<false>/p' gpurun_out/r06_phase_fact.log >> $L
cat $L
