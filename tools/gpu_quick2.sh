#!/bin/bash
# quick A/B: bench line without the CPU baseline (kernel_ms), optionally a subset of gpu tests first ($1 = -k expression)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
if [ -n "$1" ]; then timeout 600 python -m pytest tests -m gpu -q -x -k "$1" 2>&1 | tail -3; fi
for i in 1 2; do
timeout 200 python bench.py --no-cpu-baseline --steps 30 2> /dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['ms_per_step'],3), {k: round(v,3) for k,v in r['kernel_ms'].items()}, 'kkt', r['kkt_over_max_1_g_inf'])"
done
