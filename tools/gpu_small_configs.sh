#!/bin/bash
# bench lines of the single-instance configs (1, 2, 3) and the 32-instance strong leg: the latency-bound paths
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for spec in "cfg3 --batch 1 --nodes 100 --no-perturb" "cfg3_serial --batch 1 --nodes 100 --no-perturb --riccati serial" "cent1 --formulation centroidal --nodes 20" "cent2 --formulation centroidal --nodes 100" \
            "strong32 --force-strong --global-batch 32 --batch 32"; do
  set -- $spec; name=$1; shift
  timeout 300 python bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d.get('strong_scaling')
print('$name', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['kernel_ms'].items()}, 'kkt/|g|', '%.2e' % d['kkt_over_max_1_g_inf'], (s and s.get('two_level_sweep') and round(s['two_level_sweep']['ms_per_step'], 3)))"
done
timeout 900 python -m pytest tests -m gpu -x -q -k "scan or parallel_in_time or config3 or centroidal" 2>&1 | tail -2
