#!/bin/bash
# Parallel-in-time backward sweep on the GPU: tests, timings (scan vs serial), kernel stats of the scan run; optional WB regression.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
timeout 150 python -m pytest tests/test_gpu_centroidal.py -q 2>&1 | tail -6
HSQP_RICCATI=parallel timeout 60 python tools/cent_timing.py 2 2>&1 | sed -E 's/"perf_before.*//' | tail -2
export TMPDIR=/tmp
R=$PWD
(cd /tmp && HSQP_RICCATI=parallel timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_scan -o scan -- python $R/tools/cent_timing.py 2 > $R/gpurun_out/prof_scan.log 2>&1)
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/prof_scan/scan_kernel_stats.csv')):
    if float(r['AverageNs']) > 15e3: print(r['Name'][:60].ljust(60), r['Calls'], 'avg us %.1f' % (float(r['AverageNs']) / 1e3))
PY
if [ "${WB:-0}" = 1 ]; then
  timeout 120 python -m pytest tests/test_gpu_parity.py -q 2>&1 | tail -3
  timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WB', round(d['value'],1), d['kernel_ms'])"
fi
