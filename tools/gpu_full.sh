#!/bin/bash
# Full GPU regression: every -m gpu test, the default bench line (with CPU baseline) and the centroidal bench line (config 2).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > "$OUT/pytest_gpu.log"
tail -4 "$OUT/pytest_gpu.log"
timeout 200 python bench.py --formulation centroidal --nodes 100 > "$OUT/bench_cent_cfg2.log" 2> "$OUT/bench_cent_cfg2.err"
tail -c 600 "$OUT/bench_cent_cfg2.log"; tail -3 "$OUT/bench_cent_cfg2.err"
if [ "${WB_BENCH:-1}" = 1 ]; then
  timeout 300 python bench.py > "$OUT/bench_wb.log" 2> "$OUT/bench_wb.err"
  tail -c 400 "$OUT/bench_wb.log"; tail -3 "$OUT/bench_wb.err"
fi
