#!/bin/bash
# round 2, first GPU call: regression of every -m gpu test, measured parity table (tools/parity_report.py), default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
timeout 500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > "$OUT/pytest_gpu.log"
tail -4 "$OUT/pytest_gpu.log"
timeout 600 python tools/parity_report.py --out "$OUT/parity_report.json" > "$OUT/parity_report.log" 2>&1
cat "$OUT/parity_report.log" | tail -12
timeout 300 python bench.py > "$OUT/bench_wb.log" 2> "$OUT/bench_wb.err"
tail -c 600 "$OUT/bench_wb.log"; tail -3 "$OUT/bench_wb.err"
nproc; grep -m1 "model name" /proc/cpuinfo
