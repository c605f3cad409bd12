#!/bin/bash
# kernel stats of the parallel-in-time sweeps: config 3 (whole-body, n = 58) and config 2 (centroidal, n = 35)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"; rm -rf "$OUT/prof_scan_wb" "$OUT/prof_scan_cent"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_scan_wb" -o scan -- python "$OUT/../bench.py" --batch 1 --no-perturb --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/prof_scan_wb.log" 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_scan_cent" -o scan -- python "$OUT/../bench.py" --formulation centroidal --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/prof_scan_cent.log" 2>&1
cd "$OUT/.."
for d in wb cent; do f=$(find "$OUT/prof_scan_$d" -name "*kernel_stats.csv" | head -1); cp "$f" "$OUT/scan_${d}_kernel_stats.csv"; echo "== $d"; cut -d, -f1-4,7 "$OUT/scan_${d}_kernel_stats.csv" | sed 's/void (anonymous namespace):://; s/(.*)//' | head -16; done
