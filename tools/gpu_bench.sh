#!/bin/bash
# bench lines: default (config 4 + CPU baseline), the data-path leg forced on one GPU, config 3, centroidal config 2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
timeout 400 python bench.py > "$OUT/bench_wb.log" 2> "$OUT/bench_wb.err"; echo "rc=$?"; tail -c 2500 "$OUT/bench_wb.log"; tail -3 "$OUT/bench_wb.err"
timeout 200 python bench.py --force-strong --no-cpu-baseline --global-batch 64 > "$OUT/bench_strong1.log" 2> "$OUT/bench_strong1.err"; echo "rc=$?"; tail -c 1500 "$OUT/bench_strong1.log"; tail -3 "$OUT/bench_strong1.err"
timeout 200 python bench.py --batch 1 --no-perturb --no-cpu-baseline > "$OUT/bench_cfg3.log" 2> "$OUT/bench_cfg3.err"; echo "rc=$?"; tail -c 700 "$OUT/bench_cfg3.log"
timeout 200 python bench.py --formulation centroidal --no-cpu-baseline > "$OUT/bench_cent_cfg2.log" 2> "$OUT/bench_cent_cfg2.err"; echo "rc=$?"; tail -c 700 "$OUT/bench_cent_cfg2.log"
