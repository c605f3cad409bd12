#!/bin/bash
# End-of-round GPU run: every -m gpu test, smoke, centroidal kernel stats (rocprofv3), centroidal and whole-body bench lines.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
timeout 200 python -m pytest tests -m gpu -q 2>&1 | tail -6 > "$OUT/pytest_gpu.log"; tail -3 "$OUT/pytest_gpu.log"
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cent -o cent -- python $R/tools/cent_timing.py > $R/gpurun_out/cent_timing.log 2>&1)
sed -E 's/"perf_before.*//' "$OUT/cent_timing.log" | grep -E "^config|^N100" 
timeout 100 python bench.py --formulation centroidal --nodes 100 > "$OUT/bench_cent_cfg2.log" 2> "$OUT/bench_cent_cfg2.err"; tail -c 300 "$OUT/bench_cent_cfg2.log"; echo
if [ "${WB_BENCH:-1}" = 1 ]; then timeout 200 python bench.py > "$OUT/bench_wb.log" 2> "$OUT/bench_wb.err"; tail -c 300 "$OUT/bench_wb.log"; echo; fi
