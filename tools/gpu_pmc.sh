#!/bin/bash
# HBM traffic counters of the bench kernels (GPU box, via gpurun).  Counters are collected in their own runs, one
# counter per pass (FETCH_SIZE and WRITE_SIZE do not fit the TCC slots together), with --kernel-trace only.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o pmc -- \
    python "$OUT/../bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_$C.log" 2>&1
  echo "rocprof $C exit: $?" >> "$OUT/pmc_$C.log"
done
cd "$OUT/.."
python tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json"
cat "$OUT/pmc_summary.json"
