#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, a short bench, and a rocprofv3 kernel trace of the bench.
# Everything lands under gpurun_out/ (merged back into the repo copy).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > "$OUT/device.txt"
nproc >> "$OUT/device.txt"; grep -m1 "model name" /proc/cpuinfo >> "$OUT/device.txt"
date > "$OUT/times.txt"
if [ "${PYTEST:-1}" = "1" ]; then timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -60 > "$OUT/pytest_gpu.log"; fi
echo "pytest exit: $?" >> "$OUT/pytest_gpu.log"
date >> "$OUT/times.txt"
timeout 600 python bench.py --steps ${STEPS:-10} --warmup ${WARMUP:-3} ${BENCH_ARGS:-} > "$OUT/bench.log" 2> "$OUT/bench.err"
BRC=$?
echo "bench exit: $BRC" >> "$OUT/bench.err"
date >> "$OUT/times.txt"
if [ "${PROFILE:-1}" = "1" ] && [ "$BRC" = "0" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout -s KILL 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python "$OUT/../bench.py" --steps ${PSTEPS:-3} --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > "$OUT/prof.log" 2>&1
  echo "rocprof exit: $?" >> "$OUT/prof.log"
  find "$OUT/prof" -type f | head -20 > "$OUT/prof_files.txt"
  date >> "$OUT/times.txt"
fi
tail -5 "$OUT/pytest_gpu.log"; cat "$OUT/bench.log" | cut -c1-1500; tail -3 "$OUT/bench.err"
