#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, a short bench, and a rocprofv3 kernel trace of the bench.
# Everything lands under gpurun_out/ (merged back into the repo copy).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > "$OUT/device.txt"
nproc >> "$OUT/device.txt"; grep -m1 "model name" /proc/cpuinfo >> "$OUT/device.txt"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -60 > "$OUT/pytest_gpu.log"
echo "pytest exit: $?" >> "$OUT/pytest_gpu.log"
timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 > "$OUT/bench.log" 2> "$OUT/bench.err"
echo "bench exit: $?" >> "$OUT/bench.err"
if [ "${PROFILE:-1}" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench -- python "$OUT/../bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/prof.log" 2>&1
  echo "rocprof exit: $?" >> "$OUT/prof.log"
  find "$OUT/prof" -name "*stats*" | head > "$OUT/prof_files.txt"
fi
tail -5 "$OUT/pytest_gpu.log"; cat "$OUT/bench.log" | cut -c1-1500; tail -3 "$OUT/bench.err"
