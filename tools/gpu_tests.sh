#!/bin/bash
# every -m gpu test; "$@" = extra pytest arguments (e.g. -k convergence -s)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
timeout 1200 python -m pytest tests -m gpu -q --durations=8 "$@" > "$OUT/pytest_gpu.log" 2>&1
grep -E "^E  |^FAILED|^ERROR|passed|failed|^[a-z0-9_]+: violation" "$OUT/pytest_gpu.log" | head -80
