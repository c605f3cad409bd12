#!/bin/bash
# Quick GPU iteration loop (via gpurun): GPU parity tests, bench without the CPU baseline leg, per-phase tick profile.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > "$OUT/pytest_gpu.log"
timeout 300 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline > "$OUT/bench_quick.log" 2> "$OUT/bench_quick.err"
timeout 300 python tools/phase_profile.py ${PB:-256} ${PN:-100} > "$OUT/phase.log" 2>&1
tail -3 "$OUT/pytest_gpu.log"; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_quick.log").read().strip().splitlines()[-1])
    print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["kernel_ms"].items()})
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/bench_quick.err").read()[-1500:])
PY
cat "$OUT/phase.log" | tail -120
