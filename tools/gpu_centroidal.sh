#!/bin/bash
# One short GPU call for the centroidal formulation: parity tests, kernel times, rocprofv3 kernel stats (each step bounded).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
timeout 170 python -m pytest tests/test_gpu_centroidal.py tests/test_device_params.py -m gpu -q 2>&1 | tail -40 > "$OUT/pytest_cent.log"
tail -5 "$OUT/pytest_cent.log"
timeout 60 python tools/cent_timing.py > "$OUT/cent_timing.log" 2>&1
tail -4 "$OUT/cent_timing.log"
timeout 60 python -m pytest tests/test_gpu_parity.py -q -k "golden or lq_blocks" 2>&1 | tail -5 > "$OUT/pytest_wb_quick.log"
tail -2 "$OUT/pytest_wb_quick.log"
export TMPDIR=/tmp
(cd /tmp && timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_cent" -o cent -- python "$OLDPWD/tools/cent_timing.py" 2 > "$OUT/prof_cent.log" 2>&1)
find "$OUT/prof_cent" -name "*kernel_stats*" | head -2 | xargs -r head -12
timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_wb_quick.log" 2> "$OUT/bench_wb_quick.err"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_wb_quick.log").read().strip().splitlines()[-1])
    print("WB value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["kernel_ms"].items()})
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/bench_wb_quick.err").read()[-1500:])
PY
