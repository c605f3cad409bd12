#!/bin/bash
# BASELINE configs 3 and 5 on one GPU (config 4 is the default bench line).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python bench.py --batch 1 --nodes 100 --no-perturb --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_cfg3.log 2> gpurun_out/bench_cfg3.err
timeout 900 python bench.py --batch 1024 --nodes 200 --gait slow_walk --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cfg5.log 2> gpurun_out/bench_cfg5.err
for f in gpurun_out/bench_cfg3.log gpurun_out/bench_cfg5.log; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["kernel_ms"].items()}, "kkt", d["kkt_residual_max"])
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done
tail -3 gpurun_out/bench_cfg5.err
