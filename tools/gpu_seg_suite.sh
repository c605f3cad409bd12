#!/bin/bash
# two-level sweep: GPU tests + the strong leg of bench.py at 32 instances per GPU (the shard of BASELINE config 4 on 8 GPUs)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "two_level or scan or until_converged" -s 2>&1 | tail -25 > gpurun_out/seg_tests.log
timeout 600 python bench.py --force-strong --global-batch 32 --no-cpu-baseline --steps 20 > gpurun_out/bench_strong32.json 2> gpurun_out/bench_strong32.err
tail -5 gpurun_out/seg_tests.log; python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_strong32.json").read().strip().splitlines()[-1])
s = d.get("strong_scaling") or d
print({k: s[k] for k in ("ms_per_step", "kernel_ms", "gathered_solution_equals_single_gpu_solve") if k in s})
print(s.get("two_level_sweep"))
PY
