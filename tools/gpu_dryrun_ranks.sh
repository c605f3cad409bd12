#!/bin/bash
# the N-rank code path of bench.py dry-run on a ONE-GPU box (HSQP_DIST_BACKEND=gloo: ranks share the device, collectives staged through the
# host): a logic check of what the driver launches on 2 / 4 / 8 GPUs — not a measurement
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for n in 2 8; do
  HSQP_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 3 --warmup 1 \
    > gpurun_out/dryrun_$n.log 2> gpurun_out/dryrun_$n.err
  echo "ranks $n rc=$?"
  python - gpurun_out/dryrun_$n.log <<'PY'
import json, sys
try:
    lines = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"{len(lines)} JSON lines"
    d = json.loads(lines[0])
    s = d["strong_scaling"]
    print("   value", round(d["value"], 1), d["scaling"], "n_gpus", d["n_gpus"], "rccl_ranks", d["rccl_ranks"], "batch/gpu", s["batch_per_gpu"], "gathered == single", s["gathered_solution_equals_single_gpu_solve"],
          "two-level", (s.get("two_level_sweep") or {}).get("ms_per_step"), "weak", round(d["weak_scaling"]["value"], 1), "n1_eq", round(d["n1_equivalent"]["value"], 1))
except Exception as e:
    print("   failed:", e); print(open(sys.argv[1].replace(".log", ".err")).read()[-1500:])
PY
done
