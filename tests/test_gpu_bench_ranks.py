"""bench.py's N-rank code path on a ONE-GPU box: HSQP_DIST_BACKEND=gloo lets the ranks share the device and stages the collectives through
the host (RCCL refuses two ranks on one GPU), so the whole `--gpus 2` flow the driver launches — image broadcast, scatter, per-rank solve,
gather, equality with the single-GPU solve, the weak leg, ONE JSON line from rank 0 — runs here.  A logic check, not a measurement."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_rank_bench_line_on_one_gpu():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSQP_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--global-batch", "12", "--batch", "6", "--nodes", "16"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "strong" and d["steps"] == 2 and d["warmup"] == 1
    s = d["strong_scaling"]
    assert s["global_batch"] == 12 and s["batch_per_gpu"] == 6 and s["gathered_solution_equals_single_gpu_solve"] is True
    assert d["value"] == s["value"] and d["weak_scaling"]["global_batch"] == 12 and d["weak_scaling"]["batch_per_gpu"] == 6
    assert d["kkt_over_max_1_g_inf"] <= 1e-9
