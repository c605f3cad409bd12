"""The N > 1 paths of bench.py on CPU (two and three gloo ranks):
  * weak scaling: per-rank shards, barrier + max-over-ranks timing, gather of the per-rank summaries;
  * the north star's data path: broadcast of the shared problem image, scatter of the instance blocks from rank 0, per-rank solve
    (the CPU oracle stands in for the per-rank GPU solve), gather of the solutions — the gathered solution must equal rank 0's own
    solve of the whole batch bit for bit, also when the batch does not divide evenly."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import ctypes, json, os, sys, time
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle"))
    import numpy as np, torch
    from wb_humanoid_mpc_amd import _abi, load_model
    from wb_humanoid_mpc_amd.distributed import BatchShards, Group, aggregate_throughput, broadcast_image, env_rank, shard_range, shard_seed
    from wb_humanoid_mpc_amd.reference import BENCH_SEED, make_problem
    from hsqp_oracle import Oracle
    rank, local_rank, world = env_rank()
    g = Group(world, backend="gloo")
    model = load_model()
    # ---- weak scaling: own shard per rank
    x0, x, u, par, dt = make_problem(model, n_nodes=4, batch=2, perturb=True, seed=shard_seed(BENCH_SEED, rank))
    oracle = Oracle(model)                      # CPU stand-in for the per-rank GPU solve
    g.barrier(); t0 = time.perf_counter()
    kkt = max(oracle.sqp_iteration(dt, x0[b], x[b], u[b], par[b], want_perf=False)["kkt"].max() for b in range(2))
    g.barrier(); elapsed = time.perf_counter() - t0 + 0.01 * rank
    emax, kmax = g.max([elapsed, kkt])
    table = g.gather([rank, x0[0, 6], elapsed])
    # ---- the north star's data path: rank 0 owns the global batch
    GB, N = 5, 3
    image = bytes(ctypes.string_at(ctypes.addressof(model.desc), ctypes.sizeof(model.desc))) if rank == 0 else None
    image = broadcast_image(g, image)
    desc = _abi.ModelDesc.from_buffer_copy(image)
    same_image = image == bytes(ctypes.string_at(ctypes.addressof(model.desc), ctypes.sizeof(model.desc)))
    if rank == 0:
        gx0, gx, gu, gpar, dt = make_problem(model, n_nodes=N, batch=GB, perturb=True, seed=7)
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        glob = [T(gx0), T(gx), T(gu), T(gpar)]
    else:
        glob = [None] * 4
    sh = BatchShards(g, GB)
    lx0 = sh.scatter(glob[0], (_abi.NX,)).numpy(); lx = sh.scatter(glob[1], (N + 1, _abi.NX)).numpy()
    lu = sh.scatter(glob[2], (N, _abi.NU)).numpy(); lpar = sh.scatter(glob[3], (N + 1, _abi.NODE_PARAMS)).numpy()
    sol_x, sol_u, sol_k = np.zeros_like(lx), np.zeros_like(lu), np.zeros((sh.per, 2))
    for b in range(sh.per):                     # padded rows are solved too (and dropped by the gather)
        r = oracle.sqp_iteration(0.035, lx0[b], lx[b], lu[b], lpar[b], want_perf=False)
        sol_x[b], sol_u[b], sol_k[b] = r["x"], r["u"], r["kkt"]
    gsx, gsu, gsk = sh.gather(torch.from_numpy(sol_x)), sh.gather(torch.from_numpy(sol_u)), sh.gather(torch.from_numpy(sol_k))
    if rank == 0:
        ref_x = np.stack([oracle.sqp_iteration(0.035, gx0[b], gx[b], gu[b], gpar[b], want_perf=False)["x"] for b in range(GB)])
        ref_u = np.stack([oracle.sqp_iteration(0.035, gx0[b], gx[b], gu[b], gpar[b], want_perf=False)["u"] for b in range(GB)])
        print(json.dumps({"world": world, "emax": emax, "kmax": kmax, "own": elapsed, "table": table.tolist(),
                          "value": aggregate_throughput([2] * world, 1, emax), "ranges": [shard_range(5, world, r) for r in range(world)],
                          "same_image": same_image, "gathered_shape": list(gsx.shape),
                          "bitwise_x": bool(np.array_equal(gsx.numpy(), ref_x)), "bitwise_u": bool(np.array_equal(gsu.numpy(), ref_u)),
                          "kkt_rows": int(gsk.shape[0]), "per": sh.per, "count": sh.count}))
    g.close()
''') % (ROOT, ROOT)


def _run(tmp_path, world, port):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


def test_two_rank_gloo_run(tmp_path):
    r = _run(tmp_path, 2, 29541)
    assert r["world"] == 2
    table = np.array(r["table"])
    assert list(table[:, 0]) == [0.0, 1.0]
    assert table[0, 1] != table[1, 1]                       # different seeds -> different instances per rank
    assert abs(r["emax"] - table[:, 2].max()) < 1e-12 and r["emax"] >= r["own"]
    assert abs(r["value"] - 4.0 / r["emax"]) < 1e-9          # whole-job throughput over the slowest rank
    assert r["kmax"] < 1e-6
    assert r["ranges"] == [[0, 3], [3, 5]]
    # data path: image broadcast, scatter (3 + 2 instances, the second block padded), gather, bitwise equality with the single-rank solve
    assert r["same_image"] and r["gathered_shape"] == [5, 4, 58] and r["kkt_rows"] == 5 and r["per"] == 3 and r["count"] == 3
    assert r["bitwise_x"] and r["bitwise_u"]


def test_three_rank_gloo_run_with_an_uneven_split(tmp_path):
    r = _run(tmp_path, 3, 29547)
    assert r["world"] == 3 and r["ranges"] == [[0, 2], [2, 4], [4, 5]] and r["per"] == 2
    assert r["same_image"] and r["gathered_shape"] == [5, 4, 58] and r["bitwise_x"] and r["bitwise_u"]
