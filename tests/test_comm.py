"""hsqp_comm_* (include/hsqp.h): the batch axis over the GPUs of one node behind the C ABI — a C++ host's counterpart of
wb_humanoid_mpc_amd/distributed.py.  One GPU per test box: the communicator of a world of ONE rank goes through the real RCCL for its creation, the
broadcast and the reductions (scatter / gather of the root's own block are device copies).  Two ranks cannot share a device under RCCL, so the
N-RANK paths — the grouped ncclSend / ncclRecv loops of scatter and gather, the broadcast and the reduction across processes — run here with 2 and 3
rank PROCESSES that share the box's one GPU over a stand-in transport bound through HSQP_RCCL_LIB (tests/stubs/rccl_standin: the ten entry points of
RCCL's public C interface over POSIX shared memory), uneven split included, gathered solution bitwise equal to the single-process hsqp_solve."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from wb_humanoid_mpc_amd import _abi, solver
from wb_humanoid_mpc_amd.reference import make_problem

pytestmark = pytest.mark.gpu


@pytest.fixture()
def comm():
    lib = solver.load_library()
    ident = C.create_string_buffer(_abi.COMM_ID_BYTES)
    assert lib.hsqp_comm_unique_id(ident) == _abi.OK, lib.hsqp_comm_create_error()
    c = C.c_void_p()
    assert lib.hsqp_comm_create(C.byref(c), ident, 0, 1, 0) == _abi.OK, lib.hsqp_comm_create_error()
    yield lib, c
    lib.hsqp_comm_destroy(c)


def test_single_rank_communicator_round_trip(comm):
    import torch
    lib, c = comm
    assert lib.hsqp_comm_rank(c) == 0 and lib.hsqp_comm_world(c) == 1
    lo, hi = C.c_int(), C.c_int()
    assert lib.hsqp_comm_shard(c, 37, C.byref(lo), C.byref(hi)) == _abi.OK and (lo.value, hi.value) == (0, 37)
    dev = torch.device("cuda:0")
    g = torch.arange(37 * 11, dtype=torch.float64, device=dev).reshape(37, 11).contiguous()
    loc = torch.zeros_like(g)
    assert lib.hsqp_comm_scatter_rows(c, g.data_ptr(), loc.data_ptr(), 11, 37, 0) == _abi.OK, lib.hsqp_comm_last_error(c)
    assert torch.equal(loc, g)
    back = torch.zeros_like(g)
    assert lib.hsqp_comm_gather_rows(c, loc.data_ptr(), back.data_ptr(), 11, 37, 0) == _abi.OK, lib.hsqp_comm_last_error(c)
    assert torch.equal(back, g)
    img = torch.full((1000,), 3.25, dtype=torch.float64, device=dev)
    assert lib.hsqp_comm_broadcast(c, img.data_ptr(), img.numel() * 8, 0) == _abi.OK, lib.hsqp_comm_last_error(c)
    assert bool((img == 3.25).all())
    v = np.array([1.5, -2.0, 7.0])
    assert lib.hsqp_comm_max(c, v.ctypes.data_as(C.POINTER(C.c_double)), 3) == _abi.OK, lib.hsqp_comm_last_error(c)
    assert np.array_equal(v, [1.5, -2.0, 7.0])
    assert lib.hsqp_comm_barrier(c) == _abi.OK
    # argument errors name themselves
    assert lib.hsqp_comm_scatter_rows(c, None, loc.data_ptr(), 11, 37, 0) == _abi.ERR_BAD_ARG
    assert b"scatter" in lib.hsqp_comm_last_error(c)
    assert lib.hsqp_comm_broadcast(c, img.data_ptr(), 8, 1) == _abi.ERR_BAD_ARG      # root outside the world


def test_scattered_shard_through_the_device_entry_points_equals_the_host_solve(comm, model):
    """The data path a C++ host runs per solve: scatter every array of hsqp_problem -> hsqp_upload_device -> one iteration -> hsqp_download_device ->
    gather: the same numbers, bit for bit, as hsqp_solve on host buffers."""
    import torch
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    lib, c = comm
    B, N = 6, 12
    x0, x, u, par, dt = make_problem(model, n_nodes=N, batch=B, perturb=True, seed=5)
    dev = torch.device("cuda:0")
    s = HipSqpSolver(model, max_nodes=N, max_batch=B)
    try:
        ref = s.run(x0, x, u, par, dt)
        glob = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (x0, x, u, par)]
        loc = [torch.zeros_like(t) for t in glob]
        for gt, lt in zip(glob, loc):
            assert lib.hsqp_comm_scatter_rows(c, gt.data_ptr(), lt.data_ptr(), gt[0].numel(), B, 0) == _abi.OK, lib.hsqp_comm_last_error(c)
        s.upload_device(B, N, dt, *[t.data_ptr() for t in loc])
        s.iterate(1, take_step=True, kkt=True)
        xs, us = torch.zeros_like(glob[1]), torch.zeros_like(glob[2])
        s.download_device(x_ptr=xs.data_ptr(), u_ptr=us.data_ptr())
        gx, gu = torch.zeros_like(xs), torch.zeros_like(us)
        assert lib.hsqp_comm_gather_rows(c, xs.data_ptr(), gx.data_ptr(), xs[0].numel(), B, 0) == _abi.OK
        assert lib.hsqp_comm_gather_rows(c, us.data_ptr(), gu.data_ptr(), us[0].numel(), B, 0) == _abi.OK
        assert np.array_equal(gx.cpu().numpy(), ref["x"]) and np.array_equal(gu.cpu().numpy(), ref["u"])
    finally:
        s.close()


HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def standin(tmp_path_factory):
    out = tmp_path_factory.mktemp("rccl_standin") / "librccl_standin.so"
    src = os.path.join(HERE, "stubs", "rccl_standin", "rccl_standin.cpp")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-shared", "-fPIC", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src,
                           "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-o", str(out)])
    return str(out)


@pytest.mark.parametrize("world", [2, 3])
def test_scatter_solve_gather_over_rank_processes(tmp_path, standin, world):
    """hsqp_comm.hip with world > 1 (round 5 review, item 5): `world` processes, one communicator each, 5 instances split unevenly (3 + 2, 2 + 2 + 1):
    broadcast of the problem image, scatter of every hsqp_problem array (the root's grouped sends, the others' receives), hsqp_upload_device ->
    iterate -> hsqp_download_device per rank, gather (the root's grouped receives), max-reduction and barrier.  Rank 0 checks the gathered
    trajectories against its own single-process hsqp_solve of the whole batch: bit for bit (instances are independent)."""
    B, N = 5, 10
    env = dict(os.environ, HSQP_RCCL_LIB=standin, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "comm_worker.py"), str(r), str(world), str(tmp_path), str(B), str(N)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(outs)
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    per = (B + world - 1) // world
    assert [r["shard"] for r in res] == [[min(r * per, B), min(r * per + per, B)] for r in range(world)]
    assert res[0]["finite"] and res[0]["x_equal"] and res[0]["u_equal"], res[0]
