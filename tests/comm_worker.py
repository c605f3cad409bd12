"""One rank of the multi-rank hsqp_comm_* test (tests/test_comm.py): python comm_worker.py RANK WORLD WORKDIR BATCH NODES.
The ranks share the box's one GPU; the transport is the stand-in named by HSQP_RCCL_LIB (tests/stubs/rccl_standin)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, work, B, N = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    import torch
    from wb_humanoid_mpc_amd import _abi, load_model, solver
    from wb_humanoid_mpc_amd.reference import make_problem
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    lib = solver.load_library()
    idfile = os.path.join(work, "id.bin")
    ident = C.create_string_buffer(_abi.COMM_ID_BYTES)
    if rank == 0:
        assert lib.hsqp_comm_unique_id(ident) == _abi.OK, lib.hsqp_comm_create_error()
        with open(idfile + ".tmp", "wb") as f:
            f.write(ident.raw)
        os.rename(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            assert time.time() - t0 < 120, "rank 0 never published the identifier"
            time.sleep(0.05)
        ident = C.create_string_buffer(open(idfile, "rb").read(), _abi.COMM_ID_BYTES)
    c = C.c_void_p()
    assert lib.hsqp_comm_create(C.byref(c), ident, rank, world, 0) == _abi.OK, lib.hsqp_comm_create_error()
    assert lib.hsqp_comm_rank(c) == rank and lib.hsqp_comm_world(c) == world
    model = load_model()
    # the shared problem image: rank 0's model description, broadcast as bytes in device memory; every rank builds its handle from what arrives
    nbytes = C.sizeof(model.desc)
    if rank == 0:
        img = torch.frombuffer(bytearray(C.string_at(C.addressof(model.desc), nbytes)), dtype=torch.uint8).to(dev)
    else:
        img = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    assert lib.hsqp_comm_broadcast(c, img.data_ptr(), nbytes, 0) == _abi.OK, lib.hsqp_comm_last_error(c)
    model.desc = _abi.ModelDesc.from_buffer_copy(bytes(img.cpu().numpy().tobytes()))
    dt = model.sqp["dt"]
    lo, hi = C.c_int(), C.c_int()
    assert lib.hsqp_comm_shard(c, B, C.byref(lo), C.byref(hi)) == _abi.OK
    per, nloc = (B + world - 1) // world, hi.value - lo.value
    shapes = [(_abi.NX,), (N + 1, _abi.NX), (N, _abi.NU), (N + 1, _abi.NODE_PARAMS)]
    glob = [None] * 4
    if rank == 0:
        x0, x, u, par, dt = make_problem(model, n_nodes=N, batch=B, perturb=True, seed=5)
        glob = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (x0, x, u, par)]
    loc = [torch.zeros((per,) + s, dtype=torch.float64, device=dev) for s in shapes]
    for g, l, s in zip(glob, loc, shapes):
        assert lib.hsqp_comm_scatter_rows(c, g.data_ptr() if g is not None else None, l.data_ptr(), int(np.prod(s)), B, 0) == _abi.OK, lib.hsqp_comm_last_error(c)
    xs = torch.zeros((per, N + 1, _abi.NX), dtype=torch.float64, device=dev)
    us = torch.zeros((per, N, _abi.NU), dtype=torch.float64, device=dev)
    ms = np.array([float(rank), 10.0 - rank, 0.0])
    if nloc > 0:
        s = HipSqpSolver(model, max_nodes=N, max_batch=per)
        s.upload_device(nloc, N, dt, *[t.data_ptr() for t in loc])
        s.iterate(1, take_step=True, kkt=True)
        s.download_device(x_ptr=xs.data_ptr(), u_ptr=us.data_ptr())
        ms[2] = s.kernel_ms()["total"]
        s.close()
    gx = torch.zeros((B, N + 1, _abi.NX), dtype=torch.float64, device=dev) if rank == 0 else None
    gu = torch.zeros((B, N, _abi.NU), dtype=torch.float64, device=dev) if rank == 0 else None
    assert lib.hsqp_comm_gather_rows(c, xs.data_ptr(), gx.data_ptr() if rank == 0 else None, (N + 1) * _abi.NX, B, 0) == _abi.OK, lib.hsqp_comm_last_error(c)
    assert lib.hsqp_comm_gather_rows(c, us.data_ptr(), gu.data_ptr() if rank == 0 else None, N * _abi.NU, B, 0) == _abi.OK, lib.hsqp_comm_last_error(c)
    assert lib.hsqp_comm_max(c, ms.ctypes.data_as(C.POINTER(C.c_double)), 3) == _abi.OK, lib.hsqp_comm_last_error(c)
    assert ms[0] == world - 1 and ms[1] == 10.0 and ms[2] > 0.0, ms
    assert lib.hsqp_comm_barrier(c) == _abi.OK
    res = {"rank": rank, "shard": [lo.value, hi.value]}
    if rank == 0:
        ref_solver = HipSqpSolver(model, max_nodes=N, max_batch=B)
        ref = ref_solver.run(x0, x, u, par, dt)
        ref_solver.close()
        res["x_equal"] = bool(np.array_equal(gx.cpu().numpy(), ref["x"]))
        res["u_equal"] = bool(np.array_equal(gu.cpu().numpy(), ref["u"]))
        res["finite"] = bool(np.isfinite(ref["x"]).all())
    lib.hsqp_comm_destroy(c)
    with open(os.path.join(work, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
