#!/usr/bin/env python3
"""bench.py — SQP iterations/sec of the G1 whole-body MPC (N = 100) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched as
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU).
One "step" = one SQP iteration of every resident MPC instance: LQ approximation at all nodes,
equality projection, Riccati QP, full step (alpha = 1) and the performance index before/after
(SURVEY.md §8d).  Inputs are resident in HBM before the timed region (hsqp_upload); nothing is
skipped inside it.  Rank 0 prints ONE JSON line.

Workload: whole-body G1, N = 100 nodes, dt = 0.035 s, gait `walk`,
v_cmd = (0.3, 0, 0.7925, 0), 256 perturbed instances PER GPU (numpy PCG64 seed 20250808 + rank),
cold-start trajectory.  Weak scaling: per-GPU work is fixed, instances are independent, there is no
data-path collective (RCCL is used only for the barrier / result gather around the timed region).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

# algorithmic flops per node (SURVEY.md §8d / BASELINE.md §5; dense count, mul+add, no symmetry credit)
def node_flops(NX, NU, NE, NUT, MROWS):
    f_rk4 = 6 * NX * NX * (NX + NU)
    f_gn = 2 * (2 * MROWS * (NX + NU) ** 2 + 2 * MROWS * (NX + NU))
    f_proj = (2 * NU * NE ** 2 + 2 * NX * NU * NUT + 2 * NX ** 2 * NU + 2 * NU ** 2 * NUT + 2 * NUT ** 2 * NU + 2 * NU ** 2 * NX +
              2 * NUT * NU * NX + 4 * NX ** 2 * NU + 2 * NU ** 2 * NX)
    f_ric = 7.0 / 3.0 * NX ** 3 + 4 * NX ** 2 * NUT + 2 * NX * NUT ** 2 + NUT ** 3 / 3.0
    return f_rk4, f_gn, f_proj, f_ric


F_RK4, F_GN, F_PROJ, F_RIC = node_flops(58, 35, 12, 23, 18)      # whole-body double support: 4.62 Mflop per node
F_NODE = F_RK4 + F_GN + F_PROJ + F_RIC
BYTES_NODE = 404 * 1024          # unfused dataflow bytes per node
CENT_F = node_flops(35, 35, 12, 23, 12)                          # centroidal stance: 1.65 Mflop per node
CENT_BYTES_NODE = 201 * 1024
PEAK_FP64_TFLOPS = 78.6          # = FP32 vector/matrix peak 157.3 TF / 2 (MI355X_MICROARCH.md chip table; FP64 runs at half the FP32 rate)
PEAK_HBM_TBS = 8.0               # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 measured)


def pmc_traffic(kernel_key):
    """HBM bytes per launch of one kernel from the committed rocprofv3 --pmc passes of this same command
    (tools/gpu_pmc.sh -> profiles/r01_pmc_summary.json; FETCH_SIZE doubled per MI355X_MICROARCH.md).  None if absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    try:
        with open(path) as fh:
            k = json.load(fh)["kernels"][kernel_key]
        return k["FETCH_SIZE"]["mean"] + k["WRITE_SIZE"]["mean"]
    except Exception:
        return None


def cpu_baseline_centroidal(model, n_nodes, seed):
    """Centroidal workload: the CPU oracle's centroidal SQP iteration (oracle/centroidal.hpp, kind 'port'), node-parallel LQ on
    OpenMP threads + serial Riccati, single instance, ~10 s; all host cores and 4 threads (the reference's nThreads)."""
    from hsqp_oracle import Oracle
    from wb_humanoid_mpc_amd.reference import make_centroidal_problem
    cores = os.cpu_count() or 1
    x0, x, u, par, dt = make_centroidal_problem(model, n_nodes=n_nodes, batch=1)
    oracle = Oracle(model)

    def leg(threads, t_min):
        oracle.cent_sqp_iteration(dt, x0[0], x[0], u[0], par[0], threads=threads)
        t0, done = time.perf_counter(), 0
        while done < 1 or time.perf_counter() - t0 < t_min:
            oracle.cent_sqp_iteration(dt, x0[0], x[0], u[0], par[0], threads=threads)
            done += 1
        return done, time.perf_counter() - t0

    omp = min(16, cores)
    done, wall = leg(omp, 8.0)
    done4, wall4 = leg(4, 5.0)
    return {"value": done / wall, "unit": "SQP iters/s", "cores": omp, "kind": "port",
            "sample": f"{done} single-instance iterations (centroidal, N={n_nodes}) in {wall:.1f} s on {omp} OpenMP threads (node-parallel LQ, "
                      "serial Riccati on the padded 58-state layout); forward-mode dual-number oracle, not the reference's CppAD/HPIPM build",
            "value_4_threads": done4 / wall4, "sample_4_threads": f"{done4} iterations in {wall4:.1f} s on 4 threads"}


def cpu_baseline(model, n_nodes, seed):
    """The CPU oracle (oracle/oracle.cpp, kind 'port') on a bounded sample of the same workload: single-instance SQP
    iterations of the same perturbed walk inputs, node-parallel LQ on OpenMP threads, serial Riccati.  Timed at all host
    cores (`value`) and at 4 threads (the reference's nThreads, g1_wb_mpc/config/mpc/task.info:79)."""
    from hsqp_oracle import Oracle
    from wb_humanoid_mpc_amd.reference import make_problem
    cores = os.cpu_count() or 1
    n_inst = 4
    x0, x, u, par, dt = make_problem(model, n_nodes=n_nodes, batch=n_inst, perturb=True, seed=seed)
    oracle = Oracle(model)

    def leg(workers, omp_threads, t_min):
        """`workers` host threads, each running single-instance iterations with `omp_threads` OpenMP threads (ctypes releases the GIL)."""
        from concurrent.futures import ThreadPoolExecutor
        oracle.sqp_iteration(dt, x0[0], x[0], u[0], par[0], threads=omp_threads, want_perf=True)  # warm-up
        t0 = time.perf_counter()

        def work(wid):
            done = 0
            while done < 1 or time.perf_counter() - t0 < t_min:
                b = (wid + done) % n_inst
                oracle.sqp_iteration(dt, x0[b], x[b], u[b], par[b], threads=omp_threads, want_perf=True)
                done += 1
            return done

        with ThreadPoolExecutor(max_workers=workers) as ex:
            done = sum(ex.map(work, range(workers)))
        return done, time.perf_counter() - t0

    omp = min(8, cores)
    workers = max(1, cores // omp)
    done, wall = leg(workers, omp, 10.0)
    done4, wall4 = leg(1, 4, 6.0)
    return {"value": done / wall, "unit": "SQP iters/s", "cores": workers * omp, "kind": "port",
            "sample": f"{done} single-instance iterations (N={n_nodes}, same perturbed walk inputs) in {wall:.1f} s: {workers} concurrent instances x "
                      f"{omp} OpenMP threads (node-parallel LQ, serial Riccati); forward-mode dual-number oracle (93 tangents), not the "
                      f"reference's CppAD/HPIPM build",
            "value_4_threads": done4 / wall4, "sample_4_threads": f"{done4} iterations in {wall4:.1f} s on 4 threads (the reference's nThreads)"}


def cpu_kernel_sources_on_host(model, n_nodes, seed, cent=False):
    """Second, stronger CPU figure (reported beside `cpu_baseline`, never instead of it): the SAME kernel sources
    (wb_humanoid_mpc_amd/csrc/*.h: analytic derivatives, structured RK4 chain, projection, Riccati) compiled for the host
    with a one-thread context (tests/hostemu, g++ -O2), one instance per host thread, all cores."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    from wb_humanoid_mpc_amd.reference import make_problem
    path = os.path.join(ROOT, "tests", "hostemu", "libhsqp_hostemu.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    lib.emu_create.restype = C.c_void_p
    err = C.create_string_buffer(256)
    h = C.c_void_p(lib.emu_create(C.byref(model.desc), err, 256))
    if not h.value:
        return None
    cores = os.cpu_count() or 1
    n_inst = 8
    if cent:   # centroidal: every lane of the LQ kernel is emulated one after the other (70 tangent lanes per node), serial Riccati sweep
        from wb_humanoid_mpc_amd.reference import make_centroidal_problem
        x0, x, u, par, dt = make_centroidal_problem(model, n_nodes=n_nodes, batch=n_inst, perturb=True, seed=seed)
    else:
        x0, x, u, par, dt = make_problem(model, n_nodes=n_nodes, batch=n_inst, perturb=True, seed=seed)
    dp = C.POINTER(C.c_double)
    t0 = time.perf_counter()

    def work(wid):
        P = lambda a: a.ctypes.data_as(dp)  # noqa: E731
        xn, un, dx, du = np.zeros_like(x[0]), np.zeros_like(u[0]), np.zeros_like(x[0]), np.zeros_like(u[0])
        kkt, pb, pa = np.zeros(2), np.zeros(3), np.zeros(3)
        done = 0
        while done < 1 or time.perf_counter() - t0 < 8.0:
            b = (wid + done) % n_inst
            lib.emu_sqp_iteration(h, n_nodes, C.c_double(dt), P(x0[b]), P(x[b]), P(u[b]), P(par[b]), P(xn), P(un), P(dx), P(du), P(kkt), P(pb), P(pa), None)
            done += 1
        return done

    with ThreadPoolExecutor(max_workers=cores) as ex:
        done = sum(ex.map(work, range(cores)))
    wall = time.perf_counter() - t0
    return {"value": done / wall, "unit": "SQP iters/s", "cores": cores, "kind": "kernel sources compiled for the host (tests/hostemu)",
            "sample": f"{done} single-instance iterations (N={n_nodes}) in {wall:.1f} s, one instance per thread on {cores} threads, "
                      "each iteration incl. the KKT check and the performance pass"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--formulation", default="wb", choices=["wb", "centroidal"],
                    help="wb: BASELINE configs 3-5 (default, the metric's workload); centroidal: configs 1-2 (default batch 1)")
    ap.add_argument("--batch", type=int, default=None, help="MPC instances per GPU (default 256; centroidal: 1)")
    ap.add_argument("--nodes", type=int, default=100)
    ap.add_argument("--gait", default="walk", help="gait of the synthetic schedule (config 5: slow_walk)")
    ap.add_argument("--no-perturb", action="store_true", help="config 3: the unperturbed initial state")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from wb_humanoid_mpc_amd.distributed import Group, aggregate_throughput, env_rank, shard_seed
    rank, local_rank, world = env_rank()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # torch FIRST: its bundled libamdhip64.so.7 must be the one HIP runtime of the process; libhsqp_hip.so then
    # binds to it by soname (measured on the GPU box: the other order leaves torch.cuda unavailable)
    import torch
    from wb_humanoid_mpc_amd import load_model
    from wb_humanoid_mpc_amd.reference import BENCH_SEED, make_problem
    from wb_humanoid_mpc_amd.solver import HipSqpSolver, load_library
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path)")
    torch.cuda.set_device(local_rank)
    torch.zeros(1, device="cuda")  # initialise the HIP runtime through torch before the library touches it
    load_library()
    group = Group(world, backend="nccl", device=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm

    cent = args.formulation == "centroidal"
    model = load_model(formulation=args.formulation)
    B, N = args.batch if args.batch else (1 if cent else 256), args.nodes
    if cent:
        from wb_humanoid_mpc_amd.reference import make_centroidal_problem
        x0, x, u, par, dt = make_centroidal_problem(model, n_nodes=N, batch=B, gait=args.gait, perturb=B > 1 and not args.no_perturb,
                                                    seed=shard_seed(BENCH_SEED, rank))
    else:
        x0, x, u, par, dt = make_problem(model, n_nodes=N, batch=B, gait=args.gait, perturb=not args.no_perturb, seed=shard_seed(BENCH_SEED, rank))
    solver = HipSqpSolver(model, max_nodes=N, max_batch=B, device=local_rank)
    solver.upload(x0, x, u, par, dt)      # inputs resident in HBM before the timed region

    def sync():
        torch.cuda.synchronize()
        group.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        solver.iterate(1, take_step=False)
    sync()
    kms = np.zeros(5)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        solver.iterate(1, take_step=False)   # blocking: synchronises the library's stream
        k = solver.kernel_ms()
        kms += [k["lq"], k["project"], k["riccati"], k["step_perf"], k["total"]]
    sync()
    elapsed = time.perf_counter() - t0
    kms /= args.steps
    solver.iterate(1, take_step=False, kkt=True)   # outside the timed region: KKT residual of the QP for the report
    out = solver.download()
    kkt = float(np.max(out["kkt"]))
    # outside the timed region: the same iteration through hsqp_solve with HOST buffers (upload + iterate + download over PCIe)
    t1 = time.perf_counter()
    for _ in range(2):
        solver.run(x0, x, u, par, dt)
    pcie_ms = 1e3 * (time.perf_counter() - t1) / 2

    elapsed, kkt = group.max([elapsed, kkt])   # max over ranks

    if rank == 0:
        value = aggregate_throughput([B] * world, args.steps, elapsed)
        cfg = {(100, 256): "4", (100, 1): "3", (200, 1024): "5"}.get((N, B), "4-like")
        if cent:
            cfg = {(20, 1): "1", (100, 1): "2"}.get((N, B), "2-like")
        nodes = B * N
        f_rk4, f_gn, f_proj, f_ric = CENT_F if cent else (F_RK4, F_GN, F_PROJ, F_RIC)
        f_node, bytes_node = f_rk4 + f_gn + f_proj + f_ric, CENT_BYTES_NODE if cent else BYTES_NODE
        # dominant kernel of one step, its algorithmic work and measured duration (HIP events on the library's stream)
        kern = {"lq_approximation(k_lq)": (kms[0], f_rk4 + f_gn, "k_lq_cent" if cent else "k_lq<true>"), "projection(k_project)": (kms[1], f_proj, "k_project"),
                ("backward_sweep(k_scan_*: parallel-in-time scan)" if cent and B <= 2 and N >= 48 else "riccati(k_riccati)"): (kms[2], f_ric, "k_riccati")}
        dom = max(kern, key=lambda n: kern[n][0])
        dom_ms, dom_flops, dom_key = kern[dom]
        traffic = pmc_traffic(dom_key) if (B, N) == (256, 100) and not cent else None
        ach_tf = nodes * dom_flops / (dom_ms * 1e-3) / 1e12
        step_tf = nodes * f_node / (elapsed / args.steps) / 1e12
        step_tbs = nodes * bytes_node / (elapsed / args.steps) / 1e12
        res = {
            "metric": "SQP iters/sec (G1 centroidal MPC)" if cent else "SQP iters/sec (G1 WB-MPC, N=100)", "value": value, "unit": "SQP iters/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE config {cfg}: G1 {'centroidal' if cent else 'whole-body'} MPC, N={N}, dt={dt}, gait {args.gait}, {B} {'perturbed ' if not args.no_perturb else ''}instances per GPU, "
                                   "1 SQP iteration per step (LQ + projection + Riccati + full step + performance index), cold-start trajectory",
                       "batch_per_gpu": B, "global_batch": B * world, "nodes": N, "parallelism": f"batch-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": ach_tf, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tf / PEAK_FP64_TFLOPS, "traffic": traffic,
                         "traffic_note": "HBM bytes per launch of the dominant kernel (FETCH_SIZE x2 + WRITE_SIZE) from the committed rocprofv3 --pmc "
                                         "passes of this command (profiles/r01_pmc_summary.json); null for other shapes",
                         "note": "achieved = ALGORITHMIC (dense-count) flops of the dominant kernel / its HIP-event duration; the kernels exploit "
                                 "the flow map's structure and execute fewer flops than the dense count (DESIGN.md)",
                         "whole_step_algorithmic_TFLOPs": step_tf, "whole_step_frac_fp64": step_tf / PEAK_FP64_TFLOPS,
                         "whole_step_unfused_TBs": step_tbs, "whole_step_frac_hbm": step_tbs / PEAK_HBM_TBS},
            "kernel_ms": {"lq": kms[0], "project": kms[1], "riccati": kms[2], "step_perf": kms[3], "sum": kms[4]},
            "kkt_residual_max": kkt,
            "pcie_inclusive": {"ms_per_step": pcie_ms, "value": B / (pcie_ms * 1e-3), "unit": "SQP iters/s per GPU",
                               "note": "hsqp_solve with host buffers (upload 36 MB + iterate incl. KKT check + download 39 MB at B=256, N=100); never `value`"},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_centroidal(model, N, BENCH_SEED) if cent else cpu_baseline(model, N, BENCH_SEED)
            res["speedup_vs_cpu_baseline"] = value / res["cpu_baseline"]["value"]
            host = cpu_kernel_sources_on_host(model, N, BENCH_SEED, cent=cent)
            if host:
                res["cpu_kernel_sources_on_host"] = host
                res["speedup_vs_kernel_sources_on_host"] = value / host["value"]
        print(json.dumps(res))
    solver.close()
    group.close()


if __name__ == "__main__":
    main()
