#!/usr/bin/env python3
"""bench.py — SQP iterations/sec of the G1 whole-body MPC (N = 100) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched as
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU).
One "step" = one SQP iteration of every resident MPC instance: LQ approximation at all nodes,
equality projection, Riccati QP, full step (alpha = 1) and the performance index before/after
(SURVEY.md §8d).  Inputs are resident in HBM before the timed region (hsqp_upload); nothing is
skipped inside it.  Rank 0 prints ONE JSON line.

Workload: whole-body G1, N = 100 nodes, dt = 0.035 s, gait `walk`, v_cmd = (0.3, 0, 0.7925, 0), ONE global
batch of 256 perturbed instances (numpy PCG64 seed 20250808), cold-start trajectory.  `value` is BASELINE config 4 AS WRITTEN for every
--gpus N: the 256 instances sharded 256/N per GPU ("scaling": "strong") along the north star's data path — rank 0 owns the problem:
RCCL broadcast of the shared problem image, scatter of the contiguous instance blocks into HBM, hsqp_upload_device — and timed with the
shards resident in HBM (the contract's "inputs already resident"); `collective_inclusive` repeats it with scatter + gather every step,
and rank 0 checks that the gathered solution equals its own solve of the whole batch bit for bit.  At N = 1 this is the plain
256-instance run.  For N > 1 the line also carries `weak_scaling` (256 instances PER GPU, per-rank seeds, no data-path collective:
the contract's reading of a partitioned path), `rccl_ranks` (an all-reduce of ones over the GPUs) and `n1_equivalent` (the per-GPU
rate of the weak leg, which must reproduce the N = 1 line).

Started without a torchrun environment, `--gpus N` (N > 1) re-launches itself under torch.distributed.run with N ranks.
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


PROFILE_ROUND = "r06"            # which committed PMC passes the per-kernel HBM bytes / matrix-pipe shares are read from (profiles/<round>_pmc_*)


def pmc_kernel_info():
    """Per kernel: HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE) and the matrix-pipe busy share from the COMMITTED rocprofv3 --pmc
    passes of this same command (tools/gpu_profiles.sh -> profiles/<round>_pmc_summary.json, <round>_pmc_sq_summary.json): read, not
    measured in this run — the returned provenance says so.  A summary whose kernel names are not all present in the library this run
    loads (a profile of another build) is dropped.  -> (info per kernel, provenance)"""
    from wb_humanoid_mpc_amd import build as _build
    out, prov = {}, {"source": [], "measured_in_this_run": False, "round": PROFILE_ROUND}
    try:
        lib_bytes = open(os.environ.get("HSQP_LIB", _build.LIB), "rb").read()
    except Exception:
        lib_bytes = b""

    def names_match(keys):
        ours = [k for k in keys if k.startswith("k_")]
        missing = [k for k in ours if k.split("<")[0].encode() not in lib_bytes]
        if missing or not ours:
            prov.setdefault("dropped", []).append({"missing_in_library": missing})
            return False
        return True

    try:
        path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_pmc_summary.json")
        with open(path) as fh:
            kernels = json.load(fh)["kernels"]
        if names_match(kernels):
            # a kernel launched more than once per iteration (the limb-lane LQ kernels run in two node ranges, k_perf_reduce twice): its bytes per
            # ITERATION = mean per launch x launches per iteration, counted against k_project's dispatches (one per iteration)
            ref = max(1, int(kernels.get("k_project", {}).get("FETCH_SIZE", {}).get("dispatches", 1)))
            for k, v in kernels.items():
                per_it = max(1, round(v["FETCH_SIZE"].get("dispatches", ref) / ref)) if k.startswith("k_") else 1
                out.setdefault(k, {})["hbm_bytes"] = per_it * (v["FETCH_SIZE"]["mean"] + v["WRITE_SIZE"]["mean"])
                out[k]["launches_per_iteration"] = per_it
            prov["source"].append(os.path.relpath(path, ROOT))
    except Exception:
        pass
    try:
        path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_pmc_sq_summary.json")
        with open(path) as fh:
            kernels = json.load(fh)
        if names_match(kernels):
            for k, v in kernels.items():
                out.setdefault(k, {})["mfma_busy"] = v.get("mfma_busy")
            prov["source"].append(os.path.relpath(path, ROOT))
    except Exception:
        pass
    return out, prov


def gpu_clocks(device):
    """Shader / memory clock (MHz) of the device as rocm-smi reports them right now, or None: printed before and after the timed region so that a
    box that runs at other clocks (round 2: one box 3 - 50 % slower) is visible in the record."""
    import re
    import subprocess
    try:
        txt = subprocess.run(["rocm-smi", "-d", str(device), "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        out = {}
        for key in ("sclk", "mclk", "fclk"):
            m = re.search(key + r"\s+clock level:?\s*\d*:?\s*\((\d+)Mhz\)", txt, re.I)
            if m:
                out[key + "_mhz"] = int(m.group(1))
        return out or None
    except Exception:  # noqa: BLE001
        return None


def strong_prediction(batch_per_gpu, nodes, riccati):
    """The committed one-GPU shard timings (profiles/r06_strong_prediction.json, tools/strong_prediction.py: the batch ONE GPU of an N-GPU run of config 4
    holds, timed on one GPU with the collectives degenerate to copies) -> what `--gpus N` should measure per step.  Printed next to the measured value
    so that the first real multi-GPU run confirms or refutes it."""
    path = os.path.join(ROOT, "profiles", "r06_strong_prediction.json")
    try:
        with open(path) as f:
            pred = json.load(f)
        row = pred["config4_shards"].get(str(batch_per_gpu)) if nodes == pred.get("nodes", 100) else None
        if not row:
            return None
        key = "two_level_sweep" if riccati == "segmented" else "exact"
        if key not in row:
            return None
        return {"predicted_ms_per_step": row[key]["ms_per_step"], "from": os.path.relpath(path, ROOT), "measured_on": pred.get("measured_on"),
                "note": "one-GPU timing of this shard size (resident shards, default exact sweep unless --riccati segmented); RCCL moves tens of microseconds per step on top"}
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline(model, n_nodes, seed, cent=False):
    """The timed CPU baseline (kind "port", oracle/cpu_baseline.cpp): this repository's arithmetic for the same iteration — the kernel
    sources' analytic derivatives, structured RK4 chain, QR projection, Riccati recursion, value pass — built HERE with
    g++ -O3 -march=native -fopenmp, on a bounded sample of the same workload:
      * all host threads: one instance per thread (batch across cores), `value`;
      * 4 threads on one instance (node-parallel LQ + value pass, serial Riccati): the reference's nThreads (task.info:79).
    The forward-mode dual-number oracle (the correctness reference) is timed for a few seconds as a footnote."""
    from cpu_baseline import CpuBaseline, cpu_model, physical_cores, usable_cores
    from wb_humanoid_mpc_amd.reference import make_centroidal_problem, make_problem
    cores, hw_threads, quota = usable_cores()
    n_inst = min(256, max(2 * cores, 4))
    x0, x, u, par, dt = (make_centroidal_problem if cent else make_problem)(model, n_nodes=n_nodes, batch=n_inst, perturb=True, seed=seed)
    base = CpuBaseline(model)
    base.iterate(x0[:cores], x[:cores], u[:cores], par[:cores], dt, outer=cores, inner=1, iterations=1)      # warm-up (page in, spin up the pool)

    def leg(outer, inner, batch, t_min):
        done, t0 = 0, time.perf_counter()
        while done == 0 or time.perf_counter() - t0 < t_min:
            base.iterate(x0[:batch], x[:batch], u[:batch], par[:batch], dt, outer=outer, inner=inner, iterations=1)
            done += batch
        return done, time.perf_counter() - t0

    done, wall = leg(cores, 1, n_inst, 8.0)
    done4, wall4 = leg(1, 4, 1, 4.0)
    # does the per-core rate hold as the cores fill up?  (the all-core extrapolation below assumes it does): 1, 2, 4, 8, .. concurrent instances,
    # ~1.5 s each; efficiency = rate(c) / (c x rate(1))
    scaling = {}
    for c in [c for c in (1, 2, 4, 8, 16, 32, 64) if c < cores] + [cores]:
        d, w = (done, wall) if c == cores else leg(c, 1, max(c, 2), 1.5)
        scaling[c] = d / w
    eff = {str(c): round(scaling[c] / (c * scaling[1]), 3) for c in scaling}
    res = {"value": done / wall, "unit": "SQP iters/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
           "hardware_threads_visible": hw_threads, "cgroup_cpu_quota": quota, "value_per_core": done / wall / cores,
           "cores_note": "cores = hardware threads this process can use = min(affinity mask, cgroup cpu.max quota); the GPU boxes of this pool cap the "
                         "container at 16 CPUs of a 2 x 64-core host, and running more threads than the quota lowers the rate",
           "build": "g++ -O3 -march=native -fopenmp oracle/cpu_baseline.cpp (the kernel sources' arithmetic compiled for this host)",
           "sample": f"{done} single-instance iterations ({'centroidal' if cent else 'whole-body'}, N={n_nodes}, the bench's perturbed instances) in {wall:.1f} s: "
                     f"{cores} instances concurrently, one per usable core; each iteration = LQ + projection + serial Riccati + step + "
                     "performance index before/after (no KKT check)",
           "all_core_extrapolation": {"extrapolated": True, "factor": (physical_cores() or hw_threads) / cores,
                                      "value": done / wall / cores * (physical_cores() or hw_threads), "cores": physical_cores() or hw_threads,
                                      "value_all_hardware_threads": done / wall / cores * hw_threads,
                                      "measured_scaling": {"concurrent_instances": {str(c): round(v, 1) for c, v in scaling.items()}, "efficiency_vs_one_core": eff,
                                                           "note": "iters/s with c single-threaded instances side by side on this container's cores; the "
                                                                   "extrapolation multiplies the rate per core AT THE LARGEST c measured, so memory-bandwidth "
                                                                   "saturation up to that c is already in it"},
                                      "note": "per-core rate x physical cores of this host (linear scaling assumed; value_all_hardware_threads counts SMT siblings as cores: "
                                              "an upper bound).  BASELINE.md asks >= 10x the ALL-core baseline: compare `value` of the line with THIS number; the "
                                              "measured `cpu_baseline.value` is what the container's cgroup quota allows"},
           "value_4_threads": done4 / wall4,
           "sample_4_threads": f"{done4} iterations of one instance in {wall4:.1f} s on 4 threads (node-parallel LQ and value pass, serial Riccati): the reference's nThreads"}
    try:   # footnote: the dual-number oracle (what tests compare against), a few seconds on 16 threads
        from hsqp_oracle import Oracle
        oracle = Oracle(model)
        fn = oracle.cent_sqp_iteration if cent else oracle.sqp_iteration
        thr = min(16, cores)
        fn(dt, x0[0], x[0], u[0], par[0], threads=thr)
        t0, k = time.perf_counter(), 0
        while k == 0 or time.perf_counter() - t0 < 3.0:
            fn(dt, x0[0], x[0], u[0], par[0], threads=thr)
            k += 1
        res["footnote_dual_number_oracle"] = {"value": k / (time.perf_counter() - t0), "unit": "SQP iters/s", "threads": thr,
                                              "note": "forward-mode dual-number restatement used as the correctness oracle; not a performance baseline"}
    except Exception as e:  # noqa: BLE001
        res["footnote_dual_number_oracle"] = {"error": str(e)}
    return res


def strong_scaling_leg(args, torch, group, model_local, rank, local_rank, world, sync):
    """BASELINE config 4 as written: one global batch of 256 instances, 256 / world per GPU, along the north star's data path."""
    import ctypes
    from wb_humanoid_mpc_amd import _abi
    from wb_humanoid_mpc_amd.distributed import BatchShards, broadcast_image
    from wb_humanoid_mpc_amd.reference import BENCH_SEED, make_problem
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    GB, N = args.global_batch, args.nodes
    dev = torch.device("cuda", local_rank)
    # shared problem image: rank 0's model description, broadcast; every rank builds its handle from the broadcast bytes
    image = bytes(ctypes.string_at(ctypes.addressof(model_local.desc), ctypes.sizeof(model_local.desc))) if rank == 0 else None
    image = broadcast_image(group, image)
    model_local.desc = _abi.ModelDesc.from_buffer_copy(image)
    dt = model_local.sqp["dt"]
    if rank == 0:
        gx0, gx, gu, gpar, dt = make_problem(model_local, n_nodes=N, batch=GB, gait=args.gait, perturb=not args.no_perturb, seed=BENCH_SEED)
        glob = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (gx0, gx, gu, gpar)]
    else:
        glob = [None] * 4
    sh = BatchShards(group, GB)
    shapes = [(_abi.NX,), (N + 1, _abi.NX), (N, _abi.NU), (N + 1, _abi.NODE_PARAMS)]
    solver = HipSqpSolver(model_local, max_nodes=N, max_batch=sh.per, device=local_rank, riccati=args.riccati)
    sol = dict(x=torch.empty((sh.per, N + 1, _abi.NX), dtype=torch.float64, device=dev), u=torch.empty((sh.per, N, _abi.NU), dtype=torch.float64, device=dev),
               perf=torch.empty((sh.per, 4), dtype=torch.float64, device=dev), kkt=torch.empty((sh.per, 2), dtype=torch.float64, device=dev))

    def scatter_upload():
        loc = [sh.scatter(g, shp) for g, shp in zip(glob, shapes)]
        torch.cuda.synchronize()
        solver.upload_device(sh.per, N, dt, *[t.data_ptr() for t in loc])
        return loc

    def download_gather():
        solver.download_device(x_ptr=sol["x"].data_ptr(), u_ptr=sol["u"].data_ptr(), perf_after_ptr=sol["perf"].data_ptr(), kkt_ptr=sol["kkt"].data_ptr())
        return {k: sh.gather(v) for k, v in sol.items()}

    keep = scatter_upload()
    for _ in range(args.warmup):
        solver.iterate(1, take_step=False)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        solver.iterate(1, take_step=False)
    sync()
    resident = time.perf_counter() - t0
    kms = solver.kernel_ms()
    # the whole data path every step: scatter + upload + iterate (with the KKT check) + download + gather
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        keep = scatter_upload()
        solver.iterate(1, take_step=False, kkt=True)
        gathered = download_gather()
    sync()
    inclusive = time.perf_counter() - t0
    resident, inclusive = group.max([resident, inclusive])
    out = None
    if rank == 0:
        # the gathered solution equals rank 0's own solve of the whole batch (instances are independent: bit for bit)
        ref = HipSqpSolver(model_local, max_nodes=N, max_batch=GB, device=local_rank)
        ref.upload_device(GB, N, dt, *[t.data_ptr() for t in glob])
        ref.iterate(1, take_step=False, kkt=True)
        rx = torch.empty_like(gathered["x"]); ru = torch.empty_like(gathered["u"])
        ref.download_device(x_ptr=rx.data_ptr(), u_ptr=ru.data_ptr())
        same = bool(torch.equal(rx, gathered["x"]) and torch.equal(ru, gathered["u"]))
        ref.close()
        per_gpu_bytes = sum(int(np.prod(shp)) for shp in shapes) * 8 * sh.per + sum(v.numel() * 8 for v in sol.values())
        out = {"scaling": "strong", "global_batch": GB, "batch_per_gpu": sh.per, "value": GB * args.steps / resident, "unit": "SQP iters/s",
               "ms_per_step": 1e3 * resident / args.steps, "kernel_ms": kms,
               "collective_inclusive": {"value": GB * args.steps / inclusive, "ms_per_step": 1e3 * inclusive / args.steps,
                                        "bytes_per_gpu_per_step": per_gpu_bytes,
                                        "note": "every step: RCCL scatter of x_init / x / u / node parameters from rank 0 into HBM, hsqp_upload_device, one SQP "
                                                "iteration incl. the KKT check, hsqp_download_device, RCCL gather of x / u / performance / KKT to rank 0"},
               "gathered_solution_equals_single_gpu_solve": same, "kkt_residual_max": float(gathered["kkt"].max().item()),
               "note": "BASELINE config 4 as written (256 instances over the GPUs); bounded by the serial Riccati sweep: one workgroup per instance, "
                       "~17 us per stage whatever the batch (DESIGN.md §6)"}
    # the same shard through the opt-in two-level sweep (hsqp_segment.h): what lifts the serial sweep's wall at 32 instances per GPU, reported
    # NEXT to the headline because it is a declared relaxation (its distance from the default path's solution is measured here)
    seg = None
    segP = 7 if min(256 // sh.per - 1, N // 4) >= 7 else 3 if min(256 // sh.per - 1, N // 4) >= 3 else 0   # segment_count() of hsqp_capi.hip
    if segP:
        s2 = HipSqpSolver(model_local, max_nodes=N, max_batch=sh.per, device=local_rank, riccati="segmented")
        s2.upload_device(sh.per, N, dt, *[t.data_ptr() for t in keep])
        for _ in range(args.warmup):
            s2.iterate(1, take_step=False)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            s2.iterate(1, take_step=False)
        sync()
        seg_t = group.max([time.perf_counter() - t0])[0]
        kms2 = s2.kernel_ms()
        sx, su = torch.empty_like(sol["x"]), torch.empty_like(sol["u"])
        s2.download_device(x_ptr=sx.data_ptr(), u_ptr=su.data_ptr())
        solver_x, solver_u = torch.empty_like(sol["x"]), torch.empty_like(sol["u"])
        solver.iterate(1, take_step=False)
        solver.download_device(x_ptr=solver_x.data_ptr(), u_ptr=solver_u.data_ptr())
        diff = group.max([float(torch.maximum((sx - solver_x).abs().max(), (su - solver_u).abs().max()).item()), float(s2.scan_fallbacks())])
        s2.close()
        if rank == 0:
            seg = {"value": GB * args.steps / seg_t, "unit": "SQP iters/s", "ms_per_step": 1e3 * seg_t / args.steps, "kernel_ms": kms2,
                   "max_abs_difference_from_the_default_path": diff[0], "gate_fallbacks_max_over_ranks": int(diff[1]),
                   "segments_per_instance": segP,
                   "note": "HSQP_FLAG_SEGMENTED_RICCATI: segment elements + suffix scan + per-segment recursions (hsqp_segment.h); "
                           "declared relaxation of BASELINE.md §6 (trajectories within 4e-10 of the step's scale of the serial recursion's), so it is NOT the headline"}
    if out is not None and seg is not None:
        out["two_level_sweep"] = seg
    solver.close()
    return out


def self_spawn(args):
    """`python bench.py --gpus N` outside torchrun: re-launch under torch.distributed.run with N ranks (one per GPU)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


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
    ap.add_argument("--riccati", default="auto", choices=["auto", "serial", "parallel", "segmented"],
                    help="backward sweep: auto (serial; scan for <= 2 instances), serial, parallel (the associative scan over the stages, hsqp_scan.h), "
                         "segmented (the two-level sweep, hsqp_segment.h: opt-in, declared relaxation of the trajectory tolerance)")
    ap.add_argument("--global-batch", type=int, default=256, help="strong-scaling leg (N > 1): instances of the one global batch")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling / data-path leg")
    ap.add_argument("--sustained", type=int, default=-1, help="steps of the sustained leg behind the timed region (default: max(300, --steps); 0: none)")
    ap.add_argument("--force-strong", action="store_true", help="run the data-path leg on one GPU too (scatter / gather degenerate to copies): exercises hsqp_upload_device / hsqp_download_device")
    args = ap.parse_args()

    from wb_humanoid_mpc_amd.distributed import Group, aggregate_throughput, env_rank, shard_seed
    rank, local_rank, world = env_rank()
    if args.gpus > 1 and "RANK" not in os.environ:
        self_spawn(args)                         # does not return
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or plainly (it re-launches itself)")

    # torch FIRST: its bundled libamdhip64.so.7 must be the one HIP runtime of the process; libhsqp_hip.so then
    # binds to it by soname (measured on the GPU box: the other order leaves torch.cuda unavailable)
    import torch
    from wb_humanoid_mpc_amd import load_model
    from wb_humanoid_mpc_amd.reference import BENCH_SEED, make_problem
    from wb_humanoid_mpc_amd.solver import HipSqpSolver, load_library
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path)")
    # HSQP_DIST_BACKEND=gloo: dry run of the N-rank code path on a box with fewer GPUs than ranks (RCCL refuses two ranks on one device);
    # ranks then share the devices round-robin and the collectives are staged through the host — a logic check, never a measurement
    backend = os.environ.get("HSQP_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    torch.zeros(1, device="cuda")  # initialise the HIP runtime through torch before the library touches it
    load_library()
    group = Group(world, backend=backend, device=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm

    cent = args.formulation == "centroidal"
    model = load_model(formulation=args.formulation)
    B, N = args.batch if args.batch else (1 if cent else 256), args.nodes
    if cent:
        from wb_humanoid_mpc_amd.reference import make_centroidal_problem
        x0, x, u, par, dt = make_centroidal_problem(model, n_nodes=N, batch=B, gait=args.gait, perturb=B > 1 and not args.no_perturb,
                                                    seed=shard_seed(BENCH_SEED, rank))
    else:
        x0, x, u, par, dt = make_problem(model, n_nodes=N, batch=B, gait=args.gait, perturb=not args.no_perturb, seed=shard_seed(BENCH_SEED, rank))
    solver = HipSqpSolver(model, max_nodes=N, max_batch=B, device=local_rank, riccati=args.riccati)
    solver.upload(x0, x, u, par, dt)      # inputs resident in HBM before the timed region

    def sync():
        torch.cuda.synchronize()
        group.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        solver.iterate(1, take_step=False)
    sync()
    clocks_before = gpu_clocks(local_rank)
    kms = np.zeros(5)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        solver.iterate(1, take_step=False)   # blocking: synchronises the library's stream
        k = solver.kernel_ms()
        kms += [k["lq"], k["project"], k["riccati"], k["step_perf"], k["total"]]
    sync()
    elapsed = time.perf_counter() - t0
    kms /= args.steps
    clocks_after = gpu_clocks(local_rank)
    # sustained rate: the same handle, the same resident problem, >= 300 more steps (the timed region above is the driver's --steps: 83 ms at the
    # default 20 — too short to show a box that throttles); never `value`
    sus_steps = max(300, args.steps) if args.sustained < 0 else args.sustained
    sustained = None
    if sus_steps > 0:
        sync()
        t1 = time.perf_counter()
        for _ in range(sus_steps):
            solver.iterate(1, take_step=False)
        sync()
        sus_elapsed = group.max([time.perf_counter() - t1])[0]
        sustained = {"steps": sus_steps, "ms_per_step": 1e3 * sus_elapsed / sus_steps, "value": B * world * sus_steps / sus_elapsed, "unit": "SQP iters/s",
                     "clocks_after": gpu_clocks(local_rank),
                     "note": "same handle and resident problem as the timed region, run right behind it; max over ranks; reported next to `value`, never as it"}
    solver.iterate(1, take_step=False, kkt=True)   # outside the timed region: KKT residual of the QP for the report
    out = solver.download()
    kkt = float(np.max(out["kkt"]))
    kkt_norm = float(np.max(out["kkt"].max(1) / np.maximum(1.0, out["grad_inf"])))   # r <= 1e-9 max(1, |g|_inf), per instance
    g_inf_min, g_inf_max = float(out["grad_inf"].min()), float(out["grad_inf"].max())
    # outside the timed region: the same iteration through hsqp_solve with HOST buffers (upload + iterate + download over PCIe)
    # (solution buffers allocated once and reused, as an MPC loop would; first from pageable memory, then with every buffer page-locked
    #  through hsqp_host_register)
    into = solver.alloc_solution(B, N)
    xs = [np.ascontiguousarray(a) for a in (x0, x, u, par)]
    solver.run(*xs, dt, into=into)
    t1 = time.perf_counter()
    for _ in range(2):
        solver.run(*xs, dt, into=into)
    pcie_ms = 1e3 * (time.perf_counter() - t1) / 2
    pinned = xs + [into[1][k] for k in ("x", "u", "dx", "du")]
    pcie_pinned_ms = None
    try:
        solver.pin(*pinned)
        solver.run(*xs, dt, into=into)
        t1 = time.perf_counter()
        for _ in range(2):
            solver.run(*xs, dt, into=into)
        pcie_pinned_ms = 1e3 * (time.perf_counter() - t1) / 2
        solver.unpin(*pinned)
    except Exception:  # noqa: BLE001  (page-locking 75 MB can be refused by the container's memlock limit)
        pcie_pinned_ms = None

    elapsed, kkt, kkt_norm = group.max([elapsed, kkt, kkt_norm])   # max over ranks
    strong = None
    if (world > 1 or args.force_strong) and not args.no_strong and not cent:
        strong = strong_scaling_leg(args, torch, group, model, rank, local_rank, world, sync)
    # RCCL really spans `world` GPUs: an all-reduce of ones over the ranks' devices
    rccl_ranks = 1
    if world > 1:
        ones = torch.ones(1, dtype=torch.float64, device=torch.device("cuda", local_rank) if backend == "nccl" else "cpu")
        group.dist.all_reduce(ones)
        rccl_ranks = int(round(float(ones.item())))

    if rank == 0:
        weak_value = aggregate_throughput([B] * world, args.steps, elapsed)
        cfg = {(100, 256): "4", (100, 1): "3", (200, 1024): "5"}.get((N, B), "4-like")
        if cent:
            cfg = {(20, 1): "1", (100, 1): "2"}.get((N, B), "2-like")
        nodes = B * N
        f_rk4, f_gn, f_proj, f_ric = CENT_F if cent else (F_RK4, F_GN, F_PROJ, F_RIC)
        f_node, bytes_node = f_rk4 + f_gn + f_proj + f_ric, CENT_BYTES_NODE if cent else BYTES_NODE
        seg_used = args.riccati == "segmented" and min(256 // B - 1, N // 4) >= 3   # segment_count() of hsqp_capi.hip
        scan_used = (args.riccati == "parallel" or (args.riccati == "auto" and B <= 2 and N >= 48)) and not seg_used   # HSQP_SCAN_AUTO_BATCH / _MIN_NODES
        # Per-kernel algorithmic work (SURVEY §8d dense counts) and measured duration (HIP events on the library's stream).  The
        # Gauss-Newton contraction J^T J (F_gn) runs in k_project (hsqp_project.h), not in the LQ kernel; the LQ kernel's dense count is
        # the RK4 sensitivity product only — its real work (four analytic rigid-body model evaluations per node) is vector FP64 and is
        # not part of SURVEY's count, so its fraction is reported against the same FP64 peak but its bound is VALU issue, not the matrix pipe.
        pmc, pmc_prov = pmc_kernel_info()
        forms = solver.kernel_forms()   # which kernels this handle runs (decided at hsqp_create from its size: HSQP_BLK_FORMS)
        # (chain_fused: the RK4 chain of the [A|B] columns runs inside k_project and the defect on the lanes of k_lq_rows — k_lq_chain is not launched; the dense RK4 count
        #  stays with the LQ bucket, where the stage Jacobians it is made of are formed: k_project's algorithmic flops are NOT raised by the move)
        lq_name = "k_lq_cent2" if cent else (("k_lq_limb + k_lq_rows" if forms.get("chain_fused") else "k_lq_limb + k_lq_rows + k_lq_chain") if forms["lq_limb"] else "k_lq<true>")
        step_name = ("k_step + k_lq_cent2_value (+ reductions)" if cent else
                     ("k_step + k_value_quad (+ reductions)" if forms["value_quad"] else "k_step_value (+ reductions)"))
        kern = {lq_name: (kms[0], f_rk4, "valu-issue / scattered stores (limb lanes)" if forms["lq_limb"] and not cent else "valu-issue"),
                "k_project": (kms[1], f_proj + f_gn, "mfma"),
                ("k_scan_*" if scan_used else ("k_seg_*" if seg_used else ("k_riccati_fact" if forms.get("ric_fact") and not cent else "k_riccati"))): (kms[2], f_ric, "latency (serial stage chain; matrix pipe)" if not scan_used else "mfma"),
                step_name: (kms[3], 0.0, "hbm (step) / valu-issue (value pass)")}
        per_kernel = {}
        for name, (ms, fl, bound) in kern.items():
            # a bucket of several kernels (timed together by the library's HIP events): their HBM bytes add up, matrix-pipe share from its first MFMA kernel
            parts = [q.strip().split(" ")[0] for q in name.split("+")]
            parts = [q.replace("k_scan_*", "k_scan_combine") if q != "k_riccati" else "k_riccati<58>" for q in parts]
            infos = [pmc.get(q, {}) for q in parts] if (B, N) == (256, 100) and not cent else []
            hbm = sum(i.get("hbm_bytes", 0.0) for i in infos) if any("hbm_bytes" in i for i in infos) else None
            busy = next((i.get("mfma_busy") for i in infos if i.get("mfma_busy")), 0.0 if infos and any(i for i in infos) else None)
            per_kernel[name] = {"ms": ms, "bound": bound, "algorithmic_TFLOPs": nodes * fl / (ms * 1e-3) / 1e12 if ms > 0 else None,
                                "frac_fp64": nodes * fl / (ms * 1e-3) / 1e12 / PEAK_FP64_TFLOPS if ms > 0 else None,
                                "mfma_busy": busy, "hbm_bytes": hbm}
        dom = max((k for k in kern if kern[k][1] > 0), key=lambda n: kern[n][0])
        dom_ms, dom_flops, dom_bound = kern[dom]
        ach_tf = nodes * dom_flops / (dom_ms * 1e-3) / 1e12
        step_tf = nodes * f_node / (elapsed / args.steps) / 1e12
        step_tbs = nodes * bytes_node / (elapsed / args.steps) / 1e12
        headline_strong = strong is not None and world > 1
        value = strong["value"] if headline_strong else weak_value
        ms_step = strong["ms_per_step"] if headline_strong else 1e3 * elapsed / args.steps
        GB = args.global_batch if headline_strong else B * world
        res = {
            "metric": "SQP iters/sec (G1 centroidal MPC)" if cent else "SQP iters/sec (G1 WB-MPC, N=100)", "value": value, "unit": "SQP iters/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong" if (headline_strong or world == 1) else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic", "rccl_ranks": rccl_ranks,
            "collective_backend": "nccl (RCCL over xGMI)" if backend == "nccl" else f"{backend} (DRY RUN of the N-rank path through the host: no RCCL involved, rccl_ranks counts {backend} ranks)",
            "sustained": sustained, "gpu_clocks": {"before_timed_region": clocks_before, "after_timed_region": clocks_after},
            "config": {"workload": f"BASELINE config {cfg}: G1 {'centroidal' if cent else 'whole-body'} MPC, N={N}, dt={dt}, gait {args.gait}, "
                                   f"{GB} {'perturbed ' if not args.no_perturb else ''}instances in all ({GB // world if headline_strong else B} per GPU), "
                                   "1 SQP iteration per step (LQ + projection + Riccati + full step + performance index), cold-start trajectory",
                       "batch_per_gpu": GB // world if headline_strong else B, "global_batch": GB, "nodes": N,
                       "parallelism": (f"one global batch sharded over {world} GPUs: RCCL broadcast of the problem image + scatter of the instance blocks, shards resident while timed"
                                       if headline_strong else f"batch-sharded x{world}, no data-path collective"),
                       "backward_sweep": "parallel-in-time scan over the stages (hsqp_scan.h)" if scan_used else ("two-level (segmented) sweep (hsqp_segment.h)" if seg_used else "serial Riccati recursion")},
            "roofline": {"bound": dom_bound, "kernel": dom, "achieved": ach_tf, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tf / PEAK_FP64_TFLOPS, "traffic": per_kernel[dom]["hbm_bytes"],
                         "peak_note": "FP64 vector = FP64 matrix peak (78.6 TFLOP/s; tools/microbench/f64_rates.hip measured 77.4 / 73)",
                         "traffic_note": "HBM bytes per launch of the dominant kernel (FETCH_SIZE x2 + WRITE_SIZE) from the committed rocprofv3 --pmc "
                                         "passes of this command (see counters_provenance); null for other shapes or when the passes are absent",
                         "counters_provenance": pmc_prov,
                         "note": "achieved = ALGORITHMIC (dense-count, SURVEY §8d) flops of the dominant kernel / its HIP-event duration; the kernels exploit "
                                 "the flow map's structure and execute fewer flops than the dense count (DESIGN.md); the N = 1 measurement of this rank",
                         "per_kernel": per_kernel,
                         "whole_step_algorithmic_TFLOPs": step_tf, "whole_step_frac_fp64": step_tf / PEAK_FP64_TFLOPS,
                         "whole_step_unfused_TBs": step_tbs, "whole_step_frac_hbm": step_tbs / PEAK_HBM_TBS},
            "kernel_ms": {"lq": kms[0], "project": kms[1], "riccati": kms[2], "step_perf": kms[3], "sum": kms[4]},
            "kkt_residual_max": kkt, "kkt_over_max_1_g_inf": kkt_norm, "g_inf_range": [g_inf_min, g_inf_max],
            "kkt_note": "max over instances of max(r_stat, r_prim) / max(1, |g|_inf), g = gradient of the projected QP; BASELINE.md §6 asks <= 1e-9",
            "assumptions": "every number depends on the oracle-level assumptions A1 (PieceWisePolynomialBarrierPenalty), A2 (orientation error to plane)" +
                           (" and A7 (centroidal flow map)" if cent else "") + ": DESIGN.md §2",
            "pcie_inclusive": {"ms_per_step": pcie_ms, "value": B / (pcie_ms * 1e-3), "unit": "SQP iters/s per GPU",
                               "pinned_ms_per_step": pcie_pinned_ms, "pinned_value": None if not pcie_pinned_ms else B / (pcie_pinned_ms * 1e-3),
                               "note": "hsqp_solve with host buffers reused every call (upload 36 MB + iterate incl. KKT check + download 39 MB at B=256, N=100), from pageable memory and "
                                       "with the buffers page-locked through hsqp_host_register; never `value`"},
        }
        if world > 1:
            res["weak_scaling"] = {"scaling": "weak", "value": weak_value, "unit": "SQP iters/s", "ms_per_step": 1e3 * elapsed / args.steps, "batch_per_gpu": B,
                                   "global_batch": B * world, "kernel_ms": res["kernel_ms"],
                                   "note": "256 instances PER GPU (per-rank seeds), no data-path collective; RCCL only for the barrier and the max-over-ranks time"}
            res["n1_equivalent"] = {"value": weak_value / world, "unit": "SQP iters/s per GPU",
                                    "note": "per-GPU rate with a full 256-instance batch resident (the weak leg / N): the same work as the --gpus 1 line, must reproduce its `value`"}
        pred = strong_prediction(GB // world if headline_strong else B, N, args.riccati) if not cent else None
        if pred is not None:
            res["strong_prediction"] = dict(pred, measured_ms_per_step=ms_step, measured_over_predicted=ms_step / pred["predicted_ms_per_step"])
        if strong is not None:
            res["strong_scaling"] = strong
            if headline_strong:
                res["kernel_ms_strong"] = strong["kernel_ms"]
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(model, N, BENCH_SEED, cent=cent)
            res["speedup_vs_cpu_baseline"] = value / res["cpu_baseline"]["value"]
            ace = res["cpu_baseline"]["all_core_extrapolation"]
            res["extrapolated_not_measured"] = {"speedup_vs_all_core_extrapolation": value / ace["value"], "extrapolation_factor": ace["factor"],
                                                "note": "NOT a measurement: the measured per-core rate of the container's cores times the host's physical core count "
                                                        "(linear scaling assumed beyond the cores the cgroup allows).  The >= 10x target of BASELINE.md is against the all-core "
                                                        "host, which this container cannot time; speedup_vs_cpu_baseline is the measured ratio against the cores it may use"}
        print(json.dumps(res))
    solver.close()
    group.close()


if __name__ == "__main__":
    main()
