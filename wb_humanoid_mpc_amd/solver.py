"""Host-side mirror of the reference's solver interface over the HIP C ABI.

The reference drives its solver through ocs2::SolverBase / MPC_BASE
(humanoid_nmpc/humanoid_wb_mpc_ros2/src/WBMpcSqpNode.cpp:64-89,
 humanoid_nmpc/humanoid_wb_mpc/src/mrt/WBMpcMrtJointController.cpp:200-213):
reset(), run(t0, x0, tf), getPrimalSolution(), getPerformanceIndeces(), and the fork's
SqpSolver::getBenchmarks() (humanoid_common_mpc_ros2/src/benchmarks/SqpBenchmarksPublisher.cpp:44-57).
HipSqpSolver keeps those names and meanings for a batch of independent MPC instances; the C++
adaptor of INTEGRATION.md is the same thin layer in the reference's own language.

There is no CPU path: constructing a solver without the built HIP library or without a GPU raises.
"""
import ctypes as C
import os

import numpy as np

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HSQP_LIB") or os.path.join(_HERE, "libhsqp_hip.so")   # HSQP_LIB: debug builds only
_dp = C.POINTER(C.c_double)
_lib = None


class HsqpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"hsqp error {code}: {msg}")
        self.code = code


def load_library():
    """Load libhsqp_hip.so (built in-tree by wb_humanoid_mpc_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m wb_humanoid_mpc_amd.build` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.hsqp_create.argtypes = [C.POINTER(_abi.ModelDesc), C.POINTER(_abi.Settings), C.POINTER(C.c_void_p)]
    lib.hsqp_destroy.argtypes = [C.c_void_p]
    lib.hsqp_solve.argtypes = [C.c_void_p, C.POINTER(_abi.Problem), C.POINTER(_abi.Solution)]
    lib.hsqp_upload.argtypes = [C.c_void_p, C.POINTER(_abi.Problem)]
    lib.hsqp_upload_reference.argtypes = [C.c_void_p, C.POINTER(_abi.Problem), C.POINTER(_abi.Reference)]
    lib.hsqp_iterate_device.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.hsqp_last_iterations.argtypes = [C.c_void_p]
    lib.hsqp_iteration_log.argtypes = [C.c_void_p, C.c_int, C.POINTER(_abi.Perf), _dp, C.POINTER(C.c_int32)]
    lib.hsqp_update_weights.argtypes = [C.c_void_p, _dp, _dp, _dp]
    lib.hsqp_host_register.argtypes = [C.c_void_p, C.c_size_t]
    lib.hsqp_host_unregister.argtypes = [C.c_void_p]
    lib.hsqp_download.argtypes = [C.c_void_p, C.POINTER(_abi.Solution)]
    lib.hsqp_upload_device.argtypes = [C.c_void_p, C.POINTER(_abi.Problem)]
    lib.hsqp_download_device.argtypes = [C.c_void_p, C.POINTER(_abi.Solution)]
    lib.hsqp_debug_read.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]
    lib.hsqp_debug_read.restype = C.c_longlong
    lib.hsqp_last_kernel_ms.argtypes = [C.c_void_p, _dp]
    lib.hsqp_last_error.argtypes = [C.c_void_p]
    lib.hsqp_last_error.restype = C.c_char_p
    lib.hsqp_scan_fallbacks.argtypes = [C.c_void_p]
    lib.hsqp_scan_fallbacks.restype = C.c_longlong
    lib.hsqp_get_term_weights.argtypes = [C.c_void_p, C.c_void_p]
    lib.hsqp_update_term_weights.argtypes = [C.c_void_p, C.c_void_p]
    lib.hsqp_scan_backoffs.argtypes = [C.c_void_p]
    lib.hsqp_scan_backoffs.restype = C.c_longlong
    lib.hsqp_version.restype = C.c_char_p
    lib.hsqp_set_scan_backoff_persistent.argtypes = [C.c_void_p, C.c_int]
    # hsqp_comm_*: the batch axis over the GPUs of one node behind the C ABI (the Python host uses torch.distributed instead: distributed.py)
    lib.hsqp_comm_unique_id.argtypes = [C.c_void_p]
    lib.hsqp_comm_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.hsqp_comm_destroy.argtypes = [C.c_void_p]
    lib.hsqp_comm_destroy.restype = None
    lib.hsqp_comm_create_error.restype = C.c_char_p
    lib.hsqp_comm_last_error.argtypes = [C.c_void_p]
    lib.hsqp_comm_last_error.restype = C.c_char_p
    lib.hsqp_comm_rank.argtypes = [C.c_void_p]
    lib.hsqp_comm_world.argtypes = [C.c_void_p]
    lib.hsqp_comm_shard.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.hsqp_comm_shard_of.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.hsqp_comm_broadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int]
    lib.hsqp_comm_scatter_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int]
    lib.hsqp_comm_gather_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int]
    lib.hsqp_comm_max.argtypes = [C.c_void_p, _dp, C.c_int]
    lib.hsqp_comm_barrier.argtypes = [C.c_void_p]
    if lib.hsqp_abi_version() != _abi.ABI_VERSION:
        raise RuntimeError(f"libhsqp_hip ABI {lib.hsqp_abi_version()} != the binding's {_abi.ABI_VERSION} (include/hsqp.h: HSQP_ABI_VERSION)")
    lib.hsqp_joint_torques.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp]
    lib.hsqp_evaluate_policy.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
    lib.hsqp_linesearch_defaults.argtypes = [C.POINTER(_abi.LinesearchSettings)]
    lib.hsqp_linesearch_defaults.restype = None
    lib.hsqp_set_linesearch.argtypes = [C.c_void_p, C.POINTER(_abi.LinesearchSettings)]
    _lib = lib
    return lib


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class HipSqpSolver:
    def __init__(self, model, max_nodes, max_batch=1, device=0, linesearch=False, riccati="auto"):
        """linesearch=True: run() uses the filter line search (ocs2 SqpSolver behaviour) instead of the full step.
        riccati: "auto" (serial recursion; KKT-gated parallel-in-time scan for <= 2 instances on >= 48 nodes), "serial", "parallel" (the scan
        for every size), "segmented" (the KKT-gated two-level sweep of hsqp_segment.h for mid-sized batches: faster, but a declared
        relaxation — its step is up to 4e-10 of the step's scale from the serial recursion's)."""
        self.lib = load_library()
        self.model = model
        flags = (_abi.FLAG_LINESEARCH if linesearch else 0) | {"auto": 0, "serial": _abi.FLAG_SERIAL_RICCATI, "parallel": _abi.FLAG_PARALLEL_RICCATI, "segmented": _abi.FLAG_SEGMENTED_RICCATI}[riccati]
        st = _abi.Settings(max_nodes=max_nodes, max_batch=max_batch, device=device, flags=flags)
        h = C.c_void_p()
        rc = self.lib.hsqp_create(C.byref(model.desc), C.byref(st), C.byref(h))
        if rc != 0:
            raise HsqpError(rc, self.lib.hsqp_last_error(None).decode())
        self.h = h
        self.max_nodes, self.max_batch = max_nodes, max_batch
        self._shape = None
        self._sol = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.hsqp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise HsqpError(rc, self.lib.hsqp_last_error(self.h).decode())

    # ---- SolverBase::reset
    def reset(self):
        self._shape = None
        self._sol = None

    def _problem(self, x_init, x_traj, u_traj, params, dt):
        x_traj, u_traj, params, x_init = _c(x_traj), _c(u_traj), _c(params), _c(x_init)
        if x_traj.ndim == 2:
            x_traj, u_traj, params, x_init = x_traj[None], u_traj[None], params[None], x_init[None]
        B, N = u_traj.shape[0], u_traj.shape[1]
        if x_traj.shape != (B, N + 1, _abi.NX) or u_traj.shape != (B, N, _abi.NU) or \
                params.shape != (B, N + 1, _abi.NODE_PARAMS) or x_init.shape != (B, _abi.NX):
            raise ValueError("inconsistent problem array shapes")
        dt, grid = self._grid(dt, B, N)
        keep = (x_init, x_traj, u_traj, params, grid)
        p = _abi.Problem(batch=B, n_nodes=N, dt=dt, x_init=x_init.ctypes.data_as(_dp), x_traj=x_traj.ctypes.data_as(_dp),
                         u_traj=u_traj.ctypes.data_as(_dp), node_params=params.ctypes.data_as(_dp),
                         dt_nodes=None if grid is None else grid.ctypes.data_as(_dp))
        return p, keep, (B, N)

    @staticmethod
    def _grid(dt, B, N):
        """dt: the uniform node spacing, or the interval lengths of a non-uniform grid ([N] shared by all instances or [B][N];
        a zero marks an event interval, hsqp_problem::dt_nodes)."""
        if np.ndim(dt) == 0:
            return float(dt), None
        grid = np.ascontiguousarray(np.broadcast_to(np.asarray(dt, dtype=np.float64), (B, N)))
        return 0.0, grid

    def _alloc_solution(self, B, N):
        out = dict(x=np.zeros((B, N + 1, _abi.NX)), u=np.zeros((B, N, _abi.NU)), dx=np.zeros((B, N + 1, _abi.NX)),
                   du=np.zeros((B, N, _abi.NU)), kkt=np.zeros((B, 2)), alpha=np.zeros(B), step_type=np.zeros(B, dtype=np.int32),
                   armijo=np.zeros(B), grad_inf=np.zeros(B))
        pb, pa = (_abi.Perf * B)(), (_abi.Perf * B)()
        s = _abi.Solution(x=out["x"].ctypes.data_as(_dp), u=out["u"].ctypes.data_as(_dp), dx=out["dx"].ctypes.data_as(_dp),
                          du=out["du"].ctypes.data_as(_dp), perf_before=pb, perf_after=pa, kkt=out["kkt"].ctypes.data_as(_dp),
                          alpha=out["alpha"].ctypes.data_as(_dp), step_type=out["step_type"].ctypes.data_as(C.POINTER(C.c_int32)),
                          armijo=out["armijo"].ctypes.data_as(_dp), grad_inf=out["grad_inf"].ctypes.data_as(_dp))
        return s, out, pb, pa

    @staticmethod
    def _perf(arr):
        return [dict(merit=p.merit, cost=p.cost, dynamics_sse=p.dynamics_sse, equality_sse=p.equality_sse) for p in arr]

    def _finish(self, s, out, pb, pa):
        out["perf_before"], out["perf_after"] = self._perf(pb), self._perf(pa)
        t = s.timings
        out["benchmarks"] = dict(linearQuadraticApproximationTime=t.lq_approximation, solveQpTime=t.solve_qp,
                                 linesearchTime=t.linesearch, computeControllerTime=t.compute_controller, total=t.total)
        self._sol = out
        return out

    # ---- SolverBase::run (one SQP iteration, sqpIteration = 1 as in task.info:81)
    def pin(self, *arrays):
        """hsqp_host_register on caller-owned numpy arrays (page-locked: one DMA per transfer); unpin() before they are freed."""
        for a in arrays:
            self._check(self.lib.hsqp_host_register(a.ctypes.data_as(C.c_void_p), a.nbytes))

    def unpin(self, *arrays):
        for a in arrays:
            self._check(self.lib.hsqp_host_unregister(a.ctypes.data_as(C.c_void_p)))

    def alloc_solution(self, B, N):
        """Solution buffers for run(..., into=...): allocate (and pin) once, reuse every cycle."""
        return self._alloc_solution(B, N)

    def run(self, x_init, x_traj, u_traj, params, dt, into=None):
        p, keep, (B, N) = self._problem(x_init, x_traj, u_traj, params, dt)
        s, out, pb, pa = into if into is not None else self._alloc_solution(B, N)
        self._check(self.lib.hsqp_solve(self.h, C.byref(p), C.byref(s)))
        self._shape = (B, N)
        return self._finish(s, out, pb, pa)

    # ---- device-resident loop (bench): upload once, iterate, download
    def upload(self, x_init, x_traj, u_traj, params, dt):
        p, keep, self._shape = self._problem(x_init, x_traj, u_traj, params, dt)
        self._check(self.lib.hsqp_upload(self.h, C.byref(p)))

    def upload_reference(self, x_init, x_traj, u_traj, dt, t0, n_events, event_times, mode_sequence, target_times, target_states,
                         swing, terrain_height=0.0, arm_swing=True, node_times=None):
        """hsqp_upload_reference: the per-node parameter table is generated on the device from the compact reference
        (mode schedule + target knots per instance; see reference.pack_reference).  Non-uniform grid: dt = interval lengths
        ([N] or [B][N]) together with node_times ([N+1] or [B][N+1])."""
        x_traj, u_traj, x_init = _c(x_traj), _c(u_traj), _c(x_init)
        if x_traj.ndim == 2:
            x_traj, u_traj, x_init = x_traj[None], u_traj[None], x_init[None]
        B, N = u_traj.shape[0], u_traj.shape[1]
        n_events = np.ascontiguousarray(n_events, dtype=np.int32)
        mode_sequence = np.ascontiguousarray(mode_sequence, dtype=np.int32)
        event_times, target_times, target_states = _c(event_times), _c(target_times), _c(target_states)
        if event_times.shape[0] != B or mode_sequence.shape != (B, event_times.shape[1] + 1) or target_states.shape != (B, target_times.shape[1], _abi.NX):
            raise ValueError("inconsistent reference array shapes")
        ip = C.POINTER(C.c_int32)
        dt, grid = self._grid(dt, B, N)
        nt = None if node_times is None else np.ascontiguousarray(np.broadcast_to(np.asarray(node_times, dtype=np.float64), (B, N + 1)))
        p = _abi.Problem(batch=B, n_nodes=N, dt=dt, x_init=x_init.ctypes.data_as(_dp), x_traj=x_traj.ctypes.data_as(_dp),
                         u_traj=u_traj.ctypes.data_as(_dp), node_params=None, dt_nodes=None if grid is None else grid.ctypes.data_as(_dp))
        r = _abi.Reference(batch=B, n_nodes=N, t0=t0, dt=dt, node_times=None if nt is None else nt.ctypes.data_as(_dp), max_events=event_times.shape[1], n_events=n_events.ctypes.data_as(ip),
                           event_times=event_times.ctypes.data_as(_dp), mode_sequence=mode_sequence.ctypes.data_as(ip),
                           n_knots=target_times.shape[1], target_times=target_times.ctypes.data_as(_dp),
                           target_states=target_states.ctypes.data_as(_dp), swing=swing, terrain_height=terrain_height,
                           arm_swing=1 if arm_swing else 0, reserved=0)
        self._check(self.lib.hsqp_upload_reference(self.h, C.byref(p), C.byref(r)))
        self._shape = (B, N)

    def device_params(self):
        """The per-node parameter table resident on the device ([B][N+1][72])."""
        B, N = self._shape
        a = np.zeros((B, N + 1, _abi.NODE_PARAMS))
        n = self.lib.hsqp_debug_read(self.h, _abi.BLK_PARAMS, a.ctypes.data_as(C.c_void_p), a.nbytes)
        if n < 0:
            raise HsqpError(n, self.lib.hsqp_last_error(self.h).decode())
        return a

    def iterate(self, n_iterations=1, take_step=False, kkt=False, linesearch=False, until_converged=False):
        """until_converged: n_iterations is an upper bound; the call ends when every instance's step is below deltaTol (or no step length
        was accepted): ocs2's SqpSolver::checkConvergence.  Returns the number of iterations run."""
        self._check(self.lib.hsqp_iterate_device(self.h, n_iterations, (1 if take_step else 0) | (2 if kkt else 0) | (4 if linesearch else 0) |
                                                 (8 if until_converged else 0)))
        return int(self.lib.hsqp_last_iterations(self.h))

    def iteration_log(self, iteration):
        """(perf[B], alpha[B], step_type[B]) iteration `iteration` of the last until_converged call ended with."""
        B, _ = self._shape
        perf, alpha, st = (_abi.Perf * B)(), np.zeros(B), np.zeros(B, dtype=np.int32)
        self._check(self.lib.hsqp_iteration_log(self.h, iteration, perf, alpha.ctypes.data_as(_dp), st.ctypes.data_as(C.POINTER(C.c_int32))))
        return self._perf(perf), alpha, st

    def update_weights(self, Q=None, R=None, Qf=None):
        """hsqp_update_weights: new diagonal Q[58] / R[35] / Qf[58] of the live handle (None keeps the current one)."""
        arrs = [None if a is None else _c(a) for a in (Q, R, Qf)]
        self._check(self.lib.hsqp_update_weights(self.h, *[None if a is None else a.ctypes.data_as(_dp) for a in arrs]))

    def term_weights(self):
        """hsqp_get_term_weights: the handle's current foot-cost weights, foot-constraint gains, barrier parameters (an _abi.TermWeights)."""
        w = _abi.TermWeights()
        self._check(self.lib.hsqp_get_term_weights(self.h, C.byref(w)))
        return w

    def update_term_weights(self, w):
        """hsqp_update_term_weights: replace them (what the reference's gains updaters retune at run time)."""
        self._check(self.lib.hsqp_update_term_weights(self.h, C.byref(w)))

    # ---- sqp::Settings of the line search (task.info sqp block + upstream defaults)
    def linesearch_settings(self):
        s = _abi.LinesearchSettings()
        self.lib.hsqp_linesearch_defaults(C.byref(s))
        return s

    def set_linesearch(self, **kw):
        s = self.linesearch_settings()
        for k, v in kw.items():
            if not hasattr(s, k):
                raise ValueError(f"unknown line-search setting {k}")
            setattr(s, k, float(v))
        self._check(self.lib.hsqp_set_linesearch(self.h, C.byref(s)))
        return s

    # ---- multi-GPU data path: problem shards / solutions that are already in this GPU's HBM (e.g. the tensors RCCL
    #      scattered / will gather); pointers are raw device addresses (torch: tensor.data_ptr())
    def upload_device(self, batch, n_nodes, dt, x_init_ptr, x_ptr, u_ptr, params_ptr, dt_nodes_ptr=0):
        cast = lambda a: C.cast(C.c_void_p(int(a)), _dp) if a else None  # noqa: E731
        p = _abi.Problem(batch=batch, n_nodes=n_nodes, dt=float(dt), x_init=cast(x_init_ptr), x_traj=cast(x_ptr), u_traj=cast(u_ptr),
                         node_params=cast(params_ptr), dt_nodes=cast(dt_nodes_ptr))
        self._check(self.lib.hsqp_upload_device(self.h, C.byref(p)))
        self._shape = (batch, n_nodes)

    def download_device(self, x_ptr=0, u_ptr=0, dx_ptr=0, du_ptr=0, perf_before_ptr=0, perf_after_ptr=0, kkt_ptr=0, grad_inf_ptr=0):
        cast = lambda a, t=_dp: C.cast(C.c_void_p(int(a)), t) if a else None  # noqa: E731
        s = _abi.Solution(x=cast(x_ptr), u=cast(u_ptr), dx=cast(dx_ptr), du=cast(du_ptr), perf_before=cast(perf_before_ptr, C.POINTER(_abi.Perf)),
                          perf_after=cast(perf_after_ptr, C.POINTER(_abi.Perf)), kkt=cast(kkt_ptr), grad_inf=cast(grad_inf_ptr))
        self._check(self.lib.hsqp_download_device(self.h, C.byref(s)))

    def download(self):
        B, N = self._shape
        s, out, pb, pa = self._alloc_solution(B, N)
        self._check(self.lib.hsqp_download(self.h, C.byref(s)))
        return self._finish(s, out, pb, pa)

    def kernel_ms(self):
        ms = np.zeros(5)
        self._check(self.lib.hsqp_last_kernel_ms(self.h, ms.ctypes.data_as(_dp)))
        return dict(lq=ms[0], project=ms[1], riccati=ms[2], step_perf=ms[3], total=ms[4])

    def kernel_forms(self):
        """Which kernel forms the handle runs (HSQP_BLK_FORMS): {"lq_limb": bool, "value_quad": bool, "lq_ranges": int, "ric_fact": bool, "chain_fused": bool}."""
        f = np.zeros(5, dtype=np.int32)
        n = self.lib.hsqp_debug_read(self.h, _abi.BLK_FORMS, f.ctypes.data_as(C.c_void_p), f.nbytes)
        if n < 0:
            self._check(int(n))
        return {"lq_limb": bool(f[0]), "value_quad": bool(f[1]), "lq_ranges": int(f[2]), "ric_fact": bool(f[3]), "chain_fused": bool(f[4])}

    def set_scan_backoff_persistent(self, on=True):
        """hsqp_set_scan_backoff_persistent: the KKT gate's back-off survives uploads of the same shape (receding-horizon use)."""
        self._check(self.lib.hsqp_set_scan_backoff_persistent(self.h, 1 if on else 0))

    def scan_backoffs(self):
        """Iterations that ran the serial recursion because the automatic sweep choice was backing off (hsqp_scan_backoffs)."""
        return int(self.lib.hsqp_scan_backoffs(self.h))

    def scan_fallbacks(self):
        """Iterations whose parallel-in-time sweep failed the KKT gate and were redone with the serial recursion (hsqp_scan_fallbacks)."""
        return int(self.lib.hsqp_scan_fallbacks(self.h))

    # ---- the step after the solve: MPC_MRT_Interface::evaluatePolicy + computeJointTorques (WBMpcMrtJointController.cpp:136-147)
    def evaluate_policy(self, s):
        """Feed-forward policy of the device-resident solution at s[b] seconds after the first node: (x[B,58], u[B,35], tau[B,23])."""
        B, _ = self._shape
        s = _c(np.broadcast_to(s, (B,)))
        x, u, tau = np.zeros((B, _abi.NX)), np.zeros((B, _abi.NU)), np.zeros((B, _abi.NJ))
        self._check(self.lib.hsqp_evaluate_policy(self.h, s.ctypes.data_as(_dp), x.ctypes.data_as(_dp), u.ctypes.data_as(_dp), tau.ctypes.data_as(_dp)))
        return x, u, tau

    def joint_torques(self, x, u):
        x, u = _c(np.atleast_2d(x)), _c(np.atleast_2d(u))
        tau = np.zeros((x.shape[0], _abi.NJ))
        self._check(self.lib.hsqp_joint_torques(self.h, x.shape[0], x.ctypes.data_as(_dp), u.ctypes.data_as(_dp), tau.ctypes.data_as(_dp)))
        return tau

    # ---- reference-named accessors
    def getPrimalSolution(self):
        return self._sol["x"], self._sol["u"]

    def getPerformanceIndeces(self):
        return self._sol["perf_after"]

    def getBenchmarks(self):
        return self._sol["benchmarks"]

    # ---- parity/debug access to intermediate blocks of the last iteration
    def debug_read(self, what):
        B, N = self._shape
        shapes = {_abi.BLK_AB: (B, N, _abi.NX, _abi.NZ), _abi.BLK_BVEC: (B, N, _abi.NX), _abi.BLK_H: (B, N, _abi.NZ, _abi.NZ),
                  _abi.BLK_G: (B, N, _abi.NZ), _abi.BLK_CDE: (B, N, _abi.NE_MAX, _abi.NZ + 1), _abi.BLK_NE: (B, N),
                  _abi.BLK_COST: (B, N + 1), _abi.BLK_DX: (B, N + 1, _abi.NX), _abi.BLK_DU: (B, N, _abi.NU),
                  _abi.BLK_FLOW: (B, N, _abi.NX)}
        a = np.zeros(shapes[what], dtype=np.int32 if what == _abi.BLK_NE else np.float64)
        n = self.lib.hsqp_debug_read(self.h, what, a.ctypes.data_as(C.c_void_p), a.nbytes)
        if n < 0:
            raise HsqpError(n, self.lib.hsqp_last_error(self.h).decode())
        assert n == a.nbytes, (n, a.nbytes)
        return a
