"""The C-ABI library: every symbol of include/hsqp.h is exported, struct layouts agree, and the product
path refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from wb_humanoid_mpc_amd import _abi, solver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "hsqp.h")).read()
    return sorted(set(re.findall(r"\b(hsqp_[a-z_]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    assert _header_functions() == sorted([
        "hsqp_create", "hsqp_destroy", "hsqp_solve", "hsqp_upload", "hsqp_iterate_device", "hsqp_download",
        "hsqp_debug_read", "hsqp_last_kernel_ms", "hsqp_last_error", "hsqp_scan_fallbacks", "hsqp_version", "hsqp_device_count",
        "hsqp_linesearch_defaults", "hsqp_set_linesearch", "hsqp_upload_reference", "hsqp_joint_torques", "hsqp_evaluate_policy",
        "hsqp_upload_device", "hsqp_download_device", "hsqp_last_iterations", "hsqp_iteration_log", "hsqp_update_weights", "hsqp_host_register", "hsqp_host_unregister",
        "hsqp_scan_backoffs", "hsqp_get_term_weights", "hsqp_update_term_weights", "hsqp_set_scan_backoff_persistent", "hsqp_abi_version",
        "hsqp_comm_unique_id", "hsqp_comm_create", "hsqp_comm_destroy", "hsqp_comm_create_error", "hsqp_comm_last_error", "hsqp_comm_rank", "hsqp_comm_world",
        "hsqp_comm_shard", "hsqp_comm_shard_of", "hsqp_comm_broadcast", "hsqp_comm_scatter_rows", "hsqp_comm_gather_rows", "hsqp_comm_max", "hsqp_comm_barrier"])


def test_library_exports_every_declared_symbol():
    lib = solver.load_library()
    for name in _header_functions():
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.hsqp_version()
    # the binary interface revision: header, library and Python binding agree (a caller checks this before it passes structs: ADVICE r4)
    hdr = int(re.search(r"#define\s+HSQP_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "hsqp.h")).read()).group(1))
    assert lib.hsqp_abi_version() == hdr == _abi.ABI_VERSION and f"abi {hdr}".encode() in lib.hsqp_version()


def test_comm_split_equals_the_python_hosts_and_creation_fails_loudly_without_a_gpu():
    """hsqp_comm_* (the batch axis over GPUs behind the C ABI): the block of a rank is the one wb_humanoid_mpc_amd/distributed.py::shard_range
    gives the torch.distributed host (contiguous ceil(B / world) instances, empty blocks for the last ranks of a small batch); arguments are
    checked; and without a HIP device hsqp_comm_create reports HSQP_ERR_NO_DEVICE (no host path) — RCCL is not even loaded for that."""
    import ctypes as C
    from wb_humanoid_mpc_amd.distributed import shard_range
    lib = solver.load_library()
    lo, hi = C.c_int(), C.c_int()
    for B in (0, 1, 5, 32, 256, 1000, 1024):
        for world in (1, 2, 3, 8):
            for rank in range(world):
                assert lib.hsqp_comm_shard_of(B, world, rank, C.byref(lo), C.byref(hi)) == _abi.OK
                assert (lo.value, hi.value) == shard_range(B, world, rank), (B, world, rank)
    assert lib.hsqp_comm_shard_of(8, 2, 2, C.byref(lo), C.byref(hi)) == _abi.ERR_BAD_ARG
    assert lib.hsqp_comm_shard_of(-1, 2, 0, C.byref(lo), C.byref(hi)) == _abi.ERR_BAD_ARG
    assert lib.hsqp_comm_unique_id(None) == _abi.ERR_BAD_ARG
    comm = C.c_void_p()
    ident = C.create_string_buffer(_abi.COMM_ID_BYTES)
    assert lib.hsqp_comm_create(C.byref(comm), ident, 2, 2, 0) == _abi.ERR_BAD_ARG and not comm.value     # rank outside the world
    assert lib.hsqp_comm_create(C.byref(comm), None, 0, 1, 0) == _abi.ERR_BAD_ARG
    if lib.hsqp_device_count() == 0:
        assert lib.hsqp_comm_create(C.byref(comm), ident, 0, 1, 0) == _abi.ERR_NO_DEVICE and not comm.value
        assert b"no HIP device" in lib.hsqp_comm_create_error()
    assert lib.hsqp_comm_rank(None) == -1 and lib.hsqp_comm_world(None) == 0
    lib.hsqp_comm_destroy(None)


def test_struct_sizes_match_the_c_compiler(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "hsqp.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(hsqp_body),sizeof(hsqp_frame),sizeof(hsqp_model_desc),sizeof(hsqp_settings),sizeof(hsqp_problem),"
                   "sizeof(hsqp_perf),sizeof(hsqp_timings),sizeof(hsqp_solution),sizeof(hsqp_linesearch_settings),sizeof(hsqp_swing_config),sizeof(hsqp_reference),sizeof(hsqp_term_weights));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(s) for s in subprocess.check_output([str(exe)]).split()]
    assert sizes == [C.sizeof(t) for t in (_abi.Body, _abi.Frame, _abi.ModelDesc, _abi.Settings, _abi.Problem, _abi.Perf,
                                           _abi.Timings, _abi.Solution, _abi.LinesearchSettings, _abi.SwingConfig, _abi.Reference, _abi.TermWeights)]


def test_linesearch_defaults_follow_task_info():
    """g_max / g_min / deltaTol come from the reference's task.info (sqp block), the rest are the upstream ocs2 defaults."""
    lib = solver.load_library()
    s = _abi.LinesearchSettings()
    lib.hsqp_linesearch_defaults(C.byref(s))
    assert (s.g_max, s.g_min, s.delta_tol) == (1e-2, 1e-6, 1e-4)
    assert (s.gamma_c, s.armijo_factor, s.alpha_decay, s.alpha_min, s.cost_tol) == (1e-6, 1e-4, 0.5, 1e-4, 1e-4)


def test_no_cpu_fallback(model):
    """Without a HIP device hsqp_create must fail loudly with HSQP_ERR_NO_DEVICE (skipped when a GPU is present)."""
    lib = solver.load_library()
    if lib.hsqp_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(solver.HsqpError) as ei:
        solver.HipSqpSolver(model, max_nodes=4)
    assert ei.value.code == _abi.ERR_NO_DEVICE


def test_product_sources_never_reference_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "wb_humanoid_mpc_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "hsqp_oracle" not in text and "liborc" not in text and "hostemu" not in text.replace("tests/hostemu", ""), f


def test_parallel_riccati_flag_is_validated_before_the_device_is_touched(model):
    """HSQP_FLAG_PARALLEL_RICCATI excludes HSQP_FLAG_SERIAL_RICCATI: BAD_ARG, also without a GPU; on its own it is valid for both formulations
    (whole-body: opt-in only, csrc/hsqp_scan.h)."""
    import ctypes as C
    from wb_humanoid_mpc_amd import _abi, load_model, solver
    lib = solver.load_library()
    h = C.c_void_p()
    cm = load_model(formulation="centroidal")
    for m in (model, cm):
        st = _abi.Settings(max_nodes=4, max_batch=1, device=0, flags=_abi.FLAG_PARALLEL_RICCATI | _abi.FLAG_SERIAL_RICCATI)
        assert lib.hsqp_create(C.byref(m.desc), C.byref(st), C.byref(h)) == _abi.ERR_BAD_ARG
        assert b"PARALLEL_RICCATI" in lib.hsqp_last_error(None)
        st = _abi.Settings(max_nodes=4, max_batch=1, device=0, flags=_abi.FLAG_PARALLEL_RICCATI)
        rc = lib.hsqp_create(C.byref(m.desc), C.byref(st), C.byref(h))
        assert rc != _abi.ERR_BAD_ARG            # no GPU here: HSQP_ERR_NO_DEVICE; on a GPU box: HSQP_OK
        if rc == 0:
            lib.hsqp_destroy(h)


def test_missing_rccl_is_an_error_code_not_a_crash(tmp_path):
    """with no RCCL to bind, hsqp_comm_unique_id must return HSQP_ERR_HIP with a message (round 5 read dlerror() twice: the second
    read is NULL and std::string + NULL is a crash).  HSQP_RCCL_LIB names the library exclusively, so a path that does not exist is a host without
    RCCL; run in a process of its own (the binding is cached per process).  No GPU needed: the identifier is made on the host."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "from wb_humanoid_mpc_amd import _abi, solver\n"
        "lib = solver.load_library()\n"
        "ident = C.create_string_buffer(_abi.COMM_ID_BYTES)\n"
        "rc = lib.hsqp_comm_unique_id(ident)\n"
        "msg = lib.hsqp_comm_create_error().decode()\n"
        "print(rc, '|', msg)\n"
        "sys.exit(0 if rc == _abi.ERR_HIP and 'librccl not found' in msg and 'HSQP_RCCL_LIB' in msg else 1)\n")
    env = dict(os.environ, HSQP_RCCL_LIB=str(tmp_path / "no_such_librccl.so"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, (r.stdout, r.stderr)
