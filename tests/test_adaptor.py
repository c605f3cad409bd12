"""The ocs2::SolverBase / MPC_BASE adaptor (wb_humanoid_mpc_amd/host/HipSqpSolverAdaptor.h) compiled
against stand-in ocs2 headers (tests/stubs/ocs2) and RUN: without a GPU its construction fails loudly; on the GPU three receding-horizon
MPC_BASE::run(t, x) calls — time grid with event nodes, interpolated warm start + WeightCompInitializer tail, device-generated node
parameters, one SQP iteration with the filter line search, PrimalSolution — equal the same loop driven through the ctypes path."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from wb_humanoid_mpc_amd import _abi, solver
from wb_humanoid_mpc_amd.reference import (EVENT_EPS, TargetTrajectories, event_grid, mode_to_contact_flags, pack_reference, pad_targets, swing_config,
                                           tile_gait, velocity_command_targets, weight_compensating_input)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "wb_humanoid_mpc_amd")


def build_driver(tmp_path):
    exe = tmp_path / "adaptor_driver"
    solver.load_library()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "tests", "stubs", "ocs2"), "-I", os.path.join(LIBDIR, "host"),
                           os.path.join(ROOT, "tests", "adaptor", "adaptor_driver.cpp"), "-L", LIBDIR, "-lhsqp_hip", "-Wl,-rpath," + LIBDIR,
                           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", str(exe)])
    return exe


def write_case(tmp_path, model, schedule, targets, x0, horizon, period, calls, state_dim, event_nodes=True, max_nodes=64):
    (tmp_path / "model.bin").write_bytes(bytes(ctypes.string_at(ctypes.addressof(model.desc), ctypes.sizeof(model.desc))))
    sw = swing_config(model)
    vals = [state_dim, model.sqp["dt"], horizon, period, 0.0, calls, int(event_nodes), max_nodes]
    vals += [sw.lift_off_velocity, sw.touch_down_velocity, sw.swing_height, sw.touch_down_height_offset, sw.swing_time_scale, sw.impact_mid,
             sw.impact_lift_velocity, sw.impact_touch_velocity]
    vals += [len(schedule.event_times), *schedule.event_times, *schedule.mode_sequence]
    vals += [len(targets.times), *targets.times, *np.asarray(targets.states)[:, :state_dim].ravel(), *x0[:state_dim]]
    (tmp_path / "case.txt").write_text(" ".join(repr(float(v)) if not isinstance(v, (int, np.integer)) else str(int(v)) for v in vals))


def read_output(path, state_dim):
    tok = open(path).read().split()
    pos, calls = 0, []
    while pos < len(tok):
        n = int(tok[pos]); head = [float(v) for v in tok[pos + 1:pos + 8]]; pos += 8
        rows = np.array(tok[pos:pos + n * (1 + state_dim + _abi.NU)], dtype=float).reshape(n, -1); pos += rows.size
        tau = np.array(tok[pos:pos + _abi.NJ], dtype=float); pos += _abi.NJ
        calls.append(dict(n=n, t=head[0], alpha=head[1], step_type=int(head[2]), cost=head[3], dyn=head[4], eq=head[5], n_post=int(head[6]),
                          times=rows[:, 0], x=rows[:, 1:1 + state_dim], u=rows[:, 1 + state_dim:], tau=tau))
    return calls


def test_adaptor_compiles_and_fails_loudly_without_a_device(tmp_path, model):
    exe = build_driver(tmp_path)
    if solver.load_library().hsqp_device_count() > 0:
        pytest.skip("a GPU is visible: the run itself is covered by the gpu test")
    schedule = tile_gait(model.gaits["walk"], 0.3, 6.0)
    targets = velocity_command_targets(model, (0.3, 0.0, 0.7925, 0.0), 0.0, model.initial_state, 3.0)
    write_case(tmp_path, model, schedule, targets, model.initial_state, 1.05, 0.02, 3, _abi.NX)
    r = subprocess.run([str(exe), str(tmp_path / "model.bin"), str(tmp_path / "case.txt"), str(tmp_path / "out.txt")], capture_output=True, text=True)
    assert r.returncode == 3 and "runtime_error" in r.stdout and "(-2)" in r.stdout, (r.returncode, r.stdout, r.stderr)


PUBLISHER = "/root/reference/humanoid_nmpc/humanoid_common_mpc_ros2/src/benchmarks/SqpBenchmarksPublisher.cpp"
PUBLISHER_MAIN = r"""
#include <cstdio>
#include <humanoid_common_mpc_ros2/benchmarks/SqpBenchmarksPublisher.h>
// the reference's publisher object file is linked in; this driver only proves that its types resolve to the adaptor's
static_assert(std::is_same<ocs2::SqpSolver, ocs2::humanoid::HipSqpSolverAdaptor>::value, "alias");
static_assert(std::is_same<ocs2::SqpSolver::Benchmarks, ocs2::humanoid::HipSqpBenchmarks>::value, "nested Benchmarks type");
int main() {
  auto node = std::make_shared<rclcpp::Node>();
  ocs2::humanoid::SqpBenchmarksPublisher pub(node, nullptr);       // (stores the pointer; a solver needs a GPU)
  ocs2::PrimalSolution empty;
  pub.postSolverRun(empty);                                        // empty trajectory: returns before it touches the solver
  std::printf("%s\n", node->topic.c_str());
  return 0;
}
"""


@pytest.mark.skipif(not os.path.exists(PUBLISHER), reason="/root/reference is only present in the build container")
def test_reference_benchmarks_publisher_compiles_unchanged_against_the_adaptor(tmp_path):
    """SURVEY §8f-3 / VERDICT r4 'missing 5': humanoid_common_mpc_ros2's SqpBenchmarksPublisher takes a `const ocs2::SqpSolver*` and reads
    `SqpSolver::Benchmarks` (SqpBenchmarksPublisher.cpp:34,44-57).  With the one-line alias of tests/stubs/ros2/ocs2_sqp/SqpSolver.h and the adaptor's
    nested `Benchmarks` type the reference's file compiles UNCHANGED, in place, and links; ROS 2 itself is a stand-in (tests/stubs/ros2)."""
    solver.load_library()
    inc = ["-I", os.path.join(ROOT, "tests", "stubs", "ros2"), "-I", os.path.join(ROOT, "tests", "stubs", "ocs2"), "-I", os.path.join(LIBDIR, "host"),
           "-I", os.path.join(ROOT, "include"), "-I", "/root/reference/humanoid_nmpc/humanoid_common_mpc_ros2/include"]
    obj = tmp_path / "publisher.o"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-c", *inc, PUBLISHER, "-o", str(obj)])
    main = tmp_path / "main.cpp"
    main.write_text(PUBLISHER_MAIN)
    exe = tmp_path / "pub"
    subprocess.check_call(["g++", "-std=c++17", "-O1", *inc, str(main), str(obj), "-L", LIBDIR, "-lhsqp_hip", "-Wl,-rpath," + LIBDIR,
                           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", str(exe)])
    assert subprocess.check_output([str(exe)], text=True).strip() == "/humanoid/mpc_benchmarks"


def interp_rows(t, v, time):
    """ocs2 LinearInterpolation with duplicated (event) stamps: the post-event row at an event time."""
    if time <= t[0]:
        return v[0].copy()
    if time >= t[-1]:
        return v[-1].copy()
    i = int(np.searchsorted(t, time, side="right"))
    h = t[i] - t[i - 1]
    a = (time - t[i - 1]) / h if h > 0 else 1.0
    return (1.0 - a) * v[i - 1] + a * v[i]


@pytest.mark.gpu
@pytest.mark.parametrize("formulation", ["wb", "centroidal"])
def test_adaptor_receding_horizon_equals_the_ctypes_path(tmp_path, model, cmodel, formulation):
    from wb_humanoid_mpc_amd.reference import centroidal_velocity_command_targets
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    cent = formulation == "centroidal"
    m = cmodel if cent else model
    nx = _abi.CNX if cent else _abi.NX
    dt, horizon, period, calls = m.sqp["dt"], (0.6 if cent else 1.05), 0.02, 3
    schedule = tile_gait(m.gaits["walk"], 0.3, 6.0)                 # stance, then walking from t = 0.3 s: switches inside every horizon
    x0 = m.initial_state.copy()
    targets = (centroidal_velocity_command_targets if cent else velocity_command_targets)(m, (0.3, 0.0, 0.7925, 0.0), 0.0, x0, 3.0)
    write_case(tmp_path, m, schedule, targets, x0, horizon, period, calls, nx)
    exe = build_driver(tmp_path)
    # the model description comes from the exported image through the C++ loader (host/HipSqpModelIO.h): no Python-made struct in the loop
    image = os.path.join(LIBDIR, "data", "g1_centroidal.json" if cent else "g1_wb.json")
    r = subprocess.run([str(exe), image, str(tmp_path / "case.txt"), str(tmp_path / "out.txt")], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "ok calls=3 preSolverRun=3" in r.stdout and "LQ Approximation" in r.stdout
    got = read_output(tmp_path / "out.txt", nx)
    assert len(got) == calls
    # ---- the same loop through the ctypes path
    s = HipSqpSolver(m, max_nodes=64, max_batch=1)
    padded = pad_targets(targets) if cent else targets
    try:
        t, xm = 0.0, x0.copy()
        prev = None
        for c in range(calls):
            dts, node_times = event_grid(t, t + horizon, dt, schedule.event_times)
            times = node_times - EVENT_EPS * (np.concatenate([[1.0], dts]) == 0.0)
            N = len(dts)
            x, u = np.zeros((N + 1, _abi.NX)), np.zeros((N, _abi.NU))
            for k in range(N + 1):
                if prev is not None and times[k] <= prev["times"][-1]:
                    x[k, :nx] = interp_rows(prev["times"], prev["x"], times[k])
                    if k < N:
                        u[k] = interp_rows(prev["times"], prev["u"], times[k])
                else:
                    x[k, :nx] = xm if k == 0 else x[k - 1, :nx]
                    if k < N:
                        u[k] = weight_compensating_input(m, mode_to_contact_flags(schedule.mode_at(node_times[k])))
            x0p = np.zeros(_abi.NX); x0p[:nx] = xm
            s.upload_reference(x0p, x, u, dts, t, *pack_reference([schedule], [padded]), swing_config(m), node_times=node_times)
            s.iterate(1, take_step=True, kkt=True, linesearch=True)
            out = s.download()
            g = got[c]
            assert g["n"] == N + 1 and g["n_post"] == int((dts == 0.0).sum()) and g["n_post"] >= 1
            np.testing.assert_allclose(g["times"], times, rtol=0, atol=1e-12)
            assert g["alpha"] == out["alpha"][0] and g["step_type"] == out["step_type"][0]
            assert np.abs(g["x"] - out["x"][0][:, :nx]).max() <= 1e-8
            uo = out["u"][0].copy()
            pre = np.flatnonzero(dts == 0.0)
            pre = pre[pre >= 1]
            assert len(pre) >= 1
            uo[pre] = uo[pre - 1]             # pre-event nodes repeat the input before them (upstream toPrimalSolution)
            assert np.abs(g["u"][:-1] - uo).max() <= 1e-7 and np.array_equal(g["u"][-1], g["u"][-2])
            assert all(np.array_equal(g["u"][k], g["u"][k - 1]) for k in pre)
            assert np.isclose(g["cost"], out["perf_after"][0]["cost"], rtol=1e-9, atol=1e-9)
            xs, us, tau = s.evaluate_policy(np.array([period]))
            np.testing.assert_allclose(g["tau"], tau[0], rtol=0, atol=1e-6 * max(1.0, np.abs(tau).max()))
            prev = dict(times=times, x=out["x"][0][:, :nx].copy(), u=np.vstack([uo, uo[-1:]]))
            xm = xs[0][:nx].copy()
            t += period
    finally:
        s.close()
