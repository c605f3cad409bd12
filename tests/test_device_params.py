"""Device-side per-node parameter generation (hsqp_params.h) against the host-side
generators of wb_humanoid_mpc_amd/reference.py: the kernel source compiled for the host (tests/hostemu) on CPU, and the
HIP kernel through the C ABI on the GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from wb_humanoid_mpc_amd import _abi
from wb_humanoid_mpc_amd.reference import make_problem, pack_reference, swing_config

HERE = os.path.dirname(os.path.abspath(__file__))
_dp, _ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "hostemu")])
    return C.CDLL(os.path.join(HERE, "hostemu", "libhsqp_hostemu.so"))


CASES = [("walk", 24, (0.3, 0.0, 0.7925, 0.0)), ("run", 30, (0.8, 0.1, 0.7925, 0.2)), ("stance", 6, (0.0, 0.0, 0.7925, 0.0)),
         ("slow_walk", 40, (0.2, 0.0, 0.78, -0.1)), ("trot", 20, (0.4, 0.0, 0.7925, 0.0))]


@pytest.mark.parametrize("gait,n,v_cmd", CASES)
def test_generated_table_equals_the_host_generators(model, emu, gait, n, v_cmd):
    B = 3
    x0, x, u, par, dt, (schedules, targets, t0) = make_problem(model, n_nodes=n, batch=B, gait=gait, v_cmd=v_cmd, perturb=True, seed=21,
                                                                with_reference=True)
    n_events, ev, seq, tt, ts = pack_reference(schedules, targets)
    cfg = swing_config(model)
    for b in range(B):
        out = np.zeros((n + 1, _abi.NODE_PARAMS))
        evb, seqb, ttb, tsb = np.ascontiguousarray(ev[b]), np.ascontiguousarray(seq[b]), np.ascontiguousarray(tt[b]), np.ascontiguousarray(ts[b])
        bad = emu.emu_node_params(C.byref(cfg), C.c_double(0.0), 1, int(n_events[b]), evb.ctypes.data_as(_dp), seqb.ctypes.data_as(_ip),
                                  tt.shape[1], ttb.ctypes.data_as(_dp), tsb.ctypes.data_as(_dp), C.c_double(t0), C.c_double(dt), n,
                                  out.ctypes.data_as(_dp))
        assert bad == 0
        assert np.array_equal(out[:, _abi.P_CONTACT:_abi.P_CONTACT + 2], par[b][:, _abi.P_CONTACT:_abi.P_CONTACT + 2])
        np.testing.assert_allclose(out, par[b], rtol=0, atol=1e-13)
    # the swing references are exercised: at least one node of a moving gait has a foot in the air with a non-zero height target
    if gait != "stance":
        assert (par[..., _abi.P_CONTACT:_abi.P_CONTACT + 2] == 0).any() and np.abs(par[..., _abi.P_SWING]).max() > 1e-3


def test_unbracketed_swing_phase_is_reported(model, emu):
    """A schedule that ends in the air has no touch-down: the reference's planner throws, the generator reports it."""
    cfg = swing_config(model)
    ev = np.array([0.0, 0.5])
    seq = np.array([3, 2, 0], dtype=np.int32)       # STANCE, LF, FLY: the final flight never lands
    tt, ts = np.array([0.0]), np.zeros((1, _abi.NX))
    out = np.zeros((5, _abi.NODE_PARAMS))
    bad = emu.emu_node_params(C.byref(cfg), C.c_double(0.0), 1, 2, ev.ctypes.data_as(_dp), seq.ctypes.data_as(_ip), 1, tt.ctypes.data_as(_dp),
                              ts.ctypes.data_as(_dp), C.c_double(0.0), C.c_double(0.2), 4, out.ctypes.data_as(_dp))
    assert bad == 1


@pytest.mark.gpu
@pytest.mark.parametrize("gait,n,v_cmd", CASES[:2])
def test_device_table_and_solve_equal_the_uploaded_table(model, gait, n, v_cmd):
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    B = 4
    x0, x, u, par, dt, (schedules, targets, t0) = make_problem(model, n_nodes=n, batch=B, gait=gait, v_cmd=v_cmd, perturb=True, seed=21,
                                                                with_reference=True)
    n_events, ev, seq, tt, ts = pack_reference(schedules, targets)
    s = HipSqpSolver(model, max_nodes=n, max_batch=B)
    try:
        ref = s.run(x0, x, u, par, dt)
        s.upload_reference(x0, x, u, dt, t0, n_events, ev, seq, tt, ts, swing_config(model))
        np.testing.assert_allclose(s.device_params(), par, rtol=0, atol=1e-13)
        s.iterate(1, take_step=True, kkt=True)
        out = s.download()
        sc = max(1.0, np.abs(ref["dx"]).max(), np.abs(ref["du"]).max())
        assert np.abs(out["x"] - ref["x"]).max() <= 1e-9 * sc and np.abs(out["u"] - ref["u"]).max() <= 1e-9 * sc
    finally:
        s.close()


# ---------------------------------------------------------------------------------------------- centroidal formulation
@pytest.mark.parametrize("gait,n,v_cmd", CASES)
def test_centroidal_table_equals_the_host_generators(cmodel, emu, gait, n, v_cmd):
    """The torso task-space reference is generated on the device too (tree pass of hsqp_cent.h at the interpolated target state)
    and must equal the independent numpy kinematics of reference.torso_reference."""
    from wb_humanoid_mpc_amd.reference import make_centroidal_problem
    B = 2
    x0, x, u, par, dt, (schedules, targets, t0) = make_centroidal_problem(cmodel, n_nodes=n, batch=B, gait=gait, v_cmd=v_cmd, perturb=True, seed=21,
                                                                           with_reference=True)
    n_events, ev, seq, tt, ts = pack_reference(schedules, targets)
    cfg = swing_config(cmodel)
    emu.emu_create.restype = C.c_void_p
    err = C.create_string_buffer(256)
    h = C.c_void_p(emu.emu_create(C.byref(cmodel.desc), err, 256))
    assert h.value, err.value
    for b in range(B):
        out = np.zeros((n + 1, _abi.NODE_PARAMS))
        evb, seqb, ttb, tsb = np.ascontiguousarray(ev[b]), np.ascontiguousarray(seq[b]), np.ascontiguousarray(tt[b]), np.ascontiguousarray(ts[b])
        bad = emu.emu_cent_node_params(h, C.byref(cfg), C.c_double(0.0), 1, int(n_events[b]), evb.ctypes.data_as(_dp), seqb.ctypes.data_as(_ip),
                                       tt.shape[1], ttb.ctypes.data_as(_dp), tsb.ctypes.data_as(_dp), C.c_double(t0), C.c_double(dt), n,
                                       out.ctypes.data_as(_dp))
        assert bad == 0
        np.testing.assert_allclose(out, par[b], rtol=0, atol=1e-12)
    assert np.abs(par[..., _abi.PC_TORSO + 7]).max() > 1e-3 or gait == "stance"   # the torso velocity reference is exercised


@pytest.mark.gpu
def test_centroidal_device_table_and_solve_equal_the_uploaded_table(cmodel):
    from wb_humanoid_mpc_amd.reference import make_centroidal_problem
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    gait, n, v_cmd = CASES[0]
    B = 3
    x0, x, u, par, dt, (schedules, targets, t0) = make_centroidal_problem(cmodel, n_nodes=n, batch=B, gait=gait, v_cmd=v_cmd, perturb=True, seed=21,
                                                                           with_reference=True)
    n_events, ev, seq, tt, ts = pack_reference(schedules, targets)
    s = HipSqpSolver(cmodel, max_nodes=n, max_batch=B)
    try:
        ref = s.run(x0, x, u, par, dt)
        s.upload_reference(x0, x, u, dt, t0, n_events, ev, seq, tt, ts, swing_config(cmodel))
        np.testing.assert_allclose(s.device_params(), par, rtol=0, atol=1e-12)
        s.iterate(1, take_step=True, kkt=True)
        out = s.download()
        sc = max(1.0, np.abs(ref["dx"]).max(), np.abs(ref["du"]).max())
        assert np.abs(out["x"] - ref["x"]).max() <= 1e-8 * sc and np.abs(out["u"] - ref["u"]).max() <= 1e-8 * sc
    finally:
        s.close()
