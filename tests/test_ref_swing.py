"""SURVEY.md §8 rows a16/a17 PINNED against reference-compiled code.

oracle/_ref/libref_swing.so is the reference's own CubicSpline.cpp, SplineCpg.cpp, SwingTrajectoryPlanner.cpp and
GaitSchedule.cpp compiled in place from /root/reference (oracle/Makefile target `ref`; stand-ins only for the absent
third-party headers).  tests/golden/ref_swing.npz holds its outputs for every gait template of gait.info
(tests/golden/make_ref_swing_golden.py).  Checked against them:
  * the host mirror wb_humanoid_mpc_amd/reference.py (CubicSpline, SplineCpg, SwingTrajectoryPlanner, tile_gait, mode flags),
  * the device generator hsqp_params.h — kernel source compiled for the host here, the HIP kernel k_params through the C ABI on the GPU.
Where the library itself is present (build container: always; GPU box: it travels with the snapshot) the mirror is also
compared with it directly on random schedules."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from wb_humanoid_mpc_amd import _abi
from wb_humanoid_mpc_amd.reference import (MODE_BY_NAME, CubicSpline, ModeSchedule, SplineCpg, SwingTrajectoryPlanner, TargetTrajectories,
                                           mode_to_contact_flags, pack_reference, swing_config, tile_gait)

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "ref_swing.npz"))
GAITS = [str(g) for g in G["gaits"]]
_dp, _ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
# gait.info's "skip" template has non-monotonic switchingTimes (0.75, 0.08, 1.0): its tiled event times are unsorted and a binary search
# on them is implementation-defined (std::lower_bound, bisect and the kernel's loop may legitimately disagree), so the per-time
# look-ups of the device generator are compared on the fifteen well-formed gaits
GAITS_SORTED = [g for g in GAITS if np.all(np.diff(G[f"{g}.event_times"]) > 0)]
TOL = 1e-12   # the mirror evaluates the same cubic in the same order; z is O(0.1), zddot O(10)


def test_fixture_covers_every_gait_of_gait_info(model):
    assert sorted(model.gaits) == GAITS and len(GAITS) == 16 and set(GAITS) - set(GAITS_SORTED) == {"skip"}


def test_cubic_spline_and_cpg_against_reference_compiled_values():
    a, b = G["cubic.args"]
    s = CubicSpline(a[0], a[1], a[2], b[0], b[1], b[2])
    got = np.array([[s.position(t), s.velocity(t), s.acceleration(t)] for t in G["cubic.t"]])
    np.testing.assert_allclose(got, G["cubic.values"], rtol=0, atol=1e-13)
    l, r = G["cpg.args"]
    c = SplineCpg(tuple(l), float(G["cpg.mid"]), tuple(r))
    got = np.array([[c.position(t), c.velocity(t), c.acceleration(t)] for t in G["cpg.t"]])
    np.testing.assert_allclose(got, G["cpg.values"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("gait", GAITS)
def test_host_planner_on_the_reference_schedule(model, gait):
    """reference.py's planner on the mode schedule the reference's GaitSchedule produced: flags, z, zdot, zddot, impact proximity."""
    sched = ModeSchedule(G[f"{gait}.event_times"], G[f"{gait}.mode_sequence"])
    planner = SwingTrajectoryPlanner(model.swing, sched)
    want, modes = G[f"{gait}.values"], G[f"{gait}.modes"]
    for i, t in enumerate(G["t"]):
        assert sched.mode_at(t) == modes[i]
        for leg in range(2):
            got = (*planner.z_refs(leg, t), planner.impact_proximity(leg, t))
            np.testing.assert_allclose(got, want[i, leg], rtol=0, atol=TOL, err_msg=f"{gait} t={t} leg={leg}")


@pytest.mark.parametrize("gait", GAITS)
def test_tile_gait_equals_the_reference_gait_schedule(model, gait):
    """tile_gait (the mirror of GaitSchedule::tileModeSequenceTemplate used by make_problem) gives the same contact flags and the
    same swing references as the schedule the reference builds by insertModeSequenceTemplate + getModeSchedule."""
    t_insert = float(G["t_insert"])
    sched = tile_gait(model.gaits[gait], t_insert, float(G["upper"]))
    # structurally: the reference's schedule is the initial STANCE phase (event 0.5 of reference.info) followed by exactly tile_gait's
    np.testing.assert_allclose(sched.event_times, G[f"{gait}.event_times"][1:], rtol=0, atol=1e-12)
    assert list(sched.mode_sequence) == list(G[f"{gait}.mode_sequence"][1:])
    if gait == "skip":
        # gait.info's "skip" template has non-monotonic switchingTimes (0.75, 0.08, 1.0): its event times are unsorted and a binary
        # search on them is implementation-defined, so per-time lookups are only compared on the reference's own array above
        return
    planner = SwingTrajectoryPlanner(model.swing, sched)
    want, modes = G[f"{gait}.values"], G[f"{gait}.modes"]
    for i, t in enumerate(G["t"]):
        assert mode_to_contact_flags(sched.mode_at(t)) == mode_to_contact_flags(modes[i])
        for leg in range(2):
            got = (*planner.z_refs(leg, t), planner.impact_proximity(leg, t))
            np.testing.assert_allclose(got, want[i, leg], rtol=0, atol=TOL, err_msg=f"{gait} t={t} leg={leg}")


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "hostemu")])
    return C.CDLL(os.path.join(HERE, "hostemu", "libhsqp_hostemu.so"))


def _device_columns(par):
    """{z, zdot, zddot, impact} per leg and the contact flags from a node-parameter table [n][72]."""
    vals = np.stack([np.concatenate([par[:, _abi.P_SWING + 3 * leg:_abi.P_SWING + 3 * leg + 3], par[:, _abi.P_IMPACT + leg:_abi.P_IMPACT + leg + 1]], axis=1)
                     for leg in range(2)], axis=1)
    return vals, par[:, _abi.P_CONTACT:_abi.P_CONTACT + 2] > 0.5


@pytest.mark.parametrize("gait", GAITS_SORTED)
def test_kernel_source_generator_against_reference_compiled_values(model, emu, gait):
    """hsqp_params.h (the source of k_params) compiled for the host, on the reference's schedule and time grid."""
    ev = np.ascontiguousarray(G[f"{gait}.event_times"])
    seq = np.ascontiguousarray(G[f"{gait}.mode_sequence"], dtype=np.int32)
    t = G["t"]
    n = len(t) - 1
    out = np.zeros((n + 1, _abi.NODE_PARAMS))
    tt, ts = np.array([0.0]), np.zeros((1, _abi.NX))
    cfg = swing_config(model)
    bad = emu.emu_node_params(C.byref(cfg), C.c_double(0.0), 1, len(ev), ev.ctypes.data_as(_dp), seq.ctypes.data_as(_ip), 1, tt.ctypes.data_as(_dp),
                              ts.ctypes.data_as(_dp), C.c_double(t[0]), C.c_double(t[1] - t[0]), n, out.ctypes.data_as(_dp))
    assert bad == 0
    vals, flags = _device_columns(out)
    # the kernel's node times are t0 + k dt, the fixture's np.linspace: equal to ~1e-16 relative; exclude nodes that sit within
    # 1e-9 s of a mode switch (either side is a correct answer there)
    keep = np.array([np.abs(ev - tk).min() > 1e-9 for tk in t])
    want_flags = np.array([mode_to_contact_flags(m) for m in G[f"{gait}.modes"]])
    assert np.array_equal(flags[keep], want_flags[keep])
    np.testing.assert_allclose(vals[keep], G[f"{gait}.values"][keep], rtol=0, atol=1e-11)


@pytest.mark.gpu
def test_device_kernel_against_reference_compiled_values(model):
    """k_params on the GPU (hsqp_upload_reference) for all sixteen gaits as one batch, against the reference-compiled fixture."""
    from wb_humanoid_mpc_amd.solver import HipSqpSolver
    t = G["t"]
    GAITS = GAITS_SORTED
    n, B = 100, len(GAITS)            # the handle's node grid: 101 consecutive points of the fixture's grid
    dt = float(t[1] - t[0])
    scheds = [ModeSchedule(G[f"{g}.event_times"], G[f"{g}.mode_sequence"]) for g in GAITS]
    targets = [TargetTrajectories([0.0], [model.initial_state]) for _ in GAITS]
    x0 = np.tile(model.initial_state, (B, 1))
    x = np.tile(model.initial_state, (B, n + 1, 1))
    u = np.zeros((B, n, _abi.NU))
    s = HipSqpSolver(model, max_nodes=n, max_batch=B)
    try:
        for k0 in (0, 100, 200, 300):
            s.upload_reference(x0, x, u, dt, float(t[k0]), *pack_reference(scheds, targets), swing_config(model))
            par = s.device_params()
            for b, g in enumerate(GAITS):
                vals, flags = _device_columns(par[b])
                sl = slice(k0, k0 + n + 1)
                keep = np.array([np.abs(G[f"{g}.event_times"] - tk).min() > 1e-9 for tk in t[sl]])
                want_flags = np.array([mode_to_contact_flags(m) for m in G[f"{g}.modes"][sl]])
                assert np.array_equal(flags[keep], want_flags[keep]), g
                np.testing.assert_allclose(vals[keep], G[f"{g}.values"][sl][keep], rtol=0, atol=1e-11, err_msg=g)
    finally:
        s.close()


def _ref_or_skip():
    import ref_swing
    if not ref_swing.available():
        pytest.skip("oracle/_ref/libref_swing.so not present and /root/reference not mounted")
    return ref_swing.RefSwing()


def test_mode_numbers_against_reference_compiled_code():
    ref = _ref_or_skip()
    for name, m in MODE_BY_NAME.items():
        assert ref.stance_legs(m) == mode_to_contact_flags(m), name


def test_host_planner_against_the_reference_library_on_random_schedules(model, rng):
    """Random mode sequences (all four modes, random durations) with random planner settings, compared with the compiled
    reference directly; sequences the reference rejects (a swing without lift-off / touch-down) must be rejected by the mirror too."""
    ref = _ref_or_skip()
    accepted = rejected = 0
    for trial in range(60):
        n_ev = int(rng.integers(2, 14))
        ev = np.cumsum(rng.uniform(0.05, 0.7, n_ev))
        seq = rng.integers(0, 4, n_ev + 1)
        if trial % 3:
            seq[0] = seq[-1] = 3      # most schedules start and end in stance like the reference's own
        cfg = dict(model.swing)
        cfg.update(liftOffVelocity=rng.uniform(0, 0.3), touchDownVelocity=-rng.uniform(0, 0.3), swingHeight=rng.uniform(0.02, 0.2),
                   touchDownHeightOffset=rng.uniform(-0.01, 0.01), swingTimeScale=rng.uniform(0.1, 0.6))
        terrain = rng.uniform(-0.05, 0.05)
        t = np.sort(rng.uniform(ev[0] - 0.2, ev[-1] + 0.2, 64))
        ok, want, modes = ref.swing_planner(cfg, ev, seq, t, terrain)
        try:
            planner = SwingTrajectoryPlanner(cfg, ModeSchedule(ev, seq), terrain)
        except RuntimeError:
            assert not ok
            rejected += 1
            continue
        assert ok
        accepted += 1
        for i, tk in enumerate(t):
            assert planner.schedule.mode_at(tk) == modes[i]
            for leg in range(2):
                got = (*planner.z_refs(leg, tk), planner.impact_proximity(leg, tk))
                np.testing.assert_allclose(got, want[i, leg], rtol=1e-12, atol=1e-11)
    assert accepted >= 15 and rejected >= 5
