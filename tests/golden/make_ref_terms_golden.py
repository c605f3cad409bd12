"""Generates tests/golden/ref_terms.npz from MORE of the reference's own compiled code (oracle/_ref/libref_terms.so, built from
/root/reference in place by `make -C oracle ref`; oracle/ref_terms_driver.cpp).  Run in the build container:

    python tests/golden/make_ref_terms_golden.py

Contents: the whole-body robot model's layout and accessors on a random (x, u); FrictionForceConeConstraint (value, dfdu, dfduu, dfdxx
diagonal, isActive) and ZeroWrenchConstraint (value, dfdu, isActive) of both contacts on random inputs at times inside stance / single
support / flight phases of a run-gait schedule; SwitchedModelReferenceManager::getDesiredState (arm-swing reference on the current yaw),
getPhaseVariable and getContactFlags on a walk schedule; EndEffectorDynamicsWeights::getWeights(task.info).toVector();
WBMpcTargetTrajectoriesCalculator::commandedVelocityToTargetTrajectories on reference.info (filter at steady state + one first call)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

from ref_terms import RefTerms  # noqa: E402
from wb_humanoid_mpc_amd import load_model  # noqa: E402
from wb_humanoid_mpc_amd.reference import tile_gait, velocity_command_targets  # noqa: E402

TASK = "/root/reference/robot_models/unitree_g1/g1_wb_mpc/config/mpc/task.info"
REFERENCE_INFO = "/root/reference/robot_models/unitree_g1/g1_wb_mpc/config/command/reference.info"


def main():
    m = load_model()
    ref = RefTerms(m.nj)
    rng = np.random.default_rng(20250926)
    out = dict(layout_keys=np.array(sorted(ref.layout())), layout_values=np.array([ref.layout()[k] for k in sorted(ref.layout())]))
    x, u = rng.standard_normal(m.nx), 50.0 * rng.standard_normal(m.nu)
    out["acc.x"], out["acc.u"] = x, u
    for k, v in ref.accessors(x, u).items():
        out[f"acc.{k}"] = v
    # constraints on a run-gait schedule (stance, single support and flight phases)
    sched = tile_gait(m.gaits["run"], 0.3, 3.0)
    ev, seq = np.asarray(sched.event_times), np.asarray(sched.mode_sequence, dtype=np.int32)
    out["con.event_times"], out["con.mode_sequence"] = ev, seq
    cfg = np.array([m.desc.friction_mu, m.desc.friction_reg, m.desc.friction_grip, m.desc.friction_hess_shift])
    out["con.cfg"] = cfg
    times = np.concatenate([[0.1], 0.5 * (ev[:-1] + ev[1:])[:8]])
    U = 80.0 * rng.standard_normal((len(times), m.nu))
    U[0, :2] = 0.0                       # Fx = Fy = 0: the regularised apex of the cone
    out["con.times"], out["con.u"] = times, U
    fr = {k: [] for k in ("f", "dfdu", "dfduu", "dfdxx_diag", "active")}
    zw = {k: [] for k in ("f", "dfdu", "active")}
    for t, uu in zip(times, U):
        for c in range(2):
            r = ref.friction_cone(cfg, c, ev, seq, t, x, uu)
            for k in fr:
                fr[k].append(r[k])
            r = ref.zero_wrench(c, ev, seq, t, x, uu)
            for k in zw:
                zw[k].append(r[k])
    for k, v in fr.items():
        out[f"fric.{k}"] = np.array(v).reshape(len(times), 2, *np.shape(v[0]))
    for k, v in zw.items():
        out[f"zw.{k}"] = np.array(v).reshape(len(times), 2, *np.shape(v[0]))
    # reference manager on a walk schedule: desired state with the arm-swing reference, phase variable, contact flags
    sched = tile_gait(m.gaits["walk"], 0.3, 4.0)
    ev, seq = np.asarray(sched.event_times), np.asarray(sched.mode_sequence, dtype=np.int32)
    x0 = m.initial_state.copy()
    targets = velocity_command_targets(m, (0.4, 0.1, 0.7925, 0.2), 0.0, x0, 2.0)
    tt, ts = np.asarray(targets.times), np.asarray(targets.states)
    arm = np.array(list(m.desc.arm_swing_joint), dtype=np.int32)
    out["des.event_times"], out["des.mode_sequence"], out["des.tt"], out["des.ts"], out["des.arm"] = ev, seq, tt, ts, arm
    qt = np.linspace(0.05, 1.9, 38)
    states = np.tile(x0, (len(qt), 1))
    states[:, 3] = rng.uniform(-3.0, 3.0, len(qt))          # current yaw
    out["des.times"], out["des.states"] = qt, states
    xn, ph, fl, xn_off = [], [], [], []
    for t, s in zip(qt, states):
        a, b, c = ref.desired_state(arm, ev, seq, tt, ts, True, 0.0, 2.0, s, t)
        xn.append(a); ph.append(b); fl.append(c)
        xn_off.append(ref.desired_state(arm, ev, seq, tt, ts, False, 0.0, 2.0, s, t)[0])
    out["des.xnom"], out["des.phase"], out["des.flags"], out["des.xnom_no_arm_swing"] = np.array(xn), np.array(ph), np.array(fl), np.array(xn_off)
    out["foot_weights"] = ref.foot_weights(TASK, "task_space_foot_cost_weights.")
    # velocity-command target generator on the reference's own reference.info: random initial states (yaw, base velocity), commands and
    # horizons; 200 identical calls each (the command filter is a function-local static of the reference)
    cmds = np.column_stack([rng.uniform(-0.8, 0.8, 6), rng.uniform(-0.4, 0.4, 6), rng.uniform(0.6, 0.85, 6), rng.uniform(-0.6, 0.6, 6)])
    x0s = np.tile(m.initial_state, (6, 1)) + 0.2 * rng.standard_normal((6, m.nx))
    hor, t0s = rng.uniform(0.5, 3.0, 6), rng.uniform(0.0, 5.0, 6)
    tt, ts = [], []
    for c, xx, h, t0 in zip(cmds, x0s, hor, t0s):
        a, b = ref.velocity_targets(REFERENCE_INFO, h, c, t0, xx)
        tt.append(a); ts.append(b)
    out["tgt.cmd"], out["tgt.x0"], out["tgt.horizon"], out["tgt.t0"], out["tgt.times"], out["tgt.states"] = cmds, x0s, hor, t0s, np.array(tt), np.array(ts)
    # the first call after a fresh command: the filter's transient (documented, not reproduced by the host mirror)
    a, b = ref.velocity_targets(REFERENCE_INFO, 2.0, np.array([5.0, 0.0, 0.7925, 0.0]), 0.0, m.initial_state, calls=1)
    out["tgt.first_call_states"] = b
    path = os.path.join(ROOT, "tests", "golden", "ref_terms.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
