"""Generates tests/golden/ref_assembly.npz from the reference's ASSEMBLY files compiled in place (oracle/_ref/libref_terms.so, second half of
oracle/ref_terms_driver.cpp; `make -C oracle ref`).  Run in the build container:

    python tests/golden/make_ref_assembly_golden.py

What is recorded:
* nodes of a perturbed walk-gait problem (double support, left swing, right swing): (x, u, node parameters), the ORACLE's foot kinematics
  and their Jacobians at (x, u) — the reference's own end-effector kinematics are CppAD-generated and unbuildable here —, and what the
  reference's EndEffectorDynamicsAccelerationsConstraint (behind ZeroAccelerationConstraintCppAd) / EndEffectorDynamicsLinearAccConstraint
  return when those kinematics are handed to them through the EndEffectorDynamics interface, with the configurations WBMpcInterface /
  WBMpcPreComputation build from the task file's foot-constraint gains and the planner's z references;
* StateInputQuadraticCost's deviation (x - x_nom(t) with the arm-swing reference, u - weight compensation on the contact flags) on the walk
  schedule of the ref_terms fixture;
* JointLimitsSoftConstraint (value, gradient, Hessian diagonal) on joint angles inside, at and beyond the limits, with the stand-in penalty;
* WeightCompInitializer::compute on the walk schedule."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

from hsqp_oracle import Oracle  # noqa: E402
from ref_terms import RefTerms  # noqa: E402
from wb_humanoid_mpc_amd import _abi, load_model  # noqa: E402
from wb_humanoid_mpc_amd.reference import LF, RF, STANCE, FLY, make_problem, tile_gait, velocity_command_targets  # noqa: E402


def mode_of(flags):
    return {(True, True): STANCE, (True, False): LF, (False, True): RF, (False, False): FLY}[tuple(bool(f) for f in flags)]


def main():
    m = load_model()
    ref, orc = RefTerms(m.nj), Oracle(m)
    rng = np.random.default_rng(20250927)
    out = {}
    gains = np.array([getattr(m.desc, k) for k in RefTerms.GAIN_KEYS])
    out["gains"] = gains
    # ---- a9 / a10: nodes of a perturbed walk problem
    x0, x, u, par, dt = make_problem(m, n_nodes=40, batch=1, perturb=True, seed=4)
    x, u, par = x[0], u[0], par[0]
    u = u + 5.0 * rng.standard_normal(u.shape)                       # off the weight-compensating input
    x = x + 0.05 * rng.standard_normal(x.shape)
    flags = par[:-1, _abi.P_CONTACT:_abi.P_CONTACT + 2] > 0.5
    picks = []
    for want in ((True, True), (True, False), (False, True)):
        ks = [k for k in range(len(u)) if tuple(flags[k]) == want]
        picks += ks[:: max(1, len(ks) // 3)][:3]
    picks = sorted(picks)
    assert len({tuple(flags[k]) for k in picks}) == 3
    out["node.x"], out["node.u"], out["node.par"], out["node.dt"] = x[picks], u[picks], par[picks], np.array(dt)
    kin, jac = [], []
    st = {k: [] for k in ("value", "f", "dfdx", "dfdu", "active")}
    sw = {k: [] for k in ("value", "f", "dfdx", "dfdu")}
    for k in picks:
        kk, _, J = orc.foot_kinematics(x[k], u[k], jac=True)
        kin.append(kk); jac.append(J)
        ev, seq = np.array([1.0]), np.array([mode_of(flags[k]), STANCE], dtype=np.int32)
        for f in range(2):
            r = ref.stance_foot_constraint(gains, f, ev, seq, 0.5, kk[f], J[f])
            for key in st:
                st[key].append(r[key])
            zref = par[k, _abi.P_SWING + 3 * f:_abi.P_SWING + 3 * f + 3]
            r = ref.swing_foot_constraint(gains, zref, kk[f], J[f])
            for key in sw:
                sw[key].append(r[key])
    out["node.kin"], out["node.jac"] = np.array(kin), np.array(jac)
    n = len(picks)
    for key, v in st.items():
        out[f"stance.{key}"] = np.array(v).reshape(n, 2, *np.shape(v[0]))
    for key, v in sw.items():
        out[f"swing.{key}"] = np.array(v).reshape(n, 2, *np.shape(v[0]))
    # ---- a5: the quadratic cost's deviation on a walk schedule (as the ref_terms fixture's reference-manager cases)
    sched = tile_gait(m.gaits["walk"], 0.3, 4.0)
    ev, seq = np.asarray(sched.event_times), np.asarray(sched.mode_sequence, dtype=np.int32)
    targets = velocity_command_targets(m, (0.4, 0.1, 0.7925, 0.2), 0.0, m.initial_state.copy(), 2.0)
    tt, ts = np.asarray(targets.times), np.asarray(targets.states)
    arm = np.array(list(m.desc.arm_swing_joint), dtype=np.int32)
    qt = np.linspace(0.05, 1.9, 12)
    xs = np.tile(m.initial_state, (len(qt), 1)) + 0.1 * rng.standard_normal((len(qt), m.nx))
    us = 30.0 * rng.standard_normal((len(qt), m.nu))
    Q, R = np.array(m.desc.Q), np.array(m.desc.R)
    out["cost.event_times"], out["cost.mode_sequence"], out["cost.tt"], out["cost.ts"] = ev, seq, tt, ts
    out["cost.times"], out["cost.x"], out["cost.u"], out["cost.total_mass"] = qt, xs, us, np.array(orc.total_mass())
    dxs, dus, vals = [], [], []
    for t, xx, uu in zip(qt, xs, us):
        a, b, c = ref.state_input_quadratic_cost(arm, orc.total_mass(), ev, seq, tt, ts, True, 0.0, 2.0, Q, R, xx, uu, t)
        dxs.append(a); dus.append(b); vals.append(c)
    out["cost.dx"], out["cost.du"], out["cost.value"] = np.array(dxs), np.array(dus), np.array(vals)
    # ---- a18: the initializer on the same schedule
    ini_u, ini_x = [], []
    for t, xx in zip(qt, xs):
        a, b = ref.weight_comp_initializer(orc.total_mass(), ev, seq, t, t + 0.035, xx)
        ini_u.append(a); ini_x.append(b)
    out["init.u"], out["init.next_state"] = np.array(ini_u), np.array(ini_x)
    # ---- a14: joint limits with the stand-in penalty: states inside, near (within delta) and beyond the limits
    mu, delta = m.desc.joint_limit_barrier.mu, m.desc.joint_limit_barrier.delta
    lo, hi = np.asarray(m.q_lo), np.asarray(m.q_hi)
    jl_x = np.tile(m.initial_state, (8, 1))
    for i in range(8):
        q = lo + rng.uniform(0.0, 1.0, m.nj) * (hi - lo)
        near = rng.random(m.nj) < 0.4
        q[near] = np.where(rng.random(near.sum()) < 0.5, lo[near], hi[near]) + rng.uniform(-1.5 * delta, 1.5 * delta, near.sum())
        jl_x[i, 6:6 + m.nj] = q
    out["jl.x"], out["jl.mu_delta"] = jl_x, np.array([mu, delta])
    f, g, h = [], [], []
    for xx in jl_x:
        a, b, c = ref.joint_limits(lo, hi, mu, delta, xx)
        f.append(a); g.append(b); h.append(c)
    out["jl.f"], out["jl.dfdx"], out["jl.dfdxx_diag"] = np.array(f), np.array(g), np.array(h)
    path = os.path.join(ROOT, "tests", "golden", "ref_assembly.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
