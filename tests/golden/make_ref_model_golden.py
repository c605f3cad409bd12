"""Generates tests/golden/ref_model.npz from the reference's ASSEMBLY of rigid-body quantities compiled in place against a mock of Pinocchio
(oracle/_ref/libref_model.so, oracle/ref_model_driver.cpp; `make -C oracle ref`).  Run in the build container:

    python tests/golden/make_ref_model_golden.py

What is recorded at a handful of perturbed states / inputs, two of them with the feet close enough for active
collision rows:
* the ORACLE's joint-space inertia matrix M (projected Newton-Euler, all 29 x 29), nle, and the LOCAL_WORLD_ALIGNED Jacobians of the two contact frames —
  handed to the reference's computeStateDerivative / computeBaseAcceleration (humanoid_wb_mpc/src/dynamics/DynamicsHelperFunctions.cpp:52-134 over
  humanoid_common_mpc/src/pinocchio_model/DynamicsHelperFunctions.cpp:197-218) through the mock's crba / nonLinearEffects / computeFrameJacobian — and what
  the reference returns: xdot, a_b;
* the ten collision-frame positions (oracle placements + the model's frame offsets) and FootCollisionConstraint's 16 values and activity per contact mode;
* the contact frames' rotations and ContactMomentXYConstraintCppAd's four rows per foot;
* the contact frames' velocity / classical acceleration (oracle) and EndEffectorDynamicsFootCost's residual with the parameter vector its getParameters
  builds (zero references, plane normal e_z, sqrt weights, impact proximity scaler)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from hsqp_oracle import Oracle  # noqa: E402
from ref_model import RefModel  # noqa: E402
from wb_humanoid_mpc_amd import _abi, load_model  # noqa: E402
from conftest import random_state_input  # noqa: E402

NV = _abi.NV
MODES = {"FLY": 0, "RF": 1, "LF": 2, "STANCE": 3}     # MotionPhaseDefinition.h:47-56; contact flags (left, right): LF = left foot in contact


def collision_frames(desc):
    return [desc.ankle[0], desc.ankle[1], desc.contact[0], desc.contact[1], desc.collision_p1[0], desc.collision_p1[1], desc.collision_p2[0],
            desc.collision_p2[1], desc.knee[0], desc.knee[1]]


def handed_in(model, orc, x, u):
    """What the mock Pinocchio is given: the oracle's own quantities at (x, u)."""
    M, nle = orc.full_dynamics(x)
    kin, R, J = orc.foot_kinematics(x, u, jac=True)
    Jf = np.stack([np.vstack([J[f][6:9, NV:2 * NV], J[f][9:12, NV:2 * NV]]) for f in range(2)])     # d(v_lin, omega) / d(generalized velocities)
    Rb, pb = orc.body_placements(x[:NV])
    pos = np.array([pb[f.body] + Rb[f.body] @ np.array(list(f.p)) for f in collision_frames(model.desc)])
    return dict(M=M, nle=nle, J=Jf, kin=kin, R=R, pos=pos)


def foot_cost_parameters(model, impact):
    return np.concatenate([[0.0, 0.0, 0.0, 0.0, 0.0, 1.0], np.zeros(12), np.array(list(model.desc.foot_sqrt_w)), [impact]])   # EndEffectorDynamicsFootCost.cpp:139-149


def reference_outputs(model, ref, h, x, u, impact):
    d = model.desc
    out = {}
    out["xdot"], out["ab"] = ref.state_derivative(h["M"], h["nle"], h["J"][0], h["J"][1], x, u)
    coll, act = [], []
    for name in ("FLY", "RF", "LF", "STANCE"):
        c, a = ref.foot_collision(h["pos"], [d.r_foot, d.r_knee], [1e9], [MODES[name]] * 2, 0.1, x)
        coll.append(c); act.append(a)
    assert all(np.array_equal(coll[0], c) for c in coll)                     # the values do not depend on the mode, the activity does
    out["coll"], out["coll_active"] = coll[0], np.array(act)
    rect = [d.rect_x_min, d.rect_x_max, d.rect_y_min, d.rect_y_max]
    out["mom"] = np.stack([ref.contact_moment(f, h["R"][f], rect, [1e9], [MODES["STANCE"]] * 2, 0.1, x, u)[0] for f in range(2)])
    out["mom_active"] = np.array([[ref.contact_moment(f, h["R"][f], rect, [1e9], [MODES[n]] * 2, 0.1, x, u)[1] for f in range(2)] for n in ("FLY", "RF", "LF", "STANCE")])
    par = foot_cost_parameters(model, impact)
    out["foot_r"] = np.stack([ref.foot_cost(f, h["R"][f], h["kin"][f][6:9], h["kin"][f][9:12], h["kin"][f][12:15], h["kin"][f][15:18], par, x, u) for f in range(2)])
    return out


def samples(model):
    rng = np.random.default_rng(20250928)
    xs, us = [], []
    for k in range(6):
        x, u = random_state_input(model, rng)
        if k >= 4:      # feet close together: hip roll inwards, so that some collision distances fall below the barrier's delta
            names = list(model.joint_names)
            x[6 + names.index("left_hip_roll_joint")] = -0.22 - 0.03 * (k - 4)
            x[6 + names.index("right_hip_roll_joint")] = 0.22 + 0.03 * (k - 4)
        xs.append(x); us.append(u)
    return np.array(xs), np.array(us), 0.35 + 0.1 * np.arange(6)


def main():
    m = load_model()
    ref, orc = RefModel(m.nj), Oracle(m)
    xs, us, impact = samples(m)
    rec = {"x": xs, "u": us, "impact": impact}
    keys_in, keys_out = ("M", "nle", "J", "kin", "R", "pos"), ("xdot", "ab", "coll", "coll_active", "mom", "mom_active", "foot_r")
    cols = {k: [] for k in keys_in + keys_out}
    for x, u, ip in zip(xs, us, impact):
        h = handed_in(m, orc, x, u)
        o = reference_outputs(m, ref, h, x, u, ip)
        for k in keys_in:
            cols[k].append(h[k])
        for k in keys_out:
            cols[k].append(o[k])
    for k, v in cols.items():
        rec[("in." if k in keys_in else "ref.") + k] = np.array(v)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_model.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, {k: v.shape for k, v in rec.items()})
    print("min collision distance per sample:", rec["ref.coll"].min(axis=1), " delta =", m.desc.collision_barrier.delta)


if __name__ == "__main__":
    main()
