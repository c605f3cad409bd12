"""G1 whole-body MPC problem constants -> hsqp_model_desc.

The JSON fixture is produced by tools/export_g1_model.py from the reference's
URDF / task.info / reference.info / gait.info (robot_models/unitree_g1/...).
In a real deployment the ocs2 adaptor fills hsqp_model_desc from the
pinocchio::Model and ModelSettings it already holds.
"""
import json
import math
import os

import numpy as np

from . import _abi

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "g1_wb.json")
_DATA_CENTROIDAL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "g1_centroidal.json")


class G1Model:
    def __init__(self, path=None):
        with open(path or _DATA) as f:
            d = json.load(f)
        self.raw = d
        self.nj, self.nx, self.nu = d["nj"], d["nx"], d["nu"]
        self.centroidal = d.get("formulation", "wb") == "centroidal"
        assert (self.nj, self.nx, self.nu) == (_abi.NJ, _abi.CNX if self.centroidal else _abi.NX, _abi.NU)
        self.joint_names = d["joint_names"]
        self.total_mass = d["total_mass"]
        self.initial_state = np.array(d["initial_state"])
        self.default_joint_state = np.array(d["default_joint_state"])
        self.default_base_height = d["default_base_height"]
        self.gaits = d["gaits"]
        self.swing = d["swing"]
        self.sqp = d["sqp"]
        self.desc = self._build_desc(d)

    @staticmethod
    def _build_desc(d):
        m = _abi.ModelDesc()
        cent = d.get("formulation", "wb") == "centroidal"
        m.formulation = _abi.FORM_CENTROIDAL if cent else _abi.FORM_WB
        m.n_joints = d["nj"]
        for i, b in enumerate(d["bodies"]):
            mb = m.bodies[i]
            mb.parent = b["parent"]
            mb.R[:] = b["R"]
            mb.p[:] = b["p"]
            mb.axis[:] = b["axis"]
            mb.mass = b["mass"]
            mb.com[:] = b["com"]
            mb.inertia[:] = b["inertia"]
            mb.q_lo = b["lo"] if b["lo"] is not None else 0.0
            mb.q_hi = b["hi"] if b["hi"] is not None else 0.0
        for name in ("contact", "collision_p1", "collision_p2", "ankle", "knee"):
            for i in range(2):
                fr = getattr(m, name)[i]
                fr.body = d["frames"][name][i]["body"]
                fr.p[:] = d["frames"][name][i]["p"]
        m.gravity = d["gravity"]
        pad = [0.0] * (_abi.NX - len(d["Q"]))   # centroidal: 35 weights, the padding states carry none
        m.Q[:] = d["Q"] + pad
        m.R[:] = d["R"]
        m.Qf[:] = d["Qf"] + pad
        if cent:
            t = d["torso"]
            m.torso.body = t["body"]
            m.torso.p[:] = t["p"]
            m.torso_R[:] = t["R"]
            m.torso_sqrt_w[:] = [math.sqrt(w) for w in t["weights"]]
            m.cent_foot_sqrt_w[:] = [math.sqrt(w) for w in d["cent_foot_cost_weights"]]
            for f in range(2):
                m.ext_torque_sqrt_w[f][:] = [math.sqrt(w) for w in d["ext_torque"]["weights"][f]]
                m.ext_torque_joint[f][:] = d["ext_torque"]["joints"][f]
        m.foot_sqrt_w[:] = [math.sqrt(w) for w in d["foot_cost_weights"]]
        fc = d["foot_constraint"]
        m.gain_pos_z = fc["positionErrorGain_z"]
        m.gain_ori = fc["orientationErrorGain"]
        m.gain_linvel_z = fc["linearVelocityErrorGain_z"]
        m.gain_linvel_xy = fc["linearVelocityErrorGain_xy"]
        m.gain_angvel = fc["angularVelocityErrorGain"]
        m.gain_linacc_z = fc["linearAccelerationErrorGain_z"]
        m.gain_linacc_xy = fc["linearAccelerationErrorGain_xy"]
        m.gain_angacc = fc["angularAccelerationErrorGain"]
        fr = d["friction"]
        m.friction_mu, m.friction_reg = fr["mu"], fr["regularization"]
        m.friction_grip, m.friction_hess_shift = fr["gripper_force"], fr["hessian_diagonal_shift"]
        m.friction_barrier.mu, m.friction_barrier.delta = fr["barrier_mu"], fr["barrier_delta"]
        r = d["contact_rectangle"]
        m.rect_x_min, m.rect_x_max, m.rect_y_min, m.rect_y_max = r["x_min"], r["x_max"], r["y_min"], r["y_max"]
        m.moment_barrier.mu, m.moment_barrier.delta = d["moment_xy"]["barrier_mu"], d["moment_xy"]["barrier_delta"]
        m.joint_limit_barrier.mu = d["joint_limits"]["barrier_mu"]
        m.joint_limit_barrier.delta = d["joint_limits"]["barrier_delta"]
        c = d["collision"]
        m.r_foot, m.r_knee = c["r_foot"], c["r_knee"]
        m.collision_barrier.mu, m.collision_barrier.delta = c["barrier_mu"], c["barrier_delta"]
        m.arm_swing_joint[:] = d["arm_swing_joints"]
        return m

    @property
    def q_lo(self):
        return np.array([b["lo"] for b in self.raw["bodies"][1:]])

    @property
    def q_hi(self):
        return np.array([b["hi"] for b in self.raw["bodies"][1:]])


def load_model(path=None, formulation="wb"):
    """formulation: "wb" (whole-body acceleration-level, g1_wb_mpc) or "centroidal" (g1_centroidal_mpc)."""
    if path is None and formulation == "centroidal":
        path = _DATA_CENTROIDAL
    return G1Model(path)
