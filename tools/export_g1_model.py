#!/usr/bin/env python3
"""Export the Unitree G1 whole-body MPC problem constants to a flat JSON fixture.

Runs ONLY in the build container (needs /root/reference); the GPU box and the
tests consume the committed JSON (wb_humanoid_mpc_amd/data/g1_wb.json).

What it restates (reference file:line, relative to /root/reference):
  * URDF -> MPC kinematic tree: non-MPC joints fixed and their links lumped into
    the parent body, composite Translation+SphericalZYX base
    (humanoid_nmpc/humanoid_common_mpc/src/pinocchio_model/createPinocchioModel.cpp:60-67,139-182).
  * contact / collision frames (createPinocchioModel.cpp:77-128,
    robot_models/unitree_g1/g1_wb_mpc/config/mpc/task.info:403-419).
  * weights / gains / barrier settings (task.info), foot-cost weight quirk
    (humanoid_nmpc/humanoid_wb_mpc/src/cost/EndEffectorDynamicsCostHelpers.cpp:100-108).
  * gait templates (humanoid_nmpc/humanoid_common_mpc/config/command/gait.info) and
    reference.info defaults.
"""
import json
import math
import os
import re
import sys
import xml.etree.ElementTree as ET

import numpy as np

REF = "/root/reference"
URDF = f"{REF}/robot_models/unitree_g1/g1_description/urdf/g1_29dof.urdf"
TASK = f"{REF}/robot_models/unitree_g1/g1_wb_mpc/config/mpc/task.info"
REFERENCE = f"{REF}/robot_models/unitree_g1/g1_wb_mpc/config/command/reference.info"
GAIT = f"{REF}/humanoid_nmpc/humanoid_common_mpc/config/command/gait.info"


# --------------------------------------------------------------------------- INFO parser
def parse_info(path):
    """Minimal boost::property_tree INFO reader: nested dict of str -> (str | dict)."""
    toks = []
    with open(path) as f:
        for line in f:
            line = line.split(";")[0].split("//")[0]
            toks += re.findall(r'"[^"]*"|[{}]|[^\s{}]+', line)
    pos = 0

    def block():
        nonlocal pos
        d = {}
        while pos < len(toks):
            t = toks[pos]
            if t == "}":
                pos += 1
                return d
            key = t.strip('"')
            pos += 1
            if pos < len(toks) and toks[pos] == "{":
                pos += 1
                d[key] = block()
            else:
                val = toks[pos].strip('"')
                pos += 1
                if pos < len(toks) and toks[pos] == "{":  # "key value {children}" (unused here)
                    pos += 1
                    d[key] = block()
                else:
                    d[key] = val
        return d

    return block()


def info_diag(node, n):
    """loadData::loadEigenMatrix semantics for '(i,j) v' entries with optional 'scaling'."""
    scaling = float(node.get("scaling", 1.0))
    m = np.zeros((n, n))
    for k, v in node.items():
        mm = re.match(r"\((\d+),(\d+)\)", k)
        if mm:
            m[int(mm.group(1)), int(mm.group(2))] = float(v) * scaling
    return m


def info_vec(node, n):
    v = np.zeros(n)
    for k, val in node.items():
        mm = re.match(r"\((\d+),(\d+)\)", k)
        if mm:
            v[int(mm.group(1))] = float(val)
    return v


def info_list(node):
    items = sorted(((int(re.match(r"\[(\d+)\]", k).group(1)), v) for k, v in node.items()), key=lambda t: t[0])
    return [v for _, v in items]


# --------------------------------------------------------------------------- URDF
def rpy_to_R(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def floats(s):
    return np.array([float(x) for x in s.split()])


class Inertia:
    """mass, com (frame of owner), rotational inertia about com expressed in owner frame axes."""

    def __init__(self, m=0.0, c=None, I=None):
        self.m = m
        self.c = np.zeros(3) if c is None else c
        self.I = np.zeros((3, 3)) if I is None else I

    def transformed(self, R, p):
        return Inertia(self.m, R @ self.c + p, R @ self.I @ R.T)

    def __add__(self, o):
        m = self.m + o.m
        if m == 0.0:
            return Inertia()
        c = (self.m * self.c + o.m * o.c) / m

        def shift(b):
            d = b.c - c
            return b.I + b.m * (d @ d * np.eye(3) - np.outer(d, d))

        return Inertia(m, c, shift(self) + shift(o))


def load_urdf(path, fixed_joint_names):
    root = ET.parse(path).getroot()
    links = {}
    for l in root.findall("link"):
        ine = l.find("inertial")
        if ine is None:
            links[l.get("name")] = Inertia()
            continue
        o = ine.find("origin")
        xyz = floats(o.get("xyz", "0 0 0")) if o is not None else np.zeros(3)
        rpy = floats(o.get("rpy", "0 0 0")) if o is not None else np.zeros(3)
        i = ine.find("inertia")
        I = np.array([[float(i.get("ixx")), float(i.get("ixy")), float(i.get("ixz"))],
                      [float(i.get("ixy")), float(i.get("iyy")), float(i.get("iyz"))],
                      [float(i.get("ixz")), float(i.get("iyz")), float(i.get("izz"))]])
        R = rpy_to_R(*rpy)
        links[l.get("name")] = Inertia(float(ine.find("mass").get("value")), xyz, R @ I @ R.T)
    joints = {}
    children = {}
    child_links = set()
    for j in root.findall("joint"):
        o = j.find("origin")
        xyz = floats(o.get("xyz", "0 0 0")) if o is not None else np.zeros(3)
        rpy = floats(o.get("rpy", "0 0 0")) if o is not None else np.zeros(3)
        a = j.find("axis")
        lim = j.find("limit")
        jt = j.get("type")
        if j.get("name") in fixed_joint_names:
            jt = "fixed"
        joints[j.get("name")] = dict(
            name=j.get("name"), type=jt, parent=j.find("parent").get("link"), child=j.find("child").get("link"),
            R=rpy_to_R(*rpy), p=xyz, axis=floats(a.get("xyz")) if a is not None else None,
            lo=float(lim.get("lower")) if lim is not None and jt != "fixed" else None,
            hi=float(lim.get("upper")) if lim is not None and jt != "fixed" else None)
        children.setdefault(j.find("parent").get("link"), []).append(j.get("name"))
        child_links.add(j.find("child").get("link"))
    root_link = [n for n in links if n not in child_links]
    assert len(root_link) == 1
    return links, joints, children, root_link[0]


def build_tree(links, joints, children, root_link):
    """Depth-first, siblings in alphabetical joint-name order (urdfdom keeps joints in a
    std::map, pinocchio::urdf::buildModel walks link->child_joints in that order)."""
    bodies = []  # dict(name, joint, parent, R, p, axis, inertia, lo, hi)
    frames = {}  # name -> (body index, R, p) placement of every URDF link/joint frame in its body

    def visit(link, body_idx, R_bl, p_bl):
        # link frame expressed in body frame: (R_bl, p_bl)
        bodies[body_idx]["inertia"] = bodies[body_idx]["inertia"] + links[link].transformed(R_bl, p_bl)
        frames[link] = (body_idx, R_bl.copy(), p_bl.copy())
        for jn in sorted(children.get(link, [])):
            j = joints[jn]
            R_bj = R_bl @ j["R"]
            p_bj = R_bl @ j["p"] + p_bl
            if j["type"] == "fixed":
                frames[jn] = (body_idx, R_bj.copy(), p_bj.copy())
                visit(j["child"], body_idx, R_bj, p_bj)
            else:
                assert j["type"] == "revolute"
                bodies.append(dict(name=j["child"], joint=jn, parent=body_idx, R=R_bj, p=p_bj, axis=j["axis"],
                                   inertia=Inertia(), lo=j["lo"], hi=j["hi"]))
                idx = len(bodies) - 1
                frames[jn] = (idx, np.eye(3), np.zeros(3))
                visit(j["child"], idx, np.eye(3), np.zeros(3))

    bodies.append(dict(name=root_link, joint="root_joint", parent=-1, R=np.eye(3), p=np.zeros(3), axis=None,
                       inertia=Inertia(), lo=None, hi=None))
    visit(root_link, 0, np.eye(3), np.zeros(3))
    return bodies, frames


def main(out_path, formulation="wb"):
    cent = formulation == "centroidal"
    task = parse_info(TASK.replace("g1_wb_mpc", "g1_centroidal_mpc") if cent else TASK)
    ref = parse_info(REFERENCE.replace("g1_wb_mpc", "g1_centroidal_mpc") if cent else REFERENCE)
    gait = parse_info(GAIT)
    ms = task["model_settings"]
    fixed = info_list(ms["fixedJointNames"])
    links, joints, children, root_link = load_urdf(URDF, fixed)
    bodies, frames = build_tree(links, joints, children, root_link)
    joint_names = [b["joint"] for b in bodies[1:]]
    expected = ["left_hip_pitch_joint", "left_hip_roll_joint", "left_hip_yaw_joint", "left_knee_joint",
                "left_ankle_pitch_joint", "left_ankle_roll_joint", "right_hip_pitch_joint", "right_hip_roll_joint",
                "right_hip_yaw_joint", "right_knee_joint", "right_ankle_pitch_joint", "right_ankle_roll_joint",
                "waist_yaw_joint", "waist_roll_joint", "waist_pitch_joint", "left_shoulder_pitch_joint",
                "left_shoulder_roll_joint", "left_shoulder_yaw_joint", "left_elbow_joint",
                "right_shoulder_pitch_joint", "right_shoulder_roll_joint", "right_shoulder_yaw_joint",
                "right_elbow_joint"]
    assert joint_names == expected, joint_names  # task.info:130-152 order
    nj = len(joint_names)
    total_mass = sum(b["inertia"].m for b in bodies)

    # contact frames (createPinocchioModel.cpp:77-83) and collision points (:92-106)
    contact_parents = info_list(ms["contactParentJointNames"])
    cft = task["contacts"]["contact_frame_translation"]
    t_c = np.array([float(cft["x"]), float(cft["y"]), float(cft["z"])])
    rect = task["contacts"]["contact_rectangle"]
    scale = float(rect.get("scale_factor", 1.0))
    bounds = dict(x_min=float(rect["x_min"]) * scale, x_max=float(rect["x_max"]) * scale,
                  y_min=float(rect["y_min"]) * scale, y_max=float(rect["y_max"]) * scale)

    def frame_on_joint(jn, offset):
        b, R, p = frames[jn]
        assert np.allclose(R, np.eye(3))
        return dict(body=b, p=(p + offset).tolist())

    contacts = [frame_on_joint(jn, t_c) for jn in contact_parents]
    coll_p1 = [frame_on_joint(jn, t_c + np.array([bounds["x_max"] * 0.6, 0, 0])) for jn in contact_parents]
    coll_p2 = [frame_on_joint(jn, t_c + np.array([bounds["x_min"] * 0.6, 0, 0])) for jn in contact_parents]
    cc = task["collision_constraint"]
    ankles = [frame_on_joint(cc["foot"]["leftAnkleFrame"], np.zeros(3)),
              frame_on_joint(cc["foot"]["rightAnkleFrame"], np.zeros(3))]
    knees = [frame_on_joint(cc["knee"]["leftKneeFrame"], np.zeros(3)),
             frame_on_joint(cc["knee"]["rightKneeFrame"], np.zeros(3))]

    nx, nu = (12 + nj if cent else 2 * (6 + nj)), 12 + nj
    Q = info_diag(task["Q"], nx)
    R = info_diag(task["R"], nu)
    Qf = info_diag(task["Q_final"], nx) * float(task["terminalCostScaling"])
    assert np.count_nonzero(Q - np.diag(np.diag(Q))) == 0 and np.count_nonzero(R - np.diag(np.diag(R))) == 0

    # EndEffectorDynamicsWeights::getWeights quirk (EndEffectorDynamicsCostHelpers.cpp:100-108):
    # lin/ang *velocity* weights are overwritten by the *acceleration* entries, acceleration
    # weights keep their struct defaults 0.01 (EndEffectorDynamicsCostHelpers.h:45-50).
    fw = task["task_space_foot_cost_weights"]
    g = lambda k: float(fw[k])
    foot_w = [g("pos_x"), g("pos_y"), g("pos_z"), g("orientation_x"), g("orientation_y"), g("orientation_z"),
              g("lin_acceleration_x"), g("lin_acceleration_y"), g("lin_acceleration_z"),
              g("ang_acceleration_x"), g("ang_acceleration_y"), g("ang_acceleration_z"),
              0.01, 0.01, 0.01, 0.01, 0.01, 0.01]
    extra = {}
    if cent:
        # centroidal problem (humanoid_nmpc/humanoid_centroidal_mpc/src/CentroidalMpcInterface.cpp:139-215):
        # EndEffectorKinematicsWeights::getWeights (humanoid_common_mpc/src/cost/EndEffectorKinematicCostHelpers.cpp:52-97) for the
        # feet and the torso link, ExternalTorqueQuadraticCostAD::loadConfigFromFile (…/ExternalTorqueQuadraticCostAD.cpp:140-170)
        kin_keys = ["pos_x", "pos_y", "pos_z", "orientation_x", "orientation_y", "orientation_z", "lin_velocity_x",
                    "lin_velocity_y", "lin_velocity_z", "ang_velocity_x", "ang_velocity_y", "ang_velocity_z"]
        torso = task["task_space_costs"]["torso"]
        tb, tR, tp = frames[torso["link_name"]]
        ext_w, ext_j = [], []
        for side in ("left_leg_torque_cost", "right_leg_torque_cost"):
            names = info_list(task[side]["activeJointNames"])
            w = info_vec(task[side]["weights"], len(names)) * float(task[side]["weights"].get("scaling", 1.0))
            ext_w.append(w.tolist())
            ext_j.append([joint_names.index(n) for n in names])
        extra = dict(cent_foot_cost_weights=[g(k) for k in kin_keys],
                     torso=dict(link=torso["link_name"], body=tb, R=tR.reshape(-1).tolist(), p=tp.tolist(),
                                weights=[float(torso["weights"][k]) for k in kin_keys]),
                     ext_torque=dict(weights=ext_w, joints=ext_j),
                     icp_weight=float(task["icp_cost_weights"]["icpErrorWeight"]),
                     centroidal_model_type=int(task["centroidalModelType"]))

    fc = ms["foot_constraint"]
    fr = task["contacts"]["frictionForceConeSoftConstraint"]
    mxy = task["contacts"]["contactMomentXYSoftConstraint"]
    jl = task["jointLimits"]
    st = task["swing_trajectory_config"]
    msq = task["multiple_shooting"]
    arm = ms["armJointNames"]

    gaits = {}
    for name in info_list(gait["list"]) + ["very_slow_trot"]:
        if name in gait:
            gaits[name] = dict(modeSequence=info_list(gait[name]["modeSequence"]),
                               switchingTimes=[float(x) for x in info_list(gait[name]["switchingTimes"])])

    out = dict(
        _generated_by="tools/export_g1_model.py from manumerous/wb_humanoid_mpc @ 2025-08-08 (URDF + task.info + reference.info + gait.info)",
        robot="g1", formulation=formulation, nj=nj, nx=nx, nu=nu, gravity=9.81, total_mass=total_mass,
        joint_names=joint_names,
        bodies=[dict(name=b["name"], joint=b["joint"], parent=b["parent"], R=b["R"].reshape(-1).tolist(),
                     p=b["p"].tolist(), axis=(b["axis"].tolist() if b["axis"] is not None else [0, 0, 0]),
                     mass=b["inertia"].m, com=b["inertia"].c.tolist(), inertia=b["inertia"].I.reshape(-1).tolist(),
                     lo=b["lo"], hi=b["hi"]) for b in bodies],
        frames=dict(contact=contacts, collision_p1=coll_p1, collision_p2=coll_p2, ankle=ankles, knee=knees),
        contact_rectangle=bounds,
        Q=np.diag(Q).tolist(), R=np.diag(R).tolist(), Qf=np.diag(Qf).tolist(),
        foot_cost_weights=foot_w,
        foot_constraint=dict((k, float(v)) for k, v in fc.items()),
        friction=dict(mu=float(fr["frictionCoefficient"]), regularization=25.0, gripper_force=0.0,
                      hessian_diagonal_shift=1e-6, barrier_mu=float(fr["mu"]), barrier_delta=float(fr["delta"])),
        moment_xy=dict(barrier_mu=float(mxy["mu"]), barrier_delta=float(mxy["delta"])),
        joint_limits=dict(barrier_mu=float(jl["mu"]), barrier_delta=float(jl["delta"])),
        collision=dict(r_foot=float(cc["foot"]["footCollisionSphereRadius"]),
                       r_knee=float(cc["knee"]["kneeCollisionSphereRadius"]),
                       barrier_mu=float(cc["mu"]), barrier_delta=float(cc["delta"])),
        arm_swing_joints=[joint_names.index(arm[k]) for k in
                          ("left_shoulder_y", "right_shoulder_y", "left_elbow_y", "right_elbow_y")],
        swing=dict((k, float(v)) for k, v in st.items()),
        sqp=dict(dt=float(msq["dt"]), sqpIteration=int(msq["sqpIteration"]), deltaTol=float(msq["deltaTol"]),
                 g_max=float(msq["g_max"]), g_min=float(msq["g_min"]), nThreads=int(msq["nThreads"])),
        mpc=dict(timeHorizon=float(task["mpc"]["timeHorizon"])),
        initial_state=info_vec(task["initialState"], nx).tolist(),
        default_joint_state=info_vec(ref["defaultJointState"], nj).tolist(),
        default_base_height=float(ref["defaultBaseHeight"]),
        phase_transition_stance_time=float(ms["phaseTransitionStanceTime"]),
        gaits=gaits,
        **extra,
    )
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    print(f"wrote {out_path}: {len(bodies)} bodies, total mass {total_mass:.4f} kg")


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    if len(sys.argv) > 1 and sys.argv[1] == "--centroidal":
        main(sys.argv[2] if len(sys.argv) > 2 else os.path.join(here, "..", "wb_humanoid_mpc_amd", "data", "g1_centroidal.json"), "centroidal")
    else:
        main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "wb_humanoid_mpc_amd", "data", "g1_wb.json"))
