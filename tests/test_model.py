"""Known-answer checks on the exported G1 model: layout, mass, frames."""
import numpy as np

from wb_humanoid_mpc_amd import _abi


def test_dimensions_match_reference_layout(model):
    # WBAccelMpcRobotModel: state = 2*(6+nj), input = 6*N_CONTACTS + nj (WBAccelMpcRobotModel.h:77)
    assert model.nj == 23 and model.nx == 2 * (6 + 23) == _abi.NX and model.nu == 12 + 23 == _abi.NU
    # joint order of task.info:130-152
    assert model.joint_names[0] == "left_hip_pitch_joint" and model.joint_names[6] == "right_hip_pitch_joint"
    assert model.joint_names[12] == "waist_yaw_joint" and model.joint_names[-1] == "right_elbow_joint"
    assert [model.joint_names[i] for i in model.raw["arm_swing_joints"]] == [
        "left_shoulder_pitch_joint", "right_shoulder_pitch_joint", "left_elbow_joint", "right_elbow_joint"]


def test_total_mass(model, oracle):
    # sum of URDF inertials = 35.115 kg, no mass scaling
    assert abs(model.total_mass - 35.115) < 1e-3
    assert abs(oracle.total_mass() - model.total_mass) < 1e-12


def test_tree_is_topologically_sorted(model):
    for i, b in enumerate(model.raw["bodies"]):
        assert b["parent"] < i
        if i:
            assert abs(np.linalg.norm(b["axis"]) - 1.0) < 1e-12
            R = np.array(b["R"]).reshape(3, 3)
            assert np.allclose(R @ R.T, np.eye(3), atol=1e-12)
            I = np.array(b["inertia"]).reshape(3, 3)
            assert np.allclose(I, I.T) and np.all(np.linalg.eigvalsh(I) > 0)


def test_foot_cost_weight_quirk(model):
    # EndEffectorDynamicsCostHelpers.cpp:105-108: velocity weights take the acceleration entries, acc keep 0.01
    w = model.raw["foot_cost_weights"]
    assert w[:3] == [0, 0, 0] and w[3:6] == [1e4, 1e4, 0.0]
    assert w[6:9] == [5.0, 5.0, 0.0] and w[9:12] == [2.0, 2.0, 2.0] and w[12:] == [0.01] * 6


def test_contact_frames(model, oracle):
    # contact frames: ankle_roll joint frame translated by (0.035, 0, -0.035), identity rotation
    for side in range(2):
        fr = model.raw["frames"]["contact"][side]
        assert np.allclose(fr["p"], [0.035, 0.0, -0.035])
        assert model.raw["bodies"][fr["body"]]["joint"].endswith("ankle_roll_joint")
    # frame-rotation identity at q = 0 (testPinocchioFrameConversions.cpp:61-90): R_contact = I
    x = np.zeros(model.nx)
    x[2] = 0.8415
    out, R = oracle.foot_kinematics(x, np.zeros(model.nu))
    assert np.allclose(R[0], np.eye(3), atol=1e-12) and np.allclose(R[1], np.eye(3), atol=1e-12)
    # left/right symmetric, below the pelvis
    assert np.allclose(out[0, :3] * [1, -1, 1], out[1, :3], atol=1e-9)
    assert 0.0 < out[0, 2] < 0.1 and abs(out[0, 0] - 0.035) < 1e-3


def test_reference_known_answers_for_the_contact_frames(model, oracle):
    """The two known answers the reference's tests hold for the contact frames (round 5 review, "missing 6").
    (1) testPinocchioFrameConversions.cpp:61-90: with q[6] = 0.5 — generalized coordinate 6 is the first joint, left_hip_pitch, a rotation about y —
        the left contact frame's local -> world rotation is R_y(0.5) (= getRotationMatrixFromZyxEulerAngles(0, 0.5, 0)); the right one stays the identity;
        local -> world -> local is the identity for random configurations (:92-110).
    (2) humanoid_centroidal_mpc/test/testDynamicsHelperFunctions.cpp:67-93 — DISABLED by its author ("change with model updates"): contact positions
        (0.04286, +-0.0895, 0) at q = 0, z = 0.8415.  They belong to an earlier model: the G1 URDF of this repository's reference snapshot puts the
        contact frames at (0.0350, +-0.1185, 0.0496) there (hip offset 0.1185 m; straight-leg pelvis height 0.7919 m = 0.8415 - 0.0496, the number
        DESIGN.md §2b derives independently).  Asserted: what carries over (x = the contact frame's 0.035 m offset within 8e-3, left / right mirror
        images, a pure translation of the base moves them rigidly: the second half of the disabled test), and that the stale y, z do NOT hold."""
    x = np.zeros(model.nx)
    x[2] = 0.8415
    u = np.zeros(model.nu)
    out, R = oracle.foot_kinematics(x, u)
    assert np.allclose(R[0], np.eye(3), atol=1e-12) and np.allclose(R[1], np.eye(3), atol=1e-12)
    assert abs(out[0, 0] - 0.04286) < 8e-3 and np.allclose(out[0, :3] * [1, -1, 1], out[1, :3], atol=1e-12)
    assert abs(out[0, 1] - 0.0895) > 2e-2 and abs(out[0, 2]) > 4e-2            # the disabled goldens are another model's
    assert abs((0.8415 - out[0, 2]) - 0.7919) < 1e-3
    x2 = x.copy(); x2[0] = -1.0
    out2, _ = oracle.foot_kinematics(x2, u)
    assert np.allclose(out2[:, :3] - out[:, :3], [[-1.0, 0.0, 0.0]] * 2, atol=1e-12)
    xk = x.copy(); xk[6] = 0.5
    _, Rk = oracle.foot_kinematics(xk, u)
    c, s = np.cos(0.5), np.sin(0.5)
    assert np.allclose(Rk[0], [[c, 0, s], [0, 1, 0], [-s, 0, c]], atol=1e-12) and np.allclose(Rk[1], np.eye(3), atol=1e-12)
    rng = np.random.default_rng(3)
    for _ in range(25):
        xr = np.zeros(model.nx); xr[:29] = rng.uniform(-1, 1, 29); xr[2] = 0.88
        _, Rr = oracle.foot_kinematics(xr, u)
        v = rng.uniform(-1, 1, 3)
        for side in range(2):
            assert np.allclose(Rr[side].T @ (Rr[side] @ v), v, atol=1e-13) and abs(np.linalg.det(Rr[side]) - 1.0) < 1e-12
