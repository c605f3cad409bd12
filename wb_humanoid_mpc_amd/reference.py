"""Host-side per-node parameter generation.

Mirrors, for the solver's node grid, what the reference's solver pulls through its
callbacks every iteration:
  * mode schedule / contact flags  — GaitSchedule::tileModeSequenceTemplate
    (humanoid_nmpc/humanoid_common_mpc/src/gait/GaitSchedule.cpp:101-131),
    ModeSchedule::modeAtTime (upstream ocs2: lower_bound on the event times),
    modeNumber2StanceLeg (…/gait/MotionPhaseDefinition.h:58-76)
  * swing-foot height splines and impact-proximity factor — SwingTrajectoryPlanner::update
    (…/swing_foot_planner/SwingTrajectoryPlanner.cpp:87-191), SplineCpg.cpp:38-62, CubicSpline.cpp:38-80
  * gait phase variable for the arm-swing reference — SwitchedModelReferenceManager::getPhaseVariable
    (…/reference_manager/SwitchedModelReferenceManager.cpp:62-80)
  * target trajectories from a commanded base velocity — WBMpcTargetTrajectoriesCalculator::
    commandedVelocityToTargetTrajectories (humanoid_nmpc/humanoid_wb_mpc/src/command/WBMpcTargetTrajectoriesCalculator.cpp:80-136)
  * cold-start trajectory — WeightCompInitializer::compute (…/initialization/WeightCompInitializer.cpp:66-70)

The result is the POD table hsqp_problem::node_params (include/hsqp.h).  In the ROS
deployment the ocs2 adaptor fills the same table from the reference's own
ReferenceManager / SwingTrajectoryPlanner objects.
"""
import bisect
import math

import numpy as np

from . import _abi

FLY, RF, LF, STANCE = 0, 1, 2, 3
MODE_BY_NAME = {"FLY": FLY, "RF": RF, "LF": LF, "STANCE": STANCE}


def mode_to_contact_flags(mode):
    """{left, right} stance flags (MotionPhaseDefinition.h:58-76)."""
    return (mode in (LF, STANCE), mode in (RF, STANCE))


class ModeSchedule:
    def __init__(self, event_times, mode_sequence):
        assert len(mode_sequence) == len(event_times) + 1
        self.event_times = list(event_times)
        self.mode_sequence = list(mode_sequence)

    def index_at(self, t):
        # lookup::findIndexInTimeArray == std::lower_bound
        return bisect.bisect_left(self.event_times, t)

    def mode_at(self, t):
        return self.mode_sequence[self.index_at(t)]


def tile_gait(gait, t_start, t_final):
    """STANCE until t_start, then the template tiled until the tiling passes t_final, then STANCE
    (GaitSchedule::tileModeSequenceTemplate: 'add a initial time', tile, 'default final phase')."""
    modes = [MODE_BY_NAME[m] for m in gait["modeSequence"]]
    times = gait["switchingTimes"]
    event_times = [t_start]
    seq = [STANCE]
    while event_times[-1] < t_final:
        for i, m in enumerate(modes):
            seq.append(m)
            event_times.append(event_times[-1] + (times[i + 1] - times[i]))
    seq.append(STANCE)
    return ModeSchedule(event_times, seq)


# ------------------------------------------------------------------------------------ splines
class CubicSpline:
    def __init__(self, t0, p0, v0, t1, p1, v1):
        self.t0, self.dt = t0, t1 - t0
        dp, dv = p1 - p0, v1 - v0
        dc1, dc2, dc3 = v0, -(3.0 * v0 + dv), (2.0 * v0 + dv)
        self.c0 = p0
        self.c1 = dc1 * self.dt
        self.c2 = dc2 * self.dt + 3.0 * dp
        self.c3 = dc3 * self.dt - 2.0 * dp

    def _tn(self, t):
        return (t - self.t0) / self.dt

    def position(self, t):
        tn = self._tn(t)
        return self.c3 * tn ** 3 + self.c2 * tn ** 2 + self.c1 * tn + self.c0

    def velocity(self, t):
        tn = self._tn(t)
        return (3.0 * self.c3 * tn * tn + 2.0 * self.c2 * tn + self.c1) / self.dt

    def acceleration(self, t):
        tn = self._tn(t)
        return (6.0 * self.c3 * tn + 2.0 * self.c2) / (self.dt * self.dt)


class SplineCpg:
    def __init__(self, lift_off, mid_height, touch_down):
        (t0, p0, v0), (t1, p1, v1) = lift_off, touch_down
        self.mid = 0.5 * (t0 + t1)
        self.left = CubicSpline(t0, p0, v0, self.mid, mid_height, 0.0)
        self.right = CubicSpline(self.mid, mid_height, 0.0, t1, p1, v1)

    def _s(self, t):
        return self.left if t < self.mid else self.right

    def position(self, t):
        return self._s(t).position(t)

    def velocity(self, t):
        return self._s(t).velocity(t)

    def acceleration(self, t):
        return self._s(t).acceleration(t)


class SwingTrajectoryPlanner:
    """Per-leg z-height and impact-proximity splines over a mode schedule, flat terrain."""

    def __init__(self, cfg, schedule, terrain_height=0.0):
        self.schedule = schedule
        seq, ev = schedule.mode_sequence, schedule.event_times
        n = len(seq)
        self.height = [[], []]
        self.impact = [[], []]
        for leg in range(2):
            flags = [mode_to_contact_flags(m)[leg] for m in seq]
            lift = terrain_height
            touch = terrain_height + cfg["touchDownHeightOffset"]
            for p in range(n):
                if flags[p]:
                    self.height[leg].append(SplineCpg((0.0, lift, 0.0), lift, (1.0, lift, 0.0)))
                    self.impact[leg].append(SplineCpg((0.0, 1.0, 0.0), 1.0, (1.0, 1.0, 0.0)))
                    continue
                start = next((ip for ip in range(p - 1, -1, -1) if flags[ip]), -1)
                final = next((ip - 1 for ip in range(p + 1, n) if flags[ip]), n - 1)
                if start < 0 or final >= n - 1:
                    raise RuntimeError(f"swing phase {p} of leg {leg} has no lift-off/touch-down inside the schedule")
                t0, t1 = ev[start], ev[final]
                scaling = min(1.0, (t1 - t0) / cfg["swingTimeScale"])
                mp = cfg["impactProximityFactorMidPointValue"]
                if flags[p - 1] and flags[p + 1]:
                    mid = min(lift, touch) + scaling * cfg["swingHeight"]
                    self.height[leg].append(SplineCpg((t0, lift, scaling * cfg["liftOffVelocity"]), mid,
                                                      (t1, touch, scaling * cfg["touchDownVelocity"])))
                    self.impact[leg].append(SplineCpg((t0, 1.0, scaling * cfg["impactProximityFactorLiftOffVelocity"]), mp,
                                                      (t1, 1.0, scaling * cfg["impactProximityFactorTouchDownVelocity"])))
                elif flags[p - 1]:
                    mid = lift + cfg["swingHeight"]
                    self.height[leg].append(SplineCpg((t0, lift, cfg["liftOffVelocity"]), mid, (t1, mid, 0.0)))
                    self.impact[leg].append(SplineCpg((t0, 1.0, cfg["impactProximityFactorLiftOffVelocity"]), mp, (t1, mp, 0.0)))
                elif flags[p + 1]:
                    mid = touch + cfg["swingHeight"]
                    self.height[leg].append(SplineCpg((t0, mid, 0.0), mid, (t1, touch, cfg["touchDownVelocity"])))
                    self.impact[leg].append(SplineCpg((t0, mp, 0.0), mp, (t1, 1.0, cfg["impactProximityFactorTouchDownVelocity"])))
                else:
                    mid = touch + cfg["swingHeight"]
                    self.height[leg].append(SplineCpg((t0, mid, 0.0), mid, (t1, mid, 0.0)))
                    self.impact[leg].append(SplineCpg((t0, mp, 0.0), mp, (t1, mp, 0.0)))

    def z_refs(self, leg, t):
        s = self.height[leg][self.schedule.index_at(t)]
        return s.position(t), s.velocity(t), s.acceleration(t)

    def impact_proximity(self, leg, t):
        return self.impact[leg][self.schedule.index_at(t)].position(t)


def phase_variable(schedule, t):
    """SwitchedModelReferenceManager::getPhaseVariable (upper_bound on the event times)."""
    ev = schedule.event_times
    it = bisect.bisect_right(ev, t)
    if it <= 0 or it >= len(ev):
        return 0.0
    nxt, prv = ev[it], ev[it - 1]
    mode = schedule.mode_at(t)
    if mode == LF:
        return 0.5 * (t - prv) / (nxt - prv)
    if mode == RF:
        return 0.5 + 0.5 * (t - prv) / (nxt - prv)
    return 0.5 if schedule.mode_at(prv - 0.01) == LF else 0.0


# ------------------------------------------------------------------------------------ targets
class TargetTrajectories:
    def __init__(self, times, states):
        self.times = np.asarray(times, dtype=float)
        self.states = np.asarray(states, dtype=float)

    def desired_state(self, t):
        """ocs2 LinearInterpolation: clamped piece-wise linear."""
        if t <= self.times[0]:
            return self.states[0].copy()
        if t >= self.times[-1]:
            return self.states[-1].copy()
        i = int(np.searchsorted(self.times, t, side="right")) - 1
        a = (self.times[i + 1] - t) / (self.times[i + 1] - self.times[i])
        return a * self.states[i] + (1.0 - a) * self.states[i + 1]


def velocity_command_targets(model, v_cmd, t0, x0, horizon):
    """commandedVelocityToTargetTrajectories with the command filter at steady state
    (the reference's first-call transient of the function-local static filter —
    TargetTrajectoriesCalculatorBase.cpp:117-119 — is deliberately not reproduced)."""
    nj = model.nj
    vx, vy, height, wz = v_cmd
    pose = np.array(x0[:6], dtype=float)
    pose[4] = pose[5] = 0.0
    yaw = pose[3]
    gvx = math.cos(yaw) * vx - math.sin(yaw) * vy
    gvy = math.sin(yaw) * vx + math.cos(yaw) * vy
    target_vel = np.array([gvx, gvy, 0.0, wz, 0.0, 0.0])
    base_vel = np.asarray(x0[6 + nj: 12 + nj])
    t_mid = 0.7 * horizon
    pose[2] = height

    def integrate(p, v3, h, dt):
        q = p.copy()
        q[0] += v3[0] * dt
        q[1] += v3[1] * dt
        q[2] = h
        q[3] += v3[2] * dt
        q[4] = q[5] = 0.0
        return q

    mid = integrate(pose, [(base_vel[0] + gvx) / 2, (base_vel[1] + gvy) / 2, (base_vel[5] + wz) / 2], height, t_mid)
    fin = integrate(mid, [gvx, gvy, wz], height, horizon - t_mid)
    jt = model.default_joint_state
    mk = lambda p: np.concatenate([p, jt, target_vel, np.zeros(nj)])
    return TargetTrajectories([t0, t0 + t_mid, t0 + horizon], [mk(pose), mk(mid), mk(fin)])


# ------------------------------------------------------------------------------------ node table
def weight_compensating_input(model, flags):
    u = np.zeros(model.nu)
    ns = int(flags[0]) + int(flags[1])
    if ns:
        fz = model.total_mass * 9.81 / ns
        if flags[0]:
            u[2] = fz
        if flags[1]:
            u[8] = fz
    return u


def build_node_params(model, schedule, targets, t0, dt, n_nodes, arm_swing=True):
    """[N+1][HSQP_NODE_PARAMS] table for one instance on the uniform grid t_k = t0 + k dt."""
    planner = SwingTrajectoryPlanner(model.swing, schedule)
    par = np.zeros((n_nodes + 1, _abi.NODE_PARAMS))
    for k in range(n_nodes + 1):
        t = t0 + k * dt
        flags = mode_to_contact_flags(schedule.mode_at(t))
        par[k, _abi.P_XDES:_abi.P_XDES + model.nx] = targets.desired_state(t)
        par[k, _abi.P_ARMSWING] = math.sin(2.0 * math.pi * (phase_variable(schedule, t) - 0.15)) if arm_swing else 0.0
        par[k, _abi.P_CONTACT:_abi.P_CONTACT + 2] = flags
        for leg in range(2):
            par[k, _abi.P_SWING + 3 * leg:_abi.P_SWING + 3 * leg + 3] = planner.z_refs(leg, t)
            par[k, _abi.P_IMPACT + leg] = planner.impact_proximity(leg, t)
    return par


def swing_config(model):
    """hsqp_swing_config from task.info's swing_trajectory_config."""
    c = model.swing
    return _abi.SwingConfig(lift_off_velocity=c["liftOffVelocity"], touch_down_velocity=c["touchDownVelocity"], swing_height=c["swingHeight"],
                            touch_down_height_offset=c["touchDownHeightOffset"], swing_time_scale=c["swingTimeScale"],
                            impact_mid=c["impactProximityFactorMidPointValue"], impact_lift_velocity=c["impactProximityFactorLiftOffVelocity"],
                            impact_touch_velocity=c["impactProximityFactorTouchDownVelocity"])


def pack_reference(schedules, targets):
    """Compact per-instance reference for hsqp_upload_reference: (n_events[B], event_times[B,E], mode_sequence[B,E+1],
    target_times[B,K], target_states[B,K,58]) from ModeSchedule / TargetTrajectories objects."""
    B = len(schedules)
    E = max(len(s.event_times) for s in schedules)
    K = len(targets[0].times)
    n_events = np.array([len(s.event_times) for s in schedules], dtype=np.int32)
    ev = np.zeros((B, E))
    seq = np.full((B, E + 1), STANCE, dtype=np.int32)
    for b, s in enumerate(schedules):
        ev[b, :n_events[b]] = s.event_times
        ev[b, n_events[b]:] = s.event_times[-1]
        seq[b, :n_events[b] + 1] = s.mode_sequence
    tt = np.stack([t.times for t in targets])
    ts = np.stack([t.states for t in targets])
    assert tt.shape == (B, K) and ts.shape[:2] == (B, K)
    return n_events, ev, seq, tt, ts


def cold_start(model, x0, par):
    """WeightCompInitializer: x_k = x0, u_k = weight compensation for the node's contact flags."""
    n_nodes = par.shape[0] - 1
    x = np.tile(np.asarray(x0, dtype=float), (n_nodes + 1, 1))
    u = np.stack([weight_compensating_input(model, par[k, _abi.P_CONTACT:_abi.P_CONTACT + 2] > 0.5) for k in range(n_nodes)])
    return x, u


# ------------------------------------------------------------------------------------ time grid with event nodes
EVENT_EPS = 1e-9   # a post-event node is sampled at t_event + EVENT_EPS (ocs2 getIntervalStart adds an epsilon of its own)


def time_discretization_with_events(t0, tf, dt, event_times, dt_min=1e-4):
    """ocs2 multiple-shooting time grid (upstream ocs2_oc timeDiscretizationWithEvents, restated from the published source; the fork's
    copy is absent): nodes every dt from t0, an event time inside (t0, tf) becomes a PRE-event node and a POST-event node with the same
    time stamp, the node before it is dropped if it would be closer than dt_min, the last interval is shortened to hit tf.
    Returns (times[n+1], is_post_event[n+1])."""
    times, post = [float(t0)], [False]
    ev = [e for e in event_times if t0 < e < tf]
    ie = 0
    while times[-1] < tf:
        t_next, is_event = times[-1] + dt, False
        if ie < len(ev) and t_next >= ev[ie]:
            t_next, is_event = ev[ie], True
            ie += 1
        if t_next >= tf:
            t_next, is_event = float(tf), False
        if t_next > times[-1] + dt_min or post[-1]:
            times.append(t_next); post.append(False)
        else:
            times[-1] = t_next
        if is_event:
            times.append(t_next); post.append(True)
    return np.array(times), np.array(post)


def event_grid(t0, tf, dt, event_times):
    """(dt_nodes[N], node_times[N+1]) for hsqp_problem::dt_nodes / hsqp_reference::node_times: interval lengths with 0 on the
    pre -> post event intervals, and the sampling time of every node (post-event nodes at t + EVENT_EPS, so that look-ups by
    lower_bound see the new mode)."""
    times, post = time_discretization_with_events(t0, tf, dt, event_times)
    return np.diff(times), times + EVENT_EPS * post


def build_node_params_at(model, schedule, targets, node_times, arm_swing=True):
    """build_node_params on explicit node times (non-uniform grid / event nodes)."""
    planner = SwingTrajectoryPlanner(model.swing, schedule)
    par = np.zeros((len(node_times), _abi.NODE_PARAMS))
    for k, t in enumerate(node_times):
        flags = mode_to_contact_flags(schedule.mode_at(t))
        par[k, _abi.P_XDES:_abi.P_XDES + model.nx] = targets.desired_state(t)
        par[k, _abi.P_ARMSWING] = math.sin(2.0 * math.pi * (phase_variable(schedule, t) - 0.15)) if arm_swing else 0.0
        par[k, _abi.P_CONTACT:_abi.P_CONTACT + 2] = flags
        for leg in range(2):
            par[k, _abi.P_SWING + 3 * leg:_abi.P_SWING + 3 * leg + 3] = planner.z_refs(leg, t)
            par[k, _abi.P_IMPACT + leg] = planner.impact_proximity(leg, t)
    return par


# ------------------------------------------------------------------------------------ benchmark configs
BENCH_SEED = 20250808


def make_problem(model, n_nodes=100, batch=1, gait="walk", v_cmd=(0.3, 0.0, 0.7925, 0.0), dt=None, perturb=False,
                 seed=BENCH_SEED, t0=0.0, with_reference=False):
    """Synthetic inputs of BASELINE.md configs 3-5: (x_init[B,58], x[B,N+1,58], u[B,N,35], params[B,N+1,72], dt).

    perturb=False: every instance starts at task.info's initialState with gait phase offset 0 (config 3).
    with_reference=True additionally returns (schedules, targets, t0) for the device-side parameter generation.
    perturb=True : x0 + N(0, sigma^2) (0.02 m base position, 0.05 rad euler/joints, 0.1 velocities; joints clipped to
                   limits - 0.05 rad) and a per-instance gait phase offset U[0, 1.4 s) from numpy PCG64(seed) (configs 4-5).
    """
    dt = model.sqp["dt"] if dt is None else dt
    horizon = n_nodes * dt
    rng = np.random.Generator(np.random.PCG64(seed))
    nj = model.nj
    xs, us, ps, x0s = [], [], [], []
    schedules, targets_all = [], []
    for _ in range(batch):
        x0 = model.initial_state.copy()
        offset = 0.0
        if perturb:
            sig = np.concatenate([np.full(3, 0.02), np.full(3 + nj, 0.05), np.full(6 + nj, 0.1)])
            x0 = x0 + sig * rng.standard_normal(model.nx)
            x0[6:6 + nj] = np.clip(x0[6:6 + nj], model.q_lo + 0.05, model.q_hi - 0.05)
            offset = rng.uniform(0.0, 1.4)
        # the gait starts `offset` seconds before t0, preceded by stance; tile far enough past the horizon that
        # every swing phase has a touch-down inside the schedule
        schedule = tile_gait(model.gaits[gait], t0 - offset - 1e-9, t0 + 2.0 * horizon + 3.0)
        targets = velocity_command_targets(model, v_cmd, t0, x0, horizon)
        par = build_node_params(model, schedule, targets, t0, dt, n_nodes)
        x, u = cold_start(model, x0, par)
        xs.append(x); us.append(u); ps.append(par); x0s.append(x0)
        schedules.append(schedule); targets_all.append(targets)
    if with_reference:
        return np.stack(x0s), np.stack(xs), np.stack(us), np.stack(ps), dt, (schedules, targets_all, t0)
    return np.stack(x0s), np.stack(xs), np.stack(us), np.stack(ps), dt


# ------------------------------------------------------------------------------------ centroidal formulation
# Host-side parameter generation of the centroidal problem.  States are handled unpadded (35) in this section and padded to the
# ABI's 58-double rows by make_centroidal_problem().
def _rot_axis(axis, q):
    a = np.asarray(axis, dtype=float)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + math.sin(q) * K + (1.0 - math.cos(q)) * (K @ K)


def body_placements(model, q):
    """World placements (R[NB], p[NB]) of the MPC model's bodies at q = [p_b, eulerZYX, q_j] (composite Translation +
    SphericalZYX base: createPinocchioModel.cpp:60-67)."""
    bodies = model.raw["bodies"]
    R, p = [None] * len(bodies), [None] * len(bodies)
    R[0] = _rot_axis([0, 0, 1], q[3]) @ _rot_axis([0, 1, 0], q[4]) @ _rot_axis([1, 0, 0], q[5])
    p[0] = np.array(q[:3], dtype=float)
    for i in range(1, len(bodies)):
        b = bodies[i]
        par = b["parent"]
        R[i] = R[par] @ np.array(b["R"]).reshape(3, 3) @ _rot_axis(b["axis"], q[5 + i])
        p[i] = p[par] + R[par] @ np.array(b["p"])
    return R, p


def euler_rate_axes(q):
    """World axes of the euler Z, Y, X rates: omega = E @ [zdot, ydot, xdot]."""
    Rz = _rot_axis([0, 0, 1], q[3])
    Rzy = Rz @ _rot_axis([0, 1, 0], q[4])
    return np.stack([np.array([0.0, 0.0, 1.0]), Rz[:, 1], Rzy[:, 0]], axis=1)


def centroidal_base_velocity(model, q, h_normalized, qd_j=None):
    """v_b = A_b^-1 (m h - A_j qd_j): CentroidalModelPinocchioMapping::getPinocchioJointVelocity (upstream ocs2_centroidal_model,
    FullCentroidalDynamics) evaluated from the composite inertia of the whole robot about its centre of mass."""
    bodies = model.raw["bodies"]
    R, p = body_placements(model, q)
    m = model.total_mass
    cs = [p[i] + R[i] @ np.array(b["com"]) for i, b in enumerate(bodies)]
    com = sum(b["mass"] * c for b, c in zip(bodies, cs)) / m
    Ic = np.zeros((3, 3))
    for i, b in enumerate(bodies):
        d = cs[i] - com
        Ic += R[i] @ np.array(b["inertia"]).reshape(3, 3) @ R[i].T + b["mass"] * (d @ d * np.eye(3) - np.outer(d, d))
    lin_j, ang_j = np.zeros(3), np.zeros(3)
    if qd_j is not None:   # momentum of the joint motion about the centre of mass
        om, v = [np.zeros(3)] * len(bodies), [np.zeros(3)] * len(bodies)
        for i in range(1, len(bodies)):
            b = bodies[i]
            par = b["parent"]
            w = R[par] @ np.array(b["R"]).reshape(3, 3) @ np.array(b["axis"])
            om[i] = om[par] + w * qd_j[i - 1]
            v[i] = v[par] + np.cross(om[par], p[i] - p[par])
            vc = v[i] + np.cross(om[i], cs[i] - p[i])
            lin_j = lin_j + b["mass"] * vc
            ang_j = ang_j + R[i] @ np.array(b["inertia"]).reshape(3, 3) @ R[i].T @ om[i] + np.cross(cs[i] - com, b["mass"] * vc)
    E = euler_rate_axes(q)
    rhs_lin, rhs_ang = m * np.asarray(h_normalized[:3]) - lin_j, m * np.asarray(h_normalized[3:6]) - ang_j
    euler_rates = np.linalg.solve(Ic @ E, rhs_ang)
    omega = E @ euler_rates
    vlin = rhs_lin / m - np.cross(omega, com - p[0])
    return np.concatenate([vlin, euler_rates])


def matrix_to_quaternion(R):
    """x, y, z, w (Eigen coefficient order), w >= 0 branch first (oracle ASSUMPTION A8)."""
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0.0:
        s = math.sqrt(tr + 1.0) * 2.0
        return np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = math.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2.0
    qv = np.zeros(4)
    qv[i] = 0.25 * s
    qv[j] = (R[i, j] + R[j, i]) / s
    qv[k] = (R[i, k] + R[k, i]) / s
    qv[3] = (R[k, j] - R[j, k]) / s
    return qv


def torso_reference(model, x_ref):
    """EndEffectorKinematicsQuadraticCost::getReferenceCostElement (EndEffectorKinematicsQuadraticCost.cpp:92-104): the torso
    link's position, orientation quaternion and LOCAL_WORLD_ALIGNED velocities at (xRef, uRef = 0)."""
    t = model.raw["torso"]
    q = np.asarray(x_ref[6:35], dtype=float)
    R, p = body_placements(model, q)
    b = t["body"]
    vb = centroidal_base_velocity(model, q, x_ref[:6])
    omega = euler_rate_axes(q) @ vb[3:6]
    r = R[b] @ np.array(t["p"])
    pos = p[b] + r
    vlin = vb[:3] + np.cross(omega, pos - p[0])
    return np.concatenate([pos, matrix_to_quaternion(R[b] @ np.array(t["R"]).reshape(3, 3)), vlin, omega])


def centroidal_velocity_command_targets(model, v_cmd, t0, x0, horizon):
    """CentroidalMpcTargetTrajectoriesCalculator::commandedVelocityToTargetTrajectories
    (humanoid_centroidal_mpc/src/command/CentroidalMpcTargetTrajectoriesCalculator.cpp:88-158), command filter at steady state."""
    vx, vy, height, wz = v_cmd
    pose = np.array(x0[6:12], dtype=float)
    pose[4] = pose[5] = 0.0
    yaw = pose[3]
    gvx = math.cos(yaw) * vx - math.sin(yaw) * vy
    gvy = math.sin(yaw) * vx + math.cos(yaw) * vy
    target_momentum = np.array([gvx, gvy, 0.0, 0.0, 0.0, wz / model.total_mass])
    base_vel = centroidal_base_velocity(model, np.asarray(x0[6:35]), x0[:6])   # "assumes no joint velocities"
    # the reference reads baseVel[5] (the euler X rate) as the yaw rate here; reproduced
    t_mid = 0.7 * horizon
    pose[2] = height

    def integrate(pp, v3, h, dt):
        qq = pp.copy()
        qq[0] += v3[0] * dt
        qq[1] += v3[1] * dt
        qq[2] = h
        qq[3] += v3[2] * dt
        qq[4] = qq[5] = 0.0
        return qq

    mid = integrate(pose, [(base_vel[0] + gvx) / 2, (base_vel[1] + gvy) / 2, (base_vel[5] + wz) / 2], height, t_mid)
    fin = integrate(mid, [gvx, gvy, wz], height, horizon - t_mid)
    jt = model.default_joint_state
    mk = lambda pp: np.concatenate([target_momentum, pp, jt])
    return TargetTrajectories([t0, t0 + t_mid, t0 + horizon], [mk(pose), mk(mid), mk(fin)])


def build_centroidal_node_params(model, schedule, targets, t0, dt, n_nodes, arm_swing=True):
    """[N+1][HSQP_NODE_PARAMS] table of the centroidal problem (layout: include/hsqp.h)."""
    planner = SwingTrajectoryPlanner(model.swing, schedule)
    par = np.zeros((n_nodes + 1, _abi.NODE_PARAMS))
    for k in range(n_nodes + 1):
        t = t0 + k * dt
        flags = mode_to_contact_flags(schedule.mode_at(t))
        xd = targets.desired_state(t)
        par[k, _abi.P_XDES:_abi.P_XDES + _abi.CNX] = xd
        par[k, _abi.PC_TORSO:_abi.PC_TORSO + 13] = torso_reference(model, xd)
        par[k, _abi.P_ARMSWING] = math.sin(2.0 * math.pi * (phase_variable(schedule, t) - 0.15)) if arm_swing else 0.0
        par[k, _abi.P_CONTACT:_abi.P_CONTACT + 2] = flags
        for leg in range(2):
            z, zd, _ = planner.z_refs(leg, t)
            par[k, _abi.P_SWING + 3 * leg:_abi.P_SWING + 3 * leg + 3] = (z, zd, 0.0)
            par[k, _abi.P_IMPACT + leg] = planner.impact_proximity(leg, t)
    return par


def pad_targets(targets):
    """TargetTrajectories with the 35-wide centroidal knots padded to the ABI's 58-double rows (for pack_reference)."""
    st = np.zeros((len(targets.times), _abi.NX))
    st[:, :_abi.CNX] = targets.states
    return TargetTrajectories(targets.times, st)


def make_centroidal_problem(model, n_nodes=100, batch=1, gait="walk", v_cmd=(0.3, 0.0, 0.7925, 0.0), dt=None, perturb=False,
                            seed=BENCH_SEED, t0=0.0, with_reference=False):
    """Synthetic inputs of BASELINE.md configs 1-2 in the ABI's padded layout: (x_init[B,58], x[B,N+1,58], u[B,N,35],
    params[B,N+1,72], dt); only the first 35 entries of a state row are used.  Cold start as CentroidalWeightCompInitializer
    (x_k = x0 with the momentum extended, u_k = weight compensation)."""
    assert model.centroidal
    dt = model.sqp["dt"] if dt is None else dt
    horizon = n_nodes * dt
    rng = np.random.Generator(np.random.PCG64(seed))
    nj = model.nj
    xs, us, ps, x0s = [], [], [], []
    schedules, targets_all = [], []
    for _ in range(batch):
        x0 = model.initial_state.copy()
        offset = 0.0
        if perturb:
            sig = np.concatenate([np.full(6, 0.1), np.full(3, 0.02), np.full(3 + nj, 0.05)])
            x0 = x0 + sig * rng.standard_normal(model.nx)
            x0[12:12 + nj] = np.clip(x0[12:12 + nj], model.q_lo + 0.05, model.q_hi - 0.05)
            offset = rng.uniform(0.0, 1.4)
        schedule = tile_gait(model.gaits[gait], t0 - offset - 1e-9, t0 + 2.0 * horizon + 3.0)
        targets = centroidal_velocity_command_targets(model, v_cmd, t0, x0, horizon)
        par = build_centroidal_node_params(model, schedule, targets, t0, dt, n_nodes)
        x0p = np.zeros(_abi.NX)
        x0p[:_abi.CNX] = x0
        x, u = cold_start(model, x0p, par)
        xs.append(x); us.append(u); ps.append(par); x0s.append(x0p)
        schedules.append(schedule); targets_all.append(pad_targets(targets))
    if with_reference:
        return np.stack(x0s), np.stack(xs), np.stack(us), np.stack(ps), dt, (schedules, targets_all, t0)
    return np.stack(x0s), np.stack(xs), np.stack(us), np.stack(ps), dt
