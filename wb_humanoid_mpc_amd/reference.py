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
