// TEST INFRASTRUCTURE — part of the CPU oracle (included at the end of oracle/oracle.cpp; see its header).
//
// Groundwork for SURVEY.md §8 row a22 (the centroidal formulation of the same MPC): the flow map and the equality constraints of
//   x = [h/m (6: v_com, L/m), q_b (6: p, eulerZYX), q_j (23)],  u = [W_l (6), W_r (6), qd_j (23)]
// (layout: humanoid_nmpc/humanoid_centroidal_mpc/include/humanoid_centroidal_mpc/common/CentroidalMpcRobotModel.h:49-71,89-95).
// There is NO product (HIP) path for this formulation yet; nothing outside tests/ uses these functions.
//
// PARITY STATUS: unpinned, like the rest of the oracle.  The reference's CentroidalDynamicsAD
// (humanoid_nmpc/humanoid_centroidal_mpc/src/dynamics/CentroidalDynamicsAD.cpp:52-63) only forwards to
// ocs2_centroidal_model::PinocchioCentroidalDynamicsAD, which lives in the absent submodule lib/ocs2_ros2.
// ASSUMPTION A7 (restated from the published leggedrobotics/ocs2 ocs2_centroidal_model sources; the fork's copy is absent):
//   * state derivative = [normalized centroidal momentum rate; generalized velocity of the Pinocchio model]
//     (PinocchioCentroidalDynamicsAD::getValueCppAd);
//   * normalized momentum rate = ( m g + sum_c f_c ;  sum_c (p_c - com) x f_c + tau_c ) / m  with g = (0, 0, -9.81)
//     hard-coded and p_c the contact FRAME positions (ModelHelperFunctions: getNormalizedCentroidalMomentumRate);
//   * FullCentroidalDynamics (task.info:1 `centroidalModelType 0`): v_b = A_b^-1 (m h - A_j qd_j) with A = [A_b | A_j] the
//     centroidal momentum matrix of the model whose base joint is Translation + SphericalZYX, i.e. v_b = [pdot, euler rates]
//     directly (CentroidalModelPinocchioMapping::getPinocchioJointVelocity); A_b^-1 by the block formula
//     [[I/m, -A_12 A_22^-1 / m], [0, A_22^-1]] (computeFloatingBaseCentroidalMomentumMatrixInverse), which is the exact
//     inverse because a base translation carries no angular momentum about the centre of mass.
// The golden numbers in humanoid_nmpc/humanoid_centroidal_mpc/test/testCentroidalConversions.cpp:40-131 belong to a
// 40-state robot whose URDF is not in the reference (the G1 model has 35 states): not usable.  The known answers that
// ARE usable are checked in tests/test_oracle_centroidal.py: the index layout and "weight compensation gives a zero
// normalized momentum rate" (humanoid_nmpc/humanoid_centroidal_mpc/test/testDynamicsHelperFunctions.cpp:95-127).
#pragma once
#include <memory>

namespace {

constexpr int CNX = 12 + NJ, CNU = 12 + NJ, CNZ = CNX + CNU;   // 35, 35, 70
static_assert(CNZ <= NDIR, "tangent directions");

// Centroidal momentum matrix A(q) (6 x NV: rows = [linear; angular about the centre of mass], world-aligned) for the
// generalized velocity [pdot, euler ZYX rates, qd_j], and the centre of mass.  Column c = momentum of the bodies moved by
// a unit rate of coordinate c (what pinocchio::computeCentroidalMap / ccrba leave in data.Ag).
template <class T>
void centroidal_map(const Oracle& o, const Kin<T>& k, T A[6][NV], V3<T>& com_out) {
  V3<T> com;
  V3<T> cb[NB];
  for (int i = 0; i < NB; ++i) {
    cb[i] = k.p[i] + k.R[i] * const_v3<T>(o.md.bodies[i].com);
    com = com + cb[i] * T(o.md.bodies[i].mass);
  }
  com = com * T(1.0 / o.total_mass);
  for (int r = 0; r < 6; ++r) for (int c = 0; c < NV; ++c) A[r][c] = T(0.0);
  for (int i = 0; i < NB; ++i) {
    const hsqp_body& b = o.md.bodies[i];
    const M3<T> Iw = k.R[i] * const_m3<T>(b.inertia) * transpose(k.R[i]);
    const V3<T> rc = cb[i] - com;
    for (int c = 0; c < NV; ++c) {
      V3<T> lin, ang;
      if (!jacobian_column(o, k, i, c, cb[i], lin, ang)) continue;
      const V3<T> pl = lin * T(b.mass);
      const V3<T> pa = Iw * ang + cross(rc, pl);
      for (int r = 0; r < 3; ++r) { A[r][c] = A[r][c] + pl[r]; A[3 + r][c] = A[3 + r][c] + pa[r]; }
    }
  }
  com_out = com;
}

// getNormalizedCentroidalMomentumRate (ASSUMPTION A7)
template <class T>
void cent_momentum_rate(const Oracle& o, const Kin<T>& k, const V3<T>& com, const T* u, T out[6]) {
  const T invm(1.0 / o.total_mass);
  V3<T> lin(T(0.0), T(0.0), T(-9.81 * o.total_mass)), ang;
  for (int f = 0; f < 2; ++f) {
    const hsqp_frame& fr = o.md.contact[f];
    const V3<T> pt = k.p[fr.body] + k.R[fr.body] * const_v3<T>(fr.p);
    const V3<T> force(u[6 * f], u[6 * f + 1], u[6 * f + 2]), moment(u[6 * f + 3], u[6 * f + 4], u[6 * f + 5]);
    lin = lin + force;
    ang = ang + cross(pt - com, force) + moment;
  }
  for (int r = 0; r < 3; ++r) { out[r] = lin[r] * invm; out[3 + r] = ang[r] * invm; }
}

// CentroidalModelPinocchioMapping::getPinocchioJointVelocity, FullCentroidalDynamics (ASSUMPTION A7):
// v = [A_b^-1 (m h - A_j qd_j); qd_j]
template <class T>
void cent_generalized_velocity(const T A[6][NV], const T* x, const T* u, T* v /*[NV]*/) {
  const T mass = A[0][0];
  T rhs[6];
  for (int r = 0; r < 6; ++r) {
    T s = mass * x[r];
    for (int j = 0; j < NJ; ++j) s = s - A[r][6 + j] * u[12 + j];
    rhs[r] = s;
  }
  M3<T> A22, A12;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { A22.m[i][j] = A[3 + i][3 + j]; A12.m[i][j] = A[i][3 + j]; }
  const V3<T> vang = inverse3(A22) * V3<T>(rhs[3], rhs[4], rhs[5]);
  const V3<T> vlin = (V3<T>(rhs[0], rhs[1], rhs[2]) - A12 * vang) * (T(1.0) / mass);
  for (int i = 0; i < 3; ++i) { v[i] = vlin[i]; v[3 + i] = vang[i]; }
  for (int j = 0; j < NJ; ++j) v[6 + j] = u[12 + j];
}

// PinocchioCentroidalDynamicsAD::getValueCppAd (ASSUMPTION A7)
template <class T>
void cent_flow_map(const Oracle& o, const T* x, const T* u, T* xdot) {
  const T* q = x + 6;
  Kin<T> k;
  forward_kinematics<T>(o, q, static_cast<const T*>(nullptr), static_cast<const T*>(nullptr), k);
  T A[6][NV];
  V3<T> com;
  centroidal_map<T>(o, k, A, com);
  cent_momentum_rate<T>(o, k, com, u, xdot);
  cent_generalized_velocity<T>(A, x, u, xdot + 6);
}

// Foot frame position, orientation error to the ground plane and twist [linear; angular] (LOCAL_WORLD_ALIGNED) at the
// velocity-level model: PinocchioEndEffectorKinematicsCppAd with the centroidal mapping and the velocity update callback of
// humanoid_nmpc/humanoid_centroidal_mpc/src/CentroidalMpcInterface.cpp:153-156,188-191 (positions from q, velocities from
// v = getPinocchioJointVelocity(x, u)).  The class itself lives in humanoid_common_mpc; its whole-body twin is restated in
// foot_kinematics() above.
template <class T>
void cent_foot_kinematics(const Oracle& o, const T* x, const T* u, FootKin<T> out[2]) {
  const T* q = x + 6;
  T v[NV];
  {
    Kin<T> k0;
    forward_kinematics<T>(o, q, static_cast<const T*>(nullptr), static_cast<const T*>(nullptr), k0);
    T A[6][NV];
    V3<T> com;
    centroidal_map<T>(o, k0, A, com);
    cent_generalized_velocity<T>(A, x, u, v);
  }
  Kin<T> k;
  forward_kinematics<T>(o, q, v, static_cast<const T*>(nullptr), k);
  const V3<T> n(T(0.0), T(0.0), T(1.0));
  for (int f = 0; f < 2; ++f) {
    const hsqp_frame& fr = o.md.contact[f];
    const int b = fr.body;
    const V3<T> r = k.R[b] * const_v3<T>(fr.p);
    out[f].R = k.R[b];
    out[f].pos = k.p[b] + r;
    out[f].ori = rotation_distance_to_plane(k.R[b], n);
    out[f].vlin = k.v[b] + cross(k.om[b], r);
    out[f].vang = k.om[b];
  }
}

// Equality constraints of the centroidal problem, stacked per foot in the order the terms are added
// (humanoid_nmpc/humanoid_centroidal_mpc/src/CentroidalMpcInterface.cpp:203-207): zeroWrench (swing, 6 rows: W_f = 0,
// humanoid_common_mpc/src/constraint/ZeroWrenchConstraint.cpp:59-84), zeroVelocity (stance, 6 rows:
// b + Ax [pos; oriErr] + Av twist with Av = I, Ax(2,2) = positionErrorGain_z, Ax(3:6,3:6) = orientationErrorGain I,
// b[2] = -Ax(2,2) footReferenceHeight — CentroidalMpcInterface.cpp:232-257, ZeroVelocityConstraintCppAd.cpp:69-77,
// EndEffectorKinematicsTwistConstraint.cpp:83-99), normalVelocity (swing, 1 row: v_z - zvel* + positionErrorGain_z (z - zpos*)
// — HumanoidPreComputation.cpp:104-115, EndEffectorKinematicsLinearVelConstraint.cpp:75-86).
template <class T>
int cent_equalities(const Oracle& o, const T* x, const T* u, const int contact[2], const double zpos[2], const double zvel[2],
                    double gain_pos_z, double gain_ori, T* eq /*[NE_MAX]*/) {
  FootKin<T> fk[2];
  cent_foot_kinematics<T>(o, x, u, fk);
  int ne = 0;
  for (int f = 0; f < 2; ++f) {
    if (!contact[f]) for (int i = 0; i < 6; ++i) eq[ne++] = u[6 * f + i];
    if (contact[f]) {
      for (int i = 0; i < 3; ++i) eq[ne++] = fk[f].vlin[i] + (i == 2 ? (fk[f].pos[2] - T(zpos[f])) * T(gain_pos_z) : T(0.0));
      for (int i = 0; i < 3; ++i) eq[ne++] = fk[f].vang[i] + fk[f].ori[i] * T(gain_ori);
    } else {
      eq[ne++] = fk[f].vlin[2] - T(zvel[f]) + (fk[f].pos[2] - T(zpos[f])) * T(gain_pos_z);
    }
  }
  return ne;
}


// ===================================================================================== the centroidal OCP
// Terms in the order humanoid_nmpc/humanoid_centroidal_mpc/src/CentroidalMpcInterface.cpp:139-215 adds them.  The problem is
// EMBEDDED in the whole-body array layout (include/hsqp.h): state slots 0..34 of 58, the other 23 are decoupled padding states
// (A = I, no cost), so that NodeLQ / project_node / riccati / kkt_residual above are reused unchanged; tangent direction of
// state i is i, of input i is NX + i.
//
// ASSUMPTION A8 (sources absent, restated from upstream ocs2_robotic_tools/common/RotationTransforms.h):
//   matrixToQuaternion = the usual trace-based conversion (w >= 0 branch for the torso's near-upright attitudes),
//   quaternionDistance(q, qRef) = q.w qRef.vec - qRef.w q.vec + q.vec x qRef.vec.
// Not restated because they cannot contribute: ICPCost (icpErrorWeight 0 in g1_centroidal_mpc/config/mpc/task.info — the
// residual is multiplied by sqrt(0)), the position rows of the foot and torso task-space costs (weights 0 there as well; the
// HIP path rejects non-zero ones), mimic-joint constraints (no mimicJoints block in the task file).
template <class T>
struct CentTerms {
  FootKin<T> fk[2];
  V3<T> torso_err[3];    // orientation error (quaternionDistance), linear velocity error, angular velocity error
  T coll[16];
  T mxy[2][4];
  T tau[2][6];           // (J_ee^T W)[6 + joint] of the active leg joints
};

template <class T>
void quat_from_matrix(const M3<T>& R, T q[4] /* x y z w */) {
  const double tr = value_of(R.m[0][0]) + value_of(R.m[1][1]) + value_of(R.m[2][2]);
  if (tr > 0.0) {
    const T s = sqrt(R.m[0][0] + R.m[1][1] + R.m[2][2] + T(1.0)) * T(2.0);
    q[3] = s * T(0.25); q[0] = (R.m[2][1] - R.m[1][2]) / s; q[1] = (R.m[0][2] - R.m[2][0]) / s; q[2] = (R.m[1][0] - R.m[0][1]) / s;
  } else if (value_of(R.m[0][0]) > value_of(R.m[1][1]) && value_of(R.m[0][0]) > value_of(R.m[2][2])) {
    const T s = sqrt(T(1.0) + R.m[0][0] - R.m[1][1] - R.m[2][2]) * T(2.0);
    q[3] = (R.m[2][1] - R.m[1][2]) / s; q[0] = s * T(0.25); q[1] = (R.m[0][1] + R.m[1][0]) / s; q[2] = (R.m[0][2] + R.m[2][0]) / s;
  } else if (value_of(R.m[1][1]) > value_of(R.m[2][2])) {
    const T s = sqrt(T(1.0) + R.m[1][1] - R.m[0][0] - R.m[2][2]) * T(2.0);
    q[3] = (R.m[0][2] - R.m[2][0]) / s; q[0] = (R.m[0][1] + R.m[1][0]) / s; q[1] = s * T(0.25); q[2] = (R.m[1][2] + R.m[2][1]) / s;
  } else {
    const T s = sqrt(T(1.0) + R.m[2][2] - R.m[0][0] - R.m[1][1]) * T(2.0);
    q[3] = (R.m[1][0] - R.m[0][1]) / s; q[0] = (R.m[0][2] + R.m[2][0]) / s; q[1] = (R.m[1][2] + R.m[2][1]) / s; q[2] = s * T(0.25);
  }
}

template <class T>
void cent_terms_eval(const Oracle& o, const T* x, const T* u, const double* par, CentTerms<T>& out) {
  const hsqp_model_desc& md = o.md;
  const T* q = x + 6;
  cent_foot_kinematics<T>(o, x, u, out.fk);
  // full kinematics once more for the torso frame and the leg Jacobians (velocity-level model)
  T v[NV];
  Kin<T> k;
  {
    Kin<T> k0;
    forward_kinematics<T>(o, q, static_cast<const T*>(nullptr), static_cast<const T*>(nullptr), k0);
    T A[6][NV];
    V3<T> com;
    centroidal_map<T>(o, k0, A, com);
    cent_generalized_velocity<T>(A, x, u, v);
  }
  forward_kinematics<T>(o, q, v, static_cast<const T*>(nullptr), k);
  // EndEffectorKinematicsQuadraticCost::costVectorFunction (humanoid_common_mpc/src/cost/EndEffectorKinematicsQuadraticCost.cpp:110-138)
  {
    const int b = md.torso.body;
    const M3<T> Rt = k.R[b] * const_m3<T>(md.torso_R);
    const V3<T> r = k.R[b] * const_v3<T>(md.torso.p);
    T qc[4];
    quat_from_matrix<T>(Rt, qc);
    const double* ref = par + HSQP_PC_TORSO;   // pos(3) quat xyzw(4) vlin(3) vang(3)
    const V3<T> qv(qc[0], qc[1], qc[2]);
    const V3<T> rv = const_v3<T>(ref + 3);
    out.torso_err[0] = rv * qc[3] - qv * T(ref[6]) + cross(qv, rv);
    const V3<T> vl = k.v[b] + cross(k.om[b], r);
    out.torso_err[1] = vl - const_v3<T>(ref + 7);
    out.torso_err[2] = k.om[b] - const_v3<T>(ref + 10);
  }
  collision_distances<T>(o, q, out.coll);
  for (int f = 0; f < 2; ++f) {
    moment_xy<T>(o, out.fk[f].R, u, f, out.mxy[f]);
    // ExternalTorqueQuadraticCostAD::costVectorFunction (…/ExternalTorqueQuadraticCostAD.cpp:110-135): (J_ee^T W)[6 + joint]
    const V3<T> force(u[6 * f], u[6 * f + 1], u[6 * f + 2]), moment(u[6 * f + 3], u[6 * f + 4], u[6 * f + 5]);
    for (int a = 0; a < 6; ++a) {
      V3<T> lin, ang;
      const int c = 6 + md.ext_torque_joint[f][a];
      if (jacobian_column(o, k, md.contact[f].body, c, out.fk[f].pos, lin, ang)) out.tau[f][a] = dot(lin, force) + dot(ang, moment);
      else out.tau[f][a] = T(0.0);
    }
  }
}

void cent_seed(const double* x, const double* u, AD* xa, AD* ua) {
  for (int i = 0; i < CNX; ++i) xa[i] = AD::seed(x[i], i);
  for (int i = 0; i < CNU; ++i) ua[i] = AD::seed(u[i], NX + i);
}

// x_nom, u_nom: StateInputQuadraticCost::getStateInputDeviation with the centroidal model's accessors
// (getBaseComLinearVelocity = x[0:3], getBasePose(state)[3] = x[9], joint angles = x[12:]; CentroidalMpcRobotModel.h:100-145)
void cent_nominal(const Oracle& o, const double* x, const double* par, double* xnom, double* unom) {
  for (int i = 0; i < CNX; ++i) xnom[i] = par[HSQP_P_XDES + i];
  const double yaw = x[9];
  const double gcf = par[HSQP_P_ARMSWING] * (std::cos(yaw) * xnom[0] + std::sin(yaw) * xnom[1]);
  const double sgn[4] = {-0.15, 0.15, -0.15, 0.15};
  for (int a = 0; a < 4; ++a) xnom[12 + o.md.arm_swing_joint[a]] += sgn[a] * gcf;
  for (int i = 0; i < CNU; ++i) unom[i] = 0.0;
  const bool c0 = par[HSQP_P_CONTACT] > 0.5, c1 = par[HSQP_P_CONTACT + 1] > 0.5;
  const int ns = int(c0) + int(c1);
  if (ns > 0) {
    const double fz = o.total_mass * 9.81 / ns;
    if (c0) unom[2] = fz;
    if (c1) unom[8] = fz;
  }
}

// Stage cost + equality constraints of the centroidal problem at (x, u) (35 + 35 doubles): value only (lq = nullptr) or with
// the quadratic model in the padded 58 / 35 layout.  Returns l (unscaled).
double cent_stage_terms(const Oracle& o, const double* x, const double* u, const double* par, NodeLQ* lq, double* eq_out, int* ne_out) {
  const hsqp_model_desc& md = o.md;
  const bool contact[2] = {par[HSQP_P_CONTACT] > 0.5, par[HSQP_P_CONTACT + 1] > 0.5};
  double cost = 0.0;
  double* H = lq ? lq->H : nullptr;
  double* g = lq ? lq->g : nullptr;
  double* CDe = lq ? lq->CDe : nullptr;
  if (lq) { std::fill(H, H + NZ * NZ, 0.0); std::fill(g, g + NZ, 0.0); std::fill(CDe, CDe + NE_MAX * (NZ + 1), 0.0); }
  // (1) StateInputQuadraticCost
  double xnom[CNX], unom[CNU];
  cent_nominal(o, x, par, xnom, unom);
  for (int i = 0; i < CNX; ++i) {
    const double d = x[i] - xnom[i];
    cost += 0.5 * md.Q[i] * d * d;
    if (lq) { g[i] += md.Q[i] * d; H[i * NZ + i] += md.Q[i]; }
  }
  for (int i = 0; i < CNU; ++i) {
    const double d = u[i] - unom[i];
    cost += 0.5 * md.R[i] * d * d;
    if (lq) { g[NX + i] += md.R[i] * d; H[(NX + i) * NZ + NX + i] += md.R[i]; }
  }
  // all kinematic quantities, with tangents when the model is requested
  std::unique_ptr<CentTerms<AD>> ta;
  CentTerms<double> tv;
  if (lq) {
    std::vector<AD> xa(CNX), ua(CNU);
    cent_seed(x, u, xa.data(), ua.data());
    ta.reset(new CentTerms<AD>);
    cent_terms_eval<AD>(o, xa.data(), ua.data(), par, *ta);
  }
  cent_terms_eval<double>(o, x, u, par, tv);
  auto add_gn = [&](double r, const AD* ra, double scale) {   // 0.5 (scale r)^2
    cost += 0.5 * scale * scale * r * r;
    if (lq && scale != 0.0)
      for (int a = 0; a < NZ; ++a) {
        const double ja = scale * ra->d[a];
        if (ja == 0.0) continue;
        g[a] += ja * scale * r;
        for (int b = 0; b < NZ; ++b) H[a * NZ + b] += ja * scale * ra->d[b];
      }
  };
  auto add_pen = [&](const Pen& p, const AD* ha) {           // penalty on a linear-order constraint
    cost += p.p;
    if (lq)
      for (int a = 0; a < NZ; ++a) {
        if (ha->d[a] == 0.0) continue;
        g[a] += p.d1 * ha->d[a];
        for (int b = 0; b < NZ; ++b) H[a * NZ + b] += p.d2 * ha->d[a] * ha->d[b];
      }
  };
  // (2) torso task-space cost (position rows: weight 0, see the header)
  for (int blk = 0; blk < 3; ++blk)
    for (int c = 0; c < 3; ++c) add_gn(tv.torso_err[blk][c], lq ? &ta->torso_err[blk][c] : nullptr, md.torso_sqrt_w[3 + 3 * blk + c]);
  // (3) state soft constraints: joint limits, foot collision
  for (int j = 0; j < NJ; ++j) {
    const double qj = x[12 + j];
    const Pen lo = pwp_barrier(md.joint_limit_barrier.mu, md.joint_limit_barrier.delta, qj - md.bodies[1 + j].q_lo);
    const Pen hi = pwp_barrier(md.joint_limit_barrier.mu, md.joint_limit_barrier.delta, md.bodies[1 + j].q_hi - qj);
    cost += lo.p + hi.p;
    if (lq) { g[12 + j] += lo.d1 - hi.d1; H[(12 + j) * NZ + 12 + j] += lo.d2 + hi.d2; }
  }
  if (!(contact[0] && contact[1]))
    for (int r = 0; r < 16; ++r) add_pen(pwp_barrier(md.collision_barrier.mu, md.collision_barrier.delta, tv.coll[r]), lq ? &ta->coll[r] : nullptr);
  // (4) per foot
  int ne = 0;
  double eq[NE_MAX];
  auto eq_row = [&](double val, const AD* va, const AD* vb, double gb) {   // row = va + gb vb
    eq[ne] = val;
    if (lq) {
      for (int a = 0; a < NZ; ++a) CDe[ne * (NZ + 1) + a] = va->d[a] + (vb ? gb * vb->d[a] : 0.0);
      CDe[ne * (NZ + 1) + NZ] = val;
    }
    ++ne;
  };
  for (int f = 0; f < 2; ++f) {
    if (contact[f]) {   // friction cone (FrictionForceConeConstraint.cpp:78-224), relaxed barrier; same as the whole-body problem
      const double Fx = u[6 * f], Fy = u[6 * f + 1], Fz = u[6 * f + 2];
      const double T2 = Fx * Fx + Fy * Fy + md.friction_reg, Tn = std::sqrt(T2), T32 = Tn * T2;
      const Pen p = relaxed_barrier(md.friction_barrier.mu, md.friction_barrier.delta, md.friction_mu * (Fz + md.friction_grip) - Tn);
      cost += p.p;
      if (lq) {
        const double dh[3] = {-Fx / Tn, -Fy / Tn, md.friction_mu};
        const double d2[3][3] = {{-(Fy * Fy + md.friction_reg) / T32, Fx * Fy / T32, 0.0}, {Fx * Fy / T32, -(Fx * Fx + md.friction_reg) / T32, 0.0}, {0.0, 0.0, 0.0}};
        const int o0 = NX + 6 * f;
        for (int a = 0; a < 3; ++a) {
          g[o0 + a] += p.d1 * dh[a];
          for (int b = 0; b < 3; ++b) H[(o0 + a) * NZ + o0 + b] += p.d2 * dh[a] * dh[b] + p.d1 * d2[a][b];
        }
        // hessianDiagonalShift on every state and input of THIS problem (35 + 35)
        for (int a = 0; a < CNX; ++a) H[a * NZ + a] += p.d1 * (-md.friction_hess_shift);
        for (int a = 0; a < CNU; ++a) H[(NX + a) * NZ + NX + a] += p.d1 * (-md.friction_hess_shift);
      }
      for (int r = 0; r < 4; ++r) add_pen(relaxed_barrier(md.moment_barrier.mu, md.moment_barrier.delta, tv.mxy[f][r]), lq ? &ta->mxy[f][r] : nullptr);
    }
    const FootKin<double>& fk = tv.fk[f];
    if (!contact[f])    // zeroWrench
      for (int r = 0; r < 6; ++r) {
        eq[ne] = u[6 * f + r];
        if (lq) { CDe[ne * (NZ + 1) + NX + 6 * f + r] = 1.0; CDe[ne * (NZ + 1) + NZ] = eq[ne]; }
        ++ne;
      }
    if (contact[f]) {   // zeroVelocity: b + Ax [pos; oriErr] + twist, b[2] = -Ax(2,2) zpos*  (see cent_equalities)
      const double zp = par[HSQP_P_SWING + 3 * f];
      for (int i = 0; i < 3; ++i)
        eq_row(fk.vlin[i] + (i == 2 ? md.gain_pos_z * (fk.pos[2] - zp) : 0.0), lq ? &ta->fk[f].vlin[i] : nullptr, (lq && i == 2) ? &ta->fk[f].pos[2] : nullptr, md.gain_pos_z);
      for (int i = 0; i < 3; ++i) eq_row(fk.vang[i] + md.gain_ori * fk.ori[i], lq ? &ta->fk[f].vang[i] : nullptr, lq ? &ta->fk[f].ori[i] : nullptr, md.gain_ori);
    } else {            // normalVelocity
      const double zp = par[HSQP_P_SWING + 3 * f], zv = par[HSQP_P_SWING + 3 * f + 1];
      eq_row(fk.vlin[2] - zv + md.gain_pos_z * (fk.pos[2] - zp), lq ? &ta->fk[f].vlin[2] : nullptr, lq ? &ta->fk[f].pos[2] : nullptr, md.gain_pos_z);
    }
    // CentroidalMpcEndEffectorFootCost (…/CentroidalMpcEndEffectorFootCost.cpp:90-152): [pos - 0, oriErr, (v - 0) ip, w - 0] .* sqrtW
    {
      const double ip = par[HSQP_P_IMPACT + f];
      for (int c = 0; c < 3; ++c) add_gn(fk.ori[c], lq ? &ta->fk[f].ori[c] : nullptr, md.cent_foot_sqrt_w[3 + c]);
      for (int c = 0; c < 3; ++c) add_gn(fk.vlin[c], lq ? &ta->fk[f].vlin[c] : nullptr, md.cent_foot_sqrt_w[6 + c] * ip);
      for (int c = 0; c < 3; ++c) add_gn(fk.vang[c], lq ? &ta->fk[f].vang[c] : nullptr, md.cent_foot_sqrt_w[9 + c]);
    }
    // ExternalTorqueQuadraticCostAD (active in contact): tau .* sqrtW * (1 - impactProximity of the OTHER foot)
    if (contact[f]) {
      const double mid = 1.0 - par[HSQP_P_IMPACT + (1 - f)];
      for (int a = 0; a < 6; ++a) add_gn(tv.tau[f][a], lq ? &ta->tau[f][a] : nullptr, md.ext_torque_sqrt_w[f][a] * mid);
    }
  }
  if (lq) lq->ne = ne;
  if (ne_out) *ne_out = ne;
  if (eq_out) for (int i = 0; i < ne; ++i) eq_out[i] = eq[i];
  return cost;
}

// RK4 sensitivity of the centroidal flow map into the padded [A|B] (58 x 93): padding states keep A = I
void cent_rk4_sensitivity(const Oracle& o, const double* x, const double* u, double dt, double* AB, double* xnext, double* flow0) {
  AD xa[CNX], ua[CNU], k1[CNX], k2[CNX], k3[CNX], k4[CNX], xs[CNX];
  cent_seed(x, u, xa, ua);
  cent_flow_map<AD>(o, xa, ua, k1);
  for (int i = 0; i < CNX; ++i) xs[i] = xa[i] + k1[i] * (0.5 * dt);
  cent_flow_map<AD>(o, xs, ua, k2);
  for (int i = 0; i < CNX; ++i) xs[i] = xa[i] + k2[i] * (0.5 * dt);
  cent_flow_map<AD>(o, xs, ua, k3);
  for (int i = 0; i < CNX; ++i) xs[i] = xa[i] + k3[i] * dt;
  cent_flow_map<AD>(o, xs, ua, k4);
  if (AB) { std::fill(AB, AB + NX * NZ, 0.0); for (int i = CNX; i < NX; ++i) AB[i * NZ + i] = 1.0; }
  for (int i = 0; i < CNX; ++i) {
    const AD xn = xa[i] + (k1[i] + k2[i] * 2.0 + k3[i] * 2.0 + k4[i]) * (dt / 6.0);
    xnext[i] = xn.v;
    if (AB) for (int c = 0; c < NZ; ++c) AB[i * NZ + c] = xn.d[c];
    if (flow0) flow0[i] = k1[i].v;
  }
}
void cent_rk4_value(const Oracle& o, const double* x, const double* u, double dt, double* xnext) {
  double k1[CNX], k2[CNX], k3[CNX], k4[CNX], xs[CNX];
  cent_flow_map<double>(o, x, u, k1);
  for (int i = 0; i < CNX; ++i) xs[i] = x[i] + 0.5 * dt * k1[i];
  cent_flow_map<double>(o, xs, u, k2);
  for (int i = 0; i < CNX; ++i) xs[i] = x[i] + 0.5 * dt * k2[i];
  cent_flow_map<double>(o, xs, u, k3);
  for (int i = 0; i < CNX; ++i) xs[i] = x[i] + dt * k3[i];
  cent_flow_map<double>(o, xs, u, k4);
  for (int i = 0; i < CNX; ++i) xnext[i] = x[i] + dt / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
}

// x, xnext: padded 58-double rows (first 35 used); u: 35
void cent_node_lq(const Oracle& o, const double* x, const double* u, const double* xnext, const double* par, double dt, NodeLQ& lq) {
  if (dt == 0.0) { jump_node_lq(x, xnext, lq); return; }   // event interval (the padding states are zero on both sides)
  double phi[CNX];
  std::fill(lq.flow, lq.flow + NX, 0.0);
  cent_rk4_sensitivity(o, x, u, dt, lq.AB, phi, lq.flow);
  for (int i = 0; i < NX; ++i) lq.b[i] = i < CNX ? phi[i] - xnext[i] : 0.0;
  const double l = cent_stage_terms(o, x, u, par, &lq, nullptr, nullptr);
  lq.cost = dt * l;
  for (int i = 0; i < NZ * NZ; ++i) lq.H[i] *= dt;
  for (int i = 0; i < NZ; ++i) lq.g[i] *= dt;
}

void cent_performance(const Oracle& o, int N, double dt, const double* x, const double* u, const double* par, int threads, hsqp_perf* out) {
  double cost = 0.0, dyn = 0.0, eqs = 0.0;
#pragma omp parallel for num_threads(threads) reduction(+ : cost, dyn, eqs) schedule(dynamic)
  for (int k = 0; k < N; ++k) {
    double phi[CNX], eq[NE_MAX];
    int ne = 0;
    const double dtk = dt_at(o, k, dt);
    if (dtk == 0.0) {
      for (int i = 0; i < CNX; ++i) { const double d = x[k * NX + i] - x[(k + 1) * NX + i]; dyn += d * d; }
      continue;
    }
    cent_rk4_value(o, x + k * NX, u + k * NU, dtk, phi);
    for (int i = 0; i < CNX; ++i) { const double d = phi[i] - x[(k + 1) * NX + i]; dyn += dtk * d * d; }
    cost += dtk * cent_stage_terms(o, x + k * NX, u + k * NU, par + k * NP, nullptr, eq, &ne);
    for (int i = 0; i < ne; ++i) eqs += dtk * eq[i] * eq[i];
  }
  cost += terminal_cost(o, x + N * NX, par + N * NP, nullptr, nullptr);   // Qf is zero on the padding states
  out->cost = cost; out->dynamics_sse = dyn; out->equality_sse = eqs; out->merit = cost;
}

}  // namespace

extern "C" {

// A (6 x 29, row-major) and the centre of mass at q = [p, eulerZYX, q_j]
void orc_cent_momentum_matrix(void* h, const double* q, double* A, double* com) {
  const Oracle& o = *static_cast<Oracle*>(h);
  Kin<double> k;
  forward_kinematics<double>(o, q, nullptr, nullptr, k);
  double Am[6][NV];
  V3<double> c;
  centroidal_map<double>(o, k, Am, c);
  for (int r = 0; r < 6; ++r) for (int j = 0; j < NV; ++j) A[r * NV + j] = Am[r][j];
  for (int r = 0; r < 3; ++r) com[r] = c[r];
}

void orc_cent_momentum_rate(void* h, const double* q, const double* u, double* out) {
  const Oracle& o = *static_cast<Oracle*>(h);
  Kin<double> k;
  forward_kinematics<double>(o, q, nullptr, nullptr, k);
  double Am[6][NV];
  V3<double> c;
  centroidal_map<double>(o, k, Am, c);
  cent_momentum_rate<double>(o, k, c, u, out);
}

void orc_cent_flow_map(void* h, const double* x, const double* u, double* xdot) { cent_flow_map<double>(*static_cast<Oracle*>(h), x, u, xdot); }

// xdot (35) and its Jacobian wrt [x; u] (35 x 70)
void orc_cent_flow_map_jac(void* h, const double* x, const double* u, double* xdot, double* J) {
  const Oracle& o = *static_cast<Oracle*>(h);
  AD xa[CNX], ua[CNU], f[CNX];
  for (int i = 0; i < CNX; ++i) xa[i] = AD::seed(x[i], i);
  for (int i = 0; i < CNU; ++i) ua[i] = AD::seed(u[i], CNX + i);
  cent_flow_map<AD>(o, xa, ua, f);
  for (int i = 0; i < CNX; ++i) { xdot[i] = f[i].v; for (int c = 0; c < CNZ; ++c) J[i * CNZ + c] = f[i].d[c]; }
}

// Foot frame kinematics at the velocity-level model: out[f] = {pos(3), oriErr(3), vlin(3), vang(3)}
void orc_cent_foot_kinematics(void* h, const double* x, const double* u, double* out /*[2][12]*/) {
  FootKin<double> fk[2];
  cent_foot_kinematics<double>(*static_cast<Oracle*>(h), x, u, fk);
  for (int f = 0; f < 2; ++f)
    for (int i = 0; i < 3; ++i) { out[12 * f + i] = fk[f].pos[i]; out[12 * f + 3 + i] = fk[f].ori[i]; out[12 * f + 6 + i] = fk[f].vlin[i]; out[12 * f + 9 + i] = fk[f].vang[i]; }
}

// Equality constraints (value, and if J != null the Jacobian wrt [x; u], rows x 70); returns the number of rows
int orc_cent_equalities(void* h, const double* x, const double* u, const int* contact, const double* zpos, const double* zvel,
                        double gain_pos_z, double gain_ori, double* eq, double* J) {
  const Oracle& o = *static_cast<Oracle*>(h);
  if (!J) return cent_equalities<double>(o, x, u, contact, zpos, zvel, gain_pos_z, gain_ori, eq);
  AD xa[CNX], ua[CNU], e[NE_MAX];
  for (int i = 0; i < CNX; ++i) xa[i] = AD::seed(x[i], i);
  for (int i = 0; i < CNU; ++i) ua[i] = AD::seed(u[i], CNX + i);
  const int ne = cent_equalities<AD>(o, xa, ua, contact, zpos, zvel, gain_pos_z, gain_ori, e);
  for (int r = 0; r < ne; ++r) { eq[r] = e[r].v; for (int c = 0; c < CNZ; ++c) J[r * CNZ + c] = e[r].d[c]; }
  return ne;
}


// ---- the centroidal OCP: padded layout (x rows of 58 doubles, the first 35 used)
double orc_cent_stage_cost(void* h, const double* x, const double* u, const double* par, double* eq, int* ne) {
  return cent_stage_terms(*static_cast<Oracle*>(h), x, u, par, nullptr, eq, ne);
}
void orc_cent_rk4(void* h, const double* x, const double* u, double dt, double* xnext) { cent_rk4_value(*static_cast<Oracle*>(h), x, u, dt, xnext); }

// raw term values for unit tests: torso errors (9), collision (16), moment xy (8), external torques (12)
void orc_cent_terms(void* h, const double* x, const double* u, const double* par, double* out /*[45]*/) {
  CentTerms<double> t;
  cent_terms_eval<double>(*static_cast<Oracle*>(h), x, u, par, t);
  int n = 0;
  for (int b = 0; b < 3; ++b) for (int c = 0; c < 3; ++c) out[n++] = t.torso_err[b][c];
  for (int r = 0; r < 16; ++r) out[n++] = t.coll[r];
  for (int f = 0; f < 2; ++f) for (int r = 0; r < 4; ++r) out[n++] = t.mxy[f][r];
  for (int f = 0; f < 2; ++f) for (int r = 0; r < 6; ++r) out[n++] = t.tau[f][r];
}

void orc_cent_lq(void* h, int N, double dt, const double* x, const double* u, const double* par, int threads,
                 double* AB, double* b, double* H, double* g, double* CDe, int* ne, double* cost, double* flow) {
  const Oracle& o = *static_cast<Oracle*>(h);
#pragma omp parallel for num_threads(threads) schedule(dynamic)
  for (int k = 0; k < N; ++k) {
    std::unique_ptr<NodeLQ> lq(new NodeLQ);
    cent_node_lq(o, x + k * NX, u + k * NU, x + (k + 1) * NX, par + k * NP, dt_at(o, k, dt), *lq);
    if (AB) std::copy(lq->AB, lq->AB + NX * NZ, AB + (size_t)k * NX * NZ);
    if (b) std::copy(lq->b, lq->b + NX, b + (size_t)k * NX);
    if (H) std::copy(lq->H, lq->H + NZ * NZ, H + (size_t)k * NZ * NZ);
    if (g) std::copy(lq->g, lq->g + NZ, g + (size_t)k * NZ);
    if (CDe) std::copy(lq->CDe, lq->CDe + NE_MAX * (NZ + 1), CDe + (size_t)k * NE_MAX * (NZ + 1));
    if (ne) ne[k] = lq->ne;
    if (cost) cost[k] = lq->cost;
    if (flow) std::copy(lq->flow, lq->flow + NX, flow + (size_t)k * NX);
  }
}

// One SQP iteration of the centroidal problem (same outputs as orc_sqp_iteration); 0 ok, -4 numeric failure
int orc_cent_sqp_iteration(void* h, int N, double dt, const double* x_init, const double* x, const double* u, const double* par, int threads,
                           double* dx, double* du, double* x_new, double* u_new, hsqp_perf* before, hsqp_perf* after, double* kkt, double* armijo_out) {
  const Oracle& o = *static_cast<Oracle*>(h);
  std::vector<Projected> st(N);
  int fail = 0;
#pragma omp parallel for num_threads(threads) schedule(dynamic)
  for (int k = 0; k < N; ++k) {
    std::unique_ptr<NodeLQ> lq(new NodeLQ);
    cent_node_lq(o, x + k * NX, u + k * NU, x + (k + 1) * NX, par + k * NP, dt_at(o, k, dt), *lq);
    if (!project_node(*lq, st[k])) {
#pragma omp atomic write
      fail = 1;
    }
  }
  if (fail) return HSQP_ERR_NUMERIC;
  double HN[NX], gN[NX], dx0[NX];
  terminal_cost(o, x + N * NX, par + N * NP, HN, gN);
  for (int i = 0; i < NX; ++i) dx0[i] = x_init[i] - x[i];
  RiccatiOut r;
  if (!riccati(st, HN, gN, dx0, N, r)) return HSQP_ERR_NUMERIC;
  for (int k = 0; k <= N; ++k) for (int i = 0; i < NX; ++i) { dx[k * NX + i] = r.dx[k * NX + i]; x_new[k * NX + i] = x[k * NX + i] + r.dx[k * NX + i]; }
  for (int k = 0; k < N; ++k) {
    const Projected& p = st[k];
    for (int i = 0; i < NU; ++i) {
      double a = p.Pe[i];
      for (int j = 0; j < NX; ++j) a += p.Px[i * NX + j] * r.dx[k * NX + j];
      for (int j = 0; j < p.nut; ++j) a += p.Pu[i * p.nut + j] * r.ut[k * NU + j];
      du[k * NU + i] = a; u_new[k * NU + i] = u[k * NU + i] + a;
    }
  }
  if (kkt) kkt_residual(st, HN, gN, dx0, N, r, &kkt[0], &kkt[1]);
  if (armijo_out) {  // ASSUMPTION A5 (as orc_sqp_iteration)
    double am = 0.0;
    for (int k = 0; k < N; ++k) {
      const Projected& p = st[k];
      for (int i = 0; i < NX; ++i) am += p.qt[i] * r.dx[k * NX + i];
      for (int j = 0; j < p.nut; ++j) am += p.rt[j] * r.ut[k * NU + j];
    }
    for (int i = 0; i < NX; ++i) am += gN[i] * r.dx[N * NX + i];
    *armijo_out = am;
  }
  if (before) cent_performance(o, N, dt, x, u, par, threads, before);
  if (after) cent_performance(o, N, dt, x_new, u_new, par, threads, after);
  return 0;
}
// filter line search of the centroidal problem (ASSUMPTION A6, the code of orc_linesearch on the centroidal performance index)
void orc_cent_linesearch(void* h, int N, double dt, const double* x, const double* u, const double* dx, const double* du, const double* par,
                         int threads, const double* settings, double armijo, double* alpha_out, int* type_out, int* trials_out,
                         double* x_new, double* u_new, hsqp_perf* perf_new) {
  const Oracle& o = *static_cast<Oracle*>(h);
  filter_linesearch([&](const double* xx, const double* uu, hsqp_perf* out) { cent_performance(o, N, dt, xx, uu, par, threads, out); }, N, x, u, dx, du,
                    settings, armijo, alpha_out, type_out, trials_out, x_new, u_new, perf_new);
}
void orc_cent_performance(void* h, int N, double dt, const double* x, const double* u, const double* par, int threads, hsqp_perf* out) {
  cent_performance(*static_cast<Oracle*>(h), N, dt, x, u, par, threads, out);
}

}  // extern "C"
