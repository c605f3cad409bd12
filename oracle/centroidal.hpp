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

}  // extern "C"
