// =====================================================================================
// TEST INFRASTRUCTURE — CPU ORACLE.  NOT PART OF THE PRODUCT PATH.
//
// Plain C++ (g++, no third-party libraries) restatement of one multiple-shooting SQP
// iteration of the whole-body humanoid MPC of manumerous/wb_humanoid_mpc for the
// Unitree G1.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// may load this library; the HIP library never calls into it.
//
// PARITY STATUS: **unpinned**.  The arithmetic of the reference's hot path lives in the
// un-vendored, empty submodule lib/ocs2_ros2 (manumerous/ocs2_ros2, commit unknown:
// /root/reference/.gitmodules:13-15) and in Pinocchio / CppAD / HPIPM, none of which
// exist in the reference mount or in this image, and no reference test pins a solver
// result.  The problem definition (state/input layout, flow map,
// every cost and constraint term, task weights) is restated from the files cited on
// each function below (paths relative to /root/reference); the solver semantics
// (RK4 sensitivity discretisation, dt-scaled costs, QR constraint projection,
// Riccati recursion, performance index) follow upstream leggedrobotics/ocs2 as
// recorded in SURVEY.md Appendix A.  What *is* pinned is checked in tests/:
// total mass, weight compensation, contact positions, centre of pressure, frame
// rotation identities, state layout.
//
// Documented assumptions (each is marked ASSUMPTION where it is used):
//   A1  PieceWisePolynomialBarrierPenalty(mu,delta): fork-only, source absent.
//       p(h) = 0 for h >= delta, mu*((delta-h)/delta)^3 below (C2, p(0) = mu).
//   A2  rotationMatrixDistanceToPlane(R,n) == quaternionDistanceToPlane(q(R),n) ==
//       quaternionDistance(shortestArc(R*ez -> n), Identity) with upstream's
//       quaternionDistance(q,qRef) = q.w*qRef.vec - qRef.w*q.vec + q.vec x qRef.vec
//       (hint: robot_models/unitree_g1/g1_centroidal_mpc/test/testPinocchioModel.cpp:200-227).
//   A3  Uniform time grid without event nodes; PerformanceIndex terms scaled by dt as
//       in upstream multiple_shooting::computeIntermediatePerformance.
//   A4  Derivatives by forward-mode dual numbers instead of CppAD tapes.
//   A5  Armijo descent metric = sum over nodes of (projected cost gradient) . (dx, u~) + terminal gradient . dx_N:
//       upstream SqpSolver::getOCPSolution evaluates multiple_shooting::armijoDescentMetric(cost_, dx, du) on the
//       PROJECTED cost before the inputs are mapped back (restated from the published ocs2_sqp sources; the fork's
//       copy is absent).
//   A6  Filter line search = upstream ocs2::FilterLinesearch::acceptStep (three branches on the total constraint
//       violation sqrt(dynamicsSSE + equalitySSE)) inside the back-tracking loop of SqpSolver::takeStep:
//       alpha = 1; evaluate; accept -> done; else alpha *= alpha_decay; escape with a ZERO step when
//       alpha*|dx| and alpha*|du| are both below deltaTol, or when alpha < alpha_min.  |.| = sqrt of the sum of
//       squared node norms (multiple_shooting::trajectoryNorm).  g_max, g_min, deltaTol: reference task.info
//       (robot_models/unitree_g1/g1_wb_mpc/config/mpc/task.info:81-90); gamma_c, armijoFactor, alpha_decay,
//       alpha_min: upstream sqp::Settings defaults.
// =====================================================================================
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../include/hsqp.h"
#include "dual.hpp"

namespace {

constexpr int NJ = HSQP_NJ, NV = HSQP_NV, NX = HSQP_NX, NU = HSQP_NU, NB = HSQP_NB, NZ = NX + NU;
constexpr int NDIR = 96;  // tangent directions (93 used) — multiple of 8 for AVX-512
constexpr int NE_MAX = 14;
constexpr int NP = HSQP_NODE_PARAMS;
using AD = Dual<NDIR>;

// ----------------------------------------------------------------------------- small LA
template <class T>
struct V3 {
  T x, y, z;
  V3() : x(0.0), y(0.0), z(0.0) {}
  V3(T a, T b, T c) : x(a), y(b), z(c) {}
  T& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  const T& operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
template <class T> V3<T> operator+(const V3<T>& a, const V3<T>& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T> V3<T> operator-(const V3<T>& a, const V3<T>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> V3<T> operator*(const V3<T>& a, const T& s) { return {a.x * s, a.y * s, a.z * s}; }
template <class T> V3<T> cross(const V3<T>& a, const V3<T>& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class T> T dot(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

template <class T>
struct M3 {
  T m[3][3];
  M3() { for (auto& r : m) for (auto& e : r) e = T(0.0); }
  static M3 identity() { M3 r; r.m[0][0] = r.m[1][1] = r.m[2][2] = T(1.0); return r; }
  V3<T> col(int j) const { return {m[0][j], m[1][j], m[2][j]}; }
};
template <class T> M3<T> operator*(const M3<T>& a, const M3<T>& b) {
  M3<T> r;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    T s(0.0);
    for (int k = 0; k < 3; ++k) s = s + a.m[i][k] * b.m[k][j];
    r.m[i][j] = s;
  }
  return r;
}
template <class T> V3<T> operator*(const M3<T>& a, const V3<T>& v) {
  return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
          a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
template <class T> M3<T> transpose(const M3<T>& a) {
  M3<T> r;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
  return r;
}
template <class T> V3<T> tmul(const M3<T>& a, const V3<T>& v) { return transpose(a) * v; }  // a^T v
template <class T> M3<T> inverse3(const M3<T>& a) {  // Eigen's Matrix3::inverse() is the cofactor formula too
  M3<T> c;
  c.m[0][0] = a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1];
  c.m[0][1] = a.m[0][2] * a.m[2][1] - a.m[0][1] * a.m[2][2];
  c.m[0][2] = a.m[0][1] * a.m[1][2] - a.m[0][2] * a.m[1][1];
  c.m[1][0] = a.m[1][2] * a.m[2][0] - a.m[1][0] * a.m[2][2];
  c.m[1][1] = a.m[0][0] * a.m[2][2] - a.m[0][2] * a.m[2][0];
  c.m[1][2] = a.m[0][2] * a.m[1][0] - a.m[0][0] * a.m[1][2];
  c.m[2][0] = a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0];
  c.m[2][1] = a.m[0][1] * a.m[2][0] - a.m[0][0] * a.m[2][1];
  c.m[2][2] = a.m[0][0] * a.m[1][1] - a.m[0][1] * a.m[1][0];
  T det = a.m[0][0] * c.m[0][0] + a.m[0][1] * c.m[1][0] + a.m[0][2] * c.m[2][0];
  T inv = T(1.0) / det;
  for (auto& r : c.m) for (auto& e : r) e = e * inv;
  return c;
}
template <class T> M3<T> rot_axis(const double axis[3], const T& q) {  // Rodrigues for a unit axis
  const T c = cos(q), s = sin(q), t = T(1.0) - c;
  const double x = axis[0], y = axis[1], z = axis[2];
  M3<T> r;
  r.m[0][0] = t * (x * x) + c;       r.m[0][1] = t * (x * y) - s * z;   r.m[0][2] = t * (x * z) + s * y;
  r.m[1][0] = t * (x * y) + s * z;   r.m[1][1] = t * (y * y) + c;       r.m[1][2] = t * (y * z) - s * x;
  r.m[2][0] = t * (x * z) - s * y;   r.m[2][1] = t * (y * z) + s * x;   r.m[2][2] = t * (z * z) + c;
  return r;
}
template <class T> M3<T> const_m3(const double* a) {
  M3<T> r;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = T(a[3 * i + j]);
  return r;
}
template <class T> V3<T> const_v3(const double* a) { return {T(a[0]), T(a[1]), T(a[2])}; }

// ----------------------------------------------------------------------------- model
struct Oracle {
  hsqp_model_desc md;
  bool anc[NB][NB];  // anc[j][i]: body j is i or an ancestor of i
  double total_mass;
  // optional non-uniform time grid with event intervals (hsqp_problem::dt_nodes of ONE instance; set by orc_set_grid): empty = uniform
  std::vector<double> grid;
};
// length of interval k: the grid's, or the uniform dt
inline double dt_at(const Oracle& o, int k, double dt) { return o.grid.empty() ? dt : o.grid[(size_t)k]; }

void init_oracle(Oracle& o, const hsqp_model_desc* md) {
  o.md = *md;
  for (int j = 0; j < NB; ++j) for (int i = 0; i < NB; ++i) o.anc[j][i] = false;
  o.total_mass = 0.0;
  for (int i = 0; i < NB; ++i) {
    o.total_mass += md->bodies[i].mass;
    int k = i;
    while (k >= 0) { o.anc[k][i] = true; k = md->bodies[k].parent; }
  }
}

// ----------------------------------------------------------------------------- kinematics
// Pinocchio conventions restated: q = [p, eulerZ, eulerY, eulerX, q_j];
// v = [pdot (world), euler rates, qd_j]; base joint = composite Translation + SphericalZYX
// (humanoid_nmpc/humanoid_common_mpc/src/pinocchio_model/createPinocchioModel.cpp:60-67), which is
// kinematically the chain prismatic x,y,z -> revolute z -> revolute y' -> revolute x''.
template <class T>
struct Kin {
  M3<T> R[NB];
  V3<T> p[NB];    // body frame origin, world
  V3<T> w[NB];    // joint axis, world (w[0] unused)
  V3<T> om[NB], v[NB], al[NB], a[NB];  // angular vel, origin linear vel, angular acc, origin classical linear acc
  V3<T> eax[3];   // world axes of the euler Z, Y, X rates
};

template <class T>
void forward_kinematics(const Oracle& o, const T* q, const T* v, const T* vd, Kin<T>& k) {
  const double ez[3] = {0, 0, 1}, ey[3] = {0, 1, 0}, ex[3] = {1, 0, 0};
  const M3<T> Rz = rot_axis<T>(ez, q[3]), Ry = rot_axis<T>(ey, q[4]), Rx = rot_axis<T>(ex, q[5]);
  const M3<T> Rzy = Rz * Ry;
  k.R[0] = Rzy * Rx;
  k.p[0] = {q[0], q[1], q[2]};
  k.eax[0] = {T(0.0), T(0.0), T(1.0)};
  k.eax[1] = Rz.col(1);
  k.eax[2] = Rzy.col(0);
  const T zero(0.0);
  const T vz = v ? v[3] : zero, vy = v ? v[4] : zero, vx = v ? v[5] : zero;
  const V3<T> om_z = k.eax[0] * vz, om_zy = om_z + k.eax[1] * vy;
  k.om[0] = om_zy + k.eax[2] * vx;
  k.v[0] = v ? V3<T>(v[0], v[1], v[2]) : V3<T>();
  // alpha = sum w_k * edd_k + (angular velocity of the frame carrying w_k) x w_k * ed_k
  V3<T> al = cross(om_z, k.eax[1] * vy) + cross(om_zy, k.eax[2] * vx);
  if (vd) al = al + k.eax[0] * vd[3] + k.eax[1] * vd[4] + k.eax[2] * vd[5];
  k.al[0] = al;
  k.a[0] = vd ? V3<T>(vd[0], vd[1], vd[2]) : V3<T>();
  for (int i = 1; i < NB; ++i) {
    const hsqp_body& b = o.md.bodies[i];
    const int par = b.parent;
    const M3<T> Rj = k.R[par] * const_m3<T>(b.R);
    k.R[i] = Rj * rot_axis<T>(b.axis, q[5 + i]);
    const V3<T> r = k.R[par] * const_v3<T>(b.p);
    k.p[i] = k.p[par] + r;
    k.w[i] = Rj * const_v3<T>(b.axis);
    const T qd = v ? v[5 + i] : zero;
    k.om[i] = k.om[par] + k.w[i] * qd;
    k.v[i] = k.v[par] + cross(k.om[par], r);
    V3<T> ali = k.al[par] + cross(k.om[par], k.w[i] * qd);
    if (vd) ali = ali + k.w[i] * vd[5 + i];
    k.al[i] = ali;
    k.a[i] = k.a[par] + cross(k.al[par], r) + cross(k.om[par], cross(k.om[par], r));
  }
}

// Column c of the LOCAL_WORLD_ALIGNED Jacobian of a point fixed to body i (pinocchio::computeFrameJacobian).
template <class T>
bool jacobian_column(const Oracle& o, const Kin<T>& k, int i, int c, const V3<T>& point, V3<T>& lin, V3<T>& ang) {
  if (c < 3) {
    lin = V3<T>(); lin[c] = T(1.0); ang = V3<T>();
    return true;
  }
  if (c < 6) {
    ang = k.eax[c - 3];
    lin = cross(ang, point - k.p[0]);
    return true;
  }
  const int j = c - 5;
  if (!o.anc[j][i]) return false;
  ang = k.w[j];
  lin = cross(ang, point - k.p[j]);
  return true;
}

// ----------------------------------------------------------------------------- flow map
// computeBaseAcceleration(state,input,...) —
//   humanoid_nmpc/humanoid_wb_mpc/src/dynamics/DynamicsHelperFunctions.cpp:52-82  (crba, nonLinearEffects, foot Jacobians)
//   humanoid_nmpc/humanoid_common_mpc/src/pinocchio_model/DynamicsHelperFunctions.cpp:197-218
//     (M_lin and M_ang 3x3 blocks inverted separately; the lin/ang coupling block is ignored — reproduced literally)
// Only rows 0..5 of M and nle are ever read by the reference, so only those are formed:
//   M[r][c]  = sum_i  Jc_i[:,r]^T m_i Jc_i[:,c] + Jw_i[:,r]^T I_i Jw_i[:,c]            (crba)
//   nle[r]   = sum_i  Jc_i[:,r]^T m_i (acom_i|qdd=0 + g e_z) + Jw_i[:,r]^T (I_i alpha_i|qdd=0 + om_i x I_i om_i)   (rnea(q,v,0))
template <class T>
void base_acceleration(const Oracle& o, const T* x, const T* u, T* ab /*[6]*/, T* M6out /*[6*NV] or null*/, T* nle6out /*[6] or null*/) {
  const T* q = x;
  const T* v = x + NV;
  Kin<T> k;
  forward_kinematics<T>(o, q, v, static_cast<const T*>(nullptr), k);
  T M6[6][NV];
  T nle6[6];
  for (int r = 0; r < 6; ++r) { nle6[r] = T(0.0); for (int c = 0; c < NV; ++c) M6[r][c] = T(0.0); }
  const V3<T> grav(T(0.0), T(0.0), T(o.md.gravity));  // -g_vector
  for (int i = 0; i < NB; ++i) {
    const hsqp_body& b = o.md.bodies[i];
    const V3<T> rc = k.R[i] * const_v3<T>(b.com);
    const V3<T> com = k.p[i] + rc;
    const M3<T> Iw = k.R[i] * const_m3<T>(b.inertia) * transpose(k.R[i]);
    const V3<T> acom = k.a[i] + cross(k.al[i], rc) + cross(k.om[i], cross(k.om[i], rc));
    const V3<T> f = (acom + grav) * T(b.mass);
    const V3<T> n = Iw * k.al[i] + cross(k.om[i], Iw * k.om[i]);
    V3<T> rl[6], ra[6];
    for (int r = 0; r < 6; ++r) {
      jacobian_column(o, k, i, r, com, rl[r], ra[r]);
      nle6[r] = nle6[r] + dot(rl[r], f) + dot(ra[r], n);
    }
    for (int c = 0; c < NV; ++c) {
      V3<T> cl, ca;
      if (!jacobian_column(o, k, i, c, com, cl, ca)) continue;
      const V3<T> ml = cl * T(b.mass);
      const V3<T> Ia = Iw * ca;
      for (int r = 0; r < 6; ++r) M6[r][c] = M6[r][c] + dot(rl[r], ml) + dot(ra[r], Ia);
    }
  }
  // foot Jacobians (LOCAL_WORLD_ALIGNED), base 6x6 blocks: tau_b = sum_i J_b,i^T W_i
  T inter[6];
  for (int r = 0; r < 6; ++r) {
    T s = -nle6[r];
    for (int j = 0; j < NJ; ++j) s = s - M6[r][6 + j] * u[12 + j];
    inter[r] = s;
  }
  for (int f = 0; f < 2; ++f) {
    const hsqp_frame& fr = o.md.contact[f];
    const V3<T> pt = k.p[fr.body] + k.R[fr.body] * const_v3<T>(fr.p);
    const V3<T> force(u[6 * f], u[6 * f + 1], u[6 * f + 2]), moment(u[6 * f + 3], u[6 * f + 4], u[6 * f + 5]);
    for (int r = 0; r < 6; ++r) {
      V3<T> l, a;
      jacobian_column(o, k, fr.body, r, pt, l, a);
      inter[r] = inter[r] + dot(l, force) + dot(a, moment);
    }
  }
  M3<T> Mlin, Mang;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { Mlin.m[i][j] = M6[i][j]; Mang.m[i][j] = M6[3 + i][3 + j]; }
  const V3<T> alin = inverse3(Mlin) * V3<T>(inter[0], inter[1], inter[2]);
  const V3<T> aang = inverse3(Mang) * V3<T>(inter[3], inter[4], inter[5]);
  for (int i = 0; i < 3; ++i) { ab[i] = alin[i]; ab[3 + i] = aang[i]; }
  if (M6out) for (int r = 0; r < 6; ++r) for (int c = 0; c < NV; ++c) M6out[r * NV + c] = M6[r][c];
  if (nle6out) for (int r = 0; r < 6; ++r) nle6out[r] = nle6[r];
}

// computeStateDerivative — humanoid_nmpc/humanoid_wb_mpc/src/dynamics/DynamicsHelperFunctions.cpp:121-134
template <class T>
void flow_map(const Oracle& o, const T* x, const T* u, T* xdot) {
  for (int i = 0; i < NV; ++i) xdot[i] = x[NV + i];
  T ab[6];
  base_acceleration<T>(o, x, u, ab, static_cast<T*>(nullptr), static_cast<T*>(nullptr));
  for (int i = 0; i < 6; ++i) xdot[NV + i] = ab[i];
  for (int j = 0; j < NJ; ++j) xdot[NV + 6 + j] = u[12 + j];
}

// ----------------------------------------------------------------------------- foot kinematics
// PinocchioEndEffectorDynamicsCppAd: position :241-255, orientation error wrt plane :497-517,
// twist :580-599 ([linear; angular], LOCAL_WORLD_ALIGNED), classical accelerations :761-781
// (forwardKinematics(q, v, a) with a = computeGeneralizedAccelerations(state, input)).
template <class T>
struct FootKin {
  V3<T> pos, ori, vlin, vang, alin, aang;
  M3<T> R;
};

// ASSUMPTION A2.  e = -(vec part of the shortest-arc quaternion taking a = R*ez onto n)
//               = (n x a) / sqrt(2 (1 + a.n)).
template <class T>
V3<T> rotation_distance_to_plane(const M3<T>& R, const V3<T>& n) {
  const V3<T> a = R.col(2);
  const T s = sqrt((T(1.0) + dot(a, n)) * T(2.0));
  return cross(n, a) * (T(1.0) / s);
}

template <class T>
void foot_kinematics(const Oracle& o, const T* x, const T* u, FootKin<T> out[2]) {
  T vd[NV];
  base_acceleration<T>(o, x, u, vd, static_cast<T*>(nullptr), static_cast<T*>(nullptr));
  for (int j = 0; j < NJ; ++j) vd[6 + j] = u[12 + j];
  Kin<T> k;
  forward_kinematics<T>(o, x, x + NV, vd, k);
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
    out[f].alin = k.a[b] + cross(k.al[b], r) + cross(k.om[b], cross(k.om[b], r));
    out[f].aang = k.al[b];
  }
}

// FootCollisionConstraint::constraintFunction — humanoid_nmpc/humanoid_common_mpc/src/constraint/FootCollisionConstraint.cpp:92-144
template <class T>
void collision_distances(const Oracle& o, const T* x, T h[16]) {
  Kin<T> k;
  forward_kinematics<T>(o, x, static_cast<const T*>(nullptr), static_cast<const T*>(nullptr), k);
  auto fp = [&](const hsqp_frame& fr) { return k.p[fr.body] + k.R[fr.body] * const_v3<T>(fr.p); };
  const V3<T> ankle_l = fp(o.md.ankle[0]), ankle_r = fp(o.md.ankle[1]);
  const V3<T> f_l = fp(o.md.contact[0]), f_r = fp(o.md.contact[1]);
  const V3<T> l1 = fp(o.md.collision_p1[0]), r1 = fp(o.md.collision_p1[1]);
  const V3<T> l2 = fp(o.md.collision_p2[0]), r2 = fp(o.md.collision_p2[1]);
  const V3<T> k_l = fp(o.md.knee[0]), k_r = fp(o.md.knee[1]);
  const T df(2.0 * o.md.r_foot), dk(2.0 * o.md.r_knee);
  auto dist = [](const V3<T>& a, const V3<T>& b) { const V3<T> d = a - b; return sqrt(dot(d, d)); };
  h[0] = dist(l1, r1) - df;  h[1] = dist(l1, r2) - df;  h[2] = dist(l2, r1) - df;  h[3] = dist(l2, r2) - df;
  h[4] = dist(f_l, r1) - df; h[5] = dist(f_l, r2) - df; h[6] = dist(f_r, l1) - df; h[7] = dist(f_r, l2) - df;
  h[8] = dist(f_l, f_r) - df;
  h[9] = dist(k_l, k_r) - dk;
  h[10] = dist(f_l, ankle_r) - df; h[11] = dist(l1, ankle_r) - df; h[12] = dist(l2, ankle_r) - df;
  h[13] = dist(f_r, ankle_l) - df; h[14] = dist(r1, ankle_l) - df; h[15] = dist(r2, ankle_l) - df;
}

// ContactMomentXYConstraintCppAd::constraintFunction — humanoid_nmpc/humanoid_common_mpc/src/constraint/ContactMomentXYConstraintCppAd.cpp:77-104
template <class T>
void moment_xy(const Oracle& o, const M3<T>& Rfoot, const T* u, int f, T h[4]) {
  const V3<T> lf = tmul(Rfoot, V3<T>(u[6 * f], u[6 * f + 1], u[6 * f + 2]));
  const V3<T> lm = tmul(Rfoot, V3<T>(u[6 * f + 3], u[6 * f + 4], u[6 * f + 5]));
  h[0] = lm.x - lf.z * T(o.md.rect_y_min);
  h[1] = lf.z * T(o.md.rect_y_max) - lm.x;
  h[2] = T(0.0) - lm.y - lf.z * T(o.md.rect_x_min);
  h[3] = lm.y + lf.z * T(o.md.rect_x_max);
}

// ----------------------------------------------------------------------------- penalties
// RelaxedBarrierPenalty (upstream ocs2_core/penalties/penalties/RelaxedBarrierPenalty.cpp)
struct Pen { double p, d1, d2; };
Pen relaxed_barrier(double mu, double delta, double h) {
  if (h > delta) return {-mu * std::log(h), -mu / h, mu / (h * h)};
  const double t = (h - 2.0 * delta) / delta;
  return {mu * (-std::log(delta) + 0.5 * t * t - 0.5), mu * (h - 2.0 * delta) / (delta * delta), mu / (delta * delta)};
}
// ASSUMPTION A1 (fork-only PieceWisePolynomialBarrierPenalty, formula unknown).
Pen pwp_barrier(double mu, double delta, double h) {
  if (h >= delta) return {0.0, 0.0, 0.0};
  const double t = (delta - h) / delta;
  return {mu * t * t * t, -3.0 * mu * t * t / delta, 6.0 * mu * t / (delta * delta)};
}

// ----------------------------------------------------------------------------- node LQ
struct NodeLQ {
  double AB[NX * NZ];   // [A|B]
  double b[NX];         // Phi(x_k,u_k) - x_{k+1}
  double H[NZ * NZ];    // dt * hessian wrt z=[x;u]
  double g[NZ];         // dt * gradient
  double cost;          // dt * l
  int ne;
  double CDe[NE_MAX * (NZ + 1)];  // rows [C|D|e]
  double flow[NX];
};

void seed(const double* x, const double* u, AD* xa, AD* ua) {
  for (int i = 0; i < NX; ++i) xa[i] = AD::seed(x[i], i);
  for (int i = 0; i < NU; ++i) ua[i] = AD::seed(u[i], NX + i);
}

// RK4 sensitivity discretisation (upstream ocs2_core/integration/SensitivityIntegrator RK4, u held constant).
void rk4_sensitivity(const Oracle& o, const double* x, const double* u, double dt, double* AB, double* xnext, double* flow0) {
  AD xa[NX], ua[NU], k1[NX], k2[NX], k3[NX], k4[NX], xs[NX];
  seed(x, u, xa, ua);
  flow_map<AD>(o, xa, ua, k1);
  for (int i = 0; i < NX; ++i) xs[i] = xa[i] + k1[i] * (0.5 * dt);
  flow_map<AD>(o, xs, ua, k2);
  for (int i = 0; i < NX; ++i) xs[i] = xa[i] + k2[i] * (0.5 * dt);
  flow_map<AD>(o, xs, ua, k3);
  for (int i = 0; i < NX; ++i) xs[i] = xa[i] + k3[i] * dt;
  flow_map<AD>(o, xs, ua, k4);
  for (int i = 0; i < NX; ++i) {
    const AD xn = xa[i] + (k1[i] + k2[i] * 2.0 + k3[i] * 2.0 + k4[i]) * (dt / 6.0);
    xnext[i] = xn.v;
    if (AB) for (int c = 0; c < NZ; ++c) AB[i * NZ + c] = xn.d[c];
    if (flow0) flow0[i] = k1[i].v;
  }
}
void rk4_value(const Oracle& o, const double* x, const double* u, double dt, double* xnext) {
  double k1[NX], k2[NX], k3[NX], k4[NX], xs[NX];
  flow_map<double>(o, x, u, k1);
  for (int i = 0; i < NX; ++i) xs[i] = x[i] + 0.5 * dt * k1[i];
  flow_map<double>(o, xs, u, k2);
  for (int i = 0; i < NX; ++i) xs[i] = x[i] + 0.5 * dt * k2[i];
  flow_map<double>(o, xs, u, k3);
  for (int i = 0; i < NX; ++i) xs[i] = x[i] + dt * k3[i];
  flow_map<double>(o, xs, u, k4);
  for (int i = 0; i < NX; ++i) xnext[i] = x[i] + dt / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
}

// x_nom, u_nom of StateInputQuadraticCost::getStateInputDeviation —
//   humanoid_nmpc/humanoid_common_mpc/src/cost/StateInputQuadraticCost.cpp:67-78,
//   SwitchedModelReferenceManager::getDesiredState (…/reference_manager/SwitchedModelReferenceManager.cpp:110-135; the arm-swing
//   offsets read the CURRENT state's yaw but ocs2 treats x_nom as constant when differentiating),
//   weightCompensatingInput (…/pinocchio_model/DynamicsHelperFunctions.h:178-193).
void nominal(const Oracle& o, const double* x, const double* par, double* xnom, double* unom) {
  for (int i = 0; i < NX; ++i) xnom[i] = par[HSQP_P_XDES + i];
  const double yaw = x[3];
  const double vloc = std::cos(yaw) * xnom[NV + 0] + std::sin(yaw) * xnom[NV + 1];
  const double gcf = par[HSQP_P_ARMSWING] * vloc;
  xnom[6 + o.md.arm_swing_joint[0]] += -0.15 * gcf;
  xnom[6 + o.md.arm_swing_joint[1]] += 0.15 * gcf;
  xnom[6 + o.md.arm_swing_joint[2]] += -0.15 * gcf;
  xnom[6 + o.md.arm_swing_joint[3]] += 0.15 * gcf;
  for (int i = 0; i < NU; ++i) unom[i] = 0.0;
  const bool c0 = par[HSQP_P_CONTACT] > 0.5, c1 = par[HSQP_P_CONTACT + 1] > 0.5;
  const int ns = int(c0) + int(c1);
  if (ns > 0) {
    const double fz = o.total_mass * 9.81 / ns;  // the reference hard-codes 9.81 here
    if (c0) unom[2] = fz;
    if (c1) unom[8] = fz;
  }
}

// Stage cost + equality constraints at (x,u): value only (lq = nullptr) or with the
// Gauss-Newton / penalty quadratic model.  Returns l (unscaled); fills lq->H,g (unscaled) and C,D,e.
// Terms and their order follow WBMpcInterface::setupOptimalControlProblem (humanoid_nmpc/humanoid_wb_mpc/src/WBMpcInterface.cpp:131-199).
// Friction cone h(F) = mu (Fz + gripperForce) - sqrt(Fx^2 + Fy^2 + regularization) of one contact force F (world = contact frame,
// t_R_w = I) with its first and second derivatives: FrictionForceConeConstraint.cpp:141-185 (coneConstraint, computeConeLocalDerivatives).
static void friction_cone(const hsqp_model_desc& md, const double* F, double& h, double dh[3], double d2[3][3]) {
  const double Fx = F[0], Fy = F[1], Fz = F[2];
  const double T2 = Fx * Fx + Fy * Fy + md.friction_reg, Tn = std::sqrt(T2), T32 = Tn * T2;
  h = md.friction_mu * (Fz + md.friction_grip) - Tn;
  dh[0] = -Fx / Tn; dh[1] = -Fy / Tn; dh[2] = md.friction_mu;
  const double d[3][3] = {{-(Fy * Fy + md.friction_reg) / T32, Fx * Fy / T32, 0.0}, {Fx * Fy / T32, -(Fx * Fx + md.friction_reg) / T32, 0.0}, {0.0, 0.0, 0.0}};
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) d2[a][b] = d[a][b];
}

double stage_terms(const Oracle& o, const double* x, const double* u, const double* par, NodeLQ* lq, double* eq_out, int* ne_out) {
  const hsqp_model_desc& md = o.md;
  const bool contact[2] = {par[HSQP_P_CONTACT] > 0.5, par[HSQP_P_CONTACT + 1] > 0.5};
  double cost = 0.0;
  double* H = lq ? lq->H : nullptr;
  double* g = lq ? lq->g : nullptr;
  if (lq) { std::fill(H, H + NZ * NZ, 0.0); std::fill(g, g + NZ, 0.0); }

  // (1) StateInputQuadraticCost: 0.5 dx'Q dx + 0.5 du'R du
  double xnom[NX], unom[NU];
  nominal(o, x, par, xnom, unom);
  for (int i = 0; i < NX; ++i) {
    const double d = x[i] - xnom[i];
    cost += 0.5 * md.Q[i] * d * d;
    if (lq) { g[i] += md.Q[i] * d; H[i * NZ + i] += md.Q[i]; }
  }
  for (int i = 0; i < NU; ++i) {
    const double d = u[i] - unom[i];
    cost += 0.5 * md.R[i] * d * d;
    if (lq) { g[NX + i] += md.R[i] * d; H[(NX + i) * NZ + NX + i] += md.R[i]; }
  }

  // foot kinematics, value and (if requested) Jacobians by forward AD
  FootKin<double> fk[2];
  std::vector<FootKin<AD>> fka;
  if (lq) {
    std::vector<AD> xa(NX), ua(NU);
    seed(x, u, xa.data(), ua.data());
    fka.resize(2);
    foot_kinematics<AD>(o, xa.data(), ua.data(), fka.data());
    for (int f = 0; f < 2; ++f) {
      for (int i = 0; i < 3; ++i) {
        fk[f].pos[i] = fka[f].pos[i].v; fk[f].ori[i] = fka[f].ori[i].v; fk[f].vlin[i] = fka[f].vlin[i].v;
        fk[f].vang[i] = fka[f].vang[i].v; fk[f].alin[i] = fka[f].alin[i].v; fk[f].aang[i] = fka[f].aang[i].v;
        for (int j = 0; j < 3; ++j) fk[f].R.m[i][j] = fka[f].R.m[i][j].v;
      }
    }
  } else {
    foot_kinematics<double>(o, x, u, fk);
  }

  // (2) per-foot terms, in the order they are added (WBMpcInterface.cpp:164-188)
  int ne = 0;
  double eq[NE_MAX];
  double* CDe = lq ? lq->CDe : nullptr;
  if (lq) std::fill(CDe, CDe + NE_MAX * (NZ + 1), 0.0);
  auto add_gn_row = [&](double r, const AD* ra, double scale) {  // 0.5*(scale*r)^2 with Jacobian row scale*ra->d
    cost += 0.5 * scale * scale * r * r;
    if (lq && scale != 0.0) {
      for (int a = 0; a < NZ; ++a) {
        const double ja = scale * ra->d[a];
        if (ja == 0.0) continue;
        g[a] += ja * scale * r;
        for (int b = 0; b < NZ; ++b) H[a * NZ + b] += ja * scale * ra->d[b];
      }
    }
  };
  auto add_penalty_row = [&](const Pen& p, const double* jac /*NZ or null*/) {
    cost += p.p;
    if (lq && jac) {
      for (int a = 0; a < NZ; ++a) {
        if (jac[a] == 0.0) continue;
        g[a] += p.d1 * jac[a];
        for (int b = 0; b < NZ; ++b) H[a * NZ + b] += p.d2 * jac[a] * jac[b];
      }
    }
  };
  for (int f = 0; f < 2; ++f) {
    // --- friction cone soft constraint (active in contact): FrictionForceConeConstraint.cpp:78-224, relaxed barrier
    if (contact[f]) {
      double h, dh[3], d2[3][3];
      friction_cone(md, u + 6 * f, h, dh, d2);     // (pinned against the reference-compiled constraint: tests/test_ref_terms.py)
      const Pen p = relaxed_barrier(md.friction_barrier.mu, md.friction_barrier.delta, h);
      cost += p.p;
      if (lq) {
        const int o0 = NX + 6 * f;
        for (int a = 0; a < 3; ++a) {
          g[o0 + a] += p.d1 * dh[a];
          for (int b = 0; b < 3; ++b) H[(o0 + a) * NZ + o0 + b] += p.d2 * dh[a] * dh[b] + p.d1 * d2[a][b];
        }
        // ConstraintOrder::Quadratic: dfdxx = dfduu -= hessianDiagonalShift on the whole diagonal (:213-224)
        for (int a = 0; a < NZ; ++a) H[a * NZ + a] += p.d1 * (-md.friction_hess_shift);
      }
    }
    // --- contact moment XY soft constraint (active in contact), relaxed barrier
    if (contact[f]) {
      double h[4];
      double jac[4][NZ];
      if (lq) {
        AD ha[4];
        std::vector<AD> ua(NU);
        for (int i = 0; i < NU; ++i) ua[i] = AD::seed(u[i], NX + i);
        moment_xy<AD>(o, fka[f].R, ua.data(), f, ha);
        for (int r = 0; r < 4; ++r) { h[r] = ha[r].v; for (int a = 0; a < NZ; ++a) jac[r][a] = ha[r].d[a]; }
      } else {
        moment_xy<double>(o, fk[f].R, u, f, h);
      }
      for (int r = 0; r < 4; ++r) add_penalty_row(relaxed_barrier(md.moment_barrier.mu, md.moment_barrier.delta, h[r]), lq ? jac[r] : nullptr);
    }
    // --- equality: zero wrench (swing) — ZeroWrenchConstraint.cpp:59-84
    if (!contact[f]) {
      for (int r = 0; r < 6; ++r) {
        eq[ne] = u[6 * f + r];
        if (lq) { CDe[ne * (NZ + 1) + NX + 6 * f + r] = 1.0; CDe[ne * (NZ + 1) + NZ] = eq[ne]; }
        ++ne;
      }
    }
    // --- equality: stance foot zero acceleration (contact) —
    //     EndEffectorDynamicsAccelerationsConstraint.cpp:109-146, gains WBMpcInterface.cpp:205-229
    if (contact[f]) {
      const double Ax[6] = {0.0, 0.0, md.gain_pos_z, md.gain_ori, md.gain_ori, md.gain_ori};
      const double Av[6] = {md.gain_linvel_xy, md.gain_linvel_xy, md.gain_linvel_z, md.gain_angvel, md.gain_angvel, md.gain_angvel};
      const double Aa[6] = {md.gain_linacc_xy, md.gain_linacc_xy, md.gain_linacc_z, md.gain_angacc, md.gain_angacc, md.gain_angacc};
      for (int r = 0; r < 6; ++r) {
        const int c = r % 3;
        const double pose = r < 3 ? fk[f].pos[c] : fk[f].ori[c];
        const double tw = r < 3 ? fk[f].vlin[c] : fk[f].vang[c];
        const double ac = r < 3 ? fk[f].alin[c] : fk[f].aang[c];
        eq[ne] = Ax[r] * pose + Av[r] * tw + Aa[r] * ac;
        if (lq) {
          const AD& posea = r < 3 ? fka[f].pos[c] : fka[f].ori[c];
          const AD& twa = r < 3 ? fka[f].vlin[c] : fka[f].vang[c];
          const AD& aca = r < 3 ? fka[f].alin[c] : fka[f].aang[c];
          for (int a = 0; a < NZ; ++a) {
            // pose terms are state-only in the reference (getPositionLinearApproximation(state)); d/du of them is 0 anyway
            CDe[ne * (NZ + 1) + a] = Ax[r] * posea.d[a] + Av[r] * twa.d[a] + Aa[r] * aca.d[a];
          }
          CDe[ne * (NZ + 1) + NZ] = eq[ne];
        }
        ++ne;
      }
    }
    // --- equality: swing foot vertical acceleration tracking (swing) —
    //     EndEffectorDynamicsLinearAccConstraint.cpp:97-127, config WBMpcPreComputation.cpp:91-104
    if (!contact[f]) {
      const double zs = par[HSQP_P_SWING + 3 * f], zds = par[HSQP_P_SWING + 3 * f + 1], zdds = par[HSQP_P_SWING + 3 * f + 2];
      const double bb = -md.gain_linvel_z * zds - md.gain_linacc_z * zdds - md.gain_pos_z * zs;
      eq[ne] = bb + md.gain_pos_z * fk[f].pos.z + md.gain_linvel_z * fk[f].vlin.z + md.gain_linacc_z * fk[f].alin.z;
      if (lq) {
        for (int a = 0; a < NZ; ++a)
          CDe[ne * (NZ + 1) + a] = md.gain_pos_z * fka[f].pos.z.d[a] + md.gain_linvel_z * fka[f].vlin.z.d[a] + md.gain_linacc_z * fka[f].alin.z.d[a];
        CDe[ne * (NZ + 1) + NZ] = eq[ne];
      }
      ++ne;
    }
    // --- foot task-space Gauss-Newton cost (always active) — EndEffectorDynamicsFootCost.cpp:91-152; r = [0; ori; vlin; vang; alin; aang] .* sqrtW * impactProximity
    {
      const double ip = par[HSQP_P_IMPACT + f];
      const V3<double>* vals[5] = {&fk[f].ori, &fk[f].vlin, &fk[f].vang, &fk[f].alin, &fk[f].aang};
      for (int blk = 0; blk < 5; ++blk) {
        for (int c = 0; c < 3; ++c) {
          const double sw = md.foot_sqrt_w[3 + 3 * blk + c] * ip;
          const AD* ra = nullptr;
          if (lq) {
            const V3<AD>* va[5] = {&fka[f].ori, &fka[f].vlin, &fka[f].vang, &fka[f].alin, &fka[f].aang};
            ra = &(*va[blk])[c];
          }
          add_gn_row((*vals[blk])[c], ra, sw);
        }
      }
    }
  }

  // (3) state soft constraints: joint limits (always) — JointLimitsSoftConstraint.cpp:64-100
  for (int j = 0; j < NJ; ++j) {
    const double qj = x[6 + j];
    const Pen lo = pwp_barrier(md.joint_limit_barrier.mu, md.joint_limit_barrier.delta, qj - md.bodies[1 + j].q_lo);
    const Pen hi = pwp_barrier(md.joint_limit_barrier.mu, md.joint_limit_barrier.delta, md.bodies[1 + j].q_hi - qj);
    cost += lo.p + hi.p;
    if (lq) { g[6 + j] += lo.d1 - hi.d1; H[(6 + j) * NZ + 6 + j] += lo.d2 + hi.d2; }
  }
  // foot collision (inactive in double support) — FootCollisionConstraint.cpp:80-144
  if (!(contact[0] && contact[1])) {
    double h[16];
    std::vector<double> jac;
    if (lq) {
      std::vector<AD> xa(NX);
      for (int i = 0; i < NX; ++i) xa[i] = AD::seed(x[i], i);
      AD ha[16];
      collision_distances<AD>(o, xa.data(), ha);
      jac.assign(16 * NZ, 0.0);
      for (int r = 0; r < 16; ++r) { h[r] = ha[r].v; for (int a = 0; a < NX; ++a) jac[r * NZ + a] = ha[r].d[a]; }
    } else {
      collision_distances<double>(o, x, h);
    }
    for (int r = 0; r < 16; ++r) add_penalty_row(pwp_barrier(md.collision_barrier.mu, md.collision_barrier.delta, h[r]), lq ? &jac[r * NZ] : nullptr);
  }
  if (lq) lq->ne = ne;
  if (ne_out) *ne_out = ne;
  if (eq_out) for (int i = 0; i < ne; ++i) eq_out[i] = eq[i];
  return cost;
}

// Intermediate node: multiple_shooting::setupIntermediateNode semantics: cost and soft constraints scaled by dt.
// Event interval (dt = 0): the identity jump map x+ = x between the pre- and the post-event node, no cost, no
// constraints.  The inputs of the pre-event node do not enter anything; a unit Hessian pins their step to zero.
void jump_node_lq(const double* x, const double* xnext, NodeLQ& lq) {
  std::fill(lq.AB, lq.AB + NX * NZ, 0.0);
  std::fill(lq.H, lq.H + NZ * NZ, 0.0);
  std::fill(lq.g, lq.g + NZ, 0.0);
  std::fill(lq.CDe, lq.CDe + NE_MAX * (NZ + 1), 0.0);
  std::fill(lq.flow, lq.flow + NX, 0.0);
  for (int i = 0; i < NX; ++i) { lq.AB[i * NZ + i] = 1.0; lq.b[i] = x[i] - xnext[i]; }
  for (int i = NX; i < NZ; ++i) lq.H[i * NZ + i] = 1.0;
  lq.cost = 0.0;
  lq.ne = 0;
}

void node_lq(const Oracle& o, const double* x, const double* u, const double* xnext, const double* par, double dt, NodeLQ& lq) {
  if (dt == 0.0) { jump_node_lq(x, xnext, lq); return; }
  double phi[NX];
  rk4_sensitivity(o, x, u, dt, lq.AB, phi, lq.flow);
  for (int i = 0; i < NX; ++i) lq.b[i] = phi[i] - xnext[i];
  const double l = stage_terms(o, x, u, par, &lq, nullptr, nullptr);
  lq.cost = dt * l;
  for (int i = 0; i < NZ * NZ; ++i) lq.H[i] *= dt;
  for (int i = 0; i < NZ; ++i) lq.g[i] *= dt;
}

// Terminal node: QuadraticStateCost(Q_final*scaling) on x - x_des(t_N) — HumanoidCostConstraintFactory.cpp:218-228
double terminal_cost(const Oracle& o, const double* x, const double* par, double* Hd /*[NX] diag*/, double* g /*[NX]*/) {
  double c = 0.0;
  for (int i = 0; i < NX; ++i) {
    const double d = x[i] - par[HSQP_P_XDES + i];
    c += 0.5 * o.md.Qf[i] * d * d;
    if (Hd) Hd[i] = o.md.Qf[i];
    if (g) g[i] = o.md.Qf[i] * d;
  }
  return c;
}

// ----------------------------------------------------------------------------- projection
// Householder QR of D^T (nu x ne): D^T = [Q1 Q2][R1;0].  Px = -Q1 R1^-T C, Pe = -Q1 R1^-T e, Pu = Q2.
struct Projected {
  int nut;                          // nu - ne
  std::vector<double> Px, Pu, Pe;   // NU x NX, NU x nut, NU
  std::vector<double> At, Bt, bt;   // NX x NX, NX x nut, NX
  std::vector<double> Qt, Pt, Rt, qt, rt;  // NX x NX, nut x NX, nut x nut, NX, nut
};

bool project_node(const NodeLQ& lq, Projected& pr) {
  const int ne = lq.ne, nut = NU - ne;
  pr.nut = nut;
  // Dt = D^T (NU x ne)
  std::vector<double> Rm(NU * std::max(ne, 1), 0.0), Qm(NU * NU, 0.0);
  for (int r = 0; r < ne; ++r) for (int c = 0; c < NU; ++c) Rm[c * ne + r] = lq.CDe[r * (NZ + 1) + NX + c];
  for (int i = 0; i < NU; ++i) Qm[i * NU + i] = 1.0;
  for (int k = 0; k < ne; ++k) {
    double nrm = 0.0;
    for (int i = k; i < NU; ++i) nrm += Rm[i * ne + k] * Rm[i * ne + k];
    nrm = std::sqrt(nrm);
    if (nrm < 1e-12) return false;
    std::vector<double> v(NU, 0.0);
    const double alpha = Rm[k * ne + k] >= 0 ? -nrm : nrm;
    for (int i = k; i < NU; ++i) v[i] = Rm[i * ne + k];
    v[k] -= alpha;
    double vn = 0.0;
    for (int i = k; i < NU; ++i) vn += v[i] * v[i];
    if (vn < 1e-300) continue;
    for (int c = 0; c < ne; ++c) {  // R <- (I - 2 v v^T / vn) R
      double s = 0.0;
      for (int i = k; i < NU; ++i) s += v[i] * Rm[i * ne + c];
      s *= 2.0 / vn;
      for (int i = k; i < NU; ++i) Rm[i * ne + c] -= s * v[i];
    }
    for (int r = 0; r < NU; ++r) {  // Q <- Q (I - 2 v v^T / vn)
      double s = 0.0;
      for (int i = k; i < NU; ++i) s += Qm[r * NU + i] * v[i];
      s *= 2.0 / vn;
      for (int i = k; i < NU; ++i) Qm[r * NU + i] -= s * v[i];
    }
  }
  // W = R1^-T [C | e]  (ne x (NX+1)); R1^T is lower triangular
  std::vector<double> W(std::max(ne, 1) * (NX + 1), 0.0);
  for (int c = 0; c <= NX; ++c) {
    for (int i = 0; i < ne; ++i) {
      double s = (c < NX) ? lq.CDe[i * (NZ + 1) + c] : lq.CDe[i * (NZ + 1) + NZ];
      for (int j = 0; j < i; ++j) s -= Rm[j * ne + i] * W[j * (NX + 1) + c];
      W[i * (NX + 1) + c] = s / Rm[i * ne + i];
    }
  }
  pr.Px.assign(NU * NX, 0.0); pr.Pe.assign(NU, 0.0); pr.Pu.assign(NU * std::max(nut, 1), 0.0);
  for (int r = 0; r < NU; ++r) {
    for (int c = 0; c < NX; ++c) {
      double s = 0.0;
      for (int j = 0; j < ne; ++j) s += Qm[r * NU + j] * W[j * (NX + 1) + c];
      pr.Px[r * NX + c] = -s;
    }
    double s = 0.0;
    for (int j = 0; j < ne; ++j) s += Qm[r * NU + j] * W[j * (NX + 1) + NX];
    pr.Pe[r] = -s;
    for (int c = 0; c < nut; ++c) pr.Pu[r * nut + c] = Qm[r * NU + ne + c];
  }
  // change of variables
  const double* AB = lq.AB;
  auto A = [&](int i, int j) { return AB[i * NZ + j]; };
  auto B = [&](int i, int j) { return AB[i * NZ + NX + j]; };
  auto Hxx = [&](int i, int j) { return lq.H[i * NZ + j]; };
  auto Hux = [&](int i, int j) { return lq.H[(NX + i) * NZ + j]; };
  auto Huu = [&](int i, int j) { return lq.H[(NX + i) * NZ + NX + j]; };
  pr.At.assign(NX * NX, 0.0); pr.Bt.assign(NX * std::max(nut, 1), 0.0); pr.bt.assign(NX, 0.0);
  for (int i = 0; i < NX; ++i) {
    for (int j = 0; j < NX; ++j) { double s = A(i, j); for (int k = 0; k < NU; ++k) s += B(i, k) * pr.Px[k * NX + j]; pr.At[i * NX + j] = s; }
    for (int j = 0; j < nut; ++j) { double s = 0.0; for (int k = 0; k < NU; ++k) s += B(i, k) * pr.Pu[k * nut + j]; pr.Bt[i * nut + j] = s; }
    double s = lq.b[i]; for (int k = 0; k < NU; ++k) s += B(i, k) * pr.Pe[k]; pr.bt[i] = s;
  }
  // T1 = Hux + Huu Px (NU x NX); t1 = gu + Huu Pe (NU)
  std::vector<double> T1(NU * NX), t1(NU);
  for (int i = 0; i < NU; ++i) {
    for (int j = 0; j < NX; ++j) { double s = Hux(i, j); for (int k = 0; k < NU; ++k) s += Huu(i, k) * pr.Px[k * NX + j]; T1[i * NX + j] = s; }
    double s = lq.g[NX + i]; for (int k = 0; k < NU; ++k) s += Huu(i, k) * pr.Pe[k]; t1[i] = s;
  }
  pr.Qt.assign(NX * NX, 0.0); pr.Pt.assign(std::max(nut, 1) * NX, 0.0); pr.Rt.assign(std::max(nut * nut, 1), 0.0);
  pr.qt.assign(NX, 0.0); pr.rt.assign(std::max(nut, 1), 0.0);
  for (int i = 0; i < NX; ++i) {
    for (int j = 0; j < NX; ++j) {
      double s = Hxx(i, j);
      for (int k = 0; k < NU; ++k) s += Hux(k, i) * pr.Px[k * NX + j] + pr.Px[k * NX + i] * T1[k * NX + j];
      pr.Qt[i * NX + j] = s;
    }
    double s = lq.g[i];
    for (int k = 0; k < NU; ++k) s += pr.Px[k * NX + i] * t1[k] + Hux(k, i) * pr.Pe[k];
    pr.qt[i] = s;
  }
  for (int i = 0; i < nut; ++i) {
    for (int j = 0; j < NX; ++j) { double s = 0.0; for (int k = 0; k < NU; ++k) s += pr.Pu[k * nut + i] * T1[k * NX + j]; pr.Pt[i * NX + j] = s; }
    double s = 0.0; for (int k = 0; k < NU; ++k) s += pr.Pu[k * nut + i] * t1[k]; pr.rt[i] = s;
    for (int j = 0; j < nut; ++j) {
      double r = 0.0;
      for (int k = 0; k < NU; ++k) { double hk = 0.0; for (int l = 0; l < NU; ++l) hk += Huu(k, l) * pr.Pu[l * nut + j]; r += pr.Pu[k * nut + i] * hk; }
      pr.Rt[i * nut + j] = r;
    }
  }
  return true;
}

// ----------------------------------------------------------------------------- Riccati
bool cholesky(std::vector<double>& L, int n) {  // in place, lower
  for (int j = 0; j < n; ++j) {
    double d = L[j * n + j];
    for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
    if (d <= 0.0) return false;
    d = std::sqrt(d);
    L[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = L[i * n + j];
      for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = s / d;
    }
  }
  return true;
}
void chol_solve(const std::vector<double>& L, int n, double* rhs, int nrhs, int ld) {  // rhs (n x nrhs) <- Lambda^-1 rhs
  for (int c = 0; c < nrhs; ++c) {
    for (int i = 0; i < n; ++i) { double s = rhs[i * ld + c]; for (int k = 0; k < i; ++k) s -= L[i * n + k] * rhs[k * ld + c]; rhs[i * ld + c] = s / L[i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double s = rhs[i * ld + c]; for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * rhs[k * ld + c]; rhs[i * ld + c] = s / L[i * n + i]; }
  }
}

struct RiccatiOut {
  std::vector<double> dx;   // (N+1) x NX
  std::vector<double> ut;   // N x NU (first nut entries used)
  std::vector<double> lam;  // (N+1) x NX costates
};

bool riccati(const std::vector<Projected>& st, const double* HN /*NX diag*/, const double* gN, const double* dx0, int N, RiccatiOut& out) {
  std::vector<double> S(NX * NX, 0.0), s(NX);
  for (int i = 0; i < NX; ++i) { S[i * NX + i] = HN[i]; s[i] = gN[i]; }
  std::vector<std::vector<double>> Ks(N), ks(N), Ss(N + 1), ss(N + 1);
  Ss[N] = S; ss[N] = s;
  for (int k = N - 1; k >= 0; --k) {
    const Projected& p = st[k];
    const int m = p.nut;
    // SA = S A~ (NX x NX), SB = S B~ (NX x m), sb = s + S b~
    std::vector<double> SA(NX * NX), SB(NX * std::max(m, 1)), sb(NX);
    for (int i = 0; i < NX; ++i) {
      for (int j = 0; j < NX; ++j) { double a = 0.0; for (int l = 0; l < NX; ++l) a += S[i * NX + l] * p.At[l * NX + j]; SA[i * NX + j] = a; }
      for (int j = 0; j < m; ++j) { double a = 0.0; for (int l = 0; l < NX; ++l) a += S[i * NX + l] * p.Bt[l * m + j]; SB[i * m + j] = a; }
      double a = s[i]; for (int l = 0; l < NX; ++l) a += S[i * NX + l] * p.bt[l]; sb[i] = a;
    }
    std::vector<double> Lam(std::max(m * m, 1)), G(std::max(m, 1) * NX), gv(std::max(m, 1));
    for (int i = 0; i < m; ++i) {
      for (int j = 0; j < m; ++j) { double a = p.Rt[i * m + j]; for (int l = 0; l < NX; ++l) a += p.Bt[l * m + i] * SB[l * m + j]; Lam[i * m + j] = a; }
      for (int j = 0; j < NX; ++j) { double a = p.Pt[i * NX + j]; for (int l = 0; l < NX; ++l) a += p.Bt[l * m + i] * SA[l * NX + j]; G[i * NX + j] = a; }
      double a = p.rt[i]; for (int l = 0; l < NX; ++l) a += p.Bt[l * m + i] * sb[l]; gv[i] = a;
    }
    for (int i = 0; i < m; ++i) for (int j = i + 1; j < m; ++j) { const double a = 0.5 * (Lam[i * m + j] + Lam[j * m + i]); Lam[i * m + j] = Lam[j * m + i] = a; }
    std::vector<double> L = Lam;
    if (m > 0 && !cholesky(L, m)) return false;
    std::vector<double> K = G, kv = gv;  // K = -Lam^-1 G, k = -Lam^-1 g
    if (m > 0) { chol_solve(L, m, K.data(), NX, NX); chol_solve(L, m, kv.data(), 1, 1); }
    for (auto& e : K) e = -e;
    for (auto& e : kv) e = -e;
    // S = Q~ + A~^T SA + G^T K ; s = q~ + A~^T sb + G^T k
    std::vector<double> Sn(NX * NX), sn(NX);
    for (int i = 0; i < NX; ++i) {
      for (int j = 0; j < NX; ++j) {
        double a = p.Qt[i * NX + j];
        for (int l = 0; l < NX; ++l) a += p.At[l * NX + i] * SA[l * NX + j];
        for (int l = 0; l < m; ++l) a += G[l * NX + i] * K[l * NX + j];
        Sn[i * NX + j] = a;
      }
      double a = p.qt[i];
      for (int l = 0; l < NX; ++l) a += p.At[l * NX + i] * sb[l];
      for (int l = 0; l < m; ++l) a += G[l * NX + i] * kv[l];
      sn[i] = a;
    }
    for (int i = 0; i < NX; ++i) for (int j = i + 1; j < NX; ++j) { const double a = 0.5 * (Sn[i * NX + j] + Sn[j * NX + i]); Sn[i * NX + j] = Sn[j * NX + i] = a; }
    S = Sn; s = sn;
    Ks[k] = K; ks[k] = kv; Ss[k] = S; ss[k] = s;
  }
  out.dx.assign((N + 1) * NX, 0.0); out.ut.assign(N * NU, 0.0); out.lam.assign((N + 1) * NX, 0.0);
  for (int i = 0; i < NX; ++i) out.dx[i] = dx0[i];
  for (int k = 0; k < N; ++k) {
    const Projected& p = st[k];
    const int m = p.nut;
    const double* dx = &out.dx[k * NX];
    double* ut = &out.ut[k * NU];
    for (int i = 0; i < m; ++i) { double a = ks[k][i]; for (int j = 0; j < NX; ++j) a += Ks[k][i * NX + j] * dx[j]; ut[i] = a; }
    double* dxn = &out.dx[(k + 1) * NX];
    for (int i = 0; i < NX; ++i) {
      double a = p.bt[i];
      for (int j = 0; j < NX; ++j) a += p.At[i * NX + j] * dx[j];
      for (int j = 0; j < m; ++j) a += p.Bt[i * m + j] * ut[j];
      dxn[i] = a;
    }
  }
  for (int k = 0; k <= N; ++k)
    for (int i = 0; i < NX; ++i) { double a = ss[k][i]; for (int j = 0; j < NX; ++j) a += Ss[k][i * NX + j] * out.dx[k * NX + j]; out.lam[k * NX + i] = a; }
  return true;
}

void kkt_residual(const std::vector<Projected>& st, const double* HN, const double* gN, const double* dx0, int N, const RiccatiOut& r, double* stat, double* prim) {
  double rs = 0.0, rp = 0.0;
  for (int i = 0; i < NX; ++i) rp = std::max(rp, std::fabs(r.dx[i] - dx0[i]));
  for (int k = 0; k < N; ++k) {
    const Projected& p = st[k];
    const int m = p.nut;
    const double* dx = &r.dx[k * NX]; const double* dxn = &r.dx[(k + 1) * NX]; const double* ut = &r.ut[k * NU];
    const double* lam = &r.lam[k * NX]; const double* lamn = &r.lam[(k + 1) * NX];
    for (int i = 0; i < NX; ++i) {
      double a = p.qt[i] - lam[i];
      for (int j = 0; j < NX; ++j) a += p.Qt[i * NX + j] * dx[j] + p.At[j * NX + i] * lamn[j];
      for (int j = 0; j < m; ++j) a += p.Pt[j * NX + i] * ut[j];
      rs = std::max(rs, std::fabs(a));
      double d = dxn[i] - p.bt[i];
      for (int j = 0; j < NX; ++j) d -= p.At[i * NX + j] * dx[j];
      for (int j = 0; j < m; ++j) d -= p.Bt[i * m + j] * ut[j];
      rp = std::max(rp, std::fabs(d));
    }
    for (int i = 0; i < m; ++i) {
      double a = p.rt[i];
      for (int j = 0; j < NX; ++j) a += p.Pt[i * NX + j] * dx[j] + p.Bt[j * m + i] * lamn[j];
      for (int j = 0; j < m; ++j) a += p.Rt[i * m + j] * ut[j];
      rs = std::max(rs, std::fabs(a));
    }
  }
  for (int i = 0; i < NX; ++i) rs = std::max(rs, std::fabs(HN[i] * r.dx[N * NX + i] + gN[i] - r.lam[N * NX + i]));
  *stat = rs; *prim = rp;
}

// PerformanceIndex of a trajectory (value-only pass): ASSUMPTION A3 for the dt scaling of the SSE terms.
void performance(const Oracle& o, int N, double dt, const double* x, const double* u, const double* par, int threads, hsqp_perf* out) {
  double cost = 0.0, dyn = 0.0, eqs = 0.0;
#pragma omp parallel for num_threads(threads) reduction(+ : cost, dyn, eqs) schedule(dynamic)
  for (int k = 0; k < N; ++k) {
    double phi[NX], eq[NE_MAX];
    int ne = 0;
    const double dtk = dt_at(o, k, dt);
    if (dtk == 0.0) {   // event interval: jump defect, unscaled (upstream computeEventPerformance)
      for (int i = 0; i < NX; ++i) { const double d = x[k * NX + i] - x[(k + 1) * NX + i]; dyn += d * d; }
      continue;
    }
    rk4_value(o, x + k * NX, u + k * NU, dtk, phi);
    for (int i = 0; i < NX; ++i) { const double d = phi[i] - x[(k + 1) * NX + i]; dyn += dtk * d * d; }
    cost += dtk * stage_terms(o, x + k * NX, u + k * NU, par + k * NP, nullptr, eq, &ne);
    for (int i = 0; i < ne; ++i) eqs += dtk * eq[i] * eq[i];
  }
  cost += terminal_cost(o, x + N * NX, par + N * NP, nullptr, nullptr);
  out->cost = cost; out->dynamics_sse = dyn; out->equality_sse = eqs; out->merit = cost;
}

}  // namespace

// ================================================================================= C API (ctypes)
extern "C" {

void* orc_create(const hsqp_model_desc* md) {
  Oracle* o = new Oracle;
  init_oracle(*o, md);
  return o;
}
void orc_destroy(void* h) { delete static_cast<Oracle*>(h); }
// non-uniform grid / event intervals of the next calls (N interval lengths, 0 = event); dts == null: back to the uniform dt argument
void orc_set_grid(void* h, int N, const double* dts) {
  Oracle& o = *static_cast<Oracle*>(h);
  if (dts) o.grid.assign(dts, dts + N); else o.grid.clear();
}
double orc_total_mass(void* h) { return static_cast<Oracle*>(h)->total_mass; }

void orc_flow_map(void* h, const double* x, const double* u, double* xdot) { flow_map<double>(*static_cast<Oracle*>(h), x, u, xdot); }

// xdot and its Jacobian wrt [x;u] (58 x 93)
void orc_flow_map_jac(void* h, const double* x, const double* u, double* xdot, double* J) {
  const Oracle& o = *static_cast<Oracle*>(h);
  AD xa[NX], ua[NU], f[NX];
  seed(x, u, xa, ua);
  flow_map<AD>(o, xa, ua, f);
  for (int i = 0; i < NX; ++i) { xdot[i] = f[i].v; for (int c = 0; c < NZ; ++c) J[i * NZ + c] = f[i].d[c]; }
}

// rows 0..5 of the joint-space inertia matrix and of nle (for the known-answer tests)
void orc_base_dynamics(void* h, const double* x, const double* u, double* ab, double* M6, double* nle6) {
  base_acceleration<double>(*static_cast<Oracle*>(h), x, u, ab, M6, nle6);
}

// Full joint-space inertia matrix M (29x29) and nle (29) by projected Newton-Euler — used only to
// cross-check the 6-row shortcut against a "complete" CRBA/RNEA restatement.
void orc_full_dynamics(void* h, const double* x, double* M, double* nle) {
  const Oracle& o = *static_cast<Oracle*>(h);
  Kin<double> k;
  forward_kinematics<double>(o, x, x + NV, static_cast<const double*>(nullptr), k);
  std::fill(M, M + NV * NV, 0.0);
  std::fill(nle, nle + NV, 0.0);
  const V3<double> grav(0.0, 0.0, o.md.gravity);
  for (int i = 0; i < NB; ++i) {
    const hsqp_body& b = o.md.bodies[i];
    const V3<double> rc = k.R[i] * const_v3<double>(b.com), com = k.p[i] + rc;
    const M3<double> Iw = k.R[i] * const_m3<double>(b.inertia) * transpose(k.R[i]);
    const V3<double> acom = k.a[i] + cross(k.al[i], rc) + cross(k.om[i], cross(k.om[i], rc));
    const V3<double> f = (acom + grav) * b.mass, n = Iw * k.al[i] + cross(k.om[i], Iw * k.om[i]);
    V3<double> l[NV], a[NV];
    bool ok[NV];
    for (int c = 0; c < NV; ++c) ok[c] = jacobian_column(o, k, i, c, com, l[c], a[c]);
    for (int r = 0; r < NV; ++r) {
      if (!ok[r]) continue;
      nle[r] += dot(l[r], f) + dot(a[r], n);
      for (int c = 0; c < NV; ++c) if (ok[c]) M[r * NV + c] += b.mass * dot(l[r], l[c]) + dot(a[r], Iw * a[c]);
    }
  }
}

// foot frame kinematics: out[f] = {pos3, ori3, vlin3, vang3, alin3, aang3} (18), R[f] (9), optional Jacobians J[f][18][93]
void orc_foot_kinematics(void* h, const double* x, const double* u, double* out, double* Rout, double* J) {
  const Oracle& o = *static_cast<Oracle*>(h);
  std::vector<AD> xa(NX), ua(NU);
  seed(x, u, xa.data(), ua.data());
  std::vector<FootKin<AD>> fk(2);
  foot_kinematics<AD>(o, xa.data(), ua.data(), fk.data());
  for (int f = 0; f < 2; ++f) {
    const V3<AD>* blocks[6] = {&fk[f].pos, &fk[f].ori, &fk[f].vlin, &fk[f].vang, &fk[f].alin, &fk[f].aang};
    for (int b = 0; b < 6; ++b) for (int c = 0; c < 3; ++c) {
      const AD& e = (*blocks[b])[c];
      out[f * 18 + 3 * b + c] = e.v;
      if (J) for (int a = 0; a < NZ; ++a) J[(f * 18 + 3 * b + c) * NZ + a] = e.d[a];
    }
    if (Rout) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rout[f * 9 + 3 * i + j] = fk[f].R.m[i][j].v;
  }
}

// body frame placements (R row-major 9, p 3) for all bodies
void orc_body_placements(void* h, const double* q, double* R, double* p) {
  const Oracle& o = *static_cast<Oracle*>(h);
  Kin<double> k;
  forward_kinematics<double>(o, q, static_cast<const double*>(nullptr), static_cast<const double*>(nullptr), k);
  for (int i = 0; i < NB; ++i) {
    for (int a = 0; a < 3; ++a) { p[3 * i + a] = k.p[i][a]; for (int b = 0; b < 3; ++b) R[9 * i + 3 * a + b] = k.R[i].m[a][b]; }
  }
}

void orc_collision(void* h, const double* x, double* hval) { collision_distances<double>(*static_cast<Oracle*>(h), x, hval); }

double orc_stage_cost(void* h, const double* x, const double* u, const double* par, double* eq, int* ne) {
  return stage_terms(*static_cast<Oracle*>(h), x, u, par, nullptr, eq, ne);
}

void orc_rk4(void* h, const double* x, const double* u, double dt, double* xnext) { rk4_value(*static_cast<Oracle*>(h), x, u, dt, xnext); }

// LQ approximation of every node of one instance.  Output blocks use the HSQP_BLK_* layouts of include/hsqp.h.
void orc_lq(void* h, int N, double dt, const double* x, const double* u, const double* par, int threads,
            double* AB, double* bvec, double* H, double* g, double* CDe, int* ne, double* cost, double* flow) {
  const Oracle& o = *static_cast<Oracle*>(h);
#pragma omp parallel for num_threads(threads) schedule(dynamic)
  for (int k = 0; k < N; ++k) {
    NodeLQ* lq = new NodeLQ;
    node_lq(o, x + k * NX, u + k * NU, x + (k + 1) * NX, par + k * NP, dt_at(o, k, dt), *lq);
    if (AB) std::memcpy(AB + (size_t)k * NX * NZ, lq->AB, sizeof(lq->AB));
    if (bvec) std::memcpy(bvec + k * NX, lq->b, sizeof(lq->b));
    if (H) std::memcpy(H + (size_t)k * NZ * NZ, lq->H, sizeof(lq->H));
    if (g) std::memcpy(g + k * NZ, lq->g, sizeof(lq->g));
    if (CDe) std::memcpy(CDe + (size_t)k * NE_MAX * (NZ + 1), lq->CDe, sizeof(lq->CDe));
    if (ne) ne[k] = lq->ne;
    if (cost) cost[k] = lq->cost;
    if (flow) std::memcpy(flow + k * NX, lq->flow, sizeof(lq->flow));
    delete lq;
  }
  if (cost) cost[N] = terminal_cost(o, x + N * NX, par + N * NP, nullptr, nullptr);
}

// One full SQP iteration of one instance (alpha = 1).  Returns 0 on success, HSQP_ERR_NUMERIC otherwise.
// proj_out (optional): per node packed {Px[35*58], Pe[35], PuPuT[35*35]} — the basis-independent projection data.
int orc_sqp_iteration(void* h, int N, double dt, const double* x_init, const double* x, const double* u, const double* par, int threads,
                      double* x_new, double* u_new, double* dx_out, double* du_out, hsqp_perf* perf_before, hsqp_perf* perf_after,
                      double* kkt, double* proj_out, double* armijo_out) {
  const Oracle& o = *static_cast<Oracle*>(h);
  std::vector<Projected> st(N);
  int bad = 0;
#pragma omp parallel for num_threads(threads) schedule(dynamic)
  for (int k = 0; k < N; ++k) {
    NodeLQ* lq = new NodeLQ;
    node_lq(o, x + k * NX, u + k * NU, x + (k + 1) * NX, par + k * NP, dt_at(o, k, dt), *lq);
    if (!project_node(*lq, st[k])) {
#pragma omp atomic write
      bad = 1;
    }
    delete lq;
  }
  if (bad) return HSQP_ERR_NUMERIC;
  double HN[NX], gN[NX], dx0[NX];
  terminal_cost(o, x + N * NX, par + N * NP, HN, gN);
  for (int i = 0; i < NX; ++i) dx0[i] = x_init[i] - x[i];
  RiccatiOut r;
  if (!riccati(st, HN, gN, dx0, N, r)) return HSQP_ERR_NUMERIC;
  std::vector<double> du(N * NU);
  for (int k = 0; k < N; ++k) {
    const Projected& p = st[k];
    for (int i = 0; i < NU; ++i) {
      double a = p.Pe[i];
      for (int j = 0; j < NX; ++j) a += p.Px[i * NX + j] * r.dx[k * NX + j];
      for (int j = 0; j < p.nut; ++j) a += p.Pu[i * p.nut + j] * r.ut[k * NU + j];
      du[k * NU + i] = a;
    }
    if (proj_out) {
      double* po = proj_out + (size_t)k * (NU * NX + NU + NU * NU);
      std::memcpy(po, p.Px.data(), sizeof(double) * NU * NX);
      std::memcpy(po + NU * NX, p.Pe.data(), sizeof(double) * NU);
      for (int i = 0; i < NU; ++i) for (int j = 0; j < NU; ++j) {
        double a = 0.0;
        for (int l = 0; l < p.nut; ++l) a += p.Pu[i * p.nut + l] * p.Pu[j * p.nut + l];
        po[NU * NX + NU + i * NU + j] = a;
      }
    }
  }
  if (kkt) kkt_residual(st, HN, gN, dx0, N, r, &kkt[0], &kkt[1]);
  if (armijo_out) {  // ASSUMPTION A5
    double am = 0.0;
    for (int k = 0; k < N; ++k) {
      const Projected& p = st[k];
      for (int i = 0; i < NX; ++i) am += p.qt[i] * r.dx[k * NX + i];
      for (int j = 0; j < p.nut; ++j) am += p.rt[j] * r.ut[k * NU + j];
    }
    for (int i = 0; i < NX; ++i) am += gN[i] * r.dx[N * NX + i];
    *armijo_out = am;
  }
  std::vector<double> xn((N + 1) * NX), un(N * NU);
  for (int i = 0; i < (N + 1) * NX; ++i) xn[i] = x[i] + r.dx[i];
  for (int i = 0; i < N * NU; ++i) un[i] = u[i] + du[i];
  if (x_new) std::memcpy(x_new, xn.data(), sizeof(double) * xn.size());
  if (u_new) std::memcpy(u_new, un.data(), sizeof(double) * un.size());
  if (dx_out) std::memcpy(dx_out, r.dx.data(), sizeof(double) * r.dx.size());
  if (du_out) std::memcpy(du_out, du.data(), sizeof(double) * du.size());
  if (perf_before) performance(o, N, dt, x, u, par, threads, perf_before);
  if (perf_after) performance(o, N, dt, xn.data(), un.data(), par, threads, perf_after);
  return 0;
}

// Riccati on externally supplied projected LQ data is not exposed: tests validate the QP solution with a dense KKT solve
// of the UNPROJECTED problem instead (tests/test_oracle_qp.py), which also covers the projection.

// Filter line search on the step (dx, du) from (x, u): ASSUMPTION A6.  settings = {g_max, g_min, gamma_c, armijoFactor,
// alpha_decay, alpha_min, deltaTol}.  step type: 0 COST, 1 DUAL, 2 CONSTRAINT, 3 ZERO.
}  // extern "C"
template <class Perf>
static void filter_linesearch(const Perf& performance_of, int N, const double* x, const double* u, const double* dx, const double* du,
                              const double* settings, double armijo, double* alpha_out, int* type_out, int* trials_out,
                              double* x_new, double* u_new, hsqp_perf* perf_new) {
  const double g_max = settings[0], g_min = settings[1], gamma_c = settings[2], armijo_factor = settings[3], decay = settings[4],
               alpha_min = settings[5], delta_tol = settings[6];
  hsqp_perf base;
  performance_of(x, u, &base);
  double nx2 = 0.0, nu2 = 0.0;
  for (int i = 0; i < (N + 1) * NX; ++i) nx2 += dx[i] * dx[i];
  for (int i = 0; i < N * NU; ++i) nu2 += du[i] * du[i];
  const double dxn = std::sqrt(nx2), dun = std::sqrt(nu2);
  const double g = std::sqrt(base.dynamics_sse + base.equality_sse);
  std::vector<double> xn((N + 1) * NX), un(N * NU);
  double alpha = 1.0;
  int trials = 0;
  do {
    for (int i = 0; i < (N + 1) * NX; ++i) xn[i] = x[i] + alpha * dx[i];
    for (int i = 0; i < N * NU; ++i) un[i] = u[i] + alpha * du[i];
    hsqp_perf pn;
    performance_of(xn.data(), un.data(), &pn);
    ++trials;
    const double gn = std::sqrt(pn.dynamics_sse + pn.equality_sse);
    const double am = alpha * armijo;
    bool ok;
    int type;
    if (gn > g_max) { ok = gn < (1.0 - gamma_c) * g; type = 2; }
    else if (gn < g_min && g < g_min && am < 0.0) { ok = pn.merit < base.merit + armijo_factor * am; type = 0; }
    else { ok = (pn.merit < base.merit - gamma_c * g) || (gn < (1.0 - gamma_c) * g); type = 1; }
    if (ok) {
      *alpha_out = alpha; *type_out = type; *trials_out = trials;
      std::memcpy(x_new, xn.data(), sizeof(double) * xn.size());
      std::memcpy(u_new, un.data(), sizeof(double) * un.size());
      *perf_new = pn;
      return;
    }
    alpha *= decay;
    if (alpha * dun < delta_tol && alpha * dxn < delta_tol) break;
  } while (alpha >= alpha_min);
  *alpha_out = 0.0; *type_out = 3; *trials_out = trials;
  std::memcpy(x_new, x, sizeof(double) * (N + 1) * NX);
  std::memcpy(u_new, u, sizeof(double) * N * NU);
  *perf_new = base;
}
extern "C" {
void orc_linesearch(void* h, int N, double dt, const double* x, const double* u, const double* dx, const double* du, const double* par,
                    int threads, const double* settings, double armijo, double* alpha_out, int* type_out, int* trials_out,
                    double* x_new, double* u_new, hsqp_perf* perf_new) {
  const Oracle& o = *static_cast<Oracle*>(h);
  filter_linesearch([&](const double* xx, const double* uu, hsqp_perf* out) { performance(o, N, dt, xx, uu, par, threads, out); }, N, x, u, dx, du,
                    settings, armijo, alpha_out, type_out, trials_out, x_new, u_new, perf_new);
}

void orc_performance(void* h, int N, double dt, const double* x, const double* u, const double* par, int threads, hsqp_perf* out) {
  performance(*static_cast<Oracle*>(h), N, dt, x, u, par, threads, out);
}

// the nominal state / input the quadratic cost uses at (x, node parameters): StateInputQuadraticCost::getStateInputDeviation's xNominal, uNominal
void orc_nominal(void* hh, const double* x, const double* par, double* xnom, double* unom) { nominal(*static_cast<Oracle*>(hh), x, par, xnom, unom); }
// the friction-cone value / derivatives the LQ code uses, and the Hessian diagonal shift it applies to every state and input
void orc_friction_cone(void* hh, const double* F, double* h, double* dh, double* d2, double* shift) {
  const Oracle& o = *static_cast<Oracle*>(hh);
  double d[3][3];
  friction_cone(o.md, F, *h, dh, d);
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) d2[3 * a + b] = d[a][b];
  *shift = o.md.friction_hess_shift;
}
double orc_penalty(int kind, double mu, double delta, double hval, double* d1, double* d2) {
  const Pen p = kind == 0 ? relaxed_barrier(mu, delta, hval) : pwp_barrier(mu, delta, hval);
  if (d1) *d1 = p.d1;
  if (d2) *d2 = p.d2;
  return p.p;
}

}  // extern "C"

#include "centroidal.hpp"   // groundwork for the centroidal formulation: flow map only
