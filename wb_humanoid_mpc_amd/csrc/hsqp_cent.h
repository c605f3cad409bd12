// Centroidal formulation of the humanoid MPC (BASELINE configs 1-2): LQ approximation of one shooting node.
//
//   x = [h/m (6) | p_b (3) eulerZYX (3) | q_j (23)],  u = [W_l (6) W_r (6) | qd_j (23)]
//   (humanoid_nmpc/humanoid_centroidal_mpc/include/humanoid_centroidal_mpc/common/CentroidalMpcRobotModel.h:49-71,89-95)
//
// The problem is embedded in the whole-body array layout (include/hsqp.h): a state row keeps 58 doubles of which the first 35
// are the centroidal state and the other 23 are decoupled padding states (A = I, no cost, dx = 0), so the projection, Riccati,
// step, KKT and performance kernels run unchanged on the record this file writes; only the structured [A|B] differs
// (project_node(..., cent = true)).
//
// This header holds what the centroidal node function (hsqp_cent_lq.h) is made of: the number types (double / one-tangent dual numbers),
// the closing solve of a model pass, the cost / constraint terms.  Round 4 replaced the first form of the kernels (every lane walked the
// kinematic tree on dual numbers with private copies of every body record) by values once per node in an LDS workspace and tangent lanes on
// closed-form seeds: hsqp_cent_lq.h.
//
// The flow map is NOT evaluated as the reference / the oracle write it (centroidal momentum matrix column by column,
// ocs2_centroidal_model — oracle ASSUMPTION A7) but from the momentum balance directly:
//   one tree pass with the joint rates only gives every body's placement, its velocity relative to the base (om_i, v_i), the
//   momentum of the joint motion (lin_J, ang_J about the centre of mass), the centre of mass and the composite rotational
//   inertia I_c of the whole robot about it; with E the euler-rate axes,
//     A_b = [[m 1, -m [com - p_b]x E], [0, I_c E]],   A_j qd_j = [lin_J; ang_J]
//   so   euler rates = (I_c E)^-1 (m h_ang - ang_J),   pdot = (m h_lin - lin_J)/m - (E euler rates) x (com - p_b),
//   and  d(h/m)/dt = [g + sum f / m ;  sum ((p_c - com) x f + tau) / m].
// Body velocities of the velocity-level model follow from the same pass plus the rigid base motion.
#pragma once
#include <type_traits>
#include "hsqp_lq.h"

namespace hsqp {

constexpr int CNX = HSQP_CNX, CNZ = CNX + NU;    // 35, 70
// residual row slots (NRS = 64).  Position rows of the foot / torso task-space costs are not carried: their weights are zero in
// the G1 task file and build_dev_model() rejects non-zero ones.
constexpr int CROW_FOOT = 0;     // 9 f + {ori(3), vlin(3), vang(3)}
constexpr int CROW_TORSO = 18;   // ori(3), vlin(3), vang(3)
constexpr int CROW_FRIC = 27;    // 4 f + k
constexpr int CROW_MXY = 35;     // 4 f + k
constexpr int CROW_COLL = 43;    // 16 rows (inactive in double support)
// external-torque rows (6 per stance foot): double support -> the collision slots 43 + 6 f; single support -> the swing foot's
// friction and moment slots (both inactive then)
HSQP_HD int crow_ext(int f, bool both, int k) { return both ? CROW_COLL + 6 * f + k : (k < 4 ? CROW_FRIC + 4 * (1 - f) + k : CROW_MXY + 4 * (1 - f) + k - 4); }

// ---- dual numbers with one tangent
struct Dual1 { double v, d; };
HSQP_HD Dual1 mk(double v, double d = 0.0) { Dual1 r; r.v = v; r.d = d; return r; }
HSQP_HD Dual1 operator+(Dual1 a, Dual1 b) { return mk(a.v + b.v, a.d + b.d); }
HSQP_HD Dual1 operator-(Dual1 a, Dual1 b) { return mk(a.v - b.v, a.d - b.d); }
HSQP_HD Dual1 operator-(Dual1 a) { return mk(-a.v, -a.d); }
HSQP_HD Dual1 operator*(Dual1 a, Dual1 b) { return mk(a.v * b.v, a.v * b.d + a.d * b.v); }
HSQP_HD Dual1 operator*(Dual1 a, double b) { return mk(a.v * b, a.d * b); }
HSQP_HD Dual1 operator*(double b, Dual1 a) { return mk(a.v * b, a.d * b); }
HSQP_HD Dual1 operator+(Dual1 a, double b) { return mk(a.v + b, a.d); }
HSQP_HD Dual1 operator-(Dual1 a, double b) { return mk(a.v - b, a.d); }
HSQP_HD Dual1 operator/(Dual1 a, Dual1 b) { const double q = a.v / b.v; return mk(q, (a.d - q * b.d) / b.v); }
HSQP_HD Dual1 dsqrt(Dual1 a) { const double s = sqrt(a.v); return mk(s, 0.5 * a.d / s); }
HSQP_HD double dsqrt(double a) { return sqrt(a); }
HSQP_HD void dsincos(Dual1 a, Dual1& s, Dual1& c) { double sv, cv; sincos(a.v, &sv, &cv); s = mk(sv, cv * a.d); c = mk(cv, -sv * a.d); }   // one range reduction
HSQP_HD void dsincos(double a, double& s, double& c) { sincos(a, &s, &c); }
HSQP_HD double val(Dual1 a) { return a.v; }
HSQP_HD double val(double a) { return a; }
HSQP_HD double tan_of(Dual1 a) { return a.d; }
HSQP_HD double tan_of(double) { return 0.0; }
HSQP_HD void set_tan(Dual1& a) { a.d = 1.0; }
HSQP_HD void set_tan(double&) {}
template <class T> HSQP_HD T cst(double v);
template <> HSQP_HD Dual1 cst<Dual1>(double v) { return mk(v); }
template <> HSQP_HD double cst<double>(double v) { return v; }

template <class T> HSQP_HD void t_cross(const T* a, const T* b, T* r) {
  const T x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
template <class T> HSQP_HD T t_dot(const T* a, const T* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <class T> HSQP_HD void t_mulc(const T* M, const double* v, T* r) {   // r = M v, v constant
  for (int i = 0; i < 3; ++i) r[i] = M[3 * i] * v[0] + M[3 * i + 1] * v[1] + M[3 * i + 2] * v[2];
}
template <class T> HSQP_HD void t_mulv(const T* M, const T* v, T* r) {
  for (int i = 0; i < 3; ++i) r[i] = M[3 * i] * v[0] + M[3 * i + 1] * v[1] + M[3 * i + 2] * v[2];
}
template <class T> HSQP_HD void t_tmulv(const T* M, const T* v, T* r) {       // r = M^T v
  for (int i = 0; i < 3; ++i) r[i] = M[i] * v[0] + M[3 + i] * v[1] + M[6 + i] * v[2];
}
template <class T> HSQP_HD void t_inverse3(const T* a, T* c) {
  c[0] = a[4] * a[8] - a[5] * a[7]; c[1] = a[2] * a[7] - a[1] * a[8]; c[2] = a[1] * a[5] - a[2] * a[4];
  c[3] = a[5] * a[6] - a[3] * a[8]; c[4] = a[0] * a[8] - a[2] * a[6]; c[5] = a[2] * a[3] - a[0] * a[5];
  c[6] = a[3] * a[7] - a[4] * a[6]; c[7] = a[1] * a[6] - a[0] * a[7]; c[8] = a[0] * a[4] - a[1] * a[3];
  const T inv = cst<T>(1.0) / (a[0] * c[0] + a[1] * c[3] + a[2] * c[6]);
  for (int i = 0; i < 9; ++i) c[i] = c[i] * inv;
}

// placement of a body and its angular / origin velocity RELATIVE to the base (joint rates only), as one lane sees them
template <class T>
struct BodyRec { T R[9], p[3], om[3], v[3]; };

// momentum sums of the walk: total m c, momentum of the joint motion about the world origin, composite inertia about the origin
template <class T>
struct CentSums { T mc[3], lin[3], angO[3], IO[6]; };

// The closing solve of a model pass: from the momentum sums of the joint motion (hsqp_cent_lq.h: cent_stage_values; positions relative to p0, or
// absolute with p0 given), the euler-rate axes E, the contact points pc (same origin as the sums), the normalized momentum h and the wrenches W:
//   com, the base velocity vb = [pdot; euler rates], its angular velocity wb, and xdot[0..11] = [d(h/m)/dt ; pdot ; euler rates].
template <class T>
HSQP_HD void cent_finish(const DevModel& dm, const CentSums<T>& sums, const T* E, const T (*pc)[3], const T* h, const T* W, T* xdot, T* vb, T* wb, T* com,
                         const T* p0 = nullptr) {
  const T zero = cst<T>(0.0);
  const double M = dm.total_mass, iM = 1.0 / M;
  for (int r = 0; r < 3; ++r) com[r] = sums.mc[r] * iM;
  const T* cm = com;
  const T* IO = sums.IO;
  const T cm2 = t_dot(cm, cm);
  T Ic[9];
  Ic[0] = IO[0] - (cm2 - cm[0] * cm[0]) * M; Ic[1] = IO[1] + (cm[0] * cm[1]) * M; Ic[2] = IO[2] + (cm[0] * cm[2]) * M;
  Ic[4] = IO[3] - (cm2 - cm[1] * cm[1]) * M; Ic[5] = IO[4] + (cm[1] * cm[2]) * M; Ic[8] = IO[5] - (cm2 - cm[2] * cm[2]) * M;
  Ic[3] = Ic[1]; Ic[6] = Ic[2]; Ic[7] = Ic[5];
  T angJ[3], t[3];
  t_cross(cm, sums.lin, t);
  for (int r = 0; r < 3; ++r) angJ[r] = sums.angO[r] - t[r];
  // euler rates = (Ic E)^-1 (M h_ang - angJ)
  T A22[9], A22i[9], rhs[3];
  for (int r = 0; r < 3; ++r)
    for (int e = 0; e < 3; ++e) A22[3 * r + e] = Ic[3 * r] * E[e] + Ic[3 * r + 1] * E[3 + e] + Ic[3 * r + 2] * E[6 + e];
  t_inverse3(A22, A22i);
  for (int r = 0; r < 3; ++r) rhs[r] = h[3 + r] * M - angJ[r];
  t_mulv(A22i, rhs, vb + 3);
  t_mulv(E, vb + 3, wb);
  T d[3];
  for (int r = 0; r < 3; ++r) d[r] = p0 ? cm[r] - p0[r] : cm[r];
  t_cross(wb, d, t);
  for (int r = 0; r < 3; ++r) vb[r] = h[r] - sums.lin[r] * iM - t[r];
  // normalized momentum rate (getNormalizedCentroidalMomentumRate; gravity 9.81 hard-coded as upstream does)
  T fs[3] = {zero, zero, cst<T>(-9.81 * M)}, ns[3] = {zero, zero, zero};
  for (int f = 0; f < 2; ++f) {
    T arm[3];
    for (int r = 0; r < 3; ++r) arm[r] = pc[f][r] - cm[r];
    t_cross(arm, W + 6 * f, t);
    for (int r = 0; r < 3; ++r) { fs[r] = fs[r] + W[6 * f + r]; ns[r] = ns[r] + t[r] + W[6 * f + 3 + r]; }
  }
  for (int r = 0; r < 3; ++r) { xdot[r] = fs[r] * iM; xdot[3 + r] = ns[r] * iM; }
  for (int r = 0; r < 6; ++r) xdot[6 + r] = vb[r];
}

// world position and LOCAL_WORLD_ALIGNED velocity of a point fixed to a body (velocity-level model: base motion + joint motion)
template <class T>
HSQP_HD void cent_point(const T* p0, const T* vb, const T* wb, const BodyRec<T>& b, const double* pl, T* pos, T* vlin, T* vang) {
  T rp[3], d[3], t1[3], t2[3];
  t_mulc(b.R, pl, rp);
  for (int r = 0; r < 3; ++r) { pos[r] = b.p[r] + rp[r]; d[r] = pos[r] - p0[r]; vang[r] = wb[r] + b.om[r]; }
  t_cross(wb, d, t1);            // rigid base motion of the point
  t_cross(b.om, rp, t2);         // joint motion relative to the base
  for (int r = 0; r < 3; ++r) vlin[r] = vb[r] + t1[r] + b.v[r] + t2[r];
}
// quaternion (x, y, z, w) of a rotation matrix, branch chosen on the values (oracle ASSUMPTION A8)
template <class T>
HSQP_HD void cent_quat(const T* R, T* q) {
  const double tr = val(R[0]) + val(R[4]) + val(R[8]);
  const T one = cst<T>(1.0);
  if (tr > 0.0) {
    const T s = dsqrt(R[0] + R[4] + R[8] + one) * 2.0;
    q[3] = s * 0.25; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s;
  } else if (val(R[0]) > val(R[4]) && val(R[0]) > val(R[8])) {
    const T s = dsqrt(one + R[0] - R[4] - R[8]) * 2.0;
    q[3] = (R[7] - R[5]) / s; q[0] = s * 0.25; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s;
  } else if (val(R[4]) > val(R[8])) {
    const T s = dsqrt(one + R[4] - R[0] - R[8]) * 2.0;
    q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = s * 0.25; q[2] = (R[5] + R[7]) / s;
  } else {
    const T s = dsqrt(one + R[8] - R[0] - R[4]) * 2.0;
    q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = s * 0.25;
  }
}

// A SINK of cent_terms (hsqp_cent_lq.h: CentTangentSink, CentValueSink) has: header, fric_d1, gn (Gauss-Newton row: value r, weight
// sqrt(w)), pen (penalty row: constraint value h, penalty p), raw (a row with an explicit scale and no gradient / cost contribution: the
// friction cone's curvature rows), eq (equality row).  KIN answers p0 / vb / wb (base position, [pdot; euler rates], angular velocity),
// foot(f) / torso() (body records), point(p) (collision point p), ext_arm(f, a) (external-torque joint a of foot f).
// Cost / constraint terms of the node from the stage-1 kinematics `kin` at (x, u) (W = the twelve wrench entries of u); order and sources as
// oracle/centroidal.hpp.
template <class T, class Kin, class Sink>
HSQP_HD void cent_terms(const DevModel& dm, const Kin& kin, const T* W, const double* par, Sink& o) {
  const T zero = cst<T>(0.0);
  const int c0 = par[HSQP_P_CONTACT] > 0.5, c1 = par[HSQP_P_CONTACT + 1] > 0.5;
  const bool both = c0 && c1;
  const int contact[2] = {c0, c1}, eq_off[2] = {0, c0 ? 6 : 7};
  o.header(eq_off[1] + (c1 ? 6 : 7), c0, c1, eq_off[0], eq_off[1]);
  // ---- torso task-space cost: EndEffectorKinematicsQuadraticCost.cpp:110-138 (quaternionDistance, velocity differences)
  {
    const BodyRec<T> tb = kin.torso();
    T Rt[9], qc[4], pos[3], vl[3], va[3];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Rt[3 * r + c] = tb.R[3 * r] * dm.torso_R[c] + tb.R[3 * r + 1] * dm.torso_R[3 + c] + tb.R[3 * r + 2] * dm.torso_R[6 + c];
    cent_quat(Rt, qc);
    cent_point(kin.p0(), kin.vb(), kin.wb(), tb, dm.torso_p, pos, vl, va);
    const double* ref = par + HSQP_PC_TORSO;
    const T rv[3] = {cst<T>(ref[3]), cst<T>(ref[4]), cst<T>(ref[5])};
    T cr[3];
    t_cross(qc, rv, cr);
    for (int c = 0; c < 3; ++c) {
      o.gn(CROW_TORSO + c, rv[c] * qc[3] - qc[c] * ref[6] + cr[c], dm.torso_sqrt_w[3 + c]);
      o.gn(CROW_TORSO + 3 + c, vl[c] - ref[7 + c], dm.torso_sqrt_w[6 + c]);
      o.gn(CROW_TORSO + 6 + c, va[c] - ref[10 + c], dm.torso_sqrt_w[9 + c]);
    }
  }
  // ---- foot collision (FootCollisionConstraint.cpp:92-144), inactive in double support
  if (!both) {
    for (int r = 0; r < 16; ++r) {
      int a, b;
      coll_pair(r, a, b);
      T pa[3], pb[3], dd[3];
      kin.point(a, pa);
      kin.point(b, pb);
      for (int c = 0; c < 3; ++c) dd[c] = pa[c] - pb[c];
      const T hh = dsqrt(t_dot(dd, dd)) - 2.0 * (r == 9 ? dm.r_knee : dm.r_foot);
      o.pen(CROW_COLL + r, hh, pwp_barrier(dm.coll_bmu, dm.coll_bdelta, val(hh)));
    }
  }
  // ---- per foot
  for (int f = 0; f < 2; ++f) {
    const int ct = contact[f];
    const BodyRec<T> fb = kin.foot(f);
    T pos[3], vl[3], va[3], ori[3];
    cent_point(kin.p0(), kin.vb(), kin.wb(), fb, dm.contact_p[f], pos, vl, va);
    {  // orientation error to the ground plane (oracle ASSUMPTION A2): (n x a) / sqrt(2 (1 + a.n)), a = R e_z, n = e_z
      const T s = dsqrt((fb.R[8] + 1.0) * 2.0);
      ori[0] = -fb.R[5] / s; ori[1] = fb.R[2] / s; ori[2] = zero;
    }
    const T* Wf = W + 6 * f;
    if (ct) {
      // friction cone (FrictionForceConeConstraint.cpp:78-224): relaxed barrier of h; its second-order term p' d2h as three rows
      const T T2 = Wf[0] * Wf[0] + Wf[1] * Wf[1] + dm.friction_reg;
      const T hh = (Wf[2] + dm.friction_grip) * dm.friction_mu - dsqrt(T2);
      const Pen3 p = relaxed_barrier(dm.friction_bmu, dm.friction_bdelta, val(hh));
      o.pen(CROW_FRIC + 4 * f, hh, p);
      o.fric_d1(f, p.d1);
      const double T3 = val(T2) * sqrt(val(T2));
      o.raw(CROW_FRIC + 4 * f + 1, Wf[0], sqrt(-p.d1 * dm.friction_reg / T3));
      o.raw(CROW_FRIC + 4 * f + 2, Wf[1], sqrt(-p.d1 * dm.friction_reg / T3));
      o.raw(CROW_FRIC + 4 * f + 3, Wf[0] * val(Wf[1]) - Wf[1] * val(Wf[0]), sqrt(-p.d1 / T3));
      // contact moment XY (ContactMomentXYConstraintCppAd.cpp:77-104): wrench in the contact frame
      T lf[3], lm[3];
      t_tmulv(fb.R, Wf, lf);
      t_tmulv(fb.R, Wf + 3, lm);
      const T hm[4] = {lm[0] - lf[2] * dm.rect_y_min, lf[2] * dm.rect_y_max - lm[0], -lm[1] - lf[2] * dm.rect_x_min, lm[1] + lf[2] * dm.rect_x_max};
      for (int r = 0; r < 4; ++r) o.pen(CROW_MXY + 4 * f + r, hm[r], relaxed_barrier(dm.moment_bmu, dm.moment_bdelta, val(hm[r])));
      // external torque cost (ExternalTorqueQuadraticCostAD.cpp:110-135): (J_ee^T W)[6 + j] .* sqrtW * (1 - impactProximity of the other foot)
      const double mid = 1.0 - par[HSQP_P_IMPACT + (1 - f)];
      for (int a = 0; a < 6; ++a) {
        T ea[4];
        kin.ext_arm(f, a, ea);
        const T tau = pos[0] * ea[0] + pos[1] * ea[1] + pos[2] * ea[2] + ea[3];
        o.gn(crow_ext(f, both, a), tau, dm.ext_sqrt_w[f][a] * mid);
      }
    }
    // equalities: zeroWrench (swing), zeroVelocity (stance), normalVelocity (swing) — CentroidalMpcInterface.cpp:203-207,232-257
    const int r0 = eq_off[f];
    const double zp = par[HSQP_P_SWING + 3 * f], zv = par[HSQP_P_SWING + 3 * f + 1];
    if (ct) {
      for (int i = 0; i < 3; ++i) o.eq(r0 + i, i == 2 ? vl[2] + (pos[2] - zp) * dm.gain_pos_z : vl[i]);
      for (int i = 0; i < 3; ++i) o.eq(r0 + 3 + i, va[i] + ori[i] * dm.gain_ori);
    } else {
      for (int i = 0; i < 6; ++i) o.eq(r0 + i, Wf[i]);
      o.eq(r0 + 6, vl[2] - zv + (pos[2] - zp) * dm.gain_pos_z);
    }
    // foot task-space cost (CentroidalMpcEndEffectorFootCost.cpp:90-152): [oriErr, v * impactProximity, w] .* sqrtW
    const double ip = par[HSQP_P_IMPACT + f];
    for (int c = 0; c < 3; ++c) {
      o.gn(CROW_FOOT + 9 * f + c, ori[c], dm.cent_foot_sqrt_w[3 + c]);
      o.gn(CROW_FOOT + 9 * f + 3 + c, vl[c], dm.cent_foot_sqrt_w[6 + c] * ip);
      o.gn(CROW_FOOT + 9 * f + 6 + c, va[c], dm.cent_foot_sqrt_w[9 + c]);
    }
  }
}

// Expand the centroidal record into the dense padded [A|B] (58 x 93) — debug / parity path and tests.
inline void cent_expand_AB(const double* rec, double dt, double* AB) {
  for (int i = 0; i < NX * NZ; ++i) AB[i] = 0.0;
  for (int i = 0; i < NX; ++i) AB[i * NZ + i] = 1.0;
  for (int j = 0; j < NJ; ++j) AB[(12 + j) * NZ + NX + 12 + j] = dt;
  for (int r = 0; r < 12; ++r)
    for (int c = 0; c < NZ; ++c) AB[r * NZ + c] += rec[REC_PV + r * LDJ + c];
}

}  // namespace hsqp
