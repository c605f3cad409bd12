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
// Device path: forward-mode derivatives with ONE tangent direction per lane.  Lane c of a lane group evaluates a scalar program of
// the node on dual numbers whose tangent is d/dz_c, z = [x(35); u(35)]; lane 70 carries no tangent and writes the values; lanes
// 71..96 zero-fill the padding columns of the record.  The program has two independent halves that run on different waves of the
// workgroup: the four RK4 stages of the flow map (group 0) and one tree pass plus every cost / constraint term (group 1).  Lanes
// never exchange data: there are no barriers and no LDS traffic; the tree pass keeps the parent's record in registers (chain
// walk) and what does not fit lives in private memory.  The value-only pass (performance index, line search) runs the same two
// halves on plain doubles, two lanes per node.
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
constexpr int CENT_LANES = 97;                   // 70 tangent lanes + 1 value lane + 26 zero-fill lanes (columns 35..57, 93..95)
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

// ---- per-lane state of one model pass.  The tree is walked chain by chain (DevModel::chain_*: maximal single-child paths): inside
// a chain the parent is the previous body and stays in registers; only the records of the chain ENDS (where other chains attach)
// are kept in a small table.  What the cost / constraint terms need later is collected in side tables while the walk passes the
// bodies concerned (stage 1 only: TERMS = true).
template <class T>
struct BodyRec { T R[9], p[3], om[3], v[3]; };   // placement; angular / origin velocity RELATIVE to the base (joint rates only)
constexpr int CENT_MAX_SLOTS = 8;                // base + chains of the kinematic tree (build_dev_model checks)

template <class T>
struct CentSide {
  BodyRec<T> foot[2], torso;
  T pts[10][3];                  // collision points (order of DevModel::coll_body)
  T ea[2][6][4];                 // external-torque joints: tau = pos_foot . ea[0..2] + ea[3]  (ea = {f x w_j, w_j . m - p_j . (f x w_j)})
};
template <class T>
struct CentKin {
  BodyRec<T> ends[CENT_MAX_SLOTS];   // slot 0 = base, slot 1 + c = last body of chain c
  T E[9];                        // E[3 r + e]: world axis of euler rate e (z, y, x)
  T p0[3], com[3];
  T vb[6];                       // [pdot; euler rates]
  T wb[3];                       // angular velocity of the base = E euler rates
  CentSide<T> side;
};

// momentum sums of the walk: total m c, momentum of the joint motion about the world origin, composite inertia about the origin
template <class T>
struct CentSums { T mc[3], lin[3], angO[3], IO[6]; };

template <class T>
HSQP_HD void cent_accumulate(const DevModel& dm, int i, const BodyRec<T>& b, CentSums<T>& a) {
  const double m = dm.mass[i];
  T rc[3], c[3], t[3], vc[3];
  t_mulc(b.R, dm.com[i], rc);
  for (int r = 0; r < 3; ++r) c[r] = b.p[r] + rc[r];
  t_cross(b.om, rc, t);
  for (int r = 0; r < 3; ++r) vc[r] = b.v[r] + t[r];
  // Iw = R I R^T (symmetric): RI = R I first
  T RI[9], Iw[6];
  for (int r = 0; r < 3; ++r)
    for (int cc = 0; cc < 3; ++cc) RI[3 * r + cc] = b.R[3 * r] * dm.inertia[i][cc] + b.R[3 * r + 1] * dm.inertia[i][3 + cc] + b.R[3 * r + 2] * dm.inertia[i][6 + cc];
  {
    int n = 0;
    for (int r = 0; r < 3; ++r)
      for (int cc = r; cc < 3; ++cc) Iw[n++] = RI[3 * r] * b.R[3 * cc] + RI[3 * r + 1] * b.R[3 * cc + 1] + RI[3 * r + 2] * b.R[3 * cc + 2];
  }
  const T c2 = t_dot(c, c);
  a.IO[0] = a.IO[0] + Iw[0] + (c2 - c[0] * c[0]) * m; a.IO[1] = a.IO[1] + Iw[1] - (c[0] * c[1]) * m; a.IO[2] = a.IO[2] + Iw[2] - (c[0] * c[2]) * m;
  a.IO[3] = a.IO[3] + Iw[3] + (c2 - c[1] * c[1]) * m; a.IO[4] = a.IO[4] + Iw[4] - (c[1] * c[2]) * m; a.IO[5] = a.IO[5] + Iw[5] + (c2 - c[2] * c[2]) * m;
  T mv[3], cxmv[3];
  for (int r = 0; r < 3; ++r) { mv[r] = vc[r] * m; a.mc[r] = a.mc[r] + c[r] * m; a.lin[r] = a.lin[r] + mv[r]; }
  t_cross(c, mv, cxmv);
  const T* o = b.om;
  a.angO[0] = a.angO[0] + Iw[0] * o[0] + Iw[1] * o[1] + Iw[2] * o[2] + cxmv[0];
  a.angO[1] = a.angO[1] + Iw[1] * o[0] + Iw[3] * o[1] + Iw[4] * o[2] + cxmv[1];
  a.angO[2] = a.angO[2] + Iw[2] * o[0] + Iw[4] * o[1] + Iw[5] * o[2] + cxmv[2];
}

// side tables for body i (world joint axis w; W = contact wrenches of the node)
template <class T>
HSQP_HD void cent_collect(const DevModel& dm, int i, const BodyRec<T>& b, const T* w, const T* W, CentSide<T>& sd) {
#pragma unroll
  for (int f = 0; f < 2; ++f)
    if (i == dm.contact_body[f]) sd.foot[f] = b;
  if (i == dm.torso_body) sd.torso = b;
#pragma unroll
  for (int p = 0; p < 10; ++p)
    if (i == dm.coll_body[p]) {
      T rp[3];
      t_mulc(b.R, dm.coll_p[p], rp);
      for (int r = 0; r < 3; ++r) sd.pts[p][r] = b.p[r] + rp[r];
    }
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int a = 0; a < 6; ++a)
      if (i == 1 + dm.ext_joint[f][a]) {
        // tau_j = w_j . (m + (pos - p_j) x f) = pos . (f x w_j) + (w_j . m - p_j . (f x w_j)); zero unless joint j carries the foot
        const int cb = dm.contact_body[f];
        const bool carries = cb >= i && cb < i + dm.subtree_size[i];
        T fxw[3];
        t_cross(W + 6 * f, w, fxw);
        const T rest = t_dot(w, W + 6 * f + 3) - t_dot(b.p, fxw);
        for (int r = 0; r < 3; ++r) sd.ea[f][a][r] = carries ? fxw[r] : cst<T>(0.0);
        sd.ea[f][a][3] = carries ? rest : cst<T>(0.0);
      }
}

// The closing solve of a model pass: from the momentum sums of the joint motion (cent_accumulate; positions relative to p0, or absolute with
// p0 given), the euler-rate axes E, the contact points pc (same origin as the sums), the normalized momentum h and the wrenches W:
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

// One pass over the kinematic tree at q = [p_b, euler, q_j] with joint rates qd, then the base velocity from the normalized
// momentum h and the normalized momentum rate for the contact wrenches W.  xdot[0..11] = [d(h/m)/dt ; pdot ; euler rates].
// WAVE_TRIG (device, the LQ kernel's tangent lanes only): the 26 angles of the pass (3 euler + 23 joints) have the SAME value in every lane
// of a wave — the lanes differ in their tangents only —, and their sines / cosines are where the instructions of a pass went (52 double-
// precision library calls per lane and pass, ~ 6 k instructions, four passes in the RK4 half).  Lane a < 26 of the wave evaluates angle a
// once; every lane picks the pair up with v_readlane (uniform index) and attaches its own tangent.  Needs lanes 0 .. 25 of every wave
// that runs the pass to be active: cent_lq_node's lane layout guarantees it.
template <class T, bool TERMS, bool WAVE_TRIG = false>
HSQP_HD void cent_pass(const DevModel& dm, const T* h, const T* q, const T* W, const T* qd, CentKin<T>& k, T* xdot) {
  const T zero = cst<T>(0.0), one = cst<T>(1.0);
#if defined(__HIP_DEVICE_COMPILE__)
  double wt_s = 0.0, wt_c = 1.0;
  if constexpr (WAVE_TRIG) {
    const int wl = (int)(threadIdx.x & 63);
    if (wl < 3 + NJ) sincos(val(q[3 + wl]), &wt_s, &wt_c);
  }
#endif
  auto trig = [&](int a /* angle index: q[3 + a] */, T& sn, T& cs) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (WAVE_TRIG) {
      const double sv = readlane_f64(wt_s, a), cv = readlane_f64(wt_c, a);
      if constexpr (std::is_same<T, double>::value) { sn = sv; cs = cv; }
      else { const T ang = q[3 + a]; sn = mk(sv, cv * ang.d); cs = mk(cv, -sv * ang.d); }
      return;
    }
#endif
    dsincos(q[3 + a], sn, cs);
  };
  CentSums<T> sums;
  for (int r = 0; r < 3; ++r) { sums.mc[r] = zero; sums.lin[r] = zero; sums.angO[r] = zero; sums.IO[r] = zero; sums.IO[3 + r] = zero; }
  T pc[2][3];   // contact points
  {
    T sz, cz, sy, cy, sx, cx;
    trig(0, sz, cz); trig(1, sy, cy); trig(2, sx, cx);
    BodyRec<T> b0;   // R_0 = Rz Ry Rx
    b0.R[0] = cz * cy; b0.R[1] = cz * sy * sx - sz * cx; b0.R[2] = cz * sy * cx + sz * sx;
    b0.R[3] = sz * cy; b0.R[4] = sz * sy * sx + cz * cx; b0.R[5] = sz * sy * cx - cz * sx;
    b0.R[6] = -sy;     b0.R[7] = cy * sx;                b0.R[8] = cy * cx;
    k.E[0] = zero; k.E[3] = zero; k.E[6] = one;            // z axis
    k.E[1] = -sz;  k.E[4] = cz;   k.E[7] = zero;           // Rz e_y
    k.E[2] = cz * cy; k.E[5] = sz * cy; k.E[8] = -sy;      // Rz Ry e_x
    for (int r = 0; r < 3; ++r) { b0.p[r] = q[r]; k.p0[r] = q[r]; b0.om[r] = zero; b0.v[r] = zero; }
    cent_accumulate(dm, 0, b0, sums);
    if constexpr (TERMS) { const T w0[3] = {zero, zero, zero}; cent_collect(dm, 0, b0, w0, W, k.side); }
    for (int f = 0; f < 2; ++f)
      if (dm.contact_body[f] == 0) { T rp[3]; t_mulc(b0.R, dm.contact_p[f], rp); for (int r = 0; r < 3; ++r) pc[f][r] = b0.p[r] + rp[r]; }
    k.ends[0] = b0;
  }
  for (int c = 0; c < dm.n_chains; ++c) {
    BodyRec<T> cur = k.ends[dm.chain_par_slot[c]];
    const int i0 = dm.chain_start[c], len = dm.chain_len[c];
    for (int i = i0; i < i0 + len; ++i) {
      BodyRec<T> nb;
      T Rj[9];
      for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc)
          Rj[3 * r + cc] = cur.R[3 * r] * dm.Rfix[i][cc] + cur.R[3 * r + 1] * dm.Rfix[i][3 + cc] + cur.R[3 * r + 2] * dm.Rfix[i][6 + cc];
      T sn, cs;
      trig(2 + i, sn, cs);
      const double* a = dm.axis[i];
      // Rodrigues: Rot = I + s K + (1 - c) K^2,  K = [a]x (unit axis)
      const T omc = one - cs;
      T Rot[9];
      Rot[0] = one + omc * (a[0] * a[0] - 1.0);     Rot[1] = omc * (a[0] * a[1]) - sn * a[2];     Rot[2] = omc * (a[0] * a[2]) + sn * a[1];
      Rot[3] = omc * (a[0] * a[1]) + sn * a[2];     Rot[4] = one + omc * (a[1] * a[1] - 1.0);     Rot[5] = omc * (a[1] * a[2]) - sn * a[0];
      Rot[6] = omc * (a[0] * a[2]) - sn * a[1];     Rot[7] = omc * (a[1] * a[2]) + sn * a[0];     Rot[8] = one + omc * (a[2] * a[2] - 1.0);
      for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) nb.R[3 * r + cc] = Rj[3 * r] * Rot[cc] + Rj[3 * r + 1] * Rot[3 + cc] + Rj[3 * r + 2] * Rot[6 + cc];
      T off[3], t[3], w[3];
      t_mulc(cur.R, dm.pfix[i], off);
      t_mulc(Rj, dm.axis[i], w);
      t_cross(cur.om, off, t);
      for (int r = 0; r < 3; ++r) {
        nb.p[r] = cur.p[r] + off[r];
        nb.om[r] = cur.om[r] + w[r] * qd[i - 1];
        nb.v[r] = cur.v[r] + t[r];
      }
      cent_accumulate(dm, i, nb, sums);
      if constexpr (TERMS) cent_collect(dm, i, nb, w, W, k.side);
      for (int f = 0; f < 2; ++f)
        if (dm.contact_body[f] == i) { T rp[3]; t_mulc(nb.R, dm.contact_p[f], rp); for (int r = 0; r < 3; ++r) pc[f][r] = nb.p[r] + rp[r]; }
      cur = nb;
    }
    k.ends[1 + c] = cur;
  }
  cent_finish<T>(dm, sums, k.E, pc, h, W, xdot, k.vb, k.wb, k.com, k.p0);
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
template <class T>
HSQP_HD void cent_point(const CentKin<T>& k, const BodyRec<T>& b, const double* pl, T* pos, T* vlin, T* vang) { cent_point(k.p0, k.vb, k.wb, b, pl, pos, vlin, vang); }
// what cent_terms asks of the kinematics, answered from the side tables of a tree pass (cent_pass<T, true>); the LQ kernel answers the same
// questions from its LDS workspace (hsqp_cent_lq.h: CentLaneKin)
template <class T>
struct CentPassKin {
  const CentKin<T>& k;
  HSQP_HD const T* p0() const { return k.p0; }
  HSQP_HD const T* vb() const { return k.vb; }
  HSQP_HD const T* wb() const { return k.wb; }
  HSQP_HD const BodyRec<T>& foot(int f) const { return k.side.foot[f]; }
  HSQP_HD const BodyRec<T>& torso() const { return k.side.torso; }
  HSQP_HD void point(int p, T* out) const { for (int r = 0; r < 3; ++r) out[r] = k.side.pts[p][r]; }
  HSQP_HD void ext_arm(int f, int a, T* ea) const { for (int r = 0; r < 4; ++r) ea[r] = k.side.ea[f][a][r]; }
};

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

// Everything one lane produces for its node, kept in arrays (the single-lane value kernels and the tree-pass form; the LQ kernel's lanes
// write their rows straight to the record through sinks of their own, hsqp_cent_lq.h).  A SINK of cent_terms has: header, fric_d1, gn
// (Gauss-Newton row: value r, weight sqrt(w)), pen (penalty row: constraint value h, penalty p), raw (a row with an explicit scale and no
// gradient / cost contribution: the friction cone's curvature rows), eq (equality row).
template <class T>
struct CentOut {
  T row[NRS];            // residual rows (unscaled: the quantity whose square / penalty is the cost term)
  double sc[NRS];        // row scale: sqrt(w) (Gauss-Newton) or sqrt(p'') (penalty); 0 = inactive
  double rho[NRS];       // sc * value (Gauss-Newton) or p' / sqrt(p'') (penalty)
  double pen_[NRS];      // cost contribution of the row: 0.5 rho^2 (Gauss-Newton) or the penalty value
  T eq_[NE_MAX];
  int ne, contact[2], eq_off[2];
  double hfric_d1[2];    // p' of the friction barrier per foot (0 if not in contact): Hessian diagonal shift
  HSQP_HD void header(int ne_, int c0, int c1, int o0, int o1) {
    ne = ne_; contact[0] = c0; contact[1] = c1; eq_off[0] = o0; eq_off[1] = o1; hfric_d1[0] = hfric_d1[1] = 0.0;
    for (int s = 0; s < NRS; ++s) { row[s] = cst<T>(0.0); sc[s] = 0.0; rho[s] = 0.0; pen_[s] = 0.0; }
    for (int r = 0; r < NE_MAX; ++r) eq_[r] = cst<T>(0.0);
  }
  HSQP_HD void fric_d1(int f, double d1) { hfric_d1[f] = d1; }
  HSQP_HD void gn(int s, const T& r, double w) { row[s] = r; sc[s] = w; rho[s] = w * val(r); pen_[s] = 0.5 * rho[s] * rho[s]; }
  HSQP_HD void pen(int s, const T& hh, const Pen3& p) {
    row[s] = hh; pen_[s] = p.p;
    if (p.d2 > 0.0) { sc[s] = sqrt(p.d2); rho[s] = p.d1 / sc[s]; }
  }
  HSQP_HD void raw(int s, const T& r, double scale) { row[s] = r; sc[s] = scale; }
  HSQP_HD void eq(int r, const T& v) { eq_[r] = v; }
};

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

// nominal state / input of the quadratic cost (StateInputQuadraticCost.cpp:67-78 with the centroidal model's accessors)
HSQP_HD void cent_nominal(const DevModel& dm, const double* x, const double* par, double* xnom, double* unom) {
  for (int i = 0; i < CNX; ++i) xnom[i] = par[HSQP_P_XDES + i];
  const double gcf = par[HSQP_P_ARMSWING] * (cos(x[9]) * xnom[0] + sin(x[9]) * xnom[1]);
  xnom[12 + dm.arm_swing_joint[0]] += -0.15 * gcf;
  xnom[12 + dm.arm_swing_joint[1]] += 0.15 * gcf;
  xnom[12 + dm.arm_swing_joint[2]] += -0.15 * gcf;
  xnom[12 + dm.arm_swing_joint[3]] += 0.15 * gcf;
  const int c0 = par[HSQP_P_CONTACT] > 0.5, c1 = par[HSQP_P_CONTACT + 1] > 0.5;
  for (int i = 0; i < NU; ++i) unom[i] = 0.0;
  if (c0 + c1 > 0) {
    const double fz = dm.total_mass * 9.81 / (c0 + c1);
    if (c0) unom[2] = fz;
    if (c1) unom[8] = fz;
  }
}

// The scalar program of one node on the number type T with tangent direction `dir` (0..69, or -1 for none) comes in two halves that
// do not depend on each other and run on different lanes: the RK4 step of the flow map (u held constant; joint rows
// q_j+ = q_j + dt qd_j exactly) and the cost / constraint terms at (x, u).
template <class T>
HSQP_HD void cent_seed(const double* x, const double* u, int dir, T* xs, T* us) {
  for (int i = 0; i < CNX; ++i) xs[i] = cst<T>(x[i]);
  for (int i = 0; i < NU; ++i) us[i] = cst<T>(u[i]);
  if (dir >= 0 && dir < CNX) set_tan(xs[dir]);
  else if (dir >= CNX && dir < CNZ) set_tan(us[dir - CNX]);
}
// RK4: x_next (35) and the flow (12 dense rows) at (x, u)
template <class T, bool WAVE_TRIG = false>
HSQP_HD void cent_rk4(const DevModel& dm, const double* x, const double* u, double dt, int dir, CentKin<T>& k, T* xn /*[CNX]*/, T* flow /*[12]*/) {
  T xs[CNX], us[NU], x0[CNX], k1[12], ks[12], acc[12];
  cent_seed<T>(x, u, dir, xs, us);
  for (int i = 0; i < CNX; ++i) x0[i] = xs[i];
  cent_pass<T, false, WAVE_TRIG>(dm, xs, xs + 6, us, us + 12, k, k1);
  for (int r = 0; r < 12; ++r) { flow[r] = k1[r]; acc[r] = k1[r]; }
  // stages 2..4: x_s = x + c k_{s-1}; the joint rows of every k are qd_j
  for (int s = 1; s < 4; ++s) {
    const double c = s == 3 ? dt : 0.5 * dt;
    const T* kp = s == 1 ? k1 : ks;
    for (int r = 0; r < 12; ++r) xs[r] = x0[r] + kp[r] * c;
    for (int j = 0; j < NJ; ++j) xs[12 + j] = x0[12 + j] + us[12 + j] * c;
    T kn[12];
    cent_pass<T, false, WAVE_TRIG>(dm, xs, xs + 6, us, us + 12, k, kn);
    const double wgt = s == 3 ? 1.0 : 2.0;
    for (int r = 0; r < 12; ++r) { ks[r] = kn[r]; acc[r] = acc[r] + kn[r] * wgt; }
  }
  for (int r = 0; r < 12; ++r) xn[r] = x0[r] + acc[r] * (dt / 6.0);
  for (int j = 0; j < NJ; ++j) xn[12 + j] = x0[12 + j] + us[12 + j] * dt;
}
// terms: one tree pass with the side tables, then every cost / constraint term
template <class T, bool WAVE_TRIG = false>
HSQP_HD void cent_terms_program(const DevModel& dm, const double* x, const double* u, const double* par, int dir, CentKin<T>& k, CentOut<T>& o) {
  T xs[CNX], us[NU], k1[12];
  cent_seed<T>(x, u, dir, xs, us);
  cent_pass<T, true, WAVE_TRIG>(dm, xs, xs + 6, us, us + 12, k, k1);
  cent_terms<T>(dm, CentPassKin<T>{k}, us, par, o);
}

// value lane of the RK4 half: defect, flow; misc[3] = dt |b|^2
template <class T>
HSQP_HD void cent_write_dynamics(const T* xn, const T* flow, const double* u, const double* xnext, double dt, double* rec, double* misc) {
  double dyn = 0.0;
  for (int i = 0; i < CNX; ++i) { const double b = val(xn[i]) - xnext[i]; dyn += b * b; if (rec) rec[REC_B + i] = b; }
  misc[3] = (dt > 0.0 ? dt : 1.0) * dyn;   // event interval (dt = 0): unscaled, as hsqp_lq.h
  if (!rec) return;
  for (int i = CNX; i < 64; ++i) rec[REC_B + i] = 0.0;
  for (int i = 0; i < 64; ++i) rec[REC_FLOW + i] = i < 12 ? val(flow[i]) : (i < CNX ? u[i] : 0.0);   // joint rows: qd_j = u[12 + (i - 12)]
}
// value lane of the terms half: cost, equality values, row data; misc[0..2, 4..7]
template <class T>
HSQP_HD void cent_write_terms(const DevModel& dm, const CentOut<T>& o, const double* x, const double* u, const double* par, double dt, double* rec,
                              double* misc) {
  double xnom[CNX], unom[NU];
  cent_nominal(dm, x, par, xnom, unom);
  const double sdt = sqrt(dt);
  double cost = 0.0;
  for (int i = 0; i < CNX; ++i) { const double d = x[i] - xnom[i]; cost += 0.5 * dm.Q[i] * d * d; }
  for (int i = 0; i < NU; ++i) { const double d = u[i] - unom[i]; cost += 0.5 * dm.R[i] * d * d; }
  for (int s = 0; s < NRS; ++s) cost += o.pen_[s];
  for (int j = 0; j < NJ; ++j)   // JointLimitsSoftConstraint.cpp:64-100
    cost += pwp_barrier(dm.jl_bmu, dm.jl_bdelta, x[12 + j] - dm.q_lo[j]).p + pwp_barrier(dm.jl_bmu, dm.jl_bdelta, dm.q_hi[j] - x[12 + j]).p;
  double eq = 0.0;
  for (int r = 0; r < o.ne; ++r) eq += val(o.eq_[r]) * val(o.eq_[r]);
  misc[0] = (double)o.ne; misc[1] = dt * cost; misc[2] = dt * eq;
  misc[4] = (double)o.contact[0]; misc[5] = (double)o.contact[1]; misc[6] = (double)o.eq_off[0]; misc[7] = (double)o.eq_off[1];
  if (!rec) return;
  rec[REC_NROWS] = (double)NRS;   // this kernel fills every row slot (no compaction)
  for (int s = 0; s < NRS; ++s) rec[REC_RHO + s] = sdt * o.rho[s];
  const double shift = -(o.hfric_d1[0] + o.hfric_d1[1]) * dm.friction_hess_shift;   // hessianDiagonalShift on every state and input
  for (int i = 0; i < LDJ; ++i) {
    double d = 0.0, g = 0.0;
    if (i < CNX) { d = dm.Q[i] + shift; g = dm.Q[i] * (x[i] - xnom[i]); }
    else if (i >= NX && i < NZ) { d = dm.R[i - NX] + shift; g = dm.R[i - NX] * (u[i - NX] - unom[i - NX]); }
    if (i >= 12 && i < CNX) {
      const int j = i - 12;
      const Pen3 lo = pwp_barrier(dm.jl_bmu, dm.jl_bdelta, x[i] - dm.q_lo[j]), hi = pwp_barrier(dm.jl_bmu, dm.jl_bdelta, dm.q_hi[j] - x[i]);
      d += lo.d2 + hi.d2; g += lo.d1 - hi.d1;
    }
    rec[REC_D + i] = dt * d;
    rec[REC_GD + i] = dt * g;
  }
  for (int r = 0; r < NE_MAX; ++r) rec[REC_CDE + r * LDJ + NZ] = r < o.ne ? val(o.eq_[r]) : 0.0;
}

// ---- the LQ kernel body: lane = tangent direction, two lane groups of 128 (group 0: RK4 -> [A|B], defect, flow; group 1: terms
//      -> residual rows, equality rows, diagonals).  Workspace: none (private memory only).
constexpr int CENT_GROUP = 128, CENT_THREADS = 2 * CENT_GROUP;
// SPLIT (the latency form, chosen for launches of a few hundred nodes: BASELINE configs 1-2 are one instance): the computing lanes are
// split evenly over the group's two waves so that both can share their sines / cosines (cent_pass<.., WAVE_TRIG>): config 1's kernel
// 0.233 -> 0.209 ms.  With thousands of nodes in flight the packed form (64 + 7 computing lanes, every lane its own trigonometry)
// touches fewer private-memory lines per wave and is 8 % faster (64 x 100 nodes: 7.0 against 7.6 ms).
template <bool SPLIT>
HSQP_HD void cent_lq_node(const Ctx& ctx, const DevModel& dm, const double* x, const double* u, const double* xnext, const double* par, double dt,
                          double* rec) {
  WG_FOR(ctx, it, CENT_THREADS) {
    // SPLIT: the 71 computing lanes (70 tangents + the value lane) are split 36 + 35 over the group's two waves so that lanes
    // 0 .. 25 of BOTH waves run the model passes (cent_pass<.., WAVE_TRIG>); the 26 zero-fill lanes sit behind the first 36
    const int grp = it / CENT_GROUP, gl = it % CENT_GROUP;
    constexpr int CL0 = 36;
    const int lane = !SPLIT ? gl : (gl < CL0 ? gl : (gl < CL0 + (CENT_LANES - CNZ - 1) ? CNZ + 1 + (gl - CL0) : (gl >= 64 && gl < 64 + (CNZ + 1 - CL0) ? CL0 + (gl - 64) : CENT_LANES)));
    if (lane >= CENT_LANES) continue;
    if (lane > CNZ) {   // zero-fill lanes: padding columns 35..57 and 93..95 of the rows of this group
      const int col = lane - CNZ - 1 < NX - CNX ? CNX + (lane - CNZ - 1) : NZ + (lane - CNZ - 1 - (NX - CNX));
      if (grp == 0) {
        for (int r = 0; r < 12; ++r) rec[REC_PV + r * LDJ + col] = 0.0;
      } else {
        for (int s = 0; s < NRS; ++s) rec[REC_J + s * LDJ + col] = 0.0;
        if (col != NZ) for (int r = 0; r < NE_MAX; ++r) rec[REC_CDE + r * LDJ + col] = 0.0;
      }
      continue;
    }
    const int dir = lane < CNZ ? lane : -1;
    const int col = lane < CNX ? lane : NX + (lane - CNX);
    CentKin<Dual1> k;
    if (grp == 0) {
      Dual1 xn[CNX], flow[12];
      cent_rk4<Dual1, SPLIT>(dm, x, u, dt, dir, k, xn, flow);
      if (lane == CNZ) { cent_write_dynamics<Dual1>(xn, flow, u, xnext, dt, rec, rec + REC_MISC); continue; }
      // [A|B] - [I|0] on the 12 dense rows (the joint rows are q_j+ = q_j + dt qd_j: structure known to the projection)
      for (int r = 0; r < 12; ++r) rec[REC_PV + r * LDJ + col] = xn[r].d - (r == lane ? 1.0 : 0.0);
    } else {
      CentOut<Dual1> o;
      cent_terms_program<Dual1, SPLIT>(dm, x, u, par, dir, k, o);
      if (lane == CNZ) { cent_write_terms<Dual1>(dm, o, x, u, par, dt, rec, rec + REC_MISC); continue; }
      const double sdt = sqrt(dt);
      for (int s = 0; s < NRS; ++s) rec[REC_J + s * LDJ + col] = sdt * o.sc[s] * o.row[s].d;
      for (int r = 0; r < NE_MAX; ++r) rec[REC_CDE + r * LDJ + col] = r < o.ne ? o.eq_[r].d : 0.0;
    }
  }
  WG_SYNC(ctx);
}

// value-only evaluation of one node (performance index / line search), in the same two halves: part 0 writes misc[3], part 1 the rest
// WAVE_TRIG: the caller runs the node on the lanes 0 .. 3 + NJ - 1 of a wave, all with the same inputs: lane a evaluates sincos of angle a
// for everybody (cent_pass); every lane writes the same results.
template <bool WAVE_TRIG = false>
HSQP_HD void cent_value_node(const DevModel& dm, const double* x, const double* u, const double* xnext, const double* par, double dt, double* misc, int part) {
  CentKin<double> k;
  if (part == 0) {
    double xn[CNX], flow[12];
    cent_rk4<double, WAVE_TRIG>(dm, x, u, dt, -1, k, xn, flow);
    cent_write_dynamics<double>(xn, flow, u, xnext, dt, nullptr, misc);
  } else {
    CentOut<double> o;
    cent_terms_program<double, WAVE_TRIG>(dm, x, u, par, -1, k, o);
    cent_write_terms<double>(dm, o, x, u, par, dt, nullptr, misc);
  }
}

// Device-side parameter generation, centroidal part (after node_params_eval has filled the desired state, contact flags, swing
// references, impact proximity and arm-swing factor of the row): the torso task-space reference = kinematics of the torso link at
// (xRef, uRef = 0) — EndEffectorKinematicsQuadraticCost::getParameters / getReferenceCostElement
// (humanoid_common_mpc/src/cost/EndEffectorKinematicsQuadraticCost.cpp:80-104) — from the same tree pass the LQ kernel uses.
HSQP_HD void cent_params_finish(const DevModel& dm, double* par) {
  CentKin<double> k;
  double xd[12], W[12], qd[NJ];
  for (int i = 0; i < 12; ++i) W[i] = 0.0;
  for (int i = 0; i < NJ; ++i) qd[i] = 0.0;
  cent_pass<double, true>(dm, par + HSQP_P_XDES, par + HSQP_P_XDES + 6, W, qd, k, xd);
  const BodyRec<double>& tb = k.side.torso;
  double Rt[9], pos[3], vl[3], va[3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Rt[3 * r + c] = tb.R[3 * r] * dm.torso_R[c] + tb.R[3 * r + 1] * dm.torso_R[3 + c] + tb.R[3 * r + 2] * dm.torso_R[6 + c];
  cent_point(k, tb, dm.torso_p, pos, vl, va);
  double* ref = par + HSQP_PC_TORSO;
  cent_quat(Rt, ref + 3);
  for (int r = 0; r < 3; ++r) { ref[r] = pos[r]; ref[7 + r] = vl[r]; ref[10 + r] = va[r]; }
  for (int i = HSQP_PC_TORSO + 13; i < NX; ++i) par[HSQP_P_XDES + i] = 0.0;
  par[HSQP_P_SWING + 2] = 0.0; par[HSQP_P_SWING + 5] = 0.0;   // the velocity-level constraints have no acceleration reference
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
