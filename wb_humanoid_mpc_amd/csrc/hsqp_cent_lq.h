// Centroidal LQ approximation, second form (round 4): values ONCE per node in an LDS workspace, tangents in closed form.
//
// The first form (rounds 1-3: a scalar program on dual numbers, one tangent direction per lane) made every lane walk the whole kinematic
// tree with its own copy of every body record: 9 KB of private memory per lane, one wave per SIMD, ~160 k wave instructions per node.  Here
//   * the model pass of a stage runs once per node, cooperatively, in workgroup phases over an LDS workspace (the whole-body kernel's
//     scheme, hsqp_model.h): one sincos per angle, the placement walk per (chain, row), joint-rate velocities per body, per-body momentum
//     sums, a 15-entry reduction;
//   * a tangent lane never walks the tree.  What a coordinate q_c changes is a RIGID ROTATION of the subtree of joint c about its axis:
//         dR_i = [w_c]x R_i,  dr_i = w_c x (r_i - r_c),  d om_i = w_c x (om_i - om_c),  d v_i = w_c x (v_i - v_c) + (om_c x w_c) x (r_i - r_c)
//     for the bodies i of the subtree (a contiguous range of the depth-first order), nothing else; a joint rate qd_c adds w_c to their
//     angular and w_c x (r_i - r_c) to their origin velocities; an euler angle rotates the whole robot about the base origin, so the
//     momentum sums themselves rotate (no loop); h, W and the base position enter the closing 3 x 3 solve only.  The lane turns these
//     seeds into the tangents of the momentum sums by the product rule on the per-body values of the stage (1 .. 11 bodies instead of 24),
//     then runs the closing solve (hsqp_cent.h: cent_finish) and the terms (cent_terms) on one-tangent dual numbers — the arithmetic that
//     was verified against the oracle — without any per-lane array;
//   * the four stage Jacobians (12 x 70 each) stay in LDS and are chained per column at the end (12 x 12 blocks), as the whole-body kernel
//     chains its 6 x 6 blocks;
//   * the cost / constraint rows are produced by the same lanes right behind their stage-1 column (the closing solve's base velocity
//     tangent is what the velocity rows need) and go straight to the record through a sink — no row arrays.
// 128 threads per node: x columns on lanes 0 .. 34 of wave 0, u columns on lanes 0 .. 34 of wave 1 (q_j and qd_j — the lanes with a
// subtree loop — on different waves), value items behind them.
#pragma once
#include "hsqp_cent.h"

namespace hsqp {

constexpr int CLQ_THREADS = 128;
constexpr int CLQ_LDG = 70;            // leading dimension of a stage Jacobian (70 columns: x 0 .. 34, u 35 .. 69)

template <bool D>
struct CentWST {
  unsigned char anc[NB][NANC], n_anc[NB], chain_start[NB], chain_len[NB], sub[NB];
  int n_chains;
  double x[CNX], u[NU], par[NP], xnext[CNX];
  double xs[12];                       // [h/m ; p_b ; euler] of the stage
  double q[6 + NJ];                    // [p_b ; euler ; q_j] of the stage (angle a = q[3 + a], as the whole-body workspace)
  double ecs[3][2];
  double E[9];                         // E[3 r + e]: world axis of euler rate e
  union {
    struct { double Mq[NB + 1][9], pa[NB + 1][6]; };   // joint rotations / offsets and axes in the parent frame: dead after the placement walk
    double part[NB][16];               // per-body momentum sums (CentSums order: mc, lin, angO, IO)
  };
  double R[NB + 1][9], r[NB + 1][3], w[NB + 1][3];   // world rotation, origin relative to the base origin, world joint axis (row NB: dump row)
  double om[NB][3], vo[NB][3];         // angular / origin velocity relative to the base (joint rates only)
  double sums[16];                     // totals of the per-body sums
  double bc[NB][3], bvc[NB][3], bIw[NB][6];   // per body: centre of mass (relative to the base origin), its velocity, rotational inertia in world axes
  double pc[2][3];                     // contact points relative to the base origin
  double kv[4][12];                    // stage values of the 12 dense rows of xdot
  double xnom[CNX], unom[NU], gcf;     // nominal state / input of the quadratic cost (gcf: the arm-swing term's factor)
  double tv[8];                        // from the terms' value lane: cost of the rows, equality SSE, friction p' sum, ne, contact flags, row offsets
  double G[D ? 4 : 1][12][D ? CLQ_LDG : 1];
};

// column kinds of a tangent lane
constexpr int CK_NONE = 0, CK_H = 1, CK_P = 2, CK_EUL = 3, CK_Q = 4, CK_W = 5, CK_QD = 6;
struct CentCol { int kind, idx; };     // idx: component (H, P, EUL, W) or joint (Q, QD)
HSQP_HD CentCol cent_col_x(int c) { return c < 6 ? CentCol{CK_H, c} : (c < 9 ? CentCol{CK_P, c - 6} : (c < 12 ? CentCol{CK_EUL, c - 9} : CentCol{CK_Q, c - 12})); }
HSQP_HD CentCol cent_col_u(int c) { return c < 12 ? CentCol{CK_W, c} : CentCol{CK_QD, c - 12}; }

template <class T> HSQP_HD T cent_mk(double v, double d);
template <> HSQP_HD Dual1 cent_mk<Dual1>(double v, double d) { return mk(v, d); }
template <> HSQP_HD double cent_mk<double>(double v, double) { return v; }

// Body record of body i as the tangent lane `col` sees it: values from the workspace, tangent from the rigid-motion rules above.
// p0 (optional): absolute base position — the cost / constraint terms need absolute heights; the flow map runs on relative positions.
template <class T, class WS>
HSQP_HD BodyRec<T> cent_seed_body(const WS& ws, int i, CentCol col, const double* p0 = nullptr) {
  BodyRec<T> b;
  double dR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dp[3] = {0, 0, 0}, dom[3] = {0, 0, 0}, dv[3] = {0, 0, 0};
  if constexpr (!std::is_same<T, double>::value) {
    const int c = col.idx + 1;
    const bool in_sub = (col.kind == CK_Q || col.kind == CK_QD) && i >= c && i < c + ws.sub[c];
    if (col.kind == CK_EUL || (col.kind == CK_Q && in_sub)) {
      const bool eul = col.kind == CK_EUL;
      double wa[3], d[3], rel_om[3], rel_v[3];
      for (int k = 0; k < 3; ++k) {
        wa[k] = eul ? ws.E[3 * k + col.idx] : ws.w[c][k];
        d[k] = ws.r[i][k] - (eul ? 0.0 : ws.r[c][k]);
        rel_om[k] = ws.om[i][k] - (eul ? 0.0 : ws.om[c][k]);
        rel_v[k] = ws.vo[i][k] - (eul ? 0.0 : ws.vo[c][k]);
      }
      for (int cc = 0; cc < 3; ++cc) {
        const double colv[3] = {ws.R[i][cc], ws.R[i][3 + cc], ws.R[i][6 + cc]};
        double t[3];
        v3_cross(wa, colv, t);
        dR[cc] = t[0]; dR[3 + cc] = t[1]; dR[6 + cc] = t[2];
      }
      v3_cross(wa, d, dp);
      v3_cross(wa, rel_om, dom);
      v3_cross(wa, rel_v, dv);
      if (!eul) {   // the parent's rotation does not turn with the subtree: (om_c x w_c) x d
        double ow[3], t[3];
        v3_cross(ws.om[c], wa, ow);
        v3_cross(ow, d, t);
        for (int k = 0; k < 3; ++k) dv[k] += t[k];
      }
    } else if (col.kind == CK_QD && in_sub) {
      double d[3];
      for (int k = 0; k < 3; ++k) { d[k] = ws.r[i][k] - ws.r[c][k]; dom[k] = ws.w[c][k]; }
      v3_cross(ws.w[c], d, dv);
    } else if (col.kind == CK_P && p0) {
      dp[col.idx] = 1.0;
    }
  }
  for (int k = 0; k < 9; ++k) b.R[k] = cent_mk<T>(ws.R[i][k], dR[k]);
  for (int k = 0; k < 3; ++k) {
    b.p[k] = cent_mk<T>(ws.r[i][k] + (p0 ? p0[k] : 0.0), dp[k]);
    b.om[k] = cent_mk<T>(ws.om[i][k], dom[k]);
    b.v[k] = cent_mk<T>(ws.vo[i][k], dv[k]);
  }
  return b;
}
// world axis of joint i (body i) as the lane sees it
template <class T, class WS>
HSQP_HD void cent_seed_axis(const WS& ws, int i, CentCol col, T* w) {
  double dw[3] = {0, 0, 0};
  if constexpr (!std::is_same<T, double>::value) {
    const int c = col.idx + 1;
    if (col.kind == CK_EUL) { const double wa[3] = {ws.E[col.idx], ws.E[3 + col.idx], ws.E[6 + col.idx]}; v3_cross(wa, ws.w[i], dw); }
    else if (col.kind == CK_Q && i >= c && i < c + ws.sub[c]) v3_cross(ws.w[c], ws.w[i], dw);
  }
  for (int k = 0; k < 3; ++k) w[k] = cent_mk<T>(ws.w[i][k], dw[k]);
}
// a point fixed to body `body` (local coordinates pl), relative to the base origin (+ p0)
template <class T, class WS>
HSQP_HD void cent_seed_point(const WS& ws, int body, const double* pl, CentCol col, const double* p0, T* out) {
  const BodyRec<T> b = cent_seed_body<T>(ws, body, col, p0);
  T rp[3];
  t_mulc(b.R, pl, rp);
  for (int k = 0; k < 3; ++k) out[k] = b.p[k] + rp[k];
}

// The momentum sums of the stage as lane `col` sees them: total values from the workspace, tangents from the bodies the lane's
// coordinate moves (closed form for an euler angle: the sums about the base origin rotate with the robot).
template <class T, class WS>
HSQP_HD void cent_lane_sums(const DevModel& dm, const WS& ws, CentCol col, CentSums<T>& s) {
  double d[15];
  for (int e = 0; e < 15; ++e) d[e] = 0.0;
  if constexpr (!std::is_same<T, double>::value) {
    if (col.kind == CK_EUL) {
      const double wa[3] = {ws.E[col.idx], ws.E[3 + col.idx], ws.E[6 + col.idx]};
      v3_cross(wa, ws.sums + 0, d + 0);
      v3_cross(wa, ws.sums + 3, d + 3);
      v3_cross(wa, ws.sums + 6, d + 6);
      // d IO = [w]x IO - IO [w]x = WI + WI^T with WI = [w]x IO (IO symmetric, [w]x antisymmetric)
      const double* I6 = ws.sums + 9;
      const double I[9] = {I6[0], I6[1], I6[2], I6[1], I6[3], I6[4], I6[2], I6[4], I6[5]};
      double WI[9];
      for (int cc = 0; cc < 3; ++cc) {
        const double colv[3] = {I[cc], I[3 + cc], I[6 + cc]};
        double t[3];
        v3_cross(wa, colv, t);
        WI[cc] = t[0]; WI[3 + cc] = t[1]; WI[6 + cc] = t[2];
      }
      d[9] = 2.0 * WI[0]; d[10] = WI[1] + WI[3]; d[11] = WI[2] + WI[6]; d[12] = 2.0 * WI[4]; d[13] = WI[5] + WI[7]; d[14] = 2.0 * WI[8];
    } else if (col.kind == CK_Q || col.kind == CK_QD) {
      const int c = col.idx + 1, n = ws.sub[c];
      // the tangents of the per-body sums under the lane's rigid motion of the subtree, from the per-body values of the
      // stage (product rule by hand: a quarter of the instructions of the generic dual-number accumulation, and no value is recomputed)
      const double* wa = ws.w[c];
      const double* a0 = ws.r[c];
      double owc[3];
      v3_cross(ws.om[c], wa, owc);
      for (int i = c; i < c + n; ++i) {
        const double m = dm.mass[i];
        const double* ci = ws.bc[i];
        const double* vci = ws.bvc[i];
        const double* I6 = ws.bIw[i];
        double ca[3], dvc[3], dc[3] = {0.0, 0.0, 0.0}, dIo[3] = {0.0, 0.0, 0.0}, dIw[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, dom[3];
        for (int k = 0; k < 3; ++k) ca[k] = ci[k] - a0[k];
        if (col.kind == CK_QD) {
          for (int k = 0; k < 3; ++k) dom[k] = wa[k];
          v3_cross(wa, ca, dvc);
        } else {
          double rel[3], ra[3], rc[3], t1[3], t2[3], t3[3], t4[3], wrc[3];
          v3_cross(wa, ca, dc);
          for (int k = 0; k < 3; ++k) { rel[k] = ws.om[i][k] - ws.om[c][k]; ra[k] = ws.r[i][k] - a0[k]; rc[k] = ci[k] - ws.r[i][k]; }
          v3_cross(wa, rel, dom);
          for (int k = 0; k < 3; ++k) rel[k] = ws.vo[i][k] - ws.vo[c][k];
          v3_cross(wa, rel, t1);
          v3_cross(owc, ra, t2);
          v3_cross(dom, rc, t3);
          v3_cross(wa, rc, wrc);
          v3_cross(ws.om[i], wrc, t4);
          for (int k = 0; k < 3; ++k) dvc[k] = t1[k] + t2[k] + t3[k] + t4[k];
          // d Iw = [w]x Iw - Iw [w]x = WI + WI^T
          const double I[9] = {I6[0], I6[1], I6[2], I6[1], I6[3], I6[4], I6[2], I6[4], I6[5]};
          double WI[9];
          for (int cc = 0; cc < 3; ++cc) {
            const double colv[3] = {I[cc], I[3 + cc], I[6 + cc]};
            double t[3];
            v3_cross(wa, colv, t);
            WI[cc] = t[0]; WI[3 + cc] = t[1]; WI[6 + cc] = t[2];
          }
          dIw[0] = 2.0 * WI[0]; dIw[1] = WI[1] + WI[3]; dIw[2] = WI[2] + WI[6]; dIw[3] = 2.0 * WI[4]; dIw[4] = WI[5] + WI[7]; dIw[5] = 2.0 * WI[8];
          sym3_mulv(dIw, ws.om[i], dIo);
        }
        double Idom[3], mvc[3], mdvc[3], x1[3], x2[3];
        sym3_mulv(I6, dom, Idom);
        for (int k = 0; k < 3; ++k) { mvc[k] = m * vci[k]; mdvc[k] = m * dvc[k]; }
        v3_cross(dc, mvc, x1);
        v3_cross(ci, mdvc, x2);
        const double cdc = 2.0 * v3_dot(ci, dc);
        for (int k = 0; k < 3; ++k) { d[k] += m * dc[k]; d[3 + k] += mdvc[k]; d[6 + k] += dIo[k] + Idom[k] + x1[k] + x2[k]; }
        d[9] += dIw[0] + m * (cdc - 2.0 * ci[0] * dc[0]); d[10] += dIw[1] - m * (ci[0] * dc[1] + dc[0] * ci[1]); d[11] += dIw[2] - m * (ci[0] * dc[2] + dc[0] * ci[2]);
        d[12] += dIw[3] + m * (cdc - 2.0 * ci[1] * dc[1]); d[13] += dIw[4] - m * (ci[1] * dc[2] + dc[1] * ci[2]); d[14] += dIw[5] + m * (cdc - 2.0 * ci[2] * dc[2]);
      }
    }
  }
  for (int r = 0; r < 3; ++r) { s.mc[r] = cent_mk<T>(ws.sums[r], d[r]); s.lin[r] = cent_mk<T>(ws.sums[3 + r], d[3 + r]); s.angO[r] = cent_mk<T>(ws.sums[6 + r], d[6 + r]); }
  for (int r = 0; r < 6; ++r) s.IO[r] = cent_mk<T>(ws.sums[9 + r], d[9 + r]);
}

// What the closing solve leaves for the terms: base position, [pdot; euler rates], angular velocity of the base
template <class T> struct CentBase { T p0[3], vb[6], wb[3]; };

// The stage's 12 dense rows of xdot for lane `col` (T = Dual1) or their values (T = double, col.kind = CK_NONE)
template <class T, class WS>
HSQP_HD void cent_lane_flow(const DevModel& dm, const WS& ws, CentCol col, T* xdot, CentBase<T>& base) {
  CentSums<T> s;
  cent_lane_sums<T>(dm, ws, col, s);
  T E[9], pc[2][3], h[6], W[12];
  for (int k = 0; k < 9; ++k) E[k] = cst<T>(ws.E[k]);
  for (int k = 0; k < 6; ++k) h[k] = cent_mk<T>(ws.xs[k], (col.kind == CK_H && col.idx == k) ? 1.0 : 0.0);
  for (int k = 0; k < 12; ++k) W[k] = cent_mk<T>(ws.u[k], (col.kind == CK_W && col.idx == k) ? 1.0 : 0.0);
  for (int f = 0; f < 2; ++f)
    for (int k = 0; k < 3; ++k) pc[f][k] = cst<T>(ws.pc[f][k]);
  if constexpr (!std::is_same<T, double>::value) {
    if (col.kind == CK_EUL) {
      const double wa[3] = {ws.E[col.idx], ws.E[3 + col.idx], ws.E[6 + col.idx]};
      for (int e = col.idx + 1; e < 3; ++e) {   // the axes of the later euler joints turn with this one
        const double ax[3] = {ws.E[e], ws.E[3 + e], ws.E[6 + e]};
        double t[3];
        v3_cross(wa, ax, t);
        for (int k = 0; k < 3; ++k) E[3 * k + e] = cent_mk<T>(ws.E[3 * k + e], t[k]);
      }
      for (int f = 0; f < 2; ++f) { double t[3]; v3_cross(wa, ws.pc[f], t); for (int k = 0; k < 3; ++k) pc[f][k] = cent_mk<T>(ws.pc[f][k], t[k]); }
    } else if (col.kind == CK_Q) {
      const int c = col.idx + 1;
      for (int f = 0; f < 2; ++f) {
        const int cb = dm.contact_body[f];
        if (cb >= c && cb < c + ws.sub[c]) {
          double d[3], t[3];
          for (int k = 0; k < 3; ++k) d[k] = ws.pc[f][k] - ws.r[c][k];
          v3_cross(ws.w[c], d, t);
          for (int k = 0; k < 3; ++k) pc[f][k] = cent_mk<T>(ws.pc[f][k], t[k]);
        }
      }
    }
  }
  T com[3];
  cent_finish<T>(dm, s, E, pc, h, W, xdot, base.vb, base.wb, com);
}

// ---- terms: the kinematics cent_terms asks for, produced on demand from the workspace (stage 1) with the lane's seeds
template <class T, class WS>
struct CentLaneKin {
  const DevModel& dm;
  const WS& ws;
  CentCol col;
  CentBase<T> base;
  T Wd[12];                       // the node's wrenches as the lane sees them
  HSQP_HD const T* p0() const { return base.p0; }
  HSQP_HD const T* vb() const { return base.vb; }
  HSQP_HD const T* wb() const { return base.wb; }
  HSQP_HD BodyRec<T> foot(int f) const { return cent_seed_body<T>(ws, dm.contact_body[f], col, ws.x + 6); }
  HSQP_HD BodyRec<T> torso() const { return cent_seed_body<T>(ws, dm.torso_body, col, ws.x + 6); }
  HSQP_HD void point(int p, T* out) const { cent_seed_point<T>(ws, dm.coll_body[p], dm.coll_p[p], col, ws.x + 6, out); }
  // external-torque joint a of foot f: tau = w_j . (m + (pos - p_j) x f) = pos . (f x w_j) + (w_j . m - p_j . (f x w_j)) = pos . ea[0..2] + ea[3];
  // zero unless joint j carries the foot
  HSQP_HD void ext_arm(int f, int a, T* ea) const {
    const int i = 1 + dm.ext_joint[f][a], cb = dm.contact_body[f];
    const bool carries = cb >= i && cb < i + ws.sub[i];
    if (!carries) { for (int k = 0; k < 4; ++k) ea[k] = cst<T>(0.0); return; }
    T w[3], fxw[3];
    cent_seed_axis<T>(ws, i, col, w);
    const BodyRec<T> b = cent_seed_body<T>(ws, i, col, ws.x + 6);
    t_cross(Wd + 6 * f, w, fxw);
    for (int k = 0; k < 3; ++k) ea[k] = fxw[k];
    ea[3] = t_dot(w, Wd + 6 * f + 3) - t_dot(b.p, fxw);
  }
};

// sink of a tangent lane: the row's tangent, scaled, straight into the record (column `rcol` of REC_J / REC_CDE)
struct CentTangentSink {
  double* rec;
  int rcol;
  double sdt;
  unsigned long long seen;   // residual rows written (the others are zero-filled at the end)
  int ne;
  HSQP_HD void header(int ne_, int, int, int, int) { ne = ne_; }
  HSQP_HD void fric_d1(int, double) {}
  HSQP_HD void row(int s, const Dual1& r, double sc) { rec[REC_J + s * LDJ + rcol] = sdt * sc * r.d; seen |= 1ull << s; }
  HSQP_HD void gn(int s, const Dual1& r, double w) { row(s, r, w); }
  HSQP_HD void pen(int s, const Dual1& h, const Pen3& p) { row(s, h, p.d2 > 0.0 ? sqrt(p.d2) : 0.0); }
  HSQP_HD void raw(int s, const Dual1& r, double sc) { row(s, r, sc); }
  HSQP_HD void eq(int r, const Dual1& v) { rec[REC_CDE + r * LDJ + rcol] = v.d; }
  HSQP_HD void finish() {
    for (int s = 0; s < NRS; ++s) if (!((seen >> s) & 1ull)) rec[REC_J + s * LDJ + rcol] = 0.0;
    for (int r = ne; r < NE_MAX; ++r) rec[REC_CDE + r * LDJ + rcol] = 0.0;
  }
};
// sink of the value lane: rho, equality values, cost — scalars only
struct CentValueSink {
  double* rec;               // may be null (value-only pass)
  double sdt;
  double cost, eqsse, shift_d1;
  unsigned long long seen;
  int ne, contact[2], eq_off[2];
  HSQP_HD void header(int ne_, int c0, int c1, int o0, int o1) { ne = ne_; contact[0] = c0; contact[1] = c1; eq_off[0] = o0; eq_off[1] = o1; }
  HSQP_HD void fric_d1(int, double d1) { shift_d1 += d1; }
  HSQP_HD void put(int s, double rho, double pen) { cost += pen; if (rec) rec[REC_RHO + s] = sdt * rho; seen |= 1ull << s; }
  HSQP_HD void gn(int s, double r, double w) { const double rho = w * r; put(s, rho, 0.5 * rho * rho); }
  HSQP_HD void pen(int s, double, const Pen3& p) { put(s, p.d2 > 0.0 ? p.d1 / sqrt(p.d2) : 0.0, p.p); }
  HSQP_HD void raw(int s, double, double) { put(s, 0.0, 0.0); }
  HSQP_HD void eq(int r, double v) { eqsse += v * v; if (rec) rec[REC_CDE + r * LDJ + NZ] = v; }
  HSQP_HD void finish() {
    if (!rec) return;
    for (int s = 0; s < NRS; ++s) if (!((seen >> s) & 1ull)) rec[REC_RHO + s] = 0.0;
    for (int r = ne; r < NE_MAX; ++r) rec[REC_CDE + r * LDJ + NZ] = 0.0;
  }
};

// entry i (0 .. LDJ - 1) of the diagonal part of the cost model: Hessian / gradient diagonals -> record (x dt), its cost -> returned
// (StateInputQuadraticCost on (x - x_nom, u - u_nom), JointLimitsSoftConstraint.cpp:64-100, the friction cone's hessianDiagonalShift)
template <class W>
HSQP_HD double cent_diag_entry(const DevModel& dm, const W& ws, int i, double dt, double* rec) {
  const double shift = -ws.tv[2] * dm.friction_hess_shift;   // on every state and input
  double d = 0.0, g = 0.0, cost = 0.0;
  if (i < CNX) { const double dx = ws.x[i] - ws.xnom[i]; d = dm.Q[i] + shift; g = dm.Q[i] * dx; cost = 0.5 * dm.Q[i] * dx * dx; }
  else if (i >= NX && i < NZ) { const double du = ws.u[i - NX] - ws.unom[i - NX]; d = dm.R[i - NX] + shift; g = dm.R[i - NX] * du; cost = 0.5 * dm.R[i - NX] * du * du; }
  if (i >= 12 && i < CNX) {
    const int j = i - 12;
    const Pen3 lo = pwp_barrier(dm.jl_bmu, dm.jl_bdelta, ws.x[i] - dm.q_lo[j]), hi = pwp_barrier(dm.jl_bmu, dm.jl_bdelta, dm.q_hi[j] - ws.x[i]);
    d += lo.d2 + hi.d2; g += lo.d1 - hi.d1; cost += lo.p + hi.p;
  }
  if (rec) { rec[REC_D + i] = dt * d; rec[REC_GD + i] = dt * g; }
  return cost;
}

template <class W>
HSQP_HD void cent_ws_topology(const Ctx& ctx, const DevModel& dm, W& ws) {
  WG_FOR(ctx, i, NB * NANC + NB) {
    if (i < NB * NANC) ws.anc[i / NANC][i % NANC] = dm.anc[i / NANC][i % NANC];
    else {
      const int b = i - NB * NANC;
      ws.n_anc[b] = (unsigned char)dm.n_anc[b]; ws.chain_start[b] = (unsigned char)dm.chain_start[b]; ws.chain_len[b] = (unsigned char)dm.chain_len[b];
      ws.sub[b] = (unsigned char)dm.subtree_size[b];
      if (b == 0) ws.n_chains = dm.n_chains;
    }
  }
}

// Values of RK4 stage s (0..3) into the workspace: placements, joint-rate velocities, momentum sums, contact points.
template <class W>
HSQP_HD void cent_stage_values(const Ctx& ctx, const DevModel& dm, W& ws, int s, double dt) {
  const double c = s == 0 ? 0.0 : (s == 3 ? dt : 0.5 * dt);
  WG_FOR(ctx, i, 12 + NJ) {
    if (i < 12) { const double v = ws.x[i] + (s == 0 ? 0.0 : c * ws.kv[s - 1][i]); ws.xs[i] = v; if (i >= 6) ws.q[i - 6] = v; }
    else ws.q[6 + (i - 12)] = ws.x[i] + c * ws.u[i];        // q_j + c qd_j  (u[12 + j] = qd_j sits at the same index as x[12 + j] = q_j)
  }
  WG_SYNC(ctx);
  // ---- trigonometry: one sincos per angle (hsqp_model.h, phase F0)
  WG_FOR(ctx, it, NB + 3) {
    if (it == NB + 2) {
      for (int k = 0; k < 9; ++k) ws.Mq[NB][k] = (k % 4 == 0) ? 1.0 : 0.0;
      for (int k = 0; k < 6; ++k) ws.pa[NB][k] = 0.0;
      continue;
    }
    double sn, cs;
    sincos(ws.q[3 + it], &sn, &cs);
    if (it < 3) { ws.ecs[it][0] = cs; ws.ecs[it][1] = sn; continue; }
    const int i = it - 2;
    double Rq[9];
    rot_axis_cs(dm.axis[i], cs, sn, Rq);
    m3_mul(dm.Rfix[i], Rq, ws.Mq[i]);
    for (int k = 0; k < 3; ++k) { ws.pa[i][k] = dm.pfix[i][k]; ws.pa[i][3 + k] = dm.axis_p[i][k]; }
  }
  WG_SYNC(ctx);
  // ---- placement walk per (chain, row) (hsqp_model.h, phase F1) + the euler axes
  WG_FOR(ctx, it, ws.n_chains * 3 + 3 + 1) {
    const double cz = ws.ecs[0][0], sz = ws.ecs[0][1], cy = ws.ecs[1][0], sy = ws.ecs[1][1], cx = ws.ecs[2][0], sx = ws.ecs[2][1];
    if (it == ws.n_chains * 3 + 3) {
      const double wz[3] = {0.0, 0.0, 1.0}, wy[3] = {-sz, cz, 0.0}, wx[3] = {cz * cy, sz * cy, -sy};
      for (int r = 0; r < 3; ++r) { ws.E[3 * r] = wz[r]; ws.E[3 * r + 1] = wy[r]; ws.E[3 * r + 2] = wx[r]; }
      continue;
    }
    const int r = it % 3, ch = it / 3;
    double Rp[3];
    if (r == 0) { Rp[0] = cz * cy; Rp[1] = cz * sy * sx - sz * cx; Rp[2] = cz * sy * cx + sz * sx; }
    else if (r == 1) { Rp[0] = sz * cy; Rp[1] = sz * sy * sx + cz * cx; Rp[2] = sz * sy * cx - cz * sx; }
    else { Rp[0] = -sy; Rp[1] = cy * sx; Rp[2] = cy * cx; }
    if (ch == ws.n_chains) {
      for (int cc = 0; cc < 3; ++cc) ws.R[0][3 * r + cc] = Rp[cc];
      ws.r[0][r] = 0.0; ws.w[0][r] = 0.0;
      continue;
    }
    const int b0 = ws.chain_start[ch], end = b0 + ws.chain_len[ch] - 1, na = ws.n_anc[end];
    double rp = 0.0;
    const unsigned long long pk = anc_packed(ws.anc[end]);
#pragma unroll
    for (int n = 0; n < NANC; ++n) {
      const int ia = anc_at(pk, n);
      const int i = n < na ? ia : NB;
      const int d = (n < na && ia >= b0) ? ia : NB;
      const double* M = ws.Mq[i];
      const double* pa = ws.pa[i];
      const double rn0 = Rp[0] * M[0] + Rp[1] * M[3] + Rp[2] * M[6];
      const double rn1 = Rp[0] * M[1] + Rp[1] * M[4] + Rp[2] * M[7];
      const double rn2 = Rp[0] * M[2] + Rp[1] * M[5] + Rp[2] * M[8];
      const double rr = rp + Rp[0] * pa[0] + Rp[1] * pa[1] + Rp[2] * pa[2];
      const double wv = Rp[0] * pa[3] + Rp[1] * pa[4] + Rp[2] * pa[5];
      ws.R[d][3 * r] = rn0; ws.R[d][3 * r + 1] = rn1; ws.R[d][3 * r + 2] = rn2; ws.r[d][r] = rr; ws.w[d][r] = wv;
      Rp[0] = rn0; Rp[1] = rn1; Rp[2] = rn2; rp = rr;
    }
  }
  WG_SYNC(ctx);
  // ---- velocities relative to the base from the joint rates (one item per body: sums over its ancestor path); contact points
  WG_FOR(ctx, it, NB + 2) {
    if (it >= NB) {
      const int f = it - NB, b = dm.contact_body[f];
      double t[3];
      m3_mulv(ws.R[b], dm.contact_p[f], t);
      for (int k = 0; k < 3; ++k) ws.pc[f][k] = ws.r[b][k] + t[k];
      continue;
    }
    const int i = it, na = ws.n_anc[i];
    double om[3] = {0.0, 0.0, 0.0}, vo[3] = {0.0, 0.0, 0.0};
    const unsigned long long pk = anc_packed(ws.anc[i]);
#pragma unroll
    for (int n = 0; n < NANC; ++n) {
      const int a = n < na ? anc_at(pk, n) : 0;
      const double qa = (n < na && a > 0) ? ws.u[12 + a - 1] : 0.0;
      double d[3], t[3];
      for (int k = 0; k < 3; ++k) d[k] = ws.r[i][k] - ws.r[a][k];
      v3_cross(ws.w[a], d, t);
      for (int k = 0; k < 3; ++k) { om[k] += qa * ws.w[a][k]; vo[k] += qa * t[k]; }
    }
    for (int k = 0; k < 3; ++k) { ws.om[i][k] = om[k]; ws.vo[i][k] = vo[k]; }
  }
  WG_SYNC(ctx);
  // ---- per-body momentum sums, then their totals
  WG_FOR(ctx, i, NB) {
    // centre of mass, its velocity, rotational inertia in world axes (kept for the tangent lanes), and the body's share of the sums
    const double m = dm.mass[i];
    const double* Rb = ws.R[i];
    double rc[3], c[3], t[3], vc[3];
    m3_mulv(Rb, dm.com[i], rc);
    for (int r = 0; r < 3; ++r) c[r] = ws.r[i][r] + rc[r];
    v3_cross(ws.om[i], rc, t);
    for (int r = 0; r < 3; ++r) vc[r] = ws.vo[i][r] + t[r];
    double RI[9], Iw[6];
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) RI[3 * r + cc] = Rb[3 * r] * dm.inertia[i][cc] + Rb[3 * r + 1] * dm.inertia[i][3 + cc] + Rb[3 * r + 2] * dm.inertia[i][6 + cc];
    {
      int n = 0;
      for (int r = 0; r < 3; ++r)
        for (int cc = r; cc < 3; ++cc) Iw[n++] = RI[3 * r] * Rb[3 * cc] + RI[3 * r + 1] * Rb[3 * cc + 1] + RI[3 * r + 2] * Rb[3 * cc + 2];
    }
    for (int r = 0; r < 3; ++r) { ws.bc[i][r] = c[r]; ws.bvc[i][r] = vc[r]; }
    for (int r = 0; r < 6; ++r) ws.bIw[i][r] = Iw[r];
    double* p = ws.part[i];
    const double c2 = v3_dot(c, c);
    p[9] = Iw[0] + (c2 - c[0] * c[0]) * m; p[10] = Iw[1] - (c[0] * c[1]) * m; p[11] = Iw[2] - (c[0] * c[2]) * m;
    p[12] = Iw[3] + (c2 - c[1] * c[1]) * m; p[13] = Iw[4] - (c[1] * c[2]) * m; p[14] = Iw[5] + (c2 - c[2] * c[2]) * m;
    double mv[3], cxmv[3], Io[3];
    for (int r = 0; r < 3; ++r) { mv[r] = vc[r] * m; p[r] = c[r] * m; p[3 + r] = mv[r]; }
    v3_cross(c, mv, cxmv);
    sym3_mulv(Iw, ws.om[i], Io);
    for (int r = 0; r < 3; ++r) p[6 + r] = Io[r] + cxmv[r];
  }
  WG_SYNC(ctx);
  WG_FOR(ctx, e, 15) {
    double sacc = 0.0;
    for (int i = 0; i < NB; ++i) sacc += ws.part[i][e];
    ws.sums[e] = sacc;
  }
  WG_SYNC(ctx);
}

// Full LQ data of one centroidal node -> record (DERIV) / performance terms only (!DERIV: rec may be null, misc = 8 doubles).
template <bool DERIV>
HSQP_HD void cent_lq_node2(const Ctx& ctx, const DevModel& dm, CentWST<DERIV>& ws, const double* x, const double* u, const double* xnext, const double* par,
                           double dt, double* rec, double* misc) {
  cent_ws_topology(ctx, dm, ws);
  WG_FOR(ctx, i, CNX + NU + NP + CNX) {
    if (i < CNX) ws.x[i] = x[i];
    else if (i < CNX + NU) ws.u[i - CNX] = u[i - CNX];
    else if (i < CNX + NU + NP) ws.par[i - CNX - NU] = par[i - CNX - NU];
    else ws.xnext[i - CNX - NU - NP] = xnext[i - CNX - NU - NP];
  }
  WG_FOR(ctx, it, 65) {   // (item 64: a wave of its own next to the loads)
    if (it != 64) continue;
    double sn, cs;
    sincos(x[9], &sn, &cs);
    ws.gcf = par[HSQP_P_ARMSWING] * (cs * par[HSQP_P_XDES] + sn * par[HSQP_P_XDES + 1]);
  }
  WG_SYNC(ctx);
  // nominal state / input (cent_nominal, one entry per item)
  WG_FOR(ctx, i, CNX + NU) {
    if (i < CNX) {
      double v = ws.par[HSQP_P_XDES + i];
      const int j = i - 12;
      if (j == dm.arm_swing_joint[0] || j == dm.arm_swing_joint[2]) v += -0.15 * ws.gcf;
      if (j == dm.arm_swing_joint[1] || j == dm.arm_swing_joint[3]) v += 0.15 * ws.gcf;
      ws.xnom[i] = v;
    } else {
      const int k = i - CNX, c0 = ws.par[HSQP_P_CONTACT] > 0.5, c1 = ws.par[HSQP_P_CONTACT + 1] > 0.5;
      ws.unom[k] = ((k == 2 && c0) || (k == 8 && c1)) ? dm.total_mass * 9.81 / (c0 + c1) : 0.0;
    }
  }
  const double sdt = sqrt(dt);
  for (int s = 0; s < 4; ++s) {
    cent_stage_values(ctx, dm, ws, s, dt);
    // ---- columns of the stage Jacobian (and, at stage 1, of every cost / constraint row), the stage's values, the terms' values
    WG_FOR(ctx, it, CLQ_THREADS) {
      const bool xcol = it < CNX, ucol = it >= 64 && it < 64 + NU;
      if (DERIV && (xcol || ucol)) {
        const CentCol col = xcol ? cent_col_x(it) : cent_col_u(it - 64);
        const int gcol = xcol ? it : CNX + (it - 64);
        CentLaneKin<Dual1, CentWST<DERIV>> kin{dm, ws, col, {}, {}};
        Dual1 xdot[12];
        cent_lane_flow<Dual1>(dm, ws, col, xdot, kin.base);
        for (int r = 0; r < 12; ++r) ws.G[DERIV ? s : 0][r][DERIV ? gcol : 0] = xdot[r].d;
        if (s == 0) {
          for (int k = 0; k < 3; ++k) kin.base.p0[k] = mk(ws.x[6 + k], (col.kind == CK_P && col.idx == k) ? 1.0 : 0.0);
          for (int k = 0; k < 12; ++k) kin.Wd[k] = mk(ws.u[k], (col.kind == CK_W && col.idx == k) ? 1.0 : 0.0);
          CentTangentSink sink{rec, xcol ? it : NX + (it - 64), sdt, 0ull, 0};
          cent_terms<Dual1>(dm, kin, kin.Wd, ws.par, sink);
          sink.finish();
        }
      } else if (it == CNX) {          // values of the stage
        CentBase<double> base;
        double xdot[12];
        cent_lane_flow<double>(dm, ws, CentCol{CK_NONE, 0}, xdot, base);
        for (int r = 0; r < 12; ++r) ws.kv[s][r] = xdot[r];
      } else if (it == 64 + NU && s == 0) {   // values of the terms
        CentLaneKin<double, CentWST<DERIV>> kin{dm, ws, CentCol{CK_NONE, 0}, {}, {}};
        double xdot[12];
        cent_lane_flow<double>(dm, ws, kin.col, xdot, kin.base);
        for (int k = 0; k < 3; ++k) kin.base.p0[k] = ws.x[6 + k];
        for (int k = 0; k < 12; ++k) kin.Wd[k] = ws.u[k];
        CentValueSink sink{DERIV ? rec : nullptr, sdt, 0.0, 0.0, 0.0, 0ull, 0, {0, 0}, {0, 0}};
        cent_terms<double>(dm, kin, kin.Wd, ws.par, sink);
        sink.finish();
        ws.tv[0] = sink.cost; ws.tv[1] = sink.eqsse; ws.tv[2] = sink.shift_d1; ws.tv[3] = (double)sink.ne;
        ws.tv[4] = (double)sink.contact[0]; ws.tv[5] = (double)sink.contact[1]; ws.tv[6] = (double)sink.eq_off[0]; ws.tv[7] = (double)sink.eq_off[1];
      } else if (DERIV && s == 0 && it > CNX && it <= CNX + 26) {   // zero-fill of the padding columns 35..57 and 93..95
        const int z = it - CNX - 1;
        const int col = z < NX - CNX ? CNX + z : NZ + (z - (NX - CNX));
        for (int r = 0; r < 12; ++r) rec[REC_PV + r * LDJ + col] = 0.0;
        for (int sr = 0; sr < NRS; ++sr) rec[REC_J + sr * LDJ + col] = 0.0;
        if (col != NZ) for (int r = 0; r < NE_MAX; ++r) rec[REC_CDE + r * LDJ + col] = 0.0;
      }
    }
    WG_SYNC(ctx);
  }
  // ---- RK4 value, defect, flow; the diagonal part of the cost model
  WG_FOR(ctx, i0, 64 + LDJ) {
    if (i0 >= 64) { (&ws.part[0][0])[i0] = cent_diag_entry(dm, ws, i0 - 64, dt, DERIV ? rec : nullptr); continue; }   // (cost of the entry, next to the defect)
    const int i = i0;
    double b = 0.0;
    if (i < 12) b = ws.x[i] + dt / 6.0 * (ws.kv[0][i] + 2.0 * ws.kv[1][i] + 2.0 * ws.kv[2][i] + ws.kv[3][i]) - ws.xnext[i];
    else if (i < CNX) b = ws.x[i] + dt * ws.u[i] - ws.xnext[i];
    if (DERIV) { rec[REC_B + i] = b; rec[REC_FLOW + i] = i < 12 ? ws.kv[0][i] : (i < CNX ? ws.u[i] : 0.0); }
    (&ws.part[0][0])[i] = b;   // (the per-body sums are dead: scratch of the defect's sum of squares)
  }
  WG_SYNC(ctx);
  WG_FOR(ctx, it, 1) {
    double dyn = 0.0;
    for (int i = 0; i < CNX; ++i) { const double b = (&ws.part[0][0])[i]; dyn += b * b; }
    misc[3] = (dt > 0.0 ? dt : 1.0) * dyn;   // event interval (dt = 0): unscaled, as hsqp_lq.h
    double cost = ws.tv[0];
    for (int i = 0; i < LDJ; ++i) cost += (&ws.part[0][0])[64 + i];
    misc[0] = ws.tv[3]; misc[1] = dt * cost; misc[2] = dt * ws.tv[1];
    misc[4] = ws.tv[4]; misc[5] = ws.tv[5]; misc[6] = ws.tv[6]; misc[7] = ws.tv[7];
    if (DERIV) { rec[REC_NROWS] = (double)NRS; rec[REC_LAYOUT] = 0.0; }   // this kernel fills every row slot (no compaction), row-major
  }
  if constexpr (DERIV) {
    // ---- chain the stage Jacobians per column: Ab_1 = G_1,  Ab_s = G_s + c_s G_s[:, 0..11] Ab_{s-1} (+ c_s G_s[:, q_j] for the column qd_j),
    //      [A|B] - [I|0] on the 12 dense rows = dt/6 (Ab_1 + 2 Ab_2 + 2 Ab_3 + Ab_4)
    WG_FOR(ctx, it, CLQ_THREADS) {
      const bool xcol = it < CNX, ucol = it >= 64 && it < 64 + NU;
      if (!xcol && !ucol) continue;
      const int gcol = xcol ? it : CNX + (it - 64), rcol = xcol ? it : NX + (it - 64);
      const int jq = (ucol && it - 64 >= 12) ? 12 + (it - 64 - 12) : -1;   // the state column of q_j for the input column qd_j
      double a[12], acc[12];
      for (int r = 0; r < 12; ++r) { a[r] = ws.G[0][r][gcol]; acc[r] = a[r]; }
      for (int s = 1; s < 4; ++s) {
        const double c = s == 3 ? dt : 0.5 * dt, wgt = s == 3 ? 1.0 : 2.0;
        double an[12];
        for (int r = 0; r < 12; ++r) {
          double v = 0.0;
#pragma unroll
          for (int k = 0; k < 12; ++k) v += ws.G[s][r][k] * a[k];
          if (jq >= 0) v += ws.G[s][r][jq];
          an[r] = ws.G[s][r][gcol] + c * v;
        }
        for (int r = 0; r < 12; ++r) { a[r] = an[r]; acc[r] += wgt * an[r]; }
      }
      for (int r = 0; r < 12; ++r) rec[REC_PV + r * LDJ + rcol] = dt / 6.0 * acc[r];
    }
  }
  WG_SYNC(ctx);
}

// ---- the two other users of the model pass, on the same workspace (one workgroup — one wave is enough — per evaluation)
// [pdot; euler rates] of the base at a centroidal (x, u): the generalized velocities CentroidalMpcMrtJointController::computeJointControlAction
// needs (humanoid_centroidal_mpc/src/mrt/CentroidalMpcMrtJointController.cpp:155-175) -> vb6
template <class W>
HSQP_HD void cent_base_velocity(const Ctx& ctx, const DevModel& dm, W& ws, const double* x, const double* u, double* vb6) {
  cent_ws_topology(ctx, dm, ws);
  WG_FOR(ctx, i, CNX + NU) { if (i < CNX) ws.x[i] = x[i]; else ws.u[i - CNX] = u[i - CNX]; }
  WG_SYNC(ctx);
  cent_stage_values(ctx, dm, ws, 0, 0.0);
  WG_FOR(ctx, it, 1) {
    CentBase<double> base;
    double xdot[12];
    cent_lane_flow<double>(dm, ws, CentCol{CK_NONE, 0}, xdot, base);
    for (int k = 0; k < 6; ++k) vb6[k] = base.vb[k];
  }
  WG_SYNC(ctx);
}
// Torso task-space reference of a node-parameter row (after node_params_eval has filled the desired state): kinematics of the torso link at
// (xRef, uRef = 0) — EndEffectorKinematicsQuadraticCost::getParameters / getReferenceCostElement
// (humanoid_common_mpc/src/cost/EndEffectorKinematicsQuadraticCost.cpp:80-104)
template <class W>
HSQP_HD void cent_params_torso(const Ctx& ctx, const DevModel& dm, W& ws, double* par) {
  cent_ws_topology(ctx, dm, ws);
  WG_FOR(ctx, i, CNX + NU) { if (i < CNX) ws.x[i] = par[HSQP_P_XDES + i]; else ws.u[i - CNX] = 0.0; }
  WG_SYNC(ctx);
  cent_stage_values(ctx, dm, ws, 0, 0.0);
  WG_FOR(ctx, it, 1) {
    CentLaneKin<double, W> kin{dm, ws, CentCol{CK_NONE, 0}, {}, {}};
    double xdot[12];
    cent_lane_flow<double>(dm, ws, kin.col, xdot, kin.base);
    for (int k = 0; k < 3; ++k) kin.base.p0[k] = ws.x[6 + k];
    const BodyRec<double> tb = kin.torso();
    double Rt[9], pos[3], vl[3], va[3];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Rt[3 * r + c] = tb.R[3 * r] * dm.torso_R[c] + tb.R[3 * r + 1] * dm.torso_R[3 + c] + tb.R[3 * r + 2] * dm.torso_R[6 + c];
    cent_point(kin.p0(), kin.vb(), kin.wb(), tb, dm.torso_p, pos, vl, va);
    double* ref = par + HSQP_PC_TORSO;
    cent_quat(Rt, ref + 3);
    for (int r = 0; r < 3; ++r) { ref[r] = pos[r]; ref[7 + r] = vl[r]; ref[10 + r] = va[r]; }
    for (int i = HSQP_PC_TORSO + 13; i < NX; ++i) par[HSQP_P_XDES + i] = 0.0;
    par[HSQP_P_SWING + 2] = 0.0; par[HSQP_P_SWING + 5] = 0.0;   // the velocity-level constraints have no acceleration reference
  }
  WG_SYNC(ctx);
}

}  // namespace hsqp
