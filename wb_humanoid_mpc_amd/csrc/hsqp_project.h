// Equality-constraint projection and change of input variables of one node:
//   D^T = [Q1 Q2][R1; 0]  (Householder),  Px = -Q1 R1^-T C,  Pe = -Q1 R1^-T e,  Pu = Q2,
//   du = Px dx + Pu ut + Pe,   A~ = A + B Px, B~ = B Pu, b~ = b + B Pe,
//   H~ = T^T H T,  g~ = T^T (g + H t),  T = [I 0; Px Pu], t = [0; Pe]
// (ocs2: projectStateInputEqualityConstraints true — task.info:87; upstream qrConstraintProjection +
// changeOfInputVariables).  The cost arrives as diag(d) + J^T J, so the projected Hessian is formed as
// diag + (T_u^T diag(d_u) T_u) + J~^T J~ with J~ = J T — the Gauss-Newton contraction of this problem.
// The projected input dimension is padded to NUT = 23 with identity (R~ = 1, everything else 0) when
// more than 12 equality rows are active, so the Riccati kernel sees a fixed stage size.
#pragma once
#include "hsqp_lq.h"

namespace hsqp {

// ---- per-node QP record (doubles) written by the projection kernel, read by the Riccati kernel
constexpr int QP_A = 0;                        // [58][58]
constexpr int QP_B = QP_A + NX * NX;           // [58][23]
constexpr int QP_BV = QP_B + NX * NUT;         // [58]
constexpr int QP_Q = QP_BV + NX;               // [58][58]
constexpr int QP_P = QP_Q + NX * NX;           // [23][58]
constexpr int QP_R = QP_P + NUT * NX;          // [23][23]
constexpr int QP_QV = QP_R + NUT * NUT;        // [58]
constexpr int QP_RV = QP_QV + NX;              // [23]
constexpr int QP_PX = QP_RV + NUT;             // [35][58]
constexpr int QP_PU = QP_PX + NU * NX;         // [35][23]
constexpr int QP_PE = QP_PU + NU * NUT;        // [35]
constexpr int QP_NUT = QP_PE + NU;             // [1]  nu - ne, or -1 if D was rank deficient
constexpr int QP_SIZE = ((QP_NUT + 1 + 7) / 8) * 8;

constexpr int NTW = NX + NUT;                  // 81: projected stage variable [dx; ut]
constexpr int LDT = 82;                        // leading dimension of J~ in LDS

struct ProjWS {
  union {
    struct {
      double CDe[NE_MAX][LDJ];
      double Rm[NU][NE_MAX];     // D^T, overwritten by R1
      double Qm[NU][NU];
      double Wm[NE_MAX][NX + 1]; // R1^-T [C|e]
      double hv[NU];             // Householder vector
      double hs[2];              // {2/|v|^2, ok flag}
    } qr;
    double Ju[NRS][NU];          // input block of the residual rows (staged after the QR data is dead)
  };
  int ne, nut, ok;
  double Px[NU][NX], Pu[NU][NUT], Pe[NU];
  double PV[2][6][LDJ];
  double bvec[64];
  double d[LDJ], gd[LDJ], rho[NRS], gu[NU];
  double Jt[NRS][LDT];
};

HSQP_HD double tu(const ProjWS& w, int k, int a) { return a < NX ? w.Px[k][a] : w.Pu[k][a - NX]; }

HSQP_HD void project_node(const Ctx& ctx, ProjWS& w, const double* rec, double dt, double* qp) {
  // ---- load
  WG_FOR(ctx, i, NE_MAX * LDJ + 2 * 6 * LDJ + 64 + 2 * LDJ + NRS + 1) {
    int j = i;
    if (j < NE_MAX * LDJ) { w.qr.CDe[j / LDJ][j % LDJ] = rec[REC_CDE + j]; continue; }
    j -= NE_MAX * LDJ;
    if (j < 2 * 6 * LDJ) { w.PV[j / (6 * LDJ)][(j / LDJ) % 6][j % LDJ] = rec[REC_PV + j]; continue; }
    j -= 2 * 6 * LDJ;
    if (j < 64) { w.bvec[j] = rec[REC_B + j]; continue; }
    j -= 64;
    if (j < LDJ) { w.d[j] = rec[REC_D + j]; continue; }
    j -= LDJ;
    if (j < LDJ) { w.gd[j] = rec[REC_GD + j]; continue; }
    j -= LDJ;
    if (j < NRS) { w.rho[j] = rec[REC_RHO + j]; continue; }
    w.ne = (int)rec[REC_MISC];
    w.nut = NU - w.ne;
    w.ok = 1;
  }
  WG_SYNC(ctx);
  const int ne = w.ne, nut = w.nut;
  // ---- Householder QR of D^T
  WG_FOR(ctx, i, NU * NE_MAX + NU * NU) {
    if (i < NU * NE_MAX) { const int r = i / NE_MAX, c = i % NE_MAX; w.qr.Rm[r][c] = c < ne ? w.qr.CDe[c][NX + r] : 0.0; }
    else { const int r = (i - NU * NE_MAX) / NU, c = (i - NU * NE_MAX) % NU; w.qr.Qm[r][c] = r == c ? 1.0 : 0.0; }
  }
  WG_SYNC(ctx);
  for (int k = 0; k < ne; ++k) {
    WG_FOR(ctx, it, 1) {
      double nrm = 0.0;
      for (int i = k; i < NU; ++i) nrm += w.qr.Rm[i][k] * w.qr.Rm[i][k];
      nrm = sqrt(nrm);
      if (nrm < 1e-12) w.ok = 0;
      const double alpha = w.qr.Rm[k][k] >= 0.0 ? -nrm : nrm;
      double vn = 0.0;
      for (int i = 0; i < NU; ++i) {
        double v = i < k ? 0.0 : w.qr.Rm[i][k];
        if (i == k) v -= alpha;
        w.qr.hv[i] = v;
        vn += v * v;
      }
      w.qr.hs[0] = vn > 1e-300 ? 2.0 / vn : 0.0;
    }
    WG_SYNC(ctx);
    WG_FOR(ctx, it, (ne - k) + NU) {
      const double beta = w.qr.hs[0];
      if (it < ne - k) {  // R(:,c) <- (I - beta v v^T) R(:,c)
        const int c = k + it;
        double s = 0.0;
        for (int i = k; i < NU; ++i) s += w.qr.hv[i] * w.qr.Rm[i][c];
        s *= beta;
        for (int i = k; i < NU; ++i) w.qr.Rm[i][c] -= s * w.qr.hv[i];
      } else {            // Q(r,:) <- Q(r,:) (I - beta v v^T)
        const int r = it - (ne - k);
        double s = 0.0;
        for (int i = k; i < NU; ++i) s += w.qr.Qm[r][i] * w.qr.hv[i];
        s *= beta;
        for (int i = k; i < NU; ++i) w.qr.Qm[r][i] -= s * w.qr.hv[i];
      }
    }
    WG_SYNC(ctx);
  }
  // ---- W = R1^-T [C | e]: forward substitution per column
  WG_FOR(ctx, c, NX + 1) {
    for (int i = 0; i < ne; ++i) {
      double s = c < NX ? w.qr.CDe[i][c] : w.qr.CDe[i][NZ];
      for (int j = 0; j < i; ++j) s -= w.qr.Rm[j][i] * w.qr.Wm[j][c];
      w.qr.Wm[i][c] = s / w.qr.Rm[i][i];
    }
  }
  WG_SYNC(ctx);
  WG_FOR(ctx, i, NU * (NX + 1 + NUT)) {
    const int r = i / (NX + 1 + NUT), c = i % (NX + 1 + NUT);
    if (c <= NX) {
      double s = 0.0;
      for (int j = 0; j < ne; ++j) s += w.qr.Qm[r][j] * w.qr.Wm[j][c];
      if (c < NX) w.Px[r][c] = -s; else w.Pe[r] = -s;
    } else {
      const int cc = c - NX - 1;
      w.Pu[r][cc] = cc < nut ? w.qr.Qm[r][ne + cc] : 0.0;
    }
  }
  WG_SYNC(ctx);  // QR data dead from here: Ju aliases it
  // ---- stage the input block of the residual rows, write the projection
  WG_FOR(ctx, i, NRS * NU + NU * (NX + NUT + 1) + 1) {
    if (i < NRS * NU) { const int r = i / NU, k = i % NU; w.Ju[r][k] = rec[REC_J + r * LDJ + NX + k]; continue; }
    int j = i - NRS * NU;
    if (j < NU * NX) { qp[QP_PX + j] = w.Px[j / NX][j % NX]; continue; }
    j -= NU * NX;
    if (j < NU * NUT) { qp[QP_PU + j] = w.Pu[j / NUT][j % NUT]; continue; }
    j -= NU * NUT;
    if (j < NU) { qp[QP_PE + j] = w.Pe[j]; w.gu[j] = w.gd[NX + j] + w.d[NX + j] * w.Pe[j]; continue; }
    qp[QP_NUT] = w.ok ? (double)nut : -1.0;
  }
  WG_SYNC(ctx);
  // ---- dynamics: A~ = A + B Px, B~ = B Pu, b~ = b + B Pe  with the structured [A|B] (hsqp_lq.h)
  WG_FOR(ctx, i, NX * (NX + NUT + 1)) {
    const int r = i / (NX + NUT + 1), c = i % (NX + NUT + 1);
    // B row r applied to column c of [Px | Pu | Pe]
    double s = 0.0;
    const bool base = (r < 6) || (r >= NV && r < NV + 6);
    if (base) {
      const double* Brow = &w.PV[r < 6 ? 0 : 1][r < 6 ? r : r - NV][NX];
      if (c < NX) for (int k = 0; k < NU; ++k) s += Brow[k] * w.Px[k][c];
      else if (c < NX + NUT) for (int k = 0; k < NU; ++k) s += Brow[k] * w.Pu[k][c - NX];
      else for (int k = 0; k < NU; ++k) s += Brow[k] * w.Pe[k];
    } else {
      const int j = r < NV ? r - 6 : r - NV - 6;
      const double coef = r < NV ? 0.5 * dt * dt : dt;
      s = coef * (c < NX ? w.Px[12 + j][c] : (c < NX + NUT ? w.Pu[12 + j][c - NX] : w.Pe[12 + j]));
    }
    if (c < NX) {
      double a = (r == c) ? 1.0 : 0.0;
      if (r < NV && c == NV + r) a += dt;
      if (base) a += w.PV[r < 6 ? 0 : 1][r < 6 ? r : r - NV][c];
      qp[QP_A + r * NX + c] = a + s;
    } else if (c < NX + NUT) {
      qp[QP_B + r * NUT + (c - NX)] = s;
    } else {
      qp[QP_BV + r] = w.bvec[r] + s;
    }
  }
  // ---- J~ = J T (state block read from global, input block from LDS), rho' = rho + J_u Pe
  WG_FOR(ctx, i, NRS * (NTW + 1)) {
    const int r = i / (NTW + 1), a = i % (NTW + 1);
    double s = 0.0;
    if (a < NX) { s = rec[REC_J + r * LDJ + a]; for (int k = 0; k < NU; ++k) s += w.Ju[r][k] * w.Px[k][a]; w.Jt[r][a] = s; }
    else if (a < NTW) { for (int k = 0; k < NU; ++k) s += w.Ju[r][k] * w.Pu[k][a - NX]; w.Jt[r][a] = s; }
    else { s = w.rho[r]; for (int k = 0; k < NU; ++k) s += w.Ju[r][k] * w.Pe[k]; w.Jt[r][LDT - 1] = s; }
  }
  WG_SYNC(ctx);
  // ---- projected gradient and Hessian (upper triangle computed, mirrored on write)
  WG_FOR(ctx, i, NTW + NTW * (NTW + 1) / 2) {
    if (i < NTW) {
      const int a = i;
      double s = a < NX ? w.gd[a] : 0.0;
      for (int k = 0; k < NU; ++k) s += tu(w, k, a) * w.gu[k];
      for (int r = 0; r < NRS; ++r) s += w.Jt[r][a] * w.Jt[r][LDT - 1];
      if (a < NX) qp[QP_QV + a] = s; else qp[QP_RV + a - NX] = s;
      continue;
    }
    // unrank the upper-triangular pair (a <= b)
    int t = i - NTW, a = 0;
    while (t >= NTW - a) { t -= NTW - a; ++a; }
    const int b = a + t;
    double s = (a == b && a < NX) ? w.d[a] : 0.0;
    for (int k = 0; k < NU; ++k) s += w.d[NX + k] * tu(w, k, a) * tu(w, k, b);
    for (int r = 0; r < NRS; ++r) s += w.Jt[r][a] * w.Jt[r][b];
    if (b < NX) { qp[QP_Q + a * NX + b] = s; qp[QP_Q + b * NX + a] = s; }
    else if (a < NX) { qp[QP_P + (b - NX) * NX + a] = s; }
    else {
      if (a - NX >= nut || b - NX >= nut) s = (a == b) ? 1.0 : 0.0;  // identity padding of the unused projected inputs
      qp[QP_R + (a - NX) * NUT + (b - NX)] = s; qp[QP_R + (b - NX) * NUT + (a - NX)] = s;
    }
  }
  WG_SYNC(ctx);
}

}  // namespace hsqp
