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
#include "hsqp_linalg.h"
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
constexpr int LDTM = 88;                       // leading dimension of Tm = [Px | Pu | Pe | pad] and of the residual rows
// The projected residual rows (64 slots + 35 input-weight rows sqrt(d_u) [Px|Pu|Pe]) are processed in two passes so that
// the workspace stays under 80 KB (two workgroups per CU): pass A = slots 0..47, pass B = slots 48..63 + weight rows.
constexpr int NRA = 48;                        // residual row slots of pass A
constexpr int NRB = (NRS - NRA) + NU + 1;      // 16 slots + 35 input-weight rows + 1 zero row = 52
static_assert(NRA % 4 == 0 && NRB % 4 == 0 && NRB >= NRA, "pass sizes");

constexpr int LDR = 16;
struct ProjWS {
  union {
    struct {
      double CDe[NE_MAX][LDJ];
      double Rm[NU + 1][LDR];    // D^T (zero padded to 36 x 16), overwritten by R1
      double QT[NU][NU];         // Q^T of the Householder QR (rows 0..ne-1 = Q1^T, the rest Q2^T)
      double Wm[NE_MAX][NX + 2]; // R1^-T [C|e]
      double V[NE_MAX][NU + 1];  // Householder vectors
      double beta[NE_MAX], Rdiag[NE_MAX], rinv[LDR], ep[LDR];   // ep: e of the dense rows after the eliminated inputs are substituted
      double part[LDR][4], yk[LDR];   // per-step scratch: partial dots, row k of R
    } qr;                        // live until Tm is formed
    double PV[2][6][LDJ];        // then the non-trivial rows of [A|B] (loaded when the QR data is dead) ...
    double Jt[NRB][LDTM];        // ... then the projected residual rows of the current pass (column 81 = rho')
  };
  double JuT[NU][NRA];           // transposed input block of the residual rows of the current pass
  int ne, nut, ok;
  // deflation of the unit rows of D (a swing foot's zero-wrench constraints W_f = 0 fix six inputs outright): the Householder
  // QR only sees the dense rows (ned of them) restricted to the free inputs (nub of them)
  int ned, nub;
  int ub[NU];                    // free input indices, then the eliminated ones (positions nub..NU-1)
  int urow[NU];                  // for an eliminated input: its unit constraint row; -1 for a free input
  int rd[NE_MAX];                // dense constraint rows
  double Tm[NU][LDTM];           // [Px (58) | Pu (23) | Pe | 0 0]
  double bvec[64];
  double rho[NRS], d[LDJ], gd[LDJ];   // mirror one contiguous piece of the LQ record
  double gacc[LDTM];             // gradient partial sums of pass A
};

// Event interval (hsqp_problem::dt_nodes[b][k] == 0): the stage of the QP is the identity jump map
// dx+ = dx + (x_k - x_{k+1}) with no cost and the inputs pinned (R~ = I, everything else zero -> ut = 0, du = 0).  b~ is the defect
// the LQ kernel computed with dt = 0.  Overwrites what project_node wrote for this node.
HSQP_HD void jump_node_qp(const Ctx& ctx, const double* rec, double* qp) {
  WG_FOR(ctx, i, QP_SIZE) {
    double v = 0.0;
    if (i < QP_B) { const int a = i / NX, c = i % NX; v = a == c ? 1.0 : 0.0; }
    else if (i >= QP_BV && i < QP_Q) v = rec[REC_B + i - QP_BV];
    else if (i >= QP_R && i < QP_QV) { const int a = (i - QP_R) / NUT, c = (i - QP_R) % NUT; v = a == c ? 1.0 : 0.0; }
    else if (i == QP_NUT) v = (double)NUT;
    qp[i] = v;
  }
  WG_SYNC(ctx);
}

// cent = true: the record comes from the centroidal LQ kernel (hsqp_cent.h): the dense rows of [A|B] - [I|0] are rows 0..11
// (PV[0] = momentum rows, PV[1] = base pose rows), rows 12..34 are q_j+ = q_j + dt qd_j, rows 35..57 padding states (A = I).
HSQP_HD void project_node(const Ctx& ctx, ProjWS& w, const double* rec, double dt, double* qp, bool cent = false) {
  // ---- load: record pieces [REC_B, REC_J) -> bvec and [REC_RHO, REC_MISC) -> rho, d, gd, CDe; 8 loads in flight per item
  {
    static_assert(REC_B == REC_PV + 2 * 6 * LDJ && REC_J == REC_B + 64, "record layout");
    static_assert(REC_D == REC_RHO + NRS && REC_GD == REC_D + LDJ && REC_CDE == REC_GD + LDJ && REC_MISC == REC_CDE + NE_MAX * LDJ, "record layout");
    constexpr int n1 = 64, n2 = NRS + 2 * LDJ, n3 = NE_MAX * LDJ, nld = n1 + n2 + n3, nb = nbatches(nld, 8);
    double* dst1 = &w.bvec[0];
    double* dst2 = &w.rho[0];
    double* dst3 = &w.qr.CDe[0][0];
    WG_FOR(ctx, b, nb) {
      double t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int idx = b + j * nb; t[j] = idx < nld ? rec[idx < n1 ? REC_B + idx : REC_RHO + (idx - n1)] : 0.0; }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = b + j * nb;
        if (idx < n1) dst1[idx] = t[j];
        else if (idx < n1 + n2) dst2[idx - n1] = t[j];
        else if (idx < nld) dst3[idx - n1 - n2] = t[j];
      }
    }
    // structure of the equality rows (closed forms, one item per input / row): a swing foot f contributes the unit rows
    // off[f] .. off[f]+5 on the inputs 6f .. 6f+5
    WG_FOR(ctx, i, NU + NE_MAX + 1) {
      const int ne_ = (int)rec[REC_MISC];
      const int sw0 = rec[REC_MISC + 4] == 0.0 ? 1 : 0, sw1 = rec[REC_MISC + 5] == 0.0 ? 1 : 0;
      const int off0 = (int)rec[REC_MISC + 6], off1 = (int)rec[REC_MISC + 7];
      const int nel = 6 * (sw0 + sw1);
      auto clamp6 = [](int v) { return v < 0 ? 0 : (v > 6 ? 6 : v); };
      if (i < NU) {
        const int u = i;
        const int before = sw0 * clamp6(u) + sw1 * clamp6(u - 6);          // eliminated inputs below u
        const bool elim = (u < 6 && sw0) || (u >= 6 && u < 12 && sw1);
        w.urow[u] = elim ? (u < 6 ? off0 + u : off1 + u - 6) : -1;
        if (elim) w.ub[NU - nel + before] = u; else w.ub[u - before] = u;
      } else if (i < NU + NE_MAX) {
        const int r = i - NU;
        const int before = sw0 * clamp6(r - off0) + sw1 * clamp6(r - off1);  // unit rows below r
        const bool unit = (sw0 && r >= off0 && r < off0 + 6) || (sw1 && r >= off1 && r < off1 + 6);
        if (r < ne_ && !unit) w.rd[r - before] = r;
        if (r >= ne_ - nel) w.rd[r] = 0;                                      // padding of the dense-row list
      } else {
        w.ne = ne_;
        w.nut = NU - ne_;
        w.ok = 1;
        w.ned = ne_ - nel;
        w.nub = NU - nel;
      }
    }
  }
  WG_SYNC(ctx);
  const int ne = w.ne, nut = w.nut, ned = w.ned, nub = w.nub;
  PH_TICK(ctx, 1);
  // ---- Householder QR of D^T.  Step k: every remaining column recomputes the reflector from column k (which is
  // left untouched: its final diagonal goes to Rdiag, the vector to V).
  WG_FOR(ctx, i, (NU + 1) * LDR) { const int r = i / LDR, c = i % LDR; w.qr.Rm[r][c] = (c < ned && r < nub) ? w.qr.CDe[w.rd[c]][NX + w.ub[r]] : 0.0; if (r == 0) {
      w.qr.rinv[c] = 0.0;
      // e' = e - D_a e_U: the eliminated inputs are fixed at -e_U (the unit rows have C = 0)
      double ep = 0.0;
      if (c < ned) {
        const int ri = w.rd[c];
        ep = w.qr.CDe[ri][NZ];
        for (int t = nub; t < NU; ++t) { const int ue = w.ub[t]; ep -= w.qr.CDe[ri][NX + ue] * w.qr.CDe[w.urow[ue]][NZ]; }
      }
      w.qr.ep[c] = ep;
    }
    if (i < NU) w.d[NX + i] = sqrt(w.d[NX + i]); }   // d_u -> sqrt(d_u) (only used for the weight rows)
  WG_SYNC(ctx);
  PH_TICK(ctx, 8);
  // Both phases run on fixed item grids with unconditional loads (columns >= ne and row 35 are zero padding), so a
  // phase is one LDS round trip; masks are applied to values, not to control flow.
  for (int k = 0; k < ned; ++k) {
    // (a) partial dots x . R(:,c) of the pivot column x = R(k:,k) with every column (item = column c, row residue p mod 4)
    WG_FOR(ctx, it, LDR * 4) {
      const int c = it >> 2, p = it & 3;
      double xk[9], yc[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) { xk[t] = w.qr.Rm[p + 4 * t][k]; yc[t] = w.qr.Rm[p + 4 * t][c]; }
      double sdot = 0.0;
#pragma unroll
      for (int t = 0; t < 9; ++t) sdot += (p + 4 * t >= k ? xk[t] : 0.0) * yc[t];
      w.qr.part[c][p] = sdot;
      if (p == 0) w.qr.yk[c] = w.qr.Rm[k][c];
    }
    WG_SYNC(ctx);
    // (b) every item recomputes the reflector scalars (no broadcast phase); v = x - alpha e_k;
    //     R(:,c) -= beta (v . R(:,c)) v  with  v . y = x . y - alpha y_k ;  column k itself is recorded in V
    WG_FOR(ctx, it, NE_MAX * (NU + 1)) {
      const int c = it / (NU + 1), i = it % (NU + 1);
      const double p0 = w.qr.part[k][0], p1 = w.qr.part[k][1], p2 = w.qr.part[k][2], p3 = w.qr.part[k][3], rkk = w.qr.yk[k];
      const double q0 = w.qr.part[c][0], q1 = w.qr.part[c][1], q2 = w.qr.part[c][2], q3 = w.qr.part[c][3], ykc = w.qr.yk[c];
      const double xi = w.qr.Rm[i][k], rc = w.qr.Rm[i][c];
      const double nrm2 = (p0 + p1) + (p2 + p3);
      const double rs = inv_sqrt(nrm2 > 1e-300 ? nrm2 : 1e-300);
      const double nrm = nrm2 * rs;
      const double alpha = rkk >= 0.0 ? -nrm : nrm;
      const double hv = nrm2 - alpha * rkk;      // = |v|^2 / 2
      const double beta = hv > 1e-300 ? fast_rcp(hv) : 0.0;
      if (it == 0) {
        w.qr.beta[k] = beta;
        w.qr.Rdiag[k] = alpha;
        w.qr.rinv[k] = rkk >= 0.0 ? -rs : rs;
        if (!(nrm >= 1e-12)) w.ok = 0;
      }
      const double vi = i < k ? 0.0 : (xi - (i == k ? alpha : 0.0));
      const double sdot = beta * (((q0 + q1) + (q2 + q3)) - alpha * ykc);
      if (c == k) w.qr.V[k][i] = vi;
      else if (c > k && i >= k) w.qr.Rm[i][c] = rc - sdot * vi;
    }
    WG_SYNC(ctx);
  }
  PH_TICK(ctx, 9);
  // Q^T = H_{ned-1} ... H_0 of the free inputs: one column per item, the column lives in registers while the reflectors are
  // applied; it is stored at the ORIGINAL input index (column ub[c]); the columns of the eliminated inputs are zero
  WG_FOR(ctx, c, NU) {
    double col[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) col[i] = (i == c && c < nub) ? 1.0 : 0.0;
    for (int k = 0; k < (c < nub ? ned : 0); ++k) {
      double sdot = 0.0;
#pragma unroll
      for (int i = 0; i < NU; ++i) sdot += w.qr.V[k][i] * col[i];
      sdot *= w.qr.beta[k];
#pragma unroll
      for (int i = 0; i < NU; ++i) col[i] -= sdot * w.qr.V[k][i];
    }
    const int uc = w.ub[c];
#pragma unroll
    for (int i = 0; i < NU; ++i) w.qr.QT[i][uc] = col[i];
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 2);
  // ---- W = R1^-T [C | e'] over the dense rows: forward substitution per column, the column kept in registers (rows >= ned:
  //      rinv = 0 -> 0)
  WG_FOR(ctx, c, NX + 1) {
    double wc[NE_MAX];
#pragma unroll
    for (int i = 0; i < NE_MAX; ++i) {
      const int ri = w.rd[i];
      double s = c < NX ? w.qr.CDe[ri][c] : w.qr.ep[i];
#pragma unroll
      for (int j = 0; j < i; ++j) s -= w.qr.Rm[j][i] * wc[j];
      wc[i] = s * w.qr.rinv[i];
      w.qr.Wm[i][c] = wc[i];
    }
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 3);
  // ---- Tm = [Px | Pu | Pe | 0 0]:  [Px | Pe] = -Q1 W  (X^T Y with X = Q1^T),  Pu = Q2
  {
    const XtyJob job = xty_job(NU, NX, ned, &w.qr.QT[0][0], NU, &w.qr.Wm[0][0], NX + 2, &w.Tm[0][0], LDTM, nullptr, 0, -1.0);
    wg_xty_jobs<true>(ctx, &job, 1);
    WG_FOR(ctx, i, NU * (NUT + 3)) {
      const int r = i / (NUT + 3), cc = i % (NUT + 3);
      if (cc < NUT) w.Tm[r][NX + cc] = cc < nut ? w.qr.QT[ned + cc][r] : 0.0;
      else if (cc == NUT) {
        double sdot = 0.0;
        for (int j = 0; j < ned; ++j) sdot += w.qr.QT[j][r] * w.qr.Wm[j][NX];
        w.Tm[r][NTW] = (w.urow[r] >= 0 ? -w.qr.CDe[w.urow[r]][NZ] : 0.0) - sdot;   // an eliminated input is fixed at -e of its unit row
      }
      else w.Tm[r][NTW + (cc - NUT)] = 0.0;
    }
  }
  WG_SYNC(ctx);  // QR data dead from here: PV, then Jt alias it
  PH_TICK(ctx, 4);
  // ---- stage the (transposed) input block of the residual rows r0 .. r0+nr and, in pass A, the [A|B] rows
  auto stage_inputs = [&](int r0, int nr, bool with_pv) {
    const int nj = nr * NU, ntot = nj + (with_pv ? 2 * 6 * LDJ : 0), nb = (ntot + 7) / 8;
    double* pv = &w.PV[0][0][0];
    WG_FOR(ctx, b, nb) {
      double t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = b + j * nb;
        t[j] = e < nj ? rec[REC_J + (r0 + e / NU) * LDJ + NX + e % NU] : (e < ntot ? rec[REC_PV + (e - nj)] : 0.0);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = b + j * nb;
        if (e < nj) w.JuT[e % NU][e / NU] = t[j];
        else if (e < ntot) pv[e - nj] = t[j];
      }
    }
  };
  stage_inputs(0, NRA, true);
  WG_SYNC(ctx);
  // ---- write the projection
  WG_FOR(ctx, i, NU * (NX + NUT + 1) + 1) {
    int j = i;
    if (j < NU * NX) { qp[QP_PX + j] = w.Tm[j / NX][j % NX]; continue; }
    j -= NU * NX;
    if (j < NU * NUT) { qp[QP_PU + j] = w.Tm[j / NUT][NX + j % NUT]; continue; }
    j -= NU * NUT;
    if (j < NU) { qp[QP_PE + j] = w.Tm[j][NTW]; continue; }
    qp[QP_NUT] = w.ok ? (double)nut : -1.0;
  }
  // ---- dynamics: A~ = A + B Px, B~ = B Pu, b~ = b + B Pe  with the structured [A|B] (hsqp_lq.h)
  WG_FOR(ctx, i, NX * (NTW + 1)) {
    const int r = i / (NTW + 1), c = i % (NTW + 1);
    double s = 0.0;
    const bool base = cent ? r < 12 : ((r < 6) || (r >= NV && r < NV + 6));
    const int pw = cent ? r / 6 : (r < 6 ? 0 : 1), pr = cent ? r % 6 : (r < 6 ? r : r - NV);   // block / row of PV (only used if base)
    if (base) {  // B row r applied to column c of [Px | Pu | Pe]
      const double* Brow = &w.PV[pw][pr][NX];
#pragma unroll 5
      for (int k = 0; k < NU; ++k) s += Brow[k] * w.Tm[k][c];
    } else if (cent) {
      s = r < HSQP_CNX ? dt * w.Tm[r][c] : 0.0;            // input 12 + (r - 12) = r
    } else {
      const int j = r < NV ? r - 6 : r - NV - 6;
      s = (r < NV ? 0.5 * dt * dt : dt) * w.Tm[12 + j][c];
    }
    if (c < NX) {
      double a = (r == c) ? 1.0 : 0.0;
      if (!cent && r < NV && c == NV + r) a += dt;
      if (base) a += w.PV[pw][pr][c];
      qp[QP_A + r * NX + c] = a + s;
    } else if (c < NTW) {
      qp[QP_B + r * NUT + (c - NX)] = s;
    } else {
      qp[QP_BV + r] = w.bvec[r] + s;
    }
  }
  WG_SYNC(ctx);  // PV dead: Jt aliases it
  PH_TICK(ctx, 5);
  // ---- two passes over the residual rows: J~ = J T (+ rho' in column 81), then H~ (+)= J~^T J~ on the matrix cores (blocks
  //      Q~, P~, R~ straight to the QP record; pass B adds to what the same lanes wrote in pass A) and the gradient
  //      g~ = T^T gd + J~^T rho'
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int r0 = pass == 0 ? 0 : NRA, nr = pass == 0 ? NRA : NRS - NRA, nrows = pass == 0 ? NRA : NRB;
    if (pass == 1) { stage_inputs(NRA, NRS - NRA, false); WG_SYNC(ctx); PH_TICK(ctx, 13); }
    {
      const XtyJob jobs[2] = {xty_job(nr, NX, NU, &w.JuT[0][0], NRA, &w.Tm[0][0], LDTM, &w.Jt[0][0], LDTM, rec + REC_J + r0 * LDJ, LDJ),
                              xty_job(nr, NUT, NU, &w.JuT[0][0], NRA, &w.Tm[0][NX], LDTM, &w.Jt[0][NX], LDTM)};
      wg_xty_jobs<true, XTY_ADD_GLOBAL>(ctx, jobs, 2);
      WG_FOR(ctx, i, nr + (pass == 1 ? (NU + 1) * LDTM : 0)) {
        if (i < nr) {
          double sdot = w.rho[r0 + i];
#pragma unroll
          for (int k = 0; k < NU; ++k) sdot += w.JuT[k][i] * w.Tm[k][NTW];
          w.Jt[i][NTW] = sdot;
        } else {   // pass B: the input-weight rows sqrt(d_u) [Px | Pu | Pe] and the zero row
          const int k = (i - nr) / LDTM, a = (i - nr) % LDTM;
          w.Jt[nr + k][a] = (k < NU && a <= NTW) ? w.d[NX + k] * w.Tm[k][a] : 0.0;
        }
      }
    }
    WG_SYNC(ctx);
    PH_TICK(ctx, pass == 0 ? 6 : 11);
    {
      const double* addq = pass == 0 ? nullptr : qp + QP_Q;
      const double* addp = pass == 0 ? nullptr : qp + QP_P;
      const double* addr = pass == 0 ? nullptr : qp + QP_R;
      // Q~ and R~ are symmetric: only the tiles on/above the diagonal are computed (10 of 16, 3 of 4) and mirrored
      XtyJob jq = xty_job(NX, NX, nrows, &w.Jt[0][0], LDTM, &w.Jt[0][0], LDTM, qp + QP_Q, NX, addq, NX);
      XtyJob jr = xty_job(NUT, NUT, nrows, &w.Jt[0][NX], LDTM, &w.Jt[0][NX], LDTM, qp + QP_R, NUT, addr, NUT);
      jq.sym = 1; jr.sym = 1;
      const XtyJob jobs[3] = {jq, xty_job(NUT, NX, nrows, &w.Jt[0][NX], LDTM, &w.Jt[0][0], LDTM, qp + QP_P, NX, addp, NX), jr};
      wg_xty_jobs<true, XTY_ADD_GLOBAL | XTY_C_GLOBAL>(ctx, jobs, 3);
      PH_TICK(ctx, pass == 0 ? 14 : 15);
      WG_FOR(ctx, a, NTW) {
        double s = 0.0;
#pragma unroll 4
        for (int r = 0; r < nrows; ++r) s += w.Jt[r][a] * w.Jt[r][NTW];
        if (pass == 0) {
          if (a < NX) s += w.gd[a];
#pragma unroll
          for (int k = 0; k < NU; ++k) s += w.Tm[k][a] * w.gd[NX + k];
          w.gacc[a] = s;
        } else {
          s += w.gacc[a];
          if (a < NX) qp[QP_QV + a] = s; else qp[QP_RV + a - NX] = s;
        }
      }
    }
    WG_SYNC(ctx);
    PH_TICK(ctx, pass == 0 ? 7 : 12);
  }
  // diagonal of Q~ and the identity padding of the unused projected inputs (read-modify-write of this node's own record)
  WG_FOR(ctx, i, NX + NUT * NUT) {
    if (i < NX) qp[QP_Q + i * NX + i] += w.d[i];
    else {
      const int a = (i - NX) / NUT, b = (i - NX) % NUT;
      if (a >= nut || b >= nut) qp[QP_R + a * NUT + b] = (a == b) ? 1.0 : 0.0;
    }
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 10);
}

}  // namespace hsqp
