// Riccati recursion of the projected, equality-free stage QP (the reference solves
// the same QP with HPIPM through ocs2's HpipmInterface — the minimiser is unique, A.1):
//   backward  Lam = R~ + B~^T S+ B~ (Cholesky),  K = -Lam^-1 (P~ + B~^T S+ A~),  k = -Lam^-1 (r~ + B~^T (s+ + S+ b~))
//             S = Q~ + A~^T S+ A~ + G^T K,  s = q~ + A~^T (s+ + S+ b~) + G^T k,   S_N = diag(Qf), s_N = Qf (x_N - x_des)
//   forward   ut = K dx + k,  dx+ = A~ dx + B~ ut + b~,  du = Px dx + Pu ut + Pe
// One workgroup per MPC instance; the stage matrices live in LDS for the whole backward step.
#pragma once
#include "hsqp_project.h"

namespace hsqp {

constexpr int RIC_K = 0;                     // [23][58]
constexpr int RIC_KV = RIC_K + NUT * NX;     // [23]
constexpr int RIC_SIZE = ((RIC_KV + NUT + 7) / 8) * 8;

struct RicWS {
  double S[NX][NX], Sn[NX][NX], A[NX][NX], SA[NX][NX];
  double B[NX][NUT], SB[NX][NUT];
  double Gm[NUT][NX], Km[NUT][NX];
  double Lam[NUT][NUT];
  double sv[NX], sb[NX], bt[NX], gv[NUT], kv[NUT], dx[NX], dxn[NX], ut[NUT];
  int ok;
};

// Returns through w.ok whether every Lam was positive definite.  qp: [N][QP_SIZE], ric: [N][RIC_SIZE].
HSQP_HD void riccati_backward(const Ctx& ctx, RicWS& w, const double* Qf, const double* xN, const double* parN, const double* qp,
                              double* ric, int N) {
  WG_FOR(ctx, i, NX * NX + NX + 1) {
    if (i < NX * NX) { const int r = i / NX, c = i % NX; w.S[r][c] = r == c ? Qf[r] : 0.0; }
    else if (i < NX * NX + NX) { const int r = i - NX * NX; w.sv[r] = Qf[r] * (xN[r] - parN[HSQP_P_XDES + r]); }
    else w.ok = 1;
  }
  WG_SYNC(ctx);
  for (int k = N - 1; k >= 0; --k) {
    const double* q = qp + (size_t)k * QP_SIZE;
    WG_FOR(ctx, i, NX * NX + NX * NUT + NX) {
      if (i < NX * NX) w.A[i / NX][i % NX] = q[QP_A + i];
      else if (i < NX * NX + NX * NUT) { const int j = i - NX * NX; w.B[j / NUT][j % NUT] = q[QP_B + j]; }
      else w.bt[i - NX * NX - NX * NUT] = q[QP_BV + i - NX * NX - NX * NUT];
    }
    WG_SYNC(ctx);
    // SA = S A, SB = S B, sb = s + S b
    WG_FOR(ctx, i, NX * (NX + NUT + 1)) {
      const int r = i / (NX + NUT + 1), c = i % (NX + NUT + 1);
      double s = 0.0;
      if (c < NX) { for (int l = 0; l < NX; ++l) s += w.S[r][l] * w.A[l][c]; w.SA[r][c] = s; }
      else if (c < NX + NUT) { const int cc = c - NX; for (int l = 0; l < NX; ++l) s += w.S[r][l] * w.B[l][cc]; w.SB[r][cc] = s; }
      else { s = w.sv[r]; for (int l = 0; l < NX; ++l) s += w.S[r][l] * w.bt[l]; w.sb[r] = s; }
    }
    WG_SYNC(ctx);
    // Lam = R + B^T SB (symmetrised), G = P + B^T SA, g = r + B^T sb
    WG_FOR(ctx, i, NUT * (NUT + NX + 1)) {
      const int r = i / (NUT + NX + 1), c = i % (NUT + NX + 1);
      if (c < NUT) {
        if (c < r) continue;
        double s1 = q[QP_R + r * NUT + c], s2 = q[QP_R + c * NUT + r];
        for (int l = 0; l < NX; ++l) { s1 += w.B[l][r] * w.SB[l][c]; s2 += w.B[l][c] * w.SB[l][r]; }
        const double a = 0.5 * (s1 + s2);
        w.Lam[r][c] = a; w.Lam[c][r] = a;
      } else if (c < NUT + NX) {
        const int cc = c - NUT;
        double s = q[QP_P + r * NX + cc];
        for (int l = 0; l < NX; ++l) s += w.B[l][r] * w.SA[l][cc];
        w.Gm[r][cc] = s;
      } else {
        double s = q[QP_RV + r];
        for (int l = 0; l < NX; ++l) s += w.B[l][r] * w.sb[l];
        w.gv[r] = s;
      }
    }
    WG_SYNC(ctx);
    // Cholesky Lam = L L^T (lower, in place)
    for (int j = 0; j < NUT; ++j) {
      WG_FOR(ctx, it, 1) {
        double dj = w.Lam[j][j];
        for (int l = 0; l < j; ++l) dj -= w.Lam[j][l] * w.Lam[j][l];
        if (!(dj > 0.0)) { w.ok = 0; dj = 1.0; }
        w.Lam[j][j] = sqrt(dj);
      }
      WG_SYNC(ctx);
      WG_FOR(ctx, it, NUT - 1 - j) {
        const int i = j + 1 + it;
        double s = w.Lam[i][j];
        for (int l = 0; l < j; ++l) s -= w.Lam[i][l] * w.Lam[j][l];
        w.Lam[i][j] = s / w.Lam[j][j];
      }
      WG_SYNC(ctx);
    }
    // K = -Lam^-1 G, kv = -Lam^-1 g: one item per right-hand side
    WG_FOR(ctx, c, NX + 1) {   // solved in place in the item's own column of Km / kv (no private arrays)
      for (int i = 0; i < NUT; ++i) {
        double s = c < NX ? w.Gm[i][c] : w.gv[i];
        for (int l = 0; l < i; ++l) s -= w.Lam[i][l] * (c < NX ? w.Km[l][c] : w.kv[l]);
        s /= w.Lam[i][i];
        if (c < NX) w.Km[i][c] = s; else w.kv[i] = s;
      }
      for (int i = NUT - 1; i >= 0; --i) {
        double s = c < NX ? w.Km[i][c] : w.kv[i];
        for (int l = i + 1; l < NUT; ++l) s -= w.Lam[l][i] * (c < NX ? w.Km[l][c] : w.kv[l]);
        s /= w.Lam[i][i];
        if (c < NX) w.Km[i][c] = s; else w.kv[i] = s;
      }
      for (int i = 0; i < NUT; ++i) { if (c < NX) w.Km[i][c] = -w.Km[i][c]; else w.kv[i] = -w.kv[i]; }
    }
    WG_SYNC(ctx);
    // Sn = Q + A^T SA + G^T K (symmetrised on the next phase), sn = q + A^T sb + G^T kv ; store K, kv
    double* rk = ric + (size_t)k * RIC_SIZE;
    WG_FOR(ctx, i, NX * (NX + 1) + NUT * NX + NUT) {
      if (i < NX * (NX + 1)) {
        const int r = i / (NX + 1), c = i % (NX + 1);
        if (c < NX) {
          double s = q[QP_Q + r * NX + c];
          for (int l = 0; l < NX; ++l) s += w.A[l][r] * w.SA[l][c];
          for (int l = 0; l < NUT; ++l) s += w.Gm[l][r] * w.Km[l][c];
          w.Sn[r][c] = s;
        } else {
          double s = q[QP_QV + r];
          for (int l = 0; l < NX; ++l) s += w.A[l][r] * w.sb[l];
          for (int l = 0; l < NUT; ++l) s += w.Gm[l][r] * w.kv[l];
          w.dxn[r] = s;  // staged: sv is still being read by nobody, but keep phases simple
        }
      } else if (i < NX * (NX + 1) + NUT * NX) {
        const int j = i - NX * (NX + 1);
        rk[RIC_K + j] = w.Km[j / NX][j % NX];
      } else {
        const int j = i - NX * (NX + 1) - NUT * NX;
        rk[RIC_KV + j] = w.kv[j];
      }
    }
    WG_SYNC(ctx);
    WG_FOR(ctx, i, NX * NX + NX) {
      if (i < NX * NX) { const int r = i / NX, c = i % NX; w.S[r][c] = 0.5 * (w.Sn[r][c] + w.Sn[c][r]); }
      else w.sv[i - NX * NX] = w.dxn[i - NX * NX];
    }
    WG_SYNC(ctx);
  }
}

// Forward roll-out of the QP solution and the full step.  x,u: linearisation trajectory of the instance;
// outputs dx [N+1][58], du [N][35], ut [N][23], x_new, u_new (any of the output pointers may alias nothing).
HSQP_HD void riccati_forward(const Ctx& ctx, RicWS& w, const double* x_init, const double* x, const double* u, const double* qp,
                             const double* ric, int N, double alpha, double* dx_out, double* du_out, double* ut_out, double* x_new,
                             double* u_new) {
  WG_FOR(ctx, i, NX) {
    const double d = x_init[i] - x[i];
    w.dx[i] = d;
    dx_out[i] = d;
    x_new[i] = x[i] + alpha * d;
  }
  WG_SYNC(ctx);
  for (int k = 0; k < N; ++k) {
    const double* q = qp + (size_t)k * QP_SIZE;
    const double* rk = ric + (size_t)k * RIC_SIZE;
    WG_FOR(ctx, i, NUT) {
      double s = rk[RIC_KV + i];
      for (int l = 0; l < NX; ++l) s += rk[RIC_K + i * NX + l] * w.dx[l];
      w.ut[i] = s;
      ut_out[(size_t)k * NUT + i] = s;
    }
    WG_SYNC(ctx);
    WG_FOR(ctx, i, NX + NU) {
      if (i < NX) {
        double s = q[QP_BV + i];
        for (int l = 0; l < NX; ++l) s += q[QP_A + i * NX + l] * w.dx[l];
        for (int l = 0; l < NUT; ++l) s += q[QP_B + i * NUT + l] * w.ut[l];
        w.dxn[i] = s;
        dx_out[(size_t)(k + 1) * NX + i] = s;
        x_new[(size_t)(k + 1) * NX + i] = x[(size_t)(k + 1) * NX + i] + alpha * s;
      } else {
        const int r = i - NX;
        double s = q[QP_PE + r];
        for (int l = 0; l < NX; ++l) s += q[QP_PX + r * NX + l] * w.dx[l];
        for (int l = 0; l < NUT; ++l) s += q[QP_PU + r * NUT + l] * w.ut[l];
        du_out[(size_t)k * NU + r] = s;
        u_new[(size_t)k * NU + r] = u[(size_t)k * NU + r] + alpha * s;
      }
    }
    WG_SYNC(ctx);
    WG_FOR(ctx, i, NX) w.dx[i] = w.dxn[i];
    WG_SYNC(ctx);
  }
}

// KKT residual of the projected QP at (dx, ut): costates by the backward stationarity recursion
//   lam_N = Qf dx_N + g_N,  lam_k = Q~ dx + P~^T ut + q~ + A~^T lam+   (x-stationarity holds by construction)
// reported: max | R~ ut + P~ dx + r~ + B~^T lam+ |  and  max | dx+ - A~ dx - B~ ut - b~ |, | dx_0 - (x_init - x_0) |.
HSQP_HD void kkt_residual(const Ctx& ctx, RicWS& w, const double* Qf, const double* x_init, const double* x, const double* parN,
                          const double* qp, const double* dx, const double* ut, int N, double* out2) {
  WG_FOR(ctx, i, NX) {
    const double dN = dx[(size_t)N * NX + i];
    w.sv[i] = Qf[i] * dN + Qf[i] * (x[(size_t)N * NX + i] - parN[HSQP_P_XDES + i]);
    w.sb[i] = fabs(dx[i] - (x_init[i] - x[i]));   // primal residual accumulator (per lane-owned slot)
    w.bt[i] = 0.0;                                 // stationarity accumulator
  }
  WG_SYNC(ctx);
  for (int k = N - 1; k >= 0; --k) {
    const double* q = qp + (size_t)k * QP_SIZE;
    const double* dxk = dx + (size_t)k * NX;
    const double* dxn = dx + (size_t)(k + 1) * NX;
    const double* utk = ut + (size_t)k * NUT;
    WG_FOR(ctx, i, NX + NUT) {
      if (i < NX) {
        double l = q[QP_QV + i];
        double pr = dxn[i] - q[QP_BV + i];
        for (int j = 0; j < NX; ++j) { l += q[QP_Q + i * NX + j] * dxk[j] + q[QP_A + j * NX + i] * w.sv[j]; pr -= q[QP_A + i * NX + j] * dxk[j]; }
        for (int j = 0; j < NUT; ++j) { l += q[QP_P + j * NX + i] * utk[j]; pr -= q[QP_B + i * NUT + j] * utk[j]; }
        w.dxn[i] = l;
        w.sb[i] = fmax(w.sb[i], fabs(pr));
      } else {
        const int r = i - NX;
        double s = q[QP_RV + r];
        for (int j = 0; j < NX; ++j) s += q[QP_P + r * NX + j] * dxk[j] + q[QP_B + j * NUT + r] * w.sv[j];
        for (int j = 0; j < NUT; ++j) s += q[QP_R + r * NUT + j] * utk[j];
        w.bt[r] = fmax(w.bt[r], fabs(s));
      }
    }
    WG_SYNC(ctx);
    WG_FOR(ctx, i, NX) w.sv[i] = w.dxn[i];
    WG_SYNC(ctx);
  }
  WG_FOR(ctx, it, 1) {
    double st = 0.0, pr = 0.0;
    for (int i = 0; i < NX; ++i) pr = fmax(pr, w.sb[i]);
    for (int i = 0; i < NUT; ++i) st = fmax(st, w.bt[i]);
    out2[0] = st;
    out2[1] = pr;
  }
  WG_SYNC(ctx);
}

}  // namespace hsqp
