// Riccati recursion of the projected, equality-free stage QP (the reference solves
// the same QP with HPIPM through ocs2's HpipmInterface — the minimiser is unique, A.1):
//   backward  Lam = R~ + B~^T S+ B~ (Cholesky),  K = -Lam^-1 (P~ + B~^T S+ A~),  k = -Lam^-1 (r~ + B~^T (s+ + S+ b~))
//             S = Q~ + A~^T S+ A~ + G^T K,  s = q~ + A~^T (s+ + S+ b~) + G^T k,   S_N = diag(Qf), s_N = Qf (x_N - x_des)
//   forward   ut = K dx + k,  dx+ = A~ dx + B~ ut + b~,  du = Px dx + Pu ut + Pe
// One workgroup per MPC instance; the stage matrices live in LDS for the whole backward step and every
// product is a register-tiled X^T Y contraction (hsqp_linalg.h):
//   SA = S^T A, SB = S^T B (S symmetric), Lam = R + B^T SB, G = P + B^T SA, Z = (L^-T)^T G, K = -(L^-1)^T Z,
//   S <- Q + A^T SA + G^T K (upper tiles, mirrored).
#pragma once
#include "hsqp_linalg.h"
#include "hsqp_project.h"

namespace hsqp {

constexpr int RIC_K = 0;                     // [23][58]
constexpr int RIC_KV = RIC_K + NUT * NX;     // [23]
constexpr int RIC_SIZE = ((RIC_KV + NUT + 7) / 8) * 8;
constexpr int LDB = 24;                      // leading dimension of the 23-wide LDS matrices (16-byte aligned rows)

struct RicWS {
  double S[NX][NX], A[NX][NX], SA[NX][NX];
  double B[NX][LDB];
  union {
    double SB[NX][LDB];
    double Z[NUT][NX];                       // L^-1 G (SB is dead once Lam and G are formed)
  };
  double Gm[NUT][NX], Km[NUT][NX];
  double Lam[LDB][LDB], M1[LDB][LDB], M1T[LDB][LDB];   // Cholesky factor (lower, in place), L^-1 and its transpose
  double dsq[LDB];
  double sv[NX], sn[NX], sb[NX], bt[NX], gv[NUT], kv[NUT], zv[NUT], dx[NX], dxn[NX], ut[NUT];
  double part[(NX + NU) * 4];
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
    double* rk = ric + (size_t)k * RIC_SIZE;
    PH_TICK(ctx, 9);
    // ---- P1: stage data -> LDS (coalesced)
    WG_FOR(ctx, i, NX * NX + NX * LDB + NX) {
      if (i < NX * NX) w.A[i / NX][i % NX] = q[QP_A + i];
      else if (i < NX * NX + NX * LDB) { const int j = i - NX * NX, r = j / LDB, c = j % LDB; w.B[r][c] = c < NUT ? q[QP_B + r * NUT + c] : 0.0; }
      else w.bt[i - NX * NX - NX * LDB] = q[QP_BV + i - NX * NX - NX * LDB];
    }
    WG_SYNC(ctx);
    PH_TICK(ctx, 1);
    // ---- P2: SA = S A, SB = S B (S symmetric => X = S), sb = s + S b
    wg_xty<4, 4>(ctx, NX, NX, NX, &w.S[0][0], NX, &w.A[0][0], NX, AllTiles(), [&](int r, int c, double v) { w.SA[r][c] = v; });
    wg_xty<4, 4>(ctx, NX, NUT, NX, &w.S[0][0], NX, &w.B[0][0], LDB, AllTiles(), [&](int r, int c, double v) { w.SB[r][c] = v; });
    WG_FOR(ctx, r, NX) {
      double s = w.sv[r];
      for (int l = 0; l < NX; ++l) s += w.S[l][r] * w.bt[l];
      w.sb[r] = s;
    }
    WG_SYNC(ctx);
    PH_TICK(ctx, 2);
    // ---- P3: Lam = R + B^T SB, G = P + B^T SA, g = r + B^T sb
    wg_xty<4, 4>(ctx, NUT, NUT, NX, &w.B[0][0], LDB, &w.SB[0][0], LDB, AllTiles(),
                 [&](int r, int c, double v) { w.Lam[r][c] = v + q[QP_R + r * NUT + c]; });
    wg_xty<4, 4>(ctx, NUT, NX, NX, &w.B[0][0], LDB, &w.SA[0][0], NX, AllTiles(),
                 [&](int r, int c, double v) { w.Gm[r][c] = v + q[QP_P + r * NX + c]; });
    WG_FOR(ctx, r, NUT) {
      double s = q[QP_RV + r];
      for (int l = 0; l < NX; ++l) s += w.B[l][r] * w.sb[l];
      w.gv[r] = s;
    }
    WG_SYNC(ctx);
    PH_TICK(ctx, 3);
    // ---- P4: right-looking Cholesky of Lam (lower triangle; the diagonal square roots go to dsq)
    for (int j = 0; j < NUT; ++j) {
      WG_FOR(ctx, it, NUT - j) {
        const int i = j + it;
        double dj = w.Lam[j][j];
        if (!(dj > 0.0)) { dj = 1.0; if (it == 0) w.ok = 0; }
        const double sq = sqrt(dj);
        if (it == 0) w.dsq[j] = sq; else w.Lam[i][j] = w.Lam[i][j] / sq;
      }
      WG_SYNC(ctx);
      const int m = NUT - 1 - j;
      WG_FOR(ctx, it, m * m) {
        const int i = j + 1 + it / m, c = j + 1 + it % m;
        if (c <= i) w.Lam[i][c] -= w.Lam[i][j] * w.Lam[c][j];
      }
      WG_SYNC(ctx);
    }
    PH_TICK(ctx, 4);
    // ---- P5: M1 = L^-1 (lower) and its transpose, one column per item (forward substitution)
    WG_FOR(ctx, j, NUT) {
      for (int i = 0; i < NUT; ++i) {
        double s = 0.0;
        if (i >= j) {
          s = i == j ? 1.0 : 0.0;
          for (int l = j; l < i; ++l) s -= w.Lam[i][l] * w.M1[l][j];
          s /= w.dsq[i];
        }
        w.M1[i][j] = s;
        w.M1T[j][i] = s;
      }
    }
    WG_SYNC(ctx);
    PH_TICK(ctx, 5);
    // ---- P6: Z = L^-1 G = (M1T)^T G, z = L^-1 g
    wg_xty<4, 4>(ctx, NUT, NX, NUT, &w.M1T[0][0], LDB, &w.Gm[0][0], NX, AllTiles(), [&](int r, int c, double v) { w.Z[r][c] = v; });
    WG_FOR(ctx, r, NUT) {
      double s = 0.0;
      for (int l = 0; l <= r; ++l) s += w.M1[r][l] * w.gv[l];
      w.zv[r] = s;
    }
    WG_SYNC(ctx);
    PH_TICK(ctx, 6);
    // ---- P7: K = -L^-T Z = -(M1)^T Z, k = -L^-T z ; stored for the forward pass
    wg_xty<4, 4>(ctx, NUT, NX, NUT, &w.M1[0][0], LDB, &w.Z[0][0], NX, AllTiles(),
                 [&](int r, int c, double v) { w.Km[r][c] = -v; rk[RIC_K + r * NX + c] = -v; });
    WG_FOR(ctx, r, NUT) {
      double s = 0.0;
      for (int l = r; l < NUT; ++l) s += w.M1[l][r] * w.zv[l];
      w.kv[r] = -s;
      rk[RIC_KV + r] = -s;
    }
    WG_SYNC(ctx);
    PH_TICK(ctx, 7);
    // ---- P8: S <- Q + A^T SA + G^T K (upper tiles; S itself is dead since P2), s <- q + A^T sb + G^T k
    wg_xty2<4, 4>(ctx, NX, NX, NX, &w.A[0][0], NX, &w.SA[0][0], NX, NUT, &w.Gm[0][0], NX, &w.Km[0][0], NX, UpperTiles(),
                  [&](int r, int c, double v) { if (c >= r) w.S[r][c] = v + q[QP_Q + r * NX + c]; });
    WG_FOR(ctx, r, NX) {
      double s = q[QP_QV + r];
      for (int l = 0; l < NX; ++l) s += w.A[l][r] * w.sb[l];
      for (int l = 0; l < NUT; ++l) s += w.Gm[l][r] * w.kv[l];
      w.sn[r] = s;
    }
    WG_SYNC(ctx);
    PH_TICK(ctx, 8);
    // ---- P9: mirror the upper triangle, roll s
    WG_FOR(ctx, i, NX * NX + NX) {
      if (i < NX * NX) { const int r = i / NX, c = i % NX; if (r > c) w.S[r][c] = w.S[c][r]; }
      else w.sv[i - NX * NX] = w.sn[i - NX * NX];
    }
    WG_SYNC(ctx);
  }
}

// Forward roll-out of the QP solution and the step of length alpha.  x,u: linearisation trajectory of the instance;
// outputs dx [N+1][58], du [N][35], ut [N][23], x_new, u_new.  Matrix-vector products read the stage matrices from
// global memory with 4 work items per row and a deterministic two-phase reduction.
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
    wg_matvec_partial(ctx, NUT, NX, rk + RIC_K, NX, w.dx, w.part);
    WG_SYNC(ctx);
    WG_FOR(ctx, i, NUT) {
      const double s = rk[RIC_KV + i] + ((w.part[4 * i] + w.part[4 * i + 1]) + (w.part[4 * i + 2] + w.part[4 * i + 3]));
      w.ut[i] = s;
      ut_out[(size_t)k * NUT + i] = s;
    }
    WG_SYNC(ctx);
    WG_FOR(ctx, it, (NX + NU) * 4) {
      const int r = it >> 2, p = it & 3;
      const double* Ax = r < NX ? q + QP_A + r * NX : q + QP_PX + (r - NX) * NX;
      const double* Bx = r < NX ? q + QP_B + r * NUT : q + QP_PU + (r - NX) * NUT;
      double s = 0.0;
      for (int c = p; c < NX; c += 4) s += Ax[c] * w.dx[c];
      for (int c = p; c < NUT; c += 4) s += Bx[c] * w.ut[c];
      w.part[it] = s;
    }
    WG_SYNC(ctx);
    WG_FOR(ctx, i, NX + NU) {
      const double s4 = (w.part[4 * i] + w.part[4 * i + 1]) + (w.part[4 * i + 2] + w.part[4 * i + 3]);
      if (i < NX) {
        const double s = q[QP_BV + i] + s4;
        w.dxn[i] = s;
        dx_out[(size_t)(k + 1) * NX + i] = s;
        x_new[(size_t)(k + 1) * NX + i] = x[(size_t)(k + 1) * NX + i] + alpha * s;
      } else {
        const int r = i - NX;
        const double s = q[QP_PE + r] + s4;
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
    w.sb[i] = fabs(dx[i] - (x_init[i] - x[i]));   // primal residual accumulator (one slot per row)
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
