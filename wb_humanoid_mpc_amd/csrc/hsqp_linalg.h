// Workgroup-cooperative dense kernels on LDS-resident matrices.
//
// Every product of the projection and Riccati kernels is brought to the form  C = X^T Y  with X (L x M)
// and Y (L x N) row-major, i.e. the contraction runs over ROWS of both operands (S and Lam^-1 are symmetric,
// residual rows are stored row-wise, B/A/G are used transposed).  Then both operands of a register tile are
// contiguous in memory:   acc[i][j] += X[l][r0+i] * Y[l][c0+j].
// VALU implementation: one work item = one TM x TN output tile with TM*TN independent accumulators
// (latency hiding without extra waves; 2 LDS reads per TM*TN/(TM+TN) FMAs).  Host-testable.
#pragma once
#include "hsqp_common.h"

namespace hsqp {

// Generic tile loop:  C = X1^T Y1 + sign2 * X2^T Y2  (second product optional: L2 = 0; sign2 = +1 or -1).
// A work item owns the STRIDED tile  rows {tr + i*tm}, cols {tc + j*tn}  (tm = ceil(M/TM), tn = ceil(N/TN)):
// consecutive lanes then read consecutive LDS words of a row of X / Y (conflict-free ds_read_b64) instead of
// words 8*TN bytes apart.  store(r, c, value) is called for every in-range element of the tile.
template <int TM, int TN, class Store>
HSQP_HD void wg_xty2(const Ctx& ctx, int M, int N, int L1, const double* X1, int ldx1, const double* Y1, int ldy1, int L2,
                     const double* X2, int ldx2, const double* Y2, int ldy2, double sign2, Store store) {
  const int tm = (M + TM - 1) / TM, tn = (N + TN - 1) / TN;
  WG_FOR(ctx, t, tm * tn) {
    const int tr = t / tn, tc = t % tn;
    int xo[TM], yo[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) xo[i] = (tr + i * tm < M) ? tr + i * tm : M - 1;   // clamp: duplicates are computed, never stored
#pragma unroll
    for (int j = 0; j < TN; ++j) yo[j] = (tc + j * tn < N) ? tc + j * tn : N - 1;
    double acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = 0.0;
    for (int seg = 0; seg < 2; ++seg) {
      const int L = seg ? L2 : L1;
      const double* X = seg ? X2 : X1;
      const double* Y = seg ? Y2 : Y1;
      const int ldx = seg ? ldx2 : ldx1, ldy = seg ? ldy2 : ldy1;
#pragma unroll 2
      for (int l = 0; l < L; ++l) {
        double a[TM], b[TN];
        const double* xr = X + l * ldx;
        const double* yr = Y + l * ldy;
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = seg ? sign2 * xr[xo[i]] : xr[xo[i]];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = yr[yo[j]];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] += a[i] * b[j];
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        if (tr + i * tm < M && tc + j * tn < N) store(tr + i * tm, tc + j * tn, acc[i][j]);
  }
}

template <int TM, int TN, class Store>
HSQP_HD void wg_xty(const Ctx& ctx, int M, int N, int L, const double* X, int ldx, const double* Y, int ldy, Store store) {
  wg_xty2<TM, TN>(ctx, M, N, L, X, ldx, Y, ldy, 0, X, ldx, Y, ldy, 1.0, store);
}

// y[r] (+)= sum_c A[r][c] x[c] for a row-major matrix in GLOBAL memory, 4 work items per row with a
// deterministic two-phase reduction through `part` (LDS, rows x 4).
HSQP_HD void wg_matvec_partial(const Ctx& ctx, int rows, int cols, const double* A, int lda, const double* x, double* part) {
  WG_FOR(ctx, it, rows * 4) {
    const int r = it >> 2, p = it & 3;
    double s = 0.0;
    for (int c = p; c < cols; c += 4) s += A[r * lda + c] * x[c];
    part[it] = s;
  }
}

}  // namespace hsqp
