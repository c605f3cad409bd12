// Two-level (segmented) backward sweep of the stage QP — the strong-scaling form for 2 < B <= 64 instances per GPU (BASELINE config 4 as
// written: 256 instances over 8 GPUs = 32 per GPU).  One workgroup per instance leaves 256 - B CUs idle during the N
// dependent stages of k_riccati, and the parallel scan over ALL stages (hsqp_scan.h) needs B (N + 1) workgroups per level.  Here the
// horizon of every instance is cut into P = 256 / B segments, bounds k_p = p N / P:
//   1a. k_seg_elem_ric      per segment: the ordinary Riccati recursion (riccati_backward) over its stages started from J = 0, eta = 0 —
//                           the (J, eta) part of the segment's conditional value function; keeps K, k and (L^-1)^T of every stage;
//   1b. k_seg_accumulate    per segment: the (A, b, C) part by PREPENDING the stages one at a time (seg_accumulate below).  A stage is
//                           an element whose C1 = B R^-1 B' has rank nu, so M^-1 = (I + C1 J2)^-1 of the generic combination follows from
//                           the gain solve of 1a:  M^-1 A1 = A + B K = Acl,  M^-1 (b1 + C1 eta2) = b + B k = bcl,  M^-1 C1 = B Lam^-1 B', hence
//                               A_seg <- A_seg Acl,   b_seg <- A_seg bcl + b_seg,   C_seg <- A_seg B Lam^-1 B' A_seg' + C_seg
//                           (oracle/parallel_scan.py::prepend_stage; checked against the generic combination there);
//   2.  k_scan_combine      the existing combination kernel on the P segment elements + the terminal element: ceil(log2(P + 1)) levels
//                           give the value function (J, -eta) at every segment boundary;
//   3.  k_seg_riccati       per segment: riccati_backward from the boundary value function at its end -> the gains K, k of its stages;
//       k_kkt_boundaries    the gate: KKT residual of the last stage of every segment (below);
//   4.  k_ric_forward       the roll-out (riccati_forward), one workgroup per instance.
// 1a, 1b and 3 run on B P workgroups at once.  Segments are short, so their elements are far better conditioned than the late levels of
// the full scan: the numpy prototype reproduces the serial recursion to 5e-12 of the step's scale on the whole-body problem (N = 100,
// P = 8) where the full scan gives 1e-7 (tests/test_parallel_scan.py).
#pragma once
#include "hsqp_scan.h"

namespace hsqp {

HSQP_HD int seg_bound(int p, int N, int P) { return (int)(((long long)p * N) / P); }

// Gate (hsqp_capi.hip::k_kkt_boundaries, scan_gate_accepts): inside a segment the gains come from an exact recursion started at the segment's end;
// what the scanned boundary value functions got wrong shows as the KKT residual of the LAST stage of every segment (its gains were derived from
// the scanned suffix, its successor costate is the value function the next segment's own recursion arrives at).  Those B (P - 1) stages are
// checked against the scan's bounds; a rejected sweep is redone with the serial recursion.  (A cheaper check — the difference between the
// scanned suffix and the value function the next segment's recursion arrives at — was measured and dropped: the step error follows
// Lam^-1 B' ds, whose amplification varies by 1e3 between problems, so no bound on |ds| separates good sweeps from bad ones.)

// Workspace of pass 1b (n = 58: 153 KB of the 160 KB of LDS).  T = A_seg' (so that every product is X^T Y on row-major operands).
struct SegAccWS {
  double T[2][NX][NX];          // A_seg' of the stages processed so far, double-buffered (T_new reads T)
  double C[NX][NX];             // C_seg
  double A[NX][NX];             // A~ of the stage
  double B[NX][LDB];            // B~
  double K[NUT][NX];            // gain of the stage from pass 1a (terminal J = 0)
  double U[NUT][NX];            // B' T
  double Wt[NUT][NX];           // L^-1 U = (A_seg B L^-T)'
  double LinvT[LDB][LDB];
  double bseg[2][NX], bcl[NX], bt[NX], kv[LDB];
};
static_assert(sizeof(SegAccWS) <= 163400, "LDS budget of pass 1b");

// qp / ric / linv: the records of the segment's stages (stage 0 = first stage of the segment), L stages; vf0: (J, s) at the segment's
// first node from pass 1a (VF_SIZE doubles: S row-major with leading dimension NX, then s); el: the segment's element out (ScanEl<n>).
template <int n>
HSQP_HD void seg_accumulate(const Ctx& ctx, SegAccWS& w, const double* qp, const double* ric, const double* linv, int L, const double* vf0, double* el) {
  using E = ScanEl<n>;
  // identity element of the empty interval: A_seg = I, b_seg = 0, C_seg = 0
  WG_FOR(ctx, i, NX * NX + NX) {
    if (i < NX * NX) { const int r = i / NX, c = i % NX; w.T[0][r][c] = r == c ? 1.0 : 0.0; w.C[r][c] = 0.0; }
    else w.bseg[0][i - NX * NX] = 0.0;
  }
  WG_SYNC(ctx);
  int cur = 0;
  for (int k = L - 1; k >= 0; --k) {
    const double* q = qp + (size_t)k * QP_SIZE;
    const double* rk = ric + (size_t)k * RIC_SIZE;
    const double* lk = linv + (size_t)k * LDB * LDB;
    double(*T)[NX] = w.T[cur];
    double(*Tn)[NX] = w.T[1 - cur];
    // ---- stage data -> LDS (batched loads: eight in flight per item)
    {
      constexpr int na = nbatches(NX * NX, 8), nk = nbatches(NUT * NX, 8);
      WG_FOR(ctx, it, na + nk + NX * LDB + LDB * LDB + NX + LDB) {
        if (it < na) copy_batch<8>(it, NX * NX, q + QP_A, [&](int i, double v) { w.A[i / NX][i % NX] = v; });
        else if (it < na + nk) copy_batch<8>(it - na, NUT * NX, rk + RIC_K, [&](int i, double v) { w.K[i / NX][i % NX] = v; });
        else if (it < na + nk + NX * LDB) { const int i = it - na - nk, r = i / LDB, c = i % LDB; w.B[r][c] = c < NUT ? q[QP_B + r * NUT + c] : 0.0; }
        else if (it < na + nk + NX * LDB + LDB * LDB) { const int i = it - na - nk - NX * LDB; w.LinvT[i / LDB][i % LDB] = lk[i]; }
        else if (it < na + nk + NX * LDB + LDB * LDB + NX) { const int i = it - na - nk - NX * LDB - LDB * LDB; w.bt[i] = q[QP_BV + i]; }
        else { const int i = it - na - nk - NX * LDB - LDB * LDB - NX; w.kv[i] = i < NUT ? rk[RIC_KV + i] : 0.0; }
      }
    }
    WG_SYNC(ctx);
    // ---- U = B' T (nu x n); bcl = b + B k
    {
      const XtyJob job = xty_job(NUT, n, n, &w.B[0][0], LDB, &T[0][0], NX, &w.U[0][0], NX);
      wg_xty_jobs<true>(ctx, &job, 1);
      WG_FOR(ctx, i, n) {
        double s = w.bt[i];
#pragma unroll
        for (int l = 0; l < NUT; ++l) s += w.B[i][l] * w.kv[l];
        w.bcl[i] = s;
      }
    }
    WG_SYNC(ctx);
    // ---- T_new = Acl' T = A' T + K' U;  Wt = L^-1 U;  b_seg <- T' bcl + b_seg   (all read the OLD T)
    {
      XtyJob jt = xty_job(n, n, n, &w.A[0][0], NX, &T[0][0], NX, &Tn[0][0], NX);
      jt.L2 = NUT; jt.X2 = &w.K[0][0]; jt.ldx2 = NX; jt.Y2 = &w.U[0][0]; jt.ldy2 = NX; jt.sign2 = 1.0;
      const XtyJob jobs[2] = {jt, xty_job(NUT, n, NUT, &w.LinvT[0][0], LDB, &w.U[0][0], NX, &w.Wt[0][0], NX)};
      wg_xty_jobs<true>(ctx, jobs, 2);
      WG_FOR(ctx, i, n) {
        double s = w.bseg[cur][i];
        for (int l = 0; l < n; ++l) s += T[l][i] * w.bcl[l];
        w.bseg[1 - cur][i] = s;
      }
    }
    WG_SYNC(ctx);
    // ---- C_seg += Wt' Wt (symmetric: tiles on / above the diagonal, mirrored)
    {
      XtyJob jc = xty_job(n, n, NUT, &w.Wt[0][0], NX, &w.Wt[0][0], NX, &w.C[0][0], NX, &w.C[0][0], NX);
      jc.sym = 1;
      wg_xty_jobs<true>(ctx, &jc, 1);
    }
    WG_SYNC(ctx);
    cur = 1 - cur;
  }
  // ---- the element: A = T', b, C; J, eta = -s from pass 1a
  WG_FOR(ctx, i, E::SIZE) {
    double v = 0.0;
    if (i < E::C) { const int r = (i - E::A) / n, c = (i - E::A) % n; v = w.T[cur][c][r]; }
    else if (i < E::J) { const int r = (i - E::C) / n, c = (i - E::C) % n; v = w.C[r][c]; }
    else if (i < E::B) { const int r = (i - E::J) / n, c = (i - E::J) % n; v = vf0[r * NX + c]; }
    else if (i < E::ETA) v = w.bseg[cur][i - E::B];
    else if (i < E::ETA + n) v = -vf0[NX * NX + (i - E::ETA)];
    el[i] = v;
  }
  WG_SYNC(ctx);
}

}  // namespace hsqp
