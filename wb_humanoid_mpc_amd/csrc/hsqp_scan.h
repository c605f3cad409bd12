// Parallel-in-time backward sweep of the stage QP (BASELINE north star: "the serial Riccati recursion run as a cyclic-reduction /
// parallel-scan over stages"): the value functions S_k, s_k of all nodes from an associative scan over conditional value functions
// instead of N dependent Riccati stages.  Used for one or two instances on a long horizon (both formulations: n = 35 and n = 58), where
// the serial chain of N = 100 stages leaves all but one or two CUs idle (config 3: 1.67 of 1.93 ms; config 2: 0.81 of 1.29 ms).
//
// Formulation (Särkkä & García-Fernández, "Temporal parallelization of dynamic programming and linear quadratic control",
// IEEE TAC 2023; algebra restated and checked in oracle/parallel_scan.py): an element (A, b, C, eta, J) of the interval i -> j stands
// for   V_{i->j}(x_i, x_j) = const + 1/2 x_i' J x_i - eta' x_i + max_lam { -1/2 lam' C lam - lam' (x_j - A x_i - b) } ;
// i -> j combined with j -> k:
//     M = I + C1 J2,  [XA | XC | xb] = M^-1 [A1 | C1 | b1 + C1 eta2],  y = eta2 - J2 b1
//     A = A2 XA,  b = A2 xb + b2,  C = A2 XC A2' + C2,  J = A1' J2 XA + J1,  eta = A1' (y - J2 XC y) + eta1 ;
// the value function of node k is S_k = J, s_k = -eta of the suffix k -> N+1.  One stage x+ = A x + B u + b with cost
// 1/2 x'Qx + u'Px + 1/2 u'Ru + q'x + r'u is the element (A - B R^-1 P, b - B R^-1 r, B R^-1 B', -(q - P' R^-1 r), Q - P' R^-1 P), the
// terminal cost the element (0, 0, 0, -q_N, Q_N).
//
// Kernels (hsqp_capi.hip): k_scan_init (one workgroup per node: stage -> element), k_scan_combine (one per node and level:
// Hillis-Steele suffix scan, ceil(log2(N+1)) levels, ping-pong element buffers), k_scan_gains (one per node: ONE stage of the
// existing Riccati code started from S_{k+1}, s_{k+1} -> K, k, S_k for the KKT check, and the stage's closed loop), k_scan_forward
// (the roll-out dx+ = A_cl dx + b_cl, closed_loop_forward in hsqp_riccati.h).  The record the step / KKT kernels read is the one
// k_riccati writes.
//
// Numerics: M is solved by Gauss-Jordan elimination with scaled row pivoting, in registers (gauss_jordan_pipeline below; the LDS form
// gauss_jordan is what the host build runs); the symmetric results are symmetrised by averaging, which removes the antisymmetric part
// of the elimination's error.  cond(M) <= 1e5 on the projected QPs of the centroidal problem, up to 1e9 on the whole-body problem: on
// the QPs of a cold start / a tracking MPC the scan reproduces the serial recursion to 1e-11 .. 1e-10 of the step's scale, on
// far-from-feasible line-search iterates it loses up to five digits — hsqp_iterate_device therefore gates every scan result by the KKT
// residual of the QP (scan_gate_accepts below) and falls back to the serial recursion.
#pragma once
#include "hsqp_riccati.h"

namespace hsqp {
constexpr int SCAN_PF = 3;   // operand prefetch depth of the scan kernels' tile loops (hsqp_linalg.h): one workgroup per element, latency-bound like the Riccati stage


// KKT gate of the parallel-in-time sweep: the stationarity / primal residuals of the QP must be below BOTH HSQP_SCAN_GATE_REL max(1, |g|_inf)
// (BASELINE.md §6's criterion for a QP solution) and HSQP_SCAN_GATE_ABS.  The absolute bound is what separates the two populations seen on
// this problem (the QP's units are fixed by the model): where the scan reproduces the serial recursion to <= 2e-10 of the step's scale its
// stationarity is 3e-11 .. 3.4e-9 (configs 2, 3, perturbed walk instances, N = 16 .. 100); where it loses digits — far-from-feasible
// line-search iterates, randomly perturbed run-gait QPs with |du| ~ 1e3 — it is 1.6e-7 .. 2e-4 (step errors 1e-8 .. 1e-5 of the scale),
// while the residual relative to |g|_inf still looks harmless there (3e-10) because |g|_inf is 1e4 .. 1e5.
constexpr double SCAN_GATE_REL = 1e-9, SCAN_GATE_ABS = 2e-8;
// r_stat / r_prim: KKT residuals of the instance's QP with the scanned value functions as costates (k_kkt), g_inf: |gradient of the projected
// QP|_inf, flags: what the scan kernels reported (bad pivot, failed Lam, rank-deficient D).  NaN residuals do not pass.
HSQP_HD bool scan_gate_accepts(double r_stat, double r_prim, double g_inf, int flags) {
  const double rel = SCAN_GATE_REL * (g_inf > 1.0 ? g_inf : 1.0), lim = rel < SCAN_GATE_ABS ? rel : SCAN_GATE_ABS;
  return flags == 0 && g_inf == g_inf && r_stat <= lim && r_prim <= lim;
}

// ---- element layout in global memory (doubles), dimension n = NXE, row-major with leading dimension n
template <int n> struct ScanEl {
  static constexpr int A = 0, C = A + n * n, J = C + n * n, B = J + n * n, ETA = B + n, SIZE = ((ETA + n + 7) / 8) * 8;
};

// Gauss-Jordan elimination of the n x ncol matrix G (LDS, leading dimension ld) whose first n columns hold M: afterwards row
// piv[j] holds (unscaled) the solution row j, G[piv[j]][j] its pivot.  PIVOT = false: diagonal pivots (symmetric positive definite
// M).  PIVOT = true: row pivoting scaled by the rows' initial magnitudes (implicit equilibration): the first unused row of largest
// scaled magnitude in column j.  ONE barrier per column in both cases: the items that update column j + 1 also publish the rows'
// pivot candidates for the next step (double-buffered; -1 for used rows), and after the barrier every thread takes the maximum of
// the n candidates itself (the same addresses in every lane: LDS broadcasts) instead of waiting for a dedicated search phase.
// Measured on the combination kernel (35 steps on 35 x 106, 512 threads, tools/phase_profile.py): search by one item or redundantly
// on the raw column 8 k cycles per step; search by wave 0 with cross-lane exchanges behind a second barrier 4.2 k; this form 3.8 k
// (1.6 k candidate scan, 2.2 k update, 0.3 k barrier) — still the largest part (60 %) of the combination.
struct GjWS { int piv[64]; double rscale[64]; float cand[2][64]; int ok; };
// ld, ncol are compile-time constants: the item -> (row, residue) map and every address offset fold into immediates
template <int n, int ld, int ncol, bool PIVOT>
HSQP_HD void gauss_jordan(const Ctx& ctx, double* G, GjWS& g) {
  static_assert(n <= 64, "row bookkeeping is a 64-bit mask");
  WG_FOR(ctx, i, 64 + 1) {
    if (i < 64) {
      double m = 0.0;
      if (PIVOT && i < n) for (int c = 0; c < n; ++c) m = fmax(m, fabs(G[i * ld + c]));
      const double sc = m > 0.0 ? 1.0 / m : 1.0;
      g.rscale[i] = sc;
      g.cand[0][i] = (PIVOT && i < n) ? (float)(fabs(G[i * ld]) * sc) : -1.0f;
      g.piv[i] = 0;
    } else g.ok = 1;
  }
  WG_SYNC(ctx);
  // fixed item grid over ALL columns: item (row i, residue ch) owns the columns ch + c NCH, c < CH, of its row for the whole
  // elimination (consecutive lanes = consecutive words of a row; no index arithmetic that depends on the step); columns <= j are
  // simply skipped.  The candidates are compared in single precision: the pivot only has to be large, not the largest.
  constexpr int CH = 8;
  constexpr int NCH = (ncol + CH - 1) / CH;
  unsigned long long used = 0ull;             // rows used as pivots so far: the same value in every thread
  for (int j = 0; j < n; ++j) {
    const float* cd = g.cand[j & 1];
    float* cn = g.cand[(j + 1) & 1];
    int p = j;
    if (PIVOT) {
      float c[n];
#pragma unroll
      for (int i = 0; i < n; ++i) c[i] = cd[i];
      float best = -1.0f;
      p = 0;
#pragma unroll
      for (int i = 0; i < n; ++i) if (c[i] > best) { best = c[i]; p = i; }
    }
    double pv = G[p * ld + j];
    const bool bad = !(fabs(pv) > 1e-300);
    if (bad) pv = 1.0;
    used |= 1ull << p;
    const double rpv = fast_rcp(pv);
    WG_FOR(ctx, it, n * NCH) {
      if (it == 0) { g.piv[j] = p; if (bad) g.ok = 0; }
      const int i = it / NCH, ch = it - i * NCH;
      if (i != p) {
        const double f = G[i * ld + j] * rpv;
        double a[CH], b[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) { const int cc = ch + c * NCH < ncol ? ch + c * NCH : ncol - 1; a[c] = G[i * ld + cc]; b[c] = G[p * ld + cc]; }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int cc = ch + c * NCH;
          if (cc > j && cc < ncol) {
            const double v = a[c] - f * b[c];
            G[i * ld + cc] = v;
            if (PIVOT && cc == j + 1 && cc < n) cn[i] = ((used >> i) & 1ull) ? -1.0f : (float)(fabs(v) * g.rscale[i]);
          }
        }
      } else if (PIVOT && j + 1 < n && ch == (j + 1) % NCH) cn[i] = -1.0f;
    }
    WG_SYNC(ctx);
  }
}

// LDS mailbox of gauss_jordan_pipeline (below)
struct GjPipe {
  static constexpr int CH = 8;
  double f[2][CH][64];     // multiplier of row (lane) i at step j of the chunk (0 for the pivot row itself and for the padding lanes)
  int piv[2][CH];          // pivot lane of the step
  int mycol[64];           // which solution row the lane's row became
  double rdiag[64];        // 1 / its pivot
};

#if defined(__HIP_DEVICE_COMPILE__)
// The same elimination with the matrix in REGISTERS (device, combination step): lane i of a wave holds row i — the n columns of M and
// a share NO of the right-hand-side columns — and NW waves each eliminate M redundantly (identical arithmetic, identical pivots) on
// their own share, so a step needs no barrier, no LDS traffic and no dynamic register index: the pivot row is a LANE, its elements come
// through v_readlane with that (wave-uniform) lane number, the pivot search is a DPP wave maximum of the packed
// (magnitude | lane) candidates, and each lane's multiplier is its own element of column j.  Rows are never swapped: a used row stays
// in its lane and remembers which solution row it is.  X[r][c] (leading dimension ldx)
// receives the solution row r of right-hand-side column c < nrhs.  Steps: ~0.9 k cycles instead of 3.8 k for the LDS form.
// load_m(row, c) / load_r(row, c): element of M / of the right-hand sides (from LDS or global memory).
// maximum of an unsigned value over the 64 lanes of the wave (wave-uniform result): Hillis-Steele inside the rows of 16 lanes with DPP
// row shifts, then the two row broadcasts of the GFX9 DPP set; a lane without a source keeps 0, the identity of the unsigned maximum
__device__ inline unsigned wave_umax(unsigned v) {
  auto step = [](unsigned x, int y) { return x > (unsigned)y ? x : (unsigned)y; };
  v = step(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));   // row_shr:1
  v = step(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));   // row_shr:2
  v = step(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));   // row_shr:4
  v = step(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));   // row_shr:8   -> lane 15 of a row: the row's maximum
  v = step(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1 and 3
  v = step(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2 and 3 -> lane 63: all
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

template <int n, int nrhs, int NW, bool PIVOT, class LoadM, class LoadR>
__device__ inline void gauss_jordan_rows(const Ctx& ctx, LoadM load_m, LoadR load_r, double* X, int ldx, int* okflag) {
  constexpr int NO = (nrhs + NW - 1) / NW;
  const int wave = wave_index(ctx.tid), lane = ctx.tid & 63;
  if (wave >= NW) return;
  const int row = lane < n ? lane : n - 1;
  const int c0 = wave * NO;
  double m[n], o[NO];
#pragma clang loop unroll(full)
  for (int c = 0; c < n; ++c) m[c] = load_m(row, c);
#pragma clang loop unroll(full)
  for (int c = 0; c < NO; ++c) o[c] = load_r(row, c0 + c < nrhs ? c0 + c : nrhs - 1);
  double mx = 0.0;
#pragma clang loop unroll(full)
  for (int c = 0; c < n; ++c) mx = fmax(mx, fabs(m[c]));
  const double rscale = mx > 0.0 ? 1.0 / mx : 1.0;
  bool used = lane >= n;          // the padding lanes never become pivots
  int mycol = 0;
  double dpiv = 1.0;
  bool good = true;
#pragma clang loop unroll(full)
  for (int j = 0; j < n; ++j) {
    // candidates: the magnitude as a float with its low 6 bits replaced by the lane number; used rows offer 0
    int p = j;                    // PIVOT = false: diagonal pivots (symmetric positive definite M)
    if (PIVOT) {
      const unsigned cand = used ? 0u : ((__float_as_uint((float)(fabs(m[j]) * rscale)) & ~63u) | (unsigned)lane);
      p = (int)(wave_umax(cand) & 63u);
    }
    double pv = readlane_f64(m[j], p);
    if (!(fabs(pv) > 1e-300)) { good = false; pv = 1.0; }
    const double rpv = fast_rcp(pv);
    const bool isp = lane == p;
    const double f = (isp || lane >= n) ? 0.0 : m[j] * rpv;
    if (isp) { used = true; mycol = j; dpiv = pv; }
#pragma clang loop unroll(full)
    for (int c = j + 1; c < n; ++c) m[c] -= f * readlane_f64(m[c], p);
#pragma clang loop unroll(full)
    for (int c = 0; c < NO; ++c) o[c] -= f * readlane_f64(o[c], p);
  }
  if (lane < n) {
    const double rd = 1.0 / dpiv;
#pragma clang loop unroll(full)
    for (int c = 0; c < NO; ++c) if (c0 + c < nrhs) X[mycol * ldx + c0 + c] = o[c] * rd;
  }
  if (!good && ctx.tid == 0) *okflag = 0;
}

// The same elimination as a PIPELINE over the waves of an 8-wave workgroup (combination step).  In gauss_jordan_rows every wave repeats
// the elimination of M (two thirds of its instructions at n = 58); here wave 0 alone eliminates M, CH steps at a time, and publishes
// per step the pivot lane and every row's multiplier (LDS, double-buffered per chunk); the other seven waves hold only right-hand-side
// columns and apply the chunk wave 0 finished one barrier earlier: o[c] -= f_row * (pivot row's o[c], by v_readlane).  One barrier per
// chunk, both sides take about the same time per step (M: 58 - j column updates on one wave; right-hand sides: 17 on each of seven).

template <int n, int nrhs, class LoadM, class LoadR>
__device__ inline void gauss_jordan_pipeline(const Ctx& ctx, GjPipe& g, LoadM load_m, LoadR load_r, double* X, int ldx, int* okflag) {
  constexpr int CH = GjPipe::CH, NCHK = (n + CH - 1) / CH, NRW = 7, NO = (nrhs + NRW - 1) / NRW;
  const int wave = wave_index(ctx.tid), lane = ctx.tid & 63;
  const int row = lane < n ? lane : n - 1;
  const int c0 = (wave - 1) * NO;
  double m[n], o[NO];
  double rscale = 1.0, dpiv = 1.0;
  bool used = lane >= n, good = true;
  int mycol = 0;
  if (wave == 0) {
#pragma clang loop unroll(full)
    for (int c = 0; c < n; ++c) m[c] = load_m(row, c);
    double mx = 0.0;
#pragma clang loop unroll(full)
    for (int c = 0; c < n; ++c) mx = fmax(mx, fabs(m[c]));
    rscale = mx > 0.0 ? 1.0 / mx : 1.0;
  } else {
#pragma clang loop unroll(full)
    for (int c = 0; c < NO; ++c) o[c] = load_r(row, c0 + c < nrhs ? c0 + c : nrhs - 1);
  }
#pragma clang loop unroll(full)
  for (int k = 0; k <= NCHK; ++k) {
    if (wave == 0) {
      if (k < NCHK) {
#pragma clang loop unroll(full)
        for (int jj = 0; jj < CH; ++jj) {
          const int j = k * CH + jj;
          if (j < n) {
            const unsigned cand = used ? 0u : ((__float_as_uint((float)(fabs(m[j]) * rscale)) & ~63u) | (unsigned)lane);
            const int p = (int)(wave_umax(cand) & 63u);
            double pv = readlane_f64(m[j], p);
            if (!(fabs(pv) > 1e-300)) { good = false; pv = 1.0; }
            const double rpv = fast_rcp(pv);
            const bool isp = lane == p;
            const double f = (isp || lane >= n) ? 0.0 : m[j] * rpv;
            if (isp) { used = true; mycol = j; dpiv = pv; }
            g.f[k & 1][jj][lane] = f;
            if (lane == 0) g.piv[k & 1][jj] = p;
#pragma clang loop unroll(full)
            for (int c = j + 1; c < n; ++c) m[c] -= f * readlane_f64(m[c], p);
          }
        }
        if (k == NCHK - 1) {
          g.mycol[lane] = mycol;
          g.rdiag[lane] = 1.0 / dpiv;
          if (!good && lane == 0) *okflag = 0;
        }
      }
    } else if (k > 0) {
      const int kc = k - 1, nst = (kc + 1) * CH <= n ? CH : n - kc * CH;
      for (int jj = 0; jj < nst; ++jj) {
        const int p = __builtin_amdgcn_readfirstlane(g.piv[kc & 1][jj]);
        const double f = g.f[kc & 1][jj][lane];
#pragma clang loop unroll(full)
        for (int c = 0; c < NO; ++c) o[c] -= f * readlane_f64(o[c], p);
      }
    }
    __syncthreads();
  }
  if (wave > 0 && lane < n) {
    const int r = g.mycol[lane];
    const double rd = g.rdiag[lane];
#pragma clang loop unroll(full)
    for (int c = 0; c < NO; ++c) if (c0 + c < nrhs) X[r * ldx + c0 + c] = o[c] * rd;
  }
}
#endif

// ---- stage -> element
// The element of one stage is its conditional value function with the input eliminated by R~ ALONE, and cond(R~) reaches 1e7 on the whole-body
// problem (input weights 1e-3 dt against the rows the projection leaves).  Round 2-4 formed R~^-1 explicitly (Gauss-Jordan) and multiplied:
// A - B R^-1 P, B R^-1 B', Q - P' R^-1 P.  That is where the scan lost its digits, not in the combinations (oracle/parallel_scan.py, numpy:
// the step error against the serial recursion on config 3's iterates 1.5e-11 / 4.7e-5 / 1.6e-3 with the explicit inverse, 1.8e-11 / 1.6e-8 /
// 2.9e-9 with the form below; symmetric "square-root" variants of the combination changed nothing: 6.2e-8 -> 6.3e-8).  The element is now
// formed the way the Riccati stage forms its own quantities — the stage PREPENDED to the empty interval (hsqp_segment.h), S = 0:
//     R~ = L L',  Z = L^-1 P~,  z = L^-1 r~,  W' = L^-1 B~'      (the blocked elimination of the Riccati stage, hsqp_elim.h, on [R~ | I | P~ | r~])
//     A = A~ - W Z,  b = b~ - W z,  C = W W',  J = Q~ - Z' Z,  eta = -(q~ - Z' z)
// every product a Gram-type contraction with L^-1 applied to both factors; C and J are symmetric by construction.
template <int n>
struct ScanInitWS {
  double Ef[LDB][LDF];            // [R~ | . | L^-1]
  double LinvT[LDB][LDB];
  double PG[NUT][n + 2];          // [P~ | r~]
  double Zs[NUT][n + 2];          // [Z | z]
  double BT[NUT][n + 1], Wt[NUT][n + 1];   // B~', W' = L^-1 B~'
  double zv[LDB];
  int ok;
};

// the element of the terminal cost 1/2 x'diag(Qf)x + qN'x
template <int n>
HSQP_HD void scan_terminal_element(const Ctx& ctx, double* el, const double* Qf, const double* xN, const double* parN) {
  using E = ScanEl<n>;
  WG_FOR(ctx, i, E::SIZE) {
    double v = 0.0;
    if (i >= E::J && i < E::J + n * n) { const int r = (i - E::J) / n, c = (i - E::J) % n; v = r == c ? Qf[r] : 0.0; }
    else if (i >= E::ETA && i < E::ETA + n) { const int r = i - E::ETA; v = -Qf[r] * (xN[r] - parN[HSQP_P_XDES + r]); }
    el[i] = v;
  }
  WG_SYNC(ctx);
}

// q: QP record of the stage (hsqp_project.h), el: element out.  terminal: the element of the terminal cost instead.
// *ok (optional) is cleared when R~ has a pivot that is not positive.
template <int n>
HSQP_HD void scan_init_node(const Ctx& ctx, ScanInitWS<n>& w, const double* q, double* el, bool terminal, const double* Qf, const double* xN,
                            const double* parN, int* ok = nullptr) {
  using E = ScanEl<n>;
  if (terminal) { scan_terminal_element<n>(ctx, el, Qf, xN, parN); return; }
  constexpr int LP = n + 2;
  WG_FOR(ctx, i, NUT * LDF + LDB * LDB + NUT * LP + NUT * (n + 1) + 1) {
    if (i < NUT * LDF) {
      const int r = i / LDF, c = i % LDF;
      w.Ef[r][c] = c < NUT ? q[QP_R + r * NUT + c] : 0.0;
    } else if (i < NUT * LDF + LDB * LDB) {
      const int j = i - NUT * LDF;
      w.LinvT[j / LDB][j % LDB] = 0.0;
    } else if (i < NUT * LDF + LDB * LDB + NUT * LP) {
      const int j = i - NUT * LDF - LDB * LDB, r = j / LP, c = j % LP;
      w.PG[r][c] = c < n ? q[QP_P + r * NX + c] : (c == n ? q[QP_RV + r] : 0.0);
    } else if (i < NUT * LDF + LDB * LDB + NUT * LP + NUT * (n + 1)) {
      const int j = i - NUT * LDF - LDB * LDB - NUT * LP, r = j / (n + 1), c = j % (n + 1);
      w.BT[r][c] = c < n ? q[QP_B + c * NUT + r] : 0.0;
    } else w.ok = 1;
  }
  WG_SYNC(ctx);
  {
    const ElimIO io{&w.Ef[0][0], LDF, &w.PG[0][0], &w.PG[0][n], LP, &w.Ef[0][EF_MI], LDF, &w.LinvT[0][0], LDB, &w.Zs[0][0], LP, w.zv, &w.ok};
#if defined(__HIP_DEVICE_COMPILE__)
    if (ctx.nthreads >= 128) {   // two waves, as in the Riccati stage
      const int wv = wave_index(ctx.tid);
      const DevWave dw{ctx.tid & 63};
      if (wv == 0) eliminate_blocked<n, 0>(dw, io);
      else if (wv == 1) eliminate_blocked<n, 1>(dw, io);
    } else
#endif
    {
#if !defined(__HIP_DEVICE_COMPILE__)
      if (ctx.tid == 0) { const HostWave hw; eliminate_blocked<n, 0>(hw, io); eliminate_blocked<n, 1>(hw, io); }
#endif
    }
  }
  WG_SYNC(ctx);
  {  // W' = L^-1 B~'  (X^T Y with X = (L^-1)^T)
    const XtyJob job = xty_job(NUT, n, NUT, &w.LinvT[0][0], LDB, &w.BT[0][0], n + 1, &w.Wt[0][0], n + 1);
    wg_xty_jobs<true, 0, SCAN_PF>(ctx, &job, 1);
  }
  WG_SYNC(ctx);
  {  // A~ - W Z, C = W W', J = Q~ - Z' Z  (X^T Y with X = W' resp. Z; the additive terms come from the QP record)
    const XtyJob jobs[3] = {xty_job(n, n, NUT, &w.Wt[0][0], n + 1, &w.Zs[0][0], LP, el + E::A, n, q + QP_A, NX, -1.0),
                            xty_job(n, n, NUT, &w.Wt[0][0], n + 1, &w.Wt[0][0], n + 1, el + E::C, n),
                            xty_job(n, n, NUT, &w.Zs[0][0], LP, &w.Zs[0][0], LP, el + E::J, n, q + QP_Q, NX, -1.0)};
    wg_xty_jobs<true, 0, SCAN_PF>(ctx, jobs, 3);
    WG_FOR(ctx, i, 2 * n + (E::SIZE - E::ETA - n) + 1) {
      if (i < n) { double s = q[QP_BV + i]; for (int l = 0; l < NUT; ++l) s -= w.Wt[l][i] * w.zv[l]; el[E::B + i] = s; }
      else if (i < 2 * n) { const int r = i - n; double s = q[QP_QV + r]; for (int l = 0; l < NUT; ++l) s -= w.Zs[l][r] * w.zv[l]; el[E::ETA + r] = -s; }
      else if (i < 2 * n + (E::SIZE - E::ETA - n)) el[E::ETA + n + (i - 2 * n)] = 0.0;
      else if (ok && !w.ok) *ok = 0;
    }
  }
  WG_SYNC(ctx);
}

// ---- combination of two elements
template <int n>
struct ScanCombWS {
  static constexpr int LD = n + 1, LX = 2 * n + 2;
  double Mb[n][LD];                   // M = I + C1 J2; once the elimination has read it: V = J2 XA
  double J2[n][LD];                   // J2; once T and t are formed: A1
  double A2T[n][LD];
  union {
    double C1[n][LD];                 // staging of C1 for the product that forms M (the elimination reads C1 from global memory)
    double X[n][LX];                  // [XA | XC | xb . ] in natural row order; XA is overwritten by T = A2 XC
  };
  double b1[n], eta1[n], b2[n], eta2[n], y[n], z[n], t[n], rh[n];
  int ok;
  GjPipe gp;
};

// e1 (i -> j), e2 (j -> k) -> out (i -> k); returns through *ok whether every pivot was usable.
// 137 KB of LDS for n = 58: the augmented matrix [M | A1 | C1 | rhs] of the elimination lives in registers (gauss_jordan_rows), the
// two symmetric results are staged for their symmetrisation in buffers that are dead by then.
template <int n>
HSQP_HD void scan_combine(const Ctx& ctx, ScanCombWS<n>& w, const double* e1, const double* e2, double* out, int* ok) {
  using E = ScanEl<n>;
  constexpr int LD = ScanCombWS<n>::LD, LX = ScanCombWS<n>::LX;
  PH_TICK(ctx, 126);
  // item = (column c, row group g of NG): walks down its column of the three matrices with constant strides (no division per element;
  // a wave reads 64 consecutive words of a row); eight loads in flight before the first store
  {
    constexpr int NG = 8, RPG = (n + NG - 1) / NG;
    WG_FOR(ctx, it, 64 * NG) {
      const int c = it & 63, g = it >> 6;
      if (c < n) {
        double t[3][RPG];
#pragma unroll
        for (int k = 0; k < RPG; ++k) {
          const int r = g + NG * k, rr = r < n ? r : n - 1;
          t[0][k] = e1[E::C + rr * n + c]; t[1][k] = e2[E::J + rr * n + c]; t[2][k] = e2[E::A + rr * n + c];
        }
#pragma unroll
        for (int k = 0; k < RPG; ++k) {
          const int r = g + NG * k;
          if (r < n) { w.C1[r][c] = t[0][k]; w.J2[r][c] = t[1][k]; w.A2T[c][r] = t[2][k]; }
        }
      }
    }
    WG_FOR(ctx, j, 4 * n + 1) {
      const int r = j % n;
      if (j < n) w.b1[r] = e1[E::B + r];
      else if (j < 2 * n) w.eta1[r] = e1[E::ETA + r];
      else if (j < 3 * n) w.b2[r] = e2[E::B + r];
      else if (j < 4 * n) w.eta2[r] = e2[E::ETA + r];
      else w.ok = 1;
    }
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 20);
  {  // M - I = C1 J2 (C1 symmetric: X = C1), right-hand side b1 + C1 eta2, y = eta2 - J2 b1
    const XtyJob job = xty_job(n, n, n, &w.C1[0][0], LD, &w.J2[0][0], LD, &w.Mb[0][0], LD);
    wg_xty_jobs<true, 0, SCAN_PF>(ctx, &job, 1);
    WG_FOR(ctx, i, 2 * n) {
      if (i < n) { double s = w.b1[i]; for (int l = 0; l < n; ++l) s += w.C1[i][l] * w.eta2[l]; w.rh[i] = s; }
      else { const int r = i - n; double s = w.eta2[r]; for (int l = 0; l < n; ++l) s -= w.J2[r][l] * w.b1[l]; w.y[r] = s; }
    }
  }
  WG_SYNC(ctx);   // the staging of C1 is dead from here on (X aliases it)
  PH_TICK(ctx, 21);
  // [XA | XC | xb] = M^-1 [A1 | C1 | rhs]
#if defined(__HIP_DEVICE_COMPILE__)
  {   // k_scan_combine runs with 512 threads (8 waves)
    gauss_jordan_pipeline<n, 2 * n + 1>(
        ctx, w.gp, [&](int r, int c) { return w.Mb[r][c] + (r == c ? 1.0 : 0.0); },
        [&](int r, int c) { return c < n ? e1[E::A + r * n + c] : (c < 2 * n ? e1[E::C + r * n + (c - n)] : w.rh[r]); }, &w.X[0][0], LX, &w.ok);
    WG_FOR(ctx, r, n) w.X[r][2 * n + 1] = 0.0;
  }
#else
  {
    constexpr int LG = 3 * n + 2;
    std::vector<double> Gv((size_t)n * LG);
    GjWS gj;
    WG_FOR(ctx, i, n * LG) {
      const int r = i / LG, c = i % LG;
      Gv[i] = c < n ? w.Mb[r][c] + (r == c ? 1.0 : 0.0) : (c < 2 * n ? e1[E::A + r * n + (c - n)] : (c < 3 * n ? e1[E::C + r * n + (c - 2 * n)] : (c == 3 * n ? w.rh[r] : 0.0)));
    }
    gauss_jordan<n, LG, 3 * n + 1, true>(ctx, Gv.data(), gj);
    WG_FOR(ctx, i, n * LX) {
      const int r = i / LX, c = i % LX;
      const int p = gj.piv[r];
      w.X[r][c] = c <= 2 * n ? Gv[(size_t)p * LG + n + c] / Gv[(size_t)p * LG + r] : 0.0;
    }
    if (!gj.ok) w.ok = 0;
  }
#endif
  WG_SYNC(ctx);
  PH_TICK(ctx, 22);
  {  // A = A2 XA (to the output), V = J2 XA (over M);  z = XC y, b = A2 xb + b2
    const XtyJob jobs[2] = {xty_job(n, n, n, &w.A2T[0][0], LD, &w.X[0][0], LX, out + E::A, n),
                            xty_job(n, n, n, &w.J2[0][0], LD, &w.X[0][0], LX, &w.Mb[0][0], LD)};
    wg_xty_jobs<true, 0, SCAN_PF>(ctx, jobs, 2);
    WG_FOR(ctx, i, 2 * n) {
      if (i < n) { double s = 0.0; for (int l = 0; l < n; ++l) s += w.X[i][n + l] * w.y[l]; w.z[i] = s; }
      else { const int r = i - n; double s = w.b2[r]; for (int l = 0; l < n; ++l) s += w.A2T[l][r] * w.X[l][2 * n]; out[E::B + r] = s; }
    }
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 23);
  // T = A2 XC over XA (dead);  t = y - J2 z;  A1 is fetched meanwhile and lands in the J2 buffer after the barrier
  constexpr int NPA1 = (n * n + 511) / 512;
#if defined(__HIP_DEVICE_COMPILE__)
  double ta1[NPA1];
  const bool hoist = ctx.nthreads >= 512;
  if (hoist) {
#pragma unroll
    for (int j = 0; j < NPA1; ++j) { const int e = ctx.tid + j * ctx.nthreads; ta1[j] = e1[E::A + (e < n * n ? e : 0)]; }
  }
#else
  const bool hoist = false;
#endif
  {
    const XtyJob job = xty_job(n, n, n, &w.A2T[0][0], LD, &w.X[0][n], LX, &w.X[0][0], LX);
    wg_xty_jobs<true, 0, SCAN_PF>(ctx, &job, 1);
    WG_FOR(ctx, r, n) { double s = w.y[r]; for (int l = 0; l < n; ++l) s -= w.J2[r][l] * w.z[l]; w.t[r] = s; }
  }
  WG_SYNC(ctx);   // J2 is dead
#if defined(__HIP_DEVICE_COMPILE__)
  if (hoist) {
#pragma unroll
    for (int j = 0; j < NPA1; ++j) { const int e = ctx.tid + j * ctx.nthreads; if (e < n * n) w.J2[e / n][e % n] = ta1[j]; }
  }
#endif
  if (!hoist) { WG_FOR(ctx, e, n * n) w.J2[e / n][e % n] = e1[E::A + e]; }
  WG_SYNC(ctx);
  PH_TICK(ctx, 24);
  // C = T A2' + C2 (X^T Y with X[l][i] = T[i][l]) and J = A1' V + J1 are symmetric in exact arithmetic, but the computed X carries the
  // rounding of an elimination whose condition number reaches 1e5 (centroidal) .. 1e9 (whole-body): the ANTISYMMETRIC part of that error
  // is removed by averaging the two triangles, which is worth three digits on ill-conditioned iterates (measured: input error 1.3e-4
  // vs 2e-7 on the line-search iterations of config 2) — so both products are formed in full and symmetrised, one after the other
  // through the two buffers that are free by then: the XC columns of X (T is formed, z and b are out), then A2'.
  {
    XtyJob jc = xty_job(n, n, n, &w.X[0][0], 1, &w.A2T[0][0], LD, &w.X[0][n], LX, e2 + E::C, n);
    jc.sx1 = LX;
    wg_xty_jobs<true, XTY_ADD_GLOBAL, SCAN_PF>(ctx, &jc, 1);
    WG_FOR(ctx, i, n) { double s = w.eta1[i]; for (int l = 0; l < n; ++l) s += w.J2[l][i] * w.t[l]; out[E::ETA + i] = s; }
  }
  WG_SYNC(ctx);   // A2' is dead
  {
    const XtyJob jj = xty_job(n, n, n, &w.J2[0][0], LD, &w.Mb[0][0], LD, &w.A2T[0][0], LD, e1 + E::J, n);
    wg_xty_jobs<true, XTY_ADD_GLOBAL, SCAN_PF>(ctx, &jj, 1);
    WG_FOR(ctx, i, n * n) { const int r = i / n, c = i % n; out[E::C + i] = 0.5 * (w.X[r][n + c] + w.X[c][n + r]); }
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 25);
  WG_FOR(ctx, i, n * n + (E::SIZE - E::ETA - n) + 1) {
    if (i < n * n) { const int r = i / n, c = i % n; out[E::J + i] = 0.5 * (w.A2T[r][c] + w.A2T[c][r]); }
    else if (i < n * n + (E::SIZE - E::ETA - n)) out[E::ETA + n + (i - n * n)] = 0.0;
    else if (!w.ok) *ok = 0;
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 26);
}

}  // namespace hsqp
