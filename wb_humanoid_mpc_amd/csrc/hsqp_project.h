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
#include <cstddef>
#include <vector>
#include "hsqp_linalg.h"
#ifndef HSQP_PEXP
#define HSQP_PEXP 0   /* timing experiments of tuning builds (WRONG results): bit 0 no Px / Pu / Pe and b~ stores, 1 no Gram store, 2 no second pass, 3 no QR, 4 no dense-row jobs, 5 no weight rows, 6 no first pass */
#endif
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
#ifndef HSQP_LDTM
#define HSQP_LDTM 82    /* (88 until the end of round 6: row groups of a tile read 48 banks apart overlap in 16 banks; 82: in 4.  k_project 1.410 -> 1.397 ms) */
#endif
#ifndef HSQP_LDP
#define HSQP_LDP 98     /* leading dimension of PV in LDS (the record's is LDJ = 96: sixteen lanes LDJ apart are ONE bank) */
#endif
constexpr int LDP = HSQP_LDP;
constexpr int LDTM = HSQP_LDTM;                // leading dimension of Tm = [Px | Pu | Pe | pad] and of the residual rows (>= NTW + 1; tuning builds: -DHSQP_LDTM)
// The 64 residual row slots are projected in passes of NRP rows (24 + 24 + 16) through one small LDS block; the projected
// Gauss-Newton Hessian J~^T J~ is accumulated ACROSS the passes in registers (the matrix-core accumulators of the 15 tiles on /
// above the diagonal of its leading 80 x 80 block, dealt to the four waves; the last projected input and the gradient column
// as per-thread sums), and the 35 input-weight rows sqrt(d_u) [Px | Pu | Pe] are accumulated straight from Tm.  Nothing of the
// Hessian makes a round trip through memory between the passes, and the workspace is 49.5 KB: three workgroups per CU.
constexpr int NRP = 24;
static_assert(NRP % 4 == 0 && NRS == 2 * NRP + 16, "pass sizes");
constexpr int NGT = 5;                         // 16-column tiles of the Gram matrix on the matrix cores: columns 0..79
static_assert(16 * NGT == NTW - 1, "the matrix cores take the leading 80 columns");

constexpr int LDR = 16;
struct ProjWS {
  union {
    struct {
      double Rm[NU + 1][LDR];    // D^T (zero padded to 36 x 16), overwritten by R1
      double V[NE_MAX][NU + 1];  // Householder vectors
      double beta[NE_MAX], Rdiag[NE_MAX], rinv[LDR], ep[LDR];   // ep: e of the dense rows after the eliminated inputs are substituted
      double part[LDR][4], yk[LDR];   // per-step scratch: partial dots, row k of R
      double Q1T[NE_MAX][NU];    // rows 0..ned-1 of Q^T (Q1^T); the rows of Q2^T go straight into Tm
      double Wm[NE_MAX][NX + 2]; // R1^-T [C|e]
    } qr;                        // live until Tm is formed
    double PV[2][6][LDP];        // the non-trivial rows of [A|B]: stored while Tm is being formed, over Rm / V / the scalars (dead by then)
    struct {
      double Jt[NRP][LDTM];      // the projected residual rows of the current pass (column 81 = rho')
      double JuT[NU + 1][NRP];   // transposed input block of the residual rows of the current (next) pass (row NU: their rho); overlaps Wm
    } ps;
#if defined(HSQP_PEXP_ALIAS)   /* timing experiment only (WRONG results): T over the first union — 28.5 KB, four to five workgroups per CU (profiles/r06_project_experiments.txt (3)) */
    double Tm[NU + 1][LDTM];
    double CDe[NE_MAX][LDJ];
  };
#else
  };
  union {
    double Tm[NU + 1][LDTM];     // [Px (58) | Pu (23) | Pe | 0 ...]; row NU = e_NTW: with rho as row NU of JuT the product J_u [Pu | Pe] also adds rho to its last column
    double CDe[NE_MAX][LDJ];     // the equality rows, until W = R1^-T [C|e] is formed (Tm is written after that)
  };
#endif
  int ne, nut, ok;
  int jt;                        // the record's residual / equality rows are stored transposed (REC_LAYOUT)
  int nrows;                     // residual rows of the record in use (REC_NROWS); the third pass is skipped when they fit two
  // deflation of the unit rows of D (a swing foot's zero-wrench constraints W_f = 0 fix six inputs outright): the Householder
  // QR only sees the dense rows (ned of them) restricted to the free inputs (nub of them)
  int ned, nub;
  int ub[NU];                    // free input indices, then the eliminated ones (positions nub..NU-1)
  int urow[NU];                  // for an eliminated input: its unit constraint row; -1 for a free input
  int rd[NE_MAX];                // dense constraint rows
  double eu[NU];                 // -e of the unit row of an eliminated input (0 for a free input)
  double bvec[64];
  double rho[NRS], d[LDJ], gd[LDJ];   // mirror one contiguous piece of the LQ record
};
static_assert(LDP >= LDJ && sizeof(double) * 2 * 6 * LDP <= offsetof(ProjWS, qr.Q1T), "PV may only overlap what is dead while Tm is formed");
static_assert(sizeof(ProjWS) <= 163840 / 3 - 256, "three workgroups per CU");
// fused RK4 chain (project_node, chain = true): the chain's threads — waves 2, 3 minus their last 32 lanes; wave 0 runs the factorisation meanwhile.  The 216 entries of the
// 6 x 6 blocks have wave-uniform addresses and come through the scalar cache; the A/B form (HSQP_PROJ_CHAIN_BLK_LDS) stages them in rows of Tm that lie behind the equality
// rows (CDe) and are first written when Q2^T goes into Tm, two phases after the chain
#ifndef HSQP_PROJ_CHAIN_BLK_LDS
#define HSQP_PROJ_CHAIN_BLK_LDS 0   /* 1: the chain's 6 x 6 blocks staged in LDS (rows of Tm) instead of read through the scalar cache (A/B builds) */
#endif
#ifndef HSQP_PROJ_CHAIN_T0
#define HSQP_PROJ_CHAIN_T0 128   /* first thread of the fused chain (tuning builds: 64 = waves 1, 2) */
#endif
constexpr int PROJ_CHAIN_T0 = HSQP_PROJ_CHAIN_T0, PROJ_CHAIN_ROW = 24;
static_assert(PROJ_CHAIN_T0 >= 64 && PROJ_CHAIN_T0 + LDJ <= 256, "the chain's threads: not wave 0 (the factorisation), inside the workgroup");
static_assert(PROJ_CHAIN_ROW * LDTM >= NE_MAX * LDJ && PROJ_CHAIN_ROW * LDTM + 3 * 72 <= NU * LDTM, "the chain's blocks must not touch the equality rows or row NU of Tm");
HSQP_HD double* proj_chain_blk(ProjWS& w) { return &w.Tm[PROJ_CHAIN_ROW][0]; }

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

// ---- Gram accumulation across the passes: H (+)= X^T diag(s) X over the rows of X (columns 0..81 of Jt or Tm; column 81 carries
// rho', so the same sums also give the gradient J~^T rho').  Device: each wave owns the accumulators of up to four 16 x 16 tiles on /
// above the diagonal of the leading 80 x 80 block; threads 0..161 own one element each of column 80 (the last projected input) and
// of the gradient column.  Host build: a plain upper-triangular array.
struct GramAcc {
#if defined(__HIP_DEVICE_COMPILE__)
  hsqp_d4 acc[4];
  int xr[4], yc[4], nt;
  double vs;
#else
  std::vector<double> h;         // [NTW][NTW + 1]: h[a][c] for a <= c <= 80, h[a][81] = gradient
#endif
};

HSQP_HD void gram_init(const Ctx& ctx, GramAcc& g) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int wave = ctx.tid >> 6, i = ctx.tid & 15;
  g.nt = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int id = wave + 4 * t;                       // 15 tiles dealt round-robin to four waves
    int tr = 0, tc = id < NGT * (NGT + 1) / 2 ? id : 0;
    while (tc >= NGT - tr) { tc -= NGT - tr; ++tr; }
    tc += tr;
    g.xr[t] = 16 * tr + i;
    g.yc[t] = 16 * tc + i;
    g.acc[t] = hsqp_d4{0.0, 0.0, 0.0, 0.0};
    if (id < NGT * (NGT + 1) / 2) g.nt = t + 1;
  }
  g.vs = 0.0;
#else
  (void)ctx;
  g.h.assign((size_t)NTW * (NTW + 1), 0.0);
#endif
}

// WEIGHT = false: the rows are taken as they are.  WEIGHT = true (the input-weight rows, X = Tm): row k counts with weight du[k], and
// the gradient column gets gdu[k] on top: sum_k X[k][a] (du[k] X[k][81] + gdu[k]) = (T_u^T (D_u Pe + g_u))[a].
template <bool WEIGHT, int NR>
HSQP_HD void gram_rows(const Ctx& ctx, GramAcc& g, const double* X, int ldx, const double* du, const double* gdu) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int kk = (ctx.tid & 63) >> 4;
#pragma unroll
  for (int k0 = 0; k0 < NR; k0 += 4) {
    const int k = k0 + kk;
    const bool ok = (NR % 4 == 0) || k < NR;
    const double* row = X + (ok ? k : 0) * ldx;
    const double sc = WEIGHT ? (ok ? du[ok ? k : 0] : 0.0) : 1.0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < 3 || g.nt == 4) {
        const double a = row[g.xr[t]];
        double b = row[g.yc[t]];
        if (WEIGHT || NR % 4 != 0) b *= (WEIGHT ? sc : (ok ? 1.0 : 0.0));
        g.acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, g.acc[t], 0, 0, 0);
      }
    }
  }
  // (the 162 per-thread sums sit on the LAST threads of the workgroup: wave 3 carries three of the fifteen tiles, the others four)
  const int vt = ctx.nthreads - 1 - ctx.tid;
  if (vt < 2 * NTW) {
    const int a = vt < NTW ? vt : vt - NTW, cb = vt < NTW ? NTW - 1 : NTW;
    double s = g.vs;
#pragma unroll 4
    for (int r = 0; r < NR; ++r) {
      double xb = X[r * ldx + cb];
      if (WEIGHT) xb = du[r] * xb + (cb == NTW ? gdu[r] : 0.0);
      s += X[r * ldx + a] * xb;
    }
    g.vs = s;
  }
#else
  (void)ctx;
  for (int r = 0; r < NR; ++r)
    for (int a = 0; a < NTW; ++a) {
      const double xa = X[r * ldx + a];
      double* ha = &g.h[(size_t)a * (NTW + 1)];
      for (int c = a; c <= NTW; ++c) {
        double xb = X[r * ldx + c];
        if (WEIGHT) xb = du[r] * xb + (c == NTW ? gdu[r] : 0.0);
        ha[c] += xa * xb;
      }
    }
#endif
}

// The accumulated blocks to the QP record: Q~ (+ diag d_x), P~, R~ (identity on the unused projected inputs), q~ (+ g_x), r~.
// lower_q = false: the strictly lower triangle of Q~ is not written (the factored Riccati sweep reads the tiles on / above the diagonal only; the mirrored
// stores are the scattered ones of this epilogue: 13 KB per node)
HSQP_HD void gram_store(const Ctx& ctx, const GramAcc& g, const ProjWS& w, int nut, double* qp, bool lower_q = true) {
  auto put = [&](int r, int c, double v) {             // r <= c <= 80
    if (c < NX) {
      if (r == c) v += w.d[r];
      qp[QP_Q + r * NX + c] = v;
      if (r != c && lower_q) qp[QP_Q + c * NX + r] = v;
    } else if (r < NX) {
      qp[QP_P + (c - NX) * NX + r] = v;
    } else {
      const int a = r - NX, b = c - NX;
      if (a >= nut || b >= nut) v = (a == b) ? 1.0 : 0.0;
      qp[QP_R + a * NUT + b] = v;
      if (a != b) qp[QP_R + b * NUT + a] = v;
    }
  };
  auto put_grad = [&](int a, double v) {
    if (a < NX) qp[QP_QV + a] = v + w.gd[a]; else qp[QP_RV + a - NX] = v;
  };
#if defined(__HIP_DEVICE_COMPILE__)
  const int i = ctx.tid & 15, kk = (ctx.tid & 63) >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t < 3 || g.nt == 4) {
      const int r0 = g.xr[t] - i, c = g.yc[t];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = r0 + kk + 4 * reg;
        if (r <= c) put(r, c, g.acc[t][reg]);
      }
    }
  }
  const int vt = ctx.nthreads - 1 - ctx.tid;
  if (vt < NTW) put(vt, NTW - 1, g.vs);
  else if (vt < 2 * NTW) put_grad(vt - NTW, g.vs);
#else
  (void)ctx;
  for (int a = 0; a < NTW; ++a) {
    for (int c = a; c < NTW; ++c) put(a, c, g.h[(size_t)a * (NTW + 1) + c]);
    put_grad(a, g.h[(size_t)a * (NTW + 1) + NTW]);
  }
#endif
}

// cent = true: the record comes from the centroidal LQ kernel (hsqp_cent.h): the dense rows of [A|B] - [I|0] are rows 0..11
// (PV[0] = momentum rows, PV[1] = base pose rows), rows 12..34 are q_j+ = q_j + dt qd_j, rows 35..57 padding states (A = I).
// joint_rows = false: the 46 joint rows of A~ / B~ (scaled copies of rows 12 .. of [Px | Pu]: q_j+ = q_j + dt v_j + dt^2/2 qdd_j, v_j+ = v_j + dt qdd_j) and the
// strictly lower triangle of Q~ are NOT written — the factored Riccati sweep (hsqp_riccati_fact.h) forms them from Px, Pu on the fly; b~ is always complete.  30 of the record's 101 KB.
// chain = true (device; limb-lane LQ form): the RK4 chain of the columns of [A|B] (lq_chain_column_pv, hsqp_lq.h) runs HERE, from the stage Jacobians REC_GS, on two waves
// that would otherwise wait for the factorisation's one wave — REC_PV is neither written by the LQ kernels nor read (P6, V6 made a round trip of 18 KB per node through
// memory and k_lq_chain read 18 KB of stage Jacobians for them; the defect moves to the lanes of k_lq_rows and k_lq_chain is not launched).
HSQP_HD void project_node(const Ctx& ctx, ProjWS& w, const double* rec, double dt, double* qp, bool cent = false, bool joint_rows = true, bool chain = false) {
  (void)chain;
  // ---- load: record pieces [REC_B, REC_J) -> bvec and [REC_RHO, REC_MISC) -> rho, d, gd, CDe; 8 loads in flight per item
  {
    static_assert(REC_B == REC_PV + 2 * 6 * LDJ && REC_J == REC_B + 64, "record layout");
    static_assert(REC_D == REC_RHO + NRS && REC_GD == REC_D + LDJ && REC_CDE == REC_GD + LDJ && REC_MISC == REC_CDE + CDE_ROWS * LDJ, "record layout");
    constexpr int n1 = 64, n2 = NRS + 2 * LDJ, n3 = CDE_ROWS * LDJ, nld = n1 + n2 + n3, nb = nbatches(nld, 8);
    double* dst1 = &w.bvec[0];
    double* dst2 = &w.rho[0];
    double* dst3 = &w.CDe[0][0];
    // the scalars of the record that steer the structure items below: fetched WITH the block loads (behind the stores of the load loop they were a
    // second, dependent HBM round trip in front of the factorisation)
    const double m_ne = rec[REC_MISC], m_c0 = rec[REC_MISC + 4], m_c1 = rec[REC_MISC + 5], m_o0 = rec[REC_MISC + 6], m_o1 = rec[REC_MISC + 7], m_nr = rec[REC_NROWS];
    const bool jt0 = rec[REC_LAYOUT] != 0.0;   // transposed residual / equality rows (limb-lane LQ kernel, hsqp_lql.h)
    WG_FOR(ctx, b, nb) {
      double t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int idx = b + j * nb; t[j] = idx < nld ? rec[idx < n1 ? REC_B + idx : REC_RHO + (idx - n1)] : 0.0; }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = b + j * nb;
        if (idx < n1) dst1[idx] = t[j];
        else if (idx < n1 + n2) dst2[idx - n1] = t[j];
        else if (idx < nld) {   // the equality rows; transposed record: element (row e % 16, column e / 16)
          const int e = idx - n1 - n2;
          if (!jt0) { if (e < NE_MAX * LDJ) dst3[e] = t[j]; }
          else if (e % CDE_ROWS < NE_MAX) dst3[(e % CDE_ROWS) * LDJ + e / CDE_ROWS] = t[j];
        }
      }
    }
    // structure of the equality rows (closed forms, one item per input / row): a swing foot f contributes the unit rows
    // off[f] .. off[f]+5 on the inputs 6f .. 6f+5
    WG_FOR(ctx, i, LDTM) w.Tm[NU][i] = i == NTW ? 1.0 : 0.0;   // (beyond the equality rows that share the block until Tm is formed)
#if defined(__HIP_DEVICE_COMPILE__)
    // (A/B form only) the 6 x 6 blocks G_s[:, v_b], G_s[:, q_b] of stages 2 .. 4 that every column's chain multiplies with: in rows of Tm nobody touches before the W phase is over
    if (chain && HSQP_PROJ_CHAIN_BLK_LDS) WG_FOR(ctx, i, 3 * 72) {
      const int sg = i / 72, which = (i / 36) % 2, r = (i / 6) % 6, k = i % 6;
      proj_chain_blk(w)[i] = rec[REC_GS + lq_chain_blk_offset(sg, which, r, k)];
    }
#endif
    WG_FOR(ctx, i, NU + NE_MAX + 1) {
      const int ne_ = (int)m_ne;
      const int sw0 = m_c0 == 0.0 ? 1 : 0, sw1 = m_c1 == 0.0 ? 1 : 0;
      const int off0 = (int)m_o0, off1 = (int)m_o1;
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
        w.nrows = (int)m_nr;
        w.jt = jt0 ? 1 : 0;
      }
    }
  }
  WG_SYNC(ctx);
  const int ne = w.ne, nut = w.nut, ned = w.ned, nub = w.nub;
  const bool jt = w.jt != 0;
  (void)ne;
  PH_TICK(ctx, 1);
  // ---- Householder QR of D^T.  Step k: every remaining column recomputes the reflector from column k (which is
  // left untouched: its final diagonal goes to Rdiag, the vector to V).
  WG_FOR(ctx, i, (NU + 1) * LDR) { const int r = i / LDR, c = i % LDR; w.qr.Rm[r][c] = (c < ned && r < nub) ? w.CDe[w.rd[c]][NX + w.ub[r]] : 0.0; if (r == 0) {
      w.qr.rinv[c] = 0.0;
      // e' = e - D_a e_U: the eliminated inputs are fixed at -e_U (the unit rows have C = 0)
      double ep = 0.0;
      if (c < ned) {
        // (only the twelve wrench inputs can be eliminated; a fixed loop over them in ascending order — the order of ub's tail — keeps its
        //  loads independent: the walk over ub[nub ..] was twelve dependent LDS round trips on the phase's critical path)
        const int ri = w.rd[c];
        ep = w.CDe[ri][NZ];
#pragma unroll
        for (int ue = 0; ue < 12; ++ue) { const int ur = w.urow[ue]; const double eu_ = w.CDe[ur >= 0 ? ur : 0][NZ]; if (ur >= 0) ep -= w.CDe[ri][NX + ue] * eu_; }
      }
      w.qr.ep[c] = ep;
    }
    if (i < NU) w.eu[i] = w.urow[i] >= 0 ? -w.CDe[w.urow[i]][NZ] : 0.0; }   // an eliminated input is fixed at -e of its unit row
  WG_SYNC(ctx);
  PH_TICK(ctx, 8);
#if defined(__HIP_DEVICE_COMPILE__)
  // device: the whole factorisation in the registers of ONE wave, no barrier and no LDS traffic per step.  Lane (g << 4) | c holds the rows
  // 4 t + g (t = 0 .. 8) of column c of the 36 x 16 matrix — four row groups, one per DPP row of 16 lanes (round 3 kept a whole column in a
  // lane: 16 of 64 lanes busy, 36 registers walked per step, 3.5 k vector instructions = 28 % of the workgroup's).  The pivot column comes to
  // every lane of a row through one DPP row broadcast per register (row_newbcast:k — lane k of the own row holds the same rows of column k), the
  // norm, the dot product and the own column's row-k entry through one butterfly over the four row groups.  The partial sums are those of the
  // round-3 form (rows i mod 4, ascending; ((0 + 1) + (2 + 3))).  Same arithmetic per element as the two-phase form below
  // (which the host build runs), summation order aside.
  double chP[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, chV[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // chain: column tid - PROJ_CHAIN_T0 of P6, V6 (threads PROJ_CHAIN_T0 .. + LDJ - 1), held until store_pv
  if (chain && ctx.tid >= PROJ_CHAIN_T0 && ctx.tid < PROJ_CHAIN_T0 + LDJ)
    lq_chain_column_pv<true, !HSQP_PROJ_CHAIN_BLK_LDS>(reinterpret_cast<const double (*)[2][6][6]>(proj_chain_blk(w)), rec + REC_GS, ctx.tid - PROJ_CHAIN_T0, dt, chP, chV);
  if (ctx.tid < 64 && !(HSQP_PEXP & 8)) {
    const int lane = ctx.tid, g = lane >> 4, c = lane & 15;
    constexpr int NT9 = (NU + 1) / 4;
    static_assert(NT9 * 4 == NU + 1 && LDR == 16, "four row groups of nine rows, one column per lane of a DPP row");
    double e[NT9];
#pragma unroll
    for (int t = 0; t < NT9; ++t) e[t] = w.qr.Rm[4 * t + g][c];
    auto xsum = [](double v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; };   // over the four row groups, in every lane
    static_for<NE_MAX>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if (k < ned) {
        double x[NT9];
        double pn = 0.0, pd = 0.0;
#pragma unroll
        for (int t = k >> 2; t < NT9; ++t) {
          x[t] = row_bcast_f64<k>(e[t]);
          // (only the register that holds row k mixes finished rows — i < k — with live ones; g < 4 is not known to the compiler)
          const double xm = (t > (k >> 2) || g >= (k & 3)) ? x[t] : 0.0;
          pn += xm * xm;
          pd += xm * e[t];
        }
        const double ekp = g == (k & 3) ? e[k >> 2] : 0.0;        // row k of the own column (one row group holds it)
        const double nrm2 = xsum(pn), dot = xsum(pd), ek = xsum(ekp);
        const double rkk = readlane_f64(e[k >> 2], ((k & 3) << 4) | k);
        const double rs = inv_sqrt(nrm2 > 1e-300 ? nrm2 : 1e-300);
        const double nrm = nrm2 * rs;
        const double alpha = rkk >= 0.0 ? -nrm : nrm;
        const double hv = nrm2 - alpha * rkk;      // = |v|^2 / 2
        const double beta = hv > 1e-300 ? fast_rcp(hv) : 0.0;
        const double sdot = beta * (dot - alpha * ek);
        if (lane == 0) {
          w.qr.beta[k] = beta;
          w.qr.Rdiag[k] = alpha;
          w.qr.rinv[k] = rkk >= 0.0 ? -rs : rs;
          if (!(nrm >= 1e-12)) w.ok = 0;
        }
        if (c == k) {
#pragma unroll
          for (int t = 0; t < NT9; ++t) {
            double v = 0.0;
            if (t > (k >> 2)) v = x[t];
            else if (t == (k >> 2)) v = g == (k & 3) ? rkk - alpha : (g > (k & 3) ? x[t] : 0.0);
            w.qr.V[k][4 * t + g] = v;
          }
        } else if (c > k) {
#pragma unroll
          for (int t = k >> 2; t < NT9; ++t) {
            if (t > (k >> 2)) e[t] -= sdot * x[t];
            else if (g == (k & 3)) e[t] -= sdot * (rkk - alpha);
            else if (g > (k & 3)) e[t] -= sdot * x[t];
          }
        }
      }
    });
    if (c < LDR) {   // R1 above the diagonal (the W phase reads Rm[j][i], j < i)
#pragma unroll
      for (int t = 0; t < NT9; ++t) { const int j = 4 * t + g; if (j < NE_MAX && j < c) w.qr.Rm[j][c] = e[t]; }
    }
  }
  WG_SYNC(ctx);
#else
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
      if (c < k && it != 0) continue;            // finished columns: whole waves drop out as k grows (item 0 keeps the step's bookkeeping)
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
#endif
  PH_TICK(ctx, 9);
  // ---- W = R1^-T [C | e'] over the dense rows: forward substitution per column, the column kept in registers (rows >= ned:
  //      rinv = 0 -> 0).  Last use of the equality rows: Tm may be written from the next phase on.
  WG_FOR(ctx, c, NX + 1) {
    double wc[NE_MAX];
#pragma unroll
    for (int i = 0; i < NE_MAX; ++i) {
      const int ri = w.rd[i];
      double s = c < NX ? w.CDe[ri][c] : w.qr.ep[i];
#pragma unroll
      for (int j = 0; j < i; ++j) s -= w.qr.Rm[j][i] * wc[j];
      wc[i] = s * w.qr.rinv[i];
      w.qr.Wm[i][c] = wc[i];
    }
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 3);
  // Q^T = H_{ned-1} ... H_0 of the free inputs: one column per item, the column lives in registers while the reflectors are
  // applied; it belongs to the ORIGINAL input index ub[c].  Rows 0..ned-1 are Q1^T (-> Q1T), rows ned.. are Q2^T = Pu^T, written
  // straight into Tm; the columns of the eliminated inputs are zero
#if defined(__HIP_DEVICE_COMPILE__)
  // device: FOUR adjacent lanes per column, lane p holds the entries i = p + 4 t; the dot product with a reflector is closed by two DPP
  // quad permutes (the one-lane-per-column form below keeps 35 lanes of one wave busy for 12 x 105 dependent operations)
  if (ctx.tid < 4 * (NU + 1)) {
    constexpr int NT4 = (NU + 4) / 4;            // 9 entries per lane cover i = 0..35 (V has NU + 1 columns, the last one zero)
    const int c = ctx.tid >> 2, p = ctx.tid & 3;
    const bool livec = c < nub;                  // c == NU (the 36th quad) and the eliminated inputs: zero columns
    double col[NT4];
#pragma unroll
    for (int t = 0; t < NT4; ++t) col[t] = (p + 4 * t == c && livec) ? 1.0 : 0.0;
    // (the next reflector is fetched while this one is applied: its loads do not depend on the column)
    double vk[NT4], bk = 0.0;
    auto fetch_v = [&](int k, double* v, double& b) {
#pragma unroll
      for (int t = 0; t < NT4; ++t) { const int i = p + 4 * t; v[t] = i <= NU ? w.qr.V[k][i < NU + 1 ? i : NU] : 0.0; }
      b = w.qr.beta[k];
    };
    if (ned > 0) fetch_v(0, vk, bk);
    for (int k = 0; k < ned; ++k) {
      double vn[NT4], bn = 0.0;
      fetch_v(k + 1 < ned ? k + 1 : k, vn, bn);
      double sdot = 0.0;
#pragma unroll
      for (int t = 0; t < NT4; ++t) sdot += vk[t] * col[t];
      sdot += quad_perm_f64<0xB1>(sdot);
      sdot += quad_perm_f64<0x4E>(sdot);
      sdot *= bk;
#pragma unroll
      for (int t = 0; t < NT4; ++t) { col[t] -= sdot * vk[t]; vk[t] = vn[t]; }
      bk = bn;
    }
    if (c < NU) {
      const int uc = w.ub[c];
#pragma unroll
      for (int t = 0; t < NT4; ++t) {
        const int i = p + 4 * t;
        if (i < NU) {
          if (i < ned) w.qr.Q1T[i][uc] = col[t];
          const int cc = i - ned;
          if (cc >= 0 && cc < nut) w.Tm[uc][NX + cc] = col[t];
        }
      }
      for (int cc = nut + p; cc < NUT; cc += 4) w.Tm[uc][NX + cc] = 0.0;
    }
  }
#else
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
    for (int i = 0; i < NU; ++i) {
      if (i < NE_MAX && i < ned) w.qr.Q1T[i][uc] = col[i];
      const int cc = i - ned;
      if (cc >= 0 && cc < nut) w.Tm[uc][NX + cc] = col[i];
    }
#pragma unroll
    for (int cc = 0; cc < NUT; ++cc) if (cc >= nut) w.Tm[uc][NX + cc] = 0.0;
  }
#endif
  WG_SYNC(ctx);
  PH_TICK(ctx, 2);
  // ---- staging of the record's [A|B] rows and of the (transposed) input block of the residual rows r0 .. r0+nr.  Device: the
  //      loads are issued a phase ahead of the stores (registers), so their HBM round trip runs under the phase's other work.
  constexpr int NPV = 2 * 6 * LDJ;
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TPV = (NPV + 255) / 256, TJU = (NRP * NU + 255) / 256;   // k_project runs with >= 256 threads (hsqp_capi.hip)
  double tpv[TPV], tju[TJU];
#endif
  auto load_pv = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
    if (chain) return;
#pragma unroll
    for (int j = 0; j < TPV; ++j) { const int e = ctx.tid + j * ctx.nthreads; tpv[j] = rec[REC_PV + (e < NPV ? e : 0)]; }
#endif
  };
  // the constant part of the dense rows of A (identity; dt on the q-v coupling of the whole-body base rows) is folded in here, so
  // that A~ of those rows is PV_x + B Px exactly as the matrix cores form it
  auto pv_const = [&](int e) {
    const int p = e / (6 * LDJ), r = (e / LDJ) % 6, c = e % LDJ;
    const int row = cent ? 6 * p + r : (p == 0 ? r : NV + r);
    return (c == row ? 1.0 : 0.0) + ((!cent && p == 0 && c == NV + r) ? dt : 0.0);
  };
  auto store_pv = [&]() {
    double* pv = &w.PV[0][0][0];
#if defined(__HIP_DEVICE_COMPILE__)
    if (chain) {   // the chain's threads hold their column's twelve entries
      if (ctx.tid >= PROJ_CHAIN_T0 && ctx.tid < PROJ_CHAIN_T0 + LDJ) {
        const int c = ctx.tid - PROJ_CHAIN_T0;
#pragma unroll
        for (int r = 0; r < 6; ++r) { pv[r * LDP + c] = chP[r] + pv_const(r * LDJ + c); pv[(6 + r) * LDP + c] = chV[r] + pv_const((6 + r) * LDJ + c); }
      }
    } else
#pragma unroll
    for (int j = 0; j < TPV; ++j) { const int e = ctx.tid + j * ctx.nthreads; if (e < NPV) pv[(e / LDJ) * LDP + e % LDJ] = tpv[j] + pv_const(e); }
#else
    WG_FOR(ctx, e, NPV) pv[(e / LDJ) * LDP + e % LDJ] = rec[REC_PV + e] + pv_const(e);
#endif
  };
  auto load_ju = [&](int r0, int nr) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int j = 0; j < TJU; ++j) {
      const int e = ctx.tid + j * ctx.nthreads, ec = e < nr * NU ? e : 0;
      tju[j] = jt ? rec[REC_J + (NX + ec / nr) * NRS + r0 + ec % nr] : rec[REC_J + (r0 + ec / NU) * LDJ + NX + ec % NU];   // (transposed: item = (input, row), rows contiguous)
    }
#else
    (void)r0; (void)nr;
#endif
  };
  auto store_ju = [&](int r0, int nr) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int j = 0; j < TJU; ++j) { const int e = ctx.tid + j * ctx.nthreads; if (e < nr * NU) { if (jt) w.ps.JuT[e / nr][e % nr] = tju[j]; else w.ps.JuT[e % NU][e / NU] = tju[j]; } }
#else
    WG_FOR(ctx, e, nr * NU) w.ps.JuT[e % NU][e / NU] = jt ? rec[REC_J + (NX + e % NU) * NRS + r0 + e / NU] : rec[REC_J + (r0 + e / NU) * LDJ + NX + e % NU];
#endif
    WG_FOR(ctx, e, NRP) w.ps.JuT[NU][e] = e < nr ? w.rho[r0 + e] : 0.0;   // rho rides as one more row of the contraction (against e_NTW, row NU of Tm)
  };
  // ---- Tm = [Px | Pu | Pe | 0 ...]:  [Px | Pe] = -Q1 W  (X^T Y with X = Q1^T);  Pu is in place
  load_pv();
  load_ju(0, NRP);
  {
    const XtyJob job = xty_job(NU, NX, ned, &w.qr.Q1T[0][0], NU, &w.qr.Wm[0][0], NX + 2, &w.Tm[0][0], LDTM, nullptr, 0, -1.0);
    wg_xty_jobs<true>(ctx, &job, 1);
    WG_FOR(ctx, i, NU * (LDTM - NTW)) {
      const int r = i / (LDTM - NTW), cc = i % (LDTM - NTW);
      if (cc == 0) {
        double sdot = 0.0;
#pragma unroll
        for (int j = 0; j < NE_MAX; ++j) sdot += (j < ned ? w.qr.Q1T[j][r] : 0.0) * w.qr.Wm[j][NX];   // (fixed trip count: independent loads; rows >= ned of W are zero)
        w.Tm[r][NTW] = w.eu[r] - sdot;
      }
      else w.Tm[r][NTW + cc] = 0.0;
    }
  }
  store_pv();   // over Rm / V / the QR scalars: nothing reads them any more
  WG_SYNC(ctx);  // QR data dead from here
  PH_TICK(ctx, 4);
  store_ju(0, NRP);
  // ---- the projection and the dynamics A~ = A + B Px, B~ = B Pu, b~ = b + B Pe with the structured [A|B] (hsqp_lq.h) to the record.
  // The twelve dense rows go through the matrix cores, straight to the record (PV_x + B_row [Px | Pu]); every other row is one
  // scaled row of Tm: item = (column c of [Px | Pu | Pe], row group), which walks down its column with constant strides
  {
    const int r1 = cent ? 6 : NV;   // first row of the second dense block
    // both dense row blocks in ONE product each (twelve rows of a 16-row tile instead of twice six): the rows of the second block land r1 - 6
    // rows further down in the record (XTY_ROW_JUMP)
    XtyJob ja = xty_job(12, NX, NU, &w.PV[0][0][NX], 1, &w.Tm[0][0], LDTM, qp + QP_A, NX, &w.PV[0][0][0], LDP);
    XtyJob jb = xty_job(12, NUT, NU, &w.PV[0][0][NX], 1, &w.Tm[0][NX], LDTM, qp + QP_B, NUT);
    ja.sx1 = LDP; jb.sx1 = LDP;
    ja.rsplit = 6; ja.rjump = r1 - 6; jb.rsplit = 6; jb.rjump = r1 - 6;
    const XtyJob jobs[2] = {ja, jb};
    if (!(HSQP_PEXP & 16)) wg_xty_jobs<true, XTY_C_GLOBAL | XTY_ROW_JUMP>(ctx, jobs, 2);
  }
  constexpr int NCG = 3;   // row groups per column
  constexpr int NCI = NCG * (NTW + 1), NBI = 256 - NCI;   // column items; the items left of one 256-thread round share the twelve b~ rows
  static_assert(NBI >= 1 && NBI <= 12, "one round of the 256-thread workgroup (a second round would be wave 0's alone)");
  if (!(HSQP_PEXP & 1))
  WG_FOR(ctx, it, 256) {
    if (it >= NCI) {   // b~ of the dense rows
      for (int i = it - NCI; i < 12; i += NBI) {
        const int pw = i / 6, pr = i % 6, r = cent ? i : (pw == 0 ? pr : NV + pr);
        const double* Brow = &w.PV[pw][pr][NX];
        double s = w.bvec[r];
#pragma unroll 5
        for (int k = 0; k < NU; ++k) s += Brow[k] * w.Tm[k][NTW];
        qp[QP_BV + r] = s;
      }
      continue;
    }
    const int c = it % (NTW + 1), g = it / (NTW + 1);
    if (it == 0) qp[QP_NUT] = w.ok ? (double)nut : -1.0;
    {
      double* dst = c < NX ? qp + QP_PX + c : (c < NTW ? qp + QP_PU + (c - NX) : qp + QP_PE);
      const int st = c < NX ? NX : (c < NTW ? NUT : 1);
      for (int k = g; k < NU; k += NCG) dst[k * st] = w.Tm[k][c];
    }
    double* dst = c < NX ? qp + QP_A + c : (c < NTW ? qp + QP_B + (c - NX) : qp + QP_BV);
    const int st = c < NX ? NX : (c < NTW ? NUT : 1);
    if (!cent) {
      // joint j: q_j+ = q_j + dt v_j + dt^2/2 qdd_j (row 6 + j), v_j+ = v_j + dt qdd_j (row NV + 6 + j), qdd_j = input 12 + j: both rows
      // are scaled copies of row 12 + j of T, read once
      const double hq = 0.5 * dt * dt;
      if (joint_rows || c == NTW)
      for (int j = g; j < NJ; j += NCG) {
        const int rq = 6 + j, rv = NV + 6 + j;
        const double t = w.Tm[12 + j][c];
        double sq = hq * t, sv = dt * t;
        if (c < NX) { sq += (c == rq ? 1.0 : 0.0) + (c == rv ? dt : 0.0); sv += (c == rv ? 1.0 : 0.0); }
        else if (c == NTW) { sq += w.bvec[rq]; sv += w.bvec[rv]; }
        dst[rq * st] = sq;
        dst[rv * st] = sv;
      }
    } else {
      for (int r = 12 + g; r < NX; r += NCG) {   // rows 12..34: q_j+ = q_j + dt qd_j (input r); rows 35..57: padding states (identity)
        double s = r < HSQP_CNX ? dt * w.Tm[r][c] : 0.0;
        if (c < NX) s += (r == c) ? 1.0 : 0.0;
        else if (c == NTW) s += w.bvec[r];
        dst[r * st] = s;
      }
    }
  }
  WG_SYNC(ctx);  // PV dead: Jt aliases it
  PH_TICK(ctx, 5);
  // ---- passes over the residual row slots: J~ = J T (+ rho' in column 81) on the matrix cores, then its Gram matrix onto the
  //      accumulators while the input block of the next pass is fetched
  GramAcc g;
  gram_init(ctx, g);
  // In the limb-lane layout the row slots are fixed (hsqp_lql.h: ROWQ_FOOT + 16 f: ori, vlin, vang, alin, aang, one zero row; ROWQ_FM + 8 f: friction cone and
  // contact moment of foot f; ROWQ_COLL: collision), and most of the INPUT block of J is zero for every state: only the six acceleration rows of a foot
  // depend on all inputs, a foot's friction / moment rows on its own wrench (inputs 6 f .. 6 f + 5), nothing else on any input.  So the rows of a pass
  // beyond its first 16-row tile need no product with [Px | Pu | Pe] at all (pass 0: rows 16 .. 23 = orientation / velocities of foot 1; pass 2: the
  // collision rows) or one over six inputs (pass 1: rows 40 .. 47 = friction / moment of foot 1): J~ = J_x (+ rho) there.  96 of the 216 matrix
  // instructions of J~ = J T per node; the skipped products are sums of exact zeros.  head = rows of the pass with the full contraction; the tail
  // contracts the inputs [tk0, tk0 + tL).
  auto project_rows = [&](int r0, int nr, int head, int tk0, int tL) {
    // (the second product carries rho' = rho + J_u Pe as its 24th column: 36 = 9 x 4 contraction steps, as many as for 35 — round 3 formed it
    //  as 24 serial 35-term sums on one wave behind the tiles)
    const double* addp = jt ? rec + REC_J + r0 : rec + REC_J + r0 * LDJ;
    const int ldadd = jt ? NRS : LDJ;
    if (!jt || head >= nr) {
      XtyJob jx = xty_job(nr, NX, NU, &w.ps.JuT[0][0], NRP, &w.Tm[0][0], LDTM, &w.ps.Jt[0][0], LDTM, addp, ldadd);
      jx.addt = jt ? 1 : 0;
      const XtyJob jobs[2] = {jx, xty_job(nr, NUT + 1, NU + 1, &w.ps.JuT[0][0], NRP, &w.Tm[0][NX], LDTM, &w.ps.Jt[0][NX], LDTM)};
      if (jt) wg_xty_jobs<true, XTY_ADD_GLOBAL | XTY_ADD_T>(ctx, jobs, 2);
      else wg_xty_jobs<true, XTY_ADD_GLOBAL>(ctx, jobs, 2);
      return;
    }
    // (transposed layout only: the additive term of rows >= head starts `head` elements further along a column)
    XtyJob jt0 = xty_job(nr - head, NX, tL, &w.ps.JuT[tk0][head], NRP, &w.Tm[tk0][0], LDTM, &w.ps.Jt[head][0], LDTM, addp + head, ldadd);
    jt0.addt = 1;
    XtyJob jt1 = xty_job(nr - head, NUT + 1, tL, &w.ps.JuT[tk0][head], NRP, &w.Tm[tk0][NX], LDTM, &w.ps.Jt[head][NX], LDTM);
    jt1.L2 = 1; jt1.X2 = &w.ps.JuT[NU][head]; jt1.ldx2 = NRP; jt1.Y2 = &w.Tm[NU][NX]; jt1.ldy2 = LDTM; jt1.sign2 = 1.0;   // rho against e_NTW
    // (job counts are compile-time constants per branch: the unrolled form keeps the descriptors in registers)
    if (head > 0) {
      XtyJob jh0 = xty_job(head, NX, NU, &w.ps.JuT[0][0], NRP, &w.Tm[0][0], LDTM, &w.ps.Jt[0][0], LDTM, addp, ldadd);
      jh0.addt = 1;
      const XtyJob jobs[4] = {jh0, xty_job(head, NUT + 1, NU + 1, &w.ps.JuT[0][0], NRP, &w.Tm[0][NX], LDTM, &w.ps.Jt[0][NX], LDTM), jt0, jt1};
      wg_xty_jobs<true, XTY_ADD_GLOBAL | XTY_ADD_T>(ctx, jobs, 4);
    } else {
      const XtyJob jobs[2] = {jt0, jt1};
      wg_xty_jobs<true, XTY_ADD_GLOBAL | XTY_ADD_T>(ctx, jobs, 2);
    }
  };
  if (!(HSQP_PEXP & 64)) project_rows(0, NRP, 16, 0, 0);
  WG_SYNC(ctx);
  PH_TICK(ctx, 6);
  load_ju(NRP, NRP);
  if (!(HSQP_PEXP & 64)) gram_rows<false, NRP>(ctx, g, &w.ps.Jt[0][0], LDTM, nullptr, nullptr);
  store_ju(NRP, NRP);
  WG_SYNC(ctx);
  PH_TICK(ctx, 14);
  if (!(HSQP_PEXP & 4)) project_rows(NRP, NRP, 16, 6, 6);
  WG_SYNC(ctx);
  PH_TICK(ctx, 11);
  // the third pass only if rows beyond the first two passes are in use (the whole-body LQ kernel writes its rows compactly: 46 in
  // double support, 38 in single support; the collision rows, which would make it 54 / 62, are absent unless one of them is active)
  const bool pass3 = w.nrows > 2 * NRP;
  if (pass3) load_ju(2 * NRP, NRS - 2 * NRP);
  if (!(HSQP_PEXP & 4)) gram_rows<false, NRP>(ctx, g, &w.ps.Jt[0][0], LDTM, nullptr, nullptr);
  if (pass3) {
    store_ju(2 * NRP, NRS - 2 * NRP);
    WG_SYNC(ctx);
    PH_TICK(ctx, 15);
    project_rows(2 * NRP, NRS - 2 * NRP, 0, 0, 0);
    WG_SYNC(ctx);
    PH_TICK(ctx, 7);
    gram_rows<false, NRS - 2 * NRP>(ctx, g, &w.ps.Jt[0][0], LDTM, nullptr, nullptr);
  }
  // the input-weight rows sqrt(d_u) [Px | Pu | Pe] and the diagonal part of the gradient, straight from Tm
  if (!(HSQP_PEXP & 32)) gram_rows<true, NU>(ctx, g, &w.Tm[0][0], LDTM, &w.d[NX], &w.gd[NX]);
  PH_TICK(ctx, 12);
  if (!(HSQP_PEXP & 2)) gram_store(ctx, g, w, nut, qp, joint_rows);
  WG_SYNC(ctx);
  PH_TICK(ctx, 10);
}

}  // namespace hsqp
