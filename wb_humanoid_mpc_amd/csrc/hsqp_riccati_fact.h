// Riccati recursion of the whole-body stage QP on the FACTORS of its dynamics (the reference solves the same QP with
// HPIPM through ocs2's HpipmInterface, task.info:79-93 — the minimiser is unique, A.1).
//
// The projected dynamics of the whole-body formulation are almost entirely structure (hsqp_project.h, hsqp_lq.h):
//     [A~ | B~] = [E_J | 0] + F [Vx | Vu],      F = [e_base (12 columns) | T_J (23 columns)]   (58 x 35),
// with  E_J   the identity on the joint states plus dt on the (q_j, v_j) couplings (base rows zero),
//       T_J   dt^2/2 on row q_j and dt on row v_j of column j  (q_j+ = q_j + dt v_j + dt^2/2 qdd_j,  v_j+ = v_j + dt qdd_j  exactly),
//       V     the twelve dense base rows of [A~ | B~]  and  rows 12 .. 34 of [Px | Pu]  (qdd_j = input 12 + j)          (35 x 81).
// So every product of the stage with A~ or B~ is a product with F^T (a row COMBINATION: copy a base row, or hq * row q_j + dt * row v_j,
// formed while the operand is fetched) and a 35-deep contraction with V — nine 16 x 16 x 4 steps instead of fifteen — plus an E_J term
// that is a shifted copy added in the tile epilogue.  What that buys is not the flops as such (a phase of the stage is overhead-bound)
// but a different SCHEDULE: S A~ and W' = A~^T S A~ together are 117 matrix instructions per SIMD instead of 195 and run (all but a quarter of S A~) UNDER the
// elimination on the two SIMDs it leaves free, with no synchronisation between the four waves that form them: a wave owns one column
// tile of S A~ (four tiles) and forms, from it alone, the tiles of W' in that column (accumulators in registers, written over the
// upper-triangle tiles of S A~ once the column is consumed).  The stage (eight waves; per SIMD the waves w and w + 4):
//   Ph1  S B~ = (F^T S)^T Vu  (8 tiles x 9 steps: waves 0 .. 3 run both column tiles of their row tile in one call, the A operand formed once; the
//        combinations F^T S are KEPT: FS),  sb = s + S b~  (vector items: all that waves 4 .. 7 do in the phase)
//   Ph2  G = P~ + SB^T E_J + (F^T SB)^T Vx  (8 tiles),  [Lam | g] = [R~ | r~] + Vu^T (F^T [SB | sb])  (4 tiles, second tile of waves 4 .. 7's call);
//        P~, R~, r~ from LDS: no global load in the phase.  Waves 0 .. 3, one tile short of the others, form the first row tile of S A~ (HSQP_SA_EARLY)
//   Ph3  waves 0, 1: blocked elimination of [Lam | I | G | g] (hsqp_elim.h, unchanged); wave 0's hook issues every asynchronous copy of the next
//        stage (Vx, P~, R~) and both fetch Q~ of their S tiles into registers  ||  waves 2, 3, 6, 7: the other three row tiles of S A~, then W' — whose
//        last column tile carries sb as column NX of S A~ and leaves A~^T sb there  ||  waves 4, 5: the next stage's Vu, b~, r~ through address
//        tables formed once in front of the stage loop
//   Ph4  waves 0 .. 3: S = Q~ + W' - Z^T Z  (six 23-deep steps; 3 / 2 / 3 / 2 tiles, Q~ from registers) — the lane whose column is z forms the new
//        s = (A~^T sb + q~) - Z^T z by the same expression (A~^T sb through the W' fetch, q~ through the Q~ fetch);  waves 4 .. 7: two tiles each of
//        [K | k] = -L^-T [Z | z]
// i.e. 633 matrix instructions per stage against 909 (1.56 x the algorithmic count instead of 2.24 x), the two 58-deep phases gone
// from the serial path, and per stage 35 x 81 + 58 numbers of dynamics read instead of 58 x 82.  The forward sweep applies the factors
// as well (34 instead of 48 KB per stage) and hands rows 12 .. 34 of Px dx + Pu ut to the step kernel.  The centroidal formulation (35 states:
// F would have full rank) keeps hsqp_riccati.h.  What the device taught while this was built: DESIGN.md §4 "k_riccati_fact".
#pragma once
#include <cstddef>
#include "hsqp_riccati.h"
#ifndef HSQP_LAP_WAVE
#define HSQP_LAP_WAVE 4   /* -DHSQP_PHASE_PROFILE builds: the wave whose Ph4 is split into laps (slots 20 .. 25) */
#endif
#ifndef HSQP_EXP
#ifndef HSQP_FACT_PF
#define HSQP_FACT_PF RIC_PF   /* operand prefetch depth of this stage's tile loops (tuning builds) */
#endif
#ifndef HSQP_PH1_NT2
#define HSQP_PH1_NT2 1    /* Ph1: the two column tiles of a row tile of S B~ in one call on waves 0 .. 3 (0: one tile per wave; A/B builds) */
#endif
#ifndef HSQP_SA_EARLY
#define HSQP_SA_EARLY 1   /* the first row tile of SA on waves 0 .. 3 in Ph2 (0: all of SA in Ph3; A/B builds) */
#endif
#define HSQP_EXP 0   /* timing experiments of tuning builds (WRONG results): bit 0 memory waves without trailing copies / Q~, 1 no Vx copy, 2 memory waves idle, 3 no W' tiles, 4 no hook; (correct results:) 5 no s_setprio */
#endif

namespace hsqp {
constexpr int FACT_PF = HSQP_FACT_PF;

constexpr int NF = 12 + NJ;                   // rank of the factored part
constexpr int NFS = (NF + 3) / 4;             // contraction steps of four (index 35 is padding)
static_assert(NX == 2 * NV && NV == 6 + NJ && NU == 12 + NJ && NUT == NJ, "whole-body layout: x = [q_b(6) q_j | v_b(6) v_j], u = [W_l W_r qdd_j]");
static_assert(NFS * 4 == 36 && (12 % 4) == 0, "the base rows fill whole contraction steps: a step is either copies or joint combinations");

// state row of factor index k < 12 (dense base rows: q_b, then v_b); factor index k >= 12 is joint k - 12: rows k - 6 (q_j) and k + 23 (v_j)
HSQP_HD int fact_base_row(int k) { return k < 6 ? k : NV + k - 6; }
// (F^T M)[k][col] for a row-major M with 58 rows; `step` = k / 4 (a compile-time constant in the unrolled device loops: steps 0 .. 2 are
// plain copies).  k beyond the factors is clamped (the caller zeroes the operand).
template <class SC>
HSQP_HD double fact_combo(SC step, int kk, const double* M, int ld, int col, double dt, double hq) {
  const int s = step, k = 4 * s + kk;
  if (s < 3) return M[fact_base_row(k) * ld + col];
  const int kc = k < NF ? k : NF - 1;
  return hq * M[(kc - 6) * ld + col] + dt * M[(kc + 23) * ld + col];
}
HSQP_HD bool fact_is_jq(int i) { return i >= 6 && i < NV; }
HSQP_HD bool fact_is_jv(int i) { return i >= NV + 6; }
// (M E_J)[r][c]: column c of the product from the rows' entries
HSQP_HD double fact_ej_cols(const double* Mrow, int c, double dt) { return fact_is_jq(c) ? Mrow[c] : (fact_is_jv(c) ? Mrow[c] + dt * Mrow[c - NV] : 0.0); }
// (E_J^T M)[r][c]
HSQP_HD double fact_ej_rows(const double* M, int ld, int r, int c, double dt) {
  return fact_is_jq(r) ? M[r * ld + c] : (fact_is_jv(r) ? M[r * ld + c] + dt * M[(r - NV) * ld + c] : 0.0);
}
// where the factor rows sit in the QP record (hsqp_project.h): dense rows of A~ / B~ at their state rows, rows 12 .. of Px / Pu
HSQP_HD const double* fact_va_row(const double* q, int k) { return k < 12 ? q + QP_A + fact_base_row(k) * NX : q + QP_PX + k * NX; }
HSQP_HD const double* fact_vb_row(const double* q, int k) { return k < 12 ? q + QP_B + fact_base_row(k) * NUT : q + QP_PU + k * NUT; }

constexpr int LDG = NX + 2;                     // [G (58) | g] -> [K | k]: the block the elimination reads and the gains land in
constexpr int LDA = NX + 1;                     // SA carries one more column: sb (the W' tiles of the last column tile then form A~^T sb next to W')
constexpr int FG_GV = NX;                       // column of g (then of k)
struct RicFWS {
  double VA[2][NF][NX];                        // Vx of this / the next stage (stage k uses VA[k & 1]); the record's rows byte for byte
  double Pn[NUT][NX];                          // P~ of the stage (device: copied a stage ahead, asynchronously; a 16-byte multiple)
  union {
    double Rn[LDB][LDB];                       // R~ of the stage as the record holds it ([23][23] contiguous; device: asynchronous copy a stage ahead)
    double LinvT[LDB][LDB];                    // host build: (L^-1)^T of the generic elimination (the device's elimination does not form it)
  };
  double S[NX][NX];                            // value function
  double SA[NX][LDA];                          // [S A~ | sb]; its tiles on / above the diagonal are then overwritten with W' = E_J^T SA + Vx^T (F^T SA)
  double FS[NF][NX];                           // F^T S
  double VB[NF][LDB];                          // Vu (column 23 unused, zero)
  union {
    double SB[NX][LDB];                        // [S B~ | sb]
    double Zs[NUT + 1][LDZ];                   // [L^-1 G | z] (SB is dead once Lam, G are formed); row NUT: padding of the last contraction step (finite, multiplied by zero)
  };
  double PG[NUT][LDG];                         // [G | g] -> [K | k]
  double Ef[LDB][LDF];                         // [Lam | . | L^-1]
  double bt[NX], sb[NX], dx[NX], sv[NX], zv[LDB], kv[LDB], fsb[NFS * 4];
  double part[NX * 4];
  int ok;
};
static_assert(sizeof(RicFWS) <= 163400, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) accepts 163 400 bytes on gfx950");
static_assert(offsetof(RicFWS, VA) % 16 == 0 && (sizeof(double) * NF * NX) % 16 == 0 && offsetof(RicFWS, Pn) % 16 == 0 && offsetof(RicFWS, Rn) % 16 == 0,
              "16-byte chunks of the asynchronous copies");
static_assert(QP_A % 2 == 0 && (NV * NX) % 2 == 0 && (QP_PX + 12 * NX) % 2 == 0 && (6 * NX) % 2 == 0 && QP_SIZE % 2 == 0 && QP_P % 2 == 0 && (NUT * NX) % 2 == 0 && QP_R % 2 == 0 &&
              QP_QV == QP_R + NUT * NUT && (NUT * NUT + 1) / 2 * 2 <= LDB * LDB,
              "16-byte alignment of the factor rows and of P~, R~ in the record (the copy of R~ takes one double of q~ along)");

// branch-free forms of the E_J terms for the tile epilogues: every load unconditional (a per-lane branch around an LDS read is one
// dependent round trip per element — the first device build spent 5 k cycles of the stage in such epilogues)
HSQP_HD double fact_ej_mix(double a, double b, int i, double dt) {   // a = entry at index i, b = entry at the partner index i - 29 (any valid entry if there is none)
  const double m1 = (fact_is_jq(i) || fact_is_jv(i)) ? 1.0 : 0.0, m2 = fact_is_jv(i) ? dt : 0.0;
  return m1 * a + m2 * b;
}
HSQP_HD int fact_partner(int i) { return i >= NV ? i - NV : i; }

#if defined(__HIP_DEVICE_COMPILE__)
// NT tiles of  acc += X^T Y  over NS contraction steps of four, operands through functors: xf(step, t) / yf(step, t) = this lane's A / B
// operand of tile t (`step` a compile-time constant), travelling PF steps ahead in a ring of register sets as in xty_job_tiles_mfma
template <int NT, int PF, int NS, class XF, class YF>
__attribute__((always_inline)) HSQP_D void fact_mfma(hsqp_d4 (&acc)[NT], XF xf, YF yf) {
  double pa[PF][NT], pb[PF][NT];
  static_for<PF>([&](auto uc) {
    constexpr int u = decltype(uc)::value;
    if constexpr (u < NS) {
#pragma unroll
      for (int t = 0; t < NT; ++t) { pa[u][t] = xf(uc, t); pb[u][t] = yf(uc, t); }
    }
  });
  __builtin_amdgcn_sched_barrier(0);
  static_for<NS>([&](auto sc) {
    constexpr int s = decltype(sc)::value, u = s % PF;
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[u][t], pb[u][t], acc[t], 0, 0, 0);
    if constexpr (s + PF < NS) {
      constexpr std::integral_constant<int, s + PF> nx{};
#pragma unroll
      for (int t = 0; t < NT; ++t) { pa[u][t] = xf(nx, t); pb[u][t] = yf(nx, t); }
    }
    __builtin_amdgcn_sched_barrier(0);
  });
}
// The same with ONE A operand for all NT tiles (tiles of one row tile: xf(step) is fetched — and, where it is a combination of rows, formed — once)
template <int NT, int PF, int NS, class XF, class YF>
__attribute__((always_inline)) HSQP_D void fact_mfma_sx(hsqp_d4 (&acc)[NT], XF xf, YF yf) {
  double pa[PF], pb[PF][NT];
  static_for<PF>([&](auto uc) {
    constexpr int u = decltype(uc)::value;
    if constexpr (u < NS) {
      pa[u] = xf(uc);
#pragma unroll
      for (int t = 0; t < NT; ++t) pb[u][t] = yf(uc, t);
    }
  });
  __builtin_amdgcn_sched_barrier(0);
  static_for<NS>([&](auto sc) {
    constexpr int s = decltype(sc)::value, u = s % PF;
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[u], pb[u][t], acc[t], 0, 0, 0);
    if constexpr (s + PF < NS) {
      constexpr std::integral_constant<int, s + PF> nx{};
      pa[u] = xf(nx);
#pragma unroll
      for (int t = 0; t < NT; ++t) pb[u][t] = yf(nx, t);
    }
    __builtin_amdgcn_sched_barrier(0);
  });
}
// 16-byte chunks [0, n) of src -> dst (LDS) as asynchronous copies: wave instructions of 64 chunks, ALL lanes active in every one — the last
// instruction of a piece starts at n - 64 and re-copies what it overlaps.  (With a per-lane tail predicate `c0 + lane < n` the compiler merged
// the tail of one piece with the head of the next under the predicate and took the LDS base of the merged copy from the first active lane:
// lanes beyond the tail wrote the next piece to the wrong place.  Found by the GPU parity run; the builtin's LDS pointer must stay wave-uniform
// through every transformation, which straight-line code guarantees.)  NW = 2: `turn` alternates the instructions between two waves.
template <int N, int NW>
HSQP_D void fact_async_piece(const double* src, double* dst, int wave, int lane, int& turn) {
  static_assert(N >= 64 && (NW == 1 || NW == 2), "a piece is at least one full wave instruction; one or two waves share the pieces");
  constexpr int NI = (N + 63) / 64;
  static_for<NI>([&](auto ic) {
    constexpr int i = decltype(ic)::value, c0 = (i + 1) * 64 <= N ? i * 64 : N - 64;
    if (NW == 1 || ((turn + i) & 1) == wave) __builtin_amdgcn_global_load_lds((hsqp_gcptr)src + 2 * (c0 + lane), (hsqp_ldsptr)(dst + 2 * c0), 16, 0, 0);
  });
  turn += NI;
}
// the next stage's Vx -> LDS (17 wave instructions) and P~, R~ (16), all issued by wave 0: the faster of the two eliminating waves
HSQP_D void fact_next_vx_to_lds(const double* qn, double* va, int lane) {
  if (!qn) return;
  int turn = 0;
  fact_async_piece<6 * NX / 2, 1>(qn + QP_A, va, 0, lane, turn);
  fact_async_piece<6 * NX / 2, 1>(qn + QP_A + NV * NX, va + 6 * NX, 0, lane, turn);
  fact_async_piece<NJ * NX / 2, 1>(qn + QP_PX + 12 * NX, va + 12 * NX, 0, lane, turn);
}
HSQP_D void fact_next_cost_to_lds(const double* qn, double* pn, double* rn, int lane) {
  if (!qn) return;
  int turn = 0;
  fact_async_piece<NUT * NX / 2, 1>(qn + QP_P, pn, 0, lane, turn);
  fact_async_piece<(NUT * NUT + 1) / 2, 1>(qn + QP_R, rn, 0, lane, turn);
}
// tiles on / above the diagonal of a 4 x 4 tile grid, row by row
HSQP_D int fact_sym_tr(int id) { return id < 4 ? 0 : (id < 7 ? 1 : (id < 9 ? 2 : 3)); }
HSQP_D int fact_sym_tc(int id) { return id < 4 ? id : (id < 7 ? id - 3 : (id < 9 ? id - 5 : 3)); }
#endif

// qp: [N][QP_SIZE] of this instance (dense rows of A~ / B~, Px, Pu, b~ and the cost blocks are read; the joint rows of A~ / B~ are not),
// dts: [N] interval lengths, ric: [N][RIC_SIZE].  w.ok reports whether every Lam was positive definite.  vf (optional, [N+1][VF_SIZE]).
HSQP_HD void riccati_backward_fact(const Ctx& ctx, RicFWS& w, const double* Qf, const double* xN, const double* parN, const double* qp, const double* dts,
                                   double* ric, int N, double* vf = nullptr) {
  constexpr int NXE = NX;
  WG_FOR(ctx, i, NX * NX + NX + 1 + LDB * LDB + NF) {
    if (i < NX * NX) { const int r = i / NX, c = i % NX; w.S[r][c] = r == c ? Qf[r] : 0.0; }
    else if (i < NX * NX + NX) {
      const int r = i - NX * NX;
      const double sr = Qf[r] * (xN[r] - parN[HSQP_P_XDES + r]);
      w.part[4 * r] = sr; w.part[4 * r + 1] = 0.0; w.part[4 * r + 2] = 0.0; w.part[4 * r + 3] = 0.0;   // s travels as four partial sums (Ph4 -> Ph1)
    }
    else if (i == NX * NX + NX) w.ok = 1;
    else if (i < NX * NX + NX + 1 + LDB * LDB) { const int t = i - NX * NX - NX - 1; w.LinvT[t / LDB][t % LDB] = 0.0; }
    else w.VB[i - (NX * NX + NX + 1 + LDB * LDB)][NUT] = 0.0;
  }
  // Row NUT of Zs is the Y operand of the last contraction step's fourth lane group in Ph4 (X is zero there): it has to be FINITE.  Its first columns
  // lie over SB's last row (rewritten with finite values every stage), the rest is beyond SB and written by nobody — whatever the previous kernel
  // left in LDS, NaN bit patterns included (an intermittent "not positive definite" on the device until this was zeroed)
  WG_FOR(ctx, i, LDZ) w.Zs[NUT][i] = 0.0;
  WG_SYNC(ctx);
  if (vf) {
    WG_FOR(ctx, i, VF_SIZE) vf[(size_t)N * VF_SIZE + i] = i < NX * NX ? w.S[i / NX][i % NX] : w.part[4 * (i - NX * NX)];
  }
  // factors and cost blocks of stage N - 1 -> LDS (the later stages are fetched while the previous one is being processed)
  {
    const double* q = qp + (size_t)(N - 1) * QP_SIZE;
    double(*Vn)[NX] = w.VA[(N - 1) & 1];
    constexpr int n1 = NF * NX, n2 = n1 + NF * NUT, n3 = n2 + NX, n4 = n3 + NUT, n5 = n4 + NUT * NX, n6 = n5 + NUT * NUT;
    WG_FOR(ctx, it, n6) {
      if (it < n1) { const int k = it / NX, c = it % NX; Vn[k][c] = fact_va_row(q, k)[c]; }
      else if (it < n2) { const int e = it - n1, k = e / NUT, c = e % NUT; w.VB[k][c] = fact_vb_row(q, k)[c]; }
      else if (it < n3) w.bt[it - n2] = q[QP_BV + it - n2];
      else if (it < n4) w.kv[it - n3] = q[QP_RV + it - n3];
      else if (it < n5) (&w.Pn[0][0])[it - n4] = q[QP_P + it - n4];
      else (&w.Rn[0][0])[it - n5] = q[QP_R + it - n5];
    }
  }
  WG_SYNC(ctx);
  const Ctx& ctx_outer = ctx;
#if defined(__HIP_DEVICE_COMPILE__)
  // Loop-invariant address tables of the waves that only move data in Ph3 (and of the Q~ fetch), formed ONCE: offsets relative to the record of the
  // NEXT stage (records are contiguous: this stage's q~ is QP_SIZE further) and LDS destinations.  Inside the stage loop the thread index is opaque
  // (see below), so everything derived from it is recomputed per stage — for these waves that was ~400 vector instructions of index arithmetic per
  // stage, issued in the gaps the eliminating wave of the same SIMD leaves (it is older and nearly always ready): the memory waves reached the
  // phase's barrier 1 - 2 k cycles AFTER the elimination, whatever they loaded (profiles/r06_ric_experiments.txt).
  constexpr int FM_NPB = 7, FM_NVB = NF * NUT;
  static_assert(128 * FM_NPB >= FM_NVB + NX + NUT, "one pass of the two memory waves");
  // ONE table of 25 integers per lane, overlaid by role (a wave's role never changes, and the registers are allotted to the kernel, not to a role):
  //   waves 0 .. 3 (S tiles of Ph4; three tiles t):  FS_Q(t) Q~ fetch offset (from the stage's record; rows + 4 r are 4 NX further), FS_ZX / FS_ZY(t) operand offsets into Zs
  //     (steps + 4 s are 4 LDZ further), FS_W(t) offset of the tile's element in SA / S, FS_M(t) of its mirror image in S (z lane: the first of its rows), FS_MASK(t) which of
  //     its four elements are stored (bit r; bits 4 + r: the lane's column is z and row r exists — the new s leaves through it; bit 8: z lane — its FS_Q is q~, rows 4 apart,
  //     its FS_W column NX of SA, where Ph3 left A~^T sb)
  //   waves 4 .. 7 (K tiles of Ph4, column tile wv - 4, both row tiles t):  FK_X(t) operand offset into Ef (L^-1), FK_Y into Zs, FK_PG(t) / FK_RK(t) offsets of the tile's element
  //     in PG and in the gains record (rows + 4 r: 4 LDG / 4 NX further), FK_MASK(t) which elements exist (bit r; bit 4 + r: also in the record)
  //   waves 4, 5 (the memory waves of Ph3):  FM_SRC(t) source offset (doubles, from the next stage's record), FM_DST(t) LDS destination (doubles, from &w.VA[0][0][0]; -1: none)
  int ftab[25];
#define FS_Q(t) ftab[(t)]
#define FS_ZX(t) ftab[3 + (t)]
#define FS_ZY(t) ftab[6 + (t)]
#define FS_W(t) ftab[9 + (t)]
#define FS_M(t) ftab[12 + (t)]
#define FS_MASK(t) ftab[15 + (t)]
#define FK_X(t) ftab[(t)]
#define FK_Y ftab[2]
#define FK_PG(t) ftab[3 + (t)]
#define FK_RK(t) ftab[5 + (t)]
#define FK_MASK(t) ftab[7 + (t)]
#define FM_SRC(t) ftab[9 + (t)]
#define FM_DST(t) ftab[17 + (t)]
  int f_sfirst, f_scount;
  {
    const int tid0 = ctx_outer.tid, wv0 = tid0 >> 6, lane0 = tid0 & 63, li0 = lane0 & 15, kk0 = lane0 >> 4, pt = tid0 - 256;
    double* const lbase = &w.VA[0][0][0];
#pragma unroll
    for (int i = 0; i < 25; ++i) ftab[i] = 0;
    // Ph4 per SIMD: ONE wave forms S tiles (ids 0 .. 9 of the upper triangle, row by row: waves 0, 2 three each, waves 1, 3 two each — wave 1 carries the
    // larger share of the elimination), the other one (waves 4 .. 7) two K tiles
    f_sfirst = wv0 == 0 ? 0 : (wv0 == 1 ? 3 : (wv0 == 2 ? 5 : 8)); f_scount = wv0 >= 4 ? 0 : ((wv0 & 1) ? 2 : 3);
    if (wv0 < 4) {
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int id = t < f_scount ? f_sfirst + t : f_sfirst, tr = fact_sym_tr(id), tc = fact_sym_tc(id);
        const int c = 16 * tc + li0, cc = c < NX ? c : NX - 1, rr = 16 * tr + li0 < NX ? 16 * tr + li0 : NX - 1, row0 = 16 * tr + kk0;
        const bool zl = c == NX && t < f_scount;  // the lane whose column is z (column NX of [Z | z]): its accumulators are (Z^T z)[rows] — the new s leaves through it
        FS_Q(t) = zl ? QP_QV + row0 : QP_Q + row0 * NX + cc;   // (rows beyond the matrix — the last tile row — address valid memory of the record; their values are never stored; z lane: q~, rows 4 apart)
        FS_ZX(t) = kk0 * LDZ + rr;
        FS_ZY(t) = kk0 * LDZ + (c <= NX ? c : NX - 1);
        FS_W(t) = row0 * NX + (zl ? NX : cc);     // (z lane: column NX of SA — A~^T sb, left there by Ph3; it stores nothing through this entry)
        FS_M(t) = zl ? row0 : cc * NX + row0;
        int mask = 0;
        for (int r = 0; r < 4; ++r) { const int row = row0 + 4 * r; if (t < f_scount && c < NX && row < NX && (tr != tc || row <= c)) mask |= 1 << r; if (zl && row < NX) mask |= 16 << r; }
        if (zl) mask |= 256;
        FS_MASK(t) = mask;
      }
    } else {
      const int ct = wv0 - 4, c = 16 * ct + li0, cz = c <= NX ? c : NX;
      FK_Y = kk0 * LDZ + cz;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int rr = 16 * t + li0 < NUT ? 16 * t + li0 : NUT - 1, row0 = 16 * t + kk0;
        FK_X(t) = kk0 * LDF + EF_MI + rr;
        FK_PG(t) = row0 * LDG + cz;
        FK_RK(t) = RIC_K + row0 * NX + (c < NX ? c : NX - 1);
        int mask = 0;
        for (int r = 0; r < 4; ++r) { const int row = row0 + 4 * r; if (row < NUT && c <= NX) mask |= (1 << r) | (c < NX ? 16 << r : 0); }
        FK_MASK(t) = mask;
      }
#pragma unroll
      for (int t = 0; t < FM_NPB; ++t) {
        const int idx = pt + 128 * t;
        int so = 0, dd = -1;
        if (idx >= 0 && idx < FM_NVB) { const int kf = idx / NUT, c2 = idx - kf * NUT; so = (int)(fact_vb_row(qp, kf) - qp) + c2; dd = (int)(&w.VB[kf][c2] - lbase); }
        else if (idx >= FM_NVB && idx < FM_NVB + NX) { so = QP_BV + idx - FM_NVB; dd = (int)(&w.bt[idx - FM_NVB] - lbase); }
        else if (idx >= FM_NVB + NX && idx < FM_NVB + NX + NUT) { so = QP_RV + idx - FM_NVB - NX; dd = (int)(&w.kv[idx - FM_NVB - NX] - lbase); }
        FM_SRC(t) = so; FM_DST(t) = dd;
      }
    }
  }
#endif
  for (int k = N - 1; k >= 0; --k) {
    // (the thread index is made opaque once per stage: see riccati_backward)
    Ctx ctx = ctx_outer;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HSQP_FACT_NO_OPAQUE)
    asm volatile("" : "+v"(ctx.tid));
#endif
    const double* q = qp + (size_t)k * QP_SIZE;
    const double* qn = k > 0 ? qp + (size_t)(k - 1) * QP_SIZE : nullptr;   // next stage to be processed
    double* rk = ric + (size_t)k * RIC_SIZE;
    const double dt = dts[k], hq = 0.5 * dt * dt;
    double(*VA)[NX] = w.VA[k & 1];
    double(*VAn)[NX] = w.VA[(k + 1) & 1];
    const double* Rn = &w.Rn[0][0];               // R~[r][c] = Rn[r * NUT + c]
    PH_TICK(ctx, 1);
    PH_MARK(ctx);
#if defined(__HIP_DEVICE_COMPILE__)
    const int wv = wave_index(ctx.tid), lane = ctx.tid & 63, li = lane & 15, kk = lane >> 4;
    // ---- Ph1: SB = (F^T S)^T Vu — one tile per wave; the waves of the first column tile also keep the combinations they fetch (FS);
    //      the helper half first: sb = s + S b~ (four partial sums per row, closed by DPP quad permutes), k of the previous stage -> record
    {
      // (ONE wave per SIMD carries the vector items, all 64 lanes: a SIMD runs the vector instructions of its two waves one after the other, so what a
      //  phase costs a SIMD is the SUM over its waves — the same items dealt to all eight waves at 32 lanes each were twice the instructions per SIMD
      //  and measurably slower)
      if (wv >= 4) {
        const int it = ctx.tid - 256;
        if (it < 4 * NX) {
          const int r = it >> 2, p = it & 3;
          constexpr int LA = (NX + 3) / 4;
          double sacc = 0.0;
          if (p == 0) { const double* sp = &w.part[4 * r]; sacc = (sp[0] + sp[1]) + (sp[2] + sp[3]); }
#pragma unroll
          for (int l = 0; l < LA; ++l) { const int ll = p * LA + l, lc = ll < NX ? ll : NX - 1; const double a = w.S[r][lc], b = w.bt[lc]; sacc += ll < NX ? a * b : 0.0; }
          sacc += quad_perm_f64<0xB1>(sacc);
          sacc += quad_perm_f64<0x4E>(sacc);
          if (p == 0) { w.sb[r] = sacc; w.SB[r][NUT] = sacc; w.SA[r][NX] = sacc; }
        } else if (it < 4 * NX + NUT && k < N - 1) ric[(size_t)(k + 1) * RIC_SIZE + RIC_KV + it - 4 * NX] = w.PG[it - 4 * NX][FG_GV];
      }
      if (HSQP_PH1_NT2) {
        // waves 0 .. 3: BOTH column tiles of row tile wv in one call — the A operand (a combination of rows of S: two or three LDS reads and two
        // multiply-adds per element) is formed once for the two tiles instead of once per tile on two waves of the same SIMD
        if (wv < 4) {
          const int r0 = wv << 4, xr = r0 + li < NX ? r0 + li : NX - 1;
          hsqp_d4 acc[2] = {hsqp_d4{0.0, 0.0, 0.0, 0.0}, hsqp_d4{0.0, 0.0, 0.0, 0.0}};
          auto xf = [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            double v = fact_combo(sc, kk, &w.S[0][0], NX, xr, dt, hq);
            if (4 * s + 3 >= NF) v = 4 * s + kk < NF ? v : 0.0;
            if (r0 + li < NX && 4 * s + kk < NF) w.FS[4 * s + kk][xr] = v;
            return v;
          };
          auto yf = [&](auto sc, int t) { constexpr int s = decltype(sc)::value; const int kc = 4 * s + kk < NF ? 4 * s + kk : NF - 1; return w.VB[kc][16 * t + li < LDB ? 16 * t + li : LDB - 1]; };
          fact_mfma_sx<2, FACT_PF, NFS>(acc, xf, yf);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int col = 16 * t + li;
            if (col < NUT) {
#pragma unroll
              for (int r = 0; r < 4; ++r) { const int row = r0 + kk + 4 * r; if (row < NX) w.SB[row][col] = acc[t][r]; }
            }
          }
        }
      } else {
      const int rt = wv & 3, ct = wv >> 2, r0 = rt << 4, c0 = ct << 4;
      const int xr = r0 + li < NX ? r0 + li : NX - 1, yc = c0 + li < LDB ? c0 + li : LDB - 1;
      hsqp_d4 acc[1] = {hsqp_d4{0.0, 0.0, 0.0, 0.0}};
      auto xf = [&](auto sc, int) {
        constexpr int s = decltype(sc)::value;
        double v = fact_combo(sc, kk, &w.S[0][0], NX, xr, dt, hq);
        if (4 * s + 3 >= NF) v = 4 * s + kk < NF ? v : 0.0;
        if (ct == 0 && r0 + li < NX && 4 * s + kk < NF) w.FS[4 * s + kk][xr] = v;
        return v;
      };
      auto yf = [&](auto sc, int) { constexpr int s = decltype(sc)::value; const int kc = 4 * s + kk < NF ? 4 * s + kk : NF - 1; return w.VB[kc][yc]; };
      fact_mfma<1, FACT_PF, NFS>(acc, xf, yf);
      const int col = c0 + li;
      if (col < NUT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = r0 + kk + 4 * r; if (row < NX) w.SB[row][col] = acc[0][r]; }
      }
      }
    }
#else
    WG_FOR(ctx, it, NF * NX) { const int kf = it / NX, r = it % NX; w.FS[kf][r] = fact_combo(kf >> 2, kf & 3, &w.S[0][0], NX, r, dt, hq); }
    WG_FOR(ctx, it, NX + NUT) {
      if (it < NX) {
        const double* sp = &w.part[4 * it];
        double sacc = (sp[0] + sp[1]) + (sp[2] + sp[3]);
        for (int l = 0; l < NX; ++l) sacc += w.S[it][l] * w.bt[l];
        w.sb[it] = sacc;
      } else if (k < N - 1) ric[(size_t)(k + 1) * RIC_SIZE + RIC_KV + it - NX] = w.PG[it - NX][FG_GV];
    }
    WG_SYNC(ctx);
    WG_FOR(ctx, it, NX * LDB) {
      const int r = it / LDB, c = it % LDB;
      double sacc = 0.0;
      if (c < NUT) { for (int kf = 0; kf < NF; ++kf) sacc += w.FS[kf][r] * w.VB[kf][c]; } else sacc = w.sb[r];
      w.SB[r][c] = sacc;
    }
#endif
    PH_ARRIVE(ctx, 0);
    WG_SYNC(ctx);
    PH_TICK(ctx, 2);
    PH_MARK(ctx);
#if defined(__HIP_DEVICE_COMPILE__)
    // SA tiles (rt0 .. rt0 + NT - 1, ct) = FS^T Vx + S E_J: nine 35-deep steps, the E_J term from S's own column (and its partner column) in the epilogue
    auto sa_tiles = [&](auto ntc, int rt0, int ct) {
      constexpr int NT = decltype(ntc)::value;
      const int col = (ct << 4) + li, yc = col < NX ? col : NX - 1, ycp = fact_partner(yc);
      hsqp_d4 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = hsqp_d4{0.0, 0.0, 0.0, 0.0};
      auto xf = [&](auto sc, int t) {
        constexpr int s = decltype(sc)::value;
        const int kc = 4 * s + kk < NF ? 4 * s + kk : NF - 1, r = 16 * (rt0 + t) + li;
        const double v = w.FS[kc][r < NX ? r : NX - 1];
        return 4 * s + kk < NF ? v : 0.0;
      };
      auto yf = [&](auto sc, int) { constexpr int s = decltype(sc)::value; const int kc = 4 * s + kk < NF ? 4 * s + kk : NF - 1; return VA[kc][yc]; };
      fact_mfma<NT, FACT_PF, NFS>(acc, xf, yf);
      double ea[NT][4], eb[NT][4];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = 16 * (rt0 + t) + kk + 4 * r, rc = row < NX ? row : NX - 1; ea[t][r] = w.S[rc][yc]; eb[t][r] = w.S[rc][ycp]; }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = 16 * (rt0 + t) + kk + 4 * r; if (col < NX && row < NX) w.SA[row][col] = acc[t][r] + fact_ej_mix(ea[t][r], eb[t][r], yc, dt); }
    };
#endif
    // ---- Ph2: G = P~ + SB^T E_J + (F^T SB)^T Vx;  [Lam | g] = [R~ | r~] + Vu^T (F^T [SB | sb]);  fsb = F^T sb
#if defined(__HIP_DEVICE_COMPILE__)
    {
      // G: one tile per wave (2 row tiles x 4 column tiles); [Lam | g]: four tiles on waves 4 .. 7 (every SIMD then carries three tiles) — as the
      // second tile of ONE call (a tile call is ~2.5 k cycles whatever it contracts: two calls in a row were the phase's critical path).  No global
      // load in the phase: P~, R~ arrived in LDS a stage ahead.
      const int grt = wv >> 2, gct = wv & 3, gr0 = grt << 4, gc0 = gct << 4;
      const int gxr = gr0 + li < NUT ? gr0 + li : NUT - 1, gyc = gc0 + li < NX ? gc0 + li : NX - 1, gycp = fact_partner(gyc);
      const int lrt = (wv - 4) >> 1, lct = (wv - 4) & 1, lr0 = lrt << 4, lc0 = lct << 4;                // (waves 4 .. 7 only)
      const int lxr = lr0 + li < NUT ? lr0 + li : NUT - 1, lyc = lc0 + li < LDB ? lc0 + li : LDB - 1;
      auto xf = [&](auto sc, int t) {
        constexpr int s = decltype(sc)::value;
        const int kc = 4 * s + kk < NF ? 4 * s + kk : NF - 1;
        const double v = t == 0 ? fact_combo(sc, kk, &w.SB[0][0], LDB, gxr, dt, hq) : w.VB[kc][lxr];
        return 4 * s + kk < NF ? v : 0.0;
      };
      auto yf = [&](auto sc, int t) {
        constexpr int s = decltype(sc)::value;
        const int kc = 4 * s + kk < NF ? 4 * s + kk : NF - 1;
        return t == 0 ? VA[kc][gyc] : fact_combo(sc, kk, &w.SB[0][0], LDB, lyc, dt, hq);
      };
      hsqp_d4 acc[2] = {hsqp_d4{0.0, 0.0, 0.0, 0.0}, hsqp_d4{0.0, 0.0, 0.0, 0.0}};
      if (wv >= 4) fact_mfma<2, FACT_PF, NFS>(acc, xf, yf);
      else { hsqp_d4 a1[1] = {acc[0]}; fact_mfma<1, FACT_PF, NFS>(a1, xf, yf); acc[0] = a1[0]; }
      {
        // (SB^T E_J)[row][col]: column `col` of E_J picks row col (and row col - 29) of SB
        double pv[4], ea[4], eb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = gr0 + kk + 4 * r, rc = row < NUT ? row : NUT - 1; pv[r] = w.Pn[rc][gyc]; ea[r] = w.SB[gyc][rc]; eb[r] = w.SB[gycp][rc]; }
        const int col = gc0 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = gr0 + kk + 4 * r; if (col < NX && row < NUT) w.PG[row][col] = (acc[0][r] + pv[r]) + fact_ej_mix(ea[r], eb[r], gyc, dt); }
      }
      if (wv >= 4) {
        double av[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // (both reads unconditional, then a select)
          const int row = lr0 + kk + 4 * r, rc = row < NUT ? row : NUT - 1;
          const double a1 = Rn[rc * NUT + (lyc < NUT ? lyc : NUT - 1)], a2 = w.kv[rc];
          av[r] = lyc < NUT ? a1 : a2;
        }
        const int col = lc0 + li;
        if (col <= NUT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = lr0 + kk + 4 * r;
            if (row < NUT) {
              if (col < NUT) w.Ef[row][col] = acc[1][r] + av[r];
              else w.PG[row][FG_GV] = acc[1][r] + av[r];   // g = r~ + B~^T sb (r~ staged in kv a stage ahead)
            }
          }
        }
      }
      // waves 0 .. 3 have one tile in this phase, waves 4 .. 7 two: the first row tile of SA's column tile wv (S, FS, Vx: all there since Ph1) moves
      // from Ph3 — whose tile waves are 2 k cycles behind the elimination — into the difference
      else if (HSQP_SA_EARLY) sa_tiles(std::integral_constant<int, 1>{}, 0, wv);
    }
#else
    WG_FOR(ctx, it, NFS * 4) w.fsb[it] = it < NF ? fact_combo(it >> 2, it & 3, w.sb, 1, 0, dt, hq) : 0.0;
    WG_FOR(ctx, it, NUT * NX + NUT * LDB) {
      if (it < NUT * NX) {
        const int r = it / NX, c = it % NX;
        double sacc = 0.0;
        for (int kf = 0; kf < NF; ++kf) sacc += fact_combo(kf >> 2, kf & 3, &w.SB[0][0], LDB, r, dt, hq) * VA[kf][c];
        w.PG[r][c] = (sacc + w.Pn[r][c]) + fact_ej_mix(w.SB[c][r], w.SB[fact_partner(c)][r], c, dt);
      } else {
        const int e = it - NUT * NX, r = e / LDB, c = e % LDB;
        double sacc = 0.0;
        for (int kf = 0; kf < NF; ++kf) sacc += w.VB[kf][r] * fact_combo(kf >> 2, kf & 3, &w.SB[0][0], LDB, c, dt, hq);
        if (c < NUT) w.Ef[r][c] = sacc + Rn[r * NUT + c];
        else w.PG[r][FG_GV] = sacc + w.kv[r];
      }
    }
#endif
    PH_ARRIVE(ctx, 1);
    WG_SYNC(ctx);
    PH_TICK(ctx, 3);
    PH_MARK(ctx);
    // ---- Ph3: factorisation (leaves L^-1 in Ef columns 24.., Z = L^-1 G in Zs, z in zv) || SA = S A~, then W' (into SA) || next stage -> LDS
#if defined(__HIP_DEVICE_COMPILE__)
    // Q~ of the S-update tiles of Ph4 travels in registers of the waves that form them there (0, 1, 4, 5): fetched here, where those waves
    // have no LDS read behind the fetch for thousands of cycles (the compiler makes every LDS read wait for ALL outstanding vector-memory
    // loads of the wave once asynchronous LDS copies are in the kernel: a global load in front of a tile loop costs the tile an HBM round trip)
    double qpre[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) qpre[t][r] = 0.0;
    const int sfirst = __builtin_amdgcn_readfirstlane(f_sfirst), scount = __builtin_amdgcn_readfirstlane(f_scount);
    auto fetch_qpre = [&]() {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) qpre[t][r] = ((hsqp_gcptr)q)[FS_Q(t) + r * ((FS_MASK(t) & 256) ? 4 : 4 * NX)];
    };
    {
      if (wv < 2) {
        if (!(HSQP_EXP & 32)) __builtin_amdgcn_s_setprio(3);
        const DevWave dw{lane};
        const ElimIO io{&w.Ef[0][0], LDF, &w.PG[0][0], &w.PG[0][FG_GV], LDG, &w.Ef[0][EF_MI], LDF, nullptr, 0, &w.Zs[0][0], LDZ, w.zv, &w.ok};
        double* v_dst = &VAn[0][0];
        // wave 0 (the faster of the two) issues every asynchronous copy of the next stage: Vx, P~, R~
        auto prefetch = [&]() {
          if (HSQP_EXP & 16) return;
          fetch_qpre();
          if (wv == 0 && !(HSQP_EXP & 2)) { fact_next_vx_to_lds(qn, v_dst, lane); fact_next_cost_to_lds(qn, &w.Pn[0][0], &w.Rn[0][0], lane); }
        };
        if (wv == 0) eliminate_blocked<NXE, 0>(dw, io, prefetch);
        else eliminate_blocked<NXE, 1>(dw, io, prefetch);
        __builtin_amdgcn_s_setprio(0);
      } else if ((wv & 3) < 2) {
        if (HSQP_EXP & 4) {} else {
        // waves 4, 5 share their SIMDs with the eliminating waves: memory only.  The next stage's Vu, b~, r~ through the
        // address tables: every load in flight before the first store, no index arithmetic in here
        double pb[FM_NPB];
        double* const lbase = &w.VA[0][0][0];
        if (qn) {
#pragma unroll
          for (int t = 0; t < FM_NPB; ++t) pb[t] = ((hsqp_gcptr)qn)[FM_SRC(t)];
#pragma unroll
          for (int t = 0; t < FM_NPB; ++t) if (FM_DST(t) >= 0) lbase[FM_DST(t)] = pb[t];
        }
        }
      } else {
        // waves 2, 6 (SIMD 2) own column tiles 0, 3, waves 3, 7 (SIMD 3) column tiles 1, 2: 45 + 72 and 54 + 63 matrix instructions.  No global
        // access at all in here.
        const int ct = wv == 2 ? 0 : (wv == 6 ? 3 : (wv == 3 ? 1 : 2)), c0 = ct << 4;
        const int col = c0 + li, yc = col < NX ? col : NX - 1, ycp = fact_partner(yc);
        hsqp_d4 acc[4];
        if (HSQP_SA_EARLY) sa_tiles(std::integral_constant<int, 3>{}, 1, ct);   // (row tile 0 of every column tile: formed in Ph2)
        else sa_tiles(std::integral_constant<int, 4>{}, 0, ct);
        WV_SYNC();
        // W' tiles (rt <= ct, ct) from the wave's own column of SA; the accumulators wait in registers until the column is consumed
        if (!(HSQP_EXP & 8)) {
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = hsqp_d4{0.0, 0.0, 0.0, 0.0};
          // Column NX of the last column tile is not a column of S A~: it carries sb (as if sb were column NX of SA).  The same tiles then leave
          // A~^T sb = E_J^T sb + Vx^T (F^T sb) in that column — the vector part of the new s, which rounds 3 - 6 summed by 232 items in Ph4
          const bool sbl = col == NX;                                   // (lane 10 of every row group of wave 6)
          const double* ym = &w.SA[0][0];
          constexpr int yld = LDA;
          const int ycs = col < NX ? col : NX;
          auto yf = [&](auto sc, int) { return fact_combo(sc, kk, ym, yld, ycs, dt, hq); };
          auto xf = [&](auto sc, int t) {
            constexpr int s = decltype(sc)::value;
            const int kc = 4 * s + kk < NF ? 4 * s + kk : NF - 1, r = 16 * t + li;
            const double v = VA[kc][r < NX ? r : NX - 1];
            return 4 * s + kk < NF ? v : 0.0;
          };
          if (ct == 0) { hsqp_d4 a1[1] = {acc[0]}; fact_mfma<1, FACT_PF, NFS>(a1, xf, yf); acc[0] = a1[0]; }
          else if (ct == 1) { hsqp_d4 a2[2] = {acc[0], acc[1]}; fact_mfma<2, FACT_PF, NFS>(a2, xf, yf); acc[0] = a2[0]; acc[1] = a2[1]; }
          else if (ct == 2) { hsqp_d4 a3[3] = {acc[0], acc[1], acc[2]}; fact_mfma<3, FACT_PF, NFS>(a3, xf, yf); acc[0] = a3[0]; acc[1] = a3[1]; acc[2] = a3[2]; }
          else fact_mfma<4, FACT_PF, NFS>(acc, xf, yf);
          // (E_J^T SA)[row][col]: row `row` of E_J^T picks row `row` (and row - 29) of SA — unconditional reads of the own column, then the stores
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int row = 16 * t + kk + 4 * r, rc = row < NX ? row : NX - 1; acc[t][r] += fact_ej_mix(ym[rc * yld + ycs], ym[fact_partner(rc) * yld + ycs], rc, dt); }
          WV_SYNC();   // every read of the column is done before it is overwritten
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = 16 * t + kk + 4 * r;
              if (t <= ct && col < NX && row < NX) w.SA[row][col] = acc[t][r];
            }
          // (A~^T sb)[rows] of the sb lanes: picked up by the S tiles of the last column in Ph4.  ONE branch of the wave around all of them (a lane
          // condition per store was 16 branches on every tile wave: 2.2 k cycles of wave 2's epilogue)
          if (ct == 3 && sbl) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) { const int row = 16 * t + kk + 4 * r; if (row < NX) w.SA[row][NX] = acc[t][r]; }
          }
        }
        // waves 2, 3 form one S tile each in Ph4: its Q~ is fetched here, behind the wave's last LDS read of the phase (they finish 2 - 3 k cycles
        // before the elimination: the round trip runs under their wait at the barrier)
        if (wv < 4) fetch_qpre();
      }
    }
#else
    {
      WG_FOR(ctx, it, NX * NX) {
        const int r = it / NX, c = it % NX;
        double sacc = 0.0;
        for (int kf = 0; kf < NF; ++kf) sacc += w.FS[kf][r] * VA[kf][c];
        w.SA[r][c] = sacc + fact_ej_mix(w.S[r][c], w.S[r][fact_partner(c)], c, dt);
      }
      WG_SYNC(ctx);
      // W' on / above the diagonal (the device keeps it in SA's upper tiles; here it goes to S, which is dead, and the S-update reads it there)
      WG_FOR(ctx, it, NX * NX) {
        const int r = it / NX, c = it % NX;
        if (r > c) continue;
        double sacc = 0.0;
        for (int kf = 0; kf < NF; ++kf) sacc += VA[kf][r] * fact_combo(kf >> 2, kf & 3, &w.SA[0][0], LDA, c, dt, hq);
        w.S[r][c] = sacc + fact_ej_mix(w.SA[r][c], w.SA[fact_partner(r)][c], r, dt);
      }
      // elimination of [Lam | I] column by column, the Cholesky scaling, then Z, z as products (the generic form of riccati_backward)
      WG_FOR(ctx, j, NUT * (LDF - NUT)) { const int r = j / (LDF - NUT), c = NUT + j % (LDF - NUT); w.Ef[r][c] = (c - EF_MI == r) ? 1.0 : 0.0; }
      WG_FOR(ctx, it, NX) w.dx[it] = q[QP_QV + it];
      WG_SYNC(ctx);
      for (int j = 0; j < NUT - 1; ++j) {
        WG_FOR(ctx, it, NUT * 16) {
          const int i = it >> 4, c0 = it & 15;
          if (i <= j) continue;
          const double pj = w.Ef[j][j], fji = w.Ef[j][i];
          const double ej0 = w.Ef[j][c0], ej1 = w.Ef[j][c0 + 16], ej2 = w.Ef[j][c0 + 32];
          const double ei0 = w.Ef[i][c0], ei1 = w.Ef[i][c0 + 16], ei2 = w.Ef[i][c0 + 32];
          const double f = fji * fast_rcp(pj);
          w.Ef[i][c0] = ei0 - f * ej0;
          w.Ef[i][c0 + 16] = ei1 - f * ej1;
          w.Ef[i][c0 + 32] = ei2 - f * ej2;
        }
        WG_SYNC(ctx);
      }
      WG_FOR(ctx, it, NUT * LDB) {   // L^-1 = D^-1/2 Mi (in place) and its transpose (over R~, which Ph2 has consumed)
        const int r = it / LDB, c = it % LDB;
        double dj = w.Ef[r][r];
        if (!(dj > 0.0)) { dj = 1.0; if (c == 0) w.ok = 0; }
        const double v = c <= r ? w.Ef[r][EF_MI + c] * inv_sqrt(dj) : 0.0;
        w.Ef[r][EF_MI + c] = v;
        w.LinvT[c][r] = v;
      }
      WG_SYNC(ctx);
      {
        const XtyJob job = xty_job(NUT, NXE, NUT, &w.LinvT[0][0], LDB, &w.PG[0][0], LDG, &w.Zs[0][0], LDZ);
        wg_xty_jobs(ctx, &job, 1);
        WG_FOR(ctx, it, NUT) {
          double s = 0.0;
          for (int l = 0; l < NUT; ++l) s += w.Ef[it][EF_MI + l] * w.PG[l][FG_GV];
          w.zv[it] = s;
          w.Zs[it][NXE] = s;
        }
      }
      if (qn) {
        WG_SYNC(ctx);
        constexpr int n1 = NF * NX, n2 = n1 + NF * NUT, n3 = n2 + NX, n4 = n3 + NUT, n5 = n4 + NUT * NX, n6 = n5 + NUT * NUT;
        WG_FOR(ctx, it, n6) {
          if (it < n1) { const int kf = it / NX, c = it % NX; VAn[kf][c] = fact_va_row(qn, kf)[c]; }
          else if (it < n2) { const int e = it - n1, kf = e / NUT, c = e % NUT; w.VB[kf][c] = fact_vb_row(qn, kf)[c]; }
          else if (it < n3) w.bt[it - n2] = qn[QP_BV + it - n2];
          else if (it < n4) w.kv[it - n3] = qn[QP_RV + it - n3];
          else if (it < n5) (&w.Pn[0][0])[it - n4] = qn[QP_P + it - n4];
          else (&w.Rn[0][0])[it - n5] = qn[QP_R + it - n5];
        }
      }
    }
#endif
    PH_ARRIVE(ctx, 3);
    WG_SYNC(ctx);
    PH_TICK(ctx, 4);
    PH_MARK(ctx);
    // ---- Ph4: S = Q~ + W' - Z^T Z, [K | k] = -L^-T [Z | z] (into the G block and K to the record), s <- q~ + E_J^T sb + Vx^T fsb - Z^T z (host: four partial sums; device: the z lane of the S tiles)
    {
      auto s_item = [&](int it) {
        const int r = it >> 2, p = it & 3;
        constexpr int LF = NFS, LZ = (NUT + 3) / 4;
        double s = 0.0;
        if (p == 0) s = w.dx[r] + fact_ej_mix(w.sb[r], w.sb[fact_partner(r)], r, dt);
#pragma unroll
        for (int l = 0; l < LF; ++l) { const int ll = p * LF + l, lc = ll < NF ? ll : NF - 1; const double a = VA[lc][r], b = w.fsb[ll]; s += ll < NF ? a * b : 0.0; }
#pragma unroll
        for (int l = 0; l < LZ; ++l) { const int ll = p * LZ + l, lc = ll < NUT ? ll : NUT - 1; const double a = w.Zs[lc][r], b = w.zv[lc]; s -= ll < NUT ? a * b : 0.0; }
        w.part[it] = s;
      };
      const XtyJob jk = xty_also_to(xty_job(NUT, NXE + 1, NUT, &w.Ef[0][EF_MI], LDF, &w.Zs[0][0], LDZ, &w.PG[0][0], LDG, nullptr, 0, -1.0), rk + RIC_K, NX, NXE);
#if defined(__HIP_DEVICE_COMPILE__)
      constexpr int NSZ = (NUT + 3) / 4;
      PH_LAP0(ctx, HSQP_LAP_WAVE);
      if (wv < 4) {
        // S tiles: six 23-deep steps, W' from LDS and Q~ from the registers loaded in Ph3
        auto run = [&](auto ntc) {
          constexpr int NT = decltype(ntc)::value;
          const double* lz = &w.Zs[0][0];
          const double* lsa = &w.SA[0][0];
          double* ls = &w.S[0][0];
          hsqp_d4 acc[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = hsqp_d4{0.0, 0.0, 0.0, 0.0};
          auto xf = [&](auto sc, int t) {
            constexpr int s = decltype(sc)::value;
            const double v = lz[FS_ZX(t) + 4 * s * LDZ];
            if constexpr (4 * s + 3 >= NUT) return 4 * s + kk < NUT ? v : 0.0; else return v;
          };
          auto yf = [&](auto sc, int t) { constexpr int s = decltype(sc)::value; return lz[FS_ZY(t) + 4 * s * LDZ]; };
          double wp[NT][4];
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) wp[t][r] = lsa[FS_W(t) + (16 * fact_sym_tr(sfirst + t) + kk) * (LDA - NX) + 4 * LDA * r];   // (SA's rows are LDA apart, S's NX)
          PH_LAP(ctx, HSQP_LAP_WAVE, 21);
          fact_mfma<NT, FACT_PF, NSZ>(acc, xf, yf);
          // (the z lane's addresses are formed HERE: left to the compiler they are formed in front of the stage loop, one register per element — a spill)
          int fm[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) { fm[t] = FS_M(t); asm volatile("" : "+v"(fm[t])); }
          PH_LAP(ctx, HSQP_LAP_WAVE, 22);
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double v = (wp[t][r] + qpre[t][r]) - acc[t][r];
              // tiles above the diagonal: stored and mirrored; diagonal tiles: the elements on / above the diagonal, each also at its mirrored
              // place, so that S is symmetric to the bit (which elements: the mask formed in front of the stage loop)
              if ((FS_MASK(t) >> r) & 1) { ls[FS_W(t) + 4 * NX * r] = v; ls[fm[t] + 4 * r] = v; }
            }
          // the new s = (A~^T sb + q~) - Z^T z from the lane whose column is z (tiles of the last column: every row tile has one): the SAME expression as the
          // matrix elements' — A~^T sb came in through the W' fetch (column NX of SA), q~ through the Q~ fetch; it travels in the first partial-sum slot
#pragma unroll
          for (int t = 0; t < NT; ++t)
            if (fact_sym_tc(sfirst + t) == 3 && (FS_MASK(t) >> 4)) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if ((FS_MASK(t) >> (4 + r)) & 1) w.part[4 * (fm[t] + 4 * r)] = (wp[t][r] + qpre[t][r]) - acc[t][r];
            }
        };
        if (scount == 3) run(std::integral_constant<int, 3>{});
        else run(std::integral_constant<int, 2>{});
        PH_LAP(ctx, HSQP_LAP_WAVE, 23);
      } else {
        // waves 4 .. 7: [K | k] = -L^-T [Z | z], two of the eight tiles each, one call (the vector items of the new s are gone: it leaves through the S tiles)
        PH_LAP(ctx, HSQP_LAP_WAVE, 20);
        {
          const double* lz = &w.Zs[0][0];
          const double* le = &w.Ef[0][0];
          double* lpg = &w.PG[0][0];
          hsqp_d4 acc[2] = {hsqp_d4{0.0, 0.0, 0.0, 0.0}, hsqp_d4{0.0, 0.0, 0.0, 0.0}};
          auto xf = [&](auto sc, int t) {
            constexpr int s = decltype(sc)::value;
            if constexpr (4 * s + 3 >= NUT) { const double v = le[FK_X(t) + (4 * s + kk < NUT ? 4 * s : 4 * s - 1) * LDF]; return 4 * s + kk < NUT ? v : 0.0; }
            else return le[FK_X(t) + 4 * s * LDF];
          };
          auto yf = [&](auto sc, int) { constexpr int s = decltype(sc)::value; return lz[FK_Y + 4 * s * LDZ]; };
          fact_mfma<2, FACT_PF, NSZ>(acc, xf, yf);
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double v = -acc[t][r];
              if ((FK_MASK(t) >> r) & 1) lpg[FK_PG(t) + 4 * LDG * r] = v;                       // (k in column NX: picked up by the next stage's Ph1)
              if ((FK_MASK(t) >> (4 + r)) & 1) ((hsqp_gptr)rk)[FK_RK(t) + 4 * NX * r] = v;
            }
        }
        PH_LAP(ctx, HSQP_LAP_WAVE, 23);
      }
#else
      WG_FOR(ctx, it, 4 * NX) s_item(it);
      // (host: W' sits in S's upper triangle; Q~ is added and Z^T Z subtracted element by element, then mirrored)
      WG_FOR(ctx, it, NX * NX) {
        const int r = it / NX, c = it % NX;
        if (r > c) continue;
        double sacc = 0.0;
        for (int l = 0; l < NUT; ++l) sacc += w.Zs[l][r] * w.Zs[l][c];
        w.SA[r][c] = (w.S[r][c] + q[QP_Q + r * NX + c]) - sacc;
      }
      WG_SYNC(ctx);
      WG_FOR(ctx, it, NX * NX) { const int r = it / NX, c = it % NX; w.S[r][c] = r <= c ? w.SA[r][c] : w.SA[c][r]; }
      wg_xty_jobs(ctx, &jk, 1);
#endif
    }
    PH_ARRIVE(ctx, 2);
    WG_SYNC(ctx);
    PH_TICK(ctx, 5);
    if (vf) {
      WG_FOR(ctx, i, VF_SIZE) {
        const double* sp = &w.part[4 * (i >= NX * NX ? i - NX * NX : 0)];
        vf[(size_t)k * VF_SIZE + i] = i < NX * NX ? w.S[i / NX][i % NX] : (sp[0] + sp[1]) + (sp[2] + sp[3]);
      }
    }
    PH_TICK(ctx, 6);
  }
  WG_FOR(ctx, r, NUT) { const double kr = w.PG[r][FG_GV]; w.kv[r] = kr; ric[RIC_KV + r] = kr; }   // k of stage 0
  WG_SYNC(ctx);
}

// Forward sweep on the factors: ut = k + K dx;  f = Vx dx + Vu ut  (35 numbers: the base rows' share and qdd_j);  dx+ = E_J dx + F f + b~.
// Two barriers per stage as riccati_forward; a stage reads 35 x 81 + 23 x 58 numbers instead of 58 x 81 + 23 x 58.
// fj_out (optional, [N][NJ]): f of the joint rows = rows 12 .. 34 of Px dx + Pu ut, for step_node (which then reads the twelve wrench rows of Px / Pu only).
HSQP_HD void riccati_forward_fact(const Ctx& ctx, RicFWS& w, const double* x_init, const double* x, const double* qp, const double* dts, const double* ric, int N,
                                  double* dx_out, double* ut_out = nullptr, double* fj_out = nullptr) {
  WG_FOR(ctx, i, NX) {
    const double d = x_init[i] - x[i];
    w.dx[i] = d;
    dx_out[i] = d;
  }
  WG_SYNC(ctx);
  constexpr int NC = (NX + 3) / 4, NCB = (NUT + 3) / 4;
  constexpr int NR1 = 4 * (NF + NUT);            // rows of [Vx | Vu] (NF), then rows of K (NUT), four partial sums each
#if defined(__HIP_DEVICE_COMPILE__)
  {
    // Waves 0 .. 3 carry the sweep, the others only meet them at the barriers.  Item it < NR1 owns row it >> 2 of [Vx; K] (columns p + 4c) and, if
    // it < 4 NF, row it >> 2 of Vu as well.  A K row's first lane adds k and writes ut; after the second quad sum every lane of a factor row's quad
    // holds f: lane p = 0 writes the row's first state (base row, or q_j), lane p = 1 the second one of a joint row (v_j); each carries the b~ of the
    // state it writes.  The slices travel PF stages ahead in registers.
    // EVERY load of a slice is unconditional (lanes without a row, columns beyond a row: a valid address, and the OTHER operand of the product is
    // zeroed; the fetch index is clamped to the last stage instead of a branch around the fetch; the groups of PF stages run without a test of k):
    // with a branch anywhere around a load the compiler cannot count the loads in flight and waits for ALL of them (vmcnt(0)) before the first use —
    // the youngest was issued one stage earlier, so every third stage paid a memory round trip (4.1 k cycles per stage against 3.3 k now).
    // What is left is memory: a stage reads 33 KB; at 256 instances the sweep moves 850 MB in 139 us (6.1 TB/s: the HBM roofline), and a single CU
    // does not stream faster than ~10.6 B per cycle whatever the request pattern (loader waves writing coalesced lines to LDS slots were tried:
    // 3.2 k cycles per stage at 32 instances, 3.9 k at 256; without their loads the sweep takes 1.8 k — profiles/r06_ric_experiments.txt).
    constexpr int PF = 3;
    const int it = ctx.tid, row = it >> 2, p = it & 3;
    const bool rowV = it < 4 * NF, rowK = it >= 4 * NF && it < NR1;
    const bool joint = rowV && row >= 12;
    const int out_state = !rowV ? 0 : (row < 12 ? fact_base_row(row) : (p == 0 ? row - 6 : row + 23));
    const bool writes = rowV && (p == 0 || (p == 1 && joint));
    const bool ksum = rowK && p == 0;
    // where the lane's slices start in stage 0's records, and how far the next stage's are
    const int krow = rowK ? row - NF : 0, vrow = rowV ? row : 0;
    const double* a0 = rowK ? ric + RIC_K + krow * NX : fact_va_row(qp, vrow);
    const double* b0 = fact_vb_row(qp, vrow);
    const double* s0 = ksum ? ric + RIC_KV + krow : qp + QP_BV + out_state;
    const size_t astep = rowK ? (size_t)RIC_SIZE : (size_t)QP_SIZE, sstep = ksum ? (size_t)RIC_SIZE : (size_t)QP_SIZE;
    const bool a_last_ok = p + 4 * (NC - 1) < NX, b_last_ok = p + 4 * (NCB - 1) < NUT;
    const int ca_last = a_last_ok ? p + 4 * (NC - 1) : p, cb_last = b_last_ok ? p + 4 * (NCB - 1) : p;   // (clamped: column p again)
    // the two states of dx a joint row's writer combines with f (lanes that write nothing read state 0)
    const int xa = joint ? (p == 0 ? row - 6 : row + 23) : 0, xb = joint && p == 0 ? row + 23 : 0;
    double a[PF][NC], bq[PF][NCB], sc[PF], dtk[PF];
    auto fetch = [&](int k, double* av, double* bv, double& s1, double& d1) {
      const double* ap = a0 + (size_t)k * astep;
      const double* bp = b0 + (size_t)k * QP_SIZE;
#pragma unroll
      for (int c = 0; c < NC - 1; ++c) av[c] = ap[p + 4 * c];
      av[NC - 1] = ap[ca_last];
#pragma unroll
      for (int c = 0; c < NCB - 1; ++c) bv[c] = bp[p + 4 * c];
      bv[NCB - 1] = bp[cb_last];
      s1 = s0[(size_t)k * sstep];
      d1 = dts[k];
    };
    auto stage = [&](int k, double* av, double* bv, double& scu, double& dtu) {
      const double* dcur = (k & 1) ? w.sv : w.dx;
      double xv[NC];
#pragma unroll
      for (int c = 0; c < NC - 1; ++c) xv[c] = dcur[p + 4 * c];
      { const double xl = dcur[ca_last]; xv[NC - 1] = a_last_ok ? xl : 0.0; }
      const double da = dcur[xa], db = dcur[xb];
      double s1 = 0.0;
#pragma unroll
      for (int c = 0; c < NC; ++c) s1 += av[c] * xv[c];
      double sk = s1;
      sk += quad_perm_f64<0xB1>(sk);
      sk += quad_perm_f64<0x4E>(sk);
      if (ksum) {          // row of K: ut_j = k_j + K_j dx
        const double ut = scu + sk;
        w.zv[krow] = ut;
        if (ut_out) ut_out[(size_t)k * NUT + krow] = ut;
      }
      WG_SYNC(ctx);
      double* dnxt = (k & 1) ? w.dx : w.sv;
      const double dt = dtu, hq = 0.5 * dt * dt;
      double uv[NCB];
#pragma unroll
      for (int c = 0; c < NCB - 1; ++c) uv[c] = w.zv[p + 4 * c];
      { const double ul = w.zv[cb_last]; uv[NCB - 1] = b_last_ok ? ul : 0.0; }
      double s = s1;
#pragma unroll
      for (int c = 0; c < NCB; ++c) s += bv[c] * uv[c];
      s += quad_perm_f64<0xB1>(s);
      s += quad_perm_f64<0x4E>(s);
      if (writes) {
        double v;
        if (!joint) v = s + scu;
        else if (p == 0) v = ((da + dt * db) + hq * s) + scu;
        else v = (da + dt * s) + scu;
        dnxt[out_state] = v;
        dx_out[(size_t)(k + 1) * NX + out_state] = v;
        if (fj_out && joint && p == 0) fj_out[(size_t)k * NJ + row - 12] = s;
      }
      fetch(k + PF < N ? k + PF : N - 1, av, bv, scu, dtu);
      WG_SYNC(ctx);
    };
    if (it < 256) {
#pragma unroll
      for (int u = 0; u < PF; ++u) fetch(u < N ? u : N - 1, a[u], bq[u], sc[u], dtk[u]);
      int k0 = 0;
      for (; k0 + PF <= N; k0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) stage(k0 + u, a[u], bq[u], sc[u], dtk[u]);
      }
      // (N % PF stages left: their slices sit in sets 0, 1)
      if (k0 < N) stage(k0, a[0], bq[0], sc[0], dtk[0]);
      if (k0 + 1 < N) stage(k0 + 1, a[1], bq[1], sc[1], dtk[1]);
    } else {
      for (int k = 0; k < N; ++k) { WG_SYNC(ctx); WG_SYNC(ctx); }
    }
    return;
  }
#else
  for (int k = 0; k < N; ++k) {
    const double* q = qp + (size_t)k * QP_SIZE;
    const double* rk = ric + (size_t)k * RIC_SIZE;
    const double dt = dts[k], hq = 0.5 * dt * dt;
    WG_FOR(ctx, j, NUT) {   // (the device's partial sums: columns p + 4c, ((0 + 1) + (2 + 3)))
      const double* kr = rk + RIC_K + j * NX;
      const double p0 = matvec_part<NX>(kr, w.dx, 0), p1 = matvec_part<NX>(kr, w.dx, 1), p2 = matvec_part<NX>(kr, w.dx, 2), p3 = matvec_part<NX>(kr, w.dx, 3);
      w.zv[j] = rk[RIC_KV + j] + ((p0 + p1) + (p2 + p3));
    }
    WG_SYNC(ctx);
    if (ut_out) WG_FOR(ctx, j, NUT) ut_out[(size_t)k * NUT + j] = w.zv[j];
    WG_FOR(ctx, kf, NF) {
      const double* va = fact_va_row(q, kf);
      const double* vb = fact_vb_row(q, kf);
      double pp[4];
      for (int p = 0; p < 4; ++p) pp[p] = matvec_part<NX>(va, w.dx, p) + matvec_part<NUT>(vb, w.zv, p);
      w.fsb[kf] = (pp[0] + pp[1]) + (pp[2] + pp[3]);
      if (fj_out && kf >= 12) fj_out[(size_t)k * NJ + kf - 12] = w.fsb[kf];
    }
    WG_SYNC(ctx);
    WG_FOR(ctx, i, NX) {
      double v;
      if (i < 6) v = w.fsb[i] + q[QP_BV + i];
      else if (i < NV) v = ((w.dx[i] + dt * w.dx[i + NV]) + hq * w.fsb[i + 6]) + q[QP_BV + i];
      else if (i < NV + 6) v = w.fsb[i - NV + 6] + q[QP_BV + i];
      else v = (w.dx[i] + dt * w.fsb[i - 23]) + q[QP_BV + i];
      w.sv[i] = v;
      dx_out[(size_t)(k + 1) * NX + i] = v;
    }
    WG_SYNC(ctx);
    WG_FOR(ctx, i, NX) w.dx[i] = w.sv[i];
    WG_SYNC(ctx);
  }
#endif
}

}  // namespace hsqp
