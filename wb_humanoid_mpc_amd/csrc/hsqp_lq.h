// LQ approximation of one intermediate shooting node: RK4 sensitivity discretisation of the flow
// map (upstream ocs2 SensitivityIntegrator, integratorType RK4 — task.info:92) plus the
// node's cost model and equality rows (hsqp_node.h).
//
// Structure that is exploited (and is exact, not an approximation): the flow map is
//   xdot = [ v ; a_b(q,v,u) ; qdd_j ],
// so of the 58 rows of every stage Jacobian only the 6 base-acceleration rows are non-trivial, and
// [A|B] = [I|0] + dt/6 (dk1 + 2 dk2 + 2 dk3 + dk4) is fully described by two 6 x 93 blocks
//   P6 = dt^2/6 (Ab1 + Ab2 + Ab3)            (base position rows 0..5)
//   V6 = dt/6   (Ab1 + 2 Ab2 + 2 Ab3 + Ab4)  (base velocity rows 29..34)
// where Ab_s = d a_b(x_s,u)/dz chains the stage Jacobians G_s; the joint rows are
//   q_j+ = q_j + dt v_j + dt^2/2 qdd_j ,  v_j+ = v_j + dt qdd_j   exactly.
#pragma once
#include "hsqp_node.h"

namespace hsqp {

// ---- per-node record written by the LQ kernel (doubles), read by the projection kernel
constexpr int REC_PV = 0;                         // [2][6][LDJ]  P6, V6
constexpr int REC_B = REC_PV + 2 * 6 * LDJ;       // [64]         defect b = Phi(x,u) - x_next
constexpr int REC_J = REC_B + 64;                 // [NRS][LDJ]   residual rows (x sqrt(dt))
constexpr int REC_RHO = REC_J + NRS * LDJ;        // [NRS]
constexpr int REC_D = REC_RHO + NRS;              // [LDJ]        Hessian diagonal (x dt)
constexpr int REC_GD = REC_D + LDJ;               // [LDJ]        gradient, diagonal part (x dt)
constexpr int REC_CDE = REC_GD + LDJ;             // [NE_MAX][LDJ] rows [C|D|e]
constexpr int CDE_ROWS = 16;                      //              rows the equality block is allocated for (its transposed form keeps 16 per column)
constexpr int REC_MISC = REC_CDE + CDE_ROWS * LDJ;  // [16] ne, cost (x dt), eq_sse (x dt), dyn_sse (x dt), contact flags (2), first equality row of each foot (2), [8] = NROWS, [9] = LAYOUT
constexpr int REC_NROWS = REC_MISC + 8;           //      residual rows in use (compact layout, hsqp_node.h); the rows up to the end of their 24-row pass are zero
constexpr int REC_LAYOUT = REC_MISC + 9;          //      0: REC_J [row][LDJ], REC_CDE [row][LDJ] (phase form, centroidal); 1: transposed, REC_J [column][NRS] with the row slots of
                                                  //      hsqp_lql.h (ROWQ_*), REC_CDE [column][CDE_ROWS] — written column by column by the limb lanes
constexpr int REC_FLOW = REC_MISC + 16;           // [64] xdot at (x,u)
constexpr int REC_GS = REC_FLOW + 64;             // [4][6][LDJ] stage Jacobians d a_b/dz (scratch of the LQ kernel; limb-lane form: transposed, [4][LDJ][6])
constexpr int REC_AS = REC_GS + 4 * 6 * LDJ;      // [4][6]      base accelerations of the RK4 stages (limb-lane form: from the model kernel to the chain kernel)
constexpr int REC_SIZE = REC_AS + 24;
static_assert(REC_J % 2 == 0 && REC_CDE % 2 == 0 && REC_GS % 2 == 0 && REC_SIZE % 2 == 0, "16-byte aligned pieces");

// Model constants: read from global memory through the vector L1 (every workgroup of a CU reads the same 9 KB), or,
// with -DHSQP_DM_LDS=1, from a per-workgroup LDS copy (costs 9 KB of LDS = one workgroup of occupancy per CU).
#ifndef HSQP_DM_LDS
#define HSQP_DM_LDS 0
#endif
struct NoDevModelCopy {};
template <bool D>
struct LqWST {
#if HSQP_DM_LDS
  DevModel dml;
#else
  NoDevModelCopy dml;
#endif
  StageWST<D> st;
  double blk[D ? 3 : 1][2][6][6];   // RK4 chain: the blocks G_s[:, v_b] and G_s[:, q_b] of stages 2..4 (everything else of the chain lives in registers)
  NodeWST<D> nw;
  double vs[4][NV];    // velocity part of the stage states
  double as[4][6];     // base accelerations of the stages
  double xnext[NX];
  double bvec[64];
};
using LqWS = LqWST<true>;

// set up the model evaluation inputs of RK4 stage s (0..3)
template <class LW>
HSQP_HD void rk4_stage_inputs(const Ctx& ctx, LW& w, int s, double dt) {
  const double c = (s == 0) ? 0.0 : (s == 3 ? dt : 0.5 * dt);
  WG_FOR(ctx, i, NV + NV + NJ + 12) {
    if (i < NV) {
      w.st.q[i] = w.nw.x[i] + (s == 0 ? 0.0 : c * w.vs[s - 1][i]);
    } else if (i < 2 * NV) {
      const int k = i - NV;
      double a = 0.0;
      if (s > 0) a = k < 6 ? w.as[s - 1][k] : w.nw.u[12 + k - 6];
      const double v = w.nw.x[NV + k] + c * a;
      w.st.v[k] = v;
      w.vs[s][k] = v;
    } else if (i < 2 * NV + NJ) {
      w.st.qddj[i - 2 * NV] = w.nw.u[12 + i - 2 * NV];
    } else {
      w.st.W[i - 2 * NV - NJ] = w.nw.u[i - 2 * NV - NJ];
    }
  }
  WG_SYNC(ctx);
}

// X (6 x 29 block starting at column c0 of G_s) times Vd_t = [Ab_t ; E_qdd], element (r, col)
HSQP_HD double times_vd(const double (*Gs)[LDJ], int c0, const double (*Abt)[LDJ], int r, int col) {
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) s += Gs[r][c0 + k] * Abt[k][col];
  if (col >= NX + 12) s += Gs[r][c0 + 6 + (col - NX - 12)];
  return s;
}

constexpr int GT_LD = 6;   // limb-lane form (hsqp_lql.h): the stage Jacobians travel TRANSPOSED, [stage][column][6] — a lane owns a column
// where entry (r, k) of the 6 x 6 block G_{sg+2}[:, v_b] (which = 0) / G_{sg+2}[:, q_b] (which = 1) sits in the transposed stage Jacobians (offset from REC_GS)
HSQP_HD constexpr int lq_chain_blk_offset(int sg, int which, int r, int k) { return ((sg + 1) * LDJ + (which == 0 ? NV : 0) + k) * GT_LD + r; }

// The RK4 sensitivity chain of ONE column of [A|B] (see lq_node): Ab_1 = G_1, Ab_s = direct(G_s) + c_s G_s[:, v_b] Ab_{s-1} + c_s c_{s-1} G_s[:, q_b] Ab_{s-2};
// P6 = dt^2/6 (Ab_1 + Ab_2 + Ab_3), V6 = dt/6 (Ab_1 + 2 Ab_2 + 2 Ab_3 + Ab_4) -> rec[REC_PV].  gs = rec + REC_GS: the stage Jacobians as the
// column phases wrote them; the own-column entries of a stage and its selection partners are fetched one stage ahead of their use.
// GT: the stage Jacobians are stored transposed, [stage][column][6] (written by the limb lanes of hsqp_lql.h), else [stage][6][LDJ].
// (lq_chain_column_pv: the column's twelve entries P6[r], V6[r] as values — k_project forms them itself when the chain is fused into it, hsqp_project.h)
// BG (with GT): the 6 x 6 blocks are read from the transposed stage Jacobians themselves (blk unused) — their addresses are the same in every lane, so a device build fetches them
// through the scalar cache and they never occupy a vector register or LDS (k_project); the six entries of a column are fetched as three 16-byte pieces.
template <bool GT = false, bool BG = false>
HSQP_HD void lq_chain_column_pv(const double (*blk)[2][6][6], const double* gs, int col, double dt, double* P6, double* V6) {
  static_assert(GT || !BG, "blocks from the record: transposed stage Jacobians only");
  constexpr int RS = GT ? 1 : LDJ, CS = GT ? 6 : 1, SS = 6 * LDJ;   // strides of a row, a column, a stage
  auto B = [&](int sg, int which, int r, int k) -> double { if constexpr (BG) return gs[lq_chain_blk_offset(sg, which, r, k)]; else return blk[sg][which][r][k]; };
  // the six entries of column i of stage st (element r at gs[st * SS + r * RS + i * CS])
  auto col6 = [&](int st, int i, double* o) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (GT) {   // contiguous, 16-byte aligned (REC_GS and 6 * i are even)
      const double2* p2 = reinterpret_cast<const double2*>(gs + st * SS + i * CS);
      const double2 q0 = p2[0], q1 = p2[1], q2 = p2[2];
      o[0] = q0.x; o[1] = q0.y; o[2] = q1.x; o[3] = q1.y; o[4] = q2.x; o[5] = q2.y;
      return;
    }
#endif
#pragma unroll
    for (int r = 0; r < 6; ++r) o[r] = gs[st * SS + r * RS + i * CS];
  };
  const bool vcol = col >= NV && col < NX, acol = col >= NX + 12 && col < NZ;
  const int j = col - NX - 12;
  const int iA = col < NZ ? col : 0, iB = vcol ? col - NV : (acol ? NV + 6 + j : iA), iC = acol ? 6 + j : iA;
  const double live = col < NZ ? 1.0 : 0.0;
  const double cs[3] = {0.5 * dt, 0.5 * dt, dt};
  double a2[6], a1[6], P[6], V[6];   // Ab_{s-2}, Ab_{s-1}
  double gA[6], gB[6], gC[6];
  col6(0, iA, a1);
#pragma unroll
  for (int r = 0; r < 6; ++r) { a1[r] = live * a1[r]; a2[r] = 0.0; P[r] = a1[r]; V[r] = a1[r]; }
  col6(1, iA, gA); col6(1, iB, gB); col6(1, iC, gC);
#pragma unroll
  for (int sg = 0; sg < 3; ++sg) {
    const double c = cs[sg], cprev = sg == 0 ? 0.0 : cs[sg - 1];
    const double wB = (vcol || acol) ? c : 0.0, wC = (acol && sg >= 1) ? c * cprev : 0.0;
    double an[6], nA[6], nB[6], nC[6];
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);   // keep the ILP scheduler from hoisting the block loads of all three stages to the top (1.3 KB of spills per lane)
#endif
    if (sg < 2) { col6(sg + 2, iA, nA); col6(sg + 2, iB, nB); col6(sg + 2, iC, nC); }
#pragma unroll 2
    for (int r = 0; r < 6; ++r) {   // (two rows at a time: fully unrolled, the scheduler fetches all 72 block entries first and spills)
      const double v = live * gA[r] + wB * gB[r] + wC * gC[r];
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) { s1 += B(sg, 0, r, k) * a1[k]; s2 += B(sg, 1, r, k) * a2[k]; }
      an[r] = v + c * (s1 + cprev * s2);
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      a2[r] = a1[r]; a1[r] = an[r];
      if (sg < 2) { P[r] += an[r]; V[r] += 2.0 * an[r]; gA[r] = nA[r]; gB[r] = nB[r]; gC[r] = nC[r]; } else V[r] += an[r];
    }
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    P6[r] = live * (dt * dt / 6.0 * P[r]);
    V6[r] = live * (dt / 6.0 * V[r]);
  }
}
template <bool GT = false>
HSQP_HD void lq_chain_column(const double (*blk)[2][6][6], const double* gs, int col, double dt, double* rec) {
  double P6[6], V6[6];
  lq_chain_column_pv<GT>(blk, gs, col, dt, P6, V6);
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    rec[REC_PV + r * LDJ + col] = P6[r];
    rec[REC_PV + (6 + r) * LDJ + col] = V6[r];
  }
}

// Full LQ data of node (x, u, x_next, par) -> record `rec` (global memory); misc[0..3] = {ne, dt*cost, dt*|eq|^2, dt*|b|^2}.
// DERIV = false: values only (performance index); rec is not touched and may be null.
// PRELOADED: w.nw.x, w.nw.u and w.xnext already hold the node's (x, u, x_next) — the fused step + value kernel writes the stepped
// values there — and x / u / xnext are not read.
template <bool DERIV, bool PRELOADED = false>
HSQP_HD void lq_node(const Ctx& ctx, const DevModel& dm_global, LqWST<DERIV>& w, const double* x, const double* u, const double* xnext,
                     const double* par, double dt, double* rec, double* misc) {
#if HSQP_DM_LDS
  {
    constexpr int nw = (int)(sizeof(DevModel) / sizeof(double));
    static_assert(sizeof(DevModel) % sizeof(double) == 0, "DevModel must be a whole number of doubles");
    const double* src = reinterpret_cast<const double*>(&dm_global);
    double* dst = reinterpret_cast<double*>(&w.dml);
    WG_FOR(ctx, i, nw) dst[i] = src[i];
  }
  const DevModel& dm = w.dml;
#else
  const DevModel& dm = dm_global;
#endif
  stage_topology(ctx, dm, w.st, /*sync*/ false);   // one phase with the node's inputs: the two global round trips overlap
  WG_FOR(ctx, i, NX + NU + NP + NX) {
    if (i < NX) { if (!PRELOADED) w.nw.x[i] = x[i]; }
    else if (i < NX + NU) { if (!PRELOADED) w.nw.u[i - NX] = u[i - NX]; }
    else if (i < NX + NU + NP) w.nw.par[i - NX - NU] = par[i - NX - NU];
    else if (!PRELOADED) w.xnext[i - NX - NU - NP] = xnext[i - NX - NU - NP];
  }
  WG_SYNC(ctx);
  for (int s = 0; s < 4; ++s) {
    PH_TICK(ctx, 1);
    rk4_stage_inputs(ctx, w, s, dt);
    PH_TICK(ctx, 27);
    // the stage Jacobians of stages 2..4 go straight to the record (they are only chained later); stage 1 stays in LDS
    // for the node terms and is copied out
    stage_eval<DERIV>(ctx, dm, w.st, (DERIV && s > 0) ? rec + REC_GS + s * 6 * LDJ : nullptr);
    PH_TICK(ctx, 2);
    WG_FOR(ctx, i, 6 + ((DERIV && s == 0) ? 6 * LDJ : 0)) {
      if (i < 6) w.as[s][i] = w.st.ab[i];
      else { const int r = (i - 6) / LDJ, c = (i - 6) % LDJ; rec[REC_GS + r * LDJ + c] = c < NZ ? w.st.G[r][c] : 0.0; }
    }
    WG_SYNC(ctx);
    if (s == 0) {
      PH_TICK(ctx, 3);
      node_values(ctx, dm, w.st, w.nw);
      PH_TICK(ctx, 4);
      node_scalars(ctx, dm, w.st, w.nw);
      PH_TICK(ctx, 5);
      if constexpr (DERIV) {
        WG_FOR(ctx, i, 64) {   // (no barrier needed before the next phase: record writes only)
          double f = 0.0;
          if (i < NV) f = w.nw.x[NV + i];
          else if (i < NV + 6) f = w.st.ab[i - NV];
          else if (i < NX) f = w.nw.u[12 + i - NV - 6];
          rec[REC_FLOW + i] = f;
        }
        node_derivatives(ctx, dm, w.st, w.nw, dt, rec + REC_J, rec + REC_CDE, rec + REC_RHO);
      }
      PH_TICK(ctx, 6);
    }
  }
  PH_TICK(ctx, 7);
  // ---- RK4 value: x_next = x + dt/6 (k1 + 2 k2 + 2 k3 + k4), defect, performance terms
  WG_FOR(ctx, i, 64) {
    double b = 0.0;
    if (i < NV) {
      b = w.nw.x[i] + dt / 6.0 * (w.vs[0][i] + 2.0 * w.vs[1][i] + 2.0 * w.vs[2][i] + w.vs[3][i]) - w.xnext[i];
    } else if (i < NX) {
      const int k = i - NV;
      const double a = k < 6 ? (w.as[0][k] + 2.0 * w.as[1][k] + 2.0 * w.as[2][k] + w.as[3][k]) / 6.0 : w.nw.u[12 + k - 6];
      b = w.nw.x[i] + dt * a - w.xnext[i];
    }
    if (DERIV) rec[REC_B + i] = b;
    w.bvec[i] = b;
  }
  WG_SYNC(ctx);
  WG_FOR(ctx, it, 1) {
    double dyn = 0.0, eq = 0.0;
    for (int i = 0; i < NX; ++i) dyn += w.bvec[i] * w.bvec[i];
    for (int r = 0; r < w.nw.ne; ++r) eq += w.nw.eqv[r] * w.nw.eqv[r];
    misc[0] = (double)w.nw.ne;
    misc[1] = dt * node_cost(w.nw);
    if (DERIV) { misc[8] = (double)w.nw.nrows; misc[9] = 0.0; }   // (DERIV: misc = rec + REC_MISC, 16 wide; [9]: row-major layout)
    misc[2] = dt * eq;
    misc[3] = (dt > 0.0 ? dt : 1.0) * dyn;   // an event interval (dt = 0, identity jump map) counts its defect unscaled
    // structure of the equality rows for the projection: a swing foot's zero-wrench rows are unit rows of D (and have C = 0)
    misc[4] = (double)w.nw.contact[0]; misc[5] = (double)w.nw.contact[1];
    misc[6] = (double)w.nw.eq_off[0]; misc[7] = (double)w.nw.eq_off[1];
  }
  if constexpr (DERIV) {
  // ---- d, gd to the record; the chain's shared blocks G_s[:, v_b], G_s[:, q_b] (s = 2..4) from the record (written by this
  //      workgroup, L2-resident) to LDS
  WG_FOR(ctx, i, 2 * LDJ + 3 * 72) {
    if (i < LDJ) rec[REC_D + i] = w.nw.d[i];
    else if (i < 2 * LDJ) rec[REC_GD + i - LDJ] = w.nw.gd[i - LDJ];
    else {
      const int e = i - 2 * LDJ, sg = e / 72, which = (e / 36) % 2, r = (e / 6) % 6, k = e % 6;
      w.blk[sg][which][r][k] = rec[REC_GS + ((sg + 1) * 6 + r) * LDJ + (which == 0 ? NV : 0) + k];
    }
  }
  // ---- chain the stage Jacobians:  Ab_s = d a_b(x_s, u) / dz = G_s dz_s/dz.  dz_s/dz only couples a COLUMN of Ab_s with the same
  // column of Ab_{s-1}, Ab_{s-2} (through the 6 x 6 blocks of G_s that multiply d v_b and d q_b) and with a few columns of G_s (the
  // selection structure), so one item per column runs the whole chain in registers and writes its column of P6, V6: one phase
  // (round 2: LDS copies of all four G_s, a direct-part phase and a matrix-core job per stage — eight barriers, 22 k cycles).
  WG_SYNC(ctx);
#ifndef HSQP_NO_CHAIN
  WG_FOR(ctx, col, LDJ) lq_chain_column(w.blk, rec + REC_GS, col, dt, rec);
#endif
  PH_TICK(ctx, 8);
  }
}

// element (row, col) of the record's residual rows / equality rows in either layout (REC_LAYOUT) — debug / parity paths and tests
inline double rec_J_at(const double* rec, int row, int col) { return rec[REC_LAYOUT] != 0.0 ? rec[REC_J + col * NRS + row] : rec[REC_J + row * LDJ + col]; }
inline double rec_CDe_at(const double* rec, int row, int col) { return rec[REC_LAYOUT] != 0.0 ? rec[REC_CDE + col * CDE_ROWS + row] : rec[REC_CDE + row * LDJ + col]; }

// Expand the structured record into the dense [A|B] (58 x 93) — used by the debug/parity path and by tests.
inline void expand_AB(const double* rec, double dt, double* AB) {
  for (int i = 0; i < NX * NZ; ++i) AB[i] = 0.0;
  for (int i = 0; i < NX; ++i) AB[i * NZ + i] = 1.0;
  for (int i = 0; i < NV; ++i) AB[i * NZ + NV + i] = dt;
  for (int j = 0; j < NJ; ++j) {
    AB[(6 + j) * NZ + NX + 12 + j] = 0.5 * dt * dt;
    AB[(NV + 6 + j) * NZ + NX + 12 + j] = dt;
  }
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < NZ; ++c) {
      AB[r * NZ + c] += rec[REC_PV + r * LDJ + c];
      AB[(NV + r) * NZ + c] += rec[REC_PV + 6 * LDJ + r * LDJ + c];
    }
}

}  // namespace hsqp
