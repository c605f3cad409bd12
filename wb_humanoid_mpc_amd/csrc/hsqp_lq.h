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
constexpr int REC_MISC = REC_CDE + NE_MAX * LDJ;  // [16] ne, cost (x dt), eq_sse (x dt), dyn_sse (x dt), contact flags (2), first equality row of each foot (2), [8] = NROWS
constexpr int REC_NROWS = REC_MISC + 8;           //      residual rows in use (compact layout, hsqp_node.h); the rows up to the end of their 24-row pass are zero
constexpr int REC_FLOW = REC_MISC + 16;           // [64] xdot at (x,u)
constexpr int REC_GS = REC_FLOW + 64;             // [4][6][LDJ] stage Jacobians d a_b/dz (scratch of the LQ kernel)
constexpr int REC_SIZE = REC_GS + 4 * 6 * LDJ;

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
  union {
    StageWST<D> st;
    struct {
      double Gs[D ? 3 : 1][D ? 6 : 1][LDJ];   // stage Jacobians of stages 2..4 (stage 1 is Ab[0])
      double Ab[D ? 4 : 1][D ? 6 : 1][LDJ];
    } ch;              // RK4 chain workspace: aliases the stage workspace, which is dead after stage 4
  };
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
    if (DERIV) misc[8] = (double)w.nw.nrows;   // (DERIV: misc = rec + REC_MISC, 16 wide)
    misc[2] = dt * eq;
    misc[3] = (dt > 0.0 ? dt : 1.0) * dyn;   // an event interval (dt = 0, identity jump map) counts its defect unscaled
    // structure of the equality rows for the projection: a swing foot's zero-wrench rows are unit rows of D (and have C = 0)
    misc[4] = (double)w.nw.contact[0]; misc[5] = (double)w.nw.contact[1];
    misc[6] = (double)w.nw.eq_off[0]; misc[7] = (double)w.nw.eq_off[1];
  }
  if constexpr (DERIV) {
  // ---- write d, gd (the equality rows went straight to the record)
  WG_FOR(ctx, i, 2 * LDJ) {
    if (i < LDJ) rec[REC_D + i] = w.nw.d[i];
    else rec[REC_GD + i - LDJ] = w.nw.gd[i - LDJ];
  }
  WG_SYNC(ctx);  // the stage workspace is dead from here on: Ab aliases it
  PH_TICK(ctx, 33);
  // ---- chain the stage Jacobians:  Ab_s = d a_b(x_s, u) / dz
  const double c2 = 0.5 * dt, c3 = 0.5 * dt, c4 = dt;
  {   // stage Jacobians back from the record (written by this workgroup, L2-resident), 9 loads in flight per item
    constexpr int ng = 4 * 6 * LDJ, nbg = nbatches(ng, 9);
    WG_FOR(ctx, b, nbg) {
      copy_batch<9>(b, ng, rec + REC_GS, [&](int i, double g) {
        if (i < 6 * LDJ) w.ch.Ab[0][i / LDJ][i % LDJ] = g;
        else w.ch.Gs[i / (6 * LDJ) - 1][(i / LDJ) % 6][i % LDJ] = g;
      });
    }
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 34);
  for (int s = 1; s < 4; ++s) {
    const double c = s == 1 ? c2 : (s == 2 ? c3 : c4);
    const double cprev = s == 2 ? c2 : c3;  // coefficient of stage s-1 (used for s >= 2)
    // direct part: G_s composed with the selection structure of d z_s / d z
    WG_FOR(ctx, i, 6 * LDJ) {
      const int r = i / LDJ, col = i % LDJ;
      const double* G = w.ch.Gs[s - 1][r];
      double val = 0.0;
      if (col < NZ) {
        val = G[col];
        if (col >= NV && col < NX) val += c * G[col - NV];
        if (col >= NX + 12) {
          val += c * G[NV + 6 + (col - NX - 12)];
          if (s >= 2) val += c * cprev * G[6 + (col - NX - 12)];
        }
      }
      w.ch.Ab[s][r][col] = val;
    }
    WG_SYNC(ctx);
    // chained part on the matrix cores: Ab_s += c G_s[:, v_b] Ab_{s-1} + c c_{s-1} G_s[:, q_b] Ab_{s-2}  (6x6 by 6x96 products;
    // X^T Y with X read through an element stride: X[l][r] = G_s[r][c0 + l])
    {
      XtyJob job = xty_job(6, LDJ, 6, &w.ch.Gs[s - 1][0][NV], 1, &w.ch.Ab[s - 1][0][0], LDJ, &w.ch.Ab[s][0][0], LDJ, &w.ch.Ab[s][0][0], LDJ, c);
      job.sx1 = LDJ;
      if (s >= 2) { job.L2 = 6; job.X2 = &w.ch.Gs[s - 1][0][0]; job.ldx2 = 1; job.sx2 = LDJ; job.Y2 = &w.ch.Ab[s - 2][0][0]; job.ldy2 = LDJ; job.sign2 = cprev; }
      wg_xty_jobs<true>(ctx, &job, 1);
    }
    WG_SYNC(ctx);
  }
  PH_TICK(ctx, 8);
  WG_FOR(ctx, i, 2 * 6 * LDJ) {
    const int which = i / (6 * LDJ), r = (i / LDJ) % 6, col = i % LDJ;
    double v;
    if (which == 0) v = dt * dt / 6.0 * (w.ch.Ab[0][r][col] + w.ch.Ab[1][r][col] + w.ch.Ab[2][r][col]);
    else v = dt / 6.0 * (w.ch.Ab[0][r][col] + 2.0 * w.ch.Ab[1][r][col] + 2.0 * w.ch.Ab[2][r][col] + w.ch.Ab[3][r][col]);
    rec[REC_PV + i] = v;
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 9);
  }
}

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
