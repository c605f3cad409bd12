// Stage cost / soft-constraint quadratic model and equality-constraint linearisation of one
// shooting node, evaluated from the stage-1 model workspace (hsqp_model.h).
//
// Terms and their order follow WBMpcInterface::setupOptimalControlProblem
// (humanoid_nmpc/humanoid_wb_mpc/src/WBMpcInterface.cpp:131-199); each block cites its source.
// The cost model is emitted in "diagonal + residual rows" form
//       H = diag(d) + J^T J ,     g = gd + J^T rho ,
// which is exact for every term of this problem: Gauss-Newton costs (J = sqrt(w) dr/dz), penalties on
// linear-order constraints (J = sqrt(p'') dh/dz, rho = p'/sqrt(p'')) and the friction cone's
// second-order term p' d2h (p' < 0, d2h negative semidefinite => three more rows).
// Row slots (NR = 64): [15 f + k] foot f task-space rows k = ori(3) vlin(3) vang(3) alin(3) aang(3);
// [30 + 4 f + k] friction cone; [38 + 4 f + k] contact moment XY; [46 + k] foot collision (16); 62,63 unused.
#pragma once
#include "hsqp_model.h"

namespace hsqp {

constexpr int NRS = 64;          // residual row slots
constexpr int ROW_FOOT = 0, ROW_FRIC = 30, ROW_MXY = 38, ROW_COLL = 46;

template <bool D>
struct NodeWST {
  static constexpr bool DERIV = D;
  double x[NX], u[NU], par[NP];
  double xnom[NX], unom[NU];
  int contact[2];
  int eq_off[2];                 // first equality row of each foot
  int ne;
  // compact residual rows of the record: the 30 foot rows, then 8 rows (friction cone, contact moment) per foot in contact, then
  // the 16 collision rows if one of them is active; nrows of them, nrows_pad = the end of the last 24-row pass that holds one
  int row_fm[2], row_coll, nrows, nrows_pad;   // first record row of a foot's friction / moment rows, of the collision rows (-1: absent)
  // values
  double fkv[2][18];             // pos, ori, vlin, vang, alin, aang per foot
  double hfric[2], hmxy[2][4], hcoll[16];
  double fshift[2];              // friction cone: -p' * hessianDiagonalShift of a foot in contact (0 otherwise)
  double scale[D ? NRS : 1];     // sqrt(p'') (or sqrt(w)*ip) per row slot, 0 if the slot is inactive (derivative rows only)
  double rho[NRS];
  double eqv[NE_MAX];
  union {
    struct { double terms[208], tsum[16]; };          // stage-cost terms and their partial sums (node_scalars) ...
    struct { double d[D ? LDJ : 1], gd[D ? LDJ : 1]; };   // ... then the Hessian / gradient diagonals (node_derivatives; tsum stays live)
  };
  double cost;
};
using NodeWS = NodeWST<true>;
static_assert(2 * LDJ <= 208, "d / gd may not reach the partial sums of the cost");

// orientation error wrt the ground plane (oracle ASSUMPTION A2): e = (n x a)/sqrt(2(1+a.n)), a = R e_z, n = e_z
HSQP_HD void ori_error(const double* R, double* e) {
  const double a[3] = {R[2], R[5], R[8]};
  const double s = sqrt(2.0 * (1.0 + a[2]));
  e[0] = -a[1] / s; e[1] = a[0] / s; e[2] = 0.0;
}
HSQP_HD void ori_error_d(const double* R, const double* da, double* de) {  // differential for a -> a + da
  const double a[3] = {R[2], R[5], R[8]};
  const double s = sqrt(2.0 * (1.0 + a[2]));
  const double ds = da[2] / s;
  de[0] = -da[1] / s + a[1] * ds / (s * s);
  de[1] = da[0] / s - a[0] * ds / (s * s);
  de[2] = 0.0;
}

// does revolute coordinate jc move body b?  (sub: the subtree sizes in the stage workspace — an LDS read in front of the item's LDS operands, not a
// lane-indexed load from the model image in global memory)
template <class SW>
HSQP_HD bool supports(const SW& ws, int jc, int b) {
  if (jc < 3) return true;
  const int bi = jc - 2;
  return b >= bi && b < bi + (int)ws.sub[bi];
}

// pairs of collision points per constraint row (FootCollisionConstraint.cpp:118-141); point ids follow DevModel::coll_body:
// 0 ankle_l, 1 ankle_r, 2 f_l, 3 f_r, 4 l1, 5 r1, 6 l2, 7 r2, 8 k_l, 9 k_r
HSQP_HD void coll_pair(int row, int& a, int& b) {
  // (packed into two words: a table indexed by a run-time row would live in scratch memory)
  constexpr unsigned long long A = 0x7536428233226644ull, B = 0x0001119364757575ull;
  a = (int)((A >> (4 * row)) & 0xfull); b = (int)((B >> (4 * row)) & 0xfull);
}
// collision point p relative to O
template <class SW>
HSQP_HD void coll_point(const DevModel& dm, const SW& ws, int p, double* out) {
  const int b = dm.coll_body[p];
  double t[3];
  m3_mulv(ws.R[b], dm.coll_p[p], t);
  for (int k = 0; k < 3; ++k) out[k] = ws.r[b][k] + t[k];
}

// Values of everything that does not need derivatives — ONE phase: nominal state / input, contact flags and row layout, foot frame
// quantities, constraint values.  Requires stage_eval() results in ws for (x,u).  (Round 2 had these in two phases with a single
// item looping over the 93 nominal entries.)
template <class SW, class NW>
HSQP_HD void node_values(const Ctx& ctx, const DevModel& dm, const SW& ws, NW& nw) {
  constexpr int I_FLAGS = NZ, I_FEET = NZ + 1, I_COLL = I_FEET + 2, I_FRIC = I_COLL + 16, I_MXY = I_FRIC + 2, I_END = I_MXY + 8;
  static_assert(I_END <= 128, "one round of a two-wave workgroup");
  WG_FOR(ctx, it, I_END) {
    const int c0 = nw.par[HSQP_P_CONTACT] > 0.5, c1 = nw.par[HSQP_P_CONTACT + 1] > 0.5;
    if (it < NX) {
      // StateInputQuadraticCost::getStateInputDeviation (humanoid_common_mpc/src/cost/StateInputQuadraticCost.cpp:67-78)
      double xn = nw.par[HSQP_P_XDES + it];
      const int j = it - 6;
      const bool a0 = j == dm.arm_swing_joint[0], a1 = j == dm.arm_swing_joint[1], a2 = j == dm.arm_swing_joint[2], a3 = j == dm.arm_swing_joint[3];
      if (a0 || a1 || a2 || a3) {
        const double yaw = nw.x[3];
        const double vloc = cos(yaw) * nw.par[HSQP_P_XDES + NV] + sin(yaw) * nw.par[HSQP_P_XDES + NV + 1];
        const double gcf = nw.par[HSQP_P_ARMSWING] * vloc;  // SwitchedModelReferenceManager.cpp:110-135
        // (a joint listed twice collects both terms, in list order, as the serial form did)
        if (a0) xn += -0.15 * gcf;
        if (a1) xn += 0.15 * gcf;
        if (a2) xn += -0.15 * gcf;
        if (a3) xn += 0.15 * gcf;
      }
      nw.xnom[it] = xn;
    } else if (it < NZ) {   // weightCompensatingInput (DynamicsHelperFunctions.h:178-193), 9.81 hard-coded there
      const int i = it - NX;
      double un = 0.0;
      if ((i == 2 && c0) || (i == 8 && c1)) un = dm.total_mass * 9.81 / (c0 + c1);
      nw.unom[i] = un;
    } else if (it == I_FLAGS) {
      nw.contact[0] = c0; nw.contact[1] = c1;
      nw.eq_off[0] = 0;
      nw.eq_off[1] = c0 ? 6 : 7;
      nw.ne = nw.eq_off[1] + (c1 ? 6 : 7);
      nw.row_fm[0] = c0 ? ROW_FRIC : -1;
      nw.row_fm[1] = c1 ? ROW_FRIC + 8 * c0 : -1;
    } else if (it < I_COLL) {   // foot frames: value quantities
      const int f = it - I_FEET, b = dm.contact_body[f], jc = b + 2;
      const double* rP = ws.rP[f];
      const double* vl = ws.vl[jc];
      // full spatial acceleration of the body: trick acceleration (vd_base = 0) + {E a_ang, a_lin} - gravity
      double a[6];
      for (int k = 0; k < 3; ++k) { a[k] = ws.al[jc][k] + ws.y[k]; a[3 + k] = ws.al[jc][3 + k] + ws.ab[k]; }
      a[5] -= dm.gravity;
      double* o = nw.fkv[f];
      for (int k = 0; k < 3; ++k) o[k] = nw.x[k] + rP[k];
      ori_error(ws.R[b], o + 3);
      double t[3], t2[3];
      v3_cross(vl, rP, t);
      for (int k = 0; k < 3; ++k) { o[6 + k] = vl[3 + k] + t[k]; o[9 + k] = vl[k]; }
      v3_cross(a, rP, t);
      v3_cross(vl, o + 6, t2);
      for (int k = 0; k < 3; ++k) { o[12 + k] = a[3 + k] + t[k] + t2[k]; o[15 + k] = a[k]; }
    } else if (it < I_FRIC) {   // foot collision distances (FootCollisionConstraint.cpp:118-141)
      const int r = it - I_COLL;
      int a, b;
      coll_pair(r, a, b);
      double pa[3], pb[3], dd[3];
      coll_point(dm, ws, a, pa);
      coll_point(dm, ws, b, pb);
      for (int k = 0; k < 3; ++k) dd[k] = pa[k] - pb[k];
      nw.hcoll[r] = sqrt(v3_dot(dd, dd)) - 2.0 * (r == 9 ? dm.r_knee : dm.r_foot);
    } else if (it < I_MXY) {   // friction cone value (FrictionForceConeConstraint.cpp:180-185)
      const int f = it - I_FRIC;
      const double Fx = nw.u[6 * f], Fy = nw.u[6 * f + 1], Fz = nw.u[6 * f + 2];
      nw.hfric[f] = dm.friction_mu * (Fz + dm.friction_grip) - sqrt(Fx * Fx + Fy * Fy + dm.friction_reg);
    } else {   // contact moment XY (ContactMomentXYConstraintCppAd.cpp:87-104); the contact frame has the rotation of its body
      const int f = (it - I_MXY) / 4, r = (it - I_MXY) % 4;
      const double* Rf = ws.R[dm.contact_body[f]];
      double lf[3], lm[3];
      m3_tmulv(Rf, nw.u + 6 * f, lf);
      m3_tmulv(Rf, nw.u + 6 * f + 3, lm);
      double h;
      if (r == 0) h = lm[0] - dm.rect_y_min * lf[2];
      else if (r == 1) h = -lm[0] + dm.rect_y_max * lf[2];
      else if (r == 2) h = -lm[1] - dm.rect_x_min * lf[2];
      else h = lm[1] + dm.rect_x_max * lf[2];
      nw.hmxy[f][r] = h;
    }
  }
  WG_SYNC(ctx);
}

// Penalties, row scalings, equality values and the stage-cost terms in ONE phase (every item forms its penalty once and derives
// the row scaling, rho and the cost term from it), then the partial sums of the cost; the final sum is taken by the caller
// (node_cost) when it writes the node's performance terms.  After node_values().
template <class SW, class NW>
HSQP_HD void node_scalars(const Ctx& ctx, const DevModel& dm, const SW& ws, NW& nw) {
  (void)ws;
  constexpr int T_FOOT = NZ, T_FRIC = NZ + ROW_FRIC, T_MXY = T_FRIC + 2, T_JL = T_MXY + 8, T_COLL = T_JL + 2 * NJ, T_END = T_COLL + 16;
  static_assert(T_END <= 208, "cost-term slots");
  auto set_row = [&](int s, double sc, double rho) { if (NW::DERIV) nw.scale[NW::DERIV ? s : 0] = sc; nw.rho[s] = rho; };
  WG_FOR(ctx, t, 208 + NE_MAX) {
    double c = 0.0;
    if (t < NX) { const double dxx = nw.x[t] - nw.xnom[t]; c = 0.5 * dm.Q[t] * dxx * dxx; }
    else if (t < NZ) { const int i = t - NX; const double duu = nw.u[i] - nw.unom[i]; c = 0.5 * dm.R[i] * duu * duu; }
    else if (t < T_FRIC) {   // EndEffectorDynamicsFootCost.cpp:91-124: r = err .* sqrtW * impactProximity
      const int s = t - T_FOOT, f = s / 15, k = s % 15;
      const double sc = dm.foot_sqrt_w[3 + k] * nw.par[HSQP_P_IMPACT + f];
      const double rho = sc * nw.fkv[f][3 + k];
      set_row(s, sc, rho);
      c = 0.5 * rho * rho;
    } else if (t < T_MXY) {   // friction cone of foot f: its four row slots (first-order row + the three rows of p' d2h), the diagonal shift
      const int f = t - T_FRIC;
      double sc[4] = {0.0, 0.0, 0.0, 0.0}, rho0 = 0.0, shift = 0.0;
      if (nw.contact[f]) {
        const Pen3 p = relaxed_barrier(dm.friction_bmu, dm.friction_bdelta, nw.hfric[f]);
        const double Fx = nw.u[6 * f], Fy = nw.u[6 * f + 1];
        const double T2 = Fx * Fx + Fy * Fy + dm.friction_reg, T3 = T2 * sqrt(T2);
        sc[0] = sqrt(p.d2); rho0 = p.d1 / sc[0];
        sc[1] = sc[2] = sqrt(-p.d1 * dm.friction_reg / T3);
        sc[3] = sqrt(-p.d1 / T3);
        shift = -p.d1 * dm.friction_hess_shift;   // hessianDiagonalShift (FrictionForceConeConstraint.cpp:213-224)
        c = p.p;
      }
      for (int k = 0; k < 4; ++k) set_row(ROW_FRIC + 4 * f + k, sc[k], k == 0 ? rho0 : 0.0);
      nw.fshift[f] = shift;
    } else if (t < T_JL) {
      const int k = t - T_MXY, f = k / 4;
      double sc = 0.0, rho = 0.0;
      if (nw.contact[f]) {
        const Pen3 p = relaxed_barrier(dm.moment_bmu, dm.moment_bdelta, nw.hmxy[f][k % 4]);
        sc = sqrt(p.d2); rho = p.d1 / sc;
        c = p.p;
      }
      set_row(ROW_MXY + k, sc, rho);
    } else if (t < T_COLL) {   // JointLimitsSoftConstraint.cpp:64-100
      const int k = t - T_JL, j = k >> 1;
      c = (k & 1) ? pwp_barrier(dm.jl_bmu, dm.jl_bdelta, dm.q_hi[j] - nw.x[6 + j]).p : pwp_barrier(dm.jl_bmu, dm.jl_bdelta, nw.x[6 + j] - dm.q_lo[j]).p;
    } else if (t < T_END) {
      const int r = t - T_COLL;
      double sc = 0.0, rho = 0.0;
      if (!(nw.contact[0] && nw.contact[1])) {
        const Pen3 p = pwp_barrier(dm.coll_bmu, dm.coll_bdelta, nw.hcoll[r]);
        if (p.d2 > 0.0) { sc = sqrt(p.d2); rho = p.d1 / sc; }
        c = p.p;
      }
      set_row(ROW_COLL + r, sc, rho);
    } else if (t >= 208) {   // equality values
      const int r = t - 208;
      if (r < nw.ne) {
        const int f = r >= nw.eq_off[1] ? 1 : 0, k = r - nw.eq_off[f];
        const double* o = nw.fkv[f];
        double e;
        if (nw.contact[f]) {  // EndEffectorDynamicsAccelerationsConstraint.cpp:82-103, gains WBMpcInterface.cpp:205-229
          const int cc = k % 3;
          const double gp = k < 2 ? 0.0 : (k == 2 ? dm.gain_pos_z : dm.gain_ori);
          const double gv = k < 2 ? dm.gain_linvel_xy : (k == 2 ? dm.gain_linvel_z : dm.gain_angvel);
          const double ga = k < 2 ? dm.gain_linacc_xy : (k == 2 ? dm.gain_linacc_z : dm.gain_angacc);
          e = k < 3 ? gp * o[cc] + gv * o[6 + cc] + ga * o[12 + cc] : gp * o[3 + cc] + gv * o[9 + cc] + ga * o[15 + cc];
        } else if (k < 6) {   // ZeroWrenchConstraint.cpp:59-84
          e = nw.u[6 * f + k];
        } else {              // EndEffectorDynamicsLinearAccConstraint.cpp:69-83, config WBMpcPreComputation.cpp:91-104
          const double* sw = nw.par + HSQP_P_SWING + 3 * f;
          e = -dm.gain_linvel_z * sw[1] - dm.gain_linacc_z * sw[2] - dm.gain_pos_z * sw[0] + dm.gain_pos_z * o[2] +
              dm.gain_linvel_z * o[8] + dm.gain_linacc_z * o[14];
        }
        nw.eqv[r] = e;
      }
      continue;
    }
    nw.terms[t] = c;
  }
  WG_SYNC(ctx);
  WG_FOR(ctx, p, 17) {
    if (p == 16) {   // are the collision rows needed at all?  (they are inactive unless two collision points come within delta)
      bool any = false;
      if (NW::DERIV) for (int r = 0; r < 16; ++r) any = any || nw.scale[NW::DERIV ? ROW_COLL + r : 0] != 0.0;
      const int base = ROW_FRIC + 8 * (nw.contact[0] + nw.contact[1]);
      nw.row_coll = any ? base : -1;
      nw.nrows = base + (any ? 16 : 0);
      nw.nrows_pad = nw.nrows <= 24 ? 24 : (nw.nrows <= 48 ? 48 : NRS);
      continue;
    }
    double c = 0.0;
#pragma unroll
    for (int k = 0; k < 13; ++k) c += nw.terms[p * 13 + k];
    nw.tsum[p] = c;
  }
  WG_SYNC(ctx);
}

// the stage cost: sum of the 16 partial sums (any one item may call it)
template <class NW>
HSQP_HD double node_cost(const NW& nw) {
  double c = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) c += nw.tsum[k];
  return c;
}

// partial derivatives of one foot frame's kinematic quantities w.r.t. ONE column of z = [x;u]
// out[18] = d{pos, ori, vlin, vang, alin, aang}/dz_col
template <class SW>
HSQP_HD void foot_column(const DevModel& dm, const SW& ws, const NodeWS& nw, int f, int col, double* out) {
  const int b = dm.contact_body[f], jb = b + 2;
  const double* rP = ws.rP[f];
  const double* vi = ws.vl[jb];
  const double* o = nw.fkv[f];
  double dv[6] = {0, 0, 0, 0, 0, 0}, da[6] = {0, 0, 0, 0, 0, 0}, drP[3] = {0, 0, 0}, dpos_extra[3] = {0, 0, 0};
  bool rot = false;  // whether the foot frame rotates with this column (orientation partial)
  double wax[3] = {0, 0, 0};
  if (col < 3) {
    dpos_extra[col] = 1.0;
  } else if (col < NV) {
    const int jc = col - 3;
    if (supports(ws, jc, b)) {
      const double* Sx = ws.S[jc];
      double dvv[6], daa[6], t[6], pc[3];
      for (int k = 0; k < 6; ++k) dvv[k] = vi[k] - ws.vl[jc][k];
      for (int k = 0; k < 6; ++k) daa[k] = ws.al[jb][k] - ws.al[jc][k];
      if (jc < 2)  // euler links z, y do not carry the later euler accelerations: add {sum_{e>jc} w_e a_e, 0}
        for (int e = jc + 1; e < 3; ++e)
          for (int k = 0; k < 3; ++k) daa[k] += ws.E[3 * k + e] * ws.ab[3 + e];
      mxm(Sx, dvv, dv);
      mxm(Sx, daa, da);
      mxm(ws.Sd[jc], dvv, t);
      for (int k = 0; k < 6; ++k) da[k] += t[k];
      for (int k = 0; k < 3; ++k) pc[k] = rP[k] - (jc < 3 ? 0.0 : ws.r[jc - 2][k]);
      v3_cross(Sx, pc, drP);
      rot = true;
      for (int k = 0; k < 3; ++k) wax[k] = Sx[k];
    }
  } else if (col < NX) {
    const int c = col - NV;
    if (c < 3) {
      dv[3 + c] = 1.0;  // prismatic S = {0, e}; classical acceleration does not depend on it
      // spatial-acceleration partial -v_i x S cancels in the classical acceleration; keep da consistent:
      double Sx[6] = {0, 0, 0, 0, 0, 0}, t[6];
      Sx[3 + c] = 1.0;
      mxm(vi, Sx, t);
      for (int k = 0; k < 6; ++k) da[k] = -t[k];
    } else {
      const int jc = c - 3;
      if (supports(ws, jc, b)) {
        const double* Sx = ws.S[jc];
        double t[6];
        for (int k = 0; k < 6; ++k) dv[k] = Sx[k];
        mxm(vi, Sx, t);
        for (int k = 0; k < 6; ++k) da[k] = 2.0 * ws.Sd[jc][k] - t[k];
      }
    }
  } else if (col >= NX + 12) {
    const int jc = 3 + (col - NX - 12);
    if (supports(ws, jc, b))
      for (int k = 0; k < 6; ++k) da[k] = ws.S[jc][k];
  }
  // chain through the base acceleration a_b(z): d alpha += E G[3:6], d aO += G[0:3]  (S_e pass through O)
  {
    double gy[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k)
      for (int e = 0; e < 3; ++e) gy[k] += ws.E[3 * k + e] * ws.G[3 + e][col];
    for (int k = 0; k < 3; ++k) { da[k] += gy[k]; da[3 + k] += ws.G[k][col]; }
  }
  // frame level
  const double* om = vi;           // omega_i
  const double* al = o + 15;       // full angular acceleration
  const double* vP = o + 6;
  double t1[3], t2[3], t3[3], t4[3], dvP[3];
  for (int k = 0; k < 3; ++k) out[k] = drP[k] + dpos_extra[k];
  if (rot) {
    const double* Rf = ws.R[b];   // the contact frame has the rotation of its body
    const double a[3] = {Rf[2], Rf[5], Rf[8]};
    double dvec[3];
    v3_cross(wax, a, dvec);
    ori_error_d(Rf, dvec, out + 3);
  } else {
    out[3] = out[4] = out[5] = 0.0;
  }
  v3_cross(dv, rP, t1);
  v3_cross(om, drP, t2);
  for (int k = 0; k < 3; ++k) { dvP[k] = dv[3 + k] + t1[k] + t2[k]; out[6 + k] = dvP[k]; out[9 + k] = dv[k]; }
  v3_cross(da, rP, t1);
  v3_cross(al, drP, t2);
  v3_cross(dv, vP, t3);
  v3_cross(om, dvP, t4);
  for (int k = 0; k < 3; ++k) { out[12 + k] = da[3 + k] + t1[k] + t2[k] + t3[k] + t4[k]; out[15 + k] = da[k]; }
}

// derivative of a body-fixed point position w.r.t. generalized coordinate c (0..28); returns false if zero
template <class SW>
HSQP_HD bool point_column(const DevModel& dm, const SW& ws, int body, const double* rpt, int c, double* d) {
  if (c < 3) { d[0] = d[1] = d[2] = 0.0; d[c] = 1.0; return true; }
  const int jc = c - 3;
  if (!supports(ws, jc, body)) return false;
  double pc[3];
  for (int k = 0; k < 3; ++k) pc[k] = rpt[k] - (jc < 3 ? 0.0 : ws.r[jc - 2][k]);
  v3_cross(ws.S[jc], pc, d);
  return true;
}

// First-order data: residual rows J (compact row layout of NodeWST, written to Jout[nrows_pad][LDJ], scaled by sqrt(dt)) and their
// rho (rho_out[NRS]), d, gd (x dt), CDe.  After node_values() and node_scalars(); ws.G must hold the stage-1 Jacobian.
// Everything here reads LDS and writes the record (and nw.d / nw.gd): one phase.
template <class SW>
HSQP_HD void node_derivatives(const Ctx& ctx, const DevModel& dm, const SW& ws, NodeWS& nw, double dt, double* Jout, double* CDe /*[NE_MAX][LDJ], global*/,
                              double* rho_out) {
  const double sdt = sqrt(dt);
  // ---- foot columns: task-space cost rows + stance / swing equality rows.  Items are grouped by the KIND of their column so that a
  // wave round runs one branch of foot_column: [0, 52) the q columns of both feet, [64, 116) the qd columns, [128, 186) qdd, base
  // position and base linear velocity, [192, 222) the wrench and padding columns (a two-wave workgroup: rounds 1 / 2 of waves 0 / 1).
  WG_FOR(ctx, it, 222) {
    int f, col;
    if (it < 52) { f = it / 26; col = 3 + it % 26; }
    else if (it < 64) continue;
    else if (it < 116) { f = (it - 64) / 26; col = NV + 3 + (it - 64) % 26; }
    else if (it < 128) continue;
    else if (it < 174) { f = (it - 128) / 23; col = NX + 12 + (it - 128) % 23; }
    else if (it < 186) { const int j = (it - 174) % 6; f = (it - 174) / 6; col = j < 3 ? j : NV + j - 3; }
    else if (it < 192) continue;
    else if (it < 216) { f = (it - 192) / 12; col = NX + (it - 192) % 12; }
    else { f = (it - 216) / 3; col = NZ + (it - 216) % 3; }
    if (col >= NZ) {
      for (int k = 0; k < 15; ++k) Jout[(ROW_FOOT + 15 * f + k) * LDJ + col] = 0.0;
      continue;
    }
    double dq[18];
    foot_column(dm, ws, nw, f, col, dq);
    for (int k = 0; k < 15; ++k) Jout[(ROW_FOOT + 15 * f + k) * LDJ + col] = sdt * nw.scale[ROW_FOOT + 15 * f + k] * dq[3 + k];
    const int r0 = nw.eq_off[f];
    if (nw.contact[f]) {
      for (int k = 0; k < 6; ++k) {
        const int c = k % 3;
        const double gp = k < 2 ? 0.0 : (k == 2 ? dm.gain_pos_z : dm.gain_ori);
        const double gv = k < 2 ? dm.gain_linvel_xy : (k == 2 ? dm.gain_linvel_z : dm.gain_angvel);
        const double ga = k < 2 ? dm.gain_linacc_xy : (k == 2 ? dm.gain_linacc_z : dm.gain_angacc);
        CDe[(r0 + k) * LDJ + col] = k < 3 ? gp * dq[c] + gv * dq[6 + c] + ga * dq[12 + c] : gp * dq[3 + c] + gv * dq[9 + c] + ga * dq[15 + c];
      }
    } else {
      for (int k = 0; k < 6; ++k) CDe[(r0 + k) * LDJ + col] = (col == NX + 6 * f + k) ? 1.0 : 0.0;
      CDe[(r0 + 6) * LDJ + col] = dm.gain_pos_z * dq[2] + dm.gain_linvel_z * dq[8] + dm.gain_linacc_z * dq[14];
    }
  }
  PH_TICK(ctx, 31);
  // ---- friction cone and contact moment rows of the feet in contact: one item per column, its eight rows of a foot unrolled
  WG_FOR(ctx, col, LDJ) {
    for (int f = 0; f < 2; ++f) {
      const int r0 = nw.row_fm[f];
      if (r0 < 0) continue;
      double v[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      const int cu = col - (NX + 6 * f);
      if (cu >= 0 && cu < 3) {   // FrictionForceConeConstraint.cpp:153-178 (first and second derivative of the cone)
        const double Fx = nw.u[6 * f], Fy = nw.u[6 * f + 1];
        const double Tn = sqrt(Fx * Fx + Fy * Fy + dm.friction_reg);
        v[0] = cu == 0 ? -Fx / Tn : (cu == 1 ? -Fy / Tn : dm.friction_mu);
        v[1] = cu == 0 ? 1.0 : 0.0;
        v[2] = cu == 1 ? 1.0 : 0.0;
        v[3] = cu == 0 ? Fy : (cu == 1 ? -Fx : 0.0);
      }
      {   // d/dz of (R^T m)_x,y and (R^T f)_z
        const double* Rf = ws.R[dm.contact_body[f]];
        double dlf[3] = {0, 0, 0}, dlm[3] = {0, 0, 0};
        if (col >= 3 && col < NV) {
          const int jc = col - 3;
          if (supports(ws, jc, dm.contact_body[f])) {
            double t[3];
            v3_cross(nw.u + 6 * f, ws.S[jc], t);
            m3_tmulv(Rf, t, dlf);
            v3_cross(nw.u + 6 * f + 3, ws.S[jc], t);
            m3_tmulv(Rf, t, dlm);
          }
        } else if (cu >= 0 && cu < 3) {
          for (int r = 0; r < 3; ++r) dlf[r] = Rf[3 * cu + r];
        } else if (cu >= 3 && cu < 6) {
          for (int r = 0; r < 3; ++r) dlm[r] = Rf[3 * (cu - 3) + r];
        }
        v[4] = dlm[0] - dm.rect_y_min * dlf[2];
        v[5] = -dlm[0] + dm.rect_y_max * dlf[2];
        v[6] = -dlm[1] - dm.rect_x_min * dlf[2];
        v[7] = dlm[1] + dm.rect_x_max * dlf[2];
      }
      for (int k = 0; k < 8; ++k) {
        const double sc = nw.scale[(k < 4 ? ROW_FRIC + 4 * f : ROW_MXY + 4 * f - 4) + k];
        Jout[(r0 + k) * LDJ + col] = col < NZ ? sc * sdt * v[k] : 0.0;
      }
    }
  }
  // ---- collision rows (present only while one of them is active) and the zero rows that pad the last 24-row pass
  if (nw.row_coll >= 0) {
    WG_FOR(ctx, it, 16 * LDJ) {
      const int r = it / LDJ, col = it % LDJ;
      const double sc = nw.scale[ROW_COLL + r];
      double val = 0.0;
      if (sc != 0.0 && col < NV) {
        int a, b;
        coll_pair(r, a, b);
        double pa[3], pb[3], da[3] = {0, 0, 0}, db[3] = {0, 0, 0}, dd[3];
        coll_point(dm, ws, a, pa);
        coll_point(dm, ws, b, pb);
        point_column(dm, ws, dm.coll_body[a], pa, col, da);
        point_column(dm, ws, dm.coll_body[b], pb, col, db);
        for (int k = 0; k < 3; ++k) dd[k] = pa[k] - pb[k];
        const double n = sqrt(v3_dot(dd, dd));
        val = sc * sdt * (dd[0] * (da[0] - db[0]) + dd[1] * (da[1] - db[1]) + dd[2] * (da[2] - db[2])) / n;
      }
      Jout[(nw.row_coll + r) * LDJ + col] = val;
    }
  }
  WG_FOR(ctx, it, (nw.nrows_pad - nw.nrows) * LDJ) Jout[nw.nrows * LDJ + it] = 0.0;
  // ---- rho of the compact rows (row slots -> record rows), zero beyond them
  WG_FOR(ctx, it, 2 * NRS) {
    if (it < NRS) {
      const int s = it;
      int row = -1;
      if (s < ROW_FRIC) row = s;
      else if (s < ROW_MXY) { const int f = (s - ROW_FRIC) / 4; if (nw.row_fm[f] >= 0) row = nw.row_fm[f] + (s - ROW_FRIC) % 4; }
      else if (s < ROW_COLL) { const int f = (s - ROW_MXY) / 4; if (nw.row_fm[f] >= 0) row = nw.row_fm[f] + 4 + (s - ROW_MXY) % 4; }
      else if (s < ROW_COLL + 16) { if (nw.row_coll >= 0) row = nw.row_coll + s - ROW_COLL; }
      if (row >= 0) rho_out[row] = sdt * nw.rho[s];
    } else if (it - NRS >= nw.nrows) {
      rho_out[it - NRS] = 0.0;
    }
  }
  PH_TICK(ctx, 32);
  // ---- diagonal part
  WG_FOR(ctx, i, LDJ) {
    double d = 0.0, g = 0.0;
    if (i < NX) { d = dm.Q[i]; g = dm.Q[i] * (nw.x[i] - nw.xnom[i]); }
    else if (i < NZ) { d = dm.R[i - NX]; g = dm.R[i - NX] * (nw.u[i - NX] - nw.unom[i - NX]); }
    if (i >= 6 && i < NV) {
      const int j = i - 6;
      const Pen3 lo = pwp_barrier(dm.jl_bmu, dm.jl_bdelta, nw.x[i] - dm.q_lo[j]);
      const Pen3 hi = pwp_barrier(dm.jl_bmu, dm.jl_bdelta, dm.q_hi[j] - nw.x[i]);
      d += lo.d2 + hi.d2;
      g += lo.d1 - hi.d1;
    }
    if (i < NZ) { d += nw.fshift[0]; d += nw.fshift[1]; }   // friction cone: hessianDiagonalShift on every state and input (FrictionForceConeConstraint.cpp:213-224)
    nw.d[i] = dt * d;
    nw.gd[i] = dt * g;
  }
  WG_FOR(ctx, r, NE_MAX) {
    if (r < nw.ne) CDe[r * LDJ + NZ] = nw.eqv[r];
    for (int c = NZ + (r < nw.ne ? 1 : 0); c < LDJ; ++c) CDe[r * LDJ + c] = 0.0;
    if (r >= nw.ne) for (int c = 0; c < NZ + 1; ++c) CDe[r * LDJ + c] = 0.0;
  }
  WG_SYNC(ctx);
}

}  // namespace hsqp
