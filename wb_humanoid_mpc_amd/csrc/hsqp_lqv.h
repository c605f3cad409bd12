// Value-only evaluation of a whole-body shooting node — RK4 value of the flow map, stage cost, equality values: what the performance
// index and the line-search trials need (lq_node<false>, hsqp_lq.h, computes the same numbers) — on a QUAD of lanes.
//
// The phase form (hsqp_model.h: one wave per node, every quantity of the tree in LDS, a barrier between dependent quantities) keeps
// 20 - 48 of a wave's 64 lanes busy through ~60 short phases per node: 8.2 k vector instructions per node, issue-bound.  The value pass
// needs no per-body quantity afterwards — only sums over the bodies (net force and inertia about the base origin) and the two foot
// frames — so here a LANE walks one LIMB (a root-to-leaf path of the kinematic tree: two legs, waist + arm twice) with the running
// placement / velocity / acceleration in registers, adds the bodies it owns to its partial sums, and the four lanes of a node meet in
// two DPP quad permutes per RK4 stage.  A wave evaluates 16 nodes; no barrier, no LDS traffic but the node's (x, u) and the body constants;
// the shared waist bodies are re-walked by both arm lanes (and owned by the first).  ~1 k vector instructions per node.
//
// Same formulas as stage_eval<false> / node_values / node_scalars (each block cites them); the sums over bodies and cost terms are
// taken in another order, so the results agree with lq_node<false> to rounding (host build: tests/hostemu; device: tests/test_gpu_parity.py).
#pragma once
#include <cstddef>
#include "hsqp_lq.h"

namespace hsqp {

#if defined(__HIP_DEVICE_COMPILE__)
#define QV_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)   /* keeps the ILP scheduler from hoisting a later block's operand loads over this point (register pressure) */
#else
#define QV_SCHED_FENCE() ((void)0)
#endif

constexpr int QV_NODES = 16, QV_THREADS = 64;   // a wave evaluates 16 nodes, four lanes each
static_assert(QV_NODES * QV_LIMBS == QV_THREADS && QV_LIMBS == 4, "a node is a DPP quad");

struct QvConst {   // body constants (one copy per workgroup: lanes index them by the body of their step).  Two pieces in DevModel's own order and
  double Rfix[NB][9], pfix[NB][3], axis[NB][3], axis_p[NB][3];   // layout — {Rfix .. axis_p} and {mass .. inertia} — so that loading them is two flat copies
  double mass[NB], com[NB][3], inertia[NB][9];
};
constexpr int QV_WAVES = 2;         // waves per workgroup: they share the body constants (37 KB per workgroup: eight waves per CU)
struct QvWS {
  QvConst k;
  struct {
    double x[QV_NODES][NX], u[QV_NODES][NU];
    double cp[QV_NODES][10][3];    // collision points relative to the base origin (stage 1), written by the lanes that own their bodies
  } wv[QV_WAVES];
};
static_assert(sizeof(QvWS) * 4 <= 163840, "four workgroups per CU");

template <class F>
HSQP_HD void qv_load_const(const Ctx& ctx, const DevModel& dm, QvConst& k, F&& sync) {
  constexpr int NA = NB * 18, NBK = NB * 13;
  static_assert(offsetof(DevModel, axis_p) - offsetof(DevModel, Rfix) == (NA - NB * 3) * sizeof(double) && offsetof(DevModel, inertia) - offsetof(DevModel, mass) == (NBK - NB * 9) * sizeof(double) &&
                offsetof(QvConst, mass) == NA * sizeof(double) && sizeof(QvConst) == (NA + NBK) * sizeof(double), "two flat pieces");
  const double* sa = &dm.Rfix[0][0];
  const double* sb = &dm.mass[0];
  double* d = &k.Rfix[0][0];
  WG_FOR(ctx, i, NA + NBK) d[i] = i < NA ? sa[i] : sb[i - NA];
  sync();
}

// what a lane carries from one RK4 stage to the next and to the node terms (registers on the device)
struct QvCarry {
  double vb[6], ap[6];       // base part of the previous stage's velocity, base acceleration of the previous stage
  double sv[6], sa[6];       // RK4 sums v1 + 2 v2 + 2 v3 + v4, a1 + 2 a2 + 2 a3 + a4 of the base rows
  double ecs[4];             // cos, sin of the euler angles z, y of the stage being evaluated (the euler-rate map of its base solve)
  double fR[9], fr[3], fv[6], fa[6];   // stage 1: placement, velocity and (gravity-trick) acceleration of the lane's foot body
  double y0[3], ab0[6];      // stage 1: E a_ang, base acceleration
};
HSQP_HD void qv_carry_init(QvCarry& c) {
  for (int k = 0; k < 6; ++k) { c.vb[k] = 0.0; c.ap[k] = 0.0; c.sv[k] = 0.0; c.sa[k] = 0.0; c.fv[k] = 0.0; c.fa[k] = 0.0; c.ab0[k] = 0.0; }
  for (int k = 0; k < 9; ++k) c.fR[k] = 0.0;
  for (int k = 0; k < 3; ++k) { c.fr[k] = 0.0; c.y0[k] = 0.0; }
  for (int k = 0; k < 4; ++k) c.ecs[k] = 0.0;
}

// One RK4 stage s (0..3) of limb L: part[0..5] = this lane's share of F_ext - F {moment, force} about the base origin, part[6..15] of the
// total spatial inertia (stage_eval, hsqp_model.h:143-394; the stage inputs: rk4_stage_inputs, hsqp_lq.h:57).  x, u: the node's state /
// input; cp: the node's collision points (written at stage 0).
HSQP_HD void qv_limb_stage(const DevModel& dm, const QvConst& kc, const double* x, const double* u, int L, int s, double dt, QvCarry& c, double* part,
                           double (*cp)[3]) {
  const double cs = s == 0 ? 0.0 : (s == 3 ? dt : 0.5 * dt);
  const double cprev = s <= 1 ? 0.0 : 0.5 * dt;   // the factor of the stage before (whose velocity moves this stage's q)
  double qe[3], vb[6];
  for (int k = 0; k < 3; ++k) qe[k] = x[3 + k] + (s == 0 ? 0.0 : cs * c.vb[3 + k]);
  for (int k = 0; k < 6; ++k) vb[k] = x[NV + k] + cs * (s == 0 ? 0.0 : c.ap[k]);
  for (int e = 0; e < 16; ++e) part[e] = 0.0;
  // ---- the base: euler chain z -> y' -> x'' (F0 - F4 of stage_eval for the three euler links)
  double sz, cz, sy, cy, sx, cx;
  sincos(qe[0], &sz, &cz);
  sincos(qe[1], &sy, &cy);
  sincos(qe[2], &sx, &cx);
  const double wz[3] = {0.0, 0.0, 1.0}, wy[3] = {-sz, cz, 0.0}, wx[3] = {cz * cy, sz * cy, -sy};
  c.ecs[0] = cz; c.ecs[1] = sz; c.ecs[2] = cy; c.ecs[3] = sy;
  double R[9] = {cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx, sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx, -sy, cy * sx, cy * cx};
  double r[3] = {0.0, 0.0, 0.0};
  double vl[6], al[6];
  {
    const double S0[6] = {wz[0], wz[1], wz[2], 0.0, 0.0, 0.0}, S1[6] = {wy[0], wy[1], wy[2], 0.0, 0.0, 0.0}, S2[6] = {wx[0], wx[1], wx[2], 0.0, 0.0, 0.0};
    double Sd0[6], Sd1[6], Sd2[6];
    for (int k = 0; k < 3; ++k) { vl[k] = S0[k] * vb[3]; vl[3 + k] = vb[k]; }
    mxm(vl, S0, Sd0);
    for (int k = 0; k < 3; ++k) vl[k] += S1[k] * vb[4];
    mxm(vl, S1, Sd1);
    for (int k = 0; k < 3; ++k) vl[k] += S2[k] * vb[5];
    mxm(vl, S2, Sd2);
    for (int k = 0; k < 6; ++k) al[k] = (k == 5 ? dm.gravity : 0.0) + Sd0[k] * vb[3] + Sd1[k] * vb[4] + Sd2[k] * vb[5];
  }
  // a body's spatial inertia about the base origin and its net force (stage_eval: "per-body spatial inertia about O and net force"); the
  // collision points and the contact wrench / foot frame it carries
  auto body = [&](int i, bool own) {
    const double* Rb = R;
    double cc3[3], t[9], Iw[9], In[10];
    m3_mulv(Rb, kc.com[i], cc3);
    for (int k = 0; k < 3; ++k) cc3[k] += r[k];
    m3_mul(Rb, kc.inertia[i], t);
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) Iw[3 * a + b] = t[3 * a] * Rb[3 * b] + t[3 * a + 1] * Rb[3 * b + 1] + t[3 * a + 2] * Rb[3 * b + 2];
    const double m = kc.mass[i], cc = v3_dot(cc3, cc3);
    In[0] = m; In[1] = m * cc3[0]; In[2] = m * cc3[1]; In[3] = m * cc3[2];
    In[4] = Iw[0] + m * (cc - cc3[0] * cc3[0]); In[5] = Iw[1] - m * cc3[0] * cc3[1]; In[6] = Iw[2] - m * cc3[0] * cc3[2];
    In[7] = Iw[4] + m * (cc - cc3[1] * cc3[1]); In[8] = Iw[5] - m * cc3[1] * cc3[2]; In[9] = Iw[8] + m * (cc - cc3[2] * cc3[2]);
    QV_SCHED_FENCE();
    double h[6], fa[6], fv[6];
    inertia_apply(In, vl, h);
    inertia_apply(In, al, fa);
    mxf(vl, h, fv);
    const double mk = own ? 1.0 : 0.0;
    for (int k = 0; k < 6; ++k) part[k] -= mk * (fa[k] + fv[k]);
    for (int e = 0; e < 10; ++e) part[6 + e] += mk * In[e];
    QV_SCHED_FENCE();
    for (int f = 0; f < 2; ++f) {
      if (dm.contact_body[f] != i || dm.foot_limb[f] != L) continue;
      // contact point of foot f relative to O and its wrench about O {moment, force}
      double rr[3], mom[3];
      m3_mulv(Rb, dm.contact_p[f], rr);
      for (int k = 0; k < 3; ++k) rr[k] += r[k];
      v3_cross(rr, u + 6 * f, mom);
      for (int k = 0; k < 3; ++k) { part[k] += u[6 * f + 3 + k] + mom[k]; part[3 + k] += u[6 * f + k]; }
      if (s == 0) {
        for (int k = 0; k < 9; ++k) c.fR[k] = R[k];
        for (int k = 0; k < 3; ++k) c.fr[k] = r[k];
        for (int k = 0; k < 6; ++k) { c.fv[k] = vl[k]; c.fa[k] = al[k]; }
      }
    }
    if (s == 0 && own) {
      for (int p = 0; p < 10; ++p) {
        if (dm.coll_body[p] != i) continue;
        double tt[3];
        m3_mulv(Rb, dm.coll_p[p], tt);
        for (int k = 0; k < 3; ++k) cp[p][k] = r[k] + tt[k];
      }
    }
  };
  if (L == 0) body(0, true);
  // ---- the limb's bodies, root first
  const unsigned long long path = dm.limb_path[L];
  const int len = dm.limb_len[L];
  const unsigned own = dm.limb_own[L];
  for (int st = 0; st < dm.limb_max_len; ++st) {
    if (st >= len) continue;
    const int i = (int)((path >> (8 * st)) & 0xffull), j = i - 1;
    const double qdd = u[12 + j];
    const double qj = x[6 + j] + (s == 0 ? 0.0 : cs * (x[NV + 6 + j] + cprev * qdd));
    const double qd = x[NV + 6 + j] + cs * qdd;
    double sn, cn, Rq[9], Mq[9], Rn[9], w[3], rn[3];
    sincos(qj, &sn, &cn);
    rot_axis_cs(kc.axis[i], cn, sn, Rq);
    m3_mul(kc.Rfix[i], Rq, Mq);
    m3_mulv(R, kc.axis_p[i], w);       // joint axis in world axes (from the parent's rotation)
    m3_mulv(R, kc.pfix[i], rn);
    for (int k = 0; k < 3; ++k) rn[k] += r[k];
    m3_mul(R, Mq, Rn);
    double S[6], Sd[6];
    for (int k = 0; k < 3; ++k) S[k] = w[k];
    v3_cross(rn, w, S + 3);
    for (int k = 0; k < 6; ++k) vl[k] += S[k] * qd;
    mxm(vl, S, Sd);
    for (int k = 0; k < 6; ++k) al[k] += S[k] * qdd + Sd[k] * qd;
    for (int k = 0; k < 9; ++k) R[k] = Rn[k];
    for (int k = 0; k < 3; ++k) r[k] = rn[k];
    QV_SCHED_FENCE();
    body(i, ((own >> st) & 1u) != 0);
  }
}

// totals of a stage -> base acceleration (stage_eval: "totals and the block-diagonal base solve"); moves the carry on to the next stage
HSQP_HD void qv_base_solve(const double* tot, const double* x, int s, double dt, QvCarry& c) {
  const double cs = s == 0 ? 0.0 : (s == 3 ? dt : 0.5 * dt);
  double Einv[9], vcur[6];
  {
    const double cz = c.ecs[0], sz = c.ecs[1], cy = c.ecs[2], sy = c.ecs[3];
    const double E[9] = {0.0, -sz, cz * cy, 0.0, cz, sz * cy, 1.0, 0.0, -sy};   // columns: the world axes of the euler rates z, y, x (stage_eval F1)
    m3_inverse(E, Einv);
  }
  for (int k = 0; k < 6; ++k) vcur[k] = x[NV + k] + cs * (s == 0 ? 0.0 : c.ap[k]);   // the stage's base velocity (as qv_limb_stage formed it)
  const double* I6 = tot + 10;
  const double Ib[9] = {I6[0], I6[1], I6[2], I6[1], I6[3], I6[4], I6[2], I6[4], I6[5]};
  double Iinv[9], y[3], ab[6];
  m3_inverse(Ib, Iinv);
  m3_mulv(Iinv, tot, y);
  const double minv = 1.0 / tot[6];
  for (int k = 0; k < 3; ++k) ab[k] = tot[3 + k] * minv;
  m3_mulv(Einv, y, ab + 3);
  const double wg = (s == 0 || s == 3) ? 1.0 : 2.0;
  for (int k = 0; k < 6; ++k) {
    c.sv[k] += wg * vcur[k];
    c.sa[k] += wg * ab[k];
    c.vb[k] = vcur[k];
    c.ap[k] = ab[k];
  }
  if (s == 0) {
    for (int k = 0; k < 3; ++k) c.y0[k] = y[k];
    for (int k = 0; k < 6; ++k) c.ab0[k] = ab[k];
  }
}

// This lane's share of the stage cost and of the squared equality values (node_values / node_scalars, hsqp_node.h:90-267), after stage 1's
// base solve.  cp: the node's collision points (all four lanes' writes visible).
HSQP_HD void qv_node_terms(const DevModel& dm, const double* x, const double* u, const double* par, int L, const QvCarry& c, const double (*cp)[3],
                           double& cost, double& eq) {
  const int c0 = par[HSQP_P_CONTACT] > 0.5, c1 = par[HSQP_P_CONTACT + 1] > 0.5;
  double cst = 0.0, eqs = 0.0;
  // StateInputQuadraticCost (StateInputQuadraticCost.cpp:67-78; arm swing: SwitchedModelReferenceManager.cpp:110-135)
  const double yaw = x[3];
  const double vloc = cos(yaw) * par[HSQP_P_XDES + NV] + sin(yaw) * par[HSQP_P_XDES + NV + 1];
  const double gcf = par[HSQP_P_ARMSWING] * vloc;
  for (int t = L; t < NZ; t += QV_LIMBS) {
    if (t < NX) {
      double xn = par[HSQP_P_XDES + t];
      const int j = t - 6;
      if (j == dm.arm_swing_joint[0]) xn += -0.15 * gcf;
      if (j == dm.arm_swing_joint[1]) xn += 0.15 * gcf;
      if (j == dm.arm_swing_joint[2]) xn += -0.15 * gcf;
      if (j == dm.arm_swing_joint[3]) xn += 0.15 * gcf;
      const double dxx = x[t] - xn;
      cst += 0.5 * dm.Q[t] * dxx * dxx;
    } else {   // weightCompensatingInput (DynamicsHelperFunctions.h:178-193)
      const int i = t - NX;
      double un = 0.0;
      if ((i == 2 && c0) || (i == 8 && c1)) un = dm.total_mass * 9.81 / (c0 + c1);
      const double duu = u[i] - un;
      cst += 0.5 * dm.R[i] * duu * duu;
    }
  }
  // JointLimitsSoftConstraint.cpp:64-100
  for (int j = L; j < NJ; j += QV_LIMBS) {
    cst += pwp_barrier(dm.jl_bmu, dm.jl_bdelta, x[6 + j] - dm.q_lo[j]).p;
    cst += pwp_barrier(dm.jl_bmu, dm.jl_bdelta, dm.q_hi[j] - x[6 + j]).p;
  }
  // foot collision distances (FootCollisionConstraint.cpp:118-141)
  if (!(c0 && c1)) {
    for (int rw = L; rw < 16; rw += QV_LIMBS) {
      int a, b;
      coll_pair(rw, a, b);
      double dd[3];
      for (int k = 0; k < 3; ++k) dd[k] = cp[a][k] - cp[b][k];
      const double h = sqrt(v3_dot(dd, dd)) - 2.0 * (rw == 9 ? dm.r_knee : dm.r_foot);
      cst += pwp_barrier(dm.coll_bmu, dm.coll_bdelta, h).p;
    }
  }
  // the lane's foot: frame quantities, task-space cost, friction cone, contact moment, equality values
  for (int f = 0; f < 2; ++f) {
    if (dm.foot_limb[f] != L) continue;
    const int cf = f == 0 ? c0 : c1;
    double rP[3], a[6], o[18], t[3], t2[3];
    m3_mulv(c.fR, dm.contact_p[f], rP);
    for (int k = 0; k < 3; ++k) rP[k] += c.fr[k];
    for (int k = 0; k < 3; ++k) { a[k] = c.fa[k] + c.y0[k]; a[3 + k] = c.fa[3 + k] + c.ab0[k]; }
    a[5] -= dm.gravity;
    for (int k = 0; k < 3; ++k) o[k] = x[k] + rP[k];
    ori_error(c.fR, o + 3);
    v3_cross(c.fv, rP, t);
    for (int k = 0; k < 3; ++k) { o[6 + k] = c.fv[3 + k] + t[k]; o[9 + k] = c.fv[k]; }
    v3_cross(a, rP, t);
    v3_cross(c.fv, o + 6, t2);
    for (int k = 0; k < 3; ++k) { o[12 + k] = a[3 + k] + t[k] + t2[k]; o[15 + k] = a[k]; }
    // EndEffectorDynamicsFootCost.cpp:91-124
    for (int k = 0; k < 15; ++k) {
      const double rho = dm.foot_sqrt_w[3 + k] * par[HSQP_P_IMPACT + f] * o[3 + k];
      cst += 0.5 * rho * rho;
    }
    if (cf) {
      // FrictionForceConeConstraint.cpp:180-185, ContactMomentXYConstraintCppAd.cpp:87-104
      const double Fx = u[6 * f], Fy = u[6 * f + 1], Fz = u[6 * f + 2];
      const double hf = dm.friction_mu * (Fz + dm.friction_grip) - sqrt(Fx * Fx + Fy * Fy + dm.friction_reg);
      cst += relaxed_barrier(dm.friction_bmu, dm.friction_bdelta, hf).p;
      double lf[3], lm[3];
      m3_tmulv(c.fR, u + 6 * f, lf);
      m3_tmulv(c.fR, u + 6 * f + 3, lm);
      const double hm[4] = {lm[0] - dm.rect_y_min * lf[2], -lm[0] + dm.rect_y_max * lf[2], -lm[1] - dm.rect_x_min * lf[2], lm[1] + dm.rect_x_max * lf[2]};
      for (int k = 0; k < 4; ++k) cst += relaxed_barrier(dm.moment_bmu, dm.moment_bdelta, hm[k]).p;
      // EndEffectorDynamicsAccelerationsConstraint.cpp:82-103, gains WBMpcInterface.cpp:205-229
      for (int k = 0; k < 6; ++k) {
        const int cc = k % 3;
        const double gp = k < 2 ? 0.0 : (k == 2 ? dm.gain_pos_z : dm.gain_ori);
        const double gv = k < 2 ? dm.gain_linvel_xy : (k == 2 ? dm.gain_linvel_z : dm.gain_angvel);
        const double ga = k < 2 ? dm.gain_linacc_xy : (k == 2 ? dm.gain_linacc_z : dm.gain_angacc);
        const double e = k < 3 ? gp * o[cc] + gv * o[6 + cc] + ga * o[12 + cc] : gp * o[3 + cc] + gv * o[9 + cc] + ga * o[15 + cc];
        eqs += e * e;
      }
    } else {
      // ZeroWrenchConstraint.cpp:59-84, EndEffectorDynamicsLinearAccConstraint.cpp:69-83 (config WBMpcPreComputation.cpp:91-104)
      for (int k = 0; k < 6; ++k) eqs += u[6 * f + k] * u[6 * f + k];
      const double* sw = par + HSQP_P_SWING + 3 * f;
      const double e = -dm.gain_linvel_z * sw[1] - dm.gain_linacc_z * sw[2] - dm.gain_pos_z * sw[0] + dm.gain_pos_z * o[2] + dm.gain_linvel_z * o[8] +
                       dm.gain_linacc_z * o[14];
      eqs += e * e;
    }
  }
  cost = cst;
  eq = eqs;
}

// this lane's share of the squared RK4 defect (lq_node: "RK4 value")
HSQP_HD double qv_defect(const double* x, const double* u, const double* xnext, int L, double dt, const QvCarry& c) {
  double dyn = 0.0;
  for (int j = L; j < NJ; j += QV_LIMBS) {
    const double v0 = x[NV + 6 + j], qdd = u[12 + j];
    const double v1 = v0 + 0.5 * dt * qdd, v3 = v0 + dt * qdd;   // stage velocities (stages 2 and 3 share v1)
    const double bq = x[6 + j] + dt / 6.0 * (v0 + 2.0 * v1 + 2.0 * v1 + v3) - xnext[6 + j];
    const double bv = v0 + dt * qdd - xnext[NV + 6 + j];
    dyn += bq * bq + bv * bv;
  }
  if (L == 0) {
    for (int k = 0; k < 6; ++k) {
      const double bq = x[k] + dt / 6.0 * c.sv[k] - xnext[k];
      const double bv = x[NV + k] + dt * (c.sa[k] / 6.0) - xnext[NV + k];
      dyn += bq * bq + bv * bv;
    }
  }
  return dyn;
}

// misc[0..7] of a node from the node's totals (lq_node's layout)
HSQP_HD void qv_write_misc(const double* par, double dt, double cost, double eq, double dyn, double* misc) {
  const int c0 = par[HSQP_P_CONTACT] > 0.5, c1 = par[HSQP_P_CONTACT + 1] > 0.5;
  const int off1 = c0 ? 6 : 7;
  misc[0] = (double)(off1 + (c1 ? 6 : 7));
  misc[1] = dt * cost;
  misc[2] = dt * eq;
  misc[3] = (dt > 0.0 ? dt : 1.0) * dyn;
  misc[4] = (double)c0; misc[5] = (double)c1;
  misc[6] = 0.0; misc[7] = (double)off1;
}

#if !defined(__HIP_DEVICE_COMPILE__)
// The four lanes of a node one after the other, the quad sums in between (host builds: tests/hostemu)
inline void qv_node_host(const DevModel& dm, const double* x, const double* u, const double* xnext, const double* par, double dt, double* misc) {
  QvConst* kc = new QvConst;
  const Ctx ctx{0, 1, nullptr};
  qv_load_const(ctx, dm, *kc, [] {});
  QvCarry c[QV_LIMBS];
  double cp[10][3] = {};
  for (int L = 0; L < QV_LIMBS; ++L) qv_carry_init(c[L]);
  for (int s = 0; s < 4; ++s) {
    double part[QV_LIMBS][16], tot[16];
    for (int L = 0; L < QV_LIMBS; ++L) qv_limb_stage(dm, *kc, x, u, L, s, dt, c[L], part[L], cp);
    for (int e = 0; e < 16; ++e) tot[e] = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
    for (int L = 0; L < QV_LIMBS; ++L) qv_base_solve(tot, x, s, dt, c[L]);
  }
  double cost[QV_LIMBS], eq[QV_LIMBS], dyn[QV_LIMBS];
  for (int L = 0; L < QV_LIMBS; ++L) {
    qv_node_terms(dm, x, u, par, L, c[L], cp, cost[L], eq[L]);
    dyn[L] = qv_defect(x, u, xnext, L, dt, c[L]);
  }
  qv_write_misc(par, dt, (cost[0] + cost[1]) + (cost[2] + cost[3]), (eq[0] + eq[1]) + (eq[2] + eq[3]), (dyn[0] + dyn[1]) + (dyn[2] + dyn[3]), misc);
  delete kc;
}
#endif

}  // namespace hsqp
