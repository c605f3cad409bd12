// LQ approximation of a whole-body shooting node on LIMB LANES — the derivative half of the rigid-body model
// (hsqp_model.h: stage_eval<true>) re-formulated so that a wave carries 16 nodes and no lane idles.
//
// What it replaces: the phase form runs one 128-thread workgroup per node through ~70 barrier-separated phases whose item
// counts are 1 .. 156 — 18.4 k vector instructions per node, a third of the lanes busy, issue-bound at 2 waves per SIMD.
// Here a LANE walks one LIMB (a root-to-leaf path of the kinematic tree, DevModel::limb_*, as the value pass of hsqp_lqv.h does)
//   forward  (root -> leaf): placement, spatial velocity, gravity-trick acceleration in registers; the bodies the lane owns go
//            into its share of F_ext - F and of the total inertia; the four lanes of a node meet in DPP quad sums, every lane
//            solves the 3 x 3 base systems itself (same code as the value pass);
//   backward (leaf -> root): the body state is UNWOUND joint by joint (R_p = R_i Mq_i^T, v_p = v_i - S_i qd_i, ...: no per-body
//            storage), the composites of the subtree accumulate in registers, and the three Jacobian columns of the joint
//            (d/dq, d/dqd, d/dqdd of the base acceleration) are formed on the spot and stored.  Where two limbs share their
//            root-side bodies (waist of the two arms) the lanes exchange composites once (DevModel::limb_merge); the base
//            columns (euler angles / rates, contact wrenches) follow from the quad sum of the limbs' composites.
//
// The composite of the velocity-product term needs 12 numbers per body, not 36: with I = [[Ibar, h x], [-h x, m]], v = (w, u),
//   BB m = -I (v x m) + m x* (I v) + v x* (I m)  =  ( Bn m_ang ;  -2 f_v x m_ang ),      (I v) = (n_v ; f_v),
//   Bn = C + C^T - u h^T - h u^T + 2 (h.u) 1 - [n_v]x ,   C = [w]x Ibar
// (the blocks that multiply m_lin cancel identically: [w]x[h]x - [h]x[w]x = [w x h]x and f_v = m u + w x h).
//
// Formulas: stage_eval (hsqp_model.h:143-496), cited per block.  Reference: computeBaseAcceleration,
// humanoid_nmpc/humanoid_wb_mpc/src/dynamics/DynamicsHelperFunctions.cpp:52-134 (the CppAD tape this replaces).
#pragma once
#include "hsqp_lqv.h"

namespace hsqp {

constexpr int QL_NODES = 16, QL_THREADS = 64;   // a wave evaluates 16 nodes, four lanes each
constexpr int QL_MAXLEN = NANC;                 // steps of the longest limb
constexpr int NCMP = 28;                        // composite of a subtree: spatial inertia (10), net force (6), Bn (9), f_v (3)
constexpr int CMP_I = 0, CMP_F = 10, CMP_BN = 16, CMP_FV = 25;
constexpr int GT_LD = 6;                        // the stage Jacobians travel TRANSPOSED: [stage][column][6] (a lane owns a column)

// kinematics of RK4 stage 1 as the node terms need them (written by the limb lanes, read back by the node-term phases)
struct KinImg {
  double S[NJC + 1][6], Sd[NJC + 1][6], vl[NJC + 1][6], al[NJC + 1][6];
  double R[NB + 1][9], r[NB + 1][3];
  double E[9], rP[2][3], y[3], ab[6];
};
constexpr int KIN_SIZE = (int)(sizeof(KinImg) / sizeof(double));

// what every lane of a node knows after the base solve of a stage
struct QlShared {
  double M[9];        // E^-1 Itot^-1: moment balance -> euler-rate acceleration
  double y[3];        // E a_ang
  double minv;        // 1 / total mass
  double ab[6];       // base acceleration {lin, euler-rate acc}
};

struct QlCarry {      // from one RK4 stage to the next
  double vb[6], ap[6];   // base velocity / base acceleration of the previous stage
};

struct QlState { double R[9], r[3], vl[6], al[6]; };   // the body a lane stands on

// ---- per-body quantities (stage_eval: "per-body spatial inertia about O and net force")
HSQP_HD void ql_inertia(const QvConst& kc, int i, const double* Rb, const double* r, double* In) {
  double c[3], t[9], Iw[9];
  m3_mulv(Rb, kc.com[i], c);
  for (int k = 0; k < 3; ++k) c[k] += r[k];
  m3_mul(Rb, kc.inertia[i], t);
  for (int a = 0; a < 3; ++a)
    for (int b = a; b < 3; ++b) Iw[3 * a + b] = t[3 * a] * Rb[3 * b] + t[3 * a + 1] * Rb[3 * b + 1] + t[3 * a + 2] * Rb[3 * b + 2];
  const double m = kc.mass[i], cc = v3_dot(c, c);
  In[0] = m; In[1] = m * c[0]; In[2] = m * c[1]; In[3] = m * c[2];
  In[4] = Iw[0] + m * (cc - c[0] * c[0]); In[5] = Iw[1] - m * c[0] * c[1]; In[6] = Iw[2] - m * c[0] * c[2];
  In[7] = Iw[4] + m * (cc - c[1] * c[1]); In[8] = Iw[5] - m * c[1] * c[2]; In[9] = Iw[8] + m * (cc - c[2] * c[2]);
}
// f = I a + v x* (I v); hv = I v
HSQP_HD void ql_force(const double* In, const double* vl, const double* al, double* f, double* hv) {
  double fa[6], fv[6];
  inertia_apply(In, vl, hv);
  inertia_apply(In, al, fa);
  mxf(vl, hv, fv);
  for (int k = 0; k < 6; ++k) f[k] = fa[k] + fv[k];
}
// Bn of one body (header comment), row-major 3 x 3
HSQP_HD void ql_bn(const double* In, const double* vl, const double* hv, double* Bn) {
  const double* h = In + 1;
  const double* Ib = In + 4;   // xx xy xz yy yz zz
  const double w0 = vl[0], w1 = vl[1], w2 = vl[2], u0 = vl[3], u1 = vl[4], u2 = vl[5];
  // C = [w]x Ibar: column j = w x Ibar[:, j]
  const double I0[3] = {Ib[0], Ib[1], Ib[2]}, I1[3] = {Ib[1], Ib[3], Ib[4]}, I2[3] = {Ib[2], Ib[4], Ib[5]};
  const double C00 = w1 * I0[2] - w2 * I0[1], C10 = w2 * I0[0] - w0 * I0[2], C20 = w0 * I0[1] - w1 * I0[0];
  const double C01 = w1 * I1[2] - w2 * I1[1], C11 = w2 * I1[0] - w0 * I1[2], C21 = w0 * I1[1] - w1 * I1[0];
  const double C02 = w1 * I2[2] - w2 * I2[1], C12 = w2 * I2[0] - w0 * I2[2], C22 = w0 * I2[1] - w1 * I2[0];
  const double hu2 = 2.0 * (h[0] * u0 + h[1] * u1 + h[2] * u2);
  const double n0 = hv[0], n1 = hv[1], n2 = hv[2];
  Bn[0] = 2.0 * C00 - 2.0 * u0 * h[0] + hu2;
  Bn[4] = 2.0 * C11 - 2.0 * u1 * h[1] + hu2;
  Bn[8] = 2.0 * C22 - 2.0 * u2 * h[2] + hu2;
  const double s01 = C01 + C10 - u0 * h[1] - h[0] * u1, s02 = C02 + C20 - u0 * h[2] - h[0] * u2, s12 = C12 + C21 - u1 * h[2] - h[1] * u2;
  // - [n]x = [[0, n2, -n1], [-n2, 0, n0], [n1, -n0, 0]]
  Bn[1] = s01 + n2; Bn[3] = s01 - n2;
  Bn[2] = s02 - n1; Bn[6] = s02 + n1;
  Bn[5] = s12 + n0; Bn[7] = s12 - n0;
}
// a body's contribution to a composite
HSQP_HD void ql_body_comp(const double* In, const double* vl, const double* al, double* own) {
  double hv[6];
  for (int e = 0; e < 10; ++e) own[CMP_I + e] = In[e];
  ql_force(In, vl, al, own + CMP_F, hv);
  ql_bn(In, vl, hv, own + CMP_BN);
  for (int k = 0; k < 3; ++k) own[CMP_FV + k] = hv[3 + k];
}
// BB^c m (only the angular part of m enters)
HSQP_HD void ql_bb_apply(const double* cmp, const double* m, double* out) {
  m3_mulv(cmp + CMP_BN, m, out);
  double t[3];
  v3_cross(cmp + CMP_FV, m, t);
  for (int k = 0; k < 3; ++k) out[3 + k] = -2.0 * t[k];
}

// ---- Jacobian columns of one revolute coordinate from the composite of its subtree (stage_eval: "Jacobian columns", kinds q / qd / qdd).
// S, Sd, Sdd: the joint's motion axis and its derivatives; dext: d(F_ext moment)/dq (contact points that move with the joint);
// xtra: additional moment term (the euler joints z, y: d(S_E)/dq a_ang).  g6 = {lin(3), euler-rate acc(3)} of the column.
HSQP_HD void ql_finish(const QlShared& sh, const double* rhs, const double* dFf, double* g6) {
  for (int k = 0; k < 3; ++k) g6[k] = -dFf[k] * sh.minv;
  m3_mulv(sh.M, rhs, g6 + 3);
}
HSQP_HD void ql_col_q(const double* S, const double* Sd, const double* Sdd, const double* cmp, const QlShared& sh, const double* dext, const double* xtra,
                      double* g6) {
  double t1[6], t2[6], t3[6];
  mxf(S, cmp + CMP_F, t1);
  ql_bb_apply(cmp, Sd, t2);
  inertia_apply(cmp + CMP_I, Sdd, t3);
  // (dI_tot/dq_c) y6 = S x* (I^c y6) - I^c (S x y6), y6 = {y, 0}: moment rows
  const double* h = cmp + CMP_I + 1;
  double Iy[3], hy[3], a1[3], a1b[3], sya[3], syl[3], a2[3], a2b[3];
  sym3_mulv(cmp + CMP_I + 4, sh.y, Iy);
  v3_cross(h, sh.y, hy);
  v3_cross(S, Iy, a1);
  v3_cross(S + 3, hy, a1b);
  v3_cross(S, sh.y, sya);
  v3_cross(S + 3, sh.y, syl);
  sym3_mulv(cmp + CMP_I + 4, sya, a2);
  v3_cross(h, syl, a2b);
  double rhs[3], dFf[3];
  for (int k = 0; k < 3; ++k) {
    const double extra = (a1[k] - a1b[k]) - (a2[k] + a2b[k]) + xtra[k];
    rhs[k] = dext[k] - (t1[k] + t2[k] + t3[k]) - extra;
    dFf[k] = t1[3 + k] + t2[3 + k] + t3[3 + k];
  }
  ql_finish(sh, rhs, dFf, g6);
}
HSQP_HD void ql_col_qd(const double* S, const double* Sd, const double* cmp, const QlShared& sh, double* g6) {
  double t1[6], t2[6], rhs[3], dFf[3];
  inertia_apply(cmp + CMP_I, Sd, t1);
  ql_bb_apply(cmp, S, t2);
  for (int k = 0; k < 3; ++k) { rhs[k] = -(2.0 * t1[k] + t2[k]); dFf[k] = 2.0 * t1[3 + k] + t2[3 + k]; }
  ql_finish(sh, rhs, dFf, g6);
}
HSQP_HD void ql_col_qdd(const double* S, const double* cmp, const QlShared& sh, double* g6) {
  double t1[6], rhs[3];
  inertia_apply(cmp + CMP_I, S, t1);
  for (int k = 0; k < 3; ++k) rhs[k] = -t1[k];
  ql_finish(sh, rhs, t1 + 3, g6);
}

// stage inputs of joint j (rk4_stage_inputs, hsqp_lq.h:57): angle, rate at RK4 stage s
HSQP_HD void ql_joint_inputs(const double* x, const double* u, int j, int s, double dt, double& qj, double& qd, double& qdd) {
  const double cs = s == 0 ? 0.0 : (s == 3 ? dt : 0.5 * dt);
  const double cprev = s <= 1 ? 0.0 : 0.5 * dt;
  qdd = u[12 + j];
  qj = x[6 + j] + (s == 0 ? 0.0 : cs * (x[NV + 6 + j] + cprev * qdd));
  qd = x[NV + 6 + j] + cs * qdd;
}

// ---- the base: euler chain z -> y' -> x'' (F0 - F4 of stage_eval for the three euler links).  All lanes of a node compute the same.
struct QlBaseKin {
  double R[9];
  double w[3][3];                 // angular axes of the euler joints (their linear part is zero: they pass through O)
  double vl[3][6], al[3][6];      // link velocity / trick acceleration after euler joint 0, 1, 2
  double Sd[3][6];
  double cz, sz, cy, sy;
};
HSQP_HD void ql_base_kin(const DevModel& dm, const double* x, int s, double dt, const QlCarry& c, QlBaseKin& b) {
  const double cs = s == 0 ? 0.0 : (s == 3 ? dt : 0.5 * dt);
  double qe[3], vb[6];
  for (int k = 0; k < 3; ++k) qe[k] = x[3 + k] + (s == 0 ? 0.0 : cs * c.vb[3 + k]);
  for (int k = 0; k < 6; ++k) vb[k] = x[NV + k] + cs * (s == 0 ? 0.0 : c.ap[k]);
  double sz, cz, sy, cy, sx, cx;
  sincos(qe[0], &sz, &cz);
  sincos(qe[1], &sy, &cy);
  sincos(qe[2], &sx, &cx);
  b.cz = cz; b.sz = sz; b.cy = cy; b.sy = sy;
  const double R[9] = {cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx, sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx, -sy, cy * sx, cy * cx};
  for (int k = 0; k < 9; ++k) b.R[k] = R[k];
  b.w[0][0] = 0.0; b.w[0][1] = 0.0; b.w[0][2] = 1.0;
  b.w[1][0] = -sz; b.w[1][1] = cz; b.w[1][2] = 0.0;
  b.w[2][0] = cz * cy; b.w[2][1] = sz * cy; b.w[2][2] = -sy;
  double vl[6], al[6];
  for (int k = 0; k < 3; ++k) { vl[k] = 0.0; vl[3 + k] = vb[k]; }
  for (int k = 0; k < 6; ++k) al[k] = k == 5 ? dm.gravity : 0.0;
  for (int e = 0; e < 3; ++e) {
    const double S[6] = {b.w[e][0], b.w[e][1], b.w[e][2], 0.0, 0.0, 0.0};
    for (int k = 0; k < 3; ++k) vl[k] += S[k] * vb[3 + e];
    mxm(vl, S, b.Sd[e]);
    for (int k = 0; k < 6; ++k) al[k] += b.Sd[e][k] * vb[3 + e];
    for (int k = 0; k < 6; ++k) { b.vl[e][k] = vl[k]; b.al[e][k] = al[k]; }
  }
}

// ---- forward pass of limb L at RK4 stage s: leaves the lane on its leaf (st), its share of the totals in part[16] (qv_limb_stage's layout:
// F_ext - F {moment, force}, inertia), the cos / sin of its joints in csn[step * csn_ld + {0, 1}] for the way back, the contact point and
// force of the lane's foot in rP / Ff.  kin (stage 1 only, may be null): the kinematics image of the node.
HSQP_HD void ql_forward(const DevModel& dm, const QvConst& kc, const double* x, const double* u, int L, int s, double dt, const QlBaseKin& bk, QlState& st,
                        double* part, double* csn, int csn_ld, double* rP, double* Ff, KinImg* kin) {
  for (int e = 0; e < 16; ++e) part[e] = 0.0;
  for (int k = 0; k < 9; ++k) st.R[k] = bk.R[k];
  for (int k = 0; k < 3; ++k) st.r[k] = 0.0;
  for (int k = 0; k < 6; ++k) { st.vl[k] = bk.vl[2][k]; st.al[k] = bk.al[2][k]; }
  for (int k = 0; k < 3; ++k) { rP[k] = 0.0; Ff[k] = 0.0; }
  auto body = [&](int i, bool own) {
    double In[10], f[6], hv[6];
    ql_inertia(kc, i, st.R, st.r, In);
    QV_SCHED_FENCE();
    ql_force(In, st.vl, st.al, f, hv);
    const double mk = own ? 1.0 : 0.0;
    for (int k = 0; k < 6; ++k) part[k] -= mk * f[k];
    for (int e = 0; e < 10; ++e) part[6 + e] += mk * In[e];
    QV_SCHED_FENCE();
    for (int fo = 0; fo < 2; ++fo) {
      if (dm.contact_body[fo] != i || dm.foot_limb[fo] != L) continue;
      double rr[3], mom[3];
      m3_mulv(st.R, dm.contact_p[fo], rr);
      for (int k = 0; k < 3; ++k) rr[k] += st.r[k];
      v3_cross(rr, u + 6 * fo, mom);
      for (int k = 0; k < 3; ++k) { part[k] += u[6 * fo + 3 + k] + mom[k]; part[3 + k] += u[6 * fo + k]; rP[k] = rr[k]; Ff[k] = u[6 * fo + k]; }
      if (kin) for (int k = 0; k < 3; ++k) kin->rP[fo][k] = rr[k];
    }
    if (kin && own) {
      for (int k = 0; k < 9; ++k) kin->R[i][k] = st.R[k];
      for (int k = 0; k < 3; ++k) kin->r[i][k] = st.r[k];
    }
  };
  if (L == 0) {
    body(0, true);
    if (kin) {
      for (int e = 0; e < 3; ++e)
        for (int k = 0; k < 6; ++k) { kin->S[e][k] = k < 3 ? bk.w[e][k] : 0.0; kin->Sd[e][k] = bk.Sd[e][k]; kin->vl[e][k] = bk.vl[e][k]; kin->al[e][k] = bk.al[e][k]; }
    }
  }
  const unsigned long long path = dm.limb_path[L];
  const int len = dm.limb_len[L];
  const unsigned own = dm.limb_own[L];
  for (int t = 0; t < dm.limb_max_len; ++t) {
    if (t >= len) continue;
    const int i = (int)((path >> (8 * t)) & 0xffull), j = i - 1;
    double qj, qd, qdd;
    ql_joint_inputs(x, u, j, s, dt, qj, qd, qdd);
    double sn, cn, Rq[9], Mq[9], Rn[9], w[3], rn[3];
    sincos(qj, &sn, &cn);
    csn[t * csn_ld] = cn; csn[t * csn_ld + 1] = sn;
    rot_axis_cs(kc.axis[i], cn, sn, Rq);
    m3_mul(kc.Rfix[i], Rq, Mq);
    m3_mulv(st.R, kc.axis_p[i], w);
    m3_mulv(st.R, kc.pfix[i], rn);
    for (int k = 0; k < 3; ++k) rn[k] += st.r[k];
    m3_mul(st.R, Mq, Rn);
    double S[6], Sd[6];
    for (int k = 0; k < 3; ++k) S[k] = w[k];
    v3_cross(rn, w, S + 3);
    for (int k = 0; k < 6; ++k) st.vl[k] += S[k] * qd;
    mxm(st.vl, S, Sd);
    for (int k = 0; k < 6; ++k) st.al[k] += S[k] * qdd + Sd[k] * qd;
    for (int k = 0; k < 9; ++k) st.R[k] = Rn[k];
    for (int k = 0; k < 3; ++k) st.r[k] = rn[k];
    const bool mine = ((own >> t) & 1u) != 0;
    if (kin && mine)
      for (int k = 0; k < 6; ++k) { kin->S[i + 2][k] = S[k]; kin->Sd[i + 2][k] = Sd[k]; kin->vl[i + 2][k] = st.vl[k]; kin->al[i + 2][k] = st.al[k]; }
    QV_SCHED_FENCE();
    body(i, mine);
  }
}

// totals of a stage -> what the columns need and the base acceleration (qv_base_solve / stage_eval "totals")
HSQP_HD void ql_base_solve(const double* tot, const QlBaseKin& bk, QlShared& sh) {
  const double E[9] = {0.0, -bk.sz, bk.cz * bk.cy, 0.0, bk.cz, bk.sz * bk.cy, 1.0, 0.0, -bk.sy};   // columns: the world axes of the euler rates z, y, x
  double Einv[9], Iinv[9];
  m3_inverse(E, Einv);
  const double* I6 = tot + 10;
  const double Ib[9] = {I6[0], I6[1], I6[2], I6[1], I6[3], I6[4], I6[2], I6[4], I6[5]};
  m3_inverse(Ib, Iinv);
  m3_mulv(Iinv, tot, sh.y);
  sh.minv = 1.0 / tot[6];
  for (int k = 0; k < 3; ++k) sh.ab[k] = tot[3 + k] * sh.minv;
  m3_mulv(Einv, sh.y, sh.ab + 3);
  m3_mul(Einv, Iinv, sh.M);
}
// the carry moves on to the next stage (after the stage's last use of the base kinematics)
HSQP_HD void ql_carry_advance(const double* x, int s, double dt, const QlShared& sh, QlCarry& c) {
  const double cs = s == 0 ? 0.0 : (s == 3 ? dt : 0.5 * dt);
  for (int k = 0; k < 6; ++k) c.vb[k] = x[NV + k] + cs * (s == 0 ? 0.0 : c.ap[k]);
  for (int k = 0; k < 6; ++k) c.ap[k] = sh.ab[k];
}

// ---- one step of the way back: the lane stands on body i = path[t] with the composite of i's strict descendants in cmp; adds i, forms the
// three columns of joint i (emit(column, g6)), and steps up to the parent.  Lanes whose limb is shorter than t + 1 do nothing.
template <class Emit>
HSQP_HD void ql_back_step(const DevModel& dm, const QvConst& kc, const double* x, const double* u, int L, int s, double dt, int t, QlState& st, double* cmp,
                          const QlShared& sh, const double* csn, int csn_ld, const double* rP, const double* Ff, int foot_step, Emit&& emit) {
  if (t >= dm.limb_len[L]) return;
  const int i = (int)((dm.limb_path[L] >> (8 * t)) & 0xffull), j = i - 1;
  double qj, qd, qdd;
  ql_joint_inputs(x, u, j, s, dt, qj, qd, qdd);
  (void)qj;
  {
    double In[10], own[NCMP];
    ql_inertia(kc, i, st.R, st.r, In);
    QV_SCHED_FENCE();
    ql_body_comp(In, st.vl, st.al, own);
    for (int e = 0; e < NCMP; ++e) cmp[e] += own[e];
  }
  QV_SCHED_FENCE();
  double S[6], Sd[6], Sdd[6];
  m3_mulv(st.R, kc.axis[i], S);          // R_i axis_i = R_p (Rfix axis): the joint axis in world axes
  v3_cross(st.r, S, S + 3);
  mxm(st.vl, S, Sd);
  {
    double t1[6], t2[6];
    mxm(st.al, S, t1);
    mxm(st.vl, Sd, t2);
    for (int k = 0; k < 6; ++k) Sdd[k] = t1[k] + t2[k];
  }
  const bool mine = ((dm.limb_own[L] >> t) & 1u) != 0;
  {
    // d(F_ext moment)/dq: the lane's contact point moves with every joint above it
    double dext[3] = {0.0, 0.0, 0.0};
    const double zero3[3] = {0.0, 0.0, 0.0};
    if (t <= foot_step) {
      double d[3], dr[3];
      for (int k = 0; k < 3; ++k) d[k] = rP[k] - st.r[k];
      v3_cross(S, d, dr);
      v3_cross(dr, Ff, dext);
    }
    double g[6];
    ql_col_q(S, Sd, Sdd, cmp, sh, dext, zero3, g);
    if (mine) emit(3 + i + 2, g);
    QV_SCHED_FENCE();
    ql_col_qd(S, Sd, cmp, sh, g);
    if (mine) emit(NV + 3 + i + 2, g);
    ql_col_qdd(S, cmp, sh, g);
    if (mine) emit(NX + 12 + j, g);
  }
  QV_SCHED_FENCE();
  // unwind to the parent: v_p = v_i - S qd, a_p = a_i - S qdd - Sd qd, R_p = R_i Mq^T, r_p = r_i - R_p pfix
  for (int k = 0; k < 6; ++k) { st.al[k] -= S[k] * qdd + Sd[k] * qd; st.vl[k] -= S[k] * qd; }
  double Rq[9], Mq[9], Rp[9], t3[3];
  rot_axis_cs(kc.axis[i], csn[t * csn_ld], csn[t * csn_ld + 1], Rq);
  m3_mul(kc.Rfix[i], Rq, Mq);
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) Rp[3 * a + b] = st.R[3 * a] * Mq[3 * b] + st.R[3 * a + 1] * Mq[3 * b + 1] + st.R[3 * a + 2] * Mq[3 * b + 2];
  m3_mulv(Rp, kc.pfix[i], t3);
  for (int k = 0; k < 9; ++k) st.R[k] = Rp[k];
  for (int k = 0; k < 3; ++k) st.r[k] -= t3[k];
}

// ---- the base columns.  cmp: composite of the whole robot (every lane the same); dext_e[jc][3]: d(F_ext moment)/d(euler angle jc), summed
// over both feet.  The lanes share the work: lane L < 3 forms the q and qd columns of euler joint L; the lane that carries foot f its six
// wrench columns; lane 0 the zero columns.
template <class Emit>
HSQP_HD void ql_base_columns(const DevModel& dm, int L, const QlBaseKin& bk, const double* cmp, const QlShared& sh, const double* dext_e, const double* rP,
                             Emit&& emit) {
  if (L < 3) {
    const int jc = L;
    // (the lane's euler joint picked by selects: an array indexed by the lane would live in scratch memory)
    auto sel = [jc](double a0, double a1, double a2) { return jc == 0 ? a0 : (jc == 1 ? a1 : a2); };
    double S[6], Sdj[6], vlj[6], alj[6], dxt[3];
    for (int k = 0; k < 3; ++k) { S[k] = sel(bk.w[0][k], bk.w[1][k], bk.w[2][k]); S[3 + k] = 0.0; dxt[k] = sel(dext_e[k], dext_e[3 + k], dext_e[6 + k]); }
    for (int k = 0; k < 6; ++k) { Sdj[k] = sel(bk.Sd[0][k], bk.Sd[1][k], bk.Sd[2][k]); vlj[k] = sel(bk.vl[0][k], bk.vl[1][k], bk.vl[2][k]); alj[k] = sel(bk.al[0][k], bk.al[1][k], bk.al[2][k]); }
    double Sdd[6], t1[6], t2[6];
    mxm(alj, S, t1);
    mxm(vlj, Sdj, t2);
    for (int k = 0; k < 6; ++k) Sdd[k] = t1[k] + t2[k];
    // d(S_E)/dq_c a_ang = {w_c x sum_{e>c} w_e a_e, 0}
    double xtra[3], z[3], wz[3];
    for (int k = 0; k < 3; ++k) z[k] = (jc < 1 ? bk.w[1][k] * sh.ab[4] : 0.0) + (jc < 2 ? bk.w[2][k] * sh.ab[5] : 0.0);
    v3_cross(S, z, wz);
    sym3_mulv(cmp + CMP_I + 4, wz, xtra);
    double g[6];
    ql_col_q(S, Sdj, Sdd, cmp, sh, dxt, xtra, g);
    emit(3 + jc, g);
    ql_col_qd(S, Sdj, cmp, sh, g);
    emit(NV + 3 + jc, g);
  }
  for (int f = 0; f < 2; ++f) {
    if (dm.foot_limb[f] != L) continue;
    for (int k6 = 0; k6 < 6; ++k6) {
      double e[3] = {0.0, 0.0, 0.0}, rhs[3], nlin[3] = {0.0, 0.0, 0.0}, g[6];
      e[k6 % 3] = 1.0;
      if (k6 < 3) { v3_cross(rP, e, rhs); for (int k = 0; k < 3; ++k) nlin[k] = -e[k]; }
      else for (int k = 0; k < 3; ++k) rhs[k] = e[k];
      ql_finish(sh, rhs, nlin, g);
      emit(NX + 6 * f + k6, g);
    }
  }
  if (L == 0) {
    const double z6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int c = 0; c < 3; ++c) { emit(c, z6); emit(NV + c, z6); emit(NZ + c, z6); }
  }
}

// the step of the lane's limb on which its foot body sits (-1: the limb carries no foot)
HSQP_HD int ql_foot_step(const DevModel& dm, int L) {
  for (int f = 0; f < 2; ++f)
    if (dm.foot_limb[f] == L)
      for (int t = 0; t < dm.limb_len[L]; ++t)
        if ((int)((dm.limb_path[L] >> (8 * t)) & 0xffull) == dm.contact_body[f]) return t;
  return -1;
}

// ------------------------------------------------------------------------------------------------
// Node terms and the RK4 chain behind the limb lanes: one workgroup per node in the phase form of hsqp_node.h / hsqp_lq.h, its kinematic inputs
// LOADED from the image the limb lanes left in the record instead of evaluated in ~60 phases (stage_eval): seven barriers per node.
static_assert(KIN_SIZE == KIN_DOUBLES, "kinematics image: record layout");
struct KinWS : KinImg {
  double G[6][LDJ];                 // stage-1 Jacobian d a_b / d[x;u]
  unsigned char sub[NB];            // subtree sizes (supports())
};
struct LqbWS {
  KinWS st;
  double blk[3][2][6][6];           // RK4 chain: the blocks G_s[:, v_b] and G_s[:, q_b] of stages 2..4
  NodeWS nw;
  double as[4][6];                  // base accelerations of the stages
  double xnext[NX];
  double bvec[64];
};

// the rest of the LQ record of node (x, u, x_next, par): everything lq_node<true> writes except REC_GS (read here).  rec must hold REC_GS
// (transposed), REC_AS and REC_KIN of the node (k_lq_limb / ql_node_host).
HSQP_HD void lqb_node(const Ctx& ctx, const DevModel& dm, LqbWS& w, const double* x, const double* u, const double* xnext, const double* par, double dt,
                      double* rec) {
  double* misc = rec + REC_MISC;
  WG_FOR(ctx, i, NX + NU + NP + NX) {
    if (i < NX) w.nw.x[i] = x[i];
    else if (i < NX + NU) w.nw.u[i - NX] = u[i - NX];
    else if (i < NX + NU + NP) w.nw.par[i - NX - NU] = par[i - NX - NU];
    else w.xnext[i - NX - NU - NP] = xnext[i - NX - NU - NP];
  }
  {
    double* img = reinterpret_cast<double*>(static_cast<KinImg*>(&w.st));
    WG_FOR(ctx, i, KIN_SIZE) img[i] = rec[REC_KIN + i];
  }
  WG_FOR(ctx, i, 6 * LDJ) w.st.G[i % 6][i / 6] = rec[REC_GS + i];   // stage 1, stored [column][6]
  WG_FOR(ctx, i, NB + 24) {
    if (i < NB) w.st.sub[i] = (unsigned char)dm.subtree_size[i];
    else w.as[(i - NB) / 6][(i - NB) % 6] = rec[REC_AS + i - NB];
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 3);
  node_values(ctx, dm, w.st, w.nw);
  PH_TICK(ctx, 4);
  node_scalars(ctx, dm, w.st, w.nw);
  PH_TICK(ctx, 5);
  WG_FOR(ctx, i, 64) {   // (no barrier needed before the next phase: record writes only)
    double f = 0.0;
    if (i < NV) f = w.nw.x[NV + i];
    else if (i < NV + 6) f = w.st.ab[i - NV];
    else if (i < NX) f = w.nw.u[12 + i - NV - 6];
    rec[REC_FLOW + i] = f;
  }
  node_derivatives(ctx, dm, w.st, w.nw, dt, rec + REC_J, rec + REC_CDE, rec + REC_RHO);
  PH_TICK(ctx, 6);
  // ---- RK4 value: x_next = x + dt/6 (k1 + 2 k2 + 2 k3 + k4), defect (lq_node, hsqp_lq.h "RK4 value")
  WG_FOR(ctx, i, 64) {
    double b = 0.0;
    if (i < NV) {
      const double v0 = w.nw.x[NV + i];
      double a0, a1, a2;   // what moves the stage velocities: the previous stage's acceleration
      if (i < 6) { a0 = w.as[0][i]; a1 = w.as[1][i]; a2 = w.as[2][i]; }
      else a0 = a1 = a2 = w.nw.u[12 + i - 6];
      const double v1 = v0 + 0.5 * dt * a0, v2 = v0 + 0.5 * dt * a1, v3 = v0 + dt * a2;
      b = w.nw.x[i] + dt / 6.0 * (v0 + 2.0 * v1 + 2.0 * v2 + v3) - w.xnext[i];
    } else if (i < NX) {
      const int k = i - NV;
      const double a = k < 6 ? (w.as[0][k] + 2.0 * w.as[1][k] + 2.0 * w.as[2][k] + w.as[3][k]) / 6.0 : w.nw.u[12 + k - 6];
      b = w.nw.x[i] + dt * a - w.xnext[i];
    }
    rec[REC_B + i] = b;
    w.bvec[i] = b;
  }
  WG_SYNC(ctx);
  WG_FOR(ctx, it, 1) {
    double dyn = 0.0, eq = 0.0;
    for (int i = 0; i < NX; ++i) dyn += w.bvec[i] * w.bvec[i];
    for (int r = 0; r < w.nw.ne; ++r) eq += w.nw.eqv[r] * w.nw.eqv[r];
    misc[0] = (double)w.nw.ne;
    misc[1] = dt * node_cost(w.nw);
    misc[8] = (double)w.nw.nrows;
    misc[2] = dt * eq;
    misc[3] = (dt > 0.0 ? dt : 1.0) * dyn;
    misc[4] = (double)w.nw.contact[0]; misc[5] = (double)w.nw.contact[1];
    misc[6] = (double)w.nw.eq_off[0]; misc[7] = (double)w.nw.eq_off[1];
  }
  WG_FOR(ctx, i, 2 * LDJ + 3 * 72) {
    if (i < LDJ) rec[REC_D + i] = w.nw.d[i];
    else if (i < 2 * LDJ) rec[REC_GD + i - LDJ] = w.nw.gd[i - LDJ];
    else {
      const int e = i - 2 * LDJ, sg = e / 72, which = (e / 36) % 2, r = (e / 6) % 6, k = e % 6;
      w.blk[sg][which][r][k] = rec[REC_GS + ((sg + 1) * LDJ + (which == 0 ? NV : 0) + k) * GT_LD + r];
    }
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 7);
  WG_FOR(ctx, col, LDJ) lq_chain_column<true>(w.blk, rec + REC_GS, col, dt, rec);
  PH_TICK(ctx, 8);
}

#if !defined(__HIP_DEVICE_COMPILE__)
// The four lanes of a node one after the other, the quad exchanges in between (host builds: tests/hostemu).  GT[4][LDJ][GT_LD]: transposed
// stage Jacobians, as4[4][6]: base accelerations of the stages, kin: stage-1 kinematics image.
inline void ql_node_host(const DevModel& dm, const double* x, const double* u, double dt, double* GT, double* as4, KinImg* kin) {
  QvConst* kc = new QvConst;
  const Ctx ctx{0, 1, nullptr};
  qv_load_const(ctx, dm, *kc, [] {});
  QlCarry c[QV_LIMBS];
  for (int L = 0; L < QV_LIMBS; ++L) for (int k = 0; k < 6; ++k) { c[L].vb[k] = 0.0; c[L].ap[k] = 0.0; }
  for (int s = 0; s < 4; ++s) {
    QlBaseKin bk[QV_LIMBS];
    QlState st[QV_LIMBS];
    QlShared sh[QV_LIMBS];
    double part[QV_LIMBS][16], tot[16], csn[QV_LIMBS][QL_MAXLEN][2], rP[QV_LIMBS][3], Ff[QV_LIMBS][3], cmp[QV_LIMBS][NCMP];
    for (int L = 0; L < QV_LIMBS; ++L) {
      ql_base_kin(dm, x, s, dt, c[L], bk[L]);
      ql_forward(dm, *kc, x, u, L, s, dt, bk[L], st[L], part[L], &csn[L][0][0], 2, rP[L], Ff[L], s == 0 ? kin : nullptr);
    }
    for (int e = 0; e < 16; ++e) tot[e] = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
    for (int L = 0; L < QV_LIMBS; ++L) ql_base_solve(tot, bk[L], sh[L]);
    for (int k = 0; k < 6; ++k) as4[6 * s + k] = sh[0].ab[k];
    if (s == 0 && kin) {
      for (int e = 0; e < 3; ++e) for (int k = 0; k < 3; ++k) kin->E[3 * k + e] = bk[0].w[e][k];
      for (int k = 0; k < 3; ++k) kin->y[k] = sh[0].y[k];
      for (int k = 0; k < 6; ++k) kin->ab[k] = sh[0].ab[k];
    }
    double* G = GT + (size_t)s * LDJ * GT_LD;
    auto emit = [&](int col, const double* g) { for (int k = 0; k < 6; ++k) G[col * GT_LD + k] = g[k]; };
    for (int L = 0; L < QV_LIMBS; ++L) for (int e = 0; e < NCMP; ++e) cmp[L][e] = 0.0;
    for (int t = dm.limb_max_len - 1; t >= 0; --t) {
      double snap[QV_LIMBS][NCMP];
      for (int L = 0; L < QV_LIMBS; ++L) for (int e = 0; e < NCMP; ++e) snap[L][e] = cmp[L][e];
      for (int L = 0; L < QV_LIMBS; ++L)
        for (int k = 1; k < QV_LIMBS; ++k)
          if ((dm.limb_merge[t][L] >> k) & 1u) for (int e = 0; e < NCMP; ++e) cmp[L][e] += snap[L ^ k][e];
      for (int L = 0; L < QV_LIMBS; ++L)
        ql_back_step(dm, *kc, x, u, L, s, dt, t, st[L], cmp[L], sh[L], &csn[L][0][0], 2, rP[L], Ff[L], ql_foot_step(dm, L), emit);
    }
    // the base: the limbs that own their root-side body, plus the base body itself
    double ctot[NCMP], dext_e[9];
    {
      const double r0[3] = {0.0, 0.0, 0.0};
      double In[10], own[NCMP];
      ql_inertia(*kc, 0, bk[0].R, r0, In);
      ql_body_comp(In, bk[0].vl[2], bk[0].al[2], own);
      for (int e = 0; e < NCMP; ++e) {
        double sum = 0.0;
        for (int L = 0; L < QV_LIMBS; ++L) sum += (dm.limb_own[L] & 1u) ? cmp[L][e] : 0.0;
        ctot[e] = sum + own[e];
      }
    }
    for (int jc = 0; jc < 3; ++jc)
      for (int k = 0; k < 3; ++k) dext_e[3 * jc + k] = 0.0;
    for (int L = 0; L < QV_LIMBS; ++L) {
      if (ql_foot_step(dm, L) < 0) continue;
      for (int jc = 0; jc < 3; ++jc) {
        double dr[3], tt[3];
        v3_cross(bk[L].w[jc], rP[L], dr);
        v3_cross(dr, Ff[L], tt);
        for (int k = 0; k < 3; ++k) dext_e[3 * jc + k] += tt[k];
      }
    }
    for (int L = 0; L < QV_LIMBS; ++L) ql_base_columns(dm, L, bk[L], ctot, sh[L], dext_e, rP[L], emit);
    for (int L = 0; L < QV_LIMBS; ++L) ql_carry_advance(x, s, dt, sh[L], c[L]);
  }
  delete kc;
}
#endif

}  // namespace hsqp
