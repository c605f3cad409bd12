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
// (GT_LD = 6, hsqp_lq.h: the stage Jacobians travel TRANSPOSED, [stage][column][6] — a lane owns a column)

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
// the lane's limb, read from the model image ONCE (registers): these sit in front of every step's operand addresses
struct QlLimb { unsigned long long path; int len, max_len, foot_step; unsigned own; };
HSQP_HD QlLimb ql_limb(const DevModel& dm, int L) { return QlLimb{dm.limb_path[L], dm.limb_len[L], dm.limb_max_len, dm.limb_foot_step[L], dm.limb_own[L]}; }

// node data of RK4 stage 1 shared by the four lanes of a node (LDS); see "Node terms on the limb lanes" below
struct QlFoot {                      // per (node, foot), LDS
  double R[9], rP[3], vl[6], al[6];  // forward pass: rotation of the foot body, contact point relative to O, body velocity / trick acceleration
  double vP[3], alpha[3];            // pass A: linear velocity of the contact point, full angular acceleration of the frame (node_values)
  double scfm[8];                    // sqrt(dt) x scaling of the friction (4) and moment (4) rows; 0: not in contact
  double fshift;                     // friction cone: -p' hessianDiagonalShift
  double pad_;
};
struct QlNodeLds {
  QlFoot ft[2];
  double cp[10][3];                  // collision points relative to O
  double sccoll[16];                 // sqrt(dt) x scaling of the collision rows; 0: inactive
};
struct QlRows {                      // what a lane knows about its node's rows (registers)
  int own;                           // the lane's foot, -1: none
  int c0, c1, off1;                  // contact flags, first equality row of foot 1
  int coll;                          // collision rows in use
  double sdt;                        // sqrt(dt)
  double imp[2];                     // sdt x impact proximity scaler of the feet
};


// ---- per-body quantities (stage_eval: "per-body spatial inertia about O and net force")
template <class KC>
HSQP_HD void ql_inertia(const KC& kc, int i, const double* Rb, const double* r, double* In) {
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
// force of the lane's foot in rP / Ff.
template <class KC>
HSQP_HD void ql_forward(const DevModel& dm, const KC& kc, const QlLimb& lb, const double* x, const double* u, int L, int s, double dt, const QlBaseKin& bk, QlState& st,
                        double* part, double* csn, int csn_ld, double* rP, double* Ff) {
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
    }
  };
  if (L == 0) body(0, true);
  const unsigned long long path = lb.path;
  const int len = lb.len;
  const unsigned own = lb.own;
  for (int t = 0; t < lb.max_len; ++t) {
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
    QV_SCHED_FENCE();
    body(i, ((own >> t) & 1u) != 0);
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

// step up to the parent of body i: v_p = v_i - S qd, a_p = a_i - S qdd - Sd qd, R_p = R_i Mq^T, r_p = r_i - R_p pfix (cn, sn: cos / sin of joint i)
template <class KC>
HSQP_HD void ql_unwind(const KC& kc, int i, const double* S, const double* Sd, double qd, double qdd, double cn, double sn, QlState& st) {
  for (int k = 0; k < 6; ++k) { st.al[k] -= S[k] * qdd + Sd[k] * qd; st.vl[k] -= S[k] * qd; }
  double Rq[9], Mq[9], Rp[9], t3[3];
  rot_axis_cs(kc.axis[i], cn, sn, Rq);
  m3_mul(kc.Rfix[i], Rq, Mq);
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) Rp[3 * a + b] = st.R[3 * a] * Mq[3 * b] + st.R[3 * a + 1] * Mq[3 * b + 1] + st.R[3 * a + 2] * Mq[3 * b + 2];
  m3_mulv(Rp, kc.pfix[i], t3);
  for (int k = 0; k < 9; ++k) st.R[k] = Rp[k];
  for (int k = 0; k < 3; ++k) st.r[k] -= t3[k];
}

// ---- one step of the way back: the lane stands on body i = path[t] with the composite of i's strict descendants in cmp; adds i, forms the
// three columns of joint i (emit(column, g6)), and steps up to the parent.  Lanes whose limb is shorter than t + 1 do nothing.
template <class KC, class Emit>
HSQP_HD void ql_back_step(const DevModel& dm, const KC& kc, const QlLimb& lb, const double* x, const double* u, int L, int s, double dt, int t, QlState& st, double* cmp,
                          const QlShared& sh, const double* csn, int csn_ld, const double* rP, const double* Ff, Emit&& emit) {
  if (t >= lb.len) return;
  const int foot_step = lb.foot_step;
  const int i = (int)((lb.path >> (8 * t)) & 0xffull), j = i - 1;
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
  const bool mine = ((lb.own >> t) & 1u) != 0;
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
    QV_SCHED_FENCE();
    ql_col_qdd(S, cmp, sh, g);
    if (mine) emit(NX + 12 + j, g);
  }
  QV_SCHED_FENCE();
  ql_unwind(kc, i, S, Sd, qd, qdd, csn[t * csn_ld], csn[t * csn_ld + 1], st);
}

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

// the step of the lane's limb on which its foot body sits (-1: the limb carries no foot): DevModel::limb_foot_step
HSQP_HD int ql_foot_step(const DevModel& dm, int L) { return dm.limb_foot_step[L]; }

// ------------------------------------------------------------------------------------------------
// Node terms on the limb lanes (RK4 stage 1): the values and penalties of hsqp_node.h's node_values / node_scalars dealt to the four lanes of a
// node (as hsqp_lqv.h does for the value pass), and the residual / equality ROWS of every column formed by the lane that owns the column, at the
// moment its stage-1 Jacobian column exists — the joint's motion axis, the body state and the composite are in registers then.  The record's
// rows are stored TRANSPOSED (REC_LAYOUT = 1): REC_J [column][NRS], REC_CDE [column][CDE_ROWS], so that a lane writes contiguous, 16-byte
// aligned pieces; the row slots are fixed:
constexpr int ROWQ_FOOT = 0;    // + 16 f: the 15 task-space rows of foot f (ori, vlin, vang, alin, aang) and one zero row
constexpr int ROWQ_FM = 32;     // + 8 f : friction cone (4) and contact moment (4) rows of foot f — zero while the foot is in the air
constexpr int ROWQ_COLL = 48;   // the 16 foot-collision rows, in use (REC_NROWS = 64) only while one of them is active
static_assert(ROWQ_COLL + 16 == NRS && ROWQ_FM + 16 == ROWQ_COLL, "row slots");
// Entries that are zero for every state (a column that can never move a row: an arm joint and the friction rows, ...) are never written: the
// record is zero-filled when it is allocated (hsqp_create) and only this kernel writes these regions.

#ifndef HSQP_QL_MERGED_STORES
#define HSQP_QL_MERGED_STORES 1   /* ql_put_foot: full and half blocks stored by the same wave instructions (0: one branch each; A/B builds) */
#endif
HSQP_HD void ql_st2(double* p, double a, double b) {
#if defined(HSQP_EXP_FEWSTORES)   /* timing experiment only (wrong results): one store in eight reaches memory */
  if ((reinterpret_cast<unsigned long long>(p) >> 4) & 7ull) return;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
  *reinterpret_cast<double2*>(p) = make_double2(a, b);
#else
  p[0] = a; p[1] = b;
#endif
}

// what the node terms need of stage 1's base solve, from the record: ab = REC_AS[0..5], y = E a_ang
HSQP_HD void ql_shared_from_record(const QlBaseKin& bk, const double* rec, QlShared& sh) {
  for (int k = 0; k < 6; ++k) sh.ab[k] = rec[REC_AS + k];
  for (int k = 0; k < 3; ++k) sh.y[k] = bk.w[0][k] * sh.ab[3] + bk.w[1][k] * sh.ab[4] + bk.w[2][k] * sh.ab[5];
  for (int k = 0; k < 9; ++k) sh.M[k] = 0.0;
  sh.minv = 0.0;
}

// pass A, after the base solve of stage 1: the lane's foot (frame values, penalties, rho, equality values) and its share of the cost terms.
HSQP_HD void ql_terms_a(const DevModel& dm, const double* x, const double* u, const double* par, int L, double dt, const QlShared& sh, QlNodeLds& nl,
                        double* rec, bool live, double& cost, double& eqs) {
  const int c0 = par[HSQP_P_CONTACT] > 0.5, c1 = par[HSQP_P_CONTACT + 1] > 0.5;
  const double sdt = sqrt(dt);
  double cst = 0.0, eq = 0.0;
  // StateInputQuadraticCost / joint limits: as qv_node_terms (hsqp_lqv.h), the d / gd entries follow in pass B
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
    } else {
      const int i = t - NX;
      double un = 0.0;
      if ((i == 2 && c0) || (i == 8 && c1)) un = dm.total_mass * 9.81 / (c0 + c1);
      const double duu = u[i] - un;
      cst += 0.5 * dm.R[i] * duu * duu;
    }
  }
  for (int j = L; j < NJ; j += QV_LIMBS) {
    cst += pwp_barrier(dm.jl_bmu, dm.jl_bdelta, x[6 + j] - dm.q_lo[j]).p;
    cst += pwp_barrier(dm.jl_bmu, dm.jl_bdelta, dm.q_hi[j] - x[6 + j]).p;
  }
  for (int f = 0; f < 2; ++f) {
    if (dm.foot_limb[f] != L) continue;
    QlFoot& ft = nl.ft[f];
    const int cf = f == 0 ? c0 : c1;
    double a[6], o[18], t[3], t2[3];
    #pragma unroll
    for (int k = 0; k < 3; ++k) { a[k] = ft.al[k] + sh.y[k]; a[3 + k] = ft.al[3 + k] + sh.ab[k]; }
    a[5] -= dm.gravity;
    #pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = x[k] + ft.rP[k];
    ori_error(ft.R, o + 3);
    v3_cross(ft.vl, ft.rP, t);
    #pragma unroll
    for (int k = 0; k < 3; ++k) { o[6 + k] = ft.vl[3 + k] + t[k]; o[9 + k] = ft.vl[k]; }
    v3_cross(a, ft.rP, t);
    v3_cross(ft.vl, o + 6, t2);
    #pragma unroll
    for (int k = 0; k < 3; ++k) { o[12 + k] = a[3 + k] + t[k] + t2[k]; o[15 + k] = a[k]; }
    #pragma unroll
    for (int k = 0; k < 3; ++k) { ft.vP[k] = o[6 + k]; ft.alpha[k] = o[15 + k]; }
    // EndEffectorDynamicsFootCost.cpp:91-124
    double* rho = rec + REC_RHO;
    #pragma unroll
    for (int k = 0; k < 15; ++k) {
      const double r = dm.foot_sqrt_w[3 + k] * par[HSQP_P_IMPACT + f] * o[3 + k];
      cst += 0.5 * r * r;
      if (live) rho[ROWQ_FOOT + 16 * f + k] = sdt * r;
    }
    if (live) rho[ROWQ_FOOT + 16 * f + 15] = 0.0;
    double scfm[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, rfm[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, shift = 0.0, e[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (cf) {
      // FrictionForceConeConstraint.cpp:153-224 (node_scalars, hsqp_node.h:191-205)
      const double Fx = u[6 * f], Fy = u[6 * f + 1], Fz = u[6 * f + 2];
      const double T2 = Fx * Fx + Fy * Fy + dm.friction_reg, T1 = sqrt(T2), T3 = T2 * T1;
      const Pen3 p = relaxed_barrier(dm.friction_bmu, dm.friction_bdelta, dm.friction_mu * (Fz + dm.friction_grip) - T1);
      scfm[0] = sqrt(p.d2); rfm[0] = p.d1 / scfm[0];
      scfm[1] = scfm[2] = sqrt(-p.d1 * dm.friction_reg / T3);
      scfm[3] = sqrt(-p.d1 / T3);
      shift = -p.d1 * dm.friction_hess_shift;
      cst += p.p;
      // ContactMomentXYConstraintCppAd.cpp:87-104
      double lf[3], lm[3];
      m3_tmulv(ft.R, u + 6 * f, lf);
      m3_tmulv(ft.R, u + 6 * f + 3, lm);
      const double hm[4] = {lm[0] - dm.rect_y_min * lf[2], -lm[0] + dm.rect_y_max * lf[2], -lm[1] - dm.rect_x_min * lf[2], lm[1] + dm.rect_x_max * lf[2]};
      #pragma unroll
      for (int k = 0; k < 4; ++k) {
        const Pen3 pm = relaxed_barrier(dm.moment_bmu, dm.moment_bdelta, hm[k]);
        scfm[4 + k] = sqrt(pm.d2); rfm[4 + k] = pm.d1 / scfm[4 + k];
        cst += pm.p;
      }
      // EndEffectorDynamicsAccelerationsConstraint.cpp:82-103, gains WBMpcInterface.cpp:205-229
      #pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int cc = k % 3;
        const double gp = k < 2 ? 0.0 : (k == 2 ? dm.gain_pos_z : dm.gain_ori);
        const double gv = k < 2 ? dm.gain_linvel_xy : (k == 2 ? dm.gain_linvel_z : dm.gain_angvel);
        const double ga = k < 2 ? dm.gain_linacc_xy : (k == 2 ? dm.gain_linacc_z : dm.gain_angacc);
        e[k] = k < 3 ? gp * o[cc] + gv * o[6 + cc] + ga * o[12 + cc] : gp * o[3 + cc] + gv * o[9 + cc] + ga * o[15 + cc];
      }
    } else {
      // ZeroWrenchConstraint.cpp:59-84, EndEffectorDynamicsLinearAccConstraint.cpp:69-83 (config WBMpcPreComputation.cpp:91-104)
      #pragma unroll
      for (int k = 0; k < 6; ++k) e[k] = u[6 * f + k];
      const double* sw = par + HSQP_P_SWING + 3 * f;
      e[6] = -dm.gain_linvel_z * sw[1] - dm.gain_linacc_z * sw[2] - dm.gain_pos_z * sw[0] + dm.gain_pos_z * o[2] + dm.gain_linvel_z * o[8] + dm.gain_linacc_z * o[14];
    }
    #pragma unroll
    for (int k = 0; k < 7; ++k) eq += e[k] * e[k];
    if (live) {   // the foot's part of the column of the equality values; the second foot also clears the rows behind the last one
      const int n0 = c0 ? 6 : 7, r0 = f == 0 ? 0 : n0, nf = cf ? 6 : 7;
      double* ce = rec + REC_CDE + NZ * CDE_ROWS;
#pragma unroll
      for (int k = 0; k < 7; ++k) if (k < nf) ce[r0 + k] = e[k];   // (fixed trip count: e stays in registers)
      if (f == 1) for (int k = r0 + nf; k < CDE_ROWS; ++k) ce[k] = 0.0;
    }
    #pragma unroll
    for (int k = 0; k < 8; ++k) { ft.scfm[k] = sdt * scfm[k]; if (live) rho[ROWQ_FM + 8 * f + k] = sdt * rfm[k]; }
    ft.fshift = shift;
  }
  cost = cst;
  eqs = eq;
}

// pass B (every lane's pass A of the node is visible): collision rows, d / gd, the equality values' column, flow; returns the lane's share of the
// collision cost and (any) whether one of the lane's collision rows is active
HSQP_HD void ql_terms_b(const DevModel& dm, const double* x, const double* u, const double* par, int L, double dt, const QlShared& sh, QlNodeLds& nl,
                        double* rec, bool live, double& cost, int& any) {
  const int c0 = par[HSQP_P_CONTACT] > 0.5, c1 = par[HSQP_P_CONTACT + 1] > 0.5;
  const double sdt = sqrt(dt);
  double cst = 0.0;
  int act = 0;
  // foot collision distances (FootCollisionConstraint.cpp:118-141; node_values / node_scalars)
  for (int rw = L; rw < 16; rw += QV_LIMBS) {
    double sc = 0.0, rho = 0.0;
    if (!(c0 && c1)) {
      int a, b;
      coll_pair(rw, a, b);
      double dd[3];
      for (int k = 0; k < 3; ++k) dd[k] = nl.cp[a][k] - nl.cp[b][k];
      const double h = sqrt(v3_dot(dd, dd)) - 2.0 * (rw == 9 ? dm.r_knee : dm.r_foot);
      const Pen3 p = pwp_barrier(dm.coll_bmu, dm.coll_bdelta, h);
      if (p.d2 > 0.0) { sc = sqrt(p.d2); rho = p.d1 / sc; act = 1; }
      cst += p.p;
    }
    nl.sccoll[rw] = sdt * sc;
    if (live) rec[REC_RHO + ROWQ_COLL + rw] = sdt * rho;
  }
  // diagonal part of the cost model (node_derivatives "diagonal part", hsqp_node.h:500-515)
  const double yaw = x[3];
  const double vloc = cos(yaw) * par[HSQP_P_XDES + NV] + sin(yaw) * par[HSQP_P_XDES + NV + 1];
  const double gcf = par[HSQP_P_ARMSWING] * vloc;
  const double fs = nl.ft[0].fshift + nl.ft[1].fshift;
  for (int i = L; i < LDJ; i += QV_LIMBS) {
    double d = 0.0, g = 0.0;
    if (i < NX) {
      double xn = par[HSQP_P_XDES + i];
      const int j = i - 6;
      if (j == dm.arm_swing_joint[0]) xn += -0.15 * gcf;
      if (j == dm.arm_swing_joint[1]) xn += 0.15 * gcf;
      if (j == dm.arm_swing_joint[2]) xn += -0.15 * gcf;
      if (j == dm.arm_swing_joint[3]) xn += 0.15 * gcf;
      d = dm.Q[i]; g = dm.Q[i] * (x[i] - xn);
    } else if (i < NZ) {
      const int k = i - NX;
      double un = 0.0;
      if ((k == 2 && c0) || (k == 8 && c1)) un = dm.total_mass * 9.81 / (c0 + c1);
      d = dm.R[k]; g = dm.R[k] * (u[k] - un);
    }
    if (i >= 6 && i < NV) {
      const int j = i - 6;
      const Pen3 lo = pwp_barrier(dm.jl_bmu, dm.jl_bdelta, x[i] - dm.q_lo[j]);
      const Pen3 hi = pwp_barrier(dm.jl_bmu, dm.jl_bdelta, dm.q_hi[j] - x[i]);
      d += lo.d2 + hi.d2;
      g += lo.d1 - hi.d1;
    }
    if (i < NZ) d += fs;
    if (live) { rec[REC_D + i] = dt * d; rec[REC_GD + i] = dt * g; }
  }
  if (live) {
    for (int i = L; i < 64; i += QV_LIMBS) {
      double f = 0.0;
      if (i < NV) f = x[NV + i];
      else if (i < NV + 6) continue;   // (the base acceleration: below, with compile-time indices — sh indexed by i would live in scratch memory)
      else if (i < NX) f = u[12 + i - NV - 6];
      rec[REC_FLOW + i] = f;
    }
    if (L == 1) {
#pragma unroll
      for (int k = 0; k < 6; ++k) rec[REC_FLOW + NV + k] = sh.ab[k];
    }
  }
  cost = cst;
  any = act;
}

// what a lane needs to know about its node's rows (after pass B)
HSQP_HD void ql_rows_setup(const DevModel& dm, const double* par, int L, double dt, int coll, QlRows& rw) {
  rw.c0 = par[HSQP_P_CONTACT] > 0.5; rw.c1 = par[HSQP_P_CONTACT + 1] > 0.5;
  rw.off1 = rw.c0 ? 6 : 7;
  rw.own = dm.foot_limb[0] == L ? 0 : (dm.foot_limb[1] == L ? 1 : -1);
  rw.coll = coll;
  rw.sdt = sqrt(dt);
  rw.imp[0] = rw.sdt * par[HSQP_P_IMPACT]; rw.imp[1] = rw.sdt * par[HSQP_P_IMPACT + 1];
}
// misc of the record (lq_node's layout; [3], the defect, is the chain kernel's)
HSQP_HD void ql_write_misc(const double* par, double dt, double cost, double eq, int coll, double* misc) {
  const int c0 = par[HSQP_P_CONTACT] > 0.5, c1 = par[HSQP_P_CONTACT + 1] > 0.5;
  const int off1 = c0 ? 6 : 7;
  misc[0] = (double)(off1 + (c1 ? 6 : 7));
  misc[1] = dt * cost;
  misc[2] = dt * eq;
  misc[4] = (double)c0; misc[5] = (double)c1;
  misc[6] = 0.0; misc[7] = (double)off1;
  misc[8] = coll ? (double)NRS : (double)ROWQ_COLL;   // REC_NROWS
  misc[9] = 1.0;                                      // REC_LAYOUT: transposed
}

// The RK4 value and the defect b = Phi(x, u) - x_next of a node on its four lanes: lane L forms the entries i = 4 j + L (the formulas of lq_chain_node below / lq_node, hsqp_lq.h
// "RK4 value"; as = REC_AS, the base accelerations of the four stages from k_lq_limb) and returns its share of |b|^2.  What k_lq_rows runs when the chain is fused into
// k_project (k_lq_chain is not launched then); entries NX .. 63 of REC_B are written as zeros.
HSQP_HD double ql_defect_lane(const double* x, const double* u, const double* xnext, const double* as, double dt, int L, double* rec, bool live) {
  double sq = 0.0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int i = 4 * j + L;
    double b = 0.0;
    if (i < NV) {
      const double v0 = x[NV + i];
      double a0, a1, a2;   // what moves the stage velocities: the previous stage's acceleration
      if (i < 6) { a0 = as[i]; a1 = as[6 + i]; a2 = as[12 + i]; }
      else a0 = a1 = a2 = u[12 + i - 6];
      const double v1 = v0 + 0.5 * dt * a0, v2 = v0 + 0.5 * dt * a1, v3 = v0 + dt * a2;
      b = x[i] + dt / 6.0 * (v0 + 2.0 * v1 + 2.0 * v2 + v3) - xnext[i];
    } else if (i < NX) {
      const int k = i - NV;
      const double a = k < 6 ? (as[k] + 2.0 * as[6 + k] + 2.0 * as[12 + k] + as[18 + k]) / 6.0 : u[12 + k - 6];
      b = x[i] + dt * a - xnext[i];
    }
    if (live) rec[REC_B + i] = b;
    sq += b * b;
  }
  return sq;
}

// d{pos, ori, vlin, vang, alin, aang}/dz of a contact frame from the partials of its body's spatial velocity dv, spatial acceleration da (own part +
// base chain) and of the contact point drP (foot_column "frame level", hsqp_node.h:342-365); rot: the column rotates the frame about wax
HSQP_HD void ql_frame_rows(const QlFoot& ft, const double* dv, const double* da, const double* drP, bool rot, const double* wax, double* out) {
  const double* om = ft.vl;
  const double* al = ft.alpha;
  const double* vP = ft.vP;
  double t1[3], t2[3], t3[3], t4[3], dvP[3];
  for (int k = 0; k < 3; ++k) out[k] = drP[k];
  if (rot) {
    const double a[3] = {ft.R[2], ft.R[5], ft.R[8]};
    double dvec[3];
    v3_cross(wax, a, dvec);
    ori_error_d(ft.R, dvec, out + 3);
  } else {
    out[3] = out[4] = out[5] = 0.0;
  }
  v3_cross(dv, ft.rP, t1);
  v3_cross(om, drP, t2);
  for (int k = 0; k < 3; ++k) { dvP[k] = dv[3 + k] + t1[k] + t2[k]; out[6 + k] = dvP[k]; out[9 + k] = dv[k]; }
  v3_cross(da, ft.rP, t1);
  v3_cross(al, drP, t2);
  v3_cross(dv, vP, t3);
  v3_cross(om, dvP, t4);
  for (int k = 0; k < 3; ++k) { out[12 + k] = da[3 + k] + t1[k] + t2[k] + t3[k] + t4[k]; out[15 + k] = da[k]; }
}
// ... when only the base acceleration moves the frame (dab = {E g[3:6], g[0:3]}): the acceleration rows
HSQP_HD void ql_frame_rows_base(const QlFoot& ft, const double* dab, double* out) {
  double t1[3];
  for (int k = 0; k < 12; ++k) out[k] = 0.0;
  v3_cross(dab, ft.rP, t1);
  for (int k = 0; k < 3; ++k) { out[12 + k] = dab[3 + k] + t1[k]; out[15 + k] = dab[k]; }
}

// the task-space rows of foot f in column `col` (full: all fifteen, else the six acceleration rows) and the foot's equality rows ef[7]
// zpairs: bit j set = the row pair (2 j, 2 j + 1) of a FULL block is zero for every state in this column (a d/dqd column moves no orientation row: bit 0; a d/dqdd column
// only the acceleration rows: bits 0 .. 3; a base linear velocity neither orientation nor angular velocity: bits 0, 3) — not stored, like every other structural zero of
// the record (profiles/r06_lq_store_diet.txt).  Wave-uniform at every call site.
HSQP_HD void ql_put_foot(const DevModel& dm, const QlRows& rw, int f, bool full, const double* out, int col, double* rec, bool live, double* ef, int unit_row, unsigned zpairs = 0u) {
  if (live) {
    double* jt = rec + REC_J + col * NRS + ROWQ_FOOT + 16 * f;
    const double sc = rw.imp[f];
    const double* w = dm.foot_sqrt_w + 3;
#if HSQP_QL_MERGED_STORES
    // ONE store instruction per row pair for the lanes that write the whole block and those that write its acceleration half (rows 8 .. 15; their out[11] is zero):
    // the two groups used to run their stores one after the other, twelve wave instructions per foot and column instead of eight
#pragma unroll
    for (int k = 0; k < 16; k += 2)
      if (!((zpairs >> (k / 2)) & 1u) || k >= 8) { if (full || k >= 8) ql_st2(jt + k, sc * w[k] * out[3 + k], k < 14 ? sc * w[k + 1] * out[4 + k] : 0.0); }
#else
    if (full) {
#pragma unroll
      for (int k = 0; k < 14; k += 2) if (!((zpairs >> (k / 2)) & 1u)) ql_st2(jt + k, sc * w[k] * out[3 + k], sc * w[k + 1] * out[4 + k]);
      ql_st2(jt + 14, sc * w[14] * out[17], 0.0);
    } else {
      ql_st2(jt + 8, 0.0, sc * w[9] * out[12]);
      ql_st2(jt + 10, sc * w[10] * out[13], sc * w[11] * out[14]);
      ql_st2(jt + 12, sc * w[12] * out[15], sc * w[13] * out[16]);
      ql_st2(jt + 14, sc * w[14] * out[17], 0.0);
    }
#endif
  }
  const int cf = f == 0 ? rw.c0 : rw.c1;
  if (cf) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int c = k % 3;
      const double gp = k < 2 ? 0.0 : (k == 2 ? dm.gain_pos_z : dm.gain_ori);
      const double gv = k < 2 ? dm.gain_linvel_xy : (k == 2 ? dm.gain_linvel_z : dm.gain_angvel);
      const double ga = k < 2 ? dm.gain_linacc_xy : (k == 2 ? dm.gain_linacc_z : dm.gain_angacc);
      ef[k] = k < 3 ? gp * out[c] + gv * out[6 + c] + ga * out[12 + c] : gp * out[3 + c] + gv * out[9 + c] + ga * out[15 + c];
    }
    ef[6] = 0.0;
  } else {
#pragma unroll
    for (int k = 0; k < 6; ++k) ef[k] = k == unit_row ? 1.0 : 0.0;   // ZeroWrench: unit rows on the foot's own wrench columns
    ef[6] = dm.gain_pos_z * out[2] + dm.gain_linvel_z * out[8] + dm.gain_linacc_z * out[14];
  }
}
// the equality rows of a column from the two feet's rows
HSQP_HD void ql_put_cde(const QlRows& rw, const double* e0, const double* e1, int col, double* rec, bool live) {
  if (!live) return;
  double* ct = rec + REC_CDE + col * CDE_ROWS;
  const bool c0 = rw.c0 != 0;
  ql_st2(ct + 0, e0[0], e0[1]);
  ql_st2(ct + 2, e0[2], e0[3]);
  ql_st2(ct + 4, e0[4], e0[5]);
  ql_st2(ct + 6, c0 ? e1[0] : e0[6], c0 ? e1[1] : e1[0]);
  ql_st2(ct + 8, c0 ? e1[2] : e1[1], c0 ? e1[3] : e1[2]);
  ql_st2(ct + 10, c0 ? e1[4] : e1[3], c0 ? e1[5] : e1[4]);
  ql_st2(ct + 12, c0 ? e1[6] : e1[5], c0 ? 0.0 : e1[6]);
  // (rows 14, 15 of a column's slot are padding beyond NE_MAX: never read, zero since hsqp_create — no store: LQ kernels 0.759 -> 0.733 ms, profiles/r06_lq_store_diet.txt)
}
// friction / moment rows of foot f in a column that rotates the foot frame about w (v[0..3] = 0) or that is a wrench component (cu = 0..5)
HSQP_HD void ql_put_fm(const DevModel& dm, const QlFoot& ft, int f, const double* u, const double* w, int cu, int col, double* rec, bool live) {
  double v[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, dlf[3] = {0.0, 0.0, 0.0}, dlm[3] = {0.0, 0.0, 0.0};
  if (w) {
    double t[3];
    v3_cross(u + 6 * f, w, t);
    m3_tmulv(ft.R, t, dlf);
    v3_cross(u + 6 * f + 3, w, t);
    m3_tmulv(ft.R, t, dlm);
  } else if (cu < 3) {
    // FrictionForceConeConstraint.cpp:153-178 (first and second derivative of the cone)
    const double Fx = u[6 * f], Fy = u[6 * f + 1];
    const double Tn = sqrt(Fx * Fx + Fy * Fy + dm.friction_reg);
    v[0] = cu == 0 ? -Fx / Tn : (cu == 1 ? -Fy / Tn : dm.friction_mu);
    v[1] = cu == 0 ? 1.0 : 0.0;
    v[2] = cu == 1 ? 1.0 : 0.0;
    v[3] = cu == 0 ? Fy : (cu == 1 ? -Fx : 0.0);
    for (int r = 0; r < 3; ++r) dlf[r] = ft.R[3 * cu + r];
  } else {
    for (int r = 0; r < 3; ++r) dlm[r] = ft.R[3 * (cu - 3) + r];
  }
  v[4] = dlm[0] - dm.rect_y_min * dlf[2];
  v[5] = -dlm[0] + dm.rect_y_max * dlf[2];
  v[6] = -dlm[1] - dm.rect_x_min * dlf[2];
  v[7] = dlm[1] + dm.rect_x_max * dlf[2];
  if (!live) return;
  double* jt = rec + REC_J + col * NRS + ROWQ_FM + 8 * f;
#pragma unroll
  for (int k = 0; k < 8; k += 2) ql_st2(jt + k, ft.scfm[k] * v[k], ft.scfm[k + 1] * v[k + 1]);
}
// collision rows of a column that rotates the bodies [b_lo, b_hi) about the axis w through the point rc (relative to O)
HSQP_HD void ql_put_coll(const DevModel& dm, const QlNodeLds& nl, const double* w, const double* rc, int b_lo, int b_hi, int col, double* rec, bool live) {
  double* jt = rec + REC_J + col * NRS + ROWQ_COLL;
  for (int r = 0; r < 16; ++r) {
    int a, b;
    coll_pair(r, a, b);
    double val = 0.0;
    const double sc = nl.sccoll[r];
    if (sc != 0.0) {
      double dd[3], da[3] = {0.0, 0.0, 0.0}, db[3] = {0.0, 0.0, 0.0}, pc[3];
      for (int k = 0; k < 3; ++k) dd[k] = nl.cp[a][k] - nl.cp[b][k];
      if (dm.coll_body[a] >= b_lo && dm.coll_body[a] < b_hi) { for (int k = 0; k < 3; ++k) pc[k] = nl.cp[a][k] - rc[k]; v3_cross(w, pc, da); }
      if (dm.coll_body[b] >= b_lo && dm.coll_body[b] < b_hi) { for (int k = 0; k < 3; ++k) pc[k] = nl.cp[b][k] - rc[k]; v3_cross(w, pc, db); }
      const double n = sqrt(v3_dot(dd, dd));
      val = sc * (dd[0] * (da[0] - db[0]) + dd[1] * (da[1] - db[1]) + dd[2] * (da[2] - db[2])) / n;
    }
    if (live) jt[r] = val;
  }
}

// The rows of one JOINT column (kind 0 / 1 / 2: d/dq, d/dqd, d/dqdd of joint i = body index, standing state st) of RK4 stage 1; g: the
// column of the stage Jacobian, wE: the euler-rate axes (E g[3:6] = the column's angular base acceleration), sup: the joint moves the lane's foot.
// kind is a run-time value on purpose: ONE copy of this code serves the three columns of a joint (as three template instances the kernel
// was 15 k instructions = 120 KB, twice the instruction cache).
HSQP_HD void ql_rows_joint(const DevModel& dm, const QlRows& rw, const QlNodeLds& nl, const double* u, int kind, bool sup, int i, const double* S, const double* Sd,
                           const QlState& st, const double (*wE)[3], const double* g, int col, double* rec, bool live) {
  double dab[6], ef[2][7], out[2][18];
  for (int k = 0; k < 3; ++k) { dab[k] = wE[0][k] * g[3] + wE[1][k] * g[4] + wE[2][k] * g[5]; dab[3 + k] = g[k]; }
  // every foot: the base acceleration moves its frame
#pragma unroll
  for (int f = 0; f < 2; ++f) ql_frame_rows_base(nl.ft[f], dab, out[f]);
  const bool full = sup && rw.own >= 0;
  if (full) {   // the lane's own foot hangs below the joint
    const QlFoot& ft = nl.ft[rw.own];
    double dv[6], da[6], drP[3] = {0.0, 0.0, 0.0}, o[18];
    if (kind == 0) {
      double dvv[6], daa[6], t[6], pc[3];
      for (int k = 0; k < 6; ++k) { dvv[k] = ft.vl[k] - st.vl[k]; daa[k] = ft.al[k] - st.al[k]; }
      mxm(S, dvv, dv);
      mxm(S, daa, da);
      mxm(Sd, dvv, t);
      for (int k = 0; k < 6; ++k) da[k] += t[k];
      for (int k = 0; k < 3; ++k) pc[k] = ft.rP[k] - st.r[k];
      v3_cross(S, pc, drP);
    } else if (kind == 1) {
      double t[6];
      mxm(ft.vl, S, t);
      for (int k = 0; k < 6; ++k) { dv[k] = S[k]; da[k] = 2.0 * Sd[k] - t[k]; }
    } else {
      for (int k = 0; k < 6; ++k) { dv[k] = 0.0; da[k] = S[k]; }
    }
    for (int k = 0; k < 6; ++k) da[k] += dab[k];
    ql_frame_rows(ft, dv, da, drP, kind == 0, S, o);
#pragma unroll
    for (int k = 0; k < 18; ++k) { if (rw.own == 0) out[0][k] = o[k]; else out[1][k] = o[k]; }
  }
#pragma unroll
  for (int f = 0; f < 2; ++f) ql_put_foot(dm, rw, f, full && f == rw.own, out[f], col, rec, live, ef[f], -1, kind == 0 ? 0u : (kind == 1 ? 1u : 0xFu));
  ql_put_cde(rw, ef[0], ef[1], col, rec, live);
  if (kind == 0 && full) ql_put_fm(dm, nl.ft[rw.own], rw.own, u, S, 0, col, rec, live);
  if (kind == 0 && rw.coll) ql_put_coll(dm, nl, S, st.r, i, i + dm.subtree_size[i], col, rec, live);
}

// The rows of the BASE columns of stage 1, column by column as ql_base_columns forms them.
// q (kind 0) or qd (kind 1) column of euler joint jc: both feet hang below it
HSQP_HD void ql_rows_euler(const DevModel& dm, const QlRows& rw, const QlNodeLds& nl, const double* u, int kind, int jc, const QlBaseKin& bk, const QlShared& sh,
                           const double* g, double* rec, bool live) {
  auto sel = [jc](double a0, double a1, double a2) { return jc == 0 ? a0 : (jc == 1 ? a1 : a2); };
  double S[6], Sdj[6], vlj[6], alj[6];
  for (int k = 0; k < 3; ++k) { S[k] = sel(bk.w[0][k], bk.w[1][k], bk.w[2][k]); S[3 + k] = 0.0; }
  for (int k = 0; k < 6; ++k) { Sdj[k] = sel(bk.Sd[0][k], bk.Sd[1][k], bk.Sd[2][k]); vlj[k] = sel(bk.vl[0][k], bk.vl[1][k], bk.vl[2][k]); alj[k] = sel(bk.al[0][k], bk.al[1][k], bk.al[2][k]); }
  // euler links z, y do not carry the later euler accelerations: add {sum_{e>jc} w_e a_e, 0}
  double zz[3];
  for (int k = 0; k < 3; ++k) zz[k] = (jc < 1 ? bk.w[1][k] * sh.ab[4] : 0.0) + (jc < 2 ? bk.w[2][k] * sh.ab[5] : 0.0);
  const double r0[3] = {0.0, 0.0, 0.0};
  const int col = (kind == 0 ? 3 : NV + 3) + jc;
  double dab[6], ef[2][7];
  for (int k = 0; k < 3; ++k) { dab[k] = bk.w[0][k] * g[3] + bk.w[1][k] * g[4] + bk.w[2][k] * g[5]; dab[3 + k] = g[k]; }
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const QlFoot& ft = nl.ft[f];
    double dv[6], da[6], drP[3] = {0.0, 0.0, 0.0}, out[18];
    if (kind == 0) {
      double dvv[6], daa[6], t[6];
      for (int k = 0; k < 6; ++k) { dvv[k] = ft.vl[k] - vlj[k]; daa[k] = ft.al[k] - alj[k]; }
      for (int k = 0; k < 3; ++k) daa[k] += zz[k];
      mxm(S, dvv, dv);
      mxm(S, daa, da);
      mxm(Sdj, dvv, t);
      for (int k = 0; k < 6; ++k) da[k] += t[k];
      v3_cross(S, ft.rP, drP);
    } else {
      double t[6];
      mxm(ft.vl, S, t);
      for (int k = 0; k < 6; ++k) { dv[k] = S[k]; da[k] = 2.0 * Sdj[k] - t[k]; }
    }
    for (int k = 0; k < 6; ++k) da[k] += dab[k];
    ql_frame_rows(ft, dv, da, drP, kind == 0, S, out);
    ql_put_foot(dm, rw, f, true, out, col, rec, live, ef[f], -1, kind == 0 ? 0u : 1u);
    if (kind == 0) ql_put_fm(dm, ft, f, u, S, 0, col, rec, live);
  }
  ql_put_cde(rw, ef[0], ef[1], col, rec, live);
  if (kind == 0 && rw.coll) ql_put_coll(dm, nl, S, r0, 0, NB, col, rec, live);
}
// the base linear velocity columns and the base height column (the base acceleration depends on neither)
HSQP_HD void ql_rows_base_linear(const DevModel& dm, const QlRows& rw, const QlNodeLds& nl, double* rec, bool live) {
  // base linear velocity c: S = {0, e_c}, dv = S, da = -(v_i x S) (foot_column "prismatic")
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    double ef[2][7];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const QlFoot& ft = nl.ft[f];
      double Sx[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, t[6], dv[6], da[6], out[18];
      const double drP[3] = {0.0, 0.0, 0.0};
      Sx[3 + c] = 1.0;
      mxm(ft.vl, Sx, t);
      for (int k = 0; k < 6; ++k) { dv[k] = Sx[k]; da[k] = -t[k]; }
      ql_frame_rows(ft, dv, da, drP, false, Sx, out);
      ql_put_foot(dm, rw, f, true, out, NV + c, rec, live, ef[f], -1, 0x9u);
    }
    ql_put_cde(rw, ef[0], ef[1], NV + c, rec, live);
  }
  // base height (column 2): d pos_z = 1 -> the position gains of the equality rows; columns 0, 1 move no row
  double ef[2][7];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int cf = f == 0 ? rw.c0 : rw.c1;
    for (int k = 0; k < 7; ++k) ef[f][k] = 0.0;
    if (cf) ef[f][2] = dm.gain_pos_z; else ef[f][6] = dm.gain_pos_z;
  }
  ql_put_cde(rw, ef[0], ef[1], 2, rec, live);
}
// wrench component k6 of foot f: moves both feet through the base acceleration, and the foot's own friction / moment rows
HSQP_HD void ql_rows_wrench(const DevModel& dm, const QlRows& rw, const QlNodeLds& nl, const double* u, int f, int k6, const QlBaseKin& bk, const double* g,
                            double* rec, bool live) {
  const int col = NX + 6 * f + k6;
  double dab[6], ef[2][7], out[18];
  for (int k = 0; k < 3; ++k) { dab[k] = bk.w[0][k] * g[3] + bk.w[1][k] * g[4] + bk.w[2][k] * g[5]; dab[3 + k] = g[k]; }
#pragma unroll
  for (int f2 = 0; f2 < 2; ++f2) {
    ql_frame_rows_base(nl.ft[f2], dab, out);
    ql_put_foot(dm, rw, f2, false, out, col, rec, live, ef[f2], f2 == f ? k6 : -1);
  }
  ql_put_cde(rw, ef[0], ef[1], col, rec, live);
  ql_put_fm(dm, nl.ft[f], f, u, nullptr, k6, col, rec, live);
}

// ---- the ROWS PASS of stage 1 (k_lq_rows): a walk of the limb, kinematics only, that forms the residual / equality rows of every column from the
// stage-1 Jacobian columns k_lq_limb left in the record.  A kernel of its own: next to the model kernel's composites and column arithmetic
// the row arithmetic does not fit the register file (345 spilled registers in one kernel; registers are allotted per kernel).
// placement / velocity / trick acceleration of the lane's leaf at RK4 stage 1 (ql_forward without the bodies' inertia work); leaves the cos / sin
// of the lane's joints in csn for the way back, the foot body's state and the collision points the lane owns in the node's shared data
template <class KC>
HSQP_HD void ql_kin_to_leaf(const DevModel& dm, const KC& kc, const QlLimb& lb, const double* x, const double* u, int L, const QlBaseKin& bk, double* csn, int csn_ld,
                            QlState& st, QlNodeLds& nl) {
  for (int k = 0; k < 9; ++k) st.R[k] = bk.R[k];
  for (int k = 0; k < 3; ++k) st.r[k] = 0.0;
  for (int k = 0; k < 6; ++k) { st.vl[k] = bk.vl[2][k]; st.al[k] = bk.al[2][k]; }
  auto capture = [&](int i, bool own) {
    for (int fo = 0; fo < 2; ++fo) {
      if (dm.contact_body[fo] != i || dm.foot_limb[fo] != L) continue;
      QlFoot& ft = nl.ft[fo];
      double rr[3];
      m3_mulv(st.R, dm.contact_p[fo], rr);
      for (int k = 0; k < 9; ++k) ft.R[k] = st.R[k];
      for (int k = 0; k < 3; ++k) ft.rP[k] = rr[k] + st.r[k];
      for (int k = 0; k < 6; ++k) { ft.vl[k] = st.vl[k]; ft.al[k] = st.al[k]; }
    }
    if (own) {
      for (int p = 0; p < 10; ++p) {
        if (dm.coll_body[p] != i) continue;
        double tt[3];
        m3_mulv(st.R, dm.coll_p[p], tt);
        for (int k = 0; k < 3; ++k) nl.cp[p][k] = st.r[k] + tt[k];
      }
    }
  };
  if (L == 0) capture(0, true);
  const unsigned long long path = lb.path;
  const int len = lb.len;
  const unsigned own = lb.own;
  for (int t = 0; t < lb.max_len; ++t) {
    if (t >= len) continue;
    const int i = (int)((path >> (8 * t)) & 0xffull), j = i - 1;
    const double qd = x[NV + 6 + j], qdd = u[12 + j];
    double sn, cn, Rq[9], Mq[9], Rn[9], w[3], rn[3];
    sincos(x[6 + j], &sn, &cn);
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
    QV_SCHED_FENCE();
    capture(i, ((own >> t) & 1u) != 0);
  }
}
// column `kind` (0 / 1 / 2: d/dq, d/dqd, d/dqdd) of the joint at step t of the lane's limb (a valid column if the lane has no step t)
HSQP_HD int ql_rows_column(const QlLimb& lb, int t, int kind) {
  const bool has = t >= 0 && t < lb.len;
  const int i = has ? (int)((lb.path >> (8 * t)) & 0xffull) : 1, j = i - 1;
  return kind == 0 ? 3 + i + 2 : (kind == 1 ? NV + 3 + i + 2 : NX + 12 + j);
}
HSQP_HD void ql_rows_fetch(const double* gs, int col, double* g) {
#if defined(HSQP_EXP_NOFETCH)   /* timing experiment only (wrong results): no stage-Jacobian column fetch, hence no wait behind the row stores */
#pragma unroll
  for (int k = 0; k < 6; ++k) g[k] = 1e-3 * (col + k);
#else
#pragma unroll
  for (int k = 0; k < 6; ++k) g[k] = gs[col * GT_LD + k];
#endif
}
// one step of the rows pass: the rows of the three columns of joint i = path[t], then up to the parent.  pend: the stage Jacobian columns of
// (t, kinds 0 .. 2) on entry, of step t - 1 on exit.
template <class KC>
HSQP_HD void ql_rows_back_step(const DevModel& dm, const KC& kc, const QlLimb& lb, const QlRows& rw, const QlNodeLds& nl, const double* x, const double* u, int t,
                               QlState& st, const double* csn, int csn_ld, const double (*wE)[3], const double* gs, double (*pend)[6], double* rec, bool live) {
  const bool active = t < lb.len;
  const int i = active ? (int)((lb.path >> (8 * t)) & 0xffull) : 1, j = i - 1;
  const double qd = x[NV + 6 + j], qdd = u[12 + j];
  double S[6], Sd[6];
  m3_mulv(st.R, kc.axis[i], S);
  v3_cross(st.r, S, S + 3);
  mxm(st.vl, S, Sd);
  const bool mine = active && ((lb.own >> t) & 1u) != 0;
  const bool sup = t <= lb.foot_step;
  // pend: the three stage-Jacobian columns of THIS step (kinds 0, 1, 2), fetched during the step before it, in front of that step's first stores.  The wave
  // waits for them right here — the only stores in flight are the previous step's last ones — takes them over, and sends the next step's three columns on
  // their way.  A wait behind a column's ~22 stores drains them all (the in-order memory counter; the store count of a column is not a compile-time constant,
  // so the compiler waits for zero): three such waits per step — every column used to be fetched one column ahead — were 0.12 ms of the LQ kernels' 1.0
  // (DESIGN.md §7 (4)).
  double g[3][6];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(pend[q][k]));
  QV_SCHED_FENCE();
#endif
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int k = 0; k < 6; ++k) g[q][k] = pend[q][k];
#pragma unroll
  for (int q = 0; q < 3; ++q) ql_rows_fetch(gs, ql_rows_column(lb, t - 1, q), pend[q]);
  QV_SCHED_FENCE();
#pragma unroll 1
  for (int kind = 0; kind < 3; ++kind) {
    double gcur[6];
    if (kind == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) gcur[k] = g[0][k];
    } else if (kind == 1) {
#pragma unroll
      for (int k = 0; k < 6; ++k) gcur[k] = g[1][k];
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) gcur[k] = g[2][k];
    }
    if (mine) ql_rows_joint(dm, rw, nl, u, kind, sup, i, S, Sd, st, wE, gcur, ql_rows_column(lb, t, kind), rec, live);
  }
  QV_SCHED_FENCE();
  if (active) ql_unwind(kc, i, S, Sd, qd, qdd, csn[t * csn_ld], csn[t * csn_ld + 1], st);
}
// the rows of the base columns (the division of labour of ql_base_columns)
HSQP_HD void ql_rows_base(const DevModel& dm, const QlRows& rw, const QlNodeLds& nl, const double* u, int L, const QlBaseKin& bk, const QlShared& sh,
                          const double* gs, double* rec, bool live) {
  // (the columns are fetched in two batches, each in front of the stores of the columns it serves and waited for at once: see ql_rows_back_step)
  const int fo = rw.own >= 0 ? rw.own : 0, cw0 = NX + 6 * fo, Le = L < 3 ? L : 2;
  double e0[6], e1[6], w0[6], w1[6], w2[6];
  auto arrive = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
    QV_SCHED_FENCE();
#pragma unroll
    for (int k = 0; k < 6; ++k) { asm volatile("" : "+v"(w0[k])); asm volatile("" : "+v"(w1[k])); asm volatile("" : "+v"(w2[k])); }
    QV_SCHED_FENCE();
#endif
  };
  ql_rows_fetch(gs, 3 + Le, e0);
  ql_rows_fetch(gs, NV + 3 + Le, e1);
  ql_rows_fetch(gs, cw0, w0);
  ql_rows_fetch(gs, cw0 + 1, w1);
  ql_rows_fetch(gs, cw0 + 2, w2);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int k = 0; k < 6; ++k) { asm volatile("" : "+v"(e0[k])); asm volatile("" : "+v"(e1[k])); }
#endif
  arrive();
  if (L < 3) {
#pragma unroll 1
    for (int kind = 0; kind < 2; ++kind) {
      double gc[6];
      if (kind == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) gc[k] = e0[k];
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) gc[k] = e1[k];
      }
      ql_rows_euler(dm, rw, nl, u, kind, L, bk, sh, gc, rec, live);
    }
  } else {
    ql_rows_base_linear(dm, rw, nl, rec, live);
  }
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    if (half == 1) {
      ql_rows_fetch(gs, cw0 + 3, w0);
      ql_rows_fetch(gs, cw0 + 4, w1);
      ql_rows_fetch(gs, cw0 + 5, w2);
      arrive();
    }
    if (rw.own >= 0) {
#pragma unroll 1
      for (int q = 0; q < 3; ++q) {
        double gc[6];
        if (q == 0) {
#pragma unroll
          for (int k = 0; k < 6; ++k) gc[k] = w0[k];
        } else if (q == 1) {
#pragma unroll
          for (int k = 0; k < 6; ++k) gc[k] = w1[k];
        } else {
#pragma unroll
          for (int k = 0; k < 6; ++k) gc[k] = w2[k];
        }
        ql_rows_wrench(dm, rw, nl, u, rw.own, 3 * half + q, bk, gc, rec, live);
      }
    }
  }
}

// The RK4 chain and the defect of a node behind the limb lanes: one small workgroup per node, a lane per column (lq_chain_column, hsqp_lq.h)
struct LqChainWS {
  double blk[3][2][6][6];
  double as[4][6];
  double bvec[64];
};
// columns = false: the defect and its squared norm only — k_project runs the column chain itself from REC_GS (`chain`, hsqp_project.h) and REC_PV is not written
HSQP_HD void lq_chain_node(const Ctx& ctx, LqChainWS& w, const double* x, const double* u, const double* xnext, double dt, double* rec, bool columns = true) {
  WG_FOR(ctx, i, 3 * 72 + 24) {
    if (i < 3 * 72) {
      const int sg = i / 72, which = (i / 36) % 2, r = (i / 6) % 6, k = i % 6;
      if (columns) w.blk[sg][which][r][k] = rec[REC_GS + lq_chain_blk_offset(sg, which, r, k)];
    } else w.as[(i - 216) / 6][(i - 216) % 6] = rec[REC_AS + i - 216];
  }
  WG_SYNC(ctx);
  // ---- RK4 value: x_next = x + dt/6 (k1 + 2 k2 + 2 k3 + k4), defect (lq_node, hsqp_lq.h "RK4 value")
  WG_FOR(ctx, i, 64) {
    double b = 0.0;
    if (i < NV) {
      const double v0 = x[NV + i];
      double a0, a1, a2;   // what moves the stage velocities: the previous stage's acceleration
      if (i < 6) { a0 = w.as[0][i]; a1 = w.as[1][i]; a2 = w.as[2][i]; }
      else a0 = a1 = a2 = u[12 + i - 6];
      const double v1 = v0 + 0.5 * dt * a0, v2 = v0 + 0.5 * dt * a1, v3 = v0 + dt * a2;
      b = x[i] + dt / 6.0 * (v0 + 2.0 * v1 + 2.0 * v2 + v3) - xnext[i];
    } else if (i < NX) {
      const int k = i - NV;
      const double a = k < 6 ? (w.as[0][k] + 2.0 * w.as[1][k] + 2.0 * w.as[2][k] + w.as[3][k]) / 6.0 : u[12 + k - 6];
      b = x[i] + dt * a - xnext[i];
    }
    rec[REC_B + i] = b;
#if defined(__HIP_DEVICE_COMPILE__)
    // the squared defect: a butterfly over the 64 lanes that hold it (wave 0) instead of a second barrier and one lane's 58-term sum
    double sq = b * b;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m);
    if (i == 0) rec[REC_MISC + 3] = (dt > 0.0 ? dt : 1.0) * sq;
#else
    w.bvec[i] = b;
#endif
  }
  if (columns) WG_FOR(ctx, col, LDJ) lq_chain_column<true>(w.blk, rec + REC_GS, col, dt, rec);
#if !defined(__HIP_DEVICE_COMPILE__)
  WG_FOR(ctx, it, 1) {
    double dyn = 0.0;
    for (int i = 0; i < NX; ++i) dyn += w.bvec[i] * w.bvec[i];
    rec[REC_MISC + 3] = (dt > 0.0 ? dt : 1.0) * dyn;
  }
#endif
}

#if !defined(__HIP_DEVICE_COMPILE__)
// The four lanes of a node one after the other, the quad exchanges in between (host builds: tests/hostemu): what k_lq_limb writes to the node's
// record: REC_GS (transposed), REC_AS
inline void ql_node_host(const DevModel& dm, const double* x, const double* u, double dt, double* rec) {
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
      ql_forward(dm, *kc, ql_limb(dm, L), x, u, L, s, dt, bk[L], st[L], part[L], &csn[L][0][0], 2, rP[L], Ff[L]);
    }
    for (int e = 0; e < 16; ++e) tot[e] = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
    for (int L = 0; L < QV_LIMBS; ++L) ql_base_solve(tot, bk[L], sh[L]);
    for (int k = 0; k < 6; ++k) rec[REC_AS + 6 * s + k] = sh[0].ab[k];
    double* G = rec + REC_GS + (size_t)s * LDJ * GT_LD;
    auto putg = [&](int col, const double* g) { for (int k = 0; k < 6; ++k) G[col * GT_LD + k] = g[k]; };
    for (int L = 0; L < QV_LIMBS; ++L) for (int e = 0; e < NCMP; ++e) cmp[L][e] = 0.0;
    for (int t = dm.limb_max_len - 1; t >= 0; --t) {
      double snap[QV_LIMBS][NCMP];
      for (int L = 0; L < QV_LIMBS; ++L) for (int e = 0; e < NCMP; ++e) snap[L][e] = cmp[L][e];
      for (int L = 0; L < QV_LIMBS; ++L)
        for (int k = 1; k < QV_LIMBS; ++k)
          if ((dm.limb_merge[t][L] >> k) & 1u) for (int e = 0; e < NCMP; ++e) cmp[L][e] += snap[L ^ k][e];
      for (int L = 0; L < QV_LIMBS; ++L)
        ql_back_step(dm, *kc, ql_limb(dm, L), x, u, L, s, dt, t, st[L], cmp[L], sh[L], &csn[L][0][0], 2, rP[L], Ff[L], putg);
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
    for (int L = 0; L < QV_LIMBS; ++L) ql_base_columns(dm, L, bk[L], ctot, sh[L], dext_e, rP[L], putg);
    for (int L = 0; L < QV_LIMBS; ++L) ql_carry_advance(x, s, dt, sh[L], c[L]);
  }
  delete kc;
}
// ... what k_lq_rows writes: the node terms of stage 1 (rec: REC_GS / REC_AS of the node from ql_node_host; zero-filled otherwise, as hsqp_create
// leaves the record)
inline void ql_rows_host(const DevModel& dm, const double* x, const double* u, const double* par, double dt, double* rec, const double* xnext = nullptr) {
  QvConst* kc = new QvConst;
  QlNodeLds* nl = new QlNodeLds;
  memset(nl, 0, sizeof(QlNodeLds));
  const Ctx ctx{0, 1, nullptr};
  qv_load_const(ctx, dm, *kc, [] {});
  QlCarry c0;
  for (int k = 0; k < 6; ++k) { c0.vb[k] = 0.0; c0.ap[k] = 0.0; }
  QlBaseKin bk[QV_LIMBS];
  QlState st[QV_LIMBS];
  QlShared sh[QV_LIMBS];
  QlRows rw[QV_LIMBS];
  double csn[QV_LIMBS][QL_MAXLEN][2];
  for (int L = 0; L < QV_LIMBS; ++L) {
    ql_base_kin(dm, x, 0, dt, c0, bk[L]);
    ql_kin_to_leaf(dm, *kc, ql_limb(dm, L), x, u, L, bk[L], &csn[L][0][0], 2, st[L], *nl);
    ql_shared_from_record(bk[L], rec, sh[L]);
  }
  // pass A of every lane, then pass B (the device: a wave-level fence in between)
  double cost[QV_LIMBS], eq[QV_LIMBS], cost2[QV_LIMBS];
  int any[QV_LIMBS];
  for (int L = 0; L < QV_LIMBS; ++L) ql_terms_a(dm, x, u, par, L, dt, sh[L], *nl, rec, true, cost[L], eq[L]);
  for (int L = 0; L < QV_LIMBS; ++L) ql_terms_b(dm, x, u, par, L, dt, sh[L], *nl, rec, true, cost2[L], any[L]);
  const double ctot = ((cost[0] + cost2[0]) + (cost[1] + cost2[1])) + ((cost[2] + cost2[2]) + (cost[3] + cost2[3]));
  const double etot = (eq[0] + eq[1]) + (eq[2] + eq[3]);
  const int coll = (any[0] + any[1] + any[2] + any[3]) > 0;
  for (int L = 0; L < QV_LIMBS; ++L) ql_rows_setup(dm, par, L, dt, coll, rw[L]);
  ql_write_misc(par, dt, ctot, etot, coll, rec + REC_MISC);
  if (xnext) {   // the defect on the lanes (the device: when the chain is fused into k_project)
    double sq[QV_LIMBS];
    for (int L = 0; L < QV_LIMBS; ++L) sq[L] = ql_defect_lane(x, u, xnext, rec + REC_AS, dt, L, rec, true);
    rec[REC_MISC + 3] = (dt > 0.0 ? dt : 1.0) * ((sq[0] + sq[1]) + (sq[2] + sq[3]));
  }
  const double* G = rec + REC_GS;
  for (int t = dm.limb_max_len - 1; t >= 0; --t)
    for (int L = 0; L < QV_LIMBS; ++L) {
      double gcur[3][6];
      for (int q = 0; q < 3; ++q) ql_rows_fetch(G, ql_rows_column(ql_limb(dm, L), t, q), gcur[q]);
      ql_rows_back_step(dm, *kc, ql_limb(dm, L), rw[L], *nl, x, u, t, st[L], &csn[L][0][0], 2, bk[L].w, G, gcur, rec, true);
    }
  for (int L = 0; L < QV_LIMBS; ++L) ql_rows_base(dm, rw[L], *nl, u, L, bk[L], sh[L], G, rec, true);
  delete kc;
  delete nl;
}
#endif

}  // namespace hsqp
