// Rigid-body model evaluation of the whole-body flow map and its analytic Jacobian.
//
// What it computes (reference: computeBaseAcceleration,
//   humanoid_nmpc/humanoid_wb_mpc/src/dynamics/DynamicsHelperFunctions.cpp:52-82 and
//   humanoid_nmpc/humanoid_common_mpc/src/pinocchio_model/DynamicsHelperFunctions.cpp:197-218):
//     blkdiag(M_lin, M_ang) a_b = -nle_b(q,v) - M_bj(q) qdd_j + sum_i J_b,i(q)^T W_i
// The reference differentiates this with a CppAD tape.  Here the base rows are evaluated
// as a Newton-Euler momentum balance about the base origin O (world axes),
//     F(q,v,vd) = sum_i  I_i a_i + v_i x* I_i v_i          (vd_base = 0, gravity via a_root = +g e_z)
//     m a_lin = (F_ext - F).force,     Ibar_tot (E a_ang) = (F_ext - F).moment,   E = euler-rate axes,
// and differentiated in closed form with spatial-vector identities (S_c = joint motion axis,
// Sd = v_c x S_c, Sdd = a_c x S_c + v_c x Sd; ^c = composite over the subtree of c):
//     dF/dq_c   = S_c x* f^c + BB^c Sd_c + I^c Sdd_c
//     dF/dqd_c  = 2 I^c Sd_c + BB^c S_c ,        BB_i m = -I_i (v_i x m) + m x* (I_i v_i) + v_i x* (I_i m)
//     dF/dqdd_c = I^c S_c
// The floating base is treated as the chain prismatic x,y,z -> revolute z -> y' -> x'' which
// reproduces Pinocchio's JointModelComposite(Translation, SphericalZYX) coordinates exactly.
// Derivatives w.r.t. the base position and base linear velocity vanish identically.
#pragma once
#include "hsqp_common.h"

namespace hsqp {

// D = false: value-only evaluation (performance index pass) — the derivative arrays shrink to one element.
template <bool D>
struct StageWST {
  // ---- tree topology used by the placement walk and the ancestor sums (copied from DevModel once per workgroup:
  //      these indices sit in front of dependent loads)
  unsigned char anc[NB][NANC], n_anc[NB], chain_start[NB], chain_len[NB], sub[NB];   // sub: subtree size
  int n_chains;
  // ---- inputs of one evaluation
  double q[NV], v[NV], qddj[NJ], W[12];
  double ecs[3][2];                // cos, sin of the euler angles z, y, x
  // ---- base
  double E[9];       // E[3*r+c]: column c = world axis of euler rate c (z, y, x)
  double Einv[9];
  // ---- revolute coordinates jc = 0..25 (euler z,y,x, then joints); generalized coordinate = 3 + jc
  double S[NJC + 1][6], Sd[NJC][6], Sdd[D ? NJC : 1][6];   // S[NJC]: dump row of the placement walk
  double vl[NJC + 1][6], al[NJC + 1][6];   // spatial velocity / (gravity-trick) acceleration of the link after joint jc (row NJC: dump row of the walks)
  // ---- bodies
  double R[NB + 1][9], r[NB + 1][3];   // world rotation, origin relative to the base origin O (row NB: dump row of the placement walk)
  union {
    struct {                       // placement walk (phases F0-F1):
      double Mq[NB + 1][9];        //   Rfix * Rot(axis, q): joint rotation in the parent body frame (row NB: identity, the padded steps of the walk)
      double pa[NB + 1][6];        //   joint offset and joint axis in the parent body frame (copied next to Mq: they sit on the walk's serial path; row NB: zero)
    };
    // spatial inertia about O and net force per body (from the inertia phase on).  Derivative pass: turned IN PLACE into suffix sums
    // over the depth-first order (row NB = 0), so that the composite of body i is row i minus row i + subtree_size[i]
    struct { double In[NB + 1][10], f[NB + 1][6]; };
  };
  union {
    double BB[D ? NB + 1 : 1][D ? 36 : 1];   // per-body BB, then its suffix sums like In / f (dead once the composites are formed)
    double G[D ? 6 : 1][D ? 96 : 1];         // d ab / d[x;u], columns 0..92 used (written after the composites)
  };
  double Ic[D ? NB : 1][10], fc[D ? NB : 1][6], BBc[D ? NB : 1][D ? 36 : 1];   // (the value-only workspace carries no derivative storage: 8 workgroups / CU)
  double rP[2][3];                 // contact points relative to O
  double Fx[2][6];                 // contact wrenches about O {moment, force}
  // ---- results
  double Ftil[6];                  // F_ext - F  {moment, force}
  double Iinv[9];                  // inverse of the total rotational inertia about O
  double y[3];                     // E a_ang
  double ab[6];                    // base acceleration {lin, euler-rate acc}
};
using StageWS = StageWST<true>;

HSQP_HD void rot_axis(const double* ax, double q, double* Rm) {
  const double c = cos(q), s = sin(q), t = 1.0 - c, x = ax[0], y = ax[1], z = ax[2];
  Rm[0] = t * x * x + c;     Rm[1] = t * x * y - s * z; Rm[2] = t * x * z + s * y;
  Rm[3] = t * x * y + s * z; Rm[4] = t * y * y + c;     Rm[5] = t * y * z - s * x;
  Rm[6] = t * x * z - s * y; Rm[7] = t * y * z + s * x; Rm[8] = t * z * z + c;
}

HSQP_HD void joint_motion(const double* vpar, const double* apar, const double* Sx, double qd, double qdd,
                          double* vl, double* al, double* Sd, double* Sdd) {
  for (int k = 0; k < 6; ++k) vl[k] = vpar[k] + Sx[k] * qd;
  mxm(vl, Sx, Sd);
  for (int k = 0; k < 6; ++k) al[k] = apar[k] + Sx[k] * qdd + Sd[k] * qd;
  double t1[6], t2[6];
  mxm(al, Sx, t1);
  mxm(vl, Sd, t2);
  for (int k = 0; k < 6; ++k) Sdd[k] = t1[k] + t2[k];
}

// BB_i m for one body (see header comment)
HSQP_HD void bb_apply(const double* In, const double* vl, const double* m, double* out) {
  double h[6], x[6], t1[6], t2[6], yv[6], t3[6];
  inertia_apply(In, vl, h);
  mxm(vl, m, x);
  inertia_apply(In, x, t1);
  mxf(m, h, t2);
  inertia_apply(In, m, yv);
  mxf(vl, yv, t3);
  for (int k = 0; k < 6; ++k) out[k] = -t1[k] + t2[k] + t3[k];
}

HSQP_HD void mat6_mulv(const double* M, const double* v, double* r) {
  for (int i = 0; i < 6; ++i) {
    double s = 0.0;
    for (int k = 0; k < 6; ++k) s += M[6 * i + k] * v[k];
    r[i] = s;
  }
}

HSQP_HD void rot_axis_cs(const double* ax, double c, double s, double* Rm) {
  const double t = 1.0 - c, x = ax[0], y = ax[1], z = ax[2];
  Rm[0] = t * x * x + c;     Rm[1] = t * x * y - s * z; Rm[2] = t * x * z + s * y;
  Rm[3] = t * x * y + s * z; Rm[4] = t * y * y + c;     Rm[5] = t * y * z - s * x;
  Rm[6] = t * x * z - s * y; Rm[7] = t * y * z + s * x; Rm[8] = t * z * z + c;
}

// One evaluation of a_b (and, if DERIV, of G = d a_b / d[x;u]) at ws.q, ws.v, ws.qddj, ws.W.
//
// Serial depth: only the placements (R_i, r_i, w_i) are propagated along the tree, row by row (three independent
// items per chain, DevModel::chain_* / anc); velocities and accelerations are sums over the ancestor path, one item
// per component.  Everything else (Sdd, inertia, net force, BB) runs in fully parallel phases.
// The ancestor path of a body is 8 byte indices: read as one 64-bit word (one LDS round trip instead of eight dependent
// byte loads in front of the operand loads).
static_assert(NANC == 8, "packed ancestor path");
HSQP_HD unsigned long long anc_packed(const unsigned char* row) {
  unsigned long long v;
  memcpy(&v, row, 8);
  return v;
}
HSQP_HD int anc_at(unsigned long long pk, int n) { return (int)((pk >> (8 * n)) & 0xffull); }

template <class SW>
HSQP_HD void stage_topology(const Ctx& ctx, const DevModel& dm, SW& ws, bool sync = true) {
  WG_FOR(ctx, i, NB * NANC + NB) {
    if (i < NB * NANC) ws.anc[i / NANC][i % NANC] = dm.anc[i / NANC][i % NANC];
    else {
      const int b = i - NB * NANC;
      ws.n_anc[b] = (unsigned char)dm.n_anc[b]; ws.chain_start[b] = (unsigned char)dm.chain_start[b]; ws.chain_len[b] = (unsigned char)dm.chain_len[b];
      ws.sub[b] = (unsigned char)dm.subtree_size[b];
      if (b == 0) ws.n_chains = dm.n_chains;
    }
  }
  if (sync) WG_SYNC(ctx);
}

// Gout (optional, row stride LDJ, may be global memory): if given, the Jacobian goes there instead of ws.G (columns
// NZ..LDJ-1 are written as zero).
template <bool DERIV>
HSQP_HD void stage_eval(const Ctx& ctx, const DevModel& dm, StageWST<DERIV>& ws, double* Gout = nullptr) {
  // ---- phase F0: trigonometry, parallel over the 26 angles (joint rotations in the parent frame; euler cos/sin)
  WG_FOR(ctx, it, NB + 3) {
    if (it == NB + 2) {   // the identity joint the placement walk takes on the padded steps of a path
      for (int k = 0; k < 9; ++k) ws.Mq[NB][k] = (k % 4 == 0) ? 1.0 : 0.0;
      for (int k = 0; k < 6; ++k) ws.pa[NB][k] = 0.0;
      continue;
    }
    double sn, cs;
    sincos(ws.q[3 + it], &sn, &cs);   // one evaluation for both roles: angle 3 + it is euler angle it (it < 3) or joint it - 3
    if (it < 3) { ws.ecs[it][0] = cs; ws.ecs[it][1] = sn; continue; }
    const int i = it - 2;
    double Rq[9];
    rot_axis_cs(dm.axis[i], cs, sn, Rq);
    m3_mul(dm.Rfix[i], Rq, ws.Mq[i]);
    for (int k = 0; k < 3; ++k) { ws.pa[i][k] = dm.pfix[i][k]; ws.pa[i][3 + k] = dm.axis_p[i][k]; }
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 28);
  // ---- phase F1: placements.  Row r of R_i = R_p Mq_i and component r of r_i = r_p + R_p pfix_i, w_i = R_p axis_i depend
  // only on row r of R_p, so every chain is walked by three independent items (one per row), each from the base
  // down its whole ancestor path (the shared waist bodies are recomputed, not exchanged): no barrier inside the tree.
  // The euler item (E, its inverse, the axes of the three euler joints) is a loop of its own at a wave boundary (item 64 of a
  // two-wave workgroup): it runs beside the walk, not after it.
  WG_FOR(ctx, ite, 65) {
    if (ite != 64) continue;
    const double cz = ws.ecs[0][0], sz = ws.ecs[0][1], cy = ws.ecs[1][0], sy = ws.ecs[1][1];
    const double wz[3] = {0.0, 0.0, 1.0}, wy[3] = {-sz, cz, 0.0}, wx[3] = {cz * cy, sz * cy, -sy};
    for (int r = 0; r < 3; ++r) { ws.E[3 * r] = wz[r]; ws.E[3 * r + 1] = wy[r]; ws.E[3 * r + 2] = wx[r]; }
    m3_inverse(ws.E, ws.Einv);
    for (int k = 0; k < 3; ++k) {
      ws.S[0][k] = wz[k]; ws.S[1][k] = wy[k]; ws.S[2][k] = wx[k];
      ws.S[0][3 + k] = 0.0; ws.S[1][3 + k] = 0.0; ws.S[2][3 + k] = 0.0;
    }
  }
  WG_FOR(ctx, it, ws.n_chains * 3 + 3) {
    const double cz = ws.ecs[0][0], sz = ws.ecs[0][1], cy = ws.ecs[1][0], sy = ws.ecs[1][1], cx = ws.ecs[2][0], sx = ws.ecs[2][1];
    const int r = it % 3, ch = it / 3;
    double Rp[3];   // row r of R0 = Rz Ry Rx
    if (r == 0) { Rp[0] = cz * cy; Rp[1] = cz * sy * sx - sz * cx; Rp[2] = cz * sy * cx + sz * sx; }
    else if (r == 1) { Rp[0] = sz * cy; Rp[1] = sz * sy * sx + cz * cx; Rp[2] = sz * sy * cx - cz * sx; }
    else { Rp[0] = -sy; Rp[1] = cy * sx; Rp[2] = cy * cx; }
    if (ch == ws.n_chains) {           // the base itself
      for (int c = 0; c < 3; ++c) ws.R[0][3 * r + c] = Rp[c];
      ws.r[0][r] = 0.0;
      continue;
    }
    const int b0 = ws.chain_start[ch], end = b0 + ws.chain_len[ch] - 1, na = ws.n_anc[end];
    double rp = 0.0;
    // fully unrolled over the (padded) path: the index and operand loads do not depend on the running row, only the
    // multiply-adds are chained.  No branch and no select on the data: a padded step multiplies by the identity joint (row NB of
    // Mq / pa: the running row comes back bit for bit), and what is not to be kept — padded steps, ancestors that belong to another
    // chain — is stored to the dump rows (R[NB], r[NB], S[NJC]).
    const unsigned long long pk1 = anc_packed(ws.anc[end]);
#pragma unroll
    for (int n = 0; n < NANC; ++n) {
      const int ia = anc_at(pk1, n);
      const int i = n < na ? ia : NB;
      const int d = (n < na && ia >= b0) ? ia : NB;
      const double* M = ws.Mq[i];
      const double* pa = ws.pa[i];
      const double rn0 = Rp[0] * M[0] + Rp[1] * M[3] + Rp[2] * M[6];
      const double rn1 = Rp[0] * M[1] + Rp[1] * M[4] + Rp[2] * M[7];
      const double rn2 = Rp[0] * M[2] + Rp[1] * M[5] + Rp[2] * M[8];
      const double rr = rp + Rp[0] * pa[0] + Rp[1] * pa[1] + Rp[2] * pa[2];
      const double w = Rp[0] * pa[3] + Rp[1] * pa[4] + Rp[2] * pa[5];
      ws.R[d][3 * r] = rn0; ws.R[d][3 * r + 1] = rn1; ws.R[d][3 * r + 2] = rn2; ws.r[d][r] = rr; ws.S[d + 2][r] = w;
      Rp[0] = rn0; Rp[1] = rn1; Rp[2] = rn2; rp = rr;
    }
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 20);
  // ---- phase F2: linear part of the joint axes S_i = {w_i, r_i x w_i} and the link velocities v_i = v_base + sum_a S_a qd_a.  One item
  // per (chain, component) walks down the ancestor path of the chain's last body and records the running sum at the bodies of its own
  // chain (the shared ancestors are re-summed, not exchanged; other chains' bodies and the padded steps go to the dump rows): 48 items
  // of 8 steps instead of one 8-step sum per (body, component) — 156 items, three wave rounds.
  WG_FOR(ctx, it, ws.n_chains * 6 + 18) {
    const int k = it % 6, k1 = (k + 1) % 3, k2 = (k + 2) % 3, ch = it / 6 - 3;
    // the base (= the third euler link): angular part from the euler rates, linear part = base linear velocity
    double s;
    if (k < 3) { s = ws.S[0][k] * ws.v[3]; if (ch >= -2) s += ws.S[1][k] * ws.v[4]; if (ch >= -1) s += ws.S[2][k] * ws.v[5]; }
    else s = ws.v[k - 3];
    if (ch < 0) { ws.vl[ch + 3][k] = s; continue; }      // the euler links jc = 0, 1, 2
    const int b0 = ws.chain_start[ch], end = b0 + ws.chain_len[ch] - 1, na = ws.n_anc[end];
    const unsigned long long pk2 = anc_packed(ws.anc[end]);
#pragma unroll
    for (int n = 0; n < NANC; ++n) {   // the path is padded with the body itself
      // branch-free: sa = p1 S[j1] - p2 S[j2] with (p1, j1, p2, j2) = (1, k, 0, k) for the angular rows, (r[k1], k2, r[k2], k1) for the linear
      const int a = anc_at(pk2, n);
      const int d = (n < na && a >= b0) ? a + 2 : NJC;
      const double r1 = ws.r[a][k1], r2 = ws.r[a][k2];
      const double s1 = ws.S[a + 2][k < 3 ? k : k2], s2 = ws.S[a + 2][k < 3 ? k : k1];
      const double qa = ws.v[5 + a];
      const double sa = (k < 3 ? 1.0 : r1) * s1 - (k < 3 ? 0.0 : r2) * s2;
      s += (n < na ? qa : 0.0) * sa;
      ws.S[d][k] = sa;                        // (an angular item rewrites the value it read: no branch)
      ws.vl[d][k] = s;
    }
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 29);
  // ---- phase F3: Sd_i = v_i x S_i
  WG_FOR(ctx, it, NJC * 6) {
    const int jc = it / 6, k = it % 6, k1 = (k + 1) % 3, k2 = (k + 2) % 3;
    const double* vv = ws.vl[jc];
    const double* Sx = ws.S[jc];
    double s = vv[k1] * Sx[(k < 3 ? 0 : 3) + k2] - vv[k2] * Sx[(k < 3 ? 0 : 3) + k1];
    if (k >= 3) s += vv[3 + k1] * Sx[k2] - vv[3 + k2] * Sx[k1];
    ws.Sd[jc][k] = s;
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 30);
  // ---- phase F4: link accelerations (gravity trick, base acceleration unknown -> 0):
  // a_i = a_0 + sum_a (S_a qdd_a + Sd_a qd_a); the euler joints enter with zero acceleration.  Walked like F2.
  WG_FOR(ctx, it, ws.n_chains * 6 + 18) {
    const int k = it % 6, ch = it / 6 - 3;
    double s = k == 5 ? dm.gravity : 0.0;
    s += ws.Sd[0][k] * ws.v[3];
    if (ch >= -2) s += ws.Sd[1][k] * ws.v[4];
    if (ch >= -1) s += ws.Sd[2][k] * ws.v[5];
    if (ch < 0) { ws.al[ch + 3][k] = s; continue; }
    const int b0 = ws.chain_start[ch], end = b0 + ws.chain_len[ch] - 1, na = ws.n_anc[end];
    const unsigned long long pk4 = anc_packed(ws.anc[end]);
#pragma unroll
    for (int n = 0; n < NANC; ++n) {
      const int a = anc_at(pk4, n);
      const int d = (n < na && a >= b0) ? a + 2 : NJC;
      const double t = ws.S[a + 2][k] * ws.qddj[a - 1] + ws.Sd[a + 2][k] * ws.v[5 + a];
      s += n < na ? t : 0.0;
      ws.al[d][k] = s;
    }
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 21);
  // ---- per-body spatial inertia about O and net force
  // derivative pass (two waves): the bodies on wave 0, the contact points and Sdd from item 64 on — different roles in one wave run
  // one after the other, on two waves side by side
  constexpr int IT_C = DERIV ? 64 : NB;
  WG_FOR(ctx, it, IT_C + 2 + (DERIV ? NJC : 0)) {
    if (it >= NB && it < IT_C) continue;
    if (it >= IT_C + 2) {   // Sdd = a x S + v x Sd (only the derivative columns need it)
      const int jc = it - IT_C - 2;
      double t1[6], t2[6];
      mxm(ws.al[jc], ws.S[jc], t1);
      mxm(ws.vl[jc], ws.Sd[jc], t2);
      for (int k = 0; k < 6; ++k) ws.Sdd[jc][k] = t1[k] + t2[k];
      continue;
    }
    if (it >= IT_C) {   // contact point of foot f relative to O and its wrench about O {moment, force} (off the serial totals phase)
      const int f = it - IT_C, b = dm.contact_body[f];
      double rr[3], mom[3];
      m3_mulv(ws.R[b], dm.contact_p[f], rr);
      for (int k = 0; k < 3; ++k) { rr[k] += ws.r[b][k]; ws.rP[f][k] = rr[k]; }
      v3_cross(rr, ws.W + 6 * f, mom);
      for (int k = 0; k < 3; ++k) { ws.Fx[f][k] = ws.W[6 * f + 3 + k] + mom[k]; ws.Fx[f][3 + k] = ws.W[6 * f + k]; }
      continue;
    }
    const int i = it;
    const double* Rb = ws.R[i];
    double c[3], t[9], Iw[9];
    m3_mulv(Rb, dm.com[i], c);
    for (int k = 0; k < 3; ++k) c[k] += ws.r[i][k];
    m3_mul(Rb, dm.inertia[i], t);
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) Iw[3 * a + b] = t[3 * a] * Rb[3 * b] + t[3 * a + 1] * Rb[3 * b + 1] + t[3 * a + 2] * Rb[3 * b + 2];
    const double m = dm.mass[i], cc = v3_dot(c, c);
    double* In = ws.In[i];
    In[0] = m; In[1] = m * c[0]; In[2] = m * c[1]; In[3] = m * c[2];
    In[4] = Iw[0] + m * (cc - c[0] * c[0]); In[5] = Iw[1] - m * c[0] * c[1]; In[6] = Iw[2] - m * c[0] * c[2];
    In[7] = Iw[4] + m * (cc - c[1] * c[1]); In[8] = Iw[5] - m * c[1] * c[2]; In[9] = Iw[8] + m * (cc - c[2] * c[2]);
    const int jc = i + 2;
    double h[6], fa[6], fv[6];
    inertia_apply(In, ws.vl[jc], h);
    inertia_apply(In, ws.al[jc], fa);
    mxf(ws.vl[jc], h, fv);
    for (int k = 0; k < 6; ++k) ws.f[i][k] = fa[k] + fv[k];
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 22);
  // totals and the block-diagonal base solve (one item).  Itot / ftot: the sums over all bodies
  auto totals = [&](const double* Itot, const double* ftot) {
    double Fext[6];
    for (int k = 0; k < 6; ++k) Fext[k] = ws.Fx[0][k] + ws.Fx[1][k];
    for (int k = 0; k < 6; ++k) ws.Ftil[k] = Fext[k] - ftot[k];
    const double* I6 = Itot + 4;
    const double Ib[9] = {I6[0], I6[1], I6[2], I6[1], I6[3], I6[4], I6[2], I6[4], I6[5]};
    m3_inverse(Ib, ws.Iinv);
    m3_mulv(ws.Iinv, ws.Ftil, ws.y);
    const double minv = 1.0 / Itot[0];
    for (int k = 0; k < 3; ++k) ws.ab[k] = ws.Ftil[3 + k] * minv;
    m3_mulv(ws.Einv, ws.y, ws.ab + 3);
  };
  if (DERIV) {
    WG_FOR(ctx, it, NB * 6) {
      const int i = it / 6, k = it % 6;
      double m[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, col[6];
      m[k] = 1.0;
      bb_apply(ws.In[i], ws.vl[i + 2], m, col);
      for (int r = 0; r < 6; ++r) ws.BB[i][6 * r + k] = col[r];
    }
    WG_SYNC(ctx);
    PH_TICK(ctx, 23);
    // composites over subtrees: comp_i = sum of own_j over the depth-first range [i, i + sub_i) = Sfx[i] - Sfx[i + sub_i] with the
    // suffix sums Sfx[j] = sum_{l >= j} own_l, formed IN PLACE by 52 items (one per quantity: 10 inertia entries, 6 force entries, 36
    // entries of BB): 24 independent loads, a chain of 23 additions in registers, 25 stores (row NB = 0); a second phase takes the
    // differences with unit-stride loads.  (The range-sum form — one item per (chain, quantity) with masked block loads — was the
    // longest phase of the kernel: 7.8 k cycles per stage; differences taken by the consumers while they load doubled the column phase.)
    WG_FOR(ctx, e, 52) {
      double* own = e < 10 ? &ws.In[0][e] : (e < 16 ? &ws.f[0][e - 10] : &ws.BB[0][e - 16]);
      const int st = e < 10 ? 10 : (e < 16 ? 6 : 36);
      double v[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) v[j] = own[j * st];
      double sfx = 0.0;
      own[NB * st] = 0.0;
#pragma unroll
      for (int j = NB - 1; j >= 0; --j) { sfx += v[j]; own[j * st] = sfx; }
    }
    WG_SYNC(ctx);
    // differences: an item = (quantity, half of the bodies), fully unrolled over its 12 bodies so that all loads are in flight
    // together (a rolled item loop pays two dependent LDS round trips per body: 7 k cycles per stage)
    static_assert(NB == 24, "two groups of twelve bodies");
    // (one more item, on the second wave beside its 40 difference items: the totals and the base solve — after the suffix sums row 0 of
    //  In / f IS the sum over all bodies, bit for bit the composite of the base (x - 0) — instead of a one-item phase of its own)
    WG_FOR(ctx, it, 2 * 52 + 1) {
      if (it == 2 * 52) { totals(ws.In[0], ws.f[0]); continue; }
      const int e = it % 52, i0 = (it / 52) * 12;
      const double* own = e < 10 ? &ws.In[0][e] : (e < 16 ? &ws.f[0][e - 10] : &ws.BB[0][e - 16]);
      double* comp = e < 10 ? &ws.Ic[0][e] : (e < 16 ? &ws.fc[0][e - 10] : &ws.BBc[0][e - 16]);
      const int st = e < 10 ? 10 : (e < 16 ? 6 : 36);
      unsigned long long s8;
      unsigned int s4;
      memcpy(&s8, &ws.sub[i0], 8);
      memcpy(&s4, &ws.sub[i0 + 8], 4);
      double a[12], b[12];
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const int sz = j < 8 ? (int)((s8 >> (8 * j)) & 0xffull) : (int)((s4 >> (8 * (j - 8))) & 0xffu);
        a[j] = own[(i0 + j) * st];
        b[j] = own[(i0 + j + sz) * st];
      }
#pragma unroll
      for (int j = 0; j < 12; ++j) comp[(i0 + j) * st] = a[j] - b[j];
    }
  } else {
    WG_FOR(ctx, e, 16) {
      double s = 0.0;
      if (e < 10) { for (int d = 0; d < NB; ++d) s += ws.In[d][e]; ws.Ic[0][e] = s; }
      else { for (int d = 0; d < NB; ++d) s += ws.f[d][e - 10]; ws.fc[0][e - 10] = s; }
    }
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 24);
  if (!DERIV) {
    WG_FOR(ctx, it, 1) totals(ws.Ic[0], ws.fc[0]);
    WG_SYNC(ctx);
  }
  PH_TICK(ctx, 25);
  if (!DERIV) return;
  // ---- Jacobian columns: items (jc, kind in {q, qd, qdd}) and the 12 wrench components
  // Kind-major item order, the q-columns (by far the longest) alone on wave 0, everything else from item 64 on: lanes of one wave
  // that take different branches run them one after the other, so a (jc, kind)-interleaved order made BOTH waves execute all
  // three column kinds
  static_assert(NJC <= 64 && 64 + NJC + NJ + 12 + 54 <= 192, "item layout of the Jacobian-column phase");
  WG_FOR(ctx, it, 64 + NJC + NJ + 12) {
    double rhs[3], lin[3];
    int col;
    if (it >= NJC && it < 64) continue;
    if (it < 64 + NJC + NJ) {
      const int kind = it < NJC ? 0 : (it < 64 + NJC ? 1 : 2);
      const int jc = kind == 0 ? it : (kind == 1 ? it - 64 : 3 + it - 64 - NJC);
      const int bi = jc < 3 ? 0 : jc - 2;
      const double* Sx = ws.S[jc];
      const double* Ic = ws.Ic[bi];
      double dF[6];
      if (kind == 0) {
        col = 3 + jc;
        double t1[6], t2[6], t3[6];
        mxf(Sx, ws.fc[bi], t1);
        mat6_mulv(ws.BBc[bi], ws.Sd[jc], t2);
        inertia_apply(Ic, ws.Sdd[jc], t3);
        for (int k = 0; k < 6; ++k) dF[k] = t1[k] + t2[k] + t3[k];
        // (dI_tot/dq_c) y6 = S x* (I^c y6) - I^c (S x y6),  y6 = {y, 0}
        const double y6[6] = {ws.y[0], ws.y[1], ws.y[2], 0.0, 0.0, 0.0};
        double iy[6], a1[6], sy[6], a2[6];
        inertia_apply(Ic, y6, iy);
        mxf(Sx, iy, a1);
        mxm(Sx, y6, sy);
        inertia_apply(Ic, sy, a2);
        double extra[3] = {a1[0] - a2[0], a1[1] - a2[1], a1[2] - a2[2]};
        if (jc < 2) {  // d(S_E)/dq_c a_ang = {w_c x sum_{e>c} w_e a_e, 0}
          double z[3] = {0.0, 0.0, 0.0}, wz[3], t[3];
          for (int e = jc + 1; e < 3; ++e)
            for (int k = 0; k < 3; ++k) z[k] += ws.E[3 * k + e] * ws.ab[3 + e];
          v3_cross(Sx, z, wz);
          sym3_mulv(ws.Ic[0] + 4, wz, t);
          for (int k = 0; k < 3; ++k) extra[k] += t[k];
        }
        double dext[3] = {0.0, 0.0, 0.0};
        for (int f = 0; f < 2; ++f) {
          const int cb = dm.contact_body[f];
          if (jc < 3 || (cb >= bi && cb < bi + (int)ws.sub[bi])) {
            double d[3], dr[3], t[3];
            for (int k = 0; k < 3; ++k) d[k] = ws.rP[f][k] - (jc < 3 ? 0.0 : ws.r[bi][k]);
            v3_cross(Sx, d, dr);
            v3_cross(dr, ws.W + 6 * f, t);
            for (int k = 0; k < 3; ++k) dext[k] += t[k];
          }
        }
        for (int k = 0; k < 3; ++k) rhs[k] = dext[k] - dF[k] - extra[k];
      } else if (kind == 1) {
        col = NV + 3 + jc;
        double t1[6], t2[6];
        inertia_apply(Ic, ws.Sd[jc], t1);
        mat6_mulv(ws.BBc[bi], Sx, t2);
        for (int k = 0; k < 6; ++k) dF[k] = 2.0 * t1[k] + t2[k];
        for (int k = 0; k < 3; ++k) rhs[k] = -dF[k];
      } else {
        col = NX + 12 + (jc - 3);
        inertia_apply(Ic, Sx, dF);
        for (int k = 0; k < 3; ++k) rhs[k] = -dF[k];
      }
      const double minv = 1.0 / ws.Ic[0][0];
      for (int k = 0; k < 3; ++k) lin[k] = -dF[3 + k] * minv;
    } else {
      const int wi = it - 64 - NJC - NJ, f = wi / 6, k6 = wi % 6;
      col = NX + wi;
      double e[3] = {0.0, 0.0, 0.0};
      e[k6 % 3] = 1.0;
      if (k6 < 3) {
        v3_cross(ws.rP[f], e, rhs);
        const double minv = 1.0 / ws.Ic[0][0];
        for (int k = 0; k < 3; ++k) lin[k] = e[k] * minv;
      } else {
        for (int k = 0; k < 3; ++k) { rhs[k] = e[k]; lin[k] = 0.0; }
      }
    }
    double t[3], ang[3];
    m3_mulv(ws.Iinv, rhs, t);
    m3_mulv(ws.Einv, t, ang);
    if (Gout) for (int k = 0; k < 3; ++k) { Gout[k * LDJ + col] = lin[k]; Gout[(3 + k) * LDJ + col] = ang[k]; }
    else for (int k = 0; k < 3; ++k) { ws.G[k][col] = lin[k]; ws.G[3 + k][col] = ang[k]; }
  }
  // columns with identically zero derivative: base position (0..2) and base linear velocity (29..31)
  WG_FOR(ctx, itz, 64 + 6 * 9) {
    if (itz < 64) continue;
    const int it = itz - 64, r = it / 9, c = it % 9;
    const int col = c < 3 ? c : (c < 6 ? NV + (c - 3) : NZ + (c - 6));
    if (Gout) Gout[r * LDJ + col] = 0.0;
    else if (col < NZ) ws.G[r][col] = 0.0;
  }
  WG_SYNC(ctx);
  PH_TICK(ctx, 26);
}

}  // namespace hsqp
