// Host-side construction of the device model image from the C-ABI model description.
#pragma once
#include <cmath>
#include <string.h>

#include <string>

#include "hsqp_common.h"

namespace hsqp {

// Returns an empty string on success, otherwise a description of what is wrong with the model.
inline std::string build_dev_model(const hsqp_model_desc& md, DevModel& dm) {
  memset(&dm, 0, sizeof(dm));
  if (md.formulation != HSQP_FORM_WB && md.formulation != HSQP_FORM_CENTROIDAL) return "formulation must be HSQP_FORM_WB or HSQP_FORM_CENTROIDAL";
  dm.formulation = md.formulation;
  if (md.n_joints != NJ) return "n_joints must be " + std::to_string(NJ);
  dm.total_mass = 0.0;
  for (int i = 0; i < NB; ++i) {
    const hsqp_body& b = md.bodies[i];
    if (i == 0 ? b.parent != -1 : (b.parent < 0 || b.parent >= i)) return "bodies must be listed parents-first with bodies[0] the base";
    dm.parent[i] = b.parent;
    memcpy(dm.Rfix[i], b.R, sizeof(b.R));
    memcpy(dm.pfix[i], b.p, sizeof(b.p));
    memcpy(dm.axis[i], b.axis, sizeof(b.axis));
    dm.mass[i] = b.mass;
    memcpy(dm.com[i], b.com, sizeof(b.com));
    memcpy(dm.inertia[i], b.inertia, sizeof(b.inertia));
    dm.total_mass += b.mass;
    if (i > 0) { dm.q_lo[i - 1] = b.q_lo; dm.q_hi[i - 1] = b.q_hi; }
    if (i > 0) {
      const double n = b.axis[0] * b.axis[0] + b.axis[1] * b.axis[1] + b.axis[2] * b.axis[2];
      if (fabs(n - 1.0) > 1e-9) return "joint axes must be unit vectors";
    }
  }
  // subtree sizes; depth-first (contiguous subtree) ordering is required by the composite sums
  for (int i = NB - 1; i >= 0; --i) {
    dm.subtree_size[i] += 1;
    if (i > 0) dm.subtree_size[dm.parent[i]] += dm.subtree_size[i];
  }
  for (int i = 1; i < NB; ++i) {
    const int p = dm.parent[i];
    if (!(i > p && i < p + dm.subtree_size[p])) return "bodies are not in depth-first order";
    for (int a = p; a >= 0; a = dm.parent[a])
      if (!(i >= a && i < a + dm.subtree_size[a])) return "bodies are not in depth-first order";
  }
  // chains (maximal single-child paths), ancestor paths, joint axes in the parent frame
  {
    int nchild[NB] = {0};
    for (int i = 1; i < NB; ++i) nchild[dm.parent[i]]++;
    int chain_of[NB];
    chain_of[0] = -1;
    dm.n_chains = 0;
    for (int i = 1; i < NB; ++i) {
      const int p = dm.parent[i];
      if (p != 0 && nchild[p] == 1 && i == p + 1) {
        chain_of[i] = chain_of[p];
        dm.chain_len[chain_of[i]]++;
      } else {
        const int c = dm.n_chains++;
        chain_of[i] = c;
        dm.chain_start[c] = i; dm.chain_len[c] = 1;
      }
    }
    for (int c = 0; c < dm.n_chains; ++c) if (dm.chain_len[c] > MAXCHAIN) return "a chain of the kinematic tree is longer than MAXCHAIN bodies";
    for (int i = 0; i < NB; ++i) {
      int path[NB], n = 0;
      for (int a = i; a > 0; a = dm.parent[a]) path[n++] = a;
      if (n > NANC) return "kinematic tree deeper than NANC moving bodies";
      dm.n_anc[i] = n;
      for (int k = 0; k < NANC; ++k) dm.anc[i][k] = (unsigned char)(k < n ? path[n - 1 - k] : i);   // padded with the body itself (valid index)
    }
    dm.chain_start[dm.n_chains] = 0; dm.chain_len[dm.n_chains] = 1;   // the base
    // the parent of a chain's first body is the base or the LAST body of an earlier chain (a body with several children ends its chain)
    for (int c = 0; c < dm.n_chains; ++c) {
      const int pb = dm.parent[dm.chain_start[c]];
      int slot = pb == 0 ? 0 : -1;
      for (int c2 = 0; c2 < c && slot < 0; ++c2)
        if (dm.chain_start[c2] + dm.chain_len[c2] - 1 == pb) slot = 1 + c2;
      if (slot < 0) return "internal: a chain does not attach to the end of an earlier chain";
      dm.chain_par_slot[c] = slot;
    }
    if (md.formulation == HSQP_FORM_CENTROIDAL && dm.n_chains + 1 > 8) return "centroidal: kinematic trees with more than 7 chains are not supported";
    for (int i = 1; i < NB; ++i) {
      const hsqp_body& b = md.bodies[i];
      for (int r = 0; r < 3; ++r) dm.axis_p[i][r] = b.R[3 * r] * b.axis[0] + b.R[3 * r + 1] * b.axis[1] + b.R[3 * r + 2] * b.axis[2];
    }
  }
  // limbs of the quad value pass (hsqp_lqv.h): one per leaf of the tree
  {
    int nchild[NB] = {0};
    for (int i = 1; i < NB; ++i) nchild[dm.parent[i]]++;
    bool owned[NB] = {false};
    int nl = 0;
    bool fits = true;
    for (int i = 1; i < NB && fits; ++i) {
      if (nchild[i] != 0) continue;
      if (nl == QV_LIMBS) { fits = false; break; }
      unsigned long long path = 0;
      unsigned own = 0;
      for (int k = 0; k < dm.n_anc[i]; ++k) {
        const int b = dm.anc[i][k];
        path |= (unsigned long long)b << (8 * k);
        if (!owned[b]) { own |= 1u << k; owned[b] = true; }
      }
      dm.limb_path[nl] = path; dm.limb_len[nl] = dm.n_anc[i]; dm.limb_own[nl] = own;
      if (dm.n_anc[i] > dm.limb_max_len) dm.limb_max_len = dm.n_anc[i];
      ++nl;
    }
    for (int f = 0; f < 2 && fits; ++f) {
      const int cb = md.contact[f].body;
      dm.foot_limb[f] = -1;
      for (int l = 0; l < nl && dm.foot_limb[f] < 0; ++l)
        for (int k = 0; k < dm.limb_len[l]; ++k)
          if ((int)((dm.limb_path[l] >> (8 * k)) & 0xffull) == cb) { dm.foot_limb[f] = l; break; }
      if (dm.foot_limb[f] < 0) fits = false;   // (a contact frame on the base itself)
    }
    if (fits && dm.foot_limb[0] == dm.foot_limb[1]) fits = false;
    dm.n_limbs = fits ? nl : 0;
    // merge table of the LQ kernel's way back (hsqp_lql.h): simulate which bodies every lane's composite holds
    dm.ql_ok = 0;
    memset(dm.limb_merge, 0, sizeof(dm.limb_merge));
    if (dm.n_limbs > 0) {
      bool ok = true;
      unsigned sets[QV_LIMBS] = {0, 0, 0, 0};
      auto body_at = [&](int l, int t) { return (int)((dm.limb_path[l] >> (8 * t)) & 0xffull); };
      for (int t = dm.limb_max_len - 1; t >= 0 && ok; --t) {
        unsigned snap[QV_LIMBS];
        for (int l = 0; l < QV_LIMBS; ++l) snap[l] = sets[l];
        for (int l = 0; l < nl; ++l) {
          if (t >= dm.limb_len[l]) continue;
          const int i = body_at(l, t);
          unsigned needed = 0;
          for (int b = i + 1; b < i + dm.subtree_size[i]; ++b) needed |= 1u << b;
          unsigned have = snap[l];
          for (int k = 1; k < QV_LIMBS; ++k) {
            const int p = l ^ k;
            if (p >= nl || snap[p] == 0 || (snap[p] & ~needed) != 0 || (snap[p] & have) != 0) continue;
            dm.limb_merge[t][l] |= (unsigned char)(1u << k);
            have |= snap[p];
          }
          if (have != needed) ok = false;
          sets[l] = have | (1u << i);
        }
      }
      unsigned all = 0;
      for (int l = 0; l < nl && ok; ++l) {
        if (!(dm.limb_own[l] & 1u)) continue;
        if (all & sets[l]) ok = false;
        all |= sets[l];
      }
      if (all != ((1u << NB) - 2u)) ok = false;
      // the ancestors of a foot may not be shared with another limb (the lane that forms their columns must hold the foot's contact point)
      for (int f = 0; f < 2 && ok; ++f) {
        const int l = dm.foot_limb[f];
        for (int t = 0; t < dm.limb_len[l] && ok; ++t) {
          const int b = body_at(l, t);
          for (int l2 = 0; l2 < nl; ++l2)
            if (l2 != l && t < dm.limb_len[l2] && body_at(l2, t) == b) ok = false;
          if (b == md.contact[f].body) break;
        }
      }
      for (int l = 0; l < QV_LIMBS; ++l) {
        dm.limb_foot_step[l] = -1;
        for (int f = 0; f < 2; ++f)
          if (dm.foot_limb[f] == l)
            for (int t = 0; t < dm.limb_len[l]; ++t)
              if (body_at(l, t) == md.contact[f].body) dm.limb_foot_step[l] = t;
      }
      dm.ql_ok = ok ? 1 : 0;
    }
  }
  dm.gravity = md.gravity;
  for (int f = 0; f < 2; ++f) {
    dm.contact_body[f] = md.contact[f].body;
    memcpy(dm.contact_p[f], md.contact[f].p, sizeof(double) * 3);
    if (md.contact[f].body < 0 || md.contact[f].body >= NB) return "bad contact frame body";
  }
  const hsqp_frame* cf[10] = {&md.ankle[0], &md.ankle[1], &md.contact[0], &md.contact[1], &md.collision_p1[0],
                              &md.collision_p1[1], &md.collision_p2[0], &md.collision_p2[1], &md.knee[0], &md.knee[1]};
  for (int k = 0; k < 10; ++k) {
    if (cf[k]->body < 0 || cf[k]->body >= NB) return "bad collision frame body";
    dm.coll_body[k] = cf[k]->body;
    memcpy(dm.coll_p[k], cf[k]->p, sizeof(double) * 3);
  }
  memcpy(dm.Q, md.Q, sizeof(md.Q));
  memcpy(dm.R, md.R, sizeof(md.R));
  memcpy(dm.Qf, md.Qf, sizeof(md.Qf));
  memcpy(dm.foot_sqrt_w, md.foot_sqrt_w, sizeof(md.foot_sqrt_w));
  dm.gain_pos_z = md.gain_pos_z; dm.gain_ori = md.gain_ori; dm.gain_linvel_z = md.gain_linvel_z;
  dm.gain_linvel_xy = md.gain_linvel_xy; dm.gain_angvel = md.gain_angvel; dm.gain_linacc_z = md.gain_linacc_z;
  dm.gain_linacc_xy = md.gain_linacc_xy; dm.gain_angacc = md.gain_angacc;
  dm.friction_mu = md.friction_mu; dm.friction_reg = md.friction_reg; dm.friction_grip = md.friction_grip;
  dm.friction_hess_shift = md.friction_hess_shift;
  dm.friction_bmu = md.friction_barrier.mu; dm.friction_bdelta = md.friction_barrier.delta;
  dm.rect_x_min = md.rect_x_min; dm.rect_x_max = md.rect_x_max; dm.rect_y_min = md.rect_y_min; dm.rect_y_max = md.rect_y_max;
  dm.moment_bmu = md.moment_barrier.mu; dm.moment_bdelta = md.moment_barrier.delta;
  dm.jl_bmu = md.joint_limit_barrier.mu; dm.jl_bdelta = md.joint_limit_barrier.delta;
  dm.r_foot = md.r_foot; dm.r_knee = md.r_knee;
  dm.coll_bmu = md.collision_barrier.mu; dm.coll_bdelta = md.collision_barrier.delta;
  for (int k = 0; k < 4; ++k) {
    if (md.arm_swing_joint[k] < 0 || md.arm_swing_joint[k] >= NJ) return "bad arm swing joint index";
    dm.arm_swing_joint[k] = md.arm_swing_joint[k];
  }
  if (md.formulation == HSQP_FORM_CENTROIDAL) {
    if (md.torso.body < 0 || md.torso.body >= NB) return "bad torso frame body";
    dm.torso_body = md.torso.body;
    memcpy(dm.torso_p, md.torso.p, sizeof(double) * 3);
    memcpy(dm.torso_R, md.torso_R, sizeof(md.torso_R));
    memcpy(dm.torso_sqrt_w, md.torso_sqrt_w, sizeof(md.torso_sqrt_w));
    memcpy(dm.cent_foot_sqrt_w, md.cent_foot_sqrt_w, sizeof(md.cent_foot_sqrt_w));
    // the position rows of the task-space costs are not carried by the kernels (zero weight in the G1 task file; the reference's
    // foot position reference is a placeholder (0,0,0): CentroidalMpcEndEffectorFootCost.cpp:139)
    for (int k = 0; k < 3; ++k)
      if (md.torso_sqrt_w[k] != 0.0 || md.cent_foot_sqrt_w[k] != 0.0) return "centroidal: non-zero position weights of the torso / foot task-space costs are not supported";
    for (int i = HSQP_CNX; i < NX; ++i)
      if (md.Q[i] != 0.0 || md.Qf[i] != 0.0) return "centroidal: Q / Qf beyond the 35 centroidal states must be zero";
    for (int f = 0; f < 2; ++f)
      for (int a = 0; a < 6; ++a) {
        if (md.ext_torque_joint[f][a] < 0 || md.ext_torque_joint[f][a] >= NJ) return "bad external-torque joint index";
        dm.ext_joint[f][a] = md.ext_torque_joint[f][a];
        dm.ext_sqrt_w[f][a] = md.ext_torque_sqrt_w[f][a];
      }
  }
  if (!(dm.total_mass > 0.0)) return "total mass must be positive";
  // weights, gains and barriers: ONE validation for hsqp_create and the live updaters (hsqp_update_weights / hsqp_update_term_weights).  State
  // weights finite and >= 0; input weights finite and > 0 (the reduced Hessian Lam = R~ + B~^T S B~ of every stage has to stay positive
  // definite); every task weight / gain finite; barrier mu and delta > 0
  for (int i = 0; i < NX; ++i)
    if (!std::isfinite(md.Q[i]) || md.Q[i] < 0.0 || !std::isfinite(md.Qf[i]) || md.Qf[i] < 0.0) return "Q / Qf must be finite and >= 0";
  for (int i = 0; i < NU; ++i)
    if (!std::isfinite(md.R[i]) || !(md.R[i] > 0.0)) return "R must be finite and > 0";
  for (double v : md.foot_sqrt_w) if (!std::isfinite(v)) return "foot cost weights must be finite";
  for (double v : {md.gain_pos_z, md.gain_ori, md.gain_linvel_z, md.gain_linvel_xy, md.gain_angvel, md.gain_linacc_z, md.gain_linacc_xy, md.gain_angacc})
    if (!std::isfinite(v)) return "foot constraint gains must be finite";
  // relaxed barriers (friction cone, contact moment): mu > 0 — their rows are scaled by sqrt(p'') and carry p' / sqrt(p''); the piecewise-polynomial
  // penalties (joint limits, foot collision) may be switched off with mu = 0
  for (const hsqp_barrier* b : {&md.friction_barrier, &md.moment_barrier})
    if (!std::isfinite(b->mu) || !std::isfinite(b->delta) || !(b->mu > 0.0) || !(b->delta > 0.0)) return "friction / moment barrier: mu and delta must be finite and > 0";
  for (const hsqp_barrier* b : {&md.joint_limit_barrier, &md.collision_barrier})
    if (!std::isfinite(b->mu) || !std::isfinite(b->delta) || b->mu < 0.0 || !(b->delta > 0.0)) return "joint-limit / collision barrier: mu must be finite and >= 0, delta finite and > 0";
  return "";
}

}  // namespace hsqp
