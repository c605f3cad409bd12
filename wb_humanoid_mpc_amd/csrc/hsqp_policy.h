// Policy evaluation and joint torques: the step after the solve in the reference's control loop.
//   MPC_MRT_Interface::evaluatePolicy (feed-forward policy: linear interpolation of the optimal state / input trajectories)
//       humanoid_nmpc/humanoid_wb_mpc/src/mrt/WBMpcMrtJointController.cpp:136-147
//   computeJointTorques: tau_j = M_j [a_b; qdd_j] + nle_j - (J_l^T W_l + J_r^T W_r)_j  with a_b from computeBaseAcceleration
//       humanoid_nmpc/humanoid_common_mpc/src/pinocchio_model/DynamicsHelperFunctions.cpp:233-270
// The torques come out of the same Newton-Euler quantities as the flow map (hsqp_model.h): with the base acceleration known,
// every body's spatial acceleration is the gravity-trick acceleration plus the common base term D = {E a_ang, a_lin}, and
//   tau_j = S_j . ( sum_{i in subtree(j)} (f_i + I_i D)  -  sum_{contacts in subtree(j)} {m_c + r_c x f_c, f_c} ).
#pragma once
#include "hsqp_model.h"

namespace hsqp {

// After stage_eval<false>(ctx, dm, ws) at (q, v, qdd_j, W): joint torques tau[NJ].
template <class SW>
HSQP_HD void joint_torques(const Ctx& ctx, const DevModel& dm, const SW& ws, double* tau) {
  WG_FOR(ctx, j, NJ) {
    const int b = j + 1, jc = b + 2;
    // common spatial acceleration of the base frame about O: angular part E * (euler-rate accelerations), linear part a_lin
    double D[6];
    for (int k = 0; k < 3; ++k) {
      D[k] = ws.E[3 * k] * ws.ab[3] + ws.E[3 * k + 1] * ws.ab[4] + ws.E[3 * k + 2] * ws.ab[5];
      D[3 + k] = ws.ab[k];
    }
    double F[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const int end = b + dm.subtree_size[b];
    for (int i = b; i < end; ++i) {
      double t[6];
      inertia_apply(ws.In[i], D, t);
      for (int k = 0; k < 6; ++k) F[k] += ws.f[i][k] + t[k];
    }
    for (int f = 0; f < 2; ++f) {
      const int cb = dm.contact_body[f];
      if (cb >= b && cb < end) {
        double mom[3];
        v3_cross(ws.rP[f], ws.W + 6 * f, mom);
        for (int k = 0; k < 3; ++k) { F[k] -= ws.W[6 * f + 3 + k] + mom[k]; F[3 + k] -= ws.W[6 * f + k]; }
      }
    }
    const double* Sx = ws.S[jc];
    double s = 0.0;
    for (int k = 0; k < 6; ++k) s += Sx[k] * F[k];
    tau[j] = s;
  }
  WG_SYNC(ctx);
}

// One policy evaluation: (x, u) into the stage workspace, model evaluation, torques.
template <class SW>
HSQP_HD void policy_node(const Ctx& ctx, const DevModel& dm, SW& ws, const double* x, const double* u, double* tau) {
  stage_topology(ctx, dm, ws);
  WG_FOR(ctx, i, NV + NV + NJ + 12) {
    if (i < NV) ws.q[i] = x[i];
    else if (i < 2 * NV) ws.v[i - NV] = x[i];
    else if (i < 2 * NV + NJ) ws.qddj[i - 2 * NV] = u[12 + i - 2 * NV];
    else ws.W[i - 2 * NV - NJ] = u[i - 2 * NV - NJ];
  }
  WG_SYNC(ctx);
  stage_eval<false>(ctx, dm, ws);
  joint_torques(ctx, dm, ws, tau);
}

// Feed-forward policy at time offset s (seconds after the first node) on a uniform grid: clamped linear interpolation of the
// state (N+1 nodes) and input (N nodes) trajectories — ocs2 LinearInterpolation on the PrimalSolution time stamps.
HSQP_HD void policy_interpolate(const Ctx& ctx, const double* xt, const double* ut, int N, double dt, double s, double* x, double* u) {
  double a = s / dt;
  if (a < 0.0) a = 0.0;
  int kx = (int)a;
  if (kx > N - 1) kx = N - 1;
  double ax = a - kx;
  if (ax > 1.0) ax = 1.0;
  int ku = (int)a;
  double au = a - ku;
  if (ku > N - 2) { ku = N - 2 > 0 ? N - 2 : 0; au = N >= 2 ? (a - ku > 1.0 ? 1.0 : a - ku) : 0.0; }
  WG_FOR(ctx, i, NX + NU) {
    if (i < NX) x[i] = (1.0 - ax) * xt[(size_t)kx * NX + i] + ax * xt[(size_t)(kx + 1) * NX + i];
    else { const int c = i - NX; u[c] = N >= 2 ? (1.0 - au) * ut[(size_t)ku * NU + c] + au * ut[(size_t)(ku + 1) * NU + c] : ut[c]; }
  }
  WG_SYNC(ctx);
}

// The same on a non-uniform grid (hsqp_problem::dt_nodes; zero-length event intervals): node k of the state trajectory sits at
// t_k = sum_{i<k} dts[i], the inputs at t_0 .. t_{N-1}.  At an event time the post-event node is taken (the last node with t_k <= s).
HSQP_HD void policy_interpolate_grid(const Ctx& ctx, const double* xt, const double* ut, int N, const double* dts, double s, double* x, double* u) {
  if (s < 0.0) s = 0.0;
  double tk = 0.0;       // start of interval kx
  int kx = 0;
  while (kx < N - 1 && tk + dts[kx] <= s) { tk += dts[kx]; ++kx; }
  while (kx < N - 1 && dts[kx] == 0.0) ++kx;              // never interpolate across a jump
  const double h = dts[kx];
  double ax = h > 0.0 ? (s - tk) / h : 1.0;
  if (ax > 1.0) ax = 1.0;
  // inputs: stamps t_0 .. t_{N-1}; beyond the last stamp the last input is held
  int ku = kx;
  double au = ax;
  if (ku > N - 2) { ku = N - 2 > 0 ? N - 2 : 0; au = N >= 2 ? 1.0 : 0.0; }
  if (N >= 2 && dts[ku] == 0.0) au = 1.0;
  // node ku + 1 is a PRE-event node when interval ku + 1 is an event: it carries no optimised input of its own (du = 0 there;
  // upstream multiple_shooting::toPrimalSolution gives it the input of the node before), so the input is held up to the switch
  else if (N >= 2 && ku + 1 <= N - 1 && dts[ku + 1] == 0.0) au = 0.0;
  WG_FOR(ctx, i, NX + NU) {
    if (i < NX) x[i] = (1.0 - ax) * xt[(size_t)kx * NX + i] + ax * xt[(size_t)(kx + 1) * NX + i];
    else { const int c = i - NX; u[c] = N >= 2 ? (1.0 - au) * ut[(size_t)ku * NU + c] + au * ut[(size_t)(ku + 1) * NU + c] : ut[c]; }
  }
  WG_SYNC(ctx);
}

}  // namespace hsqp
