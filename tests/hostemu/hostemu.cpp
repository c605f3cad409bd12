// TEST INFRASTRUCTURE: compiles the kernel sources of wb_humanoid_mpc_amd/csrc for the HOST with a
// one-thread execution context (hsqp_common.h) so that the arithmetic of the HIP kernels can be
// checked against the oracle in the GPU-less build container.  Never loaded by the product.
#include "../../wb_humanoid_mpc_amd/csrc/hsqp_params.h"
#include "../../wb_humanoid_mpc_amd/csrc/hsqp_policy.h"
#include <algorithm>
#include <vector>
#include <memory>

#include "../../wb_humanoid_mpc_amd/csrc/hsqp_host.h"
#include "../../wb_humanoid_mpc_amd/csrc/hsqp_riccati.h"
#include "../../wb_humanoid_mpc_amd/csrc/hsqp_riccati_fact.h"
#include "../../wb_humanoid_mpc_amd/csrc/hsqp_cent.h"
#include "../../wb_humanoid_mpc_amd/csrc/hsqp_cent_lq.h"
#include "../../wb_humanoid_mpc_amd/csrc/hsqp_lqv.h"
#include "../../wb_humanoid_mpc_amd/csrc/hsqp_lql.h"
#include "../../wb_humanoid_mpc_amd/csrc/hsqp_scan.h"
#include "../../wb_humanoid_mpc_amd/csrc/hsqp_segment.h"

using namespace hsqp;

static int g_scan_refinements = 0;   // whole-body scan: refinement passes (HSQP_SCAN_WB_REFINEMENTS in hsqp_capi.hip; emu_set_scan_refinements)
static int g_lq_limb = 1;   // whole-body LQ approximation: limb-lane form (hsqp_lql.h) as the product runs it; 0: the phase form (lq_node<true>)
static int g_ric_fact = 1;   // whole-body serial sweep on the factors of [A~ | B~] (hsqp_riccati_fact.h) as the product runs it; 0: the dense stage (riccati_backward<58>)
static int g_scan = 0;   // centroidal formulation: backward sweep by the parallel scan (hsqp_scan.h) instead of the serial recursion

// the parallel-in-time backward sweep through the kernel sources, executed level by level as the device launches it
template <int n>
static int scan_backward(const DevModel& dm, int N, const double* x, const double* par, const double* qp, double* ric, double* vf, double* acl, int refinements) {
  Ctx ctx{0, 1, nullptr};
  using E = ScanEl<n>;
  std::vector<double> ea((size_t)(N + 1) * E::SIZE), eb((size_t)(N + 1) * E::SIZE);
  auto iw = std::make_unique<ScanInitWS<n>>();
  auto cw = std::make_unique<ScanCombWS<n>>();
  auto rw = std::make_unique<RicWS>();
  for (int k = 0; k <= N; ++k)
    scan_init_node<n>(ctx, *iw, qp + (size_t)(k < N ? k : 0) * QP_SIZE, &ea[(size_t)k * E::SIZE], k == N, dm.Qf, x + (size_t)N * NX, par + (size_t)N * NP);
  int ok = 1;
  for (int d = 1; d < N + 1; d *= 2) {
    for (int k = 0; k <= N; ++k) {
      if (k + d <= N) scan_combine<n>(ctx, *cw, &ea[(size_t)k * E::SIZE], &ea[(size_t)(k + d) * E::SIZE], &eb[(size_t)k * E::SIZE], &ok);
      else std::copy(&ea[(size_t)k * E::SIZE], &ea[(size_t)(k + 1) * E::SIZE], &eb[(size_t)k * E::SIZE]);
    }
    ea.swap(eb);
  }
  if (!ok) return 0;
  // one stage of the Riccati code per node, started from the scanned value function of node k + 1 (S = J, s = -eta), then `refinements`
  // times more from the value functions of the previous pass (the exact Riccati map contracts the scan's rounding error); the last
  // pass also leaves the closed loop of every stage for the roll-out (k_scan_gains / launch_scan in hsqp_capi.hip)
  std::vector<double> va((size_t)(N + 1) * VF_SIZE), vb((size_t)(N + 1) * VF_SIZE);
  for (int pass = 0; pass <= refinements; ++pass) {
    const bool lastp = pass == refinements;
    double* vout = lastp ? vf : ((pass & 1) ? vb.data() : va.data());
    const double* vin = pass == 0 ? nullptr : (((pass - 1) & 1) ? vb.data() : va.data());
    for (int k = 0; k < N; ++k) {
      const double* en = &ea[(size_t)(k + 1) * E::SIZE];
      const double* vn = vin ? vin + (size_t)(k + 1) * VF_SIZE : nullptr;
      riccati_backward<n>(ctx, *rw, dm.Qf, x + (size_t)N * NX, par + (size_t)N * NP, qp + (size_t)k * QP_SIZE, ric + (size_t)k * RIC_SIZE, 1,
                          vout + (size_t)k * VF_SIZE, vn ? vn : en + E::J, vn ? vn + NX * NX : en + E::ETA, k == N - 1, vn ? 1.0 : -1.0, vn ? NX : n);
      if (!rw->ok) return 0;
      if (lastp) closed_loop_record<n>(ctx, *rw, acl + (size_t)k * ACL_SIZE<n>);
    }
  }
  return 1;
}

static int g_segments = 0;   // > 0: backward sweep by the two-level (segmented) form of hsqp_segment.h with this many segments

// the segmented backward sweep through the kernel sources, pass by pass as launch_segmented (hsqp_capi.hip) runs it
template <int n>
static int segmented_backward(const DevModel& dm, int N, int P, const double* x, const double* par, const double* qp, double* ric, double* vf) {
  Ctx ctx{0, 1, nullptr};
  using E = ScanEl<n>;
  std::vector<double> ea((size_t)(P + 1) * E::SIZE), eb((size_t)(P + 1) * E::SIZE), ric2((size_t)N * RIC_SIZE), linv((size_t)N * LDB * LDB), vf0((size_t)P * VF_SIZE);
  std::vector<double> zero((size_t)NX * NX + NX, 0.0);
  auto rw = std::make_unique<RicWS>();
  auto aw = std::make_unique<SegAccWS>();
  auto cw = std::make_unique<ScanCombWS<n>>();
  const double* xN = x + (size_t)N * NX;
  const double* parN = par + (size_t)N * NP;
  for (int p = 0; p < P; ++p) {
    const int k0 = seg_bound(p, N, P), L = seg_bound(p + 1, N, P) - k0;
    riccati_backward<n>(ctx, *rw, dm.Qf, xN, parN, qp + (size_t)k0 * QP_SIZE, &ric2[(size_t)k0 * RIC_SIZE], L, &vf0[(size_t)p * VF_SIZE], zero.data(), zero.data() + n * n,
                        false, 1.0, n, &linv[(size_t)k0 * LDB * LDB], 1);
    if (!rw->ok) return 0;
    seg_accumulate<n>(ctx, *aw, qp + (size_t)k0 * QP_SIZE, &ric2[(size_t)k0 * RIC_SIZE], &linv[(size_t)k0 * LDB * LDB], L, &vf0[(size_t)p * VF_SIZE], &ea[(size_t)p * E::SIZE]);
  }
  scan_terminal_element<n>(ctx, &ea[(size_t)P * E::SIZE], dm.Qf, xN, parN);
  int ok = 1;
  for (int d = 1; d < P + 1; d *= 2) {
    for (int k = 0; k <= P; ++k) {
      if (k + d <= P) scan_combine<n>(ctx, *cw, &ea[(size_t)k * E::SIZE], &ea[(size_t)(k + d) * E::SIZE], &eb[(size_t)k * E::SIZE], &ok);
      else std::copy(&ea[(size_t)k * E::SIZE], &ea[(size_t)(k + 1) * E::SIZE], &eb[(size_t)k * E::SIZE]);
    }
    ea.swap(eb);
  }
  if (!ok) return 0;
  for (int p = 0; p < P; ++p) {
    const int k0 = seg_bound(p, N, P), L = seg_bound(p + 1, N, P) - k0;
    const double* en = &ea[(size_t)(p + 1) * E::SIZE];
    riccati_backward<n>(ctx, *rw, dm.Qf, xN, parN, qp + (size_t)k0 * QP_SIZE, ric + (size_t)k0 * RIC_SIZE, L, vf + (size_t)k0 * VF_SIZE, en + E::J, en + E::ETA, p == P - 1, -1.0, n);
    if (!rw->ok) return 0;
  }
  return 1;
}

extern "C" {

void emu_set_segments(int P) { g_segments = P; }
// debug / test access to the pieces of the two-level sweep (n = 58): the element of the stage range [k0, k0 + L) of a QP record array, and
// the combination of two elements (ScanEl<58> layout)
int emu_el_size58() { return ScanEl<NX>::SIZE; }
int emu_segment_element58(void* h, const double* qp, int k0, int L, double* el) {
  const DevModel& dm = *static_cast<DevModel*>(h);
  Ctx ctx{0, 1, nullptr};
  std::vector<double> ric2((size_t)L * RIC_SIZE), linv((size_t)L * LDB * LDB), vf0(VF_SIZE), zero((size_t)NX * NX + NX, 0.0), xN(NX, 0.0), parN(NP, 0.0);
  auto rw = std::make_unique<RicWS>();
  auto aw = std::make_unique<SegAccWS>();
  riccati_backward<NX>(ctx, *rw, dm.Qf, xN.data(), parN.data(), qp + (size_t)k0 * QP_SIZE, ric2.data(), L, vf0.data(), zero.data(), zero.data() + NX * NX, false, 1.0, NX, linv.data(), 1);
  if (!rw->ok) return 0;
  seg_accumulate<NX>(ctx, *aw, qp + (size_t)k0 * QP_SIZE, ric2.data(), linv.data(), L, vf0.data(), el);
  return 1;
}
// the scan's element of ONE stage (scan_init_node, hsqp_scan.h) from its QP record
int emu_scan_stage_element58(void* h, const double* qp_record, double* el) {
  const DevModel& dm = *static_cast<DevModel*>(h);
  Ctx ctx{0, 1, nullptr};
  auto iw = std::make_unique<ScanInitWS<NX>>();
  std::vector<double> xN(NX, 0.0), parN(NP, 0.0);
  int ok = 1;
  scan_init_node<NX>(ctx, *iw, qp_record, el, false, dm.Qf, xN.data(), parN.data(), &ok);
  return ok;
}
int emu_scan_combine58(const double* e1, const double* e2, double* out) {
  Ctx ctx{0, 1, nullptr};
  auto cw = std::make_unique<ScanCombWS<NX>>();
  int ok = 1;
  scan_combine<NX>(ctx, *cw, e1, e2, out, &ok);
  return ok;
}
// The blocked (matrix-core) factorisation of the Riccati stage (hsqp_elim.h) on the host's 64-lane wave emulation: both waves one after the
// other on a workspace filled with Lam (23 x 23, symmetric positive definite), G (23 x nxe), g (23); returns L^-1 (23 x 23), Z (23 x nxe), z (23)
int emu_eliminate_blocked(int nxe, const double* lam, const double* G, const double* g, double* linv, double* linvT, double* Z, double* z) {
  auto rw = std::make_unique<RicWS>();
  RicWS& w = *rw;
  memset(&w, 0, sizeof(RicWS));
  for (int r = 0; r < NUT; ++r) {
    for (int c = 0; c < NUT; ++c) w.fac.Ef[r][c] = lam[r * NUT + c];
    for (int c = 0; c < nxe; ++c) w.Em[r][EM_G + c] = G[r * nxe + c];
    w.Em[r][EM_GVP] = g[r];
  }
  for (int r = 0; r < LDB; ++r) for (int c = 0; c < LDF; ++c) if (r >= NUT || c >= NUT) w.fac.Ef[r][c] = 1e300;   // padding must not be read as data
  w.ok = 1;
  const ElimIO io{&w.fac.Ef[0][0], LDF, &w.Em[0][EM_G], &w.Em[0][EM_GVP], LDE, &w.fac.Ef[0][EF_MI], LDF, &w.fac.LinvT[0][0], LDB, &w.Zs[0][0], LDZ, w.zv, &w.ok};
  const HostWave hw;
  if (nxe == NX) { eliminate_blocked<NX, 0>(hw, io); eliminate_blocked<NX, 1>(hw, io); }
  else if (nxe == 35) { eliminate_blocked<35, 0>(hw, io); eliminate_blocked<35, 1>(hw, io); }
  else return -1;
  for (int r = 0; r < NUT; ++r) {
    for (int c = 0; c < NUT; ++c) { linv[r * NUT + c] = w.fac.Ef[r][EF_MI + c]; linvT[r * NUT + c] = w.fac.LinvT[r][c]; }
    for (int c = 0; c < nxe; ++c) Z[r * nxe + c] = w.Zs[r][c];
    z[r] = w.zv[r];
    if (w.Zs[r][nxe] != w.zv[r]) return -2;
  }
  return w.ok;
}
int emu_scan_gate_accepts(double r_stat, double r_prim, double g_inf, int flags) { return scan_gate_accepts(r_stat, r_prim, g_inf, flags) ? 1 : 0; }
void emu_set_scan(int on) { g_scan = on; }
void emu_set_ric_fact(int on) { g_ric_fact = on; }
void emu_set_scan_refinements(int r) { g_scan_refinements = r; }

// Gauss-Jordan of hsqp_scan.h on a 35 x 106 system [M | RHS] (row-major, leading dimension 106; the shape of the combination step):
// X = M^-1 RHS (35 x 71); returns ok
int emu_gauss_jordan35(const double* Gin, int pivot, double* X) {
  constexpr int n = 35, ncol = 3 * n + 1;
  std::vector<double> G(Gin, Gin + (size_t)n * ncol);
  GjWS g;
  Ctx ctx{0, 1, nullptr};
  if (pivot) gauss_jordan<n, ncol, ncol, true>(ctx, G.data(), g);
  else gauss_jordan<n, ncol, ncol, false>(ctx, G.data(), g);
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < ncol - n; ++c) X[r * (ncol - n) + c] = G[(size_t)g.piv[r] * ncol + n + c] / G[(size_t)g.piv[r] * ncol + r];
  return g.ok;
}

void* emu_create(const hsqp_model_desc* md, char* err, int errlen) {
  DevModel* dm = new DevModel;
  std::string e = build_dev_model(*md, *dm);
  if (!e.empty()) { snprintf(err, errlen, "%s", e.c_str()); delete dm; return nullptr; }
  return dm;
}
void emu_destroy(void* h) { delete static_cast<DevModel*>(h); }

// base acceleration and its Jacobian (6 x 93) at (x, u)
void emu_stage_eval(void* h, const double* x, const double* u, int deriv, double* ab, double* G) {
  const DevModel& dm = *static_cast<DevModel*>(h);
  Ctx ctx{0, 1, nullptr};
  auto run = [&](auto& ws) {
    for (int i = 0; i < NV; ++i) { ws.q[i] = x[i]; ws.v[i] = x[NV + i]; }
    for (int i = 0; i < 12; ++i) ws.W[i] = u[i];
    for (int i = 0; i < NJ; ++i) ws.qddj[i] = u[12 + i];
  };
  if (deriv) {
    auto ws = std::make_unique<StageWST<true>>();
    run(*ws);
    stage_topology(ctx, dm, *ws);
    stage_eval<true>(ctx, dm, *ws);
    for (int i = 0; i < 6; ++i) ab[i] = ws->ab[i];
    if (G) for (int r = 0; r < 6; ++r) for (int c = 0; c < NZ; ++c) G[r * NZ + c] = ws->G[r][c];
  } else {
    auto ws = std::make_unique<StageWST<false>>();
    run(*ws);
    stage_topology(ctx, dm, *ws);
    stage_eval<false>(ctx, dm, *ws);
    for (int i = 0; i < 6; ++i) ab[i] = ws->ab[i];
  }
}


// the limb-lane form of the whole-body LQ approximation (hsqp_lql.h): the four lanes of a node one after the other (k_lq_limb), then the chain /
// defect pass (k_lq_chain); rec: REC_SIZE doubles, ZERO-FILLED by the caller (entries that are zero for every state are never written)
static double g_defect_mismatch = 0.0;   // max |b (lanes) - b (chain kernel)| of the last limb-lane node
double emu_defect_mismatch() { return g_defect_mismatch; }
void emu_set_lq_limb(int on) { g_lq_limb = on; }
int emu_ql_ok(void* h) { return static_cast<DevModel*>(h)->ql_ok; }
static void lq_limb_node(const DevModel& dm, const double* x, const double* u, const double* xnext, const double* par, double dt, double* rec) {
  Ctx ctx{0, 1, nullptr};
  ql_node_host(dm, x, u, dt, rec);
  ql_rows_host(dm, x, u, par, dt, rec, xnext);   // the defect on the lanes, as the product forms it when the chain is fused into k_project
  // the columns' chain (the device: inside k_project, lq_chain_column_pv): P6, V6 only — the chain kernel's defect must not overwrite the lanes' (the tests compare THAT with the oracle)
  std::vector<double> tmp(rec, rec + REC_SIZE);
  auto cw = std::make_unique<LqChainWS>();
  lq_chain_node(ctx, *cw, x, u, xnext, dt, tmp.data());
  for (int i = 0; i < 2 * 6 * LDJ; ++i) rec[REC_PV + i] = tmp[REC_PV + i];
  g_defect_mismatch = 0.0;
  for (int i = 0; i < 64; ++i) { const double d = std::fabs(tmp[REC_B + i] - rec[REC_B + i]); if (d > g_defect_mismatch) g_defect_mismatch = d; }
}

// LQ record of one node (REC_SIZE doubles) + dense expansions for comparison with the oracle
int emu_rec_size() { return REC_SIZE; }
int emu_rec_misc_offset() { return REC_MISC; }
int emu_rec_flow_offset() { return REC_FLOW; }
int emu_rec_gs_offset() { return REC_GS; }
void emu_lq_node(void* h, const double* x, const double* u, const double* xnext, const double* par, double dt, int deriv, double* rec) {
  const DevModel& dm = *static_cast<DevModel*>(h);
  Ctx ctx{0, 1, nullptr};
  if (deriv && g_lq_limb && dm.ql_ok) {   // the product's form for handles that fill the GPU
    for (int i = 0; i < REC_SIZE; ++i) rec[i] = 0.0;
    lq_limb_node(dm, x, u, xnext, par, dt, rec);
  } else if (deriv) { auto w = std::make_unique<LqWST<true>>(); lq_node<true>(ctx, dm, *w, x, u, xnext, par, dt, rec, rec + REC_MISC); }
  else { auto w = std::make_unique<LqWST<false>>(); lq_node<false>(ctx, dm, *w, x, u, xnext, par, dt, nullptr, rec + REC_MISC); }
}
// the limb tables of the quad value pass (build_dev_model): out = {n_limbs, max_len, foot_limb[2], len[4], times every body 1 .. NB-1 is owned (NB - 1 entries)}
void emu_limbs(void* h, int* out) {
  const DevModel& dm = *static_cast<DevModel*>(h);
  out[0] = dm.n_limbs; out[1] = dm.limb_max_len; out[2] = dm.foot_limb[0]; out[3] = dm.foot_limb[1];
  for (int l = 0; l < QV_LIMBS; ++l) out[4 + l] = dm.limb_len[l];
  for (int b = 1; b < NB; ++b) out[8 + b - 1] = 0;
  for (int l = 0; l < dm.n_limbs; ++l)
    for (int k = 0; k < dm.limb_len[l]; ++k)
      if ((dm.limb_own[l] >> k) & 1u) out[8 + (int)((dm.limb_path[l] >> (8 * k)) & 0xffull) - 1] += 1;
}
// the value pass on a quad of lanes (hsqp_lqv.h), lane by lane: misc[8]; returns the number of limbs (0: the model does not fit the form)
int emu_value_quad(void* h, const double* x, const double* u, const double* xnext, const double* par, double dt, double* misc) {
  const DevModel& dm = *static_cast<DevModel*>(h);
  if (dm.n_limbs == 0) return 0;
  qv_node_host(dm, x, u, xnext, par, dt, misc);
  return dm.n_limbs;
}
// dense blocks from a record: AB[58*93], H[93*93], g[93], CDe[14*94]
void emu_expand(const double* rec, double dt, double* AB, double* H, double* g, double* CDe) {
  expand_AB(rec, dt, AB);
  const int nrows = (int)rec[REC_NROWS];   // compact residual rows (hsqp_node.h); rows beyond them are not part of the model
  for (int a = 0; a < NZ; ++a) {
    double ga = rec[REC_GD + a];
    for (int r = 0; r < nrows; ++r) ga += rec_J_at(rec, r, a) * rec[REC_RHO + r];
    g[a] = ga;
    for (int b = 0; b < NZ; ++b) {
      double s = a == b ? rec[REC_D + a] : 0.0;
      for (int r = 0; r < nrows; ++r) s += rec_J_at(rec, r, a) * rec_J_at(rec, r, b);
      H[a * NZ + b] = s;
    }
  }
  for (int r = 0; r < NE_MAX; ++r) for (int c = 0; c <= NZ; ++c) CDe[r * (NZ + 1) + c] = rec_CDe_at(rec, r, c);
}

// per-node parameter table of one instance through the device-side generator (hsqp_params.h); returns 0 if every swing
// phase was bracketed
int emu_node_params(const hsqp_swing_config* cfg, double terrain, int arm_swing, int n_ev, const double* ev, const int* seq, int n_knots,
                    const double* tt, const double* ts, double t0, double dt, int N, double* par) {
  int bad = 0;
  for (int k = 0; k <= N; ++k)
    if (!node_params_eval(*cfg, terrain, arm_swing, n_ev, ev, seq, n_knots, tt, ts, t0 + k * dt, par + (size_t)k * NP)) bad = 1;
  return bad;
}

// the same for the centroidal formulation: the torso task-space reference is appended from the device code's tree pass
int emu_cent_node_params(void* h, const hsqp_swing_config* cfg, double terrain, int arm_swing, int n_ev, const double* ev, const int* seq, int n_knots,
                         const double* tt, const double* ts, double t0, double dt, int N, double* par) {
  const DevModel& dm = *static_cast<DevModel*>(h);
  int bad = 0;
  for (int k = 0; k <= N; ++k) {
    if (!node_params_eval(*cfg, terrain, arm_swing, n_ev, ev, seq, n_knots, tt, ts, t0 + k * dt, par + (size_t)k * NP)) bad = 1;
    { Ctx ctx{0, 1, nullptr}; auto ws = std::make_unique<CentWST<false>>(); cent_params_torso(ctx, dm, *ws, par + (size_t)k * NP); }
  }
  return bad;
}

// joint torques of one (x, u) pair through the device code path (hsqp_policy.h)
void emu_joint_torques(void* h, const double* x, const double* u, double* tau) {
  const DevModel& dm = *static_cast<DevModel*>(h);
  Ctx ctx{0, 1, nullptr};
  auto ws = std::make_unique<StageWST<false>>();
  policy_node(ctx, dm, *ws, x, u, tau);
}
void emu_policy_interpolate_grid(const double* xt, const double* ut, int N, const double* dts, double s, double* x, double* u) {
  Ctx ctx{0, 1, nullptr};
  policy_interpolate_grid(ctx, xt, ut, N, dts, s, x, u);
}
void emu_policy_interpolate(const double* xt, const double* ut, int N, double dt, double s, double* x, double* u) {
  Ctx ctx{0, 1, nullptr};
  policy_interpolate(ctx, xt, ut, N, dt, s, x, u);
}

int emu_qp_size() { return QP_SIZE; }
int emu_ric_size() { return RIC_SIZE; }
void emu_project_node(const double* rec, double dt, double* qp) {
  auto w = std::make_unique<ProjWS>();
  Ctx ctx{0, 1, nullptr};
  project_node(ctx, *w, rec, dt, qp);
}
// ---- centroidal formulation (hsqp_cent.h): padded layout, x rows of 58 doubles
void emu_cent_lq_node(void* h, const double* x, const double* u, const double* xnext, const double* par, double dt, int deriv, double* rec) {
  const DevModel& dm = *static_cast<DevModel*>(h);
  Ctx ctx{0, 1, nullptr};
  if (deriv) { auto ws = std::make_unique<CentWST<true>>(); cent_lq_node2<true>(ctx, dm, *ws, x, u, xnext, par, dt, rec, rec + REC_MISC); }
  else { auto ws = std::make_unique<CentWST<false>>(); cent_lq_node2<false>(ctx, dm, *ws, x, u, xnext, par, dt, nullptr, rec + REC_MISC); }
}
void emu_cent_expand_AB(const double* rec, double dt, double* AB) { cent_expand_AB(rec, dt, AB); }
// one full SQP iteration of one instance through the kernel sources; returns 0 or HSQP_ERR_NUMERIC
int emu_sqp_iteration(void* h, int N, double dt, const double* x_init, const double* x, const double* u, const double* par,
                      double* x_new, double* u_new, double* dx, double* du, double* kkt, double* perf_before /*3: cost,dyn,eq*/,
                      double* perf_after, double* qp_out, const double* dts /*null: uniform dt; else [N] interval lengths, 0 = event*/) {
  const DevModel& dm = *static_cast<DevModel*>(h);
  const bool cent = dm.formulation == HSQP_FORM_CENTROIDAL;
  Ctx ctx{0, 1, nullptr};
  std::vector<double> rec((size_t)N * REC_SIZE), qp((size_t)N * QP_SIZE), ric((size_t)N * RIC_SIZE, std::nan("")), ut((size_t)N * NUT);   // ric poisoned: parts nobody writes must not be read
  auto lw = std::make_unique<LqWST<true>>();
  auto lwv = std::make_unique<LqWST<false>>();
  auto pw = std::make_unique<ProjWS>();
  auto rw = std::make_unique<RicWS>();
  double pb[3] = {0, 0, 0};
  const double dt_uniform = dt;
  const bool fact = !cent && g_ric_fact && !g_scan && !g_segments;
  std::vector<double> dtv(N);
  for (int k = 0; k < N; ++k) dtv[k] = dts ? dts[k] : dt_uniform;
  auto fw = std::make_unique<RicFWS>();
  std::vector<double> fj((size_t)N * NJ, std::nan(""));   // rows 12 .. 34 of Px dx + Pu ut as the factored roll-out leaves them (step_node then skips those rows of Px / Pu)
  for (int k = 0; k < N; ++k) {
    const double dt = dts ? dts[k] : dt_uniform;
    if (cent) { auto cw = std::make_unique<CentWST<true>>(); double* r = &rec[(size_t)k * REC_SIZE]; cent_lq_node2<true>(ctx, dm, *cw, x + k * NX, u + k * NU, x + (k + 1) * NX, par + k * NP, dt, r, r + REC_MISC); }
    else if (g_lq_limb && dm.ql_ok) {
      lq_limb_node(dm, x + k * NX, u + k * NU, x + (k + 1) * NX, par + k * NP, dt, &rec[(size_t)k * REC_SIZE]);   // (rec: zero-filled vector)
    } else lq_node<true>(ctx, dm, *lw, x + k * NX, u + k * NU, x + (k + 1) * NX, par + k * NP, dt, &rec[(size_t)k * REC_SIZE], &rec[(size_t)k * REC_SIZE + REC_MISC]);
    // (the factored sweep reads the dense base rows, Px, Pu, b~ only: the joint rows of A~ / B~ stay unwritten — NaN here — until the KKT check below)
    if (fact) { for (int i = 0; i < QP_BV; ++i) qp[(size_t)k * QP_SIZE + i] = std::nan(""); for (int i = QP_Q; i < QP_P; ++i) qp[(size_t)k * QP_SIZE + i] = std::nan(""); }   // (also Q~: its strictly lower triangle stays unwritten)
    project_node(ctx, *pw, &rec[(size_t)k * REC_SIZE], dt, &qp[(size_t)k * QP_SIZE], cent, !fact);
    if (dt == 0.0) jump_node_qp(ctx, &rec[(size_t)k * REC_SIZE], &qp[(size_t)k * QP_SIZE]);
    if (qp[(size_t)k * QP_SIZE + QP_NUT] < 0) return HSQP_ERR_NUMERIC;
    pb[0] += rec[(size_t)k * REC_SIZE + REC_MISC + 1]; pb[1] += rec[(size_t)k * REC_SIZE + REC_MISC + 3]; pb[2] += rec[(size_t)k * REC_SIZE + REC_MISC + 2];
  }
  auto terminal = [&](const double* xx) { double c = 0; for (int i = 0; i < NX; ++i) { const double d = xx[N * NX + i] - par[N * NP + HSQP_P_XDES + i]; c += 0.5 * dm.Qf[i] * d * d; } return c; };
  pb[0] += terminal(x);
  std::vector<double> vf((size_t)(N + 1) * VF_SIZE);
  std::vector<double> acl(g_scan ? (size_t)N * ACL_SIZE<NX> : 0);
  if (g_segments > 0) {
    const int okk = cent ? segmented_backward<CNX>(dm, N, g_segments, x, par, qp.data(), ric.data(), vf.data()) : segmented_backward<NX>(dm, N, g_segments, x, par, qp.data(), ric.data(), vf.data());
    if (!okk) return HSQP_ERR_NUMERIC;
    rw->ok = 1;
  }
  else if (g_scan) {
    const int okk = cent ? scan_backward<CNX>(dm, N, x, par, qp.data(), ric.data(), vf.data(), acl.data(), 1) : scan_backward<NX>(dm, N, x, par, qp.data(), ric.data(), vf.data(), acl.data(), g_scan_refinements);
    if (!okk) return HSQP_ERR_NUMERIC;
    rw->ok = 1;
  }
  else if (cent) riccati_backward<CNX>(ctx, *rw, dm.Qf, x + N * NX, par + N * NP, qp.data(), ric.data(), N, vf.data());
  else if (fact) { riccati_backward_fact(ctx, *fw, dm.Qf, x + N * NX, par + N * NP, qp.data(), dtv.data(), ric.data(), N, vf.data()); rw->ok = fw->ok; }
  else riccati_backward(ctx, *rw, dm.Qf, x + N * NX, par + N * NP, qp.data(), ric.data(), N, vf.data());
  if (!rw->ok) return HSQP_ERR_NUMERIC;
  if (cent && g_scan && !g_segments) closed_loop_forward<CNX>(ctx, *rw, x_init, x, acl.data(), N, dx);   // k_scan_forward
  else if (g_scan && !g_segments) closed_loop_forward<NX>(ctx, *rw, x_init, x, acl.data(), N, dx);
  else if (cent) riccati_forward<CNX>(ctx, *rw, x_init, x, qp.data(), ric.data(), N, dx, ut.data());
  else if (fact) {
    riccati_forward_fact(ctx, *fw, x_init, x, qp.data(), dtv.data(), ric.data(), N, dx, ut.data(), fj.data());
    for (int k = 0; k < N; ++k) {   // the dense joint rows for the KKT check (what k_project writes when the report is asked for)
      project_node(ctx, *pw, &rec[(size_t)k * REC_SIZE], dtv[k], &qp[(size_t)k * QP_SIZE], cent, true);
      if (dtv[k] == 0.0) jump_node_qp(ctx, &rec[(size_t)k * REC_SIZE], &qp[(size_t)k * QP_SIZE]);
    }
  }
  else riccati_forward(ctx, *rw, x_init, x, qp.data(), ric.data(), N, dx, ut.data());
  const bool ut_given = !(g_scan && !g_segments);   // the serial roll-out leaves ut = k + K dx of every node (as k_riccati / k_ric_forward do)
  auto sw = std::make_unique<StepWS>();
  for (int k = 0; k < N; ++k)
    step_node(ctx, *sw, &qp[(size_t)k * QP_SIZE], &ric[(size_t)k * RIC_SIZE], dx + k * NX, x + k * NX, u + k * NU, 1.0, &ut[(size_t)k * NUT],
              du + k * NU, x_new + k * NX, u_new + k * NU, nullptr, ut_given ? &ut[(size_t)k * NUT] : nullptr, fact ? &fj[(size_t)k * NJ] : nullptr);
  for (int i = 0; i < NX; ++i) x_new[N * NX + i] = x[N * NX + i] + dx[N * NX + i];
  auto kw = std::make_unique<KktWS>();
  kkt[0] = kkt[1] = 0.0;
  std::vector<double> dx0(NX);
  for (int i = 0; i < NX; ++i) dx0[i] = x_init[i] - x[i];
  for (int k = 0; k < N; ++k) {
    double r2[2];
    kkt_node(ctx, *kw, &qp[(size_t)k * QP_SIZE], &vf[(size_t)k * VF_SIZE], &vf[(size_t)(k + 1) * VF_SIZE], dx + k * NX, dx + (k + 1) * NX,
             &ut[(size_t)k * NUT], k == 0 ? dx0.data() : nullptr, r2);
    kkt[0] = std::max(kkt[0], r2[0]); kkt[1] = std::max(kkt[1], r2[1]);
  }
  double pa[3] = {0, 0, 0};
  std::vector<double> r2(REC_SIZE);
  for (int k = 0; k < N; ++k) {
    const double dt = dts ? dts[k] : dt_uniform;
    if (cent) { auto cw = std::make_unique<CentWST<false>>(); cent_lq_node2<false>(ctx, dm, *cw, x_new + k * NX, u_new + k * NU, x_new + (k + 1) * NX, par + k * NP, dt, nullptr, r2.data() + REC_MISC); }
    else lq_node<false>(ctx, dm, *lwv, x_new + k * NX, u_new + k * NU, x_new + (k + 1) * NX, par + k * NP, dt, nullptr, r2.data() + REC_MISC);
    pa[0] += r2[REC_MISC + 1]; pa[1] += r2[REC_MISC + 3]; pa[2] += r2[REC_MISC + 2];
  }
  pa[0] += terminal(x_new);
  for (int i = 0; i < 3; ++i) { perf_before[i] = pb[i]; perf_after[i] = pa[i]; }
  if (qp_out) for (size_t i = 0; i < qp.size(); ++i) qp_out[i] = qp[i];
  return 0;
}
}
