// TEST INFRASTRUCTURE: compiles the kernel sources of wb_humanoid_mpc_amd/csrc for the HOST with a
// one-thread execution context (hsqp_common.h) so that the arithmetic of the HIP kernels can be
// checked against the oracle in the GPU-less build container.  Never loaded by the product.
#include <vector>
#include <memory>

#include "../../wb_humanoid_mpc_amd/csrc/hsqp_host.h"
#include "../../wb_humanoid_mpc_amd/csrc/hsqp_lq.h"

using namespace hsqp;

extern "C" {

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
  auto ws = std::make_unique<StageWS>();
  Ctx ctx{0, 1};
  for (int i = 0; i < NV; ++i) { ws->q[i] = x[i]; ws->v[i] = x[NV + i]; }
  for (int i = 0; i < 12; ++i) ws->W[i] = u[i];
  for (int i = 0; i < NJ; ++i) ws->qddj[i] = u[12 + i];
  if (deriv) stage_eval<true>(ctx, dm, *ws); else stage_eval<false>(ctx, dm, *ws);
  for (int i = 0; i < 6; ++i) ab[i] = ws->ab[i];
  if (deriv && G) for (int r = 0; r < 6; ++r) for (int c = 0; c < NZ; ++c) G[r * NZ + c] = ws->G[r][c];
}


// LQ record of one node (REC_SIZE doubles) + dense expansions for comparison with the oracle
int emu_rec_size() { return REC_SIZE; }
void emu_lq_node(void* h, const double* x, const double* u, const double* xnext, const double* par, double dt, int deriv, double* rec) {
  const DevModel& dm = *static_cast<DevModel*>(h);
  auto w = std::make_unique<LqWS>();
  Ctx ctx{0, 1};
  if (deriv) lq_node<true>(ctx, dm, *w, x, u, xnext, par, dt, rec); else lq_node<false>(ctx, dm, *w, x, u, xnext, par, dt, rec);
}
// dense blocks from a record: AB[58*93], H[93*93], g[93], CDe[14*94]
void emu_expand(const double* rec, double dt, double* AB, double* H, double* g, double* CDe) {
  expand_AB(rec, dt, AB);
  for (int a = 0; a < NZ; ++a) {
    double ga = rec[REC_GD + a];
    for (int r = 0; r < NRS; ++r) ga += rec[REC_J + r * LDJ + a] * rec[REC_RHO + r];
    g[a] = ga;
    for (int b = 0; b < NZ; ++b) {
      double s = a == b ? rec[REC_D + a] : 0.0;
      for (int r = 0; r < NRS; ++r) s += rec[REC_J + r * LDJ + a] * rec[REC_J + r * LDJ + b];
      H[a * NZ + b] = s;
    }
  }
  for (int r = 0; r < NE_MAX; ++r) for (int c = 0; c <= NZ; ++c) CDe[r * (NZ + 1) + c] = rec[REC_CDE + r * LDJ + c];
}
}
