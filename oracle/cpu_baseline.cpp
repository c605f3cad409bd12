// =====================================================================================
// TEST INFRASTRUCTURE — the TIMED CPU BASELINE of bench.py (`cpu_baseline`, kind "port").  Not part of the product path.
//
// BASELINE.md §3 / SURVEY.md §8d: the reference's CPU path (ocs2_sqp + HPIPM + CppAD tapes) cannot be built here, so the baseline
// is this repository's own arithmetic for the same iteration compiled for the host: the analytic flow-map derivatives, the
// structured RK4 sensitivity chain, the QR projection, the Riccati recursion and the value pass of wb_humanoid_mpc_amd/csrc/*.h
// (the sources of the HIP kernels, with their one-thread host context), built by bench.py ON THE MACHINE IT RUNS ON with
// `g++ -O3 -march=native -fopenmp`, and scheduled the way a CPU would be used:
//   * node-parallel LQ approximation + projection and node-parallel value pass on `inner` OpenMP threads (the reference's
//     nThreads, task.info:79), serial Riccati sweep — one instance at a time, and
//   * `outer` instances concurrently (batch across cores), each with `inner` threads.
// One iteration = what bench.py times on the GPU: LQ approximation at all nodes, projection, Riccati QP, full step, performance
// index before and after (no KKT check).  The dual-number oracle (oracle.cpp) stays the correctness reference; it is 10x slower
// than this and is reported only as a footnote.
// =====================================================================================
#include <omp.h>

#include <algorithm>
#include <memory>
#include <vector>

#include "../wb_humanoid_mpc_amd/csrc/hsqp_host.h"
#include "../wb_humanoid_mpc_amd/csrc/hsqp_riccati.h"
#include "../wb_humanoid_mpc_amd/csrc/hsqp_cent.h"
#include "../wb_humanoid_mpc_amd/csrc/hsqp_cent_lq.h"

using namespace hsqp;

namespace {
double g_t[4] = {0, 0, 0, 0};   // seconds in {LQ + projection, Riccati backward, forward + step, value pass} of thread-0 instances (profiling aid)
struct Workspaces {   // one per inner thread
  std::unique_ptr<LqWST<true>> lq{new LqWST<true>};
  std::unique_ptr<LqWST<false>> lqv{new LqWST<false>};
  std::unique_ptr<ProjWS> proj{new ProjWS};
  std::unique_ptr<StepWS> step{new StepWS};
  std::unique_ptr<CentWST<true>> clq{new CentWST<true>};
  std::unique_ptr<CentWST<false>> clqv{new CentWST<false>};
};

// one SQP iteration of one instance on `inner` threads; returns 0 or HSQP_ERR_NUMERIC.  perf = {cost, dyn, eq} before / after
int iterate_instance(const DevModel& dm, int N, double dt, const double* x_init, const double* x, const double* u, const double* par, int inner,
                     std::vector<Workspaces>& ws, RicWS& rw, std::vector<double>& rec, std::vector<double>& qp, std::vector<double>& ric,
                     std::vector<double>& ut, double* x_new, double* u_new, double* dx, double* du, double* perf_before, double* perf_after) {
  const bool cent = dm.formulation == HSQP_FORM_CENTROIDAL;
  int bad = 0;
  const bool prof = omp_get_level() == 0 || omp_get_ancestor_thread_num(1) == 0;
  double t0 = omp_get_wtime();
#pragma omp parallel for num_threads(inner) schedule(dynamic, 1) if (inner > 1)
  for (int k = 0; k < N; ++k) {
    Workspaces& w = ws[omp_get_thread_num()];
    Ctx ctx{0, 1, nullptr};
    double* r = &rec[(size_t)k * REC_SIZE];
    if (cent) cent_lq_node2<true>(ctx, dm, *w.clq, x + k * NX, u + k * NU, x + (k + 1) * NX, par + k * NP, dt, r, r + REC_MISC);
    else lq_node<true>(ctx, dm, *w.lq, x + k * NX, u + k * NU, x + (k + 1) * NX, par + k * NP, dt, r, r + REC_MISC);
    project_node(ctx, *w.proj, r, dt, &qp[(size_t)k * QP_SIZE], cent);
    if (qp[(size_t)k * QP_SIZE + QP_NUT] < 0) {
#pragma omp atomic write
      bad = 1;
    }
  }
  if (bad) return HSQP_ERR_NUMERIC;
  double t1 = omp_get_wtime();
  auto terminal = [&](const double* xx) { double c = 0; for (int i = 0; i < NX; ++i) { const double d = xx[N * NX + i] - par[N * NP + HSQP_P_XDES + i]; c += 0.5 * dm.Qf[i] * d * d; } return c; };
  double pb[3] = {terminal(x), 0, 0};
  for (int k = 0; k < N; ++k) { const double* m = &rec[(size_t)k * REC_SIZE + REC_MISC]; pb[0] += m[1]; pb[1] += m[3]; pb[2] += m[2]; }
  Ctx ctx{0, 1, nullptr};
  if (cent) riccati_backward<CNX>(ctx, rw, dm.Qf, x + N * NX, par + N * NP, qp.data(), ric.data(), N, nullptr);
  else riccati_backward(ctx, rw, dm.Qf, x + N * NX, par + N * NP, qp.data(), ric.data(), N, nullptr);
  if (!rw.ok) return HSQP_ERR_NUMERIC;
  double t2 = omp_get_wtime();
  if (cent) riccati_forward<CNX>(ctx, rw, x_init, x, qp.data(), ric.data(), N, dx, ut.data());
  else riccati_forward(ctx, rw, x_init, x, qp.data(), ric.data(), N, dx, ut.data());
  double pa0 = 0, pa1 = 0, pa2 = 0;
#pragma omp parallel for num_threads(inner) schedule(static) if (inner > 1)
  for (int k = 0; k < N; ++k) {
    Workspaces& w = ws[omp_get_thread_num()];
    Ctx c2{0, 1, nullptr};
    step_node(c2, *w.step, &qp[(size_t)k * QP_SIZE], &ric[(size_t)k * RIC_SIZE], dx + k * NX, x + k * NX, u + k * NU, 1.0, &ut[(size_t)k * NUT],
              du + k * NU, x_new + k * NX, u_new + k * NU, nullptr, &ut[(size_t)k * NUT]);
  }
  for (int i = 0; i < NX; ++i) x_new[N * NX + i] = x[N * NX + i] + dx[N * NX + i];
  double t3 = omp_get_wtime();
#pragma omp parallel for num_threads(inner) schedule(dynamic, 1) reduction(+ : pa0, pa1, pa2) if (inner > 1)
  for (int k = 0; k < N; ++k) {
    Workspaces& w = ws[omp_get_thread_num()];
    Ctx c2{0, 1, nullptr};
    double misc[8];
    if (cent) cent_lq_node2<false>(c2, dm, *w.clqv, x_new + k * NX, u_new + k * NU, x_new + (k + 1) * NX, par + k * NP, dt, nullptr, misc);
    else lq_node<false>(c2, dm, *w.lqv, x_new + k * NX, u_new + k * NU, x_new + (k + 1) * NX, par + k * NP, dt, nullptr, misc);
    pa0 += misc[1]; pa1 += misc[3]; pa2 += misc[2];
  }
  if (prof) { const double t4 = omp_get_wtime(); g_t[0] += t1 - t0; g_t[1] += t2 - t1; g_t[2] += t3 - t2; g_t[3] += t4 - t3; }
  perf_before[0] = pb[0]; perf_before[1] = pb[1]; perf_before[2] = pb[2];
  perf_after[0] = pa0 + terminal(x_new); perf_after[1] = pa1; perf_after[2] = pa2;
  return 0;
}
}  // namespace

extern "C" {

void* cpub_create(const hsqp_model_desc* md) {
  auto* dm = new DevModel;
  if (!build_dev_model(*md, *dm).empty()) { delete dm; return nullptr; }
  return dm;
}
void cpub_phase_seconds(double out[4], int reset) { for (int i = 0; i < 4; ++i) { out[i] = g_t[i]; if (reset) g_t[i] = 0; } }
void cpub_destroy(void* h) { delete static_cast<DevModel*>(h); }

// `iterations` SQP iterations of each of B instances (row-major arrays as in hsqp_problem), `outer` instances concurrently with `inner`
// node-parallel threads each.  Outputs of the LAST iteration: dx [B][N+1][58], du [B][N][35], perf [B][6].  Returns 0 or the first error.
int cpub_iterate(void* h, int B, int N, double dt, const double* x_init, const double* x, const double* u, const double* par, int outer, int inner,
                 int iterations, double* dx, double* du, double* perf) {
  const DevModel& dm = *static_cast<DevModel*>(h);
  omp_set_max_active_levels(2);
  int rc = 0;
#pragma omp parallel num_threads(outer) if (outer > 1)
  {
    std::vector<Workspaces> ws(inner);
    auto rw = std::make_unique<RicWS>();
    std::vector<double> rec((size_t)N * REC_SIZE), qp((size_t)N * QP_SIZE), ric((size_t)N * RIC_SIZE), ut((size_t)N * NUT);
    std::vector<double> xn((size_t)(N + 1) * NX), un((size_t)N * NU);
#pragma omp for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
      for (int it = 0; it < iterations; ++it) {
        const int r = iterate_instance(dm, N, dt, x_init + (size_t)b * NX, x + (size_t)b * (N + 1) * NX, u + (size_t)b * N * NU, par + (size_t)b * (N + 1) * NP, inner, ws, *rw,
                                       rec, qp, ric, ut, xn.data(), un.data(), dx + (size_t)b * (N + 1) * NX, du + (size_t)b * N * NU, perf + (size_t)b * 6, perf + (size_t)b * 6 + 3);
        if (r) {
#pragma omp atomic write
          rc = r;
        }
      }
    }
  }
  return rc;
}

}  // extern "C"
