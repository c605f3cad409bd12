// C++ host side of the drop-in boundary: a thin RAII class over the C ABI (include/hsqp.h) with the method names
// and meanings of the surface the reference consumes from its solver —
//   ocs2::SolverBase::reset / run / getPrimalSolution / getPerformanceIndeces   (lib/ocs2_ros2, missing submodule;
//     used through humanoid_nmpc/humanoid_wb_mpc_ros2/src/WBMpcSqpNode.cpp:64-89 and
//     humanoid_nmpc/humanoid_wb_mpc/src/mrt/WBMpcMrtJointController.cpp:200-213)
//   SqpSolver::getBenchmarks() {linearQuadraticApproximationTime, solveQpTime, linesearchTime, computeControllerTime}
//     (humanoid_nmpc/humanoid_common_mpc_ros2/src/benchmarks/SqpBenchmarksPublisher.cpp:44-57)
// for a batch of independent MPC instances.  It has no dependency on ocs2; the ocs2 adaptor that derives from
// ocs2::SolverBase and owns one of these is shown in INTEGRATION.md.  Failures map to std::runtime_error, which is
// what the reference's MPC thread expects from its solver (WBMpcMrtJointController.cpp:210-213).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/hsqp.h"

namespace hsqp_host {

struct Benchmarks {
  double linearQuadraticApproximationTime = 0.0, solveQpTime = 0.0, linesearchTime = 0.0, computeControllerTime = 0.0;
};

struct PrimalSolution {
  int batch = 0, nodes = 0;
  std::vector<double> stateTrajectory;   // [batch][nodes + 1][HSQP_NX]
  std::vector<double> inputTrajectory;   // [batch][nodes][HSQP_NU]
};

class HipSqpSolver {
 public:
  /** useLinesearch: run() applies the filter line search (ocs2 SqpSolver::takeStep) instead of the full step.
   *  recedingHorizon: the handle solves the SHIFTED problem of one MPC loop every cycle (MPC_BASE::run, sqpIteration = 1), so the KKT gate's back-off
   *  carries over the uploads (hsqp_set_scan_backoff_persistent).  Off by default: a batch / offline user uploads unrelated problems, and for those
   *  include/hsqp.h documents per-problem determinism (the same problem takes the same sweep whatever the handle solved before). */
  HipSqpSolver(const hsqp_model_desc& model, int maxNodes, int maxBatch = 1, int device = 0, bool useLinesearch = false, bool recedingHorizon = false) {
    hsqp_settings st{maxNodes, maxBatch, device, useLinesearch ? HSQP_FLAG_LINESEARCH : 0};
    int rc = hsqp_create(&model, &st, &h_);
    if (rc != HSQP_OK) throw std::runtime_error("[HipSqpSolver] hsqp_create failed (" + std::to_string(rc) + "): " + hsqp_last_error(nullptr));
    if (recedingHorizon && (rc = hsqp_set_scan_backoff_persistent(h_, 1)) != HSQP_OK) {
      const std::string msg = hsqp_last_error(h_);
      hsqp_destroy(h_);
      throw std::runtime_error("[HipSqpSolver] hsqp_set_scan_backoff_persistent failed (" + std::to_string(rc) + "): " + msg);
    }
  }
  ~HipSqpSolver() { hsqp_destroy(h_); }
  HipSqpSolver(const HipSqpSolver&) = delete;
  HipSqpSolver& operator=(const HipSqpSolver&) = delete;

  /** SolverBase::reset: forget the previous solution (the next run is a cold start supplied by the caller). */
  void reset() { solution_ = PrimalSolution(); perf_.clear(); }

  /**
   * SolverBase::run for `batch` instances on a uniform grid of `nodes` intervals: one SQP iteration
   * (task.info: sqpIteration 1) from the linearisation trajectory (xTraj, uTraj); xInit is the measured state.
   */
  void run(int batch, int nodes, double dt, const double* xInit, const double* xTraj, const double* uTraj, const double* nodeParams) {
    hsqp_problem p{batch, nodes, dt, xInit, xTraj, uTraj, nodeParams};
    solution_.batch = batch; solution_.nodes = nodes;
    solution_.stateTrajectory.assign((size_t)batch * (nodes + 1) * HSQP_NX, 0.0);
    solution_.inputTrajectory.assign((size_t)batch * nodes * HSQP_NU, 0.0);
    perf_.assign(batch, hsqp_perf{});
    perfBefore_.assign(batch, hsqp_perf{});
    kkt_.assign((size_t)batch * 2, 0.0);
    stepSize_.assign(batch, 0.0);
    stepType_.assign(batch, HSQP_STEP_FULL);
    hsqp_solution s{};
    s.alpha = stepSize_.data(); s.step_type = stepType_.data();
    s.x = solution_.stateTrajectory.data(); s.u = solution_.inputTrajectory.data();
    s.perf_before = perfBefore_.data(); s.perf_after = perf_.data(); s.kkt = reportKkt_ ? kkt_.data() : nullptr;
    const int rc = hsqp_solve(h_, &p, &s);
    if (rc != HSQP_OK) throw std::runtime_error("[HipSqpSolver] hsqp_solve failed (" + std::to_string(rc) + "): " + hsqp_last_error(h_));
    bench_.linearQuadraticApproximationTime = s.timings.lq_approximation;
    bench_.solveQpTime = s.timings.solve_qp;
    bench_.linesearchTime = s.timings.linesearch;
    bench_.computeControllerTime = s.timings.compute_controller;
  }

  /**
   * Same as run(), but the per-node parameter table is generated on the device from the compact reference — the adaptor hands
   * over the ocs2::ModeSchedule (event times, mode sequence) and the TargetTrajectories knots instead of sampling the reference
   * manager and the swing planner for every node (hsqp_upload_reference).
   */
  void runWithReference(int nodes, double dt, const double* xInit, const double* xTraj, const double* uTraj, const hsqp_reference& ref,
                        bool takeStepWithLinesearch = false, const double* dtNodes = nullptr /* non-uniform grid with event nodes: [batch][nodes] interval
                        lengths (0 = event), together with ref.node_times */,
                        int maxIterations = 1 /* sqp::Settings::sqpIteration: all of them in ONE device call, ended early by ocs2's step-size test */) {
    const int batch = ref.batch;
    hsqp_problem p{batch, nodes, dt, xInit, xTraj, uTraj, nullptr, dtNodes};
    int rc = hsqp_upload_reference(h_, &p, &ref);
    if (rc != HSQP_OK) throw std::runtime_error("[HipSqpSolver] hsqp_upload_reference failed (" + std::to_string(rc) + "): " + hsqp_last_error(h_));
    rc = hsqp_iterate_device(h_, maxIterations < 1 ? 1 : maxIterations,
                             HSQP_ITER_TAKE_STEP | (reportKkt_ ? HSQP_ITER_KKT : 0) | (takeStepWithLinesearch ? HSQP_ITER_LINESEARCH : 0) | HSQP_ITER_UNTIL_CONVERGED);
    if (rc != HSQP_OK) throw std::runtime_error("[HipSqpSolver] hsqp_iterate_device failed (" + std::to_string(rc) + "): " + hsqp_last_error(h_));
    iterations_ = hsqp_last_iterations(h_);
    solution_.batch = batch; solution_.nodes = nodes;
    solution_.stateTrajectory.assign((size_t)batch * (nodes + 1) * HSQP_NX, 0.0);
    solution_.inputTrajectory.assign((size_t)batch * nodes * HSQP_NU, 0.0);
    perf_.assign(batch, hsqp_perf{}); perfBefore_.assign(batch, hsqp_perf{});
    kkt_.assign((size_t)batch * 2, 0.0); stepSize_.assign(batch, 0.0); stepType_.assign(batch, HSQP_STEP_FULL);
    hsqp_solution s{};
    s.x = solution_.stateTrajectory.data(); s.u = solution_.inputTrajectory.data();
    s.perf_before = perfBefore_.data(); s.perf_after = perf_.data(); s.kkt = reportKkt_ ? kkt_.data() : nullptr;
    s.alpha = stepSize_.data(); s.step_type = stepType_.data();
    rc = hsqp_download(h_, &s);
    if (rc != HSQP_OK) throw std::runtime_error("[HipSqpSolver] hsqp_download failed (" + std::to_string(rc) + "): " + hsqp_last_error(h_));
    bench_.linearQuadraticApproximationTime = s.timings.lq_approximation;
    bench_.solveQpTime = s.timings.solve_qp;
    bench_.linesearchTime = s.timings.linesearch;
    bench_.computeControllerTime = s.timings.compute_controller;
  }

  /** MPC_MRT_Interface::evaluatePolicy + computeJointTorques for the solution of the last run: per instance, at `secondsAfterStart`. */
  void evaluatePolicy(const std::vector<double>& secondsAfterStart, std::vector<double>& state, std::vector<double>& input, std::vector<double>& jointTorques) {
    const size_t B = (size_t)solution_.batch;
    if (secondsAfterStart.size() != B) throw std::runtime_error("[HipSqpSolver] evaluatePolicy: one time per instance expected");
    state.assign(B * HSQP_NX, 0.0); input.assign(B * HSQP_NU, 0.0); jointTorques.assign(B * HSQP_NJ, 0.0);
    const int rc = hsqp_evaluate_policy(h_, secondsAfterStart.data(), state.data(), input.data(), jointTorques.data());
    if (rc != HSQP_OK) throw std::runtime_error("[HipSqpSolver] hsqp_evaluate_policy failed (" + std::to_string(rc) + "): " + hsqp_last_error(h_));
  }

  const PrimalSolution& getPrimalSolution() const { return solution_; }
  const std::vector<hsqp_perf>& getPerformanceIndeces() const { return perf_; }
  const std::vector<hsqp_perf>& getPerformanceIndecesBeforeStep() const { return perfBefore_; }
  /** KKT residuals {stationarity, primal} of the projected QP per instance — a DIAGNOSTIC that costs a kernel and, at 256 instances, 4.5 GB
   *  of reads per call: evaluated only after setReportKkt(true) (zeros otherwise); getBenchmarks() does not depend on it. */
  const std::vector<double>& getKktResiduals() const { return kkt_; }
  void setReportKkt(bool on) { reportKkt_ = on; }
  /** Step length and FilterLinesearch::StepType (HSQP_STEP_*) per instance of the last run. */
  const std::vector<double>& getStepSizes() const { return stepSize_; }
  const std::vector<int32_t>& getStepTypes() const { return stepType_; }
  void setLinesearchSettings(const hsqp_linesearch_settings& ls) {
    if (hsqp_set_linesearch(h_, &ls) != HSQP_OK) throw std::runtime_error(std::string("[HipSqpSolver] ") + hsqp_last_error(h_));
  }
  Benchmarks getBenchmarks() const { return bench_; }
  /** SQP iterations the last runWithReference ran, and what iteration `it` of them ended with (instance 0 .. batch-1). */
  int getNumIterations() const { return iterations_; }
  void getIterationLog(int it, std::vector<hsqp_perf>& perf, std::vector<double>& alpha, std::vector<int32_t>& stepType) const {
    perf.assign(solution_.batch, hsqp_perf{}); alpha.assign(solution_.batch, 0.0); stepType.assign(solution_.batch, HSQP_STEP_ZERO);
    if (hsqp_iteration_log(h_, it, perf.data(), alpha.data(), stepType.data()) != HSQP_OK) throw std::runtime_error("[HipSqpSolver] no such iteration in the log");
  }
  /** Live weight update (the centroidal node's gains receiver): diagonal Q / R / Qf, nullptr keeps the current one. */
  void updateWeights(const double* Q, const double* R, const double* Qf) {
    if (hsqp_update_weights(h_, Q, R, Qf) != HSQP_OK) throw std::runtime_error(std::string("[HipSqpSolver] ") + hsqp_last_error(h_));
  }
  hsqp_handle* handle() { return h_; }

 private:
  hsqp_handle* h_ = nullptr;
  PrimalSolution solution_;
  std::vector<hsqp_perf> perf_, perfBefore_;
  std::vector<double> kkt_, stepSize_;
  std::vector<int32_t> stepType_;
  Benchmarks bench_;
  int iterations_ = 0;
  bool reportKkt_ = false;
};

}  // namespace hsqp_host
