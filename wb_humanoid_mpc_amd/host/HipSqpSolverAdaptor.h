// The reference-side binding of the drop-in boundary: an ocs2::SolverBase / ocs2::MPC_BASE pair over the HIP library, to be
// instantiated where the reference instantiates ocs2::SqpMpc —
//   humanoid_nmpc/humanoid_wb_mpc_ros2/src/WBMpcSqpNode.cpp:64        SqpMpc mpc(mpcSettings, sqpSettings, ocp, initializer);
//   humanoid_nmpc/humanoid_wb_mpc_ros2/src/WBMpcRobotSim.cpp:65, humanoid_centroidal_mpc_ros2/src/CentroidalMpcSqpNode.cpp:63
//   then mpc.getSolverPtr()->setReferenceManager(...) / addSynchronizedModule(...)  (WBMpcSqpNode.cpp:85-86)
// becomes   HipSqpMpc mpc(mpcSettings, sqpSettings, adaptorConfig, initializer);   everything after it is unchanged.
//
// What one SolverBase::run(t0, x0, tf) does here, mirroring upstream ocs2_sqp SqpSolver::runImpl (source absent from /root/reference;
// restated from the published ocs2):
//   1. time grid: nodes every sqp.dt from t0, split at the mode schedule's event times into pre- / post-event nodes, last interval
//      shortened to tf (timeDiscretizationWithEvents);
//   2. warm start: the previous primal solution interpolated onto the new grid, the uncovered tail (or everything on the first
//      call / after reset) from the Initializer — the reference's WeightCompInitializer
//      (humanoid_common_mpc/src/initialization/WeightCompInitializer.cpp:66-70);
//   3. the reference manager's ModeSchedule and TargetTrajectories go to the device as they are (hsqp_upload_reference builds the
//      per-node parameter table there: contact flags, swing-foot references, impact proximity, arm-swing phase, target interpolation);
//   4. sqpIteration x { LQ approximation, projection, Riccati QP, filter line search } on the GPU (hsqp_iterate_device), stopping
//      early when the step falls below deltaTol;
//   5. PrimalSolution (time stamps, states, inputs, mode schedule, FeedforwardController — useFeedbackPolicy false, task.info:91),
//      PerformanceIndex log, SqpSolver::getBenchmarks() buckets.
// Compiled and run in this repository against stand-in ocs2 headers (tests/stubs/ocs2, tests/test_adaptor.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>

#include <ocs2_core/initialization/Initializer.h>
#include <ocs2_mpc/MPC_BASE.h>
#include <ocs2_oc/oc_solver/SolverBase.h>
#include <ocs2_sqp/SqpSettings.h>

#include "HipSqpSolver.h"

namespace ocs2 {
namespace humanoid {

struct HipSqpAdaptorConfig {
  hsqp_model_desc model;          // filled from the PinocchioInterface and task.info
  hsqp_swing_config swing;        // task.info swing_trajectory_config
  int stateDim = HSQP_NX;         // 58 whole-body, 35 centroidal (rows are padded to 58 on the way in)
  int maxNodes = 256;
  double terrainHeight = 0.0;
  bool armSwingReference = true;
  bool eventNodes = true;         // false: uniform grid, mode switches snap to the nodes (the bench's configuration)
  int device = 0;
};

struct HipSqpBenchmarks {         // SqpSolver::getBenchmarks() of the fork (SqpBenchmarksPublisher.cpp:44-57), accumulated seconds
  scalar_t linearQuadraticApproximationTime = 0.0, solveQpTime = 0.0, linesearchTime = 0.0, computeControllerTime = 0.0;
  size_t numCalls = 0;
};

/** upstream timeDiscretizationWithEvents: (node times, post-event flags). */
inline void timeDiscretizationWithEvents(scalar_t initTime, scalar_t finalTime, scalar_t dt, const scalar_array_t& eventTimes, scalar_array_t& times,
                                         std::vector<char>& postEvent, scalar_t dtMin = 1e-4) {
  times.assign(1, initTime);
  postEvent.assign(1, 0);
  size_t ie = 0;
  while (ie < eventTimes.size() && eventTimes[ie] <= initTime) ++ie;
  while (times.back() < finalTime) {
    scalar_t next = times.back() + dt;
    bool isEvent = false;
    if (ie < eventTimes.size() && eventTimes[ie] < finalTime && next >= eventTimes[ie]) { next = eventTimes[ie]; isEvent = true; ++ie; }
    if (next >= finalTime) { next = finalTime; isEvent = false; }
    if (next > times.back() + dtMin || postEvent.back()) { times.push_back(next); postEvent.push_back(0); }
    else times.back() = next;
    if (isEvent) { times.push_back(next); postEvent.push_back(1); }
  }
}

class HipSqpSolverAdaptor final : public SolverBase {
 public:
  static constexpr scalar_t kEventEps = 1e-9;   // a post-event node is sampled at t_event + eps (ocs2 getIntervalStart)

  HipSqpSolverAdaptor(const HipSqpAdaptorConfig& config, sqp::Settings settings, const Initializer& initializer)
      : cfg_(config), settings_(std::move(settings)), initializer_(initializer.clone()),
        impl_(config.model, config.maxNodes, /*maxBatch*/ 1, config.device, /*useLinesearch*/ true, /*recedingHorizon*/ true) {
    hsqp_linesearch_settings ls;
    hsqp_linesearch_defaults(&ls);
    ls.g_max = settings_.g_max; ls.g_min = settings_.g_min; ls.gamma_c = settings_.gamma_c; ls.armijo_factor = settings_.armijoFactor;
    ls.alpha_decay = settings_.alpha_decay; ls.alpha_min = settings_.alpha_min; ls.delta_tol = settings_.deltaTol; ls.cost_tol = settings_.costTol;
    impl_.setLinesearchSettings(ls);
  }

  void reset() override { primal_.clear(); log_.clear(); numIterations_ = 0; benchmarks_ = HipSqpBenchmarks(); }
  size_t getNumIterations() const override { return numIterations_; }
  scalar_t getFinalTime() const override { return primal_.timeTrajectory_.empty() ? 0.0 : primal_.timeTrajectory_.back(); }
  const PerformanceIndex& getPerformanceIndeces() const override {
    if (log_.empty()) throw std::runtime_error("[HipSqpSolverAdaptor] No performance log yet, no problem solved yet?");
    return log_.back();
  }
  const std::vector<PerformanceIndex>& getIterationsLog() const override {
    if (log_.empty()) throw std::runtime_error("[HipSqpSolverAdaptor] No performance log yet, no problem solved yet?");
    return log_;
  }
  void getPrimalSolution(scalar_t finalTime, PrimalSolution* out) const override {
    // the solution is returned up to finalTime (upstream truncates at the first stamp >= finalTime)
    out->clear();
    size_t n = primal_.timeTrajectory_.size();
    while (n > 1 && primal_.timeTrajectory_[n - 2] >= finalTime) --n;
    out->timeTrajectory_.assign(primal_.timeTrajectory_.begin(), primal_.timeTrajectory_.begin() + n);
    out->stateTrajectory_.assign(primal_.stateTrajectory_.begin(), primal_.stateTrajectory_.begin() + n);
    out->inputTrajectory_.assign(primal_.inputTrajectory_.begin(), primal_.inputTrajectory_.begin() + n);
    for (size_t i : primal_.postEventIndices_) if (i < n) out->postEventIndices_.push_back(i);
    out->modeSchedule_ = primal_.modeSchedule_;
    out->controllerPtr_.reset(new FeedforwardController(out->timeTrajectory_, out->inputTrajectory_));
  }
  // ---- the rest of SolverBase's pure-virtual query interface.  Upstream SqpSolver answers these the same way: it throws "not
  //      implemented" for the value function, the Hamiltonian, the Lagrangian and the multipliers (ocs2_sqp/SqpSolver.h); the
  //      optimal control problem lives on the device here (hsqp_model_desc), so there is no host OptimalControlProblem to hand out.
  const OptimalControlProblem& getOptimalControlProblem() const override {
    throw std::runtime_error("[HipSqpSolverAdaptor] getOptimalControlProblem() not available: the problem definition is device code (hsqp_model_desc)");
  }
  const DualSolution* getDualSolution() const override { return nullptr; }   // upstream SqpSolver: nullptr as well (no dual solution)
  ScalarFunctionQuadraticApproximation getValueFunction(scalar_t, const vector_t&) const override {
    throw std::runtime_error("[HipSqpSolverAdaptor] getValueFunction() not implemented");
  }
  ScalarFunctionQuadraticApproximation getHamiltonian(scalar_t, const vector_t&, const vector_t&) override {
    throw std::runtime_error("[HipSqpSolverAdaptor] getHamiltonian() not implemented");
  }
  vector_t getStateInputEqualityConstraintLagrangian(scalar_t, const vector_t&) const override {
    throw std::runtime_error("[HipSqpSolverAdaptor] getStateInputEqualityConstraintLagrangian() not implemented");
  }
  MultiplierCollection getSolutionMultipliers(scalar_t) const override {
    throw std::runtime_error("[HipSqpSolverAdaptor] getSolutionMultipliers() not implemented");
  }
  /** SqpSolver::getBenchmarks() of the fork; `Benchmarks`: the nested type name code written against ocs2::SqpSolver uses (SqpBenchmarksPublisher.cpp:44) —
   *  with `namespace ocs2 { using SqpSolver = humanoid::HipSqpSolverAdaptor; }` that file compiles unchanged (tests/stubs/ros2, tests/test_adaptor.py). */
  using Benchmarks = HipSqpBenchmarks;
  const HipSqpBenchmarks& getBenchmarks() const { return benchmarks_; }
  std::string getBenchmarkingInfo() const override {
    const scalar_t n = std::max<size_t>(benchmarks_.numCalls, 1);
    return "\n########################################################################\nThe benchmarking is computed over " + std::to_string(benchmarks_.numCalls) +
           " iterations.\nSQP Benchmarking\t   :\tAverage time [ms]\n\tLQ Approximation   :\t" + std::to_string(1e3 * benchmarks_.linearQuadraticApproximationTime / n) +
           "\n\tSolve QP           :\t" + std::to_string(1e3 * benchmarks_.solveQpTime / n) + "\n\tLinesearch         :\t" + std::to_string(1e3 * benchmarks_.linesearchTime / n) + "\n";
  }
  /** Step length / FilterLinesearch step type (HSQP_STEP_*) of the last iteration. */
  scalar_t lastStepSize() const { return stepSize_; }
  int lastStepType() const { return stepType_; }
  /** the device-side policy sample + feed-forward torques of the last solution (WBMpcMrtJointController.cpp:136-158) */
  void evaluatePolicy(scalar_t time, vector_t& state, vector_t& input, vector_t& jointTorques) {
    std::vector<double> x, u, tau;
    impl_.evaluatePolicy({time - primal_.timeTrajectory_.front()}, x, u, tau);
    state = vector_t(cfg_.stateDim); input = vector_t(HSQP_NU); jointTorques = vector_t(HSQP_NJ);
    std::copy_n(x.data(), cfg_.stateDim, state.data()); std::copy_n(u.data(), HSQP_NU, input.data()); std::copy_n(tau.data(), HSQP_NJ, jointTorques.data());
  }

 private:
  // clamped linear interpolation of the previous solution (ocs2 LinearInterpolation on the PrimalSolution stamps)
  static void interpolate(const scalar_array_t& t, const vector_array_t& v, scalar_t time, double* out, int n) {
    if (time <= t.front()) { std::copy_n(v.front().data(), n, out); return; }
    if (time >= t.back()) { std::copy_n(v.back().data(), n, out); return; }
    size_t i = std::upper_bound(t.begin(), t.end(), time) - t.begin();   // t[i-1] <= time < t[i]
    const scalar_t h = t[i] - t[i - 1], a = h > 0.0 ? (time - t[i - 1]) / h : 1.0;
    for (int k = 0; k < n; ++k) out[k] = (1.0 - a) * v[i - 1][k] + a * v[i][k];
  }

  // upstream SqpSolver: an external controller is ignored ("runImpl(initTime, initState, finalTime)"); an external primal solution
  // replaces the warm start
  void runImpl(scalar_t initTime, const vector_t& initState, scalar_t finalTime, const ControllerBase* /*externalControllerPtr*/) override {
    runImpl(initTime, initState, finalTime);
  }
  void runImpl(scalar_t initTime, const vector_t& initState, scalar_t finalTime, const PrimalSolution& primalSolution) override {
    primal_.clear();
    primal_.timeTrajectory_ = primalSolution.timeTrajectory_;
    primal_.stateTrajectory_ = primalSolution.stateTrajectory_;
    primal_.inputTrajectory_ = primalSolution.inputTrajectory_;
    primal_.postEventIndices_ = primalSolution.postEventIndices_;
    primal_.modeSchedule_ = primalSolution.modeSchedule_;
    runImpl(initTime, initState, finalTime);
  }
  void runImpl(scalar_t initTime, const vector_t& initState, scalar_t finalTime) override {
    const int nx = cfg_.stateDim;
    if ((int)initState.size() != nx) throw std::runtime_error("[HipSqpSolverAdaptor] initial state has the wrong dimension");
    const ReferenceManagerInterface& ref = getReferenceManager();
    const ModeSchedule& ms = ref.getModeSchedule();
    const TargetTrajectories& tt = ref.getTargetTrajectories();
    if (ms.eventTimes.empty() || tt.timeTrajectory.empty()) throw std::runtime_error("[HipSqpSolverAdaptor] empty mode schedule / target trajectories");
    // 1. time grid
    scalar_array_t times;
    std::vector<char> post;
    if (cfg_.eventNodes) timeDiscretizationWithEvents(initTime, finalTime, settings_.dt, ms.eventTimes, times, post);
    else {
      const int n = std::max(1, (int)std::lround((finalTime - initTime) / settings_.dt));
      for (int k = 0; k <= n; ++k) times.push_back(initTime + k * settings_.dt);
      post.assign(times.size(), 0);
    }
    const int N = (int)times.size() - 1;
    if (N > cfg_.maxNodes) throw std::runtime_error("[HipSqpSolverAdaptor] horizon needs " + std::to_string(N) + " intervals, capacity " + std::to_string(cfg_.maxNodes));
    std::vector<double> dts(N), nodeTimes(N + 1);
    for (int k = 0; k < N; ++k) dts[k] = times[k + 1] - times[k];
    for (int k = 0; k <= N; ++k) nodeTimes[k] = times[k] + (post[k] ? kEventEps : 0.0);
    // 2. warm start
    std::vector<double> x((size_t)(N + 1) * HSQP_NX, 0.0), u((size_t)N * HSQP_NU, 0.0), x0(HSQP_NX, 0.0);
    std::copy_n(initState.data(), nx, x0.data());
    const bool have = !primal_.timeTrajectory_.empty();
    const scalar_t covered = have ? primal_.timeTrajectory_.back() : initTime;
    vector_t xk(nx), uk(HSQP_NU), xn(nx);
    for (int k = 0; k <= N; ++k) {
      double* xr = &x[(size_t)k * HSQP_NX];
      if (have && times[k] <= covered) {
        interpolate(primal_.timeTrajectory_, primal_.stateTrajectory_, times[k], xr, nx);
        if (k < N) interpolate(primal_.timeTrajectory_, primal_.inputTrajectory_, times[k], &u[(size_t)k * HSQP_NU], HSQP_NU);
      } else {
        // Initializer::compute(t_k, x_k, t_{k+1}, u_k, x_{k+1}): the state is carried, the input is the weight compensation of the node's mode
        if (k == 0) std::copy_n(initState.data(), nx, xr);
        else std::copy_n(&x[(size_t)(k - 1) * HSQP_NX], nx, xr);          // x_{k} = previous nextState (WeightCompInitializer keeps the state)
        if (k < N) {
          std::copy_n(xr, nx, xk.data());
          initializer_->compute(nodeTimes[k], xk, nodeTimes[k + 1], uk, xn);
          std::copy_n(uk.data(), HSQP_NU, &u[(size_t)k * HSQP_NU]);
        }
      }
    }
    // 3. the compact reference: mode schedule + target knots (58-double rows)
    const int32_t nEvents = (int32_t)ms.eventTimes.size();
    std::vector<int32_t> modes(ms.modeSequence.begin(), ms.modeSequence.end());
    std::vector<double> knots(tt.timeTrajectory.size() * HSQP_NX, 0.0);
    for (size_t i = 0; i < tt.timeTrajectory.size(); ++i) std::copy_n(tt.stateTrajectory[i].data(), nx, &knots[i * HSQP_NX]);
    hsqp_reference r;
    std::memset(&r, 0, sizeof(r));
    r.batch = 1; r.n_nodes = N; r.t0 = initTime; r.dt = settings_.dt; r.max_events = nEvents; r.n_events = &nEvents;
    r.event_times = ms.eventTimes.data(); r.mode_sequence = modes.data(); r.n_knots = (int32_t)tt.timeTrajectory.size();
    r.target_times = tt.timeTrajectory.data(); r.target_states = knots.data(); r.swing = cfg_.swing; r.terrain_height = cfg_.terrainHeight;
    r.arm_swing = cfg_.armSwingReference ? 1 : 0; r.node_times = cfg_.eventNodes ? nodeTimes.data() : nullptr;
    // 4. SQP iterations with the filter line search: ONE upload and ONE device call for all sqpIteration of them (the trajectories stay in
    //    HBM between the iterations; the loop ends early by SqpSolver::checkConvergence's step-size test, evaluated per iteration from a
    //    small read-back of the line-search state)
    log_.clear();
    impl_.runWithReference(N, settings_.dt, x0.data(), x.data(), u.data(), r, /*line search*/ true, cfg_.eventNodes ? dts.data() : nullptr,
                           (int)std::max<size_t>(settings_.sqpIteration, 1));
    {
      const hsqp_host::PrimalSolution& s = impl_.getPrimalSolution();
      x = s.stateTrajectory; u = s.inputTrajectory;
      numIterations_ = (size_t)impl_.getNumIterations();
      std::vector<hsqp_perf> perf; std::vector<double> alpha; std::vector<int32_t> type;
      for (size_t it = 0; it < numIterations_; ++it) {
        impl_.getIterationLog((int)it, perf, alpha, type);
        PerformanceIndex pi;
        pi.merit = perf[0].merit; pi.cost = perf[0].cost; pi.dynamicsViolationSSE = perf[0].dynamics_sse; pi.equalityConstraintsSSE = perf[0].equality_sse;
        log_.push_back(pi);
        stepSize_ = alpha[0]; stepType_ = type[0];
      }
      const hsqp_host::Benchmarks b = impl_.getBenchmarks();   // summed over the iterations of the call
      benchmarks_.linearQuadraticApproximationTime += b.linearQuadraticApproximationTime; benchmarks_.solveQpTime += b.solveQpTime;
      benchmarks_.linesearchTime += b.linesearchTime; benchmarks_.computeControllerTime += b.computeControllerTime; benchmarks_.numCalls += numIterations_;
    }
    // 5. primal solution: inputs stamped at every node, the last one repeated (upstream PrimalSolution convention).  A pre-event node
    //    has no input of its own (its stage is the identity jump, du = 0): upstream multiple_shooting::toPrimalSolution copies the
    //    input of the node before it, so that the controller, the next warm start and the policy hold the last optimised input up
    //    to the switch instead of blending towards a stale one.
    for (int k = 1; k < N; ++k)
      if (dts[k] == 0.0) std::copy_n(&u[(size_t)(k - 1) * HSQP_NU], HSQP_NU, &u[(size_t)k * HSQP_NU]);
    primal_.clear();
    primal_.timeTrajectory_ = times;
    primal_.modeSchedule_ = ms;
    for (int k = 0; k <= N; ++k) {
      vector_t xs(nx), us(HSQP_NU);
      std::copy_n(&x[(size_t)k * HSQP_NX], nx, xs.data());
      std::copy_n(&u[(size_t)std::min(k, N - 1) * HSQP_NU], HSQP_NU, us.data());
      primal_.stateTrajectory_.push_back(std::move(xs)); primal_.inputTrajectory_.push_back(std::move(us));
      if (post[k]) primal_.postEventIndices_.push_back((size_t)k);
    }
  }

  HipSqpAdaptorConfig cfg_;
  sqp::Settings settings_;
  std::unique_ptr<Initializer> initializer_;
  hsqp_host::HipSqpSolver impl_;
  PrimalSolution primal_;
  std::vector<PerformanceIndex> log_;
  size_t numIterations_ = 0;
  HipSqpBenchmarks benchmarks_;
  scalar_t stepSize_ = 0.0;
  int stepType_ = HSQP_STEP_ZERO;
};

/** Mirrors ocs2::SqpMpc (ocs2_sqp/SqpMpc.h): an MPC_BASE that owns the solver. */
class HipSqpMpc final : public MPC_BASE {
 public:
  HipSqpMpc(mpc::Settings mpcSettings, sqp::Settings sqpSettings, const HipSqpAdaptorConfig& config, const Initializer& initializer)
      : MPC_BASE(std::move(mpcSettings)), solverPtr_(new HipSqpSolverAdaptor(config, std::move(sqpSettings), initializer)) {}
  HipSqpSolverAdaptor* getSolverPtr() override { return solverPtr_.get(); }
  const HipSqpSolverAdaptor* getSolverPtr() const override { return solverPtr_.get(); }

 protected:
  void calculateController(scalar_t initTime, const vector_t& initState, scalar_t finalTime) override {
    if (settings().coldStart_) solverPtr_->reset();
    solverPtr_->run(initTime, initState, finalTime);
  }

 private:
  std::unique_ptr<HipSqpSolverAdaptor> solverPtr_;
};

}  // namespace humanoid
}  // namespace ocs2
