// TEST INFRASTRUCTURE: drives wb_humanoid_mpc_amd/host/HipSqpSolverAdaptor.h the way the reference's SQP node drives ocs2::SqpMpc
// (humanoid_nmpc/humanoid_wb_mpc_ros2/src/WBMpcSqpNode.cpp:61-89): construct the MPC from settings + initializer, hand the solver a
// reference manager, then run(t, x) once per MPC period from the measured state.  Compiled against the stand-in ocs2 headers of
// tests/stubs/ocs2 and linked with libhsqp_hip.so (tests/test_adaptor.py).
//   adaptor_driver <model.bin> <case.txt> <out.txt>
#include <cstdio>
#include <fstream>
#include <iostream>

#include "HipSqpSolverAdaptor.h"
#include "HipSqpModelIO.h"

using namespace ocs2;
using namespace ocs2::humanoid;

// WeightCompInitializer (humanoid_common_mpc/src/initialization/WeightCompInitializer.cpp:66-70): input = weight compensation over the
// stance feet of the mode at `time`, next state = state.
class WeightCompInitializer final : public Initializer {
 public:
  WeightCompInitializer(const ModeSchedule* ms, double mass) : ms_(ms), mass_(mass) {}
  WeightCompInitializer* clone() const override { return new WeightCompInitializer(*this); }
  void compute(scalar_t time, const vector_t& state, scalar_t, vector_t& input, vector_t& nextState) override {
    const size_t mode = ms_->modeAtTime(time);                 // FLY 0, RF 1, LF 2, STANCE 3: {left, right} = {mode & 2, mode & 1}
    const bool left = mode & 2, right = mode & 1;
    const int ns = int(left) + int(right);
    input = vector_t::Zero(HSQP_NU);
    if (ns) { const double fz = mass_ * 9.81 / ns; if (left) input[2] = fz; if (right) input[8] = fz; }
    nextState = state;
  }
 private:
  const ModeSchedule* ms_;
  double mass_;
};

class FixedReferenceManager final : public ReferenceManagerInterface {
 public:
  void preSolverRun(scalar_t, scalar_t, const vector_t&) override { ++calls; }
  const ModeSchedule& getModeSchedule() const override { return ms; }
  const TargetTrajectories& getTargetTrajectories() const override { return tt; }
  ModeSchedule ms;
  TargetTrajectories tt;
  int calls = 0;
};

int main(int argc, char** argv) {
  if (argc != 4) { std::fprintf(stderr, "usage: adaptor_driver model.bin case.txt out.txt\n"); return 2; }
  HipSqpAdaptorConfig cfg;
  // the problem image: the exported JSON read by the C++ loader (host/HipSqpModelIO.h — no Python in the loop), or a raw struct dump
  const std::string modelPath = argv[1];
  const bool fromJson = modelPath.size() > 5 && modelPath.substr(modelPath.size() - 5) == ".json";
  if (fromJson) {
    try { cfg.model = hsqp_host::loadModelDesc(modelPath); cfg.swing = hsqp_host::loadSwingConfig(modelPath); }
    catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 2; }
  } else {
    std::ifstream f(argv[1], std::ios::binary);
    f.read(reinterpret_cast<char*>(&cfg.model), sizeof(cfg.model));
    if (!f) { std::fprintf(stderr, "cannot read the model description\n"); return 2; }
  }
  std::ifstream in(argv[2]);
  int stateDim, nEvents, nKnots, calls, eventNodes, maxNodes;
  double dt, horizon, period, t0;
  in >> stateDim >> dt >> horizon >> period >> t0 >> calls >> eventNodes >> maxNodes;
  {
    hsqp_swing_config fileSwing;
    double* sw = &fileSwing.lift_off_velocity;
    for (int i = 0; i < 8; ++i) in >> sw[i];
    if (fromJson) { if (std::memcmp(&fileSwing, &cfg.swing, sizeof(fileSwing)) != 0) { std::fprintf(stderr, "swing configuration of the image differs from the case file\n"); return 2; } }
    else cfg.swing = fileSwing;
  }
  auto rm = std::make_shared<FixedReferenceManager>();
  in >> nEvents;
  rm->ms.eventTimes.resize(nEvents); rm->ms.modeSequence.resize(nEvents + 1);
  for (auto& e : rm->ms.eventTimes) in >> e;
  for (auto& m : rm->ms.modeSequence) in >> m;
  in >> nKnots;
  rm->tt.timeTrajectory.resize(nKnots);
  for (auto& t : rm->tt.timeTrajectory) in >> t;
  for (int k = 0; k < nKnots; ++k) { vector_t s(stateDim); for (int i = 0; i < stateDim; ++i) in >> s[i]; rm->tt.stateTrajectory.push_back(s); rm->tt.inputTrajectory.push_back(vector_t::Zero(HSQP_NU)); }
  vector_t x(stateDim);
  for (int i = 0; i < stateDim; ++i) in >> x[i];
  if (!in) { std::fprintf(stderr, "malformed case file\n"); return 2; }
  cfg.stateDim = stateDim; cfg.maxNodes = maxNodes; cfg.eventNodes = eventNodes != 0;
  double mass = 0.0;
  for (const hsqp_body& b : cfg.model.bodies) mass += b.mass;

  mpc::Settings mpcSettings;
  mpcSettings.timeHorizon_ = horizon;
  sqp::Settings sqpSettings;                                   // g1_wb_mpc/config/mpc/task.info:79-93
  sqpSettings.dt = dt; sqpSettings.sqpIteration = 1; sqpSettings.deltaTol = 1e-4; sqpSettings.g_max = 1e-2; sqpSettings.g_min = 1e-6; sqpSettings.useFeedbackPolicy = false;
  WeightCompInitializer initializer(&rm->ms, mass);
  try {
    HipSqpMpc mpc(mpcSettings, sqpSettings, cfg, initializer);            // WBMpcSqpNode.cpp:64
    mpc.getSolverPtr()->setReferenceManager(rm);                           // WBMpcSqpNode.cpp:85
    std::FILE* out = std::fopen(argv[3], "w");
    double t = t0;
    for (int c = 0; c < calls; ++c) {
      mpc.run(t, x);
      const PrimalSolution sol = mpc.getSolverPtr()->primalSolution(t + horizon);
      const PerformanceIndex& p = mpc.getSolverPtr()->getPerformanceIndeces();
      const int n = (int)sol.timeTrajectory_.size();
      std::fprintf(out, "%d %.17g %.17g %d %.17g %.17g %.17g %zu\n", n, t, mpc.getSolverPtr()->lastStepSize(), mpc.getSolverPtr()->lastStepType(), p.cost,
                   p.dynamicsViolationSSE, p.equalityConstraintsSSE, sol.postEventIndices_.size());
      for (int k = 0; k < n; ++k) {
        std::fprintf(out, "%.17g", sol.timeTrajectory_[k]);
        for (int i = 0; i < stateDim; ++i) std::fprintf(out, " %.17g", sol.stateTrajectory_[k][i]);
        for (int i = 0; i < HSQP_NU; ++i) std::fprintf(out, " %.17g", sol.inputTrajectory_[k][i]);
        std::fprintf(out, "\n");
      }
      // the plant follows the plan for one MPC period: next measured state = the policy's state at t + period
      vector_t xs, us, tau;
      mpc.getSolverPtr()->evaluatePolicy(t + period, xs, us, tau);
      x = xs;
      for (int j = 0; j < HSQP_NJ; ++j) std::fprintf(out, j ? " %.17g" : "%.17g", tau[j]);
      std::fprintf(out, "\n");
      t += period;
    }
    std::fclose(out);
    const HipSqpBenchmarks& b = mpc.getSolverPtr()->getBenchmarks();
    std::printf("ok calls=%d preSolverRun=%d lq_ms=%.3f qp_ms=%.3f ls_ms=%.3f\n", calls, rm->calls, 1e3 * b.linearQuadraticApproximationTime / b.numCalls,
                1e3 * b.solveQpTime / b.numCalls, 1e3 * b.linesearchTime / b.numCalls);
    std::printf("%s", mpc.getSolverPtr()->getBenchmarkingInfo().c_str());
    return 0;
  } catch (const std::runtime_error& e) {
    std::printf("runtime_error: %s\n", e.what());
    return 3;
  }
}
