// STAND-IN for <ocs2_mpc/MPC_BASE.h> + <ocs2_mpc/MPC_Settings.h>: run(t, x) solves over [t, t + timeHorizon] through calculateController.
#pragma once
#include <ocs2_oc/oc_solver/SolverBase.h>
namespace ocs2 {
namespace mpc {
struct Settings {
  scalar_t timeHorizon_ = 1.0, solutionTimeWindow_ = -1.0, mpcDesiredFrequency_ = -1.0, mrtDesiredFrequency_ = 100.0;
  bool coldStart_ = false, debugPrint_ = false;
};
}  // namespace mpc
class MPC_BASE {
 public:
  explicit MPC_BASE(mpc::Settings mpcSettings) : mpcSettings_(std::move(mpcSettings)) {}
  virtual ~MPC_BASE() = default;
  virtual void reset() { initRun_ = true; getSolverPtr()->reset(); }
  virtual bool run(scalar_t currentTime, const vector_t& currentState) {
    const scalar_t finalTime = currentTime + mpcSettings_.timeHorizon_;
    if (mpcSettings_.coldStart_) getSolverPtr()->reset();
    calculateController(currentTime, currentState, finalTime);
    initRun_ = false;
    return true;
  }
  virtual SolverBase* getSolverPtr() = 0;
  virtual const SolverBase* getSolverPtr() const = 0;
  const mpc::Settings& settings() const { return mpcSettings_; }
 protected:
  virtual void calculateController(scalar_t initTime, const vector_t& initState, scalar_t finalTime) = 0;
  bool initRun_ = true;
 private:
  mpc::Settings mpcSettings_;
};
}  // namespace ocs2
