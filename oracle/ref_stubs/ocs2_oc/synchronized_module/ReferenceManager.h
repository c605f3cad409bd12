// STAND-IN (test infrastructure) for <ocs2_oc/synchronized_module/ReferenceManager.h> (+ <ocs2_core/reference/TargetTrajectories.h>):
// upstream ReferenceManager keeps the ModeSchedule and the TargetTrajectories and, in preSolverRun, lets the derived class modify them
// (modifyReferences) before publishing them to the getters — the same call order here.  TargetTrajectories::getDesiredState is
// upstream's LinearInterpolation of the state trajectory (clamped at the ends), restated.
#pragma once
#include <ocs2_core/Types.h>
#include <ocs2_core/reference/ModeSchedule.h>
namespace ocs2 {
struct TargetTrajectories {
  TargetTrajectories() = default;
  TargetTrajectories(scalar_array_t t, vector_array_t x, vector_array_t u = {}) : timeTrajectory(std::move(t)), stateTrajectory(std::move(x)), inputTrajectory(std::move(u)) {}
  bool empty() const { return timeTrajectory.empty(); }
  vector_t getDesiredState(scalar_t time) const {
    if (timeTrajectory.empty()) throw std::runtime_error("[TargetTrajectories] empty");
    if (time <= timeTrajectory.front() || timeTrajectory.size() == 1) return stateTrajectory.front();
    if (time >= timeTrajectory.back()) return stateTrajectory.back();
    size_t i = 1;
    while (timeTrajectory[i] < time) ++i;
    const scalar_t a = (timeTrajectory[i] - time) / (timeTrajectory[i] - timeTrajectory[i - 1]);
    return vector_t(a * stateTrajectory[i - 1] + (1.0 - a) * stateTrajectory[i]);
  }
  vector_t getDesiredInput(scalar_t time) const {   // (upstream: the same interpolation of the input trajectory)
    if (timeTrajectory.empty() || inputTrajectory.empty()) throw std::runtime_error("[TargetTrajectories] empty");
    if (time <= timeTrajectory.front() || timeTrajectory.size() == 1) return inputTrajectory.front();
    if (time >= timeTrajectory.back()) return inputTrajectory.back();
    size_t i = 1;
    while (timeTrajectory[i] < time) ++i;
    const scalar_t a = (timeTrajectory[i] - time) / (timeTrajectory[i] - timeTrajectory[i - 1]);
    return vector_t(a * inputTrajectory[i - 1] + (1.0 - a) * inputTrajectory[i]);
  }
  scalar_array_t timeTrajectory;
  vector_array_t stateTrajectory;
  vector_array_t inputTrajectory;
};
class ReferenceManager {
 public:
  ReferenceManager(TargetTrajectories t, ModeSchedule m) : targetTrajectories_(std::move(t)), modeSchedule_(std::move(m)) {}
  virtual ~ReferenceManager() = default;
  void preSolverRun(scalar_t initTime, scalar_t finalTime, const vector_t& initState) {
    modifyReferences(initTime, finalTime, initState, modeSchedule_.modeAtTime(initTime), targetTrajectories_, modeSchedule_);
  }
  const ModeSchedule& getModeSchedule() const { return modeSchedule_; }
  void setModeSchedule(ModeSchedule m) { modeSchedule_ = std::move(m); }
  const TargetTrajectories& getTargetTrajectories() const { return targetTrajectories_; }
  void setTargetTrajectories(TargetTrajectories t) { targetTrajectories_ = std::move(t); }
 protected:
  virtual void modifyReferences(scalar_t, scalar_t, const vector_t&, size_t, TargetTrajectories&, ModeSchedule&) {}
 private:
  TargetTrajectories targetTrajectories_;
  ModeSchedule modeSchedule_;
};
}  // namespace ocs2
