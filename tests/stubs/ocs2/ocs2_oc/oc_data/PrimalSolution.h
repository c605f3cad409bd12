// STAND-IN for <ocs2_oc/oc_data/PrimalSolution.h>
#pragma once
#include <ocs2_core/control/FeedforwardController.h>
#include <ocs2_core/reference/ModeSchedule.h>
namespace ocs2 {
struct PrimalSolution {
  void clear() { timeTrajectory_.clear(); stateTrajectory_.clear(); inputTrajectory_.clear(); postEventIndices_.clear(); controllerPtr_.reset(); }
  scalar_array_t timeTrajectory_;
  size_array_t postEventIndices_;
  vector_array_t stateTrajectory_;
  vector_array_t inputTrajectory_;
  ModeSchedule modeSchedule_;
  std::unique_ptr<ControllerBase> controllerPtr_;
};
}  // namespace ocs2
