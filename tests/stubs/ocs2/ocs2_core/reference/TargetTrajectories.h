// STAND-IN for <ocs2_core/reference/TargetTrajectories.h>
#pragma once
#include <ocs2_core/Types.h>
namespace ocs2 {
struct TargetTrajectories {
  scalar_array_t timeTrajectory;
  vector_array_t stateTrajectory;
  vector_array_t inputTrajectory;
};
}  // namespace ocs2
