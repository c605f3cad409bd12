// STAND-IN for <ocs2_oc/oc_data/PerformanceIndex.h> (the fields the adaptor fills)
#pragma once
#include <ocs2_core/Types.h>
namespace ocs2 {
struct PerformanceIndex {
  scalar_t merit = 0.0, cost = 0.0, dualFeasibilitiesSSE = 0.0, dynamicsViolationSSE = 0.0, equalityConstraintsSSE = 0.0, inequalityConstraintsSSE = 0.0,
           equalityLagrangian = 0.0, inequalityLagrangian = 0.0;
};
}  // namespace ocs2
