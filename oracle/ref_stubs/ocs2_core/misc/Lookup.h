// STAND-IN (test infrastructure) for <ocs2_core/misc/Lookup.h>: upstream findIndexInTimeArray is std::lower_bound on the
// time array ("partition index i such that timeArray[i-1] < t <= timeArray[i]").
#pragma once
#include <ocs2_core/Types.h>
namespace ocs2 { namespace lookup {
inline int findIndexInTimeArray(const std::vector<scalar_t>& timeArray, scalar_t time) {
  return static_cast<int>(std::lower_bound(timeArray.begin(), timeArray.end(), time) - timeArray.begin());
}
}}  // namespace ocs2::lookup
