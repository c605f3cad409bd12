// STAND-IN for <ocs2_core/reference/ModeSchedule.h>
#pragma once
#include <algorithm>
#include <ocs2_core/Types.h>
namespace ocs2 {
struct ModeSchedule {
  ModeSchedule() : modeSequence{0} {}
  ModeSchedule(scalar_array_t eventTimesInput, size_array_t modeSequenceInput) : eventTimes(std::move(eventTimesInput)), modeSequence(std::move(modeSequenceInput)) {}
  size_t modeAtTime(scalar_t t) const { return modeSequence[std::lower_bound(eventTimes.begin(), eventTimes.end(), t) - eventTimes.begin()]; }
  scalar_array_t eventTimes;
  size_array_t modeSequence;
};
}  // namespace ocs2
