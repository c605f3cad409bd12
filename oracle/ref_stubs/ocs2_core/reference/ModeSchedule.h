// STAND-IN (test infrastructure) for <ocs2_core/reference/ModeSchedule.h>: upstream's struct is {eventTimes, modeSequence}
// with modeSequence.size() == eventTimes.size() + 1 and modeAtTime(t) = modeSequence[lookup::findIndexInTimeArray(eventTimes, t)].
#pragma once
#include <ocs2_core/Types.h>
#include <ocs2_core/misc/Lookup.h>
namespace ocs2 {
struct ModeSchedule {
  ModeSchedule() : ModeSchedule(std::vector<scalar_t>{}, std::vector<size_t>{0}) {}
  ModeSchedule(std::vector<scalar_t> eventTimesInput, std::vector<size_t> modeSequenceInput)
      : eventTimes(std::move(eventTimesInput)), modeSequence(std::move(modeSequenceInput)) {}
  size_t modeAtTime(scalar_t time) const { return modeSequence[lookup::findIndexInTimeArray(eventTimes, time)]; }
  std::vector<scalar_t> eventTimes;
  std::vector<size_t> modeSequence;
};
inline std::ostream& operator<<(std::ostream& stream, const ModeSchedule&) { return stream; }   // upstream prints the schedule; unused here
}  // namespace ocs2
