#pragma once
#include "ocs2_core/Types.h"
namespace ocs2 {
struct ModeSchedule {   // upstream ocs2_core/reference/ModeSchedule.h
  scalar_array_t eventTimes;
  size_array_t modeSequence;   // eventTimes.size() + 1 entries
  size_t modeAtTime(scalar_t time) const {   // upstream: lookup::findIndexInTimeArray (lower_bound) into modeSequence
    size_t ind = 0;
    while (ind < eventTimes.size() && eventTimes[ind] < time) ++ind;
    return modeSequence[ind];
  }
};
struct TargetTrajectories {  // upstream ocs2_core/reference/TargetTrajectories.h
  scalar_array_t timeTrajectory;
  vector_array_t stateTrajectory;
  vector_array_t inputTrajectory;
};
}  // namespace ocs2
