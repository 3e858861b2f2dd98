#pragma once
#include "ocs2_core/Types.h"
namespace ocs2 {
struct ModeSchedule {   // upstream ocs2_core/reference/ModeSchedule.h
  scalar_array_t eventTimes;
  size_array_t modeSequence;   // eventTimes.size() + 1 entries
};
struct TargetTrajectories {  // upstream ocs2_core/reference/TargetTrajectories.h
  scalar_array_t timeTrajectory;
  vector_array_t stateTrajectory;
  vector_array_t inputTrajectory;
};
}  // namespace ocs2
