#pragma once
#include "ocs2_core/control/FeedforwardController.h"
#include "ocs2_core/reference/ModeSchedule.h"
namespace ocs2 {
struct PrimalSolution {   // upstream ocs2_oc/oc_data/PrimalSolution.h
  PrimalSolution() = default;
  PrimalSolution(const PrimalSolution& o)
      : timeTrajectory_(o.timeTrajectory_), stateTrajectory_(o.stateTrajectory_), inputTrajectory_(o.inputTrajectory_), postEventIndices_(o.postEventIndices_),
        modeSchedule_(o.modeSchedule_), controllerPtr_(o.controllerPtr_ ? o.controllerPtr_->clone() : nullptr) {}
  PrimalSolution& operator=(const PrimalSolution& o) { PrimalSolution t(o); swap(t); return *this; }
  void swap(PrimalSolution& o) {
    timeTrajectory_.swap(o.timeTrajectory_); stateTrajectory_.swap(o.stateTrajectory_); inputTrajectory_.swap(o.inputTrajectory_);
    postEventIndices_.swap(o.postEventIndices_); std::swap(modeSchedule_, o.modeSchedule_); controllerPtr_.swap(o.controllerPtr_);
  }
  void clear() { *this = PrimalSolution(); }
  scalar_array_t timeTrajectory_;
  vector_array_t stateTrajectory_;
  vector_array_t inputTrajectory_;
  size_array_t postEventIndices_;
  ModeSchedule modeSchedule_;
  std::unique_ptr<ControllerBase> controllerPtr_;
};
struct PerformanceIndex {   // upstream ocs2_oc/oc_data/PerformanceIndex.h
  scalar_t merit = 0.0, cost = 0.0, dualFeasibilitiesSSE = 0.0, dynamicsViolationSSE = 0.0, equalityConstraintsSSE = 0.0, inequalityConstraintsSSE = 0.0,
           equalityLagrangian = 0.0, inequalityLagrangian = 0.0;
};
}  // namespace ocs2
