#pragma once
#include "ocs2_core/reference/ModeSchedule.h"
namespace ocs2 {
class ReferenceManagerInterface {   // upstream ocs2_oc/synchronized_module/ReferenceManagerInterface.h
 public:
  virtual ~ReferenceManagerInterface() = default;
  virtual void preSolverRun(scalar_t initTime, scalar_t finalTime, const vector_t& initState) = 0;
  virtual const ModeSchedule& getModeSchedule() const = 0;
  virtual const TargetTrajectories& getTargetTrajectories() const = 0;
};
class SolverSynchronizedModule {   // upstream ocs2_oc/synchronized_module/SolverSynchronizedModule.h
 public:
  virtual ~SolverSynchronizedModule() = default;
};
}  // namespace ocs2
