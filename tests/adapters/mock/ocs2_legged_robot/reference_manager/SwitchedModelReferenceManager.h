#pragma once
#include "ocs2_legged_robot/gait/GaitSchedule.h"
#include "ocs2_oc/synchronized_module/ReferenceManagerInterface.h"
namespace ocs2 { namespace legged_robot {
class SwitchedModelReferenceManager : public ReferenceManagerInterface {   // upstream ocs2_legged_robot/reference_manager/SwitchedModelReferenceManager.h
 public:
  void preSolverRun(scalar_t, scalar_t, const vector_t&) override { ++preSolverRuns; }
  const ModeSchedule& getModeSchedule() const override { return modeSchedule; }
  const TargetTrajectories& getTargetTrajectories() const override { return targetTrajectories; }
  const std::shared_ptr<GaitSchedule>& getGaitSchedule() { return gaitSchedule; }
  ModeSchedule modeSchedule;
  TargetTrajectories targetTrajectories;
  std::shared_ptr<GaitSchedule> gaitSchedule = std::make_shared<GaitSchedule>();
  int preSolverRuns = 0;
};
}}
