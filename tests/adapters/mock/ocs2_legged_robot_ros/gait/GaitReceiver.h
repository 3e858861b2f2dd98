#pragma once
#include <ros/ros.h>
#include "ocs2_legged_robot/gait/GaitSchedule.h"
#include "ocs2_oc/synchronized_module/ReferenceManagerInterface.h"
namespace ocs2 { namespace legged_robot {
class GaitReceiver : public SolverSynchronizedModule {   // upstream ocs2_legged_robot_ros/gait/GaitReceiver.h
 public:
  GaitReceiver(::ros::NodeHandle, std::shared_ptr<GaitSchedule> gs, const std::string& robotName) : gaitSchedule(std::move(gs)), name(robotName) {}
  std::shared_ptr<GaitSchedule> gaitSchedule;
  std::string name;
};
}}
