#pragma once
#include <ros/ros.h>
#include "ocs2_oc/synchronized_module/ReferenceManagerInterface.h"
namespace ocs2 {
class RosReferenceManager final : public ReferenceManagerInterface {   // upstream decorator: forwards to the wrapped manager
 public:
  RosReferenceManager(std::string topicPrefix, std::shared_ptr<ReferenceManagerInterface> inner) : prefix(std::move(topicPrefix)), inner_(std::move(inner)) {}
  void subscribe(::ros::NodeHandle&) { subscribed = true; }
  void preSolverRun(scalar_t t0, scalar_t tf, const vector_t& x) override { inner_->preSolverRun(t0, tf, x); }
  const ModeSchedule& getModeSchedule() const override { return inner_->getModeSchedule(); }
  const TargetTrajectories& getTargetTrajectories() const override { return inner_->getTargetTrajectories(); }
  std::string prefix;
  bool subscribed = false;
 private:
  std::shared_ptr<ReferenceManagerInterface> inner_;
};
}  // namespace ocs2
