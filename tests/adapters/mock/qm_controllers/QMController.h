// TEST INFRASTRUCTURE ONLY -- the protected hooks and members of qm::QMController (qm_controllers/include/qm_controllers/QMController.h:46-92)
// that a derived controller sees.
#pragma once
#include <ros/ros.h>
#include "ocs2_legged_robot_ros/gait/GaitReceiver.h"
#include "ocs2_ros_interfaces/synchronized_module/RosReferenceManager.h"
#include "qm_interface/QMInterface.h"
#include "qm_wbc/WbcBase.h"
#include "ocs2_centroidal_model/CentroidalModelRbdConversions.h"
namespace qm {
class QMController {
 public:
  QMController() = default;
  virtual ~QMController() = default;
 protected:
  virtual void setupMpc(ros::NodeHandle& controller_nh) { (void)controller_nh; }
  virtual void setupWbc(ros::NodeHandle& controller_nh, const std::string& taskFile) { (void)controller_nh; (void)taskFile; }
  std::shared_ptr<QMInterface> qmInterface_ = std::make_shared<QMInterface>();
  std::shared_ptr<PinocchioEndEffectorKinematics> eeKinematicsPtr_ = std::make_shared<PinocchioEndEffectorKinematics>();
  std::shared_ptr<PinocchioEndEffectorKinematics> armEeKinematicsPtr_ = std::make_shared<PinocchioEndEffectorKinematics>();
  std::shared_ptr<CentroidalModelRbdConversions> rbdConversions_;
  std::shared_ptr<MPC_BASE> mpc_;
  std::shared_ptr<WbcBase> wbc_;
  ros::Publisher observationPublisher_, eeStatePublisher_;
};
}  // namespace qm
