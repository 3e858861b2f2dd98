// TEST INFRASTRUCTURE ONLY -- the virtual surface of qm::WbcBase (qm_wbc/include/qm_wbc/WbcBase.h:22-34) the GPU adapter overrides.
// As in the reference the class declares NO destructor (implicit, non-virtual: the controller holds it in a std::shared_ptr created with
// make_shared<Derived>, which remembers the right deleter) and its dynamic_reconfigure callback is private and non-virtual
// (WbcBase.h:61, WbcBase.cpp:62-67: bound to the server of namespace <controller>/wbc inside the base constructor).
#pragma once
#include <ros/ros.h>
#include <functional>
#include <memory>
#include "ocs2_core/Types.h"
#include "ocs2_pinocchio_interface/PinocchioInterface.h"
namespace qm_wbc {
struct WbcWeightConfig {   // qm_wbc/cfg/wbcWigeht.cfg:7-47 (names as WbcBase::dynamicCallback reads them, WbcBase.cpp:74-118)
  double kp_arm_joint_1 = 0, kp_arm_joint_2 = 0, kp_arm_joint_3 = 0, kp_arm_joint_4 = 0, kp_arm_joint_5 = 0, kp_arm_joint_6 = 0;
  double kd_arm_joint_1 = 0, kd_arm_joint_2 = 0, kd_arm_joint_3 = 0, kd_arm_joint_4 = 0, kd_arm_joint_5 = 0, kd_arm_joint_6 = 0;
  double kp_ee_linear_x = 0, kp_ee_linear_y = 0, kp_ee_linear_z = 0, kd_ee_linear_x = 0, kd_ee_linear_y = 0, kd_ee_linear_z = 0;
  double kp_ee_angular_x = 0, kp_ee_angular_y = 0, kp_ee_angular_z = 0, kd_ee_angular_x = 0, kd_ee_angular_y = 0, kd_ee_angular_z = 0;
  double kp_swing = 0, kd_swing = 0, baseHeightKp = 0, baseHeightKd = 0, kp_base_angular = 0, kd_base_angular = 0, kp_base_linear = 0, kd_base_linear = 0;
};
}  // namespace qm_wbc
namespace dynamic_reconfigure {
template <class Config> class Server {   // dynamic_reconfigure/server.h: the two members the adapters use
 public:
  using CallbackType = std::function<void(Config&, uint32_t)>;
  explicit Server(const ros::NodeHandle&) {}
  void setCallback(const CallbackType& cb) { cb_ = cb; }
  void fire(Config c) { if (cb_) cb_(c, 0); }   // test hook: what a reconfigure request does (on a spinner thread)
 private:
  CallbackType cb_;
};
}  // namespace dynamic_reconfigure
namespace qm {
using namespace ocs2;
class WbcBase {
 public:
  WbcBase(const PinocchioInterface&, CentroidalModelInfo, const PinocchioEndEffectorKinematics&, const PinocchioEndEffectorKinematics&, ros::NodeHandle&) {}
  virtual vector_t update(const vector_t& stateDesired, const vector_t& inputDesired, const vector_t& rbdStateMeasured, size_t mode, scalar_t period, scalar_t time) {
    (void)stateDesired; (void)inputDesired; (void)rbdStateMeasured; (void)mode; (void)period; (void)time; return vector_t();
  }
  virtual void loadTasksSetting(const std::string& taskFile, bool verbose) { (void)taskFile; (void)verbose; }
};
}  // namespace qm
