// TEST INFRASTRUCTURE ONLY -- the protected hooks and members of qm::QMController (qm_controllers/include/qm_controllers/QMController.h:46-92)
// that a derived controller sees.
#pragma once
#include <ros/ros.h>
#include <atomic>
#include <chrono>
#include <thread>
#include "ocs2_legged_robot_ros/gait/GaitReceiver.h"
#include "ocs2_ros_interfaces/synchronized_module/RosReferenceManager.h"
#include "qm_interface/QMInterface.h"
#include "qm_wbc/WbcBase.h"
#include "ocs2_centroidal_model/CentroidalModelRbdConversions.h"
namespace qm {
class QMController {
 public:
  QMController() = default;
  virtual ~QMController() {   // qm_controllers/src/QMController.cpp:343-347 -- runs AFTER a derived class's destructor
    controllerRunning_ = false;
    if (mpcThread_.joinable()) mpcThread_.join();
  }
 protected:
  // qm_controllers/src/QMController.cpp:309-335: the MPC thread calls advanceMpc (-> MPC_BASE::run on *mpc_) while mpcRunning_
  virtual void setupMrt() {
    controllerRunning_ = true;
    mpcThread_ = std::thread([&]() {
      while (controllerRunning_) {
        try {
          if (mpcRunning_) { mpc_->run(mrtTime_, mrtState_); ++mrtRuns_; }
          std::this_thread::sleep_for(std::chrono::milliseconds(1));
        } catch (const std::exception&) { controllerRunning_ = false; }
      }
    });
  }
  std::thread mpcThread_;
  std::atomic_bool controllerRunning_{}, mpcRunning_{};
  std::atomic_int mrtRuns_{0};
  ocs2::scalar_t mrtTime_ = 0.0;
  ocs2::vector_t mrtState_;
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
// qm_controllers/include/qm_controllers/QMController.h:95-110: the separated-system controller overrides setupWbc PRIVATELY (building HierarchicalMpcWbc,
// QMController.cpp:411-415) together with the arm position-interface hooks; a class derived from it can still override the virtual.
class QMMpcController : public QMController {
 public:
  QMMpcController() = default;
  ~QMMpcController() = default;
  bool baseWbcHookRan_ = false;
 private:
  void setupWbc(ros::NodeHandle& controller_nh, const std::string& taskFile) { (void)controller_nh; (void)taskFile; baseWbcHookRan_ = true; }
};
}  // namespace qm
