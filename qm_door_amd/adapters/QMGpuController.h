// QMGpuController.h -- the non-invasive way to load the GPU path: a second pluginlib class deriving from qm::QMController that
// overrides the two virtual hooks (qm_controllers/include/qm_controllers/QMController.h:52,54).  qm_controllers itself is not
// modified; select it in config/controllers.yaml with `type: qm/QMGpuController`.  In this repository it is compiled against the
// type stand-ins of tests/adapters/mock (tests/test_adapters.py).
#pragma once
#include <ocs2_centroidal_model/CentroidalModelRbdConversions.h>
#include <ocs2_legged_robot_ros/gait/GaitReceiver.h>
#include <ocs2_ros_interfaces/synchronized_module/RosReferenceManager.h>
#include <qm_controllers/QMController.h>

#include "GpuMpc.h"
#include "GpuWbc.h"
#include "qmgpu.h"

namespace qm {

class QMGpuController : public QMController {
 public:
  ~QMGpuController() override {
    // The base destructor stops the MPC thread (QMController.cpp:343-347) -- but it runs AFTER this one, and mpcMrtInterface_ holds MPC_BASE&
    // (*mpc_): an advanceMpc() still in flight would touch a freed solver, a destroyed stream and freed device memory.  Stop the thread first
    // (both members are protected, QMController.h:82-83; joining twice is harmless: the base finds the thread no longer joinable).
    controllerRunning_ = false;
    if (mpcThread_.joinable()) mpcThread_.join();
    mpc_.reset(); wbc_.reset();          // the solver / WBC objects hold the handles' streams: they go before the handles
    qmgpu_destroy(mpcHandle_); qmgpu_destroy(wbcHandle_);
  }

 protected:
  // Replaces the first statement of QMController::setupMpc (QMController.cpp:288-289, the SqpMpc object) and repeats the rest of
  // that function (:290-306) on the new object: those lines only talk to MPC_BASE / SolverBase.
  void setupMpc(ros::NodeHandle& controller_nh) override {
    ensureHandles();
    auto solver = std::make_unique<GpuSqpSolver>(mpcHandle_, problem_, kMaxNodes, qmInterface_->getOptimalControlProblem());
    mpc_ = std::make_shared<GpuMpc>(qmInterface_->mpcSettings(), std::move(solver));
    finishMpcSetup(controller_nh);
  }
  // Replaces QMController::setupWbc (QMController.cpp:273-277).  The WBC runs on the ros_control update thread while the MPC runs on
  // mpcThread_ (QMController.cpp:316): calls on one qmgpu handle must be serialised, so each side owns its own handle / stream.
  void setupWbc(ros::NodeHandle& controller_nh, const std::string& taskFile) override {
    ensureHandles();
    wbc_ = std::make_shared<GpuWbc>(qmInterface_->getPinocchioInterface(), qmInterface_->getCentroidalModelInfo(), *eeKinematicsPtr_, *armEeKinematicsPtr_, controller_nh,
                                    wbcHandle_, problem_, /*variant=*/0);
    wbc_->loadTasksSetting(taskFile, false);
  }

 private:
  static constexpr int kMaxNodes = 128;   // timeHorizon 1.0 / dt 0.015 = 67 nodes + one per mode switch (task.info:79,141)
  void ensureHandles() {
    if (mpcHandle_) return;
    std::string task, urdf, ref;
    ros::param::get("/taskFile", task); ros::param::get("/urdfFile", urdf); ros::param::get("/referenceFile", ref);  // load_controller.launch:5-14
    if (qmgpu_load_problem(task.c_str(), urdf.c_str(), ref.c_str(), nullptr, &problem_) != QMGPU_OK) throw std::invalid_argument(qmgpu_last_error());
    if (qmgpu_create(&problem_, 0, 1, kMaxNodes, &mpcHandle_) != QMGPU_OK) throw std::runtime_error(qmgpu_last_error());
    if (qmgpu_create(&problem_, 0, 1, 1, &wbcHandle_) != QMGPU_OK) throw std::runtime_error(qmgpu_last_error());
  }
  // What QMController::setupMpc does after constructing the solver (QMController.cpp:290-306): rbd conversions, the gait receiver as
  // a synchronized module, the ROS reference manager (subscribed) as the solver's reference manager, and the two observation publishers.
  void finishMpcSetup(ros::NodeHandle& /*controller_nh*/) {
    rbdConversions_ = std::make_shared<CentroidalModelRbdConversions>(qmInterface_->getPinocchioInterface(), qmInterface_->getCentroidalModelInfo());
    const std::string robotName = "qm", gaitTopicPrefix = "legged_robot";
    ros::NodeHandle nh;
    auto gaitReceiver = std::make_shared<GaitReceiver>(nh, qmInterface_->getSwitchedModelReferenceManagerPtr()->getGaitSchedule(), gaitTopicPrefix);
    auto rosReferenceManager = std::make_shared<RosReferenceManager>(robotName, qmInterface_->getReferenceManagerPtr());
    rosReferenceManager->subscribe(nh);
    mpc_->getSolverPtr()->addSynchronizedModule(gaitReceiver);
    mpc_->getSolverPtr()->setReferenceManager(rosReferenceManager);
    observationPublisher_ = nh.advertise<ocs2_msgs::mpc_observation>(robotName + "_mpc_observation", 1);
    eeStatePublisher_ = nh.advertise<qm_msgs::ee_state>(robotName + "_mpc_observation_ee_state", 1);
  }
  qmgpu_problem problem_{};
  qmgpu_handle mpcHandle_ = nullptr, wbcHandle_ = nullptr;
};

}  // namespace qm
