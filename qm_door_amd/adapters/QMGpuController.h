// QMGpuController.h -- the non-invasive way to load the GPU path: pluginlib classes deriving from the reference's two controllers
// (qm_controllers/src/QMController.cpp:450-451 exports qm::QMController and qm::QMMpcController) that override the two virtual hooks
// (qm_controllers/include/qm_controllers/QMController.h:52,54).  qm_controllers itself is not modified; select them in
// config/controllers.yaml with `type: qm/QMGpuController` (combined system, HierarchicalWbc) or `type: qm/QMGpuMpcController`
// (separated system: QMMpcController's arm position interface + HierarchicalMpcWbc, QMController.h:95-110, QMController.cpp:369-446).
// Both are ONE template: the hooks only talk to MPC_BASE / SolverBase / WbcBase, and the WBC task set (qmgpu_wbc_args::variant) follows the base.
// In this repository it is compiled against the type stand-ins of tests/adapters/mock (tests/test_adapters.py).
#pragma once
#include <ocs2_centroidal_model/CentroidalModelRbdConversions.h>
#include <ocs2_legged_robot_ros/gait/GaitReceiver.h>
#include <ocs2_ros_interfaces/synchronized_module/RosReferenceManager.h>
#include <qm_controllers/QMController.h>

#include <type_traits>

#include "GpuMpc.h"
#include "GpuWbc.h"
#include "qmgpu.h"

namespace qm {

// WBC task set of a controller class: QMMpcController::setupWbc builds HierarchicalMpcWbc (QMController.cpp:411-415) -> variant 1,
// QMController::setupWbc builds HierarchicalWbc (QMController.cpp:273-277) -> variant 0
template <class Base> struct GpuWbcVariant { static constexpr int value = std::is_base_of<QMMpcController, Base>::value ? 1 : 0; };

template <class Base>
class QMGpuControllerT : public Base {
  static_assert(std::is_base_of<QMController, Base>::value, "Base must be qm::QMController or a class derived from it");

 public:
  static constexpr int kWbcVariant = GpuWbcVariant<Base>::value;
  ~QMGpuControllerT() override {
    // The base destructor stops the MPC thread (QMController.cpp:343-347) -- but it runs AFTER this one, and mpcMrtInterface_ holds MPC_BASE&
    // (*mpc_): an advanceMpc() still in flight would touch a freed solver, a destroyed stream and freed device memory.  Stop the thread first
    // (both members are protected, QMController.h:82-83; joining twice is harmless: the base finds the thread no longer joinable).
    this->controllerRunning_ = false;
    if (this->mpcThread_.joinable()) this->mpcThread_.join();
    this->mpc_.reset(); this->wbc_.reset();          // the solver / WBC objects hold the handles' streams: they go before the handles
    qmgpu_destroy(mpcHandle_); qmgpu_destroy(wbcHandle_);
  }
  // the handles behind the two seams (per-kernel timing: qmgpu_enable_timing / qmgpu_kernel_ms_mean)
  qmgpu_handle mpcHandle() const { return mpcHandle_; }
  qmgpu_handle wbcHandle() const { return wbcHandle_; }

 protected:
  // Replaces the first statement of QMController::setupMpc (QMController.cpp:288-289, the SqpMpc object) and repeats the rest of
  // that function (:290-306) on the new object: those lines only talk to MPC_BASE / SolverBase.  QMMpcController does not override setupMpc.
  void setupMpc(ros::NodeHandle& controller_nh) override {
    ensureHandles();
    auto solver = std::make_unique<GpuSqpSolver>(mpcHandle_, problem_, kMaxNodes, this->qmInterface_->getOptimalControlProblem());
    this->mpc_ = std::make_shared<GpuMpc>(this->qmInterface_->mpcSettings(), std::move(solver));
    finishMpcSetup(controller_nh);
  }
  // Replaces QMController::setupWbc (QMController.cpp:273-277) / QMMpcController::setupWbc (QMController.cpp:411-415).  The WBC runs on the
  // ros_control update thread while the MPC runs on mpcThread_ (QMController.cpp:316): calls on one qmgpu handle must be serialised, so each
  // side owns its own handle / stream.
  void setupWbc(ros::NodeHandle& controller_nh, const std::string& taskFile) override {
    ensureHandles();
    this->wbc_ = std::make_shared<GpuWbc>(this->qmInterface_->getPinocchioInterface(), this->qmInterface_->getCentroidalModelInfo(), *this->eeKinematicsPtr_,
                                          *this->armEeKinematicsPtr_, controller_nh, wbcHandle_, problem_, kWbcVariant);
    this->wbc_->loadTasksSetting(taskFile, kWbcVariant == 1);   // verbose as the respective base passes it (QMController.cpp:276 false, :414 true)
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
    this->rbdConversions_ = std::make_shared<CentroidalModelRbdConversions>(this->qmInterface_->getPinocchioInterface(), this->qmInterface_->getCentroidalModelInfo());
    const std::string robotName = "qm", gaitTopicPrefix = "legged_robot";
    ros::NodeHandle nh;
    auto gaitReceiver = std::make_shared<GaitReceiver>(nh, this->qmInterface_->getSwitchedModelReferenceManagerPtr()->getGaitSchedule(), gaitTopicPrefix);
    auto rosReferenceManager = std::make_shared<RosReferenceManager>(robotName, this->qmInterface_->getReferenceManagerPtr());
    rosReferenceManager->subscribe(nh);
    this->mpc_->getSolverPtr()->addSynchronizedModule(gaitReceiver);
    this->mpc_->getSolverPtr()->setReferenceManager(rosReferenceManager);
    this->observationPublisher_ = nh.template advertise<ocs2_msgs::mpc_observation>(robotName + "_mpc_observation", 1);
    this->eeStatePublisher_ = nh.template advertise<qm_msgs::ee_state>(robotName + "_mpc_observation_ee_state", 1);
  }
  qmgpu_problem problem_{};
  qmgpu_handle mpcHandle_ = nullptr, wbcHandle_ = nullptr;
};

using QMGpuController = QMGpuControllerT<QMController>;         // type: qm/QMGpuController     (replaces qm/QMController)
using QMGpuMpcController = QMGpuControllerT<QMMpcController>;   // type: qm/QMGpuMpcController  (replaces qm/QMMpcController)

}  // namespace qm
