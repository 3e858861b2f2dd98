// QMGpuController.h -- the non-invasive way to load the GPU path: a second pluginlib class deriving from qm::QMController that
// overrides the two virtual hooks (qm_controllers/include/qm_controllers/QMController.h:52,54).  qm_controllers itself is not
// modified; select it in config/controllers.yaml with `type: qm/QMGpuController`.  NOT compiled in this repository (needs ROS).
#pragma once
#include <qm_controllers/QMController.h>

#include "GpuMpc.h"
#include "GpuWbc.h"
#include "qmgpu.h"

namespace qm {

class QMGpuController : public QMController {
 public:
  ~QMGpuController() override { qmgpu_destroy(handle_); }

 protected:
  void setupMpc(ros::NodeHandle& nh) override {
    ensureHandle();
    auto solver = std::make_unique<GpuSqpSolver>(handle_, problem_, qmInterface_->getSwitchedModelReferenceManagerPtr(), kMaxNodes);
    mpc_ = std::make_shared<GpuMpc>(qmInterface_->mpcSettings(), std::move(solver));
    // the remainder of QMController::setupMpc (QMController.cpp:291-306) is unchanged: gait receiver, ROS reference manager, publishers
    finishMpcSetup(nh);
  }
  void setupWbc(ros::NodeHandle& controller_nh, const std::string& taskFile) override {
    ensureHandle();
    wbc_ = std::make_shared<GpuWbc>(qmInterface_->getPinocchioInterface(), qmInterface_->getCentroidalModelInfo(), *eeKinematicsPtr_, *armEeKinematicsPtr_, controller_nh, handle_, 0);
    wbc_->loadTasksSetting(taskFile, false);
  }

 private:
  static constexpr int kMaxNodes = 128;
  void ensureHandle() {
    if (handle_) return;
    std::string task, urdf, ref;
    ros::param::get("/taskFile", task); ros::param::get("/urdfFile", urdf); ros::param::get("/referenceFile", ref);  // load_controller.launch:5-14
    if (qmgpu_load_problem(task.c_str(), urdf.c_str(), ref.c_str(), nullptr, &problem_) != QMGPU_OK) throw std::invalid_argument(qmgpu_last_error());
    if (qmgpu_create(&problem_, 0, 1, kMaxNodes, &handle_) != QMGPU_OK) throw std::runtime_error(qmgpu_last_error());
  }
  void finishMpcSetup(ros::NodeHandle& nh);  // verbatim tail of QMController::setupMpc; see INTEGRATION.md
  qmgpu_problem problem_{};
  qmgpu_handle handle_ = nullptr;
};

}  // namespace qm
