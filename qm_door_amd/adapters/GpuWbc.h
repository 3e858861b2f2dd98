// GpuWbc.h -- qm::WbcBase front end over the C ABI (include/qmgpu.h).  Header only; compiled inside the reference's catkin
// workspace (needs ROS / OCS2 / Pinocchio headers, none of which exist in the build container of this repository, so this file is
// NOT compiled or tested here -- see INTEGRATION.md).
//
// Seam: QMController::setupWbc (qm_controllers/src/QMController.cpp:273-277) creates `wbc_` as a std::shared_ptr<qm::WbcBase>;
// update() / loadTasksSetting() are virtual (qm_wbc/include/qm_wbc/WbcBase.h:31-34).  The controller only reads x.tail(18)
// (QMController.cpp:149).
#pragma once
#include <hip/hip_runtime_api.h>
#include <qm_wbc/WbcBase.h>

#include <stdexcept>
#include <string>

#include "qmgpu.h"

namespace qm {

class GpuWbc : public WbcBase {
 public:
  // `variant` 0 = HierarchicalWbc task set, 1 = HierarchicalMpcWbc task set.
  GpuWbc(const ocs2::PinocchioInterface& pinocchioInterface, ocs2::CentroidalModelInfo info, const ocs2::PinocchioEndEffectorKinematics& eeKinematics,
         const ocs2::PinocchioEndEffectorKinematics& armEeKinematics, ros::NodeHandle& nh, qmgpu_handle handle, int variant)
      : WbcBase(pinocchioInterface, std::move(info), eeKinematics, armEeKinematics, nh), h_(handle), variant_(variant) {
    check(hipMalloc(&dev_, kBytes) == hipSuccess ? QMGPU_OK : QMGPU_ERR_HIP);
    check(hipMemset(dev_, 0, kBytes) == hipSuccess ? QMGPU_OK : QMGPU_ERR_HIP);  // inputLast_ starts at zero (WbcBase.cpp:42)
  }
  ~GpuWbc() override { hipFree(dev_); }

  ocs2::vector_t update(const ocs2::vector_t& stateDesired, const ocs2::vector_t& inputDesired, const ocs2::vector_t& rbdStateMeasured, size_t mode,
                        ocs2::scalar_t period, ocs2::scalar_t time) override {
    // one instance per call: 30 + 30 + 55 doubles up, 54 doubles down (the batched entry point is for fleets / benchmarks)
    double host[kDoubles] = {0};
    std::copy(stateDesired.data(), stateDesired.data() + 30, host + kXd);
    std::copy(inputDesired.data(), inputDesired.data() + 30, host + kUd);
    std::copy(rbdStateMeasured.data(), rbdStateMeasured.data() + 55, host + kRbd);
    host[kPeriod] = period; host[kTime] = time;
    const int32_t modeI = static_cast<int32_t>(mode);
    char* d = static_cast<char*>(dev_);
    hipMemcpy(d, host, kInBytes, hipMemcpyHostToDevice);
    hipMemcpy(d + kModeOff, &modeI, sizeof(modeI), hipMemcpyHostToDevice);
    qmgpu_wbc_args a{};
    a.batch = 1; a.variant = variant_;
    double* dd = static_cast<double*>(dev_);
    a.state_desired = dd + kXd; a.input_desired = dd + kUd; a.rbd_measured = dd + kRbd; a.period = dd + kPeriod; a.time = dd + kTime;
    a.mode = reinterpret_cast<int32_t*>(d + kModeOff); a.input_last = dd + kIl; a.out = dd + kOut; a.out_status = reinterpret_cast<int32_t*>(d + kModeOff) + 1;
    check(qmgpu_wbc_solve_batch(h_, &a));
    check(qmgpu_synchronize(h_));
    ocs2::vector_t out(54);
    hipMemcpy(out.data(), dd + kOut, 54 * sizeof(double), hipMemcpyDeviceToHost);
    return out;
  }

  // Gains / limits / friction are part of the qmgpu_problem given to qmgpu_create (qmgpu_load_problem reads the same task file).
  void loadTasksSetting(const std::string&, bool) override {}

 private:
  static constexpr int kXd = 0, kUd = 30, kRbd = 60, kPeriod = 115, kTime = 116, kIl = 117, kOut = 147, kDoubles = 201;
  static constexpr size_t kInBytes = 117 * sizeof(double), kModeOff = kDoubles * sizeof(double), kBytes = kModeOff + 16;
  static void check(int st) { if (st != QMGPU_OK) throw std::runtime_error(std::string("[GpuWbc] ") + qmgpu_strerror(st) + ": " + qmgpu_last_error()); }
  qmgpu_handle h_;
  int variant_;
  void* dev_ = nullptr;
};

}  // namespace qm
