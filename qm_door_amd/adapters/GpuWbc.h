// GpuWbc.h -- qm::WbcBase front end over the C ABI (include/qmgpu.h).  Header only; compiled inside the reference's catkin
// workspace.  In this repository it is compiled against the type stand-ins of tests/adapters/mock (tests/test_adapters.py) and
// executed on the GPU box through tests/adapters/adapter_driver.cpp.
//
// Seam: QMController::setupWbc (qm_controllers/src/QMController.cpp:273-277) creates `wbc_` as a std::shared_ptr<qm::WbcBase>;
// update() / loadTasksSetting() are virtual (qm_wbc/include/qm_wbc/WbcBase.h:31-34).  The controller only reads x.tail(18)
// (QMController.cpp:149).
#pragma once
#include <hip/hip_runtime_api.h>
#include <qm_wbc/WbcBase.h>

#include <algorithm>
#include <stdexcept>
#include <string>

#include "qmgpu.h"

namespace qm {

class GpuWbc : public WbcBase {
 public:
  // `variant` 0 = HierarchicalWbc task set, 1 = HierarchicalMpcWbc task set.
  GpuWbc(const ocs2::PinocchioInterface& pinocchioInterface, ocs2::CentroidalModelInfo info, const ocs2::PinocchioEndEffectorKinematics& eeKinematics,
         const ocs2::PinocchioEndEffectorKinematics& armEeKinematics, ros::NodeHandle& nh, qmgpu_handle handle, const qmgpu_problem& problem, int variant)
      : WbcBase(pinocchioInterface, std::move(info), eeKinematics, armEeKinematics, nh), h_(handle), P_(problem), variant_(variant) {
    hip(hipMalloc(&dev_, kBytes), "hipMalloc");
    hip(hipMemset(dev_, 0, kBytes), "hipMemset");  // inputLast_ starts at zero (WbcBase.cpp:42)
    hip(hipHostMalloc(&pinned_, kBytes, hipHostMallocDefault), "hipHostMalloc");
    hip(hipStreamCreate(&stream_), "hipStreamCreate");
    check(qmgpu_set_stream(h_, stream_));
  }
  ~GpuWbc() override {
    qmgpu_set_stream(h_, nullptr);
    if (stream_) (void)hipStreamDestroy(stream_);
    if (pinned_) (void)hipHostFree(pinned_);
    if (dev_) (void)hipFree(dev_);
  }
  GpuWbc(const GpuWbc&) = delete;
  GpuWbc& operator=(const GpuWbc&) = delete;

  ocs2::vector_t update(const ocs2::vector_t& stateDesired, const ocs2::vector_t& inputDesired, const ocs2::vector_t& rbdStateMeasured, size_t mode,
                        ocs2::scalar_t period, ocs2::scalar_t time) override {
    if (stateDesired.size() != 30 || inputDesired.size() != 30 || rbdStateMeasured.size() != 55) throw std::runtime_error("[GpuWbc] bad vector sizes");
    // one instance per call: 30 + 30 + 55 + 2 doubles and the mode up, 54 doubles and the status down (the batched entry point is for fleets)
    double* host = static_cast<double*>(pinned_);
    std::copy(stateDesired.data(), stateDesired.data() + 30, host + kXd);
    std::copy(inputDesired.data(), inputDesired.data() + 30, host + kUd);
    std::copy(rbdStateMeasured.data(), rbdStateMeasured.data() + 55, host + kRbd);
    host[kPeriod] = period; host[kTime] = time;
    reinterpret_cast<int32_t*>(host + kInts)[0] = static_cast<int32_t>(mode);
    double* dd = static_cast<double*>(dev_);
    hip(hipMemcpyAsync(dd, host, kInDoubles * sizeof(double), hipMemcpyHostToDevice, stream_), "hipMemcpyAsync H2D");
    qmgpu_wbc_args a{};
    a.batch = 1; a.variant = variant_;
    a.state_desired = dd + kXd; a.input_desired = dd + kUd; a.rbd_measured = dd + kRbd; a.period = dd + kPeriod; a.time = dd + kTime;
    a.mode = reinterpret_cast<int32_t*>(dd + kInts); a.input_last = dd + kIl; a.out = dd + kOut; a.out_status = reinterpret_cast<int32_t*>(dd + kStatus);
    check(qmgpu_wbc_solve_batch(h_, &a));
    hip(hipMemcpyAsync(host + kOut, dd + kOut, (kDoubles - kOut) * sizeof(double), hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync D2H");
    hip(hipStreamSynchronize(stream_), "hipStreamSynchronize");
    lastStatus_ = reinterpret_cast<int32_t*>(host + kStatus)[0];   // the reference drops qpOASES' return value (HoQp.cpp:143); kept here for diagnostics
    ocs2::vector_t out(54);
    std::copy(host + kOut, host + kOut + 54, out.data());
    return out;
  }

  // Gains / limits / friction are part of the qmgpu_problem given to qmgpu_create (qmgpu_load_problem reads the same task file).
  void loadTasksSetting(const std::string&, bool) override {}

  // Run-time gain changes (what WbcBase::dynamicCallback does with the dynamic_reconfigure server, WbcBase.cpp:74-121): edit the
  // settings copy and push it; takes effect for the next update(), no handle re-creation.
  qmgpu_settings& settings() { return P_.settings; }
  void pushSettings() { check(qmgpu_update_settings(h_, &P_.settings)); }
  int lastStatus() const { return lastStatus_; }

 private:
  // doubles: xDes[30] uDes[30] rbd[55] period time | mode (int32 in one double slot) | inputLast[30] | out[54] status (int32 in one double slot)
  static constexpr int kXd = 0, kUd = 30, kRbd = 60, kPeriod = 115, kTime = 116, kInts = 117, kInDoubles = 118, kIl = 118, kOut = 148, kStatus = 202, kDoubles = 203;
  static constexpr size_t kBytes = kDoubles * sizeof(double);
  static void check(int st) { if (st != QMGPU_OK) throw std::runtime_error(std::string("[GpuWbc] ") + qmgpu_strerror(st) + ": " + qmgpu_last_error()); }
  static void hip(hipError_t e, const char* what) { if (e != hipSuccess) throw std::runtime_error(std::string("[GpuWbc] ") + what + ": " + hipGetErrorString(e)); }
  qmgpu_handle h_;
  qmgpu_problem P_;
  int variant_;
  void* dev_ = nullptr;
  void* pinned_ = nullptr;
  hipStream_t stream_ = nullptr;
  int lastStatus_ = 0;
};

}  // namespace qm
