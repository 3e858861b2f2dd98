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
#include <memory>
#include <mutex>
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
    // Run-time gains.  WbcBase binds ITS dynamic_reconfigure server (<controller>/wbc) to a private, non-virtual callback that edits base-class
    // members this class never reads (WbcBase.h:61, WbcBase.cpp:62-67,74-121), so that server cannot reach the GPU path.  The same
    // qm_wbc::WbcWeightConfig is therefore served a second time under <controller>/wbc_gpu; requests arrive on a spinner thread, are staged
    // under the mutex and applied at the start of the next update() on the ros_control thread (calls on one qmgpu handle must be serialised).
    ros::NodeHandle nhGains(nh, "wbc_gpu");
    dynamicSrv_ = std::make_shared<dynamic_reconfigure::Server<qm_wbc::WbcWeightConfig>>(nhGains);
    dynamicSrv_->setCallback([this](qm_wbc::WbcWeightConfig& c, uint32_t) { stageGains(c); });
    // Launch files and rqt layouts written for the reference point at <controller>/wbc: say once, loudly, that it no longer does anything here.
    ROS_WARN_STREAM("[GpuWbc] WBC gains are served on " << nhGains.getNamespace() << " (dynamic_reconfigure); the base-class server " << nh.getNamespace()
                    << "/wbc stays advertised but edits members the GPU path never reads: requests sent there have NO effect");
  }
  ~GpuWbc() {   // (the reference's WbcBase declares no virtual destructor; the controller's shared_ptr was made from this type)
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
    std::lock_guard<std::mutex> lock(mutex_);
    if (settingsDirty_) { check(qmgpu_update_settings(h_, &P_.settings)); settingsDirty_ = false; }
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
    // opt-in (carryWorkingSet(true)): the solver state of the previous tick (the rows every level ended on, the point it ended at) stays on the device next to inputLast_ and
    // is the next tick's starting guess.  Same torques whatever the path (tests/test_closed_loop.py); measured at this plugin's operating point (profiles/r06*_adapter_latency.json):
    // the average update gets ~10 % shorter, the slowest one longer (a refuted guess costs its factorisation on top of the cold solve) -- so the default is the
    // reference's: every tick cold (HoQp.cpp:136-149).
    a.working_set = carry_ ? reinterpret_cast<uint64_t*>(dd + kWs) : nullptr;
    check(qmgpu_wbc_solve_batch(h_, &a));
    hip(hipMemcpyAsync(host + kOut, dd + kOut, (kWs - kOut) * sizeof(double), hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync D2H");
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
  // Programmatic route: edit under the lock, applied by the next update().
  template <class F> void editSettings(F&& edit) { std::lock_guard<std::mutex> lock(mutex_); edit(P_.settings); settingsDirty_ = true; }
  qmgpu_settings settingsCopy() { std::lock_guard<std::mutex> lock(mutex_); return P_.settings; }
  dynamic_reconfigure::Server<qm_wbc::WbcWeightConfig>& gainServer() { return *dynamicSrv_; }
  int lastStatus() const { return lastStatus_; }
  void carryWorkingSet(bool on) { std::lock_guard<std::mutex> lock(mutex_); if (on && !carry_) hip(hipMemset(static_cast<double*>(dev_) + kWs, 0, QMGPU_WBC_STATE_WORDS * sizeof(uint64_t)), "hipMemset"); carry_ = on; }

 private:
  // doubles: xDes[30] uDes[30] rbd[55] period time | mode (int32 in one double slot) | inputLast[30] | out[54] status (int32 in one double slot) | working-set record (device only)
  static constexpr int kXd = 0, kUd = 30, kRbd = 60, kPeriod = 115, kTime = 116, kInts = 117, kInDoubles = 118, kIl = 118, kOut = 148, kStatus = 202, kWs = 203,
                       kDoubles = 203 + QMGPU_WBC_STATE_WORDS;
  static constexpr size_t kBytes = kDoubles * sizeof(double);
  static void check(int st) { if (st != QMGPU_OK) throw std::runtime_error(std::string("[GpuWbc] ") + qmgpu_strerror(st) + ": " + qmgpu_last_error()); }
  static void hip(hipError_t e, const char* what) { if (e != hipSuccess) throw std::runtime_error(std::string("[GpuWbc] ") + what + ": " + hipGetErrorString(e)); }
  // WbcBase::dynamicCallback (WbcBase.cpp:74-121), field for field, into the settings block of the C ABI
  void stageGains(const qm_wbc::WbcWeightConfig& c) {
    std::lock_guard<std::mutex> lock(mutex_);
    qmgpu_settings& s = P_.settings;
    const double kpj[6] = {c.kp_arm_joint_1, c.kp_arm_joint_2, c.kp_arm_joint_3, c.kp_arm_joint_4, c.kp_arm_joint_5, c.kp_arm_joint_6};
    const double kdj[6] = {c.kd_arm_joint_1, c.kd_arm_joint_2, c.kd_arm_joint_3, c.kd_arm_joint_4, c.kd_arm_joint_5, c.kd_arm_joint_6};
    for (int i = 0; i < 6; ++i) { s.kp_arm_joint[i] = kpj[i]; s.kd_arm_joint[i] = kdj[i]; }
    s.kp_ee_linear[0] = c.kp_ee_linear_x; s.kp_ee_linear[1] = c.kp_ee_linear_y; s.kp_ee_linear[2] = c.kp_ee_linear_z;
    s.kd_ee_linear[0] = c.kd_ee_linear_x; s.kd_ee_linear[1] = c.kd_ee_linear_y; s.kd_ee_linear[2] = c.kd_ee_linear_z;
    s.kp_ee_angular[0] = c.kp_ee_angular_x; s.kp_ee_angular[1] = c.kp_ee_angular_y; s.kp_ee_angular[2] = c.kp_ee_angular_z;
    s.kd_ee_angular[0] = c.kd_ee_angular_x; s.kd_ee_angular[1] = c.kd_ee_angular_y; s.kd_ee_angular[2] = c.kd_ee_angular_z;
    s.kp_swing = c.kp_swing; s.kd_swing = c.kd_swing;
    s.kp_base_height = c.baseHeightKp; s.kd_base_height = c.baseHeightKd;
    s.kp_base_angular = c.kp_base_angular; s.kd_base_angular = c.kd_base_angular;
    s.kp_base_linear = c.kp_base_linear; s.kd_base_linear = c.kd_base_linear;
    settingsDirty_ = true;
  }
  qmgpu_handle h_;
  qmgpu_problem P_;
  int variant_;
  std::mutex mutex_;               // update() (ros_control thread) vs gain requests (spinner thread)
  bool settingsDirty_ = false;
  std::shared_ptr<dynamic_reconfigure::Server<qm_wbc::WbcWeightConfig>> dynamicSrv_;
  void* dev_ = nullptr;
  void* pinned_ = nullptr;
  hipStream_t stream_ = nullptr;
  int lastStatus_ = 0;
  bool carry_ = false;
};

}  // namespace qm
