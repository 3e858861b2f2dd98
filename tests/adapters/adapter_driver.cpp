// TEST INFRASTRUCTURE ONLY.  Drives the reference-side adapters (qm_door_amd/adapters/*.h) exactly the way qm_controllers would:
// a controller derived from qm::QMGpuController runs its setupMpc / setupWbc hooks, then the MPC thread's `mpc_->run(t, x)` and the
// update tick's `wbc_->update(...)`.  OCS2 / ROS types are the stand-ins of tests/adapters/mock.  Reads one text file of inputs,
// writes one text file of results; tests/test_adapters.py compares them with the same solve made directly through the C ABI.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <thread>
#include <iostream>
#include <vector>

#include "QMGpuController.h"

namespace {
template <class Ctl>
struct Harness : Ctl {
  using Ctl::setupMpc;
  using Ctl::setupWbc;
  using qm::QMController::mpc_;
  using qm::QMController::qmInterface_;
  using qm::QMController::wbc_;
  using qm::QMController::rbdConversions_;
  using qm::QMController::setupMrt;
  using qm::QMController::mpcRunning_;
  using qm::QMController::mrtRuns_;
  using qm::QMController::mrtState_;
  using qm::QMController::mrtTime_;
};
ocs2::vector_t readVec(std::istream& in, int n) { ocs2::vector_t v(n); for (int i = 0; i < n; ++i) in >> v[i]; return v; }
void writeVec(std::ostream& out, const ocs2::vector_t& v) { out.precision(17); for (long i = 0; i < v.size(); ++i) out << v[i] << (i + 1 < v.size() ? ' ' : '\n'); }
}  // namespace

namespace {
// ---------------------------------------------------------------------------------------------------------------------------------------------
// Latency at the plugin's operating point: ONE robot, the reference's own instrumentation points -- mpcTimer_ around advanceMpc
// (QMController.cpp:322-324) and wbcTimer_ around wbc_->update (QMController.cpp:146-148), printed as max / average (QMController.cpp:348-355).
// timeHorizon 1.0 s, dt 0.015, trot: ~67 nodes + the mode switches inside the horizon, warm start from the previous solution, 100 Hz; ten 1 kHz WBC
// ticks per MPC run on the plan just computed.  Wall clock per call = pinned staging + H2D + the batch-1 launch chain + D2H + one stream synchronisation.
struct Stat { std::vector<double> ms; void add(double v) { ms.push_back(v); }
  void write(std::ostream& o) const { std::vector<double> s = ms; std::sort(s.begin(), s.end()); double sum = 0; for (double v : s) sum += v;
    o << "{\"calls\": " << s.size() << ", \"avg_ms\": " << sum / s.size() << ", \"max_ms\": " << s.back() << ", \"min_ms\": " << s.front() << ", \"median_ms\": " << s[s.size() / 2]
      << ", \"p99_ms\": " << s[std::min(s.size() - 1, size_t(0.99 * s.size()))] << "}"; } };

void interpPlan(const ocs2::PrimalSolution& p, double t, ocs2::vector_t& x, ocs2::vector_t& u) {
  const auto& T = p.timeTrajectory_;
  size_t k = 0;
  while (k + 2 < T.size() && T[k + 1] < t) ++k;
  const double len = T[k + 1] - T[k], a = len > 1e-12 ? std::min(1.0, std::max(0.0, (T[k + 1] - t) / len)) : 1.0;
  x.resize(30); u.resize(30);
  for (int i = 0; i < 30; ++i) { x[i] = a * p.stateTrajectory_[k][i] + (1.0 - a) * p.stateTrajectory_[k + 1][i]; u[i] = a * p.inputTrajectory_[k][i] + (1.0 - a) * p.inputTrajectory_[k + 1][i]; }
}

template <class Ctl>
int runLatency(const char* outPath, int runs, int ticksPerRun) {
  Harness<Ctl> ctl;
  ros::NodeHandle nh;
  std::string task; ros::param::get("/taskFile", task);
  qmgpu_problem P{};
  { std::string urdf, ref; ros::param::get("/urdfFile", urdf); ros::param::get("/referenceFile", ref);
    if (qmgpu_load_problem(task.c_str(), urdf.c_str(), ref.c_str(), nullptr, &P) != QMGPU_OK) throw std::runtime_error(qmgpu_last_error()); }
  auto& rm = *ctl.qmInterface_->referenceManagerPtr_;
  // stance for 0.2 s, then the trot of gait.info (0.35 s per phase) past the last horizon; hold-pose target (StartingPosition.h:13-15 offsets)
  const double tEnd = runs * 0.01 + 1.6;
  rm.modeSchedule.eventTimes.clear(); rm.modeSchedule.modeSequence = {15};
  for (double t = 0.2; t < tEnd; t += 0.35) { rm.modeSchedule.eventTimes.push_back(t); rm.modeSchedule.modeSequence.push_back(rm.modeSchedule.modeSequence.size() % 2 ? 9 : 6); }
  ocs2::vector_t x0(30); for (int i = 0; i < 30; ++i) x0[i] = P.settings.initial_state[i];
  ocs2::vector_t tgt(37); for (int i = 0; i < 30; ++i) tgt[i] = x0[i];
  tgt[30] = x0[6] + 0.6; tgt[31] = x0[7]; tgt[32] = x0[8] + 0.036; tgt[33] = 0; tgt[34] = 0; tgt[35] = 0; tgt[36] = 1;
  rm.targetTrajectories.timeTrajectory = {0.0}; rm.targetTrajectories.stateTrajectory = {tgt}; rm.targetTrajectories.inputTrajectory = {ocs2::vector_t(30)};
  ctl.qmInterface_->mpcSettings_.timeHorizon_ = 1.0;
  ctl.setupMpc(nh); ctl.setupWbc(nh, task);
  const bool coldWbc = std::getenv("QM_WBC_CARRY") == nullptr;      // default: every tick cold, as the reference's qpOASES call; QM_WBC_CARRY: the working sets travel from tick to tick
  dynamic_cast<qm::GpuWbc*>(ctl.wbc_.get())->carryWorkingSet(!coldWbc);
  std::ofstream out(outPath);
  out.precision(6);
  out << "{\"controller\": \"" << (Ctl::kWbcVariant ? "qm/QMGpuMpcController" : "qm/QMGpuController") << "\", \"wbc_variant\": " << Ctl::kWbcVariant
      << ", \"wbc_working_set\": \"" << (coldWbc ? "cold every tick" : "carried from tick to tick") << "\"";
  for (int pass = 0; pass < 2; ++pass) {       // pass 0: wall clock only; pass 1: the same loop with HIP-event kernel timing on both handles
    qmgpu_enable_timing(ctl.mpcHandle(), pass); qmgpu_enable_timing(ctl.wbcHandle(), pass);
    ctl.mpc_->reset();
    Stat mpc, wbc; double nodes = 0; int wbcStatus = 0;
    ocs2::vector_t x = x0, xd, ud;
    ocs2::PrimalSolution plan;
    for (int r = -5; r < runs; ++r) {            // five untimed warm-up cycles (first launches, lazy module load)
      const double t0 = std::max(r, 0) * 0.01;
      if (r > 0) { interpPlan(plan, t0, x, ud); }
      auto a = std::chrono::steady_clock::now();
      ctl.mpc_->run(t0, x);
      auto b = std::chrono::steady_clock::now();
      ctl.mpc_->getSolverPtr()->getPrimalSolution(t0 + 1.0, &plan);
      if (r >= 0) { mpc.add(std::chrono::duration<double, std::milli>(b - a).count()); nodes += double(plan.timeTrajectory_.size() - 1); }
      for (int j = 0; j < ticksPerRun; ++j) {
        const double t = t0 + j * 0.001;
        interpPlan(plan, t, xd, ud);
        ocs2::vector_t xh, uh; interpPlan(plan, t + 0.001, xh, uh);
        ocs2::vector_t rbd(55);
        for (int i = 0; i < 3; ++i) { rbd[i] = xd[9 + i]; rbd[3 + i] = xd[6 + i]; rbd[27 + i] = (xh[6 + i] - xd[6 + i]) / 0.001; rbd[24 + i] = (xh[11 - i] - xd[11 - i]) / 0.001; }
        for (int i = 0; i < 18; ++i) { rbd[6 + i] = xd[12 + i]; rbd[30 + i] = ud[12 + i]; }
        rbd[54] = 1.0;
        size_t k = 0; while (k + 2 < plan.timeTrajectory_.size() && plan.timeTrajectory_[k + 1] < t) ++k;
        const size_t mode = plan.modeSchedule_.modeAtTime(plan.timeTrajectory_[k]);
        auto c = std::chrono::steady_clock::now();
        (void)ctl.wbc_->update(xd, ud, rbd, mode, 0.001, 20.0 + t);
        auto d = std::chrono::steady_clock::now();
        if (r >= 0) { wbc.add(std::chrono::duration<double, std::milli>(d - c).count()); wbcStatus += dynamic_cast<qm::GpuWbc*>(ctl.wbc_.get())->lastStatus() != 0; }
      }
    }
    out << ", \"" << (pass ? "with_kernel_timing" : "wall_clock") << "\": {\"mpc_run\": "; mpc.write(out); out << ", \"wbc_update\": "; wbc.write(out);
    out << ", \"mean_nodes\": " << nodes / runs << ", \"wbc_status_nonzero\": " << wbcStatus;
    if (pass) {
      double m[6] = {0}, w[6] = {0};
      qmgpu_kernel_ms_mean(ctl.mpcHandle(), std::min(runs, 64), m); qmgpu_kernel_ms_mean(ctl.wbcHandle(), std::min(runs * ticksPerRun, 64), w);
      out << ", \"kernel_ms_mean\": {\"ad_node\": " << m[0] << ", \"lq_node\": " << m[1] << ", \"riccati\": " << m[2] << ", \"linesearch\": " << m[3] << ", \"mpc_launch_chain\": " << m[5]
          << ", \"wbc\": " << w[4] << "}";
    }
    out << "}";
  }
  out << "}\n";
  return 0;
}

template <class Ctl> int runDriver(int argc, char** argv);
}  // namespace

int main(int argc, char** argv) {
  // adapter_driver task.info robot.urdf reference.info inputs.txt outputs.txt [variant]      parity run (tests/test_adapters.py)
  // adapter_driver task.info robot.urdf reference.info --latency out.json [variant] [runs] [ticks per run]
  if (argc < 6) { std::fprintf(stderr, "usage: adapter_driver task.info robot.urdf reference.info (inputs.txt outputs.txt | --latency out.json) [variant 0|1] [runs] [ticks]\n"); return 2; }
  const int variant = argc > 6 ? std::atoi(argv[6]) : 0;
  try {
    ros::param::store()["/taskFile"] = argv[1]; ros::param::store()["/urdfFile"] = argv[2]; ros::param::store()["/referenceFile"] = argv[3];
    if (std::string(argv[4]) == "--latency") {
      const int runs = argc > 7 ? std::atoi(argv[7]) : 200, ticks = argc > 8 ? std::atoi(argv[8]) : 10;
      return variant ? runLatency<qm::QMGpuMpcController>(argv[5], runs, ticks) : runLatency<qm::QMGpuController>(argv[5], runs, ticks);
    }
    return variant ? runDriver<qm::QMGpuMpcController>(argc, argv) : runDriver<qm::QMGpuController>(argc, argv);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "adapter_driver: %s\n", e.what());
    return 1;
  }
}

namespace {
template <class Ctl>
int runDriver(int /*argc*/, char** argv) {
  {
    std::ifstream in(argv[4]);
    if (!in) throw std::runtime_error("cannot open inputs");
    std::ofstream out(argv[5]);
    Harness<Ctl> ctl;
    // reference manager contents: what the gait receiver and the target subscriber would have installed before the solver runs
    auto& rm = *ctl.qmInterface_->referenceManagerPtr_;
    int nev = 0, K = 0, runs = 0;
    in >> nev;
    rm.modeSchedule.eventTimes.resize(nev); rm.modeSchedule.modeSequence.resize(nev + 1);
    for (auto& t : rm.modeSchedule.eventTimes) in >> t;
    for (auto& m : rm.modeSchedule.modeSequence) in >> m;
    in >> K;
    rm.targetTrajectories.timeTrajectory.resize(K);
    for (auto& t : rm.targetTrajectories.timeTrajectory) in >> t;
    for (int k = 0; k < K; ++k) { rm.targetTrajectories.stateTrajectory.push_back(readVec(in, 37)); rm.targetTrajectories.inputTrajectory.push_back(ocs2::vector_t(30)); }
    double horizon = 0.0;
    in >> horizon >> runs;
    ctl.qmInterface_->mpcSettings_.timeHorizon_ = horizon;

    ros::NodeHandle nh;
    ctl.setupMpc(nh);
    ctl.setupWbc(nh, argv[1]);
    if (!ctl.mpc_ || !ctl.wbc_ || !ctl.rbdConversions_) throw std::runtime_error("hooks did not install mpc_ / wbc_ / rbdConversions_");
    if (ctl.mpc_->getSolverPtr()->getOptimalControlProblem().id != ctl.qmInterface_->getOptimalControlProblem().id) throw std::runtime_error("solver does not hand back the interface's OCP");

    double lastT = 0.0;
    ocs2::vector_t lastX;
    for (int r = 0; r < runs; ++r) {          // MPC thread: advanceMpc -> MPC_BASE::run(t, x)
      double t = 0.0;
      in >> t;
      const ocs2::vector_t x = readVec(in, 30);
      lastT = t; lastX = x;
      ctl.mpc_->run(t, x);
      ocs2::PrimalSolution p;
      ctl.mpc_->getSolverPtr()->getPrimalSolution(t + horizon, &p);
      const size_t n1 = p.timeTrajectory_.size();
      out << "run " << r << ' ' << n1 << ' ' << ctl.mpc_->getSolverPtr()->getNumIterations() << ' ' << rm.preSolverRuns << '\n';
      out.precision(17);
      for (size_t k = 0; k < n1; ++k) out << p.timeTrajectory_[k] << (k + 1 < n1 ? ' ' : '\n');
      for (size_t k = 0; k < n1; ++k) writeVec(out, p.stateTrajectory_[k]);
      for (size_t k = 0; k < n1; ++k) writeVec(out, p.inputTrajectory_[k]);
      const auto* ff = dynamic_cast<const ocs2::FeedforwardController*>(p.controllerPtr_.get());
      out << "policy " << (ff ? ff->uffArray_.size() : 0) << ' ' << ctl.mpc_->getSolverPtr()->getPerformanceIndeces().merit << '\n';
    }
    int ticks = 0;
    in >> ticks;
    ocs2::vector_t lastXd, lastUd, lastRbd;
    size_t lastMode = 15;
    for (int k = 0; k < ticks; ++k) {          // update(): wbc_->update(optimizedState, optimizedInput, measuredRbdState, plannedMode, period, time)
      const ocs2::vector_t xd = readVec(in, 30), ud = readVec(in, 30), rbd = readVec(in, 55);
      size_t mode = 0; double period = 0.0, time = 0.0;
      in >> mode >> period >> time;
      if (!in) throw std::runtime_error("input file too short");
      lastXd = xd; lastUd = ud; lastRbd = rbd; lastMode = mode;
      out << "wbc " << k << '\n';
      writeVec(out, ctl.wbc_->update(xd, ud, rbd, mode, period, time));
    }
    // run-time gain change: a dynamic_reconfigure request on <controller>/wbc_gpu arrives on ANOTHER thread while update() ticks; it is staged
    // under the adapter's mutex and applied by the next update().  Same gains -> same torques; doubled base-height gain -> different torques.
    auto* gw = dynamic_cast<qm::GpuWbc*>(ctl.wbc_.get());
    if (!gw) throw std::runtime_error("wbc_ is not a GpuWbc");
    if (ticks > 0) {
      in.clear();
      const qmgpu_settings s0 = gw->settingsCopy();
      qm_wbc::WbcWeightConfig cfg;
      cfg.kp_arm_joint_1 = s0.kp_arm_joint[0]; cfg.kp_arm_joint_2 = s0.kp_arm_joint[1]; cfg.kp_arm_joint_3 = s0.kp_arm_joint[2]; cfg.kp_arm_joint_4 = s0.kp_arm_joint[3];
      cfg.kp_arm_joint_5 = s0.kp_arm_joint[4]; cfg.kp_arm_joint_6 = s0.kp_arm_joint[5];
      cfg.kd_arm_joint_1 = s0.kd_arm_joint[0]; cfg.kd_arm_joint_2 = s0.kd_arm_joint[1]; cfg.kd_arm_joint_3 = s0.kd_arm_joint[2]; cfg.kd_arm_joint_4 = s0.kd_arm_joint[3];
      cfg.kd_arm_joint_5 = s0.kd_arm_joint[4]; cfg.kd_arm_joint_6 = s0.kd_arm_joint[5];
      cfg.kp_ee_linear_x = s0.kp_ee_linear[0]; cfg.kp_ee_linear_y = s0.kp_ee_linear[1]; cfg.kp_ee_linear_z = s0.kp_ee_linear[2];
      cfg.kd_ee_linear_x = s0.kd_ee_linear[0]; cfg.kd_ee_linear_y = s0.kd_ee_linear[1]; cfg.kd_ee_linear_z = s0.kd_ee_linear[2];
      cfg.kp_ee_angular_x = s0.kp_ee_angular[0]; cfg.kp_ee_angular_y = s0.kp_ee_angular[1]; cfg.kp_ee_angular_z = s0.kp_ee_angular[2];
      cfg.kd_ee_angular_x = s0.kd_ee_angular[0]; cfg.kd_ee_angular_y = s0.kd_ee_angular[1]; cfg.kd_ee_angular_z = s0.kd_ee_angular[2];
      cfg.kp_swing = s0.kp_swing; cfg.kd_swing = s0.kd_swing; cfg.baseHeightKp = s0.kp_base_height; cfg.baseHeightKd = s0.kd_base_height;
      cfg.kp_base_angular = s0.kp_base_angular; cfg.kd_base_angular = s0.kd_base_angular; cfg.kp_base_linear = s0.kp_base_linear; cfg.kd_base_linear = s0.kd_base_linear;
      const ocs2::vector_t before = gw->update(lastXd, lastUd, lastRbd, lastMode, 0.002, 30.0);
      std::thread spinner([&]() { gw->gainServer().fire(cfg); });            // identical gains, from another thread, while ...
      const ocs2::vector_t during = gw->update(lastXd, lastUd, lastRbd, lastMode, 0.002, 30.002);   // ... the update thread ticks
      spinner.join();
      const ocs2::vector_t same = gw->update(lastXd, lastUd, lastRbd, lastMode, 0.002, 30.004);
      cfg.baseHeightKp *= 2.0;
      std::thread spinner2([&]() { gw->gainServer().fire(cfg); });
      spinner2.join();
      const ocs2::vector_t changed = gw->update(lastXd, lastUd, lastRbd, lastMode, 0.002, 30.006);
      double dSame = 0.0, dChanged = 0.0;
      for (int i = 36; i < 54; ++i) { dSame = std::max(dSame, std::fabs(same[i] - during[i])); dChanged = std::max(dChanged, std::fabs(changed[i] - same[i])); }
      out.precision(17);
      out << "gains " << dSame << ' ' << dChanged << ' ' << (gw->settingsCopy().kp_base_height / s0.kp_base_height) << ' ' << before.size() << '\n';
    }
    // teardown with the MPC thread running (QMController::setupMrt, QMController.cpp:309-335): the derived destructor must stop and join it before the
    // solver, its stream and the handles go -- the object is destroyed at the end of this block while advanceMpc() is being called in a loop
    if (runs > 0) {
      ctl.mrtTime_ = lastT; ctl.mrtState_ = lastX;
      ctl.setupMrt();
      ctl.mpcRunning_ = true;
      while (ctl.mrtRuns_ < 3) std::this_thread::sleep_for(std::chrono::milliseconds(1));
      out << "mrt " << ctl.mrtRuns_ << '\n';
    }
    // which controller class ran, which WBC task set its hook selected, whether the reference's own (private) QMMpcController::setupWbc was bypassed, and
    // the warning about the inert base-class gain server
    out << "class " << Ctl::kWbcVariant << ' ' << ros::warnings().size() << '\n';
    out << "done\n";
    out.flush();
  }
  return 0;
}
}  // namespace
