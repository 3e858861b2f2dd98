// GpuMpc.h -- ocs2::MPC_BASE / ocs2::SolverBase front end over the C ABI (include/qmgpu.h).  Header only; compiled inside the
// reference's catkin workspace next to OCS2.  In this repository it is compiled against the minimal type stand-ins of
// tests/adapters/mock (tests/test_adapters.py: -fsyntax-only on CPU, and executed on the GPU box through tests/adapters/adapter_driver.cpp).
//
// Seam: QMController::setupMpc (qm_controllers/src/QMController.cpp:287-307) stores `mpc_` as std::shared_ptr<ocs2::MPC_BASE>
// and afterwards only uses getSolverPtr()->addSynchronizedModule / setReferenceManager (QMController.cpp:303-304) and
// MPC_MRT_Interface(*mpc_) (QMController.cpp:311).  A solver therefore has to implement SolverBase::runImpl and hand back a
// PrimalSolution; everything else (reference manager, gait receiver, MRT buffering) stays upstream code.
#pragma once
#include <hip/hip_runtime_api.h>
#include <ocs2_core/control/FeedforwardController.h>
#include <ocs2_mpc/MPC_BASE.h>
#include <ocs2_oc/oc_solver/SolverBase.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "qmgpu.h"

namespace qm {

// One call of qmgpu_mpc_solve_batch (sqp.sqpIteration iterations, task.info:77) per run, batch = 1, on the device behind `handle`.
// The mode schedule and the target trajectories come from the reference manager installed with SolverBase::setReferenceManager
// (QMController.cpp:304), exactly where upstream SqpSolver::runImpl reads them.
class GpuSqpSolver final : public ocs2::SolverBase {
 public:
  static constexpr int kMaxKnots = 32;

  // `ocp` is what upstream's SqpSolver receives (QMController.cpp:288-289 passes qmInterface_->getOptimalControlProblem()): kept as a copy only to
  // answer SolverBase::getOptimalControlProblem() -- the kernels evaluate the same problem from the qmgpu_problem loaded from the same files.
  GpuSqpSolver(qmgpu_handle handle, const qmgpu_problem& problem, int maxNodes, const ocs2::OptimalControlProblem& ocp) : h_(handle), P_(problem), maxNodes_(maxNodes), ocp_(ocp) {
    if (!handle || maxNodes < 1) throw std::invalid_argument("[GpuSqpSolver] bad arguments");
    layout();
    check(hipMalloc(&dev_, devBytes_), "hipMalloc");
    check(hipHostMalloc(&pinned_, std::max(inBytes_, outBytes_), hipHostMallocDefault), "hipHostMalloc");
    check(hipStreamCreate(&copyStream_), "hipStreamCreate");
    qmCheck(qmgpu_set_stream(h_, copyStream_));   // copies and kernels of one run are ordered on one stream
  }
  ~GpuSqpSolver() override {
    qmgpu_set_stream(h_, nullptr);
    if (copyStream_) (void)hipStreamDestroy(copyStream_);
    if (pinned_) (void)hipHostFree(pinned_);
    if (dev_) (void)hipFree(dev_);
  }
  GpuSqpSolver(const GpuSqpSolver&) = delete;
  GpuSqpSolver& operator=(const GpuSqpSolver&) = delete;

  void reset() override { haveSolution_ = false; primal_.clear(); log_.clear(); }
  ocs2::scalar_t getFinalTime() const override { return primal_.timeTrajectory_.empty() ? 0.0 : primal_.timeTrajectory_.back(); }
  void getPrimalSolution(ocs2::scalar_t /*finalTime*/, ocs2::PrimalSolution* out) const override { *out = primal_; }
  const ocs2::PerformanceIndex& getPerformanceIndeces() const override { return performance_; }
  size_t getNumIterations() const override { return iterations_; }
  const ocs2::OptimalControlProblem& getOptimalControlProblem() const override { return ocp_; }
  std::string getBenchmarkingInfo() const override { return "[GpuSqpSolver] per-kernel times: qmgpu_enable_timing / qmgpu_kernel_ms_mean on the handle"; }
  const std::vector<ocs2::PerformanceIndex>& getIterationsLog() const override { return log_; }
  // value function / multipliers are not exposed by this solver (the reference never queries them; SURVEY.md 8(b) allows the throw)
  ocs2::ScalarFunctionQuadraticApproximation getValueFunction(ocs2::scalar_t, const ocs2::vector_t&) const override { throw std::runtime_error("[GpuSqpSolver] getValueFunction not implemented"); }
  ocs2::ScalarFunctionQuadraticApproximation getHamiltonian(ocs2::scalar_t, const ocs2::vector_t&, const ocs2::vector_t&) override { throw std::runtime_error("[GpuSqpSolver] getHamiltonian not implemented"); }
  ocs2::vector_t getStateInputEqualityConstraintLagrangian(ocs2::scalar_t, const ocs2::vector_t&) const override { throw std::runtime_error("[GpuSqpSolver] multipliers not implemented"); }
  ocs2::MultiplierCollection getIntermediateDualSolution(ocs2::scalar_t) const override { throw std::runtime_error("[GpuSqpSolver] dual solution not implemented"); }
  const ocs2::DualSolution* getDualSolution() const override { return nullptr; }
  const ocs2::ProblemMetrics& getSolutionMetrics() const override { return metrics_; }
  // solver statistics of the last run (qmgpu_mpc_args::out_stats)
  const double* lastStats() const { return stats_; }

 private:
  // ---- device / pinned staging of ONE instance.  Inputs travel as one packed block, outputs as another.
  struct In { size_t t0, x0, grid, tgtT, tgtS, evT, nev, modes, end; };     // byte offsets inside the input block
  struct Out { size_t T, X, U, mode, stats, end; };                         // byte offsets inside one output block
  void layout() {
    const size_t n1 = size_t(maxNodes_) + 1, d = sizeof(double), i = sizeof(int32_t);
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 15) & ~size_t(15); return at; };
    in_.t0 = take(d); in_.x0 = take(30 * d); in_.grid = take(n1 * d); in_.tgtT = take(kMaxKnots * d); in_.tgtS = take(kMaxKnots * QMGPU_NTARGET * d);
    in_.evT = take(QMGPU_MAX_EVENTS * d); in_.nev = take(i); in_.modes = take((QMGPU_MAX_EVENTS + 1) * i); in_.end = o;
    inBytes_ = o;
    o = 0;
    out_.T = take(n1 * d); out_.X = take(n1 * 30 * d); out_.U = take(size_t(maxNodes_) * 30 * d); out_.mode = take(n1 * i); out_.stats = take(QMGPU_NSTATS * d); out_.end = o;
    outBytes_ = o;
    warmX_ = inBytes_ + 2 * outBytes_;
    warmU_ = warmX_ + n1 * 30 * d;
    devBytes_ = warmU_ + size_t(maxNodes_) * 30 * d;
  }
  char* devOut(int set) const { return static_cast<char*>(dev_) + inBytes_ + size_t(set) * outBytes_; }
  template <class T> T* at(char* base, size_t off) const { return reinterpret_cast<T*>(base + off); }

  void runImpl(ocs2::scalar_t initTime, const ocs2::vector_t& initState, ocs2::scalar_t finalTime, const ocs2::ControllerBase* /*externalControllerPtr*/) override {
    runImpl(initTime, initState, finalTime);   // an external controller only seeds upstream's rollout-based initial guess; the warm start here is the previous solution
  }
  void runImpl(ocs2::scalar_t initTime, const ocs2::vector_t& initState, ocs2::scalar_t finalTime) override {
    if (initState.size() != QMGPU_NX) throw std::runtime_error("[GpuSqpSolver] state dimension must be 30");
    const ocs2::ModeSchedule& modeSchedule = this->getReferenceManager().getModeSchedule();
    const ocs2::TargetTrajectories& target = this->getReferenceManager().getTargetTrajectories();
    stageAndSolve(initTime, finalTime, initState, modeSchedule, target);
    haveSolution_ = true;
  }

  // 1. time grid with the mode switches as nodes (upstream timeDiscretizationWithEvents; qmgpu_time_grid_with_events emits each event
  //    once: the zero-length pre/post pair is a no-op for this robot's identity jump map);
  // 2. mode schedule, target trajectories (37-dim states, QmTargetTrajectoriesPublisher_node.cpp:76-78), x0 -> one packed H2D copy;
  // 3. warm start: the previous primal solution resampled on the new grid ON THE DEVICE (qmgpu_warm_start_batch reads the previous
  //    run's output block, which stays resident), or the QMInitializer guess on the first run / after reset();
  // 4. qmgpu_mpc_solve_batch(batch = 1); one D2H copy of T, X, U, mode, stats; one stream synchronisation;
  // 5. PrimalSolution as a feed-forward policy (useFeedbackPolicy false, task.info:90) and the PerformanceIndex from out_stats.
  void stageAndSolve(double t0, double tf, const ocs2::vector_t& x0, const ocs2::ModeSchedule& ms, const ocs2::TargetTrajectories& target) {
    const int nev = static_cast<int>(ms.eventTimes.size());
    if (nev > QMGPU_MAX_EVENTS) throw std::runtime_error("[GpuSqpSolver] mode schedule has more than QMGPU_MAX_EVENTS switches");
    if (static_cast<int>(ms.modeSequence.size()) != nev + 1) throw std::runtime_error("[GpuSqpSolver] inconsistent mode schedule");
    const int K = static_cast<int>(target.timeTrajectory.size());
    if (K < 1 || K > kMaxKnots || static_cast<int>(target.stateTrajectory.size()) != K) throw std::runtime_error("[GpuSqpSolver] target trajectories need 1.." + std::to_string(kMaxKnots) + " knots");
    char* hin = static_cast<char*>(pinned_);
    std::memset(hin, 0, inBytes_);
    int32_t N = 0;
    qmCheck(qmgpu_time_grid_with_events(t0, tf, P_.settings.dt, nev, ms.eventTimes.data(), maxNodes_, &N, at<double>(hin, in_.grid)));
    *at<double>(hin, in_.t0) = t0;
    std::copy(x0.data(), x0.data() + QMGPU_NX, at<double>(hin, in_.x0));
    for (int k = 0; k < K; ++k) {
      if (target.stateTrajectory[k].size() < QMGPU_NTARGET) throw std::runtime_error("[GpuSqpSolver] target states must have 37 entries");
      at<double>(hin, in_.tgtT)[k] = target.timeTrajectory[k];
      std::copy(target.stateTrajectory[k].data(), target.stateTrajectory[k].data() + QMGPU_NTARGET, at<double>(hin, in_.tgtS) + size_t(k) * QMGPU_NTARGET);
    }
    for (int e = 0; e < QMGPU_MAX_EVENTS; ++e) at<double>(hin, in_.evT)[e] = e < nev ? ms.eventTimes[e] : 1e300;
    *at<int32_t>(hin, in_.nev) = nev;
    for (int e = 0; e <= QMGPU_MAX_EVENTS; ++e) at<int32_t>(hin, in_.modes)[e] = e <= nev ? static_cast<int32_t>(ms.modeSequence[e]) : 15;
    char* din = static_cast<char*>(dev_);
    check(hipMemcpyAsync(din, hin, inBytes_, hipMemcpyHostToDevice, copyStream_), "hipMemcpyAsync H2D");

    const int cur = outSet_ ^ 1;              // this run writes the other output block; outSet_ still holds the previous solution
    char* dPrev = devOut(outSet_);
    char* dCur = devOut(cur);
    double* warmX = nullptr;
    double* warmU = nullptr;
    if (haveSolution_) {
      warmX = at<double>(din, warmX_); warmU = at<double>(din, warmU_);
      qmCheck(qmgpu_warm_start_batch(h_, 1, prevNodes_, at<double>(dPrev, out_.T), at<double>(dPrev, out_.X), at<double>(dPrev, out_.U), N, at<double>(din, in_.grid),
                                     at<double>(din, in_.x0), warmX, warmU));
    }
    qmgpu_mpc_args a{};
    a.batch = 1; a.num_nodes = N; a.num_target_knots = K; a.line_search = 1;
    a.t0 = at<double>(din, in_.t0); a.x0 = at<double>(din, in_.x0); a.time_grid = at<double>(din, in_.grid);
    a.target_times = at<double>(din, in_.tgtT); a.target_states = at<double>(din, in_.tgtS);
    a.sched_num_events = at<int32_t>(din, in_.nev); a.sched_event_times = at<double>(din, in_.evT); a.sched_modes = at<int32_t>(din, in_.modes);
    a.warm_x = warmX; a.warm_u = warmU;
    a.out_t = at<double>(dCur, out_.T); a.out_x = at<double>(dCur, out_.X); a.out_u = at<double>(dCur, out_.U); a.out_mode = at<int32_t>(dCur, out_.mode);
    a.out_stats = at<double>(dCur, out_.stats);
    qmCheck(qmgpu_mpc_solve_batch(h_, &a));
    char* hout = static_cast<char*>(pinned_);   // the input block has been consumed by the H2D copy queued before the kernels
    check(hipMemcpyAsync(hout, dCur, outBytes_, hipMemcpyDeviceToHost, copyStream_), "hipMemcpyAsync D2H");
    check(hipStreamSynchronize(copyStream_), "hipStreamSynchronize");

    const double* T = at<double>(hout, out_.T); const double* X = at<double>(hout, out_.X); const double* U = at<double>(hout, out_.U);
    std::copy(at<double>(hout, out_.stats), at<double>(hout, out_.stats) + QMGPU_NSTATS, stats_);
    // A failed factorisation leaves the iterate where it was (the line search takes no step: out_x / out_u = the incoming iterate).  The failed
    // block is NOT committed: the next run warm-starts from the last good solution (outSet_ / prevNodes_ unchanged), as upstream's solver keeps
    // its previous primal solution when runImpl throws.
    if (stats_[7] != 0.0) throw std::runtime_error("[GpuSqpSolver] Riccati factorisation failed (projected Hessian not positive definite)");
    outSet_ = cur; prevNodes_ = N;
    ocs2::PrimalSolution p;
    p.timeTrajectory_.assign(T, T + N + 1);
    p.stateTrajectory_.resize(size_t(N) + 1);
    p.inputTrajectory_.resize(size_t(N) + 1);
    for (int k = 0; k <= N; ++k) {
      p.stateTrajectory_[k].resize(QMGPU_NX);
      std::copy(X + size_t(k) * 30, X + size_t(k) * 30 + 30, p.stateTrajectory_[k].data());
      const int ku = std::min(k, N - 1);      // upstream repeats the last input at the final time
      p.inputTrajectory_[k].resize(QMGPU_NU);
      std::copy(U + size_t(ku) * 30, U + size_t(ku) * 30 + 30, p.inputTrajectory_[k].data());
    }
    p.modeSchedule_ = ms;
    p.controllerPtr_.reset(new ocs2::FeedforwardController(p.timeTrajectory_, p.inputTrajectory_));
    primal_.swap(p);
    // out_stats: merit0, violation0, merit1, violation1, alpha, step_type, armijo, status, iterations, convergence
    performance_ = ocs2::PerformanceIndex();
    performance_.merit = stats_[2]; performance_.cost = stats_[2];
    performance_.dynamicsViolationSSE = stats_[3] * stats_[3];   // reported as the combined constraint violation (defect + equalities)
    iterations_ = static_cast<size_t>(stats_[8] > 0.0 ? stats_[8] : 1.0);
    log_.push_back(performance_);
  }

  static void check(hipError_t e, const char* what) { if (e != hipSuccess) throw std::runtime_error(std::string("[GpuSqpSolver] ") + what + ": " + hipGetErrorString(e)); }
  static void qmCheck(int st) { if (st != QMGPU_OK) throw std::runtime_error(std::string("[GpuSqpSolver] ") + qmgpu_strerror(st) + ": " + qmgpu_last_error()); }

  qmgpu_handle h_;
  qmgpu_problem P_;
  int maxNodes_;
  ocs2::OptimalControlProblem ocp_;
  In in_{};
  Out out_{};
  size_t inBytes_ = 0, outBytes_ = 0, devBytes_ = 0, warmX_ = 0, warmU_ = 0;
  void* dev_ = nullptr;
  void* pinned_ = nullptr;
  hipStream_t copyStream_ = nullptr;
  int outSet_ = 0, prevNodes_ = 0;
  bool haveSolution_ = false;
  size_t iterations_ = 0;
  double stats_[QMGPU_NSTATS] = {0};
  ocs2::PrimalSolution primal_;
  ocs2::PerformanceIndex performance_;
  std::vector<ocs2::PerformanceIndex> log_;
  ocs2::ProblemMetrics metrics_;
};

class GpuMpc final : public ocs2::MPC_BASE {
 public:
  GpuMpc(ocs2::mpc::Settings mpcSettings, std::unique_ptr<GpuSqpSolver> solver) : MPC_BASE(std::move(mpcSettings)), solver_(std::move(solver)) {}
  ocs2::SolverBase* getSolverPtr() override { return solver_.get(); }
  const ocs2::SolverBase* getSolverPtr() const override { return solver_.get(); }

 protected:
  void calculateController(ocs2::scalar_t initTime, const ocs2::vector_t& initState, ocs2::scalar_t finalTime) override { solver_->run(initTime, initState, finalTime); }

 private:
  std::unique_ptr<GpuSqpSolver> solver_;
};

}  // namespace qm
