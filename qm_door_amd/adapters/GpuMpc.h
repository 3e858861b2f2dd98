// GpuMpc.h -- ocs2::MPC_BASE / ocs2::SolverBase front end over the C ABI (include/qmgpu.h).  Header only; compiled inside the
// reference's catkin workspace next to OCS2 (not available in this repository's build container: NOT compiled or tested here).
//
// Seam: QMController::setupMpc (qm_controllers/src/QMController.cpp:287-307) stores `mpc_` as std::shared_ptr<ocs2::MPC_BASE>
// and afterwards only uses getSolverPtr()->addSynchronizedModule / setReferenceManager (QMController.cpp:303-304) and
// MPC_MRT_Interface(*mpc_) (QMController.cpp:311).  A solver therefore has to implement SolverBase::runImpl and hand back a
// PrimalSolution; everything else (reference manager, gait receiver, MRT buffering) stays upstream code.
#pragma once
#include <hip/hip_runtime_api.h>
#include <ocs2_mpc/MPC_BASE.h>
#include <ocs2_oc/oc_solver/SolverBase.h>
#include <ocs2_legged_robot/reference_manager/SwitchedModelReferenceManager.h>

#include <memory>
#include <stdexcept>
#include <vector>

#include "qmgpu.h"

namespace qm {

// One SQP iteration per run, batch = 1, on the device behind `handle`.
class GpuSqpSolver final : public ocs2::SolverBase {
 public:
  GpuSqpSolver(qmgpu_handle handle, const qmgpu_problem& problem, std::shared_ptr<ocs2::legged_robot::SwitchedModelReferenceManager> refManager, int maxNodes)
      : h_(handle), P_(problem), ref_(std::move(refManager)), maxNodes_(maxNodes) {
    // device staging: inputs and outputs of one instance
    const size_t n1 = maxNodes_ + 1;
    bytes_ = sizeof(double) * (1 + 30 + n1 + kMaxKnots * 38 + QMGPU_MAX_EVENTS + n1 * 30 * 2 + maxNodes_ * 30 * 2 + n1 + QMGPU_NSTATS) + sizeof(int32_t) * (QMGPU_MAX_EVENTS + 2 + n1);
    if (hipMalloc(&dev_, bytes_) != hipSuccess) throw std::runtime_error("[GpuSqpSolver] hipMalloc failed");
  }
  ~GpuSqpSolver() override { hipFree(dev_); }

  void reset() override { haveSolution_ = false; }
  ocs2::scalar_t getFinalTime() const override { return primal_.timeTrajectory_.empty() ? 0.0 : primal_.timeTrajectory_.back(); }
  void getPrimalSolution(ocs2::scalar_t finalTime, ocs2::PrimalSolution* out) const override { *out = primal_; }
  const ocs2::PerformanceIndex& getPerformanceIndeces() const override { return performance_; }
  size_t getNumIterations() const override { return 1; }
  const std::vector<ocs2::PerformanceIndex>& getIterationsLog() const override { return log_; }
  // value function / multipliers are not exposed by this solver (the reference never queries them)
  ocs2::ScalarFunctionQuadraticApproximation getValueFunction(ocs2::scalar_t, const ocs2::vector_t&) const override { throw std::runtime_error("[GpuSqpSolver] getValueFunction not implemented"); }
  ocs2::ScalarFunctionQuadraticApproximation getHamiltonian(ocs2::scalar_t, const ocs2::vector_t&, const ocs2::vector_t&) override { throw std::runtime_error("[GpuSqpSolver] getHamiltonian not implemented"); }
  ocs2::vector_t getStateInputEqualityConstraintLagrangian(ocs2::scalar_t, const ocs2::vector_t&) const override { throw std::runtime_error("[GpuSqpSolver] multipliers not implemented"); }
  ocs2::MultiplierCollection getIntermediateDualSolution(ocs2::scalar_t) const override { throw std::runtime_error("[GpuSqpSolver] dual solution not implemented"); }
  const ocs2::DualSolution* getDualSolution() const override { return nullptr; }
  const ocs2::ProblemMetrics& getSolutionMetrics() const override { return metrics_; }

 private:
  static constexpr int kMaxKnots = 16;

  void runImpl(ocs2::scalar_t initTime, const ocs2::vector_t& initState, ocs2::scalar_t finalTime) override {
    // 1. time grid with events: upstream timeDiscretizationWithEvents(initTime, finalTime, dt, eventTimes); the device accepts an
    //    arbitrary grid through qmgpu_mpc_args::time_grid.  qmgpu_time_grid_with_events emits each event time once (upstream's
    //    zero-length pre/post pair is a no-op for this robot's identity jump map).
    const auto& modeSchedule = ref_->getModeSchedule();
    std::vector<double> grid = makeGrid(initTime, finalTime, P_.settings.dt, modeSchedule.eventTimes);
    const int N = static_cast<int>(grid.size()) - 1;
    if (N > maxNodes_) throw std::runtime_error("[GpuSqpSolver] horizon exceeds the capacity given to qmgpu_create");
    // 2. target trajectories (37-dim states, QmTargetTrajectoriesPublisher_node.cpp:76-78) and mode schedule -> device
    // 3. warm start: previous primal solution resampled on the new grid (upstream SqpSolver) -- qmgpu_warm_start_batch does it on the device
    //    from the previous call's out_t / out_x / out_u, which stay resident between runs
    // 4. qmgpu_mpc_solve_batch(batch = 1), qmgpu_synchronize, copy X / U back into primal_ (useFeedbackPolicy false: task.info:90)
    // The staging code is mechanical (hipMemcpy of the arrays named in qmgpu_mpc_args) and is spelled out in INTEGRATION.md.
    stageAndSolve(grid, initState, modeSchedule);
    haveSolution_ = true;
  }
  static std::vector<double> makeGrid(double t0, double tf, double dt, const std::vector<double>& events) {
    std::vector<double> g(1025);
    int32_t n = 0;
    if (qmgpu_time_grid_with_events(t0, tf, dt, static_cast<int32_t>(events.size()), events.data(), 1024, &n, g.data()) != QMGPU_OK) throw std::runtime_error(qmgpu_last_error());
    g.resize(n + 1);
    return g;
  }
  void stageAndSolve(const std::vector<double>& grid, const ocs2::vector_t& x0, const ocs2::ModeSchedule& ms);

  qmgpu_handle h_;
  qmgpu_problem P_;
  std::shared_ptr<ocs2::legged_robot::SwitchedModelReferenceManager> ref_;
  int maxNodes_;
  void* dev_ = nullptr;
  size_t bytes_ = 0;
  bool haveSolution_ = false;
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
