// TEST INFRASTRUCTURE ONLY -- the virtual surface of upstream ocs2_oc/oc_solver/SolverBase.h that a solver has to implement.
// Mirrors leggedrobotics/ocs2 `main` as the reference builds against it (un-pinned, README.md:36): the state AFTER the 2022 solver-interface
// refactor that added OptimalControlProblem to the solver surface -- the reference itself overrides `getOptimalControlProblem()` on its robot
// interface (qm_interface/include/qm_interface/QMInterface.h:37) and hands the problem to SqpMpc (qm_controllers/src/QMController.cpp:288-289).
// The commit hash cannot be verified in this container (no network, OCS2 not vendored); EVERY pure virtual of that header is declared below, so a
// solver that compiles here is not abstract there:
//   reset, getNumIterations, getFinalTime, getOptimalControlProblem, getPerformanceIndeces, getIterationsLog, getPrimalSolution, getDualSolution,
//   getSolutionMetrics, getValueFunction, getHamiltonian, getStateInputEqualityConstraintLagrangian, getIntermediateDualSolution, runImpl (x2).
// Non-pure members that MPC_BASE / MPC_MRT_Interface call on a solver are provided with upstream's behaviour: run (both overloads), primalSolution,
// set / addSynchronizedModule(s), set / getReferenceManager (const and non-const), getBenchmarkingInfo.
#pragma once
#include <stdexcept>
#include <string>
#include "ocs2_oc/oc_data/PrimalSolution.h"
#include "ocs2_oc/oc_problem/OptimalControlProblem.h"
#include "ocs2_oc/synchronized_module/ReferenceManagerInterface.h"
namespace ocs2 {
class SolverBase {
 public:
  virtual ~SolverBase() = default;
  virtual void reset() = 0;
  void run(scalar_t initTime, const vector_t& initState, scalar_t finalTime) {   // preRun -> runImpl -> postRun
    if (referenceManagerPtr_) referenceManagerPtr_->preSolverRun(initTime, finalTime, initState);
    runImpl(initTime, initState, finalTime);
  }
  void run(scalar_t initTime, const vector_t& initState, scalar_t finalTime, const ControllerBase* externalControllerPtr) {
    if (referenceManagerPtr_) referenceManagerPtr_->preSolverRun(initTime, finalTime, initState);
    runImpl(initTime, initState, finalTime, externalControllerPtr);
  }
  void setReferenceManager(std::shared_ptr<ReferenceManagerInterface> p) { if (!p) throw std::runtime_error("[SolverBase] null reference manager"); referenceManagerPtr_ = std::move(p); }
  const ReferenceManagerInterface& getReferenceManager() const { return *referenceManagerPtr_; }
  ReferenceManagerInterface& getReferenceManager() { return *referenceManagerPtr_; }
  void setSynchronizedModules(const std::vector<std::shared_ptr<SolverSynchronizedModule>>& m) { synchronizedModules_ = m; }
  void addSynchronizedModule(std::shared_ptr<SolverSynchronizedModule> m) { synchronizedModules_.push_back(std::move(m)); }
  virtual size_t getNumIterations() const = 0;
  virtual scalar_t getFinalTime() const = 0;
  virtual const OptimalControlProblem& getOptimalControlProblem() const = 0;
  virtual const PerformanceIndex& getPerformanceIndeces() const = 0;
  virtual const std::vector<PerformanceIndex>& getIterationsLog() const = 0;
  virtual void getPrimalSolution(scalar_t finalTime, PrimalSolution* primalSolutionPtr) const = 0;
  PrimalSolution primalSolution(scalar_t finalTime) const { PrimalSolution p; getPrimalSolution(finalTime, &p); return p; }
  virtual const DualSolution* getDualSolution() const = 0;
  virtual const ProblemMetrics& getSolutionMetrics() const = 0;
  virtual ScalarFunctionQuadraticApproximation getValueFunction(scalar_t time, const vector_t& state) const = 0;
  virtual ScalarFunctionQuadraticApproximation getHamiltonian(scalar_t time, const vector_t& state, const vector_t& input) = 0;
  virtual vector_t getStateInputEqualityConstraintLagrangian(scalar_t time, const vector_t& state) const = 0;
  virtual MultiplierCollection getIntermediateDualSolution(scalar_t time) const = 0;
  virtual std::string getBenchmarkingInfo() const { return {}; }
 private:
  virtual void runImpl(scalar_t initTime, const vector_t& initState, scalar_t finalTime) = 0;
  virtual void runImpl(scalar_t initTime, const vector_t& initState, scalar_t finalTime, const ControllerBase* externalControllerPtr) = 0;
  std::shared_ptr<ReferenceManagerInterface> referenceManagerPtr_;
  std::vector<std::shared_ptr<SolverSynchronizedModule>> synchronizedModules_;
};
}  // namespace ocs2
