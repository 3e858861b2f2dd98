#pragma once
#include <stdexcept>
#include "ocs2_oc/oc_data/PrimalSolution.h"
#include "ocs2_oc/synchronized_module/ReferenceManagerInterface.h"
namespace ocs2 {
class SolverBase {   // upstream ocs2_oc/oc_solver/SolverBase.h: the virtual surface a solver has to implement
 public:
  virtual ~SolverBase() = default;
  virtual void reset() = 0;
  void run(scalar_t initTime, const vector_t& initState, scalar_t finalTime) {   // preRun -> runImpl -> postRun
    if (referenceManagerPtr_) referenceManagerPtr_->preSolverRun(initTime, finalTime, initState);
    runImpl(initTime, initState, finalTime);
  }
  void setReferenceManager(std::shared_ptr<ReferenceManagerInterface> p) { if (!p) throw std::runtime_error("[SolverBase] null reference manager"); referenceManagerPtr_ = std::move(p); }
  const ReferenceManagerInterface& getReferenceManager() const { return *referenceManagerPtr_; }
  void addSynchronizedModule(std::shared_ptr<SolverSynchronizedModule> m) { synchronizedModules_.push_back(std::move(m)); }
  virtual size_t getNumIterations() const = 0;
  virtual scalar_t getFinalTime() const = 0;
  virtual const PerformanceIndex& getPerformanceIndeces() const = 0;
  virtual const std::vector<PerformanceIndex>& getIterationsLog() const = 0;
  virtual void getPrimalSolution(scalar_t finalTime, PrimalSolution* primalSolutionPtr) const = 0;
  virtual const DualSolution* getDualSolution() const = 0;
  virtual const ProblemMetrics& getSolutionMetrics() const = 0;
  virtual ScalarFunctionQuadraticApproximation getValueFunction(scalar_t time, const vector_t& state) const = 0;
  virtual ScalarFunctionQuadraticApproximation getHamiltonian(scalar_t time, const vector_t& state, const vector_t& input) = 0;
  virtual vector_t getStateInputEqualityConstraintLagrangian(scalar_t time, const vector_t& state) const = 0;
  virtual MultiplierCollection getIntermediateDualSolution(scalar_t time) const = 0;
 private:
  virtual void runImpl(scalar_t initTime, const vector_t& initState, scalar_t finalTime) = 0;
  virtual void runImpl(scalar_t initTime, const vector_t& initState, scalar_t finalTime, const ControllerBase* externalControllerPtr) = 0;
  std::shared_ptr<ReferenceManagerInterface> referenceManagerPtr_;
  std::vector<std::shared_ptr<SolverSynchronizedModule>> synchronizedModules_;
};
}  // namespace ocs2
