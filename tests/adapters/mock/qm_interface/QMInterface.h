// TEST INFRASTRUCTURE ONLY -- accessors of qm::QMInterface (qm_interface/include/qm_interface/QMInterface.h:37-54) the adapters call.
#pragma once
#include "ocs2_legged_robot/reference_manager/SwitchedModelReferenceManager.h"
#include "ocs2_mpc/MPC_BASE.h"
#include "ocs2_pinocchio_interface/PinocchioInterface.h"
namespace qm {
using namespace ocs2;
using namespace legged_robot;
class QMInterface {
 public:
  const OptimalControlProblem& getOptimalControlProblem() const { return problem_; }   // QMInterface.h:37
  const mpc::Settings& mpcSettings() const { return mpcSettings_; }
  PinocchioInterface& getPinocchioInterface() { return pinocchio_; }
  const CentroidalModelInfo& getCentroidalModelInfo() const { return info_; }
  std::shared_ptr<SwitchedModelReferenceManager> getSwitchedModelReferenceManagerPtr() const { return referenceManagerPtr_; }
  std::shared_ptr<ReferenceManagerInterface> getReferenceManagerPtr() const { return referenceManagerPtr_; }
  OptimalControlProblem problem_{42};
  mpc::Settings mpcSettings_;
  PinocchioInterface pinocchio_;
  CentroidalModelInfo info_;
  std::shared_ptr<SwitchedModelReferenceManager> referenceManagerPtr_ = std::make_shared<SwitchedModelReferenceManager>();
};
}  // namespace qm
