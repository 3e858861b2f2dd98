// TEST INFRASTRUCTURE ONLY -- the virtual surface of qm::WbcBase (qm_wbc/include/qm_wbc/WbcBase.h:22-34) the GPU adapter overrides.
#pragma once
#include <ros/ros.h>
#include "ocs2_core/Types.h"
#include "ocs2_pinocchio_interface/PinocchioInterface.h"
namespace qm {
using namespace ocs2;
class WbcBase {
 public:
  WbcBase(const PinocchioInterface&, CentroidalModelInfo, const PinocchioEndEffectorKinematics&, const PinocchioEndEffectorKinematics&, ros::NodeHandle&) {}
  virtual ~WbcBase() = default;
  virtual vector_t update(const vector_t& stateDesired, const vector_t& inputDesired, const vector_t& rbdStateMeasured, size_t mode, scalar_t period, scalar_t time) {
    (void)stateDesired; (void)inputDesired; (void)rbdStateMeasured; (void)mode; (void)period; (void)time; return vector_t();
  }
  virtual void loadTasksSetting(const std::string& taskFile, bool verbose) { (void)taskFile; (void)verbose; }
};
}  // namespace qm
