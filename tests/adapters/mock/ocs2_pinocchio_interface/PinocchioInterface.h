#pragma once
namespace ocs2 {
class PinocchioInterface {};
class PinocchioEndEffectorKinematics {};
struct CentroidalModelInfo {};
}
