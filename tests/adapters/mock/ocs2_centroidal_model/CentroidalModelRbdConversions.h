#pragma once
#include "ocs2_pinocchio_interface/PinocchioInterface.h"
namespace ocs2 {
class CentroidalModelRbdConversions {   // upstream ocs2_centroidal_model/CentroidalModelRbdConversions.h
 public:
  CentroidalModelRbdConversions(PinocchioInterface, const CentroidalModelInfo&) {}
};
}
