#pragma once
#include "ocs2_core/Types.h"
namespace ocs2 {
class ControllerBase {   // upstream ocs2_core/control/ControllerBase.h
 public:
  virtual ~ControllerBase() = default;
  virtual ControllerBase* clone() const = 0;
};
class FeedforwardController final : public ControllerBase {   // upstream ocs2_core/control/FeedforwardController.h
 public:
  FeedforwardController(scalar_array_t times, vector_array_t inputs) : timeStamp_(std::move(times)), uffArray_(std::move(inputs)) {}
  FeedforwardController* clone() const override { return new FeedforwardController(*this); }
  scalar_array_t timeStamp_;
  vector_array_t uffArray_;
};
}  // namespace ocs2
