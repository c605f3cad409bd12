// STAND-IN for <ocs2_core/control/ControllerBase.h> + <ocs2_core/control/FeedforwardController.h>
#pragma once
#include <ocs2_core/Types.h>
namespace ocs2 {
class ControllerBase {
 public:
  virtual ~ControllerBase() = default;
  virtual vector_t computeInput(scalar_t t, const vector_t& x) = 0;
};
class FeedforwardController final : public ControllerBase {
 public:
  FeedforwardController(scalar_array_t times, vector_array_t inputs) : timeStamp_(std::move(times)), uffArray_(std::move(inputs)) {}
  vector_t computeInput(scalar_t t, const vector_t&) override {   // upstream: LinearInterpolation on (timeStamp_, uffArray_)
    if (t <= timeStamp_.front()) return uffArray_.front();
    if (t >= timeStamp_.back()) return uffArray_.back();
    size_t i = 1;
    while (timeStamp_[i] < t) ++i;
    const scalar_t h = timeStamp_[i] - timeStamp_[i - 1], a = h > 0.0 ? (t - timeStamp_[i - 1]) / h : 1.0;
    vector_t u(uffArray_[i].size());
    for (size_t k = 0; k < u.size(); ++k) u[k] = (1.0 - a) * uffArray_[i - 1][k] + a * uffArray_[i][k];
    return u;
  }
  scalar_array_t timeStamp_;
  vector_array_t uffArray_;
};
}  // namespace ocs2
