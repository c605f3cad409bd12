// STAND-IN for <ocs2_core/initialization/Initializer.h>
#pragma once
#include <ocs2_core/Types.h>
namespace ocs2 {
class Initializer {
 public:
  virtual ~Initializer() = default;
  virtual Initializer* clone() const = 0;
  virtual void compute(scalar_t time, const vector_t& state, scalar_t nextTime, vector_t& input, vector_t& nextState) = 0;
};
}  // namespace ocs2
