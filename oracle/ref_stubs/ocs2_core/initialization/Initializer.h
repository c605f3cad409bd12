// STAND-IN (test infrastructure) for <ocs2_core/initialization/Initializer.h> of upstream ocs2: the abstract interface.
#pragma once
#include <ocs2_core/Types.h>
namespace ocs2 {
class Initializer {
 public:
  Initializer() = default;
  virtual ~Initializer() = default;
  virtual Initializer* clone() const = 0;
  virtual void compute(scalar_t time, const vector_t& state, scalar_t nextTime, vector_t& input, vector_t& nextState) = 0;
 protected:
  Initializer(const Initializer&) = default;
};
}  // namespace ocs2
