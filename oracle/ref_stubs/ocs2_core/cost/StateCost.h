// STAND-IN (test infrastructure) for <ocs2_core/cost/StateCost.h> as the FORK uses it (its state costs override isActive(time)): the
// abstract interface only.
#pragma once
#include <ocs2_core/Types.h>
#include <ocs2_core/constraint/StateInputConstraint.h>   // PreComputation
#include <ocs2_core/reference/TargetTrajectories.h>
namespace ocs2 {
class StateCost {
 public:
  StateCost() = default;
  virtual ~StateCost() = default;
  virtual StateCost* clone() const = 0;
  virtual bool isActive(scalar_t) const { return true; }
  virtual scalar_t getValue(scalar_t time, const vector_t& state, const TargetTrajectories& targetTrajectories, const PreComputation& preComp) const = 0;
  virtual ScalarFunctionQuadraticApproximation getQuadraticApproximation(scalar_t time, const vector_t& state, const TargetTrajectories& targetTrajectories,
                                                                         const PreComputation& preComp) const = 0;
 protected:
  StateCost(const StateCost&) = default;
};
}  // namespace ocs2
