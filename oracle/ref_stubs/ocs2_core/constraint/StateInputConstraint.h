// STAND-IN (test infrastructure) for <ocs2_core/constraint/StateInputConstraint.h> as the FORK uses it (the reference's constraints
// override isActive / setActive / getActive and read a protected isActive_): the abstract interface only.
#pragma once
#include <ocs2_core/Types.h>
namespace ocs2 {
class PreComputation { public: virtual ~PreComputation() = default; };
enum class ConstraintOrder { Linear, Quadratic };
class StateInputConstraint {
 public:
  explicit StateInputConstraint(ConstraintOrder order) : order_(order) {}
  virtual ~StateInputConstraint() = default;
  virtual StateInputConstraint* clone() const = 0;
  ConstraintOrder getOrder() const { return order_; }
  virtual bool isActive(scalar_t) const { return isActive_; }
  virtual void setActive(bool active) { isActive_ = active; }
  virtual bool getActive() const { return isActive_; }
  virtual size_t getNumConstraints(scalar_t time) const = 0;
  virtual vector_t getValue(scalar_t time, const vector_t& state, const vector_t& input, const PreComputation& preComp) const = 0;
  virtual VectorFunctionLinearApproximation getLinearApproximation(scalar_t, const vector_t&, const vector_t&, const PreComputation&) const {
    throw std::runtime_error("[StateInputConstraint] Linear approximation not implemented");
  }
  virtual VectorFunctionQuadraticApproximation getQuadraticApproximation(scalar_t, const vector_t&, const vector_t&, const PreComputation&) const {
    throw std::runtime_error("[StateInputConstraint] Quadratic approximation not implemented");
  }
 protected:
  StateInputConstraint(const StateInputConstraint& rhs) = default;
  bool isActive_ = true;
 private:
  ConstraintOrder order_;
};
}  // namespace ocs2
