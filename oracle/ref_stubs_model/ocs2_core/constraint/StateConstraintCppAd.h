// STAND-IN (test infrastructure) for <ocs2_core/constraint/StateConstraintCppAd.h>: upstream records constraintFunction on a CppAD tape in
// initialize() and evaluates / differentiates the generated code.  Here initialize() is a no-op and evaluate() calls the virtual directly with
// ad_scalar_t = a plain number wrapper: what is pinned is the reference's FUNCTION BODY, not CppAD.
#pragma once
#include <string>
#include <ocs2_core/Types.h>
#include <ocs2_core/automatic_differentiation/Types.h>
#include <ocs2_core/constraint/StateInputConstraint.h>
namespace ocs2 {
class StateConstraintCppAd {
 public:
  explicit StateConstraintCppAd(ConstraintOrder order) : order_(order) {}
  virtual ~StateConstraintCppAd() = default;
  virtual StateConstraintCppAd* clone() const = 0;
  void initialize(size_t, size_t, const std::string&, const std::string&, bool = true, bool = true) {}
  virtual bool isActive(scalar_t) const { return isActive_; }
  virtual void setActive(bool active) { isActive_ = active; }   // (the fork's constraints override these two)
  virtual bool getActive() const { return isActive_; }
  virtual size_t getNumConstraints(scalar_t time) const = 0;
  virtual vector_t getParameters(scalar_t, const PreComputation&) const { return vector_t(0); }
  vector_t evaluate(scalar_t time, const vector_t& state, const vector_t& parameters) const {
    ad_vector_t s(state.size()), p(parameters.size());
    for (Eigen::Index i = 0; i < state.size(); ++i) s(i) = ad_scalar_t(state(i));
    for (Eigen::Index i = 0; i < parameters.size(); ++i) p(i) = ad_scalar_t(parameters(i));
    const ad_vector_t v = constraintFunction(ad_scalar_t(time), s, p);
    vector_t out(v.size());
    for (Eigen::Index i = 0; i < v.size(); ++i) out(i) = v(i).v;
    return out;
  }
 protected:
  StateConstraintCppAd(const StateConstraintCppAd&) = default;
  virtual ad_vector_t constraintFunction(ad_scalar_t time, const ad_vector_t& state, const ad_vector_t& parameters) const = 0;
  bool isActive_ = true;
 private:
  ConstraintOrder order_;
};
}  // namespace ocs2
