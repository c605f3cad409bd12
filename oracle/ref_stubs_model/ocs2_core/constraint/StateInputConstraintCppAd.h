// STAND-IN (test infrastructure) for <ocs2_core/constraint/StateInputConstraintCppAd.h>: see StateConstraintCppAd.h
#pragma once
#include <string>
#include <ocs2_core/Types.h>
#include <ocs2_core/automatic_differentiation/Types.h>
#include <ocs2_core/constraint/StateInputConstraint.h>
namespace ocs2 {
class StateInputConstraintCppAd {
 public:
  explicit StateInputConstraintCppAd(ConstraintOrder order) : order_(order) {}
  virtual ~StateInputConstraintCppAd() = default;
  virtual StateInputConstraintCppAd* clone() const = 0;
  void initialize(size_t, size_t, size_t, const std::string&, const std::string&, bool = true, bool = true) {}
  virtual bool isActive(scalar_t) const { return isActive_; }
  virtual void setActive(bool active) { isActive_ = active; }   // (the fork's constraints override these two)
  virtual bool getActive() const { return isActive_; }
  virtual size_t getNumConstraints(scalar_t time) const = 0;
  virtual vector_t getParameters(scalar_t, const PreComputation&) const { return vector_t(0); }
  vector_t evaluate(scalar_t time, const vector_t& state, const vector_t& input, const vector_t& parameters) const {
    auto lift = [](const vector_t& a) { ad_vector_t b(a.size()); for (Eigen::Index i = 0; i < a.size(); ++i) b(i) = ad_scalar_t(a(i)); return b; };
    const ad_vector_t v = constraintFunction(ad_scalar_t(time), lift(state), lift(input), lift(parameters));
    vector_t out(v.size());
    for (Eigen::Index i = 0; i < v.size(); ++i) out(i) = v(i).v;
    return out;
  }
 protected:
  StateInputConstraintCppAd(const StateInputConstraintCppAd&) = default;
  virtual ad_vector_t constraintFunction(ad_scalar_t time, const ad_vector_t& state, const ad_vector_t& input, const ad_vector_t& parameters) const = 0;
  bool isActive_ = true;
 private:
  ConstraintOrder order_;
};
}  // namespace ocs2
