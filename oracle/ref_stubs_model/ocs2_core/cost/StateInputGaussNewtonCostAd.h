// STAND-IN (test infrastructure) for <ocs2_core/cost/StateInputGaussNewtonCostAd.h>: upstream tapes costVectorFunction (l = 1/2 |r|^2, Gauss-Newton
// Hessian J^T J); here evaluate() returns the residual vector r of the reference's function body at plain numbers.
#pragma once
#include <string>
#include <ocs2_core/Types.h>
#include <ocs2_core/automatic_differentiation/Types.h>
#include <ocs2_core/constraint/StateInputConstraint.h>
#include <ocs2_core/reference/TargetTrajectories.h>
namespace ocs2 {
class StateInputCostGaussNewtonAd {
 public:
  StateInputCostGaussNewtonAd() = default;
  virtual ~StateInputCostGaussNewtonAd() = default;
  virtual StateInputCostGaussNewtonAd* clone() const = 0;
  void initialize(size_t, size_t, size_t, const std::string&, const std::string&, bool = true, bool = true) {}
  virtual bool isActive(scalar_t) const { return true; }
  virtual vector_t getParameters(scalar_t, const TargetTrajectories&, const PreComputation&) const { return vector_t(0); }
  vector_t evaluate(scalar_t time, const vector_t& state, const vector_t& input, const vector_t& parameters) {
    auto lift = [](const vector_t& a) { ad_vector_t b(a.size()); for (Eigen::Index i = 0; i < a.size(); ++i) b(i) = ad_scalar_t(a(i)); return b; };
    const ad_vector_t v = costVectorFunction(ad_scalar_t(time), lift(state), lift(input), lift(parameters));
    vector_t out(v.size());
    for (Eigen::Index i = 0; i < v.size(); ++i) out(i) = v(i).v;
    return out;
  }
 protected:
  StateInputCostGaussNewtonAd(const StateInputCostGaussNewtonAd&) = default;
  virtual ad_vector_t costVectorFunction(ad_scalar_t time, const ad_vector_t& state, const ad_vector_t& input, const ad_vector_t& parameters) = 0;
};
}  // namespace ocs2
