// STAND-IN (test infrastructure) for <ocs2_core/cost/QuadraticStateInputCost.h> of upstream leggedrobotics/ocs2, restated from the published
// class: l = 1/2 dx' Q dx + 1/2 du' R du (+ du' P dx) with (dx, du) = getStateInputDeviation(t, x, u, targets), which derived classes
// override (the reference's StateInputQuadraticCost does); gradient Q dx (+ P' du), R du (+ P dx); Hessian blocks Q, R, P.
#pragma once
#include <ocs2_core/Types.h>
#include <ocs2_core/constraint/StateInputConstraint.h>   // PreComputation
#include <ocs2_core/reference/TargetTrajectories.h>
namespace ocs2 {
class QuadraticStateInputCost {
 public:
  QuadraticStateInputCost(matrix_t Q, matrix_t R, matrix_t P = matrix_t()) : Q_(std::move(Q)), R_(std::move(R)), P_(std::move(P)) {}
  virtual ~QuadraticStateInputCost() = default;
  virtual QuadraticStateInputCost* clone() const = 0;
  scalar_t getValue(scalar_t time, const vector_t& state, const vector_t& input, const TargetTrajectories& targetTrajectories, const PreComputation&) const {
    const auto d = getStateInputDeviation(time, state, input, targetTrajectories);
    scalar_t v = 0.5 * d.first.dot(vector_t(Q_ * d.first)) + 0.5 * d.second.dot(vector_t(R_ * d.second));
    if (P_.size() > 0) v += d.second.dot(vector_t(P_ * d.first));
    return v;
  }
  ScalarFunctionQuadraticApproximation getQuadraticApproximation(scalar_t time, const vector_t& state, const vector_t& input, const TargetTrajectories& targetTrajectories,
                                                                 const PreComputation& preComp) const {
    const auto d = getStateInputDeviation(time, state, input, targetTrajectories);
    ScalarFunctionQuadraticApproximation L;
    L.f = getValue(time, state, input, targetTrajectories, preComp);
    L.dfdx = vector_t(Q_ * d.first); L.dfdu = vector_t(R_ * d.second);
    L.dfdxx = Q_; L.dfduu = R_;
    L.dfdux = P_.size() > 0 ? P_ : matrix_t(matrix_t::Zero(R_.rows(), Q_.rows()));
    if (P_.size() > 0) { L.dfdx += vector_t(P_.transpose() * d.second); L.dfdu += vector_t(P_ * d.first); }
    return L;
  }
 protected:
  QuadraticStateInputCost(const QuadraticStateInputCost&) = default;
  virtual std::pair<vector_t, vector_t> getStateInputDeviation(scalar_t time, const vector_t& state, const vector_t& input, const TargetTrajectories& targetTrajectories) const {
    return {vector_t(state - targetTrajectories.getDesiredState(time)), input};
  }
  matrix_t Q_, R_, P_;
};
}  // namespace ocs2
