// STAND-IN (test infrastructure) for the FORK-ONLY <ocs2_core/penalties/penalties/PieceWisePolynomialBarrierPenalty.h> (absent from
// /root/reference and from upstream ocs2): the interface the reference's JointLimitsSoftConstraint.cpp uses (Config(mu, delta), getValue /
// getDerivative / getSecondDerivative(t, h), clone, setConfig / getConfig) around ASSUMPTION A1 of this repository — p(h) = 0 for h >= delta,
// mu ((delta - h) / delta)^3 below.  What the compiled reference file pins with it is everything BUT this formula: which offsets are
// penalised, the sign of the gradient, the diagonal Hessian, the zero offset.
#pragma once
#include <ocs2_core/Types.h>
namespace ocs2 {
class PieceWisePolynomialBarrierPenalty {
 public:
  struct Config { Config(scalar_t m = 1.0, scalar_t d = 1e-3) : mu(m), delta(d) {} scalar_t mu, delta; };
  explicit PieceWisePolynomialBarrierPenalty(Config c) : c_(c) {}
  PieceWisePolynomialBarrierPenalty* clone() const { return new PieceWisePolynomialBarrierPenalty(*this); }
  scalar_t getValue(scalar_t, scalar_t h) const { if (h >= c_.delta) return 0.0; const scalar_t t = (c_.delta - h) / c_.delta; return c_.mu * t * t * t; }
  scalar_t getDerivative(scalar_t, scalar_t h) const { if (h >= c_.delta) return 0.0; const scalar_t t = (c_.delta - h) / c_.delta; return -3.0 * c_.mu * t * t / c_.delta; }
  scalar_t getSecondDerivative(scalar_t, scalar_t h) const { if (h >= c_.delta) return 0.0; const scalar_t t = (c_.delta - h) / c_.delta; return 6.0 * c_.mu * t / (c_.delta * c_.delta); }
  void setConfig(const Config& c) { c_ = c; }
  void getConfig(Config& c) const { c = c_; }
 private:
  Config c_;
};
}  // namespace ocs2
