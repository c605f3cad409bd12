// STAND-IN (test infrastructure) for <ocs2_core/automatic_differentiation/Types.h>: upstream ad_scalar_t is
// CppAD::AD<CppAD::cg::CG<double>>.  Here it is a plain number wrapper, complete enough that the reference's explicit template
// instantiations for ad_scalar_t compile.  libref_terms.so evaluates nothing through it; libref_model.so (round 5) evaluates the bodies of the
// reference's CppAD-typed functions (constraintFunction / costVectorFunction) at plain numbers through it.
#pragma once
#include <ocs2_core/Types.h>
namespace ocs2 {
struct ad_scalar_t {
  double v = 0.0;
  ad_scalar_t() = default;
  ad_scalar_t(double x) : v(x) {}
  ad_scalar_t(int x) : v(x) {}
  ad_scalar_t(long x) : v((double)x) {}
  ad_scalar_t(unsigned long x) : v((double)x) {}
  ad_scalar_t& operator+=(const ad_scalar_t& o) { v += o.v; return *this; }
  ad_scalar_t& operator-=(const ad_scalar_t& o) { v -= o.v; return *this; }
  ad_scalar_t& operator*=(const ad_scalar_t& o) { v *= o.v; return *this; }
  ad_scalar_t& operator/=(const ad_scalar_t& o) { v /= o.v; return *this; }
  ad_scalar_t operator-() const { return ad_scalar_t(-v); }
  // hidden friend: found by argument-dependent lookup only, so that an unqualified sqrt(double) in namespace ocs2 still means std::sqrt
  friend ad_scalar_t sqrt(const ad_scalar_t& a) { return ad_scalar_t(std::sqrt(a.v)); }
};
inline ad_scalar_t operator+(ad_scalar_t a, const ad_scalar_t& b) { return a += b; }
inline ad_scalar_t operator-(ad_scalar_t a, const ad_scalar_t& b) { return a -= b; }
inline ad_scalar_t operator*(ad_scalar_t a, const ad_scalar_t& b) { return a *= b; }
inline ad_scalar_t operator/(ad_scalar_t a, const ad_scalar_t& b) { return a /= b; }
inline bool operator<(const ad_scalar_t& a, const ad_scalar_t& b) { return a.v < b.v; }
inline bool operator>(const ad_scalar_t& a, const ad_scalar_t& b) { return a.v > b.v; }
using ad_vector_t = Eigen::Matrix<ad_scalar_t, Eigen::Dynamic, 1>;   // (upstream names; round 5: the assembly files evaluated through this wrapper, oracle/ref_model_driver.cpp)
using ad_matrix_t = Eigen::Matrix<ad_scalar_t, Eigen::Dynamic, Eigen::Dynamic>;
}  // namespace ocs2
