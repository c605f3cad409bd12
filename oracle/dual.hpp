// TEST INFRASTRUCTURE — part of the CPU oracle (see oracle/oracle.cpp header).
// Forward-mode dual numbers with a fixed number of tangent directions.  The
// reference obtains the same Jacobians from CppAD tapes
// (humanoid_nmpc/humanoid_wb_mpc/src/end_effector/PinocchioEndEffectorDynamicsCppAd.cpp:96-177,
//  humanoid_nmpc/humanoid_wb_mpc/src/dynamics/WBAccelDynamicsAD.cpp:40-58);
// CppAD is not available here, so the oracle differentiates the same scalar
// code with operator overloading instead.
#pragma once
#include <cmath>

template <int N>
struct Dual {
  double v;
  double d[N];
  Dual() : v(0.0) {
    for (int i = 0; i < N; ++i) d[i] = 0.0;
  }
  Dual(double c) : v(c) {  // NOLINT: implicit on purpose
    for (int i = 0; i < N; ++i) d[i] = 0.0;
  }
  static Dual seed(double value, int dir) {
    Dual r(value);
    r.d[dir] = 1.0;
    return r;
  }
};

template <int N>
inline Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r;
  r.v = a.v + b.v;
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
template <int N>
inline Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r;
  r.v = a.v - b.v;
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
template <int N>
inline Dual<N> operator-(const Dual<N>& a) {
  Dual<N> r;
  r.v = -a.v;
  for (int i = 0; i < N; ++i) r.d[i] = -a.d[i];
  return r;
}
template <int N>
inline Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r;
  r.v = a.v * b.v;
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
template <int N>
inline Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r;
  const double inv = 1.0 / b.v;
  r.v = a.v * inv;
  for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
template <int N>
inline Dual<N> operator+(const Dual<N>& a, double b) { Dual<N> r = a; r.v += b; return r; }
template <int N>
inline Dual<N> operator+(double b, const Dual<N>& a) { Dual<N> r = a; r.v += b; return r; }
template <int N>
inline Dual<N> operator-(const Dual<N>& a, double b) { Dual<N> r = a; r.v -= b; return r; }
template <int N>
inline Dual<N> operator-(double b, const Dual<N>& a) { Dual<N> r = -a; r.v += b; return r; }
template <int N>
inline Dual<N> operator*(const Dual<N>& a, double b) {
  Dual<N> r;
  r.v = a.v * b;
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b;
  return r;
}
template <int N>
inline Dual<N> operator*(double b, const Dual<N>& a) { return a * b; }
template <int N>
inline Dual<N> operator/(const Dual<N>& a, double b) { return a * (1.0 / b); }
template <int N>
inline Dual<N> operator/(double a, const Dual<N>& b) { return Dual<N>(a) / b; }
template <int N>
inline Dual<N>& operator+=(Dual<N>& a, const Dual<N>& b) { a = a + b; return a; }
template <int N>
inline Dual<N>& operator-=(Dual<N>& a, const Dual<N>& b) { a = a - b; return a; }

template <int N>
inline Dual<N> sin(const Dual<N>& a) {
  Dual<N> r;
  r.v = std::sin(a.v);
  const double c = std::cos(a.v);
  for (int i = 0; i < N; ++i) r.d[i] = c * a.d[i];
  return r;
}
template <int N>
inline Dual<N> cos(const Dual<N>& a) {
  Dual<N> r;
  r.v = std::cos(a.v);
  const double s = -std::sin(a.v);
  for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i];
  return r;
}
template <int N>
inline Dual<N> sqrt(const Dual<N>& a) {
  Dual<N> r;
  r.v = std::sqrt(a.v);
  const double s = 0.5 / r.v;
  for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i];
  return r;
}

inline double value_of(double a) { return a; }
template <int N>
inline double value_of(const Dual<N>& a) { return a.v; }
