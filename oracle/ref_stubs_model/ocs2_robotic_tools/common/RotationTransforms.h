// STAND-IN (test infrastructure) for the fork's <ocs2_robotic_tools/common/RotationTransforms.h> as libref_model.so needs it:
//   * matrixToQuaternion — upstream ocs2 (trace formula; here with the usual branches on plain numbers);
//   * quaternionDistance — upstream;
//   * quaternionDistanceToPlane — FORK-ONLY, source absent: the oracle's ASSUMPTION A2 restated on the quaternion (shortest arc from the frame's z axis
//     to the plane normal: e = (n x a) / sqrt(2 (1 + a.n)), a = R(q) e_z).  EndEffectorDynamicsFootCost.cpp calls it; what that file pins is therefore
//     everything AROUND this term (order of the 18 errors, references, weights, impact scaler), not the term.
#pragma once
#include <cmath>
#include <ocs2_core/Types.h>
namespace ocs2 {
template <typename SCALAR_T>
Eigen::Quaternion<SCALAR_T> matrixToQuaternion(const Eigen::Matrix<SCALAR_T, 3, 3>& R) {
  using std::sqrt;
  const SCALAR_T tr = R(0, 0) + R(1, 1) + R(2, 2);
  SCALAR_T w, x, y, z;
  if (tr > SCALAR_T(0.0)) {
    const SCALAR_T s = sqrt(tr + SCALAR_T(1.0)) * SCALAR_T(2.0);
    w = SCALAR_T(0.25) * s; x = (R(2, 1) - R(1, 2)) / s; y = (R(0, 2) - R(2, 0)) / s; z = (R(1, 0) - R(0, 1)) / s;
  } else if (R(0, 0) > R(1, 1) && R(0, 0) > R(2, 2)) {
    const SCALAR_T s = sqrt(SCALAR_T(1.0) + R(0, 0) - R(1, 1) - R(2, 2)) * SCALAR_T(2.0);
    w = (R(2, 1) - R(1, 2)) / s; x = SCALAR_T(0.25) * s; y = (R(0, 1) + R(1, 0)) / s; z = (R(0, 2) + R(2, 0)) / s;
  } else if (R(1, 1) > R(2, 2)) {
    const SCALAR_T s = sqrt(SCALAR_T(1.0) + R(1, 1) - R(0, 0) - R(2, 2)) * SCALAR_T(2.0);
    w = (R(0, 2) - R(2, 0)) / s; x = (R(0, 1) + R(1, 0)) / s; y = SCALAR_T(0.25) * s; z = (R(1, 2) + R(2, 1)) / s;
  } else {
    const SCALAR_T s = sqrt(SCALAR_T(1.0) + R(2, 2) - R(0, 0) - R(1, 1)) * SCALAR_T(2.0);
    w = (R(1, 0) - R(0, 1)) / s; x = (R(0, 2) + R(2, 0)) / s; y = (R(1, 2) + R(2, 1)) / s; z = SCALAR_T(0.25) * s;
  }
  return Eigen::Quaternion<SCALAR_T>(w, x, y, z);
}
template <typename SCALAR_T>
Eigen::Matrix<SCALAR_T, 3, 1> quaternionDistance(const Eigen::Quaternion<SCALAR_T>& q, const Eigen::Quaternion<SCALAR_T>& qRef) {
  Eigen::Matrix<SCALAR_T, 3, 1> e;
  const Eigen::Dyn<SCALAR_T> v = q.vec(), vr = qRef.vec();
  const Eigen::Dyn<SCALAR_T> c = v.cross(vr);
  for (int k = 0; k < 3; ++k) e(k) = q.w() * vr(k) - qRef.w() * v(k) + c(k);
  return e;
}
template <typename SCALAR_T>
Eigen::Matrix<SCALAR_T, 3, 1> quaternionDistanceToPlane(const Eigen::Quaternion<SCALAR_T>& q, const Eigen::Matrix<SCALAR_T, 3, 1>& n) {
  using std::sqrt;
  const SCALAR_T two(2.0), one(1.0);
  Eigen::Matrix<SCALAR_T, 3, 1> a, e;
  a(0) = two * (q.x() * q.z() + q.w() * q.y());
  a(1) = two * (q.y() * q.z() - q.w() * q.x());
  a(2) = one - two * (q.x() * q.x() + q.y() * q.y());
  const SCALAR_T s = sqrt(two * (one + a(0) * n(0) + a(1) * n(1) + a(2) * n(2)));
  e(0) = (n(1) * a(2) - n(2) * a(1)) / s;
  e(1) = (n(2) * a(0) - n(0) * a(2)) / s;
  e(2) = (n(0) * a(1) - n(1) * a(0)) / s;
  return e;
}
}  // namespace ocs2
