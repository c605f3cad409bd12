// STAND-IN (test infrastructure) for the fork's <ocs2_robotic_tools/common/RotationTransforms.h>: only quaternionDistance is named by
// the file compiled here (EndEffectorDynamicsCostHelpers.cpp instantiates computeTaskSpaceErrors, which the driver never calls).  The
// body below is the oracle's ASSUMPTION A2, not the fork's source — nothing is pinned through it.
#pragma once
#include <ocs2_core/Types.h>
namespace ocs2 {
template <typename SCALAR_T>
Eigen::Matrix<SCALAR_T, 3, 1> quaternionDistance(const Eigen::Quaternion<SCALAR_T>& q, const Eigen::Quaternion<SCALAR_T>& qRef) {
  Eigen::Matrix<SCALAR_T, 3, 1> e;
  const Eigen::Dyn<SCALAR_T> v = q.vec(), vr = qRef.vec();
  const Eigen::Dyn<SCALAR_T> c = v.cross(vr);
  for (int k = 0; k < 3; ++k) e(k) = q.w() * vr(k) - qRef.w() * v(k) + c(k);
  return e;
}
}  // namespace ocs2
