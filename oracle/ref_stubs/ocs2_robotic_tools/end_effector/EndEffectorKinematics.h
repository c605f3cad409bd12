// STAND-IN (test infrastructure) for <ocs2_robotic_tools/end_effector/EndEffectorKinematics.h> AS THE FORK EXTENDS IT: upstream's
// abstract interface (getIds, getPosition, getVelocity, getOrientationError + linear approximations) plus the members the reference's
// EndEffectorDynamics*Constraint.cpp call on it (orientation error wrt a plane, twist).  Only the signatures those call sites imply are
// declared; the fork's header is absent from /root/reference (un-vendored lib/ocs2_ros2 submodule).  The reference's only implementation
// is CppAD-generated (unbuildable here): oracle/ref_assembly_driver.cpp implements it with kinematics handed in by the caller.
#pragma once
#include <string>
#include <vector>
#include <ocs2_core/Types.h>
namespace ocs2 {
template <typename SCALAR_T>
class EndEffectorKinematics {
 public:
  using vector3_t = Eigen::Matrix<SCALAR_T, 3, 1>;
  using vector6_t = Eigen::Matrix<SCALAR_T, 6, 1>;
  using vector_t = Eigen::Matrix<SCALAR_T, Eigen::Dynamic, 1>;
  EndEffectorKinematics() = default;
  virtual ~EndEffectorKinematics() = default;
  virtual EndEffectorKinematics* clone() const = 0;
  virtual const std::vector<std::string>& getIds() const = 0;
  virtual std::vector<vector3_t> getPosition(const vector_t& state) const = 0;
  virtual std::vector<vector3_t> getVelocity(const vector_t& state, const vector_t& input) const = 0;
  virtual std::vector<vector3_t> getOrientationErrorWrtPlane(const vector_t& state, const std::vector<vector3_t>& planeNormals) const = 0;
  virtual std::vector<vector6_t> getTwist(const vector_t& state, const vector_t& input) const = 0;
  virtual std::vector<VectorFunctionLinearApproximation> getPositionLinearApproximation(const vector_t& state) const = 0;
  virtual std::vector<VectorFunctionLinearApproximation> getVelocityLinearApproximation(const vector_t& state, const vector_t& input) const = 0;
  virtual std::vector<VectorFunctionLinearApproximation> getOrientationErrorWrtPlaneLinearApproximation(const vector_t& state, const std::vector<vector3_t>& planeNormals) const = 0;
  virtual std::vector<VectorFunctionLinearApproximation> getTwistLinearApproximation(const vector_t& state, const vector_t& input) const = 0;
 protected:
  EndEffectorKinematics(const EndEffectorKinematics&) = default;
};
}  // namespace ocs2
