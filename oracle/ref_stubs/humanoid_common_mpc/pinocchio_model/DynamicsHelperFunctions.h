// SHADOW (test infrastructure) of the REFERENCE header humanoid_common_mpc/pinocchio_model/DynamicsHelperFunctions.h, which includes
// Pinocchio (absent here).  The one function the files compiled by oracle/Makefile take from it is computeGroundHeightEstimate, called by
// SwitchedModelReferenceManager::adaptToCurrentGroundHeight — which overwrites its result with 0.0 on the next line
// (SwitchedModelReferenceManager.cpp:84-87).  Nothing else of that header is declared here; the model-evaluation functions it holds stay
// UNPINNED.
#pragma once
#include <ocs2_pinocchio_interface/PinocchioInterface.h>
#include "humanoid_common_mpc/common/MpcRobotModelBase.h"
namespace ocs2::humanoid {
inline scalar_t computeGroundHeightEstimate(PinocchioInterface&, const MpcRobotModelBase<scalar_t>&, const vector_t&, size_t) { return 0.0; }
scalar_t& ref_stub_total_mass();   // defined by the driver
inline size_t numberOfLegsInContacts(const contact_flag_t& contactFlags) {
  size_t n = 0;
  for (auto flag : contactFlags) if (flag) ++n;
  return n;
}
inline vector_t weightCompensatingInput(const PinocchioInterface&, const contact_flag_t& contactFlags, const MpcRobotModelBase<scalar_t>& mpcRobotModel) {
  const scalar_t totalGravitationalForce = ref_stub_total_mass() * 9.81;
  const auto numStanceLegs = numberOfLegsInContacts(contactFlags);
  vector_t input = vector_t::Zero(mpcRobotModel.getInputDim());
  if (numStanceLegs > 0) {
    const vector3_t forceInInertialFrame(0.0, 0.0, totalGravitationalForce / numStanceLegs);
    for (size_t i = 0; i < contactFlags.size(); i++) if (contactFlags[i]) mpcRobotModel.setContactForce(input, forceInInertialFrame, i);
  }
  return input;
}
}  // namespace ocs2::humanoid
