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
}  // namespace ocs2::humanoid
