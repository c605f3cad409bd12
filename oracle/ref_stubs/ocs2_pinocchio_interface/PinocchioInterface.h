// STAND-IN (test infrastructure) for <ocs2_pinocchio_interface/PinocchioInterface.h>: an empty, copyable type.  The reference's
// SwitchedModelReferenceManager keeps one by value and hands it to computeGroundHeightEstimate, whose result it then overwrites with 0
// (SwitchedModelReferenceManager.cpp:84-87); nothing compiled here reads a Pinocchio model.
#pragma once
namespace ocs2 { class PinocchioInterface {}; }
