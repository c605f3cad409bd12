// STAND-IN (test infrastructure) for <ocs2_pinocchio_interface/PinocchioStateInputMapping.h>: only included
#pragma once
