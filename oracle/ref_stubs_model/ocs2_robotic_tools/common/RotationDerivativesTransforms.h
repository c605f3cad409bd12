// STAND-IN (test infrastructure) for <ocs2_robotic_tools/common/RotationDerivativesTransforms.h>: only included by the files compiled here
#pragma once
#include <ocs2_core/Types.h>
