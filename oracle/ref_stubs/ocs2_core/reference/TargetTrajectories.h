// STAND-IN (test infrastructure) for <ocs2_core/reference/TargetTrajectories.h>: the struct lives in the ReferenceManager stand-in of this
// directory tree (upstream keeps them in separate headers; the compiled reference files only need the aggregate + getDesiredState).
#pragma once
#include <ocs2_oc/synchronized_module/ReferenceManager.h>
