// STAND-IN for <ocs2_oc/synchronized_module/SolverSynchronizedModule.h>: the interface itself is stated with the reference manager's in
// tests/stubs/ocs2/ocs2_oc/synchronized_module/ReferenceManagerInterface.h (upstream: the two hooks SolverBase calls around a solve)
#pragma once
#include <ocs2_oc/synchronized_module/ReferenceManagerInterface.h>
