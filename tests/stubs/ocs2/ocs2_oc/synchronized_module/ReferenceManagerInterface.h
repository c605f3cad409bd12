// STAND-IN for <ocs2_oc/synchronized_module/ReferenceManagerInterface.h> and SolverSynchronizedModule.h
#pragma once
#include <ocs2_core/reference/ModeSchedule.h>
#include <ocs2_core/reference/TargetTrajectories.h>
namespace ocs2 {
class ReferenceManagerInterface {
 public:
  virtual ~ReferenceManagerInterface() = default;
  virtual void preSolverRun(scalar_t initTime, scalar_t finalTime, const vector_t& initState) = 0;
  virtual const ModeSchedule& getModeSchedule() const = 0;
  virtual const TargetTrajectories& getTargetTrajectories() const = 0;
};
struct PrimalSolution;
class SolverSynchronizedModule {
 public:
  virtual ~SolverSynchronizedModule() = default;
  virtual void preSolverRun(scalar_t initTime, scalar_t finalTime, const vector_t& initState, const ReferenceManagerInterface& referenceManager) = 0;
  virtual void postSolverRun(const PrimalSolution& primalSolution) = 0;
};
}  // namespace ocs2
