// STAND-IN for <ocs2_oc/oc_solver/SolverBase.h>: the call order of upstream SolverBase::run — preRun (reference manager, then the
// synchronized modules), runImpl, postRun (the modules' postSolverRun with the primal solution) — and the pure virtuals the adaptor implements.
#pragma once
#include <ocs2_core/control/FeedforwardController.h>
#include <ocs2_oc/oc_data/PerformanceIndex.h>
#include <ocs2_oc/oc_data/PrimalSolution.h>
#include <ocs2_oc/synchronized_module/ReferenceManagerInterface.h>
namespace ocs2 {
// PODs of the query interface (upstream: ocs2_core/Types.h ScalarFunctionQuadraticApproximation, ocs2_oc/oc_data/DualSolution.h,
// ocs2_oc/oc_problem/OptimalControlProblem.h, ocs2_core/model_data/Multiplier.h) — opaque here: the adaptor only has to
// name them in its overrides
struct ScalarFunctionQuadraticApproximation { scalar_t f = 0.0; vector_t dfdx, dfdu; };
struct DualSolution {};
struct OptimalControlProblem {};
struct MultiplierCollection {};
class SolverBase {
 public:
  virtual ~SolverBase() = default;
  virtual void reset() = 0;
  void run(scalar_t initTime, const vector_t& initState, scalar_t finalTime) {
    if (referenceManagerPtr_) referenceManagerPtr_->preSolverRun(initTime, finalTime, initState);
    for (auto& m : synchronizedModules_) m->preSolverRun(initTime, finalTime, initState, *referenceManagerPtr_);
    runImpl(initTime, initState, finalTime);
    if (!synchronizedModules_.empty()) { PrimalSolution s; getPrimalSolution(finalTime, &s); for (auto& m : synchronizedModules_) m->postSolverRun(s); }
  }
  void setReferenceManager(std::shared_ptr<ReferenceManagerInterface> p) { if (!p) throw std::runtime_error("[SolverBase] ReferenceManager pointer cannot be a nullptr!"); referenceManagerPtr_ = std::move(p); }
  const ReferenceManagerInterface& getReferenceManager() const { return *referenceManagerPtr_; }
  void addSynchronizedModule(std::shared_ptr<SolverSynchronizedModule> m) { synchronizedModules_.push_back(std::move(m)); }
  virtual size_t getNumIterations() const = 0;
  virtual scalar_t getFinalTime() const = 0;
  virtual void getPrimalSolution(scalar_t finalTime, PrimalSolution* primalSolutionPtr) const = 0;
  PrimalSolution primalSolution(scalar_t finalTime) const { PrimalSolution s; getPrimalSolution(finalTime, &s); return s; }
  virtual const PerformanceIndex& getPerformanceIndeces() const = 0;
  virtual const std::vector<PerformanceIndex>& getIterationsLog() const = 0;
  virtual std::string getBenchmarkingInfo() const { return {}; }
  // the rest of upstream's pure-virtual query interface (ocs2_oc/oc_solver/SolverBase.h): a solver that does not implement
  // every one of them is abstract
  virtual const OptimalControlProblem& getOptimalControlProblem() const = 0;
  virtual const DualSolution* getDualSolution() const = 0;
  virtual ScalarFunctionQuadraticApproximation getValueFunction(scalar_t time, const vector_t& state) const = 0;
  virtual ScalarFunctionQuadraticApproximation getHamiltonian(scalar_t time, const vector_t& state, const vector_t& input) = 0;
  virtual vector_t getStateInputEqualityConstraintLagrangian(scalar_t time, const vector_t& state) const = 0;
  virtual MultiplierCollection getSolutionMultipliers(scalar_t time) const = 0;
  void run(scalar_t initTime, const vector_t& initState, scalar_t finalTime, const ControllerBase* externalControllerPtr) {
    if (referenceManagerPtr_) referenceManagerPtr_->preSolverRun(initTime, finalTime, initState);
    runImpl(initTime, initState, finalTime, externalControllerPtr);
  }
  void run(scalar_t initTime, const vector_t& initState, scalar_t finalTime, const PrimalSolution& primalSolution) {
    if (referenceManagerPtr_) referenceManagerPtr_->preSolverRun(initTime, finalTime, initState);
    runImpl(initTime, initState, finalTime, primalSolution);
  }
 private:
  virtual void runImpl(scalar_t initTime, const vector_t& initState, scalar_t finalTime) = 0;
  virtual void runImpl(scalar_t initTime, const vector_t& initState, scalar_t finalTime, const ControllerBase* externalControllerPtr) = 0;
  virtual void runImpl(scalar_t initTime, const vector_t& initState, scalar_t finalTime, const PrimalSolution& primalSolution) = 0;
  std::shared_ptr<ReferenceManagerInterface> referenceManagerPtr_;
  std::vector<std::shared_ptr<SolverSynchronizedModule>> synchronizedModules_;
};
}  // namespace ocs2
