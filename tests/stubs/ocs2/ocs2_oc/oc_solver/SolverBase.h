// STAND-IN for <ocs2_oc/oc_solver/SolverBase.h>: the call order of upstream SolverBase::run — preRun (reference manager, then the
// synchronized modules), runImpl, postRun (the modules' postSolverRun with the primal solution) — and the pure virtuals the adaptor implements.
#pragma once
#include <ocs2_oc/oc_data/PerformanceIndex.h>
#include <ocs2_oc/oc_data/PrimalSolution.h>
#include <ocs2_oc/synchronized_module/ReferenceManagerInterface.h>
namespace ocs2 {
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
 private:
  virtual void runImpl(scalar_t initTime, const vector_t& initState, scalar_t finalTime) = 0;
  std::shared_ptr<ReferenceManagerInterface> referenceManagerPtr_;
  std::vector<std::shared_ptr<SolverSynchronizedModule>> synchronizedModules_;
};
}  // namespace ocs2
