// STAND-IN (test infrastructure) for <ocs2_pinocchio_interface/PinocchioInterface.h> over the mock Pinocchio (pinocchio/mock.hpp): model + data by
// value, toCppAd() = a copy with every number wrapped into ad_scalar_t (upstream: a CppAD-typed cast of the model).
#pragma once
#include <pinocchio/mock.hpp>
#include <ocs2_core/Types.h>
#include <ocs2_core/automatic_differentiation/Types.h>
namespace ocs2 {
template <class S> class PinocchioInterfaceTpl {
 public:
  using Model = pinocchio::ModelTpl<S>;
  using Data = pinocchio::DataTpl<S>;
  PinocchioInterfaceTpl() = default;
  PinocchioInterfaceTpl(const Model& m, const Data& d) : model_(m), data_(d) {}
  const Model& getModel() const { return model_; }
  Data& getData() { return data_; }
  const Data& getData() const { return data_; }
  PinocchioInterfaceTpl<ad_scalar_t> toCppAd() const { return PinocchioInterfaceTpl<ad_scalar_t>(model_.template cast<ad_scalar_t>(), data_.template cast<ad_scalar_t>()); }
 private:
  Model model_;
  mutable Data data_;
};
using PinocchioInterface = PinocchioInterfaceTpl<scalar_t>;
using PinocchioInterfaceCppAd = PinocchioInterfaceTpl<ad_scalar_t>;
}  // namespace ocs2
