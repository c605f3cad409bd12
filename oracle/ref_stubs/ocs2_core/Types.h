// STAND-IN (test infrastructure) for <ocs2_core/Types.h> of upstream leggedrobotics/ocs2: the scalar / vector / matrix aliases and the
// function-approximation PODs the reference files compiled by oracle/Makefile (_ref targets) use.  scalar_t is double upstream too;
// vector_t / matrix_t are Eigen::VectorXd / MatrixXd upstream, the Eigen stand-in of this directory here.
#pragma once
#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>
#include <Eigen/Core>
namespace ocs2 {
using scalar_t = double;
using scalar_array_t = std::vector<scalar_t>;
using size_array_t = std::vector<size_t>;
using vector_t = Eigen::Matrix<scalar_t, Eigen::Dynamic, 1>;
using matrix_t = Eigen::Matrix<scalar_t, Eigen::Dynamic, Eigen::Dynamic>;
using vector_array_t = std::vector<vector_t>;
using matrix_array_t = std::vector<matrix_t>;
// upstream ocs2_core/Types.h: value + first (+ second) derivatives of a vector-valued function of (x, u)
struct VectorFunctionLinearApproximation {
  vector_t f; matrix_t dfdx, dfdu;
  static VectorFunctionLinearApproximation Zero(size_t nv, size_t nx, size_t nu) {
    VectorFunctionLinearApproximation a; a.f = vector_t::Zero((Eigen::Index)nv); a.dfdx = matrix_t::Zero((Eigen::Index)nv, (Eigen::Index)nx); a.dfdu = matrix_t::Zero((Eigen::Index)nv, (Eigen::Index)nu); return a;
  }
};
// upstream: value, gradient and Hessian blocks of a scalar function of (x, u)
struct ScalarFunctionQuadraticApproximation { scalar_t f = 0.0; vector_t dfdx, dfdu; matrix_t dfdxx, dfdux, dfduu; };
struct VectorFunctionQuadraticApproximation { vector_t f; matrix_t dfdx, dfdu; matrix_array_t dfdxx, dfdux, dfduu; };
}  // namespace ocs2
