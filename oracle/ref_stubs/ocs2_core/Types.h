// STAND-IN (test infrastructure) for <ocs2_core/Types.h> of upstream leggedrobotics/ocs2: the scalar / array aliases the
// reference files compiled by oracle/Makefile (_ref target) use.  scalar_t is double upstream too.
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
namespace Eigen {   // only named by alias templates of humanoid_common_mpc/common/Types.h; never instantiated here
template <class S, int R, int C> class Matrix;
template <class S> class Quaternion;
}  // namespace Eigen
namespace ocs2 {
using scalar_t = double;
using scalar_array_t = std::vector<scalar_t>;
using size_array_t = std::vector<size_t>;
}  // namespace ocs2
