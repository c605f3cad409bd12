// STAND-IN for <ocs2_core/Types.h> (upstream: Eigen typedefs).  vector_t: the subset of Eigen::VectorXd the adaptor uses.
#pragma once
#include <cstddef>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
namespace ocs2 {
using scalar_t = double;
using scalar_array_t = std::vector<scalar_t>;
using size_array_t = std::vector<size_t>;
class vector_t {
 public:
  vector_t() = default;
  explicit vector_t(size_t n) : v_(n, 0.0) {}
  static vector_t Zero(size_t n) { return vector_t(n); }
  size_t size() const { return v_.size(); }
  scalar_t* data() { return v_.data(); }
  const scalar_t* data() const { return v_.data(); }
  scalar_t& operator[](size_t i) { return v_[i]; }
  const scalar_t& operator[](size_t i) const { return v_[i]; }
  void setZero(size_t n) { v_.assign(n, 0.0); }
 private:
  std::vector<scalar_t> v_;
};
using vector_array_t = std::vector<vector_t>;
}  // namespace ocs2
