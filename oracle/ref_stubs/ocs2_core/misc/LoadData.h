// STAND-IN (test infrastructure) for <ocs2_core/misc/LoadData.h>: upstream loadPtreeValue reads pt.get<T>(name) and, if the key is
// missing, leaves the value untouched (printing a note when verbose) — the same here.
#pragma once
#include <boost/property_tree/info_parser.hpp>
#include <boost/property_tree/ptree.hpp>
#include <ocs2_core/Types.h>
namespace ocs2 { namespace loadData {
template <class T>
inline void loadPtreeValue(const boost::property_tree::ptree& pt, T& value, const std::string& name, bool verbose) {
  if (pt.find(name)) value = pt.get<T>(name);
  else if (verbose) std::cerr << " #### '" << name << "' is not defined, the default is kept\n";
}
// upstream loadCppDataType: read_info(filename) then value = pt.get<T>(dataName) (throws when the key is missing)
template <class T>
inline void loadCppDataType(const std::string& filename, const std::string& dataName, T& value) {
  boost::property_tree::ptree pt;
  boost::property_tree::read_info(filename, pt);
  value = pt.get<T>(dataName);
}
// upstream loadEigenMatrix: the matrix keeps its size; entry (i,j) = pt.get(matrixName + ".(i,j)", 0.0) * pt.get(matrixName + ".scaling", 1.0)
// (a missing entry reads as zero with a warning)
template <class M>
inline void loadEigenMatrix(const std::string& filename, const std::string& matrixName, M& matrix) {
  boost::property_tree::ptree pt;
  boost::property_tree::read_info(filename, pt);
  const scalar_t scaling = pt.find(matrixName + ".scaling") ? pt.get<scalar_t>(matrixName + ".scaling") : 1.0;
  for (long i = 0; i < (long)matrix.rows(); ++i)
    for (long j = 0; j < (long)matrix.cols(); ++j) {
      const std::string key = matrixName + ".(" + std::to_string(i) + "," + std::to_string(j) + ")";
      matrix(i, j) = (pt.find(key) ? pt.get<scalar_t>(key) : 0.0) * scaling;
    }
}
}}  // namespace ocs2::loadData
