// STAND-IN (test infrastructure) for <ocs2_core/misc/LoadData.h>: upstream loadPtreeValue reads pt.get<T>(name) and, if the key is
// missing, leaves the value untouched (printing a note when verbose) — the same here.
#pragma once
#include <boost/property_tree/ptree.hpp>
#include <ocs2_core/Types.h>
namespace ocs2 { namespace loadData {
template <class T>
inline void loadPtreeValue(const boost::property_tree::ptree& pt, T& value, const std::string& name, bool verbose) {
  if (pt.find(name)) value = pt.get<T>(name);
  else if (verbose) std::cerr << " #### '" << name << "' is not defined, the default is kept\n";
}
}}  // namespace ocs2::loadData
