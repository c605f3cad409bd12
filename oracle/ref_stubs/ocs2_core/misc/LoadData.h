// STAND-IN (test infrastructure) for <ocs2_core/misc/LoadData.h>: the .info loaders are declared so that the reference files
// compile; the driver never loads a file through them (oracle/ref_driver.cpp passes the task.info values in).
#pragma once
#include <boost/property_tree/ptree.hpp>
#include <ocs2_core/Types.h>
namespace ocs2 { namespace loadData {
template <class T>
inline void loadPtreeValue(const boost::property_tree::ptree&, T&, const std::string&, bool) { throw std::runtime_error("loadPtreeValue: stand-in"); }
}}  // namespace ocs2::loadData
