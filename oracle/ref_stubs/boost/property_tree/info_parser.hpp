// STAND-IN (test infrastructure) for Boost.PropertyTree's INFO parser (never called here).
#pragma once
#include <stdexcept>
#include <boost/property_tree/ptree.hpp>
namespace boost { namespace property_tree { inline void read_info(const std::string&, ptree&) { throw std::runtime_error("read_info: stand-in"); } }}
