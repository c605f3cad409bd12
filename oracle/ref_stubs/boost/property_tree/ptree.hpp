// STAND-IN (test infrastructure) for Boost.PropertyTree: an opaque tree type, nothing is ever parsed through it here.
#pragma once
#include <string>
namespace boost { namespace property_tree { class ptree {}; }}
