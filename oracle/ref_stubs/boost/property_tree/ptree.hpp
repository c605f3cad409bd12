// STAND-IN (test infrastructure) for Boost.PropertyTree: a nested key -> (value, children) tree with the few accessors the reference's
// loaders use through ocs2 loadData (get_child / get<T> by dotted path).
#pragma once
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
namespace boost { namespace property_tree {
class ptree {
 public:
  std::string data;
  std::vector<std::pair<std::string, ptree>> children;
  const ptree* find(const std::string& path) const {
    const ptree* t = this;
    size_t pos = 0;
    while (pos <= path.size()) {
      const size_t dot = path.find('.', pos);
      const std::string key = path.substr(pos, dot == std::string::npos ? std::string::npos : dot - pos);
      const ptree* nxt = nullptr;
      for (const auto& c : t->children) if (c.first == key) { nxt = &c.second; break; }
      if (!nxt) return nullptr;
      t = nxt;
      if (dot == std::string::npos) break;
      pos = dot + 1;
    }
    return t;
  }
  const ptree& get_child(const std::string& path) const { const ptree* t = find(path); if (!t) throw std::runtime_error("ptree: no such node (" + path + ")"); return *t; }
  template <class T> T get(const std::string& path) const {
    const ptree& t = get_child(path);
    std::istringstream is(t.data);
    T v;
    if constexpr (std::is_same<T, bool>::value) { std::string s; is >> s; v = (s == "true" || s == "1"); }
    else if constexpr (std::is_same<T, std::string>::value) { v = t.data; }
    else { is >> v; if (is.fail()) throw std::runtime_error("ptree: conversion of \"" + t.data + "\" failed (" + path + ")"); }
    return v;
  }
};
}}  // namespace boost::property_tree
