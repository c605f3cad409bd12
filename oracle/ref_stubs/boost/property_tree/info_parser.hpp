// STAND-IN (test infrastructure) for Boost.PropertyTree's INFO parser: `key value`, `key { ... }`, `;` comments, quoted strings — what
// the reference's task.info / reference.info / gait.info files use.
#pragma once
#include <fstream>
#include <stdexcept>
#include <boost/property_tree/ptree.hpp>
namespace boost { namespace property_tree {
namespace info_detail {
inline std::vector<std::string> tokens(const std::string& file) {
  std::ifstream in(file);
  if (!in) throw std::runtime_error("read_info: cannot open " + file);
  std::vector<std::string> out;
  std::string line;
  while (std::getline(in, line)) {
    size_t i = 0;
    while (i < line.size()) {
      const char c = line[i];
      if (c == ';') break;
      if (c == ' ' || c == '\t' || c == '\r') { ++i; continue; }
      if (c == '{' || c == '}') { out.push_back(std::string(1, c)); ++i; continue; }
      if (c == '"') { const size_t e = line.find('"', i + 1); out.push_back("\"" + line.substr(i + 1, e - i - 1)); i = e == std::string::npos ? line.size() : e + 1; continue; }
      size_t e = i;
      while (e < line.size() && line[e] != ' ' && line[e] != '\t' && line[e] != '\r' && line[e] != ';' && line[e] != '{' && line[e] != '}') ++e;
      out.push_back(line.substr(i, e - i));
      i = e;
    }
    out.push_back("\n");
  }
  return out;
}
inline void parse(const std::vector<std::string>& tk, size_t& i, ptree& node) {
  while (i < tk.size()) {
    if (tk[i] == "\n") { ++i; continue; }
    if (tk[i] == "}") { ++i; return; }
    std::string key = tk[i++];
    if (!key.empty() && key[0] == '"') key = key.substr(1);
    ptree child;
    if (i < tk.size() && tk[i] != "\n" && tk[i] != "{" && tk[i] != "}") { child.data = tk[i][0] == '"' ? tk[i].substr(1) : tk[i]; ++i; }
    while (i < tk.size() && tk[i] == "\n") ++i;
    if (i < tk.size() && tk[i] == "{") { ++i; parse(tk, i, child); }
    node.children.emplace_back(key, std::move(child));
  }
}
}  // namespace info_detail
inline void read_info(const std::string& file, ptree& pt) { const auto tk = info_detail::tokens(file); size_t i = 0; info_detail::parse(tk, i, pt); }
}}  // namespace boost::property_tree
