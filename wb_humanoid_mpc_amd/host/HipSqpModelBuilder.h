// hsqp_model_desc straight from the reference's own input files — (taskFile, urdfFile, referenceFile), the arguments of the reference's
// WBMpcInterface / CentroidalMpcInterface constructors (humanoid_nmpc/humanoid_wb_mpc/include/humanoid_wb_mpc/WBMpcInterface.h:73-75,
// humanoid_wb_mpc_ros2/src/WBMpcSqpNode.cpp:61) — WITHOUT Python and without Pinocchio / urdfdom / Boost: a drop-in main() includes this
// header, calls buildModelDesc(...) and hands the POD to hsqp_create.  It restates, in C++17 and the standard library only, what
// tools/export_g1_model.py does (tests/test_host_cpp.py checks the two against each other field by field):
//   * URDF -> MPC kinematic tree: joints named in model_settings.fixedJointNames become fixed, fixed children are lumped into their parent
//     body (mass, centre of mass, rotational inertia), bodies depth-first with siblings in alphabetical joint-name order (urdfdom keeps
//     joints in a std::map and pinocchio::urdf::buildModel walks it), composite Translation + SphericalZYX base
//     (humanoid_common_mpc/src/pinocchio_model/createPinocchioModel.cpp:60-67,139-182);
//   * contact frames / collision points / ankle and knee frames (createPinocchioModel.cpp:77-128, task.info contacts / collision_constraint);
//   * weights, gains, barrier settings, swing-trajectory configuration (task.info), the foot-cost weight quirk
//     (humanoid_wb_mpc/src/cost/EndEffectorDynamicsCostHelpers.cpp:100-108), defaultJointState / defaultBaseHeight (reference.info).
// The intermediate product is the same problem image (a JsonValue tree) that HipSqpModelIO.h reads from the exported JSON, so both routes
// end in modelDescFromImage().
#pragma once
#include <algorithm>
#include <array>
#include <cctype>
#include <functional>
#include <set>

#include "HipSqpModelIO.h"

namespace hsqp_host {

// ------------------------------------------------------------------------------------------------ Boost INFO subset
struct InfoNode {
  std::string value;                                        // "key value"
  std::vector<std::pair<std::string, InfoNode>> children;   // "key { ... }" (insertion order kept)
  const InfoNode* find(const std::string& k) const { for (const auto& c : children) if (c.first == k) return &c.second; return nullptr; }
  const InfoNode& at(const std::string& k) const { const InfoNode* n = find(k); if (!n) throw std::runtime_error("[HipSqpModelBuilder] INFO: missing key '" + k + "'"); return *n; }
  double num(const std::string& k) const { return std::stod(at(k).value); }
  double num_or(const std::string& k, double d) const { const InfoNode* n = find(k); return n ? std::stod(n->value) : d; }
};
namespace detail {
inline std::vector<std::string> infoTokens(const std::string& path) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("[HipSqpModelBuilder] cannot open " + path);
  std::vector<std::string> toks;
  std::string line;
  while (std::getline(in, line)) {
    // Boost INFO: a comment starts at a ';' OUTSIDE a quoted string and runs to the end of the line.  "//" is NOT a comment marker there
    // (package:// URIs and paths are values) — but the reference's task files annotate entries as `key value  // remark`
    // (g1_wb_mpc/config/mpc/task.info:3), and user files also write `} // end`, `block { // remark`, `key // remark` or whole-line remarks.
    int words = 0;
    for (size_t i = 0; i < line.size();) {
      if (std::isspace((unsigned char)line[i])) { ++i; continue; }
      if (line[i] == ';') break;
      if (line.compare(i, 2, "//") == 0) {
        // a remark: "//" anywhere but in the value position of a pair (first token of a line, behind a brace, behind a complete pair), or a
        // bare "//" followed by white space even there (`key // remark`); `key //abs/dir` and `key package://...` stay values
        const bool bare = i + 2 >= line.size() || std::isspace((unsigned char)line[i + 2]);
        if (words != 1 || bare) break;
      }
      if (line[i] == '{' || line[i] == '}') { toks.emplace_back(1, line[i++]); words = 0; continue; }
      ++words;
      if (line[i] == '"') { const size_t e = line.find('"', i + 1); toks.push_back(line.substr(i + 1, e == std::string::npos ? std::string::npos : e - i - 1)); i = e == std::string::npos ? line.size() : e + 1; continue; }
      size_t e = i;
      while (e < line.size() && !std::isspace((unsigned char)line[e]) && line[e] != '{' && line[e] != '}' && line[e] != ';') ++e;
      toks.push_back(line.substr(i, e - i));
      i = e;
    }
  }
  return toks;
}
inline void infoBlock(const std::vector<std::string>& t, size_t& pos, InfoNode& node) {
  while (pos < t.size()) {
    if (t[pos] == "}") { ++pos; return; }
    const std::string key = t[pos++];
    InfoNode child;
    if (pos < t.size() && t[pos] == "{") { ++pos; infoBlock(t, pos, child); }
    else if (pos < t.size()) { child.value = t[pos++]; if (pos < t.size() && t[pos] == "{") { ++pos; infoBlock(t, pos, child); } }
    node.children.emplace_back(key, std::move(child));
  }
}
// "[i] value" lists in index order
inline std::vector<std::string> infoList(const InfoNode& n) {
  std::vector<std::pair<int, std::string>> items;
  for (const auto& c : n.children) if (c.first.size() > 2 && c.first.front() == '[') items.emplace_back(std::stoi(c.first.substr(1)), c.second.value);
  std::sort(items.begin(), items.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
  std::vector<std::string> out;
  for (auto& it : items) out.push_back(it.second);
  return out;
}
// loadData::loadEigenMatrix semantics for "(i,j) v" entries with an optional "scaling": here the diagonal / the first column
inline std::vector<double> infoEntries(const InfoNode& n, size_t size, bool diagonal, bool scaled) {
  const double scaling = scaled ? n.num_or("scaling", 1.0) : 1.0;
  std::vector<double> v(size, 0.0);
  for (const auto& c : n.children) {
    int i = 0, j = 0;
    if (std::sscanf(c.first.c_str(), "(%d,%d)", &i, &j) != 2) continue;
    if (diagonal) { if (i != j) { if (std::stod(c.second.value) != 0.0) throw std::runtime_error("[HipSqpModelBuilder] off-diagonal weight"); continue; } }
    else if (j != 0) continue;
    if ((size_t)i < size) v[i] = std::stod(c.second.value) * scaling;
  }
  return v;
}
}  // namespace detail
inline InfoNode parseInfoFile(const std::string& path) { const auto t = detail::infoTokens(path); size_t pos = 0; InfoNode root; detail::infoBlock(t, pos, root); return root; }

// ------------------------------------------------------------------------------------------------ XML subset (URDF: elements + attributes)
struct XmlNode {
  std::string tag;
  std::map<std::string, std::string> attr;
  std::vector<XmlNode> children;
  const XmlNode* child(const std::string& t) const { for (const auto& c : children) if (c.tag == t) return &c; return nullptr; }
  std::string get(const std::string& a, const std::string& dflt = "") const { auto it = attr.find(a); return it == attr.end() ? dflt : it->second; }
};
namespace detail {
class XmlParser {
 public:
  explicit XmlParser(std::string s) : s_(std::move(s)) {}
  XmlNode root() { skipMisc(); XmlNode n; if (!element(n)) throw std::runtime_error("[HipSqpModelBuilder] URDF: no root element"); return n; }
 private:
  void skipMisc() {   // whitespace, <?...?>, <!-- ... -->, <!DOCTYPE ...>
    for (;;) {
      while (i_ < s_.size() && std::isspace((unsigned char)s_[i_])) ++i_;
      if (s_.compare(i_, 4, "<!--") == 0) { const size_t e = s_.find("-->", i_); i_ = e == std::string::npos ? s_.size() : e + 3; }
      else if (s_.compare(i_, 2, "<?") == 0) { const size_t e = s_.find("?>", i_); i_ = e == std::string::npos ? s_.size() : e + 2; }
      else if (s_.compare(i_, 2, "<!") == 0) { const size_t e = s_.find('>', i_); i_ = e == std::string::npos ? s_.size() : e + 1; }
      else return;
    }
  }
  bool element(XmlNode& n) {
    if (i_ >= s_.size() || s_[i_] != '<' || s_[i_ + 1] == '/') return false;
    ++i_;
    while (i_ < s_.size() && !std::isspace((unsigned char)s_[i_]) && s_[i_] != '>' && s_[i_] != '/') n.tag.push_back(s_[i_++]);
    for (;;) {
      while (i_ < s_.size() && std::isspace((unsigned char)s_[i_])) ++i_;
      if (s_.compare(i_, 2, "/>") == 0) { i_ += 2; return true; }
      if (s_[i_] == '>') { ++i_; break; }
      std::string k;
      while (i_ < s_.size() && s_[i_] != '=' && !std::isspace((unsigned char)s_[i_])) k.push_back(s_[i_++]);
      while (i_ < s_.size() && s_[i_] != '"' && s_[i_] != '\'') ++i_;
      const char q = s_[i_++];
      const size_t e = s_.find(q, i_);
      if (e == std::string::npos) throw std::runtime_error("[HipSqpModelBuilder] URDF: unterminated attribute");
      n.attr[k] = s_.substr(i_, e - i_);
      i_ = e + 1;
    }
    for (;;) {   // content: child elements (text is ignored)
      while (i_ < s_.size() && s_[i_] != '<') ++i_;
      skipMisc();
      if (i_ >= s_.size()) throw std::runtime_error("[HipSqpModelBuilder] URDF: unexpected end inside <" + n.tag + ">");
      if (s_.compare(i_, 2, "</") == 0) { const size_t e = s_.find('>', i_); i_ = e + 1; return true; }
      XmlNode c;
      if (element(c)) n.children.push_back(std::move(c));
    }
  }
  std::string s_;
  size_t i_ = 0;
};
}  // namespace detail
inline XmlNode parseXmlFile(const std::string& path) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("[HipSqpModelBuilder] cannot open " + path);
  std::stringstream ss;
  ss << in.rdbuf();
  return detail::XmlParser(ss.str()).root();
}

// ------------------------------------------------------------------------------------------------ rigid-body bookkeeping
namespace detail {
using V3 = std::array<double, 3>;
using M3 = std::array<double, 9>;   // row-major
inline M3 eye() { return {1, 0, 0, 0, 1, 0, 0, 0, 1}; }
inline M3 mul(const M3& a, const M3& b) { M3 c{}; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += a[3 * i + k] * b[3 * k + j]; c[3 * i + j] = s; } return c; }
inline M3 tr(const M3& a) { return {a[0], a[3], a[6], a[1], a[4], a[7], a[2], a[5], a[8]}; }
inline V3 mul(const M3& a, const V3& v) { return {a[0] * v[0] + a[1] * v[1] + a[2] * v[2], a[3] * v[0] + a[4] * v[1] + a[5] * v[2], a[6] * v[0] + a[7] * v[1] + a[8] * v[2]}; }
inline V3 add(const V3& a, const V3& b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
inline M3 rpy(double r, double p, double y) {   // URDF convention: Rz(yaw) Ry(pitch) Rx(roll)
  const double cr = std::cos(r), sr = std::sin(r), cp = std::cos(p), sp = std::sin(p), cy = std::cos(y), sy = std::sin(y);
  const M3 Rx{1, 0, 0, 0, cr, -sr, 0, sr, cr}, Ry{cp, 0, sp, 0, 1, 0, -sp, 0, cp}, Rz{cy, -sy, 0, sy, cy, 0, 0, 0, 1};
  return mul(mul(Rz, Ry), Rx);
}
inline V3 floats3(const std::string& s) { V3 v{0, 0, 0}; std::istringstream is(s); is >> v[0] >> v[1] >> v[2]; return v; }
struct Inertia {   // mass, centre of mass and rotational inertia about it, in the owner's frame
  double m = 0.0; V3 c{0, 0, 0}; M3 I{};
  Inertia transformed(const M3& R, const V3& p) const { Inertia o; o.m = m; o.c = add(mul(R, c), p); o.I = mul(mul(R, I), tr(R)); return o; }
  Inertia operator+(const Inertia& o) const {
    Inertia s;
    s.m = m + o.m;
    if (s.m == 0.0) return Inertia();
    for (int k = 0; k < 3; ++k) s.c[k] = (m * c[k] + o.m * o.c[k]) / s.m;
    auto shifted = [&](const Inertia& b) {   // parallel-axis shift to the common centre of mass
      const V3 d{b.c[0] - s.c[0], b.c[1] - s.c[1], b.c[2] - s.c[2]};
      const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
      M3 r = b.I;
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r[3 * i + j] += b.m * ((i == j ? dd : 0.0) - d[i] * d[j]);
      return r;
    };
    const M3 a = shifted(*this), b2 = shifted(o);
    for (int k = 0; k < 9; ++k) s.I[k] = a[k] + b2[k];
    return s;
  }
};
struct UrdfJoint { std::string name, type, parent, child; M3 R = eye(); V3 p{0, 0, 0}, axis{0, 0, 0}; bool has_limit = false; double lo = 0, hi = 0; };
struct Body { std::string name, joint; int parent = -1; M3 R = eye(); V3 p{0, 0, 0}, axis{0, 0, 0}; Inertia inertia; bool has_limit = false; double lo = 0, hi = 0; };
struct Placed { int body = 0; M3 R = eye(); V3 p{0, 0, 0}; };
inline JsonValue jnum(double v) { JsonValue j; j.kind = JsonValue::Number; j.num = v; return j; }
inline JsonValue jstr(const std::string& v) { JsonValue j; j.kind = JsonValue::String; j.str = v; return j; }
template <class C> JsonValue jarr(const C& c) { JsonValue j; j.kind = JsonValue::Array; for (double v : c) j.arr.push_back(jnum(v)); return j; }
inline JsonValue jobj() { JsonValue j; j.kind = JsonValue::Object; return j; }
}  // namespace detail

/** The problem image of tools/export_g1_model.py — the fields modelDescFromImage / swingConfigFromImage read, plus the defaults a caller
 *  needs to build references (default joint state, base height, initial state, SQP settings, horizon) — from the reference's files. */
inline JsonValue buildProblemImage(const std::string& taskFile, const std::string& urdfFile, const std::string& referenceFile, bool centroidal = false) {
  using namespace detail;
  const InfoNode task = parseInfoFile(taskFile), ref = parseInfoFile(referenceFile);
  const InfoNode& ms = task.at("model_settings");
  const std::vector<std::string> fixedList = infoList(ms.at("fixedJointNames"));
  const std::set<std::string> fixed(fixedList.begin(), fixedList.end());
  // ---- URDF
  const XmlNode urdf = parseXmlFile(urdfFile);
  std::map<std::string, Inertia> links;
  std::vector<std::string> linkOrder;
  for (const XmlNode& l : urdf.children) {
    if (l.tag != "link") continue;
    Inertia in;
    if (const XmlNode* ine = l.child("inertial")) {
      const XmlNode* o = ine->child("origin");
      const V3 xyz = o ? floats3(o->get("xyz", "0 0 0")) : V3{0, 0, 0}, r = o ? floats3(o->get("rpy", "0 0 0")) : V3{0, 0, 0};
      const XmlNode* i = ine->child("inertia");
      const XmlNode* m = ine->child("mass");
      if (!i || !m) throw std::runtime_error("[HipSqpModelBuilder] URDF: <inertial> without mass / inertia in link " + l.get("name"));
      auto a = [&](const char* k) { return std::stod(i->get(k, "0")); };
      const M3 I{a("ixx"), a("ixy"), a("ixz"), a("ixy"), a("iyy"), a("iyz"), a("ixz"), a("iyz"), a("izz")};
      const M3 R = rpy(r[0], r[1], r[2]);
      in.m = std::stod(m->get("value", "0")); in.c = xyz; in.I = mul(mul(R, I), tr(R));
    }
    links[l.get("name")] = in;
    linkOrder.push_back(l.get("name"));
  }
  std::map<std::string, UrdfJoint> joints;
  std::map<std::string, std::vector<std::string>> children;
  std::set<std::string> childLinks;
  for (const XmlNode& j : urdf.children) {
    if (j.tag != "joint") continue;
    UrdfJoint u;
    u.name = j.get("name"); u.type = fixed.count(u.name) ? "fixed" : j.get("type");
    u.parent = j.child("parent")->get("link"); u.child = j.child("child")->get("link");
    if (const XmlNode* o = j.child("origin")) { const V3 r = floats3(o->get("rpy", "0 0 0")); u.R = rpy(r[0], r[1], r[2]); u.p = floats3(o->get("xyz", "0 0 0")); }
    if (const XmlNode* a = j.child("axis")) u.axis = floats3(a->get("xyz", "0 0 0"));
    if (const XmlNode* lim = j.child("limit")) if (u.type != "fixed") { u.has_limit = true; u.lo = std::stod(lim->get("lower", "0")); u.hi = std::stod(lim->get("upper", "0")); }
    children[u.parent].push_back(u.name);
    childLinks.insert(u.child);
    joints[u.name] = u;
  }
  std::string rootLink;
  for (const auto& n : linkOrder) if (!childLinks.count(n)) { if (!rootLink.empty()) throw std::runtime_error("[HipSqpModelBuilder] URDF: more than one root link"); rootLink = n; }
  // ---- MPC tree: depth first, siblings in alphabetical joint-name order, fixed children lumped into the parent body
  std::vector<Body> bodies;
  std::map<std::string, Placed> frames;   // every URDF link / joint frame: (body, placement in the body frame)
  bodies.emplace_back();
  bodies[0].name = rootLink; bodies[0].joint = "root_joint";
  std::function<void(const std::string&, int, const M3&, const V3&)> visit = [&](const std::string& link, int b, const M3& Rbl, const V3& pbl) {
    bodies[b].inertia = bodies[b].inertia + links.at(link).transformed(Rbl, pbl);
    frames[link] = Placed{b, Rbl, pbl};
    std::vector<std::string> kids = children[link];
    std::sort(kids.begin(), kids.end());
    for (const std::string& jn : kids) {
      const UrdfJoint& j = joints.at(jn);
      const M3 Rbj = mul(Rbl, j.R);
      const V3 pbj = add(mul(Rbl, j.p), pbl);
      if (j.type == "fixed") { frames[jn] = Placed{b, Rbj, pbj}; visit(j.child, b, Rbj, pbj); }
      else {
        if (j.type != "revolute") throw std::runtime_error("[HipSqpModelBuilder] URDF: joint type '" + j.type + "' of " + jn + " is not supported");
        Body nb;
        nb.name = j.child; nb.joint = jn; nb.parent = b; nb.R = Rbj; nb.p = pbj; nb.axis = j.axis; nb.has_limit = j.has_limit; nb.lo = j.lo; nb.hi = j.hi;
        bodies.push_back(nb);
        const int idx = (int)bodies.size() - 1;
        frames[jn] = Placed{idx, eye(), V3{0, 0, 0}};
        visit(j.child, idx, eye(), V3{0, 0, 0});
      }
    }
  };
  visit(rootLink, 0, eye(), V3{0, 0, 0});
  const int nj = (int)bodies.size() - 1;
  // the MPC joint order must be the one of task.info's jointNames (the reference re-orders nothing: createPinocchioModel.cpp:139-182)
  std::vector<std::string> jointNames;
  for (int i = 1; i <= nj; ++i) jointNames.push_back(bodies[i].joint);
  if (const InfoNode* want = ms.find("jointNames")) { const auto w = infoList(*want); if (!w.empty() && w != jointNames) throw std::runtime_error("[HipSqpModelBuilder] MPC joint order differs from model_settings.jointNames"); }
  auto jointIndex = [&](const std::string& n) { for (int i = 0; i < nj; ++i) if (jointNames[i] == n) return i; throw std::runtime_error("[HipSqpModelBuilder] unknown joint " + n); };
  double totalMass = 0.0;
  for (const Body& b : bodies) totalMass += b.inertia.m;

  JsonValue img = jobj();
  img.obj["formulation"] = jstr(centroidal ? "centroidal" : "wb");
  img.obj["robot"] = jstr(urdf.get("name"));
  const size_t nx = centroidal ? 12 + nj : 2 * (6 + nj), nu = 12 + nj;
  img.obj["nj"] = jnum(nj); img.obj["nx"] = jnum((double)nx); img.obj["nu"] = jnum((double)nu); img.obj["gravity"] = jnum(9.81); img.obj["total_mass"] = jnum(totalMass);
  { JsonValue a; a.kind = JsonValue::Array; for (const auto& n : jointNames) a.arr.push_back(jstr(n)); img.obj["joint_names"] = a; }
  {
    JsonValue a; a.kind = JsonValue::Array;
    for (const Body& b : bodies) {
      JsonValue o = jobj();
      o.obj["name"] = jstr(b.name); o.obj["joint"] = jstr(b.joint); o.obj["parent"] = jnum(b.parent); o.obj["R"] = jarr(b.R); o.obj["p"] = jarr(b.p); o.obj["axis"] = jarr(b.axis);
      o.obj["mass"] = jnum(b.inertia.m); o.obj["com"] = jarr(b.inertia.c); o.obj["inertia"] = jarr(b.inertia.I);
      o.obj["lo"] = b.has_limit ? jnum(b.lo) : JsonValue(); o.obj["hi"] = b.has_limit ? jnum(b.hi) : JsonValue();
      a.arr.push_back(o);
    }
    img.obj["bodies"] = a;
  }
  // ---- contact frames (createPinocchioModel.cpp:77-83), collision points (:92-106), ankle / knee frames
  const std::vector<std::string> contactParents = infoList(ms.at("contactParentJointNames"));
  const InfoNode& contactsCfg = task.at("contacts");
  const InfoNode& cft = contactsCfg.at("contact_frame_translation");
  const V3 tc{cft.num("x"), cft.num("y"), cft.num("z")};
  const InfoNode& rect = contactsCfg.at("contact_rectangle");
  const double scale = rect.num_or("scale_factor", 1.0);
  const double xmin = rect.num("x_min") * scale, xmax = rect.num("x_max") * scale, ymin = rect.num("y_min") * scale, ymax = rect.num("y_max") * scale;
  auto frameOnJoint = [&](const std::string& jn, const V3& off) {
    const auto it = frames.find(jn);
    if (it == frames.end()) throw std::runtime_error("[HipSqpModelBuilder] no frame '" + jn + "' in the URDF");
    const Placed& f = it->second;
    for (int k = 0; k < 9; ++k) if (std::fabs(f.R[k] - eye()[k]) > 1e-12) throw std::runtime_error("[HipSqpModelBuilder] frame '" + jn + "' is rotated against its body");
    JsonValue o = jobj();
    o.obj["body"] = jnum(f.body); o.obj["p"] = jarr(add(f.p, off));
    return o;
  };
  auto pair = [&](const std::function<JsonValue(int)>& f) { JsonValue a; a.kind = JsonValue::Array; a.arr.push_back(f(0)); a.arr.push_back(f(1)); return a; };
  const InfoNode& cc = task.at("collision_constraint");
  JsonValue fr = jobj();
  fr.obj["contact"] = pair([&](int i) { return frameOnJoint(contactParents.at(i), tc); });
  fr.obj["collision_p1"] = pair([&](int i) { return frameOnJoint(contactParents.at(i), add(tc, V3{xmax * 0.6, 0, 0})); });
  fr.obj["collision_p2"] = pair([&](int i) { return frameOnJoint(contactParents.at(i), add(tc, V3{xmin * 0.6, 0, 0})); });
  fr.obj["ankle"] = pair([&](int i) { return frameOnJoint(cc.at("foot").at(i == 0 ? "leftAnkleFrame" : "rightAnkleFrame").value, V3{0, 0, 0}); });
  fr.obj["knee"] = pair([&](int i) { return frameOnJoint(cc.at("knee").at(i == 0 ? "leftKneeFrame" : "rightKneeFrame").value, V3{0, 0, 0}); });
  img.obj["frames"] = fr;
  { JsonValue r = jobj(); r.obj["x_min"] = jnum(xmin); r.obj["x_max"] = jnum(xmax); r.obj["y_min"] = jnum(ymin); r.obj["y_max"] = jnum(ymax); img.obj["contact_rectangle"] = r; }
  // ---- weights
  img.obj["Q"] = jarr(infoEntries(task.at("Q"), nx, true, true));
  img.obj["R"] = jarr(infoEntries(task.at("R"), nu, true, true));
  { std::vector<double> qf = infoEntries(task.at("Q_final"), nx, true, true); const double sc = task.num("terminalCostScaling"); for (double& v : qf) v *= sc; img.obj["Qf"] = jarr(qf); }
  // EndEffectorDynamicsWeights::getWeights (EndEffectorDynamicsCostHelpers.cpp:100-108): the lin / ang VELOCITY weights are overwritten with
  // the ACCELERATION entries of task.info, the acceleration weights keep their struct defaults 0.01 (EndEffectorDynamicsCostHelpers.h:45-50)
  const InfoNode& fw = task.at("task_space_foot_cost_weights");
  {
    std::vector<double> w;
    for (const char* k : {"pos_x", "pos_y", "pos_z", "orientation_x", "orientation_y", "orientation_z", "lin_acceleration_x", "lin_acceleration_y", "lin_acceleration_z",
                          "ang_acceleration_x", "ang_acceleration_y", "ang_acceleration_z"}) w.push_back(fw.num(k));
    for (int k = 0; k < 6; ++k) w.push_back(0.01);
    img.obj["foot_cost_weights"] = jarr(w);
  }
  if (centroidal) {
    // CentroidalMpcInterface.cpp:139-215: EndEffectorKinematicsWeights::getWeights for the feet and the torso link, ExternalTorqueQuadraticCostAD
    const char* kin[12] = {"pos_x", "pos_y", "pos_z", "orientation_x", "orientation_y", "orientation_z", "lin_velocity_x", "lin_velocity_y", "lin_velocity_z",
                           "ang_velocity_x", "ang_velocity_y", "ang_velocity_z"};
    std::vector<double> cw;
    for (const char* k : kin) cw.push_back(fw.num(k));
    img.obj["cent_foot_cost_weights"] = jarr(cw);
    const InfoNode& torso = task.at("task_space_costs").at("torso");
    const Placed& tf = frames.at(torso.at("link_name").value);
    JsonValue t = jobj();
    t.obj["link"] = jstr(torso.at("link_name").value); t.obj["body"] = jnum(tf.body); t.obj["R"] = jarr(tf.R); t.obj["p"] = jarr(tf.p);
    std::vector<double> tw;
    for (const char* k : kin) tw.push_back(torso.at("weights").num(k));
    t.obj["weights"] = jarr(tw);
    img.obj["torso"] = t;
    JsonValue ext = jobj(), ew, ej;
    ew.kind = ej.kind = JsonValue::Array;
    for (const char* side : {"left_leg_torque_cost", "right_leg_torque_cost"}) {
      const std::vector<std::string> names = infoList(task.at(side).at("activeJointNames"));
      ew.arr.push_back(jarr(infoEntries(task.at(side).at("weights"), names.size(), false, true)));   // (the script multiplies by the scaling once: loadEigenMatrix does)
      std::vector<double> idx;
      for (const auto& n : names) idx.push_back(jointIndex(n));
      ej.arr.push_back(jarr(idx));
    }
    ext.obj["weights"] = ew; ext.obj["joints"] = ej;
    img.obj["ext_torque"] = ext;
    img.obj["icp_weight"] = jnum(task.at("icp_cost_weights").num("icpErrorWeight"));
    img.obj["centroidal_model_type"] = jnum(task.num("centroidalModelType"));
  }
  auto copyAll = [&](const InfoNode& n) { JsonValue o = jobj(); for (const auto& c : n.children) if (c.second.children.empty()) { try { o.obj[c.first] = jnum(std::stod(c.second.value)); } catch (const std::exception&) { o.obj[c.first] = jstr(c.second.value); } } return o; };
  img.obj["foot_constraint"] = copyAll(ms.at("foot_constraint"));
  auto barrierOf = [&](const InfoNode& n) { JsonValue o = jobj(); o.obj["barrier_mu"] = jnum(n.num("mu")); o.obj["barrier_delta"] = jnum(n.num("delta")); return o; };
  {
    const InfoNode& f = contactsCfg.at("frictionForceConeSoftConstraint");
    JsonValue o = barrierOf(f);
    // FrictionForceConeConstraint::Config defaults (FrictionForceConeConstraint.h: regularization 25, gripperForce 0, hessianDiagonalShift 1e-6)
    o.obj["mu"] = jnum(f.num("frictionCoefficient")); o.obj["regularization"] = jnum(25.0); o.obj["gripper_force"] = jnum(0.0); o.obj["hessian_diagonal_shift"] = jnum(1e-6);
    img.obj["friction"] = o;
  }
  img.obj["moment_xy"] = barrierOf(contactsCfg.at("contactMomentXYSoftConstraint"));
  img.obj["joint_limits"] = barrierOf(task.at("jointLimits"));
  { JsonValue o = barrierOf(cc); o.obj["r_foot"] = jnum(cc.at("foot").num("footCollisionSphereRadius")); o.obj["r_knee"] = jnum(cc.at("knee").num("kneeCollisionSphereRadius")); img.obj["collision"] = o; }
  { const InfoNode& arm = ms.at("armJointNames"); img.obj["arm_swing_joints"] = jarr(std::vector<double>{(double)jointIndex(arm.at("left_shoulder_y").value), (double)jointIndex(arm.at("right_shoulder_y").value), (double)jointIndex(arm.at("left_elbow_y").value), (double)jointIndex(arm.at("right_elbow_y").value)}); }
  img.obj["swing"] = copyAll(task.at("swing_trajectory_config"));
  { const InfoNode& q = task.at("multiple_shooting"); JsonValue o = jobj(); for (const char* k : {"dt", "sqpIteration", "deltaTol", "g_max", "g_min", "nThreads"}) o.obj[k] = jnum(q.num(k)); img.obj["sqp"] = o; }
  { JsonValue o = jobj(); o.obj["timeHorizon"] = jnum(task.at("mpc").num("timeHorizon")); img.obj["mpc"] = o; }
  img.obj["initial_state"] = jarr(infoEntries(task.at("initialState"), nx, false, false));
  img.obj["default_joint_state"] = jarr(infoEntries(ref.at("defaultJointState"), nj, false, false));
  img.obj["default_base_height"] = jnum(ref.num("defaultBaseHeight"));
  img.obj["phase_transition_stance_time"] = jnum(ms.num("phaseTransitionStanceTime"));
  return img;
}

/** (taskFile, urdfFile, referenceFile) -> the POD of hsqp_create: what the reference's interface constructors take. */
inline hsqp_model_desc buildModelDesc(const std::string& taskFile, const std::string& urdfFile, const std::string& referenceFile, bool centroidal = false) {
  return modelDescFromImage(buildProblemImage(taskFile, urdfFile, referenceFile, centroidal));
}
inline hsqp_swing_config buildSwingConfig(const std::string& taskFile, const std::string& urdfFile, const std::string& referenceFile, bool centroidal = false) {
  return swingConfigFromImage(buildProblemImage(taskFile, urdfFile, referenceFile, centroidal));
}

}  // namespace hsqp_host
