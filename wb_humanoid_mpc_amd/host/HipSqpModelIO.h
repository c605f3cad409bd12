// hsqp_model_desc from the exported problem image WITHOUT Python: reads wb_humanoid_mpc_amd/data/g1_wb.json / g1_centroidal.json (what
// tools/export_g1_model.py writes from the reference's URDF + task.info + reference.info + gait.info) into the POD the C ABI takes.
// The reference builds the same constants in WBMpcInterface / createPinocchioModel from (taskFile, urdfFile, referenceFile)
// (humanoid_nmpc/humanoid_wb_mpc/include/humanoid_wb_mpc/WBMpcInterface.h:73-75, humanoid_common_mpc/src/pinocchio_model/
// createPinocchioModel.cpp:60-182); a drop-in main() links this header instead of embedding Python.  No dependency beyond the C++17
// standard library: the JSON subset of the image (objects, arrays, numbers, strings, true / false / null) is parsed here.
#pragma once
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/hsqp.h"

namespace hsqp_host {

struct JsonValue {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  double num = 0.0;
  bool b = false;
  std::string str;
  std::vector<JsonValue> arr;
  std::map<std::string, JsonValue> obj;
  const JsonValue& at(const std::string& k) const {
    auto it = obj.find(k);
    if (kind != Object || it == obj.end()) throw std::runtime_error("[HipSqpModelIO] missing key '" + k + "'");
    return it->second;
  }
  bool has(const std::string& k) const { return kind == Object && obj.count(k) > 0; }
  const JsonValue& at(size_t i) const { if (kind != Array || i >= arr.size()) throw std::runtime_error("[HipSqpModelIO] array index out of range"); return arr[i]; }
  double number() const { if (kind != Number) throw std::runtime_error("[HipSqpModelIO] number expected"); return num; }
  double number_or(double dflt) const { return kind == Number ? num : dflt; }
  size_t size() const { return kind == Array ? arr.size() : obj.size(); }
};

class JsonParser {
 public:
  explicit JsonParser(std::string text) : s_(std::move(text)) {}
  JsonValue parse() { JsonValue v = value(); ws(); if (i_ != s_.size()) fail("trailing characters"); return v; }

 private:
  [[noreturn]] void fail(const std::string& m) const { throw std::runtime_error("[HipSqpModelIO] JSON: " + m + " at offset " + std::to_string(i_)); }
  void ws() { while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\n' || s_[i_] == '\t' || s_[i_] == '\r')) ++i_; }
  bool eat(char c) { ws(); if (i_ < s_.size() && s_[i_] == c) { ++i_; return true; } return false; }
  JsonValue value() {
    ws();
    if (i_ >= s_.size()) fail("unexpected end");
    const char c = s_[i_];
    JsonValue v;
    if (c == '{') {
      ++i_; v.kind = JsonValue::Object;
      if (eat('}')) return v;
      do { ws(); const JsonValue k = value(); if (k.kind != JsonValue::String) fail("string key expected"); if (!eat(':')) fail("':' expected"); v.obj[k.str] = value(); } while (eat(','));
      if (!eat('}')) fail("'}' expected");
    } else if (c == '[') {
      ++i_; v.kind = JsonValue::Array;
      if (eat(']')) return v;
      do { v.arr.push_back(value()); } while (eat(','));
      if (!eat(']')) fail("']' expected");
    } else if (c == '"') {
      ++i_; v.kind = JsonValue::String;
      while (i_ < s_.size() && s_[i_] != '"') { if (s_[i_] == '\\' && i_ + 1 < s_.size()) ++i_; v.str.push_back(s_[i_++]); }
      if (i_ >= s_.size()) fail("unterminated string");
      ++i_;
    } else if (s_.compare(i_, 4, "true") == 0) { i_ += 4; v.kind = JsonValue::Bool; v.b = true; }
    else if (s_.compare(i_, 5, "false") == 0) { i_ += 5; v.kind = JsonValue::Bool; }
    else if (s_.compare(i_, 4, "null") == 0) { i_ += 4; }
    else {
      size_t n = 0;
      try { v.num = std::stod(s_.substr(i_, 64), &n); } catch (const std::exception&) { fail("number expected"); }
      v.kind = JsonValue::Number; i_ += n;
    }
    return v;
  }
  std::string s_;
  size_t i_ = 0;
};

inline JsonValue loadJsonFile(const std::string& path) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("[HipSqpModelIO] cannot open " + path);
  std::stringstream ss;
  ss << in.rdbuf();
  return JsonParser(ss.str()).parse();
}

namespace detail {
template <size_t N> void fill(double (&dst)[N], const JsonValue& a, bool sqrt_of = false, size_t pad_to = N) {
  if (a.size() > N || pad_to != N) throw std::runtime_error("[HipSqpModelIO] array length mismatch");
  for (size_t i = 0; i < N; ++i) dst[i] = i < a.size() ? (sqrt_of ? std::sqrt(a.at(i).number()) : a.at(i).number()) : 0.0;
}
template <size_t N> void fill_exact(double (&dst)[N], const JsonValue& a, bool sqrt_of = false) {
  if (a.size() != N) throw std::runtime_error("[HipSqpModelIO] array length mismatch (" + std::to_string(a.size()) + " != " + std::to_string(N) + ")");
  fill(dst, a, sqrt_of);
}
inline void frame(hsqp_frame& f, const JsonValue& v) { f.body = (int32_t)v.at("body").number(); fill_exact(f.p, v.at("p")); }
inline void barrier(hsqp_barrier& b, const JsonValue& v) { b.mu = v.at("barrier_mu").number(); b.delta = v.at("barrier_delta").number(); }
}  // namespace detail

/** The problem image (JSON) -> hsqp_model_desc, field for field as wb_humanoid_mpc_amd/model.py builds it (tests/test_host_cpp.py: bit-identical). */
inline hsqp_model_desc modelDescFromImage(const JsonValue& d) {
  hsqp_model_desc m;
  std::memset(&m, 0, sizeof(m));
  const bool cent = d.has("formulation") && d.at("formulation").str == "centroidal";
  m.formulation = cent ? HSQP_FORM_CENTROIDAL : HSQP_FORM_WB;
  m.n_joints = (int32_t)d.at("nj").number();
  if (m.n_joints != HSQP_NJ || d.at("bodies").size() != HSQP_NB) throw std::runtime_error("[HipSqpModelIO] the library is built for 23 joints / 24 bodies");
  for (size_t i = 0; i < HSQP_NB; ++i) {
    const JsonValue& b = d.at("bodies").at(i);
    hsqp_body& mb = m.bodies[i];
    mb.parent = (int32_t)b.at("parent").number();
    detail::fill_exact(mb.R, b.at("R")); detail::fill_exact(mb.p, b.at("p")); detail::fill_exact(mb.axis, b.at("axis"));
    mb.mass = b.at("mass").number();
    detail::fill_exact(mb.com, b.at("com")); detail::fill_exact(mb.inertia, b.at("inertia"));
    mb.q_lo = b.at("lo").number_or(0.0); mb.q_hi = b.at("hi").number_or(0.0);
  }
  const JsonValue& fr = d.at("frames");
  for (int i = 0; i < 2; ++i) {
    detail::frame(m.contact[i], fr.at("contact").at(i)); detail::frame(m.collision_p1[i], fr.at("collision_p1").at(i));
    detail::frame(m.collision_p2[i], fr.at("collision_p2").at(i)); detail::frame(m.ankle[i], fr.at("ankle").at(i)); detail::frame(m.knee[i], fr.at("knee").at(i));
  }
  m.gravity = d.at("gravity").number();
  detail::fill(m.Q, d.at("Q")); detail::fill_exact(m.R, d.at("R")); detail::fill(m.Qf, d.at("Qf"));   // centroidal: 35 weights, zero padded
  if (cent) {
    const JsonValue& t = d.at("torso");
    m.torso.body = (int32_t)t.at("body").number();
    detail::fill_exact(m.torso.p, t.at("p")); detail::fill_exact(m.torso_R, t.at("R")); detail::fill_exact(m.torso_sqrt_w, t.at("weights"), true);
    detail::fill_exact(m.cent_foot_sqrt_w, d.at("cent_foot_cost_weights"), true);
    for (int f = 0; f < 2; ++f) {
      detail::fill_exact(m.ext_torque_sqrt_w[f], d.at("ext_torque").at("weights").at(f), true);
      for (int k = 0; k < 6; ++k) m.ext_torque_joint[f][k] = (int32_t)d.at("ext_torque").at("joints").at(f).at(k).number();
    }
  }
  detail::fill_exact(m.foot_sqrt_w, d.at("foot_cost_weights"), true);
  const JsonValue& fc = d.at("foot_constraint");
  m.gain_pos_z = fc.at("positionErrorGain_z").number(); m.gain_ori = fc.at("orientationErrorGain").number();
  m.gain_linvel_z = fc.at("linearVelocityErrorGain_z").number(); m.gain_linvel_xy = fc.at("linearVelocityErrorGain_xy").number();
  m.gain_angvel = fc.at("angularVelocityErrorGain").number(); m.gain_linacc_z = fc.at("linearAccelerationErrorGain_z").number();
  m.gain_linacc_xy = fc.at("linearAccelerationErrorGain_xy").number(); m.gain_angacc = fc.at("angularAccelerationErrorGain").number();
  const JsonValue& f = d.at("friction");
  m.friction_mu = f.at("mu").number(); m.friction_reg = f.at("regularization").number(); m.friction_grip = f.at("gripper_force").number();
  m.friction_hess_shift = f.at("hessian_diagonal_shift").number();
  detail::barrier(m.friction_barrier, f);
  const JsonValue& r = d.at("contact_rectangle");
  m.rect_x_min = r.at("x_min").number(); m.rect_x_max = r.at("x_max").number(); m.rect_y_min = r.at("y_min").number(); m.rect_y_max = r.at("y_max").number();
  detail::barrier(m.moment_barrier, d.at("moment_xy"));
  detail::barrier(m.joint_limit_barrier, d.at("joint_limits"));
  const JsonValue& c = d.at("collision");
  m.r_foot = c.at("r_foot").number(); m.r_knee = c.at("r_knee").number();
  detail::barrier(m.collision_barrier, c);
  for (int k = 0; k < 4; ++k) m.arm_swing_joint[k] = (int32_t)d.at("arm_swing_joints").at(k).number();
  return m;
}

inline hsqp_model_desc loadModelDesc(const std::string& jsonPath) { return modelDescFromImage(loadJsonFile(jsonPath)); }

/** task.info swing_trajectory_config of the same image -> hsqp_swing_config (SwingTrajectoryPlanner::Config). */
inline hsqp_swing_config swingConfigFromImage(const JsonValue& image) {
  const JsonValue& s = image.at("swing");
  hsqp_swing_config c;
  c.lift_off_velocity = s.at("liftOffVelocity").number(); c.touch_down_velocity = s.at("touchDownVelocity").number(); c.swing_height = s.at("swingHeight").number();
  c.touch_down_height_offset = s.at("touchDownHeightOffset").number(); c.swing_time_scale = s.at("swingTimeScale").number();
  c.impact_mid = s.at("impactProximityFactorMidPointValue").number(); c.impact_lift_velocity = s.at("impactProximityFactorLiftOffVelocity").number();
  c.impact_touch_velocity = s.at("impactProximityFactorTouchDownVelocity").number();
  return c;
}
inline hsqp_swing_config loadSwingConfig(const std::string& jsonPath) { return swingConfigFromImage(loadJsonFile(jsonPath)); }

}  // namespace hsqp_host
