// =====================================================================================
// TEST INFRASTRUCTURE — C entry points over the reference's ASSEMBLY of rigid-body quantities, compiled from /root/reference in place
// (oracle/Makefile, target _ref/libref_model.so) against a MOCK of Pinocchio (ref_stubs_model/pinocchio/mock.hpp: every "algorithm" returns what
// the caller handed in) and number-wrapper CppAD base classes (ref_stubs_model/ocs2_core/…).  Pinocchio and CppAD are absent from /root/reference
// and from this image, so the rigid-body model itself stays unpinned; what these entry points pin:
//   a2   humanoid_wb_mpc/src/dynamics/DynamicsHelperFunctions.cpp:52-134 (computeBaseAcceleration over crba / nonLinearEffects / the two foot
//        Jacobians, computeGeneralizedAccelerations, computeStateDerivative) + humanoid_common_mpc/src/pinocchio_model/DynamicsHelperFunctions.cpp:
//        197-218 (the block-diagonal base solve): the flow map as a function of (M, nle, J_l, J_r) — the oracle's own M / nle / Jacobians handed in
//   a13  humanoid_common_mpc/src/constraint/ContactMomentXYConstraintCppAd.cpp:84-104: the four rows, which bound pairs with which moment, the signs
//   a15  humanoid_common_mpc/src/constraint/FootCollisionConstraint.cpp:80-86 (active unless BOTH feet are in contact), :92-144 (the 16 pairs, 2 r)
//   a7   humanoid_wb_mpc/src/cost/EndEffectorDynamicsFootCost.cpp:91-124: order of the 18 errors (three zeros first), references subtracted,
//        sqrt-weights and impact scaler applied — around an orientation term that is the oracle's ASSUMPTION A2 on both sides (fork-only source)
// Only tests/ and tests/golden/make_ref_model_golden.py load the library.
// =====================================================================================
#include <cstring>
#include <memory>

#include "humanoid_common_mpc/constraint/ContactMomentXYConstraintCppAd.h"
#include "humanoid_common_mpc/constraint/FootCollisionConstraint.h"
#include "humanoid_common_mpc/reference_manager/SwitchedModelReferenceManager.h"
#include "humanoid_wb_mpc/common/WBAccelMpcRobotModel.h"
#include "humanoid_wb_mpc/cost/EndEffectorDynamicsFootCost.h"
#include "humanoid_wb_mpc/dynamics/DynamicsHelperFunctions.h"

using namespace ocs2;
using namespace ocs2::humanoid;

namespace {
struct SettingsArgs { int nj = 23; } g_args;
}

namespace ocs2::humanoid {
// declared by the reference's headers, defined in files that need Boost / Pinocchio for real: never called by this driver, or (the ModelSettings
// constructor, whose real body parses the URDF through Pinocchio) given the few fields the compiled code reads
ModeSchedule loadModeSchedule(const std::string&, const std::string&, bool) { throw std::runtime_error("stand-in"); }
ModeSequenceTemplate loadModeSequenceTemplate(const std::string&, const std::string&, bool) { throw std::runtime_error("stand-in"); }
std::ostream& operator<<(std::ostream& stream, const ModeSequenceTemplate&) { return stream; }
ModelSettings::ModelSettings(const std::string&, const std::string&, const std::string&, bool) {
  mpc_joint_dim = static_cast<size_t>(g_args.nj);
  full_joint_dim = mpc_joint_dim;
  phaseTransitionStanceTime = 0.4;
  contactNames6DoF = {"foot_l_contact", "foot_r_contact"};   // robot_models/unitree_g1/g1_wb_mpc/config/mpc/task.info:30-33
  contactNames = contactNames6DoF;
}
}  // namespace ocs2::humanoid

namespace {
const char* kFrames[10] = {"ankle_l", "ankle_r", "foot_l_contact", "foot_r_contact", "foot_l_contact_collision_p_1", "foot_r_contact_collision_p_1",
                           "foot_l_contact_collision_p_2", "foot_r_contact_collision_p_2", "knee_l", "knee_r"};   // order of hsqp's collision points
struct World {
  explicit World(int nj) : settings((g_args.nj = nj, std::string()), "", ""), model(settings), adModel(settings) {}
  ModelSettings settings;
  WBAccelMpcRobotModel<scalar_t> model;
  WBAccelMpcRobotModel<ad_scalar_t> adModel;
};
vector_t to_vec(const double* p, int n) { vector_t v(n); for (int i = 0; i < n; ++i) v(i) = p[i]; return v; }
class NoPreComp final : public PreComputation {};
std::unique_ptr<SwitchedModelReferenceManager> make_manager(World& w, int n_events, const double* event_times, const int* mode_sequence) {
  const ModeSchedule ms(std::vector<scalar_t>(event_times, event_times + n_events), std::vector<size_t>(mode_sequence, mode_sequence + n_events + 1));
  auto gait = std::make_shared<GaitSchedule>(ms, ModeSequenceTemplate({0.0, 0.5}, {STANCE}), 0.4);
  SwingTrajectoryPlanner::Config c;
  auto swing = std::make_shared<SwingTrajectoryPlanner>(c, 2);
  auto mgr = std::make_unique<SwitchedModelReferenceManager>(gait, swing, PinocchioInterface(), w.model);
  mgr->setModeSchedule(ms);
  return mgr;
}
// a mock Pinocchio interface with the ten named frames (+ handed-in placements) and room for M / nle / the two foot Jacobians
PinocchioInterface make_interface(int nv) {
  pinocchio::Model m;
  m.nq = nv; m.nv = nv; m.njoints = nv - 4;
  for (const char* n : kFrames) { pinocchio::FrameTpl<scalar_t> f; f.name = n; m.frames.push_back(f); }
  pinocchio::Data d;
  d.oMf.resize(10); d.oMi.resize(1); d.fv.resize(10); d.fa.resize(10); d.J.resize(10);
  { pinocchio::Data::Mat z; z.resize(nv, nv); d.M_in = z; d.M = z; }
  { pinocchio::Data::Vec z; z.resize(nv, 1); d.nle_in = z; d.nle = z; d.tau = z; }
  for (auto& j : d.J) { pinocchio::Data::Mat z; z.resize(6, nv); j = z; }
  return PinocchioInterface(m, d);
}
void set_dynamics(PinocchioInterface& pin, int nv, const double* M, const double* nle, const double* Jl, const double* Jr) {
  auto& d = pin.getData();
  for (int i = 0; i < nv; ++i) { d.nle_in(i) = nle[i]; for (int j = 0; j < nv; ++j) d.M_in(i, j) = M[i * nv + j]; }
  for (int r = 0; r < 6; ++r) for (int c = 0; c < nv; ++c) { d.J[2](r, c) = Jl[r * nv + c]; d.J[3](r, c) = Jr[r * nv + c]; }
}
// EndEffectorDynamics<scalar_t> is only cloned by the foot cost (its kinematics come from the mock Pinocchio): an implementation that refuses to be used
class UnusedEeDynamics final : public EndEffectorDynamics<scalar_t> {
 public:
  UnusedEeDynamics* clone() const override { return new UnusedEeDynamics(*this); }
  const std::vector<std::string>& getIds() const override { return ids_; }
  std::vector<vector3_t> getPosition(const vector_t&) const override { throw std::runtime_error("unused"); }
  std::vector<vector3_t> getVelocity(const vector_t&, const vector_t&) const override { throw std::runtime_error("unused"); }
  std::vector<vector3_t> getOrientationErrorWrtPlane(const vector_t&, const std::vector<vector3_t>&) const override { throw std::runtime_error("unused"); }
  std::vector<vector6_t> getTwist(const vector_t&, const vector_t&) const override { throw std::runtime_error("unused"); }
  std::vector<vector3_t> getLinearAcceleration(const vector_t&, const vector_t&) const override { throw std::runtime_error("unused"); }
  std::vector<vector3_t> getAngularAcceleration(const vector_t&, const vector_t&) const override { throw std::runtime_error("unused"); }
  std::vector<vector6_t> getAccelerations(const vector_t&, const vector_t&) const override { throw std::runtime_error("unused"); }
  std::vector<VectorFunctionLinearApproximation> getPositionLinearApproximation(const vector_t&) const override { throw std::runtime_error("unused"); }
  std::vector<VectorFunctionLinearApproximation> getVelocityLinearApproximation(const vector_t&, const vector_t&) const override { throw std::runtime_error("unused"); }
  std::vector<VectorFunctionLinearApproximation> getOrientationErrorWrtPlaneLinearApproximation(const vector_t&, const std::vector<vector3_t>&) const override { throw std::runtime_error("unused"); }
  std::vector<VectorFunctionLinearApproximation> getTwistLinearApproximation(const vector_t&, const vector_t&) const override { throw std::runtime_error("unused"); }
  std::vector<VectorFunctionLinearApproximation> getLinearAccelerationLinearApproximation(const vector_t&, const vector_t&) const override { throw std::runtime_error("unused"); }
  std::vector<VectorFunctionLinearApproximation> getAngularAccelerationLinearApproximation(const vector_t&, const vector_t&) const override { throw std::runtime_error("unused"); }
  std::vector<VectorFunctionLinearApproximation> getAccelerationsLinearApproximation(const vector_t&, const vector_t&) const override { throw std::runtime_error("unused"); }
 private:
  std::vector<std::string> ids_{"foot"};
};
}  // namespace

extern "C" {

// a2: xdot[2 nv] = the reference's computeStateDerivative(state, input) with CRBA's M (nv x nv row-major: the mock's crba copies its UPPER triangle into the
// zero-filled data.M, as Pinocchio does), nonLinearEffects' nle (nv) and the LOCAL_WORLD_ALIGNED Jacobians of the two contact frames (6 x nv row-major,
// rows = linear then angular) handed in; ab[6] = its computeBaseAcceleration
int refm_state_derivative(int nj, const double* M, const double* nle, const double* Jl, const double* Jr, const double* x, const double* u, double* xdot, double* ab) {
  try {
    World w(nj);
    const int nv = 6 + nj;
    PinocchioInterface pin = make_interface(nv);
    set_dynamics(pin, nv, M, nle, Jl, Jr);
    const vector_t xs = to_vec(x, (int)w.model.getStateDim()), us = to_vec(u, (int)w.model.getInputDim());
    const vector_t xd = computeStateDerivative<scalar_t>(xs, us, pin, w.model);
    const vector6_t a = computeBaseAcceleration<scalar_t>(xs, us, pin, w.model);
    for (int i = 0; i < 2 * nv; ++i) xdot[i] = xd(i);
    for (int i = 0; i < 6; ++i) ab[i] = a(i);
    return 0;
  } catch (const std::exception&) { return 1; }
}

// a15: FootCollisionConstraint on handed-in frame positions pos[10][3] (the order of kFrames = hsqp's collision points), radii = {foot, knee};
// h[16] = constraintFunction, active = isActive(time) on the given mode schedule, n = getNumConstraints
int refm_foot_collision(int nj, const double* pos, const double* radii, int n_events, const double* event_times, const int* mode_sequence, double time, const double* x,
                        double* h, int* active, int* n) {
  try {
    World w(nj);
    const int nv = 6 + nj;
    PinocchioInterface pin = make_interface(nv);
    for (int p = 0; p < 10; ++p) for (int k = 0; k < 3; ++k) pin.getData().oMf[p].t(k) = pos[3 * p + k];
    auto mgr = make_manager(w, n_events, event_times, mode_sequence);
    FootCollisionConstraint::Config cfg;
    cfg.leftAnkleFrame = kFrames[0]; cfg.rightAnkleFrame = kFrames[1]; cfg.leftKneeFrame = kFrames[8]; cfg.rightKneeFrame = kFrames[9];
    cfg.footCollisionSphereRadius = radii[0]; cfg.kneeCollisionSphereRadius = radii[1];
    FootCollisionConstraint con(*mgr, pin, w.adModel, cfg, "footCollision", w.settings);
    const vector_t v = con.evaluate(time, to_vec(x, (int)w.model.getStateDim()), con.getParameters(time, NoPreComp()));
    *n = (int)con.getNumConstraints(time);
    if (v.size() != 16) return 2;
    for (int i = 0; i < 16; ++i) h[i] = v(i);
    *active = con.isActive(time) ? 1 : 0;
    return 0;
  } catch (const std::exception&) { return 1; }
}

// a13: ContactMomentXYConstraintCppAd of contact `contact` on a handed-in contact-frame rotation Rf (3 x 3 row-major, local -> world),
// rect = {x_min, x_max, y_min, y_max}; h[4], active
int refm_contact_moment(int nj, int contact, const double* Rf, const double* rect, int n_events, const double* event_times, const int* mode_sequence, double time,
                        const double* x, const double* u, double* h, int* active) {
  try {
    World w(nj);
    const int nv = 6 + nj;
    PinocchioInterface pin = make_interface(nv);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) pin.getData().oMf[2 + contact].R(i, j) = Rf[3 * i + j];
    auto mgr = make_manager(w, n_events, event_times, mode_sequence);
    const ContactRectangle rectangle(PolygonBounds(rect[0], rect[1], rect[2], rect[3]), ContactCenterPoint(kFrames[2 + contact], "ankle_roll", vector3_t(0.0, 0.0, 0.0)));
    ContactMomentXYConstraintCppAd con(*mgr, rectangle, contact, pin, w.adModel, "contactMomentXY", w.settings);
    const vector_t v = con.evaluate(time, to_vec(x, (int)w.model.getStateDim()), to_vec(u, (int)w.model.getInputDim()), vector_t(0));
    if (v.size() != 4) return 2;
    for (int i = 0; i < 4; ++i) h[i] = v(i);
    *active = con.isActive(time) ? 1 : 0;
    return 0;
  } catch (const std::exception&) { return 1; }
}

// a7: residual r[18] of EndEffectorDynamicsFootCost::costVectorFunction of contact `contact` on the handed-in frame quantities of its contact frame
// (rotation R 3 x 3 row-major, LOCAL_WORLD_ALIGNED linear / angular velocity and classical acceleration), parameters[37] = {reference (18), sqrt weights
// (18), impact proximity scaler}
int refm_foot_cost(int nj, int contact, const double* R, const double* vlin, const double* vang, const double* alin, const double* aang, const double* params37,
                   const double* x, const double* u, double* r) {
  try {
    World w(nj);
    const int nv = 6 + nj;
    PinocchioInterface pin = make_interface(nv);
    auto& d = pin.getData();
    for (int i = 0; i < nv; ++i) d.M_in(i, i) = 1.0;   // (the cost evaluates the generalized accelerations and hands them to the mock's forwardKinematics: unused)
    const int fid = 2 + contact;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) d.oMf[fid].R(i, j) = R[3 * i + j];
      d.fv[fid].lin(i) = vlin[i]; d.fv[fid].ang(i) = vang[i]; d.fa[fid].lin(i) = alin[i]; d.fa[fid].ang(i) = aang[i];
    }
    const double ev[1] = {1e9};
    const int modes[2] = {3, 3};
    auto mgr = make_manager(w, 1, ev, modes);
    EndEffectorDynamicsWeights weights;
    UnusedEeDynamics ee;
    EndEffectorDynamicsFootCost cost(*mgr, weights, pin, ee, w.adModel, contact, "footCost", w.settings);
    const vector_t v = cost.evaluate(0.0, to_vec(x, (int)w.model.getStateDim()), to_vec(u, (int)w.model.getInputDim()), to_vec(params37, 37));
    if (v.size() != 18) return 2;
    for (int i = 0; i < 18; ++i) r[i] = v(i);
    return 0;
  } catch (const std::exception&) { return 1; }
}

}  // extern "C"
