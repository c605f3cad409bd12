// =====================================================================================
// TEST INFRASTRUCTURE — C entry points over more of the REFERENCE'S OWN code, compiled from /root/reference in place (oracle/Makefile,
// target _ref/libref_terms.so) against the stand-in third-party headers of oracle/ref_stubs/ (a minimal Eigen, ocs2 interfaces, Boost
// property tree).  Only tests/ and tests/golden/make_ref_terms_golden.py load the library.  What it pins:
//   a1   humanoid_wb_mpc/include/humanoid_wb_mpc/common/WBAccelMpcRobotModel.h:47-245       state / input layout and accessors
//   a5   humanoid_common_mpc/src/reference_manager/SwitchedModelReferenceManager.cpp:55-152  getContactFlags, getPhaseVariable,
//        getDesiredState (arm-swing reference on the CURRENT yaw), modifyReferences (gait schedule window + swing planner update)
//   a11  humanoid_common_mpc/src/constraint/ZeroWrenchConstraint.cpp:59-84                  value, dfdu
//   a12  humanoid_common_mpc/src/constraint/FrictionForceConeConstraint.cpp:78-224          value, dfdu, dfduu, dfdxx (diagonal shift)
//   a7   humanoid_wb_mpc/src/cost/EndEffectorDynamicsCostHelpers.cpp:41-108                 EndEffectorDynamicsWeights::getWeights (the weight
//        overwrite quirk) + toVector, read from the reference's own task.info through the stand-in INFO parser
//   target generator (feeds a5)  humanoid_wb_mpc/src/command/WBMpcTargetTrajectoriesCalculator.cpp:80-136 + humanoid_common_mpc/src/command/
//        TargetTrajectoriesCalculatorBase.cpp:41-160: commandedVelocityToTargetTrajectories on the reference's own reference.info
//   ---- round 4: the ASSEMBLY files (how kinematics, gains and references are combined), over an EndEffectorDynamics whose kinematics the
//        caller hands in (the reference's only implementation of that interface is CppAD-generated)
//   a9   humanoid_wb_mpc/src/constraint/EndEffectorDynamicsAccelerationsConstraint.cpp:97-146 (getValue, getLinearApproximation) behind
//        ZeroAccelerationConstraintCppAd.cpp:60-87 (isActive = contact flag), config as WBMpcInterface.cpp:204-229 builds it
//   a10  humanoid_wb_mpc/src/constraint/EndEffectorDynamicsLinearAccConstraint.cpp:78-127, config as WBMpcPreComputation.cpp:91-104 builds it
//   a5   humanoid_common_mpc/src/cost/StateInputQuadraticCost.cpp:67-78 (state - x_nom(t), input - weight compensation on the contact flags)
//   a14  humanoid_common_mpc/src/constraint/JointLimitsSoftConstraint.cpp:64-100 with the stand-in penalty (everything but ASSUMPTION A1)
//   a18  humanoid_common_mpc/src/initialization/WeightCompInitializer.cpp:66-70
// =====================================================================================
#include <cstring>

#include "humanoid_common_mpc/constraint/JointLimitsSoftConstraint.h"
#include "humanoid_common_mpc/cost/StateInputQuadraticCost.h"
#include "humanoid_common_mpc/initialization/WeightCompInitializer.h"
#include "humanoid_wb_mpc/constraint/EndEffectorDynamicsLinearAccConstraint.h"
#include "humanoid_wb_mpc/constraint/ZeroAccelerationConstraintCppAd.h"

#include "humanoid_common_mpc/constraint/FrictionForceConeConstraint.h"
#include "humanoid_common_mpc/constraint/ZeroWrenchConstraint.h"
#include "humanoid_common_mpc/reference_manager/SwitchedModelReferenceManager.h"
#include "humanoid_wb_mpc/command/WBMpcTargetTrajectoriesCalculator.h"
#include "humanoid_wb_mpc/common/WBAccelMpcRobotModel.h"
#include "humanoid_wb_mpc/cost/EndEffectorDynamicsCostHelpers.h"

using namespace ocs2;
using namespace ocs2::humanoid;

namespace {
struct SettingsArgs { int nj = 23; int arm[4] = {0, 0, 0, 0}; } g_args;
}

namespace ocs2::humanoid {
// declared by the reference's headers, defined in files that need Boost / Pinocchio for real: never called by this driver, or (the
// ModelSettings constructor, whose real body parses the URDF through Pinocchio) given the few fields the compiled code reads
ModeSchedule loadModeSchedule(const std::string&, const std::string&, bool) { throw std::runtime_error("stand-in"); }
ModeSequenceTemplate loadModeSequenceTemplate(const std::string&, const std::string&, bool) { throw std::runtime_error("stand-in"); }
std::ostream& operator<<(std::ostream& stream, const ModeSequenceTemplate&) { return stream; }
ModelSettings::ModelSettings(const std::string&, const std::string&, const std::string&, bool) {
  mpc_joint_dim = static_cast<size_t>(g_args.nj);
  full_joint_dim = mpc_joint_dim;
  j_l_shoulder_y_index = g_args.arm[0]; j_r_shoulder_y_index = g_args.arm[1]; j_l_elbow_y_index = g_args.arm[2]; j_r_elbow_y_index = g_args.arm[3];
  phaseTransitionStanceTime = 0.4;
}
}  // namespace ocs2::humanoid

namespace ocs2::humanoid {
scalar_t& ref_stub_total_mass() { static scalar_t m = 0.0; return m; }   // (see ref_stubs/humanoid_common_mpc/pinocchio_model/DynamicsHelperFunctions.h)
}

namespace {
struct World {
  World(int nj, const int arm[4]) : settings((g_args.nj = nj, std::memcpy(g_args.arm, arm, sizeof(g_args.arm)), std::string()), "", ""), model(settings) {}
  ModelSettings settings;
  WBAccelMpcRobotModel<scalar_t> model;
};
vector_t to_vec(const double* p, int n) { vector_t v(n); for (int i = 0; i < n; ++i) v(i) = p[i]; return v; }
class NoPreComp final : public PreComputation {};

std::unique_ptr<SwitchedModelReferenceManager> make_manager(World& w, int n_events, const double* event_times, const int* mode_sequence) {
  // a GaitSchedule whose getModeSchedule(...) returns exactly the given schedule on the window the manager asks for: the initial mode
  // schedule IS the given one (no template is inserted; the default template only tiles beyond its last event)
  auto gait = std::make_shared<GaitSchedule>(ModeSchedule(std::vector<scalar_t>(event_times, event_times + n_events), std::vector<size_t>(mode_sequence, mode_sequence + n_events + 1)),
                                             ModeSequenceTemplate({0.0, 0.5}, {STANCE}), 0.4);
  SwingTrajectoryPlanner::Config c;
  auto swing = std::make_shared<SwingTrajectoryPlanner>(c, 2);
  return std::make_unique<SwitchedModelReferenceManager>(gait, swing, PinocchioInterface(), w.model);
}
}  // namespace

extern "C" {

// out[12] = state dim, input dim, base start, joint start, joint-velocity start, gen. coordinates dim, wrench start of contacts 0 / 1, force
// start 0 / 1, moment start 0 / 1
void ref_wb_layout(int nj, int* out) {
  const int arm[4] = {0, 0, 0, 0};
  World w(nj, arm);
  out[0] = (int)w.model.getStateDim(); out[1] = (int)w.model.getInputDim(); out[2] = (int)w.model.getBaseStartindex(); out[3] = (int)w.model.getJointStartindex();
  out[4] = (int)w.model.getJointVelocitiesStartindex(); out[5] = (int)w.model.getGenCoordinatesDim();
  for (int c = 0; c < 2; ++c) { out[6 + c] = (int)w.model.getContactWrenchStartIndices(c); out[8 + c] = (int)w.model.getContactForceStartIndices(c); out[10 + c] = (int)w.model.getContactMomentStartIndices(c); }
}

// the accessors on a state / input pair: out = [basePose 6 | jointAngles nj | baseComLinearVelocity 3 | baseComVelocity 6 | jointVelocities nj |
//                                               generalized coordinates 6+nj | generalized velocities 6+nj | wrench0 6 | wrench1 6 | force0 3 | moment1 3]
void ref_wb_accessors(int nj, const double* x, const double* u, double* out) {
  const int arm[4] = {0, 0, 0, 0};
  World w(nj, arm);
  const vector_t xs = to_vec(x, (int)w.model.getStateDim()), us = to_vec(u, (int)w.model.getInputDim());
  int o = 0;
  auto put = [&](const Eigen::Dyn<scalar_t>& v) { for (Eigen::Index i = 0; i < v.size(); ++i) out[o++] = v(i); };
  put(w.model.getBasePose(xs)); put(w.model.getJointAngles(xs)); put(w.model.getBaseComLinearVelocity(xs)); put(w.model.getBaseComVelocity(xs));
  put(w.model.getJointVelocities(xs, us)); put(w.model.getGeneralizedCoordinates(xs)); put(w.model.getGeneralizedVelocities(xs, us));
  put(w.model.getContactWrench(us, 0)); put(w.model.getContactWrench(us, 1)); put(w.model.getContactForce(us, 0)); put(w.model.getContactMoment(us, 1));
}

// FrictionForceConeConstraint of contact `contact`: cfg = {frictionCoefficient, regularization, gripperForce, hessianDiagonalShift}.
// f[1], dfdu[nu], dfduu[nu*nu], dfdxx_diag[nx] (its dfdxx is diagonal: the shift), active = isActive(time) on the given mode schedule.
int ref_friction_cone(int nj, const double cfg[4], int contact, int n_events, const double* event_times, const int* mode_sequence, double time, const double* x,
                      const double* u, double* f, double* dfdu, double* dfduu, double* dfdxx_diag, int* active) {
  try {
    const int arm[4] = {0, 0, 0, 0};
    World w(nj, arm);
    auto mgr = make_manager(w, n_events, event_times, mode_sequence);
    mgr->setModeSchedule(ModeSchedule(std::vector<scalar_t>(event_times, event_times + n_events), std::vector<size_t>(mode_sequence, mode_sequence + n_events + 1)));
    const FrictionForceConeConstraint::Config c(cfg[0], cfg[1], cfg[2], cfg[3]);
    FrictionForceConeConstraint con(*mgr, c, contact, w.model);
    const int nx = (int)w.model.getStateDim(), nu = (int)w.model.getInputDim();
    const vector_t xs = to_vec(x, nx), us = to_vec(u, nu);
    const auto q = con.getQuadraticApproximation(time, xs, us, NoPreComp());
    const auto l = con.getLinearApproximation(time, xs, us, NoPreComp());
    f[0] = q.f(0);
    if (con.getValue(time, xs, us, NoPreComp())(0) != q.f(0) || l.f(0) != q.f(0)) return 2;
    for (int i = 0; i < nu; ++i) { dfdu[i] = q.dfdu(0, i); if (l.dfdu(0, i) != q.dfdu(0, i)) return 2; }
    for (int i = 0; i < nu; ++i) for (int j = 0; j < nu; ++j) dfduu[i * nu + j] = q.dfduu[0](i, j);
    for (int i = 0; i < nx; ++i) {
      dfdxx_diag[i] = q.dfdxx[0](i, i);
      for (int j = 0; j < nx; ++j) if (i != j && q.dfdxx[0](i, j) != 0.0) return 3;
    }
    for (int i = 0; i < nx; ++i) if (q.dfdx(0, i) != 0.0) return 3;
    *active = con.isActive(time) ? 1 : 0;
    return 0;
  } catch (const std::exception& e) { std::cerr << "ref_friction_cone: " << e.what() << "\n"; return 1; }
}

// ZeroWrenchConstraint of contact `contact`: f[6], dfdu[6*nu] (dfdx is zero), active = isActive(time) (= NOT in contact)
int ref_zero_wrench(int nj, int contact, int n_events, const double* event_times, const int* mode_sequence, double time, const double* x, const double* u,
                    double* f, double* dfdu, int* active) {
  try {
    const int arm[4] = {0, 0, 0, 0};
    World w(nj, arm);
    auto mgr = make_manager(w, n_events, event_times, mode_sequence);
    mgr->setModeSchedule(ModeSchedule(std::vector<scalar_t>(event_times, event_times + n_events), std::vector<size_t>(mode_sequence, mode_sequence + n_events + 1)));
    ZeroWrenchConstraint con(*mgr, contact, w.model);
    const int nx = (int)w.model.getStateDim(), nu = (int)w.model.getInputDim();
    const vector_t xs = to_vec(x, nx), us = to_vec(u, nu);
    const auto l = con.getLinearApproximation(time, xs, us, NoPreComp());
    if ((int)con.getNumConstraints(time) != 6) return 2;
    for (int r = 0; r < 6; ++r) { f[r] = l.f(r); for (int i = 0; i < nu; ++i) dfdu[r * nu + i] = l.dfdu(r, i); }
    for (int r = 0; r < 6; ++r) for (int i = 0; i < nx; ++i) if (l.dfdx(r, i) != 0.0) return 3;
    *active = con.isActive(time) ? 1 : 0;
    return 0;
  } catch (const std::exception& e) { std::cerr << "ref_zero_wrench: " << e.what() << "\n"; return 1; }
}

// SwitchedModelReferenceManager: preSolverRun(t0, tf, state) (-> modifyReferences: gait-schedule window, swing planner update), then at
// `time`: getDesiredState(targets, state, time) -> xnom[nx], getPhaseVariable -> *phase, getContactFlags -> flags[2].
int ref_desired_state(int nj, const int arm[4], int n_events, const double* event_times, const int* mode_sequence, int n_knots, const double* tt,
                      const double* ts, int arm_swing, double t0, double tf, const double* state, double time, double* xnom, double* phase, int* flags) {
  try {
    World w(nj, arm);
    auto mgr = make_manager(w, n_events, event_times, mode_sequence);
    const int nx = (int)w.model.getStateDim();
    TargetTrajectories targets;
    for (int k = 0; k < n_knots; ++k) { targets.timeTrajectory.push_back(tt[k]); targets.stateTrajectory.push_back(to_vec(ts + (size_t)k * nx, nx)); }
    mgr->setTargetTrajectories(targets);
    mgr->setArmSwingReferenceActive(arm_swing != 0);
    const vector_t xs = to_vec(state, nx);
    mgr->preSolverRun(t0, tf, xs);
    const vector_t xn = mgr->getDesiredState(mgr->getTargetTrajectories(), xs, time);
    for (int i = 0; i < nx; ++i) xnom[i] = xn(i);
    *phase = mgr->getPhaseVariable(time);
    const contact_flag_t fl = mgr->getContactFlags(time);
    flags[0] = fl[0]; flags[1] = fl[1];
    return 0;
  } catch (const std::exception& e) { std::cerr << "ref_desired_state: " << e.what() << "\n"; return 1; }
}

// EndEffectorDynamicsWeights::getWeights(taskFile, prefix).toVector() -> w18
int ref_foot_weights(const char* task_file, const char* prefix, double* w18) {
  try {
    EndEffectorDynamicsWeights w = EndEffectorDynamicsWeights::getWeights(task_file, prefix, false);
    const auto v = w.toVector();
    for (int i = 0; i < 18; ++i) w18[i] = v(i);
    return 0;
  } catch (const std::exception& e) { std::cerr << "ref_foot_weights: " << e.what() << "\n"; return 1; }
}

// WBMpcTargetTrajectoriesCalculator(reference.info, model, horizon).commandedVelocityToTargetTrajectories(cmd, t0, x0), called `calls`
// times with the same arguments: the command passes through a first-order filter whose state is a function-local STATIC of the
// reference (TargetTrajectoriesCalculatorBase.cpp:117: it survives across calls and calculators), so the caller chooses how far it
// has converged.  Out: the 3 knot times and 3 x nx knot states of the last call.
int ref_wb_velocity_targets(int nj, const char* reference_info, double horizon, const double cmd[4], double t0, const double* x0, int calls, double* times3,
                            double* states) {
  try {
    const int arm[4] = {0, 0, 0, 0};
    World w(nj, arm);
    WBMpcTargetTrajectoriesCalculator calc(reference_info, w.model, horizon);
    const int nx = (int)w.model.getStateDim();
    const vector_t xs = to_vec(x0, nx);
    Eigen::Matrix<scalar_t, 4, 1> c;
    for (int i = 0; i < 4; ++i) c(i) = cmd[i];
    TargetTrajectories t;
    for (int k = 0; k < calls; ++k) t = calc.commandedVelocityToTargetTrajectories(c, t0, xs);
    if (t.timeTrajectory.size() != 3 || t.stateTrajectory.size() != 3 || t.inputTrajectory.size() != 3) return 2;
    for (int k = 0; k < 3; ++k) {
      times3[k] = t.timeTrajectory[k];
      if ((int)t.stateTrajectory[k].size() != nx || (int)t.inputTrajectory[k].size() != (int)w.model.getInputDim()) return 2;
      for (int i = 0; i < nx; ++i) states[(size_t)k * nx + i] = t.stateTrajectory[k](i);
      for (int i = 0; i < (int)t.inputTrajectory[k].size(); ++i) if (t.inputTrajectory[k](i) != 0.0) return 3;
    }
    return 0;
  } catch (const std::exception& e) { std::cerr << "ref_wb_velocity_targets: " << e.what() << "\n"; return 1; }
}


}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------------------------
// Round 4: the assembly files.
namespace {
// EndEffectorDynamics over kinematics the caller hands in: kin[18] = {position, orientation error wrt the plane, linear velocity, angular
// velocity, linear acceleration, angular acceleration} of ONE end effector, jac[18][nx + nu] their Jacobians wrt (x, u).  State and input
// arguments are ignored: whatever the reference's assembly code returns is a function of these numbers and of its configuration only.
class HandedInEeDynamics final : public EndEffectorDynamics<scalar_t> {
 public:
  HandedInEeDynamics(const double* kin, const double* jac, int nx, int nu) : kin_(kin, kin + 18), jac_(jac, jac + 18 * (nx + nu)), nx_(nx), nu_(nu), ids_{"foot"} {}
  HandedInEeDynamics* clone() const override { return new HandedInEeDynamics(*this); }
  const std::vector<std::string>& getIds() const override { return ids_; }
  std::vector<vector3_t> getPosition(const vector_t&) const override { return {v3(0)}; }
  std::vector<vector3_t> getVelocity(const vector_t&, const vector_t&) const override { return {v3(6)}; }
  std::vector<vector3_t> getOrientationErrorWrtPlane(const vector_t&, const std::vector<vector3_t>& normals) const override { check_normal(normals); return {v3(3)}; }
  std::vector<vector6_t> getTwist(const vector_t&, const vector_t&) const override { return {v6(6)}; }
  std::vector<vector3_t> getLinearAcceleration(const vector_t&, const vector_t&) const override { return {v3(12)}; }
  std::vector<vector3_t> getAngularAcceleration(const vector_t&, const vector_t&) const override { return {v3(15)}; }
  std::vector<vector6_t> getAccelerations(const vector_t&, const vector_t&) const override { return {v6(12)}; }
  std::vector<VectorFunctionLinearApproximation> getPositionLinearApproximation(const vector_t&) const override { return {lin(0, 3, false)}; }
  std::vector<VectorFunctionLinearApproximation> getVelocityLinearApproximation(const vector_t&, const vector_t&) const override { return {lin(6, 3, true)}; }
  std::vector<VectorFunctionLinearApproximation> getOrientationErrorWrtPlaneLinearApproximation(const vector_t&, const std::vector<vector3_t>& normals) const override {
    check_normal(normals);
    return {lin(3, 3, false)};
  }
  std::vector<VectorFunctionLinearApproximation> getTwistLinearApproximation(const vector_t&, const vector_t&) const override { return {lin(6, 6, true)}; }
  std::vector<VectorFunctionLinearApproximation> getLinearAccelerationLinearApproximation(const vector_t&, const vector_t&) const override { return {lin(12, 3, true)}; }
  std::vector<VectorFunctionLinearApproximation> getAngularAccelerationLinearApproximation(const vector_t&, const vector_t&) const override { return {lin(15, 3, true)}; }
  std::vector<VectorFunctionLinearApproximation> getAccelerationsLinearApproximation(const vector_t&, const vector_t&) const override { return {lin(12, 6, true)}; }

 private:
  static void check_normal(const std::vector<vector3_t>& n) {
    if (n.size() != 1 || n[0](0) != 0.0 || n[0](1) != 0.0 || n[0](2) != 1.0) throw std::runtime_error("the ground plane normal the reference passes is (0, 0, 1)");
  }
  vector3_t v3(int o) const { return vector3_t(kin_[o], kin_[o + 1], kin_[o + 2]); }
  vector6_t v6(int o) const { vector6_t v; for (int i = 0; i < 6; ++i) v(i) = kin_[o + i]; return v; }
  // state-only quantities (position, orientation error): dfdu stays EMPTY, as a kinematics-only approximation has no input Jacobian
  VectorFunctionLinearApproximation lin(int o, int n, bool with_u) const {
    VectorFunctionLinearApproximation a;
    a.f = vector_t::Zero(n); a.dfdx = matrix_t::Zero(n, nx_);
    if (with_u) a.dfdu = matrix_t::Zero(n, nu_);
    for (int r = 0; r < n; ++r) {
      a.f(r) = kin_[o + r];
      for (int c = 0; c < nx_; ++c) a.dfdx(r, c) = jac_[(size_t)(o + r) * (nx_ + nu_) + c];
      if (with_u) for (int c = 0; c < nu_; ++c) a.dfdu(r, c) = jac_[(size_t)(o + r) * (nx_ + nu_) + nx_ + c];
    }
    return a;
  }
  std::vector<double> kin_, jac_;
  int nx_, nu_;
  std::vector<std::string> ids_;
};
void put_lin(const VectorFunctionLinearApproximation& l, int rows, int nx, int nu, double* f, double* dfdx, double* dfdu) {
  for (int r = 0; r < rows; ++r) {
    f[r] = l.f(r);
    for (int c = 0; c < nx; ++c) dfdx[r * nx + c] = l.dfdx(r, c);
    for (int c = 0; c < nu; ++c) dfdu[r * nu + c] = l.dfdu(r, c);
  }
}
}  // namespace

extern "C" {

// Stance foot: ZeroAccelerationConstraintCppAd over EndEffectorDynamicsAccelerationsConstraint with the configuration
// WBMpcInterface::getStanceFootConstraint builds from ModelSettings::FootConstraintConfig (WBMpcInterface.cpp:204-229, restated here: that file
// needs Pinocchio).  gains[8] = {positionErrorGain_z, orientationErrorGain, linearVelocityErrorGain_z, linearVelocityErrorGain_xy,
// angularVelocityErrorGain, linearAccelerationErrorGain_z, linearAccelerationErrorGain_xy, angularAccelerationErrorGain} (the struct's order).
// Out: value[6] = getValue, f[6] / dfdx[6 nx] / dfdu[6 nu] = getLinearApproximation, active = isActive(time) on the mode schedule.
int ref_stance_foot_constraint(int nj, const double gains[8], int contact, int n_events, const double* event_times, const int* mode_sequence, double time,
                               const double* kin18, const double* jac, double* value, double* f, double* dfdx, double* dfdu, int* active) {
  try {
    const int arm[4] = {0, 0, 0, 0};
    World w(nj, arm);
    auto mgr = make_manager(w, n_events, event_times, mode_sequence);
    mgr->setModeSchedule(ModeSchedule(std::vector<scalar_t>(event_times, event_times + n_events), std::vector<size_t>(mode_sequence, mode_sequence + n_events + 1)));
    const int nx = (int)w.model.getStateDim(), nu = (int)w.model.getInputDim();
    HandedInEeDynamics ee(kin18, jac, nx, nu);
    EndEffectorDynamicsAccelerationsConstraint::Config config;
    config.b.setZero(6);
    config.Ax.setZero(6, 6);
    config.Av = matrix_t::Identity(6, 6);
    config.Aa = matrix_t::Identity(6, 6);
    if (gains[0] != 0.0) config.Ax(2, 2) = gains[0];
    if (gains[1] != 0.0) config.Ax.block(3, 3, 3, 3) = matrix_t(matrix_t::Identity(3, 3) * gains[1]);
    config.Av.block(0, 0, 2, 2) = matrix_t(matrix_t::Identity(2, 2) * gains[3]);
    config.Av(2, 2) = gains[2];
    config.Av.block(3, 3, 3, 3) = matrix_t(matrix_t::Identity(3, 3) * gains[4]);
    config.Aa.block(0, 0, 2, 2) = matrix_t(matrix_t::Identity(2, 2) * gains[6]);
    config.Aa(2, 2) = gains[5];
    config.Aa.block(3, 3, 3, 3) = matrix_t(matrix_t::Identity(3, 3) * gains[7]);
    ZeroAccelerationConstraintCppAd con(*mgr, ee, contact, config);
    const vector_t xs = vector_t::Zero(nx), us = vector_t::Zero(nu);
    const vector_t v = con.getValue(time, xs, us, NoPreComp());
    const auto l = con.getLinearApproximation(time, xs, us, NoPreComp());
    if ((int)con.getNumConstraints(time) != 6 || v.size() != 6) return 2;
    for (int r = 0; r < 6; ++r) value[r] = v(r);
    put_lin(l, 6, nx, nu, f, dfdx, dfdu);
    *active = con.isActive(time) ? 1 : 0;
    return 0;
  } catch (const std::exception& e) { std::cerr << "ref_stance_foot_constraint: " << e.what() << "\n"; return 1; }
}

// Swing foot: EndEffectorDynamicsLinearAccConstraint (one row) with the configuration WBMpcPreComputation::request builds per node
// (WBMpcPreComputation.cpp:91-104, restated here: that file needs Pinocchio) from the planner's z position / velocity / acceleration
// references zref[3] and the gains (same array as above).  Out: value[1], f[1], dfdx[nx], dfdu[nu].
int ref_swing_foot_constraint(int nj, const double gains[8], const double zref[3], const double* kin18, const double* jac, double* value, double* f, double* dfdx,
                              double* dfdu) {
  try {
    const int arm[4] = {0, 0, 0, 0};
    World w(nj, arm);
    const int nx = (int)w.model.getStateDim(), nu = (int)w.model.getInputDim();
    HandedInEeDynamics ee(kin18, jac, nx, nu);
    EndEffectorDynamicsLinearAccConstraint::Config config;
    config.b = (vector_t(1) << -gains[2] * zref[1]).finished();
    config.Av = matrix_t::Zero(1, 3); config.Av(0, 2) = gains[2];
    config.b(0) -= gains[5] * zref[2];
    config.Aa = matrix_t::Zero(1, 3); config.Aa(0, 2) = gains[5];
    if (gains[0] != 0.0) {
      config.b(0) -= gains[0] * zref[0];
      config.Ax = matrix_t::Zero(1, 3); config.Ax(0, 2) = gains[0];
    }
    EndEffectorDynamicsLinearAccConstraint con(ee, 1);
    con.configure(config);
    const vector_t xs = vector_t::Zero(nx), us = vector_t::Zero(nu);
    const vector_t v = con.getValue(0.0, xs, us, NoPreComp());
    const auto l = con.getLinearApproximation(0.0, xs, us, NoPreComp());
    if (v.size() != 1) return 2;
    value[0] = v(0);
    put_lin(l, 1, nx, nu, f, dfdx, dfdu);
    return 0;
  } catch (const std::exception& e) { std::cerr << "ref_swing_foot_constraint: " << e.what() << "\n"; return 1; }
}

// StateInputQuadraticCost (diagonal Q[nx], R[nu]) on the manager's references: dx[nx], du[nu] = the deviation its quadratic form is taken of —
// the gradient of the stand-in QuadraticStateInputCost under unit weights —, value = 1/2 dx'Q dx + 1/2 du'R du.
int ref_state_input_quadratic_cost(int nj, const int arm[4], double total_mass, int n_events, const double* event_times, const int* mode_sequence, int n_knots,
                                   const double* tt, const double* ts, int arm_swing, double t0, double tf, const double* Q, const double* R, const double* state,
                                   const double* input, double time, double* dx, double* du, double* value) {
  try {
    World w(nj, arm);
    ref_stub_total_mass() = total_mass;
    auto mgr = make_manager(w, n_events, event_times, mode_sequence);
    const int nx = (int)w.model.getStateDim(), nu = (int)w.model.getInputDim();
    TargetTrajectories targets;
    for (int k = 0; k < n_knots; ++k) { targets.timeTrajectory.push_back(tt[k]); targets.stateTrajectory.push_back(to_vec(ts + (size_t)k * nx, nx)); }
    mgr->setTargetTrajectories(targets);
    mgr->setArmSwingReferenceActive(arm_swing != 0);
    const vector_t xs = to_vec(state, nx), us = to_vec(input, nu);
    mgr->preSolverRun(t0, tf, xs);
    matrix_t Qm = matrix_t::Zero(nx, nx), Rm = matrix_t::Zero(nu, nu);
    for (int i = 0; i < nx; ++i) Qm(i, i) = Q[i];
    for (int i = 0; i < nu; ++i) Rm(i, i) = R[i];
    const PinocchioInterface pin;
    // (the deviation through unit weights — task weights may be zero —, the value through the given ones)
    StateInputQuadraticCost unit(matrix_t::Identity(nx, nx), matrix_t::Identity(nu, nu), *mgr, pin, w.model);
    const auto L1 = unit.getQuadraticApproximation(time, xs, us, mgr->getTargetTrajectories(), NoPreComp());
    for (int i = 0; i < nx; ++i) dx[i] = L1.dfdx(i);
    for (int i = 0; i < nu; ++i) du[i] = L1.dfdu(i);
    StateInputQuadraticCost cost(Qm, Rm, *mgr, pin, w.model);
    *value = cost.getValue(time, xs, us, mgr->getTargetTrajectories(), NoPreComp());
    return 0;
  } catch (const std::exception& e) { std::cerr << "ref_state_input_quadratic_cost: " << e.what() << "\n"; return 1; }
}

// JointLimitsSoftConstraint with the stand-in penalty (mu, delta): f, dfdx[nx], the diagonal of dfdxx[nx] (off-diagonal entries must be zero)
int ref_joint_limits(int nj, const double* q_lo, const double* q_hi, double mu, double delta, const double* state, double* f, double* dfdx, double* dfdxx_diag) {
  try {
    const int arm[4] = {0, 0, 0, 0};
    World w(nj, arm);
    const int nx = (int)w.model.getStateDim();
    JointLimitsSoftConstraint con({to_vec(q_lo, nj), to_vec(q_hi, nj)}, PieceWisePolynomialBarrierPenalty::Config(mu, delta), w.model);
    const vector_t xs = to_vec(state, nx);
    const TargetTrajectories none;
    const auto L = con.getQuadraticApproximation(0.0, xs, none, NoPreComp());
    if (con.getValue(0.0, xs, none, NoPreComp()) != L.f) return 2;
    *f = L.f;
    for (int i = 0; i < nx; ++i) {
      dfdx[i] = L.dfdx(i); dfdxx_diag[i] = L.dfdxx(i, i);
      for (int j = 0; j < nx; ++j) if (i != j && L.dfdxx(i, j) != 0.0) return 3;
    }
    return 0;
  } catch (const std::exception& e) { std::cerr << "ref_joint_limits: " << e.what() << "\n"; return 1; }
}

// WeightCompInitializer::compute(time, state, nextTime) on the mode schedule: input[nu], nextState[nx]
int ref_weight_comp_initializer(int nj, double total_mass, int n_events, const double* event_times, const int* mode_sequence, double time, double next_time,
                                const double* state, double* input, double* next_state) {
  try {
    const int arm[4] = {0, 0, 0, 0};
    World w(nj, arm);
    ref_stub_total_mass() = total_mass;
    auto mgr = make_manager(w, n_events, event_times, mode_sequence);
    mgr->setModeSchedule(ModeSchedule(std::vector<scalar_t>(event_times, event_times + n_events), std::vector<size_t>(mode_sequence, mode_sequence + n_events + 1)));
    const int nx = (int)w.model.getStateDim(), nu = (int)w.model.getInputDim();
    const PinocchioInterface pin;
    WeightCompInitializer init(pin, *mgr, w.model);
    vector_t u, xn;
    init.compute(time, to_vec(state, nx), next_time, u, xn);
    if ((int)u.size() != nu || (int)xn.size() != nx) return 2;
    for (int i = 0; i < nu; ++i) input[i] = u(i);
    for (int i = 0; i < nx; ++i) next_state[i] = xn(i);
    return 0;
  } catch (const std::exception& e) { std::cerr << "ref_weight_comp_initializer: " << e.what() << "\n"; return 1; }
}

}  // extern "C"
