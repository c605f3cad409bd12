// TEST INFRASTRUCTURE — a MOCK of the Pinocchio names the reference's assembly files touch (oracle/Makefile, target _ref/libref_model.so).
// Pinocchio is absent from /root/reference and from this image, so the rigid-body model itself stays unpinned; what CAN be pinned is how the
// reference combines kinematic quantities into rows and values: here every "algorithm" is a no-op and every query returns
// what the caller handed in through DataTpl — frame placements, frame velocities / classical accelerations (LOCAL_WORLD_ALIGNED), the rows of
// CRBA's M, nonLinearEffects' nle and the frame Jacobians.  Model frames are looked up by NAME, as the reference does.
#pragma once
#include <Eigen/Core>
#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>
namespace pinocchio {
using FrameIndex = std::size_t;
using JointIndex = std::size_t;
enum ReferenceFrame { WORLD = 0, LOCAL = 1, LOCAL_WORLD_ALIGNED = 2 };
template <class S> struct SE3Tpl {
  Eigen::Matrix<S, 3, 3> R = Eigen::Matrix<S, 3, 3>::Identity();
  Eigen::Matrix<S, 3, 1> t = Eigen::Matrix<S, 3, 1>::Zero();
  const Eigen::Matrix<S, 3, 3>& rotation() const { return R; }
  const Eigen::Matrix<S, 3, 1>& translation() const { return t; }
  Eigen::Matrix<S, 4, 4> toHomogeneousMatrix_impl() const { Eigen::Matrix<S, 4, 4> H = Eigen::Matrix<S, 4, 4>::Identity(); for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) H(i, j) = R(i, j); H(i, 3) = t(i); } return H; }
};
template <class S> struct MotionTpl {
  Eigen::Matrix<S, 3, 1> lin = Eigen::Matrix<S, 3, 1>::Zero(), ang = Eigen::Matrix<S, 3, 1>::Zero();
  const Eigen::Matrix<S, 3, 1>& linear() const { return lin; }
  const Eigen::Matrix<S, 3, 1>& angular() const { return ang; }
};
template <class S> struct ForceTpl {
  Eigen::Matrix<S, 3, 1> lin = Eigen::Matrix<S, 3, 1>::Zero(), ang = Eigen::Matrix<S, 3, 1>::Zero();
  static ForceTpl Zero() { return ForceTpl(); }
  Eigen::Matrix<S, 3, 1>& linear() { return lin; }
  Eigen::Matrix<S, 3, 1>& angular() { return ang; }
};
using Force = ForceTpl<double>;
template <class S> struct FrameTpl { std::string name; JointIndex parent = 0; JointIndex parentJoint = 0; SE3Tpl<S> placement; };
namespace container { template <class T> using aligned_vector = std::vector<T>; }
template <class S> struct ModelTpl {
  std::vector<FrameTpl<S>> frames;
  int nq = 0, nv = 0, njoints = 0;
  int nv_base_joint = 6;   // nv of the first joint: the composite floating base (see crba below)
  S total_mass = S(0);
  FrameIndex getFrameId(const std::string& name) const {
    for (std::size_t i = 0; i < frames.size(); ++i) if (frames[i].name == name) return i;
    throw std::runtime_error("[mock pinocchio] unknown frame " + name);
  }
  template <class T> ModelTpl<T> cast() const {
    ModelTpl<T> m; m.nq = nq; m.nv = nv; m.njoints = njoints; m.total_mass = T(total_mass); m.nv_base_joint = nv_base_joint;
    for (const auto& f : frames) { FrameTpl<T> g; g.name = f.name; m.frames.push_back(g); }
    return m;
  }
};
template <class S> struct DataTpl {
  using Mat = Eigen::Matrix<S, Eigen::Dynamic, Eigen::Dynamic>;
  using Vec = Eigen::Matrix<S, Eigen::Dynamic, 1>;
  std::vector<SE3Tpl<S>> oMf, oMi;                 // handed in
  std::vector<MotionTpl<S>> fv, fa;                // frame velocity / classical acceleration, LOCAL_WORLD_ALIGNED, handed in
  std::vector<Mat> J;                              // frame Jacobians (6 x nv, LOCAL_WORLD_ALIGNED), handed in
  Mat M_in;                                        // handed in: what CRBA would leave (its upper triangle is what crba() copies into M)
  Vec nle_in;                                      // handed in: nonLinearEffects
  Mat M;                                           // crba
  Vec nle, tau;                                    // nonLinearEffects, rnea
  template <class T> DataTpl<T> cast() const {
    DataTpl<T> d;
    auto cm = [](const Mat& a) { typename DataTpl<T>::Mat b(a.rows(), a.cols()); for (Eigen::Index i = 0; i < a.rows(); ++i) for (Eigen::Index j = 0; j < a.cols(); ++j) b(i, j) = T(a(i, j)); return b; };
    auto c3 = [](const Eigen::Matrix<S, 3, 1>& a) { Eigen::Matrix<T, 3, 1> b; for (int i = 0; i < 3; ++i) b(i) = T(a(i)); return b; };
    for (const auto& p : oMf) { SE3Tpl<T> q; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) q.R(i, j) = T(p.R(i, j)); q.t = c3(p.t); d.oMf.push_back(q); }
    d.oMi.resize(oMi.size());
    for (const auto& m : fv) { MotionTpl<T> q; q.lin = c3(m.lin); q.ang = c3(m.ang); d.fv.push_back(q); }
    for (const auto& m : fa) { MotionTpl<T> q; q.lin = c3(m.lin); q.ang = c3(m.ang); d.fa.push_back(q); }
    for (const auto& j : J) d.J.push_back(cm(j));
    d.M = cm(M); d.M_in = cm(M_in);
    { typename DataTpl<T>::Vec v(nle_in.size()); for (Eigen::Index i = 0; i < nle_in.size(); ++i) v(i) = T(nle_in(i)); d.nle_in = v; }
    { typename DataTpl<T>::Vec v(nle.size()); for (Eigen::Index i = 0; i < nle.size(); ++i) v(i) = T(nle(i)); d.nle = v; }
    { typename DataTpl<T>::Vec v(tau.size()); for (Eigen::Index i = 0; i < tau.size(); ++i) v(i) = T(tau(i)); d.tau = v; }
    return d;
  }
};
using Model = ModelTpl<double>;
using Data = DataTpl<double>;
// ---- "algorithms": the kinematics are handed in, so these do nothing / return the handed-in quantity
template <class S, class... A> void forwardKinematics(const ModelTpl<S>&, DataTpl<S>&, const A&...) {}
template <class S> void updateFramePlacements(const ModelTpl<S>&, DataTpl<S>&) {}
template <class S> const SE3Tpl<S>& updateFramePlacement(const ModelTpl<S>&, DataTpl<S>& data, FrameIndex id) { return data.oMf.at(id); }
template <class S> MotionTpl<S> getFrameVelocity(const ModelTpl<S>&, const DataTpl<S>& data, FrameIndex id, ReferenceFrame) { return data.fv.at(id); }
template <class S> MotionTpl<S> getFrameClassicalAcceleration(const ModelTpl<S>&, const DataTpl<S>& data, FrameIndex id, ReferenceFrame) { return data.fa.at(id); }
// crba (Pinocchio, crba.hxx backward step): for joint i it writes the ROW block M[idx_v(i) .. +nv(i), idx_v(i) .. end of i's subtree] = S_i^T (Y_i^c S) — the
// joint's own nv x nv diagonal block COMPLETELY, and what lies to its right; nothing below the block diagonal.  The floating base of the reference is ONE
// joint of nv = 6 (JointModelComposite(Translation, SphericalZYX), humanoid_common_mpc/src/pinocchio_model/createPinocchioModel.cpp:60-67, 167), so rows 0..5
// are complete (the 3 x 3 blocks the reference inverts are the full symmetric ones) and the 1-dof joints fill their upper part.  The reference zero-fills M
// first (humanoid_wb_mpc/src/dynamics/DynamicsHelperFunctions.cpp:62).
template <class S, class Q> void crba(const ModelTpl<S>& model, DataTpl<S>& data, const Q&) {
  if (data.M.rows() != data.M_in.rows() || data.M.cols() != data.M_in.cols()) { typename DataTpl<S>::Mat z; z.resize(data.M_in.rows(), data.M_in.cols()); data.M = z; }
  const Eigen::Index nb = model.nv_base_joint;
  for (Eigen::Index i = 0; i < data.M_in.rows(); ++i)
    for (Eigen::Index j = 0; j < data.M_in.cols(); ++j)
      if (j >= i || (i < nb && j < nb)) data.M(i, j) = data.M_in(i, j);
}
template <class S, class Q, class V> void nonLinearEffects(const ModelTpl<S>&, DataTpl<S>& data, const Q&, const V&) { data.nle = data.nle_in; }
template <class S, class Q, class V, class A> const typename DataTpl<S>::Vec& rnea(const ModelTpl<S>&, DataTpl<S>& data, const Q&, const V&, const A&) { return data.tau; }
template <class S, class Q, class V, class A, class F> const typename DataTpl<S>::Vec& rnea(const ModelTpl<S>&, DataTpl<S>& data, const Q&, const V&, const A&, const F&) { return data.tau; }
template <class S, class Q, class JM> void computeFrameJacobian(const ModelTpl<S>&, DataTpl<S>& data, const Q&, FrameIndex id, ReferenceFrame, JM& J) { J = data.J.at(id); }
template <class S> S computeTotalMass(const ModelTpl<S>& m) { return m.total_mass; }
}  // namespace pinocchio
