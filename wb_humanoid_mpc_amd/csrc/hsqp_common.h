// Shared definitions of the HIP SQP kernels (device code) — MI355X / gfx950.
//
// Kernel bodies are written as a sequence of workgroup-parallel phases over an LDS
// workspace: WG_FOR(ctx, i, n) distributes n independent work items over the
// workgroup's threads, WG_SYNC(ctx) is the barrier between dependent phases.  Rules:
//   * LDS / global writes happen only inside WG_FOR bodies (or under ctx.tid == 0),
//   * no per-thread value is carried from one phase to the next (only values every
//     thread recomputes identically from LDS).
// With those rules the same source also compiles for the host with a one-thread
// context (tests/hostemu), which is how the kernel arithmetic is checked against the
// oracle in the GPU-less build container.  The host build is test infrastructure only;
// the product library contains the device build exclusively.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/hsqp.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HSQP_HD __host__ __device__ __forceinline__
#define HSQP_D __device__ __forceinline__
#else
#define HSQP_HD inline
#define HSQP_D inline
#endif

namespace hsqp {

constexpr int NJ = HSQP_NJ, NV = HSQP_NV, NX = HSQP_NX, NU = HSQP_NU, NB = HSQP_NB, NZ = NX + NU;
constexpr int NP = HSQP_NODE_PARAMS;
constexpr int NJC = 26;        // revolute generalized coordinates: euler z,y,x + 23 joints (coordinate index = 3 + jc)
constexpr int NE_MAX = 14;     // max active equality rows (flight: 6+1+6+1)
constexpr int NUT = 23;        // max projected input dimension nu - ne (padded with identity when ne > 12)
constexpr int NR = 60;         // residual-row slots of the Gauss-Newton / penalty model (see hsqp_node.h)
constexpr int LDJ = 96;        // leading dimension of Jacobian rows over z = [x;u] in memory (cols 0..92 used; CDe col 93 = e)
constexpr int MAXCHAIN = 6;    // longest chain (maximal single-child path) of the kinematic tree
constexpr int NANC = 8;        // longest path of moving bodies from the base to a leaf
constexpr int NLEVELS = 8;     // depth of the body tree incl. the base
constexpr int QV_LIMBS = 4;    // root-to-leaf paths of the kinematic tree the quad value pass walks, one per lane (hsqp_lqv.h)

struct Ctx {
  int tid;
  int nthreads;
  long long* prof;   // phase-profile buffer of workgroup 0 (only in -DHSQP_PHASE_PROFILE builds), else null
};

// The wave a thread belongs to, as a SCALAR on the device: every lane of a wave has the same tid >> 6, but the compiler does not know it — role branches
// (which wave eliminates, which wave takes which tile) and the tile coordinates derived from the wave index would otherwise be vector arithmetic and
// exec-masked branches in every lane.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HSQP_NO_SCALAR_WAVE)
__device__ inline int wave_index(int tid) { return __builtin_amdgcn_readfirstlane(tid >> 6); }
#else
HSQP_HD int wave_index(int tid) { return tid >> 6; }
#endif

// Phase profiling (tools/phase_profile.py): thread 0 of workgroup 0 accumulates shader-clock ticks per phase id.
#if defined(HSQP_PHASE_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define PH_TICK(ctx, id)                                                             \
  do {                                                                               \
    if ((ctx).tid == 0 && (ctx).prof) {                                              \
      const long long t_ = clock64();                                                \
      (ctx).prof[id] += t_ - (ctx).prof[127];                                        \
      (ctx).prof[127] = t_;                                                          \
    }                                                                                \
  } while (0)
// per-wave arrival times at the barrier that ends a phase: PH_MARK after the barrier that starts it, PH_ARRIVE(slot 0..8)
// before the one that ends it; lane 0 of every wave accumulates into prof[40 + 8 * slot + wave] (up to 8 waves)
#define PH_MARK(ctx)                                                                                     \
  do { if (((ctx).tid & 63) == 0 && (ctx).prof) (ctx).prof[112 + ((ctx).tid >> 6)] = clock64(); } while (0)
#define PH_ARRIVE(ctx, slot)                                                                             \
  do {                                                                                                   \
    if (((ctx).tid & 63) == 0 && (ctx).prof) (ctx).prof[40 + 8 * (slot) + ((ctx).tid >> 6)] += clock64() - (ctx).prof[112 + ((ctx).tid >> 6)]; \
  } while (0)
// laps inside a phase, for ONE chosen wave (its lane 0): PH_LAP0 arms the lap clock (slot 39), PH_LAP(id) adds the time since the last lap to slot id (20 .. 38)
#define PH_LAP0(ctx, wave)                                                                               \
  do { if ((ctx).prof && (ctx).tid == 64 * (wave)) (ctx).prof[39] = clock64(); } while (0)
#define PH_LAP(ctx, wave, id)                                                                            \
  do { if ((ctx).prof && (ctx).tid == 64 * (wave)) { const long long t_ = clock64(); (ctx).prof[id] += t_ - (ctx).prof[39]; (ctx).prof[39] = t_; } } while (0)
#else
#define PH_TICK(ctx, id) ((void)0)
#define PH_MARK(ctx) ((void)0)
#define PH_ARRIVE(ctx, slot) ((void)0)
#define PH_LAP0(ctx, wave) ((void)0)
#define PH_LAP(ctx, wave, id) ((void)0)
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define WG_SYNC(ctx) __syncthreads()
#else
#define WG_SYNC(ctx) ((void)0)
#endif
#if defined(HSQP_EMU_REVERSE) && !defined(__HIP_DEVICE_COMPILE__)
// host emulation, race check: the items of a phase are executed in REVERSE order.  Phases are race-free iff their items are
// independent, i.e. iff this build produces bit-identical results to the forward build (tests/hostemu, race check).
#define WG_FOR(ctx, i, n) for (int i = (n) - 1 - (ctx).tid; i >= 0; i -= (ctx).nthreads)
#else
#define WG_FOR(ctx, i, n) for (int i = (ctx).tid; i < (n); i += (ctx).nthreads)
#endif

// Wave-local section: code executed by ONE wave (wave 0) needs no workgroup barrier between its dependent steps — the
// LDS processes a wave's instructions in order; WV_SYNC only has to stop the compiler from reordering across the step
// boundary and make the wave wait for its outstanding LDS operations.
#if defined(__HIP_DEVICE_COMPILE__)
#define WV_SYNC()                                              \
  do {                                                         \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     \
    __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0) */       \
    __builtin_amdgcn_wave_barrier();                           \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     \
  } while (0)
#else
#define WV_SYNC() ((void)0)
#endif
HSQP_HD bool is_wave0(const Ctx& c) { return c.nthreads < 128 || c.tid < 64; }
HSQP_HD Ctx wave0_ctx(const Ctx& c) { return c.nthreads < 128 ? c : Ctx{c.tid, 64, c.prof}; }

// Wave specialisation inside one phase: the LOWER half of the workgroup's waves feeds the matrix cores (one wave per
// SIMD saturates the FP64 MFMA pipe), the UPPER half runs the phase's copies / vector work concurrently.
// The one-thread host context plays both roles.
HSQP_HD bool is_mfma_half(const Ctx& c) { return c.nthreads < 128 || c.tid < c.nthreads / 2; }
HSQP_HD bool is_helper_half(const Ctx& c) { return c.nthreads < 128 || c.tid >= c.nthreads / 2; }
HSQP_HD Ctx mfma_ctx(const Ctx& c) { return c.nthreads < 128 ? c : Ctx{c.tid, c.nthreads / 2, c.prof}; }
HSQP_HD Ctx helper_ctx(const Ctx& c) { return c.nthreads < 128 ? c : Ctx{c.tid - c.nthreads / 2, c.nthreads / 2, nullptr}; }

// ------------------------------------------------------------------------------------------------
// Device image of the model constants (built on the host from hsqp_model_desc, hsqp_host.cpp).
struct DevModel {
  // kinematic tree, bodies in depth-first order (subtree of i = [i, i + subtree_size[i]))
  int parent[NB];
  int subtree_size[NB];
  double Rfix[NB][9];
  double pfix[NB][3];
  double axis[NB][3];
  double axis_p[NB][3];            // Rfix * axis: joint axis in the parent body frame
  // chains: maximal single-child paths of consecutive bodies (the placement walk runs per chain end, the composite sums per
  // chain); chain n_chains is the base alone
  int n_chains;
  int chain_start[NB], chain_len[NB];
  // ancestor path of every body (root-most moving body first, the body itself last; the base is not listed)
  int n_anc[NB];
  unsigned char anc[NB][NANC];
  double mass[NB];
  double com[NB][3];
  double inertia[NB][9];          // about com, body axes
  double q_lo[NJ], q_hi[NJ];
  double total_mass;
  double gravity;
  // frames: body + offset
  int contact_body[2];
  double contact_p[2][3];
  int coll_body[10];              // order: ankle_l, ankle_r, f_l, f_r, l1, r1, l2, r2, k_l, k_r
  double coll_p[10][3];
  // task constants
  double Q[NX], R[NU], Qf[NX];
  double foot_sqrt_w[18];
  double gain_pos_z, gain_ori, gain_linvel_z, gain_linvel_xy, gain_angvel, gain_linacc_z, gain_linacc_xy, gain_angacc;
  double friction_mu, friction_reg, friction_grip, friction_hess_shift, friction_bmu, friction_bdelta;
  double rect_x_min, rect_x_max, rect_y_min, rect_y_max, moment_bmu, moment_bdelta;
  double jl_bmu, jl_bdelta;
  double r_foot, r_knee, coll_bmu, coll_bdelta;
  int arm_swing_joint[4];
  // centroidal formulation (hsqp_cent.h); unused by the whole-body kernels
  int formulation, torso_body;
  int ext_joint[2][6];
  int chain_par_slot[NB];        // per chain: where its first body's parent record sits (0 = base, 1 + c = last body of chain c)
  double torso_p[3], torso_R[9], torso_sqrt_w[12], cent_foot_sqrt_w[12], ext_sqrt_w[2][6];
  // limbs (hsqp_lqv.h): the root-to-leaf paths of moving bodies, root first, eight body indices packed into a word; bit k of limb_own: the limb adds
  // body k of its path to the sums over bodies (a shared ancestor is owned by the first limb that passes it; limb 0 owns the base).  n_limbs = 0: the
  // tree has more than QV_LIMBS leaves (or both feet on one limb) and the value pass runs in its phase form
  unsigned long long limb_path[QV_LIMBS];
  int limb_len[QV_LIMBS];
  unsigned limb_own[QV_LIMBS];
  int limb_max_len, n_limbs;
  int foot_limb[2];
  // limb lanes of the LQ kernel (hsqp_lql.h): on the way back from the leaves, before step t lane L adds the composite of lane L ^ k for every set
  // bit k (1..3) of limb_merge[t][L] — the limbs that hang off body path_L[t] below the part of the path the two lanes share.  ql_ok = 0: the tree
  // does not fit that scheme (no limbs, or a foot's ancestors shared with another limb) and the LQ kernel keeps its phase form
  unsigned char limb_merge[NANC][QV_LIMBS];
  int limb_foot_step[QV_LIMBS];   // the step of the limb's path on which its foot body sits (-1: the limb carries no foot)
  int ql_ok;
};

// ------------------------------------------------------------------------------------------------
// 3-vectors and 6-D spatial vectors stored as plain double arrays.
// Motion m = {ang[3], lin[3]}, force f = {moment[3], force[3]}, both about the common origin O.
HSQP_HD void v3_cross(const double* a, const double* b, double* r) {
  const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
HSQP_HD double v3_dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
HSQP_HD void m3_mulv(const double* M, const double* v, double* r) {  // r = M v (row-major), r may not alias v
  r[0] = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
  r[1] = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
  r[2] = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
}
HSQP_HD void m3_tmulv(const double* M, const double* v, double* r) {  // r = M^T v
  r[0] = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
  r[1] = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
  r[2] = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
}
HSQP_HD void m3_mul(const double* A, const double* B, double* C) {  // C = A B, C may not alias
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
// motion x motion
HSQP_HD void mxm(const double* a, const double* b, double* r) {
  double w[3], l1[3], l2[3];
  v3_cross(a, b, w);
  v3_cross(a, b + 3, l1);
  v3_cross(a + 3, b, l2);
  r[0] = w[0]; r[1] = w[1]; r[2] = w[2];
  r[3] = l1[0] + l2[0]; r[4] = l1[1] + l2[1]; r[5] = l1[2] + l2[2];
}
// motion x* force
HSQP_HD void mxf(const double* m, const double* f, double* r) {
  double n1[3], n2[3], ff[3];
  v3_cross(m, f, n1);
  v3_cross(m + 3, f + 3, n2);
  v3_cross(m, f + 3, ff);
  r[0] = n1[0] + n2[0]; r[1] = n1[1] + n2[1]; r[2] = n1[2] + n2[2];
  r[3] = ff[0]; r[4] = ff[1]; r[5] = ff[2];
}
// Spatial inertia about O in world axes: In = {m, h[3] = m*c, Ibar[6] = xx,xy,xz,yy,yz,zz (about O)}
HSQP_HD void sym3_mulv(const double* S, const double* v, double* r) {
  r[0] = S[0] * v[0] + S[1] * v[1] + S[2] * v[2];
  r[1] = S[1] * v[0] + S[3] * v[1] + S[4] * v[2];
  r[2] = S[2] * v[0] + S[4] * v[1] + S[5] * v[2];
}
HSQP_HD void inertia_apply(const double* In, const double* m, double* f) {  // f = I m
  double a[3], b[3], c[3];
  sym3_mulv(In + 4, m, a);
  v3_cross(In + 1, m + 3, b);
  v3_cross(In + 1, m, c);
  f[0] = a[0] + b[0]; f[1] = a[1] + b[1]; f[2] = a[2] + b[2];
  f[3] = In[0] * m[3] - c[0]; f[4] = In[0] * m[4] - c[1]; f[5] = In[0] * m[5] - c[2];
}
// r = Minv(3x3 row-major) * v helper for the 3x3 solves
HSQP_HD void m3_inverse(const double* a, double* c) {
  c[0] = a[4] * a[8] - a[5] * a[7]; c[1] = a[2] * a[7] - a[1] * a[8]; c[2] = a[1] * a[5] - a[2] * a[4];
  c[3] = a[5] * a[6] - a[3] * a[8]; c[4] = a[0] * a[8] - a[2] * a[6]; c[5] = a[2] * a[3] - a[0] * a[5];
  c[6] = a[3] * a[7] - a[4] * a[6]; c[7] = a[1] * a[6] - a[0] * a[7]; c[8] = a[0] * a[4] - a[1] * a[3];
  const double inv = 1.0 / (a[0] * c[0] + a[1] * c[3] + a[2] * c[6]);
  for (int i = 0; i < 9; ++i) c[i] *= inv;
}

// 1/sqrt(x) for the Cholesky pivots
HSQP_HD double inv_sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return rsqrt(x);
#else
  return 1.0 / sqrt(x);
#endif
}

// 1/x: hardware reciprocal + two Newton steps on the device (the elimination steps need a reciprocal pivot per item)
HSQP_HD double fast_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = r * (2.0 - x * r);
  r = r * (2.0 - x * r);
  return r;
#else
  return 1.0 / x;
#endif
}

// Penalties (RelaxedBarrierPenalty: upstream ocs2; PieceWisePolynomialBarrierPenalty: fork-only, ASSUMPTION A1 of the oracle)
struct Pen3 { double p, d1, d2; };
HSQP_HD Pen3 relaxed_barrier(double mu, double delta, double h) {
  Pen3 r;
  if (h > delta) { r.p = -mu * log(h); r.d1 = -mu / h; r.d2 = mu / (h * h); }
  else { const double t = (h - 2.0 * delta) / delta; r.p = mu * (-log(delta) + 0.5 * t * t - 0.5); r.d1 = mu * (h - 2.0 * delta) / (delta * delta); r.d2 = mu / (delta * delta); }
  return r;
}
HSQP_HD Pen3 pwp_barrier(double mu, double delta, double h) {
  Pen3 r;
  if (h >= delta) { r.p = 0.0; r.d1 = 0.0; r.d2 = 0.0; }
  else { const double t = (delta - h) / delta; r.p = mu * t * t * t; r.d1 = -3.0 * mu * t * t / delta; r.d2 = 6.0 * mu * t / (delta * delta); }
  return r;
}

}  // namespace hsqp
