// C ABI (include/hsqp.h) of the MI355X SQP library: HIP kernels + host orchestration.
// Product path: there is NO CPU fallback — without a usable HIP device every entry point fails with
// HSQP_ERR_NO_DEVICE / HSQP_ERR_HIP.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "hsqp_host.h"
#include "hsqp_riccati.h"
#include "hsqp_riccati_fact.h"
#include "hsqp_params.h"
#include "hsqp_policy.h"
#include "hsqp_cent.h"
#include "hsqp_cent_lq.h"
#include "hsqp_scan.h"
#include "hsqp_segment.h"
#include "hsqp_lqv.h"
#include "hsqp_lql.h"

using namespace hsqp;

namespace {

// launch shapes of the LQ kernels (overridable for tuning builds: tools/build_variants.py)
#ifndef HSQP_LQ_THREADS
#define HSQP_LQ_THREADS 128   /* 40 KB workspace: 4 workgroups of 2 waves per CU, no register spills */
#endif
#ifndef HSQP_LQ_WPE
#define HSQP_LQ_WPE 2
#endif
#ifndef HSQP_LQV_THREADS
#define HSQP_LQV_THREADS 64   /* value-only pass: its phases rarely have more than one wave of work; one wave needs no barrier hardware (A/B: 0.95 vs 0.99 ms) */
#endif
#ifndef HSQP_LQV_WPE
#define HSQP_LQV_WPE 2
#endif
constexpr int LQ_THREADS = HSQP_LQ_THREADS;

#ifndef HSQP_VALUE_QUAD_MIN_NODES
#define HSQP_VALUE_QUAD_MIN_NODES 2048   /* handles sized below this keep the phase form of the whole-body value pass (hsqp_create) */
#endif
#ifndef HSQP_LQ_LIMB_MIN_NODES
#define HSQP_LQ_LIMB_MIN_NODES 2048    /* handles sized below this keep the phase form of the whole-body LQ kernel (hsqp_create) */
#endif
#ifndef HSQP_PROJ_THREADS
#define HSQP_PROJ_THREADS 256
#endif
#ifndef HSQP_PROJ_WPE
#define HSQP_PROJ_WPE 3
#endif
constexpr int PROJ_THREADS = HSQP_PROJ_THREADS;   // 49.5 KB workspace: three workgroups of four waves per CU
static_assert(PROJ_THREADS >= 256 && PROJ_THREADS % 64 == 0, "project_node hoists its staging loads assuming >= 256 threads; the Gram tiles are dealt to waves 0..3");
// Every kernel hands its workgroup size to the device functions as the CONSTANT it is launched with (Ctx::nthreads), not as blockDim.x: the item loops
// (WG_FOR) then have constant strides and trip counts, and the paths written for other workgroup shapes (the host build's) are not compiled into the
// kernels — k_riccati alone shrank from 16.7 k to 9 k instructions and from 1.518 to 1.463 ms (256 instances; 1.378 -> 1.317 at 32).
constexpr int RIC_THREADS = 512;
constexpr int LQV_THREADS = HSQP_LQV_THREADS;   // value-only LQ pass (19.2 KB workspace: 8 one-wave workgroups per CU)
static_assert(sizeof(LqWST<false>) <= 163840 / 8, "value-only workspace: eight workgroups per CU");

extern __shared__ __attribute__((aligned(16))) unsigned char hsqp_smem[];

// ---- test aid (HSQP_POISON_LDS in the environment at hsqp_create; process-wide): every kernel launch is preceded by one that fills the LDS of every CU
// with NaN bit patterns.  LDS keeps what the previous kernel left there, so a kernel that reads a word it never wrote (say a padding row that is
// "multiplied by zero") works until the leftover happens to be a NaN — tests/test_gpu_parity.py runs the iteration with and without the poison
// and asks for the same bits.
constexpr int POISON_LDS_BYTES = 163400;   // what hipFuncSetAttribute(MaxDynamicSharedMemorySize) accepts on gfx950: one such workgroup per CU at a time
__global__ __launch_bounds__(256) void k_poison_lds() {
  volatile unsigned long long* p = reinterpret_cast<volatile unsigned long long*>(hsqp_smem);
  for (int i = threadIdx.x; i < POISON_LDS_BYTES / 8; i += 256) p[i] = 0x7ff8dead7ff8deadull;   // a NaN as a double and as two floats
}
bool g_poison_lds = false;
bool g_poison_hbm = false;                 // HSQP_POISON_HBM (test aid, read at hsqp_create): every device buffer the library allocates starts as NaN bit patterns
// (0xFF bytes) instead of whatever the allocator returns — usually zeros, which hide a read of something never written.  Not the LQ record: the limb-lane
// kernels rely on its zero fill (hsqp_lql.h).
inline void poison_hbm(void* p, size_t bytes) { if (g_poison_hbm && p) (void)hipMemset(p, 0xFF, bytes); }
int g_poison_blocks = 512;                 // two per CU (set from the device's CU count at hsqp_create)
#define HSQP_LAUNCH(kernel, grid, block, lds, st, ...)                                                                  \
  do {                                                                                                                 \
    if (g_poison_lds) hipLaunchKernelGGL(k_poison_lds, dim3(g_poison_blocks), dim3(256), POISON_LDS_BYTES, (st));      \
    hipLaunchKernelGGL(kernel, (grid), (block), (lds), (st), __VA_ARGS__);                                             \
  } while (0)

// ---- LQ approximation: one workgroup per (instance, node)
template <bool DERIV>
__global__ __launch_bounds__(DERIV ? LQ_THREADS : LQV_THREADS, DERIV ? HSQP_LQ_WPE : HSQP_LQV_WPE) void k_lq(const DevModel* __restrict__ dm, const double* __restrict__ x,
                                                   const double* __restrict__ u, const double* __restrict__ par, const double* __restrict__ dts, int N,
                                                   double* __restrict__ rec, double* __restrict__ misc, long long* prof,
                                                   const LsState* __restrict__ ls) {
  const int node = blockIdx.x, b = node / N, k = node % N;
  if (ls && !ls[b].active) return;   // line search: only the instances whose trial is pending are re-evaluated
  LqWST<DERIV>& w = *reinterpret_cast<LqWST<DERIV>*>(hsqp_smem);
  const Ctx ctx{(int)threadIdx.x, (int)blockDim.x, blockIdx.x == 0 ? prof : nullptr};   // (the constant crashes this compiler's register allocator on k_lq<false>)
  PH_TICK(ctx, 126);  // re-arm the phase clock (bucket 126 is a sink)
  const double* xk = x + ((size_t)b * (N + 1) + k) * NX;
  lq_node<DERIV>(ctx, *dm, w, xk, u + ((size_t)b * N + k) * NU, xk + NX, par + ((size_t)b * (N + 1) + k) * NP, dts[node],
                 DERIV ? rec + (size_t)node * REC_SIZE : nullptr,
                 DERIV ? rec + (size_t)node * REC_SIZE + REC_MISC : misc + (size_t)node * 8);
}

// ---- centroidal LQ approximation, second form (hsqp_cent_lq.h): one 128-thread workgroup per (instance, node), values once per node in an LDS
//      workspace, tangent lanes on closed-form seeds; four workgroups per CU
static_assert(sizeof(CentWST<true>) <= 163840 / 4, "centroidal LQ workspace: four workgroups per CU");
__global__ __launch_bounds__(CLQ_THREADS, 2) void k_lq_cent2(const DevModel* __restrict__ dm, const double* __restrict__ x, const double* __restrict__ u,
                                                             const double* __restrict__ par, const double* __restrict__ dts, int N, double* __restrict__ rec) {
  const int node = blockIdx.x, b = node / N, k = node % N;
  CentWST<true>& w = *reinterpret_cast<CentWST<true>*>(hsqp_smem);
  const Ctx ctx{(int)threadIdx.x, (int)blockDim.x, nullptr};   // (this kernel is faster with the run-time value: 0.121 against 0.165 ms at N = 100 — the constant changes its unrolling and with it the spills)
  const double* xk = x + ((size_t)b * (N + 1) + k) * NX;
  double* r = rec + (size_t)node * REC_SIZE;
  cent_lq_node2<true>(ctx, *dm, w, xk, u + ((size_t)b * N + k) * NU, xk + NX, par + ((size_t)b * (N + 1) + k) * NP, dts[node], r, r + REC_MISC);
}
// ---- centroidal value-only pass (performance index, line-search trials): the same node function without derivative storage, one wave per
//      (instance, node), 13 KB of LDS
static_assert(sizeof(CentWST<false>) <= 163840 / 8, "centroidal value-only workspace: eight workgroups per CU");
__global__ __launch_bounds__(64) void k_lq_cent2_value(const DevModel* __restrict__ dm, const double* __restrict__ x, const double* __restrict__ u,
                                                       const double* __restrict__ par, const double* __restrict__ dts, int N, double* __restrict__ misc,
                                                       const LsState* __restrict__ ls) {
  const int node = blockIdx.x, b = node / N, k = node % N;
  if (ls && !ls[b].active) return;
  CentWST<false>& w = *reinterpret_cast<CentWST<false>*>(hsqp_smem);
  const Ctx ctx{(int)threadIdx.x, 64, nullptr};
  const double* xk = x + ((size_t)b * (N + 1) + k) * NX;
  cent_lq_node2<false>(ctx, *dm, w, xk, u + ((size_t)b * N + k) * NU, xk + NX, par + ((size_t)b * (N + 1) + k) * NP, dts[node], nullptr, misc + (size_t)node * 8);
}
// ---- torso task-space reference of the node parameters of a centroidal handle (behind k_params): one wave per (instance, node)
__global__ __launch_bounds__(64) void k_params_cent_torso(const DevModel* __restrict__ dm, double* __restrict__ par) {
  CentWST<false>& w = *reinterpret_cast<CentWST<false>*>(hsqp_smem);
  const Ctx ctx{(int)threadIdx.x, 64, nullptr};
  cent_params_torso(ctx, *dm, w, par + (size_t)blockIdx.x * NP);
}

// ---- projection: one workgroup per (instance, node)
__global__ __launch_bounds__(PROJ_THREADS, HSQP_PROJ_WPE) void k_project(const double* __restrict__ rec, const double* __restrict__ dts, double* __restrict__ qp, long long* prof, int cent, int joint_rows, int chain) {
  ProjWS& w = *reinterpret_cast<ProjWS*>(hsqp_smem);
  const Ctx ctx{(int)threadIdx.x, PROJ_THREADS, blockIdx.x == 0 ? prof : nullptr};
  PH_TICK(ctx, 126);  // re-arm the phase clock (bucket 126 is a sink)
  project_node(ctx, w, rec + (size_t)blockIdx.x * REC_SIZE, dts[blockIdx.x], qp + (size_t)blockIdx.x * QP_SIZE, cent != 0, joint_rows != 0, chain != 0);
}

// ---- event intervals (jump_node_qp, hsqp_project.h): one workgroup per node; only launched when the grid has such intervals
__global__ __launch_bounds__(256) void k_jump(const double* __restrict__ dts, const double* __restrict__ rec, double* __restrict__ qp) {
  const int node = blockIdx.x;
  if (dts[node] != 0.0) return;
  const Ctx ctx{(int)threadIdx.x, 256, nullptr};
  jump_node_qp(ctx, rec + (size_t)node * REC_SIZE, qp + (size_t)node * QP_SIZE);
}

// ---- Riccati backward sweep + closed-loop forward sweep (dx): one workgroup per instance
template <int NXE>
__global__ __launch_bounds__(RIC_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_riccati(const DevModel* __restrict__ dm, const double* __restrict__ x_init,
                                                         const double* __restrict__ x, const double* __restrict__ par,
                                                         const double* __restrict__ qp, double* __restrict__ ric, int N,
                                                         double* __restrict__ dx, int* __restrict__ status, long long* prof,
                                                         double* __restrict__ vf, double* __restrict__ ut) {
  const int b = blockIdx.x;
  RicWS& w = *reinterpret_cast<RicWS*>(hsqp_smem);
  const Ctx ctx{(int)threadIdx.x, RIC_THREADS, blockIdx.x == 0 ? prof : nullptr};
  PH_TICK(ctx, 126);  // re-arm the phase clock (bucket 126 is a sink)
  const double* xb = x + (size_t)b * (N + 1) * NX;
  const double* parN = par + ((size_t)b * (N + 1) + N) * NP;
  const double* qpb = qp + (size_t)b * N * QP_SIZE;
  double* ricb = ric + (size_t)b * N * RIC_SIZE;
  int mybad = 0;  // projection failures of this instance (rank-deficient D)
  for (int k = threadIdx.x; k < N; k += blockDim.x)
    if (qpb[(size_t)k * QP_SIZE + QP_NUT] < 0.0) mybad = 1;
  const int bad = __syncthreads_or(mybad);
  riccati_backward<NXE>(ctx, w, dm->Qf, xb + (size_t)N * NX, parN, qpb, ricb, N, vf ? vf + (size_t)b * (N + 1) * VF_SIZE : nullptr);
  PH_TICK(ctx, 0);
  riccati_forward<NXE>(ctx, w, x_init + (size_t)b * NX, xb, qpb, ricb, N, dx + (size_t)b * (N + 1) * NX, ut + (size_t)b * N * NUT);
  PH_TICK(ctx, 10);
  // OR-accumulated over the iterations of one hsqp_iterate_device call (the host clears it once per call): a numeric failure
  // in an early iteration must not be masked by a later clean one
  if (threadIdx.x == 0) { const int st = (bad ? 1 : 0) | (w.ok ? 0 : 2); if (st) atomicOr(&status[b], st); }
}

// ---- the whole-body serial sweep on the FACTORS of [A~ | B~] (hsqp_riccati_fact.h): what every whole-body handle's serial recursion runs
__global__ __launch_bounds__(RIC_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_riccati_fact(const DevModel* __restrict__ dm, const double* __restrict__ x_init,
                                                         const double* __restrict__ x, const double* __restrict__ par,
                                                         const double* __restrict__ qp, const double* __restrict__ dts, double* __restrict__ ric, int N,
                                                         double* __restrict__ dx, int* __restrict__ status, long long* prof,
                                                         double* __restrict__ vf, double* __restrict__ ut, double* __restrict__ fj) {
  const int b = blockIdx.x;
  RicFWS& w = *reinterpret_cast<RicFWS*>(hsqp_smem);
  const Ctx ctx{(int)threadIdx.x, RIC_THREADS, blockIdx.x == 0 ? prof : nullptr};
  PH_TICK(ctx, 126);  // re-arm the phase clock (bucket 126 is a sink)
  const double* xb = x + (size_t)b * (N + 1) * NX;
  const double* parN = par + ((size_t)b * (N + 1) + N) * NP;
  const double* qpb = qp + (size_t)b * N * QP_SIZE;
  const double* dtb = dts + (size_t)b * N;
  double* ricb = ric + (size_t)b * N * RIC_SIZE;
  int mybad = 0;  // projection failures of this instance (rank-deficient D)
  for (int k = threadIdx.x; k < N; k += RIC_THREADS)
    if (qpb[(size_t)k * QP_SIZE + QP_NUT] < 0.0) mybad = 1;
  const int bad = __syncthreads_or(mybad);
  riccati_backward_fact(ctx, w, dm->Qf, xb + (size_t)N * NX, parN, qpb, dtb, ricb, N, vf ? vf + (size_t)b * (N + 1) * VF_SIZE : nullptr);
  PH_TICK(ctx, 0);
  riccati_forward_fact(ctx, w, x_init + (size_t)b * NX, xb, qpb, dtb, ricb, N, dx + (size_t)b * (N + 1) * NX, ut + (size_t)b * N * NUT, fj + (size_t)b * N * NJ);
  PH_TICK(ctx, 10);
  if (threadIdx.x == 0) { const int st = (bad ? 1 : 0) | (w.ok ? 0 : 2); if (st) atomicOr(&status[b], st); }
}

// ---- parallel-in-time backward sweep (hsqp_scan.h).  Elements: [B][N + 1][ScanEl<n>::SIZE], two buffers (ping-pong per level).
constexpr int SCAN_INIT_THREADS = 256, SCAN_COMB_THREADS = 512, SCAN_FWD_THREADS = 256;
static_assert(4 * NX <= SCAN_FWD_THREADS, "closed_loop_forward: one item per thread");
template <int n>
__global__ __launch_bounds__(SCAN_INIT_THREADS) void k_scan_init(const DevModel* __restrict__ dm, const double* __restrict__ x, const double* __restrict__ par,
                                                                 const double* __restrict__ qp, int N, double* __restrict__ el, int* __restrict__ status) {
  ScanInitWS<n>& w = *reinterpret_cast<ScanInitWS<n>*>(hsqp_smem);
  const int id = blockIdx.x, b = id / (N + 1), k = id % (N + 1);
  const Ctx ctx{(int)threadIdx.x, SCAN_INIT_THREADS, nullptr};
  __shared__ int ok;
  if (threadIdx.x == 0) ok = 1;
  __syncthreads();
  scan_init_node<n>(ctx, w, qp + ((size_t)b * N + (k < N ? k : 0)) * QP_SIZE, el + (size_t)id * ScanEl<n>::SIZE, k == N, dm->Qf,
                    x + ((size_t)b * (N + 1) + N) * NX, par + ((size_t)b * (N + 1) + N) * NP, &ok);
  __syncthreads();
  if (threadIdx.x == 0 && !ok) atomicOr(&status[b], 2);   // R~ of a stage not positive definite (the gate sees the flag)
}
template <int n>
__global__ __launch_bounds__(SCAN_COMB_THREADS) void k_scan_combine(const double* __restrict__ ein, double* __restrict__ eout, int N, int d, int* __restrict__ status,
                                                                    long long* prof) {
  ScanCombWS<n>& w = *reinterpret_cast<ScanCombWS<n>*>(hsqp_smem);
  const int id = blockIdx.x, b = id / (N + 1), k = id % (N + 1);
  const Ctx ctx{(int)threadIdx.x, SCAN_COMB_THREADS, blockIdx.x == 0 ? prof : nullptr};
  constexpr int SZ = ScanEl<n>::SIZE;
  if (k + d <= N) {
    __shared__ int ok;
    if (threadIdx.x == 0) ok = 1;
    __syncthreads();
    scan_combine<n>(ctx, w, ein + (size_t)id * SZ, ein + (size_t)(id + d) * SZ, eout + (size_t)id * SZ, &ok);
    __syncthreads();
    if (threadIdx.x == 0 && !ok) atomicOr(&status[b], 2);
  } else {
    for (int i = threadIdx.x; i < SZ; i += blockDim.x) eout[(size_t)id * SZ + i] = ein[(size_t)id * SZ + i];
  }
}
// one stage of the Riccati code per node, started from the value function of node k + 1: the scanned one (element el; vf_in = null)
// or, in the refinement pass, the one the first pass computed (vf_in: [N + 1][VF_SIZE] per instance).  Applying the exact Riccati map
// once more contracts the scan's rounding error (oracle-level experiment: 5e-8 -> 5e-9 on the worst whole-body case).
template <int n>
__global__ __launch_bounds__(RIC_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_scan_gains(const DevModel* __restrict__ dm, const double* __restrict__ x, const double* __restrict__ par,
                                                            const double* __restrict__ qp, const double* __restrict__ el, const double* __restrict__ vf_in,
                                                            double* __restrict__ ric, int N, int* __restrict__ status, double* __restrict__ vf, double* __restrict__ acl) {
  RicWS& w = *reinterpret_cast<RicWS*>(hsqp_smem);
  const int node = blockIdx.x, b = node / N, k = node % N;
  const Ctx ctx{(int)threadIdx.x, RIC_THREADS, nullptr};
  const double* en = el + ((size_t)b * (N + 1) + k + 1) * ScanEl<n>::SIZE;
  const double* vn = vf_in ? vf_in + ((size_t)b * (N + 1) + k + 1) * VF_SIZE : nullptr;
  const double* q = qp + (size_t)node * QP_SIZE;
  riccati_backward<n>(ctx, w, dm->Qf, x + ((size_t)b * (N + 1) + N) * NX, par + ((size_t)b * (N + 1) + N) * NP, q, ric + (size_t)node * RIC_SIZE, 1,
                      vf ? vf + ((size_t)b * (N + 1) + k) * VF_SIZE : nullptr, vn ? vn : en + ScanEl<n>::J, vn ? vn + NX * NX : en + ScanEl<n>::ETA, k == N - 1,
                      vn ? 1.0 : -1.0, vn ? NX : n);
  if (threadIdx.x == 0) {
    const int st = (q[QP_NUT] < 0.0 ? 1 : 0) | (w.ok ? 0 : 2);
    if (st) atomicOr(&status[b], st);
  }
  if (acl) closed_loop_record<n>(ctx, w, acl + (size_t)node * ACL_SIZE<n>);   // the last pass: closed loop of this stage for the roll-out
}
template <int n>
__global__ __launch_bounds__(SCAN_FWD_THREADS) void k_scan_forward(const double* __restrict__ x_init, const double* __restrict__ x, const double* __restrict__ acl,
                                                              int N, double* __restrict__ dx) {
  RicWS& w = *reinterpret_cast<RicWS*>(hsqp_smem);
  const int b = blockIdx.x;
  const Ctx ctx{(int)threadIdx.x, SCAN_FWD_THREADS, nullptr};
  closed_loop_forward<n>(ctx, w, x_init + (size_t)b * NX, x + (size_t)b * (N + 1) * NX, acl + (size_t)b * N * ACL_SIZE<n>, N, dx + (size_t)b * (N + 1) * NX);
}

// ---- two-level (segmented) backward sweep (hsqp_segment.h): B P workgroups, segment p of instance b covers the stages [p N / P, (p + 1) N / P)
constexpr int SEG_ACC_THREADS = 512;
// 1a: Riccati recursion over the segment from J = 0, eta = 0 (zero: n x n + n zeros): gains and (L^-1)^T of every stage, (J, s) at its first node
template <int n>
__global__ __launch_bounds__(RIC_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_seg_elem_ric(const DevModel* __restrict__ dm, const double* __restrict__ x, const double* __restrict__ par,
                                                              const double* __restrict__ qp, const double* __restrict__ zero, double* __restrict__ ric_tmp,
                                                              double* __restrict__ linv, double* __restrict__ vf0, int N, int P, int* __restrict__ status) {
  RicWS& w = *reinterpret_cast<RicWS*>(hsqp_smem);
  const int seg = blockIdx.x, b = seg / P, p = seg % P, k0 = seg_bound(p, N, P), L = seg_bound(p + 1, N, P) - k0;
  const Ctx ctx{(int)threadIdx.x, RIC_THREADS, nullptr};
  const size_t s0 = (size_t)b * N + k0;
  riccati_backward<n>(ctx, w, dm->Qf, x + ((size_t)b * (N + 1) + N) * NX, par + ((size_t)b * (N + 1) + N) * NP, qp + s0 * QP_SIZE, ric_tmp + s0 * RIC_SIZE, L,
                      vf0 + (size_t)seg * VF_SIZE, zero, zero + n * n, false, 1.0, n, linv + s0 * LDB * LDB, 1);
  if (threadIdx.x == 0 && !w.ok) atomicOr(&status[b], 2);
}
// 1b: the (A, b, C) part of the segment's element by prepending its stages; el: [B][P + 1][ScanEl<n>::SIZE]
template <int n>
__global__ __launch_bounds__(SEG_ACC_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_seg_accumulate(const double* __restrict__ qp, const double* __restrict__ ric_tmp, const double* __restrict__ linv,
                                                                   const double* __restrict__ vf0, int N, int P, double* __restrict__ el) {
  SegAccWS& w = *reinterpret_cast<SegAccWS*>(hsqp_smem);
  const int seg = blockIdx.x, b = seg / P, p = seg % P, k0 = seg_bound(p, N, P), L = seg_bound(p + 1, N, P) - k0;
  const Ctx ctx{(int)threadIdx.x, SEG_ACC_THREADS, nullptr};
  const size_t s0 = (size_t)b * N + k0;
  seg_accumulate<n>(ctx, w, qp + s0 * QP_SIZE, ric_tmp + s0 * RIC_SIZE, linv + s0 * LDB * LDB, L, vf0 + (size_t)seg * VF_SIZE,
                    el + ((size_t)b * (P + 1) + p) * ScanEl<n>::SIZE);
}
// the terminal cost as element P of every instance
template <int n>
__global__ __launch_bounds__(256) void k_seg_terminal(const DevModel* __restrict__ dm, const double* __restrict__ x, const double* __restrict__ par, int N, int P,
                                                      double* __restrict__ el) {
  const int b = blockIdx.x;
  const Ctx ctx{(int)threadIdx.x, 256, nullptr};
  scan_terminal_element<n>(ctx, el + ((size_t)b * (P + 1) + P) * ScanEl<n>::SIZE, dm->Qf, x + ((size_t)b * (N + 1) + N) * NX, par + ((size_t)b * (N + 1) + N) * NP);
}
// 3: the gains of the segment's stages from the value function at its end (suffix element p + 1 of the scanned array)
template <int n>
__global__ __launch_bounds__(RIC_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_seg_riccati(const DevModel* __restrict__ dm, const double* __restrict__ x, const double* __restrict__ par,
                                                             const double* __restrict__ qp, const double* __restrict__ el, double* __restrict__ ric, int N, int P,
                                                             int* __restrict__ status, double* __restrict__ vf, int vf_mode) {
  RicWS& w = *reinterpret_cast<RicWS*>(hsqp_smem);
  const int seg = blockIdx.x, b = seg / P, p = seg % P, k0 = seg_bound(p, N, P), L = seg_bound(p + 1, N, P) - k0;
  const Ctx ctx{(int)threadIdx.x, RIC_THREADS, nullptr};
  const size_t s0 = (size_t)b * N + k0;
  const double* en = el + ((size_t)b * (P + 1) + p + 1) * ScanEl<n>::SIZE;
  int mybad = 0;
  for (int k = threadIdx.x; k < L; k += blockDim.x)
    if (qp[(s0 + k) * QP_SIZE + QP_NUT] < 0.0) mybad = 1;
  const int bad = __syncthreads_or(mybad);
  // vf: this segment writes the value functions of its nodes k0 .. k0 + L - 1 (vf_mode 0: all — the KKT report; 2: its first node and its last
  // stage's node — what the gate's boundary stages need); the last segment also the terminal node
  riccati_backward<n>(ctx, w, dm->Qf, x + ((size_t)b * (N + 1) + N) * NX, par + ((size_t)b * (N + 1) + N) * NP, qp + s0 * QP_SIZE, ric + s0 * RIC_SIZE, L,
                      vf ? vf + ((size_t)b * (N + 1) + k0) * VF_SIZE : nullptr, en + ScanEl<n>::J, en + ScanEl<n>::ETA, p == P - 1, -1.0, n, nullptr, vf_mode);
  if (threadIdx.x == 0) { const int st = (bad ? 1 : 0) | (w.ok ? 0 : 2); if (st) atomicOr(&status[b], st); }
}
// 4: the roll-out alone (k_riccati runs it behind its backward sweep)
template <int NXE>
__global__ __launch_bounds__(RIC_THREADS) void k_ric_forward(const double* __restrict__ x_init, const double* __restrict__ x, const double* __restrict__ qp,
                                                             const double* __restrict__ ric, int N, double* __restrict__ dx, double* __restrict__ ut) {
  RicWS& w = *reinterpret_cast<RicWS*>(hsqp_smem);
  const int b = blockIdx.x;
  const Ctx ctx{(int)threadIdx.x, RIC_THREADS, nullptr};
  riccati_forward<NXE>(ctx, w, x_init + (size_t)b * NX, x + (size_t)b * (N + 1) * NX, qp + (size_t)b * N * QP_SIZE, ric + (size_t)b * N * RIC_SIZE, N,
                       dx + (size_t)b * (N + 1) * NX, ut + (size_t)b * N * NUT);
}

// ---- input recovery + step: one 64-thread workgroup per (instance, node); the last node of an instance also steps x_N.
//      ut_given: ut holds ut = k + K dx of every node already (the serial roll-out writes it as it goes) and the gains are not read
__global__ __launch_bounds__(64) void k_step(const double* __restrict__ qp, const double* __restrict__ ric, const double* __restrict__ dx,
                                             const double* __restrict__ x, const double* __restrict__ u, int N, double alpha,
                                             double* ut, double* __restrict__ du, double* __restrict__ x_new,
                                             double* __restrict__ u_new, double* __restrict__ info, int ut_given, const double* __restrict__ fj) {
  __shared__ StepWS w;
  const int node = blockIdx.x, b = node / N, k = node % N;
  const Ctx ctx{(int)threadIdx.x, 64, nullptr};
  const size_t xo = ((size_t)b * (N + 1) + k) * NX, uo = (size_t)node * NU;
  step_node(ctx, w, qp + (size_t)node * QP_SIZE, ric + (size_t)node * RIC_SIZE, dx + xo, x + xo, u + uo, alpha, ut + (size_t)node * NUT,
            du + uo, x_new + xo, u_new + uo, info + (size_t)node * 4, ut_given ? ut + (size_t)node * NUT : nullptr, fj ? fj + (size_t)node * NJ : nullptr);
  if (k == N - 1)
    for (int i = threadIdx.x; i < NX; i += blockDim.x) x_new[xo + NX + i] = x[xo + NX + i] + alpha * dx[xo + NX + i];
}

// ---- input recovery + full-step trial + its value pass in ONE kernel (the first evaluation of every iteration): k_step alone is
//      HBM-bound (35 KB of gains / projection per node, 3.8 TB/s) while the value pass is issue-bound, so the reads of the one hide
//      under the arithmetic of the other.  One wave per node; the step's scratch aliases the stage workspace (dead until the value
//      pass starts), the stepped (x, u, x_next) go straight into the value pass's input slots — results are bit-identical to
//      k_step followed by k_lq<false>.
__global__ __launch_bounds__(LQV_THREADS, HSQP_LQV_WPE) void k_step_value(const DevModel* __restrict__ dm, const double* __restrict__ qp, const double* __restrict__ ric,
                                                                          const double* __restrict__ dx, const double* __restrict__ x, const double* __restrict__ u,
                                                                          const double* __restrict__ par, const double* __restrict__ dts, int N, double alpha,
                                                                          double* ut, double* __restrict__ du, double* __restrict__ x_new,
                                                                          double* __restrict__ u_new, double* __restrict__ info, double* __restrict__ misc, long long* prof,
                                                                          int ut_given, const double* __restrict__ fj) {
  const int node = blockIdx.x, b = node / N, k = node % N;
  LqWST<false>& w = *reinterpret_cast<LqWST<false>*>(hsqp_smem);
  static_assert(sizeof(StepWS) <= sizeof(w.st), "the step scratch aliases the stage workspace");
  StepWS& sw = *reinterpret_cast<StepWS*>(&w.st);
  const Ctx ctx{(int)threadIdx.x, LQV_THREADS, blockIdx.x == 0 ? prof : nullptr};   // phase profile (profile builds): slot 3, labelled k_lq<false>
  PH_TICK(ctx, 126);
  const size_t xo = ((size_t)b * (N + 1) + k) * NX, uo = (size_t)node * NU;
  step_node(ctx, sw, qp + (size_t)node * QP_SIZE, ric + (size_t)node * RIC_SIZE, dx + xo, x + xo, u + uo, alpha, ut + (size_t)node * NUT,
            du + uo, x_new + xo, u_new + uo, info + (size_t)node * 4, ut_given ? ut + (size_t)node * NUT : nullptr, fj ? fj + (size_t)node * NJ : nullptr);
  WG_SYNC(ctx);
  WG_FOR(ctx, i, NX + NU + NX) {
    if (i < NX) w.nw.x[i] = x[xo + i] + alpha * sw.dx[i];
    else if (i < NX + NU) w.nw.u[i - NX] = u[uo + i - NX] + alpha * sw.du[i - NX];
    else {
      const int j = i - NX - NU;
      const double v = x[xo + NX + j] + alpha * dx[xo + NX + j];
      w.xnext[j] = v;
      if (k == N - 1) x_new[xo + NX + j] = v;   // the last node of an instance also steps x_N
    }
  }
  WG_SYNC(ctx);
  lq_node<false, true>(ctx, *dm, w, nullptr, nullptr, nullptr, par + ((size_t)b * (N + 1) + k) * NP, dts[node], nullptr, misc + (size_t)node * 8);
}

// (x, u) of the 16 nodes of a wave -> LDS: lane i < 64 takes entries i and i + 64 of every node's [x; u] row, 32 loads per lane issued back to back;
// the node arithmetic is the wave's (scalar: node0 through readfirstlane, no division per element).  The first form — one entry per lane and
// iteration with its indices by division — kept one load in flight per iteration and spent ~150 instructions per element on the indices.
__device__ __forceinline__ void quad_load_xu(double (*xs)[NX], double (*us)[NU], const double* __restrict__ x, const double* __restrict__ u, int node0_, int nodes,
                                             int N, int lane) {
  static_assert(QV_NODES == 16 && NZ <= 128 && NX <= 64, "two entries per lane and node");
  const int node0 = __builtin_amdgcn_readfirstlane(node0_);
  int b = node0 / N, k = node0 - b * N;
  double t0[QV_NODES], t1[QV_NODES];
#pragma unroll
  for (int n2 = 0; n2 < QV_NODES; ++n2) {
    const bool pad = node0 + n2 >= nodes;                          // a padding quad repeats the last node
    const int nd = pad ? nodes - 1 : node0 + n2;
    const size_t xrow = pad ? (size_t)(nodes - 1) + (nodes - 1) / N : (size_t)b * (N + 1) + k;
    t0[n2] = lane < NX ? x[xrow * NX + lane] : u[(size_t)nd * NU + (lane - NX)];
    t1[n2] = u[(size_t)nd * NU + (lane + 64 - NX < NU ? lane + 64 - NX : NU - 1)];
    if (++k == N) { k = 0; ++b; }
  }
#pragma unroll
  for (int n2 = 0; n2 < QV_NODES; ++n2) {
    if (lane < NX) xs[n2][lane] = t0[n2]; else us[n2][lane - NX] = t0[n2];
    if (lane + 64 < NZ) us[n2][lane + 64 - NX] = t1[n2];
  }
}

// ---- whole-body value pass on quads of lanes (hsqp_lqv.h): a wave evaluates QV_NODES nodes, one lane per limb; misc as k_lq<false> writes it.
//      Nodes of instances whose line search is over (ls) are evaluated with the rest of their wave but not written.
__global__ __launch_bounds__(QV_THREADS * QV_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_value_quad(const DevModel* __restrict__ dm, const double* __restrict__ x,
                                                          const double* __restrict__ u, const double* __restrict__ par, const double* __restrict__ dts, int N, int nodes,
                                                          double* __restrict__ misc, const LsState* __restrict__ ls) {
  __shared__ QvWS ws;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nn = lane >> 2, L = lane & 3;
  auto& w = ws.wv[wave];
  const int node0 = (blockIdx.x * QV_WAVES + wave) * QV_NODES, node = node0 + nn < nodes ? node0 + nn : nodes - 1;   // (a padding quad repeats the last node)
  const int b = node / N, k = node % N;
  const bool live = node0 + nn < nodes && (!ls || ls[b].active);
  if (__syncthreads_or(live) == 0) return;
  const Ctx ctx{(int)threadIdx.x, QV_THREADS * QV_WAVES, nullptr};
  qv_load_const(ctx, *dm, ws.k, [] {});
  quad_load_xu(w.x, w.u, x, u, node0, nodes, N, lane);
  __syncthreads();
  const double* xs = w.x[nn];
  const double* us = w.u[nn];
  const double* pn = par + ((size_t)b * (N + 1) + k) * NP;
  const double dt = dts[node];
#if defined(__HIP_DEVICE_COMPILE__)
  auto quad_sum = [](double v) { v += quad_perm_f64<0xB1>(v); v += quad_perm_f64<0x4E>(v); return v; };   // the sum over the node's four lanes, in all of them
#else
  auto quad_sum = [](double v) { return v; };
#endif
  QvCarry c;
  qv_carry_init(c);
  double cost = 0.0, eq = 0.0;
  auto stage = [&](int s) {
    double part[16];
    qv_limb_stage(*dm, ws.k, xs, us, L, s, dt, c, part, w.cp[nn]);
#pragma unroll
    for (int e = 0; e < 16; ++e) part[e] = quad_sum(part[e]);
    qv_base_solve(part, xs, s, dt, c);
  };
  // the first stage apart from the loop: what it captures for the node terms (the foot frame, 33 doubles per lane) is dead before the second
  stage(0);
  __syncthreads();   // orders the collision points of a node's four lanes before their readers
  qv_node_terms(*dm, xs, us, pn, L, c, w.cp[nn], cost, eq);
#pragma unroll 1
  for (int s = 1; s < 4; ++s) stage(s);
  double dyn = qv_defect(xs, us, x + ((size_t)b * (N + 1) + k + 1) * NX, L, dt, c);
  cost = quad_sum(cost); eq = quad_sum(eq); dyn = quad_sum(dyn);
  if (live && L == 0) qv_write_misc(pn, dt, cost, eq, dyn, misc + (size_t)node * 8);
}

// ---- whole-body LQ approximation on limb lanes (hsqp_lql.h), kernel 1 of 3: the rigid-body model and its Jacobian at the four RK4 stages.  A wave
//      evaluates QL_NODES nodes, one lane per limb; writes REC_GS (transposed) and REC_AS of the node's record.
#ifndef HSQP_LQ_SPLIT_DEFAULT
#define HSQP_LQ_SPLIT_DEFAULT 2   /* node ranges of the limb-lane LQ kernels on streams of their own (hsqp_iterate_device) */
#endif
#ifndef HSQP_QL_WAVES
#define HSQP_QL_WAVES 2          /* waves per workgroup: they share the body constants */
#endif
#ifndef HSQP_QL_WPE
#define HSQP_QL_WPE 1            /* waves per SIMD the register budget is cut for (1: 512 registers per lane) */
#endif
#ifndef HSQP_QR_WPE
#define HSQP_QR_WPE 1
#endif
constexpr int QL_WAVES = HSQP_QL_WAVES;
struct QlWS {
  QvConst k;
  struct {
    double x[QL_NODES][NX], u[QL_NODES][NU];
    double csn[QL_MAXLEN][QL_THREADS][2];   // cos / sin of the joints a lane passed on its way to the leaf, for the way back
  } wv[QL_WAVES];
};
#if defined(__HIP_DEVICE_COMPILE__)
#define QL_QUAD_OPS                                                                                                                                    \
  auto quad_sum = [](double v) { v += quad_perm_f64<0xB1>(v); v += quad_perm_f64<0x4E>(v); return v; }; /* the sum over the node's four lanes, in all of them */ \
  auto quad_x1 = [](double v) { return quad_perm_f64<0xB1>(v); }; /* the value of lane L ^ 1 / L ^ 2 / L ^ 3 of the quad */                       \
  auto quad_x2 = [](double v) { return quad_perm_f64<0x4E>(v); };                                                                                 \
  auto quad_x3 = [](double v) { return quad_perm_f64<0x1B>(v); };                                                                                 \
  (void)quad_x1; (void)quad_x2; (void)quad_x3
#else
#define QL_QUAD_OPS                             \
  auto quad_sum = [](double v) { return v; };  \
  auto quad_x1 = quad_sum, quad_x2 = quad_sum, quad_x3 = quad_sum; (void)quad_x1; (void)quad_x2; (void)quad_x3
#endif
__global__ __launch_bounds__(QL_THREADS * QL_WAVES) __attribute__((amdgpu_waves_per_eu(HSQP_QL_WPE, HSQP_QL_WPE))) void k_lq_limb(
    const DevModel* __restrict__ dm, const double* __restrict__ x, const double* __restrict__ u, const double* __restrict__ dts, int N, int nodes,
    double* __restrict__ rec, long long* prof, int node_base) {   // the launch covers the nodes [node_base, nodes)
  __shared__ QlWS ws;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nn = lane >> 2, L = lane & 3;
  auto& w = ws.wv[wave];
  const int node0 = node_base + (blockIdx.x * QL_WAVES + wave) * QL_NODES, node = node0 + nn < nodes ? node0 + nn : nodes - 1;   // (a padding quad repeats the last node)
  const bool live = node0 + nn < nodes;
  const Ctx ctx{(int)threadIdx.x, QL_THREADS * QL_WAVES, blockIdx.x == 0 ? prof : nullptr};
  PH_TICK(ctx, 126);
  qv_load_const(ctx, *dm, ws.k, [] {});
  quad_load_xu(w.x, w.u, x, u, node0, nodes, N, lane);
  __syncthreads();
  const double* xs = w.x[nn];
  const double* us = w.u[nn];
  const double dt = dts[node];
  double* grec = rec + (size_t)node * REC_SIZE;
  double* csn = &w.csn[0][lane][0];
  constexpr int CSN_LD = QL_THREADS * 2;
  QL_QUAD_OPS;
  const QlLimb lb = ql_limb(*dm, L);
  const int max_len = lb.max_len;
  const bool own_root = (dm->limb_own[L] & 1u) != 0;
  QlCarry c;
  for (int i = 0; i < 6; ++i) { c.vb[i] = 0.0; c.ap[i] = 0.0; }
#pragma unroll 1
  for (int s = 0; s < 4; ++s) {
    QlBaseKin bk;
    QlState st;
    QlShared sh;
    double rP[3], Ff[3];
    PH_TICK(ctx, 0);
    ql_base_kin(*dm, xs, s, dt, c, bk);
    {
      double part[16];
      ql_forward(*dm, ws.k, lb, xs, us, L, s, dt, bk, st, part, csn, CSN_LD, rP, Ff);
#pragma unroll
      for (int e = 0; e < 16; ++e) part[e] = quad_sum(part[e]);
      ql_base_solve(part, bk, sh);
    }
    PH_TICK(ctx, 1);
    if (live && L == 0)
      for (int i = 0; i < 6; ++i) grec[REC_AS + 6 * s + i] = sh.ab[i];
    double* gs = grec + REC_GS + (size_t)s * LDJ * GT_LD;
    auto putg = [&](int col, const double* g) {
      if (!live) return;
      ql_st2(gs + col * GT_LD, g[0], g[1]); ql_st2(gs + col * GT_LD + 2, g[2], g[3]); ql_st2(gs + col * GT_LD + 4, g[4], g[5]);
    };
    double cmp[NCMP];
#pragma unroll
    for (int e = 0; e < NCMP; ++e) cmp[e] = 0.0;
#pragma unroll 1
    for (int t = max_len - 1; t >= 0; --t) {
      // limbs that join below this step hand over their composites (the G1 tree: the two arm lanes, once, at the torso)
      const unsigned mg = dm->limb_merge[t][L];
      const unsigned anym = (unsigned)dm->limb_merge[t][0] | dm->limb_merge[t][1] | dm->limb_merge[t][2] | dm->limb_merge[t][3];
      if (anym & 2u) {
        const double m = (mg & 2u) ? 1.0 : 0.0;
#pragma unroll
        for (int e = 0; e < NCMP; ++e) cmp[e] += m * quad_x1(cmp[e]);
      }
      if (anym & 4u) {
        const double m = (mg & 4u) ? 1.0 : 0.0;
#pragma unroll
        for (int e = 0; e < NCMP; ++e) cmp[e] += m * quad_x2(cmp[e]);
      }
      if (anym & 8u) {
        const double m = (mg & 8u) ? 1.0 : 0.0;
#pragma unroll
        for (int e = 0; e < NCMP; ++e) cmp[e] += m * quad_x3(cmp[e]);
      }
      ql_back_step(*dm, ws.k, lb, xs, us, L, s, dt, t, st, cmp, sh, csn, CSN_LD, rP, Ff, putg);
    }
    PH_TICK(ctx, 3);
    // ---- the base: composite of the whole robot = the limbs that own their root-side body + the base body; its columns
    {
      const double r0[3] = {0.0, 0.0, 0.0};
      double In[10], own[NCMP];
      ql_inertia(ws.k, 0, bk.R, r0, In);
      ql_body_comp(In, bk.vl[2], bk.al[2], own);
      const double mk = own_root ? 1.0 : 0.0;
#pragma unroll
      for (int e = 0; e < NCMP; ++e) cmp[e] = quad_sum(mk * cmp[e]) + own[e];
    }
    double dext_e[9];
#pragma unroll
    for (int jc = 0; jc < 3; ++jc) {
      double dr[3], tt[3];
      v3_cross(bk.w[jc], rP, dr);
      v3_cross(dr, Ff, tt);   // (a lane without a foot carries rP = Ff = 0)
      for (int i = 0; i < 3; ++i) dext_e[3 * jc + i] = quad_sum(tt[i]);
    }
    ql_base_columns(*dm, L, bk, cmp, sh, dext_e, rP, putg);
    ql_carry_advance(xs, s, dt, sh, c);
    PH_TICK(ctx, 4);
  }
}

// ---- ... kernel 2 of 3: the node terms of RK4 stage 1 — values, penalties, and the residual / equality rows of every column, formed by the lane
//      that owns the column from the stage-1 Jacobian columns kernel 1 left in the record (a kinematics-only walk of the limb).  Writes REC_J,
//      REC_CDE (transposed), REC_RHO, REC_D, REC_GD, REC_FLOW, REC_MISC.
struct QrWS {
  QvConst k;
  struct {
    double x[QL_NODES][NX], u[QL_NODES][NU];
    double csn[QL_MAXLEN][QL_THREADS][2];
    QlNodeLds nl[QL_NODES];                 // what the four lanes of a node share (foot frames, collision points, row scalings)
  } wv[QL_WAVES];
};
static_assert(sizeof(QrWS) * (4 / QL_WAVES) <= 163840, "four waves per CU");
__global__ __launch_bounds__(QL_THREADS * QL_WAVES) __attribute__((amdgpu_waves_per_eu(HSQP_QR_WPE, HSQP_QR_WPE))) void k_lq_rows(
    const DevModel* __restrict__ dm, const double* __restrict__ x, const double* __restrict__ u, const double* __restrict__ par, const double* __restrict__ dts,
    int N, int nodes, double* __restrict__ rec, long long* prof, int node_base, int defect) {
  __shared__ QrWS ws;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nn = lane >> 2, L = lane & 3;
  auto& w = ws.wv[wave];
  const int node0 = node_base + (blockIdx.x * QL_WAVES + wave) * QL_NODES, node = node0 + nn < nodes ? node0 + nn : nodes - 1;
  const int b = node / N, k = node % N;
  const bool live = node0 + nn < nodes;
  const Ctx ctx{(int)threadIdx.x, QL_THREADS * QL_WAVES, blockIdx.x == 0 ? prof : nullptr};   // phase profile (profile builds): slot 3, ids 10..15
  PH_TICK(ctx, 126);
  qv_load_const(ctx, *dm, ws.k, [] {});
  quad_load_xu(w.x, w.u, x, u, node0, nodes, N, lane);
  __syncthreads();
  PH_TICK(ctx, 10);
  const double* xs = w.x[nn];
  const double* us = w.u[nn];
  const double* pn = par + ((size_t)b * (N + 1) + k) * NP;
  QlNodeLds& nl = w.nl[nn];
  const double dt = dts[node];
  double* grec = rec + (size_t)node * REC_SIZE;
  double* csn = &w.csn[0][lane][0];
  constexpr int CSN_LD = QL_THREADS * 2;
  QL_QUAD_OPS;
  const QlLimb lb = ql_limb(*dm, L);
  const int max_len = lb.max_len;
  QlCarry c;
  for (int i = 0; i < 6; ++i) { c.vb[i] = 0.0; c.ap[i] = 0.0; }
  QlBaseKin bk;
  QlState st;
  QlShared sh;
  QlRows rw;
  ql_base_kin(*dm, xs, 0, dt, c, bk);
  ql_kin_to_leaf(*dm, ws.k, lb, xs, us, L, bk, csn, CSN_LD, st, nl);
  ql_shared_from_record(bk, grec, sh);
  double dsq = 0.0;   // defect = 1 (the chain is fused into k_project, k_lq_chain is not launched): the node's defect and its squared norm on the four lanes, in front of every store of the kernel
  if (defect) dsq = ql_defect_lane(xs, us, x + ((size_t)b * (N + 1) + k + 1) * NX, grec + REC_AS, dt, L, grec, live);
  PH_TICK(ctx, 11);
  {
    // the lane's foot and its share of the cost (pass A), then what needs every lane's pass A (pass B).  The four lanes of a node sit in one
    // wave: a wave-level fence orders their LDS traffic
    double cost, eq, cost2;
    int any;
    WV_SYNC();
    ql_terms_a(*dm, xs, us, pn, L, dt, sh, nl, grec, live, cost, eq);
    WV_SYNC();
    ql_terms_b(*dm, xs, us, pn, L, dt, sh, nl, grec, live, cost2, any);
    WV_SYNC();
    cost = quad_sum(cost + cost2); eq = quad_sum(eq);
    const int coll = quad_sum((double)any) > 0.0 ? 1 : 0;
    ql_rows_setup(*dm, pn, L, dt, coll, rw);
    if (live && L == 0) ql_write_misc(pn, dt, cost, eq, coll, grec + REC_MISC);
    if (defect) { dsq = quad_sum(dsq); if (live && L == 0) grec[REC_MISC + 3] = (dt > 0.0 ? dt : 1.0) * dsq; }
  }
  PH_TICK(ctx, 12);
  const double* gs = grec + REC_GS;
  double gcur[3][6];   // the three stage-Jacobian columns of the step about to run (ql_rows_back_step keeps them one step ahead)
#pragma unroll
  for (int q = 0; q < 3; ++q) ql_rows_fetch(gs, ql_rows_column(lb, max_len - 1, q), gcur[q]);
#pragma unroll 1
  for (int t = max_len - 1; t >= 0; --t) ql_rows_back_step(*dm, ws.k, lb, rw, nl, xs, us, t, st, csn, CSN_LD, bk.w, gs, gcur, grec, live);
  PH_TICK(ctx, 13);
  ql_rows_base(*dm, rw, nl, us, L, bk, sh, gs, grec, live);
  PH_TICK(ctx, 14);
}

// ---- ... kernel 3 of 3: the RK4 chain and the defect: one workgroup per (instance, node), a lane per column of [A|B] (lq_chain_node, hsqp_lql.h)
#ifndef HSQP_LQC_WPE
#define HSQP_LQC_WPE 3
#endif
constexpr int LQC_THREADS = 128;   // (the 64 defect items must be one wave: lq_chain_node sums their squares with a butterfly)
__global__ __launch_bounds__(LQC_THREADS, HSQP_LQC_WPE) void k_lq_chain(const double* __restrict__ x, const double* __restrict__ u, const double* __restrict__ dts, int N,
                                                         double* __restrict__ rec, int node_base, int columns) {
  __shared__ LqChainWS w;
  const int node = node_base + blockIdx.x, b = node / N, k = node % N;
  const Ctx ctx{(int)threadIdx.x, LQC_THREADS, nullptr};
  const double* xk = x + ((size_t)b * (N + 1) + k) * NX;
  lq_chain_node(ctx, w, xk, u + (size_t)node * NU, xk + NX, dts[node], rec + (size_t)node * REC_SIZE, columns != 0);
}

// ---- line search: per-instance reduction of the step info (+ terminal node), state initialisation
__device__ inline void ls_init_instance(int b, const DevModel* __restrict__ dm, const double* __restrict__ info, const double* __restrict__ x,
                                        const double* __restrict__ dx, const double* __restrict__ par, int N, LsState* __restrict__ ls) {
  __shared__ double red[3][64];
  double a = 0.0, nx2 = 0.0, nu2 = 0.0;
  for (int k = threadIdx.x; k < N; k += blockDim.x) {
    const double* m = info + ((size_t)b * N + k) * 4;
    a += m[0]; nx2 += m[1]; nu2 += m[2];
  }
  if (threadIdx.x < NX) {   // terminal node: gradient of the terminal cost times dx_N
    const size_t o = ((size_t)b * (N + 1) + N) * NX + threadIdx.x;
    const double d = dx[o];
    a += dm->Qf[threadIdx.x] * (x[o] - par[((size_t)b * (N + 1) + N) * NP + HSQP_P_XDES + threadIdx.x]) * d;
    nx2 += d * d;
  }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = nx2; red[2][threadIdx.x] = nu2;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < 64; ++i) { s0 += red[0][i]; s1 += red[1][i]; s2 += red[2][i]; }
    LsState st;
    st.alpha = 1.0; st.armijo = s0; st.dxnorm = sqrt(s1); st.dunorm = sqrt(s2);
    st.active = 1; st.dirty = 0; st.step_type = HSQP_STEP_FULL; st.trials = 0;
    ls[b] = st;
  }
}
__global__ __launch_bounds__(64) void k_ls_init(const DevModel* __restrict__ dm, const double* __restrict__ info, const double* __restrict__ x,
                                                const double* __restrict__ dx, const double* __restrict__ par, int N, LsState* __restrict__ ls) {
  ls_init_instance(blockIdx.x, dm, info, x, dx, par, N, ls);
}

// ---- line search: decide the pending trials; counts[0] = instances that need a new trajectory, counts[1] = still active
__global__ void k_ls_decide(LsSettings st, const hsqp_perf* __restrict__ base, hsqp_perf* __restrict__ trial, int B, LsState* __restrict__ ls,
                            int* __restrict__ counts) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  LsState s = ls[b];
  if (!s.active) { if (s.dirty) { s.dirty = 0; ls[b] = s; } return; }
  ls_decide(st, base[b], trial[b], s);
  if (s.step_type == HSQP_STEP_ZERO && !s.active) trial[b] = base[b];   // no step: the performance index is the baseline's
  if (s.dirty) { s.active = s.step_type == HSQP_STEP_ZERO ? 0 : 1; atomicAdd(&counts[0], 1); }
  if (s.active) atomicAdd(&counts[1], 1);
  ls[b] = s;
}

// ---- line search: recompute x_new, u_new of the instances whose step length changed
__global__ __launch_bounds__(64) void k_ls_retake(const double* __restrict__ x, const double* __restrict__ u, const double* __restrict__ dx,
                                                  const double* __restrict__ du, int N, const LsState* __restrict__ ls,
                                                  double* __restrict__ x_new, double* __restrict__ u_new) {
  const int node = blockIdx.x, b = node / N, k = node % N;
  if (!ls[b].dirty) return;
  const double alpha = ls[b].alpha;
  const size_t xo = ((size_t)b * (N + 1) + k) * NX, uo = (size_t)node * NU;
  for (int i = threadIdx.x; i < NX; i += blockDim.x) x_new[xo + i] = x[xo + i] + alpha * dx[xo + i];
  for (int i = threadIdx.x; i < NU; i += blockDim.x) u_new[uo + i] = u[uo + i] + alpha * du[uo + i];
  if (k == N - 1)
    for (int i = threadIdx.x; i < NX; i += blockDim.x) x_new[xo + NX + i] = x[xo + NX + i] + alpha * dx[xo + NX + i];
}

// ---- KKT residual of the projected QP (reporting only, not part of an SQP step): one workgroup per (instance, node), the
//      per-instance maxima by atomic max on the bit patterns of the (non-negative) residuals
__global__ __launch_bounds__(256, 4) void k_kkt(const double* __restrict__ x_init, const double* __restrict__ x, const double* __restrict__ qp,
                                             const double* __restrict__ vf, const double* __restrict__ dx, const double* __restrict__ ut, int N,
                                             double* __restrict__ kkt, double* __restrict__ ginf) {
  __shared__ KktWS w;
  __shared__ double dx0[NX], r2[2], gred[128];
  const int node = blockIdx.x, b = node / N, k = node % N;
  const Ctx ctx{(int)threadIdx.x, 256, nullptr};
  if (k == 0) {
    for (int i = threadIdx.x; i < NX; i += blockDim.x) dx0[i] = x_init[(size_t)b * NX + i] - x[(size_t)b * (N + 1) * NX + i];
    __syncthreads();
  }
  const double* vfb = vf + (size_t)b * (N + 1) * VF_SIZE;
  const double* dxb = dx + (size_t)b * (N + 1) * NX;
  kkt_node(ctx, w, qp + (size_t)node * QP_SIZE, vfb + (size_t)k * VF_SIZE, vfb + (size_t)(k + 1) * VF_SIZE, dxb + (size_t)k * NX,
           dxb + (size_t)(k + 1) * NX, ut + (size_t)node * NUT, k == 0 ? dx0 : nullptr, r2);
  if (threadIdx.x < 2) atomicMax(reinterpret_cast<unsigned long long*>(kkt + 2 * b + threadIdx.x), (unsigned long long)__double_as_longlong(r2[threadIdx.x]));
  // |g|_inf of the projected QP: q~ (58) and r~ (23) of this node
  {
    const double* q = qp + (size_t)node * QP_SIZE;
    const int i = threadIdx.x;
    if (i < 128) gred[i] = i < NX ? fabs(q[QP_QV + i]) : (i < NX + NUT ? fabs(q[QP_RV + i - NX]) : 0.0);
    __syncthreads();
    if (i == 0) {
      double m = 0.0;
      for (int l = 0; l < 128; ++l) m = (gred[l] != gred[l]) ? HUGE_VAL : fmax(m, gred[l]);
      atomicMax(reinterpret_cast<unsigned long long*>(ginf + b), (unsigned long long)__double_as_longlong(m));
    }
  }
}

// ---- gate of the two-level sweep: the KKT residual of the LAST stage of every segment but the last one.  Inside a segment the gains come
//      from an exact recursion started at the segment's end, so its stages are stationary up to rounding; what the scanned boundary value
//      functions got wrong shows at stage k_p - 1, whose gains were derived from the scanned suffix p while its successor costate is the
//      value function segment p's own recursion arrives at (vf: written by k_seg_riccati for exactly these nodes).  B (P - 1) workgroups
//      instead of the B N of k_kkt (135 us at 32 x 100 nodes).
__global__ __launch_bounds__(256, 4) void k_kkt_boundaries(const double* __restrict__ x_init, const double* __restrict__ x, const double* __restrict__ qp,
                                                        const double* __restrict__ vf, const double* __restrict__ dx, const double* __restrict__ ut, int N, int P,
                                                        double* __restrict__ kkt, double* __restrict__ ginf) {
  __shared__ KktWS w;
  __shared__ double r2[2], gred[128];
  const int b = blockIdx.x / (P - 1), p = 1 + blockIdx.x % (P - 1), k = seg_bound(p, N, P) - 1;
  const Ctx ctx{(int)threadIdx.x, 256, nullptr};
  const double* vfb = vf + (size_t)b * (N + 1) * VF_SIZE;
  const double* dxb = dx + (size_t)b * (N + 1) * NX;
  const size_t node = (size_t)b * N + k;
  (void)x_init; (void)x;   // (k >= 1 here: the initial-condition residual belongs to stage 0, which is no boundary stage)
  kkt_node(ctx, w, qp + node * QP_SIZE, vfb + (size_t)k * VF_SIZE, vfb + (size_t)(k + 1) * VF_SIZE, dxb + (size_t)k * NX, dxb + (size_t)(k + 1) * NX, ut + node * NUT, nullptr, r2);
  if (threadIdx.x < 2) atomicMax(reinterpret_cast<unsigned long long*>(kkt + 2 * b + threadIdx.x), (unsigned long long)__double_as_longlong(r2[threadIdx.x]));
  {
    const double* q = qp + node * QP_SIZE;
    const int i = threadIdx.x;
    if (i < 128) gred[i] = i < NX ? fabs(q[QP_QV + i]) : (i < NX + NUT ? fabs(q[QP_RV + i - NX]) : 0.0);
    __syncthreads();
    if (i == 0) {
      double m = 0.0;
      for (int l = 0; l < 128; ++l) m = (gred[l] != gred[l]) ? HUGE_VAL : fmax(m, gred[l]);
      atomicMax(reinterpret_cast<unsigned long long*>(ginf + b), (unsigned long long)__double_as_longlong(m));
    }
  }
}

// ---- per-node parameter table from the compact per-instance reference: one thread per (instance, node)
__global__ __launch_bounds__(64) void k_params(const DevModel* __restrict__ dm, hsqp_swing_config cfg, double terrain, int arm_swing, int max_events,
                                               const int* __restrict__ n_events, const double* __restrict__ ev, const int* __restrict__ seq, int n_knots,
                                               const double* __restrict__ tt, const double* __restrict__ ts, double t0, double dt, const double* __restrict__ times,
                                               int N, int B, double* __restrict__ par, int* __restrict__ bad) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= B * (N + 1)) return;
  const int b = id / (N + 1), k = id % (N + 1);
  const bool ok = node_params_eval(cfg, terrain, arm_swing, n_events[b], ev + (size_t)b * max_events, seq + (size_t)b * (max_events + 1), n_knots,
                                   tt + (size_t)b * n_knots, ts + (size_t)b * n_knots * NX, times ? times[id] : t0 + k * dt, par + (size_t)id * NP);
  if (!ok) atomicExch(bad, 1);
}

// ---- policy evaluation / joint torques in three small kernels.
//  k_policy_inputs: one workgroup per pair — xt != null: interpolate the trajectories of instance blockIdx.x at s[blockIdx.x]
//                   (uniform grid, or dts != null: the instance's interval lengths); otherwise take the pair from xin / uin.
//  k_cent_policy_map (centroidal handles): one wave per pair — the whole-body (x, u) computeJointTorques needs
//                   (CentroidalMpcMrtJointController::computeJointControlAction, humanoid_centroidal_mpc/src/mrt/
//                   CentroidalMpcMrtJointController.cpp:155-175): q = getGeneralizedCoordinates(x), qd = getGeneralizedVelocities(x, u)
//                   with the base velocity from the centroidal momentum, v_b = A_b^-1 (m h - A_j qd_j) = rows 6..11 of the flow map,
//                   the contact wrenches, and the desired joint accelerations carried in entries 35..57 of the INPUT state row.
//  k_policy_torques: one workgroup per pair — model evaluation + torques (hsqp_policy.h).
__global__ __launch_bounds__(64) void k_policy_inputs(const double* __restrict__ xt, const double* __restrict__ ut, int N, double dt, const double* __restrict__ dts,
                                                      const double* __restrict__ s, const double* __restrict__ xin, const double* __restrict__ uin,
                                                      double* __restrict__ xout, double* __restrict__ uout) {
  const int b = blockIdx.x;
  const Ctx ctx{(int)threadIdx.x, 64, nullptr};
  if (xt) {
    if (dts) policy_interpolate_grid(ctx, xt + (size_t)b * (N + 1) * NX, ut + (size_t)b * N * NU, N, dts + (size_t)b * N, s[b], xout + (size_t)b * NX, uout + (size_t)b * NU);
    else policy_interpolate(ctx, xt + (size_t)b * (N + 1) * NX, ut + (size_t)b * N * NU, N, dt, s[b], xout + (size_t)b * NX, uout + (size_t)b * NU);
  } else {
    for (int i = threadIdx.x; i < NX + NU; i += blockDim.x) { if (i < NX) xout[(size_t)b * NX + i] = xin[(size_t)b * NX + i]; else uout[(size_t)b * NU + i - NX] = uin[(size_t)b * NU + i - NX]; }
  }
}
__global__ __launch_bounds__(64) void k_cent_policy_map(const DevModel* __restrict__ dm, int n, const double* __restrict__ xc, const double* __restrict__ uc,
                                                        double* __restrict__ xwb, double* __restrict__ uwb) {
  const int b = blockIdx.x;
  CentWST<false>& w = *reinterpret_cast<CentWST<false>*>(hsqp_smem);
  const Ctx ctx{(int)threadIdx.x, 64, nullptr};
  const double* x = xc + (size_t)b * NX;
  const double* u = uc + (size_t)b * NU;
  double* xo = xwb + (size_t)b * NX;
  double* uo = uwb + (size_t)b * NU;
  cent_base_velocity(ctx, *dm, w, x, u, xo + NV);     // [pdot; euler rates] -> the base velocities of the whole-body state
  for (int i = threadIdx.x; i < NX + NU; i += blockDim.x) {
    if (i < NV) xo[i] = x[6 + i];
    else if (i >= NV + 6 && i < NX) { const int j = i - NV - 6; xo[i] = u[12 + j]; }
    else if (i >= NX && i < NX + 12) uo[i - NX] = u[i - NX];
    else if (i >= NX + 12) { const int j = i - NX - 12; uo[12 + j] = x[HSQP_CNX + j]; }
  }
}
struct PolicyWS { StageWST<false> st; double x[NX], u[NU]; };
__global__ __launch_bounds__(128) void k_policy_torques(const DevModel* __restrict__ dm, const double* __restrict__ xwb, const double* __restrict__ uwb,
                                                        double* __restrict__ tau) {
  PolicyWS& w = *reinterpret_cast<PolicyWS*>(hsqp_smem);
  const int b = blockIdx.x;
  const Ctx ctx{(int)threadIdx.x, 128, nullptr};
  for (int i = threadIdx.x; i < NX + NU; i += blockDim.x) { if (i < NX) w.x[i] = xwb[(size_t)b * NX + i]; else w.u[i - NX] = uwb[(size_t)b * NU + i - NX]; }
  __syncthreads();
  policy_node(ctx, *dm, w.st, w.x, w.u, tau + (size_t)b * NJ);
}

// ---- per-instance performance index from per-node {ne, dt*cost, dt*eq^2, dt*dyn^2} + terminal cost
__device__ inline void perf_reduce_instance(int b, const DevModel* __restrict__ dm, const double* __restrict__ misc, int misc_stride, const double* __restrict__ x,
                                            const double* __restrict__ par, int N, hsqp_perf* __restrict__ out, const LsState* __restrict__ ls) {
  if (ls && !ls[b].active) return;
  __shared__ double red[3][64];
  double c = 0.0, e = 0.0, d = 0.0;
  for (int k = threadIdx.x; k < N; k += blockDim.x) {
    const double* m = misc + ((size_t)b * N + k) * misc_stride;
    c += m[1]; e += m[2]; d += m[3];
  }
  if (threadIdx.x < NX) {  // terminal QuadraticStateCost(Q_final * scaling): HumanoidCostConstraintFactory.cpp:218-228
    const double dd = x[((size_t)b * (N + 1) + N) * NX + threadIdx.x] - par[((size_t)b * (N + 1) + N) * NP + HSQP_P_XDES + threadIdx.x];
    c += 0.5 * dm->Qf[threadIdx.x] * dd * dd;
  }
  red[0][threadIdx.x] = c; red[1][threadIdx.x] = e; red[2][threadIdx.x] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    double cs = 0.0, es = 0.0, ds = 0.0;
    for (int i = 0; i < 64; ++i) { cs += red[0][i]; es += red[1][i]; ds += red[2][i]; }
    out[b].cost = cs; out[b].merit = cs; out[b].equality_sse = es; out[b].dynamics_sse = ds;
  }
}
__global__ void k_perf_reduce(const DevModel* __restrict__ dm, const double* __restrict__ misc, int misc_stride, const double* __restrict__ x,
                              const double* __restrict__ par, int N, hsqp_perf* __restrict__ out, const LsState* __restrict__ ls) {
  perf_reduce_instance(blockIdx.x, dm, misc, misc_stride, x, par, N, out, ls);
}
// the three per-instance reductions that follow the full-step trial in ONE launch (blockIdx.y: performance index of the iterate, of the trial, the
// line-search state): three back-to-back launches of 5 us kernels cost their launch gaps
__global__ __launch_bounds__(64) void k_perf_trio(const DevModel* __restrict__ dm, const double* __restrict__ misc0, int stride0, const double* __restrict__ x,
                                                  const double* __restrict__ misc1, int stride1, const double* __restrict__ x_new, const double* __restrict__ par, int N,
                                                  hsqp_perf* __restrict__ before, hsqp_perf* __restrict__ after, const double* __restrict__ info,
                                                  const double* __restrict__ dx, LsState* __restrict__ ls) {
  if (blockIdx.y == 0) perf_reduce_instance(blockIdx.x, dm, misc0, stride0, x, par, N, before, nullptr);
  else if (blockIdx.y == 1) perf_reduce_instance(blockIdx.x, dm, misc1, stride1, x_new, par, N, after, nullptr);
  else ls_init_instance(blockIdx.x, dm, info, x, dx, par, N, ls);
}

}  // namespace

// =================================================================================================
struct hsqp_handle {
  hsqp_model_desc md;
  hsqp_settings st;
  DevModel hdm;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[6] = {};
  static constexpr int LQ_SPLIT_MAX = 8;
  int lq_round_blocks = 512;                   // workgroups of k_lq_limb / k_lq_rows the chip holds at once (4 waves per CU, QL_WAVES per workgroup)
  int lq_split = 1;                            // node ranges of the limb-lane LQ kernels, each on its own stream (HSQP_LQ_SPLIT in the environment at hsqp_create)
  hipStream_t aux[LQ_SPLIT_MAX - 1] = {};
  hipEvent_t ev_fork = nullptr, ev_join[LQ_SPLIT_MAX - 1] = {};
  DevModel* d_dm = nullptr;
  double *d_xinit = nullptr, *d_x = nullptr, *d_u = nullptr, *d_par = nullptr;
  double *d_rec = nullptr, *d_qp = nullptr, *d_ric = nullptr;
  double *d_dx = nullptr, *d_du = nullptr, *d_ut = nullptr, *d_xnew = nullptr, *d_unew = nullptr;
  double* d_fj = nullptr;         // [B][N][NJ] rows 12 .. 34 of Px dx + Pu ut as the factored roll-out forms them (k_step then reads the wrench rows of Px / Pu only)
  double *d_misc = nullptr, *d_kkt = nullptr, *d_ginf = nullptr;
  double* d_dt = nullptr;         // [B][N] length of every interval (uniform grids: filled with dt)
  std::vector<double> h_dt;       // host copy (debug reads), empty for device-resident uploads
  bool uniform_grid = true, has_events = false;
  double* d_vf = nullptr;         // [B][N+1][VF_SIZE] value function of the last Riccati sweep (allocated when a KKT check is first asked for)
  double* d_vf2 = nullptr;        // scan path: value functions of the refinement pass (the KKT check then reads these)
  double* h_gate = nullptr;       // pinned host copy of the gate block [kkt | |g|_inf | scan flags]
  long long scan_fallbacks = 0;   // iterations whose scan result failed the KKT gate and were redone with the serial recursion
  double* d_acl = nullptr;        // scan path: closed loop [B][N][ACL_SIZE] of every stage for the roll-out (allocated when the scan is first used)
  // segmented sweep (allocated when first used): gains of the J = 0 recursions, (L^-1)^T of every stage, (J, s) at the segment starts, zeros
  double *d_ric2 = nullptr, *d_linv = nullptr, *d_vf0 = nullptr, *d_zero = nullptr;
  size_t vf0_capacity = 0;
  int seg_backoff = 0, seg_backoff_len = 0;   // after a rejected gated sweep (scan or two-level) the next seg_backoff iterations go straight to the serial recursion (doubling, <= 64);
                                              // state of the AUTOMATIC sweep choice only (a sweep forced by a flag is always attempted), reset by every upload
  bool backoff_persistent = false;            // hsqp_set_scan_backoff_persistent: uploads of the same (B, N) keep the back-off
  long long backoff_iterations = 0;           // iterations that ran the serial recursion because of the back-off (hsqp_scan_backoffs)
  bool seg_debug = false;                     // HSQP_SEG_DEBUG in the environment at hsqp_create
  bool chain_fused = false;                   // limb-lane form: the RK4 chain of the columns of [A|B] runs inside k_project (project_node, chain), the defect on the lanes of k_lq_rows (ql_defect_lane), k_lq_chain is not launched; HSQP_LQ_CHAIN_SEPARATE in the environment at hsqp_create keeps the chain in k_lq_chain (A/B runs)
  bool lq_limb = false;                       // whole-body LQ approximation on limb lanes (hsqp_lql.h: k_lq_limb + k_lq_rows + k_lq_chain) instead of the phase form k_lq<true> (HSQP_LQ_PHASE_FORM / HSQP_LQ_LIMB_FORM in the environment at hsqp_create force either)
  bool ric_fact = false;                      // whole-body serial sweep on the factors of [A~ | B~] (hsqp_riccati_fact.h: k_riccati_fact; HSQP_RICCATI_DENSE in the environment at hsqp_create: the dense stage k_riccati<58>, for A/B runs)
  bool value_quad = false;                    // whole-body value pass on quads of lanes (hsqp_lqv.h): the tree has at most four limbs (HSQP_VALUE_PHASE_FORM in the environment at hsqp_create: the phase form, for A/B runs)
  hsqp_perf *d_perf_before = nullptr, *d_perf_after = nullptr;
  int* d_status = nullptr;
  int* d_scanst = nullptr;   // flags of the scan kernels (bad pivot, rank-deficient D, failed Lam) of the current attempt: part of the gate, behind d_ginf
  double* d_stepinfo = nullptr;   // [B][N][4] per-node {armijo, |dx|^2, |du|^2}
  LsState* d_ls = nullptr;
  int* d_counts = nullptr;
  hsqp_linesearch_settings ls_settings;
  std::vector<LsState> h_ls;                  // host copies for HSQP_ITER_UNTIL_CONVERGED (reused across iterations and calls)
  std::vector<hsqp_perf> h_perf_before;
  double* d_el[2] = {nullptr, nullptr};   // scan elements (allocated when the parallel-in-time sweep is first used)
  size_t el_capacity = 0;                 // in doubles per buffer
  void* d_stage = nullptr;        // grow-only staging area for the small per-call inputs (reference, policy queries)
  size_t stage_bytes = 0;
  bool ls_ran = false;
  long long* d_prof = nullptr;   // [4][128] phase-profile ticks (k_lq<true>, k_project, k_riccati, k_lq<false>)
  int B = 0, N = 0;
  double dt = 0.0;
  bool have_problem = false, have_solution = false;
  double kernel_ms[5] = {0, 0, 0, 0, 0};
  int last_iterations = 0;
  struct IterLog { std::vector<hsqp_perf> perf; std::vector<double> alpha; std::vector<int> type; };
  std::vector<IterLog> iter_log;   // HSQP_ITER_UNTIL_CONVERGED: what every iteration of the last call ended with
  std::string err;
};

static std::string g_create_error;

#define HCHECK(call)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      h->err = std::string(#call) + ": " + hipGetErrorString(e_);                                      \
      return HSQP_ERR_HIP;                                                                             \
    }                                                                                                  \
  } while (0)

// grow-only device staging area of the handle (256-byte aligned carving by the callers)
static void* stage_area(hsqp_handle* h, size_t bytes) {
  if (bytes > h->stage_bytes) {
    if (h->d_stage) (void)hipFree(h->d_stage);
    h->d_stage = nullptr; h->stage_bytes = 0;
    if (hipMalloc(&h->d_stage, bytes) != hipSuccess) return nullptr;
    poison_hbm(h->d_stage, bytes);
    h->stage_bytes = bytes;
  }
  return h->d_stage;
}
static size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

// (the KKT gate of the parallel-in-time sweep: scan_gate_accepts, hsqp_scan.h)
// Refinement passes of the gains (one more stage of the exact Riccati map from the value functions of the previous pass).  Centroidal: one
// pass gains a digit (1e-11 -> 1e-12 of the step's scale).  Whole-body: none — the closed loop of that problem has slow modes (positions
// integrate velocities over dt = 0.02), so the map barely contracts an error in S: 0, 1, 2 or 3 passes all leave 1.4e-11 .. 8e-11
// (tests/hostemu probe, N = 16 .. 100), and each costs 24 us.
constexpr int HSQP_SCAN_WB_REFINEMENTS = 0;
// parallel-in-time backward sweep (hsqp_scan.h): elements of all stages, ceil(log2(N+1)) scan levels, single-stage gains (from the
// scanned value functions, then `refinements` times from the value functions of the previous gains pass), closed-loop roll-out
template <int n>
static int launch_scan(hsqp_handle* h, int B, int N, bool want_kkt, int refinements) {
  constexpr int SZ = ScanEl<n>::SIZE;
  const int nodes = B * N;
  const size_t need = (size_t)B * (N + 1) * SZ;
  if (need > h->el_capacity) {
    for (auto& p : h->d_el) { if (p) (void)hipFree(p); p = nullptr; }
    h->el_capacity = 0;
    for (auto& p : h->d_el) {
      if (hipMalloc(&p, need * 8) != hipSuccess) { p = nullptr; h->err = "hipMalloc failed (scan elements)"; return HSQP_ERR_OOM; }
      poison_hbm(p, need * 8);
    }
    h->el_capacity = need;
  }
  HSQP_LAUNCH(k_scan_init<n>, dim3(B * (N + 1)), dim3(SCAN_INIT_THREADS), sizeof(ScanInitWS<n>), h->stream, h->d_dm, h->d_x, h->d_par, h->d_qp, N, h->d_el[0], h->d_scanst);
  int cur = 0;
  for (int d = 1; d < N + 1; d *= 2) {
    HSQP_LAUNCH(k_scan_combine<n>, dim3(B * (N + 1)), dim3(SCAN_COMB_THREADS), sizeof(ScanCombWS<n>), h->stream, h->d_el[cur], h->d_el[1 - cur], N, d, h->d_scanst, h->d_prof + 256);
    cur = 1 - cur;
  }
  for (double** pv : {&h->d_vf, &h->d_vf2})
    if (!*pv) {
      const size_t bytes = (size_t)h->st.max_batch * (h->st.max_nodes + 1) * VF_SIZE * 8;
      if (hipMalloc(pv, bytes) != hipSuccess) { *pv = nullptr; h->err = "hipMalloc failed (value functions of the scan, " + std::to_string(bytes) + " bytes)"; return HSQP_ERR_OOM; }
      poison_hbm(*pv, bytes);
    }
  if (!h->d_acl) {
    const size_t bytes = (size_t)h->st.max_batch * h->st.max_nodes * ACL_SIZE<n> * 8;
    if (hipMalloc(&h->d_acl, bytes) != hipSuccess) { h->d_acl = nullptr; h->err = "hipMalloc failed (closed loop of the scan path, " + std::to_string(bytes) + " bytes)"; return HSQP_ERR_OOM; }
    poison_hbm(h->d_acl, bytes);
  }
  // the gains passes ping-pong between the two value-function buffers; the LAST pass writes d_vf2 (what the KKT check reads) and the closed loop
  double* vbuf[2] = {(refinements & 1) ? h->d_vf : h->d_vf2, (refinements & 1) ? h->d_vf2 : h->d_vf};
  for (int pass = 0; pass <= refinements; ++pass) {
    const bool lastp = pass == refinements;
    HSQP_LAUNCH(k_scan_gains<n>, dim3(nodes), dim3(RIC_THREADS), sizeof(RicWS), h->stream, h->d_dm, h->d_x, h->d_par, h->d_qp, h->d_el[cur],
                       pass == 0 ? (const double*)nullptr : (const double*)vbuf[(pass - 1) & 1], h->d_ric, N, h->d_scanst,
                       (lastp && !want_kkt) ? (double*)nullptr : vbuf[pass & 1], lastp ? h->d_acl : (double*)nullptr);
  }
  HSQP_LAUNCH(k_scan_forward<n>, dim3(B), dim3(SCAN_FWD_THREADS), sizeof(RicWS), h->stream, h->d_xinit, h->d_x, h->d_acl, N, h->d_dx);   // 4 n <= 256 items per stage: four waves
  return HSQP_OK;
}

// segment count of the two-level sweep for B instances on N intervals (0: not applicable).  OPT-IN (HSQP_FLAG_SEGMENTED_RICCATI): measured on
// perturbed config-4 batches of 32 (tools/gpu_seg_fuzz.py) its step is 4e-11 .. 4e-10 of the step's scale (up to 7e-8 absolute) from the serial
// recursion's — the combination of two long-segment elements solves with cond(I + C1 J2) ~ 1e9 — which is outside BASELINE.md §6's 1e-8 on
// trajectories, so the default path never takes it; a caller that accepts the declared error asks for it.  P + 1 elements per instance go
// through the suffix scan: P + 1 a power of two wastes no level, B (P + 1) <= 256 workgroups run a level in one round, and every level costs
// accuracy, hence P <= 7 (B = 32: P = 7, three levels of 256 workgroups).
static int segment_count(const hsqp_handle* h, int B, int N) {
  if (!(h->st.flags & HSQP_FLAG_SEGMENTED_RICCATI)) return 0;
  const int cap = std::min(std::min(256 / std::max(B, 1) - 1, N / 4), 7);
  if (cap < 1) return 0;
  int e = 2;
  while (2 * e <= cap + 1) e *= 2;
  return e - 1 >= 3 ? e - 1 : 0;   // 3 or 7 segments; fewer do not pay
}

// two-level sweep (hsqp_segment.h): segment elements, suffix scan over them, gains per segment, roll-out
template <int n>
static int launch_segmented(hsqp_handle* h, int B, int N, int P, bool want_vf) {
  constexpr int SZ = ScanEl<n>::SIZE;
  const size_t need = (size_t)B * (P + 1) * SZ;
  if (need > h->el_capacity) {
    for (auto& p : h->d_el) { if (p) (void)hipFree(p); p = nullptr; }
    h->el_capacity = 0;
    for (auto& p : h->d_el) {
      if (hipMalloc(&p, need * 8) != hipSuccess) { p = nullptr; h->err = "hipMalloc failed (segment elements)"; return HSQP_ERR_OOM; }
      poison_hbm(p, need * 8);
    }
    h->el_capacity = need;
  }
  const size_t BN = (size_t)h->st.max_batch * h->st.max_nodes;
  if (!h->d_ric2 && hipMalloc(&h->d_ric2, BN * RIC_SIZE * 8) != hipSuccess) { h->d_ric2 = nullptr; h->err = "hipMalloc failed (segmented sweep: gains)"; return HSQP_ERR_OOM; }
  if (!h->d_linv && hipMalloc(&h->d_linv, BN * LDB * LDB * 8) != hipSuccess) { h->d_linv = nullptr; h->err = "hipMalloc failed (segmented sweep: factors)"; return HSQP_ERR_OOM; }
  if (!h->d_zero) {
    if (hipMalloc(&h->d_zero, (NX * NX + NX) * 8) != hipSuccess) { h->d_zero = nullptr; h->err = "hipMalloc failed"; return HSQP_ERR_OOM; }
    if (hipMemsetAsync(h->d_zero, 0, (NX * NX + NX) * 8, h->stream) != hipSuccess) { h->err = "memset failed"; return HSQP_ERR_HIP; }
  }
  if ((size_t)B * P > h->vf0_capacity) {
    if (h->d_vf0) (void)hipFree(h->d_vf0);
    h->d_vf0 = nullptr; h->vf0_capacity = 0;
    if (hipMalloc(&h->d_vf0, (size_t)B * P * VF_SIZE * 8) != hipSuccess) { h->d_vf0 = nullptr; h->err = "hipMalloc failed (segmented sweep: boundary value functions)"; return HSQP_ERR_OOM; }
    poison_hbm(h->d_vf0, (size_t)B * P * VF_SIZE * 8);
    h->vf0_capacity = (size_t)B * P;
  }
  if (!h->d_vf2) {   // (the gate needs the value functions of the boundary stages' nodes; want_vf: of every node, for the KKT report)
    const size_t bytes = (size_t)h->st.max_batch * (h->st.max_nodes + 1) * VF_SIZE * 8;
    if (hipMalloc(&h->d_vf2, bytes) != hipSuccess) { h->d_vf2 = nullptr; h->err = "hipMalloc failed (value functions of the segmented sweep, " + std::to_string(bytes) + " bytes)"; return HSQP_ERR_OOM; }
    poison_hbm(h->d_vf2, bytes);
  }
  const int segs = B * P;
  HSQP_LAUNCH(k_seg_elem_ric<n>, dim3(segs), dim3(RIC_THREADS), sizeof(RicWS), h->stream, h->d_dm, h->d_x, h->d_par, h->d_qp, (const double*)h->d_zero, h->d_ric2,
                     h->d_linv, h->d_vf0, N, P, h->d_scanst);
  HSQP_LAUNCH(k_seg_accumulate<n>, dim3(segs), dim3(SEG_ACC_THREADS), sizeof(SegAccWS), h->stream, (const double*)h->d_qp, (const double*)h->d_ric2,
                     (const double*)h->d_linv, (const double*)h->d_vf0, N, P, h->d_el[0]);
  HSQP_LAUNCH(k_seg_terminal<n>, dim3(B), dim3(256), 0, h->stream, h->d_dm, h->d_x, h->d_par, N, P, h->d_el[0]);
  int cur = 0;
  for (int d = 1; d < P + 1; d *= 2) {
    HSQP_LAUNCH(k_scan_combine<n>, dim3(B * (P + 1)), dim3(SCAN_COMB_THREADS), sizeof(ScanCombWS<n>), h->stream, h->d_el[cur], h->d_el[1 - cur], P, d, h->d_scanst,
                       (long long*)nullptr);
    cur = 1 - cur;
  }
  HSQP_LAUNCH(k_seg_riccati<n>, dim3(segs), dim3(RIC_THREADS), sizeof(RicWS), h->stream, h->d_dm, h->d_x, h->d_par, h->d_qp, (const double*)h->d_el[cur], h->d_ric, N, P,
                     h->d_scanst, h->d_vf2, want_vf ? 0 : 2);
  HSQP_LAUNCH(k_ric_forward<n>, dim3(B), dim3(RIC_THREADS), sizeof(RicWS), h->stream, h->d_xinit, h->d_x, h->d_qp, (const double*)h->d_ric, N, h->d_dx, h->d_ut);
  return HSQP_OK;
}

extern "C" {

const char* hsqp_version(void) { return "hsqp-hip 0.3 (gfx950, f64, abi 6)"; }
int hsqp_abi_version(void) { return HSQP_ABI_VERSION; }
int hsqp_set_scan_backoff_persistent(hsqp_handle* h, int on) {
  if (!h) return HSQP_ERR_BAD_ARG;
  h->backoff_persistent = on != 0;
  return HSQP_OK;
}

int hsqp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

long long hsqp_scan_fallbacks(const hsqp_handle* h) { return h ? h->scan_fallbacks : -1; }
long long hsqp_scan_backoffs(const hsqp_handle* h) { return h ? h->backoff_iterations : -1; }

const char* hsqp_last_error(const hsqp_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

void hsqp_destroy(hsqp_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  void* bufs[] = {h->d_dm, h->d_xinit, h->d_x, h->d_u, h->d_par, h->d_rec, h->d_qp, h->d_ric, h->d_dx, h->d_du, h->d_ut, h->d_fj, h->d_xnew,
                  h->d_unew, h->d_misc, h->d_kkt, h->d_dt, h->d_perf_before, h->d_perf_after, h->d_status, h->d_prof, h->d_stepinfo, h->d_ls, h->d_counts, h->d_vf, h->d_stage,
                  h->d_el[0], h->d_el[1], h->d_vf2, h->d_acl, h->d_ric2, h->d_linv, h->d_vf0, h->d_zero};
  for (void* p : bufs)
    if (p) (void)hipFree(p);
  if (h->h_gate) (void)hipHostFree(h->h_gate);
  for (auto& e : h->ev)
    if (e) (void)hipEventDestroy(e);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  for (auto& e : h->ev_join)
    if (e) (void)hipEventDestroy(e);
  for (auto& st : h->aux)
    if (st) (void)hipStreamDestroy(st);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

void hsqp_linesearch_defaults(hsqp_linesearch_settings* s) {
  if (!s) return;
  s->g_max = 1e-2; s->g_min = 1e-6;      // g1_wb_mpc/config/mpc/task.info:83-84
  s->gamma_c = 1e-6; s->armijo_factor = 1e-4; s->alpha_decay = 0.5; s->alpha_min = 1e-4;   // upstream ocs2 sqp::Settings defaults
  s->delta_tol = 1e-4;                   // task.info deltaTol
  s->cost_tol = 1e-4;                    // upstream ocs2 sqp::Settings::costTol default (the task file does not set it)
}

// trials after which every instance has either accepted a step or fallen below alpha_min (zero step)
static int ls_max_trials(const hsqp_linesearch_settings& s) { return (int)ceil(log(s.alpha_min) / log(s.alpha_decay)) + 2; }
#define HSQP_LS_MAX_TRIALS 4096

int hsqp_set_linesearch(hsqp_handle* h, const hsqp_linesearch_settings* s) {
  if (!h) return HSQP_ERR_BAD_ARG;
  if (!s || !(s->alpha_decay > 0.0 && s->alpha_decay < 1.0) || !(s->alpha_min > 0.0) || !(s->g_max >= s->g_min) || !(s->cost_tol >= 0.0)) {
    h->err = "line-search settings: need 0 < alpha_decay < 1, alpha_min > 0, g_max >= g_min, cost_tol >= 0";
    return HSQP_ERR_BAD_ARG;
  }
  if (s->alpha_min < 1.0 && ls_max_trials(*s) > HSQP_LS_MAX_TRIALS) {
    h->err = "line-search settings: more than 4096 back-tracking trials (alpha_decay too close to 1 for this alpha_min)";
    return HSQP_ERR_BAD_ARG;
  }
  h->ls_settings = *s;
  return HSQP_OK;
}

int hsqp_create(const hsqp_model_desc* model, const hsqp_settings* settings, hsqp_handle** out) {
  if (!model || !settings || !out) { g_create_error = "null argument"; return HSQP_ERR_BAD_ARG; }
  *out = nullptr;
  if (settings->max_nodes < 1 || settings->max_batch < 1) { g_create_error = "max_nodes and max_batch must be >= 1"; return HSQP_ERR_BAD_ARG; }
  if ((settings->flags & HSQP_FLAG_PARALLEL_RICCATI) && (settings->flags & HSQP_FLAG_SERIAL_RICCATI)) {
    g_create_error = "HSQP_FLAG_PARALLEL_RICCATI excludes HSQP_FLAG_SERIAL_RICCATI";
    return HSQP_ERR_BAD_ARG;
  }
  if ((settings->flags & HSQP_FLAG_SEGMENTED_RICCATI) && (settings->flags & (HSQP_FLAG_PARALLEL_RICCATI | HSQP_FLAG_SERIAL_RICCATI))) {
    g_create_error = "HSQP_FLAG_SEGMENTED_RICCATI excludes HSQP_FLAG_PARALLEL_RICCATI and HSQP_FLAG_SERIAL_RICCATI";
    return HSQP_ERR_BAD_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_create_error = "no HIP device visible (this library has no CPU path)"; return HSQP_ERR_NO_DEVICE; }
  if (settings->device < 0 || settings->device >= ndev) { g_create_error = "device ordinal out of range"; return HSQP_ERR_BAD_ARG; }
  hsqp_handle* h = new hsqp_handle;
  h->seg_debug = getenv("HSQP_SEG_DEBUG") != nullptr;
  h->ric_fact = model->formulation == HSQP_FORM_WB && getenv("HSQP_RICCATI_DENSE") == nullptr;
  h->md = *model;
  h->st = *settings;
  h->device = settings->device;
  hsqp_linesearch_defaults(&h->ls_settings);
  const std::string e = build_dev_model(*model, h->hdm);
  if (!e.empty()) { g_create_error = e; delete h; return HSQP_ERR_BAD_ARG; }
  // the value pass on quads of lanes (hsqp_lqv.h) is a throughput form: a wave evaluates 16 nodes in ~100 us whatever their number, the phase form one node
  // in ~40 us — so a handle sized for fewer nodes than fill the GPU once (config 3: one instance, 100 nodes) keeps the phase form.  Decided per HANDLE, not
  // per call: every solve of a handle runs the same arithmetic (an instance of a batch equals its solo solve bit for bit)
  h->value_quad = h->hdm.formulation == HSQP_FORM_WB && h->hdm.n_limbs > 0 && getenv("HSQP_VALUE_PHASE_FORM") == nullptr &&
                  (getenv("HSQP_VALUE_QUAD_FORM") != nullptr || (size_t)settings->max_batch * settings->max_nodes >= HSQP_VALUE_QUAD_MIN_NODES);
  // the LQ approximation on limb lanes (hsqp_lql.h) is a throughput form like the quad value pass; same rule, same per-handle decision
  h->lq_limb = h->hdm.formulation == HSQP_FORM_WB && h->hdm.ql_ok && getenv("HSQP_LQ_PHASE_FORM") == nullptr &&
               (getenv("HSQP_LQ_LIMB_FORM") != nullptr || (size_t)settings->max_batch * settings->max_nodes >= HSQP_LQ_LIMB_MIN_NODES);
  h->chain_fused = h->lq_limb && getenv("HSQP_LQ_CHAIN_SEPARATE") == nullptr;
  auto fail = [&](int code, const std::string& msg) { g_create_error = msg; hsqp_destroy(h); return code; };
  if (hipSetDevice(h->device) != hipSuccess) return fail(HSQP_ERR_HIP, "hipSetDevice failed");
  if (hipStreamCreate(&h->stream) != hipSuccess) return fail(HSQP_ERR_HIP, "hipStreamCreate failed");
  for (auto& ev : h->ev)
    if (hipEventCreate(&ev) != hipSuccess) return fail(HSQP_ERR_HIP, "hipEventCreate failed");
  if (h->lq_limb) {
    const char* sp = getenv("HSQP_LQ_SPLIT");
    h->lq_split = HSQP_LQ_SPLIT_DEFAULT;
    if (sp) {
      char* end = nullptr;
      const long v = strtol(sp, &end, 10);
      if (end == sp || *end != '\0' || v < 1 || v > hsqp_handle::LQ_SPLIT_MAX)
        return fail(HSQP_ERR_BAD_ARG, std::string("HSQP_LQ_SPLIT=\"") + sp + "\": expected an integer in [1, " + std::to_string(hsqp_handle::LQ_SPLIT_MAX) + "]");
      h->lq_split = (int)v;
    }
    // one round of the chip = the workgroups of the two one-wave-per-SIMD kernels it holds at once: asked of the runtime for the kernels as built
    // (tuning builds change their waves per SIMD: HSQP_QL_WPE / HSQP_QR_WPE), not assumed from the CU count
    int cus = 0, per_cu_limb = 0, per_cu_rows = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && cus > 0 &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_limb, (const void*)k_lq_limb, QL_THREADS * QL_WAVES, 0) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_rows, (const void*)k_lq_rows, QL_THREADS * QL_WAVES, 0) == hipSuccess &&
        per_cu_limb > 0 && per_cu_rows > 0)
      h->lq_round_blocks = cus * std::min(per_cu_limb, per_cu_rows);
    else if (cus > 0) h->lq_round_blocks = cus * 4 / QL_WAVES;
    if (h->lq_split > hsqp_handle::LQ_SPLIT_MAX) h->lq_split = hsqp_handle::LQ_SPLIT_MAX;
    if (h->lq_split > 1 && hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess) return fail(HSQP_ERR_HIP, "hipEventCreate failed");
    for (int s = 0; s + 1 < h->lq_split; ++s)
      if (hipStreamCreateWithFlags(&h->aux[s], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->ev_join[s], hipEventDisableTiming) != hipSuccess)
        return fail(HSQP_ERR_HIP, "hipStreamCreate failed");
  }
  const size_t B = settings->max_batch, N = settings->max_nodes;
  struct Alloc { void** p; size_t bytes; };
  const Alloc allocs[] = {
      {(void**)&h->d_dm, sizeof(DevModel)},
      {(void**)&h->d_xinit, B * NX * 8}, {(void**)&h->d_x, B * (N + 1) * NX * 8}, {(void**)&h->d_u, B * N * NU * 8},
      {(void**)&h->d_par, B * (N + 1) * NP * 8}, {(void**)&h->d_rec, B * N * (size_t)REC_SIZE * 8},
      {(void**)&h->d_qp, B * N * (size_t)QP_SIZE * 8}, {(void**)&h->d_ric, B * N * (size_t)RIC_SIZE * 8},
      {(void**)&h->d_dx, B * (N + 1) * NX * 8}, {(void**)&h->d_du, B * N * NU * 8}, {(void**)&h->d_ut, B * N * NUT * 8}, {(void**)&h->d_fj, B * N * NJ * 8},
      {(void**)&h->d_xnew, B * (N + 1) * NX * 8}, {(void**)&h->d_unew, B * N * NU * 8}, {(void**)&h->d_misc, B * N * 8 * 8},
      {(void**)&h->d_kkt, B * 3 * 8 + ((B * sizeof(int) + 7) / 8) * 8}, {(void**)&h->d_dt, B * N * 8}, {(void**)&h->d_perf_before, B * sizeof(hsqp_perf)}, {(void**)&h->d_perf_after, B * sizeof(hsqp_perf)},
      {(void**)&h->d_status, B * sizeof(int)}, {(void**)&h->d_prof, 4 * 128 * sizeof(long long)},
      {(void**)&h->d_stepinfo, B * N * 4 * 8}, {(void**)&h->d_ls, B * sizeof(LsState)}, {(void**)&h->d_counts, 2 * sizeof(int)}};
  g_poison_hbm = getenv("HSQP_POISON_HBM") != nullptr;
  for (const Alloc& a : allocs) {
    if (hipMalloc(a.p, a.bytes) != hipSuccess) return fail(HSQP_ERR_OOM, "hipMalloc failed (" + std::to_string(a.bytes) + " bytes)");
    poison_hbm(*a.p, a.bytes);
  }
  if (hipHostMalloc((void**)&h->h_gate, B * 3 * 8 + ((B * sizeof(int) + 7) / 8) * 8) != hipSuccess) { h->h_gate = nullptr; return fail(HSQP_ERR_OOM, "hipHostMalloc failed (gate block)"); }
  h->d_ginf = h->d_kkt + 2 * B;   // one block [kkt (2 per instance of max_batch) | |g|_inf | flags of the scan kernels]: one memset, one read-back for the scan's gate
  h->d_scanst = reinterpret_cast<int*>(h->d_kkt + 3 * B);
  if (hipMemcpy(h->d_dm, &h->hdm, sizeof(DevModel), hipMemcpyHostToDevice) != hipSuccess) return fail(HSQP_ERR_HIP, "model upload failed");
  if (hipMemset(h->d_prof, 0, 4 * 128 * sizeof(long long)) != hipSuccess) return fail(HSQP_ERR_HIP, "memset failed");
  // the limb-lane LQ kernel never writes record entries that are zero for every state (hsqp_lql.h)
  if (hipMemset(h->d_rec, 0, B * N * (size_t)REC_SIZE * 8) != hipSuccess) return fail(HSQP_ERR_HIP, "memset failed");
  // the kernels use up to ~158 KB of dynamic LDS (gfx950: 160 KB per workgroup)
  hipError_t a1 = hipFuncSetAttribute((const void*)k_lq<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LqWS));
  g_poison_lds = getenv("HSQP_POISON_LDS") != nullptr;
  if (g_poison_lds) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) g_poison_blocks = 2 * prop.multiProcessorCount;
    if (a1 == hipSuccess) a1 = hipFuncSetAttribute((const void*)k_poison_lds, hipFuncAttributeMaxDynamicSharedMemorySize, POISON_LDS_BYTES);
  }
  hipError_t a2 = hipFuncSetAttribute((const void*)k_lq<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LqWST<false>));
  if (a2 == hipSuccess) a2 = hipFuncSetAttribute((const void*)k_step_value, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LqWST<false>));
  if (a2 == hipSuccess) a2 = hipFuncSetAttribute((const void*)k_lq_cent2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CentWST<true>));
  hipError_t a3 = hipFuncSetAttribute((const void*)k_project, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ProjWS));
  hipError_t a4 = hipFuncSetAttribute((const void*)k_riccati<NX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RicWS));
  if (a4 == hipSuccess) a4 = hipFuncSetAttribute((const void*)k_riccati_fact, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RicFWS));
  hipError_t a5 = hipFuncSetAttribute((const void*)k_riccati<CNX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RicWS));
  if (a5 == hipSuccess) a5 = hipFuncSetAttribute((const void*)k_scan_init<CNX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ScanInitWS<CNX>));
  if (a5 == hipSuccess) a5 = hipFuncSetAttribute((const void*)k_scan_combine<CNX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ScanCombWS<CNX>));
  if (a5 == hipSuccess) a5 = hipFuncSetAttribute((const void*)k_scan_gains<CNX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RicWS));
  if (a5 == hipSuccess) a5 = hipFuncSetAttribute((const void*)k_scan_forward<CNX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RicWS));
  if (a5 == hipSuccess) a5 = hipFuncSetAttribute((const void*)k_scan_init<NX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ScanInitWS<NX>));
  if (a5 == hipSuccess) a5 = hipFuncSetAttribute((const void*)k_scan_combine<NX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ScanCombWS<NX>));
  if (a5 == hipSuccess) a5 = hipFuncSetAttribute((const void*)k_scan_gains<NX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RicWS));
  if (a5 == hipSuccess) a5 = hipFuncSetAttribute((const void*)k_scan_forward<NX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RicWS));
  for (const void* f : {(const void*)k_seg_elem_ric<NX>, (const void*)k_seg_elem_ric<CNX>, (const void*)k_seg_riccati<NX>, (const void*)k_seg_riccati<CNX>,
                        (const void*)k_ric_forward<NX>, (const void*)k_ric_forward<CNX>})
    if (a5 == hipSuccess) a5 = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RicWS));
  for (const void* f : {(const void*)k_seg_accumulate<NX>, (const void*)k_seg_accumulate<CNX>})
    if (a5 == hipSuccess) a5 = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SegAccWS));
  if (a1 != hipSuccess || a2 != hipSuccess || a3 != hipSuccess || a4 != hipSuccess || a5 != hipSuccess) return fail(HSQP_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
  *out = h;
  g_create_error.clear();
  return HSQP_OK;
}

// centroidal formulation: the padding states of every state row must be zero (include/hsqp.h)
static bool padding_is_zero(hsqp_handle* h, const hsqp_problem* p) {
  if (h->hdm.formulation != HSQP_FORM_CENTROIDAL) return true;
  for (size_t r = 0; r < (size_t)p->batch * (p->n_nodes + 2); ++r) {
    const double* row = r < (size_t)p->batch ? p->x_init + r * NX : p->x_traj + (r - p->batch) * NX;
    for (int i = HSQP_CNX; i < NX; ++i)
      if (row[i] != 0.0) { h->err = "centroidal formulation: entries 35..57 of every state row must be zero"; return false; }
  }
  return true;
}

// interval lengths of the problem: hsqp_problem::dt_nodes, or the uniform dt.  Host arrays are validated (finite, >= 0, no event
// as the last interval); device-resident ones are the caller's responsibility and are scanned for events on the device side later.
static int set_grid(hsqp_handle* h, const hsqp_problem* p, bool device_src) {
  const size_t B = p->batch, N = p->n_nodes;
  h->has_events = false;
  if (!p->dt_nodes) {
    h->h_dt.assign(B * N, p->dt);
    h->uniform_grid = true;
    HCHECK(hipMemcpyAsync(h->d_dt, h->h_dt.data(), B * N * 8, hipMemcpyHostToDevice, h->stream));
    return HSQP_OK;
  }
  h->uniform_grid = false;
  if (device_src) {
    h->h_dt.resize(B * N);
    HCHECK(hipMemcpyAsync(h->d_dt, p->dt_nodes, B * N * 8, hipMemcpyDeviceToDevice, h->stream));
    HCHECK(hipMemcpyAsync(h->h_dt.data(), h->d_dt, B * N * 8, hipMemcpyDeviceToHost, h->stream));
    HCHECK(hipStreamSynchronize(h->stream));
  } else {
    h->h_dt.assign(p->dt_nodes, p->dt_nodes + B * N);
  }
  for (size_t b = 0; b < B; ++b)
    for (size_t k = 0; k < N; ++k) {
      const double d = h->h_dt[b * N + k];
      if (!(d >= 0.0) || d > 1e6) { h->err = "dt_nodes: interval lengths must be finite and >= 0"; return HSQP_ERR_BAD_ARG; }
      if (d == 0.0) {
        // the FIRST interval may be an event: a mode switch within dt_min after the initial time replaces the initial node
        // (timeDiscretizationWithEvents), and the identity-jump stage works at node 0 like anywhere else
        if (k == N - 1) { h->err = "dt_nodes: the last interval cannot be an event (dt = 0)"; return HSQP_ERR_BAD_ARG; }
        h->has_events = true;
      }
    }
  if (!device_src) HCHECK(hipMemcpyAsync(h->d_dt, h->h_dt.data(), B * N * 8, hipMemcpyHostToDevice, h->stream));
  return HSQP_OK;
}

static int upload_impl(hsqp_handle* h, const hsqp_problem* p, bool device_src) {
  if (!h) return HSQP_ERR_BAD_ARG;
  if (!p || !p->x_init || !p->x_traj || !p->u_traj || !p->node_params) { h->err = "null problem pointer"; return HSQP_ERR_BAD_ARG; }
  if (p->batch < 1 || p->batch > h->st.max_batch || p->n_nodes < 1 || p->n_nodes > h->st.max_nodes || (!p->dt_nodes && !(p->dt > 0.0))) {
    h->err = "batch / n_nodes outside the handle's capacity, or dt <= 0";
    return HSQP_ERR_BAD_ARG;
  }
  if (!device_src && !padding_is_zero(h, p)) return HSQP_ERR_BAD_ARG;   // device-resident inputs: the caller guarantees the zero padding
  HCHECK(hipSetDevice(h->device));
  const size_t B = p->batch, N = p->n_nodes;
  const hipMemcpyKind kind = device_src ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  // the grid is validated BEFORE any trajectory copy is queued; a rejected problem leaves no half-uploaded state behind
  h->have_problem = false; h->have_solution = false;
  { const int rc = set_grid(h, p, device_src); if (rc != HSQP_OK) return rc; }
  HCHECK(hipMemcpyAsync(h->d_xinit, p->x_init, B * NX * 8, kind, h->stream));
  HCHECK(hipMemcpyAsync(h->d_x, p->x_traj, B * (N + 1) * NX * 8, kind, h->stream));
  HCHECK(hipMemcpyAsync(h->d_u, p->u_traj, B * N * NU * 8, kind, h->stream));
  HCHECK(hipMemcpyAsync(h->d_par, p->node_params, B * (N + 1) * NP * 8, kind, h->stream));
  HCHECK(hipStreamSynchronize(h->stream));
  const bool same_shape = h->B == p->batch && h->N == p->n_nodes;
  h->B = p->batch; h->N = p->n_nodes; h->dt = p->dt;
  h->have_problem = true; h->have_solution = false;
  if (!(h->backoff_persistent && same_shape)) { h->seg_backoff = 0; h->seg_backoff_len = 0; }   // the gate's history belongs to the problem that produced it (receding-horizon callers opt out)
  return HSQP_OK;
}

int hsqp_upload(hsqp_handle* h, const hsqp_problem* p) { return upload_impl(h, p, false); }
int hsqp_upload_device(hsqp_handle* h, const hsqp_problem* p) { return upload_impl(h, p, true); }

int hsqp_upload_reference(hsqp_handle* h, const hsqp_problem* p, const hsqp_reference* r) {
  if (!h) return HSQP_ERR_BAD_ARG;
  if (!p || !r || !p->x_init || !p->x_traj || !p->u_traj || !r->n_events || !r->event_times || !r->mode_sequence || !r->target_times || !r->target_states) {
    h->err = "null problem / reference pointer";
    return HSQP_ERR_BAD_ARG;
  }
  if (p->batch < 1 || p->batch > h->st.max_batch || p->n_nodes < 1 || p->n_nodes > h->st.max_nodes || (!p->dt_nodes && !(p->dt > 0.0))) {
    h->err = "batch / n_nodes outside the handle's capacity, or dt <= 0";
    return HSQP_ERR_BAD_ARG;
  }
  if (r->batch != p->batch || r->n_nodes != p->n_nodes || (!r->node_times && r->dt != p->dt) || r->max_events < 1 || r->n_knots < 1) {
    h->err = "reference does not match the problem (batch, n_nodes, dt) or is empty";
    return HSQP_ERR_BAD_ARG;
  }
  if ((p->dt_nodes != nullptr) != (r->node_times != nullptr)) {
    h->err = "a non-uniform grid needs both hsqp_problem::dt_nodes and hsqp_reference::node_times";
    return HSQP_ERR_BAD_ARG;
  }
  for (int b = 0; b < r->batch; ++b)
    if (r->n_events[b] < 1 || r->n_events[b] > r->max_events) { h->err = "n_events outside [1, max_events]"; return HSQP_ERR_BAD_ARG; }
  if (!padding_is_zero(h, p)) return HSQP_ERR_BAD_ARG;
  HCHECK(hipSetDevice(h->device));
  const size_t B = p->batch, N = p->n_nodes, E = r->max_events, K = r->n_knots;
  // staging area for the compact reference (a few KB per instance)
  const size_t o_ne = 0, o_seq = o_ne + align256(B * 4), o_bad = o_seq + align256(B * (E + 1) * 4), o_ev = o_bad + 256,
               o_tt = o_ev + align256(B * E * 8), o_ts = o_tt + align256(B * K * 8), o_nt = o_ts + align256(B * K * NX * 8),
               total = o_nt + align256(r->node_times ? B * (N + 1) * 8 : 0);
  char* base = static_cast<char*>(stage_area(h, total));
  if (!base) { h->err = "hipMalloc failed (reference staging)"; return HSQP_ERR_OOM; }
  int* d_ne = reinterpret_cast<int*>(base + o_ne);
  int* d_seq = reinterpret_cast<int*>(base + o_seq);
  int* d_bad = reinterpret_cast<int*>(base + o_bad);
  double* d_ev = reinterpret_cast<double*>(base + o_ev);
  double* d_tt = reinterpret_cast<double*>(base + o_tt);
  double* d_ts = reinterpret_cast<double*>(base + o_ts);
  double* d_nt = r->node_times ? reinterpret_cast<double*>(base + o_nt) : nullptr;
  auto release = []() {};
  h->have_problem = false; h->have_solution = false;   // a failure below leaves no half-uploaded problem behind
  int rc = set_grid(h, p, false);
  if (rc != HSQP_OK) return rc;
  auto step = [&](hipError_t e, const char* what) { if (rc == HSQP_OK && e != hipSuccess) { h->err = std::string(what) + ": " + hipGetErrorString(e); rc = HSQP_ERR_HIP; } };
  if (d_nt) step(hipMemcpyAsync(d_nt, r->node_times, B * (N + 1) * 8, hipMemcpyHostToDevice, h->stream), "upload node_times");
  step(hipMemcpyAsync(d_ne, r->n_events, B * 4, hipMemcpyHostToDevice, h->stream), "upload n_events");
  step(hipMemcpyAsync(d_seq, r->mode_sequence, B * (E + 1) * 4, hipMemcpyHostToDevice, h->stream), "upload mode_sequence");
  step(hipMemcpyAsync(d_ev, r->event_times, B * E * 8, hipMemcpyHostToDevice, h->stream), "upload event_times");
  step(hipMemcpyAsync(d_tt, r->target_times, B * K * 8, hipMemcpyHostToDevice, h->stream), "upload target_times");
  step(hipMemcpyAsync(d_ts, r->target_states, B * K * NX * 8, hipMemcpyHostToDevice, h->stream), "upload target_states");
  step(hipMemsetAsync(d_bad, 0, 4, h->stream), "memset");
  step(hipMemcpyAsync(h->d_xinit, p->x_init, B * NX * 8, hipMemcpyHostToDevice, h->stream), "upload x_init");
  step(hipMemcpyAsync(h->d_x, p->x_traj, B * (N + 1) * NX * 8, hipMemcpyHostToDevice, h->stream), "upload x");
  step(hipMemcpyAsync(h->d_u, p->u_traj, B * N * NU * 8, hipMemcpyHostToDevice, h->stream), "upload u");
  if (rc == HSQP_OK) {
    const int total = (int)(B * (N + 1));
    HSQP_LAUNCH(k_params, dim3((total + 63) / 64), dim3(64), 0, h->stream, h->d_dm, r->swing, r->terrain_height, r->arm_swing, (int)E, d_ne, d_ev, d_seq,
                       (int)K, d_tt, d_ts, r->t0, r->dt, (const double*)d_nt, (int)N, (int)B, h->d_par, d_bad);
    if (h->hdm.formulation == HSQP_FORM_CENTROIDAL)   // torso task-space reference of every row
      HSQP_LAUNCH(k_params_cent_torso, dim3(total), dim3(64), sizeof(CentWST<false>), h->stream, h->d_dm, h->d_par);
    step(hipGetLastError(), "k_params");
  }
  int bad = 0;
  step(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, h->stream), "download status");
  step(hipStreamSynchronize(h->stream), "sync");
  release();
  if (rc != HSQP_OK) return rc;
  if (bad) { h->err = "a swing phase has no lift-off / touch-down inside the mode schedule"; return HSQP_ERR_BAD_ARG; }
  const bool same_shape = h->B == p->batch && h->N == p->n_nodes;
  h->B = p->batch; h->N = p->n_nodes; h->dt = p->dt;
  h->have_problem = true; h->have_solution = false;
  if (!(h->backoff_persistent && same_shape)) { h->seg_backoff = 0; h->seg_backoff_len = 0; }
  return HSQP_OK;
}

int hsqp_iterate_device(hsqp_handle* h, int n_iterations, int flags) {
  const int take_step = flags & HSQP_ITER_TAKE_STEP, want_kkt = (flags & HSQP_ITER_KKT) ? 1 : 0, linesearch = (flags & HSQP_ITER_LINESEARCH) ? 1 : 0;
  if (!h) return HSQP_ERR_BAD_ARG;
  if (!h->have_problem || n_iterations < 1) { h->err = "no problem uploaded or n_iterations < 1"; return HSQP_ERR_BAD_ARG; }
  HCHECK(hipSetDevice(h->device));
  const int B = h->B, N = h->N;
  const int nodes = B * N;
  const bool cent = h->hdm.formulation == HSQP_FORM_CENTROIDAL;
  HCHECK(hipMemsetAsync(h->d_status, 0, (size_t)B * sizeof(int), h->stream));   // once per call: the kernels OR into it
  const bool until_converged = (flags & HSQP_ITER_UNTIL_CONVERGED) != 0;
  h->iter_log.clear();
  h->last_iterations = 0;
  double ms_sum[4] = {0.0, 0.0, 0.0, 0.0};
  bool converged = false;
  for (int it = 0; it < n_iterations && !converged; ++it) {
    // (until_converged: any iteration may turn out to be the last one, so each is bracketed by the timing events and its times are summed)
    const bool last = it == n_iterations - 1 || until_converged;
    if (last) HCHECK(hipEventRecord(h->ev[0], h->stream));
    if (cent) {
      HSQP_LAUNCH(k_lq_cent2, dim3(nodes), dim3(CLQ_THREADS), sizeof(CentWST<true>), h->stream, h->d_dm, h->d_x, h->d_u, h->d_par, h->d_dt, N, h->d_rec);
    }
    else if (h->lq_limb) {   // limb lanes for the model and the node terms (16 nodes per wave), then the RK4 chain, a lane per column (hsqp_lql.h)
      // Two of the three kernels run one wave per SIMD (limb, rows), so a launch of config 4 is 1.56 rounds of the chip's 1024 SIMDs and the
      // last round of each leaves 44 % of them idle.  With lq_split = 2 the nodes are cut into two ranges, each going through its three
      // kernels on a stream of its own: a range's next kernel fills the SIMDs the other range's previous kernel leaves (measured, 256 x 100:
      // the three kernels 1.07 -> 0.97 ms; the bound of the arrangement, every SIMD busy throughout, is 0.92; three and more ranges lose
      // again: 1.02 / 1.10 / 1.38 ms at 3 / 4 / 8).  Only when a launch is more than one round.
      const dim3 qblock(QL_THREADS * QL_WAVES);
      constexpr int QG = QL_NODES * QL_WAVES;
      const int S = h->lq_split > 1 && (nodes + QG - 1) / QG > h->lq_round_blocks ? h->lq_split : 1;
      if (S > 1) {
        HCHECK(hipEventRecord(h->ev_fork, h->stream));
        for (int s = 1; s < S; ++s) HCHECK(hipStreamWaitEvent(h->aux[s - 1], h->ev_fork, 0));
      }
      for (int s = 0; s < S; ++s) {
        const int n0 = (int)(((long long)nodes * s / S) / QG * QG), n1 = s == S - 1 ? nodes : (int)(((long long)nodes * (s + 1) / S) / QG * QG);
        hipStream_t st = s == 0 ? h->stream : h->aux[s - 1];
        const dim3 qgrid((n1 - n0 + QG - 1) / QG);
        HSQP_LAUNCH(k_lq_limb, qgrid, qblock, 0, st, h->d_dm, h->d_x, h->d_u, h->d_dt, N, n1, h->d_rec, h->d_prof + 384, n0);
        HSQP_LAUNCH(k_lq_rows, qgrid, qblock, 0, st, h->d_dm, h->d_x, h->d_u, h->d_par, h->d_dt, N, n1, h->d_rec, h->d_prof + 384, n0, h->chain_fused ? 1 : 0);
        if (!h->chain_fused) HSQP_LAUNCH(k_lq_chain, dim3(n1 - n0), dim3(LQC_THREADS), 0, st, h->d_x, h->d_u, h->d_dt, N, h->d_rec, n0, 1);   // (fused: the columns' chain runs in k_project, the defect on the lanes of k_lq_rows)
        if (s > 0) { HCHECK(hipEventRecord(h->ev_join[s - 1], st)); HCHECK(hipStreamWaitEvent(h->stream, h->ev_join[s - 1], 0)); }
      }
    } else
      HSQP_LAUNCH(k_lq<true>, dim3(nodes), dim3(LQ_THREADS), sizeof(LqWS), h->stream, h->d_dm, h->d_x, h->d_u, h->d_par, h->d_dt, N,
                         h->d_rec, (double*)nullptr, h->d_prof, (const LsState*)nullptr);
    if (last) HCHECK(hipEventRecord(h->ev[1], h->stream));
    // The backward sweep: serial recursion, or — one or two instances on a long horizon, or on request — the associative scan over the
    // stages (hsqp_scan.h).  The scan inverts I + C1 J2 of partial horizons (condition number up to 1e5 centroidal, 1e9 whole-body): on
    // the QPs of a cold start or of a tracking MPC it reproduces the serial recursion to 1e-11 of the step's scale, on a far-from-
    // feasible line-search iterate it can lose five digits.  Its result is therefore GATED: the KKT residual of the QP is evaluated
    // (k_kkt, one small kernel + one 3 B-double read-back) and, if a residual exceeds the gate (scan_gate_accepts, hsqp_scan.h), the
    // iteration is redone with the serial recursion (hsqp_scan_fallbacks counts these).
    // > 0: the two-level sweep with segP segments per instance (hsqp_segment.h; opt-in).  A problem class whose sweeps keep failing the gate
    // (badly scaled QPs: |S| ~ 1e6 on perturbed centroidal batches) would pay sweep + fallback every iteration: after a rejection the handle
    // backs off to the serial recursion for 1, 3, 7, .. 63 iterations before it tries again
    int segP = segment_count(h, B, N);
    // (a sweep the caller FORCED by a flag is attempted every iteration: the back-off is part of the automatic choice only, so forced
    //  timings and the parity tests of the forced sweeps never silently measure the serial recursion)
    const bool forced_seg = (h->st.flags & HSQP_FLAG_SEGMENTED_RICCATI) != 0, forced_scan = (h->st.flags & HSQP_FLAG_PARALLEL_RICCATI) != 0;
    if (segP > 0 && !forced_seg && h->seg_backoff > 0) { --h->seg_backoff; ++h->backoff_iterations; segP = 0; }
    bool pscan = segP == 0 && !(h->st.flags & HSQP_FLAG_SERIAL_RICCATI) && !(h->st.flags & HSQP_FLAG_SEGMENTED_RICCATI) &&
                 ((h->st.flags & HSQP_FLAG_PARALLEL_RICCATI) || (B <= HSQP_SCAN_AUTO_BATCH && N >= HSQP_SCAN_AUTO_MIN_NODES));
    // the same back-off for the scan: a rejected scan costs scan + serial sweep (0.55 + 1.45 ms at one whole-body instance), and the
    // iterates that fail the gate — far-from-feasible line-search iterates — come in runs
    if (pscan && !forced_scan && h->seg_backoff > 0) { --h->seg_backoff; ++h->backoff_iterations; pscan = false; }
    const bool scan = pscan || segP > 0;        // either way a KKT-gated sweep with the serial recursion as fallback
    // The joint rows of A~ / B~ (46 of 58: scaled copies of rows of [Px | Pu]) are written only for those who read A~ / B~ as dense blocks: the
    // parallel-in-time and two-level sweeps, the KKT report, the centroidal stage.  The whole-body serial sweep works on the factors.
    const bool joint_rows = cent || !h->ric_fact || scan || want_kkt;
    HSQP_LAUNCH(k_project, dim3(nodes), dim3(PROJ_THREADS), sizeof(ProjWS), h->stream, h->d_rec, h->d_dt, h->d_qp, h->d_prof + 128, cent ? 1 : 0, joint_rows ? 1 : 0, (!cent && h->lq_limb && h->chain_fused) ? 1 : 0);
    if (h->has_events) HSQP_LAUNCH(k_jump, dim3(nodes), dim3(256), 0, h->stream, h->d_dt, h->d_rec, h->d_qp);
    if (last) HCHECK(hipEventRecord(h->ev[2], h->stream));
    if (want_kkt && !h->d_vf) {
      const size_t bytes = (size_t)h->st.max_batch * (h->st.max_nodes + 1) * VF_SIZE * 8;
      if (hipMalloc(&h->d_vf, bytes) != hipSuccess) { h->d_vf = nullptr; h->err = "hipMalloc failed (value function for the KKT check, " + std::to_string(bytes) + " bytes)"; return HSQP_ERR_OOM; }
      poison_hbm(h->d_vf, bytes);
    }
    const int Bm = h->st.max_batch;
    const size_t gate_bytes = (size_t)Bm * 3 * 8 + (((size_t)Bm * sizeof(int) + 7) / 8) * 8;   // [kkt | |g|_inf | scan flags]
    int ut_given = 0;   // the last sweep's roll-out left ut = k + K dx of every node in d_ut (the serial roll-out does, the scan's closed-loop roll-out does not)
    int fj_given = 0;   // ... and rows 12 .. 34 of Px dx + Pu ut in d_fj (the factored roll-out does)
    auto launch_sweep = [&](bool use_scan, bool need_vf) -> int {
      ut_given = (use_scan && segP == 0) ? 0 : 1;
      fj_given = 0;
      if (use_scan && segP > 0) return cent ? launch_segmented<CNX>(h, B, N, segP, want_kkt != 0) : launch_segmented<NX>(h, B, N, segP, want_kkt != 0);
      if (use_scan) return cent ? launch_scan<CNX>(h, B, N, need_vf, 1) : launch_scan<NX>(h, B, N, need_vf, HSQP_SCAN_WB_REFINEMENTS);
      if (cent)   // the serial recursion on the 35 centroidal states only (the padding states are decoupled)
        HSQP_LAUNCH(k_riccati<CNX>, dim3(B), dim3(RIC_THREADS), sizeof(RicWS), h->stream, h->d_dm, h->d_xinit, h->d_x, h->d_par, h->d_qp,
                           h->d_ric, N, h->d_dx, h->d_status, h->d_prof + 256, need_vf ? h->d_vf : (double*)nullptr, h->d_ut);
      else if (h->ric_fact)
        { HSQP_LAUNCH(k_riccati_fact, dim3(B), dim3(RIC_THREADS), sizeof(RicFWS), h->stream, h->d_dm, h->d_xinit, h->d_x, h->d_par, h->d_qp, h->d_dt,
                           h->d_ric, N, h->d_dx, h->d_status, h->d_prof + 256, need_vf ? h->d_vf : (double*)nullptr, h->d_ut, h->d_fj); fj_given = 1; }
      else
        HSQP_LAUNCH(k_riccati<NX>, dim3(B), dim3(RIC_THREADS), sizeof(RicWS), h->stream, h->d_dm, h->d_xinit, h->d_x, h->d_par, h->d_qp,
                           h->d_ric, N, h->d_dx, h->d_status, h->d_prof + 256, need_vf ? h->d_vf : (double*)nullptr, h->d_ut);
      return HSQP_OK;
    };
    auto launch_step = [&]() {
      if (cent)
        HSQP_LAUNCH(k_step, dim3(nodes), dim3(64), 0, h->stream, h->d_qp, h->d_ric, h->d_dx, h->d_x, h->d_u, N, 1.0, h->d_ut, h->d_du,
                           h->d_xnew, h->d_unew, h->d_stepinfo, ut_given, fj_given ? (const double*)h->d_fj : (const double*)nullptr);
      else if (h->value_quad) {   // whole-body: the step (HBM-bound), then the value pass on quads of lanes (hsqp_lqv.h)
        HSQP_LAUNCH(k_step, dim3(nodes), dim3(64), 0, h->stream, h->d_qp, h->d_ric, h->d_dx, h->d_x, h->d_u, N, 1.0, h->d_ut, h->d_du,
                           h->d_xnew, h->d_unew, h->d_stepinfo, ut_given, fj_given ? (const double*)h->d_fj : (const double*)nullptr);
        HSQP_LAUNCH(k_value_quad, dim3((nodes + QV_NODES * QV_WAVES - 1) / (QV_NODES * QV_WAVES)), dim3(QV_THREADS * QV_WAVES), 0, h->stream, h->d_dm, h->d_xnew, h->d_unew, h->d_par, h->d_dt,
                           N, nodes, h->d_misc, (const LsState*)nullptr);
      } else   // a tree with more than four limbs: the phase form of the value pass, fused with the step (k_step_value)
        HSQP_LAUNCH(k_step_value, dim3(nodes), dim3(LQV_THREADS), sizeof(LqWST<false>), h->stream, h->d_dm, h->d_qp, h->d_ric, h->d_dx, h->d_x, h->d_u,
                           h->d_par, h->d_dt, N, 1.0, h->d_ut, h->d_du, h->d_xnew, h->d_unew, h->d_stepinfo, h->d_misc, h->d_prof + 384, ut_given,
                           fj_given ? (const double*)h->d_fj : (const double*)nullptr);
    };
    auto launch_kkt = [&](bool from_scan) -> int {
      if (!from_scan) HCHECK(hipMemsetAsync(h->d_kkt, 0, gate_bytes, h->stream));   // kkt, |g|_inf (and the scan flags) are one block; the scan path has zeroed it before its kernels
      HSQP_LAUNCH(k_kkt, dim3(nodes), dim3(256), 0, h->stream, h->d_xinit, h->d_x, h->d_qp, from_scan ? h->d_vf2 : h->d_vf, h->d_dx, h->d_ut, N, h->d_kkt, h->d_ginf);
      return HSQP_OK;
    };
    if (scan) HCHECK(hipMemsetAsync(h->d_kkt, 0, gate_bytes, h->stream));   // also the flags the scan kernels OR into
    { const int rc = launch_sweep(scan, want_kkt || scan); if (rc != HSQP_OK) return rc; }
    if (last) HCHECK(hipEventRecord(h->ev[3], h->stream));   // kernel_ms buckets: {lq, project, riccati (backward + forward sweep), step + value pass + reductions}
    launch_step();
    if (scan) {   // the gate's inputs: KKT residuals, |g|_inf and the scan kernels' flags travel to pinned host memory while the kernels below run
      // (two-level sweep: the gate block holds its boundary-consistency numbers instead — no KKT kernel; the KKT report, if asked for, follows the verdict)
      if (segP == 0) { const int rc = launch_kkt(true); if (rc != HSQP_OK) return rc; }
      else HSQP_LAUNCH(k_kkt_boundaries, dim3(B * (segP - 1)), dim3(256), 0, h->stream, h->d_xinit, h->d_x, h->d_qp, (const double*)h->d_vf2, h->d_dx, h->d_ut, N, segP,
                              h->d_kkt, h->d_ginf);
      HCHECK(hipMemcpyAsync(h->h_gate, h->d_kkt, gate_bytes, hipMemcpyDeviceToHost, h->stream));
    } else if (want_kkt) {
      const int rc = launch_kkt(false);
      if (rc != HSQP_OK) return rc;
    }
    auto launch_perf = [&]() {   // value pass of the centroidal trial, performance indices before / after, line-search state of the full-step trial
      if (cent)
        HSQP_LAUNCH(k_lq_cent2_value, dim3(nodes), dim3(64), sizeof(CentWST<false>), h->stream, h->d_dm, h->d_xnew, h->d_unew, h->d_par, h->d_dt, N, h->d_misc,
                           (const LsState*)nullptr);
      HSQP_LAUNCH(k_perf_trio, dim3(B, 3), dim3(64), 0, h->stream, h->d_dm, h->d_rec + REC_MISC, REC_SIZE, h->d_x, h->d_misc, 8, h->d_xnew, h->d_par, N,
                         h->d_perf_before, h->d_perf_after, h->d_stepinfo, h->d_dx, h->d_ls);
    };
    launch_perf();   // speculatively on the scan's step: the gate is read only now, so the host round trip hides behind these kernels
    const bool ev4_early = last && !linesearch;
    if (ev4_early) HCHECK(hipEventRecord(h->ev[4], h->stream));
    if (scan) {
      HCHECK(hipStreamSynchronize(h->stream));
      const double* hk = h->h_gate;
      const int* flags = reinterpret_cast<const int*>(hk + 3 * Bm);
      bool accept = true;
      for (int b = 0; b < B; ++b) {
        // (a flag = a bad pivot / failed factorisation inside the scan: the serial recursion decides what is reported in d_status)
        if (!scan_gate_accepts(hk[2 * b], hk[2 * b + 1], hk[2 * Bm + b], flags[b])) accept = false;
      }
      if (h->seg_debug && segP > 0) {
        double m0 = 0, m1 = 0, m2 = 0; int fl = 0;
        for (int b = 0; b < B; ++b) { m0 = fmax(m0, hk[2 * b]); m1 = fmax(m1, hk[2 * b + 1]); m2 = fmax(m2, hk[2 * Bm + b]); fl |= flags[b]; }
        fprintf(stderr, "[hsqp seg gate] P=%d boundary-stage KKT stat %.3e prim %.3e |g| %.3e flags %d accept %d\n", segP, m0, m1, m2, fl, (int)accept);
      }
      if (accept && segP > 0 && want_kkt) {   // the KKT report of an accepted two-level sweep (the gate block is reused: zero it first)
        HCHECK(hipMemsetAsync(h->d_kkt, 0, gate_bytes, h->stream));
        HSQP_LAUNCH(k_kkt, dim3(nodes), dim3(256), 0, h->stream, h->d_xinit, h->d_x, h->d_qp, h->d_vf2, h->d_dx, h->d_ut, N, h->d_kkt, h->d_ginf);
      }
      if (accept) h->seg_backoff_len = 0;
      else { h->seg_backoff_len = std::min(2 * h->seg_backoff_len + 1, 63); h->seg_backoff = h->seg_backoff_len; }
      if (!accept) {
        ++h->scan_fallbacks;
        if (want_kkt && !h->d_vf) { h->err = "internal: value-function buffer missing"; return HSQP_ERR_HIP; }
        { const int rc = launch_sweep(false, want_kkt != 0); if (rc != HSQP_OK) return rc; }
        if (last) HCHECK(hipEventRecord(h->ev[3], h->stream));
        launch_step();
        if (want_kkt) { const int rc = launch_kkt(false); if (rc != HSQP_OK) return rc; }
        launch_perf();
        if (ev4_early) HCHECK(hipEventRecord(h->ev[4], h->stream));
      }
    }
    if (linesearch) {
      // back-tracking: decide the pending trials on the device, shorten the rejected steps, re-evaluate only those instances
      LsSettings lst{h->ls_settings.g_max, h->ls_settings.g_min, h->ls_settings.gamma_c, h->ls_settings.armijo_factor, h->ls_settings.alpha_decay,
                     h->ls_settings.alpha_min, h->ls_settings.delta_tol};
      // alpha_decay^n < alpha_min ends every instance's back-tracking with a zero step (ls_decide), so this bound is never the
      // reason the loop ends; hsqp_set_linesearch keeps it <= HSQP_LS_MAX_TRIALS
      const int max_trials = ls_max_trials(h->ls_settings);
      int still_active = 0;
      // Two trials per host round trip: the follow-up of a rejected trial (shortened trajectory, its value pass, its performance index) is
      // launched WITHOUT waiting for the verdict — every one of these kernels leaves the instances alone whose search is over (LsState::
      // active / dirty), so after an accepted trial they are no-ops of a few microseconds, and the host reads the counters of the second
      // decision only.  (One synchronisation per trial cost 40 us each at one instance — review item of round 1.)
      constexpr int LS_SPECULATIVE = 2;
      for (int trial = 0; trial < max_trials;) {
        int counts[2] = {0, 0};
        for (int r = 0; r < LS_SPECULATIVE && trial < max_trials; ++r, ++trial) {
          HCHECK(hipMemsetAsync(h->d_counts, 0, 2 * sizeof(int), h->stream));
          HSQP_LAUNCH(k_ls_decide, dim3((B + 63) / 64), dim3(64), 0, h->stream, lst, h->d_perf_before, h->d_perf_after, B, h->d_ls, h->d_counts);
          HSQP_LAUNCH(k_ls_retake, dim3(nodes), dim3(64), 0, h->stream, h->d_x, h->d_u, h->d_dx, h->d_du, N, h->d_ls, h->d_xnew, h->d_unew);
          if (cent)
            HSQP_LAUNCH(k_lq_cent2_value, dim3(nodes), dim3(64), sizeof(CentWST<false>), h->stream, h->d_dm, h->d_xnew, h->d_unew, h->d_par, h->d_dt, N, h->d_misc,
                               (const LsState*)h->d_ls);
          else if (h->value_quad)
            HSQP_LAUNCH(k_value_quad, dim3((nodes + QV_NODES * QV_WAVES - 1) / (QV_NODES * QV_WAVES)), dim3(QV_THREADS * QV_WAVES), 0, h->stream, h->d_dm, h->d_xnew, h->d_unew, h->d_par, h->d_dt,
                               N, nodes, h->d_misc, (const LsState*)h->d_ls);
          else
            HSQP_LAUNCH(k_lq<false>, dim3(nodes), dim3(LQV_THREADS), sizeof(LqWST<false>), h->stream, h->d_dm, h->d_xnew, h->d_unew, h->d_par, h->d_dt,
                               N, (double*)nullptr, h->d_misc, (long long*)nullptr, (const LsState*)h->d_ls);
          HSQP_LAUNCH(k_perf_reduce, dim3(B), dim3(64), 0, h->stream, h->d_dm, h->d_misc, 8, h->d_xnew, h->d_par, N, h->d_perf_after,
                             (const LsState*)h->d_ls);
        }
        HCHECK(hipMemcpyAsync(counts, h->d_counts, sizeof(counts), hipMemcpyDeviceToHost, h->stream));
        HCHECK(hipStreamSynchronize(h->stream));
        still_active = counts[1];
        if (counts[1] == 0) break;
      }
      if (still_active) { h->err = "line search: trials exhausted with instances still undecided (internal error)"; return HSQP_ERR_NUMERIC; }
    }
    h->ls_ran = linesearch != 0;
    if (last && !ev4_early) HCHECK(hipEventRecord(h->ev[4], h->stream));
    h->last_iterations = it + 1;
    if (until_converged) {
      // SqpSolver::checkConvergence (upstream ocs2_sqp, restated from the published source; the fork's copy is absent): after ITERATIONS
      // (the loop bound) STEPSIZE (no step length accepted), then METRICS (|merit after - merit before| < costTol and the constraint
      // violation after the step below g_min), then PRIMAL (alpha |dx| and alpha |du| below deltaTol); the loop ends when EVERY instance
      // meets one of them.  The record of the iteration goes to the log.  The read-back lands in the handle's host buffers (no
      // allocation per iteration); with n_iterations == 1 the loop ends regardless of the verdict, but the log is still filled (the
      // adaptor reads step length and type from it).
      hsqp_handle::IterLog rec;
      rec.perf.resize(B); rec.alpha.resize(B); rec.type.resize(B);
      h->h_ls.resize(B); h->h_perf_before.resize(B);
      HCHECK(hipMemcpyAsync(h->h_ls.data(), h->d_ls, (size_t)B * sizeof(LsState), hipMemcpyDeviceToHost, h->stream));
      HCHECK(hipMemcpyAsync(rec.perf.data(), h->d_perf_after, (size_t)B * sizeof(hsqp_perf), hipMemcpyDeviceToHost, h->stream));
      HCHECK(hipMemcpyAsync(h->h_perf_before.data(), h->d_perf_before, (size_t)B * sizeof(hsqp_perf), hipMemcpyDeviceToHost, h->stream));
      HCHECK(hipStreamSynchronize(h->stream));
      const std::vector<LsState>& ls = h->h_ls;
      converged = true;
      for (int b = 0; b < B; ++b) {
        rec.alpha[b] = linesearch ? ls[b].alpha : 1.0;
        rec.type[b] = linesearch ? ls[b].step_type : HSQP_STEP_FULL;
        const bool zero_step = linesearch && ls[b].step_type == HSQP_STEP_ZERO;
        const bool metrics = fabs(rec.perf[b].merit - h->h_perf_before[b].merit) < h->ls_settings.cost_tol && ls_violation(rec.perf[b]) < h->ls_settings.g_min;
        const bool primal = rec.alpha[b] * ls[b].dxnorm < h->ls_settings.delta_tol && rec.alpha[b] * ls[b].dunorm < h->ls_settings.delta_tol;
        if (!zero_step && !metrics && !primal) converged = false;
      }
      h->iter_log.push_back(std::move(rec));
      float msi[4];
      for (int i = 0; i < 4; ++i) { HCHECK(hipEventElapsedTime(&msi[i], h->ev[i], h->ev[i + 1])); ms_sum[i] += msi[i]; }
    }
    const bool more = it + 1 < n_iterations && !converged;
    if (take_step && more) {
      HCHECK(hipMemcpyAsync(h->d_x, h->d_xnew, (size_t)B * (N + 1) * NX * 8, hipMemcpyDeviceToDevice, h->stream));
      HCHECK(hipMemcpyAsync(h->d_u, h->d_unew, (size_t)B * N * NU * 8, hipMemcpyDeviceToDevice, h->stream));
    }
  }
  HCHECK(hipGetLastError());
  HCHECK(hipStreamSynchronize(h->stream));
  if (until_converged) {
    for (int i = 0; i < 4; ++i) h->kernel_ms[i] = ms_sum[i];
  } else {
    float ms[4];
    for (int i = 0; i < 4; ++i) HCHECK(hipEventElapsedTime(&ms[i], h->ev[i], h->ev[i + 1]));
    for (int i = 0; i < 4; ++i) h->kernel_ms[i] = ms[i];
  }
  h->kernel_ms[4] = h->kernel_ms[0] + h->kernel_ms[1] + h->kernel_ms[2] + h->kernel_ms[3];
  h->have_solution = true;
  return HSQP_OK;
}

int hsqp_last_iterations(const hsqp_handle* h) { return h ? h->last_iterations : -1; }

int hsqp_iteration_log(const hsqp_handle* h, int iteration, hsqp_perf* perf, double* alpha, int32_t* step_type) {
  if (!h || iteration < 0 || iteration >= (int)h->iter_log.size()) return HSQP_ERR_BAD_ARG;
  const hsqp_handle::IterLog& r = h->iter_log[iteration];
  for (size_t b = 0; b < r.perf.size(); ++b) {
    if (perf) perf[b] = r.perf[b];
    if (alpha) alpha[b] = r.alpha[b];
    if (step_type) step_type[b] = r.type[b];
  }
  return HSQP_OK;
}

int hsqp_update_weights(hsqp_handle* h, const double* Q, const double* R, const double* Qf) {
  if (!h) return HSQP_ERR_BAD_ARG;
  // explicit lengths (Q: 58, R: 35, Qf: 58; the same buffer may be passed twice) and the conditions build_dev_model (hsqp_create, hsqp_update_term_weights)
  // applies to them: state weights finite and >= 0, input weights finite and > 0 (the reduced Hessian
  // Lam = R~ + B~^T S B~ of every stage has to stay positive definite)
  const struct { const double* w; int n; bool positive; const char* name; } sets[3] = {{Q, NX, false, "Q"}, {R, NU, true, "R"}, {Qf, NX, false, "Qf"}};
  for (const auto& st : sets) {
    if (!st.w) continue;
    for (int i = 0; i < st.n; ++i) {
      const bool must_be_positive = st.positive;   // (both formulations use all 35 inputs)
      if (!std::isfinite(st.w[i]) || st.w[i] < 0.0 || (must_be_positive && !(st.w[i] > 0.0))) {
        h->err = std::string("hsqp_update_weights: ") + st.name + " must be finite and " + (st.positive ? "> 0" : ">= 0");
        return HSQP_ERR_BAD_ARG;
      }
    }
  }
  if (h->hdm.formulation == HSQP_FORM_CENTROIDAL)
    for (const double* w : {Q, Qf})
      for (int i = HSQP_CNX; w && i < NX; ++i)
        if (w[i] != 0.0) { h->err = "hsqp_update_weights: centroidal Q / Qf beyond the 35 centroidal states must be zero"; return HSQP_ERR_BAD_ARG; }
  HCHECK(hipSetDevice(h->device));
  if (Q) { memcpy(h->hdm.Q, Q, NX * 8); memcpy(h->md.Q, Q, NX * 8); }
  if (R) { memcpy(h->hdm.R, R, NU * 8); memcpy(h->md.R, R, NU * 8); }
  if (Qf) { memcpy(h->hdm.Qf, Qf, NX * 8); memcpy(h->md.Qf, Qf, NX * 8); }
  HCHECK(hipStreamSynchronize(h->stream));   // no kernel of an earlier call may still be reading the image
  HCHECK(hipMemcpy(h->d_dm, &h->hdm, sizeof(DevModel), hipMemcpyHostToDevice));
  return HSQP_OK;
}

int hsqp_get_term_weights(const hsqp_handle* h, hsqp_term_weights* out) {
  if (!h || !out) return HSQP_ERR_BAD_ARG;
  const hsqp_model_desc& m = h->md;
  memcpy(out->foot_sqrt_w, m.foot_sqrt_w, sizeof(out->foot_sqrt_w));
  out->gain_pos_z = m.gain_pos_z; out->gain_ori = m.gain_ori; out->gain_linvel_z = m.gain_linvel_z; out->gain_linvel_xy = m.gain_linvel_xy;
  out->gain_angvel = m.gain_angvel; out->gain_linacc_z = m.gain_linacc_z; out->gain_linacc_xy = m.gain_linacc_xy; out->gain_angacc = m.gain_angacc;
  out->friction_barrier = m.friction_barrier; out->moment_barrier = m.moment_barrier; out->joint_limit_barrier = m.joint_limit_barrier;
  out->collision_barrier = m.collision_barrier;
  memcpy(out->torso_sqrt_w, m.torso_sqrt_w, sizeof(out->torso_sqrt_w));
  memcpy(out->cent_foot_sqrt_w, m.cent_foot_sqrt_w, sizeof(out->cent_foot_sqrt_w));
  memcpy(out->ext_torque_sqrt_w, m.ext_torque_sqrt_w, sizeof(out->ext_torque_sqrt_w));
  return HSQP_OK;
}

int hsqp_update_term_weights(hsqp_handle* h, const hsqp_term_weights* w) {
  if (!h) return HSQP_ERR_BAD_ARG;
  if (!w) { h->err = "hsqp_update_term_weights: null argument"; return HSQP_ERR_BAD_ARG; }
  const double* all = reinterpret_cast<const double*>(w);
  for (size_t i = 0; i < sizeof(hsqp_term_weights) / sizeof(double); ++i)
    if (!std::isfinite(all[i])) { h->err = "hsqp_update_term_weights: every entry must be finite"; return HSQP_ERR_BAD_ARG; }
  // (the barrier parameters, weights and gains are validated by build_dev_model below: the same checks as hsqp_create)
  hsqp_model_desc m = h->md;
  memcpy(m.foot_sqrt_w, w->foot_sqrt_w, sizeof(m.foot_sqrt_w));
  m.gain_pos_z = w->gain_pos_z; m.gain_ori = w->gain_ori; m.gain_linvel_z = w->gain_linvel_z; m.gain_linvel_xy = w->gain_linvel_xy;
  m.gain_angvel = w->gain_angvel; m.gain_linacc_z = w->gain_linacc_z; m.gain_linacc_xy = w->gain_linacc_xy; m.gain_angacc = w->gain_angacc;
  m.friction_barrier = w->friction_barrier; m.moment_barrier = w->moment_barrier; m.joint_limit_barrier = w->joint_limit_barrier;
  m.collision_barrier = w->collision_barrier;
  memcpy(m.torso_sqrt_w, w->torso_sqrt_w, sizeof(m.torso_sqrt_w));
  memcpy(m.cent_foot_sqrt_w, w->cent_foot_sqrt_w, sizeof(m.cent_foot_sqrt_w));
  memcpy(m.ext_torque_sqrt_w, w->ext_torque_sqrt_w, sizeof(m.ext_torque_sqrt_w));
  DevModel dm;
  const std::string e = build_dev_model(m, dm);       // the validation of hsqp_create on the updated description
  if (!e.empty()) { h->err = "hsqp_update_term_weights: " + e; return HSQP_ERR_BAD_ARG; }
  HCHECK(hipSetDevice(h->device));
  HCHECK(hipStreamSynchronize(h->stream));   // no kernel of an earlier call may still be reading the image
  h->md = m; h->hdm = dm;
  HCHECK(hipMemcpy(h->d_dm, &h->hdm, sizeof(DevModel), hipMemcpyHostToDevice));
  return HSQP_OK;
}

static int download_impl(hsqp_handle* h, hsqp_solution* s, bool device_dst) {
  if (!h) return HSQP_ERR_BAD_ARG;
  if (!s || !h->have_solution) { h->err = "no solution on the device"; return HSQP_ERR_BAD_ARG; }
  if (device_dst && (s->alpha || s->step_type || s->armijo)) { h->err = "hsqp_download_device: alpha / step_type / armijo must be NULL (host-side fields)"; return HSQP_ERR_BAD_ARG; }
  HCHECK(hipSetDevice(h->device));
  const size_t B = h->B, N = h->N;
  const hipMemcpyKind kind = device_dst ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  if (s->x) HCHECK(hipMemcpy(s->x, h->d_xnew, B * (N + 1) * NX * 8, kind));
  if (s->u) HCHECK(hipMemcpy(s->u, h->d_unew, B * N * NU * 8, kind));
  if (s->dx) HCHECK(hipMemcpy(s->dx, h->d_dx, B * (N + 1) * NX * 8, kind));
  if (s->du) HCHECK(hipMemcpy(s->du, h->d_du, B * N * NU * 8, kind));
  if (s->perf_before) HCHECK(hipMemcpy(s->perf_before, h->d_perf_before, B * sizeof(hsqp_perf), kind));
  if (s->perf_after) HCHECK(hipMemcpy(s->perf_after, h->d_perf_after, B * sizeof(hsqp_perf), kind));
  if (s->kkt) HCHECK(hipMemcpy(s->kkt, h->d_kkt, B * 2 * 8, kind));
  if (s->grad_inf) HCHECK(hipMemcpy(s->grad_inf, h->d_ginf, B * 8, kind));
  if (s->alpha || s->step_type || s->armijo) {
    std::vector<LsState> ls(B);
    HCHECK(hipMemcpy(ls.data(), h->d_ls, B * sizeof(LsState), hipMemcpyDeviceToHost));
    for (size_t b = 0; b < B; ++b) {
      if (s->alpha) s->alpha[b] = h->ls_ran ? ls[b].alpha : 1.0;
      if (s->step_type) s->step_type[b] = h->ls_ran ? ls[b].step_type : HSQP_STEP_FULL;
      if (s->armijo) s->armijo[b] = ls[b].armijo;
    }
  }
  std::vector<int> status(B);
  HCHECK(hipMemcpy(status.data(), h->d_status, B * sizeof(int), hipMemcpyDeviceToHost));
  s->timings.lq_approximation = 1e-3 * (h->kernel_ms[0] + h->kernel_ms[1]);
  s->timings.solve_qp = 1e-3 * h->kernel_ms[2];
  s->timings.linesearch = 1e-3 * h->kernel_ms[3];
  s->timings.compute_controller = 0.0;
  s->timings.total = 1e-3 * h->kernel_ms[4];
  for (size_t b = 0; b < B; ++b)
    if (status[b]) {
      h->err = "instance " + std::to_string(b) + ": " + ((status[b] & 1) ? "rank-deficient equality Jacobian D " : "") +
               ((status[b] & 2) ? "reduced Hessian not positive definite" : "");
      return HSQP_ERR_NUMERIC;
    }
  return HSQP_OK;
}

int hsqp_download(hsqp_handle* h, hsqp_solution* s) { return download_impl(h, s, false); }
int hsqp_download_device(hsqp_handle* h, hsqp_solution* s) { return download_impl(h, s, true); }

int hsqp_host_register(void* buffer, size_t bytes) {
  if (!buffer || bytes == 0) return HSQP_ERR_BAD_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return HSQP_ERR_NO_DEVICE;
  return hipHostRegister(buffer, bytes, hipHostRegisterDefault) == hipSuccess ? HSQP_OK : ((void)hipGetLastError(), HSQP_ERR_HIP);
}
int hsqp_host_unregister(void* buffer) {
  if (!buffer) return HSQP_ERR_BAD_ARG;
  return hipHostUnregister(buffer) == hipSuccess ? HSQP_OK : ((void)hipGetLastError(), HSQP_ERR_HIP);
}

int hsqp_solve(hsqp_handle* h, const hsqp_problem* problem, hsqp_solution* solution) {
  int rc = hsqp_upload(h, problem);
  if (rc != HSQP_OK) return rc;
  // the KKT residual is a report, not part of the step: it is only computed when the caller asks for it (solution->kkt / grad_inf)
  const int want_kkt = (solution && (solution->kkt || solution->grad_inf)) ? HSQP_ITER_KKT : 0;
  rc = hsqp_iterate_device(h, 1, HSQP_ITER_TAKE_STEP | want_kkt | ((h->st.flags & HSQP_FLAG_LINESEARCH) ? HSQP_ITER_LINESEARCH : 0));
  if (rc != HSQP_OK) return rc;
  return hsqp_download(h, solution);
}

static int run_policy(hsqp_handle* h, int n, bool from_solution, const double* s_or_x, const double* u_in, double* x_out, double* u_out, double* tau) {
  HCHECK(hipSetDevice(h->device));
  const bool cent = h->hdm.formulation == HSQP_FORM_CENTROIDAL;
  const size_t nin = from_solution ? (size_t)n : (size_t)n * (NX + NU);
  const size_t o_in = 0, o_x = o_in + align256(nin * 8), o_u = o_x + align256((size_t)n * NX * 8), o_tau = o_u + align256((size_t)n * NU * 8),
               o_xw = o_tau + align256((size_t)n * NJ * 8), o_uw = o_xw + align256((size_t)n * NX * 8), total = o_uw + align256((size_t)n * NU * 8);
  char* base = static_cast<char*>(stage_area(h, total));
  if (!base) { h->err = "hipMalloc failed (policy evaluation staging)"; return HSQP_ERR_OOM; }
  double* d_in = reinterpret_cast<double*>(base + o_in);
  double* d_x = reinterpret_cast<double*>(base + o_x);
  double* d_u = reinterpret_cast<double*>(base + o_u);
  double* d_tau = reinterpret_cast<double*>(base + o_tau);
  double* d_xw = reinterpret_cast<double*>(base + o_xw);
  double* d_uw = reinterpret_cast<double*>(base + o_uw);
  int rc = HSQP_OK;
  auto step = [&](hipError_t e, const char* what) { if (rc == HSQP_OK && e != hipSuccess) { h->err = std::string(what) + ": " + hipGetErrorString(e); rc = HSQP_ERR_HIP; } };
  if (from_solution) {
    step(hipMemcpyAsync(d_in, s_or_x, (size_t)n * 8, hipMemcpyHostToDevice, h->stream), "upload s");
  } else {
    step(hipMemcpyAsync(d_in, s_or_x, (size_t)n * NX * 8, hipMemcpyHostToDevice, h->stream), "upload x");
    step(hipMemcpyAsync(d_in + (size_t)n * NX, u_in, (size_t)n * NU * 8, hipMemcpyHostToDevice, h->stream), "upload u");
  }
  if (rc == HSQP_OK) {
    step(hipFuncSetAttribute((const void*)k_policy_torques, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PolicyWS)), "hipFuncSetAttribute");
    if (from_solution)
      HSQP_LAUNCH(k_policy_inputs, dim3(n), dim3(64), 0, h->stream, (const double*)h->d_xnew, (const double*)h->d_unew, h->N, h->dt,
                         h->uniform_grid ? (const double*)nullptr : (const double*)h->d_dt, (const double*)d_in, (const double*)nullptr, (const double*)nullptr, d_x, d_u);
    else
      HSQP_LAUNCH(k_policy_inputs, dim3(n), dim3(64), 0, h->stream, (const double*)nullptr, (const double*)nullptr, 0, 0.0, (const double*)nullptr,
                         (const double*)nullptr, (const double*)d_in, (const double*)(d_in + (size_t)n * NX), d_x, d_u);
    if (cent) {
      HSQP_LAUNCH(k_cent_policy_map, dim3(n), dim3(64), sizeof(CentWST<false>), h->stream, h->d_dm, n, (const double*)d_x, (const double*)d_u, d_xw, d_uw);
      HSQP_LAUNCH(k_policy_torques, dim3(n), dim3(128), sizeof(PolicyWS), h->stream, h->d_dm, (const double*)d_xw, (const double*)d_uw, d_tau);
    } else {
      HSQP_LAUNCH(k_policy_torques, dim3(n), dim3(128), sizeof(PolicyWS), h->stream, h->d_dm, (const double*)d_x, (const double*)d_u, d_tau);
    }
    step(hipGetLastError(), "k_policy");
  }
  if (x_out) step(hipMemcpyAsync(x_out, d_x, (size_t)n * NX * 8, hipMemcpyDeviceToHost, h->stream), "download x");
  if (u_out) step(hipMemcpyAsync(u_out, d_u, (size_t)n * NU * 8, hipMemcpyDeviceToHost, h->stream), "download u");
  if (tau) step(hipMemcpyAsync(tau, d_tau, (size_t)n * NJ * 8, hipMemcpyDeviceToHost, h->stream), "download tau");
  step(hipStreamSynchronize(h->stream), "sync");
  return rc;
}

int hsqp_joint_torques(hsqp_handle* h, int n, const double* x, const double* u, double* tau) {
  if (!h) return HSQP_ERR_BAD_ARG;
  if (n < 1 || !x || !u || !tau) { h->err = "hsqp_joint_torques: n < 1 or null pointer"; return HSQP_ERR_BAD_ARG; }
  return run_policy(h, n, false, x, u, nullptr, nullptr, tau);
}

int hsqp_evaluate_policy(hsqp_handle* h, const double* s, double* x, double* u, double* tau) {
  if (!h) return HSQP_ERR_BAD_ARG;
  if (!h->have_solution) { h->err = "no solution on the device"; return HSQP_ERR_BAD_ARG; }
  if (!s) { h->err = "hsqp_evaluate_policy: null time offsets"; return HSQP_ERR_BAD_ARG; }
  return run_policy(h, h->B, true, s, nullptr, x, u, tau);
}

int hsqp_last_kernel_ms(hsqp_handle* h, double out_ms[5]) {
  if (!h || !out_ms) return HSQP_ERR_BAD_ARG;
  for (int i = 0; i < 5; ++i) out_ms[i] = h->kernel_ms[i];
  return HSQP_OK;
}

long long hsqp_debug_read(hsqp_handle* h, int what, void* dst, long long bytes) {
  if (!h) return HSQP_ERR_BAD_ARG;
  if (what == HSQP_BLK_FORMS) {
    const int forms[5] = {h->lq_limb ? 1 : 0, h->value_quad ? 1 : 0, h->lq_limb ? h->lq_split : 0, h->ric_fact ? 1 : 0, h->chain_fused ? 1 : 0};
    if (dst && bytes > 0) memcpy(dst, forms, (size_t)(bytes < 20 ? bytes : 20));
    return 20;
  }
  if (what == HSQP_BLK_PARAMS) {   // available as soon as a problem is resident
    if (!h->have_problem) { h->err = "no problem uploaded"; return HSQP_ERR_BAD_ARG; }
    if (hipSetDevice(h->device) != hipSuccess) return HSQP_ERR_HIP;
    const long long size = (long long)h->B * (h->N + 1) * NP * 8;
    if (dst && bytes > 0 && hipMemcpy(dst, h->d_par, (size_t)(bytes < size ? bytes : size), hipMemcpyDeviceToHost) != hipSuccess) return HSQP_ERR_HIP;
    return size;
  }
  if (!h->have_solution) { h->err = "no iteration has run"; return HSQP_ERR_BAD_ARG; }
  if (hipSetDevice(h->device) != hipSuccess) return HSQP_ERR_HIP;
  const size_t B = h->B, N = h->N, nodes = B * N;
  std::vector<double> out;
  std::vector<int> iout;
  auto fetch_rec = [&](std::vector<double>& rec) {
    rec.resize(nodes * (size_t)REC_SIZE);
    return hipMemcpy(rec.data(), h->d_rec, rec.size() * 8, hipMemcpyDeviceToHost) == hipSuccess;
  };
  std::vector<double> rec;
  switch (what) {
    case HSQP_BLK_AB: {
      if (!fetch_rec(rec)) return HSQP_ERR_HIP;
      out.resize(nodes * NX * NZ);
      for (size_t n = 0; n < nodes; ++n) {
        if (h->hdm.formulation == HSQP_FORM_CENTROIDAL) cent_expand_AB(&rec[n * REC_SIZE], h->h_dt[n], &out[n * NX * NZ]);
        else {
          if (h->lq_limb && h->chain_fused) {   // REC_PV is not written on this handle (k_project chains the columns itself): the same chain here, from the stage Jacobians
            double* r = &rec[n * REC_SIZE];
            double blk[3][2][6][6];
            for (int i = 0; i < 3 * 72; ++i) blk[i / 72][(i / 36) % 2][(i / 6) % 6][i % 6] = r[REC_GS + lq_chain_blk_offset(i / 72, (i / 36) % 2, (i / 6) % 6, i % 6)];
            for (int col = 0; col < LDJ; ++col) lq_chain_column<true>(blk, r + REC_GS, col, h->h_dt[n], r);
          }
          expand_AB(&rec[n * REC_SIZE], h->h_dt[n], &out[n * NX * NZ]);
        }
      }
      break;
    }
    case HSQP_BLK_BVEC: case HSQP_BLK_FLOW: {
      if (!fetch_rec(rec)) return HSQP_ERR_HIP;
      out.resize(nodes * NX);
      const int off = what == HSQP_BLK_BVEC ? REC_B : REC_FLOW;
      for (size_t n = 0; n < nodes; ++n) memcpy(&out[n * NX], &rec[n * REC_SIZE + off], NX * 8);
      break;
    }
    case HSQP_BLK_H: case HSQP_BLK_G: {
      if (!fetch_rec(rec)) return HSQP_ERR_HIP;
      const bool isH = what == HSQP_BLK_H;
      out.assign(nodes * (isH ? NZ * NZ : NZ), 0.0);
      for (size_t n = 0; n < nodes; ++n) {
        const double* r = &rec[n * REC_SIZE];
        const int nrows = (int)r[REC_NROWS];
        for (int a = 0; a < NZ; ++a) {
          if (isH) {
            for (int b = 0; b < NZ; ++b) {
              double s = a == b ? r[REC_D + a] : 0.0;
              for (int k = 0; k < nrows; ++k) s += rec_J_at(r, k, a) * rec_J_at(r, k, b);
              out[n * NZ * NZ + a * NZ + b] = s;
            }
          } else {
            double s = r[REC_GD + a];
            for (int k = 0; k < nrows; ++k) s += rec_J_at(r, k, a) * r[REC_RHO + k];
            out[n * NZ + a] = s;
          }
        }
      }
      break;
    }
    case HSQP_BLK_CDE: {
      if (!fetch_rec(rec)) return HSQP_ERR_HIP;
      out.resize(nodes * NE_MAX * (NZ + 1));
      for (size_t n = 0; n < nodes; ++n)
        for (int r = 0; r < NE_MAX; ++r) for (int c = 0; c <= NZ; ++c) out[(n * NE_MAX + r) * (NZ + 1) + c] = rec_CDe_at(&rec[n * REC_SIZE], r, c);
      break;
    }
    case HSQP_BLK_NE: {
      if (!fetch_rec(rec)) return HSQP_ERR_HIP;
      iout.resize(nodes);
      for (size_t n = 0; n < nodes; ++n) iout[n] = (int)rec[n * REC_SIZE + REC_MISC];
      break;
    }
    case HSQP_BLK_COST: {
      if (!fetch_rec(rec)) return HSQP_ERR_HIP;
      std::vector<double> x(B * (N + 1) * NX), par(B * (N + 1) * NP);
      if (hipMemcpy(x.data(), h->d_x, x.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return HSQP_ERR_HIP;
      if (hipMemcpy(par.data(), h->d_par, par.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return HSQP_ERR_HIP;
      out.resize(B * (N + 1));
      for (size_t b = 0; b < B; ++b) {
        for (size_t k = 0; k < N; ++k) out[b * (N + 1) + k] = rec[(b * N + k) * REC_SIZE + REC_MISC + 1];
        double c = 0.0;
        for (int i = 0; i < NX; ++i) { const double d = x[(b * (N + 1) + N) * NX + i] - par[(b * (N + 1) + N) * NP + HSQP_P_XDES + i]; c += 0.5 * h->hdm.Qf[i] * d * d; }
        out[b * (N + 1) + N] = c;
      }
      break;
    }
    case HSQP_BLK_DX: out.resize(B * (N + 1) * NX); if (hipMemcpy(out.data(), h->d_dx, out.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return HSQP_ERR_HIP; break;
    case HSQP_BLK_DU: out.resize(B * N * NU); if (hipMemcpy(out.data(), h->d_du, out.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return HSQP_ERR_HIP; break;
    case 100: {  // phase-profile ticks (only meaningful in -DHSQP_PHASE_PROFILE builds)
      std::vector<long long> t(4 * 128);
      if (hipMemcpy(t.data(), h->d_prof, t.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return HSQP_ERR_HIP;
      if (dst && bytes > 0) memcpy(dst, t.data(), (size_t)(bytes < (long long)t.size() * 8 ? bytes : (long long)t.size() * 8));
      return (long long)t.size() * 8;
    }
    case 101: out.resize(nodes * (size_t)RIC_SIZE); if (hipMemcpy(out.data(), h->d_ric, out.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return HSQP_ERR_HIP; break;   // raw gains record (debug tools)
    case 102: out.resize(nodes * (size_t)QP_SIZE); if (hipMemcpy(out.data(), h->d_qp, out.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return HSQP_ERR_HIP; break;    // raw QP record (debug tools)
    default: h->err = "unknown block id"; return HSQP_ERR_BAD_ARG;
  }
  const long long size = iout.empty() ? (long long)out.size() * 8 : (long long)iout.size() * 4;
  const void* src = iout.empty() ? (const void*)out.data() : (const void*)iout.data();
  if (dst && bytes > 0) memcpy(dst, src, (size_t)(bytes < size ? bytes : size));
  return size;
}

}  // extern "C"
