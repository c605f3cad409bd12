// Workgroup-cooperative dense kernels on LDS-resident matrices.
//
// Every product of the projection and Riccati kernels is brought to the form  C = X^T Y  with X (L x M)
// and Y (L x N) row-major, i.e. the contraction runs over ROWS of both operands (S and Lam^-1 are symmetric,
// residual rows are stored row-wise, B/A/G are used transposed).  Then both operands of a register tile are
// contiguous in memory:   acc[i][j] += X[l][r0+i] * Y[l][c0+j].
// VALU implementation: one work item = one TM x TN output tile with TM*TN independent accumulators
// (latency hiding without extra waves; 2 LDS reads per TM*TN/(TM+TN) FMAs).  Host-testable.
#pragma once
#include <utility>
#include "hsqp_common.h"

#if !defined(__HIP_DEVICE_COMPILE__)
#include <vector>
#endif
namespace hsqp {

// Generic tile loop:  C = X1^T Y1 + sign2 * X2^T Y2  (second product optional: L2 = 0; sign2 = +1 or -1).
// A work item owns the STRIDED tile  rows {tr + i*tm}, cols {tc + j*tn}  (tm = ceil(M/TM), tn = ceil(N/TN)):
// consecutive lanes then read consecutive LDS words of a row of X / Y (conflict-free ds_read_b64) instead of
// words 8*TN bytes apart.  store(r, c, value) is called for every in-range element of the tile.
// One tile (index t of ceil(M/TM) * ceil(N/TN)) of  C = X1^T Y1 + sign2 * X2^T Y2.
template <int TM, int TN, class Store>
HSQP_HD void xty_tile2(int t, int M, int N, int L1, const double* X1, int ldx1, const double* Y1, int ldy1, int L2, const double* X2,
                       int ldx2, const double* Y2, int ldy2, double sign2, Store store) {
  const int tm = (M + TM - 1) / TM, tn = (N + TN - 1) / TN;
  const int tr = t / tn, tc = t % tn;
  int xo[TM], yo[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) xo[i] = (tr + i * tm < M) ? tr + i * tm : M - 1;   // clamp: duplicates are computed, never stored
#pragma unroll
  for (int j = 0; j < TN; ++j) yo[j] = (tc + j * tn < N) ? tc + j * tn : N - 1;
  double acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = 0.0;
  for (int seg = 0; seg < 2; ++seg) {
    const int L = seg ? L2 : L1;
    const double* X = seg ? X2 : X1;
    const double* Y = seg ? Y2 : Y1;
    const int ldx = seg ? ldx2 : ldx1, ldy = seg ? ldy2 : ldy1;
#pragma unroll 2
    for (int l = 0; l < L; ++l) {
      double a[TM], b[TN];
      const double* xr = X + l * ldx;
      const double* yr = Y + l * ldy;
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = seg ? sign2 * xr[xo[i]] : xr[xo[i]];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = yr[yo[j]];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] += a[i] * b[j];
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
      if (tr + i * tm < M && tc + j * tn < N) store(tr + i * tm, tc + j * tn, acc[i][j]);
}
template <int TM, int TN, class Store>
HSQP_HD void xty_tile(int t, int M, int N, int L, const double* X, int ldx, const double* Y, int ldy, Store store) {
  xty_tile2<TM, TN>(t, M, N, L, X, ldx, Y, ldy, 0, X, ldx, Y, ldy, 1.0, store);
}
constexpr int ntiles(int M, int N, int TM, int TN) { return ((M + TM - 1) / TM) * ((N + TN - 1) / TN); }

// NOTE: independent products of one phase must share ONE WG_FOR item space (item ranges -> jobs); two consecutive
// wg_xty calls are executed one after the other by every thread.
template <int TM, int TN, class Store>
HSQP_HD void wg_xty2(const Ctx& ctx, int M, int N, int L1, const double* X1, int ldx1, const double* Y1, int ldy1, int L2,
                     const double* X2, int ldx2, const double* Y2, int ldy2, double sign2, Store store) {
  WG_FOR(ctx, t, ntiles(M, N, TM, TN)) xty_tile2<TM, TN>(t, M, N, L1, X1, ldx1, Y1, ldy1, L2, X2, ldx2, Y2, ldy2, sign2, store);
}
template <int TM, int TN, class Store>
HSQP_HD void wg_xty(const Ctx& ctx, int M, int N, int L, const double* X, int ldx, const double* Y, int ldy, Store store) {
  wg_xty2<TM, TN>(ctx, M, N, L, X, ldx, Y, ldy, 0, X, ldx, Y, ldy, 1.0, store);
}

// sum_l X[l*ldx] * v[l], l < L, with the loads issued ahead of the dependent FMA chain
template <int L>
HSQP_HD double dot_strided(const double* X, int ldx, const double* v) {
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int l = 0; l + 1 < L; l += 2) { s0 += X[l * ldx] * v[l]; s1 += X[(l + 1) * ldx] * v[l + 1]; }
  if (L & 1) s0 += X[(L - 1) * ldx] * v[L - 1];
  return s0 + s1;
}

// Partial products of y = A x for a row-major matrix (global or LDS), 4 work items per row; the caller adds the
// four partials in a second phase (deterministic reduction).  COLS is a compile-time constant so that the
// loads are issued ahead of the dependent FMA chain.
template <int COLS>
HSQP_HD double matvec_part(const double* Arow, const double* x, int p) {
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < (COLS + 3) / 4; ++c) {
    const int cc = p + 4 * c;
    if (cc < COLS) s += Arow[cc] * x[cc];
  }
  return s;
}

// Batched global -> LDS copy: work item b of nb = ceil(count / NBATCH) moves elements {b + j*nb}; the NBATCH loads are
// issued before the first store so that their latencies overlap.
template <int NBATCH, class Dst>
HSQP_HD void copy_batch(int b, int count, const double* src, Dst dst) {
  const int nb = (count + NBATCH - 1) / NBATCH;
  double t[NBATCH];
#pragma unroll
  for (int j = 0; j < NBATCH; ++j) { const int idx = b + j * nb; t[j] = idx < count ? src[idx] : 0.0; }
#pragma unroll
  for (int j = 0; j < NBATCH; ++j) { const int idx = b + j * nb; if (idx < count) dst(idx, t[j]); }
}
// The two halves of copy_batch, for items that do other work between issuing the loads and storing the values.
template <int NBATCH>
HSQP_HD void load_batch(int b, int count, const double* src, double* t) {
  const int nb = (count + NBATCH - 1) / NBATCH;
#pragma unroll
  for (int j = 0; j < NBATCH; ++j) { const int idx = b + j * nb; t[j] = (b < nb && idx < count) ? src[idx] : 0.0; }
}
template <int NBATCH, class Dst>
HSQP_HD void store_batch(int b, int count, const double* t, Dst dst) {
  const int nb = (count + NBATCH - 1) / NBATCH;
#pragma unroll
  for (int j = 0; j < NBATCH; ++j) { const int idx = b + j * nb; if (b < nb && idx < count) dst(idx, t[j]); }
}
constexpr int nbatches(int count, int nbatch) { return (count + nbatch - 1) / nbatch; }


// ------------------------------------------------------------------------------------------------------------
// FP64 matrix-core path.  Measured on MI355X (tools/microbench/f64_rates.hip): v_mfma_f64_16x16x4_f64 issues every
// 64 cycles per SIMD = 32 flop/clk/SIMD already from ONE wave per SIMD, whereas v_fma_f64 issues every 8 cycles per
// wave and needs >= 2 waves per SIMD for the same rate.  The LDS-resident stage kernels run one 4-8 wave workgroup
// per CU, so their dense products go through MFMA.  Operand fragments of  C = X^T Y  are plain row reads:
//   A[i][k] = X[k0 + (lane >> 4)][r0 + (lane & 15)],  B[k][j] = Y[k0 + (lane >> 4)][c0 + (lane & 15)],
//   D: lane holds rows (lane >> 4) + 4 reg, column lane & 15  (reg = 0..3).
// A job list is executed cooperatively: 16x16 output tiles are dealt round-robin to the waves of the workgroup.
constexpr int XTY_ADD_GLOBAL = 1, XTY_C_GLOBAL = 2, XTY_ROW_JUMP = 4, XTY_ADD_T = 8, XTY_ADD_LDS = 16;   // (XTY_ADD_LDS: the additive term of every job of the call is in LDS; XTY_ROW_JUMP: the jobs of the call use XtyJob::rsplit / rjump; XTY_ADD_T: the additive
// term of every job of the call is stored transposed, Add[c * ldadd + r] — XtyJob::addt on the host)
struct XtyJob {
  int M, N;                       // output size
  int L1; const double* X1; int ldx1; const double* Y1; int ldy1;
  int L2; const double* X2; int ldx2; const double* Y2; int ldy2;   // optional second product, added with sign2
  double sign2;
  double scale;                   // C = scale * acc + Add
  const double* Add; int ldadd;   // optional (may be global memory)
  int addt;                       // Add is stored transposed: element (r, c) at Add[c * ldadd + r] (device tile path: with XTY_ADD_T)
  double* C; int ldc;             // destination (LDS or global)
  int sx1, sx2;                   // element stride of X along the output-row index (1 = row-major X[l][r]; ld = 1, sx = ld' reads X'[r][l])
  int sym;                        // M == N and the product is symmetric: only tiles on/above the diagonal are computed, C is mirrored
  double* C2; int ldc2, nc2;      // optional second destination in GLOBAL memory for the columns < nc2 (device tile path; the host build copies it from C)
  int rsplit, rjump;              // rsplit > 0: output rows >= rsplit land rjump rows further down in C (two row blocks of one product written to
                                  // separate places: the dense rows of [A~ | B~]); device tile path only with XTY_ROW_JUMP
};

HSQP_HD XtyJob xty_job(int M, int N, int L, const double* X, int ldx, const double* Y, int ldy, double* C, int ldc,
                       const double* Add = nullptr, int ldadd = 0, double scale = 1.0) {
  XtyJob j;
  j.M = M; j.N = N; j.L1 = L; j.X1 = X; j.ldx1 = ldx; j.Y1 = Y; j.ldy1 = ldy;
  j.L2 = 0; j.X2 = X; j.ldx2 = ldx; j.Y2 = Y; j.ldy2 = ldy; j.sign2 = 1.0;
  j.scale = scale; j.Add = Add; j.ldadd = ldadd; j.C = C; j.ldc = ldc; j.sym = 0; j.sx1 = 1; j.sx2 = 1; j.C2 = nullptr; j.ldc2 = 0; j.nc2 = 0; j.rsplit = 0; j.rjump = 0; j.addt = 0;
  return j;
}

HSQP_HD XtyJob xty_sym(XtyJob j) { j.sym = 1; return j; }
HSQP_HD XtyJob xty_also_to(XtyJob j, double* C2, int ldc2, int nc2) { j.C2 = C2; j.ldc2 = ldc2; j.nc2 = nc2; return j; }

#if defined(HSQP_PHASE_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define XTY_PROF_T(name) const long long name = clock64()
#else
#define XTY_PROF_T(name) ((void)0)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// DPP quad permute of a double (two 32-bit moves): CTRL = the four 2-bit source selectors of a quad, e.g. 0xB1 = [1,0,3,2], 0x4E = [2,3,0,1]
template <int CTRL>
__device__ inline double quad_perm_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// a loop whose index is a compile-time constant in the body (DPP controls are immediates): f(std::integral_constant<int, 0>{}), ..., <N - 1>
template <class F, int... K>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, K...>) { (f(std::integral_constant<int, K>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
// DPP row broadcast of a double: every lane of a 16-lane row receives the value of lane K of ITS row (row_newbcast:K, gfx90a and later)
template <int K>
__device__ inline double row_bcast_f64(double v) {
  static_assert(K >= 0 && K < 16, "lane within the row");
  const long long b = __double_as_longlong(v);
  const int l0 = (int)(b & 0xffffffffll), h0 = (int)(b >> 32);   // (every lane is written: the source doubles as the "old" value, no zero to set up)
  const int lo = __builtin_amdgcn_update_dpp(l0, l0, 0x150 + K, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(h0, h0, 0x150 + K, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// a double of lane `lane` (compile-time constant after unrolling) as a wave-uniform value
__device__ inline double readlane_f64(double v, int lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
typedef double hsqp_d4 __attribute__((ext_vector_type(4)));
typedef const double __attribute__((address_space(1))) * hsqp_gcptr;
typedef double __attribute__((address_space(1))) * hsqp_gptr;
typedef const double __attribute__((address_space(3))) * hsqp_lcptr;
// NT = 1 or 2 output tiles of one job processed together: two independent accumulator chains keep the FP64 matrix
// pipe busy from a single wave (a dependent v_mfma_f64 chain alone leaves it half idle).
// SPACES (XTY_ADD_GLOBAL | XTY_C_GLOBAL): the additive term / the destination of every job of the call is known to be in
// GLOBAL memory -> address-space-qualified accesses.  A generic pointer compiles to FLAT instructions, whose loads also
// count on lgkmcnt and make the LDS operand waits of the MFMA loop wait for the L2/HBM round trip.
template <int NT, int SPACES, int XTY_PF = 1>
__attribute__((always_inline)) HSQP_D void xty_job_tiles_mfma(const XtyJob& j, const int* tiles, int lane, long long* prof = nullptr) {
  XTY_PROF_T(t_begin);
  const int tn = (j.N + 15) >> 4;
  const int i = lane & 15, kk = lane >> 4;
  int r0[NT], c0[NT], xr[NT], yc[NT];
  hsqp_d4 acc[NT];
  double addv[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    // tile row / column without an integer division (a runtime div or mod costs ~40 instructions; tile ids are < 40)
    int tr = 0, tc = tiles[t];
    while (tc >= tn) { tc -= tn; ++tr; }
    r0[t] = tr << 4; c0[t] = tc << 4;
    xr[t] = r0[t] + i < j.M ? r0[t] + i : j.M - 1;   // clamped: the duplicate rows / columns are never stored
    yc[t] = c0[t] + i < j.N ? c0[t] + i : j.N - 1;
    acc[t] = hsqp_d4{0.0, 0.0, 0.0, 0.0};
  }
  // the additive term may live in HBM: its loads are issued before the MFMA loop and consumed in the epilogue.  They are
  // UNCONDITIONAL (clamped indices, one uniform test of the pointer): a per-lane conditional load becomes an exec-masked
  // branch per element and the compiler drains vmcnt at every join, i.e. one L2 round trip after the other.
  if (j.Add) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int cc = c0[t] + i < j.N ? c0[t] + i : j.N - 1;
      const int rb = r0[t] + kk;
      const bool inside = r0[t] + 16 <= j.M;   // uniform
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rb + 4 * r, rc = (inside || row < j.M) ? row : j.M - 1;
        if constexpr ((SPACES & XTY_ADD_T) != 0) addv[t][r] = ((hsqp_gcptr)j.Add)[cc * j.ldadd + rc];
        else if (SPACES & XTY_ADD_GLOBAL) addv[t][r] = ((hsqp_gcptr)j.Add)[rc * j.ldadd + cc];
        else if constexpr ((SPACES & XTY_ADD_LDS) != 0) addv[t][r] = ((hsqp_lcptr)j.Add)[rc * j.ldadd + cc];
        else addv[t][r] = j.Add[rc * j.ldadd + cc];
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) addv[t][r] = 0.0;
  }
  XTY_PROF_T(t_loop);
  // The contraction as ONE sequence of 4-row steps (n1 of the first product, then n2 of the second) with the operands travelling XTY_PF
  // steps ahead in a ring of register sets (template parameter; 1 = the reads of step s + 1 are issued behind the matrix instructions of
  // step s, which is what a plain loop compiles to as well).  An LDS round trip is ~200 cycles under the load of eight waves, a matrix
  // instruction holds the pipe for 64: with one step in flight a one-tile call keeps its pipe a third busy — the serial Riccati stage,
  // whose phases are one or two tile calls long, asks for 3 (1.73 -> 1.62 ms); the throughput kernels (k_project: 3+ workgroups per CU
  // hide the latency with other waves, and the extra registers cost occupancy) stay at 1.  A step beyond the end of a product reads a
  // clamped (valid) row and contributes a * 0.
  const int n1 = (j.L1 + 3) >> 2, n2 = (j.L2 + 3) >> 2, ns = n1 + n2;
  auto fetch = [&](int s, double* a, double* b) {
    if (s < n1) {
      const int k = 4 * s + kk, kc = k < j.L1 ? k : j.L1 - 1;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const double av = j.X1[kc * j.ldx1 + xr[t] * j.sx1];
        a[t] = k < j.L1 ? av : 0.0;
        b[t] = j.Y1[kc * j.ldy1 + yc[t]];
      }
    } else {
      const int k = 4 * (s - n1) + kk, kc = k < j.L2 ? k : j.L2 - 1;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const double av = j.sign2 * j.X2[kc * j.ldx2 + xr[t] * j.sx2];
        a[t] = k < j.L2 ? av : 0.0;
        b[t] = j.Y2[kc * j.ldy2 + yc[t]];
      }
    }
  };
  double pa[XTY_PF][NT], pb[XTY_PF][NT];
#pragma unroll
  for (int u = 0; u < XTY_PF; ++u) { if (u < ns) fetch(u, pa[u], pb[u]); }
  __builtin_amdgcn_sched_barrier(0);
  for (int s0 = 0; s0 < ns; s0 += XTY_PF) {
#pragma unroll
    for (int u = 0; u < XTY_PF; ++u) {
      const int s = s0 + u;
      if (s < ns) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[u][t], pb[u][t], acc[t], 0, 0, 0);
        if (s + XTY_PF < ns) fetch(s + XTY_PF, pa[u], pb[u]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  XTY_PROF_T(t_epi);
  // epilogue: one address per tile (rows r0 + kk + 4 r are 4 ldc apart: constant offsets), a uniform fast path for tiles that
  // lie inside the matrix in the row direction (no per-element predicate), no arithmetic when there is nothing to add
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int c = c0[t] + i;
    if (c < j.N) {
      const int rb = r0[t] + kk;
      const bool mirror = j.sym && r0[t] != c0[t];
      // a diagonal tile of a symmetric job: only the elements on / above the diagonal are stored, each also at its mirrored place, so
      // the result is symmetric to the bit (the two triangles of the accumulator differ by rounding)
      const bool diag = j.sym && r0[t] == c0[t];
      const int o1 = rb * j.ldc + c, o2 = c * j.ldc + rb;
      double v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = j.Add ? j.scale * acc[t][r] + addv[t][r] : j.scale * acc[t][r];
      auto put = [&](int r) {
        if (diag && rb + 4 * r > c) return;
        if constexpr ((SPACES & XTY_ROW_JUMP) != 0) {
          const int row = rb + 4 * r + ((j.rsplit > 0 && rb + 4 * r >= j.rsplit) ? j.rjump : 0);
          if (SPACES & XTY_C_GLOBAL) ((hsqp_gptr)j.C)[row * j.ldc + c] = v[r];
          else j.C[row * j.ldc + c] = v[r];
          return;
        }
        if (SPACES & XTY_C_GLOBAL) {
          ((hsqp_gptr)j.C)[o1 + 4 * r * j.ldc] = v[r];
          if (mirror || diag) ((hsqp_gptr)j.C)[o2 + 4 * r] = v[r];
        } else {
          j.C[o1 + 4 * r * j.ldc] = v[r];
          if (mirror || diag) j.C[o2 + 4 * r] = v[r];
        }
        if (j.C2 && c < j.nc2) ((hsqp_gptr)j.C2)[(rb + 4 * r) * j.ldc2 + c] = v[r];
      };
      if (r0[t] + 16 <= j.M) {
#pragma unroll
        for (int r = 0; r < 4; ++r) put(r);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (rb + 4 * r < j.M) put(r);
      }
    }
  }
#if defined(HSQP_PHASE_PROFILE)
  if (prof && lane == 0) {   // wave 0 of workgroup 0: prologue / matrix loop / epilogue ticks, number of calls, matrix instructions
    const long long t_end = clock64();
    prof[90] += t_loop - t_begin; prof[91] += t_epi - t_loop; prof[92] += t_end - t_epi; prof[93] += 1; prof[94] += NT * (((j.L1 + 3) >> 2) + ((j.L2 + 3) >> 2));
  }
#endif
}
#endif

// symmetric jobs enumerate the tiles on/above the diagonal row by row
HSQP_HD int xty_tile_id(int sym, int tn, int t) {
  if (!sym) return t;
  int tr = 0;
  while (t >= tn - tr) { t -= tn - tr; ++tr; }
  return tr * tn + tr + t;
}

#if defined(__HIP_DEVICE_COMPILE__)
// tiles are numbered globally (base + t) and dealt round-robin to the waves; a wave takes its tiles two at a time
template <int SPACES, int PF = 1>
HSQP_D int xty_run_job(const XtyJob& j, int base, int wave, int nwaves, int lane, long long* prof = nullptr) {
  const int tm = (j.M + 15) >> 4, tn = (j.N + 15) >> 4;
  const int nt = j.sym ? tn * (tn + 1) / 2 : tm * tn;
  const int sym = j.sym;
  int t = (wave - base) & (nwaves - 1);   // round-robin over the waves (their number is a power of two)
  for (; t + nwaves < nt; t += 2 * nwaves) { const int pair[2] = {xty_tile_id(sym, tn, t), xty_tile_id(sym, tn, t + nwaves)}; xty_job_tiles_mfma<2, SPACES, PF>(j, pair, lane, prof); }
  if (t < nt) { const int one = xty_tile_id(sym, tn, t); xty_job_tiles_mfma<1, SPACES, PF>(j, &one, lane, prof); }
  return nt;
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// The tiles of `njobs` jobs dealt round-robin over W waves of ANY count (rank = this wave's position among them, < 0: the wave does not
// take part); a wave takes its tiles of a job two at a time.  Lets a phase spread its matrix work over all the waves that have
// nothing else on its critical path (k_riccati: the helper waves take tiles after their copies, seven waves form S A~ while the
// eighth eliminates).
// one job of a dealt list: g0 = number of tiles of the jobs before it; returns its own tile count
template <int SPACES = 0, int PF = 1>
__attribute__((always_inline)) HSQP_D int xty_deal_one(const XtyJob& j, int g0, int rank, int W, int lane, long long* prof = nullptr) {
  const int tm = (j.M + 15) >> 4, tn = (j.N + 15) >> 4;
  const int nt = j.sym ? tn * (tn + 1) / 2 : tm * tn;
  if (rank < 0) return nt;
  int t = rank - g0 % W;
  if (t < 0) t += W;
  for (; t + W < nt; t += 2 * W) { const int pair[2] = {xty_tile_id(j.sym, tn, t), xty_tile_id(j.sym, tn, t + W)}; xty_job_tiles_mfma<2, SPACES, PF>(j, pair, lane, prof); }
  if (t < nt) { const int one = xty_tile_id(j.sym, tn, t); xty_job_tiles_mfma<1, SPACES, PF>(j, &one, lane, prof); }
  return nt;
}
template <int SPACES = 0>
HSQP_D void xty_deal(const XtyJob* jobs, int njobs, int rank, int W, int lane) {
  if (rank < 0) return;
  int g0 = 0;
  for (int jn = 0; jn < njobs; ++jn) g0 += xty_deal_one<SPACES>(jobs[jn], g0, rank, W, lane);
}
#endif

// Executes `njobs` independent products; must be called by every thread of the workgroup (no barrier inside).
// UNROLL: the loop over the jobs is unrolled so that each job gets code specialised for its (constant) shape and the
// descriptors stay in registers — pays off for long contractions in throughput kernels, not inside the Riccati stage loop.
template <bool UNROLL = false, int SPACES = 0, int PF = 1>
HSQP_HD void wg_xty_jobs(const Ctx& ctx, const XtyJob* jobs, int njobs) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int wave = ctx.tid >> 6, nwaves = ctx.nthreads >> 6, lane = ctx.tid & 63;
  int base = 0;
#if defined(HSQP_PHASE_PROFILE)
  long long* prof = wave == 0 ? ctx.prof : nullptr;
#else
  long long* prof = nullptr;
#endif
  if constexpr (UNROLL) {
#pragma unroll
    for (int jn = 0; jn < njobs; ++jn) base += xty_run_job<SPACES, PF>(jobs[jn], base, wave, nwaves, lane, prof);
  } else {
    for (int jn = 0; jn < njobs; ++jn) base += xty_run_job<SPACES, PF>(jobs[jn], base, wave, nwaves, lane, prof);
  }
#else
  // host build (tests/hostemu, oracle/cpu_baseline.cpp): rank-1 updates over contiguous rows of Y so that the inner loop
  // vectorises; every element still accumulates its products in the order l = 0 .. L1-1, then L2 (the order of the naive
  // dot product), so the result does not depend on this loop structure
  static thread_local std::vector<double> accbuf;
  for (int jn = 0; jn < njobs; ++jn) {
    const XtyJob& j = jobs[jn];
    const int M = j.M, N = j.N;
    if (accbuf.size() < (size_t)M * N) accbuf.resize((size_t)M * N);
    double* acc = accbuf.data();
    for (int e = 0; e < M * N; ++e) acc[e] = 0.0;
    for (int part = 0; part < 2; ++part) {
      const int L = part ? j.L2 : j.L1, ldx = part ? j.ldx2 : j.ldx1, ldy = part ? j.ldy2 : j.ldy1, sx = part ? j.sx2 : j.sx1;
      const double* X = part ? j.X2 : j.X1;
      const double* Y = part ? j.Y2 : j.Y1;
      const double sg = part ? j.sign2 : 1.0;
      for (int l = 0; l < L; ++l) {
        const double* xr = X + (size_t)l * ldx;
        const double* yr = Y + (size_t)l * ldy;
        for (int r = 0; r < M; ++r) {
          const double a = sg * xr[r * sx];
          double* ar = acc + (size_t)r * N;
          for (int c = j.sym ? r : 0; c < N; ++c) ar[c] += a * yr[c];
        }
      }
    }
    for (int r = 0; r < M; ++r)
      for (int c = j.sym ? r : 0; c < N; ++c) {   // symmetric jobs: elements on / above the diagonal, mirrored
        double v = j.scale * acc[(size_t)r * N + c];
        if (j.Add) v += j.addt ? j.Add[c * j.ldadd + r] : j.Add[r * j.ldadd + c];
        j.C[(r + ((j.rsplit > 0 && r >= j.rsplit) ? j.rjump : 0)) * j.ldc + c] = v;
        if (j.sym && c != r) j.C[c * j.ldc + r] = v;
        if (j.C2 && c < j.nc2) j.C2[r * j.ldc2 + c] = v;
      }
  }
#endif
}

}  // namespace hsqp
