// Blocked factorisation of the Riccati stage (SURVEY.md A.4:  Lam = R~ + B~^T S+ B~ = L L^T,  Z = L^-1 G,  z = L^-1 g,  L^-1) on the
// FP64 matrix cores — the CPU analogue in the reference is HPIPM's blocked Riccati (task.info:79-93 selects it).
//
// Round 3 eliminated [Lam | I | G | g] column by column: 22 dependent steps (pivot -> reciprocal -> multipliers -> 22 row updates through
// v_readlane), ~540 cycles each, 12 k cycles — the longest phase of the stage.  Here the augmented matrix lives in the ACCUMULATOR layout of
// v_mfma_f64_16x16x4 (tile = 16 x 16, register r of a lane = rows 4r .. 4r + 3, lane = (row within the group << 4) | column) and is eliminated
// in SIX panels of four rows:
//   1. the 4 x 4 pivot block P (ten v_readlane pairs), its LDL^T factors and the unit-lower inverse M~ = L~^-1 (every lane, redundantly:
//      three reciprocals on the dependent chain instead of twelve);
//   2. Y = M~ U for the panel's four rows of every column tile: ONE matrix instruction per tile (A operand = M~ in rows 0 .. 3, B operand =
//      the tile's register r as it stands — register r of an accumulator tile IS a 4 x 16 B operand);
//   3. trailing update  E_i <- E_i - (Lam_i,panel P^-1) U = E_i - (D^-1 Y_Lam)_i^T Y: ONE matrix instruction per tile, a rank-4 update; the A
//      operand is register r of the panel's Lam tile scaled by -1/d_k (the trailing matrix stays symmetric, so the column slab a row tile
//      needs is the transposed row slab: register r of an accumulator tile is also a 16 x 4 A operand), rows that are final masked to zero.
// Rows leave unscaled (as in round 3) and are multiplied with d^-1/2 on the way out, which turns them into rows of L^-1 / Z / z.
// Two waves as before, each carrying the Lam tiles (the multipliers of a panel come from them, so the waves never talk to each other):
// wave 0 also [I | G columns 0 .. 15], wave 1 the other columns of G and g (g rides as column NXE of G).
//
// The code is written once over a wave abstraction W (device: a lane's own value; host: 64 lanes in an array, tests/hostemu), so the index
// logic and the arithmetic are checked in the GPU-less container against the column-by-column form (tests/hostemu).
#pragma once
#include "hsqp_linalg.h"

namespace hsqp {

#if defined(__HIP_DEVICE_COMPILE__)
struct DevWave {
  using V = double;
  using V4 = hsqp_d4;
  int lane;
  template <class F> HSQP_D V lanes(F f) const { return f(lane); }
  template <class M> HSQP_D V select(M m, V a, V b) const { return m(lane) ? a : b; }
  HSQP_D double readlane(V v, int l) const { return readlane_f64(v, l); }
  HSQP_D V4 mfma(V a, V b, V4 c) const { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  HSQP_D V4 zero4() const { return hsqp_d4{0.0, 0.0, 0.0, 0.0}; }
  static HSQP_D V get(const V4& t, int r) { return t[r]; }
  static HSQP_D void set(V4& t, int r, V v) { t[r] = v; }
  template <class F> HSQP_D void each(F f) const { f(lane); }
  static HSQP_D double at(V v, int) { return v; }
  static HSQP_D void pin(double& x) { asm volatile("" : "+v"(x)); }
};
#else
// host emulation of one wave (test infrastructure): a value is 64 lanes wide
struct HostV {
  double v[64];
  HostV operator*(const HostV& o) const { HostV r; for (int l = 0; l < 64; ++l) r.v[l] = v[l] * o.v[l]; return r; }
  HostV operator-() const { HostV r; for (int l = 0; l < 64; ++l) r.v[l] = -v[l]; return r; }
};
inline HostV operator*(double a, const HostV& o) { HostV r; for (int l = 0; l < 64; ++l) r.v[l] = a * o.v[l]; return r; }
struct HostV4 { HostV r[4]; };
struct HostWave {
  using V = HostV;
  using V4 = HostV4;
  template <class F> V lanes(F f) const { V r; for (int l = 0; l < 64; ++l) r.v[l] = f(l); return r; }
  template <class M> V select(M m, const V& a, const V& b) const { V r; for (int l = 0; l < 64; ++l) r.v[l] = m(l) ? a.v[l] : b.v[l]; return r; }
  double readlane(const V& v, int l) const { return v.v[l]; }
  // D[i][j] = C[i][j] + sum_k A[i][k] B[k][j];  A[i][k] in lane (k << 4) | i, B[k][j] in lane (k << 4) | j, D[(lane >> 4) + 4 r][lane & 15] in register r
  V4 mfma(const V& a, const V& b, const V4& c) const {
    V4 d = c;
    for (int r = 0; r < 4; ++r)
      for (int l = 0; l < 64; ++l) {
        const int i = (l >> 4) + 4 * r, j = l & 15;
        double s = c.r[r].v[l];
        for (int k = 0; k < 4; ++k) s += a.v[(k << 4) | i] * b.v[(k << 4) | j];
        d.r[r].v[l] = s;
      }
    return d;
  }
  V4 zero4() const { V4 z; for (int r = 0; r < 4; ++r) for (int l = 0; l < 64; ++l) z.r[r].v[l] = 0.0; return z; }
  static const V& get(const V4& t, int r) { return t.r[r]; }
  static void set(V4& t, int r, const V& v) { t.r[r] = v; }
  template <class F> void each(F f) const { for (int l = 0; l < 64; ++l) f(l); }
  static double at(const V& v, int l) { return v.v[l]; }
  static void pin(double&) {}
};
#endif

// ---- column tiles of the augmented matrix [Lam (23 -> 32) | I (23 -> 32) | G, g (NXE + 1 -> 16 NGT)]
constexpr int EB_L = 0, EB_I = 1, EB_G = 2;
constexpr int EB_NPANEL = 6;                       // six panels of four rows: 23 rows + one identity padding row
template <int NXE> constexpr int eb_ngt() { return (NXE + 1 + 15) / 16; }
constexpr int EB_G0 = 1;                           // G column tiles [0, EB_G0) in wave 0, the others in wave 1
template <int NXE, int WV> constexpr int eb_nct() { return WV == 0 ? 4 + EB_G0 : 2 + eb_ngt<NXE>() - EB_G0; }
template <int WV> constexpr int eb_kind(int c) { return c < 2 ? EB_L : (WV == 0 ? (c < 4 ? EB_I : EB_G) : EB_G); }
template <int WV> constexpr int eb_idx(int c) { return c < 2 ? c : (WV == 0 ? (c < 4 ? c - 2 : c - 4) : EB_G0 + c - 2); }
// which tiles hold anything: Lam is used through its upper triangle (tile (1, 0) never), L^-1 is lower triangular (tile (0, 1) is zero)
constexpr bool eb_exists(int kind, int idx, int rt) { return kind == EB_L ? idx >= rt : (kind == EB_I ? idx <= rt : true); }

// Workspace accessors (row-major LDS arrays of the Riccati stage): Lam[r][c] = lam[r * ldl + c], G[r][c] = gm[r * ldg + c], g[r] = gv[r * ldg];
// results: L^-1[r][c] -> linv[r * ldl + c] and (L^-1)^T -> linvT[c * ldt + r], Z[r][c] -> z[r * ldz + c] with z[r] in column NXE and in zv[r]
struct ElimIO {
  const double* lam; int ldl;
  const double* gm; const double* gv; int ldg;
  double* linv; int ldli;
  double* linvT; int ldt;
  double* z; int ldz;
  double* zv;
  int* ok;
};

struct ElimNoHook { HSQP_HD void operator()() const {} };
// after_load(): called once the wave has read its part of the workspace (the device kernel starts its asynchronous copies of the next
// stage there: a wave waits for such copies before its next LDS access, and this wave has none until the write-out)
template <int NXE, int WV, class W, class Hook = ElimNoHook>
HSQP_HD void eliminate_blocked(const W& wv, const ElimIO& io, Hook after_load = Hook()) {
  using V = typename W::V;
  using V4 = typename W::V4;
  constexpr int NC = eb_nct<NXE, WV>();
  V4 T[2][NC];
  // ---- load: tile (rt, c), register r, lane (k, cc) = E[16 rt + 4 r + k][16 idx + cc]
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int kind = eb_kind<WV>(c), idx = eb_idx<WV>(c);
      if (!eb_exists(kind, idx, rt)) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        W::set(T[rt][c], r, wv.lanes([&](int lane) -> double {
          const int row = 16 * rt + 4 * r + (lane >> 4), col = 16 * idx + (lane & 15);
          const int rc = row < NUT ? row : NUT - 1;
          if (kind == EB_L) {
            // (upper triangle of the stage's Lam, mirrored: the tile job that formed it rounds the two triangles differently)
            const int cc = col < NUT ? col : NUT - 1;
            const double v = io.lam[(rc < cc ? rc : cc) * io.ldl + (rc < cc ? cc : rc)];
            return (row < NUT && col < NUT) ? v : (row == col ? 1.0 : 0.0);
          } else if (kind == EB_I) {
            return row == col ? 1.0 : 0.0;
          } else {
            const double v = col < NXE ? io.gm[rc * io.ldg + col] : io.gv[rc * io.ldg];
            return (row < NUT && col <= NXE) ? v : 0.0;
          }
        }));
      }
    }
  after_load();
  V rsl[EB_NPANEL];      // per panel: d_k^-1/2 of the lane's row group
  bool bad = false;
  const V zero = wv.lanes([](int) { return 0.0; });
#pragma unroll
  for (int p = 0; p < EB_NPANEL; ++p) {
    const int rtp = p >> 2, pr = p & 3;
    // ---- 1. pivot block (lower triangle), its LDL^T factors, the unit-lower inverse.  Every lane computes all of it (the values are
    //      wave-uniform); W::pin keeps the compiler from sinking the values a few lanes select below into lane-divergent branches
    const V& lt = W::get(T[rtp][rtp], pr);            // column tile rtp of Lam holds the columns 16 rtp ..
    auto P = [&](int k, int k2) { return wv.readlane(lt, (k2 << 4) | (4 * pr + k)); };   // = P[k][k2] by symmetry: row k2 of the panel, column k
    const double p00 = P(0, 0), p10 = P(1, 0), p20 = P(2, 0), p30 = P(3, 0), p11 = P(1, 1), p21 = P(2, 1), p31 = P(3, 1), p22 = P(2, 2), p32 = P(3, 2),
                 p33 = P(3, 3);
    const double d0 = p00, rd0 = fast_rcp(d0);
    const double l10 = p10 * rd0, l20 = p20 * rd0, l30 = p30 * rd0;
    const double d1 = p11 - l10 * p10, rd1 = fast_rcp(d1);
    const double t21 = p21 - l20 * p10, t31 = p31 - l30 * p10;
    const double l21 = t21 * rd1, l31 = t31 * rd1;
    const double d2 = (p22 - l20 * p20) - l21 * t21, rd2 = fast_rcp(d2);
    const double t32 = (p32 - l30 * p20) - l31 * t21;
    const double l32 = t32 * rd2;
    double d3 = ((p33 - l30 * p30) - l31 * t31) - l32 * t32;
    double m10 = -l10, m21 = -l21, m32 = -l32;
    double m20 = l21 * l10 - l20, m31 = l32 * l21 - l31;
    double m30 = (l31 * l10 - l30) - l32 * m20;
    W::pin(m10); W::pin(m20); W::pin(m21); W::pin(m30); W::pin(m31); W::pin(m32); W::pin(d3);
    // A operand of  rows(panel) += (M~ - I) U : lane (kq << 4) | (4 pr + i) holds M~[i][kq] for kq < i (strictly lower), zero elsewhere — the
    // matrix instruction then updates the panel's four rows of a tile IN PLACE (no copy of its result into the tile's registers)
    const V mop = wv.lanes([&](int lane) -> double {
      const int i = (lane & 15) - 4 * pr, kq = lane >> 4;
      double v = 0.0;
      v = (kq == 0 && i == 1) ? m10 : v;
      v = (kq == 0 && i == 2) ? m20 : v;
      v = (kq == 1 && i == 2) ? m21 : v;
      v = (kq == 1 && i == 3) ? m31 : v;
      v = (kq == 2 && i == 3) ? m32 : v;
      v = (kq == 0 && i == 3) ? m30 : v;
      return v;
    });
    // a pivot that is not positive: reported (the caller's `ok`), scaled with 1 (as the column-by-column form)
    const bool real3 = 4 * p + 3 < NUT;
    bad = bad || !(d0 > 0.0) || !(d1 > 0.0) || !(d2 > 0.0) || (real3 && !(d3 > 0.0));
    // the lane's own pivot (row group kq): its reciprocal scales the A operand of the trailing update, its inverse root the row on the way out
    // (two levels of two-way selects on the bits of kq: a four-way select by equality becomes a table of stack addresses)
    const V dsel = wv.lanes([=](int lane) -> double { const double lo = (lane & 16) ? d1 : d0, hi = (lane & 16) ? d3 : d2; return (lane & 32) ? hi : lo; });
    const V nrd = wv.lanes([&](int lane) -> double { return -fast_rcp(W::at(dsel, lane)); });
    rsl[p] = wv.lanes([&](int lane) -> double { const double d = W::at(dsel, lane); return inv_sqrt(d > 0.0 ? d : 1.0); });
    // ---- 2. Y = M~ U on the panel's rows of every column tile (the Lam tiles first: the next panel's pivot block waits for them)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (!eb_exists(eb_kind<WV>(c), eb_idx<WV>(c), rtp)) continue;
      T[rtp][c] = wv.mfma(mop, W::get(T[rtp][c], pr), T[rtp][c]);
    }
    // ---- 3. trailing update of the rows below the panel
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      if (rt < rtp || 4 * p + 3 >= (rt == 0 ? 15 : NUT - 1)) continue;       // no row of this row tile is left
      // A operand: lane (k, i) = -1/d_k Y[k][16 rt + i], zero for the rows that are final (<= 4 p + 3)
      const V sc = nrd * W::get(T[rtp][rt], pr);
      const V aop = wv.select([&](int lane) { return 16 * rt + (lane & 15) > 4 * p + 3; }, sc, zero);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int kind = eb_kind<WV>(c), idx = eb_idx<WV>(c);
        if (!eb_exists(kind, idx, rt) || !eb_exists(kind, idx, rtp)) continue;
        T[rt][c] = wv.mfma(aop, W::get(T[rtp][c], pr), T[rt][c]);
      }
    }
  }
  if (bad) wv.each([&](int lane) { if (WV == 0 && lane == 0) *io.ok = 0; });
  // ---- write-out: every row scaled with d^-1/2
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int kind = eb_kind<WV>(c), idx = eb_idx<WV>(c);
      if (kind == EB_L) continue;
      const bool live = eb_exists(kind, idx, rt);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (4 * rt + r >= EB_NPANEL) continue;      // rows 24 .. 31: padding
        const V out = live ? W::get(T[rt][c], r) * rsl[4 * rt + r] : zero;
        wv.each([&](int lane) {
          const int row = 16 * rt + 4 * r + (lane >> 4), col = 16 * idx + (lane & 15);
          if (row >= NUT) return;
          const double v = W::at(out, lane);
          if (kind == EB_I) {
            if (col < NUT) {
              const double vv = col <= row ? v : 0.0;
              io.linv[row * io.ldli + col] = vv;
              if (io.linvT) io.linvT[col * io.ldt + row] = vv;   // (optional: the factored stage, hsqp_riccati_fact.h, has no use for the transpose)
            }
          } else if (col < NXE) {
            io.z[row * io.ldz + col] = v;
          } else if (col == NXE) {
            io.z[row * io.ldz + NXE] = v;
            io.zv[row] = v;
          }
        });
      }
    }
}

}  // namespace hsqp
