// Round 3's factorisation of the Riccati stage — [Lam | I | G | g] eliminated column by column in registers, a lane per column, 22 dependent steps
// through v_readlane — kept for the two probes of this directory (elim_probe.hip, elim_blocked_probe.hip), which time it against the blocked
// matrix-core form the kernels run (wb_humanoid_mpc_amd/csrc/hsqp_elim.h).  Not part of the library.
#pragma once
#include "../../wb_humanoid_mpc_amd/csrc/hsqp_riccati.h"

namespace hsqp {
#if defined(__HIP_DEVICE_COMPILE__)
constexpr int ELIM_G0 = 64 - 2 * NUT;          // columns of G in wave 0 (18)
constexpr int ELIM_SPLIT = 8;                  // steps before the mid-phase barrier (58 % of the row updates: S A~ finishes about then)
struct ElimLane { bool isLam, isI, isG, isg; int gcol; };
template <int NXE>
HSQP_D ElimLane elim_lane(int wave, int lane) {
  static_assert(NXE - ELIM_G0 <= 64 - NUT - 1, "the remaining columns of G and g fit the second wave");
  ElimLane l;
  l.isLam = lane < NUT;
  l.isI = wave == 0 && lane >= NUT && lane < 2 * NUT;
  l.gcol = wave == 0 ? lane - 2 * NUT : ELIM_G0 + lane - NUT;
  l.isG = !l.isLam && !l.isI && (wave == 0 ? true : (l.gcol < NXE));
  l.isg = wave == 1 && lane == 63;
  return l;
}
// first part: the lane's column -> registers, steps [0, ELIM_SPLIT); second part: the remaining steps, then every row i scaled with d_i^-1/2 on its way to LDS
template <int NXE>
HSQP_D void eliminate_begin(const RicWS& w, int wave, int lane, double (&e)[NUT], const double* a_next, double* a_dst) {
  const ElimLane l = elim_lane<NXE>(wave, lane);
  const double* src = l.isLam ? &w.fac.Ef[0][lane] : (l.isg ? &w.Em[0][EM_GVP] : (l.isG ? &w.Em[0][EM_G + l.gcol] : &w.fac.Ef[0][0]));
  const int stride = (l.isLam || !(l.isG || l.isg)) ? LDF : LDE;
#pragma unroll
  for (int i = 0; i < NUT; ++i) {
    const double v = src[i * stride];
    e[i] = l.isI ? (i == lane - NUT ? 1.0 : 0.0) : v;
  }
  next_a_to_lds(a_next, a_dst, wave, lane);
#pragma unroll
  for (int j = 0; j < ELIM_SPLIT; ++j) {
    const double fv = e[j] * fast_rcp(readlane_f64(e[j], j));   // lane i: the multiplier of row i (one multiply per step, not per row)
#pragma unroll
    for (int i = j + 1; i < NUT; ++i) e[i] -= readlane_f64(fv, i) * e[j];
  }
}
// second part: the remaining steps, then every row i scaled with d_i^-1/2 on its way to LDS
template <int NXE>
HSQP_D void eliminate_end(RicWS& w, int wave, int lane, double (&e)[NUT]) {
  const ElimLane l = elim_lane<NXE>(wave, lane);
#pragma unroll
  for (int j = ELIM_SPLIT; j < NUT - 1; ++j) {
    const double fv = e[j] * fast_rcp(readlane_f64(e[j], j));
#pragma unroll
    for (int i = j + 1; i < NUT; ++i) e[i] -= readlane_f64(fv, i) * e[j];
  }
  // pivots: d_i sits in lane i (row i of column i is final after step i - 1)
  double dv = 1.0;
#pragma unroll
  for (int i = 0; i < NUT; ++i) dv = lane == i ? e[i] : dv;
  const bool bad = l.isLam && !(dv > 0.0);
  if (bad) dv = 1.0;
  if (__builtin_amdgcn_ballot_w64(bad) != 0 && wave == 0 && lane == 0) w.ok = 0;
  const double rs = inv_sqrt(dv);
  const int ic = lane - NUT;
#pragma unroll
  for (int i = 0; i < NUT; ++i) {
    const double v = e[i] * readlane_f64(rs, i);
    if (l.isI) {
      const double vv = ic <= i ? v : 0.0;
      w.fac.Ef[i][EF_MI + ic] = vv;
      w.fac.LinvT[ic][i] = vv;
    } else if (l.isg) {
      w.zv[i] = v;
      w.Zs[i][NXE] = v;
    } else if (l.isG) {
      w.Zs[i][l.gcol] = v;
    }
  }
}
#endif
}  // namespace hsqp
