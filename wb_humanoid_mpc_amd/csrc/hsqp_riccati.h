// Riccati recursion of the projected, equality-free stage QP (the reference solves
// the same QP with HPIPM through ocs2's HpipmInterface — the minimiser is unique, A.1).
//
// Backward sweep, one workgroup (eight waves, FOUR matrix pipes) per MPC instance, stage matrices in LDS, four phases per stage:
//   Ph1  [SB | S b~] = S+ [B~ | b~]                                      (8 tiles over the eight waves; b~ rides as the 24th column of B~)
//   Ph2  G = P~ + SB^T A~,  Lam = R~ + B~^T SB,  sb = s+ + S+ b~,  g = r~ + B~^T sb                                    (12 tiles)
//   Ph3  Gauss-Jordan elimination of [Lam | I | G | g] on unscaled rows in registers, lane = column, by TWO waves (Lam duplicated in
//        both): with the Cholesky scaling D^-1/2 applied on the way out it leaves L^-1, Z = L^-1 G and z = L^-1 g at once.  The matrix
//        pipes are idle during it, so the other six waves form what the S-update needs but the factorisation does not:
//        SA = S+ A~ (16 tiles), one barrier in the middle of the elimination, W = Q~ + A~^T SA (10 symmetric tiles, into S)
//   Ph4  S = W - Z^T Z,  s = q~ + A~^T sb - Z^T z,  K = -L^-T Z,  k = -L^-T z          (= Q + A^T S A - G^T Lam^-1 G; 18 short tiles)
// so the serial critical path of a stage is 300 matrix instructions of 58-deep tiles, the elimination and 108 of 23-deep tiles; no
// triangular back-substitution and no product with L^-1 sits on it, and 390 of the stage's 846 matrix instructions run under the
// elimination (round 2's form had seven phases: SB | G, Lam | elimination of [Lam | I]
// under S A~ | scaling | Z = L^-1 G | K | S-update with both contractions).  The forward sweep is the mat-vec chain
//   dx+ = A~ dx + B~ (K dx + k) + b~   (A~ dx and K dx in one phase, then the 23-term B~ ut rows);
// the closed-loop matrix A~ + B~ K is never formed: it is not needed by the backward recursion, and forming it there put
// 96 matrix instructions and four tile calls per stage on the serial path.  The feed-forward/feedback inputs
//   ut = -U^-1 (Z dx + z),  du = Px dx + Pu ut + Pe
// are then recovered for all nodes in parallel (step_node).  Every product is a register-tiled
// X^T Y contraction (hsqp_linalg.h).
#pragma once
#include "hsqp_linalg.h"
#include "hsqp_elim.h"
#include "hsqp_project.h"

namespace hsqp {

constexpr int RIC_K = 0;                       // [23][58] feedback gain  K = -Lam^-1 G
constexpr int RIC_KV = RIC_K + NUT * NX;       // [23]     feed-forward   k = -Lam^-1 g
constexpr int RIC_SIZE = ((RIC_KV + NUT + 7) / 8) * 8;
constexpr int LDB = 24;                        // leading dimension of the 23-wide LDS matrices; column 23 of B carries b~ (and of SB: S b~)
static_assert(LDB == NUT + 1, "b~ is the 24th column of the B~ workspace");
constexpr int LDF = 48, EF_MI = LDB;           // elimination matrix [Lam (23) | 0 | I (23) | 0]
constexpr int EM_GVP = 0, EM_G = NUT, LDE = NUT + NX + 1;   // 82 columns; 0..3: g (host build: four partial sums, device: the sum in column 0)

constexpr int LDZ = NX + 2;                     // Z carries z as one more column: K = -L^-T [Z | z] then yields k in the same tile job
constexpr int RIC_HELPERS = 256;                // helper half of the 512-thread workgroup: item count of the fused helper passes
struct RicWS {
  double S[NX][NX];                            // value function; Ph2 overwrites it with W = Q~ + A~^T S A~ (S is dead after Ph1)
  struct {                                     // scratch of the factorisation
    double Ef[LDB][LDF];                       // [Lam | . | L^-1]  (host build: [Lam | 0 | I] -> ... -> columns 24.. scaled to L^-1)
    double LinvT[LDB][LDB];                    // (L^-1)^T
  } fac;
  double A2[2][NX][NX], SA[NX][NX];            // A2: double-buffered A~ (stage k uses A2[k & 1])
  double B[NX][LDB];                           // [B~ | b~]
  union {
    double SB[NX][LDB];                        // [S B~ | S b~]
    double Zs[NUT][LDZ];                       // [L^-1 G | z = L^-1 g in column NXE] (SB is dead once Lam, G and sb are formed)
  };
  double Em[NUT][LDE];                         // [g . | G -> K | .]
  double sv[NX], sb[NX], dx[NX], zv[LDB], kv[LDB];
  double part[NX * 4];                         // four partial sums per row: of the new s (backward sweep), of A~ dx (forward sweep)
  int ok;
};

static_assert(sizeof(RicWS) <= 163400, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) accepts 163 400 bytes on gfx950 and rejects 163 592");

// The matrix products of one phase of the sweep.  In the 512-thread kernels the waves [first_wave, first_wave + n_waves) take the tiles
// (dealt round-robin; a wave with copies / vector work in the same phase calls this AFTER that work); any other context (host build,
// smaller workgroups) runs them on its matrix half.
// SPACES = XTY_ADD_GLOBAL: the additive terms of the jobs are in global memory (the QP record) -> global instead of flat loads, which
// would also count on lgkmcnt and hold up the LDS operand waits of the matrix loop
// (the jobs are separate by-value arguments, not an array or pointers: a descriptor whose address is taken lives in scratch memory on the
//  device and its tile loops stay generic)
HSQP_HD XtyJob xty_no_job() { XtyJob j = xty_job(0, 0, 0, nullptr, 0, nullptr, 0, nullptr, 0); return j; }
constexpr int RIC_PF = 3;   // operand prefetch depth of the stage's tile loops (hsqp_linalg.h)
#ifndef PROF_WAVE
#define PROF_WAVE 2     // -DHSQP_PHASE_PROFILE builds: the wave whose tile calls are split into prologue / matrix loop / epilogue ticks
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// the tiles of one job over `W` waves chosen by the caller (rank < 0: this wave takes none)
template <int SPACES = 0>
HSQP_D void ric_products_ranked(const Ctx& ctx, int rank, int W, const XtyJob j0) {
  if (rank < 0) return;
#if defined(HSQP_PHASE_PROFILE)
  long long* prof = (ctx.tid >> 6) == PROF_WAVE ? ctx.prof : nullptr;
#else
  long long* prof = nullptr;
#endif
  xty_deal_one<SPACES, RIC_PF>(j0, 0, rank, W, ctx.tid & 63, prof);
}
#endif
template <int SPACES = 0, int NJ = 1>
HSQP_HD void ric_products(const Ctx& ctx, int first_wave, int n_waves, const XtyJob j0, const XtyJob j1 = xty_no_job(), const XtyJob j2 = xty_no_job()) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (ctx.nthreads == 512 && n_waves > 0) {
    const int r0 = wave_index(ctx.tid) - first_wave, r = r0 < n_waves ? r0 : -1, lane = ctx.tid & 63;
    if (r < 0) return;
#if defined(HSQP_PHASE_PROFILE)
    long long* prof = (ctx.tid >> 6) == PROF_WAVE ? ctx.prof : nullptr;
#else
    long long* prof = nullptr;
#endif
    int g0 = xty_deal_one<SPACES, RIC_PF>(j0, 0, r, n_waves, lane, prof);
    if constexpr (NJ > 1) g0 += xty_deal_one<SPACES, RIC_PF>(j1, g0, r, n_waves, lane, prof);
    if constexpr (NJ > 2) xty_deal_one<SPACES, RIC_PF>(j2, g0, r, n_waves, lane, prof);
    return;
  }
#endif
  if (is_mfma_half(ctx)) {
    const Ctx mc = mfma_ctx(ctx);
    wg_xty_jobs<false, SPACES>(mc, &j0, 1);
    if constexpr (NJ > 1) wg_xty_jobs<false, SPACES>(mc, &j1, 1);
    if constexpr (NJ > 2) wg_xty_jobs<false, SPACES>(mc, &j2, 1);
  }
}

#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) void* hsqp_ldsptr;
// Ph3 on the device: the whole elimination of [Lam | I | G | g] inside two waves on the FP64 matrix cores (eliminate_blocked, hsqp_elim.h), no barrier and
// no LDS traffic per step; both waves carry Lam, so they never talk to each other.  On the way out every row i is scaled with d_i^-1/2
// (Cholesky scaling of the unit-lower inverse), which turns the eliminated blocks into L^-1, Z = L^-1 G and z = L^-1 g.
// The eliminating waves also start the copy of the next stage's A~ into the other LDS buffer — asynchronous 16-byte copies straight into LDS
// (no staging registers, no store pass): the LDS image of A~ is the record's [58][58] block byte for byte; a wave instruction moves 64 x 16 B
// to (wave-uniform base) + lane x 16.  The compiler makes a wave wait for its copies before that wave's next LDS access; these two waves have
// none until the write-out (every other wave reads LDS all the time).
HSQP_D void next_a_to_lds(const double* a_next, double* a_dst, int wave, int lane) {
  if (a_next) {
    constexpr int NCH = NX * NX / 2;                         // 16-byte chunks
    static_assert((NX * NX) % 2 == 0 && QP_A % 2 == 0 && QP_SIZE % 2 == 0, "16-byte alignment of A~ in the record");
#pragma unroll
    for (int t = 0; t < (NCH + 127) / 128; ++t) {
      const int c0 = (t * 2 + wave) * 64;                    // first chunk of this wave instruction
      if (c0 + lane < NCH) __builtin_amdgcn_global_load_lds((hsqp_gcptr)a_next + 2 * (c0 + lane), (hsqp_ldsptr)(a_dst + 2 * c0), 16, 0, 0);
    }
  }
}
#endif

// qp: [N][QP_SIZE] of this instance, ric: [N][RIC_SIZE].  w.ok reports whether every Lam was positive definite.
// vf (optional, [N+1][VF_SIZE]): the value function S_k, s_k of every node, for the KKT check (lam_k = S_k dx_k + s_k).
constexpr int VF_SIZE = NX * NX + NX;
// NXE: number of leading states that take part in the recursion.  NX for the whole-body problem; 35 for the centroidal problem
// embedded in the 58-state layout, whose padding states are decoupled (A~ = I, B~ = 0, no cost there), so S, s, K
// vanish on them identically: every product is restricted to the leading NXE x NXE blocks (leading dimensions stay NX) and the
// padding parts of the outputs are written as zeros / left untouched where nothing reads them.
// termS / terms (optional): start the recursion from the value function 1/2 x' termS x + terms' x (NXE x NXE, leading dimension term_ld; NXE) instead
// of the terminal cost — the parallel-in-time path (hsqp_scan.h) runs single stages from the scanned value functions (terms_sign = -1:
// the scan carries eta = -s); then vf_terminal = false leaves vf[N] to the workgroup that owns it.
template <int NXE = NX>
HSQP_HD void riccati_backward(const Ctx& ctx, RicWS& w, const double* Qf, const double* xN, const double* parN, const double* qp,
                              double* ric, int N, double* vf = nullptr, const double* termS = nullptr, const double* terms = nullptr,
                              bool vf_terminal = true, double terms_sign = 1.0, int term_ld = NXE,
                              double* linv_out = nullptr /* optional [N][LDB * LDB]: (L^-1)^T of every stage (segmented sweep, hsqp_segment.h) */,
                              int vf_mode = 0 /* which nodes vf receives: 0 every node k (vf[k]); 1 node 0 only; 2 node 0 and the last stage's node N - 1
                                                 (the two-level sweep's gate evaluates the KKT residual of the stages at the segment boundaries only) */) {
#if defined(__HIP_DEVICE_COMPILE__)
  const bool dev512 = ctx.nthreads == 512;     // the kernels' shape: tiles over all eight waves, two-wave elimination, direct global -> LDS copies
#else
  const bool dev512 = false;
#endif
  WG_FOR(ctx, i, NX * NX + NX + 1 + LDB * LDB) {
    if (i < NX * NX) {
      const int r = i / NX, c = i % NX;
      w.S[r][c] = termS ? ((r < NXE && c < NXE) ? termS[r * term_ld + c] : 0.0) : (r == c ? Qf[r] : 0.0);
    } else if (i < NX * NX + NX) {
      const int r = i - NX * NX;
      const double sr = termS ? (r < NXE ? terms_sign * terms[r] : 0.0) : Qf[r] * (xN[r] - parN[HSQP_P_XDES + r]);
      w.sv[r] = sr;
      w.part[4 * r] = sr; w.part[4 * r + 1] = 0.0; w.part[4 * r + 2] = 0.0; w.part[4 * r + 3] = 0.0;   // s travels as four partial sums (Ph4 -> Ph1)
    }
    else if (i == NX * NX + NX) w.ok = 1;
    else { const int t = i - NX * NX - NX - 1; w.fac.LinvT[t / LDB][t % LDB] = 0.0; }   // (its padding row / column is never written)
  }
  WG_SYNC(ctx);
  if (vf && vf_terminal) {
    WG_FOR(ctx, i, VF_SIZE) vf[(size_t)N * VF_SIZE + i] = i < NX * NX ? w.S[i / NX][i % NX] : w.sv[i - NX * NX];
  }
  // stage N-1 data -> LDS (the later stages are prefetched while the previous one is being processed)
  {
    const double* q = qp + (size_t)(N - 1) * QP_SIZE;
    double(*An)[NX] = w.A2[(N - 1) & 1];
    constexpr int na = nbatches(NX * NX, 8);
    WG_FOR(ctx, it, na + NX * LDB) {
      if (it < na) copy_batch<8>(it, NX * NX, q + QP_A, [&](int i, double v) { An[i / NX][i % NX] = v; });
      else { const int i = it - na, r = i / LDB, c = i % LDB; w.B[r][c] = c < NUT ? q[QP_B + r * NUT + c] : q[QP_BV + r]; }
    }
    if (dev512) WG_FOR(ctx, r, NUT) w.kv[r] = q[QP_RV + r];   // r~ of the first stage (see Ph3)
  }
  WG_SYNC(ctx);
  const Ctx& ctx_outer = ctx;
  for (int k = N - 1; k >= 0; --k) {
    // The thread index is made opaque once per stage: everything a lane derives from it (tile coordinates, LDS offsets of its operand
    // rows, the column it holds in the elimination) is loop-invariant, and hoisted out of the stage loop those values — well over a
    // hundred registers — are spilled to scratch and re-loaded inside the serial chain.  Recomputing them costs a few hundred integer
    // instructions per stage.
    Ctx ctx = ctx_outer;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(ctx.tid));
#endif
    const double* q = qp + (size_t)k * QP_SIZE;
    const double* qn = qp + (size_t)(k > 0 ? k - 1 : 0) * QP_SIZE;   // next stage to be processed
    double* rk = ric + (size_t)k * RIC_SIZE;
    double(*A)[NX] = w.A2[k & 1];
    double(*An)[NX] = w.A2[(k + 1) & 1];
    PH_TICK(ctx, 1);
    PH_MARK(ctx);
    // ---- Ph1: [SB | S b~] = S [B~ | b~] (S symmetric => X = S); s of this stage from its partial sums
    {
      const XtyJob jsb = xty_job(NXE, NUT + 1, NXE, &w.S[0][0], NX, &w.B[0][0], LDB, &w.SB[0][0], LDB);
      if (is_helper_half(ctx)) {
        const Ctx hc = helper_ctx(ctx);
        WG_FOR(hc, it, NX + NUT) {
          if (it < NX) {   // s of this stage = the four partial sums the previous stage's Ph4 left (no roll-up phase in between)
            const double* sp = &w.part[4 * it];
            w.sv[it] = (sp[0] + sp[1]) + (sp[2] + sp[3]);
          } else if (k < N - 1) ric[(size_t)(k + 1) * RIC_SIZE + RIC_KV + it - NX] = w.Em[it - NX][EM_G + NXE];   // k of the previous stage
        }
        if (!dev512 && k > 0) {
          constexpr int na = nbatches(NX * NX, 8);
          WG_FOR(hc, it, na) copy_batch<8>(it, NX * NX, qn + QP_A, [&](int i, double v) { An[i / NX][i % NX] = v; });
        }
      }
      ric_products(ctx, 0, 8, jsb);
    }
    PH_ARRIVE(ctx, 0);
    WG_SYNC(ctx);
    PH_TICK(ctx, 2);
    PH_MARK(ctx);
    // ---- Ph2: G = P~ + SB^T A~, Lam = R~ + B~^T SB;  sb = s + S b~;  g = r~ + B~^T sb in four partial sums per row
    {
      const XtyJob jg = xty_job(NUT, NXE, NXE, &w.SB[0][0], LDB, &A[0][0], NX, &w.Em[0][EM_G], LDE, q + QP_P, NX);
      const XtyJob jl = xty_job(NUT, NUT, NXE, &w.B[0][0], LDB, &w.SB[0][0], LDB, &w.fac.Ef[0][0], LDF, q + QP_R, NUT);
      if (is_helper_half(ctx)) {
        const Ctx hc = helper_ctx(ctx);
        static_assert(128 + 4 * NUT <= RIC_HELPERS && NX <= 128, "one pass; the g items sit on the helper waves with the fewest tiles");
        WG_FOR(hc, it0, RIC_HELPERS) {
          const int it = it0 - 128;
          if (it0 < NX) w.sb[it0] = it0 < NXE ? w.sv[it0] + w.SB[it0][NUT] : 0.0;
          if (it >= 0 && it < 4 * NUT) {
            const int r = it >> 2, p = it & 3;
            constexpr int LA = (NXE + 3) / 4;
            double sg = p == 0 ? (dev512 ? w.kv[r] : q[QP_RV + r]) : 0.0;   // (device kernels: r~ staged in kv by the memory waves a stage ahead)
#pragma unroll
            for (int l = 0; l < LA; ++l) {
              const int ll = p * LA + l, lc = ll < NXE ? ll : NXE - 1;
              const double a = w.B[lc][r], b = w.sv[lc] + w.SB[lc][NUT];
              sg += ll < NXE ? a * b : 0.0;
            }
#if defined(__HIP_DEVICE_COMPILE__)
            if (dev512) {   // the four partial sums of a row sit in adjacent lanes: two DPP quad permutes leave the row's sum in column 0
              sg += quad_perm_f64<0xB1>(sg);
              sg += quad_perm_f64<0x4E>(sg);
              if (p == 0) w.Em[r][EM_GVP] = sg;
            } else
#endif
            w.Em[r][EM_GVP + p] = sg;
          }
        }
      }
      ric_products<XTY_ADD_GLOBAL, 2>(ctx, 0, 8, jg, jl);
    }
    PH_ARRIVE(ctx, 1);
    WG_SYNC(ctx);
    PH_TICK(ctx, 3);
    PH_MARK(ctx);
    // ---- Ph3: factorisation.  Leaves L^-1 (Ef columns 24.., LinvT), Z = L^-1 G (Zs), z = L^-1 g (zv).  Under it: SA = S A~ (needed by the
    //      S-update only), the next stage's A~ and [B~ | b~] -> LDS (B is dead since Ph2)
    const XtyJob jsa = xty_job(NXE, NXE, NXE, &w.S[0][0], NX, &A[0][0], NX, &w.SA[0][0], NX);
#if defined(__HIP_DEVICE_COMPILE__)
    if (dev512) {
      // wave w sits on SIMD w % 4, and FP64 matrix and vector instructions of one SIMD do not overlap (measured: the elimination took
      // 18 k cycles next to a wave with 75 matrix instructions, 12 k alone): SIMDs 0, 1 eliminate (waves 0, 1; waves 4, 5 only move the
      // next stage's [B~ | b~]), SIMDs 2, 3 carry the tiles of S A~ (waves 2, 3, 6, 7)
      const int wv = wave_index(ctx.tid);
      const int trank = (wv & 3) >= 2 ? (wv & 1) + (wv >> 2) * 2 : -1;
      constexpr int NPB = 12;                       // 128 threads x 12 >= 58 * 23 + 58
      const int pt = ctx.tid - 256;
      if (wv < 2) {
        __builtin_amdgcn_s_setprio(3);
        const int lane = ctx.tid & 63;
        const DevWave dw{lane};
        const ElimIO io{&w.fac.Ef[0][0], LDF, &w.Em[0][EM_G], &w.Em[0][EM_GVP], LDE, &w.fac.Ef[0][EF_MI], LDF, &w.fac.LinvT[0][0], LDB, &w.Zs[0][0], LDZ, w.zv, &w.ok};
        const double* a_next = k > 0 ? qn + QP_A : nullptr;
        double* a_dst = &An[0][0];
        auto prefetch = [&]() { next_a_to_lds(a_next, a_dst, wv, lane); };
        if (wv == 0) eliminate_blocked<NXE, 0>(dw, io, prefetch);
        else eliminate_blocked<NXE, 1>(dw, io, prefetch);
        __builtin_amdgcn_s_setprio(0);
      } else if (trank < 0) {
        // These two waves share their SIMDs with the eliminating waves, and a SIMD runs the FP64 vector and matrix instructions of its waves
        // one after the other: arithmetic done here (round 3: q~ + A~^T sb, 232 fifteen-term sums) lengthens the phase by its own duration.
        // They are the stage's MEMORY waves now: global -> LDS staging of everything the later phases would otherwise fetch from HBM inside
        // the serial chain (a round trip is 3.5 - 5 k cycles in this kernel; waiting costs no issue slot): the next stage's [B~ | b~], and
        // q~ of this stage for the partial sums of Ph4 (into dx, unused by the backward sweep).
        {
          double pb[NPB];
          const double qv = pt < NX ? q[QP_QV + pt] : 0.0;
          // (r~ of the next stage with the other loads: issued behind the LDS stores below it was a second HBM round trip of wave 5, which then
          //  arrived at the phase's barrier with the slower eliminating wave — 12.0 instead of 10.2 k cycles; k_riccati 1.535 -> 1.516 ms)
          const bool has_rv = k > 0 && pt >= 64 && pt < 64 + NUT;
          const double rvn = qn[QP_RV + (has_rv ? pt - 64 : 0)];
          if (k > 0) {
#pragma unroll
            for (int t = 0; t < NPB; ++t) { const int idx = pt + 128 * t; pb[t] = idx < NX * NUT ? qn[QP_B + idx] : (idx < NX * NUT + NX ? qn[QP_BV + idx - NX * NUT] : 0.0); }
#pragma unroll
            for (int t = 0; t < NPB; ++t) {
              const int idx = pt + 128 * t;
              if (idx < NX * NUT) w.B[idx / NUT][idx % NUT] = pb[t];
              else if (idx < NX * NUT + NX) w.B[idx - NX * NUT][NUT] = pb[t];
            }
          }
          if (pt < NX) w.dx[pt] = qv;
          if (has_rv) w.kv[pt - 64] = rvn;   // r~ of the next stage (Ph2's g sums)
        }
      } else ric_products_ranked(ctx, trank, 4, jsa);
    } else
#endif
    {
      // generic form (host build): elimination of [Lam | I] on full rows in LDS (rows stay unscaled, so a step needs no pivot broadcast
      // phase: one barrier per column; the multiplier is read from the symmetric upper part), the Cholesky scaling, then Z, z as products
      WG_FOR(ctx, j, NUT * (LDF - NUT)) { const int r = j / (LDF - NUT), c = NUT + j % (LDF - NUT); w.fac.Ef[r][c] = (c - EF_MI == r) ? 1.0 : 0.0; }
      ric_products(ctx, 0, 0, jsa);
      WG_SYNC(ctx);
      for (int j = 0; j < NUT - 1; ++j) {
        WG_FOR(ctx, it, NUT * 16) {
          const int i = it >> 4, c0 = it & 15;
          if (i <= j) continue;
          const double pj = w.fac.Ef[j][j], fji = w.fac.Ef[j][i];
          const double ej0 = w.fac.Ef[j][c0], ej1 = w.fac.Ef[j][c0 + 16], ej2 = w.fac.Ef[j][c0 + 32];
          const double ei0 = w.fac.Ef[i][c0], ei1 = w.fac.Ef[i][c0 + 16], ei2 = w.fac.Ef[i][c0 + 32];
          const double f = fji * fast_rcp(pj);
          w.fac.Ef[i][c0] = ei0 - f * ej0;
          w.fac.Ef[i][c0 + 16] = ei1 - f * ej1;
          w.fac.Ef[i][c0 + 32] = ei2 - f * ej2;
        }
        WG_SYNC(ctx);
      }
      WG_FOR(ctx, it, NUT * LDB) {   // L^-1 = D^-1/2 Mi (in place) and its transpose
        const int r = it / LDB, c = it % LDB;
        double dj = w.fac.Ef[r][r];
        if (!(dj > 0.0)) { dj = 1.0; if (c == 0) w.ok = 0; }
        const double v = c <= r ? w.fac.Ef[r][EF_MI + c] * inv_sqrt(dj) : 0.0;
        w.fac.Ef[r][EF_MI + c] = v;
        if (c < NUT) w.fac.LinvT[c][r] = v;
      }
      WG_SYNC(ctx);
      {
        const XtyJob job = xty_job(NUT, NXE, NUT, &w.fac.LinvT[0][0], LDB, &w.Em[0][EM_G], LDE, &w.Zs[0][0], LDZ);
        if (is_mfma_half(ctx)) wg_xty_jobs(mfma_ctx(ctx), &job, 1);
        if (is_helper_half(ctx)) {
          const Ctx hc = helper_ctx(ctx);
          WG_FOR(hc, it, NUT + NX * LDB) {
            if (it < NUT) {
              double s = 0.0;
              for (int l = 0; l < NUT; ++l) s += w.fac.Ef[it][EF_MI + l] * ((w.Em[l][EM_GVP] + w.Em[l][EM_GVP + 1]) + (w.Em[l][EM_GVP + 2] + w.Em[l][EM_GVP + 3]));
              w.zv[it] = s;
              w.Zs[it][NXE] = s;
            } else if (k > 0) {
              const int i = it - NUT, r = i / LDB, c = i % LDB;
              w.B[r][c] = c < NUT ? qn[QP_B + r * NUT + c] : qn[QP_BV + r];
            }
          }
        }
      }
    }
    PH_ARRIVE(ctx, 3);
    WG_SYNC(ctx);
    PH_TICK(ctx, 4);
    PH_MARK(ctx);
    // ---- Ph4: S = Q~ + A~^T SA - Z^T Z, K = -L^-T Z (into the G block and the record), s <- q~ + A~^T sb - Z^T z (left as four partial
    //      sums), k = -L^-T z
    {
      XtyJob js0 = xty_job(NXE, NXE, NXE, &A[0][0], NX, &w.SA[0][0], NX, &w.S[0][0], NX, q + QP_Q, NX);
      js0.L2 = NUT; js0.X2 = &w.Zs[0][0]; js0.ldx2 = LDZ; js0.Y2 = &w.Zs[0][0]; js0.ldy2 = LDZ; js0.sign2 = -1.0;
      const XtyJob js = xty_sym(js0);   // S is symmetric: tiles on/above the diagonal, mirrored into the LDS copy
      // ([K | k] = -L^-T [Z | z] goes to the G block — closed_loop_record reads K there, k is picked up from its column NXE in the next
      //  stage's Ph1 — and K straight to the record)
      const XtyJob jk = xty_also_to(xty_job(NUT, NXE + 1, NUT, &w.fac.Ef[0][EF_MI], LDF, &w.Zs[0][0], LDZ, &w.Em[0][EM_G], LDE, nullptr, 0, -1.0), rk + RIC_K, NX, NXE);
      // s <- q~ + A^T sb - Z^T z in four partial sums per row (added in the next stage's Ph1): short chains, 232 items
      auto s_item = [&](int it) {
        const int r = it >> 2, p = it & 3;
        constexpr int LA = (NXE + 3) / 4, LZ = (NUT + 3) / 4;
        double s;
        {
#if defined(__HIP_DEVICE_COMPILE__)
          s = p == 0 ? (dev512 ? w.dx[r] : q[QP_QV + r]) : 0.0;   // (staged by the memory waves in Ph3)
#else
          s = p == 0 ? q[QP_QV + r] : 0.0;
#endif
#pragma unroll
          for (int l = 0; l < LA; ++l) { const int ll = p * LA + l, lc = ll < NXE ? ll : NXE - 1; const double a = A[lc][r], b = w.sb[lc]; s += ll < NXE ? a * b : 0.0; }
        }
#pragma unroll
        for (int l = 0; l < LZ; ++l) { const int ll = p * LZ + l, lc = ll < NUT ? ll : NUT - 1; const double a = w.Zs[lc][r], b = w.zv[lc]; s -= ll < NUT ? a * b : 0.0; }
        w.part[it] = (NXE == NX || r < NXE) ? s : 0.0;
      };
#if defined(__HIP_DEVICE_COMPILE__)
      if (dev512) {   // waves 6, 7 (two short tiles in this phase), two passes; waves 4, 5 go straight to their tiles
        if (ctx.tid >= 384) for (int it = ctx.tid - 384; it < 4 * NX; it += 128) s_item(it);
      }
#endif
      if (is_helper_half(ctx)) {
        const Ctx hc = helper_ctx(ctx);
        static_assert(4 * NX <= RIC_HELPERS, "one pass");
#if defined(__HIP_DEVICE_COMPILE__)
        if (!dev512)
#endif
        WG_FOR(hc, it, 4 * NX) s_item(it);
        if (linv_out) WG_FOR(hc, i, LDB * LDB) linv_out[(size_t)k * LDB * LDB + i] = w.fac.LinvT[i / LDB][i % LDB];
        if (NXE < NX) WG_FOR(hc, i, NUT * (NX - NXE)) rk[RIC_K + (i / (NX - NXE)) * NX + NXE + i % (NX - NXE)] = 0.0;   // K vanishes on the padding states
      }
#if defined(__HIP_DEVICE_COMPILE__)
      if (dev512) {
        // per SIMD (waves w, w + 4) about the same number of matrix instructions, fewer on the waves with the vector items:
        // S: waves 0-3 two tiles, 4-5 one; [K | k]: waves 6-7 two tiles, 2-5 one
        const int wv = wave_index(ctx.tid);
        ric_products_ranked<XTY_ADD_GLOBAL>(ctx, wv < 6 ? wv : -1, 6, js);
        ric_products_ranked(ctx, wv >= 6 ? wv - 6 : (wv >= 4 ? wv - 2 : (wv >= 2 ? wv + 2 : -1)), 6, jk);
      } else
#endif
        ric_products<0, 2>(ctx, 0, 0, js, jk);
    }
    PH_ARRIVE(ctx, 2);
    WG_SYNC(ctx);
    PH_TICK(ctx, 5);
    // (the symmetric tile jobs write the diagonal tiles of S symmetric themselves — upper triangle computed, mirrored —, s stays
    //  in its four partial sums until the next stage's Ph1 adds them)
    if (vf && (vf_mode == 0 || k == 0 || (vf_mode == 2 && k == N - 1))) {
      WG_FOR(ctx, i, VF_SIZE) {
        const double* sp = &w.part[4 * (i >= NX * NX ? i - NX * NX : 0)];
        vf[(size_t)k * VF_SIZE + i] = i < NX * NX ? w.S[i / NX][i % NX] : (sp[0] + sp[1]) + (sp[2] + sp[3]);
      }
    }
    PH_TICK(ctx, 6);
  }
  WG_FOR(ctx, r, NUT) { const double kr = w.Em[r][EM_G + NXE]; w.kv[r] = kr; ric[RIC_KV + r] = kr; }   // k of stage 0
  WG_SYNC(ctx);
}

// Forward sweep dx+ = A~ dx + B~ (K dx + k) + b~ (serial over stages); writes dx [N+1][58].  qp: the stage QP records, ric: the gains.
// Three phases per stage: (1) the partial sums of A~ dx (58 rows) and K dx (23 rows), four per row; (2) the partial sums of B~ ut with
// ut = k + K dx taken from the partial sums of (1) by every item itself (no separate phase for 23 numbers); (3) dx+.
// ut_out (optional, [N][NUT]): ut = k + K dx of every stage as the sweep forms it (the device path: bit for bit what step_node would compute from the same
// dx — the same partial sums in the same order), so that the step kernel need not read the gains again (10.7 of its 33 KB per node)
template <int NXE = NX>
HSQP_HD void riccati_forward(const Ctx& ctx, RicWS& w, const double* x_init, const double* x, const double* qp, const double* ric, int N, double* dx_out,
                             double* ut_out = nullptr) {
  WG_FOR(ctx, i, NX) {
    const double d = x_init[i] - x[i];
    w.dx[i] = d;
    dx_out[i] = d;
  }
  WG_SYNC(ctx);
  constexpr int NC = (NXE + 3) / 4, NCB = (NUT + 3) / 4;
  constexpr int NR1 = 4 * (NX + NUT);            // items of phase 1: rows of A~ (NX), then rows of K (NUT), four partial sums each
  double* part1 = w.SA[0];                       // the sweep's scratch: the backward sweep's SA is dead
  double* part2 = w.SA[0] + NR1;
#if defined(__HIP_DEVICE_COMPILE__)
  if (ctx.nthreads >= NR1 + NUT) {
    // device path: the row slices of a stage are fetched into registers PF stages ahead.  A stage takes ~0.4 us, its A~, B~, K come from
    // HBM (the records of an iteration are far larger than the caches): with the loads only one stage ahead every stage waited out the
    // HBM round trip (2.3 us per stage measured), so the slices travel in a ring of PF register sets (the loop is unrolled by PF so that
    // every set has fixed registers).  Item it < NR1 owns row it >> 2 of [A~; K] (columns p + 4c) and, if it < 4 NX, row it >> 2 of
    // B~ as well; items NR1 .. NR1 + NUT - 1 carry k.
    constexpr int PF = 3;    // (4 sets spill: 22 doubles per set)
    const int it = ctx.tid, row = it >> 2, p = it & 3;
    const bool rowA = it < 4 * NX, rowK = it >= 4 * NX && it < NR1, isk = it >= NR1 && it < NR1 + NUT;
    const bool liveA = rowA && (NXE == NX || row < NXE);
    double a[PF][NC], bq[PF][NCB], sc[PF];   // sc: b~ of the row (p == 0 items of A~ rows) or k (isk items)
    auto fetch = [&](int k, double* av, double* bv, double& s1) {
      const double* q = qp + (size_t)k * QP_SIZE;
      const double* rk = ric + (size_t)k * RIC_SIZE;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int cc = p + 4 * c;
        av[c] = (cc < NXE) ? (liveA ? q[QP_A + row * NX + cc] : (rowK ? rk[RIC_K + (row - NX) * NX + cc] : 0.0)) : 0.0;
      }
#pragma unroll
      for (int c = 0; c < NCB; ++c) { const int cc = p + 4 * c; bv[c] = (liveA && cc < NUT) ? q[QP_B + row * NUT + cc] : 0.0; }
      s1 = (liveA && p == 0) ? q[QP_BV + row] : (isk ? rk[RIC_KV + it - NR1] : 0.0);
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) { if (u < N) fetch(u, a[u], bq[u], sc[u]); }
    for (int k0 = 0; k0 < N; k0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int k = k0 + u;
        if (k < N) {
          // TWO barriers per stage: the four partial sums of a row sit in four adjacent lanes and are added by DPP quad permutes (no LDS
          // round trip for them); only ut = k + K dx (23 numbers every row needs) and the new dx go through LDS; dx is double-buffered
          // (dx / sv) so that the next stage's reads do not race this stage's writes
          const double* dcur = (k & 1) ? w.sv : w.dx;
          double* dnxt = (k & 1) ? w.dx : w.sv;
          double s1 = 0.0;
#pragma unroll
          for (int c = 0; c < NC; ++c) { const int cc = p + 4 * c; if (cc < NXE) s1 += a[u][c] * dcur[cc]; }
          if (rowK) {          // row of K: ut_j = k_j + K_j dx
            double s = s1;
            s += quad_perm_f64<0xB1>(s);
            s += quad_perm_f64<0x4E>(s);
            if (p == 0) w.zv[row - NX] = s;
          } else if (isk) w.kv[it - NR1] = sc[u];
          WG_SYNC(ctx);
          if (ut_out && it < NUT) ut_out[(size_t)k * NUT + it] = w.kv[it] + w.zv[it];
          {
            double s = rowA ? s1 + (p == 0 ? sc[u] : 0.0) : 0.0;      // A~ dx slice (+ b~ on the first lane of the quad)
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
              const int j = p + 4 * c, jc = j < NUT ? j : NUT - 1;
              const double utj = w.kv[jc] + w.zv[jc];
              s += (rowA && j < NUT) ? bq[u][c] * utj : 0.0;
            }
            s += quad_perm_f64<0xB1>(s);
            s += quad_perm_f64<0x4E>(s);
            if (rowA && p == 0) {
              // padding states (NXE < NX): dx+ = dx (A~ = I there), which is zero when the caller keeps them zero
              const double v = (NXE == NX || row < NXE) ? s : dcur[row];
              dnxt[row] = v;
              dx_out[(size_t)(k + 1) * NX + row] = v;
            }
          }
          if (k + PF < N) fetch(k + PF, a[u], bq[u], sc[u]);
          WG_SYNC(ctx);
        }
      }
    }
    return;
  }
#endif
  for (int k = 0; k < N; ++k) {
    const double* q = qp + (size_t)k * QP_SIZE;
    const double* rk = ric + (size_t)k * RIC_SIZE;
    WG_FOR(ctx, it, NR1 + NUT) {
      if (it < 4 * NX) part1[it] = ((it >> 2) < NXE) ? matvec_part<NXE>(q + QP_A + (it >> 2) * NX, w.dx, it & 3) : 0.0;
      else if (it < NR1) part1[it] = matvec_part<NXE>(rk + RIC_K + ((it >> 2) - NX) * NX, w.dx, it & 3);
      else w.kv[it - NR1] = rk[RIC_KV + it - NR1];
    }
    WG_SYNC(ctx);
    if (ut_out) WG_FOR(ctx, j, NUT) { const double* pj = &part1[4 * (NX + j)]; ut_out[(size_t)k * NUT + j] = w.kv[j] + ((pj[0] + pj[1]) + (pj[2] + pj[3])); }
    WG_FOR(ctx, it, 4 * NX) {
      const int row = it >> 2, p = it & 3;
      double s = 0.0;
      for (int c = 0; c < NCB; ++c) {
        const int j = p + 4 * c;
        if (j < NUT && row < NXE) { const double* pj = &part1[4 * (NX + j)]; s += q[QP_B + row * NUT + j] * (w.kv[j] + ((pj[0] + pj[1]) + (pj[2] + pj[3]))); }
      }
      part2[it] = ((p == 0 && row < NXE) ? q[QP_BV + row] : 0.0) + s;
    }
    WG_SYNC(ctx);
    WG_FOR(ctx, i, NX) {
      const double* p1 = &part1[4 * i];
      const double* p2 = &part2[4 * i];
      const double s = (NXE == NX || i < NXE) ? ((p1[0] + p1[1]) + (p1[2] + p1[3])) + ((p2[0] + p2[1]) + (p2[2] + p2[3])) : w.dx[i];
      w.dx[i] = s;
      dx_out[(size_t)(k + 1) * NX + i] = s;
    }
    WG_SYNC(ctx);
  }
}

// ---- closed-loop roll-out of the parallel-in-time path (hsqp_scan.h).  There the gains of all stages are formed in parallel (one
// workgroup per stage), so each workgroup can also afford the closed loop A_cl = A~ + B~ K, b_cl = b~ + B~ k of its stage: the serial
// roll-out is then ONE mat-vec phase per stage (dx+ = A_cl dx + b_cl) instead of the three of riccati_forward, which applies A~, B~, K
// separately because forming A_cl inside the serial backward sweep would put it on that sweep's critical path.
template <int NXE>
constexpr int ACL_SIZE = ((NXE * NXE + NXE + 7) / 8) * 8;   // [NXE][NXE] A_cl, [NXE] b_cl

// after riccati_backward(..., N = 1, ...): the stage's A~, B~, b~ and its gains are still in the workspace
template <int NXE>
HSQP_HD void closed_loop_record(const Ctx& ctx, const RicWS& w, double* acl) {
  WG_FOR(ctx, it, NXE * NXE + NXE) {
    if (it < NXE * NXE) {
      const int i = it / NXE, j = it % NXE;
      double s = w.A2[0][i][j];
#pragma unroll
      for (int m = 0; m < NUT; ++m) s += w.B[i][m] * w.Em[m][EM_G + j];
      acl[it] = s;
    } else {
      const int i = it - NXE * NXE;
      double s = w.B[i][NUT];
#pragma unroll
      for (int m = 0; m < NUT; ++m) s += w.B[i][m] * w.kv[m];
      acl[it] = s;
    }
  }
}

// dx [N+1][58]: dx_0 = x_init - x_0, dx+ = A_cl dx + b_cl on the leading NXE states; the padding states keep dx (A~ = I there).
template <int NXE>
HSQP_HD void closed_loop_forward(const Ctx& ctx, RicWS& w, const double* x_init, const double* x, const double* acl, int N, double* dx_out) {
  WG_FOR(ctx, i, NX) {
    const double d = x_init[i] - x[i];
    w.dx[i] = d;
    dx_out[i] = d;
  }
  WG_SYNC(ctx);
  constexpr int NC = (NXE + 3) / 4;
  double* part = w.SA[0];
#if defined(__HIP_DEVICE_COMPILE__)
  if (ctx.nthreads >= 4 * NXE) {
    // item it < 4 NXE owns the columns p + 4c of row it >> 2.  The closed loops were written by N different workgroups a moment ago, so a
    // stage's slice comes from the far caches (1 - 2 us) while a stage takes 0.4 us: the slices are fetched PF stages ahead into a ring of
    // register sets (the loop is unrolled by PF so that every set has fixed registers)
    constexpr int PF = 4;
    const int it = ctx.tid, row = it >> 2, p = it & 3;
    const bool live = it < 4 * NXE;
    double a[PF][NC], sc[PF];
    auto fetch = [&](int k, double* av, double& s1) {
      const double* r = acl + (size_t)k * ACL_SIZE<NXE>;
#pragma unroll
      for (int c = 0; c < NC; ++c) { const int cc = p + 4 * c; av[c] = (live && cc < NXE) ? r[row * NXE + cc] : 0.0; }
      s1 = (live && p == 0) ? r[NXE * NXE + row] : 0.0;
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) { if (u < N) fetch(u, a[u], sc[u]); }
    for (int k0 = 0; k0 < N; k0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int k = k0 + u;
        if (k < N) {
          // ONE barrier per stage: the four partial sums of a row sit in four adjacent lanes and are added with two DPP quad
          // permutes (no LDS round trip); dx is double-buffered (dx / sv) so that the next stage's reads do not race this stage's writes
          const double* dcur = (k & 1) ? w.sv : w.dx;
          double* dnxt = (k & 1) ? w.dx : w.sv;
          double s = sc[u];
#pragma unroll
          for (int c = 0; c < NC; ++c) { const int cc = p + 4 * c; if (cc < NXE) s += a[u][c] * dcur[cc]; }
          s += quad_perm_f64<0xB1>(s);     // lanes (0,1) (2,3) of the quad exchange
          s += quad_perm_f64<0x4E>(s);     // halves of the quad exchange: every lane of the quad holds the row's sum
          if (live && p == 0) { dnxt[row] = s; dx_out[(size_t)(k + 1) * NX + row] = s; }
          if (NXE < NX && it >= 4 * NXE && it < 4 * NXE + (NX - NXE)) {   // padding states keep dx (A~ = I there)
            const int r = NXE + it - 4 * NXE;
            const double v = dcur[r];
            dnxt[r] = v;
            dx_out[(size_t)(k + 1) * NX + r] = v;
          }
          if (k + PF < N) fetch(k + PF, a[u], sc[u]);
          WG_SYNC(ctx);
        }
      }
    }
    return;
  }
#endif
  for (int k = 0; k < N; ++k) {
    const double* r = acl + (size_t)k * ACL_SIZE<NXE>;
    WG_FOR(ctx, it, 4 * NXE) {
      const int row = it >> 2, p = it & 3;
      double s = p == 0 ? r[NXE * NXE + row] : 0.0;
      for (int c = 0; c < NC; ++c) { const int cc = p + 4 * c; if (cc < NXE) s += r[row * NXE + cc] * w.dx[cc]; }
      part[it] = s;
    }
    WG_SYNC(ctx);
    WG_FOR(ctx, i, NX) {
      const double* p1 = &part[4 * (i < NXE ? i : 0)];
      const double s = (NXE == NX || i < NXE) ? (p1[0] + p1[1]) + (p1[2] + p1[3]) : w.dx[NXE == NX ? 0 : i];
      w.dx[i] = s;
      dx_out[(size_t)(k + 1) * NX + i] = s;
    }
    WG_SYNC(ctx);
  }
}

// Per-node recovery of the inputs and the step of length alpha (parallel over all nodes of all instances):
//   ut = K dx + k,  du = Px dx + Pu ut + Pe,  x_new = x + alpha dx,  u_new = u + alpha du.
struct StepWS {
  double dx[NX], t[NUT], ut[NUT], du[NU];
  double part[(NX + NU) * 4];
};
// info (optional, 3 doubles): {q~.dx + r~.ut (Armijo descent metric of the projected QP, ocs2 multiple_shooting::
// armijoDescentMetric on the projected cost), |dx|^2, |du|^2} of this node.
// ut_in (optional): ut of this node as the serial roll-out left it (riccati_forward's ut_out); the gains are then not read.
// fj_in (optional, NJ doubles; whole-body records only): rows 12 .. 34 of Px dx + Pu ut as the factored roll-out left them (riccati_forward_fact's fj_out: it forms
// exactly these sums to advance the joint states); rows 12 .. of Px / Pu are then not read — 15 of the node's 24 KB.
HSQP_HD void step_node(const Ctx& ctx, StepWS& w, const double* q, const double* rk, const double* dx, const double* x, const double* u,
                       double alpha, double* ut_out, double* du_out, double* x_new, double* u_new, double* info = nullptr, const double* ut_in = nullptr,
                       const double* fj_in = nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (ctx.nthreads == 64) {
    // One-wave kernels (k_step, k_step_value): the phase-by-phase form below exposes four dependent HBM round trips (dx; the rows of K;
    // the rows of Px / Pu; q~, r~ of the descent metric) on a node whose arithmetic is a few hundred multiply-adds.  Here EVERY global
    // load of the node is issued before the first use (93 doubles per lane in registers), so the node pays one round trip; the sums run in
    // the order of the phase form (matvec_part), the results are bit-identical to it.
    const int l = ctx.tid;
    constexpr int NCX = (NX + 3) / 4, NCU = (NUT + 3) / 4;
    const double dxl = l < NX ? dx[l] : 0.0, xl = l < NX ? x[l] : 0.0;
    const double ul = l < NU ? u[l] : 0.0, pel = l < NU ? q[QP_PE + l] : 0.0, kvl = l < NUT ? (ut_in ? ut_in[l] : rk[RIC_KV + l]) : 0.0;
    const double qvl = (info && l < NX) ? q[QP_QV + l] : 0.0, rvl = (info && l < NUT) ? q[QP_RV + l] : 0.0;
    double kk[2][NCX], px[3][NCX], pu[3][NCU];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int it = l + 64 * j, r = it >> 2, pp = it & 3;
#pragma unroll
      for (int c = 0; c < NCX; ++c) { const int cc = pp + 4 * c; kk[j][c] = (!ut_in && it < NUT * 4 && cc < NX) ? rk[RIC_K + r * NX + cc] : 0.0; }
    }
    const double fjl = (fj_in && l < NJ) ? fj_in[l] : 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int it = l + 64 * j, r = it >> 2, pp = it & 3;
      const bool rowl = it < NU * 4 && (!fj_in || r < 12);   // (with fj_in only the twelve wrench rows are formed here: the first pass of 64 lanes)
#pragma unroll
      for (int c = 0; c < NCX; ++c) { const int cc = pp + 4 * c; px[j][c] = 0.0; if (!fj_in || j == 0) px[j][c] = (rowl && cc < NX) ? q[QP_PX + r * NX + cc] : 0.0; }
#pragma unroll
      for (int c = 0; c < NCU; ++c) { const int cc = pp + 4 * c; pu[j][c] = 0.0; if (!fj_in || j == 0) pu[j][c] = (rowl && cc < NUT) ? q[QP_PU + r * NUT + cc] : 0.0; }
    }
    if (l < NX) { w.dx[l] = dxl; x_new[l] = xl + alpha * dxl; }
    WG_SYNC(ctx);
    if (ut_in) {   // (wave-uniform)
      if (l < NUT) { w.ut[l] = kvl; if (ut_out != ut_in) ut_out[l] = kvl; }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int it = l + 64 * j, pp = it & 3;
        if (it < NUT * 4) {
          double sacc = 0.0;
#pragma unroll
          for (int c = 0; c < NCX; ++c) { const int cc = pp + 4 * c; if (cc < NX) sacc += kk[j][c] * w.dx[cc]; }
          w.part[it] = sacc;
        }
      }
      WG_SYNC(ctx);
      if (l < NUT) { const double v = kvl + ((w.part[4 * l] + w.part[4 * l + 1]) + (w.part[4 * l + 2] + w.part[4 * l + 3])); w.ut[l] = v; ut_out[l] = v; }
    }
    WG_SYNC(ctx);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int it = l + 64 * j, pp = it & 3;
      if (it < NU * 4) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int c = 0; c < NCX; ++c) { const int cc = pp + 4 * c; if (cc < NX) s1 += px[j][c] * w.dx[cc]; }
#pragma unroll
        for (int c = 0; c < NCU; ++c) { const int cc = pp + 4 * c; if (cc < NUT) s2 += pu[j][c] * w.ut[cc]; }
        w.part[it] = s1 + s2;
      }
    }
    WG_SYNC(ctx);
    if (fj_in && l < NJ) w.t[l] = fjl;
    WG_SYNC(ctx);
    if (l < NU) {
      const double sv = (fj_in && l >= 12) ? pel + w.t[l - 12] : pel + ((w.part[4 * l] + w.part[4 * l + 1]) + (w.part[4 * l + 2] + w.part[4 * l + 3]));
      du_out[l] = sv; u_new[l] = ul + alpha * sv; w.du[l] = sv;
    }
    WG_SYNC(ctx);
    if (info) {
      // (the phase form sums these serially in one lane each; the same serial order here, from LDS copies of q~, r~)
      if (l < NX) w.part[l] = qvl;
      if (l < NUT) w.part[NX + l] = rvl;
      WG_SYNC(ctx);
      if (l < 3) {
        double sacc = 0.0;
        if (l == 0) { for (int i = 0; i < NX; ++i) sacc += w.part[i] * w.dx[i]; for (int j = 0; j < NUT; ++j) sacc += w.part[NX + j] * w.ut[j]; }
        else if (l == 1) { for (int i = 0; i < NX; ++i) sacc += w.dx[i] * w.dx[i]; }
        else { for (int i = 0; i < NU; ++i) sacc += w.du[i] * w.du[i]; }
        info[l] = sacc;
      }
    }
    return;
  }
#endif
  WG_FOR(ctx, i, NX) { w.dx[i] = dx[i]; x_new[i] = x[i] + alpha * dx[i]; }
  WG_SYNC(ctx);
  if (ut_in) {
    WG_FOR(ctx, i, NUT) w.ut[i] = ut_in[i];
  } else {
    WG_FOR(ctx, it, NUT * 4) w.part[it] = matvec_part<NX>(rk + RIC_K + (it >> 2) * NX, w.dx, it & 3);
    WG_SYNC(ctx);
    WG_FOR(ctx, i, NUT) w.ut[i] = rk[RIC_KV + i] + ((w.part[4 * i] + w.part[4 * i + 1]) + (w.part[4 * i + 2] + w.part[4 * i + 3]));
  }
  WG_SYNC(ctx);
  WG_FOR(ctx, it, NU * 4 + NUT) {
    if (it < NU * 4) {
      const int r = it >> 2, p = it & 3;
      w.part[it] = matvec_part<NX>(q + QP_PX + r * NX, w.dx, p) + matvec_part<NUT>(q + QP_PU + r * NUT, w.ut, p);
    } else {
      ut_out[it - NU * 4] = w.ut[it - NU * 4];
    }
  }
  WG_SYNC(ctx);
  WG_FOR(ctx, r, NU) {
    const double s = (fj_in && r >= 12) ? q[QP_PE + r] + fj_in[r - 12] : q[QP_PE + r] + ((w.part[4 * r] + w.part[4 * r + 1]) + (w.part[4 * r + 2] + w.part[4 * r + 3]));
    du_out[r] = s;
    u_new[r] = u[r] + alpha * s;
    w.du[r] = s;
  }
  WG_SYNC(ctx);
  if (info) {
    WG_FOR(ctx, it, 3) {
      double s = 0.0;
      if (it == 0) {
        for (int i = 0; i < NX; ++i) s += q[QP_QV + i] * w.dx[i];
        for (int j = 0; j < NUT; ++j) s += q[QP_RV + j] * w.ut[j];
      } else if (it == 1) {
        for (int i = 0; i < NX; ++i) s += w.dx[i] * w.dx[i];
      } else {
        for (int i = 0; i < NU; ++i) s += w.du[i] * w.du[i];
      }
      info[it] = s;
    }
  }
}

// ---- filter line search (ocs2::FilterLinesearch::acceptStep + the back-tracking loop of ocs2::SqpSolver::takeStep;
//      upstream ocs2_sqp, restated from the published algorithm: ASSUMPTIONS A5/A6 of the oracle header)
struct LsSettings { double g_max, g_min, gamma_c, armijo_factor, alpha_decay, alpha_min, delta_tol; };
struct LsState {       // per instance
  double alpha;        // step length of the current trial (final: accepted length, 0 for a zero step)
  double armijo;       // descent metric of the full step
  double dxnorm, dunorm;
  int active;          // the trial at `alpha` still has to be evaluated / decided
  int dirty;           // x_new, u_new have to be recomputed for `alpha`
  int step_type;
  int trials;
};
HSQP_HD double ls_violation(const hsqp_perf& p) { return sqrt(p.dynamics_sse + p.equality_sse); }
// decision for one evaluated trial; returns true if the line search of this instance is finished
HSQP_HD bool ls_decide(const LsSettings& st, const hsqp_perf& base, const hsqp_perf& trial, LsState& s) {
  const double g = ls_violation(base), gn = ls_violation(trial);
  const double am = s.alpha * s.armijo;
  bool accepted;
  int type;
  if (gn > st.g_max) { accepted = gn < (1.0 - st.gamma_c) * g; type = HSQP_STEP_CONSTRAINT; }
  else if (gn < st.g_min && g < st.g_min && am < 0.0) { accepted = trial.merit < base.merit + st.armijo_factor * am; type = HSQP_STEP_COST; }
  else { accepted = trial.merit < base.merit - st.gamma_c * g || gn < (1.0 - st.gamma_c) * g; type = HSQP_STEP_DUAL; }
  s.trials += 1;
  if (accepted) { s.step_type = type; s.active = 0; s.dirty = 0; return true; }
  s.alpha *= st.alpha_decay;
  const bool tiny = s.alpha * s.dunorm < st.delta_tol && s.alpha * s.dxnorm < st.delta_tol;
  if (tiny || !(s.alpha >= st.alpha_min)) { s.alpha = 0.0; s.step_type = HSQP_STEP_ZERO; s.active = 0; s.dirty = 1; return true; }
  s.dirty = 1;
  return false;
}

// KKT residual of the projected QP at (dx, ut) for one node k (parallel over the nodes), with the costates of the Riccati
// value function lam_k = S_k dx_k + s_k (the oracle's definition):
//   stationarity: | Q~ dx + P~^T ut + q~ + A~^T lam+ - lam |_inf ,  | R~ ut + P~ dx + r~ + B~^T lam+ |_inf
//   primal:       | dx+ - A~ dx - B~ ut - b~ |_inf   (node 0 also | dx_0 - (x_init - x_0) |)
// out2 = {stationarity, primal} of this node.
struct KktWS {
  double dx[NX], dxn[NX], ut[NUT], lam[NX], lamn[NX], res[2 * NX + NUT];
  double part[4 * (2 * NX + NUT)];   // four partial sums per row (short chains: the kernel is latency-bound at small batches)
};
HSQP_HD void kkt_node(const Ctx& ctx, KktWS& w, const double* q, const double* vfk, const double* vfn, const double* dxk, const double* dxn,
                      const double* utk, const double* dx0 /*null unless k == 0*/, double* out2) {
  WG_FOR(ctx, i, 2 * NX + NUT) {
    if (i < NX) w.dx[i] = dxk[i];
    else if (i < 2 * NX) w.dxn[i - NX] = dxn[i - NX];
    else w.ut[i - 2 * NX] = utk[i - 2 * NX];
  }
  WG_SYNC(ctx);
  // lam = S dx + s, lam+ = S+ dx+ + s+ : item = (row, quarter of the columns); S is stored symmetric to the bit, so the column read
  // vf[j][r] is used (consecutive rows in consecutive lanes: coalesced)
  constexpr int LQ = (NX + 3) / 4, LU = (NUT + 3) / 4;
  WG_FOR(ctx, it, 4 * 2 * NX) {
    const int p = it / (2 * NX), i = it % (2 * NX);
    const bool nxt = i >= NX;
    const int r = nxt ? i - NX : i;
    const double* vf = nxt ? vfn : vfk;
    const double* d = nxt ? w.dxn : w.dx;
    double l = p == 0 ? vf[NX * NX + r] : 0.0;
#pragma unroll
    for (int t = 0; t < LQ; ++t) { const int j = p * LQ + t; if (j < NX) l += vf[j * NX + r] * d[j]; }
    w.part[4 * i + p] = l;
  }
  WG_SYNC(ctx);
  WG_FOR(ctx, i, 2 * NX) {
    const double* pp = &w.part[4 * i];
    const double l = (pp[0] + pp[1]) + (pp[2] + pp[3]);
    if (i >= NX) w.lamn[i - NX] = l; else w.lam[i] = l;
  }
  WG_SYNC(ctx);
  WG_FOR(ctx, it, 4 * (2 * NX + NUT)) {
    const int p = it / (2 * NX + NUT), i = it % (2 * NX + NUT);
    double a = 0.0;
    if (i < NX) {              // x-stationarity (Q~ symmetric to the bit: coalesced column reads)
      if (p == 0) a = q[QP_QV + i] - w.lam[i];
#pragma unroll
      for (int t = 0; t < LQ; ++t) { const int j = p * LQ + t; if (j < NX) a += q[QP_Q + j * NX + i] * w.dx[j] + q[QP_A + j * NX + i] * w.lamn[j]; }
#pragma unroll
      for (int t = 0; t < LU; ++t) { const int j = p * LU + t; if (j < NUT) a += q[QP_P + j * NX + i] * w.ut[j]; }
    } else if (i < NX + NUT) { // u-stationarity
      const int r = i - NX;
      if (p == 0) a = q[QP_RV + r];
#pragma unroll
      for (int t = 0; t < LQ; ++t) { const int j = p * LQ + t; if (j < NX) a += q[QP_P + r * NX + j] * w.dx[j] + q[QP_B + j * NUT + r] * w.lamn[j]; }
#pragma unroll
      for (int t = 0; t < LU; ++t) { const int j = p * LU + t; if (j < NUT) a += q[QP_R + j * NUT + r] * w.ut[j]; }
    } else {                   // dynamics
      const int r = i - NX - NUT;
      if (p == 0) a = w.dxn[r] - q[QP_BV + r];
#pragma unroll
      for (int t = 0; t < LQ; ++t) { const int j = p * LQ + t; if (j < NX) a -= q[QP_A + r * NX + j] * w.dx[j]; }
#pragma unroll
      for (int t = 0; t < LU; ++t) { const int j = p * LU + t; if (j < NUT) a -= q[QP_B + r * NUT + j] * w.ut[j]; }
    }
    w.part[4 * i + p] = a;
  }
  WG_SYNC(ctx);
  WG_FOR(ctx, i, 2 * NX + NUT) {
    const double* pp = &w.part[4 * i];
    double a = (pp[0] + pp[1]) + (pp[2] + pp[3]);
    if (i >= NX + NUT && dx0) { const int r = i - NX - NUT; const double a0 = fabs(w.dx[r] - dx0[r]); a = (a0 != a0 || a0 > fabs(a)) ? a0 : fabs(a); }
    w.res[i] = fabs(a);
  }
  WG_SYNC(ctx);
  WG_FOR(ctx, it, 2) {
    double m = 0.0;
    // fmax() drops NaN: a non-finite residual must surface (as +inf, which also wins the atomic max on the bit patterns)
    const int lo = it == 0 ? 0 : NX + NUT, hi = it == 0 ? NX + NUT : 2 * NX + NUT;
    for (int i = lo; i < hi; ++i) { const double r = w.res[i]; m = (r != r) ? HUGE_VAL : fmax(m, r); }
    out2[it] = m;
  }
  WG_SYNC(ctx);
}

}  // namespace hsqp
