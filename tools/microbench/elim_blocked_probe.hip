// The Riccati stage's factorisation alone on an idle GPU: the column-by-column in-register elimination of round 3 (eliminate_begin +
// eliminate_end, elim_columnwise.h) against the blocked matrix-core form (eliminate_blocked, hsqp_elim.h).  Shader-clock ticks per call for
// one wave (wave 0's role), both roles on two SIMDs, and — what the stage looks like — both roles next to waves that stream matrix
// instructions on the same / on the other SIMDs; plus the largest difference between the two forms' L^-1, Z, z.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include "elim_columnwise.h"
using namespace hsqp;
extern __shared__ double smem[];

// mode 0: column-by-column, 1: blocked.  threads: 64 / 128 (the eliminating waves) or 512 (waves 2..7 stream matrix instructions:
// load = 1 on SIMDs 2, 3 only (waves 2, 3, 6, 7), load = 2 on all four SIMDs (waves 2..7))
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(long long* out, double* res, int reps, int mode, int load) {
  RicWS& w = *reinterpret_cast<RicWS*>(smem);
  const int tid = threadIdx.x;
  for (int i = tid; i < NUT * LDF; i += blockDim.x) { const int r = i / LDF, c = i % LDF; w.fac.Ef[r][c] = r == c ? 4.0 + r : (c < NUT ? 0.01 * ((r * 7 + c * 3) % 11) : 0.0); }
  for (int i = tid; i < NUT * LDE; i += blockDim.x) w.Em[i / LDE][i % LDE] = 0.001 * (i % 97);
  for (int i = tid; i < NX * NX; i += blockDim.x) { w.S[i / NX][i % NX] = 1e-3 * (i % 13); w.A2[0][i / NX][i % NX] = 1e-3 * (i % 7); }
  __syncthreads();
  for (int i = tid; i < NUT * NUT; i += blockDim.x) { const int r = i / NUT, c = i % NUT; if (c > r) w.fac.Ef[c][r] = w.fac.Ef[r][c]; }
  __syncthreads();
  const long long t0 = clock64();
  double acc = 0.0;
  long long telim = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  for (int it = 0; it < reps; ++it) {
    int t2 = tid;
    asm volatile("" : "+v"(t2));
    const int wv = t2 >> 6, lane = t2 & 63;
    const long long ts = clock64();
    if (wv < 2) {
      if (mode == 0) {
        double e[NUT];
        eliminate_begin<NX>(w, wv, lane, e, nullptr, nullptr);
        eliminate_end<NX>(w, wv, lane, e);
        acc += e[NUT - 1];
      } else {
        const DevWave dw{lane};
        const ElimIO io{&w.fac.Ef[0][0], LDF, &w.Em[0][EM_G], &w.Em[0][EM_GVP], LDE, &w.fac.Ef[0][EF_MI], LDF, &w.fac.LinvT[0][0], LDB, &w.Zs[0][0], LDZ, w.zv, &w.ok};
        if (wv == 0) eliminate_blocked<NX, 0>(dw, io); else eliminate_blocked<NX, 1>(dw, io);
      }
      telim += clock64() - ts;
    } else if (load && (load == 2 || (wv & 3) >= 2)) {
      const XtyJob jsa = xty_job(NX, NX, NX, &w.S[0][0], NX, &w.A2[0][0][0], NX, &w.SA[0][0], NX);
      const int nw = load == 2 ? 6 : 4, rank = load == 2 ? wv - 2 : (wv & 1) + (wv >> 2) * 2;
      xty_deal_one<0, RIC_PF>(jsa, 0, rank, nw, lane);
    }
    __syncthreads();
  }
#endif
  const long long t2 = clock64();
  if (tid == 0) { out[0] = (t2 - t0) / reps; out[1] = (long long)acc; out[2] = telim / reps; }
  __syncthreads();
  // results of the last call
  for (int i = tid; i < NUT * NUT; i += blockDim.x) res[i] = w.fac.Ef[i / NUT][EF_MI + i % NUT];
  for (int i = tid; i < NUT * (NX + 1); i += blockDim.x) res[NUT * NUT + i] = w.Zs[i / (NX + 1)][i % (NX + 1)];
}
int main() {
  long long* d; (void)hipMalloc(&d, 32);
  constexpr int NRES = NUT * NUT + NUT * (NX + 1);
  double* dr; (void)hipMalloc(&dr, NRES * sizeof(double));
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RicWS));
  static double r[2][NRES];
  for (int mode : {0, 1}) {
    for (int cfg = 0; cfg < 4; ++cfg) {
      const int threads = cfg == 0 ? 64 : (cfg == 1 ? 128 : 512), load = cfg < 2 ? 0 : cfg - 1;
      hipLaunchKernelGGL(k, 1, threads, sizeof(RicWS), 0, d, dr, 50, mode, load);
      long long h[3]; (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      printf("%s, %3d threads, matrix load %d: %lld ticks per phase, %lld for wave 0's elimination\n", mode ? "blocked      " : "column-by-col", threads, load, h[0], h[2]);
      if (cfg == 1) (void)hipMemcpy(r[mode], dr, sizeof(r[0]), hipMemcpyDeviceToHost);
    }
  }
  double dmax = 0.0, vmax = 0.0;
  for (int i = 0; i < NRES; ++i) { dmax = fmax(dmax, fabs(r[0][i] - r[1][i])); vmax = fmax(vmax, fabs(r[0][i])); }
  printf("max |blocked - column-by-column| = %.3e (max |value| %.3e)\n", dmax, vmax);
  return dmax <= 1e-12 * vmax ? 0 : 1;
}
