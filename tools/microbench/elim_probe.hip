// How long does the in-register elimination of [Lam | I | G | g] (elim_columnwise.h, eliminate_begin + eliminate_end) take by itself?
// One workgroup of 64 (wave 0's role) or 128 threads (both roles); shader-clock ticks around the call, a well-conditioned synthetic Lam.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "elim_columnwise.h"
using namespace hsqp;
extern __shared__ double smem[];
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(long long* out, int reps) {
  RicWS& w = *reinterpret_cast<RicWS*>(smem);
  const int tid = threadIdx.x;
  for (int i = tid; i < NUT * LDF; i += blockDim.x) { const int r = i / LDF, c = i % LDF; w.fac.Ef[r][c] = r == c ? 4.0 + r : (c < NUT ? 0.01 * ((r * 7 + c * 3) % 11) : 0.0); }
  for (int i = tid; i < NUT * LDE; i += blockDim.x) w.Em[i / LDE][i % LDE] = 0.001 * (i % 97);
  __syncthreads();
  // symmetrise Lam
  for (int i = tid; i < NUT * NUT; i += blockDim.x) { const int r = i / NUT, c = i % NUT; if (c > r) w.fac.Ef[c][r] = w.fac.Ef[r][c]; }
  __syncthreads();
  long long t0 = clock64();
  double acc = 0.0;
  for (int it = 0; it < reps; ++it) {
    int t2 = tid;
    asm volatile("" : "+v"(t2));
    double e[NUT];
#if defined(__HIP_DEVICE_COMPILE__)
    eliminate_begin<NX>(w, t2 >> 6, t2 & 63, e, nullptr, nullptr);
    eliminate_end<NX>(w, t2 >> 6, t2 & 63, e);
#endif
    acc += e[NUT - 1];
    __builtin_amdgcn_s_barrier();
  }
  long long t1 = clock64();
  if (tid == 0) { out[0] = (t1 - t0) / reps; out[1] = (long long)acc; }
}
int main() {
  long long* d; (void)hipMalloc(&d, 16);
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RicWS));
  for (int threads : {64, 128}) {
    hipLaunchKernelGGL(k, 1, threads, sizeof(RicWS), 0, d, 50);
    long long h[2]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%d threads: %lld ticks per elimination\n", threads, h[0]);
  }
  return 0;
}
