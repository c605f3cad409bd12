// Cost of one workgroup-parallel "phase" (LDS read -> short f64 chain -> LDS write -> barrier) on gfx950, in shader
// cycles per phase, for the workgroup sizes the solver kernels use.  One workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void k(double* out, long long* ticks, int iters) {
  __shared__ double a[2048];
  const int t = threadIdx.x;
  for (int i = t; i < 2048; i += blockDim.x) a[i] = 1.0 + 1e-3 * i;
  __syncthreads();
  const long long t0 = clock64();
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {                       // barrier only
      __syncthreads();
    } else if (MODE == 1) {                // read, fma, write, barrier
      const double x = a[(t + it) & 2047], y = a[(t * 3 + it) & 2047];
      a[t] = x - 0.5 * y;
      __syncthreads();
    } else if (MODE == 2) {                // + reciprocal chain (elimination step)
      const double p = a[it & 2047], f = a[(it + 7) & 2047];
      const double x = a[(t + it) & 2047], y = a[(t * 3 + it) & 2047];
      double r = __builtin_amdgcn_rcp(p);
      r = r * (2.0 - p * r);
      r = r * (2.0 - p * r);
      a[t] = x - f * r * y;
      __syncthreads();
    } else if (MODE == 3) {                // + rsqrt and rcp chain (Householder scalars)
      const double p = a[it & 2047], f = a[(it + 7) & 2047];
      const double x = a[(t + it) & 2047], y = a[(t * 3 + it) & 2047];
      const double n = p * rsqrt(p);
      const double hv = p + n * f;
      double r = __builtin_amdgcn_rcp(hv);
      r = r * (2.0 - hv * r);
      r = r * (2.0 - hv * r);
      a[t] = x - r * y;
      __syncthreads();
    } else if (MODE == 4) {                // 9-long dependent dot
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < 9; ++q) s += a[(t + 64 * q + it) & 2047] * a[(t * 3 + q + it) & 2047];
      a[t] = s * 1e-3;
      __syncthreads();
    } else if (MODE == 5) {                // only wave 0 works, everyone meets at the barrier
      if (t < 64) { const double x = a[(t + it) & 2047], y = a[(t * 3 + it) & 2047]; a[t] = x - 0.5 * y; }
      __syncthreads();
    } else if (MODE == 6) {                // wave-local step: no workgroup barrier (only wave 0 active)
      if (t < 64) {
        const double x = a[(t + it) & 2047], y = a[(t * 3 + it) & 2047];
        a[t] = x - 0.5 * y;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }
  const long long t1 = clock64();
  acc += a[t];
  if (t == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
  out[blockIdx.x * blockDim.x + t] = acc;
}

template <int MODE>
void run(const char* name, int threads) {
  double* out; long long* ticks;
  hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&ticks, 8);
  const int iters = 2000;
  k<MODE><<<256, threads>>>(out, ticks, iters);
  k<MODE><<<256, threads>>>(out, ticks, iters);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
  printf("%-44s threads %4d : %7.1f cycles / phase\n", name, threads, (double)h / iters);
  hipFree(out); hipFree(ticks);
}

int main() {
  for (int th : {64, 128, 256, 512, 1024}) {
    run<0>("barrier only", th);
    run<1>("read-fma-write-barrier", th);
    run<2>("read-rcp chain-fma-write-barrier", th);
    run<3>("read-rsqrt+rcp chain-fma-write-barrier", th);
    run<4>("9-long dot-write-barrier", th);
    run<5>("wave0 works, all barrier", th);
    run<6>("wave0 wave-local step (no barrier)", th);
  }
  return 0;
}
