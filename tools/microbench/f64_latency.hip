// Dependent-chain latency (shader cycles per operation, ONE wave on a CU) of the FP64 operations the serial phases
// of the solver kernels are made of.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k(double* out, long long* ticks, int iters, double seed) {
  double x = seed + threadIdx.x * 1e-9, y = 1.0 + seed * 1e-3;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) x = fma(x, y, 1e-9);
      else if (MODE == 1) x = x * y;
      else if (MODE == 2) x = x + y;
      else if (MODE == 3) x = __builtin_amdgcn_rcp(x) + 1.0;
      else if (MODE == 4) x = __builtin_amdgcn_rsq(x) + 1.0;
      else if (MODE == 5) x = rsqrt(x) + 1.0;
      else if (MODE == 6) x = sqrt(x) + 1.0;
      else if (MODE == 7) x = 1.0 / x + 1.0;
      else if (MODE == 8) { double s, c; sincos(x, &s, &c); x = s + c; }
      else if (MODE == 9) { double r = __builtin_amdgcn_rcp(x); r = r * (2.0 - x * r); r = r * (2.0 - x * r); x = r + 1.0; }
      else if (MODE == 10) { float f = (float)x; f = f * 1.0001f + 1e-3f; x = (double)f; }
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

template <int MODE>
void run(const char* name, int extra_ops) {
  double* out; long long* ticks;
  (void)hipMalloc(&out, 256 * 64 * 8); (void)hipMalloc(&ticks, 8);
  const int iters = 2000;
  k<MODE><<<256, 64>>>(out, ticks, iters, 1.25);
  k<MODE><<<256, 64>>>(out, ticks, iters, 1.25);
  (void)hipDeviceSynchronize();
  long long h; (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
  printf("%-44s %7.1f cycles per dependent op (incl. %d add)\n", name, (double)h / (iters * 8.0), extra_ops);
  (void)hipFree(out); (void)hipFree(ticks);
}

int main() {
  run<0>("v_fma_f64", 0);
  run<1>("v_mul_f64", 0);
  run<2>("v_add_f64", 0);
  run<3>("v_rcp_f64 (+add)", 1);
  run<4>("v_rsq_f64 (+add)", 1);
  run<5>("rsqrt() ocml (+add)", 1);
  run<6>("sqrt() ocml (+add)", 1);
  run<7>("1.0 / x (+add)", 1);
  run<8>("sincos() ocml (+add)", 1);
  run<9>("rcp + 2 Newton steps (+add)", 1);
  run<10>("f64->f32, f32 fma, f32->f64", 0);
  return 0;
}
