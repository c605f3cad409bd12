// v_mfma_f64_16x16x4_f64: cycles per instruction from ONE wave as a function of the number of independent accumulator
// chains it interleaves (1..6).  Answers how many output tiles a wave must hold to keep the FP64 matrix pipe busy.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NC>
__global__ void k(double* out, long long* ticks, int iters) {
  d4 c[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) c[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NC; ++i) s += c[i][0] + c[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int NC>
void run() {
  double* out; long long* ticks;
  (void)hipMalloc(&out, 256 * 64 * 8); (void)hipMalloc(&ticks, 8);
  const int iters = 4000;
  k<NC><<<256, 64>>>(out, ticks, iters);
  k<NC><<<256, 64>>>(out, ticks, iters);
  (void)hipDeviceSynchronize();
  long long h; (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
  printf("%d independent accumulator chains: %6.1f cycles per v_mfma_f64_16x16x4_f64 (one wave per SIMD)\n", NC, (double)h / (iters * (double)NC));
  (void)hipFree(out); (void)hipFree(ticks);
}
int main() { run<1>(); run<2>(); run<3>(); run<4>(); run<6>(); return 0; }
