// probe of replicate_rows01 (hsqp_riccati.h): lane l holds 1000 + l; expected lo = 1000 + (l & 15), hi = 1016 + (l & 15) in every lane
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void replicate_rows01(double v, double& lo, double& hi) {
  const long long b = __double_as_longlong(v);
  int out[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int w = h == 0 ? (int)(b & 0xffffffffll) : (int)(b >> 32);
    int a0 = w, a1 = w;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a0), "+v"(a1));
    int c0 = a0, c1 = a0, d0 = a1, d1 = a1;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c0), "+v"(c1));
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(d0), "+v"(d1));
    out[0][h] = c0; out[1][h] = d0;
  }
  lo = __longlong_as_double(((long long)out[0][1] << 32) | (unsigned int)out[0][0]);
  hi = __longlong_as_double(((long long)out[1][1] << 32) | (unsigned int)out[1][0]);
}
__global__ void k(double* out) {
  const int l = threadIdx.x;
  double lo, hi;
  replicate_rows01(1000.0 + l * 1.25, lo, hi);
  out[l] = lo; out[64 + l] = hi;
  double d = 0.0, x = 2.0;
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(lo), "v"(x));
  out[128 + l] = d;
}
int main() {
  double* d; (void)hipMalloc(&d, 192 * sizeof(double));
  hipLaunchKernelGGL(k, 1, 64, 0, 0, d);
  double h[192]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    if (h[l] != 1000.0 + (l & 15) * 1.25) ++bad;
    if (h[64 + l] != 1000.0 + (16 + (l & 15)) * 1.25) ++bad;
    if (h[128 + l] != 2.0 * (1000.0 + 5 * 1.25)) ++bad;
  }
  printf("bad %d; lo:", bad); for (int l = 0; l < 64; l += 9) printf(" %g", h[l]); printf("; hi:"); for (int l = 0; l < 64; l += 9) printf(" %g", h[64 + l]); printf("; fmac: %g %g\n", h[128], h[128 + 40]);
  return bad != 0;
}
