// gfx950 v_permlane16_swap / v_permlane32_swap semantics probe: prints, per lane, what the two operands hold after the swap.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int l = threadIdx.x;
  int a = l, b = 100 + l;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[l] = r[0]; out[64 + l] = r[1];
  auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[128 + l] = q[0]; out[192 + l] = q[1];
}
int main() {
  int* d; hipMalloc(&d, 256 * sizeof(int));
  hipLaunchKernelGGL(k, 1, 64, 0, 0, d);
  int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"swap16 first ", "swap16 second", "swap32 first ", "swap32 second"};
  for (int t = 0; t < 4; ++t) { printf("%s:", names[t]); for (int i = 0; i < 64; i += 4) printf(" %d", h[64 * t + i]); printf("\n"); }
  return 0;
}
