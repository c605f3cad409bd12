// Micro-benchmark (GPU box): FP64 VALU FMA rate, FP64 MFMA (v_mfma_f64_16x16x4_f64) rate, LDS ds_read_b64 rate.
// Used to calibrate the FP64 roofline that DESIGN.md / bench.py quote (the local guides do not list FP64 peaks).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void k_fma(double* out, int iters, long long* ticks) {
  double a0 = threadIdx.x, a1 = 1.0, a2 = 2.0, a3 = 3.0, a4 = 4.0, a5 = 5.0, a6 = 6.0, a7 = 7.0;
  const double b = 1.0000001, c = 0.5;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c);
    a4 = fma(a4, b, c); a5 = fma(a5, b, c); a6 = fma(a6, b, c); a7 = fma(a7, b, c);
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

__global__ void k_mfma(double* out, int iters, long long* ticks) {
  double4_t c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

// MFMA layout check: C = A (16x4) * B (4x16) with A[i][k] = i + 10 k, B[k][j] = (k + 1) * (j + 1) + (j == 3)
__global__ void k_mfma_layout(double* C) {
  const int l = threadIdx.x;
  const int i = l & 15, k = l >> 4;
  const double a = i + 10.0 * k;                       // A[i = l&15][k = l>>4]
  const double b = (k + 1.0) * ((l & 15) + 1.0) + ((l & 15) == 3 ? 1.0 : 0.0);   // B[k = l>>4][j = l&15]
  double4_t c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];   // guide: row = (lane>>4) + 4*reg, col = lane&15
}

__global__ void k_lds(double* out, int iters, long long* ticks) {
  __shared__ double buf[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) buf[i] = i;
  __syncthreads();
  double s = 0.0;
  int idx = threadIdx.x;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    s += buf[idx & 8191] + buf[(idx + 64) & 8191] + buf[(idx + 128) & 8191] + buf[(idx + 192) & 8191] +
         buf[(idx + 256) & 8191] + buf[(idx + 320) & 8191] + buf[(idx + 384) & 8191] + buf[(idx + 448) & 8191];
    idx += 512;
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
  double* out; long long* ticks;
  hipMalloc(&out, 1 << 26); hipMalloc(&ticks, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int threads : {256, 512, 1024}) {
    for (int which = 0; which < 3; ++which) {
      for (int blocks : {1, 256, 1024}) {
        float ms; long long t;
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          if (which == 0) hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(threads), 0, 0, out, iters, ticks);
          if (which == 1) hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(threads), 0, 0, out, iters, ticks);
          if (which == 2) hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(threads), 0, 0, out, iters, ticks);
          hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        const double waves = threads / 64.0, ops = which == 1 ? 4.0 : 8.0;
        const double flop = (double)blocks * threads * iters * (which == 0 ? 16.0 : 0.0) + (which == 1 ? (double)blocks * waves * iters * 4.0 * 2048.0 : 0.0);
        printf("%s threads %4d blocks %4d: %8.3f ms, %9lld ticks (block 0) -> %.2f ticks per wave-instr-slot per SIMD-resident wave, %.2f GHz-equiv, %.1f TFLOP/s%s\n",
               which == 0 ? "fma_f64 " : (which == 1 ? "mfma_f64" : "lds_b64 "), threads, blocks, ms, t, (double)t / (iters * ops), t / (ms * 1e6), flop / (ms * 1e-3) / 1e12,
               which == 2 ? " (8 ds_read_b64 per iter per lane)" : "");
      }
    }
  }
  std::vector<double> C(256);
  double* dC; hipMalloc(&dC, 256 * 8);
  hipLaunchKernelGGL(k_mfma_layout, dim3(1), dim3(64), 0, 0, dC);
  hipMemcpy(C.data(), dC, 256 * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    double ref = 0; for (int k = 0; k < 4; ++k) ref += (i + 10.0 * k) * ((k + 1.0) * (j + 1.0) + (j == 3 ? 1.0 : 0.0));
    if (C[i * 16 + j] != ref) ++bad;
  }
  printf("mfma_f64_16x16x4 layout check (A[l&15][l>>4], B[l>>4][l&15], C row=(l>>4)+4r col=l&15): %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
  return 0;
}
