// v_mfma_f64_16x16x4_f64 fed from LDS, as in the X^T Y tile jobs (hsqp_linalg.h): cycles per matrix instruction for one wave per
// SIMD (4 waves, each with two 16x16 output tiles of a 58 x 58 x 58 product), for different ways of ordering the operand
// loads against the matrix instructions.  Answers whether the tile loops are limited by the LDS round trip.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int N = 58, LD = 58, STEPS = 14;   // 14 full k-steps (56 rows)

// V = 0: loads and matrix instructions in program order, the compiler schedules
// V = 1: operands of step s+1 loaded before the matrix instructions of step s (two register sets, order pinned)
// V = 2: look-ahead of two steps (three register sets)
// V = 3: as 0, but the two tiles share the column operand (3 loads per 2 matrix instructions)
template <int V>
__global__ __launch_bounds__(256) void k(double* out, long long* ticks, int reps) {
  __shared__ double X[N + 6][LD], Y[N + 6][LD];
  for (int i = threadIdx.x; i < (N + 6) * LD; i += blockDim.x) { (&X[0][0])[i] = 1e-3 * (i % 97); (&Y[0][0])[i] = 1e-3 * (i % 89); }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kk = lane >> 4;
  const int r0 = 16 * (wave & 1), r1 = r0 + 32 > N - 16 ? N - 16 : r0 + 32, c0 = 16 * (wave >> 1), c1 = V == 3 ? c0 : c0 + 32 > N - 16 ? N - 16 : c0 + 32;
  d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  const double* xa = &X[kk][r0 + i];
  const double* xb = &X[kk][r1 + i];
  const double* ya = &Y[kk][c0 + i];
  const double* yb = &Y[kk][c1 + i];
  const long long t0 = clock64();
  for (int rep = 0; rep < reps; ++rep) {
    if (V == 0) {
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[4 * s * LD], ya[4 * s * LD], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[4 * s * LD], yb[4 * s * LD], acc1, 0, 0, 0);
      }
    } else if (V == 3) {
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        const double b = ya[4 * s * LD];
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[4 * s * LD], b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[4 * s * LD], b, acc1, 0, 0, 0);
      }
    } else if (V == 1) {
      double a0 = xa[0], b0 = ya[0], a1 = xb[0], b1 = yb[0];
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        const int sn = s + 1 < STEPS ? s + 1 : s;
        const double na0 = xa[4 * sn * LD], nb0 = ya[4 * sn * LD], na1 = xb[4 * sn * LD], nb1 = yb[4 * sn * LD];
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a0 = na0; b0 = nb0; a1 = na1; b1 = nb1;
      }
    } else {
      double a0 = xa[0], b0 = ya[0], a1 = xb[0], b1 = yb[0];
      double p0 = xa[4 * LD], q0 = ya[4 * LD], p1 = xb[4 * LD], q1 = yb[4 * LD];
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        const int sn = s + 2 < STEPS ? s + 2 : s;
        const double na0 = xa[4 * sn * LD], nb0 = ya[4 * sn * LD], na1 = xb[4 * sn * LD], nb1 = yb[4 * sn * LD];
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a0 = p0; b0 = q0; a1 = p1; b1 = q1;
        p0 = na0; q0 = nb0; p1 = na1; q1 = nb1;
      }
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc0[3] + acc1[1] + acc1[2];
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int V>
void run(const char* what) {
  double* out; long long* ticks;
  (void)hipMalloc(&out, 256 * 256 * 8); (void)hipMalloc(&ticks, 8);
  const int reps = 2000;
  k<V><<<256, 256>>>(out, ticks, reps);
  k<V><<<256, 256>>>(out, ticks, reps);
  (void)hipDeviceSynchronize();
  long long h; (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
  printf("%-70s %6.1f cycles per v_mfma_f64_16x16x4_f64\n", what, (double)h / (reps * 2.0 * STEPS));
  (void)hipFree(out); (void)hipFree(ticks);
}
int main() {
  run<0>("program order, compiler-scheduled (4 loads per 2 MFMA)");
  run<1>("look-ahead 1 step, pinned");
  run<2>("look-ahead 2 steps, pinned");
  run<3>("program order, shared column operand (3 loads per 2 MFMA)");
  return 0;
}
