// What the fp64 matrix pipe delivers when nothing else runs: v_mfma_f64_16x16x4_f64 back to back on NACC
// independent accumulators, W waves per SIMD.  (k_tcross is priced against this and against the 78.6 TFLOP/s spec.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, int iters, double x, double y) {
  v4d acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) acc[i] = v4d{0, 0, 0, 0};
  double a = x + threadIdx.x, b = y - threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;
}

template <int NACC>
void run(int wg_per_cu, double *out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000, blocks = 256 * wg_per_cu;
  hipLaunchKernelGGL((k<NACC>), dim3(blocks), dim3(256), 0, 0, out, 100, 1.0, 2.0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<NACC>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 2.0);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)blocks * 4 * iters * NACC * 2048.0;
  printf("%2d accumulators, %d waves per SIMD: %8.2f ms  %6.1f TFLOP/s\n", NACC, wg_per_cu, ms, flops / ms / 1e9);
}

typedef int v4i __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k8(int *out, int iters, int x) {
  v4i acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) acc[i] = v4i{0, 0, 0, 0};
  v4i a = {x + (int)threadIdx.x, x, x + 1, x + 2}, b = {x - (int)threadIdx.x, x, x, x};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123456789) out[0] = s;
}
template <int NACC>
void run8(int wg_per_cu, int *out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 40000, blocks = 256 * wg_per_cu;
  hipLaunchKernelGGL((k8<NACC>), dim3(blocks), dim3(256), 0, 0, out, 100, 1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k8<NACC>), dim3(blocks), dim3(256), 0, 0, out, iters, 1);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double ops = (double)blocks * 4 * iters * NACC * 2.0 * 16 * 16 * 64;
  printf("int8 16x16x64: %2d accumulators, %d waves per SIMD: %8.2f ms  %6.0f TOP/s\n", NACC, wg_per_cu, ms, ops / ms / 1e9);
}

int main() {
  double *out; CK(hipMalloc(&out, 64));
  run<4>(1, out); run<8>(1, out); run<16>(1, out);
  run<8>(2, out); run<8>(4, out); run<16>(2, out);
  int *o8 = (int *)out;
  run8<8>(1, o8); run8<8>(2, o8); run8<8>(4, o8); run8<16>(2, o8);
  return 0;
}
