// VALU issue-rate micro-benchmark on gfx950: ns and (estimated) cycles per wave64 instruction
// for the instruction kinds the decode uses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned seed) {
  unsigned r[16];
#pragma unroll
  for (int i = 0; i < 16; i++) r[i] = seed * (i + 1) + threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 4; rep++)
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (KIND == 0) r[i] = r[i] & (0x03030303u + rep + i);                       // v_and_b32
        if (KIND == 1) r[i] = (r[i] >> 2) | 0x40000000u;                             // v_lshrrev (+or folded?)
        if (KIND == 2) r[i] = __builtin_amdgcn_perm(0x00010002u, r[(i + 1) & 15], r[i] & 0x03030303u) | 0x01000000u;  // and + perm + or
        if (KIND == 3) r[i] = __builtin_amdgcn_perm(r[(i + 3) & 15], r[(i + 1) & 15], 0x05010400u);   // perm only
        if (KIND == 4) r[i] = r[i] * 0x9E3779B1u;                                    // v_mul_lo_u32
        if (KIND == 5) r[i] = __builtin_amdgcn_alignbit(r[(i + 1) & 15], r[i], 2);   // v_alignbit_b32
        if (KIND == 6) { float f = __uint_as_float(r[i]); f = __builtin_fmaf(f, 1.0001f, 0.5f); r[i] = __float_as_uint(f); }  // v_fma_f32
        if (KIND == 7) r[i] = (r[i] >> 2) & 0x03030303u;                             // shift + and (2 ops, or v_bfe?)
      }
  }
  unsigned a = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) a ^= r[i];
  if (a == 0x12345678u) out[threadIdx.x] = a;
}

template <int KIND>
void run(const char *name, int ops_per_inner) {
  unsigned *d; CK(hipMalloc(&d, 4096));
  const int iters = 4000, waves = 4;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<KIND>, dim3(256 * waves), dim3(256), 0, 0, d, 10, 1u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<KIND>, dim3(256 * waves), dim3(256), 0, 0, d, iters, 1u);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double n_inst = (double)iters * 64 * ops_per_inner * waves;  // wave-instructions per SIMD
  printf("%-28s %8.3f ms   %.3f ns per wave-instruction per SIMD (4 waves/SIMD)\n", name, ms, ms * 1e6 / n_inst);
  CK(hipFree(d));
}

int main() {
  run<0>("v_and_b32", 1);
  run<1>("v_lshrrev|or", 1);
  run<2>("and + v_perm_b32 + or (3)", 3);
  run<3>("v_perm_b32", 1);
  run<4>("v_mul_lo_u32", 1);
  run<5>("v_alignbit_b32", 1);
  run<6>("v_fma_f32", 1);
  run<7>("shift + and (2)", 2);
  return 0;
}
