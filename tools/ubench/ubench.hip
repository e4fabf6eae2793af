// Micro-benchmarks of the gfx950 issue rates that bound the streaming kernels:
// VALU (v_perm / shift / and), i8 MFMA 16x16x64 and 32x32x32, and their mixes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned seed, unsigned long long *clk) {
  unsigned long long t_begin = clock64();
  unsigned w = seed + threadIdx.x * 2654435761u, acc0 = 0;
  v4i a = {(int)w, (int)(w * 3), (int)(w * 5), (int)(w * 7)}, b = a;
  v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  v16i d0 = {0}, d1 = {0};
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {  // 16 dependent-free VALU: perm/shift/and mix like the decode
#pragma unroll
      for (int r = 0; r < 4; r++) {
        unsigned s0 = w & 0x03030303u, s1 = (w >> 2) & 0x03030303u;
        unsigned p0 = __builtin_amdgcn_perm(0x00010002u, 0x00010002u, s0);
        unsigned p1 = __builtin_amdgcn_perm(0x00000100u, 0x00000100u, s1);
        acc0 += p0 ^ p1;
        w = w * 1664525u + 1013904223u;
      }
    } else if (MODE == 1) {  // 4 independent MFMA 16x16x64 i8
      c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
    } else if (MODE == 2) {  // 2 independent MFMA 32x32x32 i8
      d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d1, 0, 0, 0);
    } else if (MODE == 3) {  // decode of one dword (15 VALU) + 4 MFMA 16x16x64 (NB=2 shape)
      unsigned s0 = w & 0x03030303u, s1 = (w >> 2) & 0x03030303u, s2 = (w >> 4) & 0x03030303u, s3 = (w >> 6) & 0x03030303u;
      v4i g = {(int)__builtin_amdgcn_perm(0x00010002u, 0x00010002u, s0), (int)__builtin_amdgcn_perm(0x00010002u, 0x00010002u, s1),
               (int)__builtin_amdgcn_perm(0x00010002u, 0x00010002u, s2), (int)__builtin_amdgcn_perm(0x00010002u, 0x00010002u, s3)};
      v4i n = {(int)__builtin_amdgcn_perm(0x00000100u, 0x00000100u, s0), (int)__builtin_amdgcn_perm(0x00000100u, 0x00000100u, s1),
               (int)__builtin_amdgcn_perm(0x00000100u, 0x00000100u, s2), (int)__builtin_amdgcn_perm(0x00000100u, 0x00000100u, s3)};
      c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(g, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(g, a, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(n, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(n, a, c3, 0, 0, 0);
      w = w * 1664525u + 1013904223u;
    } else if (MODE == 4) {  // decode + 2 MFMA 32x32x32
      unsigned s0 = w & 0x03030303u, s1 = (w >> 2) & 0x03030303u, s2 = (w >> 4) & 0x03030303u, s3 = (w >> 6) & 0x03030303u;
      v4i g = {(int)__builtin_amdgcn_perm(0x00010002u, 0x00010002u, s0), (int)__builtin_amdgcn_perm(0x00010002u, 0x00010002u, s1),
               (int)__builtin_amdgcn_perm(0x00010002u, 0x00010002u, s2), (int)__builtin_amdgcn_perm(0x00010002u, 0x00010002u, s3)};
      v4i n = {(int)__builtin_amdgcn_perm(0x00000100u, 0x00000100u, s0), (int)__builtin_amdgcn_perm(0x00000100u, 0x00000100u, s1),
               (int)__builtin_amdgcn_perm(0x00000100u, 0x00000100u, s2), (int)__builtin_amdgcn_perm(0x00000100u, 0x00000100u, s3)};
      d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(g, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(n, b, d1, 0, 0, 0);
      w = w * 1664525u + 1013904223u;
    } else if (MODE == 5) {  // decode only (15 VALU), results kept alive
      unsigned s0 = w & 0x03030303u, s1 = (w >> 2) & 0x03030303u, s2 = (w >> 4) & 0x03030303u, s3 = (w >> 6) & 0x03030303u;
      acc0 += __builtin_amdgcn_perm(0x00010002u, 0x00010002u, s0) ^ __builtin_amdgcn_perm(0x00010002u, 0x00010002u, s1) ^
              __builtin_amdgcn_perm(0x00010002u, 0x00010002u, s2) ^ __builtin_amdgcn_perm(0x00010002u, 0x00010002u, s3);
      acc0 += __builtin_amdgcn_perm(0x00000100u, 0x00000100u, s0) ^ __builtin_amdgcn_perm(0x00000100u, 0x00000100u, s1) ^
              __builtin_amdgcn_perm(0x00000100u, 0x00000100u, s2) ^ __builtin_amdgcn_perm(0x00000100u, 0x00000100u, s3);
      w = w * 1664525u + 1013904223u;
    }
  }
  unsigned r = acc0 ^ c0[0] ^ c1[1] ^ c2[2] ^ c3[3] ^ d0[0] ^ d1[5] ^ w;
  if (r == 0x12345678u) out[threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = clock64() - t_begin;
}

template <int MODE>
void run(const char *name, int waves_per_simd, double genos_per_iter_per_wave) {
  unsigned *d; CK(hipMalloc(&d, 4096));
  unsigned long long *dclk; CK(hipMalloc(&dclk, 8));
  int iters = 20000;
  int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 1u, dclk);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u, dclk);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double ns_per_iter = ms * 1e6 / iters / waves_per_simd;   // per wave-iteration per SIMD
  double genos = genos_per_iter_per_wave * 1024.0 * waves_per_simd * iters;  // whole chip
  printf("%-34s waves/SIMD %d: %8.3f ms  %7.2f ns per wave-iter per SIMD", name, waves_per_simd, ms, ns_per_iter);
  if (genos_per_iter_per_wave > 0) printf("  -> %.2f Tgeno/s = %.2f TB/s of 2-bit image", genos / ms / 1e9, genos / 4 / ms / 1e9);
  unsigned long long hc = 0; CK(hipMemcpy(&hc, dclk, 8, hipMemcpyDeviceToHost));
  printf("  [shader clock %.0f MHz]\n", hc / (ms * 1e3));
  CK(hipFree(d));
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("16 VALU-ish (perm/and/shift/mul)", w, 0);
    run<5>("decode only (15 VALU + lcg)", w, 1024);
    run<1>("4x mfma_i32_16x16x64_i8", w, 0);
    run<2>("2x mfma_i32_32x32x32_i8", w, 0);
    run<3>("decode + 4x mfma16 (NB=2)", w, 1024);
    run<4>("decode + 2x mfma32 (NB=2)", w, 1024);
  }
  return 0;
}
