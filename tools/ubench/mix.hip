// Issue-mix micro-benchmark: the exact per-dword instruction mix of k_cprod (7 selector
// ops + 8 v_perm + NM MFMAs), no memory traffic.  Reports ns per 1024 genotypes per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NM, bool DECODE, int TWOSETS>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned seed, unsigned lutA, unsigned lutB) {
  unsigned w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = seed * (2 * i + 1) + threadIdx.x * 2654435761u;
  v4i b0 = {(int)w[0], (int)w[1], (int)w[2], (int)w[3]}, b1 = {(int)w[4], (int)w[5], (int)w[6], (int)w[7]};
  v4i acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const unsigned x = w[i];
      v4i g, n;
      if (DECODE) {
        const unsigned s0 = x & 0x03030303u, s1 = (x >> 2) & 0x03030303u, s2 = (x >> 4) & 0x03030303u, s3 = (x >> 6) & 0x03030303u;
        g = v4i{(int)__builtin_amdgcn_perm(lutA, lutA, s0), (int)__builtin_amdgcn_perm(lutA, lutA, s1),
                (int)__builtin_amdgcn_perm(lutA, lutA, s2), (int)__builtin_amdgcn_perm(lutA, lutA, s3)};
        n = v4i{(int)__builtin_amdgcn_perm(lutB, lutB, s0), (int)__builtin_amdgcn_perm(lutB, lutB, s1),
                (int)__builtin_amdgcn_perm(lutB, lutB, s2), (int)__builtin_amdgcn_perm(lutB, lutB, s3)};
      } else {
        g = v4i{(int)x, (int)w[(i + 1) & 7], (int)w[(i + 2) & 7], (int)w[(i + 3) & 7]};
        n = v4i{(int)w[(i + 4) & 7], (int)x, (int)w[(i + 5) & 7], (int)w[(i + 6) & 7]};
      }
      if (NM >= 1) acc[(2 * i) & (TWOSETS ? 3 : 1)] = __builtin_amdgcn_mfma_i32_16x16x64_i8(g, b0, acc[(2 * i) & (TWOSETS ? 3 : 1)], 0, 0, 0);
      if (NM >= 2) acc[(2 * i + 1) & (TWOSETS ? 3 : 1)] = __builtin_amdgcn_mfma_i32_16x16x64_i8(n, b0, acc[(2 * i + 1) & (TWOSETS ? 3 : 1)], 0, 0, 0);
      if (NM >= 4) {
        acc[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(g, b1, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(n, b1, acc[3], 0, 0, 0);
      }
      if (NM == 0) asm volatile("" ::"v"(g), "v"(n));
    }
    w[it & 7] += 0x01010101u;
  }
  unsigned r = acc[0][0] ^ acc[1][1] ^ acc[2][2] ^ acc[3][3];
  if (r == 0x12345678u) out[threadIdx.x] = r;
}

template <int NACC, bool DECODE>
__global__ __launch_bounds__(256) void k32(unsigned *out, int iters, unsigned seed, unsigned lutA, unsigned lutB) {
  unsigned w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = seed * (2 * i + 1) + threadIdx.x * 2654435761u;
  v4i b0 = {(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
  v16i acc[4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[a][r] = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const unsigned x = w[i];
      v4i g, n;
      if (DECODE) {
        const unsigned s0 = x & 0x03030303u, s1 = (x >> 2) & 0x03030303u, s2 = (x >> 4) & 0x03030303u, s3 = (x >> 6) & 0x03030303u;
        g = v4i{(int)__builtin_amdgcn_perm(lutA, lutA, s0), (int)__builtin_amdgcn_perm(lutA, lutA, s1),
                (int)__builtin_amdgcn_perm(lutA, lutA, s2), (int)__builtin_amdgcn_perm(lutA, lutA, s3)};
        n = v4i{(int)__builtin_amdgcn_perm(lutB, lutB, s0), (int)__builtin_amdgcn_perm(lutB, lutB, s1),
                (int)__builtin_amdgcn_perm(lutB, lutB, s2), (int)__builtin_amdgcn_perm(lutB, lutB, s3)};
      } else {
        g = v4i{(int)x, (int)w[(i + 1) & 7], (int)w[(i + 2) & 7], (int)w[(i + 3) & 7]};
        n = v4i{(int)w[(i + 4) & 7], (int)x, (int)w[(i + 5) & 7], (int)w[(i + 6) & 7]};
      }
      acc[(2 * i) % NACC] = __builtin_amdgcn_mfma_i32_32x32x32_i8(g, b0, acc[(2 * i) % NACC], 0, 0, 0);
      acc[(2 * i + 1) % NACC] = __builtin_amdgcn_mfma_i32_32x32x32_i8(n, b0, acc[(2 * i + 1) % NACC], 0, 0, 0);
    }
    w[it & 7] += 0x01010101u;
  }
  unsigned r = acc[0][0] ^ acc[1][1] ^ acc[2][2] ^ acc[3][3];
  if (r == 0x12345678u) out[threadIdx.x] = r;
}

template <int NACC, bool DECODE>
void run32(const char *name, int waves) {
  unsigned *d; CK(hipMalloc(&d, 4096));
  const int iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k32<NACC, DECODE>), dim3(256 * waves), dim3(256), 0, 0, d, 10, 1u, 0x00010002u, 0x00000100u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k32<NACC, DECODE>), dim3(256 * waves), dim3(256), 0, 0, d, iters, 1u, 0x00010002u, 0x00000100u);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double per = ms * 1e6 / ((double)iters * 8 * waves);
  printf("%-40s waves/SIMD %d: %7.3f ms  %6.2f ns per 1024 genotypes per SIMD -> %5.2f TB/s-equivalent\n", name, waves, ms, per,
         1024.0 * 1024 / 4 / per / 1e3);
  CK(hipFree(d));
}

template <int NM, bool DECODE, int TWOSETS>
void run(const char *name, int waves) {
  unsigned *d; CK(hipMalloc(&d, 4096));
  const int iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<NM, DECODE, TWOSETS>), dim3(256 * waves), dim3(256), 0, 0, d, 10, 1u, 0x00010002u, 0x00000100u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<NM, DECODE, TWOSETS>), dim3(256 * waves), dim3(256), 0, 0, d, iters, 1u, 0x00010002u, 0x00000100u);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double per = ms * 1e6 / ((double)iters * 8 * waves);
  printf("%-40s waves/SIMD %d: %7.3f ms  %6.2f ns per 1024 genotypes per SIMD -> %5.2f TB/s-equivalent\n", name, waves, ms, per,
         1024.0 * 1024 / 4 / per / 1e3);
  CK(hipFree(d));
}

int main() {
  for (int w : {1, 2, 3, 4}) {
    run32<2, false>("2 mfma32 only (2 accs)", w);
    run32<4, false>("2 mfma32 only (4 accs)", w);
    run32<2, true>("decode + 2 mfma32 (2 accs)", w);
    run32<4, true>("decode + 2 mfma32 (4 accs)", w);
  }
  for (int w : {6}) {
    run<0, true, 1>("decode only (7 sel + 8 perm)", w);
    run<2, false, 1>("2 mfma16 only", w);
    run<2, true, 1>("decode + 2 mfma16 (NB=1)", w);
    run<4, false, 1>("4 mfma16 only", w);
    run<4, true, 1>("decode + 4 mfma16 (NB=2)", w);
  }
  return 0;
}
