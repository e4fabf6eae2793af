// Infinity-Cache residency micro-benchmark (round 4, VERDICT r3 #1b): the decode + MFMA mix of k_cprod<2> (per loaded
// dword: 7 selector ops, 4 v_perm, 4 x v_mfma_i32_16x16x64_i8) fed by a stream of 16-B loads that sweeps a FOOTPRINT of
// R bytes over and over — R = 128 MB stays in the 256-MB Infinity Cache after the first sweep, R = 16 GB comes from HBM —
// and the same loads without the compute.  One workgroup of 16 waves per CU, every wave reads contiguous 4-KB pieces
// (64 lanes x 16 B x 4 loads) in a grid-strided order, TOTAL bytes the same for every footprint.
// Prints GB/s (and, from rocm-smi during a longer run, the clock: tools/gpu/r04_l3.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <bool COMPUTE, bool VARYB>
__global__ __launch_bounds__(1024) void k(const uint4 *__restrict__ buf, size_t pieces_in_footprint, size_t pieces_per_wave,
                                          unsigned *out, unsigned lutB) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 16;
  v4i acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  // VARYB: eight different digit operands cycle through the K-steps (random bytes: the operand toggling of real digit
  // panels); otherwise two constant ones (no toggling between MFMAs: what a 0/1 panel does to the real kernels)
  v4i bb[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    unsigned h = (unsigned)(lane * 2654435761u) + 0x9e3779b9u * (VARYB ? i + 1 : (i & 1) + 1);
    for (int c = 0; c < 4; c++) { h = h * 1664525u + 1013904223u; bb[i][c] = (int)h; }
  }
  unsigned x = 0;
  // two register sets, as in the kernels: the loads of piece i + 1 fly while piece i is consumed
  uint4 ga[2][4];
  size_t p = wave % pieces_in_footprint;
#pragma unroll
  for (int l = 0; l < 4; l++) ga[0][l] = buf[p * 256 + l * 64 + lane];
  for (size_t i = 0; i < pieces_per_wave; i += 2) {
#pragma unroll
    for (int set = 0; set < 2; set++) {
      const size_t pn = (p + nwaves + 1) % pieces_in_footprint;   // (stride co-prime with the footprint: a piece comes back after a whole sweep)
#pragma unroll
      for (int l = 0; l < 4; l++) ga[set ^ 1][l] = buf[pn * 256 + l * 64 + lane];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int l = 0; l < 4; l++) {
        const uint32_t w4[4] = {ga[set][l].x, ga[set][l].y, ga[set][l].z, ga[set][l].w};
#pragma unroll
        for (int d = 0; d < 4; d++) {
          const uint32_t w = w4[d];
          if (COMPUTE) {
            const uint32_t s0 = w & 0x03030303u, s1 = (w >> 2) & 0x03030303u, s2 = (w >> 4) & 0x03030303u, s3 = (w >> 6) & 0x03030303u;
            const v4i a0 = {(int)s0, (int)s1, (int)s2, (int)s3};
            const v4i a1 = {(int)__builtin_amdgcn_perm(lutB, lutB, s0), (int)__builtin_amdgcn_perm(lutB, lutB, s1),
                            (int)__builtin_amdgcn_perm(lutB, lutB, s2), (int)__builtin_amdgcn_perm(lutB, lutB, s3)};
            const v4i b0 = bb[((l * 4 + d) & 3) * 2], b1 = bb[((l * 4 + d) & 3) * 2 + 1];
            acc[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, acc[3], 0, 0, 0);
          } else {
            x ^= w;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      p = pn;
    }
  }
  const unsigned r = x ^ (unsigned)(acc[0][0] ^ acc[1][1] ^ acc[2][2] ^ acc[3][3]);
  if (r == 0x12345679u) out[0] = r;
}

__global__ void fill(uint32_t *p, size_t n) {
  // genotype-like codes: 2-bit fields, ~1 % missing (code 3)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 7), w = 0;
    for (int e = 0; e < 16; e++) {
      h = h * 1664525u + 1013904223u;
      const uint32_t r = h >> 24;
      w |= (r < 3 ? 3u : r < 140 ? 0u : r < 220 ? 1u : 2u) << (2 * e);
    }
    p[i] = w;
  }
}

int main(int argc, char **argv) {
  const double total_gb = argc > 1 ? atof(argv[1]) : 100.0;
  const double foot_gb[] = {0.0625, 0.125, 0.1875, 0.5, 16.0};
  const size_t big = (size_t)(16.0 * (1ull << 30));
  uint4 *buf; unsigned *out;
  CK(hipMalloc(&buf, big)); CK(hipMalloc(&out, 64));
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint32_t *)buf, big / 4);
  CK(hipDeviceSynchronize());
  int dev = 0; hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, dev));
  const int ncu = pr.multiProcessorCount;
  for (int pass = 0; pass < 3; pass++)
    for (double fg : foot_gb) {
      const size_t foot_pieces = (size_t)(fg * (1ull << 30)) / 4096;
      const size_t nwaves = (size_t)ncu * 16;
      size_t ppw = (size_t)(total_gb * 1e9 / 4096 / nwaves) & ~(size_t)1;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int rep = 0; rep < 2; rep++) {   // rep 0 warms (and fills the cache), rep 1 is timed
        CK(hipEventRecord(e0));
        if (pass == 0) hipLaunchKernelGGL((k<false, false>), dim3(ncu), dim3(1024), 0, 0, buf, foot_pieces, ppw, out, 0x01000000u);
        else if (pass == 1) hipLaunchKernelGGL((k<true, false>), dim3(ncu), dim3(1024), 0, 0, buf, foot_pieces, ppw, out, 0x01000000u);
        else hipLaunchKernelGGL((k<true, true>), dim3(ncu), dim3(1024), 0, 0, buf, foot_pieces, ppw, out, 0x01000000u);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double bytes = (double)ppw * 4096 * nwaves;
      printf("%s footprint %8.1f MB  %7.2f ms per %.0f GB  %6.0f GB/s\n", pass == 0 ? "loads only                                   " : pass == 1 ? "loads + decode + 4 MFMA, constant digit operands" : "loads + decode + 4 MFMA, random digit operands  ",
             fg * 1024, ms * 100e9 / bytes, 100.0, bytes / ms / 1e6);
      fflush(stdout);
    }
  return 0;
}
