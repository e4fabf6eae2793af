// What separates k_cprod<2> (20.5 - 22.8 ms per 100 GB) from the same decode + MFMA mix on a plain contiguous stream
// (tools/ubench/l3.hip: 18.9 ms)?  The kernel's skeleton — 16 waves x 2 tiles of 16 variants, chunks of 512 samples, two
// register sets, 11 VALU + 8 MFMA per K-step pair — with two things switchable:
//   SHAPE 0: chunk-major image (the 512 rows x 128 B a workgroup reads per chunk are one contiguous 64-KB run)
//   SHAPE 1: variant-major image (rows `pitch` = 100 096 B apart: 16 rows x 64 B per load instruction)
//   LDSB  0: digit operands constant in registers (eight different ones cycling)
//   LDSB  1: the digit panel of every chunk staged through LDS by the workgroup (global load at the top of a chunk,
//            ds_write at its end, barrier) and read with two ds_read_b128 per K-step, as the kernel does
// 262 144 rows x 100 096 B (26.2 GB) per launch = 2 rounds of 256 workgroups; sustained over `reps` launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int SHAPE, int LDSB>
__global__ __launch_bounds__(1024) void k(const uint8_t *__restrict__ img, int64_t pitch, const uint4 *__restrict__ xq4,
                                          unsigned *out, unsigned lutB) {
  constexpr int XS = 1024;   // uint4 entries of a chunk's digit panel (32 blocks of 16 samples x 32 columns)
  __shared__ uint4 xs[2][XS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int nchunks = (int)(pitch / 128);
  const int64_t wg = blockIdx.x, nwg = gridDim.x;
  // byte address of (tile t, chunk ch, 64-B half it)
  auto addr = [&](int t, int ch, int it) -> const uint4 * {
    const int64_t row = wave * 32 + t * 16 + c;
    if (SHAPE == 0) return (const uint4 *)(img + ((int64_t)ch * nwg + wg) * 65536 + row * 128 + it * 64 + g * 16);
    return (const uint4 *)(img + (wg * 512 + row) * pitch + (int64_t)ch * 128 + it * 64 + g * 16);
  };
  v4i acc[2][2][2];
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
      for (int nb = 0; nb < 2; nb++) acc[t][p][nb] = v4i{0, 0, 0, 0};
  v4i bb[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    unsigned h = (unsigned)(lane * 2654435761u) + 0x9e3779b9u * (i + 1);
    for (int q = 0; q < 4; q++) { h = h * 1664525u + 1013904223u; bb[i][q] = (int)h; }
  }
  uint4 ga[2][2][2], xr = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int it = 0; it < 2; it++) { ga[0][t][it] = *addr(t, 0, it); ga[1][t][it] = *addr(t, nchunks > 1 ? 1 : 0, it); }
  if (LDSB) xs[0][tid] = xq4[tid];
  __syncthreads();
  auto chunk = [&](auto SETC, const int ch) {
    constexpr int SET = decltype(SETC)::value;
    const int ch1 = ch + 1 < nchunks ? ch + 1 : nchunks - 1, ch2 = ch + 2 < nchunks ? ch + 2 : nchunks - 1;
    if (LDSB) xr = xq4[(int64_t)ch1 * XS + tid];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < 2; it++)
#pragma unroll
      for (int d = 0; d < 4; d++) {
        v4i b[2];
        if (LDSB) {
#pragma unroll
          for (int nb = 0; nb < 2; nb++) {
            const uint4 v = xs[SET][(it * 16 + g * 4 + d) * 32 + nb * 16 + c];
            b[nb] = v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
          }
        } else {
          b[0] = bb[((it * 4 + d) & 3) * 2];
          b[1] = bb[((it * 4 + d) & 3) * 2 + 1];
        }
#pragma unroll
        for (int t = 0; t < 2; t++) {
          const uint32_t w = d == 0 ? ga[SET][t][it].x : d == 1 ? ga[SET][t][it].y : d == 2 ? ga[SET][t][it].z : ga[SET][t][it].w;
          const uint32_t s0 = w & 0x03030303u, s1 = (w >> 2) & 0x03030303u, s2 = (w >> 4) & 0x03030303u, s3 = (w >> 6) & 0x03030303u;
          const v4i a0 = {(int)s0, (int)s1, (int)s2, (int)s3};
          const v4i a1 = {(int)__builtin_amdgcn_perm(lutB, lutB, s0), (int)__builtin_amdgcn_perm(lutB, lutB, s1),
                          (int)__builtin_amdgcn_perm(lutB, lutB, s2), (int)__builtin_amdgcn_perm(lutB, lutB, s3)};
#pragma unroll
          for (int nb = 0; nb < 2; nb++) {
            acc[t][0][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b[nb], acc[t][0][nb], 0, 0, 0);
            acc[t][1][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b[nb], acc[t][1][nb], 0, 0, 0);
          }
        }
      }
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int it = 0; it < 2; it++) ga[SET][t][it] = *addr(t, ch2, it);
    __builtin_amdgcn_sched_barrier(0);
    if (LDSB) {
      xs[SET ^ 1][tid] = xr;
      __syncthreads();
    }
  };
  for (int ch = 0; ch < nchunks; ch += 2) {
    chunk(std::integral_constant<int, 0>{}, ch);
    if (ch + 1 < nchunks) chunk(std::integral_constant<int, 1>{}, ch + 1);
  }
  unsigned r = 0;
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
      for (int nb = 0; nb < 2; nb++) r ^= (unsigned)(acc[t][p][nb][0] ^ acc[t][p][nb][3]);
  if (r == 0x12345679u) out[0] = r;
}

__global__ void fill(uint32_t *p, size_t n, int genotypes) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 7), w = 0;
    if (genotypes) {
      for (int e = 0; e < 16; e++) {
        h = h * 1664525u + 1013904223u;
        const uint32_t r = h >> 24;
        w |= (r < 3 ? 3u : r < 140 ? 0u : r < 220 ? 1u : 2u) << (2 * e);
      }
    } else {
      w = h * 1664525u + 1013904223u;
    }
    p[i] = w;
  }
}

template <int SHAPE, int LDSB>
void run(const uint8_t *img, int64_t pitch, int64_t rows, const uint4 *xq, unsigned *out, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)(rows / 512);
  hipLaunchKernelGGL((k<SHAPE, LDSB>), dim3(grid), dim3(1024), 0, 0, img, pitch, xq, out, 0x01000000u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k<SHAPE, LDSB>), dim3(grid), dim3(1024), 0, 0, img, pitch, xq, out, 0x01000000u);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double bytes = (double)rows * pitch;
  printf("%-28s %-34s %7.2f ms per 100 GB  %6.0f GB/s\n", SHAPE == 0 ? "chunk-major (64-KB runs)" : "variant-major (rows 100 KB apart)",
         LDSB ? "digit panel through LDS + barrier" : "digit operands in registers", ms * 100e9 / bytes, bytes / ms / 1e6);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 12;
  const int64_t pitch = 100096, rows = 262144;
  uint8_t *img; uint4 *xq; unsigned *out;
  CK(hipMalloc(&img, (size_t)rows * pitch)); CK(hipMalloc(&xq, (size_t)(pitch / 128) * 1024 * 16)); CK(hipMalloc(&out, 64));
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint32_t *)img, (size_t)rows * pitch / 4, 1);
  hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, (uint32_t *)xq, (size_t)(pitch / 128) * 1024 * 4, 0);
  CK(hipDeviceSynchronize());
  for (int pass = 0; pass < 2; pass++) {
    run<0, 0>(img, pitch, rows, xq, out, reps);
    run<1, 0>(img, pitch, rows, xq, out, reps);
    run<0, 1>(img, pitch, rows, xq, out, reps);
    run<1, 1>(img, pitch, rows, xq, out, reps);
  }
  return 0;
}
