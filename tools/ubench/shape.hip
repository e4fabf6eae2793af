// Access-shape micro-benchmark: how fast can a wave stream a variant-major image when each
// 16-B-per-lane load instruction covers ROWS rows x (1024/ROWS) contiguous bytes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// each wave owns RT rows (RT multiple of ROWS) for the whole pitch; ROWS rows per instruction
template <int ROWS, int RT, int UNROLL>
__global__ __launch_bounds__(256) void k(const uint8_t *img, int64_t pitch, int64_t m, unsigned *out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int SEG = 1024 / ROWS;          // contiguous bytes per row per instruction
  constexpr int LPR = SEG / 16;             // lanes per row
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * RT;
  if (row0 >= m) return;
  unsigned acc = 0;
  const int rsub = lane / LPR, cl = lane % LPR;
  for (int64_t off = 0; off < pitch; off += SEG * UNROLL) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
#pragma unroll
      for (int rg = 0; rg < RT / ROWS; rg++) {
        int64_t r = row0 + rg * ROWS + rsub;
        if (r >= m) r = m - 1;
        const uint4 v = *(const uint4 *)(img + r * pitch + off + u * SEG + cl * 16);
        acc += v.x ^ v.y ^ v.z ^ v.w;
      }
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int ROWS, int RT, int UNROLL>
void run(const uint8_t *img, int64_t pitch, int64_t m, unsigned *out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int64_t blocks = (m + 4 * RT - 1) / (4 * RT);
  hipLaunchKernelGGL((k<ROWS, RT, UNROLL>), dim3((unsigned)blocks), dim3(256), 0, 0, img, pitch, m, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<ROWS, RT, UNROLL>), dim3((unsigned)blocks), dim3(256), 0, 0, img, pitch, m, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  printf("rows/instr %2d (%4d B contiguous)  rows/wave %2d  unroll %d: %7.3f ms  %6.0f GB/s\n", ROWS, 1024 / ROWS, RT, UNROLL, ms,
         (double)pitch * m / ms / 1e6);
}


// The same 16-rows-x-64-B instruction shape on a TILED image: blocks of 64 rows x 256 B stored contiguously
// ([row block][256-B column block][row in block][byte]).  A workgroup of WAVES waves owns 64 * WAVES * RT / 64 ...
// here: wave w of workgroup b owns rows (b * WAVES + w) * RT .. + RT - 1 (RT = 16 or 32) and walks along the
// row; consecutive instructions of a wave now touch one 16-KB tile instead of RT pages 100 KB apart.
template <int RT, int UNROLL>
__global__ __launch_bounds__(512) void kt(const uint8_t *img, int64_t pitch, int64_t m, unsigned *out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 8 + wave) * RT;
  if (row0 >= m) return;
  const int64_t SB = pitch / 256;
  unsigned acc = 0;
  const int rsub = lane >> 2, cl = lane & 3;   // 16 rows x 64 B per instruction
  for (int64_t off = 0; off < pitch; off += 64 * UNROLL) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
#pragma unroll
      for (int rg = 0; rg < RT / 16; rg++) {
        int64_t r = row0 + rg * 16 + rsub;
        if (r >= m) r = m - 1;
        const int64_t o = off + u * 64 + cl * 16;
        const uint4 v = *(const uint4 *)(img + (((r >> 6) * SB + (o >> 8)) * 64 + (r & 63)) * 256 + (o & 255));
        acc += v.x ^ v.y ^ v.z ^ v.w;
      }
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// plain layout with the same workgroup shape (8 waves x RT rows), for a like-for-like comparison
template <int RT, int UNROLL>
__global__ __launch_bounds__(512) void kp(const uint8_t *img, int64_t pitch, int64_t m, unsigned *out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 8 + wave) * RT;
  if (row0 >= m) return;
  unsigned acc = 0;
  const int rsub = lane >> 2, cl = lane & 3;
  for (int64_t off = 0; off < pitch; off += 64 * UNROLL) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
#pragma unroll
      for (int rg = 0; rg < RT / 16; rg++) {
        int64_t r = row0 + rg * 16 + rsub;
        if (r >= m) r = m - 1;
        const uint4 v = *(const uint4 *)(img + r * pitch + off + u * 64 + cl * 16);
        acc += v.x ^ v.y ^ v.z ^ v.w;
      }
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
template <int RT, int UNROLL, bool TILED>
void run2(const uint8_t *img, int64_t pitch, int64_t m, unsigned *out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int64_t blocks = (m + 8 * RT - 1) / (8 * RT);
  auto launch = [&] {
    if (TILED) hipLaunchKernelGGL((kt<RT, UNROLL>), dim3((unsigned)blocks), dim3(512), 0, 0, img, pitch, m, out);
    else hipLaunchKernelGGL((kp<RT, UNROLL>), dim3((unsigned)blocks), dim3(512), 0, 0, img, pitch, m, out);
  };
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 3; i++) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  printf("%s layout, 8 waves x %2d rows, 16 rows x 64 B per instruction, unroll %d: %7.3f ms  %6.0f GB/s\n",
         TILED ? "tiled" : "plain", RT, UNROLL, ms, (double)pitch * m / ms / 1e6);
}

// k_prod's shape: a workgroup of 4 waves owns a 256-B column block (1024 samples) and walks down the rows 64
// at a time; lane = (row group g = lane >> 4, 4-byte word sg = lane & 15 of the wave's 64 B), 16 dword
// loads per lane and step (rows g * 16 + r).  TILED: the step is one contiguous 16-KB tile.
template <bool TILED>
__global__ __launch_bounds__(256) void kq(const uint8_t *img, int64_t pitch, int64_t m, int64_t rows_per_slab,
                                          unsigned *out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sg = lane & 15, g = lane >> 4;
  const int64_t SB = pitch / 256, sb = blockIdx.x;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
  int64_t r1 = r0 + rows_per_slab;
  if (r1 > m) r1 = m;
  unsigned acc = 0;
  for (int64_t rb = r0; rb < r1; rb += 64) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int64_t row = rb + g * 16 + r;
      const uint8_t *p = TILED ? img + (((row >> 6) * SB + sb) * 64 + (row & 63)) * 256 + wave * 64 + sg * 4
                               : img + row * pitch + sb * 256 + wave * 64 + sg * 4;
      acc += *(const unsigned *)p;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
template <bool TILED>
void run3(const uint8_t *img, int64_t pitch, int64_t m, unsigned *out, int ky) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int64_t rps = ((m / 64 + ky - 1) / ky) * 64;
  dim3 grid((unsigned)(pitch / 256), (unsigned)((m + rps - 1) / rps));
  hipLaunchKernelGGL((kq<TILED>), grid, dim3(256), 0, 0, img, pitch, m, rps, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((kq<TILED>), grid, dim3(256), 0, 0, img, pitch, m, rps, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  printf("%s layout, k_prod shape (4 waves x 64 rows x 64 B, dword loads), %d row slabs: %7.3f ms  %6.0f GB/s\n",
         TILED ? "tiled" : "plain", ky, ms, (double)pitch * m / ms / 1e6);
}

int main() {
  const int64_t n = 400000, m = 300000, pitch = 100096;
  uint8_t *img; unsigned *out;
  CK(hipMalloc(&img, (size_t)pitch * (m + 64))); CK(hipMalloc(&out, 64));
  CK(hipMemset(img, 0x5A, (size_t)pitch * (m + 64)));
  (void)n;
  run<1, 1, 4>(img, pitch, m, out);
  run<1, 4, 1>(img, pitch, m, out);
  run<4, 4, 4>(img, pitch, m, out);
  run<4, 16, 1>(img, pitch, m, out);
  run<16, 16, 2>(img, pitch, m, out);
  run<16, 16, 4>(img, pitch, m, out);
  run<16, 32, 2>(img, pitch, m, out);
  run<16, 32, 4>(img, pitch, m, out);
  run<32, 32, 2>(img, pitch, m, out);
  run<64, 64, 2>(img, pitch, m, out);
  run2<16, 2, false>(img, pitch, m, out);
  run2<32, 2, false>(img, pitch, m, out);
  run2<32, 4, false>(img, pitch, m, out);
  run2<16, 2, true>(img, pitch, m, out);
  run2<32, 2, true>(img, pitch, m, out);
  run2<32, 4, true>(img, pitch, m, out);
  const int64_t m64 = m / 64 * 64;
  run3<false>(img, pitch, m64, out, 11);
  run3<true>(img, pitch, m64, out, 11);
  run3<false>(img, pitch, m64, out, 22);
  run3<true>(img, pitch, m64, out, 22);
  return 0;
}
