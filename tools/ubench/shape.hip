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
  return 0;
}
