// round 5: v_mfma_scale_f32_16x16x128_f8f6f4 with FP4 (E2M1) operands and unit scales as an EXACT small-integer
// matrix pipe — the question the fp4 variant of the LD kernel (ld.hip) rests on.  Checks, on the device:
//  (1) operand / result layout: lane l of A <-> row l & 15, K-group l >> 4 (32 consecutive nibbles per lane, nibble e at
//      bit 4 e of the lane's 128 bits); B likewise with columns; D: col = lane & 15, row = 4 (lane >> 4) + r.  The
//      product is invariant under any K permutation applied to A and B alike, so what matters is that A and B pair up
//      lanes of equal l >> 4 and elements of equal position — which this test pins by comparing with the host sum;
//  (2) exactness of the fp32 accumulation for integer sums up to 2^24 (values 0, 1, 2, 4: products up to 16);
//  (3) throughput: dependent and independent issue, against v_mfma_i32_16x16x64_i8.
// build: hipcc --offload-arch=gfx950 -O2 fp4_mfma.hip -o fp4_mfma ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_once(const unsigned *A, const unsigned *B, float *D, int reps) {
  const int l = threadIdx.x;
  v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int w = 0; w < 4; w++) { a[w] = (int)A[l * 4 + w]; b[w] = (int)B[l * 4 + w]; }
  v4f c = {0, 0, 0, 0};
  for (int r = 0; r < reps; r++)
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  for (int r = 0; r < 4; r++) D[l * 4 + r] = c[r];
}
template <int FP4>
__global__ __launch_bounds__(256) void k_rate(float *out, int iters) {
  v8i a = {(int)threadIdx.x, 2, 3, 4, 0, 0, 0, 0}, b = {5, 6, 7, (int)blockIdx.x, 0, 0, 0, 0};
  v4f c[8];
  v4i ci[8];
  for (int t = 0; t < 8; t++) { c[t] = v4f{0, 0, 0, 0}; ci[t] = v4i{0, 0, 0, 0}; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int t = 0; t < 8; t++) {
      if (FP4) c[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[t], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      else ci[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(v4i{a[0], a[1], a[2], a[3]}, v4i{b[0], b[1], b[2], b[3]}, ci[t], 0, 0, 0);
    }
  }
  float s = 0;
  for (int t = 0; t < 8; t++) s += FP4 ? c[t][0] : (float)ci[t][0];
  if (s == 12345.f) out[0] = s;
}
static const int val[4] = {0, 1, 2, 4};
static const unsigned nib[4] = {0x0, 0x2, 0x4, 0x6};   // E2M1: 0, 1.0, 2.0, 4.0
int main() {
  std::vector<int> Am(16 * 128), Bm(128 * 16);
  srand(7);
  for (auto &x : Am) x = rand() & 3;
  for (auto &x : Bm) x = rand() & 3;
  std::vector<unsigned> A(64 * 4, 0), B(64 * 4, 0);
  for (int l = 0; l < 64; l++)
    for (int e = 0; e < 32; e++) {
      const int k = (l >> 4) * 32 + e;
      A[l * 4 + e / 8] |= nib[Am[(l & 15) * 128 + k]] << (4 * (e & 7));
      B[l * 4 + e / 8] |= nib[Bm[k * 16 + (l & 15)]] << (4 * (e & 7));
    }
  unsigned *dA, *dB; float *dD;
  CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 1024));
  CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
  for (int reps : {1, 1000, 20000}) {
    hipLaunchKernelGGL(k_once, dim3(1), dim3(64), 0, 0, dA, dB, dD, reps);
    std::vector<float> D(256);
    CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    int bad = 0; double mx = 0;
    for (int l = 0; l < 64; l++)
      for (int r = 0; r < 4; r++) {
        const int i = 4 * (l >> 4) + r, j = l & 15;
        long long s = 0;
        for (int k = 0; k < 128; k++) s += (long long)val[Am[i * 128 + k]] * val[Bm[k * 16 + j]];
        s *= reps;
        if ((double)D[l * 4 + r] != (double)s) bad++;
        if (s > mx) mx = (double)s;
      }
    printf("reps %d: %d of 256 results differ from the exact integer sums (largest sum %.0f, 2^24 = 16777216)\n", reps, bad, mx);
  }
  float *dO; CK(hipMalloc(&dO, 4));
  for (int fp4 = 0; fp4 < 2; fp4++) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000, blocks = 256 * 8;
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0));
      if (fp4) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, dO, iters);
      else hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, dO, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double)blocks * 4 * iters * 8;
    const double ops = n * 16 * 16 * (fp4 ? 128 : 64) * 2;
    printf("%s: %.2f ms, %.0f T(FL)OP/s, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", fp4 ? "fp4 16x16x128 (scaled, unit scales)" : "i8 16x16x64",
           ms, ops / ms / 1e9, ms * 1e-3 * 2.4e9 / (n / 1024));
  }
  return 0;
}
