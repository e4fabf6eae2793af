// round 6: would the wide passes of the default solve (16 vectors x 24 bits = three int8 column blocks,
// k_cprod<3> / k_prodT<3>: 27 / 29 ms per 100 GB at the package power cap) be faster on the FP6 x FP4 form of
// v_mfma_scale_f32_16x16x128_f8f6f4?  The codes {0,1,2,3} and the missing flag {0,1} are FP4 (E2M1) numbers — a 2-bit
// code in the low bits of a nibble IS code / 2 in E2M1, so the code plane costs THREE instructions per 16 genotypes
// (w & 0x33.., (w >> 2) & 0x33..) and the missing plane two more per nibble dword (x & x >> 1) —, and a balanced base-32
// digit d in [-16, 16] is the E2M3 number d / 8 (sign-magnitude: the 5 magnitude bits ARE |d|).  A 25-bit panel is
// five digits = five column blocks at 16 pipe cycles per 128 samples (40 cycles per 64 samples against 48 for three
// int8 blocks), a 20-bit panel four (32).
// This file answers, on the device:
//  (1) layout and EXACTNESS of FP4 (A) x FP6 E2M3 (B) with unit scales: element e of a lane's A (nibble e) pairs with
//      element e of the same lane group's B (bits 6e .. 6e+5), random codes x random signed digits against the host's
//      integer sums, and sums driven towards 2^24 with and without cancellation;
//  (2) the rate of the mixed-format instruction against the FP4 x FP4 and int8 forms;
//  (3) the skeletons of the two streaming kernels — k_cprod's (ONE digit plane, two accumulators per tile: rows 100 KB
//      apart) and k_prodT's (TWO digit planes into one accumulator: chunk-major 64-KB runs) — with the int8 decode +
//      three column blocks against the FP6 form with four / five, sustained under the power cap, ms per 100 GB.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 fp6_parts.hip -o fp6_parts ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <type_traits>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// ---- (1) exactness ------------------------------------------------------------------------------------------------
__global__ void k_once(const unsigned *A, const unsigned *B, float *D, int reps) {
  const int l = threadIdx.x;
  v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int w = 0; w < 4; w++) a[w] = (int)A[l * 4 + w];
  for (int w = 0; w < 6; w++) b[w] = (int)B[l * 6 + w];
  v4f c = {0, 0, 0, 0};
  for (int r = 0; r < reps; r++)
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4 /* A: fp4 */, 2 /* B: fp6 e2m3 */, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  for (int r = 0; r < 4; r++) D[l * 4 + r] = c[r];
}
static unsigned e2m3(int d) { return (unsigned)((d < 0 ? 32 : 0) | (d < 0 ? -d : d)); }   // d / 8 for |d| <= 16

static int exactness() {
  int bad_total = 0;
  // mode 0: random codes x random digits in [-16, 15]; 1: all codes 3 x all digits +15 (the largest same-sign sum per
  // instruction: 128 * 45 = 5760); 2: codes 3 x digits alternating +16 / -16 by K (cancellation at full magnitude);
  // 3: codes 3 x digits -16
  for (int mode = 0; mode < 4; mode++) {
    std::vector<int> Am(16 * 128), Bm(128 * 16);
    srand(11 + mode);
    for (int i = 0; i < 16; i++)
      for (int k = 0; k < 128; k++) Am[i * 128 + k] = mode == 0 ? (rand() & 3) : 3;
    for (int k = 0; k < 128; k++)
      for (int j = 0; j < 16; j++)
        Bm[k * 16 + j] = mode == 0 ? (rand() % 32) - 16 : mode == 1 ? 15 : mode == 2 ? ((k & 1) ? -16 : 16) : -16;
    std::vector<unsigned> A(64 * 4, 0), B(64 * 6, 0);
    for (int l = 0; l < 64; l++)
      for (int e = 0; e < 32; e++) {
        const int k = (l >> 4) * 32 + e;
        A[l * 4 + e / 8] |= (unsigned)Am[(l & 15) * 128 + k] << (4 * (e & 7));   // nibble 00hl = code / 2 in E2M1
        const unsigned f = e2m3(Bm[k * 16 + (l & 15)]);
        const int bit = 6 * e;
        B[l * 6 + bit / 32] |= f << (bit % 32);
        if (bit % 32 > 26) B[l * 6 + bit / 32 + 1] |= f >> (32 - bit % 32);
      }
    unsigned *dA, *dB; float *dD;
    CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1536)); CK(hipMalloc(&dD, 1024));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1536, hipMemcpyHostToDevice));
    for (int reps : {1, 1000, 2900, 46000}) {
      if (mode == 0 && reps == 46000) continue;
      if (mode == 1 && reps > 2900) continue;            // 2900 * 5760 = 1.67e7 < 2^24
      if (mode == 3 && reps > 2700) continue;
      hipLaunchKernelGGL(k_once, dim3(1), dim3(64), 0, 0, dA, dB, dD, reps);
      std::vector<float> D(256);
      CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
      int bad = 0; double mx = 0;
      for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) {
          const int i = 4 * (l >> 4) + r, j = l & 15;
          long long s = 0;
          for (int k = 0; k < 128; k++) s += (long long)Am[i * 128 + k] * Bm[k * 16 + j];
          s *= reps;
          // the device sum is (code / 2) * (d / 8) = s / 16
          if ((double)D[l * 4 + r] * 16.0 != (double)s) bad++;
          if (llabs(s) > mx) mx = (double)llabs(s);
        }
      printf("exactness mode %d reps %5d: %3d of 256 differ from the integer sums (largest |sum| %.0f; 2^24 = 16777216)\n", mode, reps, bad, mx);
      bad_total += bad;
    }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dD));
  }
  return bad_total;
}

// ---- (2) instruction rate -----------------------------------------------------------------------------------------
template <int KIND>   // 0: i8 16x16x64, 1: fp4 x fp4, 2: fp4 x fp6, 3: fp6 x fp6, 4: fp4 x fp8
__global__ __launch_bounds__(256) void k_rate(float *out, int iters) {
  v8i a = {(int)threadIdx.x, 2, 3, 4, 5, 6, 0, 0}, b = {5, 6, 7, (int)blockIdx.x, 9, 10, 11, 12};
  v4f c[8];
  v4i ci[8];
  for (int t = 0; t < 8; t++) { c[t] = v4f{0, 0, 0, 0}; ci[t] = v4i{0, 0, 0, 0}; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int t = 0; t < 8; t++) {
      if (KIND == 1) c[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[t], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      else if (KIND == 2) c[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[t], 4, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      else if (KIND == 3) c[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[t], 2, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      else if (KIND == 4) c[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[t], 4, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      else ci[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(v4i{a[0], a[1], a[2], a[3]}, v4i{b[0], b[1], b[2], b[3]}, ci[t], 0, 0, 0);
    }
  }
  float s = 0;
  for (int t = 0; t < 8; t++) s += KIND ? c[t][0] : (float)ci[t][0];
  if (s == 12345.f) out[0] = s;
}
template <int KIND>
static void rate(const char *name, float *dO) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000, blocks = 256 * 8;
  for (int rep = 0; rep < 2; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, dO, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double n = (double)blocks * 4 * iters * 8;
  printf("rate %-22s %8.2f ms  %6.0f T(FL)OP/s  %5.1f cycles per MFMA per SIMD at 2.4 GHz\n", name, ms,
         n * 16 * 16 * (KIND ? 128 : 64) * 2 / ms / 1e9, ms * 1e-3 * 2.4e9 / (n / 1024));
}

typedef int v6i __attribute__((ext_vector_type(6)));
// The builtin leaves the accumulator of this instruction untied (D lands in the dying registers of the B operand, C comes
// back from scratch: hundreds of spills at 128 registers); as an asm statement with the accumulator as ONE read-write
// operand it stays in place.  s_nop 1: the two wait states between a VALU write of an operand and the MFMA reading it.
__device__ __forceinline__ void mfma_f6(v4f &acc, const v4i &a, const v6i &b, const int scale) {
  asm("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:2"
               : "+v"(acc) : "v"(a), "v"(b), "v"(scale));
}
// ---- (3) kernel skeletons -----------------------------------------------------------------------------------------
// KIND 0: int8 (v_mfma_i32_16x16x64_i8, K-step = 64 samples, 7 + 4 VALU per dword of 16 codes)
// KIND 1: FP4 x FP6 (K-step = 128 samples = two dwords per lane, 3 + 4 VALU per dword)
// P2 0: k_cprod's arithmetic (one digit plane; code plane and missing plane into their own accumulators)
// P2 1: k_prodT's (two digit planes, both into ONE accumulator)
// SHAPE 0: chunk-major image (the rows x 128 B a workgroup reads per chunk are one contiguous run), 1: rows `pitch` apart
template <int KIND, int NB, int TILES, int WAVES, int P2, int SHAPE, int PF>
__global__ __launch_bounds__(64 * WAVES) void k(const uint8_t *__restrict__ img, int64_t pitch, const uint4 *__restrict__ xq4,
                                                unsigned *out, unsigned lutB) {
  constexpr int NCOL = 16 * NB, NT = 64 * WAVES, NP = P2 ? 2 : 1, RW = WAVES * 16 * TILES;
  // digit panel of a chunk (512 samples) per plane: int8 — 32 blocks of 16 samples x NCOL x 16 B;
  // fp6 — 16 lane-groups (4 K-steps x 4) x NCOL x 24 B, kept as a 16-B part and an 8-B part (both conflict free)
  constexpr int S8 = (NB % 2 == 0) ? NCOL + 16 : NCOL;              // stride of the 8-B part in uint2 (bank spread)
  constexpr int PL = KIND ? 16 * NCOL + 8 * S8 : 32 * NCOL;          // uint4 per plane
  constexpr int XS = PL * NP;
  constexpr int NX = (XS + NT - 1) / NT;
  __shared__ uint4 xs[2][XS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int nchunks = (int)(pitch / 128);
  const int64_t wg = blockIdx.x, nwg = gridDim.x;
  auto addr = [&](int t, int ch, int it) -> const uint4 * {
    const int64_t row = wave * (16 * TILES) + t * 16 + c;
    if (SHAPE == 0) return (const uint4 *)(img + ((int64_t)ch * nwg + wg) * (RW * 128) + row * 128 + it * 64 + g * 16);
    return (const uint4 *)(img + (wg * RW + row) * pitch + (int64_t)ch * 128 + it * 64 + g * 16);
  };
  constexpr int NA = P2 ? 1 : 2;
  v4i acci[KIND ? 1 : TILES][KIND ? 1 : NA][KIND ? 1 : NB];
  v4f accf[KIND ? TILES : 1][KIND ? NA : 1][KIND ? NB : 1];
#pragma unroll
  for (int t = 0; t < TILES; t++)
#pragma unroll
    for (int p = 0; p < NA; p++)
#pragma unroll
      for (int nb = 0; nb < NB; nb++) {
        if constexpr (KIND) accf[t][p][nb] = v4f{0, 0, 0, 0};
        else acci[t][p][nb] = v4i{0, 0, 0, 0};
      }
  uint4 ga[2][TILES][2];
#pragma unroll
  for (int t = 0; t < TILES; t++)
#pragma unroll
    for (int it = 0; it < 2; it++) { ga[0][t][it] = *addr(t, 0, it); ga[1][t][it] = *addr(t, nchunks > 1 ? 1 : 0, it); }
#pragma unroll
  for (int x = 0; x < NX; x++) if (tid + x * NT < XS) xs[0][tid + x * NT] = xq4[tid + x * NT];
  __syncthreads();
  auto chunk = [&](auto SETC, const int ch) {
    constexpr int SET = decltype(SETC)::value;
    const int ch1 = ch + 1 < nchunks ? ch + 1 : nchunks - 1, ch2 = ch + 2 < nchunks ? ch + 2 : nchunks - 1;
    static_assert(NX <= 8, "staging registers");
    auto at = [&](const int x) -> int64_t { return (int64_t)ch1 * XS + (tid + x * NT < XS ? tid + x * NT : XS - 1); };
    // (scalars: an array ends up in scratch)
    uint4 xr0 = xq4[at(0)], xr1 = {0, 0, 0, 0}, xr2 = xr1, xr3 = xr1, xr4 = xr1, xr5 = xr1, xr6 = xr1, xr7 = xr1;
    if constexpr (NX > 1) xr1 = xq4[at(1)];
    if constexpr (NX > 2) xr2 = xq4[at(2)];
    if constexpr (NX > 3) xr3 = xq4[at(3)];
    if constexpr (NX > 4) xr4 = xq4[at(4)];
    if constexpr (NX > 5) xr5 = xq4[at(5)];
    if constexpr (NX > 6) xr6 = xq4[at(6)];
    if constexpr (NX > 7) xr7 = xq4[at(7)];
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (KIND == 0) {
#pragma unroll
      for (int it = 0; it < 2; it++)
#pragma unroll
        for (int d = 0; d < 4; d++) {
          v4i b[NP][NB];
#pragma unroll
          for (int p = 0; p < NP; p++)
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
              const uint4 v = xs[SET][p * PL + (it * 16 + g * 4 + d) * NCOL + nb * 16 + c];
              b[p][nb] = v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
            }
#pragma unroll
          for (int t = 0; t < TILES; t++) {
            const uint32_t w = d == 0 ? ga[SET][t][it].x : d == 1 ? ga[SET][t][it].y : d == 2 ? ga[SET][t][it].z : ga[SET][t][it].w;
            const uint32_t s0 = w & 0x03030303u, s1 = (w >> 2) & 0x03030303u, s2 = (w >> 4) & 0x03030303u, s3 = (w >> 6) & 0x03030303u;
            const v4i a0 = {(int)s0, (int)s1, (int)s2, (int)s3};
            const v4i a1 = {(int)__builtin_amdgcn_perm(lutB, lutB, s0), (int)__builtin_amdgcn_perm(lutB, lutB, s1),
                            (int)__builtin_amdgcn_perm(lutB, lutB, s2), (int)__builtin_amdgcn_perm(lutB, lutB, s3)};
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
              acci[t][0][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b[0][nb], acci[t][0][nb], 0, 0, 0);
              acci[t][NA - 1][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b[NP - 1][nb], acci[t][NA - 1][nb], 0, 0, 0);
            }
          }
        }
    } else {
      // phases of a chunk: cprod — the 4 K-steps; prodT — (K-step, digit plane) pairs.  The digit operands of phase i + 1
      // are read into the other register set while the MFMAs of phase i run (static ping-pong: everything is unrolled).
      constexpr int NPH = 4 * NP;
      v6i bb[PF ? 2 : 1][NB];
      auto readb = [&](auto PHC, v6i (&dst)[NB]) {
        constexpr int ph = decltype(PHC)::value, s = ph / NP, p = ph % NP;
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
          const uint4 lo = xs[SET][p * PL + (s * 4 + g) * NCOL + nb * 16 + c];
          const uint2 hi = ((const uint2 *)&xs[SET][p * PL + 16 * NCOL])[(s * 4 + g) * S8 + nb * 16 + c];
          dst[nb] = v6i{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y};
        }
      };
      if constexpr (PF) readb(std::integral_constant<int, 0>{}, bb[0]);
      v4i x[TILES];
      auto phase = [&](auto self, auto PHC) {
        constexpr int ph = decltype(PHC)::value;
        if constexpr (ph < NPH) {
          constexpr int s = ph / NP, p = ph % NP, it = s >> 1, h = s & 1;
          if constexpr (!PF) readb(PHC, bb[0]);
          else if constexpr (ph + 1 < NPH) readb(std::integral_constant<int, ph + 1>{}, bb[(ph + 1) & 1]);
          if constexpr (p == 0) {
#pragma unroll
            for (int t = 0; t < TILES; t++) {
              const uint32_t w0 = h == 0 ? ga[SET][t][it].x : ga[SET][t][it].z, w1 = h == 0 ? ga[SET][t][it].y : ga[SET][t][it].w;
              x[t] = v4i{(int)(w0 & 0x33333333u), (int)((w0 >> 2) & 0x33333333u), (int)(w1 & 0x33333333u), (int)((w1 >> 2) & 0x33333333u)};
            }
          }
#pragma unroll
          for (int t = 0; t < TILES; t++) {
            const v4i a1 = {x[t][0] & (int)((unsigned)x[t][0] >> 1), x[t][1] & (int)((unsigned)x[t][1] >> 1),
                            x[t][2] & (int)((unsigned)x[t][2] >> 1), x[t][3] & (int)((unsigned)x[t][3] >> 1)};
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
              if (!P2 || p == 0) mfma_f6(accf[t][0][nb], x[t], bb[PF ? (ph & 1) : 0][nb], 0x7F7F7F7F);
              if (!P2 || p == 1) mfma_f6(accf[t][NA - 1][nb], a1, bb[PF ? (ph & 1) : 0][nb], 0x7F7F7F7F);
            }
          }
          __builtin_amdgcn_sched_barrier(0);   // a phase at a time: hoisting every digit read of the chunk spills
          self(self, std::integral_constant<int, ph + 1>{});
        }
      };
      phase(phase, std::integral_constant<int, 0>{});
    }
#pragma unroll
    for (int t = 0; t < TILES; t++)
#pragma unroll
      for (int it = 0; it < 2; it++) ga[SET][t][it] = *addr(t, ch2, it);
    __builtin_amdgcn_sched_barrier(0);
    if (tid < XS) xs[SET ^ 1][tid] = xr0;
    if constexpr (NX > 1) if (tid + NT < XS) xs[SET ^ 1][tid + NT] = xr1;
    if constexpr (NX > 2) if (tid + 2 * NT < XS) xs[SET ^ 1][tid + 2 * NT] = xr2;
    if constexpr (NX > 3) if (tid + 3 * NT < XS) xs[SET ^ 1][tid + 3 * NT] = xr3;
    if constexpr (NX > 4) if (tid + 4 * NT < XS) xs[SET ^ 1][tid + 4 * NT] = xr4;
    if constexpr (NX > 5) if (tid + 5 * NT < XS) xs[SET ^ 1][tid + 5 * NT] = xr5;
    if constexpr (NX > 6) if (tid + 6 * NT < XS) xs[SET ^ 1][tid + 6 * NT] = xr6;
    if constexpr (NX > 7) if (tid + 7 * NT < XS) xs[SET ^ 1][tid + 7 * NT] = xr7;
    __syncthreads();
  };
  for (int ch = 0; ch < nchunks; ch += 2) {
    chunk(std::integral_constant<int, 0>{}, ch);
    if (ch + 1 < nchunks) chunk(std::integral_constant<int, 1>{}, ch + 1);
  }
  asm volatile("s_nop 15");   // the last MFMAs' results before compiler code reads them
  unsigned r = 0;
#pragma unroll
  for (int t = 0; t < TILES; t++)
#pragma unroll
    for (int p = 0; p < NA; p++)
#pragma unroll
      for (int nb = 0; nb < NB; nb++) {
        if constexpr (KIND) r ^= __float_as_uint(accf[t][p][nb][0]) ^ __float_as_uint(accf[t][p][nb][3]);
        else r ^= (unsigned)(acci[t][p][nb][0] ^ acci[t][p][nb][3]);
      }
  if (r == 0x12345679u) out[0] = r;
}

// ---- (3b) the column blocks DIVIDED BETWEEN WAVES (round 6, second attempt) ---------------------------------------------
// What sank the shapes above is the register budget: five column blocks x two planes x two tiles = 80 accumulators, two waves
// per SIMD.  Here the blocks of the panel are divided between two KINDS of waves over the same variants: kind A owns blocks
// [0, NBA) of TA tiles, kind B blocks [NBA, NBA + NBB) of TB tiles, with WA TA = WB TB tiles per workgroup — every genotype
// row is loaded and decoded by one wave of each kind (twice the L1 / L2 reads and twice the 14-instruction decode, the same
// HBM bytes and the same LDS operand traffic), every wave keeps <= 48 accumulators: four waves per SIMD like the int8 kernels.
//   NB = 4: 8 + 8 waves of 2 tiles x 2 blocks;   NB = 5: 9 waves of 2 tiles x 3 blocks + 6 waves of 3 tiles x 2 blocks.
template <int NBW, int T, int NBT, int NT>
__device__ __forceinline__ void fp6_role(const uint8_t *__restrict__ img, int64_t pitch, const uint4 *__restrict__ xq4, uint4 *xs,
                                         int64_t row0, int boff, unsigned *out) {
  constexpr int NCOL = 16 * NBT, S8 = (NBT % 2 == 0) ? NCOL + 16 : NCOL, XS = 16 * NCOL + 8 * S8, NX = (XS + NT - 1) / NT;
  static_assert(NX <= 4, "staging registers");
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int nchunks = (int)(pitch / 128);
  auto addr = [&](int t, int ch, int it) -> const uint4 * {
    return (const uint4 *)(img + (row0 + t * 16 + c) * pitch + (int64_t)ch * 128 + it * 64 + g * 16);
  };
  v4f acc[T][2][NBW];
#pragma unroll
  for (int t = 0; t < T; t++)
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
      for (int nb = 0; nb < NBW; nb++) acc[t][p][nb] = v4f{0, 0, 0, 0};
  uint4 ga[2][T][2];
#pragma unroll
  for (int t = 0; t < T; t++)
#pragma unroll
    for (int it = 0; it < 2; it++) { ga[0][t][it] = *addr(t, 0, it); ga[1][t][it] = *addr(t, nchunks > 1 ? 1 : 0, it); }
#pragma unroll
  for (int x = 0; x < NX; x++) if (tid + x * NT < XS) xs[tid + x * NT] = xq4[tid + x * NT];
  __syncthreads();
  auto chunk = [&](auto SETC, const int ch) {
    constexpr int SET = decltype(SETC)::value;
    const int ch1 = ch + 1 < nchunks ? ch + 1 : nchunks - 1, ch2 = ch + 2 < nchunks ? ch + 2 : nchunks - 1;
    auto at = [&](const int x) -> int64_t { return (int64_t)ch1 * XS + (tid + x * NT < XS ? tid + x * NT : XS - 1); };
    uint4 xr0 = xq4[at(0)], xr1 = {0, 0, 0, 0}, xr2 = xr1, xr3 = xr1;
    if constexpr (NX > 1) xr1 = xq4[at(1)];
    if constexpr (NX > 2) xr2 = xq4[at(2)];
    if constexpr (NX > 3) xr3 = xq4[at(3)];
    __builtin_amdgcn_sched_barrier(0);
    const uint4 *xb = xs + SET * XS;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      constexpr int dummy = 0; (void)dummy;
      const int it = s >> 1, h = s & 1;
      __builtin_amdgcn_sched_barrier(0);
      v6i b[NBW];
#pragma unroll
      for (int nb = 0; nb < NBW; nb++) {
        const uint4 lo = xb[(s * 4 + g) * NCOL + (boff + nb) * 16 + c];
        const uint2 hi = ((const uint2 *)&xb[16 * NCOL])[(s * 4 + g) * S8 + (boff + nb) * 16 + c];
        b[nb] = v6i{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y};
      }
#pragma unroll
      for (int t = 0; t < T; t++) {
        const uint32_t w0 = h == 0 ? ga[SET][t][it].x : ga[SET][t][it].z, w1 = h == 0 ? ga[SET][t][it].y : ga[SET][t][it].w;
        const v4i x = {(int)(w0 & 0x33333333u), (int)((w0 >> 2) & 0x33333333u), (int)(w1 & 0x33333333u), (int)((w1 >> 2) & 0x33333333u)};
        const v4i a1 = {x[0] & (int)((unsigned)x[0] >> 1), x[1] & (int)((unsigned)x[1] >> 1), x[2] & (int)((unsigned)x[2] >> 1), x[3] & (int)((unsigned)x[3] >> 1)};
#pragma unroll
        for (int nb = 0; nb < NBW; nb++) {
          mfma_f6(acc[t][0][nb], x, b[nb], 0x7F7F7F7F);
          mfma_f6(acc[t][1][nb], a1, b[nb], 0x7F7F7F7F);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
      for (int it = 0; it < 2; it++) ga[SET][t][it] = *addr(t, ch2, it);
    __builtin_amdgcn_sched_barrier(0);
    uint4 *xw = xs + (SET ^ 1) * XS;
    if (tid < XS) xw[tid] = xr0;
    if constexpr (NX > 1) if (tid + NT < XS) xw[tid + NT] = xr1;
    if constexpr (NX > 2) if (tid + 2 * NT < XS) xw[tid + 2 * NT] = xr2;
    if constexpr (NX > 3) if (tid + 3 * NT < XS) xw[tid + 3 * NT] = xr3;
    __syncthreads();
  };
  for (int ch = 0; ch < nchunks; ch += 2) {
    chunk(std::integral_constant<int, 0>{}, ch);
    if (ch + 1 < nchunks) chunk(std::integral_constant<int, 1>{}, ch + 1);
  }
  asm volatile("s_nop 15");
  unsigned r = 0;
#pragma unroll
  for (int t = 0; t < T; t++)
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
      for (int nb = 0; nb < NBW; nb++) r ^= __float_as_uint(acc[t][p][nb][0]) ^ __float_as_uint(acc[t][p][nb][3]);
  if (r == 0x12345679u) out[0] = r;
}
template <int NBA, int TA, int WA, int NBB, int TB, int WB>
__global__ __launch_bounds__(64 * (WA + WB)) void kr(const uint8_t *__restrict__ img, int64_t pitch, const uint4 *__restrict__ xq4, unsigned *out) {
  static_assert(WA * TA == WB * TB, "both kinds of waves cover the workgroup's tiles");
  constexpr int NBT = NBA + NBB, NT = 64 * (WA + WB), NCOL = 16 * NBT, S8 = (NBT % 2 == 0) ? NCOL + 16 : NCOL, XS = 16 * NCOL + 8 * S8;
  __shared__ uint4 xs[2 * XS];
  const int wave = threadIdx.x >> 6;
  const int64_t row_wg = (int64_t)blockIdx.x * (WA * TA * 16);
  if (wave < WA) fp6_role<NBA, TA, NBT, NT>(img, pitch, xq4, xs, row_wg + wave * (TA * 16), 0, out);
  else fp6_role<NBB, TB, NBT, NT>(img, pitch, xq4, xs, row_wg + (wave - WA) * (TB * 16), NBA, out);
}
template <int NBA, int TA, int WA, int NBB, int TB, int WB>
void run_roles(const uint8_t *img, int64_t pitch, int64_t rows, const uint4 *xq, unsigned *out, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  constexpr int RW = WA * TA * 16, NB = NBA + NBB;
  const unsigned grid = (unsigned)(rows / RW);
  auto kern = kr<NBA, TA, WA, NBB, TB, WB>;
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, (const void *)kern));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (WA + WB)), 0, 0, img, pitch, xq, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (WA + WB)), 0, 0, img, pitch, xq, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double bytes = (double)grid * RW * pitch;
  const double mfma = bytes * 4 / 128 / 16 * 2 * NB, cyc = mfma * 16 / 1024;
  printf("cprod  fp6xfp4 NB=%d (%2d-bit) column blocks divided between waves: %d waves x %d tiles x %d blocks + %d x %d x %d  regs %3d scratch %3zu lds %6zu  %7.2f ms per 100 GB  %5.0f GB/s  pipe floor %5.2f ms at 1.7 GHz\n",
         NB, 5 * NB, WA, TA, NBA, WB, TB, NBB, fa.numRegs, (size_t)fa.localSizeBytes, (size_t)fa.sharedSizeBytes, ms * 100e9 / bytes, bytes / ms / 1e6,
         cyc / 1.7e9 * 1e3 * 100e9 / bytes);
  fflush(stdout);
}

__global__ void fill(uint32_t *p, size_t n, int genotypes) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 7), w = 0;
    if (genotypes) {
      for (int e = 0; e < 16; e++) {
        h = h * 1664525u + 1013904223u;
        const uint32_t r = h >> 24;
        w |= (r < 3 ? 3u : r < 140 ? 0u : r < 220 ? 1u : 2u) << (2 * e);   // 1 % missing
      }
    } else {
      w = h * 1664525u + 1013904223u;
      w ^= w >> 15;
    }
    p[i] = w;
  }
}

template <int KIND, int NB, int TILES, int WAVES, int P2, int SHAPE, int PF = 0>
void run(const uint8_t *img, int64_t pitch, int64_t rows, const uint4 *xq, unsigned *out, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  constexpr int RW = WAVES * 16 * TILES;
  const unsigned grid = (unsigned)(rows / RW);
  auto kern = k<KIND, NB, TILES, WAVES, P2, SHAPE, PF>;
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, (const void *)kern));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), 0, 0, img, pitch, xq, out, 0x01000000u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), 0, 0, img, pitch, xq, out, 0x01000000u);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double bytes = (double)grid * RW * pitch;
  // matrix-pipe cycles per SIMD: int8 — 16 per MFMA of 64 samples; fp6 — 16 per MFMA of 128 samples; 2 planes x NB per 16 rows
  const double mfma = bytes * 4 / (KIND ? 128 : 64) / 16 * 2 * NB, cyc = mfma * 16 / 1024;
  printf("%-6s %-7s NB=%d (%2d-bit) tiles=%d waves=%2d pf=%d regs %3d scratch %3zu lds %6zu  %7.2f ms per 100 GB  %5.0f GB/s  pipe floor %5.2f ms at 1.7 GHz\n",
         P2 ? "prodT" : "cprod", KIND ? "fp6xfp4" : "int8", NB, KIND ? 5 * NB : 8 * NB, TILES, WAVES, PF, fa.numRegs, (size_t)fa.localSizeBytes, (size_t)fa.sharedSizeBytes,
         ms * 100e9 / bytes, bytes / ms / 1e6, cyc / 1.7e9 * 1e3 * 100e9 / bytes);
  fflush(stdout);
}

int main(int argc, char **argv) {
  // fp6_parts [reps] [digits: 0 random, 1 all zero (no operand toggling: what the power cap costs)] [quick: 1 = the six
  // configurations of the counter passes only, no exactness / rate part]
  const int reps = argc > 1 ? atoi(argv[1]) : 12;
  const int zero_digits = argc > 2 ? atoi(argv[2]) : 0;
  const int quick = argc > 3 ? atoi(argv[3]) : 0;
  if (!quick) {
    const int bad = exactness();
    printf("exactness: %s\n", bad ? "FAILED" : "all sums exact");
    float *dO; CK(hipMalloc(&dO, 4));
    rate<0>("i8 16x16x64", dO);
    rate<1>("fp4 x fp4 16x16x128", dO);
    rate<2>("fp4 x fp6 16x16x128", dO);
    rate<3>("fp6 x fp6 16x16x128", dO);
    rate<4>("fp4 x fp8 16x16x128", dO);
  }
  const int64_t pitch = 100096, rows = 245760;   // = 384 * 640 = 512 * 480 = 256 * 960 rows: 24.6 GB per launch
  uint8_t *img; uint4 *xq; unsigned *out;
  const size_t xq_bytes = (size_t)(pitch / 128) * 8192 * 16;
  CK(hipMalloc(&img, (size_t)rows * pitch)); CK(hipMalloc(&xq, xq_bytes)); CK(hipMalloc(&out, 64));
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint32_t *)img, (size_t)rows * pitch / 4, 1);
  if (zero_digits) CK(hipMemset(xq, 0, xq_bytes));
  else hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, (uint32_t *)xq, xq_bytes / 4, 0);
  CK(hipDeviceSynchronize());
  printf("digit panels: %s\n", zero_digits ? "all zero" : "random bits");
  if (quick) {
    run<0, 3, 2, 16, 0, 1>(img, pitch, rows, xq, out, reps);
    run<0, 2, 2, 16, 0, 1>(img, pitch, rows, xq, out, reps);
    run<1, 5, 2, 8, 0, 1, 1>(img, pitch, rows, xq, out, reps);
    run<1, 4, 1, 16, 0, 1, 0>(img, pitch, rows, xq, out, reps);
    run<0, 3, 2, 16, 1, 0>(img, pitch, rows, xq, out, reps);
    run<1, 4, 4, 8, 1, 0, 0>(img, pitch, rows, xq, out, reps);
    run_roles<2, 2, 8, 2, 2, 8>(img, pitch, rows, xq, out, reps);     // 20 bits: 8 + 8 waves
    run_roles<3, 2, 9, 2, 3, 6>(img, pitch, rows, xq, out, reps);     // 25 bits: 9 + 6 waves
    run_roles<3, 2, 6, 2, 3, 4>(img, pitch, rows, xq, out, reps);     // 25 bits: 6 + 4 waves
    run_roles<2, 1, 8, 1, 2, 4>(img, pitch, rows, xq, out, reps);     // 15 bits: 8 + 4 waves (one tile / two tiles)
    return 0;
  }
  for (int pass = 0; pass < 2; pass++) {
    printf("--- pass %d ---\n", pass);
    // k_cprod's arithmetic on the variant-major image
    run<0, 3, 2, 16, 0, 1>(img, pitch, rows, xq, out, reps);   // the shipped shape of k_cprod<3>
    run<0, 2, 2, 16, 0, 1>(img, pitch, rows, xq, out, reps);   // ... of k_cprod<2> (16-bit panels)
    run<1, 5, 2, 8, 0, 1, 0>(img, pitch, rows, xq, out, reps);
    run<1, 5, 2, 8, 0, 1, 1>(img, pitch, rows, xq, out, reps);
    run<1, 5, 1, 16, 0, 1, 0>(img, pitch, rows, xq, out, reps);
    run<1, 5, 1, 8, 0, 1, 1>(img, pitch, rows, xq, out, reps);
    run<1, 4, 2, 8, 0, 1, 0>(img, pitch, rows, xq, out, reps);
    run<1, 4, 2, 8, 0, 1, 1>(img, pitch, rows, xq, out, reps);
    run<1, 4, 2, 12, 0, 1, 0>(img, pitch, rows, xq, out, reps);
    run<1, 4, 1, 16, 0, 1, 0>(img, pitch, rows, xq, out, reps);
    run<1, 4, 1, 16, 0, 1, 1>(img, pitch, rows, xq, out, reps);
    run<1, 3, 2, 16, 0, 1, 0>(img, pitch, rows, xq, out, reps);
    run<1, 3, 2, 12, 0, 1, 1>(img, pitch, rows, xq, out, reps);
    // k_prodT's arithmetic on the chunk-major copy
    run<0, 3, 2, 16, 1, 0>(img, pitch, rows, xq, out, reps);   // the shipped shape of k_prodT<3>
    run<0, 2, 2, 16, 1, 0>(img, pitch, rows, xq, out, reps);
    run<1, 5, 4, 8, 1, 0, 0>(img, pitch, rows, xq, out, reps);
    run<1, 5, 2, 12, 1, 0, 0>(img, pitch, rows, xq, out, reps);
    run<1, 5, 2, 12, 1, 0, 1>(img, pitch, rows, xq, out, reps);
    run<1, 4, 4, 8, 1, 0, 0>(img, pitch, rows, xq, out, reps);
    run<1, 4, 4, 8, 1, 0, 1>(img, pitch, rows, xq, out, reps);
    run<1, 4, 2, 12, 1, 0, 0>(img, pitch, rows, xq, out, reps);
    run<1, 4, 2, 12, 1, 0, 1>(img, pitch, rows, xq, out, reps);
    run<1, 4, 2, 16, 1, 0, 0>(img, pitch, rows, xq, out, reps);
  }
  return 0;
}
