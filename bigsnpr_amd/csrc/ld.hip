// ld.hip — windowed pairwise-complete correlation, LD scores and clumping on gfx950.
//
// Replaces corMat0 (src/corr.cpp:11-97), ld_scores0 (src/ld-scores.cpp:11-78),
// clumping_chr (src/clumping.cpp:10-91) and bed_clumping_chr (src/clumping-bed.cpp:11-91).
//
// The reference walks every (j0, j) pair of the position window and every sample with a
// branchy scalar loop.  Here the six pairwise-complete sums of a 64 x 64 block of variant
// pairs are six exact integer GEMMs over the samples on the i8 MFMA pipe, on three planes
// decoded from the 2-bit codes: X = genotype (missing -> 0), X2 = X^2, M = non-missing:
//     sum xy = X.X      sum_{both} x = X.M     sum_{both} x^2 = X2.M
//     nona   = M.M      sum_{both} y = M.X     sum_{both} y^2 = M.X2
// (values <= 4n < 2^31, so int32 accumulation is exact, as are the reference's fp64 sums of
// small integers).  The fp64 epilogue then repeats the reference's expressions in the
// reference's operation order on identical integer inputs.
#include <algorithm>
#include <cmath>
#include <memory>
#include <unordered_map>

#include <cstdlib>

#include <cstring>
#include "bsn_internal.hpp"

namespace bsn {

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t lut4b(uint32_t lut, uint32_t sel) {
  return __builtin_amdgcn_perm(lut, lut, sel);
}
// device code (bsn_internal.hpp: 0, 1, 2 = allele count, 3 = missing) -> plane value
constexpr uint32_t kLX = 0x00020100u;   // code 0,1,2,3 -> 0, 1, 2, 0
constexpr uint32_t kLX2 = 0x00040100u;  //               -> 0, 1, 4, 0
constexpr uint32_t kLM = 0x00010101u;   //               -> 1, 1, 1, 0
constexpr int TB = 64;                  // variants per tile side

struct Planes {
  v4i x, x2, m;
};
__device__ __forceinline__ Planes decode3(uint32_t w) {
  const uint32_t s0 = w & 0x03030303u, s1 = (w >> 2) & 0x03030303u, s2 = (w >> 4) & 0x03030303u,
                 s3 = (w >> 6) & 0x03030303u;
  Planes p;
  p.x = v4i{(int)lut4b(kLX, s0), (int)lut4b(kLX, s1), (int)lut4b(kLX, s2), (int)lut4b(kLX, s3)};
  p.x2 = v4i{(int)lut4b(kLX2, s0), (int)lut4b(kLX2, s1), (int)lut4b(kLX2, s2), (int)lut4b(kLX2, s3)};
  p.m = v4i{(int)lut4b(kLM, s0), (int)lut4b(kLM, s1), (int)lut4b(kLM, s2), (int)lut4b(kLM, s3)};
  return p;
}

// The fp64 epilogue of one variant pair, in the reference's operation order (no contraction):
// mode 0: r of corMat0 with threshold (dropped -> 2.0), src/corr.cpp:76-86;  mode 1: r2 of
// ld_scores0, src/ld-scores.cpp:52-78;  mode 2: r2 of clumping_chr (raw formula with cached
// sumX / denoX, n rows), src/clumping.cpp:72-73;  mode 3: r2 of bed_clumping_chr on mean-imputed
// scaled values, src/clumping-bed.cpp:69-73 (v1 = center, v2 = scale).
#pragma clang fp contract(off)
__device__ __forceinline__ double pair_value(int mode, double xySum, double xSum, double xxSum, double ySum,
                                             double yySum, int nona, const double *__restrict__ thr,
                                             const double *__restrict__ v1, const double *__restrict__ v2,
                                             int64_t j0, int64_t j, double nrows) {
  if (mode == 0 || mode == 1) {
    const double num = xySum - xSum * ySum / nona;
    const double deno_x = xxSum - xSum * xSum / nona;
    const double deno_y = yySum - ySum * ySum / nona;
    if (mode == 0) {
      double r = num / sqrt(deno_x * deno_y);
      // src/corr.cpp:82-86; thr[nona - 1] is only read when r is not NaN (nona >= 1 then)
      if (isnan(r) || fabs(r) > thr[nona > 0 ? nona - 1 : 0]) {
        if (r > 1) r = 1; else if (r < -1) r = -1;
        return r;
      }
      return 2.0;
    }
    return num * num / (deno_x * deno_y);
  } else if (mode == 2) {
    // v1 = sumX, v2 = denoX (per position in ind_col)
    const double num = xySum - v1[j] * v1[j0] / nrows;
    return num * num / (v2[j] * v2[j0]);
  }
  // sum_i x~ y~ over rows where both are present (missing -> 0)
  const double cx = v1[j0], cy = v1[j], sx = v2[j0], sy = v2[j];
  const double s = xySum - cx * ySum - cy * xSum + cx * cy * (double)nona;
  const double r = s / (sx * sy);
  return r * r;
}
#pragma clang fp contract(on)

// what the fused epilogue of k_pair_stats needs to turn its sums into band entries
struct BandOut {
  int64_t m, W;
  const int64_t *lo;
  const double *thr, *v1, *v2;
  double nrows;
  double *band;
  int mode;
  // k_pair_stats_f4<., RAW = true> only: per-variant totals over the selected samples (sum x, sum x^2, non-missing count,
  // by position in ind.col) and the number of sample positions the kernel walks (4 pitch: dropped and pad samples included)
  const double *cx = nullptr, *cxx = nullptr, *cnn = nullptr;
  double npos = 0;
};

// stats[pair][prod][row][col], prod: 0 xy, 1 x(both), 2 xx(both), 3 y(both), 4 yy(both), 5 nona
// row = variant of tile I (the "x" / j0 side), col = variant of tile J (the "y" / j side).
// rowmask (optional): 2 bits per sample, 11 = keep; dropped samples are turned into code 11
// (missing) so that they vanish from all six sums.
// ALL = false: only product 0 (xy) is computed and stored — the case of variants without missing
// values among the selected samples, where the other five sums are per-variant constants.
// FUSE: the whole sample range is in this workgroup (no K split), so the six sums of a pair sit in
// one lane's accumulators and the fp64 epilogue runs right here: the band entry is written, the
// int32 statistics never leave the registers.
// CONTIG: the variants of a tile lie within 2 GB of its first one (a contiguous ind.col): the genotype loads are
// buffer loads with one scalar descriptor per operand tile and a 32-bit lane offset instead of four 64-bit lane
// addresses — the four registers that decide between two and three waves per SIMD (168 registers).
template <bool ALL, bool FUSE, bool CONTIG = false>
__global__ __launch_bounds__(256) void k_pair_stats(const uint8_t *__restrict__ img, int64_t pitch,
                                                    const int32_t *__restrict__ cols,
                                                    const int2 *__restrict__ pairs,
                                                    const uint32_t *__restrict__ rowmask,
                                                    int64_t kbytes_per_split,
                                                    int32_t *__restrict__ stats, int use_atomic, BandOut bo) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r16 = lane & 15, g = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int2 pr = pairs[blockIdx.x];
  const uint8_t *pa[2], *pb[2];
  uint32_t va[2], vb[2];
  const int64_t ca0 = cols[pr.x * TB], cb0 = cols[pr.y * TB];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int64_t ca = cols[pr.x * TB + wr * 32 + s * 16 + r16], cb = cols[pr.y * TB + wc * 32 + s * 16 + r16];
    pa[s] = img + ca * pitch + g * 16;
    pb[s] = img + cb * pitch + g * 16;
    va[s] = (uint32_t)((ca - ca0) * pitch + g * 16);
    vb[s] = (uint32_t)((cb - cb0) * pitch + g * 16);
  }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)(img + ca0 * pitch), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)(img + cb0 * pitch), 0, 0x7fffffff, 0x00020000);
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  int64_t b0 = (int64_t)blockIdx.y * kbytes_per_split, b1 = b0 + kbytes_per_split;
  if (b1 > pitch) b1 = pitch;

  constexpr int NP = ALL ? 6 : 1;
  v4i acc[2][2][NP];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int p = 0; p < NP; p++) acc[i][j][p] = v4i{0, 0, 0, 0};

  for (int64_t kb = b0; kb < b1; kb += 64) {  // 64 B per row = 256 samples per iteration
    uint4 a[2], b[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
      if constexpr (CONTIG) {
        const v4u x = __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)va[s], (int)kb, 0);
        const v4u y = __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)vb[s], (int)kb, 0);
        a[s] = uint4{x.x, x.y, x.z, x.w};
        b[s] = uint4{y.x, y.y, y.z, y.w};
      } else {
        a[s] = *(const uint4 *)(pa[s] + kb);
        b[s] = *(const uint4 *)(pb[s] + kb);
      }
    }
    uint4 mk = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (rowmask) mk = *(const uint4 *)((const uint8_t *)rowmask + kb + g * 16);
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const uint32_t mw = d == 0 ? mk.x : d == 1 ? mk.y : d == 2 ? mk.z : mk.w;
      Planes A[2], B[2];
#pragma unroll
      for (int s = 0; s < 2; s++) {
        uint32_t wa = d == 0 ? a[s].x : d == 1 ? a[s].y : d == 2 ? a[s].z : a[s].w;
        uint32_t wb = d == 0 ? b[s].x : d == 1 ? b[s].y : d == 2 ? b[s].z : b[s].w;
        wa |= ~mw;
        wb |= ~mw;
        A[s] = decode3(wa);
        B[s] = decode3(wb);
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          acc[i][j][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i].x, B[j].x, acc[i][j][0], 0, 0, 0);
          if constexpr (ALL) {
            acc[i][j][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i].x, B[j].m, acc[i][j][1], 0, 0, 0);
            acc[i][j][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i].x2, B[j].m, acc[i][j][2], 0, 0, 0);
            acc[i][j][3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i].m, B[j].x, acc[i][j][3], 0, 0, 0);
            acc[i][j][4] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i].m, B[j].x2, acc[i][j][4], 0, 0, 0);
            acc[i][j][5] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i].m, B[j].m, acc[i][j][5], 0, 0, 0);
          }
        }
    }
  }
  if constexpr (FUSE && ALL) {
    // The sixteen pairs of a lane go through the fp64 epilogue ONE AT A TIME in a rolled loop: their sums are
    // parked in a run-time-indexed private array (scratch memory, written once, read once).  Unrolled, the
    // sixteen inlined epilogues raised the register allocation of the WHOLE kernel from 168 to 224 — two waves
    // per SIMD instead of three for the MFMA loop, which is where the time goes.
    int32_t st[16][6];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int p = 0; p < 6; p++) st[(i * 2 + j) * 4 + r][p] = acc[i][j][p][r];
#pragma unroll 1
    for (int e = 0; e < 16; e++) {
      const int i = e >> 3, j = (e >> 2) & 1, r = e & 3;
      const int row = wr * 32 + i * 16 + 4 * g + r, col = wc * 32 + j * 16 + r16;
      const int64_t j0 = (int64_t)pr.x * TB + row, jj = (int64_t)pr.y * TB + col;
      if (j0 >= bo.m || jj >= j0 || jj < bo.lo[j0]) continue;
      bo.band[j0 * bo.W + (j0 - jj - 1)] =
          pair_value(bo.mode, (double)st[e][0], (double)st[e][1], (double)st[e][2], (double)st[e][3], (double)st[e][4],
                     st[e][5], bo.thr, bo.v1, bo.v2, j0, jj, bo.nrows);
    }
    return;
  }
  int32_t *out = stats + (int64_t)blockIdx.x * 6 * TB * TB;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int p = 0; p < NP; p++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = wr * 32 + i * 16 + 4 * g + r, col = wc * 32 + j * 16 + r16;
          int32_t *dst = out + (p * TB + row) * TB + col;
          if (use_atomic)
            atomicAdd(dst, acc[i][j][p][r]);
          else
            *dst = acc[i][j][p][r];
        }
}

// The six-product kernel with the decode of one operand SHARED through LDS (round 3).  k_pair_stats decodes
// three planes of four 16-variant sub-tiles per wave and K-step for 24 MFMAs: 3.5 VALU instructions per MFMA,
// and VALU and MFMA share the issue port (16 of an MFMA's cycles against 4 per VALU instruction: at 3.5 the port,
// not the matrix pipe, is the bound — counters: pipe 63 - 68 % busy).  Here a workgroup covers 128 x 32 variant
// pairs (the area of a 64 x 64 tile pair): every wave owns 32 of the 128 "row" variants (decoded privately, as
// before) and ALL four waves multiply them with the same 32 "column" variants, which are therefore decoded once
// per workgroup — wave w decodes K-step w of the four K-steps of an iteration and parks the three planes in LDS
// (double-buffered: one barrier per 256 samples).  2.1 VALU per MFMA; LDS moves 30 KB per wave and iteration,
// 60 % of its bandwidth at three workgroups per CU.
// pairs: (index of the 128-variant row block, index of the 32-variant column block); fused fp64 epilogue.
constexpr int TR = 128, TC = 32;
// (Round 4: an explicit MFMA : VALU schedule through __builtin_amdgcn_sched_group_barrier, as in k_cprod — 1 or 2 VALU
// offered behind every MFMA, the column operand's decode pulled into the last K-step — needs 210 registers, and forced
// back to 168 for the third wave it runs 422 - 434 ms against 292 at C5: profiles/r04_ld.txt.  Not kept.)
// PRIO (round 6, second session): s_setprio around the matrix instructions of a K-step, as in k_pair_stats_f4 (profiles/r06_ld_raw.txt).
template <int PRIO = 0>
__global__ __launch_bounds__(256, 2) void k_pair_stats_b(const uint8_t *__restrict__ img, int64_t pitch,
                                                      const int32_t *__restrict__ cols,
                                                      const int2 *__restrict__ pairs,
                                                      const uint32_t *__restrict__ rowmask, BandOut bo) {
  __shared__ uint4 sB[2][4][2][3][64];   // [buffer][K-step][sub-tile][plane x, x2, m][lane]: 48 KB
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r16 = lane & 15, g = lane >> 4;
  const int2 pr = pairs[blockIdx.x];
  const int64_t ca0 = cols[pr.x * TR], cb0 = cols[pr.y * TC];
  uint32_t va[2], vb[2];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    va[s] = (uint32_t)(((int64_t)cols[pr.x * TR + wave * 32 + s * 16 + r16] - ca0) * pitch + g * 16);
    vb[s] = (uint32_t)(((int64_t)cols[pr.y * TC + s * 16 + r16] - cb0) * pitch + g * 16);
  }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)(img + ca0 * pitch), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)(img + cb0 * pitch), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc((void *)rowmask, 0, 0x7fffffff, 0x00020000);
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  v4i acc[2][2][6];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int p = 0; p < 6; p++) acc[i][j][p] = v4i{0, 0, 0, 0};
  const int nit = (int)(pitch / 64);
  // word `wave` of a 16-byte register (uniform per wave)
  auto pick = [&](const v4u &x) -> uint32_t { return wave == 0 ? x.x : wave == 1 ? x.y : wave == 2 ? x.z : x.w; };
  auto stash = [&](int buf, const uint32_t (&b)[2], const v4u &mk) {   // this wave's K-step of the column operand -> LDS
    const uint32_t mw = pick(mk);
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const Planes P = decode3(b[s] | ~mw);
      sB[buf][wave][s][0][lane] = uint4{(uint32_t)P.x[0], (uint32_t)P.x[1], (uint32_t)P.x[2], (uint32_t)P.x[3]};
      sB[buf][wave][s][1][lane] = uint4{(uint32_t)P.x2[0], (uint32_t)P.x2[1], (uint32_t)P.x2[2], (uint32_t)P.x2[3]};
      sB[buf][wave][s][2][lane] = uint4{(uint32_t)P.m[0], (uint32_t)P.m[1], (uint32_t)P.m[2], (uint32_t)P.m[3]};
    }
  };
  // a wave needs only ITS K-step's word of the column operand: one dword per sub-tile and lane
  v4u a[2], mk, an[2], mkn;
  uint32_t b[2], bn[2];
  const int wsel = __builtin_amdgcn_readfirstlane(wave) * 4;
#pragma unroll
  for (int s = 0; s < 2; s++) {
    a[s] = __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)va[s], 0, 0);
    b[s] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsB, (int)vb[s], wsel, 0);
  }
  mk = __builtin_amdgcn_raw_buffer_load_b128(rsM, g * 16, 0, 0);
  stash(0, b, mk);
  __syncthreads();
  for (int it = 0; it < nit; it++) {
    // the next iteration's words (past the end the last one again: no branch around loads)
    const int kbn = (it + 1 < nit ? it + 1 : it) * 64;
#pragma unroll
    for (int s = 0; s < 2; s++) {
      an[s] = __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)va[s], kbn, 0);
      bn[s] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsB, (int)vb[s], kbn + wsel, 0);
    }
    mkn = __builtin_amdgcn_raw_buffer_load_b128(rsM, g * 16, kbn, 0);
    const int buf = it & 1;
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const uint32_t mw = d == 0 ? mk.x : d == 1 ? mk.y : d == 2 ? mk.z : mk.w;
      Planes A[2];
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const uint32_t wa = d == 0 ? a[s].x : d == 1 ? a[s].y : d == 2 ? a[s].z : a[s].w;
        A[s] = decode3(wa | ~mw);
      }
      if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const uint4 bx = sB[buf][d][j][0][lane], bx2 = sB[buf][d][j][1][lane], bm = sB[buf][d][j][2][lane];
        const v4i Bx = {(int)bx.x, (int)bx.y, (int)bx.z, (int)bx.w}, Bx2 = {(int)bx2.x, (int)bx2.y, (int)bx2.z, (int)bx2.w},
                  Bm = {(int)bm.x, (int)bm.y, (int)bm.z, (int)bm.w};
#pragma unroll
        for (int i = 0; i < 2; i++) {
          acc[i][j][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i].x, Bx, acc[i][j][0], 0, 0, 0);
          acc[i][j][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i].x, Bm, acc[i][j][1], 0, 0, 0);
          acc[i][j][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i].x2, Bm, acc[i][j][2], 0, 0, 0);
          acc[i][j][3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i].m, Bx, acc[i][j][3], 0, 0, 0);
          acc[i][j][4] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i].m, Bx2, acc[i][j][4], 0, 0, 0);
          acc[i][j][5] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i].m, Bm, acc[i][j][5], 0, 0, 0);
        }
      }
      if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(0);
    }
    // (Placing this decode between the K-steps above, where the matrix pipe is busy, costs 24 registers — two
    // waves per SIMD instead of three — and 12 % of the time: measured.)
    stash(buf ^ 1, bn, mkn);   // read by nobody until the barrier below; last read before the previous barrier
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; s++) a[s] = an[s];
    mk = mkn;
  }
  // fp64 epilogue, one pair at a time (see k_pair_stats)
  int32_t st[16][6];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int p = 0; p < 6; p++) st[(i * 2 + j) * 4 + r][p] = acc[i][j][p][r];
#pragma unroll 1
  for (int e = 0; e < 16; e++) {
    const int i = e >> 3, j = (e >> 2) & 1, r = e & 3;
    const int row = wave * 32 + i * 16 + 4 * g + r, col = j * 16 + r16;
    const int64_t j0 = (int64_t)pr.x * TR + row, jj = (int64_t)pr.y * TC + col;
    if (j0 >= bo.m || jj >= j0 || jj < bo.lo[j0]) continue;
    bo.band[j0 * bo.W + (j0 - jj - 1)] =
        pair_value(bo.mode, (double)st[e][0], (double)st[e][1], (double)st[e][2], (double)st[e][3], (double)st[e][4],
                   st[e][5], bo.thr, bo.v1, bo.v2, j0, jj, bo.nrows);
  }
}

// (Round 5) The same six products on the FP4 matrix pipe.  The plane values 0, 1, 2, 4 are exact in FP4 (E2M1: 0b0000,
// 0b0010, 0b0100, 0b0110), gfx950's block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales (E8M0 127) contracts
// 128 samples per instruction where v_mfma_i32_16x16x64_i8 contracts 64, at 1.85 x the rate (tools/ubench/fp4_mfma.hip:
// same issue cycles, twice the K), and its fp32 accumulation of small integers is exact up to 2^24 (checked there with
// sums of 1.1e7): the pair statistics, at most 4 n, are exact for n <= 4 194 303 samples.  Same workgroup shape as
// k_pair_stats_b (128 x 32 variant pairs, the column operand decoded once per workgroup through LDS); an iteration of
// 256 samples is TWO K-steps, wave w decodes (K-step w >> 1, column sub-tile w & 1).  A lane's 32 samples of a K-step
// are two dwords of 2-bit codes -> four dwords of nibbles per plane: the codes at positions 4 k and 4 k + 1 (2 and 3) of
// a dword share an output byte, low and high nibble, each through one byte look-up (v_perm) with the plane's table —
// the order of the samples inside the lane differs from the file's, identically for both operands, and a contraction
// does not see it.  Sums leave the registers as integers (exact conversion) through the same fp64 epilogue.
// Measured at C5 (profiles/r05_ld.txt, one box): bed_ld_scores 251 ms against 292 ms on the int8 kernel (- 14 %),
// bed_cor 297 against 339 ms; counters: 1.198e9 MFMA per launch of 4 096 blocks at 16.0 pipe cycles each, matrix pipe 42 %
// busy at 2.36 GHz, 6.65 VALU per MFMA — the decode, not the matrix pipe, paces it now.  A square
// 64 x 64 block per workgroup (each wave 16 row variants x all 64 column variants: 4.2 VALU per MFMA) was built and
// measured: 265 ms — twice the LDS operand reads per wave and ten spilled registers cost more than the decode it saves.
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr uint32_t kF4X = 0x00040200u, kF4Xh = 0x00402000u;     // code 0, 1, 2, 3 -> 0, 1.0, 2.0, 0 (low / high nibble)
constexpr uint32_t kF4M = 0x00020202u, kF4Mh = 0x00202020u;     //                 -> 1.0, 1.0, 1.0, 0
struct PlanesF4 {
  v4i x, x2, m;
};
constexpr uint32_t kF4X2 = 0x00060200u, kF4X2h = 0x00602000u;   //                 -> 0, 1.0, 4.0, 0
__device__ __forceinline__ PlanesF4 decode_f4(uint32_t w0, uint32_t w1) {
  const uint32_t a0 = w0 & 0x03030303u, a1 = (w0 >> 2) & 0x03030303u, a2 = (w0 >> 4) & 0x03030303u, a3 = (w0 >> 6) & 0x03030303u;
  const uint32_t b0 = w1 & 0x03030303u, b1 = (w1 >> 2) & 0x03030303u, b2 = (w1 >> 4) & 0x03030303u, b3 = (w1 >> 6) & 0x03030303u;
  PlanesF4 p;
  p.x = v4i{(int)(lut4b(kF4X, a0) | lut4b(kF4Xh, a1)), (int)(lut4b(kF4X, a2) | lut4b(kF4Xh, a3)),
            (int)(lut4b(kF4X, b0) | lut4b(kF4Xh, b1)), (int)(lut4b(kF4X, b2) | lut4b(kF4Xh, b3))};
  // (x^2 from x — 1.0 stays, 2.0 = 0b0100 becomes 4.0 = 0b0110: x | ((x >> 1) & 0x2222...), two instructions instead of
  // three — keeps x alive longer: ten spilled registers at three waves per SIMD, 305 instead of 251 ms.  Measured, not kept.)
  p.x2 = v4i{(int)(lut4b(kF4X2, a0) | lut4b(kF4X2h, a1)), (int)(lut4b(kF4X2, a2) | lut4b(kF4X2h, a3)),
             (int)(lut4b(kF4X2, b0) | lut4b(kF4X2h, b1)), (int)(lut4b(kF4X2, b2) | lut4b(kF4X2h, b3))};
  p.m = v4i{(int)(lut4b(kF4M, a0) | lut4b(kF4Mh, a1)), (int)(lut4b(kF4M, a2) | lut4b(kF4Mh, a3)),
            (int)(lut4b(kF4M, b0) | lut4b(kF4Mh, b1)), (int)(lut4b(kF4M, b2) | lut4b(kF4Mh, b3))};
  return p;
}
// (Round 6, second session) The same six sums from planes that cost no look-up.  A 2-bit code in the low bits of a nibble IS
// code / 2 in E2M1 (0, 0.5, 1.0, 1.5 for the codes 0, 1, 2, 3 = missing), its high bit alone is 1.0 = [code >= 2], and both
// bits set is [code = 3]: with c = the raw code, H = [c >= 2], M = [c = 3] every function of a code is a combination of
// 1, c, H, M — the allele count is x = c - 3 M, its square x^2 = c + 2 H - 5 M, "present" is 1 - M — so
//   sum x y      = cc' - 3 cM' - 3 Mc' + 9 MM'          sum x (1 - M')   = Sx  - (cM' - 3 MM')
//   sum x^2 (1 - M') = Sxx - (cM' + 2 HM' - 5 MM')      sum (1 - M)(1 - M') = Nx + Ny - npos + MM'      (and the mirror images)
// with the per-variant totals Sx, Sxx, N (non-missing count) over the selected samples: SIX products again — (c, c'),
// (c, M'), (H, M'), (M, c'), (M, H'), (M, M') — of planes that take 9 shift / and instructions per 16 genotypes where the
// look-ups of decode_f4 take 25.  M is coded 1.0; the accumulators hold cc' / 4, cM' / 2, HM', ..., MM': multiples of
// 1 / 4, exact in fp32 while 9 n < 2^24 (n <= 1 864 135; beyond that the look-up kernel, then the int8 kernel).  Dropped
// samples are ORed to code 3 as before.  The recombination in the epilogue is integer arithmetic in fp64: the six sums —
// and with them every band entry — are the ones the look-up kernel produces, bit for bit (tests/test_gpu_ld.py).
__device__ __forceinline__ PlanesF4 decode_f4_raw(uint32_t w0, uint32_t w1, bool with_h) {
  const uint32_t c0 = w0 & 0x33333333u, c1 = (w0 >> 2) & 0x33333333u, c2 = w1 & 0x33333333u, c3 = (w1 >> 2) & 0x33333333u;
  PlanesF4 p;
  p.x = v4i{(int)c0, (int)c1, (int)c2, (int)c3};
  if (with_h)
    p.x2 = v4i{(int)(c0 & 0x22222222u), (int)(c1 & 0x22222222u), (int)(c2 & 0x22222222u), (int)(c3 & 0x22222222u)};
  else
    p.x2 = v4i{0, 0, 0, 0};
  // both bits of a code set: bit 1 of the nibble survives c & (c << 1) (the upper two bits of a nibble of c are zero, so the
  // shift carries nothing into a neighbour): 1.0 for a missing code, 0 otherwise
  p.m = v4i{(int)(c0 & (c0 << 1)), (int)(c1 & (c1 << 1)), (int)(c2 & (c2 << 1)), (int)(c3 & (c3 << 1))};
  return p;
}
__device__ __forceinline__ v4f mfma_f4(const v4i &a, const v4i &b, const v4f &c) {
  const v8i a8 = {a.x, a.y, a.z, a.w, 0, 0, 0, 0}, b8 = {b.x, b.y, b.z, b.w, 0, 0, 0, 0};
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, c, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}
// SQ = false (round 6): the bed clumping formula (mode 3, src/clumping-bed.cpp:69-73 on mean-imputed scaled values) is
// sum x y - c' sum x m' - c sum m y + c c' sum m m' — FOUR of the six sums: the two with squares are neither multiplied nor
// decoded (a third of the matrix instructions and of the look-ups; their accumulators are not allocated).
// MASK = false (RAW only): every sample of the image is selected — pad samples are code 0 in the image, and code 0 adds nothing
// to any of the six raw products (c = H = M = 0), so the keep-mask is neither loaded nor ORed in (npos = n then).
// PRIO > 0: s_setprio(PRIO) around the matrix instructions of a K-step (with their LDS reads), 0 again for the decode — a wave that
// has its operands ready is issued ahead of the waves that are still decoding (measured: profiles/r06_ld_raw.txt).
template <bool SQ, bool RAW = false, bool MASK = true, int PRIO = 0>
__global__ __launch_bounds__(256, SQ ? 3 : 4) void k_pair_stats_f4(const uint8_t *__restrict__ img, int64_t pitch,
                                                       const int32_t *__restrict__ cols,
                                                       const int2 *__restrict__ pairs,
                                                       const uint32_t *__restrict__ rowmask, BandOut bo) {
  __shared__ uint4 sB[2][2][2][3][64];   // [buffer][K-step][sub-tile][plane x, x2, m][lane]: 24 KB
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r16 = lane & 15, g = lane >> 4;
  const int2 pr = pairs[blockIdx.x];
  const int64_t ca0 = cols[pr.x * TR], cb0 = cols[pr.y * TC];
  const int my_ks = __builtin_amdgcn_readfirstlane(wave >> 1), my_s = __builtin_amdgcn_readfirstlane(wave & 1);
  uint32_t va[2];
#pragma unroll
  for (int s = 0; s < 2; s++) va[s] = (uint32_t)(((int64_t)cols[pr.x * TR + wave * 32 + s * 16 + r16] - ca0) * pitch + g * 16);
  const uint32_t vb = (uint32_t)(((int64_t)cols[pr.y * TC + my_s * 16 + r16] - cb0) * pitch + g * 16);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)(img + ca0 * pitch), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)(img + cb0 * pitch), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc((void *)rowmask, 0, 0x7fffffff, 0x00020000);
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  typedef unsigned int v2u __attribute__((ext_vector_type(2)));
  v4f acc[2][2][6];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int p = 0; p < 6; p++) acc[i][j][p] = v4f{0.f, 0.f, 0.f, 0.f};
  const int nit = (int)(pitch / 64);
  static_assert(MASK || RAW, "the look-up planes count pad samples as present: they need the keep-mask");
  auto stash = [&](int buf, const v2u &b, const v4u &mk) {   // this wave's (K-step, sub-tile) of the column operand -> LDS
    const uint32_t m0 = !MASK ? ~0u : my_ks == 0 ? mk.x : mk.z, m1 = !MASK ? ~0u : my_ks == 0 ? mk.y : mk.w;
    const PlanesF4 P = RAW ? decode_f4_raw(b.x | ~m0, b.y | ~m1, SQ) : decode_f4(b.x | ~m0, b.y | ~m1);
    sB[buf][my_ks][my_s][0][lane] = uint4{(uint32_t)P.x[0], (uint32_t)P.x[1], (uint32_t)P.x[2], (uint32_t)P.x[3]};
    if constexpr (SQ) sB[buf][my_ks][my_s][1][lane] = uint4{(uint32_t)P.x2[0], (uint32_t)P.x2[1], (uint32_t)P.x2[2], (uint32_t)P.x2[3]};
    sB[buf][my_ks][my_s][2][lane] = uint4{(uint32_t)P.m[0], (uint32_t)P.m[1], (uint32_t)P.m[2], (uint32_t)P.m[3]};
  };
  v4u a[2], mk, an[2], mkn;
  v2u b, bn;
  const int wsel = my_ks * 8;
#pragma unroll
  for (int s = 0; s < 2; s++) a[s] = __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)va[s], 0, 0);
  b = __builtin_amdgcn_raw_buffer_load_b64(rsB, (int)vb, wsel, 0);
  if constexpr (MASK) mk = __builtin_amdgcn_raw_buffer_load_b128(rsM, g * 16, 0, 0);
  else mk = v4u{~0u, ~0u, ~0u, ~0u};
  mkn = mk;
  stash(0, b, mk);
  __syncthreads();
  for (int it = 0; it < nit; it++) {
    const int kbn = (it + 1 < nit ? it + 1 : it) * 64;   // (past the end the last one again: no branch around loads)
#pragma unroll
    for (int s = 0; s < 2; s++) an[s] = __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)va[s], kbn, 0);
    bn = __builtin_amdgcn_raw_buffer_load_b64(rsB, (int)vb, kbn + wsel, 0);
    if constexpr (MASK) mkn = __builtin_amdgcn_raw_buffer_load_b128(rsM, g * 16, kbn, 0);
    const int buf = it & 1;
#pragma unroll
    for (int d = 0; d < 2; d++) {
      const uint32_t m0 = !MASK ? ~0u : d == 0 ? mk.x : mk.z, m1 = !MASK ? ~0u : d == 0 ? mk.y : mk.w;
      PlanesF4 A[2];
#pragma unroll
      for (int s = 0; s < 2; s++)
        A[s] = RAW ? decode_f4_raw((d == 0 ? a[s].x : a[s].z) | ~m0, (d == 0 ? a[s].y : a[s].w) | ~m1, SQ)
                   : decode_f4((d == 0 ? a[s].x : a[s].z) | ~m0, (d == 0 ? a[s].y : a[s].w) | ~m1);
      if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const uint4 bx = sB[buf][d][j][0][lane], bm = sB[buf][d][j][2][lane];
        uint4 bx2 = {0u, 0u, 0u, 0u};
        if constexpr (SQ) bx2 = sB[buf][d][j][1][lane];
        const v4i Bx = {(int)bx.x, (int)bx.y, (int)bx.z, (int)bx.w}, Bx2 = {(int)bx2.x, (int)bx2.y, (int)bx2.z, (int)bx2.w},
                  Bm = {(int)bm.x, (int)bm.y, (int)bm.z, (int)bm.w};
#pragma unroll
        for (int i = 0; i < 2; i++) {
          acc[i][j][0] = mfma_f4(A[i].x, Bx, acc[i][j][0]);
          acc[i][j][1] = mfma_f4(A[i].x, Bm, acc[i][j][1]);
          if constexpr (SQ) acc[i][j][2] = mfma_f4(A[i].x2, Bm, acc[i][j][2]);
          acc[i][j][3] = mfma_f4(A[i].m, Bx, acc[i][j][3]);
          if constexpr (SQ) acc[i][j][4] = mfma_f4(A[i].m, Bx2, acc[i][j][4]);
          acc[i][j][5] = mfma_f4(A[i].m, Bm, acc[i][j][5]);
        }
      }
      if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(0);
    }
    stash(buf ^ 1, bn, mkn);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; s++) a[s] = an[s];
    mk = mkn;
  }
  int32_t st[16][6];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int p = 0; p < 6; p++) {
          // RAW: the accumulators hold cc' / 4, cM' / 2, HM', Mc' / 2, MH', MM' (exact: powers of two back to integers)
          const float sc = !RAW ? 1.f : p == 0 ? 4.f : (p == 1 || p == 3) ? 2.f : 1.f;
          st[(i * 2 + j) * 4 + r][p] = (int32_t)(acc[i][j][p][r] * sc);
        }
#pragma unroll 1
  for (int e = 0; e < 16; e++) {
    const int i = e >> 3, j = (e >> 2) & 1, r = e & 3;
    const int row = wave * 32 + i * 16 + 4 * g + r, col = j * 16 + r16;
    const int64_t j0 = (int64_t)pr.x * TR + row, jj = (int64_t)pr.y * TC + col;
    if (j0 >= bo.m || jj >= j0 || jj < bo.lo[j0]) continue;
    if constexpr (RAW) {
      // integers in fp64 (all below 2^53): the six pairwise-complete sums from the six raw products and the totals
      const double cc = st[e][0], cm = st[e][1], hm = st[e][2], mc = st[e][3], mh = st[e][4], mm = st[e][5];
      const double xy = cc - 3.0 * cm - 3.0 * mc + 9.0 * mm;
      const double xs = bo.cx[j0] - (cm - 3.0 * mm), ys = bo.cx[jj] - (mc - 3.0 * mm);
      const double xx = SQ ? bo.cxx[j0] - (cm + 2.0 * hm - 5.0 * mm) : 0.0, yy = SQ ? bo.cxx[jj] - (mc + 2.0 * mh - 5.0 * mm) : 0.0;
      const int nona = (int)(bo.cnn[j0] + bo.cnn[jj] - bo.npos + mm);
      bo.band[j0 * bo.W + (j0 - jj - 1)] = pair_value(bo.mode, xy, xs, xx, ys, yy, nona, bo.thr, bo.v1, bo.v2, j0, jj, bo.nrows);
    } else {
      bo.band[j0 * bo.W + (j0 - jj - 1)] =
          pair_value(bo.mode, (double)st[e][0], (double)st[e][1], (double)st[e][2], (double)st[e][3], (double)st[e][4],
                     st[e][5], bo.thr, bo.v1, bo.v2, j0, jj, bo.nrows);
    }
  }
}

// Cross product only (variants without missing values, FBM clumping): one wave owns the whole
// 64 x 64 tile pair for its share of the samples (4 x 4 MFMA sub-tiles, so each decoded operand
// feeds four MFMAs instead of two) and adds its int32 partial with atomics (K is split over
// blockIdx.y; integer sums are order-independent).  stats plane 0 must be zeroed by the caller.
__global__ __launch_bounds__(64) void k_pair_xy64(const uint8_t *__restrict__ img, int64_t pitch,
                                                  const int32_t *__restrict__ cols,
                                                  const int2 *__restrict__ pairs,
                                                  const uint32_t *__restrict__ rowmask,
                                                  int64_t kbytes_per_split, int32_t *__restrict__ stats) {
  const int lane = threadIdx.x;
  const int r16 = lane & 15, g = lane >> 4;
  const int2 pr = pairs[blockIdx.x];
  const uint8_t *pa[4], *pb[4];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    pa[s] = img + (int64_t)cols[pr.x * TB + s * 16 + r16] * pitch + g * 16;
    pb[s] = img + (int64_t)cols[pr.y * TB + s * 16 + r16] * pitch + g * 16;
  }
  int64_t b0 = (int64_t)blockIdx.y * kbytes_per_split, b1 = b0 + kbytes_per_split;
  if (b1 > pitch) b1 = pitch;
  if (b0 >= b1) return;
  v4i acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = v4i{0, 0, 0, 0};
  auto decode_x = [](uint32_t w) {
    const uint32_t s0 = w & 0x03030303u, s1 = (w >> 2) & 0x03030303u, s2 = (w >> 4) & 0x03030303u,
                   s3 = (w >> 6) & 0x03030303u;
    return v4i{(int)lut4b(kLX, s0), (int)lut4b(kLX, s1), (int)lut4b(kLX, s2), (int)lut4b(kLX, s3)};
  };
  uint4 a[4], b[4], an[4], bn[4], mk, mkn;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    a[s] = *(const uint4 *)(pa[s] + b0);
    b[s] = *(const uint4 *)(pb[s] + b0);
  }
  mk = *(const uint4 *)((const uint8_t *)rowmask + b0 + g * 16);
  for (int64_t kb = b0; kb < b1; kb += 64) {
    const int64_t kn = kb + 64 < b1 ? kb + 64 : kb;  // branch-free prefetch of the next step
#pragma unroll
    for (int s = 0; s < 4; s++) {
      an[s] = *(const uint4 *)(pa[s] + kn);
      bn[s] = *(const uint4 *)(pb[s] + kn);
    }
    mkn = *(const uint4 *)((const uint8_t *)rowmask + kn + g * 16);
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const uint32_t mw = d == 0 ? mk.x : d == 1 ? mk.y : d == 2 ? mk.z : mk.w;
      v4i A[4], B[4];
#pragma unroll
      for (int s = 0; s < 4; s++) {
        uint32_t wa = d == 0 ? a[s].x : d == 1 ? a[s].y : d == 2 ? a[s].z : a[s].w;
        uint32_t wb = d == 0 ? b[s].x : d == 1 ? b[s].y : d == 2 ? b[s].z : b[s].w;
        wa |= ~mw;  // dropped samples become missing (plane value 0)
        wb |= ~mw;
        A[s] = decode_x(wa);
        B[s] = decode_x(wb);
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i], B[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 4; s++) {
      a[s] = an[s];
      b[s] = bn[s];
    }
    mk = mkn;
  }
  int32_t *out = stats + (int64_t)blockIdx.x * 6 * TB * TB;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = i * 16 + 4 * g + r, col = j * 16 + r16;
        atomicAdd(out + row * TB + col, acc[i][j][r]);
      }
}

// (round 6) The cross product alone on the FP4 matrix pipe.  With no missing value among the selected samples the codes
// are the allele counts, and a 2-bit code in the low bits of a nibble IS code / 2 in E2M1 (0 -> 0, 1 -> 0.5, 2 -> 1.0): the
// operand of 32 samples is  w & 0x3333.., (w >> 2) & 0x3333..  of two dwords — THREE instructions per 16 genotypes where the
// int8 kernel spends twelve on its byte look-ups — and one v_mfma_scale_f32_16x16x128_f8f6f4 contracts 128 samples in the 16
// pipe cycles the int8 instruction needs for 64.  The sums are sum g g' / 4: exact in fp32 while 4 n < 2^24 (the host takes
// the int8 kernel beyond 4 194 303 samples, or with BSN_LD_I8=1); dropped samples are ANDed to code 0.  Same tile pairs,
// same K split, same integer statistics as k_pair_xy64: the band, the clumping bits and the LD scores are bit-identical.
// MASK = false when every sample of the image is selected: the pad samples are code 0 in the image and add nothing to a
// cross product, so the keep-mask (two ANDs per operand dword: a quarter of the loop's vector instructions) is not read.
// Three waves per SIMD are asked for by attribute: left to itself the compiler splits the kernel's ~ 150 registers between
// the two register files and moves accumulators back and forth (56 v_accvgpr_* per 32 MFMA; 6.1 -> 4.4 -> 3.4 VALU per MFMA).
template <bool MASK>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_pair_xy_f4(const uint8_t *__restrict__ img, int64_t pitch,
                                                   const int32_t *__restrict__ cols,
                                                   const int2 *__restrict__ pairs,
                                                   const uint32_t *__restrict__ rowmask,
                                                   int64_t kbytes_per_split, int32_t *__restrict__ stats) {
  const int lane = threadIdx.x;
  const int r16 = lane & 15, g = lane >> 4;
  const int2 pr = pairs[blockIdx.x];
  const uint8_t *pa[4], *pb[4];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    pa[s] = img + (int64_t)cols[pr.x * TB + s * 16 + r16] * pitch + g * 16;
    pb[s] = img + (int64_t)cols[pr.y * TB + s * 16 + r16] * pitch + g * 16;
  }
  int64_t b0 = (int64_t)blockIdx.y * kbytes_per_split, b1 = b0 + kbytes_per_split;   // (multiples of 128)
  if (b1 > pitch) b1 = pitch;
  if (b0 >= b1) return;
  v4f acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
  // A step is 128 bytes of every row — one whole cache line: the two halves are asked for by consecutive instructions (a
  // step of 64 bytes left the other half of 128 lines per wave to an L1 that holds 256: every line came from L2 twice)
  uint4 a[4][2], b[4][2], an[4][2], bn[4][2], mk[2] = {{~0u, ~0u, ~0u, ~0u}, {~0u, ~0u, ~0u, ~0u}}, mkn[2] = {mk[0], mk[1]};
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int h = 0; h < 2; h++) {
      a[s][h] = *(const uint4 *)(pa[s] + b0 + 64 * h);
      b[s][h] = *(const uint4 *)(pb[s] + b0 + 64 * h);
    }
  if constexpr (MASK) {
    mk[0] = *(const uint4 *)((const uint8_t *)rowmask + b0 + g * 16);
    mk[1] = *(const uint4 *)((const uint8_t *)rowmask + b0 + 64 + g * 16);
  }
  auto nib = [](uint32_t w0, uint32_t w1) {
    return v4i{(int)(w0 & 0x33333333u), (int)((w0 >> 2) & 0x33333333u), (int)(w1 & 0x33333333u), (int)((w1 >> 2) & 0x33333333u)};
  };
  for (int64_t kb = b0; kb < b1; kb += 128) {
    const int64_t kn = kb + 128 < b1 ? kb + 128 : kb;  // branch-free prefetch of the next step
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
      for (int h = 0; h < 2; h++) {
        an[s][h] = *(const uint4 *)(pa[s] + kn + 64 * h);
        bn[s][h] = *(const uint4 *)(pb[s] + kn + 64 * h);
      }
    if constexpr (MASK) {
      mkn[0] = *(const uint4 *)((const uint8_t *)rowmask + kn + g * 16);
      mkn[1] = *(const uint4 *)((const uint8_t *)rowmask + kn + 64 + g * 16);
    }
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int d = 0; d < 2; d++) {   // a lane's 16 bytes are two K-steps of 32 samples (x 4 lane groups = 128)
        const uint32_t m0 = d == 0 ? mk[h].x : mk[h].z, m1 = d == 0 ? mk[h].y : mk[h].w;
        v4i A[4], B[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
          if constexpr (MASK) {
            A[s] = nib((d == 0 ? a[s][h].x : a[s][h].z) & m0, (d == 0 ? a[s][h].y : a[s][h].w) & m1);   // dropped samples: code 0
            B[s] = nib((d == 0 ? b[s][h].x : b[s][h].z) & m0, (d == 0 ? b[s][h].y : b[s][h].w) & m1);
          } else {
            A[s] = nib(d == 0 ? a[s][h].x : a[s][h].z, d == 0 ? a[s][h].y : a[s][h].w);
            B[s] = nib(d == 0 ? b[s][h].x : b[s][h].z, d == 0 ? b[s][h].y : b[s][h].w);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[i][j] = mfma_f4(A[i], B[j], acc[i][j]);
      }
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
      for (int h = 0; h < 2; h++) {
        a[s][h] = an[s][h];
        b[s][h] = bn[s][h];
      }
    mk[0] = mkn[0];
    mk[1] = mkn[1];
  }
  int32_t *out = stats + (int64_t)blockIdx.x * 6 * TB * TB;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = i * 16 + 4 * g + r, col = j * 16 + r16;
        atomicAdd(out + row * TB + col, __float2int_rn(4.0f * acc[i][j][r]));   // (code / 2)(code' / 2) summed: exact quarters
      }
}

// The same cross product for FOUR tile pairs at once (round 6): a workgroup of four waves takes the 2 x 2 block of pairs
// (I0, I0 + 1) x (J0, J0 + 1); every wave brings ONE of the four 64-variant tiles from memory — 128 bytes of each row per step,
// whole cache lines — into LDS (raw 2-bit bytes, double-buffered, one barrier per step), and reads the two tiles of its own
// pair from there.  k_pair_xy_f4 is bound by the vector memory path, not by its arithmetic (8 KB through L1 per 32 MFMA and
// wave = the 64 B per clock and CU the L1 delivers at best; counters: matrix pipe 23 % busy, 92 % of the requests hit L2,
// HBM at 1.2 TB/s — profiles/r06_ld_pmc.txt — and neither three waves per SIMD nor 3.3 instead of 6.1 VALU per MFMA moved
// it); the block halves the bytes per MFMA.  quads[q]: {tiles I0, I1, J0, J1; pair indices of (I0,J0) (I0,J1) (I1,J0) (I1,J1)
// in this batch, -1 = not in the band}.  Same K order per pair within a split, integer-exact sums: bit-identical output.
struct QuadXY {
  int t[4], p[4];
};
template <bool MASK, int PRIO = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_quad_xy_f4(const uint8_t *__restrict__ img, int64_t pitch,
                                                                                              const int32_t *__restrict__ cols,
                                                                                              const QuadXY *__restrict__ quads,
                                                                                              const uint32_t *__restrict__ rowmask,
                                                                                              int64_t kbytes_per_split,
                                                                                              int32_t *__restrict__ stats) {
  __shared__ uint4 sT[2][4][4][2][64];   // [buffer][tile][16-row group][half line][lane]: 2 x 32 KB
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r16 = lane & 15, g = lane >> 4;
  const int my_pair = __builtin_amdgcn_readfirstlane(quads[blockIdx.x].p[wave]), my_tile = __builtin_amdgcn_readfirstlane(quads[blockIdx.x].t[wave]);
  const int ta = wave >> 1, tb = 2 + (wave & 1);
  const uint8_t *pt[4];
#pragma unroll
  for (int s = 0; s < 4; s++) pt[s] = img + (int64_t)cols[my_tile * TB + s * 16 + r16] * pitch + g * 16;
  int64_t b0 = (int64_t)blockIdx.y * kbytes_per_split, b1 = b0 + kbytes_per_split;   // (multiples of 128)
  if (b1 > pitch) b1 = pitch;
  if (b0 >= b1) return;
  v4f acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
  // two register sets: the loads of step t + 2 are issued while step t is multiplied and the set of step t + 1 — issued a
  // whole step earlier — goes to LDS at its end (one step of matrix work is ~ 1 000 cycles, an L2 miss twice that)
  uint4 ldA[4][2], ldB[4][2];
#define BSN_QUAD_FETCH(DST, KB)                                                                         \
  _Pragma("unroll") for (int s = 0; s < 4; s++) _Pragma("unroll") for (int h = 0; h < 2; h++) {          \
    uint4 v = *(const uint4 *)(pt[s] + (KB) + 64 * h);                                                  \
    if constexpr (MASK) { /* dropped samples: code 0 (the tile is masked once, by the wave that brings it) */ \
      const uint4 mk = *(const uint4 *)((const uint8_t *)rowmask + (KB) + 64 * h + g * 16);             \
      v.x &= mk.x;                                                                                      \
      v.y &= mk.y;                                                                                      \
      v.z &= mk.z;                                                                                      \
      v.w &= mk.w;                                                                                      \
    }                                                                                                   \
    DST[s][h] = v;                                                                                      \
  }
#define BSN_QUAD_STASH(SRC, BUF)                                                                        \
  _Pragma("unroll") for (int s = 0; s < 4; s++) _Pragma("unroll") for (int h = 0; h < 2; h++) mine[((BUF) * 4 * 4 * 2 + s * 2 + h) * 64] = SRC[s][h];
  uint4 *const mine = &sT[0][wave][0][0][lane];   // this wave's tile, this lane's slot
  auto nib = [](uint32_t w0, uint32_t w1) {
    return v4i{(int)(w0 & 0x33333333u), (int)((w0 >> 2) & 0x33333333u), (int)(w1 & 0x33333333u), (int)((w1 >> 2) & 0x33333333u)};
  };
#define BSN_QUAD_MULTIPLY(BUF)                                                                          \
  if (my_pair >= 0) {                                                                                   \
    _Pragma("unroll") for (int h = 0; h < 2; h++) {                                                     \
      uint4 a[4], b[4];                                                                                 \
      _Pragma("unroll") for (int s = 0; s < 4; s++) {                                                   \
        a[s] = sT[BUF][ta][s][h][lane];                                                                 \
        b[s] = sT[BUF][tb][s][h][lane];                                                                 \
      }                                                                                                 \
      _Pragma("unroll") for (int d = 0; d < 2; d++) { /* a lane's 16 bytes are two K-steps of 32 samples (x 4 lane groups = 128) */ \
        v4i A[4], B[4];                                                                                 \
        _Pragma("unroll") for (int s = 0; s < 4; s++) {                                                 \
          A[s] = nib(d == 0 ? a[s].x : a[s].z, d == 0 ? a[s].y : a[s].w);                               \
          B[s] = nib(d == 0 ? b[s].x : b[s].z, d == 0 ? b[s].y : b[s].w);                               \
        }                                                                                               \
        if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);                                         \
        _Pragma("unroll") for (int i = 0; i < 4; i++) _Pragma("unroll") for (int j = 0; j < 4; j++) acc[i][j] = mfma_f4(A[i], B[j], acc[i][j]); \
        if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(0);                                            \
      }                                                                                                 \
    }                                                                                                   \
  }
  const int64_t last = b1 - 128;
  auto at = [&](int64_t kb) { return kb < last ? kb : last; };   // (branch-free: past the end the last step again)
  BSN_QUAD_FETCH(ldA, b0)
  BSN_QUAD_STASH(ldA, 0)
  BSN_QUAD_FETCH(ldA, at(b0 + 128))
  __syncthreads();
  for (int64_t kb = b0; kb < b1; kb += 256) {
    // step at kb: LDS buffer 0 holds it, ldA holds the next one
    BSN_QUAD_FETCH(ldB, at(kb + 256))
    BSN_QUAD_MULTIPLY(0)
    BSN_QUAD_STASH(ldA, 1)
    __syncthreads();
    if (kb + 128 >= b1) break;
    // step at kb + 128: buffer 1, ldB holds the next one
    BSN_QUAD_FETCH(ldA, at(kb + 384))
    BSN_QUAD_MULTIPLY(1)
    BSN_QUAD_STASH(ldB, 0)
    __syncthreads();
  }
  if (my_pair < 0) return;
  int32_t *out = stats + (int64_t)my_pair * 6 * TB * TB;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = i * 16 + 4 * g + r, col = j * 16 + r16;
        atomicAdd(out + row * TB + col, __float2int_rn(4.0f * acc[i][j][r]));   // (code / 2)(code' / 2) summed: exact quarters
      }
}

#undef BSN_QUAD_FETCH
#undef BSN_QUAD_STASH
#undef BSN_QUAD_MULTIPLY

// Byte image (dosage grid, bsn_bed::bits == 8): the cross product of the grid indices, sum_i k_i k'_i,
// for a 64 x 64 tile pair.  The loaded bytes are the MFMA operands (no decode; `rowmask` zeroes the
// samples that are not selected, and the pad samples).  |k| <= 127, so one int32 accumulator holds at most
// 131 072 samples: the sample range is cut into slices of that size, slice s of a pair accumulates into
// plane s of its statistics block (atomics over the K splits inside a slice), and the fill kernel adds
// the slices in fp64 (exact).  Only data without missing values take this path.
__global__ __launch_bounds__(64) void k_pair_xy8(const uint8_t *__restrict__ img, int64_t pitch,
                                                 const int32_t *__restrict__ cols,
                                                 const int2 *__restrict__ pairs,
                                                 const uint8_t *__restrict__ rowmask, int64_t kbytes_per_split,
                                                 int splits_per_slice, int32_t *__restrict__ stats) {
  const int lane = threadIdx.x;
  const int r16 = lane & 15, g = lane >> 4;
  const int2 pr = pairs[blockIdx.x];
  const uint8_t *pa[4], *pb[4];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    pa[s] = img + (int64_t)cols[pr.x * TB + s * 16 + r16] * pitch + g * 16;
    pb[s] = img + (int64_t)cols[pr.y * TB + s * 16 + r16] * pitch + g * 16;
  }
  const int64_t b0 = (int64_t)blockIdx.y * kbytes_per_split;
  int64_t b1 = b0 + kbytes_per_split;
  if (b1 > pitch) b1 = pitch;
  if (b0 >= b1) return;
  v4i acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = v4i{0, 0, 0, 0};
  for (int64_t kb = b0; kb < b1; kb += 64) {  // 64 samples per step
    const uint4 mk = *(const uint4 *)(rowmask + kb + g * 16);
    v4i A[4], B[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const uint4 a = *(const uint4 *)(pa[s] + kb), b = *(const uint4 *)(pb[s] + kb);
      A[s] = v4i{(int)(a.x & mk.x), (int)(a.y & mk.y), (int)(a.z & mk.z), (int)(a.w & mk.w)};
      B[s] = v4i{(int)(b.x & mk.x), (int)(b.y & mk.y), (int)(b.z & mk.z), (int)(b.w & mk.w)};
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[i], B[j], acc[i][j], 0, 0, 0);
  }
  const int slice = (int)blockIdx.y / splits_per_slice;
  int32_t *out = stats + ((int64_t)blockIdx.x * 6 + slice) * TB * TB;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) atomicAdd(out + (i * 16 + 4 * g + r) * TB + j * 16 + r16, acc[i][j][r]);
}

// byte image: band entries from the sliced cross products and the per-variant totals of the grid
// indices.  Pearson's r is invariant under the affine map value = voff + vstep k, so modes 0 / 1 use the
// reference's expressions on the exact integer sums of k; mode 2 (clumping_chr: the caller's sumX / denoX
// are in value units, src/clumping.cpp:66-73) first maps the cross product to value units.
__global__ void k_band_fill8(const int32_t *__restrict__ stats, int nslice, const int2 *__restrict__ pairs,
                             int npairs, int64_t m, const int64_t *__restrict__ lo, int64_t W,
                             const double *__restrict__ thr, int mode, const double *__restrict__ v1,
                             const double *__restrict__ v2, double nrows, double *__restrict__ band,
                             const double *__restrict__ cx, const double *__restrict__ cxx, double vstep,
                             double voff, const double *__restrict__ nna) {
  const int pi = blockIdx.x;
  if (pi >= npairs) return;
  const int2 pr = pairs[pi];
  const int32_t *st = stats + (int64_t)pi * 6 * TB * TB;
  for (int e = threadIdx.x; e < TB * TB; e += blockDim.x) {
    const int row = e / TB, col = e % TB;
    const int64_t j0 = (int64_t)pr.x * TB + row, j = (int64_t)pr.y * TB + col;
    if (j0 >= m || j >= j0 || j < lo[j0]) continue;
    double xy = 0;
    for (int s = 0; s < nslice; s++) xy += (double)st[(s * TB + row) * TB + col];
    if (mode == 2) xy = vstep * vstep * xy + voff * (v1[j] + v1[j0]) - nrows * voff * voff;
    double val = pair_value(mode, xy, cx[j0], cxx[j0], cx[j], cxx[j], (int)nrows, thr, v1, v2, j0, j, nrows);
    // clumping_chr on an FBM with missing values: the reference's xySum is NA (a missing dosage decodes to
    // NA_real, src/clumping.cpp:66-69), r2 is NA and `r2 > thr` is false
    if (nna && (nna[j0] > 0 || nna[j] > 0)) val = __longlong_as_double(0x7ff8000000000000LL);
    band[j0 * W + (j0 - j - 1)] = val;
  }
}

// ---- byte image WITH missing values: the six pairwise-complete sums of corMat0 (src/corr.cpp:54-75) --------
// The reference recodes a missing dosage to 3 and walks the same loop (src/corr.cpp:113-118); here the marker
// -128 gives the mask plane M (1 = present) and leaves X = k with the marker zeroed; k^2 <= 16 129 does not fit
// an int8 operand, so it enters as two digit planes k^2 = 128 H + L (0 <= H <= 126, 0 <= L <= 127).  Eight
// exact int8 products per tile pair, in four groups of two (blockIdx.z) so that a wave's accumulators fit its
// registers:   0: X.X', M.M'   1: X.M', M.X'   2: H.M', L.M'   3: M.H', M.L'
// A workgroup's sample range never crosses a 131 072-sample slice (127^2 x 131 072 < 2^31); its int32 sums are
// added to int64 statistics (exact, order-independent), stats64[pair][8][64][64].
__device__ __forceinline__ uint32_t val8m_ld(uint32_t w, uint32_t &na) {
  const uint32_t t = w ^ 0x80808080u;
  const uint32_t y = (t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
  const uint32_t z = ~(y | t | 0x7F7F7F7Fu);  // 0x80 iff the byte of w is 0x80
  na = z >> 7;
  return w & ~(z | (z - na));
}
// per-byte k -> (k^2 >> 7, k^2 & 127)
__device__ __forceinline__ void square8(uint32_t w, uint32_t &h, uint32_t &l) {
  h = 0;
  l = 0;
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const int k = (int)(int8_t)(w >> (8 * b));
    const uint32_t sq = (uint32_t)(k * k);
    h |= (sq >> 7) << (8 * b);
    l |= (sq & 127u) << (8 * b);
  }
}
struct Planes8 {
  v4i x, m, h, l;
};
template <bool SQ>
__device__ __forceinline__ Planes8 decode8(uint4 a, uint4 mk) {
  Planes8 p;
  uint32_t n0, n1, n2, n3;
  // samples that are not selected (and the pad samples) count as missing
  const uint32_t x0 = val8m_ld(a.x, n0) & mk.x, x1 = val8m_ld(a.y, n1) & mk.y, x2 = val8m_ld(a.z, n2) & mk.z,
                 x3 = val8m_ld(a.w, n3) & mk.w;
  p.x = v4i{(int)x0, (int)x1, (int)x2, (int)x3};
  p.m = v4i{(int)((n0 ^ 0x01010101u) & mk.x), (int)((n1 ^ 0x01010101u) & mk.y), (int)((n2 ^ 0x01010101u) & mk.z),
            (int)((n3 ^ 0x01010101u) & mk.w)};
  if (SQ) {
    uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
    square8(x0, h0, l0); square8(x1, h1, l1); square8(x2, h2, l2); square8(x3, h3, l3);
    p.h = v4i{(int)h0, (int)h1, (int)h2, (int)h3};
    p.l = v4i{(int)l0, (int)l1, (int)l2, (int)l3};
  }
  return p;
}
__global__ __launch_bounds__(64) void k_pair_stats8(const uint8_t *__restrict__ img, int64_t pitch,
                                                    const int32_t *__restrict__ cols,
                                                    const int2 *__restrict__ pairs,
                                                    const uint8_t *__restrict__ rowmask, int64_t kbytes_per_split,
                                                    long long *__restrict__ stats) {
  const int lane = threadIdx.x, r16 = lane & 15, g = lane >> 4, grp = blockIdx.z;
  const int2 pr = pairs[blockIdx.x];
  const uint8_t *pa[4], *pb[4];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    pa[s] = img + (int64_t)cols[pr.x * TB + s * 16 + r16] * pitch + g * 16;
    pb[s] = img + (int64_t)cols[pr.y * TB + s * 16 + r16] * pitch + g * 16;
  }
  const int64_t b0 = (int64_t)blockIdx.y * kbytes_per_split;
  int64_t b1 = b0 + kbytes_per_split;
  if (b1 > pitch) b1 = pitch;
  if (b0 >= b1) return;
  v4i acc[2][4][4];
#pragma unroll
  for (int q = 0; q < 2; q++)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[q][i][j] = v4i{0, 0, 0, 0};
  for (int64_t kb = b0; kb < b1; kb += 64) {  // 64 samples per step
    const uint4 mk = *(const uint4 *)(rowmask + kb + g * 16);
    v4i A0[4], A1[4], B0[4], B1[4];   // the two operand planes of this group, per 16-variant sub-tile
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const uint4 a = *(const uint4 *)(pa[s] + kb), b = *(const uint4 *)(pb[s] + kb);
      if (grp == 0) {
        const Planes8 A = decode8<false>(a, mk), B = decode8<false>(b, mk);
        A0[s] = A.x; B0[s] = B.x; A1[s] = A.m; B1[s] = B.m;
      } else if (grp == 1) {
        const Planes8 A = decode8<false>(a, mk), B = decode8<false>(b, mk);
        A0[s] = A.x; B0[s] = B.m; A1[s] = A.m; B1[s] = B.x;
      } else if (grp == 2) {
        const Planes8 A = decode8<true>(a, mk), B = decode8<false>(b, mk);
        A0[s] = A.h; B0[s] = B.m; A1[s] = A.l; B1[s] = B.m;
      } else {
        const Planes8 A = decode8<false>(a, mk), B = decode8<true>(b, mk);
        A0[s] = A.m; B0[s] = B.h; A1[s] = A.m; B1[s] = B.l;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        acc[0][i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0[i], B0[j], acc[0][i][j], 0, 0, 0);
        acc[1][i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1[i], B1[j], acc[1][i][j], 0, 0, 0);
      }
  }
#pragma unroll
  for (int q = 0; q < 2; q++) {
    unsigned long long *out = (unsigned long long *)stats + ((int64_t)blockIdx.x * 8 + grp * 2 + q) * TB * TB;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          atomicAdd(out + (i * 16 + 4 * g + r) * TB + j * 16 + r16, (unsigned long long)(long long)acc[q][i][j][r]);
  }
}
// band entries from the eight exact sums (mode 0 / 1: Pearson's r is invariant under value = voff + vstep k)
__global__ void k_band_fill8na(const long long *__restrict__ stats, const int2 *__restrict__ pairs, int npairs,
                               int64_t m, const int64_t *__restrict__ lo, int64_t W, const double *__restrict__ thr,
                               int mode, double nrows, double *__restrict__ band) {
  const int pi = blockIdx.x;
  if (pi >= npairs) return;
  const int2 pr = pairs[pi];
  const long long *st = stats + (int64_t)pi * 8 * TB * TB;
  for (int e = threadIdx.x; e < TB * TB; e += blockDim.x) {
    const int row = e / TB, col = e % TB;
    const int64_t j0 = (int64_t)pr.x * TB + row, j = (int64_t)pr.y * TB + col;
    if (j0 >= m || j >= j0 || j < lo[j0]) continue;
    const double xy = (double)st[0 * TB * TB + e], nona = (double)st[1 * TB * TB + e], xs = (double)st[2 * TB * TB + e],
                 ys = (double)st[3 * TB * TB + e],
                 xx = 128.0 * (double)st[4 * TB * TB + e] + (double)st[5 * TB * TB + e],
                 yy = 128.0 * (double)st[6 * TB * TB + e] + (double)st[7 * TB * TB + e];
    band[j0 * W + (j0 - j - 1)] = pair_value(mode, xy, xs, xx, ys, yy, (int)nona, thr, nullptr, nullptr, j0, j, nrows);
  }
}

// second stage of the K-split / cross-product-only paths: statistics buffer -> band entries
__global__ void k_band_fill(const int32_t *__restrict__ stats, const int2 *__restrict__ pairs,
                            int npairs, int64_t m, const int64_t *__restrict__ lo, int64_t W,
                            const double *__restrict__ thr, int mode, const double *__restrict__ v1,
                            const double *__restrict__ v2, double nrows, double *__restrict__ band,
                            const double *__restrict__ cx, const double *__restrict__ cxx) {
  const int pi = blockIdx.x;
  if (pi >= npairs) return;
  const int2 pr = pairs[pi];
  const int32_t *st = stats + (int64_t)pi * 6 * TB * TB;
  for (int e = threadIdx.x; e < TB * TB; e += blockDim.x) {
    const int row = e / TB, col = e % TB;
    const int64_t j0 = (int64_t)pr.x * TB + row, j = (int64_t)pr.y * TB + col;
    if (j0 >= m || j >= j0 || j < lo[j0]) continue;
    // cx != NULL: no missing value among the selected samples of any variant, the five
    // one-sided sums are the per-variant totals (same integers the six-product path would produce)
    const double xySum = st[(0 * TB + row) * TB + col];
    const double xSum = cx ? cx[j0] : (double)st[(1 * TB + row) * TB + col],
                 xxSum = cx ? cxx[j0] : (double)st[(2 * TB + row) * TB + col],
                 ySum = cx ? cx[j] : (double)st[(3 * TB + row) * TB + col],
                 yySum = cx ? cxx[j] : (double)st[(4 * TB + row) * TB + col];
    const int nona = cx ? (int)nrows : st[(5 * TB + row) * TB + col];
    band[j0 * W + (j0 - j - 1)] = pair_value(mode, xySum, xSum, xxSum, ySum, yySum, nona, thr, v1, v2, j0, j, nrows);
  }
}

// one wave per column j0 of [c0, c1): counts kept entries (ascending j), + diagonal; cnt is indexed from c0 and so
// is the band (BandJob::band_at)
__global__ void k_cor_count(const double *__restrict__ band, const int64_t *__restrict__ lo, int64_t W,
                            int64_t c0, int64_t c1, int fill_diag, int32_t *__restrict__ cnt) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t j0 = c0 + (int64_t)blockIdx.x * 4 + wave;
  if (j0 >= c1) return;
  const int64_t width = j0 - lo[j0];
  int c = 0;
  for (int64_t w = lane; w < width; w += 64) c += band[j0 * W + w] != 2.0;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
  if (lane == 0) cnt[j0 - c0] = c + (fill_diag ? 1 : 0);
}

// p[j0 - c0] = first slot of column j0 in oi / ox (offsets within this block of columns)
__global__ void k_cor_fill(const double *__restrict__ band, const int64_t *__restrict__ lo, int64_t W,
                           int64_t c0, int64_t c1, int fill_diag, const int64_t *__restrict__ p,
                           int32_t *__restrict__ oi, double *__restrict__ ox, int *__restrict__ nan_seen) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t j0 = c0 + (int64_t)blockIdx.x * 4 + wave;
  if (j0 >= c1) return;
  const int64_t width = j0 - lo[j0];
  int64_t base = p[j0 - c0];
  // ascending j = descending w
  for (int64_t t0 = 0; t0 < width; t0 += 64) {
    const int64_t t = t0 + lane;  // t-th smallest j: w = width - 1 - t
    double v = 2.0;
    if (t < width) v = band[j0 * W + (width - 1 - t)];
    const bool keep = (t < width) && (v != 2.0);
    const unsigned long long mask = __ballot(keep);
    if (keep) {
      const int off = __popcll(mask & ((1ull << lane) - 1ull));
      oi[base + off] = (int32_t)(lo[j0] + t);
      ox[base + off] = v;
      if (v != v) *nan_seen = 1;   // (every writer stores the same value)
    }
    base += __popcll(mask);
  }
  if (fill_diag && lane == 0) {
    oi[base] = (int32_t)j0;
    ox[base] = 1.0;
  }
}

// ld[j] = 1 + sum over pairs containing j of r2 (NaN skipped), deterministic order:
// own row first (j' < j, descending j'), then later columns j0 > j in ascending order.
// The band holds the columns [c0, c1) (BandJob::band_at); blocks come in ascending order, so a variant's sum is
// started by the block that holds its own column and continued — same terms, same order, bit for bit — by the
// blocks after it whose columns still reach it.  One thread per variant j in [jlo, c1), jlo = the first variant any
// column of the block reaches.
__global__ void k_ld_sum(const double *__restrict__ band, const int64_t *__restrict__ lo, int64_t W,
                         int64_t m, int64_t jlo, int64_t c0, int64_t c1, double *__restrict__ ld) {
  const int64_t j = jlo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= c1) return;
  double s;
  if (j >= c0) {
    s = 1.0;
    const int64_t width = j - lo[j];
    for (int64_t w = 0; w < width; w++) {
      const double v = band[j * W + w];
      if (!isnan(v)) s += v;
    }
  } else {
    s = ld[j];
  }
  for (int64_t j0 = j + 1 > c0 ? j + 1 : c0; j0 < c1 && lo[j0] <= j; j0++) {
    const double v = band[j0 * W + (j0 - j - 1)];
    if (!isnan(v)) s += v;
  }
  ld[j] = s;
}

// bit w of the 64-bit word (j0 * Wq + w / 64) = band[j0 * W + w] > thr (NaN -> 0; slots outside
// the window of j0 -> 0).  One workgroup per j0; this is what the host sweep of the
// clumping functions reads (1/64 of the band's bytes per threshold).
__global__ void k_band_gt(const double *__restrict__ band, const int64_t *__restrict__ lo, int64_t W,
                          int64_t Wq, double thr, unsigned long long *__restrict__ bits) {
  const int64_t j0 = blockIdx.x;
  const int64_t width = j0 - lo[j0];
  for (int64_t w0 = (int64_t)(threadIdx.x & ~63); w0 < Wq * 64; w0 += blockDim.x) {
    const int64_t w = w0 + (threadIdx.x & 63);
    const bool gt = (w < width) && (band[j0 * W + w] > thr);
    const unsigned long long b = __ballot(gt);
    if ((threadIdx.x & 63) == 0) bits[j0 * Wq + (w0 >> 6)] = b;
  }
}

// same for the LATER neighbours of j0: bit w <-> j' = j0 + 1 + w, i.e. band[j' * W + w] > thr when
// j' exists and j0 is inside its window.  With both images the host sweep of the clumping
// functions is two word-wise ANDs against the bit set of kept variants per variant.
__global__ void k_band_gt_upper(const double *__restrict__ band, const int64_t *__restrict__ lo, int64_t W,
                                int64_t Wq, int64_t m, double thr, unsigned long long *__restrict__ bits) {
  const int64_t j0 = blockIdx.x;
  for (int64_t w0 = (int64_t)(threadIdx.x & ~63); w0 < Wq * 64; w0 += blockDim.x) {
    const int64_t w = w0 + (threadIdx.x & 63), j = j0 + 1 + w;
    const bool gt = (w < W) && (j < m) && (lo[j] <= j0) && (band[j * W + w] > thr);
    const unsigned long long b = __ballot(gt);
    if ((threadIdx.x & 63) == 0) bits[j0 * Wq + (w0 >> 6)] = b;
  }
}

// ---------------------------------------------------------------------------------------
// what bench.py --workload ld reports: set by every band_run
struct LdStats {
  double pairs = 0, tile_pairs = 0, stats_ms = 0, launches = 0;
  int kernel = 0;  // 0: six-product kernel with fused epilogue, 1: six-product + K split, 2: cross product only, 3: byte image with missing values (eight products), 4: six products, column operand decoded once per workgroup (LDS)
};
static LdStats g_ld_stats;

struct BandJob {
  bsn_bed *bed = nullptr;
  int64_t n = 0, m = 0, W = 1;
  std::vector<int64_t> lo;
  DevBuf<int32_t> d_cols, d_stats;
  DevBuf<int2> d_pairs, d_pairs_b;   // 64 x 64 tile pairs; 128 x 32 blocks of k_pair_stats_b
  DevBuf<int64_t> d_lo;
  DevBuf<uint32_t> d_mask;
  DevBuf<uint8_t> d_mask8;  // byte image: 0xFF per selected sample
  DevBuf<double> d_band, d_thr, d_v1, d_v2, d_cx, d_cxx, d_nna;   // d_nna: byte image, missing values per variant
  DevBuf<double> d_cnn;     // 2-bit image: non-missing count per variant over the selected samples (raw-plane kernel)
  DevBuf<long long> d_stats64;
  bool complete = false;  // no missing value among the selected samples of the selected variants
  bool use_mask = false;
  bool all_rows = false;  // every sample of the image is selected (2-bit image)
  bool contig = false;    // every tile's variants lie within 2 GB of its first one, in ascending order
  int64_t npairs = 0, npairs_b = 0;
  // The band is held for `chunk_cols` columns at a time (a multiple of 128; >= m: the whole band, one chunk).  The
  // reference walks any window (src/corr.cpp:52-53); a dense m x W fp64 band of a long chromosome with ONE wide
  // region does not fit HBM (1M variants x 40 000 = 320 GB), so snp_cor / snp_ld_scores process the band in blocks of
  // columns that fit a budget (free device memory, or BSN_LD_BAND_BUDGET bytes), compacting each block to its CSC
  // columns / adding its LD-score terms before the next — same kernels, same order of every sum.
  int64_t chunk_cols = 0;
  std::vector<int2> pairs_host;                   // the 64 x 64 tile pairs, as on the device
  DevBuf<QuadXY> d_quads;
  std::vector<int64_t> pair_start, pairb_start;   // first tile pair of every 64- / 128-column block of rows (+ end)
  double *band_at(int64_t c0) const { return d_band.p - c0 * W; }   // indexed with absolute columns >= c0
};

// lo[j0] = first j with pos[j] >= pos[j0] - size (src/corr.cpp:52-53).  For clumping the
// window is two-sided (src/clumping-utils.h:21-22); by symmetry the pair (j0, j), j < j0, is
// needed iff pos[j] >= pos[j0] - size, the same lower bound.
static void window_bounds(const double *pos, int64_t m, double size, bool two_sided,
                          std::vector<int64_t> &lo) {
  lo.resize((size_t)m);
  int64_t l = 0;
  for (int64_t j0 = 0; j0 < m; j0++) {
    if (j0 > 0 && pos[j0] < pos[j0 - 1]) fail("'pos' is not sorted.");
    const double pos_min = pos[j0] - size;
    // pos is sorted, so the reference's downward walk from j0-1 stops exactly at the first
    // j with pos[j] < pos_min.  For clumping the pair is also reached from j's side with
    // `pos[j0] <= pos[j] + size`; both spellings are honoured so that rounding cannot make
    // the host sweep ask for a pair outside the band.
    while (l < j0 && pos[l] < pos_min && !(two_sided && pos[j0] <= pos[l] + size)) l++;
    lo[(size_t)j0] = l;
  }
}

static void band_stats(BandJob &J, bsn_bed *bed, const int64_t *ind_row, int64_t n,
                       const int64_t *ind_col, int64_t m, const double *pos, double size,
                       bool two_sided = false, bool allow_chunks = false, double out_bytes_per_entry = 0.0) {
  if (n <= 0 || m <= 0) fail("'ind.row' and 'ind.col' can't be empty.");
  refuse_generic(bed, "windowed LD (snp_cor / snp_ld_scores / snp_clumping)");
  require_resident(bed, "windowed LD (snp_cor / snp_ld_scores / snp_clumping)");
  BSN_HIP(hipSetDevice(bed->device));
  J.bed = bed;
  J.n = n;
  J.m = m;
  window_bounds(pos, m, size, two_sided, J.lo);
  int64_t W = 1;
  for (int64_t j0 = 0; j0 < m; j0++) W = std::max(W, j0 - J.lo[(size_t)j0]);
  J.W = W;
  {
    // the band (columns x W fp64) lives in HBM next to the image: the whole of it when it fits what the device has
    // free (minus room for the statistics buffers and — snp_cor — the compacted result), else blocks of columns
    const double need = (double)m * (double)W * 8.0;
    double budget = 32e9;
    if (need > budget || getenv("BSN_LD_BAND_BUDGET")) {
      size_t free_b = 0, total_b = 0;
      BSN_HIP(hipMemGetInfo(&free_b, &total_b));
      free_b += dev_cache_held();
      const double avail = (double)free_b - 4e9;
      // a result that keeps every pair needs 12 bytes per band entry on top of the band itself
      budget = allow_chunks ? std::max(1e9, avail * (out_bytes_per_entry > 0 ? 8.0 / (8.0 + out_bytes_per_entry) : 1.0)) : avail;
      if (const char *e = getenv("BSN_LD_BAND_BUDGET")) budget = atof(e);
      if (need > budget && !allow_chunks)
        fail("LD band of %lld x %lld (%.1f GB) does not fit the %.1f GB of free device memory; use a smaller "
             "window or call it per chromosome / per block of variants",
             (long long)m, (long long)W, need / 1e9, (double)free_b / 1e9);
    }
    J.chunk_cols = (m + TR - 1) / TR * TR;
    if (need > budget) {
      J.chunk_cols = std::max<int64_t>(TR, (int64_t)(budget / ((double)W * 8.0)) / TR * TR);
      if ((double)J.chunk_cols * (double)W * 8.0 > budget * 1.0001 && !getenv("BSN_LD_BAND_BUDGET"))
        fail("LD window of %lld variants: even 128 columns of the band (%.1f GB) do not fit the free device memory",
             (long long)W, 128.0 * (double)W * 8.0 / 1e9);
    }
  }
  // columns (padded to the tile size with a valid column)
  const int64_t mt = (m + TB - 1) / TB, m_pad = (m + TR - 1) / TR * TR;   // (a multiple of 64 and of 128)
  std::vector<int32_t> cols((size_t)m_pad);
  for (int64_t j = 0; j < m_pad; j++) {
    int64_t c = ind_col ? ind_col[j < m ? j : m - 1] : (j < m ? j : m - 1);
    if (c < 0 || c >= bed->m) fail("Tested %lld < %lld. Subscript out of bounds (ind.col).", (long long)c, (long long)bed->m);
    cols[(size_t)j] = (int32_t)c;
  }
  copy_h2d(bed, J.d_cols.ensure((size_t)m_pad), cols.data(), (size_t)m_pad * 4);
  J.contig = true;   // offsets from the first variant of every 128-block (hence of every 64- and 32-block) fit 31 bits
  for (int64_t t = 0; t * TR < m_pad && J.contig; t++)
    for (int64_t j = t * TR; j < (t + 1) * TR; j++) {
      const int64_t d = (int64_t)cols[(size_t)j] - cols[(size_t)(t * TR)];
      if (d < 0 || (d + 1) * bed->pitch >= ((int64_t)1 << 31) || (j > t * TR && cols[(size_t)j] < cols[(size_t)(j - 1)]))
        J.contig = false;
    }
  if (bed->bits == 8) {
    // byte image: byte mask of the selected samples, per-variant totals of the grid indices over them
    std::vector<uint8_t> mask((size_t)bed->pitch, 0);
    std::vector<int32_t> rows32((size_t)n);
    bool ident = (n == bed->n);
    for (int64_t i = 0; i < n; i++) {
      int64_t r = ind_row ? ind_row[i] : i;
      if (r < 0 || r >= bed->n) fail("Tested %lld < %lld. Subscript out of bounds (ind.row).", (long long)r, (long long)bed->n);
      if (mask[(size_t)r]) fail("internal: repeated samples reached the keep-mask (row_view)");
      mask[(size_t)r] = 0xFF;
      rows32[(size_t)i] = (int32_t)r;
      ident = ident && r == i;
    }
    copy_h2d(bed, J.d_mask8.ensure(mask.size()), mask.data(), mask.size());
    DevBuf<int32_t> d_rows;
    if (!ident) copy_h2d(bed, d_rows.ensure((size_t)n), rows32.data(), (size_t)n * 4);
    DevBuf<long long> d_st;
    stats8(bed, ident ? nullptr : d_rows.p, n, J.d_cols.p, 0, m, d_st.ensure((size_t)3 * m));
    std::vector<long long> st((size_t)3 * m);
    copy_d2h(bed, st.data(), d_st.p, st.size() * 8);
    std::vector<double> cx((size_t)m), cxx((size_t)m), nna((size_t)m);
    long long na_total = 0;
    for (int64_t j = 0; j < m; j++) {
      na_total += st[(size_t)(3 * j + 2)];
      nna[(size_t)j] = (double)st[(size_t)(3 * j + 2)];
      cx[(size_t)j] = (double)st[(size_t)(3 * j)];
      cxx[(size_t)j] = (double)st[(size_t)(3 * j + 1)];
    }
    copy_h2d(bed, J.d_nna.ensure((size_t)m), nna.data(), (size_t)m * 8);
    J.complete = na_total == 0;
    J.use_mask = true;
    copy_h2d(bed, J.d_cx.ensure((size_t)m), cx.data(), (size_t)m * 8);
    copy_h2d(bed, J.d_cxx.ensure((size_t)m), cxx.data(), (size_t)m * 8);
  } else {
    // rows: keep-mask, 2 bits per sample (also removes the pad samples, which are coded as
    // non-missing genotype 0 in the image).  Lists with repeated samples never get here (row_view).
    J.use_mask = true;
    {
      std::vector<uint32_t> mask((size_t)(bed->pitch / 4), 0u);
      for (int64_t i = 0; i < n; i++) {
        int64_t r = ind_row ? ind_row[i] : i;
        if (r < 0 || r >= bed->n) fail("Tested %lld < %lld. Subscript out of bounds (ind.row).", (long long)r, (long long)bed->n);
        uint32_t bit = 3u << (2 * (r & 15));
        if (mask[(size_t)(r >> 4)] & bit) fail("internal: repeated samples reached the keep-mask (row_view)");
        mask[(size_t)(r >> 4)] |= bit;
      }
      copy_h2d(bed, J.d_mask.ensure(mask.size()), mask.data(), mask.size() * 4);
      BSN_HIP(hipStreamSynchronize(bed->stream));
      J.all_rows = n == bed->n;   // (no sample twice: all of them) — the FP4 cross-product kernel then skips the mask
    }
    // per-variant totals over the selected samples; when nothing is missing there, Sum x, Sum x^2 and
    // the pair count of every pair are these totals and only the cross product needs the GEMM
    {
      std::vector<int32_t> cnt((size_t)4 * m);
      counts_host(bed, ind_row, n, ind_col, m, cnt.data());
      int64_t na = 0;
      std::vector<double> cx((size_t)m), cxx((size_t)m), cnn((size_t)m);
      for (int64_t j = 0; j < m; j++) {
        const int32_t *c = &cnt[(size_t)4 * j];
        na += c[3];
        cx[(size_t)j] = (double)c[1] + 2.0 * c[2];
        cxx[(size_t)j] = (double)c[1] + 4.0 * c[2];
        cnn[(size_t)j] = (double)c[0] + (double)c[1] + (double)c[2];
      }
      J.complete = (na == 0) && !getenv("BSN_FORCE_NA_PLANE");
      // (with missing values the raw-plane kernel of the six sums reads the totals too)
      copy_h2d(bed, J.d_cx.ensure((size_t)m), cx.data(), (size_t)m * 8);
      copy_h2d(bed, J.d_cxx.ensure((size_t)m), cxx.data(), (size_t)m * 8);
      copy_h2d(bed, J.d_cnn.ensure((size_t)m), cnn.data(), (size_t)m * 8);
      BSN_HIP(hipStreamSynchronize(bed->stream));
    }
  }
  // tile pairs of the band
  std::vector<int2> pairs;
  J.pair_start.assign((size_t)mt + 1, 0);
  for (int64_t I = 0; I < mt; I++) {
    J.pair_start[(size_t)I] = (int64_t)pairs.size();
    int64_t Jlo = J.lo[(size_t)(I * TB)] / TB;
    for (int64_t Jt = Jlo; Jt <= I; Jt++) pairs.push_back(int2{(int)I, (int)Jt});
  }
  J.pair_start[(size_t)mt] = (int64_t)pairs.size();
  J.npairs = (int64_t)pairs.size();
  copy_h2d(bed, J.d_pairs.ensure(pairs.size()), pairs.data(), pairs.size() * sizeof(int2));
  J.pairs_host = pairs;
  {
    // the same band in blocks of 128 row variants x 32 column variants (k_pair_stats_b)
    std::vector<int2> pb;
    J.pairb_start.clear();
    for (int64_t I = 0; I * TR < m; I++) {
      J.pairb_start.push_back((int64_t)pb.size());
      const int64_t last = std::min<int64_t>(m, (I + 1) * TR) - 1;
      for (int64_t Jb = J.lo[(size_t)(I * TR)] / TC; Jb <= last / TC; Jb++) pb.push_back(int2{(int)I, (int)Jb});
    }
    J.pairb_start.push_back((int64_t)pb.size());
    J.npairs_b = (int64_t)pb.size();
    copy_h2d(bed, J.d_pairs_b.ensure(pb.size()), pb.data(), pb.size() * sizeof(int2));
  }
  copy_h2d(bed, J.d_lo.ensure((size_t)m), J.lo.data(), (size_t)m * 8);
  BSN_HIP(hipStreamSynchronize(bed->stream));  // host vectors go out of scope
  // statistics in batches of tile pairs (bounded scratch)
  J.d_band.ensure((size_t)std::min<int64_t>(J.chunk_cols, (m + TR - 1) / TR * TR) * (size_t)W);
}

// runs the statistics + band fill in batches; mode / aux as in pair_value
// Columns [c0, c1) of the band (c0 a multiple of 128; the whole band by default) go to J.d_band, whose first column
// is then c0.  `accumulate`: add to the statistics of the previous block instead of starting them.
static void band_run(BandJob &J, int mode, const double *d_thr, const double *d_v1, const double *d_v2,
                     double nrows, int64_t c0 = 0, int64_t c1 = -1, bool accumulate = false) {
  bsn_bed *bed = J.bed;
  if (c1 < 0 || c1 > J.m) c1 = J.m;
  if (c0 % TR != 0 || c1 - c0 > J.chunk_cols) fail("internal: LD band block [%lld, %lld)", (long long)c0, (long long)c1);
  const int64_t batch = 4096;  // 4096 x 6 x 64 x 64 x 4 B = 403 MB of int32 statistics (two-stage paths only)
  const bool xy_only = J.complete || mode == 2;  // FBM clumping (mode 2) reads the cross product only (src/clumping.cpp:66-73)
  // the tile pairs whose row variants (the j0 side) lie in the block: a contiguous run of both pair lists
  const int64_t pA = J.pair_start[(size_t)(c0 / TB)], pB = J.pair_start[(size_t)((c1 + TB - 1) / TB)];
  const int64_t qA = J.pairb_start[(size_t)(c0 / TR)], qB = J.pairb_start[(size_t)((c1 + TR - 1) / TR)];
  LdStats ls;
  if (accumulate) ls = g_ld_stats;
  ls.tile_pairs += (double)(pB - pA);
  for (int64_t j0 = c0; j0 < c1; j0++) ls.pairs += (double)(j0 - J.lo[(size_t)j0]);
  hipEvent_t e0, e1;
  BSN_HIP(hipEventCreate(&e0));
  BSN_HIP(hipEventCreate(&e1));
  float ms_total = accumulate ? (float)g_ld_stats.stats_ms : 0.f;
  double *const band = J.band_at(c0);
  BandOut bo{J.m, J.W, J.d_lo.p, d_thr, d_v1, d_v2, nrows, band, mode};
  if (bed->bits == 8) {
    if (mode == 3) fail("internal: the bed clumping formula does not apply to a byte image");
    const int64_t slice_bytes = 131072;  // samples per int32 accumulator slice
    const int nslice = (int)((bed->pitch + slice_bytes - 1) / slice_bytes);
    if (nslice > 6) fail("windowed LD on a dosage FBM supports at most %lld samples", (long long)(6 * slice_bytes));
    if (!J.complete && mode != 2) {
      // missing values among the selected samples: the six pairwise-complete sums (eight int8 products)
      const int64_t batch8 = 512;   // 512 x 8 x 64 x 64 x 8 B = 134 MB of int64 statistics
      J.d_stats64.ensure((size_t)std::min(batch8, J.npairs) * 8 * TB * TB);
      for (int64_t p0 = pA; p0 < pB; p0 += batch8) {
        const int64_t np = std::min(batch8, pB - p0);
        // enough workgroups to fill the chip; a split never crosses a 131 072-sample slice
        int64_t ks = std::max<int64_t>(nslice, std::min<int64_t>(std::max<int64_t>(1, 4096 / (np * 4)), bed->pitch / 256));
        int64_t kb = round_up((bed->pitch + ks - 1) / ks, 64);
        if (kb > slice_bytes) kb = slice_bytes;
        ks = (bed->pitch + kb - 1) / kb;
        BSN_HIP(hipMemsetAsync(J.d_stats64.p, 0, (size_t)np * 8 * TB * TB * 8, bed->stream));
        BSN_HIP(hipEventRecord(e0, bed->stream));
        hipLaunchKernelGGL(k_pair_stats8, dim3((unsigned)np, (unsigned)ks, 4), dim3(64), 0, bed->stream, bed->d_img,
                           bed->pitch, J.d_cols.p, J.d_pairs.p + p0, J.d_mask8.p, kb, J.d_stats64.p);
        BSN_HIP(hipGetLastError());
        BSN_HIP(hipEventRecord(e1, bed->stream));
        hipLaunchKernelGGL(k_band_fill8na, dim3((unsigned)np), dim3(256), 0, bed->stream, J.d_stats64.p,
                           J.d_pairs.p + p0, (int)np, J.m, J.d_lo.p, J.W, d_thr, mode, nrows, band);
        BSN_HIP(hipGetLastError());
        BSN_HIP(hipEventSynchronize(e1));
        float ms = 0;
        BSN_HIP(hipEventElapsedTime(&ms, e0, e1));
        ms_total += ms;
        ls.launches += 1;
      }
      ls.kernel = 3;   // byte image, eight products
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
      ls.stats_ms = ms_total;
      g_ld_stats = ls;
      return;
    }
    J.d_stats.ensure((size_t)std::min(batch, J.npairs) * 6 * TB * TB);
    for (int64_t p0 = pA; p0 < pB; p0 += batch) {
      const int64_t np = std::min(batch, pB - p0);
      // K splits never straddle a slice: one slice -> any 64-byte-aligned split of the row; several ->
      // a power-of-two number of splits per 131 072-byte slice
      const int64_t sl = std::min<int64_t>(slice_bytes, bed->pitch);
      const int64_t want = std::max<int64_t>(1, 8192 / (np * nslice));
      int sps;
      int64_t kb;
      if (nslice == 1) {
        sps = (int)std::min<int64_t>(want, sl / 256);
        if (sps < 1) sps = 1;
        kb = round_up((sl + sps - 1) / sps, 64);
        sps = (int)((sl + kb - 1) / kb);
      } else {
        sps = 1;
        while (sps * 2 <= want && sps < 512) sps *= 2;
        kb = sl / sps;
      }
      BSN_HIP(hipMemsetAsync(J.d_stats.p, 0, (size_t)np * 6 * TB * TB * 4, bed->stream));
      BSN_HIP(hipEventRecord(e0, bed->stream));
      hipLaunchKernelGGL(k_pair_xy8, dim3((unsigned)np, (unsigned)(sps * nslice)), dim3(64), 0, bed->stream,
                         bed->d_img, bed->pitch, J.d_cols.p, J.d_pairs.p + p0, J.d_mask8.p, kb, sps, J.d_stats.p);
      BSN_HIP(hipGetLastError());
      BSN_HIP(hipEventRecord(e1, bed->stream));
      hipLaunchKernelGGL(k_band_fill8, dim3((unsigned)np), dim3(256), 0, bed->stream, J.d_stats.p, nslice,
                         J.d_pairs.p + p0, (int)np, J.m, J.d_lo.p, J.W, d_thr, mode, d_v1, d_v2, nrows, band,
                         J.d_cx.p, J.d_cxx.p, bed->v_step, bed->v_off, J.complete ? (const double *)nullptr : J.d_nna.p);
      BSN_HIP(hipGetLastError());
      BSN_HIP(hipEventSynchronize(e1));
      float ms = 0;
      BSN_HIP(hipEventElapsedTime(&ms, e0, e1));
      ms_total += ms;
      ls.launches += 1;
    }
    ls.kernel = 2;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    ls.stats_ms = ms_total;
    g_ld_stats = ls;
    return;
  }
  if (!xy_only && J.contig && J.use_mask && J.npairs_b >= 1024 && !abl_getenv("BSN_LD_NO_SHARED_DECODE")) {
    // enough blocks to fill the chip without a K split: the kernel that shares the column operand's decode.
    // (round 6, second session) ONE launch for the whole run: the fused epilogue needs no scratch, every block walks the same
    // sample range (equal work), and a launch of 4 096 blocks on 768 resident workgroups ended with a round that was a third
    // full — 13 such tails at C5 (BSN_LD_BATCH=<n> in the profiling build: blocks per launch, for the A/B)
    int64_t fbatch = std::max<int64_t>(qB - qA, 1);
    if (const char *e = abl_getenv("BSN_LD_BATCH")) fbatch = std::max<int64_t>(1, atoll(e));
    for (int64_t p0 = qA; p0 < qB; p0 += fbatch) {
      const int64_t np = std::min(fbatch, qB - p0);
      BSN_HIP(hipEventRecord(e0, bed->stream));
      // the FP4 matrix pipe while the sums stay exact in fp32 (at most 4 n < 2^24); BSN_LD_I8=1: the int8 kernel
      static const bool i8_only = getenv("BSN_LD_I8") != nullptr;
      const bool f4 = bed->pitch * 4 <= 4194303 && !i8_only;
      // planes without look-ups while 9 n < 2^24 (BSN_LD_LUT=1: the look-up kernel, for the A/B)
      static const bool lut_only = getenv("BSN_LD_LUT") != nullptr;
      const bool raw = f4 && bed->pitch * 4 <= 1864135 && !lut_only && J.d_cnn.p != nullptr;
      ls.kernel = f4 ? (mode == 3 ? 9 : 6) : 4;
      if (raw) {
        BandOut br = bo;
        // every sample selected: no keep-mask (the pad samples are code 0 and add nothing to the raw products)
        const bool nomask = J.all_rows && !abl_getenv("BSN_LD_RAW_MASK");
        int prio = 2;   // (profiling build: BSN_LD_RAW_PRIO=0..3)
        if (const char *e = abl_getenv("BSN_LD_RAW_PRIO")) prio = atoi(e);
        (void)prio;
        br.cx = J.d_cx.p, br.cxx = J.d_cxx.p, br.cnn = J.d_cnn.p, br.npos = nomask ? (double)bed->n : (double)(bed->pitch * 4);
        ls.kernel = mode == 3 ? 11 : 10;
#define BSN_LAUNCH_RAW(SQ_, MASK_, PRIO_)                                                                                      \
  hipLaunchKernelGGL((k_pair_stats_f4<SQ_, true, MASK_, PRIO_>), dim3((unsigned)np), dim3(256), 0, bed->stream, bed->d_img, \
                     bed->pitch, J.d_cols.p, J.d_pairs_b.p + p0, J.d_mask.p, br)
#ifdef BSN_ABLATION
#define BSN_LAUNCH_RAW_P(SQ_, MASK_)                  \
  do {                                                \
    if (prio == 0) BSN_LAUNCH_RAW(SQ_, MASK_, 0);     \
    else if (prio == 1) BSN_LAUNCH_RAW(SQ_, MASK_, 1); \
    else if (prio == 3) BSN_LAUNCH_RAW(SQ_, MASK_, 3); \
    else BSN_LAUNCH_RAW(SQ_, MASK_, 2);               \
  } while (0)
#else
#define BSN_LAUNCH_RAW_P(SQ_, MASK_) BSN_LAUNCH_RAW(SQ_, MASK_, 2)
#endif
        if (mode == 3) {
          if (nomask) BSN_LAUNCH_RAW_P(false, false);
          else BSN_LAUNCH_RAW_P(false, true);
        } else {
          if (nomask) BSN_LAUNCH_RAW_P(true, false);
          else BSN_LAUNCH_RAW_P(true, true);
        }
#undef BSN_LAUNCH_RAW_P
#undef BSN_LAUNCH_RAW
      } else if (abl_getenv("BSN_LD_NOPRIO")) {   // (profiling build: the look-up / int8 kernels without the priority, for the A/B)
        if (f4 && mode == 3)
          hipLaunchKernelGGL(k_pair_stats_f4<false>, dim3((unsigned)np), dim3(256), 0, bed->stream, bed->d_img, bed->pitch,
                             J.d_cols.p, J.d_pairs_b.p + p0, J.d_mask.p, bo);
        else if (f4)
          hipLaunchKernelGGL(k_pair_stats_f4<true>, dim3((unsigned)np), dim3(256), 0, bed->stream, bed->d_img, bed->pitch,
                             J.d_cols.p, J.d_pairs_b.p + p0, J.d_mask.p, bo);
        else
          hipLaunchKernelGGL(k_pair_stats_b<0>, dim3((unsigned)np), dim3(256), 0, bed->stream, bed->d_img, bed->pitch,
                             J.d_cols.p, J.d_pairs_b.p + p0, J.d_mask.p, bo);
      } else if (f4 && mode == 3)   // the bed clumping formula reads four of the six sums
        hipLaunchKernelGGL((k_pair_stats_f4<false, false, true, 2>), dim3((unsigned)np), dim3(256), 0, bed->stream, bed->d_img, bed->pitch,
                           J.d_cols.p, J.d_pairs_b.p + p0, J.d_mask.p, bo);
      else if (f4)
        hipLaunchKernelGGL((k_pair_stats_f4<true, false, true, 2>), dim3((unsigned)np), dim3(256), 0, bed->stream, bed->d_img, bed->pitch,
                           J.d_cols.p, J.d_pairs_b.p + p0, J.d_mask.p, bo);
      else   // (the int8 kernel without the priority: with it the compiler needs 174 registers — two waves per SIMD — and it is 11 % slower)
        hipLaunchKernelGGL(k_pair_stats_b<0>, dim3((unsigned)np), dim3(256), 0, bed->stream, bed->d_img, bed->pitch,
                           J.d_cols.p, J.d_pairs_b.p + p0, J.d_mask.p, bo);
      BSN_HIP(hipGetLastError());
      BSN_HIP(hipEventRecord(e1, bed->stream));
      BSN_HIP(hipEventSynchronize(e1));
      float ms = 0;
      BSN_HIP(hipEventElapsedTime(&ms, e0, e1));
      ms_total += ms;
      ls.launches += 1;
    }
    ls.tile_pairs += (double)(qB - qA) - (double)(pB - pA);   // blocks of 128 x 32 = the area of a 64 x 64 tile pair
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    ls.stats_ms = ms_total;
    g_ld_stats = ls;
    return;
  }
  // (round 6) nothing in this loop waits for the device: the 2 x 2 blocks of tile pairs of ALL batches are grouped and uploaded
  // once, every batch has its own pair of events, and the host reads them after the last launch (a batch used to end with an
  // event wait and its own grouping + upload; measured at a million variants, same box: no difference — the band-fill kernel
  // of a batch covered the host's preparation of the next — kept because it is the simpler order of events)
  const bool xy_f4_all = xy_only && bed->bits == 2 && bed->n <= 4194303 && !getenv("BSN_LD_I8");
  const bool quad_all = xy_f4_all && !getenv("BSN_LD_NO_QUAD");
  std::vector<QuadXY> quads;
  std::vector<int64_t> quad_off;
  if (quad_all) {
    std::unordered_map<uint64_t, int> at;
    const int mt = (int)((J.m + TB - 1) / TB);
    for (int64_t p0 = pA; p0 < pB; p0 += batch) {
      const int64_t np = std::min(batch, pB - p0);
      quad_off.push_back((int64_t)quads.size());
      if (np < 64) continue;
      at.clear();
      const int64_t base = (int64_t)quads.size();
      for (int64_t p = p0; p < p0 + np; p++) {
        const int I = J.pairs_host[(size_t)p].x, Jt = J.pairs_host[(size_t)p].y;
        const uint64_t key = ((uint64_t)(uint32_t)(I >> 1) << 32) | (uint32_t)(Jt >> 1);
        auto it = at.find(key);
        if (it == at.end()) {
          QuadXY qd;
          qd.t[0] = (I >> 1) * 2;
          qd.t[1] = std::min(qd.t[0] + 1, mt - 1);
          qd.t[2] = (Jt >> 1) * 2;
          qd.t[3] = std::min(qd.t[2] + 1, mt - 1);
          qd.p[0] = qd.p[1] = qd.p[2] = qd.p[3] = -1;
          it = at.emplace(key, (int)((int64_t)quads.size() - base)).first;
          quads.push_back(qd);
        }
        quads[(size_t)(base + it->second)].p[(I & 1) * 2 + (Jt & 1)] = (int)(p - p0);
      }
    }
    quad_off.push_back((int64_t)quads.size());
    if (!quads.empty()) {
      copy_h2d(bed, J.d_quads.ensure(quads.size()), quads.data(), quads.size() * sizeof(QuadXY));
      BSN_HIP(hipStreamSynchronize(bed->stream));   // (quads is a host vector)
    }
  }
  std::vector<hipEvent_t> evs;
  int64_t ib = 0;
  for (int64_t p0 = pA; p0 < pB; p0 += batch, ib++) {
    const int64_t np = std::min(batch, pB - p0);
    hipEvent_t b0 = nullptr, b1 = nullptr;
    BSN_HIP(hipEventCreate(&b0));
    BSN_HIP(hipEventCreate(&b1));
    evs.push_back(b0);
    evs.push_back(b1);
    // K split: enough workgroups to fill the chip when there are few tile pairs
    int ksplit = (int)std::min<int64_t>(std::max<int64_t>(1, 2048 / np), bed->pitch / 256);
    if (ksplit < 1) ksplit = 1;
    int64_t kbytes = round_up((bed->pitch + ksplit - 1) / ksplit, 64);
    ksplit = (int)((bed->pitch + kbytes - 1) / kbytes);
    const bool fused = !xy_only && ksplit == 1 && !abl_getenv("BSN_LD_NOFUSE");   // (A/B switch: K split + k_band_fill)
    if (!fused) {
      J.d_stats.ensure((size_t)std::min(batch, J.npairs) * 6 * TB * TB);
      if (ksplit > 1 || xy_only) BSN_HIP(hipMemsetAsync(J.d_stats.p, 0, (size_t)np * 6 * TB * TB * 4, bed->stream));
    }
    BSN_HIP(hipEventRecord(b0, bed->stream));
    if (xy_only) {
      // one wave per workgroup here: four times the K splits of the 4-wave kernel
      int ks4 = (int)std::min<int64_t>(std::max<int64_t>(4, 8192 / np), bed->pitch / 256);
      if (ks4 < 1) ks4 = 1;
      int64_t kb4 = round_up((bed->pitch + ks4 - 1) / ks4, 128);   // (k_pair_xy_f4 walks whole cache lines)
      ks4 = (int)((bed->pitch + kb4 - 1) / kb4);
      // (round 6) on the FP4 matrix pipe while its fp32 sums are exact: 4 n < 2^24
      const bool xy_f4 = bed->n <= 4194303 && !getenv("BSN_LD_I8");
      // 2 x 2 blocks of tile pairs per workgroup (k_quad_xy_f4) once there are enough of them to fill the chip
      const bool quad = quad_all && np >= 64;
      if (quad) {
        const int64_t nq = quad_off[(size_t)ib + 1] - quad_off[(size_t)ib];
        const QuadXY *d_q = J.d_quads.p + quad_off[(size_t)ib];
        int ksq = (int)std::min<int64_t>(std::max<int64_t>(1, 4096 / nq), bed->pitch / 1024);
        if (ksq < 1) ksq = 1;
        int64_t kbq = round_up((bed->pitch + ksq - 1) / ksq, 128);
        ksq = (int)((bed->pitch + kbq - 1) / kbq);
        if (J.all_rows && abl_getenv("BSN_LD_QUAD_PRIO"))
          hipLaunchKernelGGL((k_quad_xy_f4<false, 2>), dim3((unsigned)nq, (unsigned)ksq), dim3(256), 0, bed->stream, bed->d_img, bed->pitch,
                             J.d_cols.p, d_q, J.d_mask.p, kbq, J.d_stats.p);
        else if (J.all_rows)
          hipLaunchKernelGGL(k_quad_xy_f4<false>, dim3((unsigned)nq, (unsigned)ksq), dim3(256), 0, bed->stream, bed->d_img, bed->pitch,
                             J.d_cols.p, d_q, J.d_mask.p, kbq, J.d_stats.p);
        else
          hipLaunchKernelGGL(k_quad_xy_f4<true>, dim3((unsigned)nq, (unsigned)ksq), dim3(256), 0, bed->stream, bed->d_img, bed->pitch,
                             J.d_cols.p, d_q, J.d_mask.p, kbq, J.d_stats.p);
        ls.kernel = 8;
      } else if (xy_f4 && J.all_rows)
        hipLaunchKernelGGL(k_pair_xy_f4<false>, dim3((unsigned)np, (unsigned)ks4), dim3(64), 0, bed->stream, bed->d_img,
                           bed->pitch, J.d_cols.p, J.d_pairs.p + p0, J.d_mask.p, kb4, J.d_stats.p);
      else if (xy_f4)
        hipLaunchKernelGGL(k_pair_xy_f4<true>, dim3((unsigned)np, (unsigned)ks4), dim3(64), 0, bed->stream, bed->d_img,
                           bed->pitch, J.d_cols.p, J.d_pairs.p + p0, J.d_mask.p, kb4, J.d_stats.p);
      else
        hipLaunchKernelGGL(k_pair_xy64, dim3((unsigned)np, (unsigned)ks4), dim3(64), 0, bed->stream, bed->d_img,
                           bed->pitch, J.d_cols.p, J.d_pairs.p + p0, J.d_mask.p, kb4, J.d_stats.p);
      if (!quad) ls.kernel = xy_f4 ? 7 : 2;
    } else if (fused) {
      if (J.contig)
        hipLaunchKernelGGL((k_pair_stats<true, true, true>), dim3((unsigned)np, 1), dim3(256), 0, bed->stream,
                           bed->d_img, bed->pitch, J.d_cols.p, J.d_pairs.p + p0,
                           J.use_mask ? J.d_mask.p : nullptr, kbytes, (int32_t *)nullptr, 0, bo);
      else
        hipLaunchKernelGGL((k_pair_stats<true, true>), dim3((unsigned)np, 1), dim3(256), 0, bed->stream,
                           bed->d_img, bed->pitch, J.d_cols.p, J.d_pairs.p + p0,
                           J.use_mask ? J.d_mask.p : nullptr, kbytes, (int32_t *)nullptr, 0, bo);
      ls.kernel = 0;
    } else {
      hipLaunchKernelGGL((k_pair_stats<true, false>), dim3((unsigned)np, (unsigned)ksplit), dim3(256), 0, bed->stream,
                         bed->d_img, bed->pitch, J.d_cols.p, J.d_pairs.p + p0,
                         J.use_mask ? J.d_mask.p : nullptr, kbytes, J.d_stats.p, ksplit > 1 ? 1 : 0, bo);
      ls.kernel = 1;
    }
    BSN_HIP(hipGetLastError());
    BSN_HIP(hipEventRecord(b1, bed->stream));
    if (!fused) {
      hipLaunchKernelGGL(k_band_fill, dim3((unsigned)np), dim3(256), 0, bed->stream, J.d_stats.p,
                         J.d_pairs.p + p0, (int)np, J.m, J.d_lo.p, J.W, d_thr, mode, d_v1, d_v2, nrows,
                         band, J.complete ? J.d_cx.p : (const double *)nullptr,
                         J.complete ? J.d_cxx.p : (const double *)nullptr);
      BSN_HIP(hipGetLastError());
    }
    ls.launches += 1;
  }
  if (!evs.empty()) BSN_HIP(hipStreamSynchronize(bed->stream));
  for (size_t t = 0; t + 1 < evs.size(); t += 2) {
    float ms = 0;
    BSN_HIP(hipEventElapsedTime(&ms, evs[t], evs[t + 1]));
    ms_total += ms;
  }
  for (hipEvent_t e : evs) (void)hipEventDestroy(e);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  ls.stats_ms = ms_total;
  g_ld_stats = ls;
}

}  // namespace bsn

using namespace bsn;

// ind.row with repeated samples (a bootstrap draw; the reference's accessors take any list, src/bed-acc.h:64-65): the
// band kernels select samples through a keep-mask, which cannot count a sample twice, so the selected sub-matrix is
// materialised once (rows in list order, the selected variants only) and the job runs on that image with identity
// lists.  Every statistic is an integer sum over samples: the order of the rows changes nothing.
struct RowView {
  bsn_bed *bed;
  const int64_t *ind_row, *ind_col;
  std::shared_ptr<bsn_bed> owned;
};
static RowView row_view(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m) {
  RowView v{bed, ind_row, ind_col, nullptr};
  if (!ind_row || n <= 1 || m <= 0 || bed->generic) return v;
  std::vector<uint64_t> seen((size_t)(bed->n + 63) / 64, 0ull);
  bool dup = false;
  for (int64_t i = 0; i < n && !dup; i++) {
    const int64_t r = ind_row[i];
    if (r < 0 || r >= bed->n) return v;   // reported by the job, with the reference's message
    dup = (seen[(size_t)(r >> 6)] >> (r & 63)) & 1ull;
    seen[(size_t)(r >> 6)] |= 1ull << (r & 63);
  }
  if (!dup) return v;
  v.owned.reset(image_gather(bed, ind_row, n, ind_col, m), bed_free);
  v.bed = v.owned.get();
  v.ind_row = v.ind_col = nullptr;
  return v;
}

struct bsn_cor {
  std::shared_ptr<bsn_bed> owned;   // declared first: the buffers below are released before the image's stream goes
  BandJob job;
  // the compacted columns (@i, @x), one piece per block of band columns (one piece when the whole band fits)
  struct Piece {
    DevBuf<int32_t> d_i;
    DevBuf<double> d_x;
    int64_t nnz = 0;
  };
  std::vector<std::unique_ptr<Piece>> pieces;
  int64_t nnz = 0;
  int has_nan = 0;   // an NaN among the stored correlations (a variant without variation, R/corr.R:53-54)
  // (round 5) the result of an out-of-core handle: assembled on the host from the runs of target variants
  bool on_host = false;
  std::vector<int32_t> h_i;
  std::vector<double> h_x;
};

static void cormat_resident(bsn_bed *bed_in, const int64_t *ind_row_in, int64_t n, const int64_t *ind_col_in, int64_t m,
                            double size, const double *thr, const double *pos, int fill_diag, int32_t *p_out,
                            int64_t *nnz_out, bsn_cor **out);

// (round 5) snp_cor / bed_cor on an OUT-OF-CORE handle: corMat pairs every variant with the EARLIER variants of its
// window (src/corr.cpp:52-53), so a run of target variants needs a halo to the left only.  Each run is uploaded with
// that halo into the resident slab image and goes through the resident code path; the columns of its targets — row
// indices shifted back to the caller's numbering — are appended to a result kept on the host.  Same pairs, same
// epilogue: @p, @i, @x identical to the resident handle's.
static void cormat_streamed(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m,
                            double size, const double *thr, const double *pos, int fill_diag, int32_t *p_out,
                            int64_t *nnz_out, bsn_cor **out) {
  if (m <= 0 || n <= 0) fail("'ind.row' and 'ind.col' can't be empty.");
  auto col = [&](int64_t t) -> int64_t { return ind_col ? ind_col[t] : t; };
  for (int64_t t = 0; t < m; t++) {
    if (col(t) < 0 || col(t) >= bed->m) fail("Tested %lld < %lld. Subscript out of bounds (ind.col).", (long long)col(t), (long long)bed->m);
    if (t > 0 && col(t) <= col(t - 1))
      fail("snp_cor on an out-of-core handle (it streams its file) needs 'ind.col' in increasing file order");
    if (t > 0 && pos[t] < pos[t - 1]) fail("'pos' is not sorted.");
  }
  const int64_t cap = bed->slab_cols;
  bsn_bed *img = slab_image(bed);
  std::unique_ptr<bsn_cor> C(new bsn_cor());
  C->on_host = true;
  std::vector<int64_t> loc;
  std::vector<int32_t> p_loc, i_loc;
  std::vector<double> x_loc;
  p_out[0] = 0;
  int64_t a = 0, nnz = 0;
  while (a < m) {
    int64_t lo = a;
    while (lo > 0 && pos[lo - 1] >= pos[a] - size) lo--;
    int64_t b = a;
    while (b < m && col(b) - col(lo) + 1 <= cap) b++;   // the longest run of targets behind that halo that fits
    if (b == a)
      fail("LD window before variant %lld spans %lld variants of the file: more than the %lld of the out-of-core slab "
           "image; raise BSN_IMAGE_BUDGET or use a smaller window",
           (long long)col(a), (long long)(col(a) - col(lo) + 1), (long long)cap);
    const int64_t base = col(lo), ml = b - lo;
    slab_upload_range(bed, base, col(b - 1) - base + 1);
    loc.resize((size_t)ml);
    for (int64_t t = 0; t < ml; t++) loc[(size_t)t] = col(lo + t) - base;
    p_loc.resize((size_t)ml + 1);
    int64_t nz_loc = 0;
    bsn_cor *piece = nullptr;
    cormat_resident(img, ind_row, n, loc.data(), ml, size, thr, pos + lo, fill_diag, p_loc.data(), &nz_loc, &piece);
    std::unique_ptr<bsn_cor> hold(piece);
    i_loc.resize((size_t)std::max<int64_t>(nz_loc, 1));
    x_loc.resize((size_t)std::max<int64_t>(nz_loc, 1));
    {
      int64_t at = 0;
      for (const auto &P : piece->pieces) {
        if (P->nnz > 0) {
          copy_d2h(img, i_loc.data() + at, P->d_i.p, (size_t)P->nnz * 4);
          copy_d2h(img, x_loc.data() + at, P->d_x.p, (size_t)P->nnz * 8);
        }
        at += P->nnz;
      }
      BSN_HIP(hipStreamSynchronize(img->stream));
    }
    for (int64_t t = a; t < b; t++) {           // the targets' columns, rows in the caller's numbering
      const int64_t c0 = p_loc[(size_t)(t - lo)], c1 = p_loc[(size_t)(t - lo + 1)];
      for (int64_t e = c0; e < c1; e++) {
        C->h_i.push_back((int32_t)(i_loc[(size_t)e] + lo));
        C->h_x.push_back(x_loc[(size_t)e]);
        if (x_loc[(size_t)e] != x_loc[(size_t)e]) C->has_nan = 1;
      }
      nnz += c1 - c0;
      if (nnz > 0x7fffffffLL) fail("more than 2^31 - 1 non-zero correlations");
      p_out[t + 1] = (int32_t)nnz;
    }
    a = b;
  }
  C->nnz = nnz;
  *nnz_out = nnz;
  *out = C.release();
}

extern "C" {

int bsn_cormat(bsn_bed *bed_in, const int64_t *ind_row_in, int64_t n, const int64_t *ind_col_in, int64_t m,
               double size, const double *thr, const double *pos, int fill_diag, int32_t *p_out,
               int64_t *nnz_out, bsn_cor **out) {
  return guarded([&] {
    if (bed_in->streamed()) {
      BSN_HIP(hipSetDevice(bed_in->device));
      cormat_streamed(bed_in, ind_row_in, n, ind_col_in, m, size, thr, pos, fill_diag, p_out, nnz_out, out);
      return;
    }
    cormat_resident(bed_in, ind_row_in, n, ind_col_in, m, size, thr, pos, fill_diag, p_out, nnz_out, out);
  });
}

}  // extern "C"

static void cormat_resident(bsn_bed *bed_in, const int64_t *ind_row_in, int64_t n, const int64_t *ind_col_in, int64_t m,
                            double size, const double *thr, const double *pos, int fill_diag, int32_t *p_out,
                            int64_t *nnz_out, bsn_cor **out) {
  {
    std::unique_ptr<bsn_cor> C(new bsn_cor());
    BandJob &J = C->job;
    const RowView rv = row_view(bed_in, ind_row_in, n, ind_col_in, m);
    bsn_bed *bed = rv.bed;
    const int64_t *ind_row = rv.ind_row, *ind_col = rv.ind_col;
    C->owned = rv.owned;
    band_stats(J, bed, ind_row, n, ind_col, m, pos, size, false, true, 12.0);   // @i + @x: 12 bytes per kept pair
    copy_h2d(bed, J.d_thr.ensure((size_t)n), thr, (size_t)n * 8);
    DevBuf<int32_t> d_cnt;
    DevBuf<int64_t> d_off;
    DevBuf<int> d_nan;
    BSN_HIP(hipMemsetAsync(d_nan.ensure(1), 0, sizeof(int), bed->stream));
    std::vector<int32_t> cnt;
    std::vector<int64_t> off;
    int64_t nnz = 0;
    p_out[0] = 0;
    // blocks of band columns in ascending order (one block when the band fits): statistics -> r -> count -> compact
    for (int64_t c0 = 0; c0 < m; c0 += J.chunk_cols) {
      const int64_t c1 = std::min(m, c0 + J.chunk_cols), mc = c1 - c0;
      band_run(J, 0, J.d_thr.p, nullptr, nullptr, (double)n, c0, c1, c0 > 0);
      hipLaunchKernelGGL(k_cor_count, dim3((unsigned)((mc + 3) / 4)), dim3(256), 0, bed->stream, J.band_at(c0),
                         J.d_lo.p, J.W, c0, c1, fill_diag, d_cnt.ensure((size_t)mc));
      BSN_HIP(hipGetLastError());
      cnt.resize((size_t)mc);
      copy_d2h(bed, cnt.data(), d_cnt.p, (size_t)mc * 4);
      BSN_HIP(hipStreamSynchronize(bed->stream));
      off.resize((size_t)mc);
      int64_t nz = 0;
      for (int64_t j = 0; j < mc; j++) {
        off[(size_t)j] = nz;
        nz += cnt[(size_t)j];
        if (nnz + nz > 0x7fffffffLL) fail("more than 2^31 - 1 non-zero correlations");
        p_out[c0 + j + 1] = (int32_t)(nnz + nz);
      }
      std::unique_ptr<bsn_cor::Piece> P(new bsn_cor::Piece());
      P->nnz = nz;
      P->d_i.ensure((size_t)std::max<int64_t>(nz, 1));
      P->d_x.ensure((size_t)std::max<int64_t>(nz, 1));
      copy_h2d(bed, d_off.ensure((size_t)mc), off.data(), (size_t)mc * 8);
      hipLaunchKernelGGL(k_cor_fill, dim3((unsigned)((mc + 3) / 4)), dim3(256), 0, bed->stream, J.band_at(c0),
                         J.d_lo.p, J.W, c0, c1, fill_diag, d_off.p, P->d_i.p, P->d_x.p, d_nan.p);
      BSN_HIP(hipGetLastError());
      BSN_HIP(hipStreamSynchronize(bed->stream));   // `off` is reused by the next block
      C->pieces.push_back(std::move(P));
      nnz += nz;
    }
    C->nnz = nnz;
    // (through the handle's pinned staging buffer: a copy into pageable memory leaves every later stream
    // synchronisation of the process ~4 ms late, bsn_internal.hpp)
    copy_d2h(bed, &C->has_nan, d_nan.p, sizeof(int));
    BSN_HIP(hipStreamSynchronize(bed->stream));
    J.d_band.release();
    J.d_stats.release();
    *nnz_out = nnz;
    *out = C.release();
  }
}

extern "C" {

int bsn_ld_last_stats(double *out) {
  return guarded([&] {
    out[0] = g_ld_stats.pairs;
    out[1] = g_ld_stats.tile_pairs;
    out[2] = g_ld_stats.stats_ms;
    out[3] = g_ld_stats.launches;
    out[4] = (double)g_ld_stats.kernel;
  });
}

int bsn_cormat_fetch(bsn_cor *c, int32_t *i_out, double *x_out) {
  return guarded([&] {
    if (c->on_host) {
      if (c->nnz > 0) {
        std::memcpy(i_out, c->h_i.data(), (size_t)c->nnz * 4);
        std::memcpy(x_out, c->h_x.data(), (size_t)c->nnz * 8);
      }
      return;
    }
    BSN_HIP(hipSetDevice(c->job.bed->device));
    int64_t at = 0;
    for (const auto &P : c->pieces) {
      if (P->nnz > 0) {
        copy_d2h(c->job.bed, i_out + at, P->d_i.p, (size_t)P->nnz * 4);
        copy_d2h(c->job.bed, x_out + at, P->d_x.p, (size_t)P->nnz * 8);
      }
      at += P->nnz;
    }
  });
}

int bsn_cormat_has_nan(const bsn_cor *c, int *out) {
  return guarded([&] { *out = c->has_nan; });
}

int bsn_cormat_free(bsn_cor *c) {
  return guarded([&] { delete c; });
}

static void ld_scores_resident(bsn_bed *bed_in, const int64_t *ind_row_in, int64_t n, const int64_t *ind_col_in, int64_t m,
                               double size, const double *pos, double *out);

// (round 5) LD scores on an OUT-OF-CORE handle.  The score of a variant only needs the variants inside its window,
// so the selection is cut into runs of target variants; each run is uploaded together with its halo (the window of
// its first variant to the left, of its last to the right) into the resident slab image, scored there by the resident
// code path — the same pairs in the same order: identical values — and keeps its targets' scores.
static void ld_scores_streamed(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m,
                               double size, const double *pos, double *out) {
  if (m <= 0 || n <= 0) fail("'ind.row' and 'ind.col' can't be empty.");
  auto col = [&](int64_t t) -> int64_t { return ind_col ? ind_col[t] : t; };
  for (int64_t t = 0; t < m; t++) {
    if (col(t) < 0 || col(t) >= bed->m) fail("Tested %lld < %lld. Subscript out of bounds (ind.col).", (long long)col(t), (long long)bed->m);
    if (t > 0 && col(t) <= col(t - 1))
      fail("LD scores on an out-of-core handle (it streams its file) need 'ind.col' in increasing file order");
    if (t > 0 && pos[t] < pos[t - 1]) fail("'pos' is not sorted.");
  }
  const int64_t cap = bed->slab_cols;
  bsn_bed *img = slab_image(bed);
  std::vector<int64_t> loc;
  std::vector<double> part;
  int64_t a = 0;
  while (a < m) {
    int64_t lo = a;
    while (lo > 0 && pos[lo - 1] >= pos[a] - size) lo--;
    // the longest run of targets [a, b) whose halo still fits the slab image
    int64_t b = a, hi = a;
    for (;;) {
      int64_t h = std::max(hi, b);
      while (h + 1 < m && pos[h + 1] <= pos[b] + size) h++;
      if (col(h) - col(lo) + 1 > cap) break;
      hi = h;
      b++;
      if (b >= m) break;
    }
    if (b == a)
      fail("LD window around variant %lld spans %lld variants of the file: more than the %lld of the out-of-core slab "
           "image; raise BSN_IMAGE_BUDGET or use a smaller window",
           (long long)col(a), (long long)(col(std::max(hi, a)) - col(lo) + 1), (long long)cap);
    const int64_t base = col(lo), cnt = col(hi) - base + 1, ml = hi - lo + 1;
    slab_upload_range(bed, base, cnt);
    loc.resize((size_t)ml);
    for (int64_t t = 0; t < ml; t++) loc[(size_t)t] = col(lo + t) - base;
    part.resize((size_t)ml);
    ld_scores_resident(img, ind_row, n, loc.data(), ml, size, pos + lo, part.data());
    for (int64_t t = a; t < b; t++) out[t] = part[(size_t)(t - lo)];
    a = b;
  }
}

int bsn_ld_scores(bsn_bed *bed_in, const int64_t *ind_row_in, int64_t n, const int64_t *ind_col_in, int64_t m,
                  double size, const double *pos, double *out) {
  return guarded([&] {
    if (bed_in->streamed()) {
      BSN_HIP(hipSetDevice(bed_in->device));
      ld_scores_streamed(bed_in, ind_row_in, n, ind_col_in, m, size, pos, out);
      return;
    }
    ld_scores_resident(bed_in, ind_row_in, n, ind_col_in, m, size, pos, out);
  });
}

static void ld_scores_resident(bsn_bed *bed_in, const int64_t *ind_row_in, int64_t n, const int64_t *ind_col_in, int64_t m,
                               double size, const double *pos, double *out) {
  {
    const RowView rv = row_view(bed_in, ind_row_in, n, ind_col_in, m);
    bsn_bed *bed = rv.bed;
    const int64_t *ind_row = rv.ind_row, *ind_col = rv.ind_col;
    BandJob J;
    band_stats(J, bed, ind_row, n, ind_col, m, pos, size, false, true);
    DevBuf<double> d_ld;
    d_ld.ensure((size_t)m);
    // blocks of band columns in ascending order (one block when the band fits): r2, then every variant the block's
    // columns reach takes its terms — the running sums go through the same additions as over the whole band
    for (int64_t c0 = 0; c0 < m; c0 += J.chunk_cols) {
      const int64_t c1 = std::min(m, c0 + J.chunk_cols), jlo = J.lo[(size_t)c0];
      band_run(J, 1, nullptr, nullptr, nullptr, (double)n, c0, c1, c0 > 0);
      hipLaunchKernelGGL(k_ld_sum, dim3((unsigned)((c1 - jlo + 255) / 256)), dim3(256), 0, bed->stream, J.band_at(c0),
                         J.d_lo.p, J.W, m, jlo, c0, c1, d_ld.p);
      BSN_HIP(hipGetLastError());
    }
    copy_d2h(bed, out, d_ld.p, (size_t)m * 8);
    BSN_HIP(hipStreamSynchronize(bed->stream));
  }
}

// ---- greedy clumping inside one chromosome --------------------------------------------------
// mode 0: FBM formula (aux1 = sumX, aux2 = denoX, src/clumping.cpp:66-73); mode 1: bed formula (aux1 = center,
// aux2 = scale, src/clumping-bed.cpp:62-76).
//
// The rank-ordered sweep of src/clumping.cpp:33-88 (sequential order == the reference's result for any
// ncores, tests/testthat/test-7-OpenMP.R:104-115).  When j0 is visited, the neighbours that which_to_check
// (src/clumping-utils.h:12-43) would return and that are still kept are exactly the KEPT variants visited
// before it, so "is there one with r2 > thr" is
//   (bitsL[j0] AND kept[j0-1 .. j0-nL]) OR (bitsU[j0] AND kept[j0+1 .. j0+nU])  !=  0,
// evaluated 64 neighbours per word on two bit sets of the kept variants (ascending, and reversed for the
// earlier neighbours).  bitsL / bitsU are the thresholded r2 band (k_band_gt / k_band_gt_upper).
typedef unsigned long long u64;

static bool any_and(const u64 *A, const u64 *K, int64_t kbase, int64_t len) {
  for (int64_t q = 0; q * 64 < len; q++) {
    u64 a = A[q];
    const int64_t rem = len - q * 64;
    if (rem < 64) a &= (1ull << rem) - 1ull;
    if (!a) continue;
    const int64_t bpos = kbase + q * 64;
    const int sh = (int)(bpos & 63);
    u64 kwd = K[bpos >> 6] >> sh;
    if (sh) kwd |= K[(bpos >> 6) + 1] << (64 - sh);
    if (a & kwd) return true;
  }
  return false;
}

// number of earlier / later neighbours of every variant inside the window (positions are sorted)
static void window_counts(const double *pos, int64_t m, double size, std::vector<int64_t> &nL,
                          std::vector<int64_t> &nU) {
  nL.resize((size_t)m);
  nU.resize((size_t)m);
  int64_t l = 0, u = 0;
  for (int64_t j0 = 0; j0 < m; j0++) {
    const double pos_min = pos[j0] - size, pos_max = pos[j0] + size;
    while (l < j0 && !(pos[l] >= pos_min)) l++;
    if (u < j0) u = j0;
    while (u + 1 < m && pos[u + 1] <= pos_max) u++;
    nL[(size_t)j0] = j0 - l;
    nU[(size_t)j0] = u - j0;
  }
}

// the kept variants as two bit sets (ascending and reversed), as the sweep reads them
struct KeptBits {
  int64_t m;
  std::vector<u64> fwd, rev;
  explicit KeptBits(int64_t m_) : m(m_), fwd((size_t)(m_ + 63) / 64 + 2, 0ull), rev((size_t)(m_ + 63) / 64 + 2, 0ull) {}
  void add(int64_t j) {
    fwd[(size_t)(j >> 6)] |= 1ull << (j & 63);
    const int64_t r = m - 1 - j;
    rev[(size_t)(r >> 6)] |= 1ull << (r & 63);
  }
  bool hit(const u64 *BL, const u64 *BU, int64_t Wq, int64_t j0, int64_t nl, int64_t nu) const {
    return any_and(BL + (size_t)j0 * (size_t)Wq, rev.data(), m - j0, nl) ||
           any_and(BU + (size_t)j0 * (size_t)Wq, fwd.data(), j0 + 1, nu);
  }
};

// r2 band of the listed variants at window `size`, thresholded at every uthr[t] into the two host bit
// images of the sweep; returns the words per variant
static int64_t clump_bits(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m,
                          int mode, const double *aux1, const double *aux2, const double *pos, double size,
                          const std::vector<double> &uthr, std::vector<std::vector<u64>> &bitsL,
                          std::vector<std::vector<u64>> &bitsU) {
  BandJob J;
  band_stats(J, bed, ind_row, n, ind_col, m, pos, size, true);
  copy_h2d(bed, J.d_v1.ensure((size_t)m), aux1, (size_t)m * 8);
  copy_h2d(bed, J.d_v2.ensure((size_t)m), aux2, (size_t)m * 8);
  band_run(J, mode == 0 ? 2 : 3, nullptr, J.d_v1.p, J.d_v2.p, (double)n);
  const int64_t Wq = (J.W + 63) / 64;
  bitsL.assign(uthr.size(), {});
  bitsU.assign(uthr.size(), {});
  DevBuf<u64> d_bits;
  d_bits.ensure((size_t)m * (size_t)Wq);
  for (size_t t = 0; t < uthr.size(); t++) {
    for (int side = 0; side < 2; side++) {
      if (side == 0)
        hipLaunchKernelGGL(k_band_gt, dim3((unsigned)m), dim3(256), 0, bed->stream, J.d_band.p, J.d_lo.p, J.W,
                           Wq, uthr[t], d_bits.p);
      else
        hipLaunchKernelGGL(k_band_gt_upper, dim3((unsigned)m), dim3(256), 0, bed->stream, J.d_band.p, J.d_lo.p,
                           J.W, Wq, m, uthr[t], d_bits.p);
      BSN_HIP(hipGetLastError());
      std::vector<u64> &dst = side == 0 ? bitsL[t] : bitsU[t];
      dst.resize((size_t)m * (size_t)Wq);
      copy_d2h(bed, dst.data(), d_bits.p, dst.size() * 8);
    }
  }
  BSN_HIP(hipStreamSynchronize(bed->stream));
  return Wq;
}

// (round 6) the same bit images for an OUT-OF-CORE handle (the reference maps a file of any size and clumps on it:
// src/clumping-bed.cpp:11-91, src/bed-acc.h:46).  A bit of bitsL / bitsU is RELATIVE to its variant (bit i of bitsL[j0] =
// neighbour j0 - 1 - i, of bitsU[j0] = neighbour j0 + 1 + i), so the images can be made run by run: a run of target
// variants is uploaded with the window halo on both sides into the resident slab image (as bed_ld_scores does), its band
// is computed and thresholded there by the resident code, and the rows of the targets are kept.  The rank-ordered sweep
// then runs on the host over the whole chromosome exactly as for a resident image: identical indices.
// cols == nullptr / contiguous: runs of the file; an arbitrary list (the batches of the wide-window path): the listed
// variants are gathered from the mapped file into the slab image, which must hold them.
static int64_t clump_bits_streamed(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m,
                                   int mode, const double *aux1, const double *aux2, const double *pos, double size,
                                   const std::vector<double> &uthr, std::vector<std::vector<u64>> &bitsL,
                                   std::vector<std::vector<u64>> &bitsU) {
  auto col = [&](int64_t t) -> int64_t { return ind_col ? ind_col[t] : t; };
  bool increasing = true;
  for (int64_t t = 0; t < m; t++) {
    if (col(t) < 0 || col(t) >= bed->m) fail("Tested %lld < %lld. Subscript out of bounds (ind.col).", (long long)col(t), (long long)bed->m);
    if (t > 0 && col(t) <= col(t - 1)) increasing = false;
    if (t > 0 && pos[t] < pos[t - 1]) fail("'pos' is not sorted.");
  }
  std::vector<int64_t> nL, nU;
  window_counts(pos, m, size, nL, nU);
  int64_t Wg = 1;
  for (int64_t j = 0; j < m; j++) Wg = std::max(Wg, std::max(nL[(size_t)j], nU[(size_t)j]));
  const int64_t Wq = (Wg + 63) / 64;
  bitsL.assign(uthr.size(), std::vector<u64>((size_t)m * (size_t)Wq, 0ull));
  bitsU.assign(uthr.size(), std::vector<u64>((size_t)m * (size_t)Wq, 0ull));
  const int64_t cap = bed->slab_cols;
  bsn_bed *img = slab_image(bed);
  std::vector<int64_t> loc;
  std::vector<std::vector<u64>> bl, bu;
  std::vector<uint8_t> gathered;
  int64_t a = 0;
  while (a < m) {
    int64_t lo = a;
    while (lo > 0 && pos[lo - 1] >= pos[a] - size) lo--;
    // the longest run of targets [a, b) whose halo still fits the slab image (by count: a run whose file span fits too is
    // read as one contiguous piece, any other is gathered variant by variant from the mapped file)
    int64_t b = a, hi = a, h_fail = a;
    for (;;) {
      int64_t h = std::max(hi, b);
      while (h + 1 < m && pos[h + 1] <= pos[b] + size) h++;
      if (h - lo + 1 > cap) {
        h_fail = h;
        break;
      }
      hi = h;
      b++;
      if (b >= m) break;
    }
    if (b == a)
      fail("clumping window around variant %lld holds %lld variants: more than the %lld of the out-of-core slab image; raise "
           "BSN_IMAGE_BUDGET or use a smaller window", (long long)col(a), (long long)(h_fail - lo + 1), (long long)cap);
    const int64_t ml = hi - lo + 1;
    loc.resize((size_t)ml);
    if (increasing && col(hi) - col(lo) + 1 <= cap) {
      const int64_t base = col(lo), cnt = col(hi) - base + 1;
      slab_upload_range(bed, base, cnt);
      for (int64_t t = 0; t < ml; t++) loc[(size_t)t] = col(lo + t) - base;
    } else {   // gather from the mapped file (pageable: staged by the runtime), in list order
      gathered.resize((size_t)ml * (size_t)bed->n_byte);
      for (int64_t t = 0; t < ml; t++)
        std::memcpy(gathered.data() + (size_t)t * (size_t)bed->n_byte, bed->h_map + (size_t)col(lo + t) * (size_t)bed->n_byte, (size_t)bed->n_byte);
      img->m = ml;
      img->na_cnt.clear();
      img->counts_cache.clear();
      image_from_host(img, gathered.data(), bed->n_byte);
      for (int64_t t = 0; t < ml; t++) loc[(size_t)t] = t;
    }
    // (a row list with repeated samples: the run is gathered once more, rows included, like the resident entry does)
    const RowView rv = row_view(img, ind_row, n, loc.data(), ml);
    const int64_t wq = clump_bits(rv.bed, rv.ind_row, n, rv.ind_col, ml, mode, aux1 + lo, aux2 + lo, pos + lo, size, uthr, bl, bu);
    const int64_t wc = std::min(wq, Wq);
    for (size_t t = 0; t < uthr.size(); t++)
      for (int64_t j = a; j < b; j++) {
        std::memcpy(&bitsL[t][(size_t)j * (size_t)Wq], &bl[t][(size_t)(j - lo) * (size_t)wq], (size_t)wc * 8);
        std::memcpy(&bitsU[t][(size_t)j * (size_t)Wq], &bu[t][(size_t)(j - lo) * (size_t)wq], (size_t)wc * 8);
      }
    a = b;
  }
  return Wq;
}
static int64_t clump_bits_any(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m,
                              int mode, const double *aux1, const double *aux2, const double *pos, double size,
                              const std::vector<double> &uthr, std::vector<std::vector<u64>> &bitsL,
                              std::vector<std::vector<u64>> &bitsU) {
  return bed->streamed() ? clump_bits_streamed(bed, ind_row, n, ind_col, m, mode, aux1, aux2, pos, size, uthr, bitsL, bitsU)
                         : clump_bits(bed, ind_row, n, ind_col, m, mode, aux1, aux2, pos, size, uthr, bitsL, bitsU);
}

// widest window, in variants, of a two-sided window of `size`
static int64_t window_width(const double *pos, int64_t m, double size) {
  std::vector<int64_t> lo;
  window_bounds(pos, m, size, true, lo);
  int64_t W = 1;
  for (int64_t j0 = 0; j0 < m; j0++) W = std::max(W, j0 - lo[(size_t)j0]);
  return W;
}

// One (size, thr) point whose dense band (m x W fp64) does not fit: the pairs the reference evaluates are
// only (candidate, KEPT variant inside its window) — src/clumping.cpp:52-79 skips pruned neighbours — so the
// variants are taken in rank order in batches, and each batch gets the band of the SUB-LIST "kept so far +
// this batch" (sorted by position), which is narrow because the kept variants are sparse exactly when the
// window is wide (small thr).  The batch grows with the kept set, so at most about half of a sub-band is
// kept x kept pairs that were already known.  Same sweep, same bits, same result as the dense path.
static void clump_lazy(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m,
                       int mode, const double *aux1, const double *aux2, const int32_t *ordInd,
                       const double *pos, double size, double thr, int32_t *keep, double budget_bytes,
                       int64_t batch_min) {
  std::vector<char> kept((size_t)m, 0), in_batch((size_t)m, 0);
  std::vector<int64_t> L, sub_cols, idx_in_L((size_t)m, -1), nL, nU;
  std::vector<double> sub_pos, sub_a1, sub_a2;
  std::vector<std::vector<u64>> bl, bu;
  const std::vector<double> uthr(1, thr);
  int64_t n_kept = 0, k_next = 0;
  while (k_next < m) {
    int64_t B = std::max<int64_t>(batch_min, n_kept);
    for (;;) {
      B = std::min<int64_t>(B, m - k_next);
      for (int64_t k = k_next; k < k_next + B; k++) in_batch[(size_t)ordInd[k]] = 1;
      L.clear();
      for (int64_t j = 0; j < m; j++)
        if (kept[(size_t)j] || in_batch[(size_t)j]) L.push_back(j);
      sub_pos.resize(L.size());
      for (size_t i = 0; i < L.size(); i++) sub_pos[i] = pos[L[i]];
      const int64_t Wsub = window_width(sub_pos.data(), (int64_t)L.size(), size);
      // (an out-of-core handle: the kept variants and the batch, with their windows, also have to fit the slab image)
      const bool fits_slab = !bed->streamed() || (int64_t)L.size() <= bed->slab_cols;
      if ((double)L.size() * (double)Wsub * 8.0 <= budget_bytes && fits_slab) break;
      for (int64_t k = k_next; k < k_next + B; k++) in_batch[(size_t)ordInd[k]] = 0;
      if (B <= 64 && !fits_slab)
        fail("clumping: %lld kept variants and the next candidates are more than the %lld variants of the out-of-core slab image; "
             "raise BSN_IMAGE_BUDGET or use a smaller window", (long long)n_kept, (long long)bed->slab_cols);
      if (B <= 64)
        fail("clumping: %lld kept variants within windows of up to %lld of them do not fit the device "
             "memory budget (%.1f GB); use a smaller window", (long long)n_kept, (long long)Wsub,
             budget_bytes / 1e9);
      B /= 2;
    }
    const int64_t ml = (int64_t)L.size();
    sub_cols.resize((size_t)ml);
    sub_a1.resize((size_t)ml);
    sub_a2.resize((size_t)ml);
    for (int64_t i = 0; i < ml; i++) {
      const int64_t j = L[(size_t)i];
      sub_cols[(size_t)i] = ind_col ? ind_col[j] : j;
      sub_a1[(size_t)i] = aux1[j];
      sub_a2[(size_t)i] = aux2[j];
      idx_in_L[(size_t)j] = i;
    }
    const int64_t Wq = clump_bits_any(bed, ind_row, n, sub_cols.data(), ml, mode, sub_a1.data(), sub_a2.data(),
                                      sub_pos.data(), size, uthr, bl, bu);
    window_counts(sub_pos.data(), ml, size, nL, nU);
    KeptBits kb(ml);
    for (int64_t i = 0; i < ml; i++)
      if (kept[(size_t)L[(size_t)i]]) kb.add(i);
    for (int64_t k = k_next; k < k_next + B; k++) {
      const int64_t j0 = ordInd[k], li = idx_in_L[(size_t)j0];
      const bool pruned = kb.hit(bl[0].data(), bu[0].data(), Wq, li, nL[(size_t)li], nU[(size_t)li]);
      keep[j0] = pruned ? 0 : 1;
      in_batch[(size_t)j0] = 0;
      if (!pruned) {
        kb.add(li);
        kept[(size_t)j0] = 1;
        n_kept++;
      }
    }
    k_next += B;
  }
}

// A grid of (size, thr) points (clumping_chr_cached threads a sparse r2 cache through the grid loops of
// R/SCT.R:100-131): the points whose dense band fits the device share ONE band, computed at their largest
// window and thresholded on the device once per distinct thr; the others go through clump_lazy one by
// one.  keep[g * m + j] receives 0 / 1.
static void clumping_grid(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                          int64_t m, int mode, const double *aux1, const double *aux2,
                          const int32_t *ordInd, const int32_t *rankInd, const double *pos,
                          int64_t n_grid, const double *sizes, const double *thrs, int32_t *keep_out) {
  if (n_grid <= 0) return;
  for (int64_t k = 0; k < m; k++)
    if (ordInd[k] < 0 || ordInd[k] >= m || rankInd[k] < 0 || rankInd[k] >= m)
      fail("'ordInd' / 'rankInd' out of bounds");
  BSN_HIP(hipSetDevice(bed->device));
  // memory for one band: what the device has free, less room for statistics buffers and bit images
  // (BSN_CLUMP_BAND_BUDGET, bytes, overrides it: the tests force the lazy path with it)
  size_t free_b = 0, total_b = 0;
  BSN_HIP(hipMemGetInfo(&free_b, &total_b));
  const double avail = std::max(1e9, ((double)(free_b + dev_cache_held()) - 8e9) * 0.8);
  double budget = avail;
  if (const char *e = getenv("BSN_CLUMP_BAND_BUDGET")) budget = atof(e);
  int64_t batch_min = 8192;
  if (const char *e = getenv("BSN_CLUMP_LAZY_BATCH")) batch_min = std::max<int64_t>(1, atoll(e));
  std::vector<int64_t> dense;
  for (int64_t g = 0; g < n_grid; g++) {
    const int64_t Wg = window_width(pos, m, sizes[g]);
    // (an out-of-core handle makes its band run by run: at most one slab image's worth of variants at a time)
    const int64_t mrows = bed->streamed() ? std::min(m, bed->slab_cols) : m;
    // (... and a window that holds more variants than the slab image goes the way of the wide windows: kept variants + batch)
    const bool window_fits = !bed->streamed() || 2 * Wg + 1 <= bed->slab_cols;
    if ((double)mrows * (double)Wg * 8.0 <= budget && window_fits)
      dense.push_back(g);
    else
      clump_lazy(bed, ind_row, n, ind_col, m, mode, aux1, aux2, ordInd, pos, sizes[g], thrs[g], keep_out + g * m,
                 std::min(avail, 16e9), batch_min);
  }
  if (dense.empty()) return;
  double size_max = sizes[dense[0]];
  for (int64_t g : dense) size_max = std::max(size_max, sizes[g]);
  // distinct thresholds -> one pair of bit images each
  std::vector<double> uthr;
  std::vector<int> thr_id((size_t)n_grid, 0);
  for (int64_t g : dense) {
    size_t t = 0;
    while (t < uthr.size() && !(uthr[t] == thrs[g])) t++;
    if (t == uthr.size()) uthr.push_back(thrs[g]);
    thr_id[(size_t)g] = (int)t;
  }
  std::vector<std::vector<u64>> bitsL, bitsU;
  const int64_t Wq = clump_bits_any(bed, ind_row, n, ind_col, m, mode, aux1, aux2, pos, size_max, uthr, bitsL, bitsU);
  std::vector<int64_t> nL, nU;
  for (int64_t g : dense) {
    const u64 *BL = bitsL[(size_t)thr_id[(size_t)g]].data();
    const u64 *BU = bitsU[(size_t)thr_id[(size_t)g]].data();
    int32_t *keep = keep_out + g * m;
    window_counts(pos, m, sizes[g], nL, nU);  // window of this grid point in index units
    KeptBits kb(m);
    for (int64_t k = 0; k < m; k++) {
      const int64_t j0 = ordInd[k];
      const bool pruned = kb.hit(BL, BU, Wq, j0, nL[(size_t)j0], nU[(size_t)j0]);
      keep[j0] = pruned ? 0 : 1;
      if (!pruned) kb.add(j0);
    }
  }
}

int bsn_clumping_chr(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                     int64_t m, int mode, const double *aux1, const double *aux2,
                     const int32_t *ordInd, const int32_t *rankInd, const double *pos, double size,
                     double thr, int32_t *keep) {
  return guarded([&] {
    if (bed->streamed()) {   // (round 6) out of core: the band run by run on the slab image, the sweep on the host
      clumping_grid(bed, ind_row, n, ind_col, m, mode, aux1, aux2, ordInd, rankInd, pos, 1, &size, &thr, keep);
      return;
    }
    const RowView rv = row_view(bed, ind_row, n, ind_col, m);
    clumping_grid(rv.bed, rv.ind_row, n, rv.ind_col, m, mode, aux1, aux2, ordInd, rankInd, pos, 1, &size, &thr, keep);
  });
}

int bsn_clumping_chr_cached(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                            int64_t m, int mode, const double *aux1, const double *aux2,
                            const int32_t *ordInd, const int32_t *rankInd, const double *pos,
                            int64_t n_grid, const double *sizes, const double *thrs, int32_t *keep) {
  return guarded([&] {
    if (bed->streamed()) {
      clumping_grid(bed, ind_row, n, ind_col, m, mode, aux1, aux2, ordInd, rankInd, pos, n_grid, sizes, thrs, keep);
      return;
    }
    const RowView rv = row_view(bed, ind_row, n, ind_col, m);
    clumping_grid(rv.bed, rv.ind_row, n, rv.ind_col, m, mode, aux1, aux2, ordInd, rankInd, pos, n_grid, sizes, thrs,
                  keep);
  });
}

}  // extern "C"
