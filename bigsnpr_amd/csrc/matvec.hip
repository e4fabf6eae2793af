// matvec.hip — the streaming products of the scaled genotype matrix on gfx950.
//
//   op_prod : Y = A~ X   (replaces bed_pMatVec4,  src/bed-prod-vec.cpp:15-54)
//   op_cprod: Z = A~' X  (replaces bed_cpMatVec4, src/bed-prod-vec.cpp:59-97)
//   with A~[i,j] = (g - center_j)/scale_j, missing -> 0 (src/bed-acc.h:98-111).
//
// Design: both products are HBM-bound on the 2-bit image but a
// fp64 VALU formulation needs ~7 lane-ops per genotype, 2x what the chip can issue at
// 6 TB/s.  So the multiply-accumulate is moved to the i8 MFMA pipe, exactly:
//   - the fp64 panel is scaled per vector to a fixed-point integer of 8*S bits and split
//     into S balanced base-256 digits (int8 "slices");
//   - the image holds device codes (bsn_internal.hpp: 0, 1, 2 = allele count, 3 = missing), so
//     the masked 2-bit field IS the int8 operand of the genotype plane `code` in {0,1,2,3}
//     (7 shift/and per 16 genotypes, no look-up); one v_perm_b32 byte look-up per register
//     gives the plane na in {0,1};
//   - v_mfma_i32_16x16x64_i8 accumulates  P' = sum code*digit  and  Q = sum na*digit  in int32,
//     which is exact; the 3 of a missing value leaves through Q (g0 = code - 3 na) and the
//     digits are recombined in fp64 in a finalize kernel together with the centre/scale algebra
//         A~' x = (P - c (Sx - Q)) / s,      P = P' - 3 Q = sum g0 x, Q = sum na x, Sx = sum x
//         A~ x  = sum_j code w_j + sum_j na (c w - 3 w)_j - sum_j (c w)_j,   w = x / s.
//   Results are bit-reproducible (integer sums are order-independent).
#include <cmath>
#include <cstdlib>

#include "bsn_internal.hpp"
#include <type_traits>

namespace bsn {

typedef int v4i __attribute__((ext_vector_type(4)));

// Every launch of a streaming kernel notes WHICH instantiation it was: prof_end files it under the launch's kind and
// bsn_bed_streaming_kernels reports the names, so that a committed counter record (profiles/pmc_traffic.json) can be
// matched against the kernels the running build really launches (bench.py: roofline.traffic).
static thread_local const void *g_last_kernel = nullptr;
#define BSN_KLAUNCH(kern, ...)                 \
  do {                                         \
    g_last_kernel = (const void *)(kern);      \
    hipLaunchKernelGGL(kern, __VA_ARGS__);     \
  } while (0)

// byte look-up: result byte k = lut byte (sel byte k), sel bytes in 0..3.  The LUT is
// given as both sources so the result does not depend on the S0/S1 order of v_perm_b32.
__device__ __forceinline__ uint32_t lut4(uint32_t lut, uint32_t sel) {
  return __builtin_amdgcn_perm(lut, lut, sel);
}
// byte permute of the 8 bytes {hi:lo}; selector k picks lo.byte[k] for k<4, hi.byte[k-4]
// otherwise (verified by bsn_selftest)
__device__ __forceinline__ uint32_t perm8(uint32_t hi, uint32_t lo, uint32_t sel) {
  return __builtin_amdgcn_perm(hi, lo, sel);
}

// device code -> plane value (byte k of the constant = value of code k)
constexpr uint32_t kLutG0 = 0x00020100u;   // {0,1,2,0}
constexpr uint32_t kLutNA = 0x01000000u;   // {0,0,0,1}
constexpr uint32_t kLutHom2 = 0x00010000u; // {0,0,1,0}
constexpr uint32_t kLutHet = 0x00000100u;  // {0,1,0,0}
constexpr uint32_t kLutRaw = 0xFFFFFFFFu;  // marker: the plane is the code itself (no look-up)

// ---- per-vector metadata -------------------------------------------------------
struct VecMeta {
  unsigned long long absmax_bits;  // bits of max |value| (non-negative doubles order like u64)
  long long sum_hi, sum_lo;        // sum of the fixed-point integers, split at bit 24
  long long sum2_hi, sum2_lo;      // same for the second plane (c*w) in op_prod
  unsigned long long nonfinite;    // count of non-finite inputs
  double qscale;                   // fixed-point scale actually used
  double pad;
};
static_assert(sizeof(VecMeta) == 64, "VecMeta layout");

__global__ void k_meta_clear(VecMeta *meta, int nvec) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < nvec) {
    VecMeta z = {};
    meta[v] = z;
  }
}

// rows scatter: xfull[rows[i], v] += x[i, v]   (duplicates allowed -> atomics)
__global__ void k_scatter_rows(const double *X, int64_t ldx, const int32_t *rows, int64_t n,
                               double *xfull, int64_t ldf) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int v = blockIdx.y;
  if (i >= n) return;
  atomicAdd(&xfull[(int64_t)rows[i] + v * ldf], X[i + v * ldx]);
}

// mode 0 (cprod): value = X[k, v]                       for k < len
// mode 1 (prod) : value a = X[k, v]/scale[k], b = c*a; the digit planes hold a and b - 3a
// mode 2 (prod, raw plane weights): a = X[k, v], b = center[k, v] (a second panel)
// vstep / voff / raw3: the value of a genotype is voff + vstep k (2-bit image: k = allele count, 1 / 0);
// mode 1 quantises a = vstep X / scale and b = (c - voff) X / scale; raw3: the k plane holds 3 for a
// missing value (raw 2-bit codes), so the second digit plane carries b - 3 a
// Writes the maximum (and the number of non-finite inputs) of its slice to part[v][blockIdx.x][2]: k_quant
// combines the slices itself, so the quantisation of a panel is two launches (round 2: five — clear, maximum
// with atomics, scale, memset of the digits, digits).  Workgroup 0 of a vector also clears the sums that
// k_quant accumulates.
__global__ void k_absmax(const double *X, int64_t ldx, int64_t len, const double *center,
                         const double *scale, int mode, VecMeta *meta, double *part, double vstep, double voff,
                         int raw3) {
  int v = blockIdx.y;
  double mx = 0;
  unsigned long long bad = 0;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < len;
       k += (int64_t)gridDim.x * blockDim.x) {
    double a = X[k + v * ldx];
    if (mode == 1) {
      double s = scale ? scale[k] : 1.0, c = center ? center[k] : 0.0;
      a = a / s;
      const double b = (c - voff) * a;
      a *= vstep;
      const double b3 = fabs(b - 3.0 * a);  // the second digit plane (missing-value plane weights)
      if (!(fabs(b) <= 1.79e308)) bad++; else mx = fmax(mx, fabs(b));
      if (raw3 && b3 <= 1.79e308) mx = fmax(mx, b3);
    } else if (mode == 2) {  // raw second plane: center[] is W2 (same shape as X)
      double b = center ? fabs(center[k + v * ldx]) : 0.0;
      if (!(b <= 1.79e308)) bad++; else mx = fmax(mx, b);
    }
    a = fabs(a);
    if (!(a <= 1.79e308)) bad++; else mx = fmax(mx, a);
  }
  for (int off = 32; off > 0; off >>= 1) {
    mx = fmax(mx, __shfl_down(mx, off));
    bad += __shfl_down(bad, off);
  }
  // one atomic per workgroup: thousands of same-address 64-bit atomics cost ~20 ns each
  __shared__ double smx[16];
  __shared__ unsigned long long sbad[16];
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    smx[wave] = mx;
    sbad[wave] = bad;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < nw; w++) {
      mx = fmax(mx, smx[w]);
      bad += sbad[w];
    }
    double *pp = part + ((int64_t)v * gridDim.x + blockIdx.x) * 2;
    pp[0] = mx;
    pp[1] = (double)bad;
    if (blockIdx.x == 0) meta[v].sum_hi = meta[v].sum_lo = meta[v].sum2_hi = meta[v].sum2_lo = 0;
  }
}

// One wave quantises 1024 consecutive k of one vector; a thread ends up with 16 of them as S
// digit rows of 16 bytes.
// Layout: q[(k/16) * (nplanes*ncol) + plane*ncol + col][16 B], col = v*S + s.
// PERM = 1 (cprod operand): sample e of the 16 is stored at byte (e%4)*4 + e/4, the
// order in which k_cprod's decode emits them;  PERM = 0 (prod operand): natural.
// The values are read coalesced and handed over through LDS (row stride 17 against bank
// conflicts); the digit bytes are packed in registers (S and PERM are compile-time).
template <int S, int PERM>
__global__ __launch_bounds__(64) void k_quant(const double *__restrict__ X, int64_t ldx, int64_t len,
                                              int64_t len_pad, const double *__restrict__ center,
                                              const double *__restrict__ scale, int mode, int ncol,
                                              VecMeta *meta, int8_t *__restrict__ q, double vstep, double voff,
                                              int raw3, const double *__restrict__ part, int gx) {
  __shared__ double sa[64 * 17], sb[64 * 17];
  const int tid = threadIdx.x, v = blockIdx.y;
  const int64_t kb = (int64_t)blockIdx.x * 64 + tid;  // 16-block index of this thread
  const int nplanes = mode >= 1 ? 2 : 1;
  // fixed-point scale of the vector from the slice maxima of k_absmax (gx == 0: the values are
  // integers already, scale 1): the largest power of two with absmax * qs <= 0.99 * 2^(8S-1), so that
  // x * qs is exact and only the rounding to an integer errs
  double qs = 1.0;
  if (gx > 0) {
    double mx = 0, bad = 0;
    for (int t = tid; t < gx; t += 64) {
      mx = fmax(mx, part[((int64_t)v * gx + t) * 2]);
      bad += part[((int64_t)v * gx + t) * 2 + 1];
    }
    for (int off = 32; off > 0; off >>= 1) {
      mx = fmax(mx, __shfl_xor(mx, off));
      bad += __shfl_xor(bad, off);
    }
    qs = 0.0;
    if (mx > 0) {
      int e;
      frexp(ldexp(0.99, 8 * S - 1) / mx, &e);
      qs = ldexp(1.0, e - 1);
    }
    if (blockIdx.x == 0 && tid == 0) {
      meta[v].absmax_bits = (unsigned long long)__double_as_longlong(mx);
      meta[v].nonfinite = (unsigned long long)bad;
    }
  }
  if (blockIdx.x == 0 && tid == 0) meta[v].qscale = qs;
  // all 16 x (1..3) loads of a thread are issued before the first use
  double xa[16], xc[16], xs[16];
#pragma unroll
  for (int it = 0; it < 16; it++) {
    const int64_t k = (int64_t)blockIdx.x * 1024 + it * 64 + tid;
    const bool in = k < len;
    xa[it] = in ? X[k + v * ldx] : 0.0;
    xc[it] = 0.0;
    xs[it] = 1.0;
    if (mode == 1) {
      if (in && scale) xs[it] = scale[k];
      if (in && center) xc[it] = center[k];
    } else if (mode == 2) {
      if (in && center) xc[it] = center[k + v * ldx];
    }
  }
#pragma unroll
  for (int it = 0; it < 16; it++) {
    const int t = it * 64 + tid;
    double a = xa[it], b = 0;
    if (mode == 1) {
      a = a / xs[it];
      b = (xc[it] - voff) * a;
      a *= vstep;
    } else if (mode == 2) {
      b = xc[it];
    }
    if (!(fabs(a) <= 1.79e308)) a = 0;
    if (!(fabs(b) <= 1.79e308)) b = 0;
    sa[t + (t >> 4)] = a * qs;
    sb[t + (t >> 4)] = b * qs;
  }
  __syncthreads();
  long long shi = 0, slo = 0, shi2 = 0, slo2 = 0;
  if (kb * 16 < len_pad) {
    uint32_t pk[2][S][4];
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
      for (int s = 0; s < S; s++)
#pragma unroll
        for (int w = 0; w < 4; w++) pk[p][s][w] = 0;
#pragma unroll
    for (int e = 0; e < 16; e++) {
      long long A = llrint(sa[tid * 17 + e]), B = llrint(sb[tid * 17 + e]);
      shi += A >> 24; slo += A & 0xFFFFFF;
      shi2 += B >> 24; slo2 += B & 0xFFFFFF;
      // scaled product: the genotype plane is the raw code (3 for a missing value), so the
      // missing-value plane carries c w - 3 w, in exact integers: 3 A + (B - 3 A) - B == 0
      if (raw3 & 1) B -= 3 * A;
      const int pos = PERM ? ((e & 3) * 4 + (e >> 2)) : e;
#ifdef BSN_ABLATION
      // BSN_DIGITS=2 (timing only: the finalize kernels read balanced base-256 digits): sign x 7-bit magnitude chunks
      const bool sm = (raw3 >> 8) == 2;
      const long long sgA = A < 0 ? -1 : 1, sgB = B < 0 ? -1 : 1;
      long long mA = A < 0 ? -A : A, mB = B < 0 ? -B : B;
#endif
#pragma unroll
      for (int s = 0; s < S; s++) {
#ifdef BSN_ABLATION
        const int8_t da = sm ? (int8_t)(sgA * (mA & 0x7F)) : (int8_t)(A & 0xFF), db = sm ? (int8_t)(sgB * (mB & 0x7F)) : (int8_t)(B & 0xFF);
        mA >>= 7;
        mB >>= 7;
#else
        const int8_t da = (int8_t)(A & 0xFF), db = (int8_t)(B & 0xFF);
#endif
        A = (A - da) >> 8;
        B = (B - db) >> 8;
        pk[0][s][pos >> 2] |= (uint32_t)(uint8_t)da << (8 * (pos & 3));
        pk[1][s][pos >> 2] |= (uint32_t)(uint8_t)db << (8 * (pos & 3));
      }
    }
    for (int p = 0; p < nplanes; p++)
#pragma unroll
      for (int s = 0; s < S; s++) {
#ifdef BSN_ABLATION
        // BSN_DIGITS=1 (timing only): slice-major columns — digit s of all vectors side by side, so that with 16 vectors
        // of three digits every column block holds ONE digit position
        const int col = (raw3 >> 8) == 1 ? s * (int)gridDim.y + v : v * S + s;
#else
        const int col = v * S + s;
#endif
        int8_t *dst = q + ((kb * nplanes + p) * ncol + col) * 16;
        *(uint4 *)dst = p == 0 ? uint4{pk[0][s][0], pk[0][s][1], pk[0][s][2], pk[0][s][3]}
                               : uint4{pk[1][s][0], pk[1][s][1], pk[1][s][2], pk[1][s][3]};
      }
  }
  for (int off = 32; off > 0; off >>= 1) {
    shi += __shfl_down(shi, off); slo += __shfl_down(slo, off);
    shi2 += __shfl_down(shi2, off); slo2 += __shfl_down(slo2, off);
  }
  if (tid == 0) {
    atomicAdd((unsigned long long *)&meta[v].sum_hi, (unsigned long long)shi);
    atomicAdd((unsigned long long *)&meta[v].sum_lo, (unsigned long long)slo);
    if (mode >= 1) {
      atomicAdd((unsigned long long *)&meta[v].sum2_hi, (unsigned long long)shi2);
      atomicAdd((unsigned long long *)&meta[v].sum2_lo, (unsigned long long)slo2);
    }
  }
}

// ---------------------------------------------------------------------------
// k_cprod: contraction over samples.  One wave owns 32 variants (two 16-row MFMA tiles)
// for the whole sample range; the 8 waves of a workgroup share the digit panel of the
// current sample chunk through LDS.  Per 16 B of a variant row a lane does 4 K-steps of
// {decode 15 VALU, NPLANE*NB MFMA, NB ds_read_b128}.
//   A operand: lane l -> variant row (l&15), k-group (l>>4): 16 samples of that variant
//   B operand: lane l -> digit column (l&15), same 16 samples (from LDS)
//   D        : lane l -> column (l&15), rows 4*(l>>4)+r
// RAW0: plane 0 is the device code itself (no look-up).  STATS: the per-variant counts of the
// codes 1, 2 and missing over all samples ride along (popcounts on the raw dwords; the first
// crossproduct pass of a solve, which thereby replaces the separate statistics pass).
// ABL != 0 only exists in -DBSN_ABLATION builds (profiling variants that compute wrong numbers).
// CONTIG: the variants are col0 .. col0+m-1; the genotype loads are then buffer loads with a
// scalar descriptor based at the workgroup's first row, one 32-bit lane offset per tile and a scalar
// chunk offset, so that addressing costs no VALU (global loads spend a 64-bit add on each).
// TAG only changes the kernel's name: the launches of the warm start (a fraction of the variants) run as
// <..., TAG = 1> so that a kernel trace does not average them into the full passes.
// TILED (with CONTIG, col0 a multiple of 64): `img` is the streaming-layout copy (bsn_internal.hpp): variant
// a, byte o of its row at ((a >> 6) * (pitch >> 8) + (o >> 8)) * 16384 + (a & 63) * 256 + (o & 255).
// NASKIP (round 5; RAW0 with two planes): the missing-value plane of a K-step — half its MFMAs and four of its
// decode instructions — is only issued when one of the 64 samples x 16 variants of the step carries a missing code:
// a wave-uniform branch on the ballot of (w & w >> 1 & 0x5555...).  The sums are the same integers (the plane of a
// step without a missing code is zero).  Chosen by the host from the measured share of such steps (op_na_blocks):
// at 1 % scattered missing values no step is free and the branch only costs; on nearly complete or batch-structured
// data most are.
template <int NB, int NPLANE, int KC, bool RAW0, bool STATS, bool CONTIG, int ABL = 0, int TILES = 2,
          int WAVES = 8, int MINW = 1, int TAG = 0, bool TILED = false, int SGB = 0, bool NASKIP = false>
__global__ __launch_bounds__(64 * WAVES, MINW) void k_cprod(const uint8_t *__restrict__ img, int64_t pitch,
                                               const int32_t *__restrict__ cols, int64_t col0,
                                               int64_t m, const int8_t *__restrict__ xq,
                                               int32_t *__restrict__ acc_out, int64_t m_out,
                                               uint32_t lutA, uint32_t lutB, uint32_t lutC,
                                               int32_t *__restrict__ counts, int32_t n_pad_samples) {
  constexpr int NCOL = 16 * NB;
  constexpr int LD = KC / 256;             // 16-B loads per variant row per chunk per lane
  constexpr int XS = KC / 16 * NCOL;       // uint4 entries per LDS buffer
  __shared__ uint4 xs[2][XS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c = lane & 15, g = lane >> 4;
  constexpr int NT = 64 * WAVES;
  const int64_t wg_base = (int64_t)blockIdx.x * (WAVES * 16 * TILES);
  const int64_t snp_base = wg_base + wave * (16 * TILES);
  const uint8_t *rowp[TILES];
  uint32_t voff[TILES];
#pragma unroll
  for (int t = 0; t < TILES; t++) {
    int64_t j = snp_base + t * 16 + c;
    if (j > m - 1) j = m - 1;
    int64_t col = CONTIG ? col0 + j : (int64_t)cols[j];
    rowp[t] = img + col * pitch + g * 16;
    voff[t] = TILED ? (uint32_t)((((j - wg_base) >> 6) * (pitch >> 8)) * 16384 + (j & 63) * 256 + g * 16)
                    : (uint32_t)((j - wg_base) * pitch + g * 16);
  }
  static_assert(!TILED || (CONTIG && (WAVES * 16 * TILES) % 64 == 0), "streaming layout: whole 64-variant blocks");
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(TILED ? img + (((col0 + wg_base) >> 6) * (pitch >> 8)) * 16384
                     : img + (col0 + (CONTIG ? wg_base : 0)) * pitch),
      0, 0x7fffffff, 0x00020000);
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  // 16 B of tile t at byte `off` (uniform) of the variant row
  auto gload = [&](const int t, const int off) -> uint4 {
    if constexpr (CONTIG) {
      // ablation 64 / 128: nt / sc1 cache policy on the genotype stream (correct results)
      constexpr int AUX = ((ABL & 64) ? 2 : 0) | ((ABL & 128) ? 16 : 0);
      const int soff = TILED ? ((off >> 8) << 14) + (off & 255) : off;
      const v4u r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff[t], soff, AUX);
      return uint4{r.x, r.y, r.z, r.w};
    } else {
      return *(const uint4 *)(rowp[t] + off);
    }
  };
  const int nchunks = (int)(pitch * 4 / KC);
  const uint4 *xq4 = (const uint4 *)xq;

  v4i acc[TILES][NPLANE][NB];
#pragma unroll
  for (int t = 0; t < TILES; t++)
#pragma unroll
    for (int p = 0; p < NPLANE; p++)
#pragma unroll
      for (int nb = 0; nb < NB; nb++) acc[t][p][nb] = v4i{0, 0, 0, 0};
  uint32_t st_lo[TILES], st_hi[TILES], st_na[TILES];  // popcounts of the low / all / both bits
#pragma unroll
  for (int t = 0; t < TILES; t++) st_lo[t] = st_hi[t] = st_na[t] = 0;

  constexpr int NX = (XS + NT - 1) / NT;   // staged uint4 per thread per chunk
  static_assert(NX <= 4, "staging registers");
  constexpr bool XFULL = (XS % NT == 0);   // every thread stages NX entries
  // Two genotype register sets, chunk ch lives in set ch & 1.  A register is reloaded with the
  // chunk two ahead as soon as its four K-steps are consumed (rolling prefetch): every wave
  // keeps 4 - 8 loads in flight at all times instead of draining to zero at each chunk edge.
  uint4 ga[2][TILES][LD];
  uint4 xr0 = {0, 0, 0, 0}, xr1 = {0, 0, 0, 0}, xr2 = {0, 0, 0, 0}, xr3 = {0, 0, 0, 0};
  // prologue
#pragma unroll
  for (int t = 0; t < TILES; t++)
#pragma unroll
    for (int it = 0; it < LD; it++) ga[0][t][it] = gload(t, it * 64);
  // entry x of a thread's staging set exists when the panel is a whole number of workgroups, or below its end
  // (three column blocks shared by 16 waves: 1536 entries for 1024 threads)
  auto staged = [&](const int x) -> bool { return XFULL || tid + x * NT < XS; };
  if (staged(0)) xs[0][tid] = xq4[tid];
  if constexpr (NX > 1) if (staged(1)) xs[0][tid + NT] = xq4[tid + NT];
  if constexpr (NX > 2) if (staged(2)) xs[0][tid + 2 * NT] = xq4[tid + 2 * NT];
  if constexpr (NX > 3) if (staged(3)) xs[0][tid + 3 * NT] = xq4[tid + 3 * NT];
#pragma unroll
  for (int t = 0; t < TILES; t++)
#pragma unroll
    for (int it = 0; it < LD; it++)
      ga[1][t][it] = gload(t, (nchunks > 1 ? KC / 4 : 0) + it * 64);
  __syncthreads();

  auto chunk = [&](auto SETC, const int ch) {
    constexpr int SET = decltype(SETC)::value;
    // No branches around the loads: a conditional load makes the compiler's waitcnt insertion
    // fall back to vmcnt(0) at the join, which drains the prefetch.  Past the end the last
    // chunk is simply loaded again (into registers / an LDS buffer nobody reads).
    const int ch1 = ch + 1 < nchunks ? ch + 1 : nchunks - 1;
    const int ch2 = ch + 2 < nchunks ? ch + 2 : nchunks - 1;
    {
      // the digit panel of the next chunk: into registers now, into LDS after the compute
      const uint4 *src = xq4 + (int64_t)ch1 * XS;
      // (unconditional loads at a clamped index: a branch around a load costs the prefetch its waitcnt, see below)
      auto at = [&](const int x) -> int { return XFULL ? tid + x * NT : (tid + x * NT < XS ? tid + x * NT : XS - 1); };
      xr0 = src[at(0)];
      if constexpr (NX > 1) xr1 = src[at(1)];
      if constexpr (NX > 2) xr2 = src[at(2)];
      if constexpr (NX > 3) xr3 = src[at(3)];
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the digit loads up here, a chunk ahead of their use
    if constexpr (SGB & 2) __builtin_amdgcn_s_setprio(2);  // experiment: the MFMA phase outranks waves that are loading
    const int off2 = ch2 * (KC / 4);
    // one K-step of one tile: 16 samples x 16 variants per lane-quad, decode + MFMAs
    auto kstep = [&](const int t, const uint32_t w, const uint4 (&bv)[NB]) {
      uint32_t s0 = w & 0x03030303u, s1 = (w >> 2) & 0x03030303u,
               s2 = (w >> 4) & 0x03030303u, s3 = (w >> 6) & 0x03030303u;
      if constexpr (STATS) {
        // six instructions per dword: the count of the high bits is (all bits) - (low bits), taken at the end
        const uint32_t lo = w & 0x55555555u;
        st_lo[t] += __popc(lo);
        st_hi[t] += __popc(w);              // all bits set (low + high)
        st_na[t] += __popc(lo & (w >> 1));  // both bits of a genotype
      }
      static_assert(!NASKIP || (RAW0 && NPLANE == 2 && !STATS), "the skip is for the missing-value plane of the plain pass");
      bool plane1 = true;
      if constexpr (NASKIP) plane1 = __builtin_amdgcn_ballot_w64(((w & (w >> 1)) & 0x55555555u) != 0u) != 0ull;
#pragma unroll
      for (int p = 0; p < NPLANE; p++) {
        if (NASKIP && p == 1 && !plane1) continue;
        const uint32_t lut = p == 0 ? lutA : p == 1 ? lutB : lutC;
        v4i a;
        if (ABL & 2) {  // ablation: no decode, raw bits as operand
          a = v4i{(int)w, (int)(w ^ lut), (int)s1, (int)s3};
        } else if (RAW0 && p == 0) {
          a = v4i{(int)s0, (int)s1, (int)s2, (int)s3};
        } else {
          a = v4i{(int)lut4(lut, s0), (int)lut4(lut, s1), (int)lut4(lut, s2), (int)lut4(lut, s3)};
        }
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
          v4i b = {(int)bv[nb].x, (int)bv[nb].y, (int)bv[nb].z, (int)bv[nb].w};
          if (ABL & 1) {  // ablation: no MFMA, keep the operands alive
            asm volatile("" ::"v"(a), "v"(b));
          } else {
            acc[t][p][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[t][p][nb], 0, 0, 0);
          }
        }
      }
    };
    // (three column blocks: the digit operands of a K-step are read at its start — a second register copy of them,
    // one K-step ahead, does not fit beside 48 accumulators; the other waves' MFMAs cover the LDS latency)
    constexpr bool PFB = NB <= 2;
    uint4 bv[NB], bn[PFB ? NB : 1];
#pragma unroll
    for (int nb = 0; nb < NB; nb++) bv[nb] = xs[SET][(g * 4) * NCOL + nb * 16 + c];
    if constexpr (PFB) {
      if (ABL & 4) {  // ablation: digit operand read once per chunk instead of once per K-step
#pragma unroll
        for (int nb = 0; nb < NB; nb++) bn[nb] = bv[nb];
      }
    }
#pragma unroll
    for (int it = 0; it < LD; it++) {
#pragma unroll
      for (int d = 0; d < 4; d++) {
        // prefetch the digit operand of the next K-step so that its LDS latency hides
        // under this step's decode + MFMA
        if constexpr (!PFB) {
          if (it * 4 + d > 0) {
#pragma unroll
            for (int nb = 0; nb < NB; nb++) bv[nb] = xs[SET][(it * 16 + g * 4 + d) * NCOL + nb * 16 + c];
          }
        } else if (ABL & 16) {  // ablation: no LDS read of the digit operand
#pragma unroll
          for (int nb = 0; nb < NB; nb++) { bn[nb] = bv[nb]; bn[nb].x += 1; }
        } else if (!(ABL & 4) && it * 4 + d + 1 < LD * 4) {
          const int itn = (it * 4 + d + 1) / 4, dn = (it * 4 + d + 1) % 4;
#pragma unroll
          for (int nb = 0; nb < NB; nb++) bn[nb] = xs[SET][(itn * 16 + g * 4 + dn) * NCOL + nb * 16 + c];
        }
#pragma unroll
        for (int t = 0; t < TILES; t++) {
          const uint32_t w = d == 0 ? ga[SET][t][it].x : d == 1 ? ga[SET][t][it].y
                             : d == 2 ? ga[SET][t][it].z : ga[SET][t][it].w;
          kstep(t, w, bv);
        }
        if constexpr (PFB) {
#pragma unroll
          for (int nb = 0; nb < NB; nb++) bv[nb] = bn[nb];
        }
      }
    }
    // This set is consumed: refill it with the chunk two ahead.  All loads of a tile go out
    // back to back so that both 64-B halves of every 128-B line are requested together
    // (refilling half-way through the chunk, one half at a time, costs 8 % of the bandwidth).
#pragma unroll
    for (int t = 0; t < TILES; t++)
#pragma unroll
      for (int it = 0; it < LD; it++) {
        if (ABL & 32) {  // ablation: no genotype loads after the prologue
          ga[SET][t][it].x ^= (uint32_t)off2;
        } else {
          ga[SET][t][it] = gload(t, off2 + it * 64);
        }
      }
    if constexpr (SGB & 1) {
      // Explicit software pipeline (profiles/r04_sched.txt: 2 % on the two-block kernel): the decode of the following
      // K-steps is slotted into the shadows of the MFMAs — the VALU instructions of a K-step spread evenly behind its
      // MFMAs —, the digit reads one K-step ahead, the refill loads last.  The compiler's own order issues the MFMAs
      // in back-to-back pairs, the second of which stalls its wave for the 12 remaining cycles of the first.
      // masks: 0x008 MFMA, 0x002 VALU, 0x100 DS read, 0x020 VMEM read
      constexpr int VSTEP = TILES * (7 + (NPLANE - (RAW0 ? 1 : 0)) * 4 + (STATS ? 6 : 0));  // VALU per K-step
      constexpr int MSTEP = TILES * NPLANE * NB;                                             // MFMA per K-step
      __builtin_amdgcn_sched_group_barrier(0x100, NB, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
      auto slot = [&](auto IC) {
        constexpr int i = decltype(IC)::value;
        constexpr int nv = VSTEP * (i + 1) / MSTEP - VSTEP * i / MSTEP;
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (nv > 0) __builtin_amdgcn_sched_group_barrier(0x002, nv, 0);
      };
      auto slots = [&](auto self, auto IC) {
        constexpr int i = decltype(IC)::value;
        if constexpr (i < MSTEP) {
          slot(IC);
          self(self, std::integral_constant<int, i + 1>{});
        }
      };
#pragma unroll
      for (int stp = 0; stp < LD * 4; stp++) {
        __builtin_amdgcn_sched_group_barrier(0x100, NB, 0);
        slots(slots, std::integral_constant<int, 0>{});
      }
      __builtin_amdgcn_sched_group_barrier(0x020, TILES * LD, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SGB & 2) __builtin_amdgcn_s_setprio(0);
    {
      if (staged(0)) xs[SET ^ 1][tid] = xr0;
      if constexpr (NX > 1) if (staged(1)) xs[SET ^ 1][tid + NT] = xr1;
      if constexpr (NX > 2) if (staged(2)) xs[SET ^ 1][tid + 2 * NT] = xr2;
      if constexpr (NX > 3) if (staged(3)) xs[SET ^ 1][tid + 3 * NT] = xr3;
    }
    if (!(ABL & 8)) __syncthreads();  // ablation 8: no barrier
  };
  for (int ch = 0; ch < nchunks; ch += 2) {
    chunk(std::integral_constant<int, 0>{}, ch);
    if (ch + 1 < nchunks) chunk(std::integral_constant<int, 1>{}, ch + 1);  // (odd only for KC = 1024)
  }

  // raw accumulators: acc_out[plane][variant][NCOL]
#pragma unroll
  for (int t = 0; t < TILES; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      int64_t j = snp_base + t * 16 + g * 4 + r;
      if (j < m) {
#pragma unroll
        for (int p = 0; p < NPLANE; p++)
#pragma unroll
          for (int nb = 0; nb < NB; nb++)
            acc_out[((int64_t)p * m_out + j) * NCOL + nb * 16 + c] = acc[t][p][nb][r];
      }
    }
  if constexpr (STATS) {
    // a variant row is spread over the 4 k-groups of the wave (lanes c, c+16, c+32, c+48)
#pragma unroll
    for (int t = 0; t < TILES; t++) {
      uint32_t lo = st_lo[t], hi = st_hi[t] - st_lo[t], na = st_na[t];
      lo += __shfl_xor(lo, 16); hi += __shfl_xor(hi, 16); na += __shfl_xor(na, 16);
      lo += __shfl_xor(lo, 32); hi += __shfl_xor(hi, 32); na += __shfl_xor(na, 32);
      const int64_t j = snp_base + t * 16 + c;
      if (g == 0 && j < m) {
        const int32_t n1 = (int32_t)(lo - na), n2 = (int32_t)(hi - na);
        *(int4 *)(counts + 4 * j) =
            int4{(int32_t)(pitch * 4) - n1 - n2 - (int32_t)na - n_pad_samples, n1, n2, (int32_t)na};
      }
    }
  }
}

__device__ __forceinline__ double horner(const int32_t *a, int S) {
  double r = 0;
  for (int s = S - 1; s >= 0; s--) r = r * 256.0 + (double)a[s];
  return r;
}
// sum over the slices of a[s] - k q[s] (the genotype plane sum from the raw-code plane sum:
// g0 = code - 3 na, per slice in exact integers)
__device__ __forceinline__ double horner_sub(const int32_t *a, const int32_t *q, int k, int S) {
  double r = 0;
  for (int s = S - 1; s >= 0; s--) r = r * 256.0 + (double)((long long)a[s] - (long long)k * q[s]);
  return r;
}

// z[j, v] = (P - c_j (Sx - Q)) / (s_j qs),  P = P' - 3 Q  (P' = plane sum of the raw codes)
// One thread per variant for ALL vectors of the launch: its 2 x NCOL raw sums are read once, as whole
// 64 / 128-byte rows, and parked in the thread's own LDS row (stride NCOL 2 + 1 words: conflict free), from
// where the digits of vector v are picked with run-time indices.  (Round 2 ran one thread per variant AND
// vector, each reading 2 S of the row's 2 NCOL words: the rows were fetched once per vector — 1.5 ms per
// pass of 16 vectors at 1M variants, as much as all the panel algebra of a solve.)
template <int NCOL>
__global__ __launch_bounds__(128) void k_cprod_final(const int32_t *__restrict__ acc, int64_t m, int S, int nv,
                                                     const VecMeta *__restrict__ meta,
                                                     const double *__restrict__ center,
                                                     const double *__restrict__ scale, double *Z, int64_t ldz,
                                                     int has_q) {
  constexpr int ROW = 2 * NCOL + 1;
  __shared__ int32_t srow[128 * ROW];
  const int64_t j = (int64_t)blockIdx.x * 128 + threadIdx.x;
  if (j >= m) return;
  int32_t *my = srow + threadIdx.x * ROW;
  {
    const int4 *pa = (const int4 *)(acc + j * NCOL), *pq = (const int4 *)(acc + (m + j) * NCOL);
#pragma unroll
    for (int t = 0; t < NCOL / 4; t++) {
      const int4 x = pa[t];
      my[4 * t] = x.x; my[4 * t + 1] = x.y; my[4 * t + 2] = x.z; my[4 * t + 3] = x.w;
    }
    if (has_q) {
#pragma unroll
      for (int t = 0; t < NCOL / 4; t++) {
        const int4 x = pq[t];
        my[NCOL + 4 * t] = x.x; my[NCOL + 4 * t + 1] = x.y; my[NCOL + 4 * t + 2] = x.z; my[NCOL + 4 * t + 3] = x.w;
      }
    }
  }
  const double c = center ? center[j] : 0.0, s = scale ? scale[j] : 1.0;
  for (int v = 0; v < nv; v++) {
    // no missing plane: Q == 0 and P' == P
    const double P = has_q ? horner_sub(my + v * S, my + NCOL + v * S, 3, S) : horner(my + v * S, S);
    const double Q = has_q ? horner(my + NCOL + v * S, S) : 0.0;
    const double Sx = (double)meta[v].sum_hi * 16777216.0 + (double)meta[v].sum_lo;
    const double qs = meta[v].qscale;
    double z = qs > 0 ? (P - c * (Sx - Q)) / (s * qs) : 0.0 / s;
    // a variant with no non-missing genotype contributes only bedAccScaled's NA entry (0),
    // whatever its centre / scale are (src/bed-acc.h:104); P and Sx - Q are exact integers
    if (P == 0.0 && Sx - Q == 0.0 && qs > 0) z = 0.0;
    if (meta[v].nonfinite) z = __longlong_as_double(0x7ff8000000000000LL);
    Z[j + v * ldz] = z;
  }
}

// ---------------------------------------------------------------------------
// k_prod: contraction over variants on the variant-major image.  One wave owns 256
// samples (16 sample groups x 16) and walks a range of variants 64 at a time:
//   16 dword loads/lane (16 variants x 16 samples), four 4x4 byte transposes so that a
//   register holds the same sample quad of 4 variants, then per sample: 2 VALU for the
//   selector, 2 v_perm for the planes, 2 MFMA (g0 x w-digits, na x (c w)-digits) into
//   one accumulator.
//   A operand: lane l -> digit column (l&15), k-group (l>>4): 16 variants' digits
//   B operand: lane l -> sample group (l&15), same 16 variants, sample u of the group
//   D        : lane l -> sample group (l&15), digit columns 4*(l>>4)+r
// RAWP: the P plane is the device code itself (no look-up).
// TILED: `img` is the streaming-layout copy; a step of the workgroup (64 variants x 256 B) is one tile.
// TAG only changes the kernel's name (warm-start launches).  ABL != 0: -DBSN_ABLATION builds only.
// (Shapes that were measured and dropped — 8-wave workgroups, three register sets, 2 / 4 samples decoded together,
// XCD-aware slab placement, wave pairs / lane halves with 64 accumulators, an explicit MFMA : VALU schedule, the codes as
// the A operand: profiles/r02_ablation.txt, r03_shape_sweeps.txt, r04_two_block_kernels.txt.)
template <int NB, bool CONTIG, bool RAWP, bool HASQ = true, int ABL = 0, int TAG = 0, bool TILED = false>
__global__ __launch_bounds__(256) void k_prod(const uint8_t *__restrict__ img, int64_t pitch,
                                              const int32_t *__restrict__ cols, int64_t col0,
                                              int64_t m_pad, int64_t mc,
                                              const int8_t *__restrict__ wq,
                                              int32_t *__restrict__ acc_out, int64_t n_pad,
                                              uint32_t lutP, uint32_t lutQ) {
  constexpr int NCOL = 16 * NB, WAVES = 4;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int sg = lane & 15, g = lane >> 4;
  const unsigned bx = blockIdx.x, by = blockIdx.y;
  static_assert(!TILED || CONTIG, "streaming layout: contiguous variants");
  const int64_t wbase = ((int64_t)bx * WAVES + wave) * 256;  // first sample of this wave (n_pad is a multiple of 1024)
  const int64_t wbyte = wbase / 4 + sg * 4;
  const uint32_t lane_off = (uint32_t)(g * 16 * pitch + wbyte);
  const int64_t j0 = (int64_t)by * mc;
  int64_t j1 = j0 + mc;
  if (j1 > m_pad) j1 = m_pad;
  const uint4 *wq4 = (const uint4 *)wq;

  v4i acc[16][NB];
#pragma unroll
  for (int u = 0; u < 16; u++)
#pragma unroll
    for (int nb = 0; nb < NB; nb++) acc[u][nb] = v4i{0, 0, 0, 0};

  // digit operands of one 64-variant step: (4 k-groups) x (2 planes) x NCOL entries of 16 B,
  // contiguous in wq; shared by the 4 waves through LDS, double-buffered
  constexpr int WS = 8 * NCOL;
  __shared__ uint4 ws[2][WS];
  static_assert((WS & (WS - 1)) == 0 && WS <= 64 * WAVES, "digit staging");
  // Two genotype register sets, step s lives in set s & 1 (as in k_cprod): a set is free as
  // soon as its 4x4 byte transposes are done, so it is refilled with the step two ahead
  // right there.  No branches around loads (see k_cprod); past the end the last step is
  // loaded again.
  uint32_t X[2][16];
  uint4 wreg = {0, 0, 0, 0};
  const int wtid = tid & (WS - 1);
  auto load = [&](int64_t jb, uint32_t *dst) {
    if (CONTIG) {
      // buffer loads: scalar descriptor (re-based per step) + scalar row offset + one 32-bit lane
      // offset, so the 16 addresses of a step cost no VALU (global loads took a 64-bit add each)
      if constexpr (TILED) {
        // a wave owns 64 B of a 256-B column block: the four waves of the workgroup cover one tile per step
        const int64_t sbw = (int64_t)bx;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(img + (((col0 + jb) >> 6) * (pitch >> 8) + sbw) * 16384), 0, 0x7fffffff, 0x00020000);
        const int toff = g * 4096 + wave * 64 + sg * 4;
#pragma unroll
        for (int r = 0; r < 16; r++) dst[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, toff, r * 256, 0);
      } else {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(img + (col0 + jb) * pitch), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int r = 0; r < 16; r++) dst[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)lane_off, r * (int)pitch, 0);
      }
    } else {
      const int4 *ip = (const int4 *)(cols + jb + g * 16);
#pragma unroll
      for (int r4 = 0; r4 < 4; r4++) {
        int4 id = ip[r4];
        dst[r4 * 4 + 0] = *(const uint32_t *)(img + (int64_t)id.x * pitch + wbyte);
        dst[r4 * 4 + 1] = *(const uint32_t *)(img + (int64_t)id.y * pitch + wbyte);
        dst[r4 * 4 + 2] = *(const uint32_t *)(img + (int64_t)id.z * pitch + wbyte);
        dst[r4 * 4 + 3] = *(const uint32_t *)(img + (int64_t)id.w * pitch + wbyte);
      }
    }
  };
  // j0 < j1 by the launch geometry of prod_planes (every K-slab has at least one step)
  if (j0 >= j1) return;  // never taken; without it the compiler allocates 144 instead of 80 VGPRs
  const int64_t jlast = j1 - 64;
  ws[0][wtid] = wq4[(j0 / 16) * 2 * NCOL + wtid];
  load(j0, X[0]);
  load(j0 + 64 < j1 ? j0 + 64 : jlast, X[1]);
  __syncthreads();

  auto step = [&](auto SETC, const int64_t jb) {
    constexpr int SET = decltype(SETC)::value;
    const int64_t jn1 = jb + 64 < j1 ? jb + 64 : jlast, jn2 = jb + 128 < j1 ? jb + 128 : jlast;
    wreg = wq4[(jn1 / 16) * 2 * NCOL + wtid];  // next step's digits: registers now, LDS later
    __builtin_amdgcn_sched_barrier(0);
    v4i aw[NB], awc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      uint4 t0 = ws[SET][(g * 2 + 0) * NCOL + nb * 16 + sg];
      uint4 t1 = ws[SET][(g * 2 + 1) * NCOL + nb * 16 + sg];
      aw[nb] = v4i{(int)t0.x, (int)t0.y, (int)t0.z, (int)t0.w};
      awc[nb] = v4i{(int)t1.x, (int)t1.y, (int)t1.z, (int)t1.w};
    }
    // T[q][r4]: byte b = byte q of X[4*r4 + b]
    uint32_t T[4][4];
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) {
      const uint32_t x0 = X[SET][4 * r4], x1 = X[SET][4 * r4 + 1], x2 = X[SET][4 * r4 + 2], x3 = X[SET][4 * r4 + 3];
      const uint32_t lo01 = perm8(x1, x0, 0x05010400u);  // x0.b0 x1.b0 x0.b1 x1.b1
      const uint32_t hi01 = perm8(x1, x0, 0x07030602u);  // x0.b2 x1.b2 x0.b3 x1.b3
      const uint32_t lo23 = perm8(x3, x2, 0x05010400u);
      const uint32_t hi23 = perm8(x3, x2, 0x07030602u);
      T[0][r4] = perm8(lo23, lo01, 0x05040100u);  // lo01.b0 lo01.b1 lo23.b0 lo23.b1
      T[1][r4] = perm8(lo23, lo01, 0x07060302u);
      T[2][r4] = perm8(hi23, hi01, 0x05040100u);
      T[3][r4] = perm8(hi23, hi01, 0x07060302u);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ABL & 32) {  // ablation: no genotype loads after the prologue
#pragma unroll
      for (int r = 0; r < 16; r++) X[SET][r] += (uint32_t)jn2;
    } else {
      load(jn2, X[SET]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int u0 = 0; u0 < 4; u0++) {
        v4i g0, na;
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
          const uint32_t sel = (T[q][r4] >> (2 * u0)) & 0x03030303u;
          if (ABL & 2) {  // ablation: no decode
            g0[r4] = (int)T[q][r4];
            na[r4] = (int)(T[q][r4] ^ lutQ);
          } else {
            g0[r4] = RAWP ? (int)sel : (int)lut4(lutP, sel);
            if (HASQ) na[r4] = (int)lut4(lutQ, sel);
          }
        }
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
          if (ABL & 1) {  // ablation: no MFMA
            asm volatile("" ::"v"(g0), "v"(na), "v"(aw[nb]), "v"(awc[nb]));
          } else {
            acc[q * 4 + u0][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(aw[nb], g0, acc[q * 4 + u0][nb], 0, 0, 0);
            if (HASQ)
              acc[q * 4 + u0][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(awc[nb], na, acc[q * 4 + u0][nb], 0, 0, 0);
          }
        }
      }
    ws[SET ^ 1][wtid] = wreg;
    __syncthreads();
  };
  for (int64_t jb = j0; jb < j1; jb += 128) {
    step(std::integral_constant<int, 0>{}, jb);
    if (jb + 64 < j1) step(std::integral_constant<int, 1>{}, jb + 64);
  }
  // raw accumulators: acc_out[ky][sample][NCOL], lane holds columns nb*16 + 4g .. +3
#pragma unroll
  for (int u = 0; u < 16; u++) {
    const int64_t i = wbase + sg * 16 + u;
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
      *(v4i *)(acc_out + (((int64_t)by * n_pad + i) * NCOL + nb * 16 + 4 * g)) = acc[u][nb];
  }
}

// ---------------------------------------------------------------------------
// k_prodT: Y = A~ X on the SAMPLE-MAJOR copy of the image (bsn_bed::d_smaj): the contraction index — the variants —
// is the contiguous one there, so the kernel is k_cprod with the roles of samples and variants exchanged: one wave
// owns 32 samples (two 16-row MFMA tiles) and walks its slab of variants 512 at a time, the 16 waves of a workgroup
// share the digit panel of the chunk through LDS (double-buffered).  Against k_prod on the variant-major image: no
// 4 x 4 byte transposes (11 instead of 13.3 VALU instructions per 4 MFMAs), 16 instead of 128 accumulator registers,
// four waves per SIMD instead of two.  Same integer sums -> bit-identical results.
//   A operand: lane l -> sample row (l & 15), k-group (l >> 4): 16 variants of that sample (one decoded dword)
//   B operand: lane l -> digit column (l & 15), the same 16 variants: w-digits for the code plane, (c w - 3 w)-digits
//              for the missing-value plane (k_quant mode 1 with the cprod byte order), BOTH into one accumulator
//   D        : lane l -> digit column (l & 15), sample rows 4 * (l >> 4) + r
// The variants are split into gridDim.y slabs of `cps` chunks (int32 partial sums per slab, added by k_prod_final in
// exact int64 like k_prod's): 782 workgroups of 512 samples alone would fill 3.05 rounds of 256 CUs.
// The copy is CHUNK-MAJOR (bsn_bed::d_smaj): byte of the operator's first variant in a sample row (col0 / 4, a multiple of 16).
// NASKIP: as in k_cprod — the missing-value plane of a K-step (16 samples x 64 variants) only when it has a missing code.
template <int NB, bool HASQ, int TILES = 2, int WAVES = 16, int TAG = 0, int SGB = 3, bool NASKIP = false>
__global__ __launch_bounds__(64 * WAVES) void k_prodT(const uint8_t *__restrict__ simg, int64_t rows_t, int64_t chunk0,
                                                      int nchunks, int cps,
                                                      const int8_t *__restrict__ wq, int32_t *__restrict__ acc_out,
                                                      int64_t n_pad, uint32_t lutQ, int seg_bs, int seg_stride, int seg_off) {
  constexpr int KC = 512, NCOL = 16 * NB, LD = KC / 256;
  constexpr int XS = KC / 16 * 2 * NCOL;   // uint4 entries of one chunk's digit panel (two planes)
  constexpr int NT = 64 * WAVES, NX = XS / NT;
  static_assert(XS % NT == 0 && NX >= 1 && NX <= 4, "digit staging");
  __shared__ uint4 xs[2][XS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c = lane & 15, g = lane >> 4;
  // seg_bs > 0 (a segment of the sharded solve's product pass, op_prod_segments): the launch covers blocks
  // [seg_off, seg_off + seg_bs) of every piece of seg_stride blocks
  const unsigned bx = seg_bs > 0 ? (blockIdx.x / (unsigned)seg_bs) * (unsigned)seg_stride + (unsigned)seg_off + blockIdx.x % (unsigned)seg_bs
                                 : blockIdx.x;
  const int64_t wg_base = (int64_t)bx * (WAVES * 16 * TILES);
  const int64_t row_base = wg_base + wave * (16 * TILES);
  uint32_t voff[TILES];
#pragma unroll
  for (int t = 0; t < TILES; t++) {
    int64_t i = row_base + t * 16 + c;
    if (i > rows_t - 1) i = rows_t - 1;
    voff[t] = (uint32_t)((i - wg_base) * (KC / 4) + g * 16);
  }
  const int ch_begin = (int)blockIdx.y * cps;
  int ch_end = ch_begin + cps;
  if (ch_end > nchunks) ch_end = nchunks;
  if (ch_begin >= ch_end) return;   // (never: the launch geometry gives every slab at least one chunk)
  if (wg_base >= rows_t) return;    // a workgroup wholly past the copy's rows (a segment's padding blocks): nothing to read or write
  // chunk ch of the slab: the workgroup's 512 sample rows x 128 B are ONE contiguous 64-KB run of the copy
  const int64_t chunk_stride = rows_t * (KC / 4);
  const uint8_t *const base0 = simg + ((chunk0 + ch_begin) * rows_t + wg_base) * (KC / 4);
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  auto gload = [&](const int t, const int ch, const int off) -> uint4 {
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void *)(base0 + (int64_t)ch * chunk_stride), 0, 0x7fffffff, 0x00020000);
    const v4u r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff[t], off, 0);
    return uint4{r.x, r.y, r.z, r.w};
  };
  const int nch = ch_end - ch_begin;
  const uint4 *xq4 = (const uint4 *)wq + (int64_t)ch_begin * XS;

  v4i acc[TILES][NB];
#pragma unroll
  for (int t = 0; t < TILES; t++)
#pragma unroll
    for (int nb = 0; nb < NB; nb++) acc[t][nb] = v4i{0, 0, 0, 0};
  uint4 ga[2][TILES][LD];
  uint4 xr0 = {0, 0, 0, 0}, xr1 = {0, 0, 0, 0}, xr2 = {0, 0, 0, 0}, xr3 = {0, 0, 0, 0};   // (scalars: an array captured by the lambda below ends up in scratch)
#pragma unroll
  for (int t = 0; t < TILES; t++)
#pragma unroll
    for (int it = 0; it < LD; it++) ga[0][t][it] = gload(t, 0, it * 64);
#pragma unroll
  for (int x = 0; x < NX; x++) xs[0][tid + x * NT] = xq4[tid + x * NT];
#pragma unroll
  for (int t = 0; t < TILES; t++)
#pragma unroll
    for (int it = 0; it < LD; it++) ga[1][t][it] = gload(t, nch > 1 ? 1 : 0, it * 64);
  __syncthreads();

  auto chunk = [&](auto SETC, const int ch) {
    constexpr int SET = decltype(SETC)::value;
    // branch-free prefetch as in k_cprod: past the end of the slab its last chunk is loaded again
    const int ch1 = ch + 1 < nch ? ch + 1 : nch - 1;
    const int ch2 = ch + 2 < nch ? ch + 2 : nch - 1;
    xr0 = xq4[(int64_t)ch1 * XS + tid];
    if constexpr (NX > 1) xr1 = xq4[(int64_t)ch1 * XS + tid + NT];
    if constexpr (NX > 2) xr2 = xq4[(int64_t)ch1 * XS + tid + 2 * NT];
    if constexpr (NX > 3) xr3 = xq4[(int64_t)ch1 * XS + tid + 3 * NT];
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SGB & 2) __builtin_amdgcn_s_setprio(2);
    constexpr int NP = HASQ ? 2 : 1;
    constexpr bool PFB = NB <= 2;   // (as in k_cprod: no second register copy of the digit operands with three column blocks)
    uint4 bv[NP][NB], bn[PFB ? NP : 1][PFB ? NB : 1];
#pragma unroll
    for (int p = 0; p < NP; p++)
#pragma unroll
      for (int nb = 0; nb < NB; nb++) {
        bv[p][nb] = xs[SET][((g * 4) * 2 + p) * NCOL + nb * 16 + c];
        if constexpr (PFB) bn[p][nb] = bv[p][nb];
      }
#pragma unroll
    for (int it = 0; it < LD; it++) {
#pragma unroll
      for (int d = 0; d < 4; d++) {
        if constexpr (!PFB) {
          if (it * 4 + d > 0) {
#pragma unroll
            for (int p = 0; p < NP; p++)
#pragma unroll
              for (int nb = 0; nb < NB; nb++) bv[p][nb] = xs[SET][((it * 16 + g * 4 + d) * 2 + p) * NCOL + nb * 16 + c];
          }
        } else if (it * 4 + d + 1 < LD * 4) {
          const int itn = (it * 4 + d + 1) / 4, dn = (it * 4 + d + 1) % 4;
#pragma unroll
          for (int p = 0; p < NP; p++)
#pragma unroll
            for (int nb = 0; nb < NB; nb++) bn[p][nb] = xs[SET][((itn * 16 + g * 4 + dn) * 2 + p) * NCOL + nb * 16 + c];
        }
#pragma unroll
        for (int t = 0; t < TILES; t++) {
          const uint32_t w = d == 0 ? ga[SET][t][it].x : d == 1 ? ga[SET][t][it].y
                             : d == 2 ? ga[SET][t][it].z : ga[SET][t][it].w;
          const uint32_t s0 = w & 0x03030303u, s1 = (w >> 2) & 0x03030303u, s2 = (w >> 4) & 0x03030303u,
                         s3 = (w >> 6) & 0x03030303u;
          const v4i a0 = {(int)s0, (int)s1, (int)s2, (int)s3};
#pragma unroll
          for (int nb = 0; nb < NB; nb++) {
            const v4i b = {(int)bv[0][nb].x, (int)bv[0][nb].y, (int)bv[0][nb].z, (int)bv[0][nb].w};
            acc[t][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b, acc[t][nb], 0, 0, 0);
          }
          if constexpr (HASQ) {
            static_assert(!NASKIP || HASQ, "the skip is for the missing-value plane");
            bool plane1 = true;
            if constexpr (NASKIP) plane1 = __builtin_amdgcn_ballot_w64(((w & (w >> 1)) & 0x55555555u) != 0u) != 0ull;
            if (plane1) {
              const v4i a1 = {(int)lut4(lutQ, s0), (int)lut4(lutQ, s1), (int)lut4(lutQ, s2), (int)lut4(lutQ, s3)};
#pragma unroll
              for (int nb = 0; nb < NB; nb++) {
                const v4i b = {(int)bv[1][nb].x, (int)bv[1][nb].y, (int)bv[1][nb].z, (int)bv[1][nb].w};
                acc[t][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b, acc[t][nb], 0, 0, 0);
              }
            }
          }
        }
        if constexpr (PFB) {
#pragma unroll
          for (int p = 0; p < NP; p++)
#pragma unroll
            for (int nb = 0; nb < NB; nb++) bv[p][nb] = bn[p][nb];
        }
      }
    }
#pragma unroll
    for (int t = 0; t < TILES; t++)
#pragma unroll
      for (int it = 0; it < LD; it++) ga[SET][t][it] = gload(t, ch2, it * 64);
    if constexpr (SGB & 1) {
      // the explicit MFMA / decode pipeline of k_cprod (0x008 MFMA, 0x002 VALU, 0x100 DS read, 0x020 VMEM read)
      constexpr int VSTEP = TILES * (7 + (HASQ ? 4 : 0)), MSTEP = TILES * NP * NB;
      __builtin_amdgcn_sched_group_barrier(0x100, NP * NB, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
      auto slots = [&](auto self, auto IC) {
        constexpr int i = decltype(IC)::value;
        if constexpr (i < MSTEP) {
          constexpr int nv = VSTEP * (i + 1) / MSTEP - VSTEP * i / MSTEP;
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if constexpr (nv > 0) __builtin_amdgcn_sched_group_barrier(0x002, nv, 0);
          self(self, std::integral_constant<int, i + 1>{});
        }
      };
#pragma unroll
      for (int stp = 0; stp < LD * 4; stp++) {
        __builtin_amdgcn_sched_group_barrier(0x100, NP * NB, 0);
        slots(slots, std::integral_constant<int, 0>{});
      }
      __builtin_amdgcn_sched_group_barrier(0x020, TILES * LD, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SGB & 2) __builtin_amdgcn_s_setprio(0);
    xs[SET ^ 1][tid] = xr0;
    if constexpr (NX > 1) xs[SET ^ 1][tid + NT] = xr1;
    if constexpr (NX > 2) xs[SET ^ 1][tid + 2 * NT] = xr2;
    if constexpr (NX > 3) xs[SET ^ 1][tid + 3 * NT] = xr3;
    __syncthreads();
  };
  for (int ch = 0; ch < nch; ch += 2) {
    chunk(std::integral_constant<int, 0>{}, ch);
    if (ch + 1 < nch) chunk(std::integral_constant<int, 1>{}, ch + 1);
  }
  // raw accumulators: acc_out[slab][sample][NCOL] (k_prod's layout, read by k_prod_final)
#pragma unroll
  for (int t = 0; t < TILES; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int64_t i = row_base + t * 16 + g * 4 + r;
      if (i < n_pad) {
#pragma unroll
        for (int nb = 0; nb < NB; nb++)
          acc_out[((int64_t)blockIdx.y * n_pad + i) * NCOL + nb * 16 + c] = acc[t][nb][r];
      }
    }
}

// y[i, v] = (sum_ky value(i) - C_v) / qs_v, gathered through rows[].  One thread per output
// row reads its NCOL contiguous int32 per K-chunk (16 B loads, consecutive rows adjacent).
// blk_rows > 0: blocked output for the reduce-scatter of a segment (op_prod_segments) — row i belongs to piece
// i / blk_rows and goes to Y[(piece * nv + v) * blk_rows + i % blk_rows]; rows[i] < 0 is padding (zero).
template <int NCOL>
__global__ __launch_bounds__(256) void k_prod_final(const int32_t *__restrict__ acc, int64_t n_pad, int ky, int S, int nv,
                             const VecMeta *meta, const int32_t *rows, int64_t n, double *Y,
                             int64_t ldy, int sub_const, double beta, int64_t blk_rows = 0) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t i2 = rows ? (int64_t)rows[i] : i;
  if (i2 < 0) {
    for (int v = 0; v < nv; v++) Y[(i / blk_rows * nv + v) * blk_rows + i % blk_rows] = 0.0;
    return;
  }
  long long d[NCOL];
#pragma unroll
  for (int c = 0; c < NCOL; c++) d[c] = 0;
  for (int k = 0; k < ky; k++) {
    const int4 *a = (const int4 *)(acc + ((int64_t)k * n_pad + i2) * NCOL);
#pragma unroll
    for (int c4 = 0; c4 < NCOL / 4; c4++) {
      int4 t = a[c4];
      d[4 * c4 + 0] += t.x;
      d[4 * c4 + 1] += t.y;
      d[4 * c4 + 2] += t.z;
      d[4 * c4 + 3] += t.w;
    }
  }
  for (int v = 0; v < nv; v++) {
    double r = 0;
    for (int s = S - 1; s >= 0; s--) {
      long long dv = 0;
#pragma unroll
      for (int c = 0; c < NCOL; c++) dv = (c == v * S + s) ? d[c] : dv;
      r = r * 256.0 + (double)dv;
    }
    double C = sub_const ? (double)meta[v].sum2_hi * 16777216.0 + (double)meta[v].sum2_lo : 0.0;
    double qs = meta[v].qscale;
    double y = qs > 0 ? (r - C) / qs : 0.0;
    if (meta[v].nonfinite) y = __longlong_as_double(0x7ff8000000000000LL);
    if (blk_rows > 0)
      Y[(i / blk_rows * nv + v) * blk_rows + i % blk_rows] = y;
    else
      Y[i + v * ldy] = beta != 0.0 ? beta * Y[i + v * ldy] + y : y;
  }
}

// ---------------------------------------------------------------------------
// Byte image (bsn_bed::bits == 8): one int8 grid index k per genotype, 0x80 = missing.  The loaded
// bytes ARE the MFMA operand (after zeroing the missing marker when the data has any); 1 B per genotype
// makes both products plainly HBM-bound (1 - 2 MFMA per 16 B instead of 8).
__device__ __forceinline__ uint32_t val8m(uint32_t w, uint32_t &na) {
  const uint32_t t = w ^ 0x80808080u;
  const uint32_t y = (t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
  const uint32_t z = ~(y | t | 0x7F7F7F7Fu);  // 0x80 iff the byte of w is 0x80
  na = z >> 7;
  return w & ~(z | (z - na));
}

// k_cprod8: contraction over samples.  One wave owns 16 variants for the whole sample range; the 8 waves
// of a workgroup share the digit panel of the current 256-sample chunk through LDS (double-buffered).
//   A operand: lane l -> variant row (l&15), k-group (l>>4): 16 B = 16 samples;  B: digit column (l&15)
template <int NB, bool HASNA, bool CONTIG>
__global__ __launch_bounds__(512) void k_cprod8(const uint8_t *__restrict__ img, int64_t pitch,
                                                const int32_t *__restrict__ cols, int64_t col0, int64_t m,
                                                const int8_t *__restrict__ xq, int32_t *__restrict__ acc_out,
                                                int64_t m_out) {
  constexpr int KC = 256, LD = KC / 64, NCOL = 16 * NB, XS = KC / 16 * NCOL, WAVES = 8, NT = 64 * WAVES;
  constexpr int NPL = HASNA ? 2 : 1;
  __shared__ uint4 xs[2][XS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int64_t snp_base = ((int64_t)blockIdx.x * WAVES + wave) * 16;
  int64_t j = snp_base + c;
  if (j > m - 1) j = m - 1;
  const int64_t col = CONTIG ? col0 + j : (int64_t)cols[j];
  const uint8_t *rowp = img + col * pitch + g * 16;
  const int nchunks = (int)(pitch / KC);
  const uint4 *xq4 = (const uint4 *)xq;
  v4i acc[NPL][NB];
#pragma unroll
  for (int p = 0; p < NPL; p++)
#pragma unroll
    for (int nb = 0; nb < NB; nb++) acc[p][nb] = v4i{0, 0, 0, 0};
  static_assert(XS <= NT, "digit staging");
  uint4 ga[2][LD], xr = {0, 0, 0, 0};
#pragma unroll
  for (int it = 0; it < LD; it++) ga[0][it] = *(const uint4 *)(rowp + it * 64);
  if (tid < XS) xs[0][tid] = xq4[tid];
#pragma unroll
  for (int it = 0; it < LD; it++) ga[1][it] = *(const uint4 *)(rowp + (nchunks > 1 ? KC : 0) + it * 64);
  __syncthreads();
  auto chunk = [&](auto SETC, const int ch) {
    constexpr int SET = decltype(SETC)::value;
    const int ch1 = ch + 1 < nchunks ? ch + 1 : nchunks - 1, ch2 = ch + 2 < nchunks ? ch + 2 : nchunks - 1;
    if (tid < XS) xr = xq4[(int64_t)ch1 * XS + tid];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < LD; it++) {
      const uint4 w = ga[SET][it];
      v4i val, na;
      if constexpr (HASNA) {
        uint32_t n0, n1, n2, n3;
        val = v4i{(int)val8m(w.x, n0), (int)val8m(w.y, n1), (int)val8m(w.z, n2), (int)val8m(w.w, n3)};
        na = v4i{(int)n0, (int)n1, (int)n2, (int)n3};
      } else {
        val = v4i{(int)w.x, (int)w.y, (int)w.z, (int)w.w};
      }
#pragma unroll
      for (int nb = 0; nb < NB; nb++) {
        const uint4 bv = xs[SET][(it * 4 + g) * NCOL + nb * 16 + c];
        const v4i b = {(int)bv.x, (int)bv.y, (int)bv.z, (int)bv.w};
        acc[0][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(val, b, acc[0][nb], 0, 0, 0);
        if constexpr (HASNA) acc[1][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(na, b, acc[1][nb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int it = 0; it < LD; it++) ga[SET][it] = *(const uint4 *)(rowp + (int64_t)ch2 * KC + it * 64);
    __builtin_amdgcn_sched_barrier(0);
    if (tid < XS) xs[SET ^ 1][tid] = xr;
    __syncthreads();
  };
  for (int ch = 0; ch < nchunks; ch += 2) {
    chunk(std::integral_constant<int, 0>{}, ch);
    if (ch + 1 < nchunks) chunk(std::integral_constant<int, 1>{}, ch + 1);
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int64_t jo = snp_base + g * 4 + r;
    if (jo < m) {
#pragma unroll
      for (int p = 0; p < NPL; p++)
#pragma unroll
        for (int nb = 0; nb < NB; nb++) acc_out[((int64_t)p * m_out + jo) * NCOL + nb * 16 + c] = acc[p][nb][r];
    }
  }
}

// z[j, v] = (vstep P + (voff - c_j) (Sx - Q)) / (s_j qs):  value = voff + vstep k on non-missing genotypes
__global__ void k_cprod_final8(const int32_t *acc, int64_t m, int ncol, int S, const VecMeta *meta,
                               const double *center, const double *scale, double *Z, int64_t ldz, int has_q,
                               double vstep, double voff) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int v = blockIdx.y;
  if (j >= m) return;
  const double P = horner(acc + j * ncol + v * S, S);
  const double Q = has_q ? horner(acc + (m + j) * ncol + v * S, S) : 0.0;
  const double Sx = (double)meta[v].sum_hi * 16777216.0 + (double)meta[v].sum_lo;
  const double c = center ? center[j] : 0.0, s = scale ? scale[j] : 1.0, qs = meta[v].qscale;
  double z = qs > 0 ? (vstep * P + (voff - c) * (Sx - Q)) / (s * qs) : 0.0 / s;
  if (meta[v].nonfinite) z = __longlong_as_double(0x7ff8000000000000LL);
  Z[j + v * ldz] = z;
}

// k_prod8: contraction over variants on the variant-major byte image.  One wave owns 64 samples
// (16 sample groups x 4) and walks a range of variants 64 at a time: 16 dword loads per lane (16 variants
// x 4 samples), four 4x4 byte transposes, then per sample one operand of 16 variants.
template <int NB, bool HASNA, bool CONTIG>
__global__ __launch_bounds__(256) void k_prod8(const uint8_t *__restrict__ img, int64_t pitch,
                                               const int32_t *__restrict__ cols, int64_t col0, int64_t m_pad,
                                               int64_t mc, const int8_t *__restrict__ wq,
                                               int32_t *__restrict__ acc_out, int64_t n_pad) {
  constexpr int NCOL = 16 * NB, WS = 8 * NCOL;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, sg = lane & 15, g = lane >> 4;
  const int64_t wbase = ((int64_t)blockIdx.x * 4 + wave) * 64;  // first sample of this wave
  const int64_t wbyte = wbase + sg * 4;
  const int64_t j0 = (int64_t)blockIdx.y * mc;
  int64_t j1 = j0 + mc;
  if (j1 > m_pad) j1 = m_pad;
  if (j0 >= j1) return;
  const uint4 *wq4 = (const uint4 *)wq;
  v4i acc[4][NB];
#pragma unroll
  for (int u = 0; u < 4; u++)
#pragma unroll
    for (int nb = 0; nb < NB; nb++) acc[u][nb] = v4i{0, 0, 0, 0};
  __shared__ uint4 ws[2][WS];
  static_assert(WS <= 256, "digit staging");
  const int wtid = tid & (WS - 1);
  uint32_t X[2][16];
  auto load = [&](int64_t jb, uint32_t *dst) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int64_t jj = jb + g * 16 + r;
      const int64_t col = CONTIG ? col0 + jj : (int64_t)cols[jj];
      dst[r] = *(const uint32_t *)(img + col * pitch + wbyte);
    }
  };
  const int64_t jlast = j1 - 64;
  uint4 wreg = {0, 0, 0, 0};
  ws[0][wtid] = wq4[(j0 / 16) * 2 * NCOL + wtid];
  load(j0, X[0]);
  load(j0 + 64 < j1 ? j0 + 64 : jlast, X[1]);
  __syncthreads();
  auto step = [&](auto SETC, const int64_t jb) {
    constexpr int SET = decltype(SETC)::value;
    const int64_t jn1 = jb + 64 < j1 ? jb + 64 : jlast, jn2 = jb + 128 < j1 ? jb + 128 : jlast;
    wreg = wq4[(jn1 / 16) * 2 * NCOL + wtid];
    v4i aw[NB], awc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      const uint4 t0 = ws[SET][(g * 2 + 0) * NCOL + nb * 16 + sg], t1 = ws[SET][(g * 2 + 1) * NCOL + nb * 16 + sg];
      aw[nb] = v4i{(int)t0.x, (int)t0.y, (int)t0.z, (int)t0.w};
      awc[nb] = v4i{(int)t1.x, (int)t1.y, (int)t1.z, (int)t1.w};
    }
    uint32_t T[4][4];  // T[q][r4]: byte b = sample q of variant 4 r4 + b
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) {
      const uint32_t x0 = X[SET][4 * r4], x1 = X[SET][4 * r4 + 1], x2 = X[SET][4 * r4 + 2], x3 = X[SET][4 * r4 + 3];
      const uint32_t lo01 = perm8(x1, x0, 0x05010400u), hi01 = perm8(x1, x0, 0x07030602u);
      const uint32_t lo23 = perm8(x3, x2, 0x05010400u), hi23 = perm8(x3, x2, 0x07030602u);
      T[0][r4] = perm8(lo23, lo01, 0x05040100u);
      T[1][r4] = perm8(lo23, lo01, 0x07060302u);
      T[2][r4] = perm8(hi23, hi01, 0x05040100u);
      T[3][r4] = perm8(hi23, hi01, 0x07060302u);
    }
    load(jn2, X[SET]);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      v4i val, na;
      if constexpr (HASNA) {
        uint32_t n0, n1, n2, n3;
        val = v4i{(int)val8m(T[q][0], n0), (int)val8m(T[q][1], n1), (int)val8m(T[q][2], n2), (int)val8m(T[q][3], n3)};
        na = v4i{(int)n0, (int)n1, (int)n2, (int)n3};
      } else {
        val = v4i{(int)T[q][0], (int)T[q][1], (int)T[q][2], (int)T[q][3]};
      }
#pragma unroll
      for (int nb = 0; nb < NB; nb++) {
        acc[q][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(aw[nb], val, acc[q][nb], 0, 0, 0);
        if constexpr (HASNA) acc[q][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(awc[nb], na, acc[q][nb], 0, 0, 0);
      }
    }
    ws[SET ^ 1][wtid] = wreg;
    __syncthreads();
  };
  for (int64_t jb = j0; jb < j1; jb += 128) {
    step(std::integral_constant<int, 0>{}, jb);
    if (jb + 64 < j1) step(std::integral_constant<int, 1>{}, jb + 64);
  }
  if (wbase < n_pad) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int64_t i = wbase + sg * 4 + u;
#pragma unroll
      for (int nb = 0; nb < NB; nb++)
        *(v4i *)(acc_out + (((int64_t)blockIdx.y * n_pad + i) * NCOL + nb * 16 + 4 * g)) = acc[u][nb];
    }
  }
}

void require_bits(const bsn_bed *b, int bits, const char *what) {
  if (b->bits != bits)
    fail("%s is not available for this handle: it needs %s", what,
         bits == 2 ? "a 2-bit genotype image (.bed, or an FBM.code256 whose codes decode to 0 / 1 / 2 / NA)"
                   : "a byte (dosage) image");
}

// ---------------------------------------------------------------------------
// more = true: a further piece of the launch begun before (a product pass in segments: the kernels of its segments are
// timed one by one — what lies between them, finalize kernels and the collective's hand-over, is not streaming time — and
// count as ONE launch)
void prof_begin(bsn_op *op, int kind, bool more) {
  if (!op->profile) return;
  if (op->prof_kind_override >= 0) kind = op->prof_kind_override;
  op->ev_more.push_back(more);
  hipEvent_t a, b;
  BSN_HIP(hipEventCreate(&a));
  BSN_HIP(hipEventCreate(&b));
  op->ev_begin.push_back(a);
  op->ev_end.push_back(b);
  op->ev_kind.push_back(kind);
  BSN_HIP(hipEventRecord(a, op->bed->stream));
}
void prof_end(bsn_op *op) {
  if (!op->profile) return;
  BSN_HIP(hipEventRecord(op->ev_end.back(), op->bed->stream));
  op->prof_kernel[op->ev_kind.back() % kProfKinds] = g_last_kernel;
}
void prof_collect(bsn_op *op, double ms[kProfKinds], int count[kProfKinds]) {
  for (int k = 0; k < kProfKinds; k++) ms[k] = 0, count[k] = 0;
  for (size_t i = 0; i < op->ev_begin.size(); i++) {
    float t = 0;
    // (a pair whose end was never recorded — a launch that failed, a solve that was aborted — is dropped, not an error)
    if (hipEventSynchronize(op->ev_end[i]) == hipSuccess &&
        hipEventElapsedTime(&t, op->ev_begin[i], op->ev_end[i]) == hipSuccess) {
      ms[op->ev_kind[i]] += t;
      if (!op->ev_more[i]) count[op->ev_kind[i]]++;
    }
    (void)hipEventDestroy(op->ev_begin[i]);
    (void)hipEventDestroy(op->ev_end[i]);
  }
  (void)hipGetLastError();
  op->ev_begin.clear();
  op->ev_end.clear();
  op->ev_kind.clear();
  op->ev_more.clear();
}

// Column blocks of 16 digit columns per launch.  Three (48 columns: 16 vectors x 3 slices, the early steps of a solve
// whose vectors are wanted beyond the 16-bit floor, svd_driver.hpp) exist for the two shapes such a solve runs on —
// k_cprod on the 2-bit image and k_prodT on its sample-major copy; everything else stays at two.
constexpr int kMaxCols = 48;
static int pick_nb(int ncols_needed) { return ncols_needed <= 16 ? 1 : ncols_needed <= 32 ? 2 : 3; }
static bool nb3_allowed() {
  static const bool on = getenv("BSN_NO_NB3") == nullptr;   // A/B switch: the same sums in two launches (bit-identical)
  return on;
}
constexpr int kMetaVecs = 32;   // vectors per launch at most
static VecMeta *meta_buffer(bsn_op *op) {
  return (VecMeta *)op->d_meta.ensure((size_t)kMetaVecs * 8 + (size_t)kMetaVecs * 256 * 2);
}
// samples a variant row is padded to (the digit panels and partial buffers are that long)
static inline int64_t n_padded(const bsn_bed *b) { return b->bits == 8 ? b->pitch : b->pitch * 4; }

static void quantise(bsn_op *op, const double *d_X, int64_t ldx, int64_t len, int64_t len_pad,
                     int nvec, int mode, int S, int ncol, int permute, int exact_int,
                     VecMeta *meta, int8_t *q, const double *d_W2 = nullptr) {
  hipStream_t st = op->bed->stream;
  const bool bytes = op->bed->bits == 8;
  const double vstep = bytes ? op->bed->v_step : 1.0, voff = bytes ? op->bed->v_off : 0.0;
  int raw3 = (!bytes && mode == 1) ? 1 : 0;
#ifdef BSN_ABLATION
  // energy experiments of round 5 (profiles/r05_power.txt): BSN_DIGITS=1 slice-major digit columns, =2 sign x 7-bit
  // magnitude digits.  The streaming kernels run on these operands at their real cost; the RESULTS are wrong (the
  // finalize kernels read vector-major balanced digits), so this exists in the profiling build only.
  if (const char *dg = getenv("BSN_DIGITS")) raw3 |= atoi(dg) << 8;
#endif
  if (bytes) permute = 0;  // the byte image holds the samples of a 16-block in natural order
  // meta: kMetaVecs records, then the slice maxima of k_absmax (kMetaVecs x 256 x 2 doubles)
  double *part = (double *)(meta + kMetaVecs);
  int gx = 0;
  if (exact_int) {
    hipLaunchKernelGGL(k_meta_clear, dim3(1), dim3(64), 0, st, meta, nvec);
  } else {
    gx = (int)((len + 1023) / 1024);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(k_absmax, dim3(gx, nvec), dim3(1024), 0, st, d_X, ldx, len,
                       mode == 1 ? op->d_center.p : mode == 2 ? d_W2 : nullptr,
                       mode == 1 ? op->d_scale.p : nullptr, mode, meta, part, vstep, voff, raw3);
  }
  // (no memset of q: k_quant writes every digit row of the columns in use, zeros past `len`; the columns
  // of a column block that no vector uses keep stale bytes, which only reach accumulator columns that no
  // finalize kernel reads)
  int64_t nblk = len_pad / 16;
  const double *qc = mode == 1 ? op->d_center.p : mode == 2 ? d_W2 : nullptr;
  const double *qsc = mode == 1 ? op->d_scale.p : nullptr;
  const dim3 qgrid((unsigned)((nblk + 63) / 64), nvec);
#define BSN_QUANT(SV)                                                                                  \
  case SV:                                                                                             \
    if (permute)                                                                                       \
      hipLaunchKernelGGL((k_quant<SV, 1>), qgrid, dim3(64), 0, st, d_X, ldx, len, len_pad, qc, qsc, mode, \
                         ncol, meta, q, vstep, voff, raw3, part, gx);                                  \
    else                                                                                               \
      hipLaunchKernelGGL((k_quant<SV, 0>), qgrid, dim3(64), 0, st, d_X, ldx, len, len_pad, qc, qsc, mode, \
                         ncol, meta, q, vstep, voff, raw3, part, gx);                                  \
    break;
  switch (S) {
    BSN_QUANT(1) BSN_QUANT(2) BSN_QUANT(3) BSN_QUANT(4) BSN_QUANT(5) BSN_QUANT(6) BSN_QUANT(7) BSN_QUANT(8)
    default: fail("slices must be in 1..8");
  }
#undef BSN_QUANT
  BSN_HIP(hipGetLastError());
}

static const double *scatter_rows_if_needed(bsn_op *op, const double *d_X, int64_t *ldx,
                                            int nvec) {
  if (op->rows_identity) return d_X;
  bsn_bed *b = op->bed;
  double *xf = op->d_xfull.ensure((size_t)b->n * nvec);
  BSN_HIP(hipMemsetAsync(xf, 0, (size_t)b->n * nvec * sizeof(double), b->stream));
  hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)((op->n + 255) / 256), nvec), dim3(256), 0,
                     b->stream, d_X, *ldx, op->d_rows.p, op->n, xf, b->n);
  BSN_HIP(hipGetLastError());
  *ldx = b->n;
  return xf;
}

// Ablation builds only (-DBSN_ABLATION, tools/build_ablation.py): BSN_TUNE selects profiling
// variants of the two streaming kernels that compute wrong numbers by construction.  The product
// library contains none of them.
#ifdef BSN_ABLATION
static int tune_variant() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("BSN_TUNE");
    v = e ? atoi(e) : 0;
  }
  return v;
}
#endif

// binomial scaling from the code counts (R/binom-scaling.R:133-142 on the sums of
// src/bed-fun.cpp:22-38: sumX = n1 + 2 n2, nb_nona = n - nNA), same operations in the same order
#pragma clang fp contract(off)
__global__ void k_binom_scale(const int32_t *counts, int64_t m, double *center, double *scale) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const int4 c = *(const int4 *)(counts + 4 * j);
  const double sumX = (double)(c.y + 2 * c.z);
  const double nona = (double)(c.x + c.y + c.z);
  const double af = sumX / (2.0 * nona);
  center[j] = 2.0 * af;
  scale[j] = sqrt(2.0 * af * (1.0 - af));
}
#pragma clang fp contract(on)

// by-products of the counting pass, all on the device: the total number of missing genotypes (decides
// whether the solve may drop the missing-value plane), the number of variants with more than 50 % missing
// (the warning of bed_colstats, src/bed-fun.cpp:40-41) and the per-variant missing counts the handle keeps.
// Integer atomics: exact and order-independent.  out[0] = total, out[1] = n_bad (both cleared by the caller).
__global__ __launch_bounds__(256) void k_stats_summary(const int32_t *counts, int64_t m, int64_t n, int32_t *na,
                                                       unsigned long long *out) {
  long long s = 0, bad = 0;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < m; j += (int64_t)gridDim.x * 256) {
    const int4 c = *(const int4 *)(counts + 4 * j);
    s += c.w;
    na[j] = c.w;
    if (2 * ((int64_t)c.x + c.y + c.z) < n) bad++;
  }
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off);
    bad += __shfl_down(bad, off);
  }
  __shared__ long long sw[4], sb[4];
  if ((threadIdx.x & 63) == 0) {
    sw[threadIdx.x >> 6] = s;
    sb[threadIdx.x >> 6] = bad;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s = sw[0] + sw[1] + sw[2] + sw[3];
    bad = sb[0] + sb[1] + sb[2] + sb[3];
    if (s) atomicAdd(&out[0], (unsigned long long)s);
    if (bad) atomicAdd(&out[1], (unsigned long long)bad);
  }
}

// the streaming-layout copy serves an operator over a 64-aligned contiguous range of variants
static bool use_tiled(const bsn_op *op) {
#ifdef BSN_ABLATION
  if (tune_variant() != 0) return false;   // the profiling variants exist on the plain image only
#endif
  return op->bed->d_tiled != nullptr && op->cols_contig && (op->col0 & 63) == 0;
}

template <int NPLANE, bool RAW0, bool STATS>
static void launch_cprod(bsn_op *op, int NB, const int8_t *q, int32_t *acc, uint32_t l0,
                         uint32_t l1, uint32_t l2, int32_t *counts = nullptr) {
  bsn_bed *b = op->bed;
  constexpr int KC = 512;
  // int32 accumulators over the whole sample range: a plane adds at most 4 * 128 per sample
  if (b->pitch * 4 > 4000000) fail("more than 4e6 samples are not supported by the crossproduct kernel");
  const int32_t *cols = op->cols_contig ? nullptr : op->d_cols.p;
  const int32_t npad = (int32_t)(b->pitch * 4 - b->n);
  const bool warm = op->prof_kind_override == 3;   // warm-start launches run under their own kernel name (TAG = 1)
  // k_cprod<NB, NPLANE, KC, RAW0, STATS, CONTIG, ABL, TILES, WAVES, MINW, TAG, TILED, SGB> on `image`
#define BSN_CPROD(NBV, CONTIGV, ABLV, TV, WV, TAGV, TILEDV, SGBV, image)                                              \
  BSN_KLAUNCH((k_cprod<NBV, NPLANE, KC, RAW0, STATS, CONTIGV, ABLV, TV, WV, 1, TAGV, TILEDV, SGBV>),                  \
              dim3((unsigned)((op->m + 16 * TV * WV - 1) / (16 * TV * WV))), dim3(64 * WV), 0, b->stream, image,      \
              b->pitch, cols, op->col0, op->m, q, acc, op->m, l0, l1, l2, counts, npad)
  // Shapes (profiles/r02_ablation.txt, r03_shape_sweeps.txt, r04_two_block_kernels.txt): one column block — 8 waves x 2
  // tiles on the plain image, 8 x 4 on the tiled copy (2 % faster there; the counting variant needs 150 registers with
  // 4 tiles and keeps 2); two column blocks — 16 waves x 2 tiles share one digit panel (half the L2 reads of it), with
  // the explicit MFMA / decode interleave + raised priority through the MFMA phase (SGB = 3: 2 %).
  if constexpr (NPLANE == 2 && RAW0 && !STATS) {
    if (op->na_skip_c && NB >= 2) {   // the missing-value plane only where a K-step has a missing code (op_na_blocks)
      if (NB == 3) {
        if (op->cols_contig) BSN_KLAUNCH((k_cprod<3, 2, KC, true, false, true, 0, 2, 16, 1, 0, false, 0, true>),
                                         dim3((unsigned)((op->m + 511) / 512)), dim3(1024), 0, b->stream, b->d_img, b->pitch,
                                         cols, op->col0, op->m, q, acc, op->m, l0, l1, l2, counts, npad);
        else BSN_KLAUNCH((k_cprod<3, 2, KC, true, false, false, 0, 2, 16, 1, 0, false, 0, true>),
                         dim3((unsigned)((op->m + 511) / 512)), dim3(1024), 0, b->stream, b->d_img, b->pitch, cols, op->col0,
                         op->m, q, acc, op->m, l0, l1, l2, counts, npad);
      } else {
        if (op->cols_contig) BSN_KLAUNCH((k_cprod<2, 2, KC, true, false, true, 0, 2, 16, 1, 0, false, 0, true>),
                                         dim3((unsigned)((op->m + 511) / 512)), dim3(1024), 0, b->stream, b->d_img, b->pitch,
                                         cols, op->col0, op->m, q, acc, op->m, l0, l1, l2, counts, npad);
        else BSN_KLAUNCH((k_cprod<2, 2, KC, true, false, false, 0, 2, 16, 1, 0, false, 0, true>),
                         dim3((unsigned)((op->m + 511) / 512)), dim3(1024), 0, b->stream, b->d_img, b->pitch, cols, op->col0,
                         op->m, q, acc, op->m, l0, l1, l2, counts, npad);
      }
      BSN_HIP(hipGetLastError());
      return;
    }
  }
  if (NB == 3) {   // three column blocks: 16 waves x 2 tiles on the plain image (contiguous or gathered variants)
    if constexpr (STATS || NPLANE == 3) {
      fail("internal: no three-block counting kernel");
    } else {
#ifdef BSN_ABLATION
      // shape sweep of the three-block kernel (profiling build only; correct results): BSN_NB3 bit 0: without the explicit
      // MFMA / decode pipeline, bit 1: 8-wave workgroups (two per CU), bit 3: 4 tiles per wave in 8-wave workgroups
      static const int nb3v = getenv("BSN_NB3") ? atoi(getenv("BSN_NB3")) : 0;
      if (op->cols_contig && (nb3v & 11)) {
        if ((nb3v & 11) == 1) BSN_CPROD(3, true, 0, 2, 16, 0, false, 0, b->d_img);
        else if ((nb3v & 11) == 2) BSN_CPROD(3, true, 0, 2, 8, 0, false, 3, b->d_img);
        else if ((nb3v & 11) == 3) BSN_CPROD(3, true, 0, 2, 8, 0, false, 0, b->d_img);
        else BSN_CPROD(3, true, 0, 4, 8, 0, false, 0, b->d_img);
        BSN_HIP(hipGetLastError());
        return;
      }
#endif
      if (op->cols_contig) BSN_CPROD(3, true, 0, 2, 16, 0, false, 3, b->d_img);
      else BSN_CPROD(3, false, 0, 2, 16, 0, false, 0, b->d_img);
    }
    BSN_HIP(hipGetLastError());
    return;
  }
  if (use_tiled(op)) {
    constexpr int TV = STATS ? 2 : 4;
    if (NB == 1) { if (warm) BSN_CPROD(1, true, 0, TV, 8, 1, true, 0, b->d_tiled); else BSN_CPROD(1, true, 0, TV, 8, 0, true, 0, b->d_tiled); }
    else BSN_CPROD(2, true, 0, 2, 16, 0, true, 3, b->d_tiled);
    BSN_HIP(hipGetLastError());
    return;
  }
#ifdef BSN_ABLATION
  // BSN_TUNE (profiling variants; 11 .. 19 / 111 .. 119 compute wrong numbers by construction):
  //   11 / 12 / 13 / 15 / 16 / 17 / 18 / 19   one column block: no MFMA / no decode / neither (memory skeleton) / no
  //                                          barrier / no LDS operand reads / no genotype loads / no LDS at all / compute only
  //   111 / 112 / 113 / 117 / 119            the same for two column blocks;  114: two blocks WITHOUT the explicit pipeline
  //   121 / 123                              two blocks: pipeline alone / s_setprio alone (correct results)
  if constexpr (NPLANE == 2 && RAW0 && !STATS) {
    const int abl = tune_variant();
    if (op->cols_contig && abl != 0) {
      bool done = true;
      if (NB == 1) {
        switch (abl) {
          case 11: BSN_CPROD(1, true, 1, 2, 8, 0, false, 0, b->d_img); break;
          case 12: BSN_CPROD(1, true, 2, 2, 8, 0, false, 0, b->d_img); break;
          case 13: BSN_CPROD(1, true, 3, 2, 8, 0, false, 0, b->d_img); break;
          case 15: BSN_CPROD(1, true, 8, 2, 8, 0, false, 0, b->d_img); break;
          case 16: BSN_CPROD(1, true, 16, 2, 8, 0, false, 0, b->d_img); break;
          case 17: BSN_CPROD(1, true, 32, 2, 8, 0, false, 0, b->d_img); break;
          case 18: BSN_CPROD(1, true, 24, 2, 8, 0, false, 0, b->d_img); break;
          case 19: BSN_CPROD(1, true, 56, 2, 8, 0, false, 0, b->d_img); break;
          default: done = false;
        }
      } else {
        switch (abl) {
          case 111: BSN_CPROD(2, true, 1, 2, 16, 0, false, 0, b->d_img); break;
          case 112: BSN_CPROD(2, true, 2, 2, 16, 0, false, 0, b->d_img); break;
          case 113: BSN_CPROD(2, true, 3, 2, 16, 0, false, 0, b->d_img); break;
          case 114: BSN_CPROD(2, true, 0, 2, 16, 0, false, 0, b->d_img); break;
          case 117: BSN_CPROD(2, true, 32, 2, 16, 0, false, 0, b->d_img); break;
          case 119: BSN_CPROD(2, true, 56, 2, 16, 0, false, 0, b->d_img); break;
          case 121: BSN_CPROD(2, true, 0, 2, 16, 0, false, 1, b->d_img); break;
          case 123: BSN_CPROD(2, true, 0, 2, 16, 0, false, 2, b->d_img); break;
          default: done = false;
        }
      }
      if (done) {
        BSN_HIP(hipGetLastError());
        return;
      }
    }
  }
#endif
  if (NB == 1) {
    if (!op->cols_contig) BSN_CPROD(1, false, 0, 2, 8, 0, false, 0, b->d_img);
    else if (warm) BSN_CPROD(1, true, 0, 2, 8, 1, false, 0, b->d_img);
    else BSN_CPROD(1, true, 0, 2, 8, 0, false, 0, b->d_img);
  } else {
    if (op->cols_contig) BSN_CPROD(2, true, 0, 2, 16, 0, false, 3, b->d_img);
    else BSN_CPROD(2, false, 0, 2, 16, 0, false, 0, b->d_img);
  }
#undef BSN_CPROD
  BSN_HIP(hipGetLastError());
}

// The scaling statistics of a solve ride along its first crossproduct pass: counts of the codes
// -> centre / scale on the device, before the finalize kernel needs them.
static void finish_fused_stats(bsn_op *op) {
  bsn_bed *b = op->bed;
  hipLaunchKernelGGL(k_binom_scale, dim3((unsigned)((op->m + 255) / 256)), dim3(256), 0, b->stream,
                     op->d_counts.p, op->m, op->d_center.p, op->d_scale.p);
  BSN_HIP(hipGetLastError());
  op->stats_pending = false;
  // missing-value total (and the > 50 % missing count) -> pinned host words, picked up by op_poll_stats at the
  // caller's next sync
  if (!op->h_na_total) BSN_HIP(hipHostMalloc((void **)&op->h_na_total, 2 * sizeof(long long), hipHostMallocDefault));
  op->h_na_total[0] = -1;
  op->h_na_total[1] = -1;
  unsigned long long *d_tot = (unsigned long long *)(op->d_counts.p + 4 * op->m);  // four spare words behind the counts
  BSN_HIP(hipMemsetAsync(d_tot, 0, 16, b->stream));
  int gx = (int)((op->m + 1023) / 1024);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(k_stats_summary, dim3(gx), dim3(256), 0, b->stream, op->d_counts.p, op->m, op->n,
                     op->d_na.ensure((size_t)op->m), d_tot);
  BSN_HIP(hipGetLastError());
  BSN_HIP(hipMemcpyAsync(op->h_na_total, d_tot, 16, hipMemcpyDeviceToHost, b->stream));
  op->na_poll = true;
}

void op_poll_stats(bsn_op *op) {
  if (!op->na_poll || !op->h_na_total) return;
  const long long t = *(volatile long long *)op->h_na_total;
  if (t < 0) return;
  op->na_poll = false;
  if (t == 0 && !getenv("BSN_FORCE_NA_PLANE")) op->no_na = true;  // complete data: skip the missing-value plane from now on
}

// ---------------------------------------------------------------------------
// How many K-steps of the streaming products carry no missing code?  (round 5, VERDICT r4 #5b)  The missing-value plane
// is half the MFMAs of k_cprod / k_prodT; a K-step whose 1 024 genotypes hold no missing code adds zeros through it.
// At 1 % scattered missing values no step is free (0.99^1024); on nearly complete data (1e-4: 90 %) or on data whose
// missing values come in batches (arrays merged over sample sets) most are.  k_na_blocks tests a SAMPLE of steps with
// exactly the kernels' test, in the kernels' two step shapes, on the variant-major image:
//   crossproduct  wave = 16 variants (lane & 15) x one 64-byte segment (4 x 16 B, lane >> 4): dword d of every lane is
//                 K-step d — four steps per wave
//   product       wave = one dword column (16 samples) x 256 variants of one half chunk: lane (c, g), step d reads
//                 variant j0 + 64 g + 16 d + c — four steps per wave
// out: {free, sampled} steps of the two shapes, then 1 (the arrival flag of the pinned copy).
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__global__ __launch_bounds__(64) void k_na_blocks(const uint8_t *__restrict__ img, int64_t pitch, int64_t m,
                                                  unsigned long long *out) {
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  const uint32_t h = hash32((uint32_t)blockIdx.x * 0x9E3779B1u + 0x7F4A7C15u), h2 = hash32(h ^ 0x85EBCA6Bu);
  const int64_t nseg = pitch / 64, ngrp = m / 16, nhalf = m / 256, ndw = pitch / 4;
  int free_c = 0, free_p = 0, tot_c = 0, tot_p = 0;
  if (ngrp > 0 && nseg > 0) {
    const int64_t grp = (int64_t)(((uint64_t)h * (uint64_t)ngrp) >> 32), seg = (int64_t)(((uint64_t)h2 * (uint64_t)nseg) >> 32);
    const uint4 w = *(const uint4 *)(img + (grp * 16 + c) * pitch + seg * 64 + g * 16);
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int d = 0; d < 4; d++)
      free_c += __builtin_amdgcn_ballot_w64(((ws[d] & (ws[d] >> 1)) & 0x55555555u) != 0u) == 0ull;
    tot_c = 4;
  }
  if (nhalf > 0 && ndw > 0) {
    const int64_t hc = (int64_t)(((uint64_t)h2 * (uint64_t)nhalf) >> 32), dw = (int64_t)(((uint64_t)h * (uint64_t)ndw) >> 32);
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const uint32_t w = *(const uint32_t *)(img + (hc * 256 + 64 * g + 16 * d + c) * pitch + dw * 4);
      free_p += __builtin_amdgcn_ballot_w64(((w & (w >> 1)) & 0x55555555u) != 0u) == 0ull;
    }
    tot_p = 4;
  }
  if (lane == 0) {
    atomicAdd(&out[0], (unsigned long long)free_c);
    atomicAdd(&out[1], (unsigned long long)tot_c);
    atomicAdd(&out[2], (unsigned long long)free_p);
    atomicAdd(&out[3], (unsigned long long)tot_p);
  }
}

// BSN_NA_SKIP=0 / 1: never / always the skipping kernels (A/B runs and tests; results are the same integers either
// way); BSN_NA_SKIP_MIN: the share of free steps from which they are used (default 0.30: they cost 4 VALU and a
// branch per step and do without the explicit MFMA / decode schedule, profiles/r05_power.txt).
void op_na_blocks(bsn_op *op, int digit_cols) {
  bsn_bed *b = op->bed;
  op->na_skip_c = op->na_skip_p = false;
  // (one column block — the one-shot products, 8-vector solves — has no skipping kernel: nothing to measure for)
  if (b->bits != 2 || b->generic || op->no_na || !b->d_img || digit_cols <= 16) return;
  const char *fe = getenv("BSN_NA_SKIP");   // (read at every call: the tests switch it between two products)
  const int force = fe ? atoi(fe) : -1;
  if (force == 0) return;
  if (force == 1) {
    op->na_skip_c = op->na_skip_p = true;
    return;
  }
  const char *me = getenv("BSN_NA_SKIP_MIN");
  const double min_free = me ? atof(me) : 0.30;
  if (b->na_blocks_state == 0) {
    if (!b->h_na_blocks) BSN_HIP(hipHostMalloc((void **)&b->h_na_blocks, 5 * sizeof(long long), hipHostMallocDefault));
    if (!b->d_na_blocks) BSN_HIP(hipMalloc((void **)&b->d_na_blocks, 5 * sizeof(long long)));
    for (int i = 0; i < 5; i++) b->h_na_blocks[i] = -1;   // [4] stays -1 until the second copy below has landed
    BSN_HIP(hipMemsetAsync(b->d_na_blocks, 0, 5 * sizeof(long long), b->stream));
    constexpr int kSamples = 8192;   // waves = 32 768 steps of each shape: the share to about half a percent
    hipLaunchKernelGGL(k_na_blocks, dim3(kSamples), dim3(64), 0, b->stream, b->d_img, b->pitch, b->m,
                       (unsigned long long *)b->d_na_blocks);
    BSN_HIP(hipGetLastError());
    // the counts first, the flag word (one byte set to 1) in a copy of its own behind them: it arrives last
    BSN_HIP(hipMemsetAsync((uint8_t *)b->d_na_blocks + 4 * sizeof(long long), 1, 1, b->stream));
    BSN_HIP(hipMemcpyAsync(b->h_na_blocks, b->d_na_blocks, 4 * sizeof(long long), hipMemcpyDeviceToHost, b->stream));
    BSN_HIP(hipMemcpyAsync(b->h_na_blocks + 4, b->d_na_blocks + 4, sizeof(long long), hipMemcpyDeviceToHost, b->stream));
    b->na_blocks_state = 1;
    return;
  }
  if (b->na_blocks_state == 1) {
    if (*(volatile long long *)(b->h_na_blocks + 4) != 1) return;   // not there yet: the plain kernels meanwhile
    const long long *v = b->h_na_blocks;
    b->na_free[0] = v[1] > 0 ? (double)v[0] / (double)v[1] : 0.0;
    b->na_free[1] = v[3] > 0 ? (double)v[2] / (double)v[3] : 0.0;
    b->na_blocks_state = 2;
  }
  op->na_skip_c = b->na_free[0] >= min_free;
  op->na_skip_p = b->na_free[1] >= min_free;
}

// vectors per crossproduct launch: three column blocks on a 2-bit image unless the launch also counts the codes
static int cprod_vmax(const bsn_op *op, int S) {
  const int nbmax = (op->bed->bits == 2 && !op->stats_pending && nb3_allowed()) ? 3 : 2;
  int v = 16 * nbmax / S;
  if (v > kMetaVecs) v = kMetaVecs;
  return v < 1 ? 1 : v;
}

void op_cprod_prequant(bsn_op *op, const double *d_X, int64_t ldx, int nvec) {
  bsn_bed *b = op->bed;
  const int S = op->slices;
  op->preq_X = nullptr;
  if (nvec <= 0 || nvec > cprod_vmax(op, S) || !op->rows_identity) return;
  const int64_t npad = n_padded(b);
  quantise(op, d_X, ldx, b->n, npad, nvec, 0, S, 16 * pick_nb(nvec * S), 1, 0, meta_buffer(op),
           op->d_q.ensure((size_t)npad * kMaxCols * 2));
  op->preq_X = d_X;
  op->preq_ldx = ldx;
  op->preq_nvec = nvec;
  op->preq_S = S;
}

void op_cprod(bsn_op *op, const double *d_X, int64_t ldx, int nvec, double *d_Z, int64_t ldz) {
  bsn_bed *b = op->bed;
  refuse_generic(b, "this function (it needs the streaming products)");
  const int S = op->slices;
  if (nvec <= 0) return;
  op_na_blocks(op, nvec * S);
  const bool have_digits = op->preq_X == d_X && op->preq_ldx == ldx && op->preq_nvec == nvec && d_X != nullptr &&
                           op->preq_S == S && nvec <= cprod_vmax(op, S);
  op->preq_X = nullptr;
  const double *xsrc = scatter_rows_if_needed(op, d_X, &ldx, nvec);
  const int64_t npad = n_padded(b);
  VecMeta *meta = meta_buffer(op);
  for (int v0 = 0, vmax = 0; v0 < nvec; v0 += vmax) {
    vmax = cprod_vmax(op, S);  // vectors of this launch (the launch that counts the codes is limited to two column blocks)
    int nv = nvec - v0 < vmax ? nvec - v0 : vmax;
    int NB = pick_nb(nv * S), ncol = 16 * NB;
    int8_t *q = op->d_q.ensure((size_t)npad * kMaxCols * 2);
    int32_t *acc = op->d_acc.ensure((size_t)2 * op->m * kMaxCols);
    if (!have_digits) quantise(op, xsrc + (int64_t)v0 * ldx, ldx, b->n, npad, nv, 0, S, ncol, 1, 0, meta, q);
    prof_begin(op, op->stats_pending ? 2 : NB == 3 ? 4 : 0);  // the pass that carries the code counts is timed apart
    if (b->bits == 8) {
      const dim3 grid8((unsigned)((op->m + 127) / 128));
      const int32_t *cols8 = op->cols_contig ? nullptr : op->d_cols.p;
#define BSN_CPROD8(NBV, NAV, CV)                                                                          \
  BSN_KLAUNCH((k_cprod8<NBV, NAV, CV>), grid8, dim3(512), 0, b->stream, b->d_img, b->pitch, cols8, \
                     op->col0, op->m, q, acc, op->m)
      if (NB == 1) {
        if (op->no_na) { if (op->cols_contig) BSN_CPROD8(1, false, true); else BSN_CPROD8(1, false, false); }
        else { if (op->cols_contig) BSN_CPROD8(1, true, true); else BSN_CPROD8(1, true, false); }
      } else {
        if (op->no_na) { if (op->cols_contig) BSN_CPROD8(2, false, true); else BSN_CPROD8(2, false, false); }
        else { if (op->cols_contig) BSN_CPROD8(2, true, true); else BSN_CPROD8(2, true, false); }
      }
#undef BSN_CPROD8
      BSN_HIP(hipGetLastError());
      prof_end(op);
      op->passes++;
      hipLaunchKernelGGL(k_cprod_final8, dim3((unsigned)((op->m + 255) / 256), nv), dim3(256), 0, b->stream, acc,
                         op->m, ncol, S, meta, op->d_center.p, op->d_scale.p, d_Z + (int64_t)v0 * ldz, ldz,
                         op->no_na ? 0 : 1, b->v_step, b->v_off);
      BSN_HIP(hipGetLastError());
      continue;
    }
    if (op->stats_pending) {
      if (!op->rows_identity) fail("internal: fused scaling statistics need all samples");
      launch_cprod<2, true, true>(op, NB, q, acc, kLutRaw, kLutNA, 0, op->d_counts.ensure((size_t)4 * op->m + 8));
    } else if (op->no_na) {
      launch_cprod<1, true, false>(op, NB, q, acc, kLutRaw, 0, 0);  // complete variants: code == genotype
    } else {
      launch_cprod<2, true, false>(op, NB, q, acc, kLutRaw, kLutNA, 0);
    }
    prof_end(op);
    op->passes++;
    if (op->late_center || op->late_scale) {
      // staged and sent while the streaming kernel runs; the finalize kernel waits for them
      hipStream_t up = upload_stream(b);
      if (op->late_center) copy_h2d(b, op->d_center.p, op->late_center, (size_t)op->m * 8, up);
      if (op->late_scale) copy_h2d(b, op->d_scale.p, op->late_scale, (size_t)op->m * 8, up);
      BSN_HIP(hipEventRecord(b->ev_up, up));
      BSN_HIP(hipStreamWaitEvent(b->stream, b->ev_up, 0));
      op->late_center = op->late_scale = nullptr;
    }
    const bool has_q = op->stats_pending || !op->no_na;
    if (op->stats_pending) finish_fused_stats(op);
    if (NB == 1)
      hipLaunchKernelGGL((k_cprod_final<16>), dim3((unsigned)((op->m + 127) / 128)), dim3(128), 0, b->stream, acc, op->m,
                         S, nv, meta, op->d_center.p, op->d_scale.p, d_Z + (int64_t)v0 * ldz, ldz, has_q ? 1 : 0);
    else if (NB == 2)
      hipLaunchKernelGGL((k_cprod_final<32>), dim3((unsigned)((op->m + 127) / 128)), dim3(128), 0, b->stream, acc, op->m,
                         S, nv, meta, op->d_center.p, op->d_scale.p, d_Z + (int64_t)v0 * ldz, ldz, has_q ? 1 : 0);
    else
      hipLaunchKernelGGL((k_cprod_final<48>), dim3((unsigned)((op->m + 127) / 128)), dim3(128), 0, b->stream, acc, op->m,
                         S, nv, meta, op->d_center.p, op->d_scale.p, d_Z + (int64_t)v0 * ldz, ldz, has_q ? 1 : 0);
    BSN_HIP(hipGetLastError());
  }
}

template <int NB, bool CONTIG>
static void launch_prod(bsn_op *op, dim3 grid, int64_t m_pad, int64_t mc, const int8_t *q, int32_t *acc,
                        int64_t npad, uint32_t lutP, uint32_t lutQ, bool has_q) {
  bsn_bed *b = op->bed;
  const int32_t *cols = op->d_cols.p;
  const bool warm = op->prof_kind_override == 3;   // warm-start launches run under their own kernel name (TAG = 1)
  // k_prod<NB, CONTIG, RAWP, HASQ, ABL, TAG, TILED> on `image`
#define BSN_PROD(RAWP, HASQ, ABLV, TAGV, TILEDV, image)                                                        \
  BSN_KLAUNCH((k_prod<NB, CONTIG, RAWP, HASQ, ABLV, TAGV, TILEDV>), grid, dim3(256), 0, b->stream, image,      \
              b->pitch, cols, op->col0, m_pad, mc, q, acc, npad, lutP, lutQ)
#ifdef BSN_ABLATION
  // BSN_TUNE = 61 .. 64 (one column block) / 161 .. 164 (two): no MFMA / no decode / memory skeleton / compute only
  if constexpr (CONTIG) {
    const int tv = tune_variant() - (NB == 2 ? 100 : 0);
    if (tv >= 61 && tv <= 64 && lutP == kLutRaw && has_q) {
      if (tv == 61) BSN_PROD(true, true, 1, 0, false, b->d_img);
      else if (tv == 62) BSN_PROD(true, true, 2, 0, false, b->d_img);
      else if (tv == 63) BSN_PROD(true, true, 3, 0, false, b->d_img);
      else BSN_PROD(true, true, 32, 0, false, b->d_img);
      BSN_HIP(hipGetLastError());
      return;
    }
  }
#endif
  if constexpr (CONTIG) {
    if (use_tiled(op)) {  // streaming-layout copy: same arithmetic, contiguous 16-KB steps
      if (lutP == kLutRaw) {
        if (has_q) { if (warm) BSN_PROD(true, true, 0, 1, true, b->d_tiled); else BSN_PROD(true, true, 0, 0, true, b->d_tiled); }
        else { if (warm) BSN_PROD(true, false, 0, 1, true, b->d_tiled); else BSN_PROD(true, false, 0, 0, true, b->d_tiled); }
      } else {
        if (has_q) BSN_PROD(false, true, 0, 0, true, b->d_tiled);
        else BSN_PROD(false, false, 0, 0, true, b->d_tiled);
      }
      BSN_HIP(hipGetLastError());
      return;
    }
  }
  if (lutP == kLutRaw) {
    if (has_q) { if (warm) BSN_PROD(true, true, 0, 1, false, b->d_img); else BSN_PROD(true, true, 0, 0, false, b->d_img); }
    else { if (warm) BSN_PROD(true, false, 0, 1, false, b->d_img); else BSN_PROD(true, false, 0, 0, false, b->d_img); }
  } else {
    if (has_q) BSN_PROD(false, true, 0, 0, false, b->d_img);
    else BSN_PROD(false, false, 0, 0, false, b->d_img);
  }
#undef BSN_PROD
  BSN_HIP(hipGetLastError());
}

// Y (+)= sum_j P[code_ij] W1[j, v] + sum_j lutQ[code_ij] W2[j, v]  (- sum_j W2[j, v] if sub_const)
// P = lutP look-up, or the code itself when lutP == kLutRaw.
// mode 1: W1 = X / scale, W2 = center * W1 derived from the operator's centre / scale (the
// scaled product A~ X; lutP must be kLutRaw and lutQ kLutNA: the quantiser then stores W2 - 3 W1
// in the second plane, see k_quant); mode 2: W1 = d_X, W2 = d_W2 given directly (raw plane weights).
struct ProdSegments {   // op_prod_segments
  int pieces, stride, nseg;
  const ProdSegment *segs;
  const std::function<void(int)> *after;
  bool done = false;
};
static void prod_planes(bsn_op *op, const double *d_X, const double *d_W2, int64_t ldx, int nvec,
                        double *d_Y, int64_t ldy, int mode, uint32_t lutP, uint32_t lutQ, int sub_const,
                        double beta, int S, ProdSegments *sg = nullptr) {
  bsn_bed *b = op->bed;
  refuse_generic(b, "this function (it needs the streaming products)");
  if (nvec <= 0) return;
  op->preq_X = nullptr;
  op_na_blocks(op, nvec * S);
  // k_prod addresses a 64-variant step with 32-bit offsets from its first row
  if (b->pitch >= ((int64_t)1 << 24)) fail("more than 6.7e7 samples are not supported by the product kernel");
  const int64_t npad = n_padded(b);
  // Two or three column blocks over a contiguous range of variants that starts on a 512-variant chunk, and the handle
  // has its sample-major copy: the product runs as k_prodT (k_cprod's shape, contraction over the contiguous index).
  if (b->smaj_job) (void)image_smaj_poll(b);   // a copy being made beside this solve: has it arrived?
  const bool smaj_ok = b->bits == 2 && b->d_smaj != nullptr && op->cols_contig && (op->col0 & 511) == 0 && mode == 1 &&
                       lutP == kLutRaw && lutQ == kLutNA && !getenv("BSN_NO_SMAJ");
  const int vmax_smaj = std::min(kMetaVecs, (smaj_ok && nb3_allowed() && nvec * S > 32 ? kMaxCols : 32) / S);
  const bool smaj = smaj_ok && nvec <= vmax_smaj && pick_nb(nvec * S) >= 2;   // (ONE launch: the geometry below is k_prodT's)
  // (a panel that needs several launches stays on k_prod, which has two column blocks at most: 16 vectors x 56 bits with
  // the copy in place used to cut itself into launches of three — "three column blocks without the sample-major copy")
  const int vmax = smaj ? vmax_smaj : std::max(1, std::min(kMetaVecs, 32 / S));
  if (sg && !smaj) return;   // (nothing queued: the caller takes the plain pass)
  const int64_t m_pad = round_up(op->m, smaj ? 512 : 64);
  VecMeta *meta = meta_buffer(op);
  // K split so that the grid has a few thousand workgroups
  int64_t wgx = b->bits == 8 ? npad / 256 : npad / 1024;  // workgroups along the samples
  int ky = (int)((4096 + wgx - 1) / wgx);
  int64_t steps = m_pad / 64;
  int smaj_cps = 0;   // k_prodT: chunks of 512 variants per slab
  if (smaj) {
    // one 1024-thread workgroup per CU is resident: among 6 .. 24 slabs the split whose grid fills whole rounds of
    // the chip best (782 x 17 workgroups = 51.9 rounds of 256 at 400 000 samples); a slab stays below 2.5e6 variants
    static const int ncu = [] {
      hipDeviceProp_t pr;
      int dev = 0;
      return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
                 ? pr.multiProcessorCount : 256;
    }();
    wgx = (b->n + 511) / 512;
    const int64_t nchunks = m_pad / 512;
    // ... less what the slabs cost: every slab writes its own n x 16 NB int32 partial sums and the finalize kernel reads them
    // back — 0.15 % of the image's bytes per slab at 400K x 1M, 1.2 % on the 125 000-variant shard of an 8-GPU run, where
    // 18 slabs (the best fill) measured 3.93 ms per pass against 3.80 - 3.82 with 8 - 10 (round 6, profiles/r06_shard_slabs.txt:
    // the traffic weighs about 0.3 of its bytes — the writes drain beside the stream); the full-size choice (17) is unchanged
    const double slab_cost = 0.3 * 2.0 * (double)npad * 16.0 * pick_nb((nvec < vmax ? nvec : vmax) * S) * 4.0 /
                             ((double)m_pad * (double)(b->pitch));
    int best = 1;
    double best_score = -1e300;
    for (int c = 1; c <= 24 && c <= nchunks; c++) {
      if (c < 6 && c < nchunks && nchunks >= 6) continue;
      const int64_t W = wgx * c;
      const double fill = (double)W / ((double)ncu * (double)((W + ncu - 1) / ncu));
      const double score = fill - slab_cost * c;
      if (score > best_score + 1e-9) best_score = score, best = c;
    }
    ky = best;
#ifdef BSN_ABLATION
    if (const char *e = getenv("BSN_KY_T")) ky = std::max(1, std::min(atoi(e), (int)nchunks));  // slab sweep of k_prodT (correct results)
#endif
    const int64_t ky_min2 = (m_pad + 2499999) / 2500000;
    if (ky < ky_min2) ky = (int)ky_min2;
    smaj_cps = (int)((nchunks + ky - 1) / ky);
    ky = (int)((nchunks + smaj_cps - 1) / smaj_cps);
  }
  if (!smaj) {   // (k_prodT's slab count and chunks per slab were fixed together above: clamping one would drop chunks)
    if (ky > steps) ky = (int)steps;
    if (ky > 64) ky = 64;
  }
  if (ky < 1) ky = 1;
  if (smaj && (int64_t)ky * smaj_cps < m_pad / 512) fail("internal: the slabs of k_prodT do not cover the variants");
  if (b->bits == 2 && pick_nb((nvec < vmax ? nvec : vmax) * S) == 1) {
    // (one column block: the kernel is bound by HBM; with two it is bound by instruction issue, more slabs only help
    // there — 400 000 x 125 000, 16 vectors: 4 slabs 3.45 ms, 5: 3.19, 9: 3.13, 11: 3.12 — and the rule above stays)
    // ... but every slab writes (and the finalize kernel reads back) its own n x 16 NB accumulators: on a matrix with
    // few samples per variant that is real traffic (50 000 x 200 000: 64 slabs = 16 % of the image), so the split is
    // capped at 4 % of the image bytes; and two workgroups per CU are resident, so among the splits left the one
    // whose grid fills whole rounds of 512 best wins (same matrix: 10 slabs = 490 workgroups, 0.72 - 0.75 ms per
    // call against 0.80 - 0.84 with 64 and 0.86 with 11 = 539; profiles/r03_c2_grid_sweep.txt)
    const int ncol_max = 16 * pick_nb((nvec < vmax ? nvec : vmax) * S);
    int64_t cap = (int64_t)(0.04 * (double)m_pad / (32.0 * ncol_max));
    if (cap < 1) cap = 1;
    if (ky > cap) ky = (int)cap;
    int best = ky;
    double best_fill = 0.0;
    for (int c = ky; c >= 1 && 2 * c >= ky; c--) {
      const int64_t W = wgx * c;
      const double fill = (double)W / (512.0 * (double)((W + 511) / 512));
      if (fill > best_fill + 1e-9) best_fill = fill, best = c;
    }
    ky = best;
  }
#ifdef BSN_ABLATION
  if (!smaj) {
    if (const char *e = getenv("BSN_KY")) ky = atoi(e);  // grid-shape sweep (correct results)
    if (ky > steps) ky = (int)steps;
    if (ky < 1) ky = 1;
  }
#endif
  // int32 accumulators: a slab adds at most 768 per variant (planes up to 4, digits up to 128)
  const int64_t ky_min = (m_pad + 2499999) / 2500000;
  if (ky < ky_min && !smaj) ky = (int)ky_min;
  int64_t mc = round_up((steps + ky - 1) / ky, 1) * 64;
  if (!smaj) ky = (int)((m_pad + mc - 1) / mc);
  // complete variants: the missing-value plane is all zero, skip its look-ups and MFMAs
  const bool has_q = lutQ != 0u && !(op->no_na && lutQ == kLutNA);
  for (int v0 = 0; v0 < nvec; v0 += vmax) {
    int nv = nvec - v0 < vmax ? nvec - v0 : vmax;
    int NB = pick_nb(nv * S), ncol = 16 * NB;
    int8_t *q = op->d_q.ensure((size_t)(npad > m_pad ? npad : m_pad) * kMaxCols * 2);
    size_t acc_need = (size_t)ky * npad * ncol;
    if (acc_need < (size_t)2 * op->m * kMaxCols) acc_need = (size_t)2 * op->m * kMaxCols;
    int32_t *acc = op->d_acc.ensure(acc_need);
    // (k_prodT decodes like k_cprod: its digit rows take the crossproduct's byte order)
    quantise(op, d_X + (int64_t)v0 * ldx, ldx, op->m, m_pad, nv, mode, S, ncol, smaj && NB >= 2 ? 1 : 0, 0, meta, q,
             d_W2 ? d_W2 + (int64_t)v0 * ldx : nullptr);
    dim3 grid((unsigned)wgx, (unsigned)ky);
    prof_begin(op, NB == 3 ? 5 : 1);
    if (smaj && NB >= 2) {
      const int nchunks = (int)(m_pad / 512);
      const bool warm = op->prof_kind_override == 3;
#define BSN_PRODT_(NBV, HASQV, TAGV, GRID, BS, STRIDE, OFF)                                                          \
  BSN_KLAUNCH((k_prodT<NBV, HASQV, 2, 16, TAGV>), GRID, dim3(1024), 0, b->stream, b->d_smaj, b->rows_smaj, op->col0 / 512, \
              nchunks, smaj_cps, q, acc, npad, lutQ, BS, STRIDE, OFF)
#ifdef BSN_ABLATION
      static const int nb3p = getenv("BSN_NB3") ? atoi(getenv("BSN_NB3")) : 0;   // bit 2: k_prodT<3> without the explicit pipeline
#define BSN_PRODT3(HASQV, GRID, BS, STRIDE, OFF)                                                                      \
  do {                                                                                                               \
    if (nb3p & 4)                                                                                                    \
      BSN_KLAUNCH((k_prodT<3, HASQV, 2, 16, 0, 0>), GRID, dim3(1024), 0, b->stream, b->d_smaj, b->rows_smaj, op->col0 / 512, \
                  nchunks, smaj_cps, q, acc, npad, lutQ, BS, STRIDE, OFF);                                           \
    else BSN_PRODT_(3, HASQV, 0, GRID, BS, STRIDE, OFF);                                                             \
  } while (0)
#else
#define BSN_PRODT3(HASQV, GRID, BS, STRIDE, OFF) BSN_PRODT_(3, HASQV, 0, GRID, BS, STRIDE, OFF)
#endif
#define BSN_PRODT_SKIP(NBV, GRID, BS, STRIDE, OFF)                                                                   \
  BSN_KLAUNCH((k_prodT<NBV, true, 2, 16, 0, 0, true>), GRID, dim3(1024), 0, b->stream, b->d_smaj, b->rows_smaj,       \
              op->col0 / 512, nchunks, smaj_cps, q, acc, npad, lutQ, BS, STRIDE, OFF)
#define BSN_PRODT(HASQV, TAGV, GRID, BS, STRIDE, OFF)                                                                \
  do {                                                                                                               \
    if (HASQV && op->na_skip_p) {   /* the missing-value plane only where a K-step has a missing code */             \
      if (NB == 2) BSN_PRODT_SKIP(2, GRID, BS, STRIDE, OFF);                                                         \
      else BSN_PRODT_SKIP(3, GRID, BS, STRIDE, OFF);                                                                 \
    } else if (NB == 2) BSN_PRODT_(2, HASQV, TAGV, GRID, BS, STRIDE, OFF);                                           \
    else BSN_PRODT3(HASQV, GRID, BS, STRIDE, OFF);                                                                   \
  } while (0)
      // (workgroup shapes 4 x 8 / 4 x 4 / 2 x 8 / 4 x 16 tiles x waves, chunks of 256 variants: all slower, profiles/r04_sample_major.txt)
      if (sg) {
        // the pass in segments of sample blocks: kernel + finalize of a segment, then the caller's hook (svd.hip queues the
        // segment's reduce-scatter on its second stream) while the next segment's kernel is queued behind on this one
        for (int sidx = 0; sidx < sg->nseg; sidx++) {
          const ProdSegment &sgm = sg->segs[sidx];
          const dim3 gs((unsigned)(sg->pieces * sgm.bs), (unsigned)ky);
          if (sidx > 0) prof_begin(op, NB == 3 ? 5 : 1, true);
          if (has_q) BSN_PRODT(true, 0, gs, sgm.bs, sg->stride, sgm.off);
          else BSN_PRODT(false, 0, gs, sgm.bs, sg->stride, sgm.off);
          BSN_HIP(hipGetLastError());
          prof_end(op);
          const int64_t rows_s = (int64_t)sgm.bs * 512, tot = rows_s * sg->pieces;
          if (NB == 2)
            hipLaunchKernelGGL((k_prod_final<32>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, b->stream, acc, npad, ky,
                               S, nv, meta, sgm.d_rows, tot, sgm.d_out, (int64_t)0, sub_const, 0.0, rows_s);
          else
            hipLaunchKernelGGL((k_prod_final<48>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, b->stream, acc, npad, ky,
                               S, nv, meta, sgm.d_rows, tot, sgm.d_out, (int64_t)0, sub_const, 0.0, rows_s);
          BSN_HIP(hipGetLastError());
          (*sg->after)(sidx);
        }
        op->passes++;
        sg->done = true;
        continue;
      }
      if (has_q) { if (warm) BSN_PRODT(true, 1, grid, 0, 0, 0); else BSN_PRODT(true, 0, grid, 0, 0, 0); }
      else { if (warm) BSN_PRODT(false, 1, grid, 0, 0, 0); else BSN_PRODT(false, 0, grid, 0, 0, 0); }
#undef BSN_PRODT
#undef BSN_PRODT_SKIP
#undef BSN_PRODT3
#undef BSN_PRODT_
      BSN_HIP(hipGetLastError());
    } else if (NB > 2) {
      fail("internal: three column blocks without the sample-major copy");
    } else if (b->bits == 8) {
      if (mode != 1) fail("internal: plane products are not defined on a byte image");
      const dim3 grid8 = grid;
      const int64_t npad8 = npad;
      const int32_t *cols8 = op->d_cols.p;
#define BSN_PROD8(NBV, NAV, CV)                                                                          \
  BSN_KLAUNCH((k_prod8<NBV, NAV, CV>), grid8, dim3(256), 0, b->stream, b->d_img, b->pitch, cols8, \
                     op->col0, m_pad, mc, q, acc, npad8)
      if (NB == 1) {
        if (op->no_na) { if (op->cols_contig) BSN_PROD8(1, false, true); else BSN_PROD8(1, false, false); }
        else { if (op->cols_contig) BSN_PROD8(1, true, true); else BSN_PROD8(1, true, false); }
      } else {
        if (op->no_na) { if (op->cols_contig) BSN_PROD8(2, false, true); else BSN_PROD8(2, false, false); }
        else { if (op->cols_contig) BSN_PROD8(2, true, true); else BSN_PROD8(2, true, false); }
      }
#undef BSN_PROD8
      BSN_HIP(hipGetLastError());
    } else
    if (op->cols_contig) {
      if (NB == 1) launch_prod<1, true>(op, grid, m_pad, mc, q, acc, npad, lutP, lutQ, has_q);
      else launch_prod<2, true>(op, grid, m_pad, mc, q, acc, npad, lutP, lutQ, has_q);
    } else {
      if (NB == 1) launch_prod<1, false>(op, grid, m_pad, mc, q, acc, npad, lutP, lutQ, has_q);
      else launch_prod<2, false>(op, grid, m_pad, mc, q, acc, npad, lutP, lutQ, has_q);
    }
    prof_end(op);
    op->passes++;
    if (NB == 1)
      hipLaunchKernelGGL((k_prod_final<16>), dim3((unsigned)((op->n + 255) / 256)), dim3(256), 0, b->stream,
                         acc, npad, ky, S, nv, meta, op->rows_identity ? nullptr : op->d_rows.p, op->n,
                         d_Y + (int64_t)v0 * ldy, ldy, sub_const, beta);
    else if (NB == 2)
      hipLaunchKernelGGL((k_prod_final<32>), dim3((unsigned)((op->n + 255) / 256)), dim3(256), 0, b->stream,
                         acc, npad, ky, S, nv, meta, op->rows_identity ? nullptr : op->d_rows.p, op->n,
                         d_Y + (int64_t)v0 * ldy, ldy, sub_const, beta);
    else
      hipLaunchKernelGGL((k_prod_final<48>), dim3((unsigned)((op->n + 255) / 256)), dim3(256), 0, b->stream,
                         acc, npad, ky, S, nv, meta, op->rows_identity ? nullptr : op->d_rows.p, op->n,
                         d_Y + (int64_t)v0 * ldy, ldy, sub_const, beta);
    BSN_HIP(hipGetLastError());
  }
}

// raw plane sums of the cprod kernel: P[j, v] = sum_i g0_ij x_iv, Q[j, v] = sum_i na_ij x_iv
__global__ void k_cprod_raw_final(const int32_t *acc, int64_t m, int ncol, int S, const VecMeta *meta,
                                  double *P, double *Q, int64_t ld) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int v = blockIdx.y;
  if (j >= m) return;
  const double qs = meta[v].qscale;
  const double inv = qs > 0 ? 1.0 / qs : 0.0;
  double p = horner_sub(acc + j * ncol + v * S, acc + (m + j) * ncol + v * S, 3, S) * inv,
         q = horner(acc + (m + j) * ncol + v * S, S) * inv;
  if (meta[v].nonfinite) p = q = __longlong_as_double(0x7ff8000000000000LL);
  P[j + v * ld] = p;
  Q[j + v * ld] = q;
}

void op_cprod_raw(bsn_op *op, const double *d_X, int64_t ldx, int nvec, double *d_P, double *d_Q,
                  int64_t ld) {
  require_bits(op->bed, 2, "the plane sums of multLinReg");
  op->preq_X = nullptr;
  bsn_bed *b = op->bed;
  const int S = op->slices;
  const int vmax = 32 / S;
  if (nvec <= 0) return;
  const double *xsrc = scatter_rows_if_needed(op, d_X, &ldx, nvec);
  const int64_t npad = n_padded(b);
  VecMeta *meta = meta_buffer(op);
  for (int v0 = 0; v0 < nvec; v0 += vmax) {
    int nv = nvec - v0 < vmax ? nvec - v0 : vmax;
    int NB = pick_nb(nv * S), ncol = 16 * NB;
    int8_t *q = op->d_q.ensure((size_t)npad * kMaxCols * 2);
    int32_t *acc = op->d_acc.ensure((size_t)2 * op->m * kMaxCols);
    quantise(op, xsrc + (int64_t)v0 * ldx, ldx, b->n, npad, nv, 0, S, ncol, 1, 0, meta, q);
    launch_cprod<2, true, false>(op, NB, q, acc, kLutRaw, kLutNA, 0);
    op->passes++;
    hipLaunchKernelGGL(k_cprod_raw_final, dim3((unsigned)((op->m + 255) / 256), nv), dim3(256), 0,
                       b->stream, acc, op->m, ncol, S, meta, d_P + (int64_t)v0 * ld,
                       d_Q + (int64_t)v0 * ld, ld);
    BSN_HIP(hipGetLastError());
  }
}

void op_prod(bsn_op *op, const double *d_X, int64_t ldx, int nvec, double *d_Y, int64_t ldy) {
  prod_planes(op, d_X, nullptr, ldx, nvec, d_Y, ldy, 1, kLutRaw, kLutNA, 1, 0.0, op->slices);
}

void op_prod_acc(bsn_op *op, const double *d_X, int64_t ldx, int nvec, double *d_Y, int64_t ldy, double beta) {
  prod_planes(op, d_X, nullptr, ldx, nvec, d_Y, ldy, 1, kLutRaw, kLutNA, 1, beta, op->slices);
}

bool op_prod_segments(bsn_op *op, const double *d_X, int64_t ldx, int nvec, int pieces, int stride, int nseg,
                      const ProdSegment *segs, const std::function<void(int)> &after) {
  if (!op->rows_identity || op->prof_kind_override == 3) return false;
  ProdSegments sg{pieces, stride, nseg, segs, &after};
  prod_planes(op, d_X, nullptr, ldx, nvec, nullptr, 0, 1, kLutRaw, kLutNA, 1, 0.0, op->slices, &sg);
  return sg.done;
}

// rowSumsSq[i] = sum_j A~[i, j]^2 over the non-missing genotypes (src/bed-fun.cpp:121-123):
//   ((g - c) / s)^2 = g^2 a + g b + d,  a = 1/s^2, b = -2c/s^2, d = c^2/s^2
// = X2 . a + X . b + M . d with the planes X2 = g^2, X = g, M = non-missing: two passes.
__global__ void k_rowsq_weights(const double *center, const double *scale, int64_t m, double *a,
                                double *b, double *d) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const double c = center[j], s = scale[j];
  a[j] = 1.0 / (s * s);
  b[j] = -2.0 * c / (s * s);
  d[j] = c * c / (s * s);
}

void op_row_sums_sq(bsn_op *op, double *d_out) {
  require_bits(op->bed, 2, "prod_and_rowSumsSq");
  bsn_bed *bed = op->bed;
  DevBuf<double> w;
  double *a = w.ensure((size_t)3 * op->m), *b2 = a + op->m, *d = b2 + op->m;
  hipLaunchKernelGGL(k_rowsq_weights, dim3((unsigned)((op->m + 255) / 256)), dim3(256), 0, bed->stream,
                     op->d_center.p, op->d_scale.p, op->m, a, b2, d);
  BSN_HIP(hipGetLastError());
  constexpr uint32_t kLutX2 = 0x00040100u;  // code 0,1,2,3 -> 0, 1, 4, 0
  constexpr uint32_t kLutM = 0x00010101u;   //               -> 1, 1, 1, 0
  prod_planes(op, a, b2, op->m, 1, d_out, op->n, 2, kLutX2, kLutG0, 0, 0.0, 7);
  prod_planes(op, d, nullptr, op->m, 1, d_out, op->n, 2, kLutM, 0u, 0, 1.0, 7);
  BSN_HIP(hipStreamSynchronize(bed->stream));  // `w` is released on return
}

// per-sample counts of the codes over the selected variants (bed_row_counts_cpp,
// src/bed-fun.cpp:72-99): three single-plane passes against a vector of ones — exact, the
// fixed-point image of 1.0 is a power of two and the plane sums are integers < 2^31.
// d_out: 3 x n doubles (n2, n1, nNA per selected row).
void op_row_counts(bsn_op *op, double *d_out) {
  require_bits(op->bed, 2, "bed_counts(byrow = TRUE)");
  bsn_bed *bed = op->bed;
  std::vector<double> ones((size_t)op->m, 1.0);
  DevBuf<double> w;
  copy_h2d(bed, w.ensure((size_t)op->m), ones.data(), (size_t)op->m * 8);
  prod_planes(op, w.p, nullptr, op->m, 1, d_out, op->n, 2, kLutHom2, 0u, 0, 0.0, 7);
  prod_planes(op, w.p, nullptr, op->m, 1, d_out + op->n, op->n, 2, kLutHet, 0u, 0, 0.0, 7);
  prod_planes(op, w.p, nullptr, op->m, 1, d_out + 2 * op->n, op->n, 2, kLutNA, 0u, 0, 0.0, 7);
  BSN_HIP(hipStreamSynchronize(bed->stream));  // `w` and `ones` are released on return
}

// counts of codes weighted by integer row multiplicities (general ind_row):
// three planes (hom2, het, na) against the multiplicity vector, exact.
__global__ void k_counts_final(const int32_t *acc, int64_t m, int ncol, int S, int64_t n_sub,
                               int32_t *counts) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  long long n2 = (long long)horner(acc + j * ncol, S);
  long long n1 = (long long)horner(acc + (m + j) * ncol, S);
  long long na = (long long)horner(acc + (2 * m + j) * ncol, S);
  counts[4 * j + 0] = (int32_t)(n_sub - n1 - n2 - na);
  counts[4 * j + 1] = (int32_t)n1;
  counts[4 * j + 2] = (int32_t)n2;
  counts[4 * j + 3] = (int32_t)na;
}

void counts_weighted(bsn_op *op, const double *d_w, int64_t n_sub, int32_t *d_counts) {
  require_bits(op->bed, 2, "bed_counts");
  op->preq_X = nullptr;
  bsn_bed *b = op->bed;
  const int S = 4;
  const int64_t npad = n_padded(b);
  VecMeta *meta = meta_buffer(op);
  int8_t *q = op->d_q.ensure((size_t)npad * 64);
  int32_t *acc = op->d_acc.ensure((size_t)3 * op->m * 16);
  quantise(op, d_w, b->n, b->n, npad, 1, 0, S, 16, 1, 1, meta, q);
  launch_cprod<3, false, false>(op, 1, q, acc, kLutHom2, kLutHet, kLutNA);
  hipLaunchKernelGGL(k_counts_final, dim3((unsigned)((op->m + 255) / 256)), dim3(256), 0, b->stream,
                     acc, op->m, 16, S, n_sub, d_counts);
  BSN_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------
// Instruction-level self test: v_perm_b32 source order and i8 MFMA layouts.
__global__ void k_selftest(int *out) {
  int lane = threadIdx.x;
  // perm8: selector 0..3 must pick from `lo`, 4..7 from `hi`
  uint32_t r = perm8(0x77665544u, 0x33221100u, 0x07040300u);
  int ok_perm = (r == 0x77443300u);
  // mfma: A[row=l&15][k], B[k][col=l&15], k-slots paired by (lane>>4, byte)
  // A[i][k] = (i + 1) if k == 5*i+3 (mod 64) else 0 ; B[k][j] = (k % 7) - 3 + j
  v4i a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
  int i = lane & 15, gq = lane >> 4;
  for (int e = 0; e < 16; e++) {
    int k = gq * 16 + e;
    int av = (k == (5 * i + 3) % 64) ? (i + 1) : 0;
    int bv = (k % 7) - 3 + i;  // column index of B is also lane&15
    a[e >> 2] |= (av & 0xFF) << (8 * (e & 3));
    b[e >> 2] |= (bv & 0xFF) << (8 * (e & 3));
  }
  v4i c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
  int ok_mfma = 1;
  for (int rr = 0; rr < 4; rr++) {
    int row = 4 * gq + rr, col = lane & 15;
    int k = (5 * row + 3) % 64;
    int want = (row + 1) * ((k % 7) - 3 + col);
    if (c[rr] != want) ok_mfma = 0;
  }
  int all_perm = __all(ok_perm), all_mfma = __all(ok_mfma);
  if (lane == 0) {
    out[0] = all_perm;
    out[1] = all_mfma;
    out[2] = (int)r;
  }
}

void selftest() {
  DevBuf<int> d;
  d.ensure(4);
  hipLaunchKernelGGL(k_selftest, dim3(1), dim3(64), 0, 0, d.p);
  BSN_HIP(hipGetLastError());
  int h[4] = {0, 0, 0, 0};
  BSN_HIP(hipMemcpy(h, d.p, sizeof(h), hipMemcpyDeviceToHost));
  if (!h[0]) fail("selftest: v_perm_b32 byte order differs from the assumed {hi:lo} (got %08x)", h[2]);
  if (!h[1]) fail("selftest: v_mfma_i32_16x16x64_i8 operand/accumulator layout differs");
}

}  // namespace bsn
