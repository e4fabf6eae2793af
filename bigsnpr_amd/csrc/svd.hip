// svd.hip — HIP backend of the block-Lanczos partial SVD (svd_driver.hpp) and the
// bsn_bed_randomsvd entry point (replaces bed_randomSVD -> bigstatsr::big_randomSVD,
// R/autoSVD.R:205-219).  The two matrix passes per step are op_cprod / op_prod
// (matvec.hip); the tall-skinny fp64 panel algebra below is memory-bound on the basis Q
// (n x p doubles) and is a few percent of a step.
#include <chrono>
#include <unistd.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstring>
#include <memory>
#include <functional>

#include "bsn_internal.hpp"
#include "orth_small.hpp"
#include "svd_driver.hpp"

namespace bsn {

constexpr int kMaxB = 16;
constexpr int kTP = 256;  // basis columns per LDS tile of k_gemm_nn

__device__ __forceinline__ uint32_t hmix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// rows [row0, row0 + n) of the n_total x b start block (a function of the global row index, so that
// sample blocks of different ranks are pieces of the same matrix); rows past n_total are zero
__global__ void k_random(double *W, int64_t ld, int64_t n, int b, uint32_t seed, int64_t row0, int64_t n_total) {
  int64_t il = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int j = blockIdx.y;
  if (il >= n || j >= b) return;
  const int64_t i = il + row0;
  if (i >= n_total) {
    W[il + j * ld] = 0.0;
    return;
  }
  uint32_t h = hmix((uint32_t)i * 0x9E3779B1U + hmix(seed * 0x85EBCA6BU + (uint32_t)j + 0x1234567U));
  uint32_t h2 = hmix(h ^ 0xDEADBEEFU);
  // uniform in (-1, 1) with 53 random bits
  double x = ((double)(((uint64_t)h << 21) ^ (uint64_t)h2) + 0.5) * (1.0 / 9007199254740992.0);
  W[il + j * ld] = 2.0 * x - 1.0;
}

typedef double v4d __attribute__((ext_vector_type(4)));
struct __attribute__((aligned(8))) d2 {
  double x, y;
};
constexpr int64_t kGemmRows = 1024;   // rows per workgroup of k_gemm_tn (4 waves x 256)

// C (p x cb, leading dimension ldc) = A[:, :p]' B[:, :cb]: the tall-skinny TN product of the panel algebra
// on the fp64 matrix pipe.  v_mfma_f64_16x16x4 with the 16 columns of an A tile as M, the (<= 16) columns
// of B as N and four rows as K: lane l supplies A[row, tile*16 + (l & 15)] and B[row, l & 15] for row slot
// l >> 4; which row a slot stands for is free as long as A and B agree, so a lane takes TWO consecutive
// rows per 16-byte load (rows r + 2 (l >> 4) + {0, 1} of an 8-row group, two MFMAs): every load
// instruction then reads 64 contiguous bytes per column, straight from global memory — A is read exactly
// once, B once per 16 columns of A (round 2's VALU kernel re-read B once per FOUR columns of A and was
// the largest item of a solve outside the streaming passes).
// Workgroup (tile, rc) covers rows [rc, rc + 1) * kGemmRows with four waves and writes the 16 x 16 sums to
// partial[rc][tile]; k_gemm_tn_reduce adds the chunks up.
// NT: 16-column tiles of B (cb <= 16 NT); an A tile is loaded once for all of them.
template <int NT>
__global__ __launch_bounds__(256) void k_gemm_tn(const double *__restrict__ A, int64_t lda, int p,
                                                 const double *__restrict__ B, int64_t ldb, int cb, int64_t n,
                                                 double *partial) {
  const int tile = blockIdx.x, rc = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int a = lane & 15, kq = lane >> 4;
  // columns past p (past cb) are read from a valid one and discarded (masked): no branch around the loads
  const int ca = tile * 16 + a < p ? tile * 16 + a : p - 1;
  const double *Ap = A + (int64_t)ca * lda;
  bool bval[NT];
  const double *Bp[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    bval[t] = t * 16 + a < cb;
    Bp[t] = B + (int64_t)(bval[t] ? t * 16 + a : 0) * ldb;
  }
  const int64_t c0 = (int64_t)rc * kGemmRows, c1 = c0 + kGemmRows < n ? c0 + kGemmRows : n;
  const int64_t r0 = c0 + wave * (kGemmRows / 4);
  const int64_t r1 = r0 + kGemmRows / 4 < c1 ? r0 + kGemmRows / 4 : c1;
  v4d acc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
  // groups of 32 rows through two register sets: the loads of a group are in flight while the MFMAs of the
  // previous one run (a wave that waited out every group's memory round trip left the kernel latency-bound
  // at a quarter of the bandwidth).  No branch around the loads: past the end the last group is loaded again.
  d2 av0[4], bv0[NT][4], av1[4], bv1[NT][4];
  auto ld = [&](int64_t rr, d2 (&av)[4], d2 (&bv)[NT][4]) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      av[u] = *(const d2 *)(Ap + rr + 8 * u + 2 * kq);
#pragma unroll
      for (int t = 0; t < NT; t++) bv[t][u] = *(const d2 *)(Bp[t] + rr + 8 * u + 2 * kq);
    }
  };
  auto mm = [&](const d2 (&av)[4], const d2 (&bv)[NT][4]) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int t = 0; t < NT; t++) {
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u].x, bval[t] ? bv[t][u].x : 0.0, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u].y, bval[t] ? bv[t][u].y : 0.0, acc[t], 0, 0, 0);
      }
  };
  const int64_t nfull = r1 > r0 ? (r1 - r0) / 32 : 0;
  if (nfull > 0) {
    ld(r0, av0, bv0);
    for (int64_t g = 0; g < nfull; g += 2) {
      ld(r0 + 32 * (g + 1 < nfull ? g + 1 : nfull - 1), av1, bv1);
      mm(av0, bv0);
      ld(r0 + 32 * (g + 2 < nfull ? g + 2 : nfull - 1), av0, bv0);
      if (g + 1 < nfull) mm(av1, bv1);
    }
  }
  int64_t r = r0 + 32 * nfull;
  for (; r < r1; r += 8) {   // the ragged end of the last chunk
    const int64_t i0 = r + 2 * kq, i1 = i0 + 1;
    const double a0 = i0 < r1 ? Ap[i0] : 0.0, a1 = i1 < r1 ? Ap[i1] : 0.0;
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const double b0 = (bval[t] && i0 < r1) ? Bp[t][i0] : 0.0, b1 = (bval[t] && i1 < r1) ? Bp[t][i1] : 0.0;
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[t], 0, 0, 0);
    }
  }
  // D: column of B = lane & 15, column of the A tile = (lane >> 4) + 4 * reg
  __shared__ double red[4][256];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    if (t > 0) __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; g++) red[wave][(kq + 4 * g) * 16 + a] = acc[t][g];
    __syncthreads();
    partial[(((int64_t)rc * gridDim.x + tile) * NT + t) * 256 + tid] =
        (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }
}

// C[tile*16 + mrow, ncol] = sum over the row chunks, in a fixed order (four interleaved running sums, then
// their sum): identical on every rank and in every run.  (A single launch with a "last workgroup adds up"
// tail was tried in round 3: the device-scope fences it needs write back the L2 of every XCD, 0.15 us per
// workgroup, several times the product itself.)
__global__ __launch_bounds__(1024) void k_gemm_tn_reduce(const double *__restrict__ partial, int nrc, int ntile, int p,
                                                         int cb, double *C, int ldc) {
  const int tile = blockIdx.x, bt = blockIdx.y, nt = gridDim.y, t = threadIdx.x & 255, part = threadIdx.x >> 8;
  double sum = 0;
#pragma unroll 4
  for (int c = part; c < nrc; c += 4) sum += partial[(((int64_t)c * ntile + tile) * nt + bt) * 256 + t];
  __shared__ double sp[4][256];
  sp[part][t] = sum;
  __syncthreads();
  if (part == 0) {
    sum = (sp[0][t] + sp[1][t]) + (sp[2][t] + sp[3][t]);
    const int mrow = t >> 4, ncol = bt * 16 + (t & 15);
    if (tile * 16 + mrow < p && ncol < cb) C[(tile * 16 + mrow) + (int64_t)ncol * ldc] = sum;
  }
}

// Out[i, j] = alpha * In[i, j] + beta * sum_a Q[i, a] S[a, j],  j < nc <= 8; S (p x nc) in global
__global__ __launch_bounds__(256) void k_gemm_nn(const double *__restrict__ Q, int64_t ldq, int p,
                                                 const double *__restrict__ S, int nc,
                                                 const double *In, int64_t ldi, double alpha,
                                                 double beta, double *Out, int64_t ldo, int64_t n) {
  extern __shared__ __attribute__((aligned(16))) double sS[];  // tile of S: TP x kMaxB
  constexpr int TP = kTP;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double acc[kMaxB];
#pragma unroll
  for (int j = 0; j < kMaxB; j++) acc[j] = 0;
  for (int a0 = 0; a0 < p; a0 += TP) {
    int ta = p - a0 < TP ? p - a0 : TP;
    __syncthreads();
    for (int t = threadIdx.x; t < ta * kMaxB; t += blockDim.x) {
      int a = t / kMaxB, j = t % kMaxB;
      sS[t] = j < nc ? S[(a0 + a) + (int64_t)j * p] : 0.0;
    }
    __syncthreads();
    if (i < n) {
      for (int a = 0; a < ta; a++) {
        double q = Q[i + (int64_t)(a0 + a) * ldq];
#pragma unroll
        for (int j = 0; j < kMaxB; j++) acc[j] += q * sS[a * kMaxB + j];
      }
    }
  }
  if (i < n) {
#pragma unroll
    for (int j = 0; j < kMaxB; j++)
      if (j < nc) {
        double base = alpha != 0.0 ? alpha * In[i + (int64_t)j * ldi] : 0.0;
        Out[i + (int64_t)j * ldo] = base + beta * acc[j];
      }
  }
}

__global__ void k_put_block(double *M, int ld, int r, const double *src) {   // M[:r, :r] (leading dimension ld) = src (r x r)
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < r * r) M[(t % r) + (int64_t)(t / r) * ld] = src[t];
}

// in place W[:, :r] = W[:, :cb] * M (cb x r), one thread per row
__global__ void k_right_mult(double *W, int64_t ld, int64_t n, int cb, int r, const double *M) {
  __shared__ double sM[kMaxB * kMaxB];
  if ((int)threadIdx.x < cb * r) sM[threadIdx.x] = M[threadIdx.x];
  __syncthreads();
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double w[kMaxB], o[kMaxB];
#pragma unroll
  for (int j = 0; j < kMaxB; j++) w[j] = j < cb ? W[i + j * ld] : 0.0;
#pragma unroll
  for (int c = 0; c < kMaxB; c++) {
    double s = 0;
#pragma unroll
    for (int j = 0; j < kMaxB; j++)
      if (j < cb && c < r) s += w[j] * sM[j + c * cb];
    o[c] = s;
  }
#pragma unroll
  for (int c = 0; c < kMaxB; c++)
    if (c < r) W[i + c * ld] = o[c];
}

// The small-matrix half of one pass of the two-pass orthonormalisation (orth_small.hpp) on one
// workgroup; the coefficient iterate lives in LDS.  mx_clear: the column maxima that the following
// k_update accumulates (rounding of the finished block) start from zero.
struct DevCtx {
  int tid, nt;
  __device__ void sync() { __syncthreads(); }
};
__global__ __launch_bounds__(512) void k_orth2(OrthSmall a, unsigned long long *mx_clear) {
  __shared__ double sCs[kOrthMaxP * kOrthMaxB];
  __shared__ double sG[kOrthMaxB * kOrthMaxB], sR[kOrthMaxB * kOrthMaxB], sRi[kOrthMaxB * kOrthMaxB],
      sRo[kOrthMaxB * kOrthMaxB], sD[kOrthMaxB * kOrthMaxB], sT[2 * kOrthMaxB + 4];
  a.Cs = sCs;
  a.Gs = sG;
  a.Rs = sR;
  a.Ris = sRi;
  a.Ro = sRo;
  a.Dv = sD;
  a.tmp = sT;
  if (mx_clear && threadIdx.x < kOrthMaxB) mx_clear[threadIdx.x] = 0ull;
  DevCtx cx{(int)threadIdx.x, (int)blockDim.x};
  orth_small(cx, a);
}

// W[i, :cb] <- (W[i, :cb] - Q[i, :p] C) Ri   in place, one thread per row (the tall half of a pass);
// mx != null: also the column maxima of the result (bits of a non-negative double order like u64)
template <int CBT>
__global__ __launch_bounds__(256) void k_update(const double *__restrict__ Q, int64_t ldq, int p,
                                                const double *__restrict__ C, const double *__restrict__ Ri,
                                                int cb, double *W, int64_t ldw, int64_t n,
                                                unsigned long long *mx) {
  constexpr int TP = 128, ROWS = 4;   // a workgroup covers 1024 rows: few atomics on the column maxima
  __shared__ double sS[TP * CBT], sRi[CBT * CBT], smx[4][CBT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int t = tid; t < CBT * CBT; t += 256) {
    const int j = t % CBT, c = t / CBT;
    sRi[t] = (j < cb && c < cb) ? Ri[j + c * cb] : 0.0;
  }
  if (tid < 4 * CBT) smx[tid / CBT][tid % CBT] = 0.0;
  const bool one_tile = p <= TP;
  for (int it = 0; it < ROWS; it++) {
    const int64_t i = ((int64_t)blockIdx.x * ROWS + it) * 256 + tid;
    double acc[CBT];
#pragma unroll
    for (int j = 0; j < CBT; j++) acc[j] = 0;
    for (int a0 = 0; a0 < p; a0 += TP) {
      const int ta = p - a0 < TP ? p - a0 : TP;
      if (!(one_tile && it > 0)) {   // (uniform) the coefficient tile stays in LDS when it is the only one
        __syncthreads();
        for (int t = tid; t < ta * CBT; t += 256) {
          const int a = t / CBT, j = t % CBT;
          sS[t] = j < cb ? C[(a0 + a) + (int64_t)j * p] : 0.0;
        }
        __syncthreads();
      }
      if (i < n) {
        // (unrolled by 8: the eight loads of Q go out together instead of one memory round trip per column)
#pragma unroll 8
        for (int a = 0; a < ta; a++) {
          const double q = Q[i + (int64_t)(a0 + a) * ldq];
#pragma unroll
          for (int j = 0; j < CBT; j++) acc[j] += q * sS[a * CBT + j];
        }
      }
    }
    if (p == 0 && it == 0) __syncthreads();   // sRi / smx are ready
    double w[CBT];
#pragma unroll
    for (int j = 0; j < CBT; j++) w[j] = (i < n && j < cb) ? W[i + j * ldw] - acc[j] : 0.0;
    // one output column at a time (the loop is kept rolled: 16 x 16 products unrolled need 300 registers)
#pragma unroll 1
    for (int c = 0; c < cb; c++) {
      double s = 0;
#pragma unroll
      for (int j = 0; j < CBT; j++) s += w[j] * sRi[j + c * CBT];
      if (i < n) W[i + c * ldw] = s;
      if (mx) {
        double m = fabs(s);
        for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off));
        if (lane == 0) smx[wave][c] = fmax(smx[wave][c], m);
      }
    }
  }
  if (mx) {
    __syncthreads();
    if (tid < cb) {
      const double m = fmax(fmax(smx[0][tid], smx[1][tid]), fmax(smx[2][tid], smx[3][tid]));
      atomicMax(&mx[tid], (unsigned long long)__double_as_longlong(m));
    }
  }
}

// ---- rounding of a finished basis block to the fixed-point grid of the streaming products ----
// (svd_driver.hpp: the products are then exact for the stored vectors).  The scale is the largest
// power of two with absmax * qs <= 0.98 * 2^(8S-1): one notch below the 0.99 the product's own
// quantiser uses, so that re-quantising the rounded vector can only pick the same or a finer grid.
__global__ void k_col_absmax(const double *W, int64_t ld, int64_t n, unsigned long long *mx) {
  const int v = blockIdx.y;
  double m = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = fmax(m, fabs(W[i + v * ld]));
  for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off));
  __shared__ double sm[16];
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) sm[wave] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < nw; w++) m = fmax(m, sm[w]);
    atomicMax(&mx[v], (unsigned long long)__double_as_longlong(m));
  }
}
__global__ void k_round_cols(double *W, int64_t ld, int64_t n, const unsigned long long *mx, int slices) {
  const int v = blockIdx.y;
  const double m = __longlong_as_double((long long)mx[v]);
  if (!(m > 0)) return;
  int e;
  frexp(ldexp(0.98, 8 * slices - 1) / m, &e);
  const double qs = ldexp(1.0, e - 1), iq = ldexp(1.0, 1 - e);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) W[i + v * ld] = (double)llrint(W[i + v * ld] * qs) * iq;
}

// ---- sample-block layout of a panel for the collectives ------------------------------------
// full: n x cb column-major (ld n).  blk: [rank][column][row of the rank's block], blocks of nr
// rows (the last one zero-padded) — the chunk of rank r is contiguous, which is what
// reduce-scatter / all-gather exchange.
__global__ void k_block(const double *__restrict__ full, int64_t n, int cb, int64_t nr, int world,
                        double *__restrict__ blk) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)world * cb * nr) return;
  const int64_t i = t % nr, j = (t / nr) % cb, r = t / (nr * cb);
  const int64_t gi = r * nr + i;
  blk[t] = gi < n ? full[gi + j * n] : 0.0;
}
__global__ void k_unblock(const double *__restrict__ blk, int64_t n, int cb, int64_t nr,
                          double *__restrict__ full) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * cb) return;
  const int64_t gi = t % n, j = t / n;
  full[t] = blk[((gi / nr) * cb + j) * nr + gi % nr];
}
// ---- the all-gather of a finished basis block as the 16-bit integers it is rounded to (sharded solve, round 4) ----
// mx[v] = max over the ranks of their column maxima (bit patterns of non-negative doubles order like integers)
__global__ void k_max_over_ranks(const unsigned long long *__restrict__ all, int world, int cb,
                                 unsigned long long *__restrict__ mx) {
  const int v = threadIdx.x;
  if (v >= cb) return;
  unsigned long long m = 0;
  for (int r = 0; r < world; r++) m = all[r * cb + v] > m ? all[r * cb + v] : m;
  mx[v] = m;
}
// the rounding of k_round_cols on this rank's rows: the integers go to q (column-major, nr rows), the rounded values
// back into W
template <typename INT>   // int16_t up to 16 bits, int32_t up to 32
__global__ void k_pack_int(double *__restrict__ W, int64_t nr, const unsigned long long *__restrict__ mx, int slices,
                           INT *__restrict__ q) {
  const int v = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nr) return;
  const double m = __longlong_as_double((long long)mx[v]);
  if (!(m > 0)) {
    q[i + v * nr] = 0;
    return;   // (an all-zero column stays as it is, like in k_round_cols)
  }
  int e;
  frexp(ldexp(0.98, 8 * slices - 1) / m, &e);
  const double qs = ldexp(1.0, e - 1), iq = ldexp(1.0, 1 - e);
  const long long k = llrint(W[i + v * nr] * qs);
  q[i + v * nr] = (INT)k;
  W[i + v * nr] = (double)k * iq;
}
// all ranks' integers ([rank][column][row of the block]) -> the rounded block with all n rows (column-major, ld n)
template <typename INT>
__global__ void k_unpack_int(const INT *__restrict__ q, int64_t n, int cb, int64_t nr,
                             const unsigned long long *__restrict__ mx, int slices, double *__restrict__ full) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * cb) return;
  const int64_t gi = t % n, v = t / n;
  const double m = __longlong_as_double((long long)mx[v]);
  double val = 0.0;
  if (m > 0) {
    int e;
    frexp(ldexp(0.98, 8 * slices - 1) / m, &e);
    val = (double)q[((gi / nr) * cb + v) * nr + gi % nr] * ldexp(1.0, 1 - e);
  }
  full[t] = val;
}

// a received segment (rows x cb, ld rows) into rows [r0, r0 + rows) of the panel W (ld nr)
__global__ void k_place_rows(const double *__restrict__ src, int64_t rows, int cb, int64_t r0, int64_t nr,
                             double *__restrict__ W) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * cb) return;
  W[r0 + t % rows + (t / rows) * nr] = src[t];
}
// W (nr x cb, ld nr) = rows [row0, row0 + nr) of full (n x cb, ld n), zero past n
__global__ void k_take_rows(const double *__restrict__ full, int64_t n, int cb, int64_t nr, int64_t row0,
                            double *__restrict__ W) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nr * cb) return;
  const int64_t i = t % nr, j = t / nr;
  W[t] = row0 + i < n ? full[row0 + i + j * n] : 0.0;
}

// Panels of length n (samples) are held by SAMPLE BLOCKS when the solve is distributed: rank r owns
// rows [r nr, (r+1) nr) of the basis Q and of the working panel W, orthogonalises them locally and
// shares only the small coefficient matrices (p x b, b x b) — the n-side algebra is divided by the
// number of ranks instead of being replicated.  The variants (columns of G, rows of Z) are sharded
// as before.  Per block step (round 3: two small sums over the ranks instead of eight, every
// collective on the solve's own stream):
//   Z_g = A_g' Qfull           local crossproduct pass; Qfull = newest basis block, all n rows
//   Wfull = A_g Z_g            local product pass, partial sums over this rank's variants
//   W_r = reduce-scatter(Wfull) by sample blocks
//   all-reduce #1 of [Z'Z block | Q'Q block | Q'W | W'W], project + normalise W_r (orth_small.hpp)
//   all-reduce #2 of [Q'W | W'W] of the new W_r, project + normalise again
//   Qfull = round(all-gather(W_r)); Q_r gets its rows
// Without a communicator (and without the test hook) nr = n and every collective is skipped: the
// single-GPU path is the same code.
// The working panel W is not a buffer of its own: it IS the next free columns of Q (W = Q[:, p : p+cb]),
// so [Q W]' W is one tall product over contiguous columns and a finished block needs no copy.
// device buffers of a solve; lives on the bed handle between solves (grow-only)
struct SvdWorkspace {
  DevBuf<double> Q, Z, partial, dsmall, Wsave, dorth, Wfull, Wblk, Qfull, dS, dU, dV, dUfull, dM;
  // pinned host staging for the small matrices that cross the bus every block step (Gram blocks,
  // orthogonalisation coefficients): copies to pageable memory go through the runtime's own staging
  // and were measured to cost milliseconds each once a process has run a few solves
  double *h_pin = nullptr;
  size_t h_pin_n = 0;
  hipEvent_t ev_small = nullptr;   // the small matrices of a block step have reached the host
  // sharded solve, product pass in segments (A_Zblock): the stream the reduce-scatters run on, one event per segment
  // ("its partial sums are ready") and one for "all segments have arrived"; the segments' sample lists and receive buffer
  hipStream_t st_comm = nullptr;
  hipEvent_t ev_seg[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_arrived = nullptr;
  DevBuf<int32_t> seg_rows;
  DevBuf<double> Wrecv;
  int64_t seg_n = -1, seg_nr = -1;
  int seg_world = -1;
  double *pinned(size_t count) {
    if (count > h_pin_n) {
      if (h_pin) (void)hipHostFree(h_pin);
      h_pin = nullptr;
      BSN_HIP(hipHostMalloc((void **)&h_pin, count * sizeof(double), hipHostMallocDefault));
      h_pin_n = count;
    }
    return h_pin;
  }
  ~SvdWorkspace() {
    for (auto &e : ev_seg)
      if (e) (void)hipEventDestroy(e);
    if (ev_arrived) (void)hipEventDestroy(ev_arrived);
    if (st_comm) (void)hipStreamDestroy(st_comm);
    if (ev_small) (void)hipEventDestroy(ev_small);
    if (h_pin) (void)hipHostFree(h_pin);
  }
};

struct HipSvdBackend : SvdBackend {
  SvdWorkspace &ws;
  explicit HipSvdBackend(SvdWorkspace &w)
      : ws(w), Q(w.Q), Z(w.Z), partial(w.partial), dsmall(w.dsmall), Wsave(w.Wsave), dorth(w.dorth),
        Wfull(w.Wfull), Wblk(w.Wblk), Qfull(w.Qfull) {}
  bsn_op *op = nullptr;
  hipStream_t st = nullptr;
  bsn_allreduce_fn hook = nullptr;
  void *ctx = nullptr;
  bsn_comm *comm = nullptr;
  int rank = 0, world = 1;
  bool dist = false;
  int64_t nr = 0, row0 = 0;  // rows of a sample block, first row of this rank's block
  DevBuf<double> &Q, &Z, &partial, &dsmall, &Wsave, &dorth, &Wfull, &Wblk, &Qfull;
  double *Wc = nullptr;      // the working panel: columns wcol .. of Q
  int wcol = 0;
  std::vector<double> horth;
  int cap = 0, b = 0;
  int kmax = 0;
  bool mx_valid = false;     // the column maxima of W (rounding) came out of the last k_update
  bool no_fused = false;     // BSN_NO_FUSED_STEP=1: the step-by-step path only (A/B, debugging)
  bool speculate = true;     // queue the start of the next step behind the orthonormalisation (BSN_NO_SPECULATION=1: off)
  bool spec_rounded = false; // ... and it has been: the driver's next round_W is a no-op
  int n_small_ar = 0;        // small all-reduces issued (diagnostics)
  // warm start on a leading subset of this rank's variants
  int64_t m_op_full = 0, m_sub = 0;
  bool fused_stats = false;
  int warm_launches = 0, warm_den = 16;
  bool subset(bool on) override {
    if (ooc) return false;   // (a warm start on the leading variants would walk the first slab twice: not worth the special case)
    if (on) {
      m_op_full = op->m;
      // small matrices: a full pass is cheap and the thinned matrix too noisy.  The thinned operator is
      // the sum over ranks of the shards' subsets, so the size that matters is the one over all ranks;
      // the decision is taken from quantities that are identical on every rank (a rank that skipped the
      // warm start while the others run its collectives would hang them).
      const int64_t m_lo = m_total / (world > 0 ? world : 1);
      const int64_t ms = (m_lo / warm_den) / 256 * 256;
      if (m_total / warm_den < 16384 || ms < 256) return false;
      m_sub = ms < op->m ? ms : op->m;
      op->m = m_sub;
      op->prof_kind_override = 3;
      roctx_push("svd:warm start (leading 1/16 of the variants)");
      return true;
    }
    roctx_pop();
    op->m = m_op_full;
    op->prof_kind_override = -1;
    warm_launches += 2;
    if (fused_stats) {  // the counting pass saw the subset only: count again on the first full pass
      op->stats_pending = true;
      op->na_poll = false;
      op->no_na = false;
    }
    return true;
  }
  // host wall time per phase (BSN_TIMING=1): where a solve's time outside the streaming kernels goes
  bool timing = false;
  double t_phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // alloc, At_Q, A_Z, grams, orth, round/copy, finalize, other
  int n_sync = 0;  // host synchronisations with the solve's stream
  struct Tick {
    HipSvdBackend *b;
    int ph;
    std::chrono::steady_clock::time_point t0;
    Tick(HipSvdBackend *b_, int ph_) : b(b_), ph(ph_), t0(std::chrono::steady_clock::now()) {
      static const char *names[8] = {"svd:alloc", "svd:crossprod pass (Z = A'Q)", "svd:product pass (W = A Z)", "svd:gram blocks",
                                     "svd:orthonormalise", "svd:round / copy block", "svd:form u, v", "svd:other"};
      roctx_push(names[ph & 7]);
    }
    ~Tick() {
      roctx_pop();
      if (b->timing)
        b->t_phase[ph] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
  };
  void sync_stream() {
    BSN_HIP(hipStreamSynchronize(st));
    n_sync++;
  }

  void setup_ranks() {
    if (comm) {
      rank = comm->rank;
      world = comm->world;
    }
    dist = comm != nullptr || hook != nullptr;
    if (!dist) world = 1, rank = 0;
    nr = dist ? (n + world - 1) / world : n;
    // RCCL path with sample blocks of at least 4096 rows: whole workgroup blocks of 512 samples per rank, so that the
    // product pass can be cut into segments of blocks whose reduce-scatters overlap the next segment (A_Zblock); the
    // padding (< 512 rows per rank) is zero rows like the padding of the last block has always been
    if (dist && comm && nr >= 4096) nr = (nr + 511) / 512 * 512;
    row0 = (int64_t)rank * nr;
  }
  void alloc(int cap_, int b_) override {
    Tick tk(this, 0);
    if (b_ > kMaxB) fail("block size must be <= %d", kMaxB);
    cap = cap_;
    b = b_;
    Q.ensure((size_t)nr * (cap + kMaxB));   // the working panel sits behind the basis
    Z.ensure((size_t)m_local * cap);
    {
      const int64_t rows_max = nr > m_local ? nr : m_local;
      partial.ensure((size_t)((rows_max + kGemmRows - 1) / kGemmRows) * ((cap + kMaxB + 15) / 16) * 2 * 256);
    }
    dsmall.ensure((size_t)(cap + 4) * 64);
    Wsave.ensure((size_t)nr * kMaxB);
    dorth.ensure((size_t)16 + 3 * kMaxB * kMaxB + (size_t)7 * (cap + kMaxB + 4) * kMaxB);
    ws.dM.ensure((size_t)kOrthMaxP * kOrthMaxP);
    if (dist) {
      const int wide = kmax > kMaxB ? kmax : kMaxB;
      Wfull.ensure((size_t)n * wide);
      Wblk.ensure((size_t)world * nr * wide);
      Qfull.ensure((size_t)n * kMaxB);
    }
    Wc = Q.p;
    wcol = 0;
  }

  // ---- collectives (all on the solve's stream: one communicator, one stream, issue order = stream order) ----
  // sum of a small device matrix over the ranks, in place
  void ar_small(double *d, int64_t count) {
    if (!dist) return;
    n_small_ar++;
    if (comm) {
      comm_allreduce_sum(comm, d, count, st);
    } else {
      sync_stream();
      hook(d, count, ctx);
    }
  }
  void reduce_scatter_W(int cb) {  // Wfull (n x cb partial sums) -> W (nr x cb, summed over ranks)
    const int64_t tot = (int64_t)world * cb * nr;
    hipLaunchKernelGGL(k_block, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, Wfull.p, n, cb, nr, world,
                       Wblk.p);
    BSN_HIP(hipGetLastError());
    if (comm) {
      comm_reduce_scatter_sum(comm, Wblk.p, Wc, nr * cb, st);
    } else {
      sync_stream();
      hook(Wblk.p, tot, ctx);
      BSN_HIP(hipMemcpyAsync(Wc, Wblk.p + (int64_t)rank * cb * nr, (size_t)nr * cb * 8,
                             hipMemcpyDeviceToDevice, st));
    }
  }
  // src (nr x cb local rows) -> dst (n x cb, all rows, column-major)
  void all_gather_rows(const double *src, int cb, double *dst) {
    const int64_t tot = (int64_t)world * cb * nr;
    if (comm) {
      comm_all_gather(comm, src, Wblk.p, nr * cb, st);
    } else {
      BSN_HIP(hipMemsetAsync(Wblk.p, 0, (size_t)tot * 8, st));
      BSN_HIP(hipMemcpyAsync(Wblk.p + (int64_t)rank * cb * nr, src, (size_t)nr * cb * 8, hipMemcpyDeviceToDevice, st));
      sync_stream();
      hook(Wblk.p, tot, ctx);
    }
    hipLaunchKernelGGL(k_unblock, dim3((unsigned)((n * cb + 255) / 256)), dim3(256), 0, st, Wblk.p, n, cb, nr, dst);
    BSN_HIP(hipGetLastError());
  }

  // ---- backend interface ------------------------------------------------------------------
  void random_W(int bb, uint32_t seed) override {
    Wc = Q.p;
    wcol = 0;
    mx_valid = false;
    spec_rounded = false;
    op->preq_X = nullptr;
    hipLaunchKernelGGL(k_random, dim3((unsigned)((nr + 255) / 256), bb), dim3(256), 0, st, Wc, nr, nr, bb, seed,
                       row0, n);
    BSN_HIP(hipGetLastError());
  }
  // the newest basis block with all n rows: Q itself on one GPU, the gathered copy otherwise
  const double *newest_block(int p0) const { return dist ? Qfull.p : Q.p + (int64_t)p0 * nr; }
  // ---- out-of-core handle (round 5): both passes of a block step WALK THE FILE in slabs of variants --------------
  // The reference maps a .bed of any size and solves on it (src/bed-acc.h:46, R/autoSVD.R:205-219).  Here a file
  // whose image does not fit the device is served slab by slab (bsn_internal.hpp): every slab is uploaded into the
  // one resident slab image (pread -> page-locked double buffer -> DMA, on the solve's stream) and the same streaming
  // kernels run on it — Z[slab rows] = A_slab' Q for the crossproduct pass, W += A_slab Z[slab rows] for the product
  // pass (added on the device).  The counts of the codes ride along the first crossproduct walk slab by slab.
  // Each pass moves the whole file over PCIe (about 40 GB/s against 4 - 5 TB/s from HBM): the solve is paced by the
  // link, 2.5 s per pass per 100 GB; what it buys is that no matrix is refused for its size.
  bool ooc = false;
  // (round 6) a solve over a LIST of variants in file order (ind.col = ind.keep of bed_autoSVD, R/autoSVD.R:296-301): slab
  // sl serves positions [sel_at[sl], sel_at[sl + 1]) of the list, sel_local = their indices inside the slab image
  const int64_t *ooc_cols = nullptr;
  std::vector<int64_t> sel_at, sel_local;
  std::unique_ptr<bsn_op> sop;        // the operator over the resident slab
  long long ooc_na_total = 0, ooc_n_bad = 0;
  bool ooc_stats_walk = false;        // this crossproduct walk also counts the codes (first full pass of the solve)
  // positions [*at, *at + returned count) of the operator's variants live in slab sl (uploaded here); 0: none of them do
  int64_t slab_selection(int64_t sl, int64_t *at) {
    if (!ooc_cols) {
      int64_t j0 = 0;
      const int64_t cnt = slab_upload(op->bed, sl, &j0);
      *at = j0;
      sel_local.clear();
      return cnt;
    }
    if (sel_at.empty()) {   // the list is in file order (checked by the entry point): one contiguous run of it per slab
      const int64_t nslab = slab_count(op->bed), sc = op->bed->slab_cols;
      sel_at.assign((size_t)nslab + 1, 0);
      for (int64_t t = 0; t < m_local; t++) sel_at[(size_t)(ooc_cols[t] / sc) + 1]++;
      for (int64_t q = 0; q < nslab; q++) sel_at[(size_t)q + 1] += sel_at[(size_t)q];
    }
    const int64_t a = sel_at[(size_t)sl], cnt = sel_at[(size_t)sl + 1] - a;
    *at = a;
    if (cnt <= 0) return 0;
    int64_t j0 = 0;
    slab_upload(op->bed, sl, &j0);
    sel_local.resize((size_t)cnt);
    for (int64_t t = 0; t < cnt; t++) sel_local[(size_t)t] = ooc_cols[a + t] - j0;
    return cnt;
  }
  void slab_operator(int64_t j0, int64_t cnt, bool with_scaling) {
    bsn_bed *img = slab_image(op->bed);
    if (!sop) sop.reset(new bsn_op());
    sop->bed = img;
    sop->n = n;
    sop->m = cnt;
    sop->rows_identity = true;
    sop->cols_contig = true;
    sop->col0 = 0;
    if (!sel_local.empty()) {   // a selection inside the slab: a range of it, or a gather list (padded like fill_op's)
      bool contig = true;
      for (int64_t t = 1; t < cnt && contig; t++) contig = sel_local[(size_t)t] == sel_local[0] + t;
      sop->cols_contig = contig;
      sop->col0 = contig ? sel_local[0] : 0;
      if (!contig) {
        const int64_t m_pad = round_up(cnt, 64);
        std::vector<int32_t> c((size_t)m_pad, (int32_t)sel_local[0]);
        for (int64_t t = 0; t < cnt; t++) c[(size_t)t] = (int32_t)sel_local[(size_t)t];
        copy_h2d(op->bed, sop->d_cols.ensure((size_t)m_pad), c.data(), (size_t)m_pad * 4);
        BSN_HIP(hipStreamSynchronize(st));   // (`c` is released on return)
      }
    }
    sop->no_na = false;
    sop->slices = op->slices;
    sop->profile = op->profile;
    sop->prof_kind_override = op->prof_kind_override;
    sop->preq_X = nullptr;
    sop->d_center.ensure((size_t)op->bed->slab_cols);
    sop->d_scale.ensure((size_t)op->bed->slab_cols);
    if (with_scaling) {
      BSN_HIP(hipMemcpyAsync(sop->d_center.p, op->d_center.p + j0, (size_t)cnt * 8, hipMemcpyDeviceToDevice, st));
      BSN_HIP(hipMemcpyAsync(sop->d_scale.p, op->d_scale.p + j0, (size_t)cnt * 8, hipMemcpyDeviceToDevice, st));
    }
  }
  void At_Qblock_ooc(int p0, int cb) {
    const int64_t nslab = slab_count(op->bed);
    const bool stats = op->stats_pending;
    if (stats) {
      ooc_na_total = ooc_n_bad = 0;
      op->d_na.ensure((size_t)op->m);
    }
    for (int64_t sl = 0; sl < nslab; sl++) {
      int64_t j0 = 0;
      const int64_t cnt = slab_selection(sl, &j0);
      if (cnt <= 0) continue;
      slab_operator(j0, cnt, !stats);
      sop->stats_pending = stats;
      sop->na_poll = false;
      op_cprod(sop.get(), newest_block(p0), n, cb, Z.p + (int64_t)p0 * m_local + j0, m_local);
      if (stats) {   // the slab's scaling and missing-value counts join the whole matrix's
        BSN_HIP(hipMemcpyAsync(op->d_center.p + j0, sop->d_center.p, (size_t)cnt * 8, hipMemcpyDeviceToDevice, st));
        BSN_HIP(hipMemcpyAsync(op->d_scale.p + j0, sop->d_scale.p, (size_t)cnt * 8, hipMemcpyDeviceToDevice, st));
        BSN_HIP(hipMemcpyAsync(op->d_na.p + j0, sop->d_na.p, (size_t)cnt * 4, hipMemcpyDeviceToDevice, st));
        sync_stream();
        if (sop->h_na_total && sop->h_na_total[0] >= 0) {
          ooc_na_total += sop->h_na_total[0];
          ooc_n_bad += sop->h_na_total[1];
        }
      } else {
        sync_stream();   // the slab image is overwritten by the next upload
      }
    }
    if (stats) {
      op->stats_pending = false;
      if (!op->h_na_total) BSN_HIP(hipHostMalloc((void **)&op->h_na_total, 2 * sizeof(long long), hipHostMallocDefault));
      op->h_na_total[0] = ooc_na_total;
      op->h_na_total[1] = ooc_n_bad;
    }
    op->passes++;
  }
  void A_Zblock_ooc(int p0, int cb) {
    const int64_t nslab = slab_count(op->bed);
    bool first = true;
    for (int64_t sl = 0; sl < nslab; sl++) {
      int64_t j0 = 0;
      const int64_t cnt = slab_selection(sl, &j0);
      if (cnt <= 0) continue;
      slab_operator(j0, cnt, true);
      op_prod_acc(sop.get(), Z.p + (int64_t)p0 * m_local + j0, m_local, cb, Wc, n, first ? 0.0 : 1.0);
      first = false;
      sync_stream();
    }
    op->passes++;
  }
  void At_Qblock(int p0, int cb) override {
    Tick tk(this, 1);
    if (progress) progress();
    if (ooc) return At_Qblock_ooc(p0, cb);
    op_cprod(op, newest_block(p0), n, cb, Z.p + (int64_t)p0 * m_local, m_local);
  }
  // The product pass of a sharded solve in up to four SEGMENTS of sample blocks (every rank's sample block cut at the
  // same workgroup-block boundaries): as soon as a segment's partial sums are finalised into the blocked layout its
  // reduce-scatter is queued on a second stream behind an event, and runs while the next segment's k_prodT computes;
  // the stream of the solve waits for the last arrival.  BSN_NO_OVERLAP=1: the same segments with every collective on
  // the solve's own stream (A/B: identical results).  Needs the sample-major copy (k_prodT) and sample blocks of whole
  // 512-sample workgroup blocks; otherwise — and through the host hook — the pass runs whole, followed by ONE
  // reduce-scatter.  The sums over the ranks are the same per element either way.
  bool product_in_segments(int p0, int cb) {
    if (!comm || nr % 512 != 0 || nr < 4096 || getenv("BSN_NO_SEGMENTS")) return false;
    const int B = (int)(nr / 512), nseg = B >= 16 ? 4 : 2;
    if (ws.seg_n != n || ws.seg_nr != nr || ws.seg_world != world) {   // the segments' sample lists, piece by piece
      std::vector<int32_t> rows((size_t)world * nr);
      size_t at = 0;
      for (int sgi = 0; sgi < nseg; sgi++) {
        const int b0 = (int)((int64_t)B * sgi / nseg), b1 = (int)((int64_t)B * (sgi + 1) / nseg);
        for (int r = 0; r < world; r++)
          for (int64_t t = (int64_t)b0 * 512; t < (int64_t)b1 * 512; t++) {
            const int64_t gi = (int64_t)r * nr + t;
            rows[at++] = gi < n ? (int32_t)gi : -1;
          }
      }
      copy_h2d(op->bed, ws.seg_rows.ensure(rows.size()), rows.data(), rows.size() * 4);
      BSN_HIP(hipStreamSynchronize(st));
      ws.seg_n = n; ws.seg_nr = nr; ws.seg_world = world;
    }
    const bool overlap = !getenv("BSN_NO_OVERLAP");
    if (overlap && !ws.st_comm) BSN_HIP(hipStreamCreateWithFlags(&ws.st_comm, hipStreamNonBlocking));
    for (int i = 0; i < nseg; i++)
      if (!ws.ev_seg[i]) BSN_HIP(hipEventCreateWithFlags(&ws.ev_seg[i], hipEventDisableTiming));
    if (!ws.ev_arrived) BSN_HIP(hipEventCreateWithFlags(&ws.ev_arrived, hipEventDisableTiming));
    ws.Wrecv.ensure((size_t)nr * kMaxB);
    ProdSegment segs[4];
    for (int sgi = 0; sgi < nseg; sgi++) {
      const int b0 = (int)((int64_t)B * sgi / nseg), b1 = (int)((int64_t)B * (sgi + 1) / nseg);
      segs[sgi].bs = b1 - b0;
      segs[sgi].off = b0;
      segs[sgi].d_rows = ws.seg_rows.p + (size_t)world * 512 * b0;
      segs[sgi].d_out = Wblk.p + (size_t)world * cb * 512 * b0;
    }
    hipStream_t sc = overlap ? ws.st_comm : st;
    const std::function<void(int)> after = [&](int sgi) {
      const int64_t rows_s = (int64_t)segs[sgi].bs * 512, r0 = (int64_t)segs[sgi].off * 512;
      if (overlap) {
        BSN_HIP(hipEventRecord(ws.ev_seg[sgi], st));
        BSN_HIP(hipStreamWaitEvent(sc, ws.ev_seg[sgi], 0));
      }
      double *recv = ws.Wrecv.p + (size_t)cb * r0;
      comm_reduce_scatter_sum(comm, segs[sgi].d_out, recv, rows_s * cb, sc);
      hipLaunchKernelGGL(k_place_rows, dim3((unsigned)((rows_s * cb + 255) / 256)), dim3(256), 0, sc, recv, rows_s, cb,
                         r0, nr, Wc);
      BSN_HIP(hipGetLastError());
    };
    if (overlap) {   // the second stream must not run ahead of what produced Wc / Wblk's previous contents
      BSN_HIP(hipEventRecord(ws.ev_arrived, st));
      BSN_HIP(hipStreamWaitEvent(sc, ws.ev_arrived, 0));
    }
    if (!op_prod_segments(op, Z.p + (int64_t)p0 * m_local, m_local, cb, world, B, nseg, segs, after)) return false;
    if (overlap) {
      BSN_HIP(hipEventRecord(ws.ev_arrived, sc));
      hipEvent_t t0 = comm_time_begin(comm, st);   // what the solve's stream waits here is the exposed part of the exchange
      BSN_HIP(hipStreamWaitEvent(st, ws.ev_arrived, 0));
      comm_time_end(comm, t0, 3, st);
    }
    n_seg_passes++;
    exchange_mode = std::max(exchange_mode, overlap ? 3 : 2);
    return true;
  }
  // precision schedule of the driver: the digits of the following product pass, of the rounding of the block it
  // produces and of the crossproduct pass that reads that block (everything reads op->slices when it is queued)
  int min_slices = 8;
  bool hold_round = false;
  void hold_rounding(bool hold) override { hold_round = hold; }
  void set_precision(int S) override {
    if (S < 1) S = 1;
    if (S > 7) S = 7;
    op->slices = S;
    if (sop) sop->slices = S;
    if (S < min_slices) min_slices = S;
  }
  std::function<void()> progress;   // called once per pass (the exchange watchdog's deadline moves with it)
  int n_seg_passes = 0;   // product passes that ran in segments (diagnostics / tests)
  int exchange_mode = 0;  // bsn_svd_info::exchange_mode
  int n_compact_gathers = 0;   // basis blocks all-gathered as 16-bit integers
  void A_Zblock(int p0, int cb) override {
    Tick tk(this, 2);
    mx_valid = false;
    if (ooc) return A_Zblock_ooc(p0, cb);
    if (dist && product_in_segments(p0, cb)) return;
    op_prod(op, Z.p + (int64_t)p0 * m_local, m_local, cb, dist ? Wfull.p : Wc, n);
    if (dist) {
      reduce_scatter_W(cb);
      exchange_mode = std::max(exchange_mode, 1);
    }
  }
  // dC (p x cb, leading dimension ldc) = A[:, :p]' B[:, :cb] for operands with `rows` rows (leading
  // dimension = rows); local rows only
  void gemm_tn_any(const double *A, const double *B, int64_t rows, int p, int cb, double *dC, int ldc = 0) {
    if (p <= 0 || cb <= 0) return;
    dim3 grid((unsigned)((p + 15) / 16), (unsigned)((rows + kGemmRows - 1) / kGemmRows));
    const int nt = (cb + 15) / 16;
    if (nt == 1)
      hipLaunchKernelGGL((k_gemm_tn<1>), grid, dim3(256), 0, st, A, rows, p, B, rows, cb, rows, partial.p);
    else if (nt == 2)
      hipLaunchKernelGGL((k_gemm_tn<2>), grid, dim3(256), 0, st, A, rows, p, B, rows, cb, rows, partial.p);
    else
      fail("internal: panel product with %d columns", cb);
    hipLaunchKernelGGL(k_gemm_tn_reduce, dim3(grid.x, nt), dim3(1024), 0, st, partial.p, (int)grid.y, (int)grid.x, p, cb,
                       dC, ldc > 0 ? ldc : p);
  }
  void gemm_tn(const double *A, int p, int cb, double *C_host) {
    gemm_tn_any(A, Wc, nr, p, cb, dsmall.p);
    BSN_HIP(hipGetLastError());
    ar_small(dsmall.p, (int64_t)p * cb);
    double *hp = ws.pinned((size_t)(cap + kMaxB + 4) * kMaxB * 4 + 1024);
    BSN_HIP(hipMemcpyAsync(hp, dsmall.p, (size_t)p * cb * 8, hipMemcpyDeviceToHost, st));
    sync_stream();
    std::memcpy(C_host, hp, (size_t)p * cb * 8);
  }
  void round_cols(double *X, int64_t rows, int cb, bool have_mx) {
    unsigned long long *mx = (unsigned long long *)dorth.p;
    if (!have_mx) {
      BSN_HIP(hipMemsetAsync(mx, 0, (size_t)cb * 8, st));
      hipLaunchKernelGGL(k_col_absmax, dim3(256, cb), dim3(1024), 0, st, X, rows, rows, mx);
    }
    hipLaunchKernelGGL(k_round_cols, dim3((unsigned)((rows + 255) / 256), cb), dim3(256), 0, st, X, rows, rows, mx,
                       op->slices);
    BSN_HIP(hipGetLastError());
  }
  void round_W(int cb) override {
    if (cb <= 0) return;
    if (spec_rounded) {   // already queued behind the orthonormalisation (column by column: a clipped block is fine)
      spec_rounded = false;
      return;
    }
    Tick tk(this, 5);
    if (!dist) {
      round_cols(Wc, n, cb, mx_valid);
      mx_valid = false;
      return;
    }
    if (comm && op->slices <= 4 && nr % 4 == 0 && !getenv("BSN_NO_COMPACT_GATHER")) {
      // The block is rounded to 8 * slices <= 32 bits anyway: with the column maxima over ALL rows known first (an
      // all-gather of cb numbers), every rank rounds its own rows and the all-gather ships the integers — int16 up to
      // 16 bits (a quarter of the fp64 volume), int32 up to 32 (half) —, the same rounded values bit for bit (a zero
      // column keeps an unscaled zero: as in k_round_cols).  The integers travel as doubles: an all-gather only moves bytes.
      unsigned long long *mx = (unsigned long long *)dorth.p;
      BSN_HIP(hipMemsetAsync(mx, 0, (size_t)cb * 8, st));
      hipLaunchKernelGGL(k_col_absmax, dim3(64, cb), dim3(1024), 0, st, Wc, nr, nr, mx);
      // (the segments' receive buffer is free here; it also has to hold world * cb maxima when the sample blocks are short)
      double *gmax = ws.Wrecv.ensure(std::max((size_t)nr * kMaxB, (size_t)world * kMaxB));
      comm_all_gather(comm, (const double *)mx, gmax, cb, st);
      hipLaunchKernelGGL(k_max_over_ranks, dim3(1), dim3(64), 0, st, (const unsigned long long *)gmax, world, cb, mx);
      const dim3 gp((unsigned)((nr + 255) / 256), cb), gu((unsigned)((n * cb + 255) / 256));
      if (op->slices <= 2) {
        int16_t *q_send = (int16_t *)Wfull.p, *q_all = (int16_t *)Wblk.p;
        hipLaunchKernelGGL((k_pack_int<int16_t>), gp, dim3(256), 0, st, Wc, nr, mx, op->slices, q_send);
        BSN_HIP(hipGetLastError());
        comm_all_gather(comm, (const double *)q_send, (double *)q_all, nr * cb / 4, st);
        hipLaunchKernelGGL((k_unpack_int<int16_t>), gu, dim3(256), 0, st, q_all, n, cb, nr, mx, op->slices, Qfull.p);
      } else {
        int32_t *q_send = (int32_t *)Wfull.p, *q_all = (int32_t *)Wblk.p;
        hipLaunchKernelGGL((k_pack_int<int32_t>), gp, dim3(256), 0, st, Wc, nr, mx, op->slices, q_send);
        BSN_HIP(hipGetLastError());
        comm_all_gather(comm, (const double *)q_send, (double *)q_all, nr * cb / 2, st);
        hipLaunchKernelGGL((k_unpack_int<int32_t>), gu, dim3(256), 0, st, q_all, n, cb, nr, mx, op->slices, Qfull.p);
      }
      BSN_HIP(hipGetLastError());
      n_compact_gathers++;
      return;
    }
    // every rank rounds the same gathered block (column maxima over all n rows), then keeps its rows
    all_gather_rows(Wc, cb, Qfull.p);
    round_cols(Qfull.p, n, cb, false);
    hipLaunchKernelGGL(k_take_rows, dim3((unsigned)((nr * cb + 255) / 256)), dim3(256), 0, st, Qfull.p, n, cb, nr,
                       row0, Wc);
    BSN_HIP(hipGetLastError());
  }
  void gram_to_host(const double *A, const double *B, int64_t rows, int p, int cb, double *out) {
    Tick tk(this, 3);
    gemm_tn_any(A, B, rows, p, cb, dsmall.p);
    BSN_HIP(hipGetLastError());
    ar_small(dsmall.p, (int64_t)p * cb);
    double *hp = ws.pinned((size_t)(cap + kMaxB + 4) * kMaxB * 4 + 1024);
    BSN_HIP(hipMemcpyAsync(hp, dsmall.p, (size_t)p * cb * 8, hipMemcpyDeviceToHost, st));
    sync_stream();
    std::memcpy(out, hp, (size_t)p * cb * 8);
    op_poll_stats(op);
  }
  void ZtZ(int p, int p0, int cb, double *G) override {
    gram_to_host(Z.p, Z.p + (int64_t)p0 * m_local, m_local, p, cb, G);
  }
  void QtQ(int p, int p0, int cb, double *M) override {
    gram_to_host(Q.p, Q.p + (int64_t)p0 * nr, nr, p, cb, M);
  }

  // ---- the fused block step (orth_small.hpp) -----------------------------------------------------
  // arena (doubles): [mx 16] [flag 1 | Rout B2 | ZtZ p cb | X = (Q'Qb over W'Qb | HG1) (p+cb) 2cb] [HG2 (p+cb) cb]
  // [C p cb] [Ct p cb] [Ri B2]; [flag .. left half of X] is downloaded in one piece, [ZtZ .. X] is one sum over
  // the ranks
  static constexpr int B2 = kMaxB * kMaxB;
  void update_W(int p, int cb, const double *C, const double *Ri, unsigned long long *mx) {
    const dim3 rows((unsigned)((nr + 1023) / 1024));
    if (cb <= 8)
      hipLaunchKernelGGL((k_update<8>), rows, dim3(256), 0, st, Q.p, nr, p, C, Ri, cb, Wc, nr, nr, mx);
    else
      hipLaunchKernelGGL((k_update<16>), rows, dim3(256), 0, st, Q.p, nr, p, C, Ri, cb, Wc, nr, nr, mx);
  }
  // Gram blocks of the newest basis block + orthonormalisation of W against Q[:, :p] and itself, queued
  // without returning to the host; ONE synchronisation.  with_grams = false: orthonormalisation only
  // (start block, p == 0).  Returns cb, -1 (not supported: the caller takes the step-by-step path
  // for everything) or -2 (Gram blocks delivered, W restored: the panel needs the careful path).
  int fused(int p, int p0, int cb, bool with_grams, double *blkZ, double *blkQ, std::vector<double> &Rout) {
    if (cb <= 0 || cb > kMaxB || p + cb > kOrthMaxP || p + cb > cap + kMaxB || no_fused) return -1;
    if (p > 0 && Wc != Q.p + (int64_t)p * nr) return -1;
    if (p > 0 && !with_grams) return -1;   // the device copy of Q'Q is only kept by the full step
    const double *A0 = p > 0 ? Q.p : Wc;    // [Q W] (p + cb contiguous columns); the panel alone for p == 0
    Tick tk(this, 4);
    hipEvent_t tev0 = nullptr, tev1 = nullptr;
    if (timing) {
      BSN_HIP(hipEventCreate(&tev0));
      BSN_HIP(hipEventCreate(&tev1));
      BSN_HIP(hipEventRecord(tev0, st));
    }
    unsigned long long *mx = (unsigned long long *)dorth.p;
    // X = [Q W]' [Qb W] ((p + cb) x 2 cb, leading dimension p + cb; Qb = the newest basis block, columns
    // p0 .. p-1 of Q, and W right behind it: contiguous columns): its left half holds the Gram block Q'Qb the
    // Rayleigh-Ritz step needs, its right half [H; G] of pass 0 — ONE pass over Q for both
    const size_t pc = (size_t)p * cb, hg = (size_t)(p + cb) * cb;
    double *flag = dorth.p + 16, *dRout = flag + 1, *dZtZ = dRout + B2, *X = dZtZ + pc, *HG1 = X + (p > 0 ? hg : 0),
           *HG2 = HG1 + hg, *C = HG2 + hg, *Ct = C + pc, *Ri = Ct + pc;
    BSN_HIP(hipMemsetAsync(flag, 0, (size_t)(1 + B2) * 8, st));
    if (p > 0) gemm_tn_any(Z.p, Z.p + (int64_t)p0 * m_local, m_local, p, cb, dZtZ);
    BSN_HIP(hipMemcpyAsync(Wsave.p, Wc, (size_t)nr * cb * 8, hipMemcpyDeviceToDevice, st));
    OrthSmall a;
    a.p = p;
    a.cb = cb;
    a.p0 = p0;
    a.QtQ = p > 0 ? X : nullptr;
    a.ldq = p + cb;
    a.M = ws.dM.p;
    a.ldm = kOrthMaxP;
    a.C = C;
    a.Ct = Ct;
    a.Ri = Ri;
    a.Rout = dRout;
    a.flag = flag;
    a.iters = std::min(min_slices, op->slices) >= 2 ? 2 : 4;   // |Q'Q - I| ~ 2^-8S (coarsest block): its cube (fifth power) is below rounding
    a.Cs = a.Gs = a.Rs = a.Ris = a.Ro = a.Dv = a.tmp = nullptr;
    for (int pass = 0; pass < 2; pass++) {
      double *HG = pass == 0 ? HG1 : HG2;
      if (pass == 0 && p > 0) {
        gemm_tn_any(Q.p, Q.p + (int64_t)p0 * nr, nr, p + cb, 2 * cb, X, p + cb);
        ar_small(dZtZ, (int64_t)(pc + 2 * hg));
      } else {
        gemm_tn_any(A0, Wc, nr, p + cb, cb, HG, p + cb);   // [Q W]' W: W is columns p .. p+cb-1 of Q
        ar_small(HG, (int64_t)hg);
      }
      a.pass = pass;
      a.HG = HG;
      // the column maxima of the finished panel feed the rounding; with sample blocks they would have to be
      // combined over the ranks, so the distributed solve takes them from the gathered block instead
      unsigned long long *mxp = (pass == 1 && !dist) ? mx : nullptr;
      hipLaunchKernelGGL(k_orth2, dim3(1), dim3(512), 0, st, a, mxp);
      update_W(p, cb, C, Ri, mxp);
    }
    BSN_HIP(hipGetLastError());
    const size_t nsmall = 1 + B2 + pc + (p > 0 ? hg : 0);
    double *hp = ws.pinned((size_t)(cap + kMaxB + 4) * kMaxB * 4 + 1024);
    BSN_HIP(hipMemcpyAsync(hp, flag, nsmall * 8, hipMemcpyDeviceToHost, st));
    if (timing) BSN_HIP(hipEventRecord(tev1, st));
    if (with_grams && speculate && !hold_round) {
      // The host now waits for the small matrices and then runs the Rayleigh-Ritz step; the device would idle
      // through both.  What the NEXT step starts with — rounding the new block and quantising it for the
      // crossproduct pass — depends on neither (unless the solve ends here or the panel turns out rank
      // deficient: then the two small kernels ran for nothing), so it is queued behind the copy and the host
      // waits for the copy alone.
      if (!ws.ev_small) BSN_HIP(hipEventCreateWithFlags(&ws.ev_small, hipEventDisableTiming));
      BSN_HIP(hipEventRecord(ws.ev_small, st));
      mx_valid = !dist;
      spec_rounded = false;
      round_W(cb);
      if (!ooc) op_cprod_prequant(op, dist ? Qfull.p : Wc, n, cb);
      spec_rounded = true;
      BSN_HIP(hipEventSynchronize(ws.ev_small));
      n_sync++;
    } else {
      sync_stream();
    }
    if (timing) {
      float ms = 0;
      BSN_HIP(hipEventElapsedTime(&ms, tev0, tev1));
      t_phase[7] += ms;
      (void)hipEventDestroy(tev0);
      (void)hipEventDestroy(tev1);
    }
    op_poll_stats(op);
    if (blkZ) std::memcpy(blkZ, hp + 1 + B2, pc * 8);
    if (blkQ)   // the first p rows of the left half of X (leading dimension p + cb)
      for (int j = 0; j < cb; j++) std::memcpy(blkQ + (size_t)j * p, hp + 1 + B2 + pc + (size_t)j * (p + cb), (size_t)p * 8);
    if (hp[0] != 0.0) {  // rank deficient or ill conditioned: undo, the driver takes the careful path
      BSN_HIP(hipMemcpyAsync(Wc, Wsave.p, (size_t)nr * cb * 8, hipMemcpyDeviceToDevice, st));
      mx_valid = false;
      spec_rounded = false;
      op->preq_X = nullptr;
      return -2;
    }
    Rout.assign((size_t)cb * cb, 0.0);
    for (int t = 0; t < cb * cb; t++) Rout[(size_t)t] = hp[1 + (size_t)t];
    mx_valid = spec_rounded ? false : !dist;
    return cb;
  }
  int step_fused(int p, int p0, int cb, double *blkZ, double *blkQ, std::vector<double> &Rout) override {
    return fused(p, p0, cb, true, blkZ, blkQ, Rout);
  }
  int orth_fused(int p, int cb, std::vector<double> &Cacc, std::vector<double> &Rout) override {
    if (p != 0) return -1;
    Cacc.clear();
    const int r = fused(0, 0, cb, false, nullptr, nullptr, Rout);
    return r < 0 ? -1 : r;
  }
  void QtW(int p, int cb, double *C) override { gemm_tn(Q.p, p, cb, C); }
  void WtW(int cb, double *G) override { gemm_tn(Wc, cb, cb, G); }
  void W_minus_QC(int p, int cb, const double *C) override {
    mx_valid = false;
    copy_h2d(op->bed, dsmall.p, C, (size_t)p * cb * 8);
    hipLaunchKernelGGL(k_gemm_nn, dim3((unsigned)((nr + 255) / 256)), dim3(256), kTP * kMaxB * 8, st,
                       Q.p, nr, p, dsmall.p, cb, Wc, nr, 1.0, -1.0, Wc, nr, nr);
    BSN_HIP(hipGetLastError());
    sync_stream();  // C is a host vector that may be reused
  }
  void W_times(int cb, int r, const double *M) override {
    mx_valid = false;
    copy_h2d(op->bed, dsmall.p, M, (size_t)cb * r * 8);
    hipLaunchKernelGGL(k_right_mult, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, Wc, nr, nr,
                       cb, r, dsmall.p);
    BSN_HIP(hipGetLastError());
    sync_stream();
  }
  void W_to_Q(int p0, int r) override {
    double *dst = Q.p + (int64_t)p0 * nr;
    if (dst != Wc)
      BSN_HIP(hipMemcpyAsync(dst, Wc, (size_t)nr * r * 8, hipMemcpyDeviceToDevice, st));
    wcol = p0 + r;
    Wc = Q.p + (int64_t)wcol * nr;   // the next panel goes behind the block just stored
  }
  // thick restart: Q[:, :keep] = Q[:, :pp] S and Z[:, :keep] = Z[:, :pp] S (local rows of both), the waiting
  // panel moves behind column `keep`, the device copy of Q'Q becomes the identity on the kept part
  bool restart(int pp, int keep, const double *S, int rn, const double *Mk) override {
    if (keep + rn > pp || keep <= 0) return false;   // the panel moves to the left of where it is
    DevBuf<double> &dS = ws.dS, &tQ = ws.dU, &tZ = ws.dV;
    dS.ensure((size_t)pp * keep + 16);
    tQ.ensure((size_t)nr * keep);
    tZ.ensure((size_t)m_local * keep);
    copy_h2d(op->bed, dS.p, S, (size_t)pp * keep * 8);
    for (int c0 = 0; c0 < keep; c0 += kMaxB) {
      const int nc = keep - c0 < kMaxB ? keep - c0 : kMaxB;
      hipLaunchKernelGGL(k_gemm_nn, dim3((unsigned)((nr + 255) / 256)), dim3(256), kTP * kMaxB * 8, st, Q.p, nr, pp,
                         dS.p + (size_t)c0 * pp, nc, (const double *)nullptr, (int64_t)0, 0.0, 1.0,
                         tQ.p + (int64_t)c0 * nr, nr, nr);
      hipLaunchKernelGGL(k_gemm_nn, dim3((unsigned)((m_local + 255) / 256)), dim3(256), kTP * kMaxB * 8, st, Z.p,
                         m_local, pp, dS.p + (size_t)c0 * pp, nc, (const double *)nullptr, (int64_t)0, 0.0, 1.0,
                         tZ.p + (int64_t)c0 * m_local, m_local, m_local);
    }
    BSN_HIP(hipGetLastError());
    BSN_HIP(hipMemcpyAsync(Q.p, tQ.p, (size_t)nr * keep * 8, hipMemcpyDeviceToDevice, st));
    BSN_HIP(hipMemcpyAsync(Z.p, tZ.p, (size_t)m_local * keep * 8, hipMemcpyDeviceToDevice, st));
    double *dst = Q.p + (int64_t)keep * nr;
    BSN_HIP(hipMemcpyAsync(dst, Wc, (size_t)nr * rn * 8, hipMemcpyDeviceToDevice, st));
    Wc = dst;
    wcol = keep;
    // the device copy of Q'Q on the kept part: S'(Q'Q)S as the driver computed it (the identity up to rounding
    // unless the old basis was ill conditioned)
    copy_h2d(op->bed, dS.p, Mk, (size_t)keep * keep * 8);
    hipLaunchKernelGGL(k_put_block, dim3((unsigned)((keep * keep + 255) / 256)), dim3(256), 0, st, ws.dM.p,
                       kOrthMaxP, keep, dS.p);
    BSN_HIP(hipGetLastError());
    return true;
  }
  void finalize(int pp, int k, const double *S, const double *dinv, double *u, double *v) override {
    Tick tk(this, 6);
    std::vector<double> Sv((size_t)pp * k);
    for (int t = 0; t < k; t++)
      for (int i = 0; i < pp; i++) Sv[(size_t)i + (size_t)t * pp] = S[(size_t)i + (size_t)t * pp] * dinv[t];
    DevBuf<double> &dS = ws.dS, &dU = ws.dU, &dV = ws.dV, &dUfull = ws.dUfull;
    dS.ensure((size_t)pp * k * 2 + 16);
    dU.ensure((size_t)nr * k);
    dV.ensure((size_t)m_local * k);
    copy_h2d(op->bed, dS.p, S, (size_t)pp * k * 8);
    copy_h2d(op->bed, dS.p + (size_t)pp * k, Sv.data(), (size_t)pp * k * 8);
    for (int c0 = 0; c0 < k && pp > 0; c0 += kMaxB) {
      int nc = k - c0 < kMaxB ? k - c0 : kMaxB;
      hipLaunchKernelGGL(k_gemm_nn, dim3((unsigned)((nr + 255) / 256)), dim3(256), kTP * kMaxB * 8, st,
                         Q.p, nr, pp, dS.p + (size_t)c0 * pp, nc, (const double *)nullptr, (int64_t)0,
                         0.0, 1.0, dU.p + (int64_t)c0 * nr, nr, nr);
      hipLaunchKernelGGL(k_gemm_nn, dim3((unsigned)((m_local + 255) / 256)), dim3(256),
                         kTP * kMaxB * 8, st, Z.p, m_local, pp, dS.p + (size_t)pp * k + (size_t)c0 * pp,
                         nc, (const double *)nullptr, (int64_t)0, 0.0, 1.0,
                         dV.p + (int64_t)c0 * m_local, m_local, m_local);
    }
    BSN_HIP(hipGetLastError());
    if (pp == 0) {
      BSN_HIP(hipMemsetAsync(dU.p, 0, (size_t)nr * k * 8, st));
      BSN_HIP(hipMemsetAsync(dV.p, 0, (size_t)m_local * k * 8, st));
    }
    const double *ufull = dU.p;
    if (dist) {  // every rank returns all n rows of u
      dUfull.ensure((size_t)n * k);
      all_gather_rows(dU.p, k, dUfull.p);
      ufull = dUfull.p;
    }
    if (u) copy_d2h(op->bed, u, ufull, (size_t)n * k * 8);
    if (v) copy_d2h(op->bed, v, dV.p, (size_t)m_local * k * 8);
    sync_stream();
  }
};

}  // namespace bsn

using namespace bsn;

// Watchdog of a sharded solve (bsn_svd_options::exchange_timeout_ms): when the solve has not returned in time the
// communicator is aborted — the collectives in flight end, the next one fails, the call returns an error instead
// of hanging in a stream synchronisation for ever.  If even that does not bring the call back (a transport without
// ncclCommAbort, a wedged device) the process ends with status 86 after a grace period: a run that cannot finish
// must not outlive its driver's patience either.
struct ExchangeWatchdog {
  bsn_comm *c;
  int ms;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
  std::chrono::steady_clock::time_point deadline;
  std::atomic<int> fired{0};
  ExchangeWatchdog(bsn_comm *c_, int ms_) : c(c_), ms(ms_) {
    if (!c || ms <= 0) return;
    deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(ms);
    th = std::thread([this] {
      std::unique_lock<std::mutex> lk(mu);
      // the deadline moves with the solve's progress (kick): wait until it has really passed
      for (;;) {
        const auto dl = deadline;
        if (cv.wait_until(lk, dl, [this] { return done; })) return;
        if (std::chrono::steady_clock::now() >= deadline) break;
      }
      if (done) return;   // (checked under the lock: a solve that returned while the deadline passed keeps its communicator)
      fired.store(1);
      lk.unlock();
      const bool could = comm_abort(c);
      std::fprintf(stderr, "[bsn svd] rank %d: the sharded solve made no progress for %d ms: communicator %s\n", c->rank, ms,
                   could ? "aborted" : "cannot be aborted (no ncclCommAbort)");
      lk.lock();
      if (cv.wait_for(lk, std::chrono::seconds(20), [this] { return done; })) return;
      // Still blocked 20 s after the abort (a transport without ncclCommAbort, a wedged device).  Ending the host process
      // — R, Python — from inside a library is the caller's decision: BSN_WATCHDOG_EXIT=1 (bench.py sets it: a run that
      // cannot finish must not outlive its driver's patience) asks for exit status 86; otherwise the call stays blocked
      // and says so.
      const char *ex = getenv("BSN_WATCHDOG_EXIT");
      if (ex && atoi(ex) != 0) {
        std::fprintf(stderr, "[bsn svd] rank %d: still blocked 20 s after the abort: giving up (exit status 86, BSN_WATCHDOG_EXIT)\n", c->rank);
        std::fflush(stderr);
        _exit(86);
      }
      std::fprintf(stderr, "[bsn svd] rank %d: still blocked 20 s after the abort (BSN_WATCHDOG_EXIT=1 would end the process here)\n", c->rank);
    });
  }
  // progress of the solve (a block step's pass has been queued and the previous one synchronised): the deadline is per
  // step, not per solve — a legitimately long sharded solve is not aborted
  void kick() {
    if (!th.joinable()) return;
    std::lock_guard<std::mutex> lk(mu);
    deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(ms);
  }
  ~ExchangeWatchdog() {
    if (!th.joinable()) return;
    {
      std::lock_guard<std::mutex> lk(mu);
      done = true;
    }
    cv.notify_all();
    th.join();
  }
};

// bed_scaleBinom from host code counts (the path of row subsets, where the statistics cannot ride
// along a crossproduct pass over all samples): R/binom-scaling.R:133-142
static void binom_scale_host(const std::vector<int32_t> &cnt, int64_t n, int64_t m, std::vector<double> &center,
                             std::vector<double> &scale, int32_t *n_bad) {
  center.resize((size_t)m);
  scale.resize((size_t)m);
  int32_t bad = 0;
  for (int64_t j = 0; j < m; j++) {
    const int32_t *c = &cnt[(size_t)4 * j];
    const double sumX = (double)(c[1] + 2 * c[2]), nona = (double)(c[0] + c[1] + c[2]);
    const double af = sumX / (2.0 * nona);
    center[(size_t)j] = 2.0 * af;
    scale[(size_t)j] = std::sqrt(2.0 * af * (1.0 - af));
    if (2 * (int64_t)(c[0] + c[1] + c[2]) < n) bad++;
  }
  *n_bad = bad;
}

// A solve over a list of variants that is not a contiguous range runs on a compacted copy of the selection
// (bsn_bed::sub, bsn_internal.hpp).  Returns the copy, or nullptr when the solve should go through the gather lists:
// contiguous columns, a small selection (BSN_COMPACT_MIN_BYTES, default 256 MB of 2-bit payload: below that a solve
// is milliseconds either way), no room for the copy and its sample-major twin, BSN_NO_COMPACT=1, a byte / look-up image.
static bsn_bed *compacted_view(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m) {
  if (!ind_col || m < 2 || bed->bits != 2 || bed->generic || bed->streamed() || getenv("BSN_NO_COMPACT")) return nullptr;
  bool contig = true;
  for (int64_t j = 1; j < m && contig; j++) contig = ind_col[j] == ind_col[0] + j;
  if (contig) return nullptr;
  const int64_t pitch = round_up((n + 3) / 4, 256);
  const double bytes = (double)(m + 64) * (double)pitch;
  double min_bytes = 256e6;
  if (const char *e = getenv("BSN_COMPACT_MIN_BYTES")) min_bytes = atof(e);
  if (bytes < min_bytes) return nullptr;
  // FNV-1a over the lists: the same selection again (repeated solves, bench.py) reuses the copy as it is
  uint64_t key = 1469598103934665603ull;
  auto mix = [&](const int64_t *p, size_t count) {   // (one multiply per index: a list of a million variants in a millisecond)
    for (size_t i = 0; i < count; i++) {
      key = (key ^ (uint64_t)p[i]) * 1099511628211ull;
      key ^= key >> 29;
    }
  };
  mix(&n, 1);
  mix(&m, 1);
  mix(ind_col, (size_t)m);
  bool rows_ident = n == bed->n;
  if (ind_row)
    for (int64_t i = 0; rows_ident && i < n; i++) rows_ident = ind_row[i] == i;
  if (!rows_ident && ind_row) mix(ind_row, (size_t)n);
  if (key == 0) key = 1;
  // the same selection again: the key AND the lists themselves (8 m bytes kept on the handle; a hash can collide)
  const bool keep_rows = !rows_ident && ind_row;
  if (bed->sub && bed->sub_key == key && bed->sub->n == n && bed->sub->m == m && (int64_t)bed->sub_cols.size() == m &&
      std::memcmp(bed->sub_cols.data(), ind_col, (size_t)m * 8) == 0 &&
      (keep_rows ? ((int64_t)bed->sub_rows.size() == n && std::memcmp(bed->sub_rows.data(), ind_row, (size_t)n * 8) == 0)
                 : bed->sub_rows.empty()))
    return bed->sub;
  BSN_HIP(hipSetDevice(bed->device));
  const bool in_place = bed->sub && bed->sub->n == n && bed->sub->cap_m >= m;
  if (!in_place) {
    if (bed->sub) {   // another shape: its memory goes back first
      bed_free(bed->sub);
      bed->sub = nullptr;
      bed->sub_key = 0;
      bed->sub_cols.clear();
      bed->sub_rows.clear();
    }
    size_t free_b = 0, total_b = 0;
    BSN_HIP(hipMemGetInfo(&free_b, &total_b));
    // the copy, its sample-major twin, the workspace of a solve (basis and panels: ~ 3 KB per sample + per variant)
    if ((double)(free_b + dev_cache_held()) < 2.0 * bytes + 3000.0 * (double)(n + m) + 2e9) return nullptr;
  }
  bsn_bed *fresh = nullptr;
  try {
    fresh = image_gather(bed, rows_ident ? nullptr : ind_row, n, ind_col, m, in_place ? bed->sub : nullptr);
  } catch (const std::exception &) {
    if (in_place) {   // the old copy is half overwritten: it answers to no list any more
      bed->sub_key = 0;
      bed->sub_cols.clear();
      bed->sub_rows.clear();
      throw;
    }
    (void)hipGetLastError();
    return nullptr;   // (no room after all)
  }
  bed->sub = fresh;
  bed->sub_key = key;
  bed->sub_cols.assign(ind_col, ind_col + m);
  if (keep_rows) bed->sub_rows.assign(ind_row, ind_row + n);
  else bed->sub_rows.clear();
  return fresh;
}

extern "C" int bsn_bed_randomsvd(bsn_bed *bed, const int64_t *ind_row, int64_t n,
                                 const int64_t *ind_col, int64_t m, const double *center,
                                 const double *scale, const bsn_svd_options *o, double *d, double *u,
                                 double *v, bsn_svd_info *info) {
  bool unconverged = false;
  int rc = guarded([&] {
    if (!o) fail("options must not be NULL");
    if (o->k < 1) fail("'k' must be at least 1.");
    RoctxRange whole("bsn_bed_randomsvd");
    auto t_begin = std::chrono::steady_clock::now();
    auto since = [&]() {
      return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    };
    // (round 5) an out-of-core handle: the passes of the solve walk the file in slabs (HipSvdBackend, "out-of-core handle")
    const bool ooc = bed->streamed();
    if (ooc) {
      if (o->comm || o->allreduce) fail("an out-of-core handle cannot be one shard of a sharded solve: shard the file instead");
      bool all_rows = n == bed->n;
      for (int64_t i = 0; all_rows && ind_row && i < n; i++) all_rows = ind_row[i] == i;
      bool all_cols = m == bed->m, in_order = true;
      for (int64_t j = 0; ind_col && j < m; j++) {
        if (ind_col[j] != j) all_cols = false;
        if (ind_col[j] < 0 || ind_col[j] >= bed->m) fail("Tested %lld < %lld. Subscript out of bounds (ind.col).", (long long)ind_col[j], (long long)bed->m);
        if (j > 0 && ind_col[j] <= ind_col[j - 1]) in_order = false;
      }
      // (round 6) a LIST of variants in file order is served too — what bed_autoSVD solves over (ind.keep, sorted)
      if (!all_rows || !in_order)
        fail("bed_randomSVD on an out-of-core handle (the file's image does not fit the device: it streams its file) "
             "covers all samples, and its variants in increasing file order; a subset small enough for the device is served "
             "by a handle of its own (snp_subset / bsn_bed_subset_payload)");
      ind_row = nullptr;
      if (all_cols) ind_col = nullptr;
    }
    // (round 5) a list of variants that is not a contiguous range: the solve runs on a compacted copy of the selection
    bool compacted = false;
    double t_compact = 0.0;
    bed->last_solve_on_sub = false;
    if (bsn_bed *sub = compacted_view(bed, ind_row, n, ind_col, m)) {
      bed->last_solve_on_sub = true;
      if (ind_row == nullptr && n != bed->n) fail("internal: row count of a compacted solve");
      bed = sub;
      ind_row = nullptr;
      ind_col = nullptr;
      compacted = true;
      t_compact = since();
    }
    // operator and workspace of the previous solve on this handle are reused (grow-only buffers)
    struct Lend {
      bsn_bed *bed;
      std::unique_ptr<bsn_op> op;
      ~Lend() { bed->svd_op = std::move(op); }
    } lend{bed, std::move(bed->svd_op)};
    if (!lend.op) lend.op.reset(new bsn_op());
    bsn_op *op = lend.op.get();
    op->passes = 0;
    op->prof_kind_override = -1;
    op->stats_pending = false;
    op->na_poll = false;
    op->no_na = false;
    if (!bed->svd_ws) bed->svd_ws = std::make_shared<SvdWorkspace>();
    int32_t n_bad = 0;
    bool fused = false;
    if (o->binom_scaling) {
      fill_op(op, bed, ind_row, n, ind_col, m, nullptr, nullptr, true, ooc);
      if (op->rows_identity) {
        // the counts ride along the first crossproduct pass; until they are known the general kernels run
        op->stats_pending = true;
        op->no_na = false;
        fused = true;
      } else {
        std::vector<int32_t> cnt((size_t)4 * m);
        counts_host(bed, ind_row, n, ind_col, m, cnt.data());
        std::vector<double> ce, sc;
        binom_scale_host(cnt, n, m, ce, sc, &n_bad);
        copy_h2d(bed, op->d_center.p, ce.data(), (size_t)m * 8);
        copy_h2d(bed, op->d_scale.p, sc.data(), (size_t)m * 8);
        if (o->center_out) std::copy(ce.begin(), ce.end(), o->center_out);
        if (o->scale_out) std::copy(sc.begin(), sc.end(), o->scale_out);
      }
    } else {
      fill_op(op, bed, ind_row, n, ind_col, m, center, scale, false, ooc);
    }
    const double t_create = since();
    op->profile = true;
    if (o->slices > 7) fail("slices must be in 1..7");
    HipSvdBackend bk(*bed->svd_ws);
    bk.op = op;
    bk.ooc = ooc;
    bk.ooc_cols = ooc ? ind_col : nullptr;
    bk.st = bed->stream;
    bk.n = n;
    bk.m_local = m;
    bk.m_total = o->m_total > 0 ? o->m_total : m;
    bk.hook = o->comm ? nullptr : o->allreduce;
    bk.ctx = o->allreduce_ctx;
    bk.comm = o->comm;
    bk.rank = o->hook_rank;
    bk.world = o->hook_world > 0 ? o->hook_world : 1;
    bk.kmax = o->k;
    if (bk.comm && bk.comm->device != bed->device) fail("the communicator was created on another device");
    if (bk.hook && (bk.rank < 0 || bk.rank >= bk.world)) fail("hook_rank %d of %d", bk.rank, bk.world);
    bk.setup_ranks();
    bk.timing = getenv("BSN_TIMING") != nullptr;
    bk.speculate = getenv("BSN_NO_SPECULATION") == nullptr;
    bk.no_fused = getenv("BSN_NO_FUSED_STEP") != nullptr;
    bk.fused_stats = fused;
    bk.warm_den = o->warm_denominator >= 2 ? o->warm_denominator : 16;
    int64_t dim = bk.n < bk.m_total ? bk.n : bk.m_total;
    if (o->k > dim) fail("'k' is larger than the dimensions of the matrix.");
    SvdOptions so;
    so.k = o->k;
    so.tol = o->tol > 0 ? o->tol : 1e-4;
    // Digit slices per fp64 value and vectors per pass.  Rounding a basis block to 8 S bits leaves a
    // relative residual of about 1.2 * 2^(-8 S) on a converged pair (the Ritz VALUES are not
    // affected, svd_driver.hpp), so S is the smallest width whose floor is below tol / 4; the
    // block then fills the 16 MFMA columns of one column block (8 x 2, 5 x 3, 4 x 4, 3 x 5, 2 x 7).
    // A block chosen by the caller gets as many slices as its column blocks hold anyway.
    {
      int s_tol = 2;
      while (s_tol < 7 && 1.2 * std::ldexp(1.0, -8 * s_tol) > so.tol / 4) s_tol++;
      if (o->slices > 0) s_tol = o->slices;
      // One column block (16 MFMA columns) holds b1 vectors, two hold b2.  A pass with two column blocks costs
      // 1.3x a pass with one (the two-block kernels are bound by the matrix pipe and the VALU issue port, not by
      // HBM), but the solve needs fewer block steps: measured at 400K x 1M, tol 1e-4 (profiles/r03_block_vs_k.txt):
      // k = 20: 4 steps of 16 vectors = 8.1 passes, 204 ms, against 6 steps of 8 = 12.1 passes, 219 ms;
      // k <= 10: 5 steps of 8 = 193 ms against 4 steps of 16 = 208 ms.  The default is chosen for the time to
      // the solution, not for pass throughput.
      const int b1 = std::max(1, std::min(8, 16 / s_tol)), b2 = std::max(b1, std::min(kMaxB, 32 / s_tol));
      // (Without the warm start — fewer than 262 144 variants over all ranks — the wide block saves one step of
      // seven instead of two of six and loses: 50 against 35 ms on a 125 000-variant matrix.)
      const bool warm_on = o->warm_start >= 0 && bk.m_total / bk.warm_den >= 16384;
      int bb = o->block > 0 ? (o->block > kMaxB ? kMaxB : o->block) : (4 * o->k >= 7 * b1 && warm_on ? b2 : b1);
      int ss = s_tol;
      if (o->slices <= 0) {
        const int nb = (bb * s_tol + 15) / 16;
        ss = std::max(s_tol, std::min(7, 16 * nb / bb));
      }
      so.block = bb;
      op->slices = ss;
      // precision schedule (svd_driver.hpp): wider panels for the early steps when the vectors are wanted beyond the floor
      // of `ss` digits; a caller who fixed the digits gets them at every step unless he also names a floor
      double vf = o->vec_floor;
      // (round 6: 1e-7, a tenth of north_star's 1e-6.  Round 5's 2.5e-7 left the vectors the ARITHMETIC limits — small matrices
      // that take 20 - 30 block steps, where the leading pairs converge far below tol — at 1.5 - 1.8e-6 on the oracle's
      // shapes; 1e-7 keeps the panels at 24 bits one or two steps longer there (0.5 - 0.97e-6: what uniform 24-bit panels
      // give) and leaves the 400K x 1M solve on the same three wide steps, its vectors being where the Lanczos process
      // is at tol: profiles/r06_vectors_small.txt, r06_vectors_c3.txt)
      if (vf == 0.0) vf = o->slices > 0 ? -1.0 : 1e-7;
      if (const char *e = getenv("BSN_VEC_FLOOR")) vf = atof(e);   // (A/B and the accuracy sweeps of the tests)
      so.slices_base = ss;
      so.slices_max = ss;
      so.vec_floor = 0.0;
      if (vf > 0) {
        int sm = ss;
        while (sm < 7 && 1.2 * std::ldexp(1.0, -8 * sm) > vf) sm++;
        so.slices_max = sm;
        so.vec_floor = vf;
      }
      // the START block only chooses where the iteration begins: an 8-bit grid serves, and the first full crossproduct
      // pass — the one that also counts the codes — then carries `block` digit columns instead of twice as many
      // (16 vectors: one column block, 21 instead of 26 ms per 100 GB; same residuals, same vectors)
      so.slices_start = o->slices > 0 ? 0 : 1;
      if (const char *e = abl_getenv("BSN_START_SLICES")) so.slices_start = atoi(e);
      // columns standardised by bed_scaleBinom: a random vector is amplified by sqrt(n) (svd_driver.hpp)
      // (the product pass scheduled apart from the grids, svd_driver.hpp: measured at 400K x 1M — 12 ms saved, the
      // leading vectors at 9e-7 instead of 1.6e-7; at k = 10, 6e-6: the rounding of Z is heavier-tailed than the model,
      // so the split stays an experiment, BSN_ZQ_SPLIT=1)
      so.noise_gain = (fused && abl_getenv("BSN_ZQ_SPLIT")) ? std::sqrt((double)n) : 0.0;
    }
    // a solve streams the image a dozen times: the ONE-block kernels, which are bound by HBM, get their layout (a
    // second copy in 64-variant x 256-B tiles, one extra pass of copying, kept on the handle) when the device has the
    // room: 2 - 4 % per pass.  The two-block kernels are bound by instruction issue and gain nothing from it (k_prod<2>
    // 23.65 ms on the plain image against 23.75 on the copy, k_cprod<2> 21.65 against 21.52: profiles/r03_shape_sweeps.txt),
    // so the default solve at k >= 14 leaves the other half of the HBM alone; a copy that exists is used either way.
    // ... and a solve with passes of two or three column blocks (16 vectors; 8 vectors on the 24-bit panels of the
    // precision schedule) the sample-major copy: its product passes then run as k_prodT (k_cprod's shape; DESIGN.md
    // 3.3b) instead of k_prod<2> with its transposes and 128 accumulators — and three column blocks exist on that copy
    // only.  One second copy per handle: when the sample-major one is not to be had (no room, BSN_NO_SMAJ) the
    // one-block passes of the solve still get theirs.
    bool have_smaj = false;
    // (round 6) on one GPU the FIRST such solve of a handle runs without the copy — its product passes take k_prod<2>, the
    // same sums, 55 ms more at 400K x 1M — and the copy is made BEHIND it (image_smaj_start at the end of this call: a helper
    // thread allocates 100 GB and queues the transposition on a stream of its own while the caller looks at its result);
    // later solves find it.  Inside the call the copy cost 90 ms (40 of allocation on an idle device, 51 of transposition)
    // to save 55; allocated BESIDE the running solve the same hipMalloc took 1 - 2.5 s and held up every allocation of the
    // solve (profiles/r06_cold.txt).  A caller that solves once never pays for it.  A sharded solve builds it up front:
    // every rank must take the same exchange pattern, and which passes find the copy would differ between ranks.
    const bool wants_smaj = so.block * so.slices_max > 16 && op->cols_contig && (op->col0 & 511) == 0 && m >= 4096;
    bool smaj_after = false;
    if (wants_smaj) {
      if (bk.dist || getenv("BSN_SMAJ_SYNC")) have_smaj = image_smaj(bed);
      else if (bed->d_smaj || bed->smaj_job) have_smaj = image_smaj_poll(bed) || bed->smaj_job != nullptr;
      else have_smaj = smaj_after = !bed->smaj_tried && bed->bits == 2 && !getenv("BSN_NO_SMAJ");
    }
    if (!have_smaj && so.block * so.slices_base <= 16 && op->cols_contig && (op->col0 & 63) == 0 && m >= 4096) image_tile(bed);
    so.resid_floor = 1.2 * std::ldexp(1.0, -8 * op->slices);
    // (two iterations since round 5: + 2.5 ms, every residual of the 400K x 1M solve 4 - 10 x lower at the same step)
    so.warm = o->warm_start < 0 ? 0 : (o->warm_start == 0 ? 2 : o->warm_start);
    so.max_basis = o->max_basis;
    so.max_restarts = o->max_restarts == 0 ? 100 : o->max_restarts;
    so.seed = o->seed ? o->seed : 1;
    so.verbose = o->verbose;
    BSN_HIP(hipEventRecord(bed->ev0, bed->stream));
    int wd_ms = o->exchange_timeout_ms;
    if (wd_ms == 0)
      if (const char *e = getenv("BSN_EXCHANGE_TIMEOUT_MS")) wd_ms = atoi(e);
    ExchangeWatchdog watchdog(bk.comm, wd_ms);
    if (bk.comm && wd_ms > 0) bk.progress = [&watchdog] { watchdog.kick(); };
    if (bk.comm) {
      double junk_ms[kCommClasses];
      int junk_n[kCommClasses];
      bk.comm->timing = o->exchange_timing != 0;
      comm_time_collect(bk.comm, junk_ms, junk_n);   // (leftovers of a solve that failed)
    }
    struct TimingOff {
      bsn_comm *c;
      ~TimingOff() { if (c) c->timing = false; }
    } timing_off{bk.comm};
    SvdResult r;
    try {
      r = block_lanczos_svd(bk, so, d, u, v);
    } catch (const std::exception &ex) {
      {   // the timing events of the launches that did run are of no use to anybody
        double pms[kProfKinds];
        int pc[kProfKinds];
        prof_collect(op, pms, pc);
      }
      if (watchdog.fired.load())
        fail("the exchange of the sharded solve did not finish within %d ms (exchange: %s) and the communicator was aborted; "
             "what the solve saw: %s", wd_ms,
             bk.exchange_mode == 3 ? "segments, reduce-scatters on a second stream" : bk.exchange_mode == 2 ? "segments, one stream"
             : getenv("BSN_NO_SEGMENTS") ? "whole pass" : getenv("BSN_NO_OVERLAP") ? "segments, one stream" : "segments, reduce-scatters on a second stream",
             ex.what());
      throw;
    }
    // ... and (round 6) whenever requested triplets lie below what the rounded products resolve — sigma below
    // 1.5e-4 sigma_1 on 16-bit panels: the driver leaves them out of its convergence test (they cannot meet it there) and
    // says so; their VALUES need the wide products.  Rare: k beyond the numerical rank at 16 bits.
    if (((r.exhausted && r.exhausted_resid > 1e-9) || r.below_resolution > 0) && o->slices <= 0 && op->slices < 7) {
      // a Krylov space exhausted on rounded products (svd_driver.hpp): only matrices whose rank fits the basis get
      // here, so the second solve — 56-bit digits, at most four vectors per pass — is cheap, and it is run whenever
      // the coupling block is not negligible, met tolerance or not: a matrix this small deserves its exact values
      if (o->verbose)
        std::fprintf(stderr, "[bsn svd] %s on %d-bit products (relative residual %.3g, %d requested triplets below their resolution): "
                             "again with 56-bit products\n", r.exhausted ? "Krylov space exhausted" : "triplets below the products' resolution",
                     8 * op->slices, r.max_rel_resid, r.below_resolution);
      op->slices = 7;
      so.slices_base = so.slices_max = 7;
      so.vec_floor = 0.0;
      so.block = std::min(so.block, 32 / 7);
      so.resid_floor = 1.2 * std::ldexp(1.0, -8 * op->slices);
      const SvdResult r1 = r;
      r = block_lanczos_svd(bk, so, d, u, v);
      r.niter += r1.niter;
      r.nops += r1.nops;
      r.restarts += r1.restarts;
    }
    const double t_solve = since() - t_create;
    if (o->verbose > 1 || bk.timing)
      std::fprintf(stderr,
                   "[bsn svd] host wall ms: op_create %.2f, solve %.2f = alloc %.2f + A'Q %.2f + AZ %.2f + grams %.2f + "
                   "orth %.2f (GPU events %.2f) + round %.2f + finalize %.2f + host algebra %.2f\n",
                   t_create, t_solve, bk.t_phase[0], bk.t_phase[1], bk.t_phase[2], bk.t_phase[3], bk.t_phase[4],
                   bk.t_phase[7], bk.t_phase[5], bk.t_phase[6],
                   t_solve - (bk.t_phase[0] + bk.t_phase[1] + bk.t_phase[2] + bk.t_phase[3] + bk.t_phase[4] +
                              bk.t_phase[5] + bk.t_phase[6]));
    BSN_HIP(hipEventRecord(bed->ev1, bed->stream));
    BSN_HIP(hipEventSynchronize(bed->ev1));
    float ms = 0;
    BSN_HIP(hipEventElapsedTime(&ms, bed->ev0, bed->ev1));
    if (fused) {
      // by-products of the counting pass: the scaling actually used, the > 50 % missing count of
      // bed_colstats (src/bed-fun.cpp:40-41) and the per-variant completeness of the handle.  They
      // come back through the workspace's pinned staging buffer: blocking copies into pageable memory
      // were measured to leave the runtime with ~4 ms wake-up latencies on every later stream
      // synchronisation of the process (tools/gpu/r02_h.sh).
      SvdWorkspace &ws = *bed->svd_ws;
      const int64_t chunk = 1 << 21;  // int32 per staged piece (8 MB)
      int32_t *hp = (int32_t *)ws.pinned((size_t)chunk / 2);
      if ((int64_t)bed->na_cnt.size() != bed->m) bed->na_cnt.assign((size_t)bed->m, -1);
      for (int64_t j0 = 0; j0 < m; j0 += chunk) {
        const int64_t cnt = std::min<int64_t>(chunk, m - j0);
        BSN_HIP(hipMemcpyAsync(hp, op->d_na.p + j0, (size_t)cnt * 4, hipMemcpyDeviceToHost, bed->stream));
        BSN_HIP(hipStreamSynchronize(bed->stream));
        if (ind_col) {
          for (int64_t j = 0; j < cnt; j++) bed->na_cnt[(size_t)ind_col[j0 + j]] = hp[j];
        } else {
          std::memcpy(&bed->na_cnt[(size_t)j0], hp, (size_t)cnt * 4);
        }
      }
      if (op->h_na_total && op->h_na_total[1] >= 0) n_bad += (int32_t)op->h_na_total[1];
      if (o->center_out) copy_d2h(bed, o->center_out, op->d_center.p, (size_t)m * 8);
      if (o->scale_out) copy_d2h(bed, o->scale_out, op->d_scale.p, (size_t)m * 8);
    }
    if (r.converged && r.below_resolution > 0) {
      // (the digits were fixed by the caller: no second solve) the triplets the test left out are not vouched for
      unconverged = true;
      r.converged = 0;
      char buf[320];
      std::snprintf(buf, sizeof(buf),
                    "%d of the %d requested singular triplets lie below what %d-bit products resolve (sigma below %.1e sigma_1): "
                    "their values are not converged; ask for more digits (slices) or fewer triplets",
                    r.below_resolution, so.k, 8 * op->slices, std::sqrt(64.0) * so.resid_floor);
      set_error(buf);
    } else if (!r.converged) {
      unconverged = true;
      char buf[256];
      std::snprintf(buf, sizeof(buf),
                    "the Krylov basis is full (%d vectors) but only a relative residual of %.3g (tol %.3g) was "
                    "reached for the %d requested singular triplets; increase max_basis or tol",
                    r.basis, r.max_rel_resid, so.tol, so.k);
      set_error(buf);
    }
    if (info) {
      info->n_bad = n_bad;
      info->fused_stats = fused ? 1 : 0;
      info->niter = r.niter;
      info->nops = (int32_t)op->passes;
      info->basis = r.basis;
      info->converged = r.converged;
      info->max_rel_resid = r.max_rel_resid;
      info->gpu_ms = ms;
      double pms[kProfKinds];
      int pc[kProfKinds];
      prof_collect(op, pms, pc);
      if (bk.sop) {   // out-of-core: the launches ran on the slab operator, one per slab
        double p2[kProfKinds];
        int c2[kProfKinds];
        prof_collect(bk.sop.get(), p2, c2);
        for (int t = 0; t < kProfKinds; t++) {
          pms[t] += p2[t];
          pc[t] += c2[t];
          if (bk.sop->prof_kernel[t]) op->prof_kernel[t] = bk.sop->prof_kernel[t];
        }
      }
      info->wide_cprod_ms = pms[4];
      info->wide_prod_ms = pms[5];
      info->n_wide_cprod = pc[4];
      info->n_wide_prod = pc[5];
      info->slices_max = r.slices_used_max > 0 ? r.slices_used_max : so.slices_base;
      info->wide_steps = r.wide_steps;
      info->lead_rel_resid = r.lead_rel_resid;
      info->cprod_ms = pms[0];
      info->prod_ms = pms[1];
      info->n_cprod = pc[0];
      info->n_prod = pc[1];
      info->cprod_stats_ms = pms[2];
      info->n_cprod_stats = pc[2];
      info->warm_ms = pms[3];
      info->warm_launches = bk.warm_launches;
      info->warm_fraction = bk.m_sub > 0 && bk.m_op_full > 0 ? (double)bk.m_sub / (double)bk.m_op_full : 0.0;
      info->block = so.block;
      info->slices = so.slices_base;
      info->tiled = (bed->d_tiled != nullptr && op->cols_contig && (op->col0 & 63) == 0) ? 1 : 0;
      if (bed->d_smaj != nullptr && so.block * so.slices_max > 16 && so.block * so.slices_max <= 48 && op->cols_contig &&
          (op->col0 & 511) == 0)
        info->tiled = 2;   // (one launch of two column blocks per product pass: k_prodT)
      info->segmented_passes = bk.n_seg_passes;
      info->compact_gathers = bk.n_compact_gathers;
      info->out_of_core = ooc ? 1 : 0;
      info->na_free_steps[0] = op->bed->na_free[0];
      info->na_free_steps[1] = op->bed->na_free[1];
      info->na_skip = (op->na_skip_c ? 1 : 0) | (op->na_skip_p ? 2 : 0);
      info->compacted = compacted ? 1 : 0;
      info->compact_ms = t_compact;
      info->exchange_mode = bk.exchange_mode;
      for (int c = 0; c < 4; c++) info->exchange_ms[c] = 0, info->n_exchange[c] = 0;
      if (bk.comm && o->exchange_timing) {
        double ems[kCommClasses];
        int en[kCommClasses];
        comm_time_collect(bk.comm, ems, en);
        for (int c = 0; c < 4 && c < kCommClasses; c++) info->exchange_ms[c] = ems[c], info->n_exchange[c] = en[c];
      }
    }
    // the sample-major copy for the NEXT solve, made beside the caller's own work — started last: on a box whose free
    // memory has to be scrubbed the allocation of 100 GB takes seconds and holds up every other call into the runtime
    if (smaj_after && !bed->d_smaj && !bed->smaj_job) (void)image_smaj_start(bed);
  });
  return rc != 0 ? rc : (unconverged ? 2 : 0);
}
