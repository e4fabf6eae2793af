// svd.hip — HIP backend of the block-Lanczos partial SVD (svd_driver.hpp) and the
// bsn_bed_randomsvd entry point (replaces bed_randomSVD -> bigstatsr::big_randomSVD,
// R/autoSVD.R:205-219).  The two matrix passes per step are op_cprod / op_prod
// (matvec.hip); the tall-skinny fp64 panel algebra below is memory-bound on the basis Q
// (n x p doubles) and is a few percent of a step.
#include <chrono>
#include <cmath>
#include <cstring>
#include <memory>

#include "bsn_internal.hpp"
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

// partial[rc][pt*4 + a][j] = sum over the row chunk of Q[i, pt*4+a] * W[i, j]; CB = number of
// columns of W (compile-time, so that no register or FMA is spent on absent columns)
template <int CB>
__global__ __launch_bounds__(256) void k_gemm_tn_part(const double *__restrict__ Q, int64_t ldq,
                                                      int p, const double *__restrict__ W,
                                                      int64_t ldw, int64_t n, int64_t rows_per,
                                                      double *partial) {
  const int pt = blockIdx.x, rc = blockIdx.y, tid = threadIdx.x;
  const int64_t r0 = (int64_t)rc * rows_per;
  int64_t r1 = r0 + rows_per;
  if (r1 > n) r1 = n;
  double acc[4][CB];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int j = 0; j < CB; j++) acc[a][j] = 0;
  const int pc = pt * 4;
  // columns past p are read from the last valid one and discarded at the end (no branch in the loop)
  const double *q0 = Q + (int64_t)(pc + 0 < p ? pc + 0 : p - 1) * ldq, *q1 = Q + (int64_t)(pc + 1 < p ? pc + 1 : p - 1) * ldq,
               *q2 = Q + (int64_t)(pc + 2 < p ? pc + 2 : p - 1) * ldq, *q3 = Q + (int64_t)(pc + 3 < p ? pc + 3 : p - 1) * ldq;
#pragma unroll 2
  for (int64_t i = r0 + tid; i < r1; i += 256) {
    double w[CB];
#pragma unroll
    for (int j = 0; j < CB; j++) w[j] = W[i + j * ldw];
    const double q[4] = {q0[i], q1[i], q2[i], q3[i]};
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int j = 0; j < CB; j++) acc[a][j] += q[a] * w[j];
  }
  __shared__ double red[4][4 * CB];
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int j = 0; j < CB; j++) {
      double v = acc[a][j];
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
      if (lane == 0) red[wave][a * CB + j] = v;
    }
  __syncthreads();
  if (tid < 4 * CB) {
    double v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    int a = tid / CB, j = tid % CB;
    if (pc + a < p) partial[((int64_t)rc * p + pc + a) * kMaxB + j] = v;
  }
}

static void launch_gemm_tn_part(dim3 grid, hipStream_t st, const double *A, int64_t n, int p, const double *W,
                                int cb, int64_t rows_per, double *partial) {
#define BSN_TN(CBV)                                                                                     \
  case CBV:                                                                                             \
    hipLaunchKernelGGL((k_gemm_tn_part<CBV>), grid, dim3(256), 0, st, A, n, p, W, n, n, rows_per, partial); \
    break;
  switch (cb) {
    BSN_TN(1) BSN_TN(2) BSN_TN(3) BSN_TN(4) BSN_TN(5) BSN_TN(6) BSN_TN(7) BSN_TN(8) BSN_TN(9) BSN_TN(10)
    BSN_TN(11) BSN_TN(12) BSN_TN(13) BSN_TN(14) BSN_TN(15) BSN_TN(16)
    default: fail("block size must be <= %d", kMaxB);
  }
#undef BSN_TN
}

// one wave per output element, fixed summation order (lane-strided partial sums, then a
// shuffle tree): the result is identical on every rank and in every run
__global__ __launch_bounds__(256) void k_gemm_tn_reduce(const double *partial, int nrc, int p, int cb,
                                                        double *C) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= p * cb) return;
  const int a = t % p, j = t / p;
  double s = 0;
  for (int rc = lane; rc < nrc; rc += 64) s += partial[((int64_t)rc * p + a) * kMaxB + j];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if (lane == 0) C[a + (int64_t)j * p] = s;
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

// One normalisation pass of the block orthonormalisation on the device (single thread; b <= 12):
// the arithmetic of chol_upper / inv_upper (dense_small.hpp) and of the rank tests of
// svd_driver.hpp's orth().  G0 = W'W before the projections (scale of the deficiency test,
// pass 0 only), G = W'W now.  Writes Ri = R^-1, Rout = R (pass 0) or R * Rout (pass 1), and
// raises *flag when W is not numerically of full rank (the host then redoes the step on the
// step-by-step path).
__global__ __launch_bounds__(64) void k_orth_small(const double *G0, const double *G, int b, int pass,
                                                    double *Ri, double *Rout, double *flag) {
  // inputs are staged in LDS by the whole wave; thread 0 then works on LDS only (a chain of
  // dependent global loads cost 30 us for a 5 x 5 block)
  __shared__ double sG[kMaxB * kMaxB], sG0d[kMaxB], sR[kMaxB * kMaxB], sRi[kMaxB * kMaxB],
      sRo[kMaxB * kMaxB], sRn[kMaxB * kMaxB];
  __shared__ int sbad;
  const int tid = threadIdx.x, bb = b * b;
  for (int t = tid; t < bb; t += 64) {
    sG[t] = G[t];
    sR[t] = 0.0;
    sRi[t] = 0.0;
    sRo[t] = pass == 0 ? 0.0 : Rout[t];
  }
  if (tid < b) sG0d[tid] = G0[tid + tid * b];
  __syncthreads();
  if (tid == 0) {
    bool bad = false;
    if (pass == 0) {
      double w0 = 0;
      for (int i = 0; i < b; i++) w0 = fmax(w0, sG0d[i]);
      for (int i = 0; i < b; i++)
        if (!(sG[i + i * b] > 1e-22 * w0 && w0 > 0)) bad = true;
    }
    double dmax = 0;
    for (int i = 0; i < b; i++) dmax = fmax(dmax, sG[i + i * b]);
    for (int j = 0; j < b && !bad; j++) {
      double s = sG[j + j * b];
      for (int k = 0; k < j; k++) s -= sR[k + j * b] * sR[k + j * b];
      if (!(s > 1e-22 * dmax) || !(dmax > 0)) {
        bad = true;
        break;
      }
      const double rjj = sqrt(s);
      sR[j + j * b] = rjj;
      for (int i = j + 1; i < b; i++) {
        double t = sG[j + i * b];
        for (int k = 0; k < j; k++) t -= sR[k + j * b] * sR[k + i * b];
        sR[j + i * b] = t / rjj;
      }
    }
    if (bad) {
      for (int t = 0; t < bb; t++) sRi[t] = 0.0;
      for (int i = 0; i < b; i++) sRi[i + i * b] = 1.0;  // keep the following kernels finite
    } else {
      for (int j = 0; j < b; j++) {
        sRi[j + j * b] = 1.0 / sR[j + j * b];
        for (int i = j - 1; i >= 0; i--) {
          double s = 0;
          for (int k = i + 1; k <= j; k++) s += sR[i + k * b] * sRi[k + j * b];
          sRi[i + j * b] = -s / sR[i + i * b];
        }
      }
      if (pass == 0) {
        for (int t = 0; t < bb; t++) sRn[t] = sR[t];
      } else {
        for (int j = 0; j < b; j++)
          for (int i = 0; i < b; i++) {
            double s = 0;
            for (int t = i; t < b; t++) s += sR[i + t * b] * sRo[t + j * b];
            sRn[i + j * b] = s;
          }
      }
    }
    sbad = bad ? 1 : 0;
  }
  __syncthreads();
  for (int t = tid; t < bb; t += 64) {
    Ri[t] = sRi[t];
    if (!sbad) Rout[t] = sRn[t];
  }
  if (tid == 0 && sbad) *flag = 1.0;
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
// as before.  Per block step:
//   Z_g = A_g' Qfull           local crossproduct pass; Qfull = newest basis block, all n rows
//   Wfull = A_g Z_g            local product pass, partial sums over this rank's variants
//   W_r = reduce-scatter(Wfull) by sample blocks (own stream: overlaps the local Z'Z Gram kernels)
//   orthonormalise W_r against Q_r (Gram blocks all-reduced, b x p doubles each)
//   Qfull = round(all-gather(W_r)); Q_r gets its rows
// Without a communicator (and without the test hook) nr = n and every collective is skipped: the
// single-GPU path is the same code.
// device buffers of a solve; lives on the bed handle between solves (grow-only)
struct SvdWorkspace {
  DevBuf<double> Q, Z, W, partial, dsmall, Wsave, dorth, Wfull, Wblk, Qfull, dS, dU, dV, dUfull;
  // pinned host staging for the small matrices that cross the bus every block step (Gram blocks,
  // orthogonalisation coefficients): copies to pageable memory go through the runtime's own staging
  // and were measured to cost milliseconds each once a process has run a few solves
  double *h_pin = nullptr;
  size_t h_pin_n = 0;
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
    if (h_pin) (void)hipHostFree(h_pin);
  }
};

struct HipSvdBackend : SvdBackend {
  SvdWorkspace &ws;
  explicit HipSvdBackend(SvdWorkspace &w)
      : ws(w), Q(w.Q), Z(w.Z), W(w.W), partial(w.partial), dsmall(w.dsmall), Wsave(w.Wsave), dorth(w.dorth),
        Wfull(w.Wfull), Wblk(w.Wblk), Qfull(w.Qfull) {}
  bsn_op *op = nullptr;
  hipStream_t st = nullptr;
  bsn_allreduce_fn hook = nullptr;
  void *ctx = nullptr;
  bsn_comm *comm = nullptr;
  int rank = 0, world = 1;
  bool dist = false;
  int64_t nr = 0, row0 = 0;  // rows of a sample block, first row of this rank's block
  DevBuf<double> &Q, &Z, &W, &partial, &dsmall, &Wsave, &dorth, &Wfull, &Wblk, &Qfull;
  std::vector<double> horth;
  int cap = 0, b = 0;
  int64_t rows_per = 0;
  int nrc = 0;
  bool rs_pending = false;
  int kmax = 0;
  // warm start on a leading subset of this rank's variants
  int64_t m_op_full = 0, m_sub = 0;
  bool fused_stats = false;
  int warm_launches = 0, warm_den = 16;
  bool subset(bool on) override {
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
      return true;
    }
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
  struct Tick {
    HipSvdBackend *b;
    int ph;
    std::chrono::steady_clock::time_point t0;
    Tick(HipSvdBackend *b_, int ph_) : b(b_), ph(ph_), t0(std::chrono::steady_clock::now()) {}
    ~Tick() {
      if (b->timing)
        b->t_phase[ph] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
  };

  void setup_ranks() {
    if (comm) {
      rank = comm->rank;
      world = comm->world;
    }
    dist = comm != nullptr || hook != nullptr;
    if (!dist) world = 1, rank = 0;
    nr = dist ? (n + world - 1) / world : n;
    row0 = (int64_t)rank * nr;
  }
  void alloc(int cap_, int b_) override {
    Tick tk(this, 0);
    if (b_ > kMaxB) fail("block size must be <= %d", kMaxB);
    cap = cap_;
    b = b_;
    auto t0 = std::chrono::steady_clock::now();
    Q.ensure((size_t)nr * cap);
    Z.ensure((size_t)m_local * cap);
    W.ensure((size_t)nr * kMaxB);
    rows_per = 4096;
    nrc = (int)((nr + rows_per - 1) / rows_per);
    {
      const int64_t rows_max = nr > m_local ? nr : m_local;
      partial.ensure((size_t)((rows_max + rows_per - 1) / rows_per) * (cap + 4) * kMaxB);
    }
    dsmall.ensure((size_t)(cap + 4) * 64);
    Wsave.ensure((size_t)nr * kMaxB);
    dorth.ensure((size_t)8 * kMaxB * kMaxB + (size_t)3 * (cap + 4) * kMaxB);
    if (dist) {
      const int wide = kmax > kMaxB ? kmax : kMaxB;
      Wfull.ensure((size_t)n * wide);
      Wblk.ensure((size_t)world * nr * wide);
      Qfull.ensure((size_t)n * kMaxB);
    }
    (void)t0;
  }

  // ---- collectives ------------------------------------------------------------------------
  // sum of a small device matrix over the ranks, in place, ordered on the solve's stream
  void ar_small(double *d, int64_t count) {
    if (!dist) return;
    wait_rs();
    if (comm) {
      comm_allreduce_sum(comm, d, count, st);
    } else {
      BSN_HIP(hipStreamSynchronize(st));
      hook(d, count, ctx);
    }
  }
  // the reduce-scatter of the working panel runs on the communicator's stream; everything that
  // reads W, and every later collective, waits for it here
  void wait_rs() {
    if (!rs_pending) return;
    rs_pending = false;
    if (comm) BSN_HIP(hipStreamWaitEvent(st, comm->ev_done, 0));
  }
  void reduce_scatter_W(int cb) {  // Wfull (n x cb partial sums) -> W (nr x cb, summed over ranks)
    const int64_t tot = (int64_t)world * cb * nr;
    hipLaunchKernelGGL(k_block, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, Wfull.p, n, cb, nr, world,
                       Wblk.p);
    BSN_HIP(hipGetLastError());
    if (comm) {
      BSN_HIP(hipEventRecord(comm->ev_ready, st));
      BSN_HIP(hipStreamWaitEvent(comm->stream, comm->ev_ready, 0));
      comm_reduce_scatter_sum(comm, Wblk.p, W.p, nr * cb, comm->stream);
      BSN_HIP(hipEventRecord(comm->ev_done, comm->stream));
      rs_pending = true;
    } else {
      BSN_HIP(hipStreamSynchronize(st));
      hook(Wblk.p, tot, ctx);
      BSN_HIP(hipMemcpyAsync(W.p, Wblk.p + (int64_t)rank * cb * nr, (size_t)nr * cb * 8,
                             hipMemcpyDeviceToDevice, st));
    }
  }
  // src (nr x cb local rows) -> dst (n x cb, all rows, column-major)
  void all_gather_rows(const double *src, int cb, double *dst) {
    wait_rs();
    const int64_t tot = (int64_t)world * cb * nr;
    if (comm) {
      comm_all_gather(comm, src, Wblk.p, nr * cb, st);
    } else {
      BSN_HIP(hipMemsetAsync(Wblk.p, 0, (size_t)tot * 8, st));
      BSN_HIP(hipMemcpyAsync(Wblk.p + (int64_t)rank * cb * nr, src, (size_t)nr * cb * 8, hipMemcpyDeviceToDevice, st));
      BSN_HIP(hipStreamSynchronize(st));
      hook(Wblk.p, tot, ctx);
    }
    hipLaunchKernelGGL(k_unblock, dim3((unsigned)((n * cb + 255) / 256)), dim3(256), 0, st, Wblk.p, n, cb, nr, dst);
    BSN_HIP(hipGetLastError());
  }

  // ---- backend interface ------------------------------------------------------------------
  void random_W(int bb, uint32_t seed) override {
    hipLaunchKernelGGL(k_random, dim3((unsigned)((nr + 255) / 256), bb), dim3(256), 0, st, W.p, nr, nr, bb, seed,
                       row0, n);
    BSN_HIP(hipGetLastError());
  }
  // the newest basis block with all n rows: Q itself on one GPU, the gathered copy otherwise
  const double *newest_block(int p0) const { return dist ? Qfull.p : Q.p + (int64_t)p0 * nr; }
  void At_Qblock(int p0, int cb) override {
    Tick tk(this, 1);
    op_cprod(op, newest_block(p0), n, cb, Z.p + (int64_t)p0 * m_local, m_local);
  }
  void A_Zblock(int p0, int cb) override {
    Tick tk(this, 2);
    op_prod(op, Z.p + (int64_t)p0 * m_local, m_local, cb, dist ? Wfull.p : W.p, n);
    if (dist) reduce_scatter_W(cb);
  }
  void gemm_tn(const double *A, int p, int cb, double *C_host) {
    wait_rs();
    gemm_tn_any(A, W.p, nr, p, cb, dsmall.p);
    BSN_HIP(hipGetLastError());
    ar_small(dsmall.p, (int64_t)p * cb);
    double *hp = ws.pinned((size_t)(cap + 4) * kMaxB * 4 + 1024);
    BSN_HIP(hipMemcpyAsync(hp, dsmall.p, (size_t)p * cb * 8, hipMemcpyDeviceToHost, st));
    BSN_HIP(hipStreamSynchronize(st));
    std::memcpy(C_host, hp, (size_t)p * cb * 8);
  }
  // dC (p x cb) = sum over ranks of A[:, :p]' W[:, :cb] on the local sample block
  void gemm_tn_dev(const double *A, int p, int cb, double *dC) {
    gemm_tn_any(A, W.p, nr, p, cb, dC);
    ar_small(dC, (int64_t)p * cb);
  }
  // dC (p x cb) = A[:, :p]' B[:, :cb] for operands with `rows` rows (leading dimension = rows)
  void gemm_tn_any(const double *A, const double *B, int64_t rows, int p, int cb, double *dC) {
    const int nrc_ = (int)((rows + rows_per - 1) / rows_per);
    dim3 grid((unsigned)((p + 3) / 4), (unsigned)nrc_);
    launch_gemm_tn_part(grid, st, A, rows, p, B, cb, rows_per, partial.p);
    hipLaunchKernelGGL(k_gemm_tn_reduce, dim3((unsigned)((p * cb + 3) / 4)), dim3(256), 0, st,
                       partial.p, nrc_, p, cb, dC);
  }
  void round_cols(double *X, int64_t rows, int cb) {
    unsigned long long *mx = (unsigned long long *)dorth.p;
    BSN_HIP(hipMemsetAsync(mx, 0, (size_t)cb * 8, st));
    hipLaunchKernelGGL(k_col_absmax, dim3(256, cb), dim3(1024), 0, st, X, rows, rows, mx);
    hipLaunchKernelGGL(k_round_cols, dim3((unsigned)((rows + 255) / 256), cb), dim3(256), 0, st, X, rows, rows, mx,
                       op->slices);
    BSN_HIP(hipGetLastError());
  }
  void round_W(int cb) override {
    if (cb <= 0) return;
    Tick tk(this, 5);
    if (!dist) {
      round_cols(W.p, n, cb);
      return;
    }
    // every rank rounds the same gathered block (column maxima over all n rows), then keeps its rows
    all_gather_rows(W.p, cb, Qfull.p);
    round_cols(Qfull.p, n, cb);
    hipLaunchKernelGGL(k_take_rows, dim3((unsigned)((nr * cb + 255) / 256)), dim3(256), 0, st, Qfull.p, n, cb, nr,
                       row0, W.p);
    BSN_HIP(hipGetLastError());
  }
  void gram_to_host(const double *A, const double *B, int64_t rows, int p, int cb, double *out) {
    Tick tk(this, 3);
    gemm_tn_any(A, B, rows, p, cb, dsmall.p);  // local part first: it overlaps the reduce-scatter of W
    BSN_HIP(hipGetLastError());
    ar_small(dsmall.p, (int64_t)p * cb);
    double *hp = ws.pinned((size_t)(cap + 4) * kMaxB * 4 + 1024);
    BSN_HIP(hipMemcpyAsync(hp, dsmall.p, (size_t)p * cb * 8, hipMemcpyDeviceToHost, st));
    BSN_HIP(hipStreamSynchronize(st));
    std::memcpy(out, hp, (size_t)p * cb * 8);
    op_poll_stats(op);
  }
  void ZtZ(int p, int p0, int cb, double *G) override {
    gram_to_host(Z.p, Z.p + (int64_t)p0 * m_local, m_local, p, cb, G);
  }
  void QtQ(int p, int p0, int cb, double *M) override {
    gram_to_host(Q.p, Q.p + (int64_t)p0 * nr, nr, p, cb, M);
  }
  // The whole orth() of svd_driver.hpp queued on the stream with the small matrices kept on
  // the device: one host synchronisation per block step instead of eleven.
  int orth_fused(int p, int cb, std::vector<double> &Cacc, std::vector<double> &Rout) override {
    if (cb <= 0 || cb > kMaxB) return -1;
    Tick tk(this, 4);
    wait_rs();
    hipEvent_t tev0 = nullptr, tev1 = nullptr;
    if (timing) {
      BSN_HIP(hipEventCreate(&tev0));
      BSN_HIP(hipEventCreate(&tev1));
      BSN_HIP(hipEventRecord(tev0, st));
    }
    constexpr int B2 = kMaxB * kMaxB;
    // arena: [flag | Rout | C1 | C2] is downloaded in one piece; then G0, G, Ri, C3
    double *flag = dorth.p, *dRout = flag + 1, *C1 = dRout + B2, *C2 = C1 + (size_t)p * cb,
           *G0 = C2 + (size_t)p * cb, *G = G0 + B2, *Ri = G + B2, *C3 = Ri + B2;
    const size_t nsmall = 1 + B2 + (size_t)2 * p * cb;
    BSN_HIP(hipMemsetAsync(flag, 0, nsmall * 8, st));
    BSN_HIP(hipMemcpyAsync(Wsave.p, W.p, (size_t)nr * cb * 8, hipMemcpyDeviceToDevice, st));
    const dim3 rows((unsigned)((nr + 255) / 256));
    auto project = [&](double *C) {
      gemm_tn_dev(Q.p, p, cb, C);
      hipLaunchKernelGGL(k_gemm_nn, rows, dim3(256), kTP * kMaxB * 8, st, Q.p, nr, p, C, cb, W.p, nr, 1.0, -1.0,
                         W.p, nr, nr);
    };
    gemm_tn_dev(W.p, cb, cb, G0);
    if (p > 0) {
      project(C1);
      project(C2);
    }
    for (int pass = 0; pass < 2; pass++) {
      gemm_tn_dev(W.p, cb, cb, G);
      hipLaunchKernelGGL(k_orth_small, dim3(1), dim3(64), 0, st, G0, G, cb, pass, Ri, dRout, flag);
      hipLaunchKernelGGL(k_right_mult, rows, dim3(256), 0, st, W.p, nr, nr, cb, cb, Ri);
      if (pass == 0 && p > 0) project(C3);
    }
    BSN_HIP(hipGetLastError());
    horth.resize(nsmall);
    double *hp = ws.pinned((size_t)(cap + 4) * kMaxB * 4 + 1024);
    BSN_HIP(hipMemcpyAsync(hp, flag, nsmall * 8, hipMemcpyDeviceToHost, st));
    if (timing) BSN_HIP(hipEventRecord(tev1, st));
    BSN_HIP(hipStreamSynchronize(st));
    if (timing) {
      float ms = 0;
      BSN_HIP(hipEventElapsedTime(&ms, tev0, tev1));
      t_phase[7] += ms;
      (void)hipEventDestroy(tev0);
      (void)hipEventDestroy(tev1);
    }
    std::memcpy(horth.data(), hp, nsmall * 8);
    op_poll_stats(op);
    if (horth[0] != 0.0) {  // rank deficient: undo and let the driver take the careful path
      BSN_HIP(hipMemcpyAsync(W.p, Wsave.p, (size_t)nr * cb * 8, hipMemcpyDeviceToDevice, st));
      return -1;
    }
    Rout.assign((size_t)cb * cb, 0.0);
    for (int t = 0; t < cb * cb; t++) Rout[(size_t)t] = horth[1 + (size_t)t];
    Cacc.assign((size_t)p * cb, 0.0);
    const double *h1 = horth.data() + 1 + B2, *h2 = h1 + (size_t)p * cb;
    for (size_t t = 0; t < (size_t)p * cb; t++) Cacc[t] = h1[t] + h2[t];
    return cb;
  }
  void QtW(int p, int cb, double *C) override { gemm_tn(Q.p, p, cb, C); }
  void WtW(int cb, double *G) override { gemm_tn(W.p, cb, cb, G); }
  void W_minus_QC(int p, int cb, const double *C) override {
    wait_rs();
    copy_h2d(op->bed, dsmall.p, C, (size_t)p * cb * 8);
    hipLaunchKernelGGL(k_gemm_nn, dim3((unsigned)((nr + 255) / 256)), dim3(256), kTP * kMaxB * 8, st,
                       Q.p, nr, p, dsmall.p, cb, W.p, nr, 1.0, -1.0, W.p, nr, nr);
    BSN_HIP(hipGetLastError());
    BSN_HIP(hipStreamSynchronize(st));  // C is a host vector that may be reused
  }
  void W_times(int cb, int r, const double *M) override {
    wait_rs();
    copy_h2d(op->bed, dsmall.p, M, (size_t)cb * r * 8);
    hipLaunchKernelGGL(k_right_mult, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, W.p, nr, nr,
                       cb, r, dsmall.p);
    BSN_HIP(hipGetLastError());
    BSN_HIP(hipStreamSynchronize(st));
  }
  void W_to_Q(int p0, int r) override {
    BSN_HIP(hipMemcpyAsync(Q.p + (int64_t)p0 * nr, W.p, (size_t)nr * r * 8, hipMemcpyDeviceToDevice, st));
  }
  void finalize(int pp, int k, const double *S, const double *dinv, double *u, double *v) override {
    Tick tk(this, 6);
    wait_rs();
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
    BSN_HIP(hipStreamSynchronize(st));
  }
};

}  // namespace bsn

using namespace bsn;

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

extern "C" int bsn_bed_randomsvd(bsn_bed *bed, const int64_t *ind_row, int64_t n,
                                 const int64_t *ind_col, int64_t m, const double *center,
                                 const double *scale, const bsn_svd_options *o, double *d, double *u,
                                 double *v, bsn_svd_info *info) {
  bool unconverged = false;
  int rc = guarded([&] {
    if (!o) fail("options must not be NULL");
    if (o->k < 1) fail("'k' must be at least 1.");
    auto t_begin = std::chrono::steady_clock::now();
    auto since = [&]() {
      return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    };
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
      fill_op(op, bed, ind_row, n, ind_col, m, nullptr, nullptr, true);
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
      fill_op(op, bed, ind_row, n, ind_col, m, center, scale);
    }
    // a solve streams the image a dozen times: give the two streaming kernels their layout (a second copy in
    // 64-variant x 256-B tiles, one extra pass of copying, kept on the handle) when the device has the room
    if (op->cols_contig && (op->col0 & 63) == 0 && m >= 4096) image_tile(bed);
    const double t_create = since();
    op->profile = true;
    if (o->slices > 7) fail("slices must be in 1..7");
    HipSvdBackend bk(*bed->svd_ws);
    bk.op = op;
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
      int bb = o->block > 0 ? (o->block > kMaxB ? kMaxB : o->block) : std::max(1, std::min(8, 16 / s_tol));
      int ss = s_tol;
      if (o->slices <= 0) {
        const int nb = (bb * s_tol + 15) / 16;
        ss = std::max(s_tol, std::min(7, 16 * nb / bb));
      }
      so.block = bb;
      op->slices = ss;
    }
    so.resid_floor = 1.2 * std::ldexp(1.0, -8 * op->slices);
    so.warm = o->warm_start < 0 ? 0 : (o->warm_start == 0 ? 1 : o->warm_start);
    so.max_basis = o->max_basis;
    so.seed = o->seed ? o->seed : 1;
    so.verbose = o->verbose;
    BSN_HIP(hipEventRecord(bed->ev0, bed->stream));
    SvdResult r = block_lanczos_svd(bk, so, d, u, v);
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
      const int64_t chunk = 1 << 20;  // elements per staged piece (8 MB of doubles)
      double *hp = ws.pinned((size_t)chunk);
      if ((int64_t)bed->na_cnt.size() != bed->m) bed->na_cnt.assign((size_t)bed->m, -1);
      for (int64_t j0 = 0; j0 < m; j0 += chunk / 2) {   // 4 int32 per variant
        const int64_t cnt = std::min<int64_t>(chunk / 2, m - j0);
        BSN_HIP(hipMemcpyAsync(hp, op->d_counts.p + 4 * j0, (size_t)cnt * 16, hipMemcpyDeviceToHost, bed->stream));
        BSN_HIP(hipStreamSynchronize(bed->stream));
        const int32_t *c = (const int32_t *)hp;
        for (int64_t j = 0; j < cnt; j++, c += 4) {
          bed->na_cnt[(size_t)(ind_col ? ind_col[j0 + j] : j0 + j)] = c[3];
          if (2 * (int64_t)(c[0] + c[1] + c[2]) < n) n_bad++;
        }
      }
      if (o->center_out) copy_d2h(bed, o->center_out, op->d_center.p, (size_t)m * 8);
      if (o->scale_out) copy_d2h(bed, o->scale_out, op->d_scale.p, (size_t)m * 8);
    }
    if (!r.converged) {
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
      double pms[4];
      int pc[4];
      prof_collect(op, pms, pc);
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
      info->slices = op->slices;
      info->tiled = (bed->d_tiled != nullptr && op->cols_contig && (op->col0 & 63) == 0) ? 1 : 0;
    }
  });
  return rc != 0 ? rc : (unconverged ? 2 : 0);
}
