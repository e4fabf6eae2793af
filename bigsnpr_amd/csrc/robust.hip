// robust.hip — the robust statistics behind the outlier step of snp_autoSVD / bed_autoSVD on the device.
//
// R/autoSVD.R:142-148 (and :295-301) removes, after every partial SVD, the variants whose loadings are outliers:
//   S <- bigutilsr::dist_ogk(obj.svd$v) -> rollmean -> tukey_mc_up
// dist_ogk — the orthogonalised Gnanadesikan-Kettenring estimator of Maronna & Zamar (2002) as robustbase::covOGK runs
// it, scale function scaleTau2 — is, per round, p + p (p - 1) robust scales of vectors of length m (the variants): each
// scale is two medians and two weighted sums.  On the host that is the bulk of a north_star function's wall time once
// the solves run on the GPU (0.78 of 0.99 s at 400K x 250K, k = 10; profiles/r03_autosvd_end_to_end.txt).  Here the
// scales of a whole batch of columns are computed together on the device:
//   * a median is a RADIX SELECT on the order-preserving 64-bit image of the doubles: 8 passes of 8 bits, each a
//     histogram of the current byte over the keys that match the prefix found so far (LDS histograms, one atomic add
//     per bin and workgroup), then one small kernel per pass that picks the bucket holding the wanted rank; both
//     middle order statistics of an even-length vector are selected side by side (numpy / R medians average them);
//   * the weighted mean and the truncated second moment are fixed-shape two-stage sums (64 partial sums per column,
//     added in order): the same bits run to run.
// The OGK loop itself — p x p eigen-decompositions, the hard-rejection step, the final Mahalanobis distances — is in the
// host mirror (bigsnpr_amd/autosvd.py: covrob_ogk_device, which calls these entry points on a device copy of the m x p
// loadings) AND, since round 6, here end to end (bsn_robust_dist_ogk: only p x p matrices visit the host), together with
// what follows it in the reference's loop: the Gaussian rolling mean (bsn_robust_rollmean) and, for tukey_mc_up, a device
// sort (bsn_robust_sort) and the medcouple's window of kernel values (bsn_robust_mc_window) — at a million variants the
// host versions of those (a fancy-index copy and np.cov of the kept rows, two 101-tap convolutions, np.sort, two
// searchsorted calls) cost more than the partial SVD they follow (profiles/r06_ld_autosvd.txt).
// bigutilsr is external to the reference tree: parity is pinned against the two independent host restatements
// (tests/test_autosvd_helpers_cpu.py, tests/test_gpu_autosvd.py), DESIGN.md section 6.
#include <cmath>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "bsn_internal.hpp"
#include "dense_small.hpp"

namespace bsn {
namespace {

__device__ __forceinline__ unsigned long long key_of(double x) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double val_of(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}
// the value a selection looks at: the entry itself, or its distance to the column's centre
__device__ __forceinline__ double view(const double *__restrict__ X, int64_t ld, int c, int64_t i, const double *__restrict__ centre) {
  const double x = X[i + (int64_t)c * ld];
  return centre ? fabs(x - centre[c]) : x;
}

constexpr int kSelBlocks = 64;   // workgroups per (virtual) column in the histogram and sum kernels

// One pass of the radix select.  Virtual column v = 2 c + w selects order statistic rank[v] of column c (w = 0 / 1: the
// lower / upper middle).  hist[v][256] += number of keys of column c whose bytes above `shift` equal prefix[v]'s and
// whose byte at `shift` is the bin.
__global__ __launch_bounds__(256) void k_sel_hist(const double *__restrict__ X, int64_t m, int64_t ld, const double *__restrict__ centre,
                                                  const unsigned long long *__restrict__ prefix, int shift, unsigned int *__restrict__ hist) {
  __shared__ unsigned int sh[256];
  const int v = blockIdx.y, c = v >> 1;
  sh[threadIdx.x] = 0;
  __syncthreads();
  const unsigned long long pre = prefix[v];
  const unsigned long long himask = shift >= 56 ? 0ull : ~0ull << (shift + 8);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
    const unsigned long long k = key_of(view(X, ld, c, i, centre));
    if ((k & himask) == (pre & himask)) atomicAdd(&sh[(k >> shift) & 255ull], 1u);
  }
  __syncthreads();
  if (sh[threadIdx.x]) atomicAdd(&hist[(int64_t)v * 256 + threadIdx.x], sh[threadIdx.x]);
}
// ... and the bucket that holds the wanted rank: prefix gains its byte, rank becomes the rank inside the bucket
__global__ void k_sel_pick(unsigned int *__restrict__ hist, unsigned long long *__restrict__ prefix, long long *__restrict__ rank, int shift, int nv) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  unsigned int *h = hist + (int64_t)v * 256;
  long long r = rank[v], below = 0;
  int bin = 255;
  for (int b = 0; b < 256; b++) {
    if (below + (long long)h[b] > r) {
      bin = b;
      break;
    }
    below += h[b];
  }
  rank[v] = r - below;
  prefix[v] |= (unsigned long long)bin << shift;
  for (int b = 0; b < 256; b++) h[b] = 0;
}
// median of column c from its two selected keys
__global__ void k_sel_median(const unsigned long long *__restrict__ prefix, int ncol, double *__restrict__ med) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < ncol) med[c] = 0.5 * (val_of(prefix[2 * c]) + val_of(prefix[2 * c + 1]));
}

// two-stage sums: part[(c * kSelBlocks + block) * 2 + {0, 1}]
// stage 0 of scaleTau2: sum x w and sum w, w = max(0, 1 - (|x - med| / (sigma0 c1))^2)^2
// stage 1: sum of min(((x - mu) / sigma0)^2, c2^2)
__global__ __launch_bounds__(256) void k_tau_sums(const double *__restrict__ X, int64_t m, int64_t ld, int stage,
                                                  const double *__restrict__ med, const double *__restrict__ sigma0,
                                                  const double *__restrict__ mu, double c1, double c2, double *__restrict__ part) {
  const int c = blockIdx.y;
  const double s0 = sigma0[c], me = med[c], muc = stage ? mu[c] : 0.0;
  double a = 0, b = 0;
  if (s0 > 0) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
      const double x = X[i + (int64_t)c * ld];
      if (stage == 0) {
        const double t = fabs(x - me) / (s0 * c1);
        double w = 1.0 - t * t;
        w = w > 0 ? w * w : 0.0;
        a += x * w;
        b += w;
      } else {
        const double t = (x - muc) / s0;
        a += fmin(t * t, c2 * c2);
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off);
    b += __shfl_down(b, off);
  }
  __shared__ double sa[4], sb[4];
  if ((threadIdx.x & 63) == 0) sa[threadIdx.x >> 6] = a, sb[threadIdx.x >> 6] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    double *p = part + ((int64_t)c * gridDim.x + blockIdx.x) * 2;
    p[0] = (sa[0] + sa[1]) + (sa[2] + sa[3]);
    p[1] = (sb[0] + sb[1]) + (sb[2] + sb[3]);
  }
}
__global__ void k_tau_finish(const double *__restrict__ part, int nblk, int ncol, int stage, const double *__restrict__ sigma0, int64_t m,
                             double erho, double *__restrict__ mu, double *__restrict__ s) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncol) return;
  double a = 0, b = 0;
  for (int t = 0; t < nblk; t++) {
    a += part[((int64_t)c * nblk + t) * 2];
    b += part[((int64_t)c * nblk + t) * 2 + 1];
  }
  if (stage == 0) mu[c] = sigma0[c] > 0 ? a / b : 0.0;
  else s[c] = sigma0[c] > 0 ? sigma0[c] * sqrt(a / ((double)m * erho)) : 0.0;
}

// out[:, 2 q] = Z_i + Z_j, out[:, 2 q + 1] = Z_i - Z_j for the pairs q of this chunk
__global__ void k_pairs(const double *__restrict__ Z, int64_t m, int64_t ld, const int *__restrict__ pi, const int *__restrict__ pj,
                        double *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = blockIdx.y;
  if (i >= m) return;
  const double a = Z[i + (int64_t)pi[q] * ld], b = Z[i + (int64_t)pj[q] * ld];
  out[i + (int64_t)(2 * q) * m] = a + b;
  out[i + (int64_t)(2 * q + 1) * m] = a - b;
}
__global__ void k_scale_cols(double *Z, int64_t m, int64_t ld, const double *__restrict__ div) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) Z[i + (int64_t)blockIdx.y * ld] /= div[blockIdx.y];
}
// Z <- Z E (p <= 64; a row per thread, E in LDS)
__global__ __launch_bounds__(256) void k_rotate(double *Z, int64_t m, int64_t ld, int p, const double *__restrict__ E) {
  extern __shared__ double se[];
  for (int t = threadIdx.x; t < p * p; t += 256) se[t] = E[t];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  double row[64], out[64];
  for (int j = 0; j < p; j++) row[j] = Z[i + (int64_t)j * ld];
  for (int c = 0; c < p; c++) {
    double s = 0;
    for (int j = 0; j < p; j++) s += row[j] * se[j + c * p];
    out[c] = s;
  }
  for (int c = 0; c < p; c++) Z[i + (int64_t)c * ld] = out[c];
}
__global__ void k_wdist(const double *__restrict__ Z, int64_t m, int64_t ld, int p, const double *__restrict__ mu, const double *__restrict__ sig,
                        double *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  double s = 0;
  for (int j = 0; j < p; j++) {
    const double t = (Z[i + (int64_t)j * ld] - mu[j]) / sig[j];
    s += t * t;
  }
  out[i] = s;
}

// Medcouple (bigsnpr_amd/autosvd.py: medcouple): number of regular pairs (u, l) — u from `up`, l from the ascending
// `lo` — whose kernel (u - l) / (u + l) is <= t, i.e. l >= u (1 - t) / (1 + t): one binary search per u
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void k_mc_count(const double *__restrict__ up, int64_t nu, const double *__restrict__ lo, int64_t nl, double t,
                                                  unsigned long long *__restrict__ out) {
  unsigned long long c = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nu; i += (int64_t)gridDim.x * 256) {
    const double thr = (up[i] * (1.0 - t)) / (1.0 + t);
    int64_t a = 0, b = nl;            // first index with lo[idx] >= thr (numpy.searchsorted, side = "left")
    while (a < b) {
      const int64_t mid = (a + b) >> 1;
      if (lo[mid] < thr) a = mid + 1; else b = mid;
    }
    c += (unsigned long long)(nl - a);
  }
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// Medcouple, the end of the bisection: every kernel value (u - l) / (u + l) in (a, b], -1 < a < b < 1, written out (any
// order: the caller picks by rank).  Per u the l's form the index range [first l >= u (1 - b) / (1 + b), first l >=
// u (1 - a) / (1 + a)) of the ascending `lo`; a thread reserves its range with one atomic add.  Nothing is written once the
// total passes `cap` (the caller bisects further).
__global__ __launch_bounds__(256) void k_mc_window(const double *__restrict__ up, int64_t nu, const double *__restrict__ lo, int64_t nl, double a,
                                                   double b, unsigned long long cap, unsigned long long *__restrict__ total,
                                                   double *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nu; i += (int64_t)gridDim.x * 256) {
    const double u = up[i];
    const double tb = (u * (1.0 - b)) / (1.0 + b), ta = (u * (1.0 - a)) / (1.0 + a);
    int64_t x0 = 0, x1 = nl;
    while (x0 < x1) {
      const int64_t mid = (x0 + x1) >> 1;
      if (lo[mid] < tb) x0 = mid + 1; else x1 = mid;
    }
    int64_t y0 = x0, y1 = nl;           // (ta >= tb)
    while (y0 < y1) {
      const int64_t mid = (y0 + y1) >> 1;
      if (lo[mid] < ta) y0 = mid + 1; else y1 = mid;
    }
    const int64_t cnt = y0 - x0;
    if (cnt <= 0) continue;
    const unsigned long long at = atomicAdd(total, (unsigned long long)cnt);
    if (at + (unsigned long long)cnt > cap) continue;
    for (int64_t t = 0; t < cnt; t++) {
      const double l = lo[x0 + t];
      out[at + (unsigned long long)t] = (u - l) / (u + l);
    }
  }
}

// Gaussian rolling mean (bigutilsr::rollmean): out[i] = sum_j w[j] x[i - half + j] / sum_j w[j] over the taps that fall
// inside the GROUP of i (groups = consecutive index ranges, the chromosomes; edge windows renormalised by the weights
// they contain), taps added in order
__global__ __launch_bounds__(256) void k_rollmean(const double *__restrict__ x, int64_t m, const double *__restrict__ w, int len,
                                                  const long long *__restrict__ goff, int ngroups, double *__restrict__ out) {
  extern __shared__ double sw[];
  for (int t = threadIdx.x; t < len; t += 256) sw[t] = w[t];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  int g0 = 0, g1 = ngroups;             // the group of i: last offset <= i
  while (g1 - g0 > 1) {
    const int mid = (g0 + g1) >> 1;
    if (goff[mid] <= i) g0 = mid; else g1 = mid;
  }
  const int64_t lo = goff[g0], hi = goff[g0 + 1], half = len / 2;
  double num = 0, den = 0;
  for (int j = 0; j < len; j++) {
    const int64_t k = i - half + j;
    if (k >= lo && k < hi) {
      num += sw[j] * x[k];
      den += sw[j];
    }
  }
  out[i] = num / den;
}

// hard rejection of covOGK: rows with wdist <= d0 are kept.  Fixed-shape two-stage sums over the kept rows, one
// (virtual) column per blockIdx.y: stage 0 — column c: sum of U[, c] (and the number of kept rows in slot 1); stage 1 —
// pair q = (i >= j): sum of (U[, i] - centre[i]) (U[, j] - centre[j])
__global__ __launch_bounds__(256) void k_keep_sums(const double *__restrict__ U, int64_t m, int64_t ld, const double *__restrict__ wd, double d0,
                                                   int stage, const int *__restrict__ pi, const int *__restrict__ pj,
                                                   const double *__restrict__ centre, double *__restrict__ part) {
  const int q = blockIdx.y;
  const int ci = stage ? pi[q] : q, cj = stage ? pj[q] : q;
  const double mi = stage ? centre[ci] : 0.0, mj = stage ? centre[cj] : 0.0;
  double a = 0, b = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
    if (!(wd[i] <= d0)) continue;
    if (stage == 0) {
      a += U[i + (int64_t)ci * ld];
      b += 1.0;
    } else {
      a += (U[i + (int64_t)ci * ld] - mi) * (U[i + (int64_t)cj * ld] - mj);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off);
    b += __shfl_down(b, off);
  }
  __shared__ double sa[4], sb[4];
  if ((threadIdx.x & 63) == 0) sa[threadIdx.x >> 6] = a, sb[threadIdx.x >> 6] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    double *p = part + ((int64_t)q * gridDim.x + blockIdx.x) * 2;
    p[0] = (sa[0] + sa[1]) + (sa[2] + sa[3]);
    p[1] = (sb[0] + sb[1]) + (sb[2] + sb[3]);
  }
}
// out[i] = (U[i, ] - centre)' P (U[i, ] - centre), P symmetric p x p in LDS, p <= 64
__global__ __launch_bounds__(256) void k_mahalanobis(const double *__restrict__ U, int64_t m, int64_t ld, int p, const double *__restrict__ centre,
                                                     const double *__restrict__ P, double *__restrict__ out) {
  extern __shared__ double sp[];
  for (int t = threadIdx.x; t < p * p; t += 256) sp[t] = P[t];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  double row[64];
  for (int j = 0; j < p; j++) row[j] = U[i + (int64_t)j * ld] - centre[j];
  double s = 0;
  for (int c = 0; c < p; c++) {
    double t = 0;
    for (int j = 0; j < p; j++) t += row[j] * sp[j + c * p];
    s += t * row[c];
  }
  out[i] = s;
}

// order(S, decreasing = TRUE) inside groups (R/clumping.R:106): helpers around two stable radix sorts
__global__ void k_iota_i32(int *v, int64_t m) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) v[i] = (int)i;
}
// the group of every element from the ascending group offsets
__global__ void k_group_of(const long long *__restrict__ goff, int ngroups, int64_t m, unsigned int *__restrict__ grp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  int g0 = 0, g1 = ngroups;
  while (g1 - g0 > 1) {
    const int mid = (g0 + g1) >> 1;
    if (goff[mid] <= i) g0 = mid; else g1 = mid;
  }
  grp[i] = (unsigned int)g0;
}
__global__ void k_gather_u32(const unsigned int *__restrict__ src, const int *__restrict__ idx, int64_t m, unsigned int *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) out[i] = src[idx[i]];
}
// ord (global positions, grouped) -> local order and its inverse
__global__ void k_order_finish(const int *__restrict__ val, const unsigned int *__restrict__ grp_sorted, const long long *__restrict__ goff, int64_t m,
                               int *__restrict__ ord, int *__restrict__ rank) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m) return;
  const long long o = goff[grp_sorted[t]];
  const int local = val[t] - (int)o;
  ord[t] = local;                          // position t of the concatenated output belongs to this group: t - o is its place
  rank[val[t]] = (int)(t - o);
}

// Erho(b) of robustbase::scaleTau2 (consistency = TRUE): 2 ((1 - b^2) Phi(b) - b phi(b) + b^2) - 1
double erho_of(double b) {
  const double Phi = 0.5 * std::erfc(-b / std::sqrt(2.0)), phi = std::exp(-0.5 * b * b) / std::sqrt(2.0 * M_PI);
  return 2.0 * ((1.0 - b * b) * Phi - b * phi + b * b) - 1.0;
}

struct Scratch {
  DevBuf<unsigned long long> prefix;
  DevBuf<long long> rank;
  DevBuf<unsigned int> hist;
  DevBuf<double> med, sigma0, mu, s, part;
};

// medians of the `ncol` columns of X (or of |X - centre|), device result in `out`
void select_medians(Scratch &w, const double *d_X, int64_t m, int64_t ld, int ncol, const double *d_centre, double *d_out, hipStream_t st) {
  const int nv = 2 * ncol;
  w.prefix.ensure((size_t)nv);
  w.rank.ensure((size_t)nv);
  w.hist.ensure((size_t)nv * 256);
  std::vector<long long> r((size_t)nv);
  for (int c = 0; c < ncol; c++) {
    r[(size_t)(2 * c)] = (m - 1) / 2;   // the two middle order statistics (equal for odd m)
    r[(size_t)(2 * c + 1)] = m / 2;
  }
  BSN_HIP(hipMemcpyAsync(w.rank.p, r.data(), (size_t)nv * 8, hipMemcpyHostToDevice, st));
  BSN_HIP(hipMemsetAsync(w.prefix.p, 0, (size_t)nv * 8, st));
  BSN_HIP(hipMemsetAsync(w.hist.p, 0, (size_t)nv * 256 * 4, st));
  BSN_HIP(hipStreamSynchronize(st));   // (r is a host vector)
  int gx = (int)std::min<int64_t>(kSelBlocks, (m + 255) / 256);
  for (int shift = 56; shift >= 0; shift -= 8) {
    hipLaunchKernelGGL(k_sel_hist, dim3((unsigned)gx, (unsigned)nv), dim3(256), 0, st, d_X, m, ld, d_centre, w.prefix.p, shift, w.hist.p);
    hipLaunchKernelGGL(k_sel_pick, dim3((unsigned)((nv + 63) / 64)), dim3(64), 0, st, w.hist.p, w.prefix.p, w.rank.p, shift, nv);
  }
  hipLaunchKernelGGL(k_sel_median, dim3((unsigned)((ncol + 63) / 64)), dim3(64), 0, st, w.prefix.p, ncol, d_out);
  BSN_HIP(hipGetLastError());
}

// robustbase::scaleTau2 of every column: mu (location) and s (scale) to the host
void tau2_columns(Scratch &w, const double *d_X, int64_t m, int64_t ld, int ncol, double c1, double c2, double *mu_out, double *s_out,
                  hipStream_t st) {
  if (ncol <= 0) return;
  w.med.ensure((size_t)ncol);
  w.sigma0.ensure((size_t)ncol);
  w.mu.ensure((size_t)ncol);
  w.s.ensure((size_t)ncol);
  const int gx = (int)std::min<int64_t>(kSelBlocks, (m + 255) / 256);
  w.part.ensure((size_t)ncol * gx * 2);
  select_medians(w, d_X, m, ld, ncol, nullptr, w.med.p, st);
  select_medians(w, d_X, m, ld, ncol, w.med.p, w.sigma0.p, st);
  const double q75 = 0.674489750196081743;   // qnorm(3/4): sigma0 is the raw MAD, Es2(c2) = Erho(c2 qnorm(3/4))
  const double erho = erho_of(c2 * q75);
  for (int stage = 0; stage < 2; stage++) {
    hipLaunchKernelGGL(k_tau_sums, dim3((unsigned)gx, (unsigned)ncol), dim3(256), 0, st, d_X, m, ld, stage, w.med.p, w.sigma0.p, w.mu.p, c1,
                       c2, w.part.p);
    hipLaunchKernelGGL(k_tau_finish, dim3((unsigned)((ncol + 63) / 64)), dim3(64), 0, st, w.part.p, gx, ncol, stage, w.sigma0.p, m, erho,
                       w.mu.p, w.s.p);
  }
  BSN_HIP(hipGetLastError());
  if (mu_out) BSN_HIP(hipMemcpyAsync(mu_out, w.mu.p, (size_t)ncol * 8, hipMemcpyDeviceToHost, st));
  if (s_out) BSN_HIP(hipMemcpyAsync(s_out, w.s.p, (size_t)ncol * 8, hipMemcpyDeviceToHost, st));
  BSN_HIP(hipStreamSynchronize(st));
  if (mu_out) {   // (a column whose MAD is 0: scaleTau2 returns the median and 0)
    std::vector<double> med((size_t)ncol), s0((size_t)ncol);
    BSN_HIP(hipMemcpy(med.data(), w.med.p, (size_t)ncol * 8, hipMemcpyDeviceToHost));
    BSN_HIP(hipMemcpy(s0.data(), w.sigma0.p, (size_t)ncol * 8, hipMemcpyDeviceToHost));
    for (int c = 0; c < ncol; c++)
      if (!(s0[(size_t)c] > 0)) mu_out[c] = med[(size_t)c];
  }
}

// scaleTau2 of Z_i + Z_j and Z_i - Z_j for every pair i > j (order (1,0), (2,0), (2,1), ...), to the host
void pair_scales(Scratch &w, const double *d_Z, int64_t m, int64_t ld, int p, double c1, double c2, double *s_sum_out, double *s_diff_out) {
  std::vector<int> pi, pj;
  for (int i = 0; i < p; i++)
    for (int j = 0; j < i; j++) {
      pi.push_back(i);
      pj.push_back(j);
    }
  const int npair = (int)pi.size();
  if (npair == 0) return;
  // chunks of pairs: at most 2 GB of materialised sums and differences at a time
  int chunk = (int)std::max<int64_t>(1, std::min<int64_t>(npair, ((int64_t)2 << 30) / (16 * m)));
  DevBuf<double> d_pairs;
  DevBuf<int> d_pi, d_pj;
  d_pairs.ensure((size_t)m * 2 * chunk);
  d_pi.ensure((size_t)npair);
  d_pj.ensure((size_t)npair);
  BSN_HIP(hipMemcpy(d_pi.p, pi.data(), (size_t)npair * 4, hipMemcpyHostToDevice));
  BSN_HIP(hipMemcpy(d_pj.p, pj.data(), (size_t)npair * 4, hipMemcpyHostToDevice));
  std::vector<double> s((size_t)2 * chunk);
  for (int q0 = 0; q0 < npair; q0 += chunk) {
    const int nq = std::min(chunk, npair - q0);
    hipLaunchKernelGGL(k_pairs, dim3((unsigned)((m + 255) / 256), (unsigned)nq), dim3(256), 0, nullptr, d_Z, m, ld, d_pi.p + q0, d_pj.p + q0,
                       d_pairs.p);
    BSN_HIP(hipGetLastError());
    tau2_columns(w, d_pairs.p, m, m, 2 * nq, c1, c2, nullptr, s.data(), nullptr);
    for (int q = 0; q < nq; q++) {
      s_sum_out[q0 + q] = s[(size_t)(2 * q)];
      s_diff_out[q0 + q] = s[(size_t)(2 * q + 1)];
    }
  }
}
void scale_cols(double *d_Z, int64_t m, int64_t ld, int p, const double *div) {
  DevBuf<double> d_div;
  BSN_HIP(hipMemcpy(d_div.ensure((size_t)p), div, (size_t)p * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_scale_cols, dim3((unsigned)((m + 255) / 256), (unsigned)p), dim3(256), 0, nullptr, d_Z, m, ld, d_div.p);
  BSN_HIP(hipGetLastError());
  BSN_HIP(hipStreamSynchronize(nullptr));
}
void rotate(double *d_Z, int64_t m, int64_t ld, int p, const double *E) {
  DevBuf<double> d_E;
  BSN_HIP(hipMemcpy(d_E.ensure((size_t)p * p), E, (size_t)p * p * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_rotate, dim3((unsigned)((m + 255) / 256)), dim3(256), (size_t)p * p * 8, nullptr, d_Z, m, ld, p, d_E.p);
  BSN_HIP(hipGetLastError());
  BSN_HIP(hipStreamSynchronize(nullptr));
}
// out[i] = sum_j ((Z[i, j] - mu[j]) / sig[j])^2 into the device vector d_out
void wdist(const double *d_Z, int64_t m, int64_t ld, int p, const double *mu, const double *sig, double *d_out) {
  DevBuf<double> d_mu, d_sig;
  BSN_HIP(hipMemcpy(d_mu.ensure((size_t)p), mu, (size_t)p * 8, hipMemcpyHostToDevice));
  BSN_HIP(hipMemcpy(d_sig.ensure((size_t)p), sig, (size_t)p * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_wdist, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, nullptr, d_Z, m, ld, p, d_mu.p, d_sig.p, d_out);
  BSN_HIP(hipGetLastError());
  BSN_HIP(hipStreamSynchronize(nullptr));
}

// bigutilsr::dist_ogk on the device: the statements of bigsnpr_amd/autosvd.py: covrob_ogk_device + dist_ogk, in their order
void dist_ogk(const double *d_U, int64_t m, int64_t ld, int p, int niter, double cut_ratio, double c1, double c2, double *dist_out,
              int64_t *n_kept_out) {
  Scratch w;
  DevBuf<double> Z, d_wd;
  Z.ensure((size_t)m * p);
  d_wd.ensure((size_t)m);
  BSN_HIP(hipMemcpy2D(Z.p, (size_t)m * 8, d_U, (size_t)ld * 8, (size_t)m * 8, (size_t)p, hipMemcpyDeviceToDevice));
  const int npair = p * (p - 1) / 2;
  std::vector<double> d((size_t)p), ss((size_t)std::max(npair, 1)), sd((size_t)std::max(npair, 1)), R, ev, E((size_t)p * p);
  for (int it = 0; it < niter; it++) {
    tau2_columns(w, Z.p, m, m, p, c1, c2, nullptr, d.data(), nullptr);
    for (auto &v : d)
      if (!(v > 0)) v = 1.0;
    scale_cols(Z.p, m, m, p, d.data());
    R.assign((size_t)p * p, 0.0);
    for (int i = 0; i < p; i++) R[(size_t)i + (size_t)i * p] = 1.0;
    if (npair) {
      pair_scales(w, Z.p, m, m, p, c1, c2, ss.data(), sd.data());
      int t = 0;
      for (int i = 0; i < p; i++)
        for (int j = 0; j < i; j++, t++)
          R[(size_t)i + (size_t)j * p] = R[(size_t)j + (size_t)i * p] = (ss[(size_t)t] * ss[(size_t)t] - sd[(size_t)t] * sd[(size_t)t]) / 4;
    }
    eig_sym(p, R, ev);                                   // ascending; the loop wants the largest first (the order of the
    for (int c = 0; c < p; c++)                          // columns and their signs do not reach the distances)
      for (int r = 0; r < p; r++) E[(size_t)r + (size_t)c * p] = R[(size_t)r + (size_t)(p - 1 - c) * p];
    rotate(Z.p, m, m, p, E.data());
  }
  std::vector<double> mu((size_t)p), sig((size_t)p);
  tau2_columns(w, Z.p, m, m, p, c1, c2, mu.data(), sig.data(), nullptr);
  for (auto &v : sig)
    if (!(v > 0)) v = 1.0;
  wdist(Z.p, m, m, p, mu.data(), sig.data(), d_wd.p);
  // hard rejection: keep wdist <= median(wdist) * qchisq(beta, p) / qchisq(0.5, p) (cut_ratio: the caller's quantiles)
  w.med.ensure(1);
  select_medians(w, d_wd.p, m, m, 1, nullptr, w.med.p, nullptr);
  double med = 0;
  BSN_HIP(hipMemcpy(&med, w.med.p, 8, hipMemcpyDeviceToHost));
  const double d0 = med * cut_ratio;
  // centre and covariance (denominator n_kept - 1) of the kept rows of U
  const int gx = (int)std::min<int64_t>(kSelBlocks, (m + 255) / 256);
  std::vector<int> pi, pj;
  for (int i = 0; i < p; i++)
    for (int j = 0; j <= i; j++) {
      pi.push_back(i);
      pj.push_back(j);
    }
  const int nq = (int)pi.size();
  DevBuf<int> d_pi, d_pj;
  DevBuf<double> d_part, d_centre, d_P;
  BSN_HIP(hipMemcpy(d_pi.ensure((size_t)nq), pi.data(), (size_t)nq * 4, hipMemcpyHostToDevice));
  BSN_HIP(hipMemcpy(d_pj.ensure((size_t)nq), pj.data(), (size_t)nq * 4, hipMemcpyHostToDevice));
  d_part.ensure((size_t)nq * gx * 2);
  std::vector<double> part((size_t)nq * gx * 2), centre((size_t)p), cov((size_t)p * p);
  hipLaunchKernelGGL(k_keep_sums, dim3((unsigned)gx, (unsigned)p), dim3(256), 0, nullptr, d_U, m, ld, d_wd.p, d0, 0, d_pi.p, d_pj.p,
                     (const double *)nullptr, d_part.p);
  BSN_HIP(hipGetLastError());
  BSN_HIP(hipMemcpy(part.data(), d_part.p, (size_t)p * gx * 2 * 8, hipMemcpyDeviceToHost));
  double nk = 0;
  for (int c = 0; c < p; c++) {
    double a = 0, b = 0;
    for (int t = 0; t < gx; t++) {
      a += part[((size_t)c * gx + t) * 2];
      b += part[((size_t)c * gx + t) * 2 + 1];
    }
    centre[(size_t)c] = a / b;
    nk = b;
  }
  if (n_kept_out) *n_kept_out = (int64_t)nk;
  if (!(nk >= 2)) fail("dist_ogk: fewer than two rows pass the hard rejection");
  BSN_HIP(hipMemcpy(d_centre.ensure((size_t)p), centre.data(), (size_t)p * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_keep_sums, dim3((unsigned)gx, (unsigned)nq), dim3(256), 0, nullptr, d_U, m, ld, d_wd.p, d0, 1, d_pi.p, d_pj.p,
                     d_centre.p, d_part.p);
  BSN_HIP(hipGetLastError());
  BSN_HIP(hipMemcpy(part.data(), d_part.p, (size_t)nq * gx * 2 * 8, hipMemcpyDeviceToHost));
  for (int q = 0; q < nq; q++) {
    double a = 0;
    for (int t = 0; t < gx; t++) a += part[((size_t)q * gx + t) * 2];
    cov[(size_t)pi[(size_t)q] + (size_t)pj[(size_t)q] * p] = cov[(size_t)pj[(size_t)q] + (size_t)pi[(size_t)q] * p] = a / (nk - 1.0);
  }
  // pseudo-inverse of the covariance as numpy.linalg.pinv takes it: singular values <= 1e-15 x the largest are dropped
  std::vector<double> V = cov, lam, P((size_t)p * p, 0.0);
  eig_sym(p, V, lam);
  double lmax = 0;
  for (double v : lam) lmax = std::max(lmax, std::fabs(v));
  if (!std::isfinite(lmax)) fail("dist_ogk: the covariance of the kept rows is not finite");
  for (int e = 0; e < p; e++) {
    if (!(std::fabs(lam[(size_t)e]) > 1e-15 * lmax)) continue;
    const double inv = 1.0 / lam[(size_t)e];
    for (int c = 0; c < p; c++)
      for (int r = 0; r < p; r++) P[(size_t)r + (size_t)c * p] += V[(size_t)r + (size_t)e * p] * inv * V[(size_t)c + (size_t)e * p];
  }
  BSN_HIP(hipMemcpy(d_P.ensure((size_t)p * p), P.data(), (size_t)p * p * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_mahalanobis, dim3((unsigned)((m + 255) / 256)), dim3(256), (size_t)p * p * 8, nullptr, d_U, m, ld, p, d_centre.p,
                     d_P.p, d_wd.p);
  BSN_HIP(hipGetLastError());
  BSN_HIP(hipMemcpy(dist_out, d_wd.p, (size_t)m * 8, hipMemcpyDeviceToHost));
}

}  // namespace
}  // namespace bsn

using namespace bsn;

extern "C" {

int bsn_robust_scale_tau2(const double *d_X, int64_t m, int64_t ld, int32_t ncol, double c1, double c2, double *mu_out, double *s_out) {
  return guarded([&] {
    require_gpu();
    if (m < 1 || ncol < 0 || ld < m) fail("bsn_robust_scale_tau2: dimensions");
    Scratch w;
    tau2_columns(w, d_X, m, ld, ncol, c1, c2, mu_out, s_out, nullptr);
  });
}

int bsn_robust_pair_scales(const double *d_Z, int64_t m, int64_t ld, int32_t p, double c1, double c2, double *s_sum_out, double *s_diff_out) {
  return guarded([&] {
    require_gpu();
    if (m < 1 || p < 1 || p > 64 || ld < m) fail("bsn_robust_pair_scales: dimensions (at most 64 columns)");
    Scratch w;
    pair_scales(w, d_Z, m, ld, p, c1, c2, s_sum_out, s_diff_out);
  });
}

int bsn_robust_mc_count(const double *d_up, int64_t nu, const double *d_lo, int64_t nl, double t, int64_t *count_out) {
  return guarded([&] {
    require_gpu();
    if (nu < 0 || nl < 0 || !count_out) fail("bsn_robust_mc_count: arguments");
    *count_out = 0;
    if (nu == 0 || nl == 0) return;
    DevBuf<unsigned long long> d_c;
    d_c.ensure(1);
    BSN_HIP(hipMemsetAsync(d_c.p, 0, 8, nullptr));
    const int gx = (int)std::min<int64_t>(1024, (nu + 255) / 256);
    hipLaunchKernelGGL(k_mc_count, dim3((unsigned)gx), dim3(256), 0, nullptr, d_up, nu, d_lo, nl, t, d_c.p);
    BSN_HIP(hipGetLastError());
    unsigned long long c = 0;
    BSN_HIP(hipMemcpy(&c, d_c.p, 8, hipMemcpyDeviceToHost));
    *count_out = (int64_t)c;
  });
}

int bsn_robust_scale_cols(double *d_Z, int64_t m, int64_t ld, int32_t p, const double *div) {
  return guarded([&] {
    require_gpu();
    if (m < 1 || p < 1 || ld < m) fail("bsn_robust_scale_cols: dimensions");
    scale_cols(d_Z, m, ld, p, div);
  });
}

int bsn_robust_rotate(double *d_Z, int64_t m, int64_t ld, int32_t p, const double *E) {
  return guarded([&] {
    require_gpu();
    if (m < 1 || p < 1 || p > 64 || ld < m) fail("bsn_robust_rotate: dimensions (at most 64 columns)");
    rotate(d_Z, m, ld, p, E);
  });
}

int bsn_robust_wdist(const double *d_Z, int64_t m, int64_t ld, int32_t p, const double *mu, const double *sig, double *out) {
  return guarded([&] {
    require_gpu();
    if (m < 1 || p < 1 || ld < m) fail("bsn_robust_wdist: dimensions");
    DevBuf<double> d_out;
    d_out.ensure((size_t)m);
    wdist(d_Z, m, ld, p, mu, sig, d_out.p);
    BSN_HIP(hipMemcpy(out, d_out.p, (size_t)m * 8, hipMemcpyDeviceToHost));
  });
}

int bsn_robust_dist_ogk(const double *d_U, int64_t m, int64_t ld, int32_t p, int32_t niter, double cut_ratio, double c1, double c2,
                        double *dist_out, int64_t *n_kept_out) {
  return guarded([&] {
    require_gpu();
    if (m < 2 || p < 1 || p > 64 || ld < m || niter < 0 || !dist_out) fail("bsn_robust_dist_ogk: dimensions (at most 64 columns)");
    dist_ogk(d_U, m, ld, p, niter, cut_ratio, c1, c2, dist_out, n_kept_out);
  });
}

int bsn_robust_rollmean(const double *x, int64_t m, const double *w, int32_t len, const int64_t *group_off, int32_t ngroups, double *out) {
  return guarded([&] {
    require_gpu();
    if (m < 1 || len < 1 || (len & 1) == 0 || len > 4095 || !x || !w || !out) fail("bsn_robust_rollmean: an odd number of weights (at most 4095)");
    std::vector<long long> off;
    if (group_off && ngroups > 0) {
      off.assign(group_off, group_off + ngroups + 1);
      if (off.front() != 0 || off.back() != m) fail("bsn_robust_rollmean: the groups do not cover the vector");
      for (int g = 0; g < ngroups; g++)
        if (off[(size_t)g + 1] <= off[(size_t)g]) fail("bsn_robust_rollmean: empty group");
    } else {
      off = {0, (long long)m};
    }
    const int ng = (int)off.size() - 1;
    DevBuf<double> d_x, d_w, d_out;
    DevBuf<long long> d_off;
    BSN_HIP(hipMemcpy(d_x.ensure((size_t)m), x, (size_t)m * 8, hipMemcpyHostToDevice));
    BSN_HIP(hipMemcpy(d_w.ensure((size_t)len), w, (size_t)len * 8, hipMemcpyHostToDevice));
    BSN_HIP(hipMemcpy(d_off.ensure(off.size()), off.data(), off.size() * 8, hipMemcpyHostToDevice));
    d_out.ensure((size_t)m);
    hipLaunchKernelGGL(k_rollmean, dim3((unsigned)((m + 255) / 256)), dim3(256), (size_t)len * 8, nullptr, d_x.p, m, d_w.p, len, d_off.p, ng,
                       d_out.p);
    BSN_HIP(hipGetLastError());
    BSN_HIP(hipMemcpy(out, d_out.p, (size_t)m * 8, hipMemcpyDeviceToHost));
  });
}

int bsn_robust_sort(double *x, int64_t m) {
  return guarded([&] {
    require_gpu();
    if (m < 0 || (m > 0 && !x)) fail("bsn_robust_sort: arguments");
    if (m < 2) return;
    DevBuf<double> d_in, d_out;
    DevBuf<char> d_tmp;
    BSN_HIP(hipMemcpy(d_in.ensure((size_t)m), x, (size_t)m * 8, hipMemcpyHostToDevice));
    d_out.ensure((size_t)m);
    size_t tmp = 0;
    BSN_HIP(rocprim::radix_sort_keys(nullptr, tmp, d_in.p, d_out.p, (size_t)m, 0, 64, (hipStream_t) nullptr));
    d_tmp.ensure(std::max<size_t>(tmp, 1));
    BSN_HIP(rocprim::radix_sort_keys((void *)d_tmp.p, tmp, d_in.p, d_out.p, (size_t)m, 0, 64, (hipStream_t) nullptr));
    BSN_HIP(hipMemcpy(x, d_out.p, (size_t)m * 8, hipMemcpyDeviceToHost));
  });
}

int bsn_robust_mc_window(const double *d_up, int64_t nu, const double *d_lo, int64_t nl, double a, double b, int64_t cap, double *out,
                         int64_t *count_out) {
  return guarded([&] {
    require_gpu();
    if (nu < 0 || nl < 0 || cap < 0 || !count_out || (cap > 0 && !out) || !(a > -1.0) || !(b < 1.0) || !(a < b))
      fail("bsn_robust_mc_window: arguments (-1 < a < b < 1)");
    *count_out = 0;
    if (nu == 0 || nl == 0) return;
    DevBuf<unsigned long long> d_c;
    DevBuf<double> d_out;
    d_c.ensure(1);
    d_out.ensure((size_t)std::max<int64_t>(cap, 1));
    BSN_HIP(hipMemsetAsync(d_c.p, 0, 8, nullptr));
    const int gx = (int)std::min<int64_t>(1024, (nu + 255) / 256);
    hipLaunchKernelGGL(k_mc_window, dim3((unsigned)gx), dim3(256), 0, nullptr, d_up, nu, d_lo, nl, a, b, (unsigned long long)cap, d_c.p,
                       d_out.p);
    BSN_HIP(hipGetLastError());
    unsigned long long c = 0;
    BSN_HIP(hipMemcpy(&c, d_c.p, 8, hipMemcpyDeviceToHost));
    *count_out = (int64_t)c;
    if (c > 0 && c <= (unsigned long long)cap) BSN_HIP(hipMemcpy(out, d_out.p, (size_t)c * 8, hipMemcpyDeviceToHost));
  });
}

int bsn_order_decreasing(const double *S, int64_t m, const int64_t *group_off, int32_t ngroups, int32_t *ord, int32_t *rank) {
  return guarded([&] {
    require_gpu();
    if (m < 0 || m > 0x7fffffff || (m > 0 && (!S || !ord || !rank))) fail("bsn_order_decreasing: arguments");
    if (m == 0) return;
    std::vector<long long> off;
    if (group_off && ngroups > 0) {
      off.assign(group_off, group_off + ngroups + 1);
      if (off.front() != 0 || off.back() != m) fail("bsn_order_decreasing: the groups do not cover the vector");
      for (int g = 0; g < ngroups; g++)
        if (off[(size_t)g + 1] <= off[(size_t)g]) fail("bsn_order_decreasing: empty group");
    } else {
      off = {0, (long long)m};
    }
    const int ng = (int)off.size() - 1;
    for (int64_t i = 0; i < m; i++)
      if (std::isnan(S[i])) fail("bsn_order_decreasing: NaN (R's order puts NA last; the host path handles them)");
    DevBuf<double> k_in, k_out;
    DevBuf<int> v_in, v_out, d_ord, d_rank;
    DevBuf<unsigned int> grp, g_in, g_out;
    DevBuf<long long> d_off;
    DevBuf<char> d_tmp;
    std::vector<double> key(S, S + m);
    for (auto &v : key) v += 0.0;                                        // -0 and +0 are one value to R's order
    BSN_HIP(hipMemcpy(k_in.ensure((size_t)m), key.data(), (size_t)m * 8, hipMemcpyHostToDevice));
    BSN_HIP(hipMemcpy(d_off.ensure(off.size()), off.data(), off.size() * 8, hipMemcpyHostToDevice));
    k_out.ensure((size_t)m);
    v_in.ensure((size_t)m);
    v_out.ensure((size_t)m);
    const unsigned gx = (unsigned)((m + 255) / 256);
    hipLaunchKernelGGL(k_iota_i32, dim3(gx), dim3(256), 0, nullptr, v_in.p, m);
    // 1. all values, descending, stable: ties keep their index order
    size_t tmp = 0;
    BSN_HIP(rocprim::radix_sort_pairs_desc(nullptr, tmp, k_in.p, k_out.p, v_in.p, v_out.p, (size_t)m, 0, 64, (hipStream_t) nullptr));
    d_tmp.ensure(std::max<size_t>(tmp, 1));
    BSN_HIP(rocprim::radix_sort_pairs_desc((void *)d_tmp.p, tmp, k_in.p, k_out.p, v_in.p, v_out.p, (size_t)m, 0, 64, (hipStream_t) nullptr));
    const int *val = v_out.p;
    grp.ensure((size_t)m);
    g_in.ensure((size_t)m);
    hipLaunchKernelGGL(k_group_of, dim3(gx), dim3(256), 0, nullptr, d_off.p, ng, m, grp.p);
    hipLaunchKernelGGL(k_gather_u32, dim3(gx), dim3(256), 0, nullptr, grp.p, v_out.p, m, g_in.p);
    const unsigned int *gs = g_in.p;
    if (ng > 1) {   // 2. by group, ascending, stable: every group keeps the order of step 1
      int bits = 1;
      while ((1 << bits) < ng) bits++;
      g_out.ensure((size_t)m);
      size_t tmp2 = 0;
      BSN_HIP(rocprim::radix_sort_pairs(nullptr, tmp2, g_in.p, g_out.p, v_out.p, v_in.p, (size_t)m, 0, (unsigned)bits, (hipStream_t) nullptr));
      d_tmp.ensure(std::max<size_t>(tmp2, 1));
      BSN_HIP(rocprim::radix_sort_pairs((void *)d_tmp.p, tmp2, g_in.p, g_out.p, v_out.p, v_in.p, (size_t)m, 0, (unsigned)bits, (hipStream_t) nullptr));
      val = v_in.p;
      gs = g_out.p;
    }
    d_ord.ensure((size_t)m);
    d_rank.ensure((size_t)m);
    hipLaunchKernelGGL(k_order_finish, dim3(gx), dim3(256), 0, nullptr, val, gs, d_off.p, m, d_ord.p, d_rank.p);
    BSN_HIP(hipGetLastError());
    BSN_HIP(hipMemcpy(ord, d_ord.p, (size_t)m * 4, hipMemcpyDeviceToHost));
    BSN_HIP(hipMemcpy(rank, d_rank.p, (size_t)m * 4, hipMemcpyDeviceToHost));
  });
}

}  // extern "C"
