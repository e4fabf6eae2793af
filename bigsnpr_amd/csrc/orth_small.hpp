// orth_small.hpp — the small-matrix half of the two-pass block orthonormalisation of the SVD driver.
//
// A block step has to make the new panel W (n x cb) orthonormal to the stored basis Q (n x p) and to
// itself.  The tall operands only ever enter through Gram blocks, so the whole step is two rounds of
//     [H; G] = [Q W]' W            one tall product, ONE sum over the ranks of (p + cb) x cb doubles
//     small matrices from H, G     this file: one workgroup on the device, plain loops on the host
//     W <- (W - Q C) R^-1          one tall update
// instead of the eleven tall operations (and eight small sums over the ranks) of the step-by-step
// orthonormalisation in svd_driver.hpp, which stays as the careful path for rank-deficient blocks.
//
// The stored basis is orthonormal only up to the rounding of its blocks to the product grid
// (M = Q'Q = I + E, |E| ~ 2^-8S), so the projection coefficients are C = M^-1 H — a few terms of the
// Neumann series, M is known exactly from the Gram blocks the Rayleigh-Ritz step needs anyway — which
// removes the component along span(Q) to rounding level in ONE projection (the step-by-step path needs
// three with C = H).  Pass 0 takes the Gram matrix of the projected panel by downdating,
// G1 = G - H' C, which is only as accurate as eps |W|^2: enough for a first Cholesky factor that brings
// the panel close to orthonormal; pass 1 repeats projection and factorisation on that panel, where
// H is at rounding level and the downdate exact (CGS2 + CholeskyQR2, "twice is enough").  A panel
// that lost more than six digits in the projection, or whose pass-1 Gram matrix is not close to the
// identity, raises `flag` and the driver redoes the step on the careful path.
//
// The same code runs on the device (svd.hip: Ctx = one workgroup, sync = __syncthreads) and on the
// host (tests/native: Ctx = one thread), so the CPU tests exercise the arithmetic the GPU runs.
#pragma once
#include <cmath>

#ifdef __HIPCC__
#define BSN_HD __host__ __device__
#else
#define BSN_HD
#endif

namespace bsn {

constexpr int kOrthMaxB = 16;    // columns of a panel
constexpr int kOrthMaxP = 384;   // basis columns the fused step supports (device LDS: p x cb doubles)

struct OrthSmall {
  int p, cb, pass;      // basis size, panel width, pass 0 / 1
  int p0;               // pass 0: the Gram block `QtQ` (p x cb, columns p0 .. p0+cb-1 of M) is folded into M first
  const double *QtQ;    // may be null (p == 0); leading dimension ldq
  int ldq;
  const double *HG;     // (p + cb) x cb, leading dimension p + cb: rows < p = Q'W, rows >= p = W'W
  double *M;            // p x p Gram matrix of the stored basis, leading dimension ldm (updated in pass 0)
  int ldm;
  double *C;            // out: p x cb projection coefficients
  double *Ct;           // scratch p x cb (each row is written and read by the same thread only)
  double *Ri;           // out: cb x cb inverse Cholesky factor of this pass
  double *Rout;         // pass 0 out: R1; pass 1 in/out: R2 * R1   (W_in = Q (..) + W_out Rout)
  double *flag;         // raised (1.0) when the panel needs the careful path
  int iters;            // Neumann terms beyond the first
  // scratch shared by the threads: Cs p x cb, Gs / Rs / Ris / Ro / Dv cb x cb, tmp 2 * cb + 4
  double *Cs, *Gs, *Rs, *Ris, *Ro, *Dv, *tmp;
};

// Ctx: int tid, nt; void sync();
template <class Ctx>
BSN_HD inline void orth_small(Ctx &cx, const OrthSmall &a) {
  const int p = a.p, cb = a.cb, ld = a.p + a.cb, tid = cx.tid, nt = cx.nt;
  const double *H = a.HG, *G = a.HG + p;   // G[i + j * ld]
  // fold the newest Gram block into M (both triangles)
  if (a.pass == 0 && a.QtQ && p > 0) {
    for (int t = tid; t < p * cb; t += nt) {
      const int i = t % p, j = t / p;
      const double g = a.QtQ[(long)i + (long)j * a.ldq];
      a.M[(long)i + (long)(a.p0 + j) * a.ldm] = g;
      // the diagonal block is taken as computed (one writer per element: results stay run-to-run identical)
      if (i < a.p0) a.M[(long)(a.p0 + j) + (long)i * a.ldm] = g;
    }
  }
  for (int t = tid; t < p * cb; t += nt) a.Cs[t] = H[(t % p) + (t / p) * ld];   // Cs: p x cb, ld p
  cx.sync();
  // C <- H - (M - I) C, `iters` times: C = (I - E + E^2 - ...) H
  for (int it = 0; it < a.iters && p > 0; it++) {
    // one coefficient per thread and turn (unrolled: one global load of M per term — rolled, every term
    // would wait out a memory round trip)
    for (int t = tid; t < p * cb; t += nt) {
      const int r = t % p, j = t / p;
      double acc = 0.0;
#pragma unroll 8
      for (int b = 0; b < p; b++) acc += (a.M[(long)r + (long)b * a.ldm] - (b == r ? 1.0 : 0.0)) * a.Cs[b + j * p];
      a.Ct[t] = H[r + j * ld] - acc;
    }
    cx.sync();
    for (int t = tid; t < p * cb; t += nt) a.Cs[t] = a.Ct[t];
    cx.sync();
  }
  for (int t = tid; t < p * cb; t += nt) a.C[t] = a.Cs[t];
  // Gram matrix of the projected panel by downdating, symmetrised: Gs = G - H' C
  for (int t = tid; t < cb * cb; t += nt) {
    const int i = t % cb, j = t / cb;
    if (i > j) continue;
    double s1 = 0, s2 = 0;
#pragma unroll 8
    for (int r = 0; r < p; r++) {
      s1 += H[r + i * ld] * a.Cs[r + j * p];
      s2 += H[r + j * ld] * a.Cs[r + i * p];
    }
    const double v = 0.5 * ((G[i + j * ld] - s1) + (G[j + i * ld] - s2));
    a.Gs[i + j * cb] = v;
    a.Gs[j + i * cb] = v;
  }
  for (int t = tid; t < cb * cb; t += nt) {
    a.Rs[t] = 0.0;
    a.Ris[t] = 0.0;
    a.Ro[t] = a.pass == 0 ? 0.0 : a.Rout[t];
    a.Dv[t] = G[(t % cb) + (t / cb) * ld];   // the incoming Gram matrix, for the checks below
  }
  cx.sync();
  double *bad = a.tmp + 2 * cb, *piv_min = a.tmp + 2 * cb + 1;
  if (tid == 0) {
    double g0 = 0, dev = 0;
    for (int i = 0; i < cb; i++) g0 = fmax(g0, a.Dv[i + i * cb]);
    if (a.pass == 1)
      for (int j = 0; j < cb; j++)
        for (int i = 0; i < cb; i++) dev = fmax(dev, fabs(a.Dv[i + j * cb] - (i == j ? 1.0 : 0.0)));
    // pass 0: a pivot below 1e-12 of the largest squared column norm of the incoming panel means the
    // downdated Gram matrix has no digits left; pass 1: the panel should already be near orthonormal
    *piv_min = a.pass == 0 ? 1e-12 * g0 : 1e-3;
    *bad = (!(g0 > 0) || !(dev <= 0.25)) ? 1.0 : 0.0;
  }
  cx.sync();
  // Cholesky Gs = Rs' Rs, row by row; the entries of a row are independent
  for (int j = 0; j < cb; j++) {
    for (int i = j + tid; i < cb; i += nt) {
      double t = a.Gs[j + i * cb];
      for (int k = 0; k < j; k++) t -= a.Rs[k + j * cb] * a.Rs[k + i * cb];
      a.tmp[i] = t;
    }
    cx.sync();
    const double piv = a.tmp[j];
    const bool okp = piv > *piv_min;
    const double rjj = okp ? sqrt(piv) : 1.0;
    for (int i = j + tid; i < cb; i += nt) a.Rs[j + i * cb] = i == j ? rjj : a.tmp[i] / rjj;
    if (tid == 0 && !okp) *bad = 1.0;
    cx.sync();
  }
  // inverse of the upper factor, one column per thread
  for (int c = tid; c < cb; c += nt) {
    a.Ris[c + c * cb] = 1.0 / a.Rs[c + c * cb];
    for (int i = c - 1; i >= 0; i--) {
      double s = 0;
      for (int k = i + 1; k <= c; k++) s += a.Rs[i + k * cb] * a.Ris[k + c * cb];
      a.Ris[i + c * cb] = -s / a.Rs[i + i * cb];
    }
  }
  cx.sync();
  const bool isbad = *bad != 0.0;
  for (int t = tid; t < cb * cb; t += nt) {
    const int i = t % cb, j = t / cb;
    // keep the following tall kernels finite when the step is going to be redone
    a.Ri[t] = isbad ? (i == j ? 1.0 : 0.0) : a.Ris[t];
    if (!isbad) {
      double s;
      if (a.pass == 0) {
        s = a.Rs[t];
      } else {
        s = 0;
        for (int k = i; k < cb; k++) s += a.Rs[i + k * cb] * a.Ro[k + j * cb];
      }
      a.Rout[t] = s;
    }
  }
  if (isbad) {
    for (int t = tid; t < p * cb; t += nt) a.C[t] = 0.0;
    if (tid == 0) *a.flag = 1.0;
  }
  cx.sync();
}

}  // namespace bsn
