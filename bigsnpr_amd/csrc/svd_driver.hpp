// svd_driver.hpp — partial SVD of the scaled genotype matrix by block Lanczos with full
// re-orthogonalisation on A A' (vectors of length n = number of samples).
//
// Replaces the SVD driver that bed_randomSVD delegates to (R/autoSVD.R:205-219 ->
// bigstatsr::big_randomSVD -> RSpectra::svds, both external to the reference tree): the
// same contract — the k largest singular triplets (d, u, v) of A~, found with A x / A' x
// products only — but with BLOCKS of b vectors per pass, because on MI355X one streaming
// pass over the 2-bit image costs the same for 1 or 8 vectors (the i8 MFMA pipe is idle
// otherwise), so a block method divides the number of HBM passes by ~b.
//
// Columns (variants) may be sharded over ranks: Z = A' Q is local to the shard,
// W = A Z is summed over ranks by the backend (one all-reduce of n x b doubles per
// step); everything else is replicated and deterministic, so all ranks take identical
// decisions.  The backend interface keeps this file free of HIP so that the identical
// driver is exercised on CPU (tests/native) with a host backend.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "dense_small.hpp"

namespace bsn {

struct SvdBackend {
  int64_t n = 0, m_local = 0, m_total = 0;
  virtual ~SvdBackend() {}
  virtual void alloc(int cap, int b) = 0;
  virtual void random_W(int b, uint32_t seed) = 0;          // W[:, :b] = deterministic noise
  virtual void At_Qblock(int p0, int cb) = 0;               // Z[:, p0:p0+cb] = A' Q[:, p0:p0+cb]
  virtual void A_Zblock(int p0, int cb) = 0;                // W[:, :cb] = sum_ranks A Z[:, p0:p0+cb]
  virtual void QtW(int p, int cb, double *C) = 0;           // C (p x cb) = Q[:, :p]' W[:, :cb]
  virtual void W_minus_QC(int p, int cb, const double *C) = 0;
  virtual void WtW(int cb, double *G) = 0;                  // G (cb x cb) = W' W
  virtual void W_times(int cb, int r, const double *M) = 0; // W[:, :r] = W[:, :cb] M (cb x r)
  virtual void W_to_Q(int p0, int r) = 0;                   // Q[:, p0:p0+r] = W[:, :r]
  // Optional fused form of the whole orthonormalisation step below (same arithmetic, same
  // outputs) for backends that can run it without returning to the host between its parts.
  // Returns the rank (== cb) on success; -1 if unsupported or if W turned out rank deficient,
  // in which case W is left as on entry and the driver takes the step-by-step path.
  virtual int orth_fused(int p, int cb, std::vector<double> &Cacc, std::vector<double> &Rout) {
    (void)p; (void)cb; (void)Cacc; (void)Rout;
    return -1;
  }
  // u (n x k) = Q[:, :pp] S ; v (m_local x k) = Z[:, :pp] S diag(dinv); host outputs
  virtual void finalize(int pp, int k, const double *S, const double *dinv, double *u,
                        double *v) = 0;
};

struct SvdOptions {
  int k = 10;
  double tol = 1e-4;
  int block = 8;
  int max_basis = 0;  // 0 -> chosen from k and block
  uint32_t seed = 1;
  int verbose = 0;
};

struct SvdResult {
  int niter = 0;      // block steps
  int nops = 0;       // block applications of A or A' (each is one pass over the matrix)
  int basis = 0;      // Krylov basis size at exit
  int converged = 0;  // 1 if all k residuals met tol
  double max_rel_resid = 0;
};

// d: k singular values (descending); u: n x k; v: m_local x k (column-major, host)
inline SvdResult block_lanczos_svd(SvdBackend &bk, const SvdOptions &opt, double *d, double *u,
                                   double *v) {
  const int k = opt.k;
  int b = opt.block;
  {
    const int64_t small = bk.n < bk.m_total ? bk.n : bk.m_total;
    if (b > small) b = (int)small;
  }
  // The Krylov space of A A' started from a random block of R^n has dimension at most
  // rank(A) + b: the start block is not in range(A) when n > m.
  const int64_t dim = bk.n < bk.m_total + b ? bk.n : bk.m_total + b;
  int cap = opt.max_basis > 0 ? opt.max_basis : std::max(8 * k + 4 * b, 320);
  if (cap > dim) cap = (int)dim;
  if (cap < k) cap = k;
  bk.alloc(cap + b, b);

  std::vector<double> T((size_t)cap * cap, 0.0);
  auto Tat = [&](int i, int j) -> double & { return T[(size_t)i + (size_t)j * cap]; };
  std::vector<double> C, C2, G, R, Ri, R2, Rt, evec, eval, M;
  SvdResult res;

  // orthonormalise W (cb columns) against Q[:, :p] and itself; returns rank r and the
  // cb x cb upper factor Rt with W_in = Q C + W_out Rt  (first r rows of Rt meaningful)
  auto orth = [&](int p, int cb, std::vector<double> &Cacc, std::vector<double> &Rout) -> int {
    {
      const int rf = bk.orth_fused(p, cb, Cacc, Rout);
      if (rf >= 0) return rf;
    }
    Cacc.assign((size_t)p * cb, 0.0);
    G.assign((size_t)cb * cb, 0.0);
    bk.WtW(cb, G.data());
    double w0 = 0;
    for (int i = 0; i < cb; i++) w0 = std::max(w0, G[(size_t)i + (size_t)i * cb]);
    for (int pass = 0; pass < 2 && p > 0; pass++) {
      C.assign((size_t)p * cb, 0.0);
      bk.QtW(p, cb, C.data());
      bk.W_minus_QC(p, cb, C.data());
      for (size_t t = 0; t < C.size(); t++) Cacc[t] += C[t];
    }
    Rout.assign((size_t)cb * cb, 0.0);
    for (int i = 0; i < cb; i++) Rout[(size_t)i + (size_t)i * cb] = 1.0;
    int r = cb;
    for (int pass = 0; pass < 2; pass++) {
      G.assign((size_t)r * r, 0.0);
      bk.WtW(r, G.data());
      // absolute deficiency test against the pre-projection scale (Krylov exhaustion)
      if (pass == 0) {
        int keep = 0;
        while (keep < r && G[(size_t)keep + (size_t)keep * r] > 1e-22 * w0 && w0 > 0) keep++;
        if (keep < r) {
          // shrink: keep the leading `keep` columns only
          std::vector<double> G2((size_t)keep * keep);
          for (int j = 0; j < keep; j++)
            for (int i = 0; i < keep; i++) G2[(size_t)i + (size_t)j * keep] = G[(size_t)i + (size_t)j * r];
          G.swap(G2);
          r = keep;
        }
      }
      if (r == 0) return 0;
      int rk = chol_upper(r, G, R, 1e-22);
      if (rk < r) {
        std::vector<double> G2((size_t)rk * rk);
        for (int j = 0; j < rk; j++)
          for (int i = 0; i < rk; i++) G2[(size_t)i + (size_t)j * rk] = G[(size_t)i + (size_t)j * r];
        r = rk;
        if (r == 0) return 0;
        chol_upper(r, G2, R, 0.0);
      }
      inv_upper(r, R, Ri);
      bk.W_times(r, r, Ri.data());
      // Rout <- R * Rout (leading r x cb part)
      std::vector<double> Rn((size_t)cb * cb, 0.0);
      for (int j = 0; j < cb; j++)
        for (int i = 0; i < r; i++) {
          double s = 0;
          for (int t = i; t < r; t++) s += R[(size_t)i + (size_t)t * r] * Rout[(size_t)t + (size_t)j * cb];
          Rn[(size_t)i + (size_t)j * cb] = s;
        }
      Rout.swap(Rn);
      if (pass == 0 && p > 0) {  // one more projection after the first normalisation
        C.assign((size_t)p * r, 0.0);
        bk.QtW(p, r, C.data());
        bk.W_minus_QC(p, r, C.data());
      }
    }
    return r;
  };

  // start block
  bk.random_W(b, opt.seed);
  int r = orth(0, b, C2, Rt);
  if (r == 0) r = 0;
  bk.W_to_Q(0, r);
  int p = r;      // basis size (columns of Q filled)
  int cb = r;     // size of the newest block
  int pp = 0;     // number of leading columns whose T columns are complete
  std::vector<double> Rlast;  // coupling block (r_next x cb) of the newest complete block
  int rl_rows = 0, rl_cols = 0;

  while (cb > 0) {
    const int p0 = p - cb;
    bk.At_Qblock(p0, cb);
    bk.A_Zblock(p0, cb);
    res.nops += 2;
    res.niter++;
    int rn = orth(p, cb, C2, Rt);  // C2: p x cb, Rt: cb x cb (rn rows used)
    // T[:, p0:p] = C2 (and mirror)
    for (int j = 0; j < cb; j++)
      for (int i = 0; i < p; i++) {
        Tat(i, p0 + j) = C2[(size_t)i + (size_t)j * p];
        Tat(p0 + j, i) = C2[(size_t)i + (size_t)j * p];
      }
    for (int j = 0; j < cb; j++)  // symmetrise the diagonal block
      for (int i = 0; i < j; i++) {
        double a = 0.5 * (C2[(size_t)(p0 + i) + (size_t)j * p] + C2[(size_t)(p0 + j) + (size_t)i * p]);
        Tat(p0 + i, p0 + j) = a;
        Tat(p0 + j, p0 + i) = a;
      }
    pp = p;
    const bool exhausted = (rn == 0) || (p >= dim);  // Krylov space is invariant: Ritz pairs exact
    if (rn > cap - p) rn = cap - p;
    if (rn < 0) rn = 0;
    Rlast.assign((size_t)rn * cb, 0.0);
    for (int j = 0; j < cb; j++)
      for (int i = 0; i < rn; i++) Rlast[(size_t)i + (size_t)j * rn] = Rt[(size_t)i + (size_t)j * cb];
    rl_rows = rn;
    rl_cols = cb;

    // Rayleigh-Ritz on the complete part
    evec.assign((size_t)pp * pp, 0.0);
    for (int j = 0; j < pp; j++)
      for (int i = 0; i < pp; i++) evec[(size_t)i + (size_t)j * pp] = Tat(i, j);
    eig_sym(pp, evec, eval);
    bool done = false;
    if (pp >= k) {
      double worst = 0;
      for (int t = 0; t < k; t++) {
        int col = pp - 1 - t;
        double theta = eval[col];
        double rs = 0;  // || Rlast * s[last cb rows] ||
        for (int i = 0; i < rl_rows; i++) {
          double s = 0;
          for (int j = 0; j < rl_cols; j++)
            s += Rlast[(size_t)i + (size_t)j * rl_rows] * evec[(size_t)(pp - rl_cols + j) + (size_t)col * pp];
          rs += s * s;
        }
        double rel = std::sqrt(rs) / std::max(std::fabs(theta), 1e-300);
        worst = std::max(worst, rel);
      }
      res.max_rel_resid = worst;
      if (opt.verbose)
        std::fprintf(stderr, "[bsn svd] step %d basis %d max rel resid %.3e sigma1 %.6g\n", res.niter,
                     pp, worst, std::sqrt(std::max(eval[pp - 1], 0.0)));
      if (worst <= opt.tol) {
        done = true;
        res.converged = 1;
      }
    }
    if (done || rn == 0) {
      if (exhausted) res.converged = 1;
      break;
    }
    bk.W_to_Q(p, rn);
    // T[p:p+rn, p0:p] = Rlast
    for (int j = 0; j < cb; j++)
      for (int i = 0; i < rn; i++) {
        Tat(p + i, p0 + j) = Rlast[(size_t)i + (size_t)j * rn];
        Tat(p0 + j, p + i) = Rlast[(size_t)i + (size_t)j * rn];
      }
    p += rn;
    cb = rn;
  }

  // extract the k largest
  res.basis = pp;
  int kk = k < pp ? k : pp;
  std::vector<double> S((size_t)pp * k, 0.0), dinv((size_t)k, 0.0);
  for (int t = 0; t < k; t++) {
    d[t] = 0;
    if (t < kk) {
      int col = pp - 1 - t;
      double th = eval[col] > 0 ? eval[col] : 0;
      d[t] = std::sqrt(th);
      dinv[t] = d[t] > 0 ? 1.0 / d[t] : 0.0;
      for (int i = 0; i < pp; i++) S[(size_t)i + (size_t)t * pp] = evec[(size_t)i + (size_t)col * pp];
    }
  }
  bk.finalize(pp, k, S.data(), dinv.data(), u, v);
  return res;
}

}  // namespace bsn
