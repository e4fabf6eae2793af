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
// Accuracy of the Ritz values does not depend on the precision of the products: a finished
// basis block is rounded to the fixed-point grid of the backend (round_W), so Z = A' Q is the
// EXACT product for the stored Q, and the projected problem is the generalised pair
//   (Z'Z) s = theta (Q'Q) s
// — the Galerkin condition on span(Q) in exact arithmetic, whatever rounding the expansion
// W = A Z went through.  The expansion direction and the residual estimate (from the coupling
// block of the orthonormalisation) see first-order rounding effects only, which perturb the
// subspace, not the Ritz values computed on it.
//
// Round 3: a block step asks the backend for its fused form first (step_fused: Gram blocks + two-pass
// orthonormalisation without returning to the host, orth_small.hpp) and keeps the step-by-step
// orthonormalisation below as the careful path for rank-deficient panels; a basis that fills up is compressed to
// the best Ritz vectors and the iteration continues (thick restart); a numerically dependent basis (Krylov
// exhaustion under rounded products, k > rank) is handled by canonical orthogonalisation in the Rayleigh-Ritz step.
//
// Columns (variants) may be sharded over ranks: Z = A' Q is local to the shard,
// W = A Z is summed over ranks by the backend (one all-reduce of n x b doubles per
// step); everything else is replicated and deterministic, so all ranks take identical
// decisions.  The backend interface keeps this file free of HIP so that the identical
// driver is exercised on CPU (tests/native) with a host backend.
#pragma once
#include <string>
#include <stdexcept>
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
  // Exact-product Rayleigh-Ritz (see block_lanczos_svd): the backend rounds a finished block
  // to the grid on which its products are exact, and supplies the Gram blocks of the stored
  // Z = A' Q and of the stored (rounded) Q.
  virtual void round_W(int cb) = 0;                          // W[:, :cb] <- nearest grid point
  virtual void ZtZ(int p, int p0, int cb, double *G) = 0;    // G (p x cb) = sum_ranks Z[:, :p]' Z[:, p0:p0+cb]
  virtual void QtQ(int p, int p0, int cb, double *M) = 0;    // M (p x cb) = Q[:, :p]' Q[:, p0:p0+cb]
  // Optional fused form of the whole orthonormalisation step below (same arithmetic, same
  // outputs) for backends that can run it without returning to the host between its parts.
  // Returns the rank (== cb) on success; -1 if unsupported or if W turned out rank deficient,
  // in which case W is left as on entry and the driver takes the step-by-step path.
  virtual int orth_fused(int p, int cb, std::vector<double> &Cacc, std::vector<double> &Rout) {
    (void)p; (void)cb; (void)Cacc; (void)Rout;
    return -1;
  }
  // Optional fused form of a whole block step after the two products: the Gram blocks of the newest
  // basis block (what ZtZ / QtQ deliver: blkZ, blkQ, p x cb each) AND the orthonormalisation of W,
  // without returning to the host in between.  Returns the rank (== cb) on success; -1 if unsupported
  // (nothing was done); -2 if the Gram blocks were delivered but W — left as on entry — needs the
  // step-by-step orthonormalisation.
  virtual int step_fused(int p, int p0, int cb, double *blkZ, double *blkQ, std::vector<double> &Rout) {
    (void)p; (void)p0; (void)cb; (void)blkZ; (void)blkQ; (void)Rout;
    return -1;
  }
  // Thick restart (the basis is full, the residuals are not there yet): the basis becomes
  //   Q[:, :keep] = Q[:, :pp] S,  Z[:, :keep] = Z[:, :pp] S   (S: pp x keep, Ritz vectors to keep; Z stays A' Q),
  // and the orthonormalised next block W (rn columns, waiting behind column pp) moves behind column `keep`.
  // Returns false if the backend cannot (the driver then stops, unconverged, as before).
  // Mk (keep x keep) = S' (Q'Q) S, the Gram matrix of the kept vectors (the identity up to rounding unless the old
  // basis was ill conditioned): for backends that keep their own copy of Q'Q.
  virtual bool restart(int pp, int keep, const double *S, int rn, const double *Mk) {
    (void)pp; (void)keep; (void)S; (void)rn; (void)Mk;
    return false;
  }
  // Precision of what follows: the expansion W = A Z (the stored Z is rounded to 8 S bits on its way into the
  // product) and the grid the NEXT basis block is rounded to (hence the crossproduct pass that reads it).  Called
  // between the crossproduct and the product pass of a block step; backends with exact products ignore it.
  virtual void set_precision(int slices) { (void)slices; }
  // hold = true: the fused block step must not round the block it produces ahead of the driver's round_W (the
  // driver wants this step's residuals before it chooses that block's grid)
  virtual void hold_rounding(bool hold) { (void)hold; }
  // Warm start: restrict the two products to a leading subset of the variants (on) or restore all of
  // them (off).  Returns false if the backend has no cheap subset (then the start block stays random).
  virtual bool subset(bool on) {
    (void)on;
    return false;
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
  // power iterations of the random start block on a subset of the variants before the first full
  // pass (a fraction of a pass each): the Krylov space then starts inside the dominant subspace of a
  // thinned matrix instead of at noise, which lowers every later residual by a constant factor
  int warm = 0;
  // (Only the START block may come from a thinned operator.  Running early block steps with a thinned
  // expansion W = A_thin Z was tried in round 2: the basis then no longer satisfies A A' Q_j in
  // span(Q_1 .. Q_j+1), the residual of a Ritz pair is no longer carried by the last coupling block
  // alone, and the solver "converged" to sigma_1 off by 5e-4 with an estimated residual of 7e-6.)
  // relative residual that the rounding of the basis blocks leaves on a converged pair (about
  // 1.2 * 2^(-8 slices), measured); added to the estimate before it is compared with tol.  (Combining
  // the two in quadrature was tried: at 400K x 1M it stops the default solve one block step earlier, at
  // an estimate of 9.3e-5, but the TRUE residual of that solve is 1.23e-4 > tol — the estimate itself
  // carries first-order rounding effects.  tests/test_gpu_fullsize.py checks the true residuals.)
  double resid_floor = 0.0;
  // A full basis is compressed to the k + block best Ritz vectors and the iteration continues (what the
  // implicit restart of RSpectra::svds does for the reference, R/autoSVD.R:216-218); after max_restarts
  // compressions the solve gives up (RSpectra: maxitr).  < 0: never restart (a full basis ends the solve).
  int max_restarts = 100;
  // Precision schedule (round 5).  The rounding of step j (Z_j on its way into W = A Z_j, and the grid of Q_j+1) leaves
  //   A A' Q_j = [Q_1 .. Q_j+1] T_j + F_j,   |F_j| ~ 2^(-8 S_j) |A A' Q_j|,
  // and a Ritz vector y = Q s feels  sum_j F_j s_j: step j enters with the weight |s_j| of the vector's component in
  // block j — which is what the residual of that vector was one step earlier (inexact-Krylov relaxation: the early
  // steps need the precision, the late ones do not).  With slices_base digits throughout, every vector that converges
  // early — the leading ones — keeps a residual of 1.2 * 2^(-8 slices_base) (1.8e-5 at 16 bits: angles of 1e-4 .. 3e-4
  // to the true singular vectors, where a Lanczos solve in fp64 leaves them at 1e-7).  vec_floor > 0 asks for that
  // floor instead: step j runs with the smallest S in slices_base .. slices_max for which
  //   1.2 * 2^(-8 S) * rho_(j-1) <= vec_floor,   rho = largest relative residual of the LEADING HALF of the k pairs
  // after the previous step (1 before the first Rayleigh-Ritz step, which is taken as soon as the basis holds the
  // leading half).  The Ritz values, the residual estimate and the stopping rule are those of the uniform solve; only
  // the cost of the early passes changes.  The grid of Q_j+1 enters with the weight of the residual AFTER step j
  // (E_j+1 B_j+1 s_j: |B_j+1 s_j| is that residual), so while the schedule is wide the rounding of the new block
  // waits for the step's Rayleigh-Ritz result instead of being queued ahead of it.
  // F_j has two sources and they are scheduled apart: the grid of Q_j+1 (enters as E_j+1 B_j+1: the rule above, digits
  // of the NEXT crossproduct pass) and the digits of Z_j in the product pass (enters as A dZ, dZ white noise of the size
  // of Z_j's rounding: a random vector, which A amplifies by sqrt(|A|_F^2 / m) — sqrt(n) for standardised columns —
  // where it amplifies the block itself by its singular values).  noise_gain = that factor when the backend knows it
  // (0: unknown, the product pass then follows the rule above): the product pass runs with the smallest S for which
  //   1.2 * 2^(-8 S) * rho_(j-1) * min(1, noise_gain * sigma_1 / sigma_h^2) <= vec_floor,  sigma_h = smallest of the
  // leading half's Ritz values — on a genotype matrix with population structure (sigma >> sqrt(n)) one step earlier
  // narrow than the crossproduct pass.
  int slices_base = 0;     // 0: no schedule (set_precision is never called)
  int slices_max = 0;
  int slices_start = 0;    // grid of the start block (it only chooses where the iteration starts); 0 -> slices_base
  double vec_floor = 0.0;
  double noise_gain = 0.0;
};

struct SvdResult {
  int niter = 0;      // block steps
  int nops = 0;       // block applications of A or A' (each is one pass over the matrix)
  int basis = 0;      // Krylov basis size at exit
  int converged = 0;  // 1 if all k residuals met tol
  int warm = 0;       // warm-start iterations on the variant subset actually run
  int restarts = 0;   // thick restarts
  int exhausted = 0;  // the iteration ended because the Krylov space had no direction left (rank(A) + block <= basis)
  double exhausted_resid = 0;  // ... and what the coupling block of the last step says about the triplets that are not zero
  double max_rel_resid = 0;
  double lead_rel_resid = 0;   // the same over the leading half of the k pairs
  int wide_steps = 0;          // block steps that ran with more than slices_base digits (precision schedule)
  int slices_used_max = 0;
  // of the k requested triplets, those whose Ritz value lies between the hard zero (1e-10 theta_1: below what fp64 Gram
  // matrices resolve) and what the ROUNDED products resolve ((8 resid_floor)^2 theta_1): they were left out of the
  // convergence test because on these products they cannot meet it — which says nothing about their values.  A caller
  // that can should solve again on wider products (svd.hip does, in its automatic mode; with the digits fixed by the
  // caller the solve is reported as not converged).  0 on exact products.
  int below_resolution = 0;
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

  // projected pair: Gz = Z'Z, Mq = Q'Q (both cap x cap, symmetric, filled block column by block column)
  std::vector<double> Gz((size_t)cap * cap, 0.0), Mq((size_t)cap * cap, 0.0);
  auto Gat = [&](int i, int j) -> double & { return Gz[(size_t)i + (size_t)j * cap]; };
  auto Mat = [&](int i, int j) -> double & { return Mq[(size_t)i + (size_t)j * cap]; };
  std::vector<double> C, C2, G, R, Ri, R2, Rt, evec, eval, M, blk, blk2;
  SvdResult res;

  // orthonormalise W (cb columns) against Q[:, :p] and itself; returns rank r and the
  // cb x cb upper factor Rt with W_in = Q C + W_out Rt  (first r rows of Rt meaningful)
  const double defic = 1e-22;   // "no direction left" in a projected panel: relative to the squared pre-projection scale
  auto orth = [&](int p, int cb, std::vector<double> &Cacc, std::vector<double> &Rout, bool try_fused = true) -> int {
    if (try_fused) {
      const int rf = bk.orth_fused(p, cb, Cacc, Rout);
      if (rf >= 0) return rf;
    }
    Cacc.assign((size_t)p * cb, 0.0);
    G.assign((size_t)cb * cb, 0.0);
    bk.WtW(cb, G.data());
    double w0 = 0;
    for (int i = 0; i < cb; i++) w0 = std::max(w0, G[(size_t)i + (size_t)i * cb]);
    // Projection passes.  The stored basis is orthonormal only up to the rounding of its blocks, Q'Q = I + E with
    // |E| ~ 2^-8S of the COARSEST block — 4e-3 since round 5 starts on an 8-bit grid —, and W - Q (Q'W) leaves E times
    // what it removed.  "Twice is enough" holds for |E| ~ 1e-5; with an 8-bit block and a panel that loses six digits in
    // the projection (a spectrum over eight decades, 27 samples: found by the fixed sweep of tests/test_svd_driver_sweep_cpu.py,
    // round 6) two passes left MORE of the old directions (E^2 = 1.6e-5 of the panel) than there was of the new one: the
    // next block came out 0.6 - 0.8 inside span(Q), the Gram matrix of the basis had a condition of 1e10 and the exhausted
    // space returned its tenth and eleventh singular values 44 % and 103 % off under "converged".  So: project until what a
    // pass removes is at rounding level of what is LEFT of every column (at most eight passes; two when |E| is small).
    for (int pass = 0; pass < 8 && p > 0; pass++) {
      C.assign((size_t)p * cb, 0.0);
      bk.QtW(p, cb, C.data());
      bk.W_minus_QC(p, cb, C.data());
      for (size_t t = 0; t < C.size(); t++) Cacc[t] += C[t];
      if (pass < 1) continue;
      G.assign((size_t)cb * cb, 0.0);
      bk.WtW(cb, G.data());
      bool clean = true;
      for (int j = 0; j < cb && clean; j++) {
        const double left = G[(size_t)j + (size_t)j * cb];
        if (!(left > defic * w0)) continue;   // (nothing left of this column: the deficiency test below drops it)
        double c2 = 0;
        for (int a = 0; a < p; a++) c2 += C[(size_t)a + (size_t)j * p] * C[(size_t)a + (size_t)j * p];
        clean = c2 <= 1e-24 * left;
      }
      if (clean) break;
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
        while (keep < r && G[(size_t)keep + (size_t)keep * r] > defic * w0 && w0 > 0) keep++;
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
      int rk = chol_upper(r, G, R, pass == 0 ? defic : 1e-22);
      // the rows of the factor that couple EVERY column of the panel to the directions kept: a dependent column is
      // not a new direction, but it still has components along the kept ones, and the residual estimate of the step is
      // made of them.  (Dropping them — only the kept columns' couplings survived — let a 17 x 277 matrix with
      // 16 vectors per pass end "converged" after one step, 3.4 % off: the one direction R^17 had left was coupled to
      // all 16 columns, the estimate saw the first.)
      const int rc = r;
      std::vector<double> Rrows;
      if (rk < r) {
        Rrows.assign((size_t)rk * rc, 0.0);
        for (int j = 0; j < rc; j++)
          for (int i = 0; i < rk; i++) Rrows[(size_t)i + (size_t)j * rk] = R[(size_t)i + (size_t)j * rc];
        std::vector<double> G2((size_t)rk * rk);
        for (int j = 0; j < rk; j++)
          for (int i = 0; i < rk; i++) G2[(size_t)i + (size_t)j * rk] = G[(size_t)i + (size_t)j * r];
        r = rk;
        if (r == 0) return 0;
        chol_upper(r, G2, R, 0.0);
      }
      inv_upper(r, R, Ri);
      bk.W_times(r, r, Ri.data());
      // Rout <- R * Rout (leading r x cb part; R = the r x rc coupling rows when columns were dropped)
      std::vector<double> Rn((size_t)cb * cb, 0.0);
      for (int j = 0; j < cb; j++)
        for (int i = 0; i < r; i++) {
          double s = 0;
          if (Rrows.empty()) {
            for (int t = i; t < r; t++) s += R[(size_t)i + (size_t)t * r] * Rout[(size_t)t + (size_t)j * cb];
          } else {
            for (int t = i; t < rc; t++) s += Rrows[(size_t)i + (size_t)t * r] * Rout[(size_t)t + (size_t)j * cb];
          }
          Rn[(size_t)i + (size_t)j * cb] = s;
        }
      Rout.swap(Rn);
      if (pass == 0 && p > 0) {  // project again after the first normalisation (the columns have norm 1 now: until a pass removes < 1e-12)
        for (int again = 0; again < 6; again++) {
          C.assign((size_t)p * r, 0.0);
          bk.QtW(p, r, C.data());
          bk.W_minus_QC(p, r, C.data());
          double cmax = 0;
          for (size_t t = 0; t < C.size(); t++) cmax = std::max(cmax, std::fabs(C[t]));
          if (cmax <= 1e-12) break;
        }
      }
    }
    return r;
  };

  // precision schedule (SvdOptions): digits of the next block step from the leading half's residuals
  const bool sched = opt.slices_base > 0 && opt.slices_max > opt.slices_base && opt.vec_floor > 0;
  const int klead = (k + 1) / 2;
  double rho_lead = 1.0, z_gain = 1.0;   // leading half: largest relative residual; relative weight of the product pass's rounding
  auto step_slices = [&](const double gain) -> int {
    int S = opt.slices_base;
    if (sched) {
      const double rho = rho_lead < 1.0 ? rho_lead : 1.0;
      while (S < opt.slices_max && 1.2 * std::ldexp(1.0, -8 * S) * rho * gain > opt.vec_floor) S++;
    }
    return S;
  };
  if (opt.slices_base > 0) bk.set_precision(opt.slices_start > 0 ? opt.slices_start : opt.slices_base);
  // start block
  bk.random_W(b, opt.seed);
  int r = orth(0, b, C2, Rt);
  for (int w = 0; w < opt.warm && r > 0; w++) {
    if (!bk.subset(true)) break;
    bk.round_W(r);
    bk.W_to_Q(0, r);
    bk.At_Qblock(0, r);
    bk.A_Zblock(0, r);
    bk.subset(false);
    res.warm++;
    r = orth(0, r, C2, Rt);
  }
  bk.round_W(r);
  bk.W_to_Q(0, r);
  int p = r;      // basis size (columns of Q filled)
  int cb = r;     // size of the newest block
  int pp = 0;     // number of leading columns whose T columns are complete
  std::vector<double> Rlast;  // coupling block (r_next x cb) of the newest complete block
  int rr_rank = 0;            // independent directions the last Rayleigh-Ritz step worked with (== pp normally)
  int rl_rows = 0, rl_cols = 0;
  int exhausted_restarts = 0;   // thick restarts taken while the space was exhausted by count (p >= dim)

  while (cb > 0) {
    const int p0 = p - cb;
    bk.At_Qblock(p0, cb);
    int S_next = 0;
    if (opt.slices_base > 0) {
      const int Sz = step_slices(z_gain);   // digits of Z in the product pass
      S_next = step_slices(1.0);            // grid of the block it produces = digits of the next crossproduct pass
      bk.set_precision(Sz);
      if (S_next > opt.slices_base || Sz > opt.slices_base) res.wide_steps++;
      if (std::max(Sz, S_next) > res.slices_used_max) res.slices_used_max = std::max(Sz, S_next);
      if (opt.verbose)
        std::fprintf(stderr, "[bsn svd] step %d: %d-bit product pass, next block on %d bits (leading residual %.2e, product weight %.3f)\n",
                     res.niter + 1, 8 * Sz, 8 * S_next, rho_lead, z_gain);
    }
    bk.A_Zblock(p0, cb);
    // the grid of the block this step produces: narrow already by the previous step's residuals -> queued behind the
    // orthonormalisation as always; else decided below, once this step's residuals are known
    const bool hold = sched && S_next > opt.slices_base;
    if (opt.slices_base > 0) {
      bk.set_precision(S_next);
      bk.hold_rounding(hold);
    }
    res.nops += 2;
    res.niter++;
    // Gram blocks of the new columns p0 .. p-1 (Z of this step is complete now, Q was stored rounded)
    // and the orthonormalisation of W: in one go when the backend can, else piece by piece
    blk.assign((size_t)p * cb, 0.0);
    blk2.assign((size_t)p * cb, 0.0);
    int rn = bk.step_fused(p, p0, cb, blk.data(), blk2.data(), Rt);
    const bool fused_failed = rn == -2;
    if (rn == -1) {
      bk.ZtZ(p, p0, cb, blk.data());
      bk.QtQ(p, p0, cb, blk2.data());
    }
    // NaN / Inf in the projected blocks: in the operator's centre or scale (a scale of 0), in a start block handed in,
    // or an overflow (entries beyond 1e150 square out of range).  Nothing below is defined on them — say so here
    for (size_t t = 0; t < blk.size(); t++)
      if (!std::isfinite(blk[t]) || !std::isfinite(blk2[t]))
        throw std::runtime_error("the projected matrices of the partial SVD hold NaN or Inf: non-finite values in the matrix, "
                                 "its centre or scale (a scale of 0?), or entries whose squares overflow");
    if (res.niter == 1) {
      // ... and the scale of the matrix.  Z'Z of the random start block is of the order sigma^2; the next Gram matrix
      // W'W is of the order sigma^4 and must neither overflow nor fall into the denormals: a matrix scaled by 1e120
      // came back "converged" with d = 0, one scaled by 1e-100 with d wrong by a factor of two (both found by the
      // same sweep).  Genotypes over a standard deviation are within 1e-2 .. 1e2; anything beyond 1e+-70 is refused.
      double gmax = 0;
      for (int j = 0; j < cb; j++) gmax = std::max(gmax, std::fabs(blk[(size_t)(p0 + j) + (size_t)j * p]));
      if (gmax > 1e140 || (gmax > 0 && gmax < 1e-140))
        throw std::runtime_error("the matrix of the partial SVD is of the order of 1e" +
                                 std::to_string((int)std::lround(0.5 * std::log10(gmax))) +
                                 ": outside 1e-70 .. 1e70 its Gram matrices overflow or underflow in double precision; rescale it "
                                 "(center / scale of the operator) and scale d back");
    }
    for (int j = 0; j < cb; j++)
      for (int i = 0; i < p; i++) Gat(i, p0 + j) = Gat(p0 + j, i) = blk[(size_t)i + (size_t)j * p];
    for (int j = 0; j < cb; j++)
      for (int i = 0; i < p; i++) Mat(i, p0 + j) = Mat(p0 + j, i) = blk2[(size_t)i + (size_t)j * p];
    for (int j = 0; j < cb; j++)  // symmetrise the diagonal blocks
      for (int i = 0; i < j; i++) {
        double a = 0.5 * (Gat(p0 + i, p0 + j) + Gat(p0 + j, p0 + i));
        Gat(p0 + i, p0 + j) = Gat(p0 + j, p0 + i) = a;
        a = 0.5 * (Mat(p0 + i, p0 + j) + Mat(p0 + j, p0 + i));
        Mat(p0 + i, p0 + j) = Mat(p0 + j, p0 + i) = a;
      }
    // Rt: cb x cb (rn rows used): W_in = Q C2 + W_out Rt
    if (rn < 0) rn = orth(p, cb, C2, Rt, /*try_fused=*/!fused_failed);
    // The complement of span(Q) in R^n has n - p directions, whatever the deficiency tests make of a rounded panel: on
    // 16-bit products the projected block of a nearly full space is two genuine directions and six of rounding noise,
    // all above the absolute threshold.  Without this bound a matrix with fewer rows than the basis limit (34 x 1701,
    // k = 6, block 8) asked for a restart at p = 32 for ever — the noise directions did not fit below the cap —
    // and ended on their coupling block as its "residual" (found by tests/test_gpu_random_shapes.py, offset 3000).
    // Only against n: a basis of ALL of R^n gives exact Ritz pairs of A A' whatever its vectors are (the products are
    // exact for the stored basis).  The other bound, rank(A) + block < n, is a statement about exact arithmetic: there
    // the coupling block of the noise directions stays in, it is what tells an inexact exhaustion from an exact one.
    if ((int64_t)p + rn > bk.n) rn = bk.n > p ? (int)(bk.n - p) : 0;
    pp = p;
    const bool exhausted = (rn == 0) || (p >= dim);  // Krylov space is invariant: Ritz pairs exact
    // the coupling block of the FULL next block measures the residuals, whether or not the basis has
    // room left for it (a block clipped by the cap would make them look smaller, or zero, than they are)
    Rlast.assign((size_t)rn * cb, 0.0);
    for (int j = 0; j < cb; j++)
      for (int i = 0; i < rn; i++) Rlast[(size_t)i + (size_t)j * rn] = Rt[(size_t)i + (size_t)j * cb];
    rl_rows = rn;
    rl_cols = cb;
    const int rn_full = rn;
    // no room for the next block: compress the basis first (below, once the Ritz vectors are known)
    // (also when the space is exhausted BY COUNT, p >= dim, with a block left over: in floating point — single-vector
    // steps over a wide spectrum, or rounded products — such a space is only nearly invariant, the coupling block says
    // by how much, and a thick restart continues from the Ritz vectors with what the old basis lost: 316 x 23, k = 18,
    // block 1 on exact products ended at its 24th vector with residuals of 1.5e-2, and converges after one restart)
    // (at most two restarts from the exhausted-by-count state: what floating point lost of an invariant space comes back
    // with one or two; a matrix whose ROUNDED products keep handing over noise directions there would otherwise use up
    // all max_restarts before it ends with exhausted = 1 — which is what sends the caller to the 56-bit solve anyway)
    const bool by_count = p >= dim;
    const bool want_restart = rn > cap - p && rn > 0 && pp >= k && res.restarts < opt.max_restarts &&
                              (!by_count || exhausted_restarts < 2);
    if (rn > cap - p) rn = cap - p;
    if (rn < 0) rn = 0;

    // Rayleigh-Ritz on span(Q[:, :pp]): (Gz) s = theta (Mq) s with Mq = R'R,
    // i.e. the standard problem for R^-T Gz R^-1, s = R^-1 y.  Not needed while the basis is smaller
    // than k and the iteration goes on (the device idles while the host works here).
    if (pp >= (sched ? std::min(k, klead) : k) || rn == 0 || exhausted || want_restart) {
      std::vector<double> Mp((size_t)pp * pp), Gp((size_t)pp * pp), Rm, Rmi, tmp((size_t)pp * pp);
      for (int j = 0; j < pp; j++)
        for (int i = 0; i < pp; i++) {
          Mp[(size_t)i + (size_t)j * pp] = Mat(i, j);
          Gp[(size_t)i + (size_t)j * pp] = Gat(i, j);
        }
      int rk = chol_upper(pp, Mp, Rm, 1e-14);
      if (opt.verbose > 1) {
        double dev = 0;
        for (int j = 0; j < pp; j++)
          for (int i = 0; i < pp; i++) dev = std::max(dev, std::fabs(Mat(i, j) - (i == j ? 1.0 : 0.0)));
        std::fprintf(stderr, "[bsn svd]   Q'Q: max |M - I| = %.3e, Cholesky rank %d of %d\n", dev, rk, pp);
      }
      rr_rank = pp;
      if (rk == pp) {
        inv_upper(pp, Rm, Rmi);
        // tmp = Gp * Rmi ; evec = Rmi' * tmp
        small_mm(pp, pp, pp, Gp.data(), Rmi.data(), tmp.data());
        evec.assign((size_t)pp * pp, 0.0);
        for (int j = 0; j < pp; j++)
          for (int i = 0; i <= j; i++) {
            double a = 0;
            for (int t = 0; t <= i; t++) a += Rmi[(size_t)t + (size_t)i * pp] * tmp[(size_t)t + (size_t)j * pp];
            evec[(size_t)i + (size_t)j * pp] = a;
            evec[(size_t)j + (size_t)i * pp] = a;
          }
        eig_sym(pp, evec, eval);
        // back-transform: s = Rmi * y (upper triangular)
        small_mm(pp, pp, pp, Rmi.data(), evec.data(), tmp.data());
        evec.swap(tmp);
      } else {
        // The stored basis is numerically DEPENDENT (Q'Q singular to 1e-14).  It happens when the Krylov space
        // is exhausted but rounded products keep handing over "new" directions that are amplified noise — a
        // request for more singular values than the matrix has rank, k > rank(A).  Canonical orthogonalisation:
        // Q'Q = V L V', keep the r directions with L > 1e-8 max L, T = V L^(-1/2) (pp x r); the Ritz pairs of
        // span(Q) are those of T' (Z'Z) T, s = T y; the pp - r dropped directions get Ritz value 0.
        std::vector<double> V(Mp), lam;
        eig_sym(pp, V, lam);
        const double lmax = lam[pp - 1];
        int r = 0;
        while (r < pp && lam[pp - 1 - r] > 1e-8 * lmax && lmax > 0) r++;
        std::vector<double> T((size_t)pp * std::max(r, 1), 0.0), GT((size_t)pp * std::max(r, 1)), B((size_t)r * r), bl;
        for (int c = 0; c < r; c++) {
          const double f = 1.0 / std::sqrt(lam[pp - 1 - c]);
          for (int i = 0; i < pp; i++) T[(size_t)i + (size_t)c * pp] = V[(size_t)i + (size_t)(pp - 1 - c) * pp] * f;
        }
        small_mm(pp, pp, r, Gp.data(), T.data(), GT.data());
        for (int j = 0; j < r; j++)
          for (int i = 0; i <= j; i++) {
            double a = 0;
            for (int t = 0; t < pp; t++) a += T[(size_t)t + (size_t)i * pp] * GT[(size_t)t + (size_t)j * pp];
            B[(size_t)i + (size_t)j * r] = B[(size_t)j + (size_t)i * r] = a;
          }
        eig_sym(r, B, bl);
        rr_rank = r;
        eval.assign((size_t)pp, 0.0);
        evec.assign((size_t)pp * pp, 0.0);
        for (int c = 0; c < r; c++) {   // ascending order, the r computed pairs at the top end
          const int col = pp - r + c;
          eval[(size_t)col] = bl[(size_t)c];
          for (int i = 0; i < pp; i++) {
            double a = 0;
            for (int t = 0; t < r; t++) a += T[(size_t)i + (size_t)t * pp] * B[(size_t)t + (size_t)c * r];
            evec[(size_t)i + (size_t)col * pp] = a;
          }
        }
        if (opt.verbose)
          std::fprintf(stderr, "[bsn svd]   dependent basis: %d of %d directions kept for the Rayleigh-Ritz step\n", r, pp);
      }
    }
    bool done = false;
    const double hard_zero = 1e-10;
    const double negligible = std::max(hard_zero, 64.0 * opt.resid_floor * opt.resid_floor);
    double worst_sig = 0;   // the same over the triplets that are not numerically zero (theta > negligible theta_max)
    if (pp >= k) {
      double worst = 0, lead = 0;
      int n_soft = 0;
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
        if (t < klead) lead = std::max(lead, rel);
        if (theta > negligible * eval[pp - 1]) worst_sig = std::max(worst_sig, rel);
        else if (theta > hard_zero * eval[pp - 1]) n_soft++;
      }
      res.below_resolution = n_soft;
      // Triplets whose Ritz value is numerically zero — theta below (8 resid_floor)^2 of the largest, at least 1e-10:
      // sigma below 1e-5 (24-bit products and wider) .. 1.5e-4 (16-bit) of sigma_1 — are what k > rank(A) asks for.  Their vectors are any null vectors and
      // their relative residuals mean nothing; on rounded products the directions behind them are what the rounding
      // of the stored basis left of range(A), and they never stop coming.  The convergence test is over the others:
      // a matrix of rank 4 asked for 5 - 7 triplets on 16-bit products went through 100 restarts (4 000 block steps)
      // and came back with invented singular values (98 and 296 beside the true 207 .. 151; once under "converged").
      // What the exemption must NOT do is vouch for a triplet that is merely small: sigma = 1e-4 sigma_1 is a legitimate
      // request that 16-bit products cannot serve (300 x 400, sigma = 100 .. 40 and a 1e-2 tail, k = 6 came
      // back "converged" with the tail 10 % off).  Triplets between the hard zero and the products' resolution are
      // counted (below_resolution) and the caller decides: wider products, or "not converged".
      worst = worst_sig;
      res.max_rel_resid = worst;
      res.lead_rel_resid = lead;
      rho_lead = lead;
      if (opt.noise_gain > 0) {
        const double th1 = eval[pp - 1], thh = eval[pp - klead];
        z_gain = thh > 0 ? std::min(1.0, opt.noise_gain * std::sqrt(std::max(th1, 0.0)) / thh) : 1.0;
      }
      if (opt.verbose)
        std::fprintf(stderr, "[bsn svd] step %d basis %d max rel resid %.3e sigma1 %.6g\n", res.niter,
                     pp, worst, std::sqrt(std::max(eval[pp - 1], 0.0)));
      if (worst + opt.resid_floor <= opt.tol) {
        done = true;
        res.converged = 1;
      }
    }
    if (sched && pp < k && pp >= klead && !eval.empty()) {
      // the basis does not hold k vectors yet, but it holds the leading half: their residuals steer the schedule
      double lead = 0;
      for (int t = 0; t < klead; t++) {
        const int col = pp - 1 - t;
        double rs = 0;
        for (int i = 0; i < rl_rows; i++) {
          double s = 0;
          for (int j = 0; j < rl_cols; j++)
            s += Rlast[(size_t)i + (size_t)j * rl_rows] * evec[(size_t)(pp - rl_cols + j) + (size_t)col * pp];
          rs += s * s;
        }
        lead = std::max(lead, std::sqrt(rs) / std::max(std::fabs(eval[col]), 1e-300));
      }
      rho_lead = lead;
      if (opt.noise_gain > 0) {
        const double th1 = eval[pp - 1], thh = eval[pp - klead];
        z_gain = thh > 0 ? std::min(1.0, opt.noise_gain * std::sqrt(std::max(th1, 0.0)) / thh) : 1.0;
      }
      if (opt.verbose) std::fprintf(stderr, "[bsn svd] step %d basis %d leading-half rel resid %.3e\n", res.niter, pp, lead);
    }
    if (hold) {   // (rho_lead is this step's now, unless the basis is still smaller than the leading half)
      const int S = step_slices(1.0);
      bk.set_precision(S);
      if (S > res.slices_used_max) res.slices_used_max = S;
      if (opt.verbose) std::fprintf(stderr, "[bsn svd] step %d: next block on %d bits (leading residual %.2e)\n", res.niter, 8 * S, rho_lead);
    }
    if (!done && want_restart) {
      // keep the k + b largest Ritz pairs (room for at least the next block must remain)
      int keep = std::min(std::min(pp, rr_rank), k + b);
      if (keep > cap - rn_full) keep = cap - rn_full;
      if (keep >= k) {
        std::vector<double> Sk((size_t)pp * keep);
        for (int t = 0; t < keep; t++)
          for (int i = 0; i < pp; i++) Sk[(size_t)i + (size_t)t * pp] = evec[(size_t)i + (size_t)(pp - 1 - t) * pp];
        // Gram matrices of the kept vectors from the exact Gram matrices of the old basis: S'(Z'Z)S and S'(Q'Q)S.
        // In exact arithmetic they are diag(theta) and I; with an ill-conditioned old basis (Krylov exhaustion)
        // the computed S is only (Q'Q)-orthonormal to eps * cond, and assuming I would make Z'Z and Q'Q
        // inconsistent — Ritz values above the largest singular value were the symptom.
        std::vector<double> Gk((size_t)keep * keep), Mk((size_t)keep * keep), tG((size_t)pp * keep), tM((size_t)pp * keep);
        {
          std::vector<double> Mp((size_t)pp * pp), Gp((size_t)pp * pp);
          for (int j = 0; j < pp; j++)
            for (int i = 0; i < pp; i++) {
              Mp[(size_t)i + (size_t)j * pp] = Mat(i, j);
              Gp[(size_t)i + (size_t)j * pp] = Gat(i, j);
            }
          small_mm(pp, pp, keep, Gp.data(), Sk.data(), tG.data());
          small_mm(pp, pp, keep, Mp.data(), Sk.data(), tM.data());
          for (int j = 0; j < keep; j++)
            for (int i = 0; i <= j; i++) {
              double g = 0, mm = 0;
              for (int t = 0; t < pp; t++) {
                g += Sk[(size_t)t + (size_t)i * pp] * tG[(size_t)t + (size_t)j * pp];
                mm += Sk[(size_t)t + (size_t)i * pp] * tM[(size_t)t + (size_t)j * pp];
              }
              Gk[(size_t)i + (size_t)j * keep] = Gk[(size_t)j + (size_t)i * keep] = g;
              Mk[(size_t)i + (size_t)j * keep] = Mk[(size_t)j + (size_t)i * keep] = mm;
            }
        }
        if (bk.restart(pp, keep, Sk.data(), rn_full, Mk.data())) {
          for (int j = 0; j < cap; j++)
            for (int i = 0; i < cap; i++) Gat(i, j) = Mat(i, j) = 0.0;
          for (int j = 0; j < keep; j++)
            for (int i = 0; i < keep; i++) {
              Gat(i, j) = Gk[(size_t)i + (size_t)j * keep];
              Mat(i, j) = Mk[(size_t)i + (size_t)j * keep];
            }
          res.restarts++;
          if (by_count) exhausted_restarts++;
          if (opt.verbose)
            std::fprintf(stderr, "[bsn svd] basis full at %d: restart with %d Ritz vectors\n", pp, keep);
          p = keep;
          pp = keep;
          rn = rn_full;
          bk.round_W(rn);
          bk.W_to_Q(p, rn);
          p += rn;
          cb = rn;
          continue;
        }
      }
    }
    if (done || rn == 0) {
      if (exhausted) {
        res.exhausted = 1;
        res.exhausted_resid = worst_sig;
      }
      if (exhausted && !done) {
        // An invariant Krylov space gives exact Ritz pairs — in exact arithmetic.  The products here are exact for a
        // ROUNDED basis: the last directions of a nearly exhausted space are what is left of a panel after projecting
        // out almost all of it, i.e. rounding noise amplified by that cancellation (39 x 17, k = 16, block 8, 16-bit
        // digits: the 25th of 25 directions is 98.5 % right, the singular values 1.5e-3 off), and the coupling block of
        // the step says so.  Triplets beyond the rank (theta ~ 0) are exempt: any null vector serves.  The caller
        // (svd.hip) answers an inexact exhaustion with a second solve on 56-bit products.
        res.converged = (worst_sig + opt.resid_floor <= opt.tol) ? 1 : 0;
      }
      break;
    }
    bk.round_W(rn);
    bk.W_to_Q(p, rn);
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
