// Test-only shared library: runs the product's backend-independent host code
// (bigsnpr_amd/csrc/svd_driver.hpp, dense_small.hpp) on CPU with a dense host backend so
// that the driver logic — including the column-sharded multi-rank path with an all-reduce
// hook — can be tested without a GPU (gloo, world_size 2).  Not part of the product.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "orth_small.hpp"
#include "svd_driver.hpp"

using namespace bsn;

typedef void (*allreduce_fn)(double *buf, int64_t count, void *ctx);

struct HostBackend : SvdBackend {
  const double *A = nullptr;  // n x m_local dense scaled matrix, column-major
  allreduce_fn ar = nullptr;
  void *ctx = nullptr;
  std::vector<double> Q, Z, W;
  void alloc(int cap, int b) override {
    Q.assign((size_t)n * cap, 0.0);
    Z.assign((size_t)m_local * cap, 0.0);
    W.assign((size_t)n * b, 0.0);
  }
  void random_W(int b, uint32_t seed) override {
    uint64_t s = 0x9E3779B97F4A7C15ull * (seed + 1);
    for (size_t t = 0; t < (size_t)n * b; t++) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      W[t] = (double)(s >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
    }
  }
  void At_Qblock(int p0, int cb) override {
    std::vector<double> q(Q.begin() + (size_t)p0 * n, Q.begin() + (size_t)(p0 + cb) * n);
    round_cols(q.data(), n, cb, 0.99);  // what the product's own quantiser does (exact for a rounded block)
    for (int c = 0; c < cb; c++)
      for (int64_t j = 0; j < m_local; j++) {
        double s = 0;
        for (int64_t i = 0; i < n; i++) s += A[i + j * n] * q[i + (size_t)c * n];
        Z[j + (size_t)(p0 + c) * m_local] = s;
      }
  }
  void A_Zblock(int p0, int cb) override {
    std::vector<double> z(Z.begin() + (size_t)p0 * m_local, Z.begin() + (size_t)(p0 + cb) * m_local);
    round_cols(z.data(), m_local, cb, 0.99);
    std::fill(W.begin(), W.begin() + (size_t)n * cb, 0.0);
    for (int c = 0; c < cb; c++)
      for (int64_t j = 0; j < m_local; j++) {
        double zz = z[j + (size_t)c * m_local];
        for (int64_t i = 0; i < n; i++) W[i + (size_t)c * n] += A[i + j * n] * zz;
      }
    if (ar) ar(W.data(), n * cb, ctx);
  }
  // slices > 0 emulates the product backend: vectors entering a product are rounded to a fixed-point
  // grid of 8 * slices bits (per-vector power-of-two scale); 0 = exact fp64 products
  int slices = 0;
  void round_cols(double *X, int64_t rows, int cols, double headroom) const {
    if (slices <= 0) return;
    for (int c = 0; c < cols; c++) {
      double mx = 0;
      for (int64_t i = 0; i < rows; i++) mx = std::fmax(mx, std::fabs(X[i + (size_t)c * rows]));
      if (!(mx > 0)) continue;
      int e;
      std::frexp(std::ldexp(headroom, 8 * slices - 1) / mx, &e);
      const double qs = std::ldexp(1.0, e - 1);
      for (int64_t i = 0; i < rows; i++) X[i + (size_t)c * rows] = std::nearbyint(X[i + (size_t)c * rows] * qs) / qs;
    }
  }
  void round_W(int cb) override { round_cols(W.data(), n, cb, 0.98); }
  // the driver's precision schedule: what the product backend does with it (digits of the next product pass, of the
  // rounding of the block it produces and of the crossproduct pass that reads that block)
  std::vector<int> sched_log;
  void set_precision(int S) override {
    if (slices > 0) slices = S;
    sched_log.push_back(S);
  }
  void ZtZ(int p, int p0, int cb, double *G) override {
    for (int c = 0; c < cb; c++)
      for (int a = 0; a < p; a++) {
        double s = 0;
        for (int64_t j = 0; j < m_local; j++) s += Z[j + (size_t)a * m_local] * Z[j + (size_t)(p0 + c) * m_local];
        G[a + (size_t)c * p] = s;
      }
    if (ar) ar(G, (int64_t)p * cb, ctx);
  }
  void QtQ(int p, int p0, int cb, double *M) override {
    for (int c = 0; c < cb; c++)
      for (int a = 0; a < p; a++) {
        double s = 0;
        for (int64_t i = 0; i < n; i++) s += Q[i + (size_t)a * n] * Q[i + (size_t)(p0 + c) * n];
        M[a + (size_t)c * p] = s;
      }
  }
  void QtW(int p, int cb, double *C) override {
    for (int c = 0; c < cb; c++)
      for (int a = 0; a < p; a++) {
        double s = 0;
        for (int64_t i = 0; i < n; i++) s += Q[i + (size_t)a * n] * W[i + (size_t)c * n];
        C[a + (size_t)c * p] = s;
      }
  }
  void W_minus_QC(int p, int cb, const double *C) override {
    for (int c = 0; c < cb; c++)
      for (int a = 0; a < p; a++) {
        double f = C[a + (size_t)c * p];
        for (int64_t i = 0; i < n; i++) W[i + (size_t)c * n] -= Q[i + (size_t)a * n] * f;
      }
  }
  void WtW(int cb, double *G) override {
    for (int c = 0; c < cb; c++)
      for (int a = 0; a < cb; a++) {
        double s = 0;
        for (int64_t i = 0; i < n; i++) s += W[i + (size_t)a * n] * W[i + (size_t)c * n];
        G[a + (size_t)c * cb] = s;
      }
  }
  void W_times(int cb, int r, const double *M) override {
    std::vector<double> row(cb), out(r);
    for (int64_t i = 0; i < n; i++) {
      for (int j = 0; j < cb; j++) row[j] = W[i + (size_t)j * n];
      for (int c = 0; c < r; c++) {
        double s = 0;
        for (int j = 0; j < cb; j++) s += row[j] * M[j + (size_t)c * cb];
        out[c] = s;
      }
      for (int c = 0; c < r; c++) W[i + (size_t)c * n] = out[c];
    }
  }
  void W_to_Q(int p0, int r) override {
    std::memcpy(&Q[(size_t)p0 * n], W.data(), sizeof(double) * (size_t)n * r);
  }
  // ---- the fused block step of the product's HIP backend, with host loops for the tall products and the
  // SAME small-matrix code (orth_small.hpp) the device runs in one workgroup ----
  int fused_mode = 0;   // 0 = step-by-step path only; 1 = two-pass fused path (nt_set_fused)
  int n_fused = 0, n_careful = 0;
  std::vector<double> Mdev;   // the "device" copy of Q'Q, leading dimension kOrthMaxP
  struct HostCtx {
    int tid = 0, nt = 1;
    void sync() {}
  };
  // HG ((p + cb) x cb) = [Q[:, :p] W]' W, summed over the ranks' rows (here: all rows are local)
  void hg(int p, int cb, std::vector<double> &HG) {
    HG.assign((size_t)(p + cb) * cb, 0.0);
    for (int c = 0; c < cb; c++)
      for (int a = 0; a < p + cb; a++) {
        const double *col = a < p ? &Q[(size_t)a * n] : &W[(size_t)(a - p) * n];
        double s = 0;
        for (int64_t i = 0; i < n; i++) s += col[i] * W[i + (size_t)c * n];
        HG[a + (size_t)c * (p + cb)] = s;
      }
  }
  void update(int p, int cb, const double *C, const double *Ri) {
    std::vector<double> row(cb), out(cb);
    for (int64_t i = 0; i < n; i++) {
      for (int j = 0; j < cb; j++) {
        double acc = 0;
        for (int a = 0; a < p; a++) acc += Q[i + (size_t)a * n] * C[a + (size_t)j * p];
        row[j] = W[i + (size_t)j * n] - acc;
      }
      for (int c = 0; c < cb; c++) {
        double s = 0;
        for (int j = 0; j < cb; j++) s += row[j] * Ri[j + c * cb];
        out[c] = s;
      }
      for (int c = 0; c < cb; c++) W[i + (size_t)c * n] = out[c];
    }
  }
  int fused(int p, int p0, int cb, double *blkZ, double *blkQ, std::vector<double> &Rout) {
    if (!fused_mode || cb <= 0 || cb > kOrthMaxB || p + cb > kOrthMaxP) return -1;
    if (p > 0) {
      ZtZ(p, p0, cb, blkZ);
      QtQ(p, p0, cb, blkQ);
    }
    if (Mdev.empty()) Mdev.assign((size_t)kOrthMaxP * kOrthMaxP, 0.0);
    std::vector<double> Wsave(W.begin(), W.begin() + (size_t)n * cb), HG, C((size_t)p * cb + 1), Ct((size_t)p * cb + 1),
        Ri((size_t)cb * cb), Ro((size_t)cb * cb, 0.0), Cs((size_t)p * cb + 1), Gs((size_t)cb * cb), Rs((size_t)cb * cb),
        Ris((size_t)cb * cb), Rob((size_t)cb * cb), Dv((size_t)cb * cb), tmp((size_t)2 * cb + 4);
    double flag = 0.0;
    OrthSmall a;
    a.p = p; a.cb = cb; a.p0 = p0; a.QtQ = p > 0 ? blkQ : nullptr; a.ldq = p; a.M = Mdev.data(); a.ldm = kOrthMaxP;
    a.C = C.data(); a.Ct = Ct.data(); a.Ri = Ri.data(); a.Rout = Ro.data(); a.flag = &flag; a.iters = 3;
    a.Cs = Cs.data(); a.Gs = Gs.data(); a.Rs = Rs.data(); a.Ris = Ris.data(); a.Ro = Rob.data(); a.Dv = Dv.data(); a.tmp = tmp.data();
    HostCtx cx;
    for (int pass = 0; pass < 2; pass++) {
      hg(p, cb, HG);
      // (the column-sharded host test holds all n rows of Q and W on every rank: nothing to sum here)
      a.pass = pass;
      a.HG = HG.data();
      orth_small(cx, a);
      update(p, cb, C.data(), Ri.data());
    }
    if (flag != 0.0) {
      std::copy(Wsave.begin(), Wsave.end(), W.begin());
      n_careful++;
      return -2;
    }
    Rout = Ro;
    n_fused++;
    return cb;
  }
  int step_fused(int p, int p0, int cb, double *blkZ, double *blkQ, std::vector<double> &Rout) override {
    return fused(p, p0, cb, blkZ, blkQ, Rout);
  }
  int orth_fused(int p, int cb, std::vector<double> &Cacc, std::vector<double> &Rout) override {
    if (p != 0) return -1;
    Cacc.clear();
    const int r = fused(0, 0, cb, nullptr, nullptr, Rout);
    return r < 0 ? -1 : r;
  }
  bool restart(int pp, int keep, const double *S, int rn, const double *Mk) override {
    (void)rn;   // W is a buffer of its own here
    auto combine = [&](std::vector<double> &X, int64_t rows) {
      std::vector<double> out((size_t)rows * keep, 0.0);
      for (int t = 0; t < keep; t++)
        for (int a = 0; a < pp; a++) {
          const double f = S[a + (size_t)t * pp];
          for (int64_t i = 0; i < rows; i++) out[i + (size_t)t * rows] += X[i + (size_t)a * rows] * f;
        }
      std::copy(out.begin(), out.end(), X.begin());
    };
    combine(Q, n);
    combine(Z, m_local);
    if (!Mdev.empty()) {
      for (int j = 0; j < keep; j++)
        for (int i = 0; i < keep; i++) Mdev[(size_t)i + (size_t)j * kOrthMaxP] = Mk[(size_t)i + (size_t)j * keep];
    }
    return true;
  }
  void finalize(int pp, int k, const double *S, const double *dinv, double *u, double *v) override {
    for (int t = 0; t < k; t++) {
      for (int64_t i = 0; i < n; i++) {
        double s = 0;
        for (int a = 0; a < pp; a++) s += Q[i + (size_t)a * n] * S[a + (size_t)t * pp];
        u[i + (size_t)t * n] = s;
      }
      for (int64_t j = 0; j < m_local; j++) {
        double s = 0;
        for (int a = 0; a < pp; a++) s += Z[j + (size_t)a * m_local] * S[a + (size_t)t * pp];
        v[j + (size_t)t * m_local] = s * dinv[t];
      }
    }
  }
};

static int g_slices = 0, g_fused = 0, g_max_restarts = 100;
static int g_slices_max = 0, g_slices_start = 0, g_sched_log[64], g_sched_n = 0;
static double g_vec_floor = 0.0, g_noise_gain = 0.0;
static int g_counts[2] = {0, 0};

extern "C" {

// emulate the product's fixed-point products in the host backend (0 = exact)
void nt_set_slices(int slices) { g_slices = slices; }
// precision schedule of the driver (SvdOptions::vec_floor / slices_max / slices_start); 0, 0, 0 = none
void nt_set_schedule(double vec_floor, int slices_max, int slices_start) {
  g_vec_floor = vec_floor;
  g_slices_max = slices_max;
  g_slices_start = slices_start;
}
// SvdOptions::noise_gain: amplification of a random vector by A (sqrt(|A|_F^2 / m)); 0 = unknown
void nt_set_noise_gain(double g) { g_noise_gain = g; }
// digits handed to set_precision by the last solve, in call order (first = start block); returns the count
int nt_schedule_log(int *out, int cap) {
  for (int i = 0; i < g_sched_n && i < cap; i++) out[i] = g_sched_log[i];
  return g_sched_n;
}
// 1: the host backend takes the product's fused two-pass block step (orth_small.hpp)
void nt_set_fused(int on) { g_fused = on; }
// thick restarts of a full basis (negative: none: a full basis ends the solve unconverged)
void nt_set_max_restarts(int r) { g_max_restarts = r; }
// fused steps taken / steps handed to the careful path by the last solve
void nt_fused_counts(int *out) { out[0] = g_counts[0]; out[1] = g_counts[1]; }

void nt_eig_sym(int n, double *A, double *d) {
  std::vector<double> V(A, A + (size_t)n * n), w;
  eig_sym(n, V, w);
  std::memcpy(A, V.data(), sizeof(double) * (size_t)n * n);
  std::memcpy(d, w.data(), sizeof(double) * (size_t)n);
}

int nt_chol_inv(int b, const double *G, double *R, double *Ri) {
  std::vector<double> g(G, G + (size_t)b * b), r, ri;
  int rk = chol_upper(b, g, r, 1e-22);
  if (rk == b) {
    inv_upper(b, r, ri);
    std::memcpy(R, r.data(), sizeof(double) * (size_t)b * b);
    std::memcpy(Ri, ri.data(), sizeof(double) * (size_t)b * b);
  }
  return rk;
}

// info: niter, nops, basis, converged
void nt_svd_host(const double *A, int64_t n, int64_t m_local, int64_t m_total, int k, double tol,
                 int block, int max_basis, uint32_t seed, allreduce_fn ar, void *ctx, double *d,
                 double *u, double *v, int32_t *info, double *resid) {
  HostBackend bk;
  bk.A = A;
  bk.n = n;
  bk.m_local = m_local;
  bk.m_total = m_total;
  bk.ar = ar;
  bk.ctx = ctx;
  bk.slices = g_slices;
  bk.fused_mode = g_fused;
  SvdOptions o;
  o.k = k;
  o.tol = tol;
  o.block = block;
  o.max_basis = max_basis;
  o.seed = seed;
  o.max_restarts = g_max_restarts;
  if (const char *e = getenv("NT_VERBOSE")) o.verbose = atoi(e);   // (the driver's per-step log on stderr)
  o.resid_floor = g_slices > 0 ? 1.2 * std::ldexp(1.0, -8 * g_slices) : 0.0;
  if (g_slices > 0) {
    o.slices_base = g_slices;
    o.slices_max = g_slices_max > g_slices ? g_slices_max : g_slices;
    o.slices_start = g_slices_start;
    o.vec_floor = g_vec_floor;
    o.noise_gain = g_noise_gain;
  }
  SvdResult r;
  try {
    r = block_lanczos_svd(bk, o, d, u, v);
  } catch (const std::exception &) {   // (the HIP library turns it into an error code and bsn_last_error())
    for (int t = 0; t < 8; t++) info[t] = 0;
    info[5] = 1;   // the driver refused the input
    *resid = std::nan("");
    return;
  }
  g_sched_n = 0;
  for (int S : bk.sched_log)
    if (g_sched_n < 64) g_sched_log[g_sched_n++] = S;
  g_counts[0] = bk.n_fused;
  g_counts[1] = bk.n_careful;
  info[0] = r.niter;
  info[1] = r.nops;
  info[2] = r.basis;
  info[3] = r.converged;
  info[4] = r.restarts;
  info[6] = r.exhausted;
  // (what makes the HIP wrapper solve again on 56-bit products: bit 0 an inexact exhaustion, bit 1 requested triplets
  // below what the rounded products resolve)
  info[7] = (r.exhausted && r.exhausted_resid > 1e-9 ? 1 : 0) | (r.below_resolution > 0 ? 2 : 0);
  *resid = r.max_rel_resid;
}
}
