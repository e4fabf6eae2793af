// dense_small.hpp — small dense host-side linear algebra used by the SVD driver on
// matrices of order <= a few hundred (the projected block-tridiagonal matrix and b x b
// Gram matrices).  Column-major, fp64, no external BLAS/LAPACK.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

namespace bsn {

// Symmetric eigen-decomposition: Householder tridiagonalisation with accumulated
// transforms followed by implicit-shift QL.  On entry V holds the symmetric matrix
// (n x n, column-major, both triangles); on exit its columns are the eigenvectors and
// d the eigenvalues in ascending order.
inline void eig_sym(int n, std::vector<double> &V, std::vector<double> &d) {
  d.assign((size_t)n, 0.0);
  std::vector<double> e((size_t)n, 0.0);
  auto A = [&](int i, int j) -> double & { return V[(size_t)i + (size_t)j * n]; };
  if (n == 0) return;
  // A NaN or an infinity makes every convergence test below false: the deflation search then ran past the end of d
  // (heap overflow; found by a sweep over matrices scaled by 1e120, whose Gram matrices overflow).  Such a matrix has
  // no decomposition: the eigenvalues come back NaN and the caller says so.
  for (size_t t = 0; t < (size_t)n * (size_t)n; t++)
    if (!std::isfinite(V[t])) {
      d.assign((size_t)n, std::nan(""));
      return;
    }
  // --- tridiagonalise (rows processed from the bottom) --------------------------
  for (int j = 0; j < n; j++) d[j] = A(n - 1, j);
  for (int i = n - 1; i > 0; i--) {
    double scale = 0.0, h = 0.0;
    for (int k = 0; k < i; k++) scale += std::fabs(d[k]);
    if (scale == 0.0) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; j++) {
        d[j] = A(i - 1, j);
        A(i, j) = 0.0;
        A(j, i) = 0.0;
      }
    } else {
      for (int k = 0; k < i; k++) {
        d[k] /= scale;
        h += d[k] * d[k];
      }
      double f = d[i - 1];
      double g = std::sqrt(h);
      if (f > 0) g = -g;
      e[i] = scale * g;
      h -= f * g;
      d[i - 1] = f - g;
      for (int j = 0; j < i; j++) e[j] = 0.0;
      for (int j = 0; j < i; j++) {
        f = d[j];
        A(j, i) = f;
        g = e[j] + A(j, j) * f;
        for (int k = j + 1; k <= i - 1; k++) {
          g += A(k, j) * d[k];
          e[k] += A(k, j) * f;
        }
        e[j] = g;
      }
      f = 0.0;
      for (int j = 0; j < i; j++) {
        e[j] /= h;
        f += e[j] * d[j];
      }
      double hh = f / (h + h);
      for (int j = 0; j < i; j++) e[j] -= hh * d[j];
      for (int j = 0; j < i; j++) {
        f = d[j];
        g = e[j];
        for (int k = j; k <= i - 1; k++) A(k, j) -= (f * e[k] + g * d[k]);
        d[j] = A(i - 1, j);
        A(i, j) = 0.0;
      }
    }
    d[i] = h;
  }
  // --- accumulate transformations ---------------------------------------------
  for (int i = 0; i < n - 1; i++) {
    A(n - 1, i) = A(i, i);
    A(i, i) = 1.0;
    double h = d[i + 1];
    if (h != 0.0) {
      for (int k = 0; k <= i; k++) d[k] = A(k, i + 1) / h;
      for (int j = 0; j <= i; j++) {
        double g = 0.0;
        for (int k = 0; k <= i; k++) g += A(k, i + 1) * A(k, j);
        for (int k = 0; k <= i; k++) A(k, j) -= g * d[k];
      }
    }
    for (int k = 0; k <= i; k++) A(k, i + 1) = 0.0;
  }
  for (int j = 0; j < n; j++) {
    d[j] = A(n - 1, j);
    A(n - 1, j) = 0.0;
  }
  A(n - 1, n - 1) = 1.0;
  e[0] = 0.0;
  // --- implicit QL ---------------------------------------------------------------
  for (int i = 1; i < n; i++) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  const double eps = std::ldexp(1.0, -52);
  for (int l = 0; l < n; l++) {
    tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
    int m = l;
    while (m < n - 1) {   // (e[n - 1] == 0 ends the search there at the latest; the bound holds whatever the values are)
      if (std::fabs(e[m]) <= eps * tst1) break;
      m++;
    }
    if (m > l) {
      int iter = 0;
      do {
        iter++;
        double g = d[l];
        double p = (d[l + 1] - g) / (2.0 * e[l]);
        double r = std::hypot(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; i++) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1.0, c2 = c, c3 = c;
        double el1 = e[l + 1];
        double s = 0.0, s2 = 0.0;
        for (int i = m - 1; i >= l; i--) {
          c3 = c2;
          c2 = c;
          s2 = s;
          g = c * e[i];
          h = c * p;
          r = std::hypot(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          for (int k = 0; k < n; k++) {
            h = A(k, i + 1);
            A(k, i + 1) = s * A(k, i) + c * h;
            A(k, i) = c * A(k, i) - s * h;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (std::fabs(e[l]) > eps * tst1 && iter < 200);
    }
    d[l] += f;
    e[l] = 0.0;
  }
  // --- sort ascending ---------------------------------------------------------------
  for (int i = 0; i < n - 1; i++) {
    int k = i;
    double p = d[i];
    for (int j = i + 1; j < n; j++)
      if (d[j] < p) {
        k = j;
        p = d[j];
      }
    if (k != i) {
      d[k] = d[i];
      d[i] = p;
      for (int j = 0; j < n; j++) std::swap(A(j, i), A(j, k));
    }
  }
}

// Upper Cholesky factor of a symmetric positive definite b x b matrix: G = R' R.
// Returns the number of leading columns that factorised with a pivot above
// rel_tol * max diagonal (== b when G is numerically full rank).
inline int chol_upper(int b, const std::vector<double> &G, std::vector<double> &R, double rel_tol) {
  R.assign((size_t)b * b, 0.0);
  double dmax = 0;
  for (int i = 0; i < b; i++) dmax = std::max(dmax, G[(size_t)i + (size_t)i * b]);
  for (int j = 0; j < b; j++) {
    double s = G[(size_t)j + (size_t)j * b];
    for (int k = 0; k < j; k++) s -= R[(size_t)k + (size_t)j * b] * R[(size_t)k + (size_t)j * b];
    if (!(s > rel_tol * dmax) || !(dmax > 0)) return j;
    double rjj = std::sqrt(s);
    R[(size_t)j + (size_t)j * b] = rjj;
    for (int i = j + 1; i < b; i++) {
      double t = G[(size_t)j + (size_t)i * b];
      for (int k = 0; k < j; k++) t -= R[(size_t)k + (size_t)j * b] * R[(size_t)k + (size_t)i * b];
      R[(size_t)j + (size_t)i * b] = t / rjj;
    }
  }
  return b;
}

// inverse of an upper triangular b x b matrix
inline void inv_upper(int b, const std::vector<double> &R, std::vector<double> &Ri) {
  Ri.assign((size_t)b * b, 0.0);
  for (int j = 0; j < b; j++) {
    Ri[(size_t)j + (size_t)j * b] = 1.0 / R[(size_t)j + (size_t)j * b];
    for (int i = j - 1; i >= 0; i--) {
      double s = 0;
      for (int k = i + 1; k <= j; k++) s += R[(size_t)i + (size_t)k * b] * Ri[(size_t)k + (size_t)j * b];
      Ri[(size_t)i + (size_t)j * b] = -s / R[(size_t)i + (size_t)i * b];
    }
  }
}

// C = A * B for small column-major matrices (ra x ca) * (ca x cb)
inline void small_mm(int ra, int ca, int cb, const double *A, const double *B, double *C) {
  for (int j = 0; j < cb; j++)
    for (int i = 0; i < ra; i++) {
      double s = 0;
      for (int k = 0; k < ca; k++) s += A[(size_t)i + (size_t)k * ra] * B[(size_t)k + (size_t)j * ca];
      C[(size_t)i + (size_t)j * ra] = s;
    }
}

}  // namespace bsn
