/*
 * bigsnpr_hip_shim.c — the `.Call` shim a bigsnpr maintainer adds to bind the reference's
 * R layer to libbigsnpr_hip.so (C ABI: include/bigsnpr_hip.h).
 *
 * NOT compiled in this repository's CI: the build image has no R (no Rinternals.h).  It is
 * deliberately thin — every function only converts SEXP arguments to plain pointers,
 * subtracts 1 from R's 1-based indices where the reference does (src/bed-acc.h:64-65) and
 * turns a non-zero return code into Rf_error(bsn_last_error()) — so that all logic is
 * tested through the C ABI by tests/.
 *
 * Symbol names and arities are those of the reference's registration table
 * (src/RcppExports.cpp:597-640), so R/RcppExports.R keeps working unchanged.
 *
 * Build (where R exists):
 *   R CMD SHLIB bigsnpr_hip_shim.c -I<repo>/include -L<repo>/bigsnpr_amd -lbigsnpr_hip
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <stdint.h>
#include <stdlib.h>

#include "bigsnpr_hip.h"

#define CHECK(call) do { if ((call) != 0) Rf_error("%s", bsn_last_error()); } while (0)

/* ---- handle <-> externalptr (replaces XPtr<bed>, src/bed-acc-xptr.cpp:46) ------------- */
static void bed_finalizer(SEXP xp) {
  bsn_bed *b = (bsn_bed *) R_ExternalPtrAddr(xp);
  if (b) { bsn_bed_close(b); R_ClearExternalPtr(xp); }
}

/* obj_bed is the RC object (an environment); "address" is the active binding that lazily
 * calls bedXPtr (R/bed-class.R:105-110) — read exactly like src/bed-prod-vec.cpp:23 */
static bsn_bed *get_bed(SEXP obj_bed) {
  SEXP xp = Rf_eval(Rf_lang3(Rf_install("$"), obj_bed, Rf_install("address")), R_GlobalEnv);
  bsn_bed *b = (bsn_bed *) R_ExternalPtrAddr(xp);
  if (!b) Rf_error("external pointer is not valid");
  return b;
}

/* 1-based int32 -> 0-based int64 (R_alloc memory is reclaimed at the end of .Call) */
static int64_t *ind0(SEXP v) {
  R_xlen_t n = XLENGTH(v);
  int64_t *out = (int64_t *) R_alloc((size_t) n, sizeof(int64_t));
  const int *p = INTEGER(v);
  for (R_xlen_t i = 0; i < n; i++) out[i] = (int64_t) p[i] - 1;
  return out;
}

static void assert_size(R_xlen_t a, R_xlen_t b) { /* bigstatsr myassert_size */
  if (a != b) Rf_error("Tested %ld == %ld. %s", (long) a, (long) b,
                       "Incompatibility between dimensions.");
}

/* _bigsnpr_bedXPtr(path, n, p) */
SEXP _bigsnpr_bedXPtr(SEXP path, SEXP n, SEXP p) {
  bsn_bed *b = NULL;
  CHECK(bsn_bed_open(CHAR(STRING_ELT(path, 0)), (int64_t) Rf_asInteger(n),
                     (int64_t) Rf_asInteger(p), &b));
  SEXP xp = PROTECT(R_MakeExternalPtr(b, R_NilValue, R_NilValue));
  R_RegisterCFinalizerEx(xp, bed_finalizer, TRUE);
  UNPROTECT(1);
  return xp;
}

/* _bigsnpr_bed_pMatVec4(obj_bed, ind_row, ind_col, center, scale, x, ncores) */
SEXP _bigsnpr_bed_pMatVec4(SEXP obj_bed, SEXP ind_row, SEXP ind_col, SEXP center, SEXP scale,
                           SEXP x, SEXP ncores) {
  R_xlen_t n = XLENGTH(ind_row), m = XLENGTH(ind_col);
  assert_size(XLENGTH(center), m); assert_size(XLENGTH(scale), m); assert_size(XLENGTH(x), m);
  SEXP res = PROTECT(Rf_allocVector(REALSXP, n));
  CHECK(bsn_bed_prodvec(get_bed(obj_bed), ind0(ind_row), n, ind0(ind_col), m, REAL(center),
                        REAL(scale), REAL(x), REAL(res)));
  UNPROTECT(1);
  return res;
}

/* _bigsnpr_bed_cpMatVec4(obj_bed, ind_row, ind_col, center, scale, x, ncores) */
SEXP _bigsnpr_bed_cpMatVec4(SEXP obj_bed, SEXP ind_row, SEXP ind_col, SEXP center, SEXP scale,
                            SEXP x, SEXP ncores) {
  R_xlen_t n = XLENGTH(ind_row), m = XLENGTH(ind_col);
  assert_size(XLENGTH(center), m); assert_size(XLENGTH(scale), m); assert_size(XLENGTH(x), n);
  SEXP res = PROTECT(Rf_allocVector(REALSXP, m));
  CHECK(bsn_bed_cprodvec(get_bed(obj_bed), ind0(ind_row), n, ind0(ind_col), m, REAL(center),
                         REAL(scale), REAL(x), REAL(res)));
  UNPROTECT(1);
  return res;
}

/* _bigsnpr_bed_colstats(obj_bed, ind_row, ind_col, ncores) -> list(sumX, denoX, nb_nona_col) */
SEXP _bigsnpr_bed_colstats(SEXP obj_bed, SEXP ind_row, SEXP ind_col, SEXP ncores) {
  R_xlen_t n = XLENGTH(ind_row), m = XLENGTH(ind_col);
  SEXP sumX = PROTECT(Rf_allocVector(REALSXP, m)), denoX = PROTECT(Rf_allocVector(REALSXP, m));
  SEXP nona = PROTECT(Rf_allocVector(INTSXP, m));
  int32_t n_bad = 0;
  CHECK(bsn_bed_colstats(get_bed(obj_bed), ind0(ind_row), n, ind0(ind_col), m, REAL(sumX),
                         REAL(denoX), INTEGER(nona), &n_bad));
  if (n_bad > 0) Rf_warning("%d variants have >50%% missing values.", n_bad); /* src/bed-fun.cpp:41 */
  SEXP res = PROTECT(Rf_allocVector(VECSXP, 3)), nm = PROTECT(Rf_allocVector(STRSXP, 3));
  SET_VECTOR_ELT(res, 0, sumX); SET_VECTOR_ELT(res, 1, denoX); SET_VECTOR_ELT(res, 2, nona);
  SET_STRING_ELT(nm, 0, Rf_mkChar("sumX")); SET_STRING_ELT(nm, 1, Rf_mkChar("denoX"));
  SET_STRING_ELT(nm, 2, Rf_mkChar("nb_nona_col"));
  Rf_setAttrib(res, R_NamesSymbol, nm);
  UNPROTECT(5);
  return res;
}

/* _bigsnpr_bed_col_counts_cpp(obj_bed, ind_row, ind_col, ncores) -> 4 x m integer matrix */
SEXP _bigsnpr_bed_col_counts_cpp(SEXP obj_bed, SEXP ind_row, SEXP ind_col, SEXP ncores) {
  R_xlen_t n = XLENGTH(ind_row), m = XLENGTH(ind_col);
  SEXP res = PROTECT(Rf_allocMatrix(INTSXP, 4, (int) m));
  CHECK(bsn_bed_col_counts(get_bed(obj_bed), ind0(ind_row), n, ind0(ind_col), m, INTEGER(res)));
  UNPROTECT(1);
  return res;
}

/* _bigsnpr_read_bed(obj_bed, ind_row, ind_col) -> integer matrix with NA_INTEGER */
SEXP _bigsnpr_read_bed(SEXP obj_bed, SEXP ind_row, SEXP ind_col) {
  R_xlen_t n = XLENGTH(ind_row), m = XLENGTH(ind_col);
  SEXP res = PROTECT(Rf_allocMatrix(INTSXP, (int) n, (int) m));
  CHECK(bsn_bed_read(get_bed(obj_bed), ind0(ind_row), n, ind0(ind_col), m, NA_INTEGER, INTEGER(res)));
  UNPROTECT(1);
  return res;
}

/* _bigsnpr_read_bed_scaled(obj_bed, ind_row, ind_col, center, scale) */
SEXP _bigsnpr_read_bed_scaled(SEXP obj_bed, SEXP ind_row, SEXP ind_col, SEXP center, SEXP scale) {
  R_xlen_t n = XLENGTH(ind_row), m = XLENGTH(ind_col);
  assert_size(XLENGTH(center), m); assert_size(XLENGTH(scale), m);
  SEXP res = PROTECT(Rf_allocMatrix(REALSXP, (int) n, (int) m));
  CHECK(bsn_bed_read_scaled(get_bed(obj_bed), ind0(ind_row), n, ind0(ind_col), m, REAL(center),
                            REAL(scale), REAL(res)));
  UNPROTECT(1);
  return res;
}

/* NEW entry: the whole partial SVD on the device.  R side:
 *   bed_randomSVD <- function(obj.bed, fun.scaling = bed_scaleBinom, ind.row, ind.col, k = 10,
 *                             tol = 1e-4, verbose = FALSE, ncores = 1) {
 *     ms  <- fun.scaling(obj.bed, ind.row = ind.row, ind.col = ind.col, ncores = ncores)
 *     res <- .Call(`_bigsnpr_bed_randomSVD_hip`, obj.bed$light, ind.row, ind.col,
 *                  ms$center, ms$scale, k, tol, verbose)
 *     structure(c(res, list(center = ms$center, scale = ms$scale)), class = "big_SVD")
 *   }
 * (replaces the big_randomSVD call of R/autoSVD.R:216-218; same result fields) */
SEXP _bigsnpr_bed_randomSVD_hip(SEXP obj_bed, SEXP ind_row, SEXP ind_col, SEXP center, SEXP scale,
                                SEXP k_, SEXP tol, SEXP verbose) {
  R_xlen_t n = XLENGTH(ind_row), m = XLENGTH(ind_col);
  int k = Rf_asInteger(k_);
  assert_size(XLENGTH(center), m); assert_size(XLENGTH(scale), m);
  bsn_svd_options o = {0};
  o.k = k; o.tol = Rf_asReal(tol); o.verbose = Rf_asLogical(verbose);
  bsn_svd_info info;
  SEXP d = PROTECT(Rf_allocVector(REALSXP, k));
  SEXP u = PROTECT(Rf_allocMatrix(REALSXP, (int) n, k)), v = PROTECT(Rf_allocMatrix(REALSXP, (int) m, k));
  CHECK(bsn_bed_randomsvd(get_bed(obj_bed), ind0(ind_row), n, ind0(ind_col), m, REAL(center),
                          REAL(scale), &o, REAL(d), REAL(u), REAL(v), &info));
  const char *names[] = {"d", "u", "v", "niter", "nops", ""};
  SEXP res = PROTECT(Rf_mkNamed(VECSXP, names));
  SET_VECTOR_ELT(res, 0, d); SET_VECTOR_ELT(res, 1, u); SET_VECTOR_ELT(res, 2, v);
  SET_VECTOR_ELT(res, 3, Rf_ScalarInteger(info.niter));
  SET_VECTOR_ELT(res, 4, Rf_ScalarInteger(info.nops));
  UNPROTECT(4);
  return res;
}

/* _bigsnpr_bed_row_counts_cpp(obj_bed, ind_row, ind_col, ncores) -> 4 x n integer matrix */
SEXP _bigsnpr_bed_row_counts_cpp(SEXP obj_bed, SEXP ind_row, SEXP ind_col, SEXP ncores) {
  R_xlen_t n = XLENGTH(ind_row), m = XLENGTH(ind_col);
  SEXP res = PROTECT(Rf_allocMatrix(INTSXP, 4, (int) n));
  CHECK(bsn_bed_row_counts(get_bed(obj_bed), ind0(ind_row), n, ind0(ind_col), m, INTEGER(res)));
  UNPROTECT(1);
  return res;
}

/* _bigsnpr_prod_and_rowSumsSq(obj_bed, ind_row, ind_col, center, scale, V) -> list(XV, rowSumsSq) */
SEXP _bigsnpr_prod_and_rowSumsSq(SEXP obj_bed, SEXP ind_row, SEXP ind_col, SEXP center, SEXP scale,
                                 SEXP V) {
  R_xlen_t n = XLENGTH(ind_row), m = XLENGTH(ind_col);
  assert_size(Rf_nrows(V), m);                                   /* src/bed-fun.cpp:114 */
  int K = Rf_ncols(V);
  SEXP XV = PROTECT(Rf_allocMatrix(REALSXP, (int) n, K)), rs = PROTECT(Rf_allocVector(REALSXP, n));
  CHECK(bsn_bed_prod_and_rowsumssq(get_bed(obj_bed), ind0(ind_row), n, ind0(ind_col), m, REAL(center),
                                   REAL(scale), REAL(V), K, REAL(XV), REAL(rs)));
  const char *names[] = {"XV", "rowSumsSq", ""};
  SEXP res = PROTECT(Rf_mkNamed(VECSXP, names));
  SET_VECTOR_ELT(res, 0, XV); SET_VECTOR_ELT(res, 1, rs);
  UNPROTECT(3);
  return res;
}

/* ---- LD / clumping: the accessor is either a bed object or an FBM.code256 ---------------
 * (type dispatch by the presence of the "code256" field, as src/corr.cpp:113-125 does).
 * For an FBM the .bk file (one byte per genotype, column-major, R/bigSNP-class.R:7) is mapped
 * here and repacked to the 2-bit device image once; the handle is cached per backing file. */
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
typedef struct fbm_cache { char *path; bsn_bed *img; struct fbm_cache *next; } fbm_cache;
static fbm_cache *g_fbm = NULL;

static SEXP field(SEXP obj, const char *name) {
  return Rf_eval(Rf_lang3(Rf_install("$"), obj, Rf_install(name)), R_GlobalEnv);
}
static int has_field(SEXP obj, const char *name) {
  return Rf_findVarInFrame3(obj, Rf_install(name), FALSE) != R_UnboundValue;
}
static bsn_bed *get_image(SEXP obj) {
  if (!has_field(obj, "code256")) return get_bed(obj);
  const char *bk = CHAR(STRING_ELT(field(obj, "backingfile"), 0));
  for (fbm_cache *c = g_fbm; c; c = c->next)
    if (strcmp(c->path, bk) == 0) return c->img;
  int64_t n = (int64_t) Rf_asReal(field(obj, "nrow")), m = (int64_t) Rf_asReal(field(obj, "ncol"));
  int fd = open(bk, O_RDONLY);
  if (fd < 0) Rf_error("cannot open backing file '%s'", bk);
  void *map = mmap(NULL, (size_t) (n * m), PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (map == MAP_FAILED) Rf_error("cannot map backing file '%s'", bk);
  bsn_bed *img = NULL;
  int rc = bsn_bed_from_fbm((const uint8_t *) map, n, m, n, &img);
  munmap(map, (size_t) (n * m));
  if (rc != 0) Rf_error("%s", bsn_last_error());
  fbm_cache *c = (fbm_cache *) malloc(sizeof(fbm_cache));
  c->path = strdup(bk); c->img = img; c->next = g_fbm; g_fbm = c;
  return img;
}
static int32_t *ind0_32(SEXP v) {
  R_xlen_t n = XLENGTH(v);
  int32_t *out = (int32_t *) R_alloc((size_t) n, sizeof(int32_t));
  for (R_xlen_t i = 0; i < n; i++) out[i] = INTEGER(v)[i] - 1;
  return out;
}

/* _bigsnpr_corMat(obj, rowInd, colInd, size, thr, pos, fill_diag, ncores) -> list(i, p, x)
 * (the R wrapper of R/corr.R:43-47 builds the dsCMatrix from it, unchanged) */
SEXP _bigsnpr_corMat(SEXP obj, SEXP rowInd, SEXP colInd, SEXP size, SEXP thr, SEXP pos,
                     SEXP fill_diag, SEXP ncores) {
  R_xlen_t n = XLENGTH(rowInd), m = XLENGTH(colInd);
  assert_size(XLENGTH(thr), n); assert_size(XLENGTH(pos), m);
  SEXP p = PROTECT(Rf_allocVector(INTSXP, m + 1));
  int64_t nnz = 0; bsn_cor *cor = NULL;
  CHECK(bsn_cormat(get_image(obj), ind0(rowInd), n, ind0(colInd), m, Rf_asReal(size), REAL(thr),
                   REAL(pos), Rf_asLogical(fill_diag), INTEGER(p), &nnz, &cor));
  SEXP i = PROTECT(Rf_allocVector(INTSXP, (R_xlen_t) nnz)), x = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t) nnz));
  int rc = bsn_cormat_fetch(cor, INTEGER(i), REAL(x));
  bsn_cormat_free(cor);
  if (rc != 0) Rf_error("%s", bsn_last_error());
  const char *names[] = {"i", "p", "x", ""};
  SEXP res = PROTECT(Rf_mkNamed(VECSXP, names));
  SET_VECTOR_ELT(res, 0, i); SET_VECTOR_ELT(res, 1, p); SET_VECTOR_ELT(res, 2, x);
  UNPROTECT(4);
  return res;
}

/* _bigsnpr_ld_scores(obj, rowInd, colInd, size, pos, ncores) */
SEXP _bigsnpr_ld_scores(SEXP obj, SEXP rowInd, SEXP colInd, SEXP size, SEXP pos, SEXP ncores) {
  R_xlen_t n = XLENGTH(rowInd), m = XLENGTH(colInd);
  assert_size(XLENGTH(pos), m);
  SEXP res = PROTECT(Rf_allocVector(REALSXP, m));
  CHECK(bsn_ld_scores(get_image(obj), ind0(rowInd), n, ind0(colInd), m, Rf_asReal(size), REAL(pos),
                      REAL(res)));
  UNPROTECT(1);
  return res;
}

/* `keep` of the reference is a 1 x m integer FBM written in place (R/clumping.R:116,
 * R/bed-clumping.R:53); its bytes are reached through the mapped address of BM2.  Here the
 * result comes back as an integer vector and the two R callers do `keep[] <- .Call(...)`. */
static SEXP clump(SEXP obj, int mode, SEXP rowInd, SEXP colInd, SEXP ordInd, SEXP rankInd, SEXP pos,
                  SEXP aux1, SEXP aux2, SEXP size, SEXP thr) {
  R_xlen_t n = XLENGTH(rowInd), m = XLENGTH(colInd);
  assert_size(XLENGTH(pos), m); assert_size(XLENGTH(aux1), m); assert_size(XLENGTH(aux2), m);
  SEXP keep = PROTECT(Rf_allocVector(INTSXP, m));
  CHECK(bsn_clumping_chr(get_image(obj), ind0(rowInd), n, ind0(colInd), m, mode, REAL(aux1), REAL(aux2),
                         ind0_32(ordInd), ind0_32(rankInd), REAL(pos), Rf_asReal(size), Rf_asReal(thr),
                         INTEGER(keep)));
  UNPROTECT(1);
  return keep;
}
/* _bigsnpr_clumping_chr(BM, BM2, rowInd, colInd, ordInd, rankInd, pos, sumX, denoX, size, thr, ncores) */
SEXP _bigsnpr_clumping_chr(SEXP BM, SEXP BM2, SEXP rowInd, SEXP colInd, SEXP ordInd, SEXP rankInd,
                           SEXP pos, SEXP sumX, SEXP denoX, SEXP size, SEXP thr, SEXP ncores) {
  return clump(BM, 0, rowInd, colInd, ordInd, rankInd, pos, sumX, denoX, size, thr);
}
/* _bigsnpr_bed_clumping_chr(obj_bed, BM2, ind_row, ind_col, center, scale, ordInd, rankInd, pos, size, thr, ncores) */
SEXP _bigsnpr_bed_clumping_chr(SEXP obj_bed, SEXP BM2, SEXP ind_row, SEXP ind_col, SEXP center, SEXP scale,
                               SEXP ordInd, SEXP rankInd, SEXP pos, SEXP size, SEXP thr, SEXP ncores) {
  return clump(obj_bed, 1, ind_row, ind_col, ordInd, rankInd, pos, center, scale, size, thr);
}

static const R_CallMethodDef CallEntries[] = {
  {"_bigsnpr_bedXPtr", (DL_FUNC) &_bigsnpr_bedXPtr, 3},
  {"_bigsnpr_bed_colstats", (DL_FUNC) &_bigsnpr_bed_colstats, 4},
  {"_bigsnpr_bed_col_counts_cpp", (DL_FUNC) &_bigsnpr_bed_col_counts_cpp, 4},
  {"_bigsnpr_read_bed", (DL_FUNC) &_bigsnpr_read_bed, 3},
  {"_bigsnpr_read_bed_scaled", (DL_FUNC) &_bigsnpr_read_bed_scaled, 5},
  {"_bigsnpr_bed_pMatVec4", (DL_FUNC) &_bigsnpr_bed_pMatVec4, 7},
  {"_bigsnpr_bed_cpMatVec4", (DL_FUNC) &_bigsnpr_bed_cpMatVec4, 7},
  {"_bigsnpr_bed_randomSVD_hip", (DL_FUNC) &_bigsnpr_bed_randomSVD_hip, 8},
  {"_bigsnpr_bed_row_counts_cpp", (DL_FUNC) &_bigsnpr_bed_row_counts_cpp, 4},
  {"_bigsnpr_prod_and_rowSumsSq", (DL_FUNC) &_bigsnpr_prod_and_rowSumsSq, 6},
  {"_bigsnpr_corMat", (DL_FUNC) &_bigsnpr_corMat, 8},
  {"_bigsnpr_ld_scores", (DL_FUNC) &_bigsnpr_ld_scores, 6},
  {"_bigsnpr_clumping_chr", (DL_FUNC) &_bigsnpr_clumping_chr, 12},
  {"_bigsnpr_bed_clumping_chr", (DL_FUNC) &_bigsnpr_bed_clumping_chr, 12},
  {NULL, NULL, 0}
};

void R_init_bigsnprhip(DllInfo *dll) {
  R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
