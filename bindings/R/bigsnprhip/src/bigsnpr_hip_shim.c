/*
 * bigsnpr_hip_shim.c — the `.Call` shim a bigsnpr maintainer adds to bind the reference's
 * R layer to libbigsnpr_hip.so (C ABI: include/bigsnpr_hip.h).
 *
 * The build image has no R.  tests/test_r_shim_cpu.py compiles this file (-Wall -Wextra -Werror) against
 * declarations of the R API it uses (tests/rstub/include: test infrastructure written from "Writing R
 * Extensions") and checks its registration table against the reference's; tests/test_gpu_r_shim.py runs
 * its entry points on a GPU through a small stand-in runtime (tests/rstub/rstub.c) with R-like objects and
 * compares with the oracle.  It is deliberately thin — every function only converts SEXP arguments to plain pointers,
 * subtracts 1 from R's 1-based indices where the reference does (src/bed-acc.h:64-65) and
 * turns a non-zero return code into Rf_error(bsn_last_error()) — so that all logic is
 * tested through the C ABI by tests/.
 *
 * Symbol names and arities are those of the reference's registration table
 * (src/RcppExports.cpp:597-640), so R/RcppExports.R and every R caller keep working unchanged;
 * functions with the suffix _hip are additions (whole-solve SVD, FBM mat-vec operators).
 *
 * Build (where R exists): this file is src/ of the package bindings/R/bigsnprhip —
 *   BIGSNPR_HIP_HOME=<repo> R CMD INSTALL bindings/R/bigsnprhip      (src/Makevars holds the link line)
 * or stand-alone:  R CMD SHLIB bigsnpr_hip_shim.c -I<repo>/include -L<repo>/bigsnpr_amd -lbigsnpr_hip
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "bigsnpr_hip.h"

#define CHECK(call) do { if ((call) != 0) Rf_error("%s", bsn_last_error()); } while (0)

/* ---- handle <-> externalptr (replaces XPtr<bed>, src/bed-acc-xptr.cpp:46) ------------- */
static void bed_finalizer(SEXP xp) {
  bsn_bed *b = (bsn_bed *) R_ExternalPtrAddr(xp);
  if (b) { bsn_bed_close(b); R_ClearExternalPtr(xp); }
}

/* obj$name, evaluated with the call object protected (an active binding may allocate) */
static SEXP field(SEXP obj, const char *name) {
  SEXP call = PROTECT(Rf_lang3(Rf_install("$"), obj, Rf_install(name)));
  SEXP val = Rf_eval(call, R_GlobalEnv);
  UNPROTECT(1);
  return val;
}
static int has_field(SEXP obj, const char *name) {
  return Rf_findVarInFrame3(obj, Rf_install(name), FALSE) != R_UnboundValue;
}

/* obj_bed is the RC object (an environment); "address" is the active binding that lazily
 * calls bedXPtr (R/bed-class.R:105-110) — read exactly like src/bed-prod-vec.cpp:23.  The
 * external pointer it returns is owned by the object's `extptr` field, so the handle outlives
 * this call. */
static bsn_bed *get_bed(SEXP obj_bed) {
  SEXP xp = PROTECT(field(obj_bed, "address"));
  bsn_bed *b = (bsn_bed *) R_ExternalPtrAddr(xp);
  UNPROTECT(1);
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
static int32_t *ind0_32(SEXP v) {
  R_xlen_t n = XLENGTH(v);
  int32_t *out = (int32_t *) R_alloc((size_t) n, sizeof(int32_t));
  for (R_xlen_t i = 0; i < n; i++) out[i] = INTEGER(v)[i] - 1;
  return out;
}

static void assert_size(R_xlen_t a, R_xlen_t b) { /* bigstatsr myassert_size */
  if (a != b) Rf_error("Tested %ld == %ld. %s", (long) a, (long) b,
                       "Incompatibility between dimensions.");
}
static void nan_to_na(double *x, R_xlen_t len) {
  for (R_xlen_t i = 0; i < len; i++) if (ISNAN(x[i])) x[i] = NA_REAL;
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
  const char *names[] = {"sumX", "denoX", "nb_nona_col", ""};
  SEXP res = PROTECT(Rf_mkNamed(VECSXP, names));
  SET_VECTOR_ELT(res, 0, sumX); SET_VECTOR_ELT(res, 1, denoX); SET_VECTOR_ELT(res, 2, nona);
  UNPROTECT(4);
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

/* _bigsnpr_bed_row_counts_cpp(obj_bed, ind_row, ind_col, ncores) -> 4 x n integer matrix */
SEXP _bigsnpr_bed_row_counts_cpp(SEXP obj_bed, SEXP ind_row, SEXP ind_col, SEXP ncores) {
  R_xlen_t n = XLENGTH(ind_row), m = XLENGTH(ind_col);
  SEXP res = PROTECT(Rf_allocMatrix(INTSXP, 4, (int) n));
  CHECK(bsn_bed_row_counts(get_bed(obj_bed), ind0(ind_row), n, ind0(ind_col), m, INTEGER(res)));
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

/* The partial SVD, whole, on the device (an addition; replaces the big_randomSVD call of
 * R/autoSVD.R:216-218, same result fields).  R side:
 *   bed_randomSVD <- function(obj.bed, fun.scaling = bed_scaleBinom, ind.row, ind.col, k = 10,
 *                             tol = 1e-4, verbose = FALSE, ncores = 1) {
 *     ms <- if (identical(fun.scaling, bed_scaleBinom)) NULL else   # NULL: scaling inside the solve
 *       fun.scaling(obj.bed, ind.row = ind.row, ind.col = ind.col, ncores = ncores)
 *     structure(.Call(`_bigsnpr_bed_randomSVD_hip`, obj.bed$light, ind.row, ind.col,
 *                     ms$center, ms$scale, k, tol, verbose), class = "big_SVD")
 *   }
 * center = scale = NULL selects bed_scaleBinom evaluated inside the solve (its code counts ride
 * along the first crossproduct pass); the values used come back in the list either way.
 *
 * What a caller gets (round 5): d to 1e-9 of the reference's; u and v as an fp64 Lanczos solve
 * stopped at the same tol leaves them — the early block steps run on 24-bit panels (bsn_svd_options.vec_floor,
 * 1e-7), so the rounding of the panels is not visible in the vectors: at 400K x 1M, k = 20 the leading half of the
 * vectors lies within 1.6e-7 (u) / 8e-9 (v) of a tol-1e-10 solve, all of them within 1.3e-5 / 1.2e-6 (the k-th pair
 * is converged to tol, like RSpectra's).  `tol` may carry up to three more numbers for callers who want another
 * point of the accuracy / cost frontier:  tol = c(tol, slices, block, vec.floor)  — slices: int8 digits of the panels
 * at every step (0 = automatic), block: vectors per pass (0 = automatic), vec.floor: residual floor wanted for the
 * vectors (0 = default, < 0 = none: every step on `slices` digits, round 4's 16-bit behaviour: 15 % faster, leading
 * vectors at 2e-5). */
static SEXP random_svd(bsn_bed *img, SEXP ind_row, SEXP ind_col, SEXP center, SEXP scale, SEXP k_,
                       SEXP tol, SEXP verbose) {
  R_xlen_t n = XLENGTH(ind_row), m = XLENGTH(ind_col);
  int k = Rf_asInteger(k_);
  const int binom = Rf_isNull(center) || Rf_isNull(scale);
  bsn_svd_options o;
  memset(&o, 0, sizeof(o));
  o.k = k; o.tol = Rf_asReal(tol); o.verbose = Rf_asLogical(verbose);
  if (TYPEOF(tol) == REALSXP) {   /* tol = c(tol, slices, block, vec.floor) */
    const R_xlen_t nt = XLENGTH(tol);
    if (nt > 1) o.slices = (int32_t) REAL(tol)[1];
    if (nt > 2) o.block = (int32_t) REAL(tol)[2];
    if (nt > 3) o.vec_floor = REAL(tol)[3];
  }
  bsn_svd_info info;
  SEXP d = PROTECT(Rf_allocVector(REALSXP, k));
  SEXP u = PROTECT(Rf_allocMatrix(REALSXP, (int) n, k)), v = PROTECT(Rf_allocMatrix(REALSXP, (int) m, k));
  SEXP ce = PROTECT(binom ? Rf_allocVector(REALSXP, m) : center);
  SEXP sc = PROTECT(binom ? Rf_allocVector(REALSXP, m) : scale);
  if (!binom) { assert_size(XLENGTH(center), m); assert_size(XLENGTH(scale), m); }
  o.binom_scaling = binom;
  if (binom) { o.center_out = REAL(ce); o.scale_out = REAL(sc); }
  int rc = bsn_bed_randomsvd(img, ind0(ind_row), n, ind0(ind_col), m, binom ? NULL : REAL(center),
                             binom ? NULL : REAL(scale), &o, REAL(d), REAL(u), REAL(v), &info);
  if (rc == 2) Rf_warning("%s", bsn_last_error());  /* RSpectra::svds warns when fewer than k triplets converged */
  else if (rc != 0) Rf_error("%s", bsn_last_error());
  if (binom && info.n_bad > 0) Rf_warning("%d variants have >50%% missing values.", info.n_bad);
  const char *names[] = {"d", "u", "v", "niter", "nops", "center", "scale", ""};
  SEXP res = PROTECT(Rf_mkNamed(VECSXP, names));
  SET_VECTOR_ELT(res, 0, d); SET_VECTOR_ELT(res, 1, u); SET_VECTOR_ELT(res, 2, v);
  SET_VECTOR_ELT(res, 3, Rf_ScalarInteger(info.niter));
  SET_VECTOR_ELT(res, 4, Rf_ScalarInteger(info.nops));
  SET_VECTOR_ELT(res, 5, ce); SET_VECTOR_ELT(res, 6, sc);
  UNPROTECT(6);
  return res;
}
SEXP _bigsnpr_bed_randomSVD_hip(SEXP obj_bed, SEXP ind_row, SEXP ind_col, SEXP center, SEXP scale,
                                SEXP k_, SEXP tol, SEXP verbose) {
  return random_svd(get_bed(obj_bed), ind_row, ind_col, center, scale, k_, tol, verbose);
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

/* ---- FBM.code256 objects: the accessor is either a bed object or an FBM -------------------
 * (type dispatch by the presence of the "code256" field, as src/corr.cpp:113-125 does).
 * For an FBM the .bk file (one byte per genotype, column-major, R/bigSNP-class.R:7) is mapped
 * here and repacked to the device image once.  The handle is cached per backing file together
 * with the file's size and modification time: an FBM written in place since (imputation,
 * snp_fastImpute) is uploaded again instead of serving stale genotypes. */
typedef struct fbm_cache {
  char *path; bsn_bed *img; off_t size; time_t mtime; long mtime_ns;
  double code[256];        /* the decode table the image was built with: G$copy(code = ...) shares the file */
  struct fbm_cache *next;
} fbm_cache;
static fbm_cache *g_fbm = NULL;

static bsn_bed *upload_fbm(const char *bk, int64_t n, int64_t m, const double *code256) {
  int fd = open(bk, O_RDONLY);
  if (fd < 0) Rf_error("cannot open backing file '%s'", bk);
  void *map = mmap(NULL, (size_t) (n * m), PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (map == MAP_FAILED) Rf_error("cannot map backing file '%s'", bk);
  bsn_bed *img = NULL;
  /* the object's own decode table (CODE_012, CODE_IMPUTE_PRED, CODE_DOSAGE ...; NA_real_ is a NaN) */
  int rc = bsn_fbm_open((const uint8_t *) map, n, m, n, code256, &img);
  munmap(map, (size_t) (n * m));
  if (rc != 0) Rf_error("%s", bsn_last_error());
  return img;
}
/* the device image of (backing file, decode table): one per pair — the reference routinely attaches the same
 * file with several tables (CODE_012 -> CODE_IMPUTE_PRED / CODE_DOSAGE, R/impute.R:149-201; G.round of
 * R/write-plink.R:35), and a table changes which bytes are missing, imputed or dosages.  A file written in
 * place since (snp_fastImpute) is uploaded again for every table. */
static bsn_bed *image_of(const char *bk, int64_t n, int64_t m, const double *code256) {
  struct stat st;
  if (stat(bk, &st) != 0) Rf_error("cannot stat backing file '%s'", bk);
  for (fbm_cache *c = g_fbm; c; c = c->next)
    if (strcmp(c->path, bk) == 0 && memcmp(c->code, code256, sizeof(c->code)) == 0) {
      if (c->size == st.st_size && c->mtime == st.st_mtim.tv_sec && c->mtime_ns == st.st_mtim.tv_nsec)
        return c->img;
      bsn_bed_close(c->img);                       /* the file changed under the handle */
      c->img = NULL;
      c->img = upload_fbm(bk, n, m, code256);
      c->size = st.st_size; c->mtime = st.st_mtim.tv_sec; c->mtime_ns = st.st_mtim.tv_nsec;
      return c->img;
    }
  bsn_bed *img = upload_fbm(bk, n, m, code256);     /* (an R error here leaves the cache untouched) */
  fbm_cache *c = (fbm_cache *) malloc(sizeof(fbm_cache));
  if (!c) { bsn_bed_close(img); Rf_error("out of memory"); }
  c->path = strdup(bk); c->img = img;
  memcpy(c->code, code256, sizeof(c->code));
  c->size = st.st_size; c->mtime = st.st_mtim.tv_sec; c->mtime_ns = st.st_mtim.tv_nsec;
  c->next = g_fbm; g_fbm = c;
  return c->img;
}
/* the (n, m, backing file, decode table) of an FBM.code256 object */
static const double *fbm_fields(SEXP obj, const char **bk, int64_t *n, int64_t *m) {
  *bk = CHAR(STRING_ELT(field(obj, "backingfile"), 0));
  *n = (int64_t) Rf_asReal(field(obj, "nrow"));
  *m = (int64_t) Rf_asReal(field(obj, "ncol"));
  SEXP code = PROTECT(field(obj, "code256"));
  if (TYPEOF(code) != REALSXP || XLENGTH(code) != 256) Rf_error("'code256' must be 256 doubles");
  const double *code256 = REAL(code);
  UNPROTECT(1);   /* owned by the FBM object for the duration of the call */
  return code256;
}
static bsn_bed *get_image(SEXP obj) {
  if (!has_field(obj, "code256")) return get_bed(obj);
  const char *bk; int64_t n, m;
  const double *code256 = fbm_fields(obj, &bk, &n, &m);
  return image_of(bk, n, m, code256);
}
/* drops every cached FBM image (R: .Call(`_bigsnpr_fbm_cache_clear_hip`)) */
SEXP _bigsnpr_fbm_cache_clear_hip(void) {
  while (g_fbm) { fbm_cache *c = g_fbm; g_fbm = c->next; if (c->img) bsn_bed_close(c->img); free(c->path); free(c); }
  return R_NilValue;
}

/* writes `len` values of `elt` bytes at element offset 0 of the FBM's backing file through a shared
 * mapping — what the reference does through BM$address_rw (src/clumping.cpp:28-29) */
static void write_backing(SEXP BM, const void *src, size_t bytes) {
  const char *bk = CHAR(STRING_ELT(field(BM, "backingfile"), 0));
  int fd = open(bk, O_RDWR);
  if (fd < 0) Rf_error("cannot open backing file '%s' for writing", bk);
  void *map = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (map == MAP_FAILED) Rf_error("cannot map backing file '%s'", bk);
  memcpy(map, src, bytes);
  munmap(map, bytes);
}

/* _bigsnpr_snp_colstats(BM, rowInd, colInd, ncores) -> list(sumX, denoX)   src/colstats.cpp:8-35 */
SEXP _bigsnpr_snp_colstats(SEXP BM, SEXP rowInd, SEXP colInd, SEXP ncores) {
  R_xlen_t n = XLENGTH(rowInd), m = XLENGTH(colInd);
  SEXP sumX = PROTECT(Rf_allocVector(REALSXP, m)), denoX = PROTECT(Rf_allocVector(REALSXP, m));
  CHECK(bsn_snp_colstats(get_image(BM), ind0(rowInd), n, ind0(colInd), m, REAL(sumX), REAL(denoX)));
  const char *names[] = {"sumX", "denoX", ""};
  SEXP res = PROTECT(Rf_mkNamed(VECSXP, names));
  SET_VECTOR_ELT(res, 0, sumX); SET_VECTOR_ELT(res, 1, denoX);
  UNPROTECT(3);
  return res;
}

/* _bigsnpr_multLinReg(obj, ind_row, ind_col, U, ncores) -> m x K t-scores   src/multLinReg.cpp:66-86 */
SEXP _bigsnpr_multLinReg(SEXP obj, SEXP ind_row, SEXP ind_col, SEXP U, SEXP ncores) {
  R_xlen_t n = XLENGTH(ind_row), m = XLENGTH(ind_col);
  assert_size(Rf_nrows(U), n);                                   /* src/multLinReg.cpp:14 */
  int K = Rf_ncols(U);
  SEXP res = PROTECT(Rf_allocMatrix(REALSXP, (int) m, K));
  CHECK(bsn_mult_lin_reg(get_image(obj), ind0(ind_row), n, ind0(ind_col), m, REAL(U), K, REAL(res)));
  nan_to_na(REAL(res), (R_xlen_t) m * K);
  UNPROTECT(1);
  return res;
}

/* _bigsnpr_prod_and_rowSumsSq2(BM, ind_row, ind_col, center, scale, V) -> list(XV, rowSumsSq) (unnamed,
 * src/project-utils.cpp:42) */
SEXP _bigsnpr_prod_and_rowSumsSq2(SEXP BM, SEXP ind_row, SEXP ind_col, SEXP center, SEXP scale, SEXP V) {
  R_xlen_t n = XLENGTH(ind_row), m = XLENGTH(ind_col);
  assert_size(m, Rf_nrows(V)); assert_size(m, XLENGTH(center)); assert_size(m, XLENGTH(scale));
  int K = Rf_ncols(V);
  SEXP XV = PROTECT(Rf_allocMatrix(REALSXP, (int) n, K)), rs = PROTECT(Rf_allocVector(REALSXP, n));
  CHECK(bsn_snp_prod_and_rowsumssq2(get_image(BM), ind0(ind_row), n, ind0(ind_col), m, REAL(center),
                                    REAL(scale), REAL(V), K, REAL(XV), REAL(rs)));
  nan_to_na(REAL(XV), (R_xlen_t) n * K); nan_to_na(REAL(rs), n);
  SEXP res = PROTECT(Rf_allocVector(VECSXP, 2));
  SET_VECTOR_ELT(res, 0, XV); SET_VECTOR_ELT(res, 1, rs);
  UNPROTECT(3);
  return res;
}

/* _bigsnpr_readbina(filename, BM, tab) -> logical: the whole .bed file decoded through `tab` (4 x 256 raw,
 * getCode() of R/utils.R:21-31) into the new FBM's backing file; TRUE when the file ends with the last variant
 * (src/read-plink.cpp:13-56; snp_readBed warns otherwise, R/read-plink.R:54-55).  Like the reference this does NOT
 * go through the `bed` class (no "n or p does not match" check: a longer file is the warning case): the payload is
 * mapped and handed to bsn_bed_from_host.  A file SHORTER than n x m genotypes is an error here (the reference reads
 * past the end silently and leaves stale buffer bytes in the FBM). */
SEXP _bigsnpr_readbina(SEXP filename, SEXP BM, SEXP tab) {
  if (TYPEOF(tab) != RAWSXP || XLENGTH(tab) != 1024) Rf_error("readbina: 'tab' must be a 4 x 256 raw matrix");
  const char *path = CHAR(STRING_ELT(filename, 0));
  const int64_t n = (int64_t) Rf_asInteger(field(BM, "nrow")), m = (int64_t) Rf_asInteger(field(BM, "ncol"));
  const int64_t n_byte = (n + 3) / 4;
  int fd = open(path, O_RDONLY);
  if (fd < 0) Rf_error("cannot open '%s'", path);
  struct stat st;
  if (fstat(fd, &st) != 0) { close(fd); Rf_error("cannot stat '%s'", path); }
  if ((int64_t) st.st_size < 3 + m * n_byte) {
    close(fd);
    Rf_error("readbina: '%s' holds fewer than %lld x %lld genotypes", path, (long long) n, (long long) m);
  }
  const unsigned char *map = (const unsigned char *) mmap(NULL, (size_t) st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (map == MAP_FAILED) Rf_error("cannot map '%s'", path);
  if (!(map[0] == 108 && map[1] == 27)) {          /* src/read-plink.cpp:31-32 (its third test is an assignment) */
    munmap((void *) map, (size_t) st.st_size);
    Rf_error("Wrong magic number. Aborting..");
  }
  uint8_t *bytes = (uint8_t *) R_alloc((size_t) n * (size_t) m, 1);
  bsn_bed *b = NULL;
  int rc = bsn_bed_from_host(map + 3, n, m, n_byte, &b);
  if (rc == 0) rc = bsn_bed_readbina(b, RAW(tab), bytes);
  char msg[512] = "";
  if (rc != 0) snprintf(msg, sizeof msg, "%s", bsn_last_error());
  if (b) bsn_bed_close(b);
  munmap((void *) map, (size_t) st.st_size);
  if (rc != 0) Rf_error("%s", msg);
  write_backing(BM, bytes, (size_t) n * (size_t) m);
  return Rf_ScalarLogical((int64_t) st.st_size <= 3 + m * n_byte);
}

/* _bigsnpr_readbina2(BM, obj_bed, ind_row, ind_col, ncores): decoded genotypes of the .bed sub-matrix
 * into the (new) FBM's backing file, one byte each (src/read-plink.cpp:61-80) */
SEXP _bigsnpr_readbina2(SEXP BM, SEXP obj_bed, SEXP ind_row, SEXP ind_col, SEXP ncores) {
  R_xlen_t n = XLENGTH(ind_row), m = XLENGTH(ind_col);
  uint8_t *bytes = (uint8_t *) R_alloc((size_t) n * (size_t) m, 1);
  CHECK(bsn_bed_to_fbm(get_bed(obj_bed), ind0(ind_row), n, ind0(ind_col), m, bytes));
  write_backing(BM, bytes, (size_t) n * (size_t) m);
  return R_NilValue;
}

/* _bigsnpr_writebina(filename, BM, tab, rowInd, colInd): the FBM sub-matrix as a .bed file
 * (src/write-plink.cpp:13-52).  `tab` (the byte of every 4-genotype combination, R/write-plink.R:33)
 * is the inverse of the decode table, which is what the device packer applies. */
SEXP _bigsnpr_writebina(SEXP filename, SEXP BM, SEXP tab, SEXP rowInd, SEXP colInd) {
  R_xlen_t n = XLENGTH(rowInd), m = XLENGTH(colInd);
  size_t n_byte = ((size_t) n + 3) / 4;
  uint8_t *payload = (uint8_t *) R_alloc(n_byte * (size_t) m, 1);
  /* BM is G.round: code256 = replace(round(code), is.na(code), 3) (R/write-plink.R:35) — every entry in
   * {0, 1, 2, 3} and 3 MEANS missing (`tab` maps it to the .bed code 01).  The library's table convention
   * writes a missing value as NaN, so the table handed on is G.round's with 3 -> NaN: a 2-bit image. */
  bsn_bed *img;
  if (has_field(BM, "code256")) {
    const char *bk; int64_t fn, fm;
    const double *code256 = fbm_fields(BM, &bk, &fn, &fm);
    double *code = (double *) R_alloc(256, sizeof(double));
    for (int c = 0; c < 256; c++) {
      if (!(code256[c] == 0 || code256[c] == 1 || code256[c] == 2 || code256[c] == 3))
        Rf_error("snp_writeBed: the rounded code must hold 0, 1, 2 or 3 (missing) only");
      code[c] = code256[c] == 3 ? NA_REAL : code256[c];
    }
    img = image_of(bk, fn, fm, code);
  } else {
    img = get_bed(BM);
  }
  CHECK(bsn_bed_subset_payload(img, ind0(rowInd), n, ind0(colInd), m, payload));
  FILE *f = fopen(CHAR(STRING_ELT(filename, 0)), "wb");
  if (!f) Rf_error("cannot open '%s' for writing", CHAR(STRING_ELT(filename, 0)));
  const unsigned char magic[3] = {108, 27, 1};
  int ok = fwrite(magic, 1, 3, f) == 3 && fwrite(payload, 1, n_byte * (size_t) m, f) == n_byte * (size_t) m;
  ok = (fclose(f) == 0) && ok;
  if (!ok) Rf_error("error while writing '%s'", CHAR(STRING_ELT(filename, 0)));
  return R_NilValue;
}

/* FBM mat-vec operators (additions): what bigstatsr::big_prodVec / big_cprodVec compute for an
 * FBM.code256 (callers R/PRS.R:5, R/autoSVD.R:129-134), usable as the fun.prod / fun.cprod closures of
 * big_randomSVD:   fun.prod  = function(X, x, ind.row, ind.col, center, scale)
 *                                .Call(`_bigsnpr_big_prodVec_hip`, X, x, ind.row, ind.col, center, scale) */
SEXP _bigsnpr_big_prodVec_hip(SEXP X, SEXP y_col, SEXP ind_row, SEXP ind_col, SEXP center, SEXP scale) {
  R_xlen_t n = XLENGTH(ind_row), m = XLENGTH(ind_col);
  assert_size(XLENGTH(y_col), m);
  SEXP res = PROTECT(Rf_allocVector(REALSXP, n));
  CHECK(bsn_bed_prodvec(get_image(X), ind0(ind_row), n, ind0(ind_col), m,
                        Rf_isNull(center) ? NULL : REAL(center), Rf_isNull(scale) ? NULL : REAL(scale),
                        REAL(y_col), REAL(res)));
  UNPROTECT(1);
  return res;
}
SEXP _bigsnpr_big_cprodVec_hip(SEXP X, SEXP y_row, SEXP ind_row, SEXP ind_col, SEXP center, SEXP scale) {
  R_xlen_t n = XLENGTH(ind_row), m = XLENGTH(ind_col);
  assert_size(XLENGTH(y_row), n);
  SEXP res = PROTECT(Rf_allocVector(REALSXP, m));
  CHECK(bsn_bed_cprodvec(get_image(X), ind0(ind_row), n, ind0(ind_col), m,
                         Rf_isNull(center) ? NULL : REAL(center), Rf_isNull(scale) ? NULL : REAL(scale),
                         REAL(y_row), REAL(res)));
  UNPROTECT(1);
  return res;
}
/* big_randomSVD(G, fun.scaling, ind.row, ind.col, k, tol) of an FBM.code256, whole, on the device */
SEXP _bigsnpr_big_randomSVD_hip(SEXP X, SEXP ind_row, SEXP ind_col, SEXP center, SEXP scale,
                                SEXP k_, SEXP tol, SEXP verbose) {
  return random_svd(get_image(X), ind_row, ind_col, center, scale, k_, tol, verbose);
}

/* _bigsnpr_corMat(obj, rowInd, colInd, size, thr, pos, fill_diag, ncores) -> list(i, p, x)
 * (the R wrapper of R/corr.R:43-47 builds the dsCMatrix from it, unchanged) */
SEXP _bigsnpr_corMat(SEXP obj, SEXP rowInd, SEXP colInd, SEXP size, SEXP thr, SEXP pos,
                     SEXP fill_diag, SEXP ncores) {
  R_xlen_t n = XLENGTH(rowInd), m = XLENGTH(colInd);
  assert_size(XLENGTH(pos), m);                                  /* src/corr.cpp:111 */
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

/* `keep` of the reference is a 1 x m integer FBM (initialised to -1, R/clumping.R:116,
 * R/bed-clumping.R:53) that the native code writes in place through BM2$address_rw
 * (src/clumping.cpp:28-29).  Same effect here: the 0 / 1 results go into BM2's backing file through
 * a shared mapping, so `keep[]` in the two R callers reads them without any change to the R code. */
static void clump(SEXP obj, SEXP BM2, int mode, SEXP rowInd, SEXP colInd, SEXP ordInd, SEXP rankInd, SEXP pos,
                  SEXP aux1, SEXP aux2, SEXP size, SEXP thr) {
  R_xlen_t n = XLENGTH(rowInd), m = XLENGTH(colInd);
  assert_size(XLENGTH(pos), m); assert_size(XLENGTH(aux1), m); assert_size(XLENGTH(aux2), m);
  int32_t *keep = (int32_t *) R_alloc((size_t) m, sizeof(int32_t));
  CHECK(bsn_clumping_chr(get_image(obj), ind0(rowInd), n, ind0(colInd), m, mode, REAL(aux1), REAL(aux2),
                         ind0_32(ordInd), ind0_32(rankInd), REAL(pos), Rf_asReal(size), Rf_asReal(thr),
                         keep));
  /* keep is indexed by position in colInd in the reference as well (keep[j0], src/clumping.cpp:44) */
  write_backing(BM2, keep, (size_t) m * sizeof(int32_t));
}
/* _bigsnpr_clumping_chr(BM, BM2, rowInd, colInd, ordInd, rankInd, pos, sumX, denoX, size, thr, ncores) */
SEXP _bigsnpr_clumping_chr(SEXP BM, SEXP BM2, SEXP rowInd, SEXP colInd, SEXP ordInd, SEXP rankInd,
                           SEXP pos, SEXP sumX, SEXP denoX, SEXP size, SEXP thr, SEXP ncores) {
  clump(BM, BM2, 0, rowInd, colInd, ordInd, rankInd, pos, sumX, denoX, size, thr);
  return R_NilValue;
}
/* _bigsnpr_bed_clumping_chr(obj_bed, BM2, ind_row, ind_col, center, scale, ordInd, rankInd, pos, size, thr, ncores) */
SEXP _bigsnpr_bed_clumping_chr(SEXP obj_bed, SEXP BM2, SEXP ind_row, SEXP ind_col, SEXP center, SEXP scale,
                               SEXP ordInd, SEXP rankInd, SEXP pos, SEXP size, SEXP thr, SEXP ncores) {
  clump(obj_bed, BM2, 1, ind_row, ind_col, ordInd, rankInd, pos, center, scale, size, thr);
  return R_NilValue;
}
/* _bigsnpr_clumping_chr_cached(BM, BM2, sqcor, spInd, rowInd, colInd, ordInd, rankInd, pos, sumX, denoX,
 *                              size, thr, ncores)   src/clumping-cached.cpp:11-107
 * The sparse r2 cache `sqcor` only saves the reference recomputation between the grid points of
 * snp_grid_clumping (R/SCT.R:100-131); correlations cost nothing to recompute here, so the call is the
 * plain clumping and the cache is handed back untouched — the R loop runs unchanged.  (One
 * bsn_clumping_chr_cached call per chromosome with all grid points is the faster route, INTEGRATION.md.) */
SEXP _bigsnpr_clumping_chr_cached(SEXP BM, SEXP BM2, SEXP sqcor, SEXP spInd, SEXP rowInd, SEXP colInd,
                                  SEXP ordInd, SEXP rankInd, SEXP pos, SEXP sumX, SEXP denoX, SEXP size,
                                  SEXP thr, SEXP ncores) {
  assert_size(XLENGTH(spInd), XLENGTH(colInd));                  /* src/clumping-cached.cpp:37 */
  clump(BM, BM2, 0, rowInd, colInd, ordInd, rankInd, pos, sumX, denoX, size, thr);
  return sqcor;
}

static const R_CallMethodDef CallEntries[] = {
  {"_bigsnpr_bedXPtr", (DL_FUNC) &_bigsnpr_bedXPtr, 3},
  {"_bigsnpr_bed_colstats", (DL_FUNC) &_bigsnpr_bed_colstats, 4},
  {"_bigsnpr_bed_col_counts_cpp", (DL_FUNC) &_bigsnpr_bed_col_counts_cpp, 4},
  {"_bigsnpr_bed_row_counts_cpp", (DL_FUNC) &_bigsnpr_bed_row_counts_cpp, 4},
  {"_bigsnpr_read_bed", (DL_FUNC) &_bigsnpr_read_bed, 3},
  {"_bigsnpr_read_bed_scaled", (DL_FUNC) &_bigsnpr_read_bed_scaled, 5},
  {"_bigsnpr_bed_pMatVec4", (DL_FUNC) &_bigsnpr_bed_pMatVec4, 7},
  {"_bigsnpr_bed_cpMatVec4", (DL_FUNC) &_bigsnpr_bed_cpMatVec4, 7},
  {"_bigsnpr_prod_and_rowSumsSq", (DL_FUNC) &_bigsnpr_prod_and_rowSumsSq, 6},
  {"_bigsnpr_prod_and_rowSumsSq2", (DL_FUNC) &_bigsnpr_prod_and_rowSumsSq2, 6},
  {"_bigsnpr_snp_colstats", (DL_FUNC) &_bigsnpr_snp_colstats, 4},
  {"_bigsnpr_multLinReg", (DL_FUNC) &_bigsnpr_multLinReg, 5},
  {"_bigsnpr_readbina", (DL_FUNC) &_bigsnpr_readbina, 3},
  {"_bigsnpr_readbina2", (DL_FUNC) &_bigsnpr_readbina2, 5},
  {"_bigsnpr_writebina", (DL_FUNC) &_bigsnpr_writebina, 5},
  {"_bigsnpr_corMat", (DL_FUNC) &_bigsnpr_corMat, 8},
  {"_bigsnpr_ld_scores", (DL_FUNC) &_bigsnpr_ld_scores, 6},
  {"_bigsnpr_clumping_chr", (DL_FUNC) &_bigsnpr_clumping_chr, 12},
  {"_bigsnpr_bed_clumping_chr", (DL_FUNC) &_bigsnpr_bed_clumping_chr, 12},
  {"_bigsnpr_clumping_chr_cached", (DL_FUNC) &_bigsnpr_clumping_chr_cached, 14},
  /* additions */
  {"_bigsnpr_bed_randomSVD_hip", (DL_FUNC) &_bigsnpr_bed_randomSVD_hip, 8},
  {"_bigsnpr_big_randomSVD_hip", (DL_FUNC) &_bigsnpr_big_randomSVD_hip, 8},
  {"_bigsnpr_big_prodVec_hip", (DL_FUNC) &_bigsnpr_big_prodVec_hip, 6},
  {"_bigsnpr_big_cprodVec_hip", (DL_FUNC) &_bigsnpr_big_cprodVec_hip, 6},
  {"_bigsnpr_fbm_cache_clear_hip", (DL_FUNC) &_bigsnpr_fbm_cache_clear_hip, 0},
  {NULL, NULL, 0}
};

void R_init_bigsnprhip(DllInfo *dll) {
  R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
