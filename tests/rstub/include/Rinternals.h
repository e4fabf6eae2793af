/* TEST INFRASTRUCTURE — declarations of the part of R's C API that bindings/R/bigsnpr_hip_shim.c uses,
 * written from R's documented interface ("Writing R Extensions", section 5) so that the shim can be
 * compiled with -Wall -Wextra -Werror and run against a small stand-in runtime (tests/rstub/rstub.c) on
 * a machine without R.  Not part of the product and not R: only what the shim needs, nothing else. */
#ifndef BSN_TEST_RINTERNALS_H
#define BSN_TEST_RINTERNALS_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef enum { FALSE = 0, TRUE } Rboolean;

#define NILSXP 0
#define SYMSXP 1
#define ENVSXP 4
#define LANGSXP 6
#define CHARSXP 9
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
#define EXTPTRSXP 22
#define RAWSXP 24

extern SEXP R_NilValue, R_GlobalEnv, R_UnboundValue;
extern double R_NaReal;
extern int R_NaInt;
#define NA_REAL R_NaReal
#define NA_INTEGER R_NaInt
#define NA_LOGICAL R_NaInt
int R_IsNaN_or_NA(double x);
#define ISNAN(x) R_IsNaN_or_NA(x)

int TYPEOF(SEXP x);
R_xlen_t XLENGTH(SEXP x);
int *INTEGER(SEXP x);
double *REAL(SEXP x);
typedef unsigned char Rbyte;
Rbyte *RAW(SEXP x);
const char *CHAR(SEXP x);
SEXP STRING_ELT(SEXP x, R_xlen_t i);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);

SEXP Rf_protect(SEXP x);
void Rf_unprotect(int n);
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)

SEXP Rf_allocVector(unsigned int type, R_xlen_t n);
SEXP Rf_allocMatrix(unsigned int type, int nrow, int ncol);
SEXP Rf_mkNamed(unsigned int type, const char **names);
SEXP Rf_ScalarInteger(int x);
SEXP Rf_ScalarLogical(int x);
int Rf_asInteger(SEXP x);
double Rf_asReal(SEXP x);
int Rf_asLogical(SEXP x);
Rboolean Rf_isNull(SEXP x);
int Rf_nrows(SEXP x);
int Rf_ncols(SEXP x);

SEXP Rf_install(const char *name);
SEXP Rf_lang3(SEXP a, SEXP b, SEXP c);
SEXP Rf_eval(SEXP call, SEXP env);
SEXP Rf_findVarInFrame3(SEXP env, SEXP sym, Rboolean doget);

SEXP R_MakeExternalPtr(void *p, SEXP tag, SEXP prot);
void *R_ExternalPtrAddr(SEXP x);
void R_ClearExternalPtr(SEXP x);
typedef void (*R_CFinalizer_t)(SEXP);
void R_RegisterCFinalizerEx(SEXP x, R_CFinalizer_t fin, Rboolean onexit);

char *R_alloc(size_t n, int size);

#if defined(__GNUC__)
#define BSN_NORETURN __attribute__((noreturn))
#define BSN_PRINTF(a, b) __attribute__((format(printf, a, b)))
#else
#define BSN_NORETURN
#define BSN_PRINTF(a, b)
#endif
void Rf_error(const char *fmt, ...) BSN_NORETURN BSN_PRINTF(1, 2);
void Rf_warning(const char *fmt, ...) BSN_PRINTF(1, 2);

#ifdef __cplusplus
}
#endif
#endif
