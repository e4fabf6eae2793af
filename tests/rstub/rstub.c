/* TEST INFRASTRUCTURE — a stand-in for the part of the R runtime that bindings/R/bigsnprhip/src/bigsnpr_hip_shim.c calls,
 * so that the shim's `.Call` entry points can be RUN on a machine without R: vectors, matrices, named
 * lists, environments with fields (the RC objects `bed` and `FBM.code256` as the shim sees them: obj$field),
 * external pointers with finalizers, R_alloc, and Rf_error as a non-local exit back to the caller of
 * rstub_call().  Written from the documented behaviour of those functions ("Writing R Extensions", ch. 5);
 * no garbage collector (objects live until rstub_reset), no evaluation beyond `obj$name`.
 * Not part of the product. */
#include <math.h>
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "include/R_ext/Rdynload.h"

struct envent {
  char *name;
  SEXP value;
  struct envent *next;
};
struct SEXPREC {
  int type;
  R_xlen_t len;
  void *data;          /* int / double / SEXP elements, char bytes, extptr address, symbol name */
  int nrow, ncol;      /* > 0 for matrices */
  SEXP names;          /* STRSXP for named lists */
  struct envent *vars; /* ENVSXP */
  R_CFinalizer_t fin;  /* EXTPTRSXP */
  SEXP lang[3];        /* LANGSXP */
  struct SEXPREC *all_next;
};

static struct SEXPREC nil_rec = {NILSXP, 0, NULL, 0, 0, NULL, NULL, NULL, {NULL, NULL, NULL}, NULL};
static struct SEXPREC glob_rec = {ENVSXP, 0, NULL, 0, 0, NULL, NULL, NULL, {NULL, NULL, NULL}, NULL};
static struct SEXPREC unbound_rec = {SYMSXP, 0, NULL, 0, 0, NULL, NULL, NULL, {NULL, NULL, NULL}, NULL};
SEXP R_NilValue = &nil_rec, R_GlobalEnv = &glob_rec, R_UnboundValue = &unbound_rec;
double R_NaReal;
int R_NaInt = INT32_MIN;

static SEXP g_all = NULL;           /* every object allocated since the last reset */
static int g_protect = 0, g_max_protect = 0;
static char g_error[1024], g_warnings[4096];
static jmp_buf g_jmp;
static int g_in_call = 0;
struct ralloc {
  struct ralloc *next;
};
static struct ralloc *g_ralloc = NULL;

static void __attribute__((constructor)) init_na(void) {
  /* R's NA_real_: a quiet NaN whose low word is 1954 */
  uint64_t bits = 0x7FF00000000007A2ull;
  memcpy(&R_NaReal, &bits, 8);
}

static SEXP new_obj(int type, R_xlen_t len, size_t elt) {
  SEXP x = (SEXP) calloc(1, sizeof(struct SEXPREC));
  x->type = type;
  x->len = len;
  x->data = (len > 0 && elt > 0) ? calloc((size_t) len, elt) : NULL;
  x->names = NULL;
  x->all_next = g_all;
  g_all = x;
  return x;
}

int R_IsNaN_or_NA(double x) { return isnan(x); }
int TYPEOF(SEXP x) { return x->type; }
R_xlen_t XLENGTH(SEXP x) { return x->len; }
static void need(SEXP x, int type, const char *what) {
  if (x->type != type) Rf_error("rstub: %s applied to an object of type %d", what, x->type);
}
int *INTEGER(SEXP x) {
  if (x->type != INTSXP && x->type != LGLSXP) Rf_error("rstub: INTEGER() applied to an object of type %d", x->type);
  return (int *) x->data;
}
double *REAL(SEXP x) { need(x, REALSXP, "REAL()"); return (double *) x->data; }
Rbyte *RAW(SEXP x) { need(x, RAWSXP, "RAW()"); return (Rbyte *) x->data; }
const char *CHAR(SEXP x) { need(x, CHARSXP, "CHAR()"); return (const char *) x->data; }
SEXP STRING_ELT(SEXP x, R_xlen_t i) {
  need(x, STRSXP, "STRING_ELT()");
  if (i < 0 || i >= x->len) Rf_error("rstub: STRING_ELT index out of range");
  return ((SEXP *) x->data)[i];
}
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v) {
  need(x, VECSXP, "SET_VECTOR_ELT()");
  if (i < 0 || i >= x->len) Rf_error("rstub: SET_VECTOR_ELT index out of range");
  ((SEXP *) x->data)[i] = v;
  return v;
}
SEXP Rf_protect(SEXP x) {
  g_protect++;
  if (g_protect > g_max_protect) g_max_protect = g_protect;
  return x;
}
void Rf_unprotect(int n) {
  g_protect -= n;
  if (g_protect < 0) {
    g_protect = 0;
    Rf_error("rstub: unprotect(): only %d protected items", g_protect + n);
  }
}
static SEXP mkchar(const char *s) {
  SEXP c = new_obj(CHARSXP, (R_xlen_t) strlen(s), 0);
  c->data = strdup(s);
  return c;
}
SEXP Rf_allocVector(unsigned int type, R_xlen_t n) {
  switch (type) {
    case INTSXP: case LGLSXP: return new_obj((int) type, n, sizeof(int));
    case REALSXP: return new_obj(REALSXP, n, sizeof(double));
    case RAWSXP: return new_obj(RAWSXP, n, 1);
    case VECSXP: case STRSXP: {
      SEXP x = new_obj((int) type, n, sizeof(SEXP));
      for (R_xlen_t i = 0; i < n; i++) ((SEXP *) x->data)[i] = type == STRSXP ? mkchar("") : R_NilValue;
      return x;
    }
    default: Rf_error("rstub: allocVector of type %u", type);
  }
}
SEXP Rf_allocMatrix(unsigned int type, int nrow, int ncol) {
  SEXP x = Rf_allocVector(type, (R_xlen_t) nrow * ncol);
  x->nrow = nrow;
  x->ncol = ncol;
  return x;
}
SEXP Rf_mkNamed(unsigned int type, const char **names) {
  R_xlen_t n = 0;
  while (names[n][0] != '\0') n++;
  SEXP x = Rf_allocVector(type, n), nm = Rf_allocVector(STRSXP, n);
  for (R_xlen_t i = 0; i < n; i++) ((SEXP *) nm->data)[i] = mkchar(names[i]);
  x->names = nm;
  return x;
}
SEXP Rf_ScalarInteger(int v) {
  SEXP x = Rf_allocVector(INTSXP, 1);
  INTEGER(x)[0] = v;
  return x;
}
int Rf_asInteger(SEXP x) {
  if (x->len < 1) return NA_INTEGER;
  if (x->type == INTSXP || x->type == LGLSXP) return ((int *) x->data)[0];
  if (x->type == REALSXP) return isnan(((double *) x->data)[0]) ? NA_INTEGER : (int) ((double *) x->data)[0];
  return NA_INTEGER;
}
double Rf_asReal(SEXP x) {
  if (x->len < 1) return NA_REAL;
  if (x->type == REALSXP) return ((double *) x->data)[0];
  if (x->type == INTSXP || x->type == LGLSXP)
    return ((int *) x->data)[0] == NA_INTEGER ? NA_REAL : (double) ((int *) x->data)[0];
  return NA_REAL;
}
int Rf_asLogical(SEXP x) {
  if (x->len < 1) return NA_LOGICAL;
  if (x->type == LGLSXP || x->type == INTSXP) return ((int *) x->data)[0] == NA_INTEGER ? NA_LOGICAL : ((int *) x->data)[0] != 0;
  if (x->type == REALSXP) return isnan(((double *) x->data)[0]) ? NA_LOGICAL : ((double *) x->data)[0] != 0;
  return NA_LOGICAL;
}
Rboolean Rf_isNull(SEXP x) { return x->type == NILSXP ? TRUE : FALSE; }
int Rf_nrows(SEXP x) { return x->nrow > 0 ? x->nrow : (int) x->len; }
int Rf_ncols(SEXP x) { return x->nrow > 0 ? x->ncol : 1; }

SEXP Rf_install(const char *name) {
  SEXP s = new_obj(SYMSXP, 0, 0);
  s->data = strdup(name);
  return s;
}
SEXP Rf_lang3(SEXP a, SEXP b, SEXP c) {
  SEXP l = new_obj(LANGSXP, 3, 0);
  l->lang[0] = a; l->lang[1] = b; l->lang[2] = c;
  return l;
}
SEXP Rf_findVarInFrame3(SEXP env, SEXP sym, Rboolean doget) {
  (void) doget;
  need(env, ENVSXP, "findVarInFrame3()");
  for (struct envent *e = env->vars; e; e = e->next)
    if (strcmp(e->name, (const char *) sym->data) == 0) return e->value;
  return R_UnboundValue;
}
/* only `obj$name` */
SEXP Rf_eval(SEXP call, SEXP env) {
  (void) env;
  if (call->type != LANGSXP || call->lang[0]->type != SYMSXP || strcmp((const char *) call->lang[0]->data, "$") != 0)
    Rf_error("rstub: eval() only knows `obj$name`");
  SEXP v = Rf_findVarInFrame3(call->lang[1], call->lang[2], TRUE);
  if (v == R_UnboundValue) return R_NilValue;   /* `$` on an environment: NULL for a missing name */
  return v;
}
SEXP R_MakeExternalPtr(void *p, SEXP tag, SEXP prot) {
  (void) tag; (void) prot;
  SEXP x = new_obj(EXTPTRSXP, 0, 0);
  x->data = p;
  return x;
}
void *R_ExternalPtrAddr(SEXP x) { need(x, EXTPTRSXP, "R_ExternalPtrAddr()"); return x->data; }
void R_ClearExternalPtr(SEXP x) { x->data = NULL; }
void R_RegisterCFinalizerEx(SEXP x, R_CFinalizer_t fin, Rboolean onexit) { (void) onexit; x->fin = fin; }

char *R_alloc(size_t n, int size) {
  struct ralloc *r = (struct ralloc *) malloc(sizeof(struct ralloc) + n * (size_t) size + 16);
  r->next = g_ralloc;
  g_ralloc = r;
  return (char *) (r + 1);
}
static void free_ralloc(void) {
  while (g_ralloc) { struct ralloc *r = g_ralloc; g_ralloc = r->next; free(r); }
}
void Rf_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
  if (!g_in_call) { fprintf(stderr, "rstub: error outside rstub_call: %s\n", g_error); abort(); }
  longjmp(g_jmp, 1);
}
void Rf_warning(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  size_t used = strlen(g_warnings);
  snprintf(g_warnings + used, sizeof(g_warnings) - used, "%s\n", buf);
}

/* ---- registration ---------------------------------------------------------------------------------- */
static const R_CallMethodDef *g_table = NULL;
int R_registerRoutines(DllInfo *info, const void *c, const R_CallMethodDef *call, const void *f, const void *e) {
  (void) info; (void) c; (void) f; (void) e;
  g_table = call;
  return 1;
}
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean value) { (void) info; (void) value; return TRUE; }

/* ---- what the test harness (Python, ctypes) calls ---------------------------------------------------- */
int rstub_n_routines(void) {
  int n = 0;
  while (g_table && g_table[n].name) n++;
  return n;
}
const char *rstub_routine_name(int i) { return g_table[i].name; }
int rstub_routine_nargs(int i) { return g_table[i].numArgs; }

SEXP rstub_nil(void) { return R_NilValue; }
SEXP rstub_int(const int *v, R_xlen_t n) {
  SEXP x = Rf_allocVector(INTSXP, n);
  if (n) memcpy(x->data, v, (size_t) n * sizeof(int));
  return x;
}
SEXP Rf_ScalarLogical(int v) {
  SEXP x = Rf_allocVector(LGLSXP, 1);
  ((int *) x->data)[0] = v == NA_LOGICAL ? NA_LOGICAL : (v != 0);
  return x;
}
SEXP rstub_raw_matrix(const unsigned char *v, int nrow, int ncol) {
  SEXP x = Rf_allocMatrix(RAWSXP, nrow, ncol);
  if (x->len) memcpy(x->data, v, (size_t) x->len);
  return x;
}
SEXP rstub_lgl(int v) {
  SEXP x = Rf_allocVector(LGLSXP, 1);
  ((int *) x->data)[0] = v;
  return x;
}
SEXP rstub_real(const double *v, R_xlen_t n) {
  SEXP x = Rf_allocVector(REALSXP, n);
  if (n) memcpy(x->data, v, (size_t) n * sizeof(double));
  return x;
}
SEXP rstub_real_matrix(const double *v, int nrow, int ncol) {
  SEXP x = Rf_allocMatrix(REALSXP, nrow, ncol);
  if (nrow > 0 && ncol > 0) memcpy(x->data, v, (size_t) nrow * (size_t) ncol * sizeof(double));
  return x;
}
SEXP rstub_string(const char *s) {
  SEXP x = Rf_allocVector(STRSXP, 1);
  ((SEXP *) x->data)[0] = mkchar(s);
  return x;
}
SEXP rstub_env(void) { return new_obj(ENVSXP, 0, 0); }
void rstub_env_set(SEXP env, const char *name, SEXP value) {
  for (struct envent *e = env->vars; e; e = e->next)
    if (strcmp(e->name, name) == 0) { e->value = value; return; }
  struct envent *e = (struct envent *) malloc(sizeof(struct envent));
  e->name = strdup(name);
  e->value = value;
  e->next = env->vars;
  env->vars = e;
}
int rstub_type(SEXP x) { return x->type; }
R_xlen_t rstub_length(SEXP x) { return x->len; }
int rstub_nrow(SEXP x) { return x->nrow; }
int rstub_ncol(SEXP x) { return x->ncol; }
void *rstub_data(SEXP x) { return x->data; }
SEXP rstub_list_get(SEXP x, R_xlen_t i) { return ((SEXP *) x->data)[i]; }
const char *rstub_list_name(SEXP x, R_xlen_t i) {
  return x->names ? (const char *) ((SEXP *) x->names->data)[i]->data : "";
}
const char *rstub_last_error(void) { return g_error; }
const char *rstub_warnings(void) { return g_warnings; }
int rstub_protect_depth(void) { return g_protect; }

typedef SEXP (*fn0)(void);
typedef SEXP (*fn14)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
/* .Call(name, args...): looks the routine up in the registered table, checks the arity, runs it; an R error
 * inside comes back as NULL with the message in rstub_last_error() (R unwinds the protect stack likewise).
 * Returns NULL with "balance" in the error if the routine left the protect stack unbalanced. */
SEXP rstub_call(const char *name, int nargs, SEXP *args) {
  g_error[0] = '\0';
  g_warnings[0] = '\0';
  const R_CallMethodDef *volatile d = NULL;
  for (int i = 0; g_table && g_table[i].name; i++)
    if (strcmp(g_table[i].name, name) == 0) d = &g_table[i];
  if (!d) { snprintf(g_error, sizeof(g_error), "rstub: no routine '%s' is registered", name); return NULL; }
  if (d->numArgs != nargs) {
    snprintf(g_error, sizeof(g_error), "rstub: '%s' takes %d arguments, %d given", name, d->numArgs, nargs);
    return NULL;
  }
  SEXP a[14];
  for (int i = 0; i < 14; i++) a[i] = i < nargs ? args[i] : R_NilValue;
  const int depth0 = g_protect;
  SEXP volatile res = NULL;
  g_in_call = 1;
  if (setjmp(g_jmp) == 0) {
    /* extra trailing arguments are harmless in the C calling convention used here (caller cleans up) */
    res = nargs == 0 ? ((fn0) d->fun)() : ((fn14) d->fun)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13]);
    if (g_protect != depth0) {
      snprintf(g_error, sizeof(g_error), "rstub: protect stack balance %d after '%s'", g_protect - depth0, name);
      g_protect = depth0;
      res = NULL;
    }
  } else {
    g_protect = depth0;
    res = NULL;
  }
  g_in_call = 0;
  free_ralloc();
  return res;
}
/* runs the finalizers of the external pointers and frees every object */
void rstub_reset(void) {
  for (SEXP x = g_all; x; x = x->all_next)
    if (x->type == EXTPTRSXP && x->fin && x->data) x->fin(x);
  while (g_all) {
    SEXP x = g_all;
    g_all = x->all_next;
    if (x->type != EXTPTRSXP) free(x->data);
    while (x->vars) { struct envent *e = x->vars; x->vars = e->next; free(e->name); free(e); }
    free(x);
  }
}
