/* TEST INFRASTRUCTURE — see ../Rinternals.h: the registration interface of "Writing R Extensions" 5.4. */
#ifndef BSN_TEST_RDYNLOAD_H
#define BSN_TEST_RDYNLOAD_H
#include "../Rinternals.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef void *(*DL_FUNC)(void);
typedef struct {
  const char *name;
  DL_FUNC fun;
  int numArgs;
} R_CallMethodDef;
typedef struct _DllInfo DllInfo;
int R_registerRoutines(DllInfo *info, const void *cRoutines, const R_CallMethodDef *callRoutines,
                       const void *fortranRoutines, const void *externalRoutines);
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean value);
#ifdef __cplusplus
}
#endif
#endif
