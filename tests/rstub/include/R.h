/* TEST INFRASTRUCTURE — see Rinternals.h in this directory. */
#ifndef BSN_TEST_R_H
#define BSN_TEST_R_H
#include <stdlib.h>
#include <string.h>
#endif
