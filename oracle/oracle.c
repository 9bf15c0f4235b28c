/* ORACLE -- test infrastructure only.  See oracle_core.h for what this restates and what
 * is (un)pinned.  Build: oracle/Makefile -> oracle/_build/liboracle.so */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include "oracle.h"

#define REAL double
#define ORC_IS_F32 0
#define FN(x) x##_f64
#include "oracle_core.h"
#undef REAL
#undef ORC_IS_F32
#undef FN

#define REAL float
#define ORC_IS_F32 1
#define FN(x) x##_f32
#include "oracle_core.h"
#undef REAL
#undef ORC_IS_F32
#undef FN

int orc_sizeof_params(void) { return (int)sizeof(orc_params); }
