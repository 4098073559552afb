/* CPU oracle (C restatement) of the IRBFN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * See irbfn_oracle_impl.h.  Built by oracle/Makefile into oracle/_build/libirbfn_oracle.so;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it. */
#include <math.h>
#include <stdlib.h>

#define REAL float
#define SUF f32
#define REAL_IS_FLOAT 1
#include "irbfn_oracle_impl.h"
#undef REAL
#undef SUF
#undef REAL_IS_FLOAT

#define REAL double
#define SUF f64
#define REAL_IS_FLOAT 0
#include "irbfn_oracle_impl.h"
#undef REAL
#undef SUF
#undef REAL_IS_FLOAT

#ifdef _OPENMP
#include <omp.h>
int oracle_num_threads(void) { return omp_get_max_threads(); }
void oracle_set_num_threads(int n) { omp_set_num_threads(n); }
#else
int oracle_num_threads(void) { return 1; }
void oracle_set_num_threads(int n) { (void)n; }
#endif
