/*
 * TEST INFRASTRUCTURE ONLY.  Builds the CPU oracle for the GRU forward + BPTT hot path in
 * fp32 (arithmetic type of the reference, outdated_files/TimeGANv1.py:590,667,755 `x.float()`)
 * and in fp64 (tie-breaker for tolerance decisions).  See gru_oracle_impl.h for the
 * restated algorithm and the reference lines it follows.
 *
 * Build: make -C oracle   ->  oracle/libgru_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define REAL float
#define SUFFIX _f32
#define EXPF expf
#define TANHF tanhf
#include "gru_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef EXPF
#undef TANHF

#define REAL double
#define SUFFIX _f64
#define EXPF exp
#define TANHF tanh
#include "gru_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef EXPF
#undef TANHF

/* number of parameter scalars of an L-layer stack, in the flat layout the oracle uses */
int64_t gru_oracle_param_count(int In, int H, int L) {
    int64_t n = 0;
    for (int l = 0; l < L; ++l) {
        const int64_t in_l = (l == 0) ? In : H;
        n += 3LL * H * in_l + 3LL * H * H + 6LL * H;
    }
    return n;
}
