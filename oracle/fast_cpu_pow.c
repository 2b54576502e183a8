/* TEST ORACLE — NOT PRODUCT CODE (part of oracle/fast_cpu.c's "good CPU" timing point; see there).
 * Own translation unit because glibc only declares its vector math variants under -ffast-math. */
#include <math.h>

void fc_abs_pow_row(const double* restrict s, double* restrict d, long n, double expo) {
#pragma omp simd
    for (long c = 0; c < n; ++c) d[c] = pow(fabs(s[c]), expo);
}
