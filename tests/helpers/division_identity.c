/* Property check behind divide_by() of csrc/rtk_trace.hip: with y = RN(1/a), q0 = RN(n*y), r = fma(-a, q0, n) and
 * q1 = fma(r, y, q0), q1 equals the correctly rounded quotient n/a (Markstein's division step).  The device's
 * v_fma_f64 / v_mul_f64 are IEEE operations, so the identity checked here on the host FMA unit is the one the kernel
 * relies on.  Usage: division_identity [cases]   -> prints "mismatches: K". */
#include <stdlib.h>
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
static uint64_t s = 88172645463325252ull;
static inline uint64_t rnd64(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline double rnd_double(int emin, int emax) {  // random mantissa, exponent in range, random sign
    uint64_t m = rnd64() >> 12;
    int e = emin + (int)(rnd64() % (uint64_t)(emax - emin + 1));
    uint64_t bits = ((uint64_t)(e + 1023) << 52) | m | ((rnd64() & 1) << 63);
    double d; memcpy(&d, &bits, 8); return d;
}
int main(int argc, char** argv) {
    long bad = 0, bad0 = 0; const long N = argc > 1 ? atol(argv[1]) : 400000000L;
    for (long i = 0; i < N; i++) {
        double a = fabs(rnd_double(-20, 20)), n = rnd_double(-30, 30);
        if (i % 16 == 0) { uint64_t b; memcpy(&b, &a, 8); b |= 0xFFFFFFFFFFFFFull >> (rnd64() % 8); memcpy(&a, &b, 8); }  // near all-ones significands
        double y = 1.0 / a;
        double q0 = n * y;
        double r = fma(-a, q0, n);
        double q1 = fma(r, y, q0);
        double q = n / a;
        if (q0 != q) bad0++;
        if (q1 != q) { bad++; if (bad < 10) printf("MISMATCH n=%a a=%a q=%a q1=%a\n", n, a, q, q1); }
    }
    printf("cases: %ld  uncorrected n*y differs: %ld  mismatches: %ld\n", N, bad0, bad);
    return bad != 0;
}
