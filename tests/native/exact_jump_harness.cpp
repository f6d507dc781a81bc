// CPU harness for abr_exact_jump.h: runs the jump chain and the naive loop on
// arrays of cases and reports the first mismatch.  Built by tests/test_exact_jump.py.
#include <stdint.h>
#include "abr_exact_jump.h"

template <int STOP>
static bool naive(double &x, double c, double thr, int32_t n, int32_t &a) {
    a = 0;
    while (a < n) {
        x = x + c;
        a++;
        if (abrx::stop_hit<STOP>(x, thr)) return true;
    }
    return false;
}

template <int STOP>
static int64_t run(const double *x0, const double *c, const double *thr, const int32_t *n,
                   int64_t cases, double *x_out, int32_t *a_out, uint8_t *hit_out) {
    for (int64_t i = 0; i < cases; i++) {
        double xj = x0[i], xn = x0[i];
        int32_t aj = 0, an = 0;
        bool hj = abrx::chain<STOP>(xj, c[i], thr[i], n[i], aj);
        bool hn = naive<STOP>(xn, c[i], thr[i], n[i], an);
        x_out[i] = xj; a_out[i] = aj; hit_out[i] = hj;
        uint64_t bj, bn;
        memcpy(&bj, &xj, 8); memcpy(&bn, &xn, 8);
        if (bj != bn || aj != an || hj != hn) return i;
    }
    return -1;
}

extern "C" {
int64_t fuzz_ge(const double *x0, const double *c, const double *thr, const int32_t *n,
                int64_t cases, double *x_out, int32_t *a_out, uint8_t *hit_out) {
    return run<abrx::STOP_GE>(x0, c, thr, n, cases, x_out, a_out, hit_out);
}
int64_t fuzz_le(const double *x0, const double *c, const double *thr, const int32_t *n,
                int64_t cases, double *x_out, int32_t *a_out, uint8_t *hit_out) {
    return run<abrx::STOP_LE>(x0, c, thr, n, cases, x_out, a_out, hit_out);
}
int64_t fuzz_lt(const double *x0, const double *c, const double *thr, const int32_t *n,
                int64_t cases, double *x_out, int32_t *a_out, uint8_t *hit_out) {
    return run<abrx::STOP_LT>(x0, c, thr, n, cases, x_out, a_out, hit_out);
}
}
