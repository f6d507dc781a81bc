// CPU harness for abr_exact_jump.h: runs the jump chain and the naive loop on
// arrays of cases and reports the first mismatch.  Built by tests/test_exact_jump.py.
#include <stdint.h>
// how often the jump-length estimate overshot (candidate 0 outside): the repair path of chain_segment
static int64_t g_overshoot = 0;
#define ABR_BRACKET_HOOK(ok0) (g_overshoot += (ok0) ? 0 : 1)
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

template <int STOP, int BIAS = 0>
static int64_t run(const double *x0, const double *c, const double *thr, const int32_t *n,
                   int64_t cases, double *x_out, int32_t *a_out, uint8_t *hit_out) {
    for (int64_t i = 0; i < cases; i++) {
        double xj = x0[i], xn = x0[i];
        int32_t aj = 0, an = 0;
        bool hj = abrx::chain<STOP, BIAS>(xj, c[i], thr[i], n[i], aj);
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
// the same with the jump-length estimate spoiled by +4 / -4 (kind: 0 >=, 1 <=, 2 <)
int64_t fuzz_biased(int32_t kind, int32_t bias, const double *x0, const double *c, const double *thr,
                    const int32_t *n, int64_t cases, double *x_out, int32_t *a_out, uint8_t *hit_out) {
    if (bias != 4 && bias != -4) return -2;
    switch (kind) {
        case 0: return bias > 0 ? run<abrx::STOP_GE, 4>(x0, c, thr, n, cases, x_out, a_out, hit_out)
                                : run<abrx::STOP_GE, -4>(x0, c, thr, n, cases, x_out, a_out, hit_out);
        case 1: return bias > 0 ? run<abrx::STOP_LE, 4>(x0, c, thr, n, cases, x_out, a_out, hit_out)
                                : run<abrx::STOP_LE, -4>(x0, c, thr, n, cases, x_out, a_out, hit_out);
        case 2: return bias > 0 ? run<abrx::STOP_LT, 4>(x0, c, thr, n, cases, x_out, a_out, hit_out)
                                : run<abrx::STOP_LT, -4>(x0, c, thr, n, cases, x_out, a_out, hit_out);
    }
    return -2;
}
int64_t overshoot_count(int32_t reset) {
    const int64_t v = g_overshoot;
    if (reset) g_overshoot = 0;
    return v;
}
}
