// abr_exact_jump.h -- exact closed-form stepping of "add the same float64 constant n times".
//
// The reference advances downloaded_size and buffer_level by one float64
// addition per 0.01 s tick (Simulator.py:160, :184) and branches on the rounded
// results (:163, :190, :194), so a real-number closed form n*c is not allowed:
// one flipped comparison changes a step's outputs by a whole tick.  But the
// rounded sequence itself has a closed form inside one binade.
//
// Let x be in [2^e, 2^(e+1)) with unit u = 2^(e-52), so x = X*u for an integer X,
// and write the constant as c = q*u + r with 0 <= r < u.  If fl(x + c) stays in
// the binade, round-to-nearest-even gives
//     fl(x + c) = (X + q)*u        if r <  u/2
//               = (X + q + 1)*u    if r >  u/2
//               = the even one     if r == u/2,
// i.e. a constant integer increment D, except that in the tie case D depends on
// the parity of X + q -- and a tie always produces an even X, so from the second
// consecutive in-binade step on D is constant there too.  Hence:
//     once two consecutive additions have stayed inside one binade, every
//     further in-binade addition adds exactly d = (last result) - (previous one),
//     and x + m*d is computed exactly in float64 (all terms are multiples of u
//     below 2^53 * u).
// The same holds for a negative constant.  Steps that leave the binade, and the
// first two steps inside a new one, are executed as real additions.
//
// This header is plain C++ (host + device) so that tests/ can fuzz it on the CPU
// against the naive loop.  Compile with -ffp-contract=off.
#ifndef ABR_EXACT_JUMP_H
#define ABR_EXACT_JUMP_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define ABR_HD __host__ __device__ inline
#else
#define ABR_HD inline
#endif

namespace abrx {

ABR_HD int expo(double x) {
    uint64_t b;
#if defined(__HIP_DEVICE_COMPILE__)
    b = (uint64_t)__double_as_longlong(x);
#else
    memcpy(&b, &x, 8);
#endif
    return (int)((b >> 52) & 0x7ff);
}

// 2^(biased exponent e), e in 1..2046
ABR_HD double pow2_biased(int e) {
    uint64_t b = (uint64_t)e << 52;
    double r;
#if defined(__HIP_DEVICE_COMPILE__)
    r = __longlong_as_double((long long)b);
#else
    memcpy(&r, &b, 8);
#endif
    return r;
}

enum StopKind { STOP_GE = 0 /* result >= thr */, STOP_LE = 1 /* result <= thr */,
                STOP_LT = 2 /* result <  thr */ };

template <int STOP>
ABR_HD bool stop_hit(double x, double thr) {
    return STOP == STOP_GE ? (x >= thr) : (STOP == STOP_LE ? (x <= thr) : (x < thr));
}

// Performs up to n additions x <- fl(x + c) and stops right after the first whose
// result satisfies the STOP predicate against thr.  Returns true if it stopped on
// the predicate; a_out = number of additions performed (1-based index of the
// stopping one).  c > 0 requires STOP_GE; c < 0 requires STOP_LE or STOP_LT.
// Bit-identical to the naive loop for every finite input (fuzzed in
// tests/test_exact_jump.py).
template <int STOP>
ABR_HD bool chain(double &x_io, double c, double thr, int32_t n, int32_t &a_out) {
    double x = x_io;
    int32_t a = 0;
    int inb = 0;                 // consecutive additions that stayed in one binade
    bool hit = false;
    while (a < n) {
        const double xn = x + c;
        const double d = xn - x; // exact whenever xn and x share a binade
        const int e = expo(xn);
        inb = (e == expo(x)) ? inb + 1 : 0;
        x = xn;
        a++;
        if (stop_hit<STOP>(x, thr)) { hit = true; break; }
        if (inb >= 2 && a < n && e > 0 && e < 2046) {
            // steady state: jump m additions that stay strictly inside the binade
            // and strictly before the stop
            // `lim` bounds the jumped results: they must stay inside the binade and
            // before the stop.  Going down, the binade bottom 2^e itself is excluded:
            // a steady step that lands exactly on 2^e means the exact difference lies
            // just below it, in the finer grid of the next binade, so that step has to
            // be a real subtraction.
            double lim;
            bool strict = true;  // results must be strictly beyond lim (else: may equal it)
            double mf;
            if (STOP == STOP_GE) {
                lim = pow2_biased(e + 1);           // results stay < 2^(e+1) ...
                if (thr < lim) lim = thr;           // ... and < thr
                mf = (lim - x) / d;
            } else {
                lim = pow2_biased(e);               // results stay > 2^e ...
                if (STOP == STOP_LE) { if (thr >= lim) lim = thr; }              // ... and > thr
                else if (thr > lim) { lim = thr; strict = false; }               // ... and >= thr
                mf = (x - lim) / (-d);
            }
            // clamp before converting (mf may be huge or, from rounding, slightly negative)
            const double cap = (double)(n - a);
            if (!(mf > 0.0)) mf = 0.0;
            if (mf > cap) mf = cap;
            int32_t m = (int32_t)mf;               // floor for mf >= 0
            // exact corrections: x + m*d is exact while it stays inside the binade
            if (STOP == STOP_GE) {
                while (m > 0 && !(x + (double)m * d < lim)) m--;
                while (m < n - a && (x + (double)(m + 1) * d < lim)) m++;
            } else if (strict) {
                while (m > 0 && !(x + (double)m * d > lim)) m--;
                while (m < n - a && (x + (double)(m + 1) * d > lim)) m++;
            } else {
                while (m > 0 && !(x + (double)m * d >= lim)) m--;
                while (m < n - a && (x + (double)(m + 1) * d >= lim)) m++;
            }
            x = x + (double)m * d;
            a += m;
        }
    }
    x_io = x;
    a_out = a;
    return hit;
}

}  // namespace abrx
#endif
