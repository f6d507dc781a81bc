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
// i.e. a constant integer increment D = RNE(c/u) whatever X is -- except in the tie
// case, where the first step depends on the parity of X + q; a tie always produces
// an even X, so from the second in-binade step on D is the constant q + (q odd)
// there too, which is again RNE(c/u).  D*u is obtained without touching x:
//     d = fl(2^e + |c|) - 2^e        (2^e has an even mantissa; needs |c| < 2^e)
// and c - d is computed exactly, so "tie" is the test |c - d| == u/2.  Hence:
//     inside one binade every addition adds exactly d (after one real in-binade
//     addition in the tie case), and x + m*d is computed exactly in float64 (all
//     terms are multiples of u below 2^53 * u).
// The same holds for a negative constant.  Steps that leave the binade are executed
// as real additions.
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

// Reciprocal estimate for the jump-length guess (the guess is corrected exactly).
ABR_HD double rcp_est(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(d);      // v_rcp_f64
#else
    return 1.0 / d;
#endif
}

// (int32_t)v with saturation: NaN -> 0, below INT32_MIN -> INT32_MIN, above INT32_MAX ->
// INT32_MAX.  v_cvt_i32_f64 does exactly that in one instruction; the C++ cast of an
// out-of-range double is undefined, so the host build spells it out.
ABR_HD int32_t sat_i32(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t r;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(v));
    return r;
#else
    if (!(v == v)) return 0;
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return (int32_t)v;
#endif
}

// Running state of one chain, so that a caller can interleave other work (the
// download integration advances through trace intervals between segments).
struct ChainState {
    double x;      // current value
    int32_t eb;    // biased exponent of the binade the last real addition of the CURRENT constant started in
                   // (it stayed inside iff x is still in that binade), or -1: none yet / the constant changed
};

// min / max of two non-NaN doubles in one instruction (fmin / fmax would canonicalise their operands first)
ABR_HD double min_num(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return (b < a) ? b : a;
#endif
}
ABR_HD double max_num(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return (b > a) ? b : a;
#endif
}

// Is a jumped result y on the allowed side of lim?
template <int STOP>
ABR_HD bool jump_inside(double y, double lim, bool strict) {
    if (STOP == STOP_GE) return y < lim;
    return strict ? (y > lim) : (y >= lim);
}

// Most additions one segment performs.  The jump-length estimate gap * v_rcp_f64(dm) carries the reciprocal's
// relative error eps (the hardware's double-precision reciprocal is a seed, good to a few parts in 2^24 at worst); while
// the true quotient is below 1 / (2 eps) the estimate is off by less than half a step, and a longer one is cut to the
// segment's budget <= kJumpCap, which is then certainly not too long.  So candidate 0 of the settlement below is always
// inside -- and because that is an argument about an estimate, the settlement still checks it (ok0) and repairs an
// overshoot in line; correctness never rests on eps, only speed does.  A run of more than 2^20 equal additions inside one
// binade (2.9 hours of 0.01 s ticks) takes one segment per 2^20 steps (tests/test_exact_jump_gpu.py: jumps of 2^27 steps).
constexpr int32_t kJumpCap = 1 << 20;

// min(max(v, 0), hi) for hi >= 0 in one instruction
ABR_HD uint32_t clamp0(int32_t v, int32_t hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "v"(hi));
    return r;
#else
    v = (v > 0) ? v : 0;
    return (uint32_t)((v < hi) ? v : hi);
#endif
}

// Wave-level "does any lane need the repair path?" -- the repair (an estimate that overshot)
// has never been observed, so the branch around it should cost a vote, not a divergent region.
ABR_HD bool any_lane(bool v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_expect(__any(v) != 0, 0);
#else
    return v;
#endif
}

// One SEGMENT of a chain: one exact jump towards the end of the binade / the stop / the end of
// the budget, then one real addition (the one that crosses the binade or satisfies the stop, or
// simply the last of the budget).  Performs at most n additions; returns how many (>= 1 when
// n >= 1) and sets `hit` when the last one satisfied the STOP predicate.  The jump always leaves
// one addition of the budget for the real one, so a segment with n >= 1 ends in a real addition.
// cs.eb remembers the binade that real addition started in: in the tie case a jump is allowed
// only if the previous real addition of this constant stayed inside the binade x is in now
// (header comment); a caller that changes c sets cs.eb = -1.
// Straight-line selects on purpose: this is the body of the GPU hot loop.
// BIAS (tests only) is added to the jump-length estimate: a non-zero value spoils it on purpose,
// so that the repair path (estimate too long) and the short-jump path (estimate too short) run.
template <int STOP, int BIAS = 0>
ABR_HD int32_t chain_segment(ChainState &cs, double c, double thr, int32_t n_in, bool &hit_out) {
#ifdef ABR_SEGMENT_HOOK
    ABR_SEGMENT_HOOK(STOP);              // host-side analysis builds count segments per chain kind
#endif
    const int32_t n = (n_in < kJumpCap) ? n_in : kJumpCap;   // see kJumpCap; callers loop until their budget is spent
    double x = cs.x;
    int32_t a = 0;
    const int e = expo(x);
    const bool normal = (e > 54) & (e < 2046);       // else: no jump; the values below are then unused (all stay finite
    const int ec = e;                                // or NaN-free enough: base = 0 or tiny, d = |c|, and m0 is clamped to 0)
    const double ac = (STOP == STOP_GE) ? c : -c;    // |c|: the sign of c is fixed by the stop kind (see chain())
    // steady increment of this binade and the tie test (header comment)
    const double base = pow2_biased(ec);
    const double dm = (base + ac) - base;            // RNE(|c| / u) * u
    const double rem = ac - dm;                      // exact, in [-u/2, u/2]
    const double half_u = base * 0x1p-53;            // u/2 = 2^(e-53): exact (ec > 54)
    const bool tie = ((rem < 0.0) ? -rem : rem) == half_u;
    const bool go = n > 0;
    const bool can = normal & go & (expo(ac) < e) & (!tie | (cs.eb == e));
    const double d = (STOP == STOP_GE) ? dm : -dm;
    {
        // `lim` bounds the jumped results: they must stay inside the binade and before
        // the stop.  Going down, the binade bottom 2^e itself is excluded: a steady step
        // that lands exactly on 2^e means the exact difference lies just below it, on the
        // finer grid of the next binade, so that step has to be a real subtraction.
        double lim, gap;
        bool strict = true;              // results strictly beyond lim (else: may equal it)
        if (STOP == STOP_GE) {
            lim = min_num(thr, base + base);         // results stay < 2^(e+1) and < thr
            gap = lim - x;
        } else {
            if (STOP == STOP_LE) {
                lim = max_num(thr, base);            // results stay > 2^e and > thr
            } else {
                strict = !(thr > base);              // ... and >= thr
                lim = max_num(thr, base);
            }
            gap = x - lim;
        }
        const int32_t room = can ? n - 1 : 0;        // the last addition of the budget is a real one
        // Jump length: estimate gap / dm, clamped to [0, room] (NaN and negatives -> 0); room < kJumpCap.
        int32_t m0s = sat_i32(gap * rcp_est(dm));
        if (BIAS != 0) m0s = (m0s < 0x7fffff00 && m0s > -0x7fffff00) ? m0s + BIAS : m0s;
        const uint32_t m0 = clamp0(m0s, room);
        // Exact settlement (x + m*d is exact while it stays inside the binade).  With the answer
        // m* = the longest jump that stays inside, the estimate satisfies m0 - 1 <= m* (see
        // kJumpCap), so candidate bs = m0 - 1 is inside and the answer is the last of bs, bs + 1,
        // bs + 2 that is: candidates 1 and 2 are formed by adding d to the previous one, identical
        // to x + (bs+k)*d whenever the previous candidate is inside (both are then exact), and a
        // candidate only counts if all before it are inside.  A jump SHORTER than m* is always
        // legal -- the next segment carries on -- so nothing has to bound m* from above; only
        // "candidate 0 is inside" is load-bearing, and it is not taken on trust: ok0 checks it,
        // and a lane whose estimate overshot walks back to the last inside candidate.  The vote
        // keeps that path out of the way (one compare and a scalar branch per segment).
        const uint32_t bs = (m0 > 0) ? m0 - 1 : 0;   // a saturating subtraction
        const double y0 = x + (double)bs * d;
        const double y1 = y0 + d;
        const double y2 = y1 + d;
        const bool t1 = (bs + 1 <= (uint32_t)room) & jump_inside<STOP>(y1, lim, strict);
        const bool t2 = (bs + 2 <= (uint32_t)room) & jump_inside<STOP>(y2, lim, strict) & t1;
        int32_t m = (int32_t)(bs + (t1 ? 1 : 0) + (t2 ? 1 : 0));
        double xj = t2 ? y2 : (t1 ? y1 : y0);
        const bool ok0 = (bs == 0) | jump_inside<STOP>(y0, lim, strict);
#ifdef ABR_BRACKET_HOOK
        ABR_BRACKET_HOOK(ok0);           // host-side analysis builds count overshooting estimates
#endif
        if (any_lane(!ok0)) {
            int32_t mm = ok0 ? 0 : (int32_t)bs;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang loop unroll(disable) vectorize(disable)
#endif
            while (mm > 0 && !jump_inside<STOP>(x + (double)mm * d, lim, strict)) mm--;
            if (!ok0) { m = mm; xj = x + (double)mm * d; }
        }
        x = xj;                          // == x when !can (m == 0)
        a = m;
    }
    // ---- one real addition (always, when there is a budget at all) ----
    const double xn = x + c;
    const bool hit = go & stop_hit<STOP>(xn, thr);
    x = go ? xn : x;
    a += go ? 1 : 0;
    cs.x = x;
    cs.eb = go ? e : cs.eb;              // "it stayed inside" is read off x's exponent by the next segment
    hit_out = hit;
#ifdef ABR_SEGMENT_END_HOOK
    ABR_SEGMENT_END_HOOK(STOP, a, n_in, hit, can && expo(x) == e);   // host-side analysis builds: how the segment ended
#endif
    return a;
}

// Performs up to n additions x <- fl(x + c) and stops right after the first whose
// result satisfies the STOP predicate against thr.  Returns true if it stopped on
// the predicate; a_out = number of additions performed (1-based index of the
// stopping one).  c > 0 requires STOP_GE; c < 0 requires STOP_LE or STOP_LT.
// Bit-identical to the naive loop for every finite input (fuzzed in
// tests/test_exact_jump.py).
template <int STOP, int BIAS = 0>
ABR_HD bool chain(double &x_io, double c, double thr, int32_t n, int32_t &a_out) {
    ChainState cs;
    cs.x = x_io; cs.eb = -1;
    int32_t a = 0;
    bool hit = false;
    while (a < n && !hit) a += chain_segment<STOP, BIAS>(cs, c, thr, n - a, hit);
    x_io = cs.x;
    a_out = a;
    return hit;
}

}  // namespace abrx
#endif
