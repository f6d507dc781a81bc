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

// ---------------------------------------------------------------------------------------------------------------------
// Drains with ONE constant for every lane: buffer_level -= speed*dt (Simulator.py:184) at one play speed.  (Round 6.)
//
// Going DOWN, the grid only refines, and that removes the "real addition" of a segment altogether.  Let x be a multiple
// of u_e = 2^(e-52) (any x in binade e or e+1 is) and y = x - sd the exact difference.  If y >= 2^e the result is rounded
// on the grid of binade e: fl(y) = x - S_e with S_e = RN(sd / u_e) * u_e (no tie: see make_drain_tab).  If y < 2^e it
// is rounded on the grid of binade e-1, of which x is a multiple as well: fl(y) = x - S_(e-1).  So EVERY subtraction whose
// exact result lies in binade e takes off exactly S_e -- the crossing step included -- and a drain is
//     for e = top binade downwards:  n_e = steps whose exact result stays >= 2^e;   x -= n_e * S_e
// with y >= 2^e  <=>  x - 2^e >= T_e := ceil(sd / u_e) * u_e (both sides multiples of u_e), hence
//     n_e = floor((x - B_e) / S_e) + 1   if x >= B_e := 2^e + T_e,   else 0.
// S_e and B_e depend on the binade only and are wave-uniform: S_e = fl(2^e + sd) - 2^e as in the header comment, T_e = S_e
// or S_e + u_e.  A stage is a handful of exact float64 operations -- x - B_e, n * S_e and the remainder are multiples of
// u_e below 2^53 u_e -- against the ~58 vector instructions of a chain_segment (estimate, three candidates, one real
// addition), and a buffer that runs dry (the player wave's slowest case: six segments and a 16-tick plain tail) is one
// pass over the binades and at most a few plain subtractions below them.  No table in memory: a first form read the stage
// constants from one (scalar loads through the constant address space) and waited for a load in every stage -- the compiler
// sinks a prefetch to its use -- which cost more than the six uniform instructions that compute them.
// Bit-identical to the naive loop (tests/test_lane_jump_cpu.py fuzzes it on the host, tests/test_exact_jump_gpu.py on the device).
struct DrainTab {
    int32_t e_hi = 0;     // biased exponent of the top binade covered
    int32_t n = 0;        // binades covered, downwards from e_hi; 0: no cascade (per-lane speeds, or a speed / range it does
                          // not cover): the general chains do the drains
    double top = 0.0;     // 2^(e_hi + 1): values at or above it lie above the cascade
    double rcp = 0.0;     // 1 / sd: S_e differs from sd by less than 2^-40 of it in every binade covered, and the quotient
                          // is settled with the exact remainder anyway
};

ABR_HD bool wave_any(bool v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __any(v) != 0;
#else
    return v;
#endif
}

// The cascade for subtrahend sd and values below max_level; n == 0 when it does not apply.  It ends above
//   * the binade in which sd / u_e is a round-to-even TIE (the one whose half-ulp is sd's lowest set bit): there the step
//     depends on the parity of x, and
//   * 2 * 2^expo(sd): a step from binade e must land in binade e - 1, not lower.
// Below it drain_cascade subtracts one tick at a time (a handful of ticks).
ABR_HD DrainTab make_drain_tab(double sd, double max_level) {
    DrainTab out;
    if (!(sd > 0.0) || !(max_level > 0.0) || !(sd < 1.0e300) || !(max_level < 1.0e300)) return out;
    const int es = expo(sd), eh = expo(max_level);
    if (es < 64 || es > 1900 || eh > 1900) return out;               // subnormal / absurd ranges: not worth it
    uint64_t bits;
#if defined(__HIP_DEVICE_COMPILE__)
    bits = (uint64_t)__double_as_longlong(sd);
#else
    memcpy(&bits, &sd, 8);
#endif
    const uint64_t mant = (bits & 0xfffffffffffffull) | (1ull << 52);
    int tz = 0;
    while (!((mant >> tz) & 1)) tz++;
    const int e_tie = es + 1 + tz;                                    // biased exponent of the tie binade
    int e_lo = es + 2;
    if (e_tie >= e_lo && e_tie <= eh) e_lo = e_tie + 1;
    if (e_lo > eh) return out;
    if (eh - e_lo + 1 > 64) return out;
    // the quotient guess uses 1 / sd for 1 / S_e: its error is q^2 2^-54, which the +-1 settlement covers while q < 2^26
    if (pow2_biased(eh) * 2.0 / sd >= 67108864.0) return out;
    if (pow2_biased(e_lo) * 2.0 / sd > 64.0) return out;              // the plain tail below the cascade: at most ~64 ticks
    out.e_hi = eh; out.n = eh - e_lo + 1;
    out.top = pow2_biased(eh) * 2.0;
    out.rcp = 1.0 / sd;
    return out;
}

// Up to m subtractions of sd, stopping right after the first result <= 0: the contract of chain<STOP_LE>(x, -sd, 0.0, m, a).
// `tb` must have been made for exactly this sd, and x must be below tb.top (the caller checks both).
ABR_HD bool drain_cascade(const DrainTab &tb_in, double sd, double &x_io, int32_t m, int32_t &a_out) {
    int32_t tb_n = tb_in.n, tb_e_hi = tb_in.e_hi;
    double tb_rcp = tb_in.rcp;
#if defined(__HIP_DEVICE_COMPILE__)
    // The loop's two uniform constants live in VECTOR registers: the role kernels are short of scalar registers, and a
    // constant the compiler re-reads from the kernel arguments inside the loop is a scalar-cache round trip per stage
    // (measured: the cascade then loses to the chains it replaces).
    asm volatile("" : "+v"(sd), "+v"(tb_rcp), "+v"(tb_n), "+v"(tb_e_hi));
    tb_n = __builtin_amdgcn_readfirstlane(tb_n);       // the two loop bounds: scalar values of their own, not kernel-argument
    tb_e_hi = __builtin_amdgcn_readfirstlane(tb_e_hi); // reloads
#endif
    double x = x_io;
    int32_t a = 0;
    int32_t i = 0;
    // binades above every active lane's value: nothing to do there
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    while (i < tb_n && !wave_any((a < m) & (x >= pow2_biased(tb_e_hi - i)))) i++;
    // from the wave's top binade downwards, while any lane has ticks left.  (A binade in which no active lane happens to
    // have a step still costs its stage: the lanes of a wave spread over neighbouring binades, so such gaps are rare, and a
    // test per stage costs every stage.)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (; i < tb_n && wave_any(a < m); i++) {
        const int e = tb_e_hi - i;                      // wave-uniform
        const double base = pow2_biased(e);
        const double u = pow2_biased(e - 52);
        const double S = (base + sd) - base;             // RN(sd / u) * u (not a tie: make_drain_tab)
        const double T = (S >= sd) ? S : S + u;          // ceil(sd / u) * u
        const double B = base + T;
        const double g = x - B;                          // exact (for a lane with a step here; others are masked by n <= 0)
        int32_t q = sat_i32(g * tb_rcp);                // floor(g / S) give or take one (truncation towards zero)
        const double r = g - (double)q * S;              // exact remainder of that guess
        q -= (r < 0.0) ? 1 : 0;
        q += (r >= S) ? 1 : 0;
        int32_t n = q + 1;                               // g < 0: q <= -1 after the settlement, n <= 0
        const int32_t left = m - a;                      // <= 0 for a lane that is done
        n = n < left ? n : left;
        n = n > 0 ? n : 0;
        x = x - (double)n * S;                           // exact
        a += n;
    }
    // below the cascade the grid is finer than sd's own ulp (or the one tie binade sits there): a few plain subtractions
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
#ifdef ABR_CASCADE_HOOK
    const int32_t a_stages = a;
#endif
    while (a < m && x > 0.0) { x = x - sd; a++; }
#ifdef ABR_CASCADE_HOOK
    ABR_CASCADE_HOOK(tb_e_hi, tb_n, x_io, x, a_stages, a - a_stages, m);   // host-side analysis builds: binades visited, plain ticks
#endif
    x_io = x; a_out = a;
    return a > 0 && x <= 0.0;
}

}  // namespace abrx
#endif
