// abr_lane_jump.h -- one lane of the event-driven environment step (host + device).
//
// The same tick semantics as the reference's Simulator.run() loop
// (Simulator.py:135-208 under R1-R3) without visiting every tick.  A step (ABR
// call site -> next call site) is a handful of runs in which one float64
// variable receives the same constant every tick:
//   phase A  downloaded_size += bandwidth*dt, one run per trace interval, until it
//            reaches target_size at tick k_hit            (chain<STOP_GE>, :160-163)
//            meanwhile buffer_level -= speed*dt while playing, until 0
//                                                          (chain<STOP_LE>, :184,:194)
//   hit tick buffer_level = (buffer_level + L) - speed*dt, flags, start-up exit
//                                                          (:170,:184,:190-202)
//   phase B  wait for the next chunk to become available (:143): buffer_level drains
//            for avail_tick - k ticks, then until buffer_level < max_buffer when
//            buffer_full gates the download               (chain<STOP_LT>, :144,:190)
// abrx::chain executes each run in O(binades crossed) and returns the bit-identical
// float64 value of the tick-by-tick loop (abr_exact_jump.h); the integer counters
// (ticks in start-up / rebuffering / playing, and the latency integral: sum of k
// over playing ticks) have exact closed forms over a run.
//
// State between the pieces is always "post-head": T1-T3 (:137-149) of tick s.k are
// done.  A download cannot pause inside phase A: buffer_full only turns on in a tick
// that completes a chunk (buffer_level grows nowhere else), availability is
// monotone in time.
//
// Plain C++ so that tests/native can run it on the CPU against the oracle; the
// product only ever runs it inside the HIP kernels of abr_env.hip.
#ifndef ABR_LANE_JUMP_H
#define ABR_LANE_JUMP_H

#include "abr_exact_jump.h"

#ifndef ABR_STAMP
#define ABR_STAMP(n)               // cycle stamps exist only in the diagnostic build of abr_env.hip
#endif

namespace abrx {

constexpr double kTickDt = 0.01;   // Simulator.py:133
#ifndef ABR_K_DRAIN_TAIL
#define ABR_K_DRAIN_TAIL 16
#endif
constexpr int kPrologue = 16;               // (the asynchronous pipeline's own prologue: diagnostic build only)
#ifndef ABR_PCHUNKS
#define ABR_PCHUNKS 12
#endif
#ifndef ABR_PCHECK
#define ABR_PCHECK 0x892
#endif
constexpr int kPChunks = ABR_PCHUNKS;       // prologue of a download: up to 7 single additions + this many chunks of 8
constexpr int kPCheck = ABR_PCHECK;         // bit q: a checkpoint after chunk q (the last chunk's bit must be set)

struct Tables {
    const double *G;               // G[n] = dt added n times to 0.0 (global_time, download_time, ...)
    const int32_t *interval_tick;  // first tick k with int(G[k]/interval) >= j          (:158)
    const int32_t *avail_tick;     // first tick k with int(G[k]/chunk_length) - 1 >= c  (:143)
    double L, sd, max_buffer, start_up_length;
    int32_t V, max_ticks;
    bool per_lane_speed;           // each lane carries its own speed*dt and play_time (8f rank 3)
    // speed schedule (8f rank 3, second half): speed_rows >= 2 means the lane's play speed is
    // re-read at the first playing tick of every played chunk (play_length == 0, :176-177):
    // played chunk p of lane i plays at speeds[min(p, speed_rows - 1) * speed_stride + i]
    int32_t speed_rows;
    int64_t speed_stride;
    const double *speeds;
    // the per-binade cascade of buffer_level -= sd at THE one play speed (abr_exact_jump.h: drain_cascade); n == 0 with
    // per-lane speeds or when the speed / buffer range is not covered: the general chains then do the drains
    DrainTab drain;
};

// Where a lane is in its bandwidth trace.  Only the download side (phase A) reads it.
struct Cursor {
    int32_t j;                     // interval index of the clock, int(global_time / interval) (:158)
    int32_t tpos;                  // (offset0 + j) mod tlen: bandwidths[idx] of that interval (:159)
    int32_t tlen;
    const double *trace;
};

struct LaneJ {
    double buf;                    // buffer_level
    double sd;                     // this lane's speed*dt (== Tables::sd unless per_lane_speed)
    double pt;                     // play_time, carried only when per_lane_speed (else GP[n_play])
    long long sumk;                // sum of tick indices of playing ticks (latency integral)
    int32_t k, chunk_id, n_su, n_rb, n_play, avail_k, last_action;
    bool su, be, bf;               // start_up, buffer_empty, buffer_full
    Cursor cur;
    // speed schedule only: playing ticks left in the current played chunk (0: the next playing
    // tick starts one and asks for its speed), chunks played so far (play_id), the sum of
    // play_time over the playing ticks (latency integral), and the lane's column in `speeds`
    int32_t pl_left, play_id;
    double pt_sum;
    int64_t lane;
};

ABR_HD void cursor_init(Cursor &c, int32_t offset0) {
    c.j = 0;                       // int(0.0 / interval)
    c.tpos = offset0 % c.tlen;
}

// Simulator.py:95-130, then T1-T3 of tick 0 (start_up_time += dt); everything but the cursor
ABR_HD void lanej_init_player(LaneJ &s, const Tables &t) {
    s.buf = 0.0; s.sumk = 0;
    s.k = 0; s.chunk_id = 0; s.n_su = 1; s.n_rb = 0; s.n_play = 0;
    s.last_action = -1;
    s.su = true; s.be = true; s.bf = false;
    s.avail_k = t.avail_tick[0];
    s.pt = 0.0;                    // play_time = 0 (:115); s.sd is set by the caller
    s.pl_left = 0; s.play_id = 0; s.pt_sum = 0.0;      // play_length = 0, play_id = 0 (:113-114)
}

ABR_HD void lanej_init(LaneJ &s, const Tables &t, int32_t offset0) {
    lanej_init_player(s, t);
    cursor_init(s.cur, offset0);
}

// play_time += speed*dt for `a` playing ticks (:182).  With one speed for all lanes
// play_time is the table value GP[n_play]; with per-lane speeds it is carried, advanced by
// the same exact chain machinery.
ABR_HD void lanej_play(LaneJ &s, const Tables &t, int32_t a) {
    if (!t.per_lane_speed || a <= 0) return;
    int32_t done = 0;
    double x = s.pt;
    // an unreachable threshold: the chain only counts
    chain<STOP_GE>(x, s.sd, 1.0e300, a, done);
    s.pt = x;
}

// buffer_level -= speed*dt per playing tick (:184) for up to m ticks, stopping right after the
// first result <= 0 (:194).  Same contract as chain<STOP_LE>(b, -sd, 0.0, m, a).  Far from
// zero the exact jumps do the work; within kDrainTail ticks of zero a binade lasts only a
// few ticks (8, 4, 2, 1: one segment each), so the last stretch is plain subtractions -- the
// reference's own sequence.  The switch point affects speed only, never a result; measured
// on one MI355X box at 65 536 lanes (profiles/r02_ab_drain_tail.txt): 8-32 ticks are
// equivalent (7.3-7.4e9 env-steps/s), 64 costs 5 %, 128 costs 20 %.
// (A buffer that runs dry is the player wave's slowest case -- six segments and the whole tail, one lane in 25, so 86 % of a
// wave's decisions have one -- and only the tick it ends at is needed, not its values; but that tick cannot be had from
// real arithmetic: buffer levels are sums of chunk lengths and tick-sized subtractions, so b / sd sits within rounding of
// a whole number exactly when it matters, and which side of zero the k-th result falls on is decided by the roundings
// themselves.  Built and measured in round 5, profiles/r05_experiments_not_kept.txt.)
constexpr int kDrainTail = ABR_K_DRAIN_TAIL;
ABR_HD bool drain_to_zero(double &b_io, double sd, int32_t m, int32_t &a_out) {
    ChainState cs;
    cs.x = b_io; cs.eb = -1;
    const double tail = (double)kDrainTail * sd;
    int32_t a = 0;
    bool below = false;
    while (a < m && !below) a += chain_segment<STOP_LE>(cs, -sd, tail, m - a, below);
    double b = cs.x;
    ABR_STAMP(24);
#ifdef ABR_DRAIN_HOOK
    const int32_t a_seg = a;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    while (a < m && b > 0.0) { b = b - sd; a++; }
    ABR_STAMP(25);
#ifdef ABR_DRAIN_HOOK
    ABR_DRAIN_HOOK(a > 0 && b <= 0.0, a - a_seg);   // host-side analysis builds: did it run dry, ticks of the plain tail
#endif
    b_io = b; a_out = a;
    return a > 0 && b <= 0.0;
}

// The drain of the common case -- one play speed for every lane -- goes through the per-binade cascade (round 6: a stage
// is ~25 vector instructions against the ~58 of a chain segment plus its loop, and a buffer that runs dry is one pass over
// the binades instead of six segments and a 16-tick tail).  Wave-uniform choice: a lane above the cascade (never seen: it
// covers max_buffer + chunk_length) sends the whole wave through the chains.
ABR_HD bool lanej_drain(const Tables &t, double &b_io, double sd, int32_t m, int32_t &a_out) {
    if (t.drain.n > 0 && !wave_any(!(b_io < t.drain.top))) {
        const bool zero = drain_cascade(t.drain, sd, b_io, m, a_out);
        ABR_STAMP(24);
        return zero;
    }
    return drain_to_zero(b_io, sd, m, a_out);
}

// ---- speed schedule: the play speed changes at played-chunk boundaries ----
// The first playing tick of a played chunk (play_length == 0) takes the chunk's speed
// (:176-177); the chunk then lasts until play_length, `speed*dt` added per tick from 0, is
// >= chunk_length (:183,:185-187): that many ticks, by the same exact chain.
ABR_HD void sched_begin_chunk(LaneJ &s, const Tables &t) {
    const int32_t row = s.play_id < t.speed_rows ? s.play_id : t.speed_rows - 1;
    s.sd = t.speeds[(int64_t)row * t.speed_stride + s.lane] * kTickDt;     // play_speed * dt (:182)
    double x = 0.0;
    int32_t a = 0;
    chain<STOP_GE>(x, s.sd, t.L, t.max_ticks + 1, a);
    s.pl_left = a > 0 ? a : 1;
}

// bookkeeping of `a` playing ticks inside one played chunk: play_time (exact chain), the sum of
// play_time over those ticks (real arithmetic: it only feeds average_latency), chunk boundary
ABR_HD void sched_played(LaneJ &s, const Tables &t, int32_t a) {
    if (a <= 0) return;
    s.pt_sum += (double)a * s.pt + s.sd * (double)(((long long)a * (a - 1)) / 2);
    int32_t done = 0;
    double x = s.pt;
    chain<STOP_GE>(x, s.sd, 1.0e300, a, done);
    s.pt = x;
    s.pl_left -= a;
    if (s.pl_left == 0) s.play_id++;                                       // :185-187
}

// buffer_level -= speed*dt for up to m playing ticks, stopping right after the first result that
// is <= 0 (STOP_LE, :194) or < thr (STOP_LT, :190), one played chunk at a time
template <int STOP>
ABR_HD bool sched_drain(LaneJ &s, const Tables &t, double &b, double thr, int32_t m, int32_t &a_out) {
    int32_t a_tot = 0;
    bool hit = false;
    while (a_tot < m && !hit) {
        if (s.pl_left == 0) sched_begin_chunk(s, t);
        const int32_t run = (m - a_tot < s.pl_left) ? m - a_tot : s.pl_left;
        int32_t a = 0;
        if (STOP == STOP_LE) hit = drain_to_zero(b, s.sd, run, a);
        else hit = chain<STOP_LT>(b, -s.sd, thr, run, a);
        sched_played(s, t, a);
        a_tot += a;
    }
    a_out = a_tot;
    return hit;
}

// m full iterations: T4-T9 of a tick in which no chunk completes, then T1-T3 of the next
ABR_HD void lanej_idle(LaneJ &s, const Tables &t, int32_t m) {
    if (m <= 0) return;
    // :201-202 at the end of the first of these ticks.  A no-op everywhere (start_up implies
    // buffer_level < start_up_length once any tick has run) except right after init when
    // start_up_length <= 0: tick 0 itself counts as start-up, every later one does not.
    if (s.su && s.buf >= t.start_up_length) s.su = false;
    ABR_STAMP(23);
    if (s.su) {
        s.n_su += m;                                   // :137-138; nothing plays, buffer untouched
    } else if (s.be) {
        s.n_rb += m;                                   // :139-140; buffer stays 0
    } else {
        int32_t a = 0;
        double b = s.buf;
        bool zero;
        if (t.speed_rows >= 2) zero = sched_drain<STOP_LE>(s, t, b, 0.0, m, a);
        else { zero = lanej_drain(t, b, s.sd, m, a); lanej_play(s, t, a); }       // :184,:194
        s.n_play += a;
        s.sumk += (long long)a * s.k + ((long long)a * (a - 1)) / 2;
        if (zero) { b = 0.0; s.be = true; s.n_rb += (m - a + 1); }             // :195-196, then :140
        s.buf = b;
        s.bf = b >= t.max_buffer;                      // :190, as of the last tick executed
    }
    s.k += m;
}

// From a post-head state that is not downloading: advance to the next call site
// (returns true) or to max_ticks (returns false).
ABR_HD bool lanej_wait_call(LaneJ &s, const Tables &t) {
    const int32_t mt = t.max_ticks;
    if (s.k >= s.avail_k && !s.bf) return true;
    int32_t w = s.avail_k - s.k;
    if (w < 0) w = 0;
    if (w > mt - s.k) w = mt - s.k;
    lanej_idle(s, t, w);
    if (s.k >= mt) return false;
    if (s.bf) {
        // buffer_full gates the next download (:144): drain until buffer_level < max_buffer
        if (s.su || s.be) {
            // nothing drains the buffer: the reference spins forever; run out the clock
            if (s.su) s.n_su += mt - s.k; else s.n_rb += mt - s.k;
            s.k = mt;
            return false;
        }
        int32_t a = 0;
        double b = s.buf;
        bool cleared;
        if (t.speed_rows >= 2) cleared = sched_drain<STOP_LT>(s, t, b, t.max_buffer, mt - s.k, a);
        else { cleared = chain<STOP_LT>(b, -s.sd, t.max_buffer, mt - s.k, a); lanej_play(s, t, a); }
        s.n_play += a;
        s.sumk += (long long)a * s.k + ((long long)a * (a - 1)) / 2;
        s.k += a;
        s.be = b <= 0.0;
        if (s.be) { b = 0.0; s.n_rb += 1; }
        s.buf = b;
        s.bf = !cleared;
        if (!cleared) return false;
    }
    return true;
}

struct StepResult {
    double bw;        // downloaded_size / download_time of the chunk (:164), valid when hit
    bool hit;         // the chunk completed
    bool ended;       // chunk_id >= video_length (:207-208)
    bool timeout;     // ran into max_ticks
};

// Everything a step needs from memory before it can start, fetched in ONE burst of
// independent loads (a lane alone on its SIMD cannot hide a chain of dependent loads):
// phase B moved k only, so the trace cursor (j, tpos) may be up to kCatch intervals
// behind; the candidates for all those cases are loaded at once and selected in
// registers.  The caller can issue this before it computes the action.
constexpr int kCatch = 4;
struct StepStart {
    double c;          // bandwidth * dt of the interval the call site is in (:160)
    double bw_next;    // bandwidth of the next interval
    int32_t ke;        // end tick of the current interval
    int32_t ke_next;   // end tick of the next one
    int32_t tn;        // trace position of the next interval
    int32_t avail_next;// avail_tick[chunk_id + 1]: when the chunk after this one can start (:143)
};

ABR_HD int32_t trace_wrap(int32_t pos, int32_t tlen) {
    if (pos >= tlen) pos -= tlen;
    if (pos >= tlen) pos %= tlen;             // traces shorter than the look-ahead
    return pos;
}

// The loads of lanej_begin_step on their own, so that a caller can issue them well before it
// needs the values (the role-split kernel issues them before its workgroup barrier).
struct StepLoads {
    int32_t ke[kCatch + 2];
    double bw[kCatch + 2];
    int32_t avail_next;
};

ABR_HD StepLoads lanej_begin_load(const Cursor &s, const Tables &t, int32_t chunk_id) {
    StepLoads ld;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < kCatch + 2; i++) {
        ld.ke[i] = t.interval_tick[s.j + 1 + i];
        ld.bw[i] = s.trace[trace_wrap(s.tpos + i, s.tlen)];
    }
    ld.avail_next = t.avail_tick[chunk_id + 1];
    return ld;
}

// ... and the selection among them once the call-site tick k is known.  The cursor must be the
// one the loads were issued for.
ABR_HD StepStart lanej_begin_select(Cursor &s, const Tables &t, const StepLoads &ld, int32_t k) {
    const int32_t *ke = ld.ke;
    const double *bw = ld.bw;
    StepStart st;
    st.avail_next = ld.avail_next;
    // intervals the cursor is behind: ke[] is non-decreasing
    int32_t adv = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < kCatch; i++) adv += (k >= ke[i]) ? 1 : 0;
    if (adv == kCatch && k >= ke[kCatch]) {
        // more than kCatch intervals behind (a long buffer_full wait): walk, then reload
        s.j += kCatch; s.tpos = trace_wrap(s.tpos + kCatch, s.tlen);
        int32_t e = t.interval_tick[s.j + 1];
        while (k >= e && e != 0x7fffffff) {       // the table ends in INT_MAX sentinels
            s.j++;
            s.tpos = (s.tpos + 1 == s.tlen) ? 0 : s.tpos + 1;
            e = t.interval_tick[s.j + 1];
        }
        st.ke = e; st.ke_next = t.interval_tick[s.j + 2];
        st.tn = (s.tpos + 1 == s.tlen) ? 0 : s.tpos + 1;
        st.c = s.trace[s.tpos] * kTickDt; st.bw_next = s.trace[st.tn];
        return st;
    }
    // select candidate `adv` (static indices only: the arrays stay in registers)
    double c_bw = bw[0], n_bw = bw[1];
    int32_t c_ke = ke[0], n_ke = ke[1];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 1; i <= kCatch; i++) {
        const bool pick = adv == i;
        c_bw = pick ? bw[i] : c_bw; n_bw = pick ? bw[i + 1] : n_bw;
        c_ke = pick ? ke[i] : c_ke; n_ke = pick ? ke[i + 1] : n_ke;
    }
    s.j += adv;
    s.tpos = trace_wrap(s.tpos + adv, s.tlen);
    st.c = c_bw * kTickDt; st.bw_next = n_bw; st.ke = c_ke; st.ke_next = n_ke;
    st.tn = (s.tpos + 1 == s.tlen) ? 0 : s.tpos + 1;
    return st;
}

ABR_HD StepStart lanej_begin_step(Cursor &s, const Tables &t, int32_t k, int32_t chunk_id) {
    const StepLoads ld = lanej_begin_load(s, t, chunk_id);
    return lanej_begin_select(s, t, ld, k);
}

// Phase A of one decision: the download side.  Needs nothing of the player state but
// the call-site tick k: the download cannot pause before it completes, so it is a pure
// function of (k, cursor, target) -- which is what lets a second thread run the player
// side of the SAME lane concurrently (abr_env.hip: env_split_kernel).
struct Download {
    double dl;        // downloaded_size at the completing tick (:160-163)
    int32_t n_dl;     // ticks it took: download_time = G[n_dl] (:161)
    bool hit;         // false: max_ticks reached first
};

ABR_HD Download lanej_download(Cursor &s, const Tables &t, const StepStart &st, int32_t k,
                               double target) {
    // One flat loop over chain SEGMENTS (abr_exact_jump.h); a lane moves on to its next
    // trace interval between two segments.  The next interval's bandwidth and end tick
    // are loaded one interval ahead so the loads overlap the arithmetic.  kk is the tick the
    // next addition belongs to: download_time is G[kk - k] (:161); interval ends are clamped to
    // max_ticks, so "ticks left in the interval" is also "ticks left at all".
    const int32_t mt = t.max_ticks;
    int32_t ke = st.ke < mt ? st.ke : mt, ke_next = st.ke_next, tn = st.tn;
    double c = st.c, bw_next = st.bw_next;
    ChainState cs;
    cs.x = 0.0; cs.eb = -1;                   // downloaded_size = 0 at a call site
    int32_t kk = k;
    bool hit = false;
    {
        // Prologue: downloaded_size starts at 0, so its first additions cross a binade every 1, 2, 4, 8, ... steps,
        // where a jump buys nothing (a 95-instruction trip for a handful of ticks).  The first up to 7 + 8 kPChunks additions are
        // therefore PLAIN additions -- that IS the reference's sequence -- and they have to work for every lane: a
        // wave pays its slowest lane, and with 64 lanes some call site always sits just before an interval end
        // (rounds 2-3 kept the prologue only when it fitted the current interval: 16 % of downloads got none).
        // The additions are n1 = ticks left in the current interval at c, then c2 = the next interval's constant.
        // To keep the code straight-line the first a = n1 mod 8 additions are single predicated ones, which
        // aligns the interval boundary with a chunk boundary; then kPChunks chunks of 8 with one constant each.
        // Checkpoints (bit q of kPCheck: after chunk q) are kept while they stay below the target (the sequence
        // increases, so nothing before a kept checkpoint reached it), within max_ticks, and inside the NEXT interval.
        const int32_t left = ke - kk;                                  // ticks of the current interval, >= 1
        const int32_t n1 = left > 0 ? left : 0;
        const int32_t a = n1 & 7, q1 = n1 >> 3;                        // singles, then q1 whole chunks at c
        const double c2 = bw_next * kTickDt;
        const int32_t room2 = (ke_next < mt ? ke_next : mt) - ke;      // ticks of the next interval (clamped to max_ticks)
        double x = 0.0, xb = 0.0;
        int32_t T = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 0; i < 7; i++) x = (i < a) ? x + c : x;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int q = 0; q < kPChunks; q++) {
            const double cq = (q < q1) ? c : c2;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int i = 0; i < 8; i++) x = x + cq;
            if ((kPCheck >> q) & 1) {
                // a checkpoint after Tq ticks is usable if it is below the target and either all in the current
                // interval (Tq <= n1: then also within max_ticks, ke being clamped) or its part beyond n1 ends
                // strictly inside the next interval.  Usability only ever turns off as q grows.
                const int32_t Tq = a + 8 * (q + 1);
                const bool ok = (x < target) & ((Tq <= n1) | (Tq - n1 < room2));
                xb = ok ? x : xb;
                T = ok ? Tq : T;
            }
        }
        cs.x = xb;
        // past the interval boundary?  then the cursor moves on here (the loop reloads the look-ahead from it)
        const bool crossed = T > n1;
        kk += T;
        c = crossed ? c2 : c;
        ke = crossed ? (ke_next < mt ? ke_next : mt) : ke;
        s.j += crossed ? 1 : 0;
        s.tpos = crossed ? tn : s.tpos;
    }
    while (!hit && kk < mt) {
        // Interval over?  Its successor was prefetched.  Branch-free on purpose, and the
        // prefetch of the interval after that is (re)issued in EVERY trip: a load inside
        // a divergent `if` must be waited for at the end of that `if` (the loaded
        // registers merge with the not-taken path), exposing its full latency; issued
        // unconditionally it is only needed one trip later.
        const bool adv = kk >= ke;
        c = adv ? bw_next * kTickDt : c;
        ke = adv ? (ke_next < mt ? ke_next : mt) : ke;
        s.j += adv ? 1 : 0;
        s.tpos = adv ? tn : s.tpos;
        cs.eb = adv ? -1 : cs.eb;             // new constant: the steady state is void
        tn = (s.tpos + 1 == s.tlen) ? 0 : s.tpos + 1;
        bw_next = s.trace[(uint32_t)tn];
        ke_next = t.interval_tick[(uint32_t)(s.j + 2)];
        kk += chain_segment<STOP_GE>(cs, c, target, ke - kk, hit);                // :160-163
    }
    Download d;
    d.dl = cs.x; d.n_dl = kk - k; d.hit = hit;
    return d;
}

// The rest of the decision: the player side of the download's ticks, the completing tick,
// then phase B up to the next call site.  `action` only labels the step (last_action);
// avail_next = avail_tick[chunk_id + 1].
ABR_HD StepResult lanej_after_download(LaneJ &s, const Tables &t, const Download &d,
                                       int32_t avail_next, int32_t action) {
    StepResult r;
    r.bw = 0.0; r.hit = false; r.ended = false; r.timeout = false;
    const int32_t mt = t.max_ticks;
    const bool hit = d.hit;
    const int32_t n_dl = d.n_dl;
    const double g_ndl = t.G[n_dl];           // download_time; loaded now, divided by much later
    // ---- buffer side of the ticks before the completing one ----
    lanej_idle(s, t, hit ? n_dl - 1 : n_dl);
    ABR_STAMP(10);
    if (!hit) { r.timeout = true; return r; }
    // ---- the completing tick (:163-170, then :174-202) ----
    const bool playing = !(s.be || s.su);
    double b = s.buf + t.L;                                                  // :170
    if (playing) {                                                            // :176-184
        s.sumk += s.k; s.n_play++;
        if (t.speed_rows >= 2) {
            if (s.pl_left == 0) sched_begin_chunk(s, t);
            b = b - s.sd;
            sched_played(s, t, 1);
        } else { b = b - s.sd; lanej_play(s, t, 1); }
    }
    s.bf = b >= t.max_buffer;                                                // :190
    s.be = b <= 0.0;                                                         // :194
    if (s.be) b = 0.0;
    s.buf = b;
    s.su = s.su && !(b >= t.start_up_length);                                // :201-202
    s.k++;                                                                   // :205
    r.hit = true;
    r.bw = d.dl / g_ndl;                                                     // :164
    s.last_action = action;
    s.chunk_id++;                                                            // :166
    s.avail_k = avail_next;
    r.ended = s.chunk_id >= t.V;                                             // :207-208
    r.timeout = !r.ended && s.k >= mt;
    ABR_STAMP(11);
    if (!r.ended && !r.timeout) {
        s.n_su += s.su ? 1 : 0;                                              // T1 of the next tick
        s.n_rb += (!s.su && s.be) ? 1 : 0;
        r.timeout = !lanej_wait_call(s, t);                                  // phase B
    }
    ABR_STAMP(12);
    return r;
}

// Where the NEXT download of a lane starts, from the player's state at THIS download's call site (buffer_level and the
// start_up / buffer_empty flags at tick k) and the download's length -- for the download side of the role-split kernels,
// which otherwise speculates "max(completing tick + 1, avail_next): not gated by buffer_full" (Simulator.py:143-145) and
// repeats the download when the player says otherwise.  A lane whose buffer sits near max_buffer is gated at almost every
// decision, and a workgroup is as slow as its slowest lane (round 5: ONE such lane made its workgroup, and with it the whole
// launch, 30 % longer).  Covers the steady playing state at one play speed for all lanes -- the only state in which
// buffer_level can reach max_buffer -- by running exactly what lanej_after_download and lanej_wait_call do to
// (buffer_level, buffer_full, k): the same chains on the same values, hence the same tick.  Returns false when the state is
// not covered (start-up, empty buffer, the buffer running dry, max_ticks in reach, per-lane speeds): the caller keeps its
// speculation, and the player's validation of the download's start tick stays the arbiter either way.
ABR_HD bool lanej_gate_possible(double buf, bool su, bool be, int32_t n_dl, const Tables &t) {
    // buffer_full at the completing tick needs buffer_level - (n_dl - 1) * sd + L - sd >= max_buffer; real arithmetic with a
    // margin far above the chains' rounding (<= 1e-9 over an episode), far below one tick's sd
    return !su && !be && !t.per_lane_speed && (buf + t.L) - (double)n_dl * t.sd >= t.max_buffer - 1.0e-6;
}
ABR_HD bool lanej_predict_next_call(double buf, int32_t k, int32_t n_dl, int32_t avail_next, const Tables &t,
                                    int32_t &k_next, double *buf_next = nullptr) {
    const int32_t mt = t.max_ticks;
    double b = buf;
    int32_t a = 0;
    // the ticks before the completing one (lanej_idle, playing branch)
    if (n_dl > 1 && lanej_drain(t, b, t.sd, n_dl - 1, a)) return false;
    // the completing tick (lanej_after_download): :170, :184, :190, :194
    b = b + t.L;
    b = b - t.sd;
    if (b <= 0.0) return false;
    bool bf = b >= t.max_buffer;
    k += n_dl;
    if (k >= mt) return false;
    // lanej_wait_call
    if (!(k >= avail_next && !bf)) {
        int32_t w = avail_next - k;
        if (w < 0) w = 0;
        if (w > mt - k) w = mt - k;
        if (w > 0) {
            if (lanej_drain(t, b, t.sd, w, a)) return false;
            bf = b >= t.max_buffer;
            k += w;
        }
        if (k >= mt) return false;
        if (bf) {
            a = 0;
            if (!chain<STOP_LT>(b, -t.sd, t.max_buffer, mt - k, a)) return false;
            k += a;
            if (b <= 0.0) return false;
        }
    }
    k_next = k;
    if (buf_next) *buf_next = b;               // buffer_level at that call site (the lane is still playing: b > 0)
    return true;
}

// One decision in one thread, after lanej_begin_step: download a chunk of target_size, then
// run to the next call site.
ABR_HD StepResult lanej_download_and_wait(LaneJ &s, const Tables &t, const StepStart &st,
                                          double target, int32_t action) {
    const Download d = lanej_download(s.cur, t, st, s.k, target);
    return lanej_after_download(s, t, d, st.avail_next, action);
}

// begin + download + wait in one call (host harness)
ABR_HD StepResult lanej_step(LaneJ &s, const Tables &t, double target, int32_t action) {
    const StepStart st = lanej_begin_step(s.cur, t, s.k, s.chunk_id);
    return lanej_download_and_wait(s, t, st, target, action);
}

}  // namespace abrx
#endif
