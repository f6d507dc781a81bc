// abr_env_ring.h -- K1 in role-split form WITHOUT a per-iteration workgroup barrier (round 5).  Included by abr_env.hip.
//
//   env_ring3_kernel   D | P | S   three waves per 64 lanes, the same roles and the same lane functions as
//                                  env_split3_kernel (abr_env_roles.h), coupled through K-deep LDS rings
//
// Why: in the barrier form iteration t lasts max(D_t, P_(t-1), S_(t-2)) -- the download and the player wave are level on
// average (10.8 k / 11.0 k cycles) but not step by step, and each spends ~1.5 k of an iteration's 12.5 k cycles waiting
// for the other (profiles/r04_role_stamps_split3.txt).  Here a wave waits only when its input ring is empty or its output
// ring is full; the lanes of a wave stay in lock-step (one decision per lane per iteration, so transitions run at full
// lane density -- what the asynchronous pipeline of round 3 lost on).  Priced on the host before it was written
// (tools/replay_rings.py, profiles/r05_replay_rings.txt).
//
// Protocol.  Every word of RingCtl has ONE writer (abort: anyone, one way); counters only grow.  "ctl.p_pub", "p_done" and
// the correction count below are the fields of ONE 64-bit word, RingCtl::p_state.
//   D  iteration t: waits until P has consumed record t - K (ctl.p_pub >= t - K + 1), applies P's corrections, downloads
//      step d_step of every lane from its PREDICTED call-site tick (max(completion + 1, avail_tick[chunk + 1]): "the next
//      download is not gated by buffer_full", Simulator.py:143-145), writes the record into slot t % K, then publishes
//      ctl.d_pub = t + 1 (release).
//   P  iteration u: waits for ctl.d_pub > u (acquire) and for S to have freed slot u % K2, accepts a lane's record only
//      if it is for the step the lane is at AND started at exactly the lane's call-site tick; publishes its position
//      (RingFb), the finished step for S (slot u % K2) and ctl.p_pub = u + 1.  A record for the right step that started at
//      the wrong tick (buffer_full gated the call site) is rejected and the lane's correction counter is bumped: D takes
//      P's word for (step, tick, chunk, episode), restores the trace cursor to the snapshot of the rejected record (still
//      in the ring: D cannot overwrite slot u % K before it has seen ctl.p_pub >= u + 1, and then it has seen the
//      correction) and downloads that step again.  Records D issued for later steps of the lane in between are stale: they
//      name a step the lane is not at, and P drops them.  A lane P retires (episode end without auto-reset, time-out, bad
//      action) is posted the same way with alive = 0, so that D stops downloading for it and its cursor is put back to
//      where the last accepted download left it -- the workspace ends bit-identical to the one-thread-per-lane kernel's.
//   S  iteration v: waits for ctl.p_pub > v, services slot v % K2, publishes ctl.s_pub = v + 1.
// Exit: P is the authority -- when none of its lanes has a step left it sets ctl.p_done (after its last p_pub); S leaves
// when it has serviced everything P published, D as soon as it sees p_done.
//
// Forward progress.  All waits are on monotone counters written by ONE other wave, and the wait-for graph has no cycle:
// S waits only for P (input); P waits for D (input) or S (space) -- and S never waits for space; D waits for P (space, or
// "nothing to issue": then for a correction or p_done).  D blocked on space means P has K unconsumed records, so P is
// not waiting for D; P blocked on space means S has K2 unserviced records, so S is not waiting for P.  "D has nothing to
// issue while P still wants a step" cannot arise without a posted correction (every difference between the two sides'
// view of a lane starts at a rejected record or a retired lane, and both are posted; D's own reasons to stop a lane --
// the download ran into max_ticks, the predicted call site is at or past max_ticks, the video is over -- are exactly P's
// reasons to retire it); D checks for it when it idles and raises the watchdog's flag rather than wait.  All waves of a workgroup are resident together (a workgroup is scheduled as a whole) and a
// waiting wave sleeps (s_sleep), so it does not take issue slots from the wave it waits for.  A watchdog bounds every
// wait: after kRingWatchdog polls (~0.3 s) the wave sets ctl.abort, all three leave, and every unfinished lane is
// frozen with ABR_DONE_INTERNAL -- an error the caller sees in `done`, never a hang.  It has never fired.
//
// Cross-wave visibility: see ring_ld / ring_publish below.
#ifndef ABR_ENV_RING_H
#define ABR_ENV_RING_H

#ifndef ABR_RING_K
#define ABR_RING_K 4
#endif
#ifndef ABR_RING_K2
#define ABR_RING_K2 4
#endif
constexpr int kRingK = ABR_RING_K;       // D -> P records in flight
constexpr int kRingK2 = ABR_RING_K2;     // P -> S records in flight
constexpr int kRingWatchdog = 1 << 22;   // polls of ~64-128 cycles each
// Issue priorities (s_setprio; priority outranks age in a SIMD's arbitration).  A SIMD holds one wave of each role, of
// different workgroups, and no role waits at a barrier any more, so a role that is starved throttles ITS workgroup through
// the rings: a consumer whose input ring holds kRingBoostAt records or more is what its workgroup is waiting for and
// takes priority kRingBoostTo until it has caught up (profiles/r05_ab_ring.txt).
#ifndef ABR_RING_PD
#define ABR_RING_PD 2
#endif
#ifndef ABR_RING_PP
#define ABR_RING_PP 1
#endif
#ifndef ABR_RING_PS
#define ABR_RING_PS 0
#endif
#ifndef ABR_RING_BOOST_AT
#define ABR_RING_BOOST_AT 0          // 0: static priorities
#endif
#ifndef ABR_RING_WAVES
#define ABR_RING_WAVES 3             // 4: a fourth wave that leaves at once -- one wave of each role on every SIMD
#endif
constexpr int kRingBoostAt = ABR_RING_BOOST_AT, kRingBoostTo = 3;

struct RingDP {                          // D -> P, slot = D's iteration % kRingK
    double dl[kRingK][64];
    int32_t n_dl[kRingK][64], k_start[kRingK][64], step[kRingK][64], action[kRingK][64], avail_next[kRingK][64],
        flags[kRingK][64];
    // D's own: the trace cursor before this download (where a repeat of it starts) and after it (where the lane's
    // cursor belongs if this turns out to be its last accepted record)
    int32_t snap_j[kRingK][64], snap_tpos[kRingK][64], post_j[kRingK][64], post_tpos[kRingK][64];
};
struct RingPS {                          // P -> S, slot = P's iteration % kRingK2 (the fields of SplitMail2)
    double dl[kRingK2][64], buf[kRingK2][64], lat[kRingK2][64], pt[kRingK2][64];
    int32_t meta[kRingK2][64], step[kRingK2][64], n_dl[kRingK2][64], k[kRingK2][64], nplay_o[kRingK2][64],
        nrb_o[kRingK2][64], nsu_o[kRingK2][64], nrb_r[kRingK2][64], nsu_r[kRingK2][64];
};
struct RingFb {                          // P's position after its latest iteration, per lane; seq: corrections posted,
    int32_t step[64], k[64], chunk[64], ep[64], alive[64], seq[64];      // seq: count << 8 | slot of the record << 1 | alive
    double buf[64];                      // buffer_level and start_up | buffer_empty << 1 at that call site: what D needs to
    int32_t pf[64];                      // compute EXACTLY where a download gated by buffer_full starts (lanej_predict_next_call)
};
struct RingCtl {
    int32_t d_pub;        // D: records published
    int32_t s_pub;        // S: records serviced
    // P's state in ONE 64-bit word, written and read by single DS instructions (what a reader sees is one state of P):
    //   low  = p_pub   records consumed = steps handed to S
    //   high = corr << 1 | done   corr: iterations in which P posted a correction; done: no lane has a step left
    unsigned long long p_state;
    int32_t abort;        // any: watchdog
};
__device__ __forceinline__ unsigned long long ring_pack_p(int32_t p_pub, int32_t n_corr, int32_t done) {
    return (unsigned long long)(uint32_t)p_pub | ((unsigned long long)(uint32_t)((n_corr << 1) | done) << 32);
}
struct RingP { int32_t p_pub, corr, done; };

// Counter traffic.  What MI355X does and does not promise (measured, profiles/r05_ab_ring.txt): a wave's DS instructions are
// issued in order, but a later single-lane ds_write can become visible to another wave BEFORE an earlier full-wave write of
// the same wave has been performed -- "data, then counter" without a wait tore one record in ~1 500 hand-offs under a
// tightly polling reader.  So: (1) the writer waits for its own LDS traffic (s_waitcnt lgkmcnt(0)) before it publishes a
// counter; (2) whatever must be seen together lives in ONE word written by ONE instruction (RingCtl::p_state, RingFb::seq);
// (3) a reader branches on the counter before it issues the loads the counter vouches for.  -DABR_RING_FENCES builds the
// same protocol on workgroup-scope release / acquire atomics instead (adds vmcnt(0) to every hand-off; same results).
#ifdef ABR_RING_FENCES
__device__ __forceinline__ int32_t ring_ld(const int32_t *w) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void ring_publish(int32_t *w, int32_t v) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(w, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned long long ring_ld64(const unsigned long long *w) {
    return __hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void ring_publish64(unsigned long long *w, unsigned long long v) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(w, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#else
__device__ __forceinline__ int32_t ring_ld(const int32_t *w) {
    ABR_LDS_ORDER();
    const int32_t v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ABR_LDS_ORDER();
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void ring_publish(int32_t *w, int32_t v) {
    lds_writes_done();
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ABR_LDS_ORDER();
}
__device__ __forceinline__ unsigned long long ring_ld64(const unsigned long long *w) {
    ABR_LDS_ORDER();
    const unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ABR_LDS_ORDER();
    return v;
}
__device__ __forceinline__ void ring_publish64(unsigned long long *w, unsigned long long v) {
    lds_writes_done();
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ABR_LDS_ORDER();
}
#endif
__device__ __forceinline__ RingP ring_ld_p(const RingCtl &ctl) {
    const unsigned long long w = ring_ld64(&ctl.p_state);
    RingP r;
    r.p_pub = __builtin_amdgcn_readfirstlane((int32_t)(uint32_t)w);
    const int32_t hi = __builtin_amdgcn_readfirstlane((int32_t)(uint32_t)(w >> 32));
    r.corr = hi >> 1; r.done = hi & 1;
    return r;
}

// =====================================================================================================================
// D
// =====================================================================================================================
template <int MODE>
__device__ __forceinline__ void ring_d_loop(const EnvParams &p, RingDP &m, RingFb &fb, RingCtl &ctl, ActRing &ring,
                                            const int32_t *__restrict__ actions, int32_t n_total, uint64_t seed) {
    const int l = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 64 + l;
    const int32_t V = p.video_length;
    const abrx::Tables tb = make_tables(p);
    DVars v;
    role_d_begin(v, p);
    int32_t seen_seq = 0, seen_total = 0, polls = 0;
    // An APPROXIMATE copy of the player's buffer_level and start_up / buffer_empty flags at the call site of d_step, carried
    // in float32 by D itself: D runs ahead of P, so P's state is not there when D needs to know whether buffer_full can
    // gate the download after this one (Simulator.py:143-145).  It only DETECTS that (margin 0.05 s, far above its drift);
    // when it does, D waits for P to catch up -- lock-step for that one iteration, as the barrier kernels are always --
    // and computes the exact call site from P's exact state (lanej_predict_next_call).  A wrong guess costs time, never
    // a result: P validates every download's start tick.
    float ab = 0.0f;
    bool asu = true, abe = true;
    if (i < p.n_lanes) {
        const uint8_t f0 = p.flags[i];
        ab = (float)p.buf[i]; asu = (f0 & kFlagStartUp) != 0; abe = (f0 & kFlagBufEmpty) != 0;
    }
    const float Lf = (float)tb.L, sdf = (float)tb.sd, maxf = (float)tb.max_buffer, sulf = (float)tb.start_up_length;
    ABR_STAMP(0);
    for (int32_t t = 0;; t++) {
        // ---- space: slot t % K is free once P has consumed record t - K ----
        bool leave = false;
        int32_t ct;
        for (;;) {
            const int32_t ab = ring_ld(&ctl.abort);
            const RingP ps = ring_ld_p(ctl);       // p_pub and the corrections posted up to it: one word, one state of P
            ct = ps.corr;
            if (ps.done | ab) { leave = true; break; }
            if (ps.p_pub >= t - kRingK + 1) break;
            __builtin_amdgcn_s_sleep(1);
            if (++polls > kRingWatchdog) { ring_publish(&ctl.abort, 1); leave = true; break; }
        }
        if (leave) break;
        ABR_STAMP(5);
        // ---- P's corrections: a rejected record (buffer_full gated the call site) or a retired lane ----
        if (ct != seen_total) {
            seen_total = ct;
            const int32_t sq = fb.seq[l];
            ABR_LDS_ORDER();                   // the per-lane counter before the payload it vouches for
            if (sq != seen_seq) {
                seen_seq = sq;
                // the record the correction is about is still in its slot (header).  Rejected: the cursor goes back to
                // where that download started; the lane retired: to where that download -- its last -- left it
                const int r = (sq >> 1) & (kRingK - 1);
                if (!(sq & 1)) {
                    v.d_alive = false;
                    v.cur.j = m.post_j[r][l]; v.cur.tpos = m.post_tpos[r][l];
                } else {
                    v.d_alive = true; v.d_step = fb.step[l]; v.d_k = fb.k[l]; v.d_chunk = fb.chunk[l]; v.d_ep = fb.ep[l];
                    v.cur.j = m.snap_j[r][l]; v.cur.tpos = m.snap_tpos[r][l];
                    const int32_t pf = fb.pf[l];
                    ab = (float)fb.buf[l]; asu = pf & 1; abe = pf & 2;       // the player's word for its state there, too
                }
            }
        }
        // ---- nothing to issue?  then only a correction or the end can follow ----
        if (!__any(v.d_alive && v.d_step < n_total)) {
            for (;;) {
                const int32_t ab = ring_ld(&ctl.abort);
                const RingP ps = ring_ld_p(ctl);
                if (ps.done | ab) { leave = true; break; }
                if (ps.corr != seen_total) break;
                // P has consumed everything, has posted nothing, and still wants a step?  By construction that cannot be
                // (header: forward progress); if it ever is, say so -- never resynchronise silently, never spin
                if (t > 0 && ps.p_pub == t && __any(fb.alive[l] != 0)) { ring_publish(&ctl.abort, 1); leave = true; break; }
                __builtin_amdgcn_s_sleep(2);
                if (++polls > kRingWatchdog) { ring_publish(&ctl.abort, 1); leave = true; break; }
            }
            if (leave) break;
            t--;                                                     // a correction came: look at it first
            continue;
        }
        ABR_STAMP(0);
        // ---- the download of step d_step, started at its (predicted) call site ----
        const int cb = t & (kRingK - 1);
        int32_t flags = 0;
        bool hot = false;
        int32_t hot_k = 0, hot_ndl = 0, hot_avail = 0, hot_step = -1;
        if (v.d_alive && v.d_step < n_total) {
            m.snap_j[cb][l] = v.cur.j; m.snap_tpos[cb][l] = v.cur.tpos;
            const abrx::StepStart st = abrx::lanej_begin_step(v.cur, tb, v.d_k, v.d_chunk);
            ABR_STAMP(1);
            int32_t a = -1;
            bool drawn = false;
            if (MODE == 1) a = actions[i];
            else if (MODE == 3) a = actions[(int64_t)v.d_step * p.n_lanes + i];
            else {
                // drawn ahead by S?  counter first, then the byte it vouches for
                const int32_t hi = lds_ld(&ring.act_hi);
                ABR_LDS_ORDER();
                drawn = v.d_step < hi && v.d_step >= hi - 64;
                if (drawn) a = ring.act[v.d_step & 63][l];
            }
            if (MODE == 2 && !drawn)
                a = (int32_t)philox_action(seed, (uint64_t)(p.lane_id_base + i), (uint32_t)v.d_chunk,
                                           (uint32_t)v.d_ep, (uint32_t)p.n_rates);
            flags = kRecValid;
            abrx::Download d; d.dl = 0.0; d.n_dl = 0; d.hit = false;
            ABR_STAMP(2);
            if (a < 0 || a >= p.n_rates) flags |= kRecBadAct;
            else d = abrx::lanej_download(v.cur, tb, st, v.d_k, chunk_bitrate(p, v.d_chunk, a) * p.chunk_length /* :156 */);
            ABR_STAMP(3);
            if (d.hit) flags |= kRecHit;
            m.dl[cb][l] = d.dl; m.n_dl[cb][l] = d.n_dl; m.k_start[cb][l] = v.d_k;
            m.step[cb][l] = v.d_step; m.action[cb][l] = a; m.avail_next[cb][l] = st.avail_next;
            // ---- where the NEXT download starts, if nothing gates it ----
            if (!d.hit) v.d_alive = false;           // bad action or max_ticks: the player retires the lane
            else {
                hot_k = v.d_k; hot_ndl = d.n_dl; hot_avail = st.avail_next; hot_step = v.d_step;
                // can buffer_full gate the NEXT download?  (the approximate state, before it moves on)
                hot = !asu && !abe && !tb.per_lane_speed && v.d_chunk + 1 < V &&
                      (ab + Lf) - (float)d.n_dl * sdf >= maxf - 0.05f;
                // the approximate state at the next call site: what lanej_after_download + lanej_wait_call do to it
                {
                    float b;
                    bool play = true;
                    if (asu) { b = ab + Lf; asu = b < sulf; play = !asu; }          // start-up: nothing drains (:137-138,:201)
                    else if (abe) b = Lf;                                            // rebuffering: 0 + L (:170,:194)
                    else { const float bc = ab - (float)(d.n_dl - 1) * sdf; b = bc <= 0.0f ? Lf : bc + Lf - sdf; }
                    abe = false;
                    const int32_t w = st.avail_next - (v.d_k + d.n_dl);
                    if (play && w > 0) { b -= (float)w * sdf; if (b <= 0.0f) { b = 0.0f; abe = true; } }
                    ab = b;
                }
                v.d_step++;
                v.d_chunk++;
                v.d_k = max(v.d_k + d.n_dl, st.avail_next);     // completing tick + 1, or availability (:143)
                if (v.d_chunk >= V) {
                    if (p.auto_reset) {            // a fresh episode: clock, cursor and chunk ids restart
                        v.d_chunk = 0; v.d_ep++; v.d_k = tb.avail_tick[0];
                        abrx::cursor_init(v.cur, v.offset0);
                        ab = 0.0f; asu = true; abe = true;
                    } else v.d_alive = false;
                }
                // a call site at or past max_ticks never happens (the player times the lane out)
                if (v.d_k >= tb.max_ticks) v.d_alive = false;
            }
            m.post_j[cb][l] = v.cur.j; m.post_tpos[cb][l] = v.cur.tpos;
        }
        m.flags[cb][l] = flags;
        // ---- buffer_full in reach for some lane: lock-step for this one iteration ----
        if (__any(hot)) {
            // P has consumed every record published so far (p_pub == t) <=> it stands at the call site of the download a lane
            // in step with the wave has just finished, and its published state is complete (it waits for record t now)
            for (;;) {
                const int32_t ab_ = ring_ld(&ctl.abort);
                const RingP ps = ring_ld_p(ctl);
                if (ps.done | ab_) break;
                if (ps.p_pub >= t) break;
                __builtin_amdgcn_s_sleep(1);
                if (++polls > kRingWatchdog) { ring_publish(&ctl.abort, 1); break; }
            }
            ABR_STAMP(6);
            if (hot && v.d_alive && fb.step[l] == hot_step && fb.k[l] == hot_k && fb.alive[l]) {
                const int32_t pf = fb.pf[l];
                const double pbuf = fb.buf[l];
                int32_t kn;
                double bn;
                if (abrx::lanej_gate_possible(pbuf, pf & 1, pf & 2, hot_ndl, tb) &&
                    abrx::lanej_predict_next_call(pbuf, hot_k, hot_ndl, hot_avail, tb, kn, &bn)) {
                    v.d_k = kn;
                    ab = (float)bn; asu = false; abe = false;
                    if (kn >= tb.max_ticks) v.d_alive = false;
                }
            }
        }
        ring_publish(&ctl.d_pub, t + 1);
        ABR_STAMP(4);
    }
    role_d_end(v, p);
}

// =====================================================================================================================
// P
// =====================================================================================================================
template <int MODE>
__device__ __forceinline__ void ring_p_loop(const EnvParams &p, RingDP &m, RingPS &m2, RingFb &fb, RingCtl &ctl,
                                            int32_t n_total) {
    const int l = threadIdx.x & 63;
    const abrx::Tables tb = make_tables(p);
    const bool speeds = p.lane_speeds != nullptr;
    PVars v;
    role_p3_begin(v, p);
    LaneJ &s = v.s;
    int32_t my_seq = 0, n_corr = 0, polls = 0;
    ABR_STAMP(8);
    for (int32_t u = 0;; u++) {
        // nothing left (or nothing to begin with)?  p_pub is final
        if (!__any(v.b_alive && v.b_step < n_total)) { ring_publish64(&ctl.p_state, ring_pack_p(u, n_corr, 1)); break; }
        // ---- input: D's record u;  space: slot u % K2 is free once S has serviced record u - K2 ----
        bool leave = false;
        int32_t backlog = 0;
#ifdef ABR_SPLIT_STAMPS
        bool stamp_blocked_on_s = false;
#endif
        for (;;) {
            const int32_t ab = ring_ld(&ctl.abort), dp = ring_ld(&ctl.d_pub), sp = ring_ld(&ctl.s_pub);
            backlog = dp - u;
            if (ab) { leave = true; break; }
            if (dp > u && sp >= u - kRingK2 + 1) break;
#ifdef ABR_SPLIT_STAMPS
            stamp_blocked_on_s = dp > u;
#endif
            __builtin_amdgcn_s_sleep(1);
            if (++polls > kRingWatchdog) { ring_publish(&ctl.abort, 1); leave = true; break; }
        }
        if (leave) break;
        if (kRingBoostAt > 0) {
            if (backlog >= kRingBoostAt) __builtin_amdgcn_s_setprio(kRingBoostTo);
            else __builtin_amdgcn_s_setprio(ABR_RING_PP);
        }
#ifdef ABR_SPLIT_STAMPS
        if (stamp_blocked_on_s) ABR_STAMP(19); else ABR_STAMP(18);      // diagnostic build: which ring P waited for
#endif
        const int pb = u & (kRingK - 1), cb = u & (kRingK2 - 1);
        int32_t meta = 0;
        bool post = false;
        if (v.b_alive && v.b_step < n_total) {
            const int32_t fl = m.flags[pb][l];
            if ((fl & kRecValid) && m.step[pb][l] == v.b_step) {
                if (m.k_start[pb][l] != s.k) post = true;     // gated by buffer_full: D repeats this download from s.k
                else {
                    const int32_t a = m.action[pb][l];
                    meta = kS3Valid | (a & 0xff);
                    m2.step[cb][l] = v.b_step;
                    if (fl & kRecBadAct) {
                        meta |= kS3Bad;
                        v.b_alive = false;
                    } else {
                        abrx::Download d;
                        d.dl = m.dl[pb][l]; d.n_dl = m.n_dl[pb][l]; d.hit = (fl & kRecHit) != 0;
                        const abrx::StepResult sr = abrx::lanej_after_download(s, tb, d, m.avail_next[pb][l], a);
                        if (sr.hit) meta |= kS3Hit;
                        if (sr.ended) meta |= kS3Ended;
                        if (sr.timeout) meta |= kS3Timeout;
                        m2.dl[cb][l] = d.dl; m2.n_dl[cb][l] = d.n_dl;
                        m2.nrb_r[cb][l] = s.n_rb; m2.nsu_r[cb][l] = s.n_su;
                        if (sr.ended || sr.timeout) {
                            m2.lat[cb][l] = player_latency(p, s);
                            if (p.auto_reset && sr.ended) {
                                // re-arm: this step's observation is the new episode's first call site
                                abrx::lanej_init_player(s, tb);
                                v.episode_no++;
                                meta |= kS3Reset;
                                if (!abrx::lanej_wait_call(s, tb)) { meta |= kS3Timeout2; v.b_alive = false; }
                            } else v.b_alive = false;
                        }
                        m2.buf[cb][l] = s.buf; m2.k[cb][l] = s.k; m2.nplay_o[cb][l] = s.n_play;
                        m2.nrb_o[cb][l] = s.n_rb; m2.nsu_o[cb][l] = s.n_su;
                        if (speeds) m2.pt[cb][l] = s.pt;
                    }
                    v.b_step++;
                    if (!v.b_alive) post = true;              // retired: D stops downloading for the lane
                }
            }
        }
        ABR_STAMP(13);
        m2.meta[cb][l] = meta;
        // ---- where this lane is now: the correction's payload, S's bound for drawing ahead, D's resynchronisation ----
        const bool more = v.b_alive && v.b_step < n_total;
        fb.step[l] = v.b_step; fb.k[l] = s.k; fb.chunk[l] = s.chunk_id; fb.ep[l] = v.episode_no; fb.alive[l] = more ? 1 : 0;
        fb.buf[l] = s.buf; fb.pf[l] = (s.su ? 1 : 0) | (s.be ? 2 : 0);
        // ONE word per correction: count, slot of the record it is about, alive -- read atomically by D.  (step, k, chunk,
        // ep above are the lane's position, which a rejected record did not move: the same values as an iteration ago)
        ABR_LDS_ORDER();
        if (post) fb.seq[l] = (++my_seq << 8) | (pb << 1) | (more ? 1 : 0);
        if (__any(post)) n_corr++;
        ring_publish64(&ctl.p_state, ring_pack_p(u + 1, n_corr, 0));
        ABR_STAMP(17);
    }
    role_p3_end(v, p);
}

// =====================================================================================================================
// S
// =====================================================================================================================
template <int MODE>
__device__ __forceinline__ void ring_s_loop(const EnvParams &p, RingPS &m2, RingFb &fb, RingCtl &ctl, ActRing &ring,
                                            float *__restrict__ obs_out, float *__restrict__ reward_out,
                                            uint8_t *__restrict__ done_out, int32_t *__restrict__ actions_out,
                                            int32_t n_total, uint64_t seed) {
    const int l = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 64 + l;
    SVars v;
    role_s_begin(v, p, ring);
    int32_t polls = 0;
    bool aborted = false;
    ABR_STAMP(20);
    for (int32_t w = 0;; w++) {
        if (MODE == 2) {
            // draw the policy's actions ahead of D (abr_env_roles.h: role_s_pre).  act[s % 64] may be overwritten once every
            // lane still running is past step s: steps below lo + 64, lo = the slowest live lane's step as P last published it
            // (per lane it only grows, so a torn read across lanes is merely older = smaller)
            int32_t lo = fb.alive[l] ? fb.step[l] : 0x7fffffff;
            if (w == 0) lo = 0;
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) { const int32_t o2 = __shfl_xor(lo, sh, 64); lo = o2 < lo ? o2 : lo; }
            if (lo == 0x7fffffff) lo = n_total;
            int32_t hi = lo + 60;
            if (hi > n_total) hi = n_total;
            const int32_t count = w == 0 ? 4 : 3;
            for (int32_t q = 0; q < count && v.a_next < hi; q++) {
                const uint32_t a = philox_action(seed, (uint64_t)(p.lane_id_base + i), (uint32_t)v.a_chunk, (uint32_t)v.a_ep,
                                                 (uint32_t)p.n_rates);
                ring.act[v.a_next & 63][l] = (uint8_t)a;
                v.a_next++; v.a_chunk++;
                if (v.a_chunk >= p.video_length) { v.a_chunk = 0; v.a_ep++; }
            }
            lds_writes_done();                // the bytes have been performed before the counter that vouches for them
            if (l == 0) lds_st(&ring.act_hi, v.a_next);
        }
        ABR_STAMP(20);
        // ---- input: P's record w ----
        bool leave = false;
        int32_t backlog = 0;
        for (;;) {
            const int32_t ab = ring_ld(&ctl.abort);
            const RingP ps = ring_ld_p(ctl);       // p_pub and "it is final": one word
            const int32_t dn = ps.done;
            backlog = ps.p_pub - w;
            if (ab) { leave = true; aborted = true; break; }
            if (ps.p_pub > w) break;
            if (dn) { leave = true; break; }
            __builtin_amdgcn_s_sleep(2);
            if (++polls > kRingWatchdog) { ring_publish(&ctl.abort, 1); leave = true; aborted = true; break; }
        }
        if (leave) break;
        if (kRingBoostAt > 0) {
            if (backlog >= kRingBoostAt) __builtin_amdgcn_s_setprio(kRingBoostTo);
            else __builtin_amdgcn_s_setprio(ABR_RING_PS);
        }
        ABR_STAMP(22);
        if (i < p.n_lanes) {
            const int sl = w & (kRingK2 - 1);
            if (MODE == 2 && actions_out && (m2.meta[sl][l] & kS3Valid))
                actions_out[(int64_t)m2.step[sl][l] * p.n_lanes + i] = m2.meta[sl][l] & 0xff;
            service_record(v, p, m2, sl, obs_out, reward_out, done_out);
        }
        ring_publish(&ctl.s_pub, w + 1);
        ABR_STAMP(21);
    }
    ABR_STAMP_FLUSH();
    if (i >= p.n_lanes) return;
    if (aborted && !v.done && v.s_next < n_total) v.done |= ABR_DONE_INTERNAL;      // the watchdog fired: never silently
    if (!v.was_done) {
        p.n_su_obs[i] = v.n_su_obs; p.n_rb_obs[i] = v.n_rb_obs; p.episode_no[i] = v.episode_no;
        p.last_bw[i] = v.last_bw; p.hist_n[i] = v.hist_n; p.hist_s[i] = v.hist_s; p.var_run[i] = v.var_run;
        p.done[i] = v.done;
    }
    // lanes that were already finished (or finished early) report their terminal record for the remaining steps
    for (int32_t t2 = v.s_next; t2 < n_total; t2++) {
        const int64_t o = (int64_t)t2 * p.n_lanes + i;
        if (reward_out) reward_out[o] = 0.0f;
        if (done_out) done_out[o] = v.done;
        if (MODE == 2 && actions_out) actions_out[o] = -1;
        service_write_obs(v, p, i, obs_out ? obs_out + (int64_t)t2 * ABR_OBS_DIM * p.n_lanes : nullptr);
    }
}

// MODE 1: one externally supplied action per lane; MODE 2: fused random-policy rollout; MODE 3: fused rollout of
// scripted actions [n_steps][n_lanes]
template <int MODE>
__global__ __launch_bounds__(64 * ABR_RING_WAVES) __attribute__((amdgpu_waves_per_eu(1, 3))) void env_ring3_kernel(
    EnvParams p, const int32_t *__restrict__ actions, float *__restrict__ obs_out,
    float *__restrict__ reward_out, uint8_t *__restrict__ done_out,
    int32_t *__restrict__ actions_out, int32_t n_steps, uint64_t seed) {
    __shared__ RingDP m;
    __shared__ RingPS m2;
    __shared__ RingFb fb;
    __shared__ RingCtl ctl;
    __shared__ ActRing ring;
    const int32_t n_total = (MODE >= 2) ? n_steps : 1;
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform by construction
    if (ABR_RING_WAVES == 4 && role == 3) return;       // placement only; a wave that has ended no longer counts at s_barrier
    // the ONE workgroup barrier of the kernel: the control words and the words a reader may look at before their writer
    // got there start from zero (each wave clears a share; nothing is published before this point)
    {
        const int l = threadIdx.x & 63;
        if (role == 0) {
#pragma unroll
            for (int r = 0; r < kRingK; r++) m.flags[r][l] = 0;
        } else if (role == 1) {
            fb.step[l] = 0; fb.k[l] = 0; fb.chunk[l] = 0; fb.ep[l] = 0; fb.seq[l] = 0;
            fb.alive[l] = 1;     // "at step 0" until P has published: S must not draw more than a ring ahead of it
        } else if (l == 0) {
            ctl.d_pub = 0; ctl.s_pub = 0; ctl.p_state = 0ull; ctl.abort = 0;
            ring.act_hi = 0;
        }
    }
    __syncthreads();
    ABR_WG_WHERE(role);
    if (role == 0) {
        ABR_WG_TIME(0);
        __builtin_amdgcn_s_setprio(ABR_RING_PD);
        ring_d_loop<MODE>(p, m, fb, ctl, ring, actions, n_total, seed);
        ABR_WG_TIME(1);
    } else if (role == 1) {
        __builtin_amdgcn_s_setprio(ABR_RING_PP);
        ring_p_loop<MODE>(p, m, m2, fb, ctl, n_total);
        ABR_WG_TIME(2);
    } else {
        __builtin_amdgcn_s_setprio(ABR_RING_PS);
        ring_s_loop<MODE>(p, m2, fb, ctl, ring, obs_out, reward_out, done_out, actions_out, n_total, seed);
        ABR_WG_TIME(3);
    }
}

#endif
