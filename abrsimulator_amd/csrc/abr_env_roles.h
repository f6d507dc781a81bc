// abr_env_roles.h -- K1 in role-split form: the waves of a workgroup share 64 lanes, each wave runs ONE role of a
// decision, one iteration behind the previous role.  Included by abr_env.hip.
//
//   env_split3_kernel  D | P | S   three waves (impl 5; what impl 3 = auto runs up to kSplit3MaxLanes lanes)
//   env_split_kernel   D | P+S     two waves   (impl 2; what auto runs from 65 537 up to kSplitMaxLanes = 131 072 lanes)
//
//   D  download  a pure function of (call-site tick, trace cursor, target size): Simulator.py:152-163
//   P  player    buffer_level / play_time / counters over the download's ticks, the completing tick, the wait for the
//                next call site: Simulator.py:137-149,166-202
//   S  service   everything a decision writes to global memory: bandwidth = size / time (:164), history (:165), the
//                per-step split of calculate_qoe (:79-86), done, the observation, the episode end
//
// Why roles (measured on MI355X, profiles/r02_valu_cost_microbench.txt): ONE wave on a SIMD issues a vector instruction
// every 4.1-4.5 cycles (float64: 5.4-6.4) however independent its instructions are; 65 536 lanes are 1 024 waves = one
// per SIMD with one thread per lane, which leaves more than half of every SIMD's issue slots empty.  The only thing D
// needs from P is the next call-site tick, which is max(completion tick + 1, avail_tick[chunk + 1]) unless buffer_full
// gates the download (Simulator.py:144).  D therefore SPECULATES "not gated"; P publishes its true call-site tick every
// iteration and accepts a download record only if it started at exactly that tick.  A mis-speculated lane repeats that one
// download from the cursor it started from, in the next iteration; nothing is ever rolled back in P, which only consumes
// validated records.  Mailboxes live in LDS, double-buffered by iteration parity.
//
// Barrier discipline: the iteration loop and its ONE workgroup barrier are written once, in the kernel body; a role is a
// set of plain functions (begin / one iteration's work / end) called from wave-uniform branches (the role index comes
// out of readfirstlane, so the compiler knows it is uniform).  Every wave therefore reaches the same textual
// __syncthreads() the same number of times, and all leave in the same iteration: the exit flag is written by P before
// the barrier and read by all after it.
//
// Workspace layout and results are those of the one-thread-per-lane kernels, bit for bit.
#ifndef ABR_ENV_ROLES_H
#define ABR_ENV_ROLES_H

// (output stores go through ABR_OUT, abr_env.hip: non-temporal)

// issue priorities of the three roles (s_setprio; A/B knobs: profiles/r03_ab_split3.txt, r05_experiments_not_kept.txt (3))
// the three-wave kernel's drains: the general chains (0) or the per-binade cascade (1); see make_tables
#ifndef ABR_SPLIT3_CASCADE
#define ABR_SPLIT3_CASCADE 0
#endif
#ifndef ABR_PRIO_D
#define ABR_PRIO_D 2
#endif
#ifndef ABR_PRIO_P
#define ABR_PRIO_P 1
#endif
#ifndef ABR_PRIO_S
#define ABR_PRIO_S 0
#endif

struct SplitMail {
    // D -> P, double-buffered by iteration parity
    double dl[2][64];
    int32_t n_dl[2][64], k_start[2][64], step[2][64], action[2][64], avail_next[2][64], flags[2][64];
    // P -> D, double-buffered by iteration parity
    int32_t fb_step[2][64], fb_k[2][64], fb_chunk[2][64], fb_episode[2][64], fb_alive[2][64];
    // ... and the player's buffer_level / start_up | buffer_empty << 1 at that call site: what D needs to know EXACTLY where
    // the download after next starts when buffer_full is in reach (abr_lane_jump.h: lanej_predict_next_call)
    double fb_buf[2][64];
    int32_t fb_pf[2][64];
    int32_t any_alive[2];
};
// three-wave kernel only: the policy's draws for launch steps [act_hi - 64, act_hi), made ahead by the service wave;
// act[step % 64][lane]; act_hi is published AFTER the bytes (lds_st) and read BEFORE them (lds_ld)
struct ActRing {
    uint8_t act[64][64];
    int32_t act_hi;
};
constexpr int kRecValid = 1, kRecHit = 2, kRecBadAct = 4;

struct SplitMail2 {                      // P -> S (three-wave kernel), double-buffered by iteration parity
    double dl[2][64], buf[2][64], lat[2][64], pt[2][64];
    int32_t meta[2][64], step[2][64], n_dl[2][64], k[2][64], nplay_o[2][64], nrb_o[2][64], nsu_o[2][64],
        nrb_r[2][64], nsu_r[2][64];
};
constexpr int kS3Valid = 0x10000, kS3Hit = 0x100, kS3Bad = 0x200, kS3Ended = 0x400, kS3Timeout = 0x800,
              kS3Reset = 0x1000, kS3Timeout2 = 0x2000;

// LDS words one wave writes while another reads them inside the same iteration (no barrier in between).
// A wave's DS instructions are issued in order, but that does NOT make an earlier full-wave write visible before a
// later single-lane one (round 5, measured with a tightly polling reader: profiles/r05_ab_ring.txt): the writer waits
// for its own LDS traffic (lds_writes_done) between the data and the counter that vouches for it, the reader branches
// on the counter before it loads the data.  Relaxed atomics for the counter; the compiler must not move either side.
__device__ __forceinline__ int32_t lds_ld(const int32_t *w) {
    return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_st(int32_t *w, int32_t v) {
    __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#define ABR_LDS_ORDER() asm volatile("" ::: "memory")
__device__ __forceinline__ void lds_writes_done() {
    ABR_LDS_ORDER();
    __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0), everything else untouched: this wave's LDS writes have been performed
    ABR_LDS_ORDER();
}

// The kernel's EnvParams argument, read afresh from the kernarg segment (it is the first argument of both kernels).
// Why: with ONE loop for all roles every loop-invariant uniform value any role needs -- some 130 scalar registers of
// pointers and constants -- would be live across the whole loop and spill; re-reading the block behind a compiler-only
// fence keeps each role's scalar loads inside that role's part of the iteration (a dozen s_load per iteration).
// CONTRACT: `EnvParams p` must stay the FIRST by-value argument of env_split3_kernel and env_split_kernel (the block is read
// from offset 0 of the kernarg segment).  The PRODUCT build can be asked: abr_debug_selfcheck launches instance <9> of both
// kernel templates -- the same signature, a body that only answers -- with a sentinel in the block, and each says whether
// fresh_params() saw it (tests/test_env_gpu.py); the diagnostic stamps build also traps on a mismatch in every launch.
// (A run-time branch on a field of the block at the top of the product instances cost the three-wave kernel 36 B of private
// segment: one more live scalar than it has registers for.)
__device__ __forceinline__ const EnvParams &fresh_params() {
    auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return *(const EnvParams *)kp;
}

// abr_debug_selfcheck (abr_env.hip): is the block fresh_params() reads this launch's own?  `p` is the kernel's by-value copy.
__device__ __forceinline__ void selfcheck_answer(const EnvParams &p) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const EnvParams &f = fresh_params();
        *p.selfcheck_out = (f.sentinel == p.sentinel && f.selfcheck_out == p.selfcheck_out && f.n_lanes == p.n_lanes &&
                            f.traces == p.traces && f.G == p.G) ? 1u : 2u;
    }
}

// Loop-carried state.  The kernel body owns the loop, so a role's variables cannot be locals of a role loop any more.
// Kept in registers across the shared loop they are live through the OTHER roles' code as well, and the register
// allocator pays for that with copies at every conditional of every role (measured: +19 % vector instructions, -7 %
// throughput, profiles/r04_one_barrier.txt).  So the player and the service wave PARK their variables in LDS between
// iterations -- [word][lane], which ds_write2st64_b32 / ds_read2st64_b32 address with two independent registers per
// instruction -- and only the download wave's fifteen, the critical wave's, stay in registers (parking them as well
// puts an LDS round trip on the critical path: measured slower).  Only the owning wave ever touches its parking area,
// and the LDS executes a wave's accesses in order.
struct RolePark3 { uint32_t p[24][64], s[32][64]; };                  // three-wave kernel: the player's and the service wave's variables, 14 KB
struct RolePark2 { uint32_t p[40][64]; };                        // two-wave kernel: the player carries the service state too, 10 KB

struct ParkWords {
    uint32_t w[40];
    int n;
};
__device__ __forceinline__ void pw_i32(ParkWords &k, int32_t x) { k.w[k.n++] = (uint32_t)x; }
__device__ __forceinline__ void pw_i64(ParkWords &k, long long x) {
    k.w[k.n++] = (uint32_t)x; k.w[k.n++] = (uint32_t)((unsigned long long)x >> 32);
}
__device__ __forceinline__ void pw_f64(ParkWords &k, double x) { pw_i64(k, __double_as_longlong(x)); }
__device__ __forceinline__ int32_t pr_i32(const ParkWords &k, int &at) { return (int32_t)k.w[at++]; }
__device__ __forceinline__ long long pr_i64(const ParkWords &k, int &at) {
    const unsigned long long lo = k.w[at], hi = k.w[at + 1];
    at += 2;
    return (long long)(lo | (hi << 32));
}
__device__ __forceinline__ double pr_f64(const ParkWords &k, int &at) { return __longlong_as_double(pr_i64(k, at)); }
// [word][lane]: two words of a lane are 64 dwords apart, which is what ds_write2st64_b32 / ds_read2st64_b32 address
// with two independent registers per instruction -- no gathering of values into aligned register quads.  Q = words / 4.
template <int Q>
__device__ __forceinline__ void park_store(uint32_t (*area)[64], const ParkWords &k) {
    const int l = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < 4 * Q; q++) area[q][l] = k.w[q];
}
template <int Q>
__device__ __forceinline__ void park_load(uint32_t (*area)[64], ParkWords &k) {
    const int l = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < 4 * Q; q++) k.w[q] = area[q][l];
    k.n = 0;
}

__device__ inline void lanej_store_player(const LaneJ &s, const EnvParams &p, int64_t i) {
    p.buf[i] = s.buf; p.sumk[i] = s.sumk;
    p.k[i] = s.k; p.chunk_id[i] = s.chunk_id; p.n_su[i] = s.n_su; p.n_rb[i] = s.n_rb;
    p.n_play[i] = s.n_play; p.last_action[i] = s.last_action;
    p.flags[i] = (uint8_t)((s.su ? kFlagStartUp : 0) | (s.be ? kFlagBufEmpty : 0) |
                           (s.bf ? kFlagBufFull : 0) | kFlagArmed);
    if (p.lane_speeds) { p.sd_lane[i] = s.sd; p.pt_lane[i] = s.pt; }
    if (p.lane_speeds && p.speed_rows >= 2) { p.pl_left[i] = s.pl_left; p.play_id[i] = s.play_id; p.pt_sum[i] = s.pt_sum; }
}

// =====================================================================================================================
// D: the download side of the workgroup's 64 lanes
// =====================================================================================================================
struct DVars {
    abrx::Cursor cur;
    int32_t snap_j, snap_tpos;                 // cursor before the download just issued
    int32_t d_step, d_k, d_chunk, d_ep, offset0, issued_step, issued_k;
    int32_t issued_ndl, issued_avail;          // the download just issued took this many ticks; avail_tick of the chunk after it
    bool d_alive, was_alive;
};
__device__ __forceinline__ void role_d_begin(DVars &v, const EnvParams &p) {
    const int l = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 64 + l;
    v.cur.j = 0; v.cur.tpos = 0; v.cur.tlen = 1; v.cur.trace = p.traces;
    v.snap_j = 0; v.snap_tpos = 0; v.d_step = 0; v.d_k = 0; v.d_chunk = 0; v.d_ep = 0; v.offset0 = 0;
    v.issued_step = -1; v.issued_k = -1; v.issued_ndl = 0; v.issued_avail = 0; v.d_alive = false; v.was_alive = false;
    if (i < p.n_lanes) {
        const int32_t t = p.trace_id[i];
        v.offset0 = p.offset0[i];
        v.cur.tlen = p.trace_len[t]; v.cur.trace = p.traces + p.trace_off[t];
        v.cur.j = p.j[i]; v.cur.tpos = p.tpos[i];
        v.d_k = p.k[i]; v.d_chunk = p.chunk_id[i]; v.d_ep = p.episode_no[i];
        v.d_alive = v.was_alive = !p.done[i];
    }
    ABR_STAMP_INIT();
}

// before the barrier: the download of step d_step, started at its (predicted) call site
template <int MODE, bool ACT_RING>
__device__ __forceinline__ void role_d_pre(DVars &v, const EnvParams &p, SplitMail &m, ActRing *ring,
                                           const int32_t *__restrict__ actions, int32_t *__restrict__ actions_out,
                                           int32_t n_total, uint64_t seed, int32_t t) {
    const int l = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 64 + l;
    const int32_t V = p.video_length;
    const int cb = t & 1;                      // this iteration's mailbox slot
    const abrx::Tables tb = make_tables(p);
    int32_t flags = 0;
    ABR_STAMP(0);
    if (v.d_alive && v.d_step < n_total) {
        v.snap_j = v.cur.j; v.snap_tpos = v.cur.tpos;
        // (issuing these loads one iteration ahead, before the barrier, was measured three times and lost every time:
        // two-wave kernel -1.6 %, profiles/r02_ab_prefetch.txt; three-wave kernel -3 %, profiles/r03_ab_split3.txt (6);
        // the one-barrier form of round 4 -2 %, profiles/r04_experiments_not_kept.txt (6): 133 VGPRs, and the download
        // wave's 1.4 k cycles move into the player wave's share of the SIMD)
        const abrx::StepStart st = abrx::lanej_begin_step(v.cur, tb, v.d_k, v.d_chunk);
        ABR_STAMP(1);
        int32_t a = -1;
        bool drawn = false;
        if (MODE == 1) a = actions[i];
        else if (MODE == 3) a = actions[(int64_t)v.d_step * p.n_lanes + i];
        else if (ACT_RING && t > 0) {
            // drawn ahead by S?  counter first, then the byte it vouches for (S sets the counter up before the first
            // barrier, hence t > 0)
            const int32_t hi = lds_ld(&ring->act_hi);
            ABR_LDS_ORDER();
            drawn = v.d_step < hi && v.d_step >= hi - 64;
            if (drawn) a = ring->act[v.d_step & 63][l];
        }
        if (MODE == 2 && !drawn)
            a = (int32_t)philox_action(seed, (uint64_t)(p.lane_id_base + i), (uint32_t)v.d_chunk,
                                       (uint32_t)v.d_ep, (uint32_t)p.n_rates);
        if (MODE == 2 && actions_out) actions_out[(int64_t)v.d_step * p.n_lanes + i] = a;
        flags = kRecValid;
        abrx::Download d; d.dl = 0.0; d.n_dl = 0; d.hit = false;
        ABR_STAMP(2);
        if (a < 0 || a >= p.n_rates) flags |= kRecBadAct;
        else d = abrx::lanej_download(v.cur, tb, st, v.d_k,
                                      chunk_bitrate(p, v.d_chunk, a) * p.chunk_length /* :156 */);
        ABR_STAMP(3);
        if (d.hit) flags |= kRecHit;
        m.dl[cb][l] = d.dl; m.n_dl[cb][l] = d.n_dl; m.k_start[cb][l] = v.d_k;
        m.step[cb][l] = v.d_step; m.action[cb][l] = a; m.avail_next[cb][l] = st.avail_next;
        v.issued_step = v.d_step; v.issued_k = v.d_k;
        v.issued_ndl = d.hit ? d.n_dl : 0; v.issued_avail = st.avail_next;
        // ---- where the NEXT download starts, if nothing gates it (role_d_validate settles the rest) ----
        if (!d.hit) v.d_alive = false;           // bad action or max_ticks: the player retires the lane
        else {
            v.d_step++;
            v.d_chunk++;
            v.d_k = max(v.d_k + d.n_dl, st.avail_next);     // completing tick + 1, or availability (:143)
            if (v.d_chunk >= V) {
                if (p.auto_reset) {            // a fresh episode: clock, cursor and chunk ids restart
                    v.d_chunk = 0; v.d_ep++; v.d_k = tb.avail_tick[0];
                    abrx::cursor_init(v.cur, v.offset0);
                } else v.d_alive = false;
            }
            // a call site at or past max_ticks never happens (the player times the lane out,
            // and avail_tick is INT_MAX past the table): nothing to download speculatively
            if (v.d_k >= tb.max_ticks) v.d_alive = false;
        }
    }
    m.flags[cb][l] = flags;
    ABR_STAMP(4);
}

// Validate the record issued in iteration t against the player's true call site, which P published before that
// iteration's barrier (slot t & 1 stays intact until P's iteration t + 2): first thing in iteration t + 1.
__device__ __forceinline__ void role_d_validate(DVars &v, SplitMail &m, const abrx::Tables &tb, int32_t t) {
    const int l = threadIdx.x & 63;
    const int cb = t & 1;
    // everything the common cases look at in ONE round of LDS reads: this wave is on the iteration's critical path, and a read
    // issued only inside the branch that needs it is a round trip of its own (round 5: this and the same on the player's
    // side, +3.2 % at fuse 48, +2.2 % at fuse 20: profiles/r05_ab_fixed_costs.txt)
    int32_t f_alive = m.fb_alive[cb][l], f_step = m.fb_step[cb][l], f_k = m.fb_k[cb][l], f_pf = m.fb_pf[cb][l];
    double f_buf = m.fb_buf[cb][l];
    // "all five, now": without this the compiler sinks each read into the branch that uses it, a round trip apiece.  Only together
    // with the same on the player's side does it show (+1.2 %; either alone -1 %: the two waves are level, profiles/r05_ab_fixed_costs.txt (5))
    asm volatile("" : "+v"(f_alive), "+v"(f_step), "+v"(f_k), "+v"(f_pf), "+v"(f_buf));
    if (!f_alive) v.d_alive = false;
    else if (v.issued_step != f_step || v.issued_k != f_k) {
        // gated by buffer_full: take the player's word for where the download starts and
        // redo it from the cursor it started from
        v.d_alive = true;
        v.d_step = f_step; v.d_k = f_k;
        v.d_chunk = m.fb_chunk[cb][l]; v.d_ep = m.fb_episode[cb][l];
        if (v.issued_step == v.d_step) { v.cur.j = v.snap_j; v.cur.tpos = v.snap_tpos; }
        v.issued_step = -1;
    } else if (v.d_alive && v.issued_ndl > 0 && v.d_chunk > 0) {
        // (Having the PLAYER flag "within chunk_length of max_buffer" in a spare bit of fb_alive, so that this wave reads
        // nothing more in the common case, lost 0.8 %: the player wave is as critical as this one.  profiles/r05_experiments_not_kept.txt)
        // The record issued in iteration t started where the player really was, and the player's buffer at that call site is
        // known now: if buffer_full is in reach at the completing tick, compute EXACTLY where the next download -- the one
        // this iteration is about to start -- begins, instead of speculating "not gated" and repeating it.  (A lane whose
        // buffer sits at max_buffer is gated at almost every decision; round 5: one such lane made its workgroup, and with
        // it the whole launch, 30 % longer.  d_chunk > 0: not across an episode end, whose first call site is never gated.)
        const int32_t pf = f_pf;
        const double buf = f_buf;
        if (abrx::lanej_gate_possible(buf, pf & 1, pf & 2, v.issued_ndl, tb)) {
            int32_t kn;
            if (abrx::lanej_predict_next_call(buf, v.issued_k, v.issued_ndl, v.issued_avail, tb, kn)) {
                v.d_k = kn;
                if (kn >= tb.max_ticks) v.d_alive = false;
            }
        }
    }
}

__device__ __forceinline__ void role_d_end(const DVars &v, const EnvParams &p) {
    const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    ABR_STAMP_FLUSH();
    if (i < p.n_lanes && v.was_alive) { p.j[i] = v.cur.j; p.tpos[i] = v.cur.tpos; }
}

// =====================================================================================================================
// P: the player side.  PVars is the lane's LaneJ plus the step bookkeeping; the two-wave kernel's player also carries
// the service state (SVars below) because it runs the service tail itself.
// =====================================================================================================================
struct PVars {
    LaneJ s;
    int32_t episode_no, b_step;
    bool b_alive, was_done;
};
__device__ __forceinline__ void p_words(ParkWords &k, const PVars &v) {
    const LaneJ &s = v.s;
    pw_f64(k, s.buf); pw_f64(k, s.sd); pw_f64(k, s.pt); pw_f64(k, s.pt_sum); pw_i64(k, s.sumk);
    pw_i32(k, s.k); pw_i32(k, s.chunk_id); pw_i32(k, s.n_su); pw_i32(k, s.n_rb); pw_i32(k, s.n_play);
    pw_i32(k, s.avail_k); pw_i32(k, s.last_action); pw_i32(k, s.pl_left); pw_i32(k, s.play_id);
    pw_i32(k, (s.su ? 1 : 0) | (s.be ? 2 : 0) | (s.bf ? 4 : 0) | (v.b_alive ? 8 : 0) | (v.was_done ? 16 : 0));
    pw_i32(k, v.episode_no); pw_i32(k, v.b_step);
    pw_i32(k, 0); pw_i32(k, 0);
}
__device__ __forceinline__ void p_unwords(const ParkWords &k, int &at, PVars &v, const EnvParams &p) {
    LaneJ &s = v.s;
    s.buf = pr_f64(k, at); s.sd = pr_f64(k, at); s.pt = pr_f64(k, at); s.pt_sum = pr_f64(k, at); s.sumk = pr_i64(k, at);
    s.k = pr_i32(k, at); s.chunk_id = pr_i32(k, at); s.n_su = pr_i32(k, at); s.n_rb = pr_i32(k, at);
    s.n_play = pr_i32(k, at); s.avail_k = pr_i32(k, at); s.last_action = pr_i32(k, at); s.pl_left = pr_i32(k, at);
    s.play_id = pr_i32(k, at);
    const int32_t fl = pr_i32(k, at);
    s.su = fl & 1; s.be = fl & 2; s.bf = fl & 4; v.b_alive = fl & 8; v.was_done = fl & 16;
    v.episode_no = pr_i32(k, at); v.b_step = pr_i32(k, at);
    at += 2;
    s.cur.j = 0; s.cur.tpos = 0; s.cur.tlen = 1; s.cur.trace = p.traces;      // the player never walks the trace
    s.lane = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
}
__device__ __forceinline__ void p_park(uint32_t (*area)[64], const PVars &v) {
    ParkWords k; k.n = 0;
    p_words(k, v);
    park_store<6>(area, k);
}
__device__ __forceinline__ void p_unpark(uint32_t (*area)[64], PVars &v, const EnvParams &p) {
    ParkWords k; park_load<6>(area, k);
    int at = 0;
    p_unwords(k, at, v, p);
}

__device__ __forceinline__ double player_latency(const EnvParams &p, const LaneJ &s) {
    return !p.lane_speeds ? lane_avg_latency(p, s.sumk, s.n_play)
           : (p.speed_rows >= 2 ? avg_latency_sched(s.pt, s.sumk, s.pt_sum, s.n_play)
                                : avg_latency_from(s.sd, s.pt, s.sumk, s.n_play));
}

// tell the download side where this lane really is, and all waves whether anything is left to do
__device__ __forceinline__ void player_feedback(SplitMail &m, const PVars &v, int32_t n_total, int cb) {
    const int l = threadIdx.x & 63;
    const bool more = v.b_alive && v.b_step < n_total;
    m.fb_step[cb][l] = v.b_step; m.fb_k[cb][l] = v.s.k; m.fb_chunk[cb][l] = v.s.chunk_id;
    m.fb_episode[cb][l] = v.episode_no; m.fb_alive[cb][l] = more ? 1 : 0;
    m.fb_buf[cb][l] = v.s.buf; m.fb_pf[cb][l] = (v.s.su ? 1 : 0) | (v.s.be ? 2 : 0);
    const bool any = __any(more) != 0;
    if (l == 0) m.any_alive[cb] = any ? 1 : 0;
}

// a lane out of range holds zeros and is not alive
__device__ __forceinline__ void player_clear(PVars &v, const EnvParams &p) {
    LaneJ &s = v.s;
    s.buf = 0.0; s.sd = 0.0; s.pt = 0.0; s.pt_sum = 0.0; s.sumk = 0;
    s.k = 0; s.chunk_id = 0; s.n_su = 0; s.n_rb = 0; s.n_play = 0; s.avail_k = 0; s.last_action = -1;
    s.pl_left = 0; s.play_id = 0; s.su = false; s.be = false; s.bf = false;
    s.cur.j = 0; s.cur.tpos = 0; s.cur.tlen = 1; s.cur.trace = p.traces;      // the player never walks the trace
    s.lane = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    v.episode_no = 0; v.b_step = 0; v.b_alive = false; v.was_done = true;
}

__device__ __forceinline__ void role_p3_begin(PVars &v, const EnvParams &p) {
    const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    player_clear(v, p);
    if (i < p.n_lanes) {
        v.was_done = p.done[i] != 0;
        lanej_load(v.s, p, i);
        v.episode_no = p.episode_no[i];
        v.b_alive = !v.was_done;
    }
    ABR_STAMP_INIT();
}

// Three-wave kernel, before the barrier: the player side of the step D finished in the previous iteration; the
// finished step goes to S through m2.
// (M2 / cb2: where the finished step goes for S -- SplitMail2 slot t & 1 under the barrier; a deeper ring in the diagnostic
// pair kernel, tools/diag/csrc/abr_env_pair.h)
template <int MODE, class M2>
__device__ __forceinline__ void role_p3_pre(PVars &v, const EnvParams &p, SplitMail &m, M2 &m2,
                                            int32_t n_total, int32_t t, int cb2) {
    const int l = threadIdx.x & 63;
    const int cb = t & 1, pb = (t + 1) & 1;    // this iteration's / the previous one's slot
    const abrx::Tables tb = make_tables(p, ABR_SPLIT3_CASCADE != 0);
    const bool speeds = p.lane_speeds != nullptr;
    LaneJ &s = v.s;
    int32_t meta = 0;
    ABR_STAMP(8);
    if (v.b_alive && v.b_step < n_total && t >= 1) {
        int32_t fl = m.flags[pb][l];
        // the whole record in ONE round of LDS reads, not a second one behind the validity test (this wave is as critical as
        // the download wave)
        int32_t r_step = m.step[pb][l], r_k = m.k_start[pb][l], r_a = m.action[pb][l], r_ndl = m.n_dl[pb][l],
                r_avail = m.avail_next[pb][l];
        double r_dl = m.dl[pb][l];
        asm volatile("" : "+v"(fl), "+v"(r_step), "+v"(r_k), "+v"(r_a), "+v"(r_ndl), "+v"(r_avail), "+v"(r_dl));   // all seven, now (see role_d_validate)
        // accept the download only if it started at exactly this lane's call-site tick
        if ((fl & kRecValid) && r_step == v.b_step && r_k == s.k) {
            const int32_t a = r_a;
            meta = kS3Valid | (a & 0xff);
            m2.step[cb2][l] = v.b_step;
            if (fl & kRecBadAct) {
                meta |= kS3Bad;
                v.b_alive = false;
            } else {
                abrx::Download d;
                d.dl = r_dl; d.n_dl = r_ndl; d.hit = (fl & kRecHit) != 0;
                const abrx::StepResult sr = abrx::lanej_after_download(s, tb, d, r_avail, a);
                if (sr.hit) meta |= kS3Hit;
                if (sr.ended) meta |= kS3Ended;
                if (sr.timeout) meta |= kS3Timeout;
                m2.dl[cb2][l] = d.dl; m2.n_dl[cb2][l] = d.n_dl;
                m2.nrb_r[cb2][l] = s.n_rb; m2.nsu_r[cb2][l] = s.n_su;
                if (sr.ended || sr.timeout) {
                    m2.lat[cb2][l] = player_latency(p, s);
                    if (p.auto_reset && sr.ended) {
                        // re-arm: this step's observation is the new episode's first call site
                        abrx::lanej_init_player(s, tb);
                        v.episode_no++;
                        meta |= kS3Reset;
                        if (!abrx::lanej_wait_call(s, tb)) { meta |= kS3Timeout2; v.b_alive = false; }
                    } else v.b_alive = false;
                }
                m2.buf[cb2][l] = s.buf; m2.k[cb2][l] = s.k; m2.nplay_o[cb2][l] = s.n_play;
                m2.nrb_o[cb2][l] = s.n_rb; m2.nsu_o[cb2][l] = s.n_su;
                if (speeds) m2.pt[cb2][l] = s.pt;
            }
            v.b_step++;
        }
    }
    ABR_STAMP(13);
    m2.meta[cb2][l] = meta;
    player_feedback(m, v, n_total, cb);
    ABR_STAMP(17);
}

__device__ __forceinline__ void role_p3_end(const PVars &v, const EnvParams &p) {
    const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    ABR_STAMP_FLUSH();
    if (i < p.n_lanes && !v.was_done) lanej_store_player(v.s, p, i);
}

// =====================================================================================================================
// S: the service side -- everything a decision writes to global memory
// =====================================================================================================================
struct SVars {
    double last_bw, hist_n, hist_s, g_su_obs, g_rb_obs;
    double var_run;                                         // the running episode's sum of |br[a_i] - br[a_(i+1)]| (:82)
    double o_buf, o_pt;                                     // what an observation of this lane shows right now ...
    int32_t o_chunk, o_last, o_k, o_nplay, o_nrb, o_nsu;    // ...
    int32_t n_su_obs, n_rb_obs, episode_no, s_next;
    int32_t a_next, a_chunk, a_ep;                          // next step to draw ahead, and its (chunk, episode)
    int32_t last_cb;
    uint8_t done;
    bool was_done;
};
__device__ __forceinline__ void s_park(uint32_t (*area)[64], const SVars &v) {
    ParkWords k; k.n = 0;
    pw_f64(k, v.last_bw); pw_f64(k, v.hist_n); pw_f64(k, v.hist_s); pw_f64(k, v.g_su_obs); pw_f64(k, v.g_rb_obs);
    pw_f64(k, v.o_buf); pw_f64(k, v.o_pt);
    pw_i32(k, v.o_chunk); pw_i32(k, v.o_last); pw_i32(k, v.o_k); pw_i32(k, v.o_nplay); pw_i32(k, v.o_nrb);
    pw_i32(k, v.o_nsu); pw_i32(k, v.n_su_obs); pw_i32(k, v.n_rb_obs); pw_i32(k, v.episode_no); pw_i32(k, v.s_next);
    pw_i32(k, v.a_next); pw_i32(k, v.a_chunk); pw_i32(k, v.a_ep); pw_i32(k, v.last_cb);
    pw_i32(k, (int32_t)v.done | (v.was_done ? 0x100 : 0));
    pw_f64(k, v.var_run); pw_i32(k, 0);
    park_store<8>(area, k);
}
__device__ __forceinline__ void s_unpark(uint32_t (*area)[64], SVars &v) {
    ParkWords k; park_load<8>(area, k);
    int at = 0;
    v.last_bw = pr_f64(k, at); v.hist_n = pr_f64(k, at); v.hist_s = pr_f64(k, at); v.g_su_obs = pr_f64(k, at);
    v.g_rb_obs = pr_f64(k, at); v.o_buf = pr_f64(k, at); v.o_pt = pr_f64(k, at);
    v.o_chunk = pr_i32(k, at); v.o_last = pr_i32(k, at); v.o_k = pr_i32(k, at); v.o_nplay = pr_i32(k, at);
    v.o_nrb = pr_i32(k, at); v.o_nsu = pr_i32(k, at); v.n_su_obs = pr_i32(k, at); v.n_rb_obs = pr_i32(k, at);
    v.episode_no = pr_i32(k, at); v.s_next = pr_i32(k, at); v.a_next = pr_i32(k, at); v.a_chunk = pr_i32(k, at);
    v.a_ep = pr_i32(k, at); v.last_cb = pr_i32(k, at);
    const int32_t fl = pr_i32(k, at);
    v.done = (uint8_t)(fl & 0xff); v.was_done = (fl & 0x100) != 0;
    v.var_run = pr_f64(k, at);
}

__device__ __forceinline__ void service_write_obs(const SVars &v, const EnvParams &p, int64_t i, float *obs) {
    if (!obs) return;
    const int64_t n = p.n_lanes;
    ABR_OUT(obs[ABR_OBS_CHUNK_ID * n + i], (float)v.o_chunk);
    ABR_OUT(obs[ABR_OBS_LAST_BITRATE * n + i], (float)v.o_last);
    ABR_OUT(obs[ABR_OBS_LAST_BANDWIDTH * n + i], (float)v.last_bw);
    ABR_OUT(obs[ABR_OBS_BUFFER_LEVEL * n + i], (float)v.o_buf);
    ABR_OUT(obs[ABR_OBS_GLOBAL_TIME * n + i], (float)p.G[v.o_k]);
    ABR_OUT(obs[ABR_OBS_PLAY_TIME * n + i], (float)(p.lane_speeds ? v.o_pt : p.GP[v.o_nplay]));
    ABR_OUT(obs[ABR_OBS_REBUFFER_TIME * n + i], (float)p.G[v.o_nrb]);
    ABR_OUT(obs[ABR_OBS_STARTUP_TIME * n + i], (float)p.G[v.o_nsu]);
}

// the record P left in slot `sl`: division, history, reward, done, observation, episode end
// (M2: SplitMail2, or the ring kernel's RingPS -- the same fields with more slots)
template <class M2>
__device__ __forceinline__ void service_record(SVars &v, const EnvParams &p, M2 &m2, int sl,
                                               float *__restrict__ obs_out, float *__restrict__ reward_out,
                                               uint8_t *__restrict__ done_out) {
    const int l = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 64 + l;
    const int32_t m2m = m2.meta[sl][l];
    if (!(m2m & kS3Valid)) return;
    const int32_t step = m2.step[sl][l];
    const int64_t o = (int64_t)step * p.n_lanes + i;
    float *obs = obs_out ? obs_out + (int64_t)step * ABR_OBS_DIM * p.n_lanes : nullptr;
    const int32_t a = m2m & 0xff;
    v.s_next = step + 1;
    if (m2m & kS3Bad) {
        v.done |= ABR_DONE_BADACT;
        if (reward_out) ABR_OUT(reward_out[o], 0.0f);
        if (done_out) ABR_OUT(done_out[o], v.done);
        service_write_obs(v, p, i, obs);
        return;
    }
    const int32_t chunk = v.o_chunk, prev_action = v.o_last;
    const int32_t nrb_r = m2.nrb_r[sl][l], nsu_r = m2.nsu_r[sl][l];
    double var = 0.0;
    if (m2m & kS3Hit) {
        const double bw = m2.dl[sl][l] / p.G[m2.n_dl[sl][l]];                  // :164
        const int64_t h = (int64_t)chunk * p.n_lanes + i;
        ABR_OUT(p.bw_hist[h], bw);
        ABR_OUT(p.action_hist[h], (uint8_t)a);                                       // :165
        v.last_bw = bw;
        v.hist_s = v.hist_s + 1.0 / bw;     // sum(1/x), list order (mpc.py:86-88)
        v.hist_n = v.hist_n + 1.0;
        if (prev_action >= 0)
            var = fabs(chunk_bitrate(p, chunk, a) - chunk_bitrate(p, chunk - 1, prev_action));
        v.var_run = v.var_run + var;
        v.o_last = a; v.o_chunk = chunk + 1;
    }
    // ---- step boundary: per-step split of calculate_qoe (:83-85) ----
    const double g_rb = p.G[nrb_r], g_su = p.G[nsu_r];
    const double rew = p.wr * (g_rb - v.g_rb_obs) + p.ws * (g_su - v.g_su_obs) + p.wv * var;
    if (m2m & kS3Ended) v.done |= ABR_DONE_EPISODE;
    if (m2m & kS3Timeout) v.done |= ABR_DONE_TIMEOUT;
    if (reward_out) ABR_OUT(reward_out[o], (float)rew);
    if (done_out) ABR_OUT(done_out[o], v.done);
    v.n_su_obs = nsu_r; v.n_rb_obs = nrb_r; v.g_su_obs = g_su; v.g_rb_obs = g_rb;
    if (m2m & (kS3Ended | kS3Timeout)) {
        p.ep_qoe_terms[0 * p.n_lanes + i] = g_rb;
        p.ep_qoe_terms[1 * p.n_lanes + i] = g_su;
        p.ep_qoe_terms[2 * p.n_lanes + i] = m2.lat[sl][l];
        p.ep_qoe_terms[3 * p.n_lanes + i] = v.var_run;
        if (m2m & kS3Reset) {
            v.episode_no++;
            v.n_su_obs = 0; v.n_rb_obs = 0; v.g_su_obs = 0.0; v.g_rb_obs = 0.0;
            v.last_bw = 0.0; v.hist_n = 0.0; v.hist_s = 0.0; v.var_run = 0.0;
            v.done = (m2m & kS3Timeout2) ? ABR_DONE_TIMEOUT : 0;
            v.o_chunk = 0; v.o_last = -1;
        }
    }
    v.o_buf = m2.buf[sl][l]; v.o_k = m2.k[sl][l]; v.o_nplay = m2.nplay_o[sl][l]; v.o_nrb = m2.nrb_o[sl][l];
    v.o_nsu = m2.nsu_o[sl][l];
    if (p.lane_speeds) v.o_pt = m2.pt[sl][l];
    service_write_obs(v, p, i, obs);
}

__device__ __forceinline__ void role_s_begin(SVars &v, const EnvParams &, ActRing &ring) {
    const EnvParams &p = fresh_params();
    const int l = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 64 + l;
    v.last_bw = 0.0; v.hist_n = 0.0; v.hist_s = 0.0; v.g_su_obs = 0.0; v.g_rb_obs = 0.0; v.o_buf = 0.0; v.o_pt = 0.0; v.var_run = 0.0;
    v.o_chunk = 0; v.o_k = 0; v.o_nplay = 0; v.o_nrb = 0; v.o_nsu = 0; v.n_su_obs = 0; v.n_rb_obs = 0;
    v.episode_no = 0; v.s_next = 0; v.last_cb = 0; v.done = 0;
    v.o_last = -1; v.was_done = true;
    if (i < p.n_lanes) {
        v.done = p.done[i];
        v.was_done = v.done != 0;
        v.n_su_obs = p.n_su_obs[i]; v.n_rb_obs = p.n_rb_obs[i]; v.episode_no = p.episode_no[i];
        v.last_bw = p.last_bw[i]; v.hist_n = p.hist_n[i]; v.hist_s = p.hist_s[i]; v.var_run = p.var_run[i];
        v.g_su_obs = p.G[v.n_su_obs]; v.g_rb_obs = p.G[v.n_rb_obs];
        v.o_chunk = p.chunk_id[i]; v.o_last = p.last_action[i]; v.o_k = p.k[i]; v.o_nplay = p.n_play[i];
        v.o_nrb = p.n_rb[i]; v.o_nsu = p.n_su[i]; v.o_buf = p.buf[i];
        if (p.lane_speeds) v.o_pt = p.pt_lane[i];
    }
    v.a_next = 0; v.a_chunk = v.o_chunk; v.a_ep = v.episode_no;
    if (l == 0) lds_st(&ring.act_hi, 0);  // nothing drawn yet (D looks at the ring from its second iteration on)
    ABR_STAMP_INIT();
}

// before the barrier: draw the policy's actions ahead of D, then serve what P finished in the previous iteration
template <int MODE>
__device__ __forceinline__ void role_s_pre(SVars &v, const EnvParams &, SplitMail &m, SplitMail2 &m2, ActRing &ring,
                                           float *__restrict__ obs_out, float *__restrict__ reward_out,
                                           uint8_t *__restrict__ done_out, int32_t n_total, uint64_t seed, int32_t t) {
    const EnvParams &p = fresh_params();
    const int l = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 64 + l;
    const int cb = t & 1, pb = (t + 1) & 1;
    ABR_STAMP(20);
    if (MODE == 2) {
        // Launch step s of a lane is chunk (chunk0 + s) of its episode sequence whatever is redone on the way, so the
        // draw of step s is known up front (this wave idles most of an iteration).  act[s % 64] may be overwritten once
        // every lane still running is past step s, i.e. steps below lo + 64 with lo = the slowest live lane's step (P's
        // fb_step, one iteration old); D never waits for an entry -- one that is not there it draws itself.
        int32_t lo = 0;
        if (t >= 1) {
            lo = m.fb_alive[pb][l] ? m.fb_step[pb][l] : 0x7fffffff;
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) { const int32_t o2 = __shfl_xor(lo, sh, 64); lo = o2 < lo ? o2 : lo; }
            if (lo == 0x7fffffff) lo = n_total;
        }
        int32_t hi = lo + 60;
        if (hi > n_total) hi = n_total;
        const int32_t count = t == 0 ? 4 : 3;
        for (int32_t q = 0; q < count && v.a_next < hi; q++) {
            const uint32_t a = philox_action(seed, (uint64_t)(p.lane_id_base + i), (uint32_t)v.a_chunk, (uint32_t)v.a_ep,
                                             (uint32_t)p.n_rates);
            ring.act[v.a_next & 63][l] = (uint8_t)a;
            v.a_next++; v.a_chunk++;
            if (v.a_chunk >= p.video_length) { v.a_chunk = 0; v.a_ep++; }
        }
        // publish: the bytes first -- performed, not just issued -- then the counter that vouches for them
        lds_writes_done();
        if (l == 0) lds_st(&ring.act_hi, v.a_next);
    }
    if (t >= 1 && i < p.n_lanes) service_record(v, p, m2, pb, obs_out, reward_out, done_out);
    v.last_cb = cb;
    ABR_STAMP(21);
}

template <int MODE>
__device__ __forceinline__ void role_s_end(SVars &v, const EnvParams &, SplitMail2 &m2, float *__restrict__ obs_out,
                                           float *__restrict__ reward_out, uint8_t *__restrict__ done_out,
                                           int32_t *__restrict__ actions_out, int32_t n_total) {
    const EnvParams &p = fresh_params();
    const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    ABR_STAMP_FLUSH();
    if (i >= p.n_lanes) return;
    service_record(v, p, m2, v.last_cb, obs_out, reward_out, done_out);       // P's last records
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
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(1, 3))) void env_split3_kernel(
    EnvParams p, const int32_t *__restrict__ actions, float *__restrict__ obs_out,
    float *__restrict__ reward_out, uint8_t *__restrict__ done_out,
    int32_t *__restrict__ actions_out, int32_t n_steps, uint64_t seed) {
    __shared__ SplitMail m;
    __shared__ SplitMail2 m2;
    __shared__ ActRing ring;
    const int32_t n_total = (MODE >= 2) ? n_steps : 1;
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform by construction
    if constexpr (MODE == 9) { selfcheck_answer(p); return; }                        // abr_debug_selfcheck's instance: nothing but the answer
#ifdef ABR_SPLIT_STAMPS
    if (fresh_params().n_lanes != p.n_lanes || fresh_params().traces != p.traces) __builtin_trap();   // fresh_params' contract
#endif
    __shared__ RolePark3 park;
    // The download wave is the critical one; priority outranks age in the SIMD's issue arbitration, so its instructions
    // go first whenever they are ready (+2.8 %, profiles/r03_ab_lds_staging.txt (4)); the ORDER D > P > S is worth 8 %
    // (profiles/r03_ab_split3.txt)
    DVars dv;                          // D's variables stay in registers (every wave sets them up: no value then
    role_d_begin(dv, p);               // depends on the role); P's and S's go through LDS between iterations
    if (role == 0) ABR_WG_TIME(0);
    ABR_WG_WHERE(role);
    if (role == 0) __builtin_amdgcn_s_setprio(ABR_PRIO_D);
    else if (role == 1) { __builtin_amdgcn_s_setprio(ABR_PRIO_P); PVars v; role_p3_begin(v, p); p_park(park.p, v); }
    else { __builtin_amdgcn_s_setprio(ABR_PRIO_S); SVars v; role_s_begin(v, p, ring); s_park(park.s, v); }
    for (int32_t t = 0;; t++) {
        if (role == 0) {
            if (t > 0) role_d_validate(dv, m, make_tables(p, ABR_SPLIT3_CASCADE != 0), t - 1);      // against what P published before the previous barrier
            role_d_pre<MODE, true>(dv, p, m, &ring, actions, actions_out, n_total, seed, t);
        } else if (role == 1) {
            PVars v; p_unpark(park.p, v, p);
            role_p3_pre<MODE>(v, p, m, m2, n_total, t, t & 1);
            p_park(park.p, v);
        } else {
            SVars v; s_unpark(park.s, v);
            role_s_pre<MODE>(v, p, m, m2, ring, obs_out, reward_out, done_out, n_total, seed, t);
            s_park(park.s, v);
        }
        __syncthreads();                       // THE barrier: every wave, every iteration, this one site
        ABR_STAMP(role == 0 ? 5 : (role == 1 ? 18 : 22));
        if (!m.any_alive[t & 1]) break;        // written by P before the barrier: identical in all waves
    }
    if (role == 0) { role_d_end(dv, p); ABR_WG_TIME(1); }
    else if (role == 1) { PVars v; p_unpark(park.p, v, p); role_p3_end(v, p); ABR_WG_TIME(2); }
    else { SVars v; s_unpark(park.s, v); role_s_end<MODE>(v, p, m2, obs_out, reward_out, done_out, actions_out, n_total); ABR_WG_TIME(3); }
}

// =====================================================================================================================
// Two-wave form: the player wave runs the service tail itself, one iteration behind D (auto: 65 537 - 131 072 lanes,
// kSplitMaxLanes in abr_env.hip: as long as its 2 waves per 64 lanes are all resident, four per SIMD)
// =====================================================================================================================
struct P2Vars {
    PVars pv;
    double last_bw, hist_n, hist_s, g_su_obs, g_rb_obs, var_run;
    int32_t n_su_obs, n_rb_obs;
    uint8_t done;
};
__device__ __forceinline__ void p2_park(uint32_t (*area)[64], const P2Vars &v) {
    ParkWords k; k.n = 0;
    p_words(k, v.pv);
    pw_f64(k, v.last_bw); pw_f64(k, v.hist_n); pw_f64(k, v.hist_s); pw_f64(k, v.g_su_obs); pw_f64(k, v.g_rb_obs);
    pw_i32(k, v.n_su_obs); pw_i32(k, v.n_rb_obs); pw_i32(k, v.done);
    pw_f64(k, v.var_run); pw_i32(k, 0);
    park_store<10>(area, k);
}
__device__ __forceinline__ void p2_unpark(uint32_t (*area)[64], P2Vars &v, const EnvParams &p) {
    ParkWords k; park_load<10>(area, k);
    int at = 0;
    p_unwords(k, at, v.pv, p);
    v.last_bw = pr_f64(k, at); v.hist_n = pr_f64(k, at); v.hist_s = pr_f64(k, at); v.g_su_obs = pr_f64(k, at);
    v.g_rb_obs = pr_f64(k, at);
    v.n_su_obs = pr_i32(k, at); v.n_rb_obs = pr_i32(k, at); v.done = (uint8_t)pr_i32(k, at);
    v.var_run = pr_f64(k, at);
}

__device__ __forceinline__ void role_p2_begin(P2Vars &v, const EnvParams &) {
    const EnvParams &p = fresh_params();
    const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    player_clear(v.pv, p);
    v.last_bw = 0.0; v.hist_n = 0.0; v.hist_s = 0.0; v.g_su_obs = 0.0; v.g_rb_obs = 0.0; v.var_run = 0.0;
    v.n_su_obs = 0; v.n_rb_obs = 0; v.done = 0;
    if (i < p.n_lanes) {
        v.done = p.done[i];
        v.pv.was_done = v.done != 0;
        lanej_load(v.pv.s, p, i);
        v.n_su_obs = p.n_su_obs[i]; v.n_rb_obs = p.n_rb_obs[i]; v.pv.episode_no = p.episode_no[i];
        v.last_bw = p.last_bw[i]; v.hist_n = p.hist_n[i]; v.hist_s = p.hist_s[i]; v.var_run = p.var_run[i];
        v.g_su_obs = p.G[v.n_su_obs]; v.g_rb_obs = p.G[v.n_rb_obs];
        v.pv.b_alive = !v.done;
    }
    ABR_STAMP_INIT();
}

template <int MODE>
__device__ __forceinline__ void role_p2_pre(P2Vars &v, const EnvParams &, SplitMail &m, float *__restrict__ obs_out,
                                            float *__restrict__ reward_out, uint8_t *__restrict__ done_out,
                                            int32_t n_total, int32_t t) {
    const EnvParams &p = fresh_params();
    const int l = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 64 + l;
    const int cb = t & 1, pb = (t + 1) & 1;    // this iteration's / the previous one's slot
    const abrx::Tables tb = make_tables(p);
    LaneJ &s = v.pv.s;
    ABR_STAMP(8);
    if (v.pv.b_alive && v.pv.b_step < n_total && t >= 1) {
        const int32_t fl = m.flags[pb][l];
        // the whole record in ONE round of LDS reads (as in role_p3_pre)
        const int32_t r_step = m.step[pb][l], r_k = m.k_start[pb][l], r_a = m.action[pb][l], r_ndl = m.n_dl[pb][l],
                      r_avail = m.avail_next[pb][l];
        const double r_dl = m.dl[pb][l];
        // accept the download only if it started at exactly this lane's call-site tick
        if ((fl & kRecValid) && r_step == v.pv.b_step && r_k == s.k) {
            const int64_t o = (int64_t)v.pv.b_step * p.n_lanes + i;
            float *obs = obs_out ? obs_out + (int64_t)v.pv.b_step * ABR_OBS_DIM * p.n_lanes : nullptr;
            const int32_t a = r_a;
            if (fl & kRecBadAct) {
                v.done |= ABR_DONE_BADACT;
                if (reward_out) ABR_OUT(reward_out[o], 0.0f);
                if (done_out) ABR_OUT(done_out[o], (uint8_t)v.done);
                write_obs_j(s, p, i, obs, v.last_bw);
                v.pv.b_alive = false;
            } else {
                abrx::Download d;
                d.dl = r_dl; d.n_dl = r_ndl; d.hit = (fl & kRecHit) != 0;
                const int32_t prev_action = s.last_action;
                const int32_t chunk = s.chunk_id;
                ABR_STAMP(9);
                const abrx::StepResult sr = abrx::lanej_after_download(s, tb, d, r_avail, a);
                ABR_STAMP(13);
                // everything the step reports from the tick tables -- the reward's two clocks and the observation's four -- in
                // ONE burst of loads, consumed after the divisions below (this wave is the two-wave kernel's critical one: two
                // rounds of dependent loads were two L2 round trips per iteration)
                const double g_rb = p.G[s.n_rb], g_su = p.G[s.n_su];
                double o_k = p.G[s.k], o_pl = p.lane_speeds ? s.pt : p.GP[s.n_play], o_rb = g_rb, o_su = g_su;
                double var = 0.0;
                if (sr.hit) {
                    const int64_t h = (int64_t)chunk * p.n_lanes + i;
                    ABR_OUT(p.bw_hist[h], sr.bw);                           // :164
                    ABR_OUT(p.action_hist[h], (uint8_t)a);                  // :165
                    v.last_bw = sr.bw;
                    v.hist_s = v.hist_s + 1.0 / sr.bw;  // sum(1/x), list order (mpc.py:86-88)
                    v.hist_n = v.hist_n + 1.0;
                    if (prev_action >= 0)
                        var = fabs(chunk_bitrate(p, chunk, a) - chunk_bitrate(p, chunk - 1, prev_action));
                    v.var_run = v.var_run + var;
                }
                // ---- step boundary: per-step split of calculate_qoe (:83-85) ----
                const double rew = p.wr * (g_rb - v.g_rb_obs) + p.ws * (g_su - v.g_su_obs) + p.wv * var;
                if (sr.ended) v.done |= ABR_DONE_EPISODE;
                if (sr.timeout) v.done |= ABR_DONE_TIMEOUT;
                if (reward_out) ABR_OUT(reward_out[o], (float)rew);
                if (done_out) ABR_OUT(done_out[o], (uint8_t)v.done);
                v.n_su_obs = s.n_su; v.n_rb_obs = s.n_rb;
                v.g_su_obs = g_su; v.g_rb_obs = g_rb;
                ABR_STAMP(14);
                if (sr.ended || sr.timeout) {
                    p.ep_qoe_terms[0 * p.n_lanes + i] = g_rb;
                    p.ep_qoe_terms[1 * p.n_lanes + i] = g_su;
                    p.ep_qoe_terms[2 * p.n_lanes + i] = player_latency(p, s);
                    p.ep_qoe_terms[3 * p.n_lanes + i] = v.var_run;
                    if (p.auto_reset && sr.ended) {
                        // re-arm: this step's obs is the new episode's first call site
                        abrx::lanej_init_player(s, tb);
                        v.pv.episode_no++;
                        v.n_su_obs = 0; v.n_rb_obs = 0; v.g_su_obs = 0.0; v.g_rb_obs = 0.0;
                        v.last_bw = 0.0; v.hist_n = 0.0; v.hist_s = 0.0; v.var_run = 0.0;
                        v.done = 0;
                        if (!abrx::lanej_wait_call(s, tb)) v.done |= ABR_DONE_TIMEOUT;
                        o_k = p.G[s.k]; o_pl = p.lane_speeds ? s.pt : p.GP[s.n_play];
                        o_rb = p.G[s.n_rb]; o_su = p.G[s.n_su];
                    }
                }
                ABR_STAMP(15);
                write_obs_vals(s, p, i, obs, v.last_bw, o_k, o_pl, o_rb, o_su);
                if (v.done) v.pv.b_alive = false;
                ABR_STAMP(16);
            }
            v.pv.b_step++;
        }
    }
    player_feedback(m, v.pv, n_total, cb);
    ABR_STAMP(17);
}

template <int MODE>
__device__ __forceinline__ void role_p2_end(const P2Vars &v, const EnvParams &, float *__restrict__ obs_out,
                                            float *__restrict__ reward_out, uint8_t *__restrict__ done_out,
                                            int32_t *__restrict__ actions_out, int32_t n_total) {
    const EnvParams &p = fresh_params();
    const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    ABR_STAMP_FLUSH();
    if (i >= p.n_lanes) return;
    if (!v.pv.was_done) {
        lanej_store_player(v.pv.s, p, i);
        p.n_su_obs[i] = v.n_su_obs; p.n_rb_obs[i] = v.n_rb_obs; p.episode_no[i] = v.pv.episode_no;
        p.last_bw[i] = v.last_bw; p.hist_n[i] = v.hist_n; p.hist_s[i] = v.hist_s; p.var_run[i] = v.var_run;
        p.done[i] = v.done;
    }
    // lanes that were already finished (or finished early) report their terminal record for the remaining steps
    for (int32_t t2 = v.pv.b_step; t2 < n_total; t2++) {
        const int64_t o = (int64_t)t2 * p.n_lanes + i;
        if (reward_out) ABR_OUT(reward_out[o], 0.0f);
        if (done_out) ABR_OUT(done_out[o], (uint8_t)v.done);
        if (MODE == 2 && actions_out) ABR_OUT(actions_out[o], (int32_t)-1);
        write_obs_j(v.pv.s, p, i, obs_out ? obs_out + (int64_t)t2 * ABR_OBS_DIM * p.n_lanes : nullptr, v.last_bw);
    }
}

template <int MODE>
__global__ __launch_bounds__(128) void env_split_kernel(
    EnvParams p, const int32_t *__restrict__ actions, float *__restrict__ obs_out,
    float *__restrict__ reward_out, uint8_t *__restrict__ done_out,
    int32_t *__restrict__ actions_out, int32_t n_steps, uint64_t seed) {
    __shared__ SplitMail m;
    const int32_t n_total = (MODE >= 2) ? n_steps : 1;
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform by construction
    if constexpr (MODE == 9) { selfcheck_answer(p); return; }                        // abr_debug_selfcheck's instance: nothing but the answer
#ifdef ABR_SPLIT_STAMPS
    if (fresh_params().n_lanes != p.n_lanes || fresh_params().traces != p.traces) __builtin_trap();   // fresh_params' contract
#endif
    __shared__ RolePark2 park;
    DVars dv;                          // as in env_split3_kernel: D's variables in registers, the player's through LDS
    role_d_begin(dv, p);
    if (role == 0) __builtin_amdgcn_s_setprio(2);
    else { P2Vars v; role_p2_begin(v, p); p2_park(park.p, v); }
    for (int32_t t = 0;; t++) {
        if (role == 0) {
            if (t > 0) role_d_validate(dv, m, make_tables(p), t - 1);
            role_d_pre<MODE, false>(dv, p, m, nullptr, actions, actions_out, n_total, seed, t);
        } else {
            P2Vars v; p2_unpark(park.p, v, fresh_params());
            role_p2_pre<MODE>(v, p, m, obs_out, reward_out, done_out, n_total, t);
            p2_park(park.p, v);
        }
        __syncthreads();                       // THE barrier: both waves, every iteration, this one site
        ABR_STAMP(role == 0 ? 5 : 18);
        if (!m.any_alive[t & 1]) break;        // written by P before the barrier: identical in both waves
    }
    if (role == 0) role_d_end(dv, p);
    else { P2Vars v; p2_unpark(park.p, v, fresh_params()); role_p2_end<MODE>(v, p, obs_out, reward_out, done_out, actions_out, n_total); }
}

#endif
