// abr_env_async.h -- K1 as an asynchronous three-role pipeline (impl 4), included by abr_env.hip in the
// DIAGNOSTIC build libabr_hip_async.so only (-DABR_WITH_ASYNC): measured slower than the role-split kernels, never
// picked by `auto`, not part of the product library.  Kept as evidence and as a parity-tested alternative.
//
// Same lane arithmetic as every other implementation (abr_lane_jump.h / abr_exact_jump.h,
// Simulator.py:135-208 under R1-R3) and the same workspace, bit for bit.  What changes is who
// waits for whom.  The role-split kernel above has one workgroup barrier per decision, so both of
// its waves pay the slowest of their 64 lanes every step (measured: 16.6 download segments for a
// mean of 7.2).  Here a lane's step is cut into three roles that only meet through per-lane rings
// in LDS, and nothing in the kernel waits for a whole wave:
//
//   D  download   (Simulator.py:152-170)  a FLAT loop: one chain segment per trip for every lane
//                 that is inside a download, whatever step that lane is at; lanes that finish a
//                 download park until enough of them are parked, then one PASS pushes their
//                 records, draws the next action and starts the next download for all of them.
//                 Everything a trip reads lives in LDS: the interval-end ticks, the availability
//                 ticks, the action table, and a per-lane window of the bandwidth trace that the
//                 passes refill with loads whose results are only touched one pass later -- the
//                 download loop never waits for global memory.
//   P  player     (Simulator.py:137-149,174-202)  buffer / counters over the download's ticks, the
//                 completing tick, the wait for the next call site; consumes D's records, emits
//                 one record per decision for S.  Like the role-split kernel D speculates "the
//                 next download is not gated by buffer_full" (:144); P accepts a record only if
//                 it started at exactly P's own call-site tick, and otherwise sends D back to the
//                 record's cursor snapshot with the true tick (an epoch per lane voids whatever D
//                 produced ahead of that).
//   S  service    (Simulator.py:164-165, :79-86 split per step)  bandwidth = size / time,
//                 history, reward, done, observation, episode end; also draws the policy's
//                 actions ahead of D.  Runs only when most lanes have a record waiting, so its
//                 vector instructions are nearly full.
//
// A workgroup is kAG groups of 64 lanes = 3 * kAG waves; wave w has role w / kAG on group w % kAG,
// so (waves of a workgroup go to the SIMDs in cyclic order) every SIMD holds one D, one P and one S
// wave.  No workgroup barrier after the tables are staged.
//
// Progress (no deadlock): every wait is per lane and on a monotone counter.  D(lane) waits only
// for ring-1 space (then P(lane) has input) or, once all its downloads are pushed, for P's final
// word on the lane.  P(lane) waits only for a ring-1 record (then D(lane) is not blocked: it is
// downloading, pushing, or finished -- and a finished D pushed everything) or for ring-2 space
// (then S(lane) has input).  S(lane) waits only for input.  The "run when enough lanes are ready"
// thresholds are bypassed as soon as any ring is full or a bounded patience runs out, so a
// threshold can delay, never block.
#ifndef ABR_ENV_ASYNC_H
#define ABR_ENV_ASYNC_H

constexpr int kAG = 4;                 // lane groups of 64 per workgroup
constexpr int kAW = 64 * kAG;          // lanes per workgroup
constexpr int kR1 = 4;                 // D -> P ring: records per lane
constexpr int kR2 = 2;                 // P -> S ring
constexpr int kWin = 16;               // per-lane window of the bandwidth trace (intervals)
constexpr int kStage = 8;              // bandwidths fetched per window refill
constexpr int kItickLds = 2048;        // LDS-resident prefix of interval_tick
constexpr int kAvailLds = 1024;        // LDS-resident avail_tick: video_length + 2 <= this
constexpr int kMaxFuse = 64;           // decisions per launch (rows of the action table)
#ifndef ABR_ASYNC_THETA_D
#define ABR_ASYNC_THETA_D 20           // D: parked lanes that trigger a pass
#endif
#ifndef ABR_ASYNC_THETA_P
#define ABR_ASYNC_THETA_P 40           // P: lanes with a record waiting that trigger an iteration
#endif
#ifndef ABR_ASYNC_THETA_S
#define ABR_ASYNC_THETA_S 48
#endif
#ifndef ABR_ASYNC_MAXTRIPS
#define ABR_ASYNC_MAXTRIPS 6           // D: a parked lane waits at most this many trips for a pass
#endif
constexpr int kPatience = 12;          // P / S: polls (with s_sleep) before running under the threshold
// Watchdog: a role that has polled this many times in a row without making progress (seconds of
// wall time; legitimate waits are bounded by one lane's longest download) raises sh.abort, every
// role leaves its loop, and the lanes that had not finished are frozen with ABR_DONE_INTERNAL --
// a protocol bug must end the launch loudly, never hang the GPU.
constexpr int32_t kSpinLimit = 1 << 24;

struct AsyncShared {
    int32_t itick[kItickLds];
    int32_t avail[kAvailLds];
    uint8_t act[kMaxFuse][kAW];        // action of launch-step s of lane l; 0xFF = out of range (scripted)
    double win[kWin][kAW];             // bandwidth of interval e of lane l at win[e % kWin][l]
    // ring 1, D -> P
    double r1_dl[kR1][kAW];
    int32_t r1_ndl[kR1][kAW], r1_k[kR1][kAW], r1_meta[kR1][kAW], r1_sj[kR1][kAW], r1_st[kR1][kAW];
    // ring 2, P -> S
    double r2_dl[kR2][kAW], r2_buf[kR2][kAW];
    long long r2_sumk[kR2][kAW];
    int32_t r2_ndl[kR2][kAW], r2_meta[kR2][kAW], r2_k[kR2][kAW], r2_nplay_o[kR2][kAW],
        r2_nrb_o[kR2][kAW], r2_nsu_o[kR2][kAW], r2_nrb_r[kR2][kAW], r2_nsu_r[kR2][kAW],
        r2_nplay_r[kR2][kAW];
    // per-lane monotone counters / mailboxes
    int32_t d_head[kAW];               // records D has pushed
    int32_t p_tail[kAW];               // records P has taken
    int32_t p_status[kAW];             // steps P has accepted | kPDead
    int32_t redo_epoch[kAW], redo_k[kAW], redo_step[kAW], redo_chunk[kAW], redo_ep[kAW],
        redo_j[kAW], redo_tpos[kAW];   // P -> D: "that download started at the wrong tick"
    int32_t p_head2[kAW], s_tail2[kAW];
    double lad[ABR_MAX_RATES];         // config ladder (the one Chunk run() indexes, :82,:156)
    int32_t act_ready[kAG];            // rows of `act` S has filled, per lane group
    int32_t abort;                     // watchdog (see kSpinLimit)
};

// ring-1 meta: action | flags | step << 10 | epoch << 16
constexpr int kM1Hit = 0x100, kM1Bad = 0x200;
// ring-2 meta: action | flags
constexpr int kM2Hit = 0x100, kM2Bad = 0x200, kM2Ended = 0x400, kM2Timeout = 0x800, kM2Reset = 0x1000,
              kM2Timeout2 = 0x2000;
constexpr int kPDead = 0x40000000;

// (lds_ld / lds_st / ABR_LDS_ORDER: abr_env.hip, shared with the three-wave kernel's action ring)

#ifdef ABR_ASYNC_STATS
// diagnostic build only (libabr_hip_astats.so): per-role cycle / count accumulators, lane 0 of each
// wave, flushed once at the end of the role; read with abr_debug_async_stats (tools/gpu_async_stats.py)
__device__ unsigned long long g_async_stats[48];
#define AST_DECL unsigned long long ast_[12] = {0,0,0,0,0,0,0,0,0,0,0,0}; long long ast_t_ = __builtin_amdgcn_s_memtime();
#define AST_LAP(slot) do { const long long t2_ = __builtin_amdgcn_s_memtime(); ast_[slot] += (unsigned long long)(t2_ - ast_t_); ast_t_ = t2_; } while (0)
#define AST_ADD(slot, v) do { ast_[slot] += (unsigned long long)(v); } while (0)
#define AST_FLUSH(base) do { if ((threadIdx.x & 63) == 0) for (int q_ = 0; q_ < 12; q_++) if (ast_[q_]) atomicAdd(&g_async_stats[(base) + q_], ast_[q_]); } while (0)
#else
#define AST_DECL
#define AST_LAP(slot)
#define AST_ADD(slot, v)
#define AST_FLUSH(base)
#endif

enum { DS_RUN = 0, DS_DONE = 1, DS_BEGIN = 2, DS_NEEDWIN = 3, DS_WAITWIN = 4, DS_FIN = 5, DS_EXIT = 6 };

// interval_tick[j]: from LDS, unconditionally (clamped index), and only for the rare lane past the
// staged prefix from global memory in a branch of its own -- written as `j < cap ? lds[j] : g[j]` the
// compiler selects between the two ADDRESSES and emits one flat_load, whose s_waitcnt vmcnt(0)
// lgkmcnt(0) then stalls every trip on the trace refills in flight.
__device__ __forceinline__ int32_t async_itick(const AsyncShared &sh, const int32_t *g, int32_t j) {
    int32_t v = sh.itick[j < kItickLds ? j : kItickLds - 1];
    if (j >= kItickLds) v = *(const volatile int32_t *)(g + j);     // volatile: must not be merged with the LDS load
    return v;
}

// bitrate of (chunk, rate): the one ladder from LDS, or the caller's per-chunk table
__device__ __forceinline__ double async_bitrate(const EnvParams &p, const AsyncShared &sh, int32_t chunk,
                                                int32_t rate) {
    double v = sh.lad[rate];
    if (p.br_table) v = *(const volatile double *)(p.br_table + (int64_t)chunk * p.n_rates + rate);   // as async_itick
    return v;
}

// ---------------------------------------------------------------------------------------------
// D: the download side (wave role 0)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void async_role_download(const EnvParams &p, AsyncShared &sh, const int g,
                                                    const int32_t n_total) {
    const int l = threadIdx.x & 63, wl = g * 64 + l;
    const int64_t i = (int64_t)blockIdx.x * kAW + wl;
    const bool in_range = i < p.n_lanes;
    const int32_t V = p.video_length, mt = p.max_ticks;
    const int32_t *gi = p.interval_tick;
    const double *trace = p.traces;
    int32_t tlen = 1, tpos0 = 0, j = 0, tpos = 0;
    int32_t d_step = 0, d_k = 0, d_chunk = 0;
    int state = DS_EXIT;
    bool was_alive = false;
    if (in_range) {
        const int32_t t = p.trace_id[i];
        tlen = p.trace_len[t]; trace = p.traces + p.trace_off[t];
        tpos0 = p.offset0[i] % tlen;
        j = p.j[i]; tpos = p.tpos[i];
        d_k = p.k[i]; d_chunk = p.chunk_id[i];
        was_alive = !p.done[i];
        if (was_alive && n_total > 0) state = DS_BEGIN;
    }
    // window of the trace in LDS: intervals [wlo, whi) are valid; wpos = trace position of whi
    int32_t wlo = 0, whi = 0, wpos = 0, stg_n = 0;
    double stg[kStage];
#pragma unroll
    for (int q = 0; q < kStage; q++) stg[q] = 0.0;
    // the download in flight
    double c = 0.0, bwn = 0.0, x = 0.0, target = 0.0;
    int32_t ke = 0, ken = 0, tn = 0, inb = -1 /* ChainState::eb */, n_dl = 0, kk = 0, lim = 0, k_start = 0, act = 0;
    int32_t snap_j = 0, snap_tpos = 0;
    bool hit = false, bad = false, bwn_ok = false, snapped = false, resume = false, blocked = false;
    int32_t head = 0, epoch = 0, since_pass = 0, spins = 0;
    AST_DECL

    for (;;) {
        // ======================= one trip: a chain segment per downloading lane =======================
        AST_LAP(0);
        if (__ballot(state == DS_RUN)) {
            AST_ADD(4, 1); AST_ADD(5, __popcll(__ballot(state == DS_RUN)));
            if (state == DS_RUN) {
                const bool adv = kk >= ke;
                if (adv && !bwn_ok) {
                    state = DS_NEEDWIN;                    // the next interval's bandwidth is not in the window yet
                } else {
                    // same trip as abrx::lanej_download: interval over -> its successor was prefetched
                    c = adv ? bwn * abrx::kTickDt : c;
                    ke = adv ? ken : ke;
                    j += adv ? 1 : 0;
                    tpos = adv ? tn : tpos;
                    inb = adv ? -1 : inb;
                    tn = (tpos + 1 == tlen) ? 0 : tpos + 1;
                    bwn_ok = j + 1 < whi;
                    bwn = sh.win[(j + 1) & (kWin - 1)][wl];
                    ken = async_itick(sh, gi, j + 2);
                    int32_t n = ke - kk;
                    if (n > lim - n_dl) n = lim - n_dl;
                    abrx::ChainState cs;
                    cs.x = x; cs.eb = inb;
                    bool h = false;
                    const int32_t adds = abrx::chain_segment<abrx::STOP_GE>(cs, c, target, n, h);   // :160-163
                    x = cs.x; inb = cs.eb;
                    n_dl += adds; kk += adds;
                    hit = h;
                    if (h || n_dl >= lim) state = DS_DONE;
                }
            }
        }
        // ======================= pass? =======================
        AST_LAP(1);
        since_pass++;
        const bool pend = (state == DS_DONE && !blocked) || state == DS_BEGIN || state == DS_NEEDWIN ||
                          state == DS_WAITWIN;
        const int npend = __popcll(__ballot(pend));
        const bool none_running = __ballot(state == DS_RUN) == 0;
        if (!(npend >= ABR_ASYNC_THETA_D || none_running || (npend > 0 && since_pass >= ABR_ASYNC_MAXTRIPS)))
            continue;
        since_pass = 0;
        AST_ADD(6, 1); AST_ADD(7, npend);
        // ---- (a) land the bandwidths a previous pass asked for ----
        if (stg_n > 0) {
#pragma unroll
            for (int q = 0; q < kStage; q++)
                if (q < stg_n) sh.win[(whi + q) & (kWin - 1)][wl] = stg[q];
            whi += stg_n; stg_n = 0;
            if (whi - wlo > kWin) wlo = whi - kWin;
            if (state == DS_WAITWIN) {
                if (resume) {
                    bwn = sh.win[(j + 1) & (kWin - 1)][wl];
                    bwn_ok = true;                         // the refill started at or below j + 1
                    state = DS_RUN;
                } else state = DS_BEGIN;
            }
        }
        // ---- (b) the player's word: a download that started at the wrong tick is redone ----
        if (state == DS_DONE || state == DS_FIN) {
            const int32_t ep = lds_ld(&sh.redo_epoch[wl]);
            ABR_LDS_ORDER();
            if (ep != epoch) {
                epoch = ep;
                d_step = sh.redo_step[wl]; d_k = sh.redo_k[wl]; d_chunk = sh.redo_chunk[wl];
                j = sh.redo_j[wl]; tpos = sh.redo_tpos[wl];
                snapped = false; blocked = false;
                state = DS_BEGIN;
            }
        }
        // ---- (c) push finished downloads, set up the next one ----
        if (state == DS_DONE) {
            const int32_t tail = lds_ld(&sh.p_tail[wl]);
            ABR_LDS_ORDER();
            if (head - tail < kR1) {
                const int slot = head & (kR1 - 1);
                sh.r1_dl[slot][wl] = x; sh.r1_ndl[slot][wl] = n_dl; sh.r1_k[slot][wl] = k_start;
                sh.r1_meta[slot][wl] = (act & 0xff) | (hit ? kM1Hit : 0) | (bad ? kM1Bad : 0) |
                                       ((d_step & 63) << 10) | (epoch << 16);
                sh.r1_sj[slot][wl] = snap_j; sh.r1_st[slot][wl] = snap_tpos;
                ABR_LDS_ORDER();
                lds_st(&sh.d_head[wl], ++head);
                blocked = false; snapped = false;
                if (!hit) state = DS_FIN;              // bad action or max_ticks: the player retires the lane
                else {
                    d_step++;
                    d_chunk++;
                    const int32_t av = sh.avail[d_chunk];
                    d_k = kk > av ? kk : av;               // completing tick + 1, or availability (:143)
                    state = DS_BEGIN;
                    if (d_chunk >= V) {
                        if (p.auto_reset) {                // a fresh episode: clock, cursor and chunk ids restart
                            d_chunk = 0; d_k = sh.avail[0];
                            j = 0; tpos = tpos0;
                        } else state = DS_FIN;
                    }
                    if (d_k >= mt) state = DS_FIN;         // a call site at or past max_ticks never happens
                    if (d_step >= n_total) state = DS_FIN;
                }
            } else blocked = true;
        }
        // ---- (d) lanes with nothing left to download wait for the player's last word ----
        if (state == DS_FIN) {
            const int32_t st = lds_ld(&sh.p_status[wl]);
            if ((st & kPDead) || (st & 0xffff) >= n_total) state = DS_EXIT;
        }
        // ---- (e) start the next download (abrx::lanej_begin_step + the prologue of lanej_download) ----
        bool want_fill = false;
        if (state == DS_BEGIN) {
            if (!snapped) { snap_j = j; snap_tpos = tpos; snapped = true; }
            // intervals the cursor is behind the call-site tick (ticks are non-decreasing)
            int32_t adv = 0;
#pragma unroll
            for (int q = 0; q < 5; q++) adv += (d_k >= async_itick(sh, gi, j + 1 + q)) ? 1 : 0;
            if (adv == 5) {
                while (d_k >= async_itick(sh, gi, j + 1 + adv)) adv++;      // the table ends in INT_MAX sentinels
            }
            j += adv;
            tpos = abrx::trace_wrap(tpos + adv, tlen);
            if (j >= wlo && j + 1 < whi) {
                if (d_step < lds_ld(&sh.act_ready[g])) {
                    ABR_LDS_ORDER();
                    act = sh.act[d_step][wl];
                    c = sh.win[j & (kWin - 1)][wl] * abrx::kTickDt;
                    bwn = sh.win[(j + 1) & (kWin - 1)][wl]; bwn_ok = true;
                    ke = async_itick(sh, gi, j + 1); ken = async_itick(sh, gi, j + 2);
                    tn = (tpos + 1 == tlen) ? 0 : tpos + 1;
                    k_start = d_k; kk = d_k; lim = mt - d_k;
                    x = 0.0; n_dl = 0; inb = -1; hit = false;
                    bad = act >= p.n_rates;
                    if (bad) state = DS_DONE;              // nothing downloads: the record carries the verdict
                    else {
                        target = async_bitrate(p, sh, d_chunk, act) * p.chunk_length;     // :156
                        double xx = 0.0;
#pragma unroll
                        for (int q = 0; q < abrx::kPrologue; q++) xx = xx + c;
                        const bool use = (ke - kk >= abrx::kPrologue) && (lim >= abrx::kPrologue) && (xx < target);
                        if (use) { x = xx; n_dl = abrx::kPrologue; kk += abrx::kPrologue; }
                        state = DS_RUN;
                        want_fill = j + kWin - whi >= kStage - 2;   // top the window up while it is cheap
                    }
                }
            } else {
                if (j < wlo || j >= whi) { wlo = j; whi = j; wpos = tpos; }
                want_fill = true; resume = false; state = DS_WAITWIN;
            }
        }
        if (state == DS_NEEDWIN) { want_fill = true; resume = true; state = DS_WAITWIN; }
        // ---- (f) ask for more of the trace; the values are first touched in the NEXT pass ----
        if (want_fill) {
            int32_t cnt = j + kWin - whi;
            cnt = cnt > kStage ? kStage : cnt;
#pragma unroll
            for (int q = 0; q < kStage; q++)
                if (q < cnt) { stg[q] = trace[wpos]; wpos = (wpos + 1 == tlen) ? 0 : wpos + 1; }
            stg_n = cnt > 0 ? cnt : 0;
        }
        AST_LAP(2);
        if (__ballot(state != DS_EXIT) == 0) break;
        if (__ballot(state == DS_RUN) == 0) {
            AST_ADD(8, 1);
            if (spins > 64) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(1);
            if (++spins > kSpinLimit) lds_st(&sh.abort, 1);
            if (lds_ld(&sh.abort)) break;
            AST_LAP(3);
        } else spins = 0;
    }
    AST_FLUSH(0);
    if (in_range && was_alive) { p.j[i] = j; p.tpos[i] = tpos; }
}

// "run an iteration now?" for the consumer roles: enough lanes have input, or every lane that still
// wants input has it, or a ring is full (the producer is blocked), or patience ran out
__device__ __forceinline__ bool async_should_run(bool want, bool avail, bool full, int theta, int &patience) {
    const int nav = __popcll(__ballot(avail));
    if (nav == 0) return false;
    const int nwant = __popcll(__ballot(want));
    if (nav >= theta || nav >= nwant || __ballot(full) != 0 || patience >= kPatience) { patience = 0; return true; }
    patience++;
    return false;
}

// ---------------------------------------------------------------------------------------------
// P: the player side (wave role 1)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void async_role_player(const EnvParams &p, AsyncShared &sh, const int g,
                                                  const int32_t n_total) {
    const int l = threadIdx.x & 63, wl = g * 64 + l;
    const int64_t i = (int64_t)blockIdx.x * kAW + wl;
    const bool in_range = i < p.n_lanes;
    const abrx::Tables tb = make_tables(p);
    LaneJ s;
    s.cur.j = 0; s.cur.tpos = 0; s.cur.tlen = 1; s.cur.trace = p.traces;
    s.buf = 0.0; s.sumk = 0; s.k = 0; s.chunk_id = 0; s.n_su = 0; s.n_rb = 0; s.n_play = 0; s.avail_k = 0;
    s.last_action = -1; s.su = false; s.be = false; s.bf = false; s.sd = p.sd; s.pt = 0.0;
    s.pl_left = 0; s.play_id = 0; s.pt_sum = 0.0; s.lane = i;
    int32_t b_step = 0, episode_no = 0, tail = 0, head2 = 0, epoch = 0;
    bool alive = false, was_alive = false;
    if (in_range) {
        was_alive = !p.done[i];
        lanej_load(s, p, i);
        episode_no = p.episode_no[i];
        alive = was_alive;
    }
    int patience = 0;
    int32_t spins = 0;
    AST_DECL
    for (;;) {
        const bool want = alive && b_step < n_total;
        if (__ballot(want) == 0) break;
        AST_LAP(0);
        const int32_t dh = lds_ld(&sh.d_head[wl]);
        ABR_LDS_ORDER();
        const bool avail = want && dh - tail > 0;
        if (!async_should_run(want, avail, want && dh - tail >= kR1, ABR_ASYNC_THETA_P, patience)) {
            if (spins > 64) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(2);
            if (++spins > kSpinLimit) lds_st(&sh.abort, 1);
            if (lds_ld(&sh.abort)) break;
            AST_LAP(1); AST_ADD(8, 1);
            continue;
        }
        spins = 0;
        AST_ADD(4, 1); AST_ADD(5, __popcll(__ballot(avail)));
        bool accepted = false;
        abrx::Download d; d.dl = 0.0; d.n_dl = 0; d.hit = false;
        int32_t meta = 0;
        if (avail) {
            const int slot = tail & (kR1 - 1);
            d.dl = sh.r1_dl[slot][wl]; d.n_dl = sh.r1_ndl[slot][wl];
            const int32_t ks = sh.r1_k[slot][wl];
            meta = sh.r1_meta[slot][wl];
            const int32_t sj = sh.r1_sj[slot][wl], st = sh.r1_st[slot][wl];
            ABR_LDS_ORDER();
            lds_st(&sh.p_tail[wl], ++tail);
            if (((meta >> 16) & 0xffff) == epoch) {
                // accept the download only if it started at exactly this lane's call-site tick
                if (((meta >> 10) & 63) == (b_step & 63) && ks == s.k) accepted = true;
                else {
                    epoch = (epoch + 1) & 0xffff;
                    sh.redo_step[wl] = b_step; sh.redo_k[wl] = s.k; sh.redo_chunk[wl] = s.chunk_id;
                    sh.redo_ep[wl] = episode_no; sh.redo_j[wl] = sj; sh.redo_tpos[wl] = st;
                    ABR_LDS_ORDER();
                    lds_st(&sh.redo_epoch[wl], epoch);
                }
            }   // else: produced before the redo was seen -- dropped
        }
        int32_t m2 = 0, r_nrb = 0, r_nsu = 0, r_nplay = 0;
        long long r_sumk = 0;
        if (accepted) {
            const int32_t a = meta & 0xff;
            m2 = a;
            if (meta & kM1Bad) {
                m2 |= kM2Bad;
                alive = false;
            } else {
                d.hit = (meta & kM1Hit) != 0;
                const abrx::StepResult r = abrx::lanej_after_download(s, tb, d, sh.avail[s.chunk_id + 1], a);
                if (r.hit) m2 |= kM2Hit;
                if (r.ended) m2 |= kM2Ended;
                if (r.timeout) m2 |= kM2Timeout;
                r_nrb = s.n_rb; r_nsu = s.n_su; r_nplay = s.n_play; r_sumk = s.sumk;
                if (r.ended || r.timeout) {
                    if (p.auto_reset && r.ended) {
                        // re-arm: the observation of this step is the new episode's first call site
                        abrx::lanej_init_player(s, tb);
                        episode_no++;
                        m2 |= kM2Reset;
                        if (!abrx::lanej_wait_call(s, tb)) { m2 |= kM2Timeout2; alive = false; }
                    } else alive = false;
                }
            }
        }
        AST_LAP(2);
        // ring-2 space (S runs as soon as any ring is full, so this wait ends)
        while (__ballot(accepted && head2 - lds_ld(&sh.s_tail2[wl]) >= kR2) != 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kSpinLimit) lds_st(&sh.abort, 1);
            if (lds_ld(&sh.abort)) break;
        }
        if (lds_ld(&sh.abort)) break;
        spins = 0;
        AST_LAP(3);
        if (accepted) {
            const int slot = head2 % kR2;
            sh.r2_dl[slot][wl] = d.dl; sh.r2_buf[slot][wl] = s.buf; sh.r2_sumk[slot][wl] = r_sumk;
            sh.r2_ndl[slot][wl] = d.n_dl; sh.r2_meta[slot][wl] = m2; sh.r2_k[slot][wl] = s.k;
            sh.r2_nplay_o[slot][wl] = s.n_play; sh.r2_nrb_o[slot][wl] = s.n_rb; sh.r2_nsu_o[slot][wl] = s.n_su;
            sh.r2_nrb_r[slot][wl] = r_nrb; sh.r2_nsu_r[slot][wl] = r_nsu; sh.r2_nplay_r[slot][wl] = r_nplay;
            ABR_LDS_ORDER();
            lds_st(&sh.p_head2[wl], ++head2);
            b_step++;
            lds_st(&sh.p_status[wl], b_step | (alive ? 0 : kPDead));
        }
        AST_LAP(2);
    }
    AST_FLUSH(12);
    if (in_range && was_alive) lanej_store_player(s, p, i);
}

// ---------------------------------------------------------------------------------------------
// S: the service side (wave role 2)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void async_write_obs(const EnvParams &p, int64_t i, float *obs, int32_t chunk,
                                                int32_t last_action, double last_bw, double buf, int32_t k,
                                                int32_t n_play, int32_t n_rb, int32_t n_su) {
    if (!obs) return;
    const int64_t n = p.n_lanes;
    obs[ABR_OBS_CHUNK_ID * n + i] = (float)chunk;
    obs[ABR_OBS_LAST_BITRATE * n + i] = (float)last_action;
    obs[ABR_OBS_LAST_BANDWIDTH * n + i] = (float)last_bw;
    obs[ABR_OBS_BUFFER_LEVEL * n + i] = (float)buf;
    obs[ABR_OBS_GLOBAL_TIME * n + i] = (float)p.G[k];
    obs[ABR_OBS_PLAY_TIME * n + i] = (float)p.GP[n_play];
    obs[ABR_OBS_REBUFFER_TIME * n + i] = (float)p.G[n_rb];
    obs[ABR_OBS_STARTUP_TIME * n + i] = (float)p.G[n_su];
}

// MODE 2: built-in random policy; MODE 3: scripted actions [n_total][n_lanes]
template <int MODE>
__device__ __forceinline__ void async_role_service(
    const EnvParams &p, AsyncShared &sh, const int g, const int32_t n_total, const uint64_t seed,
    const int32_t *__restrict__ script, float *__restrict__ obs_out, float *__restrict__ reward_out,
    uint8_t *__restrict__ done_out, int32_t *__restrict__ actions_out) {
    const int l = threadIdx.x & 63, wl = g * 64 + l;
    const int64_t i = (int64_t)blockIdx.x * kAW + wl;
    const bool in_range = i < p.n_lanes;
    const int32_t V = p.video_length;
    uint8_t done = 0;
    int32_t n_su_obs = 0, n_rb_obs = 0, episode_no = 0;
    double last_bw = 0.0, hist_n = 0.0, hist_s = 0.0, g_su_obs = 0.0, g_rb_obs = 0.0, var_run = 0.0;
    AST_DECL
    // what an observation of this lane shows right now
    int32_t o_chunk = 0, o_last = -1, o_k = 0, o_nplay = 0, o_nrb = 0, o_nsu = 0;
    double o_buf = 0.0;
    bool alive = false, was_alive = false;
    if (in_range) {
        done = p.done[i];
        was_alive = done == 0;
        alive = was_alive;
        n_su_obs = p.n_su_obs[i]; n_rb_obs = p.n_rb_obs[i]; episode_no = p.episode_no[i];
        last_bw = p.last_bw[i]; hist_n = p.hist_n[i]; hist_s = p.hist_s[i]; var_run = p.var_run[i];
        g_su_obs = p.G[n_su_obs]; g_rb_obs = p.G[n_rb_obs];
        o_chunk = p.chunk_id[i]; o_last = p.last_action[i]; o_k = p.k[i]; o_nplay = p.n_play[i];
        o_nrb = p.n_rb[i]; o_nsu = p.n_su[i]; o_buf = p.buf[i];
    }
    // ---- the policy's actions, drawn ahead of D: get_next_bitrate's return values (:155) ----
    {
        int32_t c = o_chunk, e = episode_no;
        for (int32_t t = 0; t < n_total; t++) {
            int32_t a = 0xff;
            if (alive) {
                if (MODE == 2) a = (int32_t)philox_action(seed, (uint64_t)(p.lane_id_base + i), (uint32_t)c,
                                                          (uint32_t)e, (uint32_t)p.n_rates);
                else {
                    const int32_t v = script[(int64_t)t * p.n_lanes + i];
                    a = (v >= 0 && v < p.n_rates) ? v : 0xff;
                }
            }
            sh.act[t][wl] = (uint8_t)a;
            c++;
            if (c >= V) { c = 0; e++; }                    // only meaningful under auto_reset: else the lane is gone
            if ((t & 3) == 3 || t == n_total - 1) {
                ABR_LDS_ORDER();
                if (l == 0) lds_st(&sh.act_ready[g], t + 1);
            }
        }
    }
    int32_t s_step = 0, tail2 = 0, spins = 0;
    int patience = 0;
    AST_LAP(6);
    for (;;) {
        const bool want = alive && s_step < n_total;
        if (__ballot(want) == 0) break;
        AST_LAP(0);
        const int32_t ph = lds_ld(&sh.p_head2[wl]);
        ABR_LDS_ORDER();
        const bool avail = want && ph - tail2 > 0;
        if (!async_should_run(want, avail, want && ph - tail2 >= kR2, ABR_ASYNC_THETA_S, patience)) {
            if (spins > 64) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(2);
            if (++spins > kSpinLimit) lds_st(&sh.abort, 1);
            if (lds_ld(&sh.abort)) {
                if (want) { done |= ABR_DONE_INTERNAL; alive = false; }
                break;
            }
            AST_LAP(1); AST_ADD(8, 1);
            continue;
        }
        spins = 0;
        AST_ADD(4, 1); AST_ADD(5, __popcll(__ballot(avail)));
        if (avail) {
            const int slot = tail2 % kR2;
            const double dl = sh.r2_dl[slot][wl], buf = sh.r2_buf[slot][wl];
            const long long sumk = sh.r2_sumk[slot][wl];
            const int32_t n_dl = sh.r2_ndl[slot][wl], m2 = sh.r2_meta[slot][wl], k = sh.r2_k[slot][wl];
            const int32_t nplay_o = sh.r2_nplay_o[slot][wl], nrb_o = sh.r2_nrb_o[slot][wl],
                          nsu_o = sh.r2_nsu_o[slot][wl], nrb_r = sh.r2_nrb_r[slot][wl],
                          nsu_r = sh.r2_nsu_r[slot][wl], nplay_r = sh.r2_nplay_r[slot][wl];
            ABR_LDS_ORDER();
            lds_st(&sh.s_tail2[wl], ++tail2);
            const int64_t o = (int64_t)s_step * p.n_lanes + i;
            float *obs = obs_out ? obs_out + (int64_t)s_step * ABR_OBS_DIM * p.n_lanes : nullptr;
            const int32_t a = m2 & 0xff;
            if (m2 & kM2Bad) {
                done |= ABR_DONE_BADACT;
                if (reward_out) reward_out[o] = 0.0f;
                if (done_out) done_out[o] = done;
                if (actions_out) actions_out[o] = MODE == 3 ? script[o] : a;
                async_write_obs(p, i, obs, o_chunk, o_last, last_bw, o_buf, o_k, o_nplay, o_nrb, o_nsu);
                alive = false;
            } else {
                const int32_t chunk = o_chunk, prev_action = o_last;
                double var = 0.0;
                if (m2 & kM2Hit) {
                    const double bw = dl / p.G[n_dl];                          // :164
                    const int64_t h = (int64_t)chunk * p.n_lanes + i;
                    p.bw_hist[h] = bw;
                    p.action_hist[h] = (uint8_t)a;                             // :165
                    last_bw = bw;
                    hist_s = hist_s + 1.0 / bw;     // sum(1/x), list order (mpc.py:86-88)
                    hist_n = hist_n + 1.0;
                    if (prev_action >= 0)
                        var = fabs(async_bitrate(p, sh, chunk, a) - async_bitrate(p, sh, chunk - 1, prev_action));
                    var_run = var_run + var;
                    o_last = a; o_chunk = chunk + 1;
                }
                // ---- step boundary: per-step split of calculate_qoe (:83-85) ----
                const double g_rb = p.G[nrb_r], g_su = p.G[nsu_r];
                const double rew = p.wr * (g_rb - g_rb_obs) + p.ws * (g_su - g_su_obs) + p.wv * var;
                if (m2 & kM2Ended) done |= ABR_DONE_EPISODE;
                if (m2 & kM2Timeout) done |= ABR_DONE_TIMEOUT;
                if (reward_out) reward_out[o] = (float)rew;
                if (done_out) done_out[o] = done;
                if (actions_out) actions_out[o] = a;
                n_su_obs = nsu_r; n_rb_obs = nrb_r; g_su_obs = g_su; g_rb_obs = g_rb;
                if (m2 & (kM2Ended | kM2Timeout)) {
                    p.ep_qoe_terms[0 * p.n_lanes + i] = g_rb;
                    p.ep_qoe_terms[1 * p.n_lanes + i] = g_su;
                    p.ep_qoe_terms[2 * p.n_lanes + i] = lane_avg_latency(p, sumk, nplay_r);
                    p.ep_qoe_terms[3 * p.n_lanes + i] = var_run;
                    if (m2 & kM2Reset) {
                        episode_no++;
                        n_su_obs = 0; n_rb_obs = 0; g_su_obs = 0.0; g_rb_obs = 0.0;
                        last_bw = 0.0; hist_n = 0.0; hist_s = 0.0; var_run = 0.0;
                        done = (m2 & kM2Timeout2) ? ABR_DONE_TIMEOUT : 0;
                        o_chunk = 0; o_last = -1;
                    }
                }
                o_buf = buf; o_k = k; o_nplay = nplay_o; o_nrb = nrb_o; o_nsu = nsu_o;
                async_write_obs(p, i, obs, o_chunk, o_last, last_bw, o_buf, o_k, o_nplay, o_nrb, o_nsu);
                if (done) alive = false;
            }
            s_step++;
        }
        AST_LAP(2);
    }
    AST_FLUSH(24);
    if (in_range) {
        if (was_alive) {
            p.n_su_obs[i] = n_su_obs; p.n_rb_obs[i] = n_rb_obs; p.episode_no[i] = episode_no;
            p.last_bw[i] = last_bw; p.hist_n[i] = hist_n; p.hist_s[i] = hist_s; p.var_run[i] = var_run;
            p.done[i] = done;
        }
        // lanes that were already finished (or finished early) report their terminal record
        // for the remaining steps
        for (int32_t t2 = s_step; t2 < n_total; t2++) {
            const int64_t o = (int64_t)t2 * p.n_lanes + i;
            if (reward_out) reward_out[o] = 0.0f;
            if (done_out) done_out[o] = done;
            if (actions_out) actions_out[o] = -1;
            async_write_obs(p, i, obs_out ? obs_out + (int64_t)t2 * ABR_OBS_DIM * p.n_lanes : nullptr, o_chunk,
                            o_last, last_bw, o_buf, o_k, o_nplay, o_nrb, o_nsu);
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(3 * kAW) void env_async_kernel(
    EnvParams p, const int32_t *__restrict__ script, float *__restrict__ obs_out,
    float *__restrict__ reward_out, uint8_t *__restrict__ done_out, int32_t *__restrict__ actions_out,
    int32_t n_steps, uint64_t seed) {
    __shared__ AsyncShared sh;
    // ---- stage the lane-independent tables, zero the counters (the only workgroup barrier) ----
    const int32_t n_it = p.n_intervals + 8 < kItickLds ? p.n_intervals + 8 : kItickLds;
    for (int32_t q = threadIdx.x; q < n_it; q += blockDim.x) sh.itick[q] = p.interval_tick[q];
    for (int32_t q = threadIdx.x; q < p.video_length + 2; q += blockDim.x) sh.avail[q] = p.avail_tick[q];
    if (threadIdx.x < kAW) {
        const int q = threadIdx.x;
        sh.d_head[q] = 0; sh.p_tail[q] = 0; sh.p_status[q] = 0; sh.redo_epoch[q] = 0;
        sh.p_head2[q] = 0; sh.s_tail2[q] = 0;
        if (q < kAG) sh.act_ready[q] = 0;
        if (q == 0) sh.abort = 0;
        if (q < ABR_MAX_RATES) sh.lad[q] = p.ladder[q];
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const int role = wave / kAG, g = wave % kAG;     // wave-uniform
    if (role == 0) async_role_download(p, sh, g, n_steps);
    else if (role == 1) async_role_player(p, sh, g, n_steps);
    else async_role_service<MODE>(p, sh, g, n_steps, seed, script, obs_out, reward_out, done_out, actions_out);
}

#endif
