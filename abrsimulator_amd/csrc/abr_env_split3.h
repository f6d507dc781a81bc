// abr_env_split3.h -- K1, role-split form with THREE waves per 64 lanes (impl 5), included by abr_env.hip.
//
// Why: once the download wave runs at raised priority (s_setprio 1, +2.8 %) the PLAYER wave becomes the
// critical one of the two-wave kernel -- profiles/r03_role_stamps.txt: D 13.4 k cycles of work + 3.0 k at the
// barrier, P 15.7 k + 0.8 k per iteration -- and a quarter of P's path is the service tail of a decision:
// bandwidth = size / time, history, reward, done, the observation, the episode end (Simulator.py:164-165 and
// the per-step split of :79-86).  That tail needs nothing but the record of the step P has just finished, so
// a third wave runs it one iteration behind P, exactly as P runs one iteration behind D:
//
//   wave 0  D  download of step s      (split_role_download, unchanged: mailbox `m`)
//   wave 1  P  player side of step s-1 (buffer / counters / completing tick / wait; validates D's guess)
//   wave 2  S  service of step s-2     (division, history, reward, done, observation, episode end, all stores)
//
// One workgroup barrier per iteration, all three waves; mailboxes double-buffered by iteration parity.  P
// hands S the step's download record plus the handful of counters an observation and a reward are made of
// (P -> S: SplitMail2).  Same lane arithmetic, same workspace, same results as every other implementation.
#ifndef ABR_ENV_SPLIT3_H
#define ABR_ENV_SPLIT3_H

struct SplitMail2 {                      // P -> S, double-buffered by iteration parity
    double dl[2][64], buf[2][64], lat[2][64], pt[2][64];
    int32_t meta[2][64], step[2][64], n_dl[2][64], k[2][64], nplay_o[2][64], nrb_o[2][64], nsu_o[2][64],
        nrb_r[2][64], nsu_r[2][64];
};
constexpr int kS3Valid = 0x10000, kS3Hit = 0x100, kS3Bad = 0x200, kS3Ended = 0x400, kS3Timeout = 0x800,
              kS3Reset = 0x1000, kS3Timeout2 = 0x2000;

// Wave 1: the player side.  split_role_player without its service tail: the finished step goes to S.
template <int MODE>
__device__ __forceinline__ void split3_role_player(const EnvParams &p, SplitMail &m, SplitMail2 &m2,
                                                   int32_t n_total) {
    const int l = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 64 + l;
    const bool in_range = i < p.n_lanes;
    const abrx::Tables tb = make_tables(p);
    __builtin_amdgcn_s_setprio(1);     // below the download wave (2), above the service wave (0): worth 8 %
    LaneJ s;
    s.cur.j = 0; s.cur.tpos = 0; s.cur.tlen = 1; s.cur.trace = p.traces;
    int32_t episode_no = 0, b_step = 0;
    bool b_alive = false, was_done = true;
    if (in_range) {
        was_done = p.done[i] != 0;
        lanej_load(s, p, i);
        episode_no = p.episode_no[i];
        b_alive = !was_done;
    }
    const bool speeds = p.lane_speeds != nullptr, sched = speeds && p.speed_rows >= 2;
    ABR_STAMP_INIT();
    for (int32_t t = 0;; t++) {
        const int cb = t & 1, pb = (t + 1) & 1;    // this iteration's / the previous one's slot
        int32_t meta = 0;
        ABR_STAMP(8);
        if (b_alive && b_step < n_total && t >= 1) {
            const int32_t fl = m.flags[pb][l];
            // accept the download only if it started at exactly this lane's call-site tick
            if ((fl & kRecValid) && m.step[pb][l] == b_step && m.k_start[pb][l] == s.k) {
                const int32_t a = m.action[pb][l];
                meta = kS3Valid | (a & 0xff);
                m2.step[cb][l] = b_step;
                if (fl & kRecBadAct) {
                    meta |= kS3Bad;
                    b_alive = false;
                } else {
                    abrx::Download d;
                    d.dl = m.dl[pb][l]; d.n_dl = m.n_dl[pb][l]; d.hit = (fl & kRecHit) != 0;
                    const abrx::StepResult r = abrx::lanej_after_download(s, tb, d, m.avail_next[pb][l], a);
                    if (r.hit) meta |= kS3Hit;
                    if (r.ended) meta |= kS3Ended;
                    if (r.timeout) meta |= kS3Timeout;
                    m2.dl[cb][l] = d.dl; m2.n_dl[cb][l] = d.n_dl;
                    m2.nrb_r[cb][l] = s.n_rb; m2.nsu_r[cb][l] = s.n_su;
                    if (r.ended || r.timeout) {
                        m2.lat[cb][l] = !speeds ? lane_avg_latency(p, s.sumk, s.n_play)
                                        : (sched ? avg_latency_sched(s.pt, s.sumk, s.pt_sum, s.n_play)
                                                 : avg_latency_from(s.sd, s.pt, s.sumk, s.n_play));
                        if (p.auto_reset && r.ended) {
                            // re-arm: this step's observation is the new episode's first call site
                            abrx::lanej_init_player(s, tb);
                            episode_no++;
                            meta |= kS3Reset;
                            if (!abrx::lanej_wait_call(s, tb)) { meta |= kS3Timeout2; b_alive = false; }
                        } else b_alive = false;
                    }
                    m2.buf[cb][l] = s.buf; m2.k[cb][l] = s.k; m2.nplay_o[cb][l] = s.n_play;
                    m2.nrb_o[cb][l] = s.n_rb; m2.nsu_o[cb][l] = s.n_su;
                    if (speeds) m2.pt[cb][l] = s.pt;
                }
                b_step++;
            }
        }
        ABR_STAMP(13);
        m2.meta[cb][l] = meta;
        // ---- tell the download side where this lane really is ----
        const bool more = b_alive && b_step < n_total;
        m.fb_step[cb][l] = b_step; m.fb_k[cb][l] = s.k; m.fb_chunk[cb][l] = s.chunk_id;
        m.fb_episode[cb][l] = episode_no; m.fb_alive[cb][l] = more ? 1 : 0;
        const bool any = __any(more) != 0;
        if (l == 0) m.any_alive[cb] = any ? 1 : 0;
        ABR_STAMP(17);
        __syncthreads();
        ABR_STAMP(18);
        if (!m.any_alive[cb]) break;               // wave-uniform, identical in all three waves
    }
    ABR_STAMP_FLUSH();
    if (in_range && !was_done) lanej_store_player(s, p, i);
}

// Wave 2: the service side -- everything a decision writes to global memory.
template <int MODE>
__device__ __forceinline__ void split3_role_service(
    const EnvParams &p, SplitMail &m, SplitMail2 &m2, float *__restrict__ obs_out,
    float *__restrict__ reward_out, uint8_t *__restrict__ done_out, int32_t *__restrict__ actions_out,
    int32_t n_total, uint64_t seed) {
    const int l = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 64 + l;
    const bool in_range = i < p.n_lanes;
    const int32_t V = p.video_length;
    const bool speeds = p.lane_speeds != nullptr;
    uint8_t done = 0;
    int32_t n_su_obs = 0, n_rb_obs = 0, episode_no = 0, s_next = 0;
    double last_bw = 0.0, hist_n = 0.0, hist_s = 0.0, g_su_obs = 0.0, g_rb_obs = 0.0;
    // what an observation of this lane shows right now
    int32_t o_chunk = 0, o_last = -1, o_k = 0, o_nplay = 0, o_nrb = 0, o_nsu = 0;
    double o_buf = 0.0, o_pt = 0.0;
    bool was_done = true;
    if (in_range) {
        done = p.done[i];
        was_done = done != 0;
        n_su_obs = p.n_su_obs[i]; n_rb_obs = p.n_rb_obs[i]; episode_no = p.episode_no[i];
        last_bw = p.last_bw[i]; hist_n = p.hist_n[i]; hist_s = p.hist_s[i];
        g_su_obs = p.G[n_su_obs]; g_rb_obs = p.G[n_rb_obs];
        o_chunk = p.chunk_id[i]; o_last = p.last_action[i]; o_k = p.k[i]; o_nplay = p.n_play[i];
        o_nrb = p.n_rb[i]; o_nsu = p.n_su[i]; o_buf = p.buf[i];
        if (speeds) o_pt = p.pt_lane[i];
    }
    auto write_obs_now = [&](float *obs) {
        if (!obs) return;
        const int64_t n = p.n_lanes;
        obs[ABR_OBS_CHUNK_ID * n + i] = (float)o_chunk;
        obs[ABR_OBS_LAST_BITRATE * n + i] = (float)o_last;
        obs[ABR_OBS_LAST_BANDWIDTH * n + i] = (float)last_bw;
        obs[ABR_OBS_BUFFER_LEVEL * n + i] = (float)o_buf;
        obs[ABR_OBS_GLOBAL_TIME * n + i] = (float)p.G[o_k];
        obs[ABR_OBS_PLAY_TIME * n + i] = (float)(speeds ? o_pt : p.GP[o_nplay]);
        obs[ABR_OBS_REBUFFER_TIME * n + i] = (float)p.G[o_nrb];
        obs[ABR_OBS_STARTUP_TIME * n + i] = (float)p.G[o_nsu];
    };
    auto service = [&](const int sl) {
        const int32_t m2m = m2.meta[sl][l];
        if (!(m2m & kS3Valid)) return;
        const int32_t step = m2.step[sl][l];
        const int64_t o = (int64_t)step * p.n_lanes + i;
        float *obs = obs_out ? obs_out + (int64_t)step * ABR_OBS_DIM * p.n_lanes : nullptr;
        const int32_t a = m2m & 0xff;
        s_next = step + 1;
        if (m2m & kS3Bad) {
            done |= ABR_DONE_BADACT;
            if (reward_out) reward_out[o] = 0.0f;
            if (done_out) done_out[o] = done;
            write_obs_now(obs);
            return;
        }
        const int32_t chunk = o_chunk, prev_action = o_last;
        const int32_t nrb_r = m2.nrb_r[sl][l], nsu_r = m2.nsu_r[sl][l];
        double var = 0.0;
        if (m2m & kS3Hit) {
            const double bw = m2.dl[sl][l] / p.G[m2.n_dl[sl][l]];                  // :164
            const int64_t h = (int64_t)chunk * p.n_lanes + i;
            p.bw_hist[h] = bw;
            p.action_hist[h] = (uint8_t)a;                                       // :165
            last_bw = bw;
            hist_s = hist_s + 1.0 / bw;         // sum(1/x), list order (mpc.py:86-88)
            hist_n = hist_n + 1.0;
            if (prev_action >= 0)
                var = fabs(chunk_bitrate(p, chunk, a) - chunk_bitrate(p, chunk - 1, prev_action));
            o_last = a; o_chunk = chunk + 1;
        }
        // ---- step boundary: per-step split of calculate_qoe (:83-85) ----
        const double g_rb = p.G[nrb_r], g_su = p.G[nsu_r];
        const double rew = p.wr * (g_rb - g_rb_obs) + p.ws * (g_su - g_su_obs) + p.wv * var;
        if (m2m & kS3Ended) done |= ABR_DONE_EPISODE;
        if (m2m & kS3Timeout) done |= ABR_DONE_TIMEOUT;
        if (reward_out) reward_out[o] = (float)rew;
        if (done_out) done_out[o] = done;
        n_su_obs = nsu_r; n_rb_obs = nrb_r; g_su_obs = g_su; g_rb_obs = g_rb;
        if (m2m & (kS3Ended | kS3Timeout)) {
            p.ep_qoe_terms[0 * p.n_lanes + i] = g_rb;
            p.ep_qoe_terms[1 * p.n_lanes + i] = g_su;
            p.ep_qoe_terms[2 * p.n_lanes + i] = m2.lat[sl][l];
            if (m2m & kS3Reset) {
                copy_episode_actions(p, i, V);
                episode_no++;
                n_su_obs = 0; n_rb_obs = 0; g_su_obs = 0.0; g_rb_obs = 0.0;
                last_bw = 0.0; hist_n = 0.0; hist_s = 0.0;
                done = (m2m & kS3Timeout2) ? ABR_DONE_TIMEOUT : 0;
                o_chunk = 0; o_last = -1;
            }
        }
        o_buf = m2.buf[sl][l]; o_k = m2.k[sl][l]; o_nplay = m2.nplay_o[sl][l]; o_nrb = m2.nrb_o[sl][l];
        o_nsu = m2.nsu_o[sl][l];
        if (speeds) o_pt = m2.pt[sl][l];
        write_obs_now(obs);
    };
    // ---- the policy's draws, made ahead of the download wave (this wave idles most of an iteration) ----
    // Launch step s of a lane is chunk (chunk0 + s) of its episode sequence whatever is redone on the way, so
    // the draw of step s is known up front.  act[s % 64] may be overwritten once every lane still running is
    // past step s, i.e. steps below lo + 64 with lo = the slowest live lane's step (P's fb_step, one iteration
    // old); D never waits for an entry -- one that is not there it draws itself.
    int32_t a_next = 0, a_chunk = o_chunk, a_ep = episode_no;      // next step to draw, and its (chunk, episode)
    auto draw_ahead = [&](const int32_t lo, const int32_t count) {
        if (MODE != 2) return;
        int32_t hi = lo + 60;
        if (hi > n_total) hi = n_total;
        for (int32_t q = 0; q < count && a_next < hi; q++) {
            const uint32_t a = philox_action(seed, (uint64_t)(p.lane_id_base + i), (uint32_t)a_chunk, (uint32_t)a_ep,
                                             (uint32_t)p.n_rates);
            m.act[a_next & 63][l] = (uint8_t)a;
            a_next++; a_chunk++;
            if (a_chunk >= V) { a_chunk = 0; a_ep++; }
        }
        // publish: the bytes first, then the counter that vouches for them (read in that order by D)
        ABR_LDS_ORDER();
        if (l == 0) lds_st(&m.act_hi, a_next);
    };
    int last_cb = 0;
    ABR_STAMP_INIT();
    for (int32_t t = 0;; t++) {
        const int cb = t & 1, pb = (t + 1) & 1;
        ABR_STAMP(20);
        {
            // the slowest live lane's step, as P published it in the previous iteration
            int32_t lo = 0;
            if (t >= 1) {
                lo = m.fb_alive[pb][l] ? m.fb_step[pb][l] : 0x7fffffff;
#pragma unroll
                for (int sh = 32; sh >= 1; sh >>= 1) { const int32_t o2 = __shfl_xor(lo, sh, 64); lo = o2 < lo ? o2 : lo; }
                if (lo == 0x7fffffff) lo = n_total;
            }
            draw_ahead(lo, t == 0 ? 4 : 3);
        }
        if (t >= 1 && in_range) service(pb);       // what P finished in the previous iteration
        last_cb = cb;
        ABR_STAMP(21);
        __syncthreads();
        ABR_STAMP(22);
        if (!m.any_alive[cb]) break;               // wave-uniform, identical in all three waves
    }
    ABR_STAMP_FLUSH();
    if (in_range) {
        service(last_cb);                          // P's last records
        if (!was_done) {
            p.n_su_obs[i] = n_su_obs; p.n_rb_obs[i] = n_rb_obs; p.episode_no[i] = episode_no;
            p.last_bw[i] = last_bw; p.hist_n[i] = hist_n; p.hist_s[i] = hist_s;
            p.done[i] = done;
        }
        // lanes that were already finished (or finished early) report their terminal record
        // for the remaining steps
        for (int32_t t2 = s_next; t2 < n_total; t2++) {
            const int64_t o = (int64_t)t2 * p.n_lanes + i;
            if (reward_out) reward_out[o] = 0.0f;
            if (done_out) done_out[o] = done;
            if (MODE == 2 && actions_out) actions_out[o] = -1;
            write_obs_now(obs_out ? obs_out + (int64_t)t2 * ABR_OBS_DIM * p.n_lanes : nullptr);
        }
    }
}

// MODE 1: one externally supplied action per lane; MODE 2: fused random-policy rollout;
// MODE 3: fused rollout of scripted actions.  Barrier discipline as env_split_kernel: each wave is
// entirely in one role, every role loop executes exactly one barrier per iteration and all leave in the
// same iteration (the exit flag is written by P before the barrier and read by all after it).
template <int MODE>
__global__ __launch_bounds__(192) void env_split3_kernel(
    EnvParams p, const int32_t *__restrict__ actions, float *__restrict__ obs_out,
    float *__restrict__ reward_out, uint8_t *__restrict__ done_out,
    int32_t *__restrict__ actions_out, int32_t n_steps, uint64_t seed) {
    __shared__ SplitMail m;
    __shared__ SplitMail2 m2;
    const int32_t n_total = (MODE >= 2) ? n_steps : 1;
    if (threadIdx.x == 0) m.act_hi = 0;
    __syncthreads();
    if (threadIdx.x < 64) split_role_download<MODE, true>(p, m, actions, actions_out, n_total, seed);
    else if (threadIdx.x < 128) split3_role_player<MODE>(p, m, m2, n_total);
    else split3_role_service<MODE>(p, m, m2, obs_out, reward_out, done_out, actions_out, n_total, seed);
}

#endif
