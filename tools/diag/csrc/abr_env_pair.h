// abr_env_pair.h -- K1 in role-split form with the SERVICE wave off the per-iteration rendezvous (round 5; impl 7 "pair3").
// Included by abr_env.hip under -DABR_WITH_RING (diagnostic build).
//
//   env_pair3_kernel   D | P | S   the three roles of env_split3_kernel (abr_env_roles.h) and literally its download and player
//                                  code; D and P meet once per iteration as there, S does not meet anybody
//
// Why (profiles/r05_role_stamps_split3_xcd.txt): at identical work the XCDs of an MI355X differ by 10 - 18 % in how long an iteration of
// the three-wave kernel lasts, and what differs is not the download or the player wave's work but how late the memory-bound service
// wave is at the workgroup barrier -- although on average it has 4.5 k of 12.5 k cycles to spare.  Here the barrier is gone: D and
// P rendezvous through two counters in LDS (D publishes "iteration t done", waits for P's, and vice versa: the lock-step the exact
// call-site prediction needs, abr_lane_jump.h: lanej_predict_next_call), and P hands finished steps to S through a K2-deep ring of
// records as in the ring pipeline (abr_env_ring.h), so that S's lateness costs nothing until it is K2 records behind.  No run-ahead
// of D over P: nothing of the ring pipeline's correction machinery is needed, the mailboxes between D and P are the barrier
// kernel's (double-buffered by iteration parity, valid for waves at most one iteration apart).
//
// Forward progress: D waits only for P's counter, P for D's counter and for ring space (S's counter), S for P's counter; S never
// waits for space, so the wait-for graph has no cycle; every counter has one writer and only grows; a waiting wave sleeps; a
// watchdog turns a stall into ABR_DONE_INTERNAL.  Counters are published after lds_writes_done() (abr_env_roles.h).
#ifndef ABR_ENV_PAIR_H
#define ABR_ENV_PAIR_H

struct PairCtl {
    int32_t d_cnt;        // D: iterations finished (its record t and everything before it are in the mailbox)
    int32_t p_cnt;        // P: iterations finished = records handed to S (slot t % K2) and feedback published
    int32_t s_pub;        // S: records serviced
    int32_t p_done;       // P: the loop has ended; p_cnt is final
    int32_t abort;        // any: watchdog
};
struct PairFb { int32_t step[64], alive[64]; };      // the player's position for S's draw-ahead bound (single copy: per lane it only grows)

template <int MODE>
__device__ __forceinline__ void pair_d_loop(const EnvParams &p, SplitMail &m, PairCtl &ctl, ActRing &ring,
                                            const int32_t *__restrict__ actions, int32_t *__restrict__ actions_out,
                                            int32_t n_total, uint64_t seed) {
    DVars v;
    role_d_begin(v, p);
    int32_t polls = 0;
    for (int32_t t = 0;; t++) {
        if (t > 0) role_d_validate(v, m, make_tables(p), t - 1);
        role_d_pre<MODE, true>(v, p, m, &ring, actions, actions_out, n_total, seed, t);
        ring_publish(&ctl.d_cnt, t + 1);
        bool leave = false;
        for (;;) {                                         // the rendezvous with P
            const int32_t ab = ring_ld(&ctl.abort), pc = ring_ld(&ctl.p_cnt);
            if (ab) { leave = true; break; }
            if (pc >= t + 1) break;
            __builtin_amdgcn_s_sleep(1);
            if (++polls > kRingWatchdog) { ring_publish(&ctl.abort, 1); leave = true; break; }
        }
        ABR_STAMP(5);
        if (leave || !m.any_alive[t & 1]) break;           // written by P before it published p_cnt = t + 1
    }
    role_d_end(v, p);
}

template <int MODE>
__device__ __forceinline__ void pair_p_loop(const EnvParams &p, SplitMail &m, RingPS &m2, PairFb &fbs, PairCtl &ctl, int32_t n_total) {
    const int l = threadIdx.x & 63;
    PVars v;
    role_p3_begin(v, p);
    int32_t polls = 0;
    for (int32_t t = 0;; t++) {
        bool leave = false;
        for (;;) {                                         // space: slot t % K2 is free once S has serviced record t - K2
            const int32_t ab = ring_ld(&ctl.abort), sp = ring_ld(&ctl.s_pub);
            if (ab) { leave = true; break; }
            if (sp >= t - kRingK2 + 1) break;
            __builtin_amdgcn_s_sleep(1);
            if (++polls > kRingWatchdog) { ring_publish(&ctl.abort, 1); leave = true; break; }
        }
        if (leave) break;
        ABR_STAMP(19);
        role_p3_pre<MODE>(v, p, m, m2, n_total, t, t & (kRingK2 - 1));
        fbs.step[l] = v.b_step; fbs.alive[l] = (v.b_alive && v.b_step < n_total) ? 1 : 0;
        ring_publish(&ctl.p_cnt, t + 1);
        for (;;) {                                         // the rendezvous with D
            const int32_t ab = ring_ld(&ctl.abort), dc = ring_ld(&ctl.d_cnt);
            if (ab) { leave = true; break; }
            if (dc >= t + 1) break;
            __builtin_amdgcn_s_sleep(1);
            if (++polls > kRingWatchdog) { ring_publish(&ctl.abort, 1); leave = true; break; }
        }
        ABR_STAMP(18);
        if (leave || !m.any_alive[t & 1]) break;
    }
    ring_publish(&ctl.p_done, 1);                          // after the last p_cnt: it is final
    role_p3_end(v, p);
}

template <int MODE>
__device__ __forceinline__ void pair_s_loop(const EnvParams &p, RingPS &m2, PairFb &fbs, PairCtl &ctl, ActRing &ring,
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
            // draw the policy's actions ahead of D (abr_env_roles.h: role_s_pre): entries below lo + 64 may be overwritten,
            // lo = the slowest live lane's step as the player last published it (per lane it only grows)
            int32_t lo = fbs.alive[l] ? fbs.step[l] : 0x7fffffff;
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
            lds_writes_done();
            if (l == 0) lds_st(&ring.act_hi, v.a_next);
        }
        ABR_STAMP(20);
        bool leave = false;
        for (;;) {                                         // input: the record of P's iteration w
            const int32_t ab = ring_ld(&ctl.abort), dn = ring_ld(&ctl.p_done);      // p_done before p_cnt: final once set
            ABR_LDS_ORDER();
            const int32_t pc = ring_ld(&ctl.p_cnt);
            if (ab) { leave = true; aborted = true; break; }
            if (pc > w) break;
            if (dn) { leave = true; break; }
            __builtin_amdgcn_s_sleep(2);
            if (++polls > kRingWatchdog) { ring_publish(&ctl.abort, 1); leave = true; aborted = true; break; }
        }
        if (leave) break;
        ABR_STAMP(22);
        if (i < p.n_lanes) service_record(v, p, m2, w & (kRingK2 - 1), obs_out, reward_out, done_out);
        ring_publish(&ctl.s_pub, w + 1);
        ABR_STAMP(21);
    }
    ABR_STAMP_FLUSH();
    if (i >= p.n_lanes) return;
    if (aborted && !v.done && v.s_next < n_total) v.done |= ABR_DONE_INTERNAL;
    if (!v.was_done) {
        p.n_su_obs[i] = v.n_su_obs; p.n_rb_obs[i] = v.n_rb_obs; p.episode_no[i] = v.episode_no;
        p.last_bw[i] = v.last_bw; p.hist_n[i] = v.hist_n; p.hist_s[i] = v.hist_s; p.var_run[i] = v.var_run;
        p.done[i] = v.done;
    }
    for (int32_t t2 = v.s_next; t2 < n_total; t2++) {      // lanes that finished early report their terminal record again
        const int64_t o = (int64_t)t2 * p.n_lanes + i;
        if (reward_out) reward_out[o] = 0.0f;
        if (done_out) done_out[o] = v.done;
        if (MODE == 2 && actions_out) actions_out[o] = -1;
        service_write_obs(v, p, i, obs_out ? obs_out + (int64_t)t2 * ABR_OBS_DIM * p.n_lanes : nullptr);
    }
}

template <int MODE>
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(1, 3))) void env_pair3_kernel(
    EnvParams p, const int32_t *__restrict__ actions, float *__restrict__ obs_out,
    float *__restrict__ reward_out, uint8_t *__restrict__ done_out,
    int32_t *__restrict__ actions_out, int32_t n_steps, uint64_t seed) {
    __shared__ SplitMail m;
    __shared__ RingPS m2;
    __shared__ PairFb fbs;
    __shared__ PairCtl ctl;
    __shared__ ActRing ring;
    const int32_t n_total = (MODE >= 2) ? n_steps : 1;
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    {
        const int l = threadIdx.x & 63;
        if (role == 1) { fbs.step[l] = 0; fbs.alive[l] = 1; }
        else if (role == 2 && l == 0) {
            ctl.d_cnt = 0; ctl.p_cnt = 0; ctl.s_pub = 0; ctl.p_done = 0; ctl.abort = 0;
            ring.act_hi = 0;
        }
    }
    __syncthreads();                                        // the ONE workgroup barrier of the kernel
    ABR_WG_WHERE(role);
    if (role == 0) {
        ABR_WG_TIME(0);
        __builtin_amdgcn_s_setprio(ABR_PRIO_D);
        pair_d_loop<MODE>(p, m, ctl, ring, actions, actions_out, n_total, seed);
        ABR_WG_TIME(1);
    } else if (role == 1) {
        __builtin_amdgcn_s_setprio(ABR_PRIO_P);
        pair_p_loop<MODE>(p, m, m2, fbs, ctl, n_total);
        ABR_WG_TIME(2);
    } else {
        __builtin_amdgcn_s_setprio(ABR_PRIO_S);
        pair_s_loop<MODE>(p, m2, fbs, ctl, ring, obs_out, reward_out, done_out, actions_out, n_total, seed);
        ABR_WG_TIME(3);
    }
}

#endif
