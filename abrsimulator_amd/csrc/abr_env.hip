// abr_env.hip -- HIP kernels (gfx950 / CDNA4) and the C ABI of include/abr_env.h.
//
// Hot path restated from the reference (file:line into Elliotshui/ABRSimulator):
//   K1/K2 env_jump_kernel<MODE>     Simulator.py:95-133,135-208   event-driven, one thread per lane: the
//         tick loop's float64 sequences advanced in exact closed form (abr_lane_jump.h,
//         abr_exact_jump.h); MODE 0 reset, 1 step, 2 fused random-policy rollout, 3 fused scripted rollout
//   K1    env_split3_kernel<MODE>   the same lane functions on three waves per 64 lanes (download / player /
//         service; abr_env_roles.h): what impl 3 (auto) runs up to kSplit3MaxLanes (65 536) lanes -- for launches of more
//         than one decision; a single decision per launch goes to env_jump_kernel at every size
//   K1    env_split_kernel<MODE>    ... on two waves per 64 lanes (download / player): up to kSplitMaxLanes (131 072) lanes
//   K1/K2 env_advance_kernel<MODE>  the same, one loop trip per 0.01 s tick (cross-check)
//   K3    mpc_select_kernel<H, B, WVM>   mpc.py:81-93,104-186   harmonic predictor (mpc_predict_kernel ahead of it when the
//         caller provides scratch) + exhaustive B^H: B^2 threads per lane; tables in LDS, depth-first enumeration with
//         prefix sharing, LDS arg-max over the threads, first-leaf search below the winning node
//   K4    episode_qoe_kernel        Simulator.py:79-86
//
// Exactness contract: every quantity that feeds a decision in the
// reference (downloaded_size >= target, buffer_level vs 0 / max_buffer / start_up_length,
// int(global_time / x), play_length >= L) is reproduced bit-for-bit: the lane-specific
// ones (downloaded_size, buffer_level) as the float64 value the serial additions produce,
// the lane-independent ones (global_time and everything that is "k additions of dt
// starting from 0") through tick tables computed once on the host in float64
// (abr_tick_tables.h).  Compile with -ffp-contract=off: an FMA would round
// bandwidth*dt + downloaded_size once instead of twice (Simulator.py:160).
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "abr_env.h"

// Cycle stamps and per-workgroup timing exist only in the diagnostic build (tools/diag/csrc: -DABR_SPLIT_STAMPS); the
// product compiles them to nothing.
#ifdef ABR_SPLIT_STAMPS
#include "abr_diag_stamps.h"
#else
#define ABR_STAMP(n)
#define ABR_STAMP_INIT()
#define ABR_STAMP_FLUSH()
#define ABR_WG_TIME(slot)
#define ABR_WG_WHERE(role)
#define K3_STAMP_DECL
#define K3_STAMP(n)
#define K3_STAMP_FLUSH()
#endif
#include "abr_lane_jump.h"
#include "abr_tick_tables.h"

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess)                                                            \
            return fail(ABR_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e__));       \
    } while (0)

extern "C" int abr_abi_version(void) { return ABR_ABI_VERSION; }
extern "C" const char *abr_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------
// device-side parameter block (passed by value to every env kernel)
// ---------------------------------------------------------------------------
constexpr double kDt = 0.01;  // Simulator.py:133
constexpr int kFlagStartUp = 1, kFlagBufEmpty = 2, kFlagBufFull = 4;
constexpr int kFlagArmed = 8;    // the lane has been reset at least once since abr_env_create

struct EnvParams {
    // config
    int32_t n_rates, video_length, max_ticks, auto_reset;
    int32_t play_ticks_per_chunk;  // P: first n with n-fold sum of speed*dt >= chunk_length (:185)
    int32_t n_intervals;           // entries in interval_tick minus sentinel
    int32_t t_block;               // ticks per block of the tick loop (<= shortest interval)
    int32_t n_traces;              // rows of trace_off / trace_len (range check of reset's trace ids)
    double chunk_length, max_buffer, start_up_length;
    double wr, wv, ws, wl;
    double sd;                     // speed * dt, the product the reference forms each tick (:182)
    double ladder[ABR_MAX_RATES];
    int64_t n_lanes, lane_id_base;
    // tables (device)
    const double *G;               // G[n] = dt added n times to 0.0
    const double *GP;              // same for sd (aliases G when speed == 1)
    const int32_t *interval_tick;  // first tick k with int(G[k]/interval) >= j   (:158)
    const int32_t *avail_tick;     // first tick k with int(G[k]/L) - 1 >= c      (:143)
    abrx::DrainTab drain;          // the per-binade cascade of buffer_level -= sd (abr_exact_jump.h); n == 0: none
    // traces (device, caller-owned)
    const double *traces;
    const int64_t *trace_off;
    const int32_t *trace_len;
    // per-lane state, SoA (device, in the workspace)
    double *buf, *last_bw, *hist_n, *hist_s;
    double *sd_lane, *pt_lane;     // per-lane speed*dt and play_time (only touched when lane_speeds != nullptr)
    const double *lane_speeds;     // caller-owned per-lane play speeds [speed_rows][n_lanes], or nullptr: one speed for all lanes
    int32_t speed_rows;            // 1: one constant speed per lane; >= 2: a schedule, one row per played chunk
    int32_t *pl_left, *play_id;    // speed schedule only: playing ticks left in the played chunk, chunks played
    double *pt_sum;                // speed schedule only: sum of play_time over the playing ticks
    const double *br_table;        // caller-owned [video_length][n_rates] per-chunk ladders, or nullptr: `ladder`
    long long *sumk;               // sum of tick indices of playing ticks (latency integral)
    int32_t *k, *chunk_id, *n_su, *n_rb, *n_play, *j, *tpos, *trace_id, *offset0;
    int32_t *last_action, *n_su_obs, *n_rb_obs, *episode_no;
    uint8_t *flags, *done;
    uint8_t *action_hist;          // [V][n_lanes]
    double *bw_hist;               // [V][n_lanes]
    double *ep_qoe_terms;          // [4][n_lanes]: rebuffer, startup, avg latency, bitrate variance of the last finished episode
    double *var_run;               // [n_lanes] sum of |br[a_i] - br[a_(i+1)]| over the running episode so far (:82), in its order
    // abr_debug_selfcheck only (nullptr otherwise): instance <9> of a role-split kernel writes 1 here if fresh_params() -- the
    // parameter block re-read from the kernarg segment -- is THIS launch's block (it carries `sentinel`), 2 if not
    uint32_t *selfcheck_out;
    uint64_t sentinel;
};

// What a workspace says about itself, in its last 256 bytes: written by abr_env_create, compared by abr_env_notify_restore
// (and by the Python wrapper's load_state_dict before it copies anything).  A checkpoint is a copy of the workspace, and the
// layout of the regions in front of this tag changes between ABI versions (3: ep_actions gone, var_run added; 4: this tag):
// restoring a workspace of another version, lane count or configuration would otherwise be reinterpreted with shifted
// offsets and no error -- sizes can coincide.  kLayoutRev counts layout changes inside one ABI version.
constexpr uint32_t kTagMagic = 0x57524241u;          // "ABRW"
constexpr uint32_t kLayoutRev = 1;
struct WorkspaceTag {
    uint32_t magic, abi_version, layout_rev, n_rates;
    int64_t n_lanes;
    int32_t video_length, max_ticks, n_intervals, reserved_;
    uint64_t total_bytes;
    double chunk_length, interval, speed;            // the tick tables in front depend on these
    uint8_t pad_[256 - 72];
};
static_assert(sizeof(WorkspaceTag) == 256, "the tag is one 256-byte block");

struct abr_env {
    EnvParams p;
    abr_env_config cfg;
    size_t workspace_bytes;
    void *tag_dev;                  // the workspace's layout tag (device) and what it must hold
    WorkspaceTag tag;
    int impl;   // 3 = auto (default): see effective_impl(); 5 / 2 = role-split event-driven kernels, three / two waves
                // per 64 lanes; 0 = event-driven, one thread per lane; 1 = tick-by-tick kernels (cross-check);
                // 4 = asynchronous role pipeline (diagnostic build only)
    int32_t *mpc_action;            // [n_lanes] scratch of abr_env_step_mpc (in the workspace)
    void *mpc_scratch;              // predictor scratch of abr_env_step_mpc (in the workspace)
    const double *pending_speeds;   // abr_env_set_lane_speeds / _speed_schedule: latched by the next full reset
    int32_t pending_speed_rows;
    bool speeds_dirty;
    const double *pending_br_table; // abr_env_set_bitrate_table: latched the same way
    bool br_table_dirty;
    bool armed;                     // abr_env_reset has run at least once: episodes may be in flight
};

// A handle on which no reset has run has no episode to protect: the setters take effect at once
// (a freshly built handle that receives a checkpointed workspace must already carry them).
static void apply_pending(abr_env *env) {
    if (env->br_table_dirty) { env->p.br_table = env->pending_br_table; env->br_table_dirty = false; }
    if (env->speeds_dirty) {
        env->p.lane_speeds = env->pending_speeds; env->p.speed_rows = env->pending_speed_rows;
        env->speeds_dirty = false;
    }
}

// mpd.chunks[chunk].bitrates[rate]: the single ladder run() indexes (Simulator.py:82,156), or --
// the evident intent of set_mpd's one-ladder-per-line file (Simulator.py:71-76) -- chunk's own
__device__ inline double chunk_bitrate(const EnvParams &p, int32_t chunk, int32_t rate) {
    return p.br_table ? p.br_table[(int64_t)chunk * p.n_rates + rate] : p.ladder[rate];
}

// ---------------------------------------------------------------------------
// philox4x32-10 (Salmon et al. 2011) -- counter-based, so shards reproduce the
// unsharded run: ctr = (global lane id lo, hi, episode step, episode number)
// ---------------------------------------------------------------------------
__host__ __device__ inline uint32_t philox_action(uint64_t seed, uint64_t lane, uint32_t step,
                                                  uint32_t episode, uint32_t n_rates) {
    uint32_t c0 = (uint32_t)lane, c1 = (uint32_t)(lane >> 32), c2 = step, c3 = episode;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return (uint32_t)(((uint64_t)c0 * n_rates) >> 32);   // multiply-shift into [0, n_rates)
}

// ---------------------------------------------------------------------------
// K1/K2: the tick loop
// ---------------------------------------------------------------------------
// "Call site" = the instant inside tick k at which the reference calls
// abr_controller.get_next_bitrate (Simulator.py:155): T1-T3 of tick k are done,
// the download is not paused and download_time == 0.  A step runs from one call
// site to the next and has two phases:
//   A  downloading: every tick adds bandwidth*dt until downloaded_size >= target.
//      The download cannot pause inside phase A: buffer_full only turns on in a
//      tick that completes a chunk (buffer_level grows nowhere else, :170), and
//      availability is monotone in time.
//   B  waiting for the next call site: chunk available (:143) and not buffer_full.
//
// Kernel structure (one lane per thread, one wave per workgroup):
//   block loop   every T ticks (T <= the shortest trace interval, so a lane
//                crosses at most one interval boundary per block):
//                - SERVICE lanes that reached a step boundary in the last block:
//                  finish the chunk record (one float64 division), emit
//                  reward/obs/done, take the next action (fused mode) or retire;
//                - PREFETCH bandwidth*dt of the current and next interval and the
//                  boundary tick, so that
//   tick loop    is straight-line predicated VALU code with no memory access and
//                no divergent branch: ~40 vector instructions per tick per wave.
//                `__any(running)` ends it early when the whole wave is waiting.
// Lanes that hit their boundary mid-block idle until the block ends (T/2 ticks
// on average, against ~400 ticks per step).
struct Lane {
    double buf, dl, target, c, c_next;   // buffer_level, downloaded_size, target_size, bandwidth*dt
    long long sumk;
    int32_t k, k_end, chunk_id, n_su, n_rb, n_play, n_dl, j, tpos, tlen;
    int32_t avail_k, avail_next, cur_action, last_action;
    bool su, be, bf;        // start_up, buffer_empty, buffer_full
    bool done_dl;           // phase B
    bool running;           // ticking inside the current block
    const double *trace;
};

// T1-T3 of tick s.k (Simulator.py:137-149 with R2/R3); sets running = not at a call site
__device__ inline void lane_head(Lane &s) {
    s.n_su += s.su ? 1 : 0;
    s.n_rb += (!s.su && s.be) ? 1 : 0;
    const bool call = s.done_dl && (s.k >= s.avail_k) && !s.bf;
    s.running = !call;
}

__device__ inline void lane_init(Lane &s, const EnvParams &p, int32_t offset0) {
    // Simulator.py:95-130
    s.buf = 0.0; s.dl = 0.0; s.target = 0.0; s.sumk = 0;
    s.k = 0; s.chunk_id = 0; s.n_su = 0; s.n_rb = 0; s.n_play = 0; s.n_dl = 0;
    s.cur_action = -1; s.last_action = -1;
    s.su = true; s.be = true; s.bf = false;
    s.done_dl = true;                  // nothing is downloading: wait for the first call site
    s.j = 0;                           // int(0.0 / interval) == 0
    s.tpos = offset0 % s.tlen;
    s.avail_k = p.avail_tick[0];
    s.avail_next = s.avail_k;
    lane_head(s);                      // T1-T3 of tick 0
}

// Per-block refresh of the interval constants.  Moves j/tpos forward while k has
// passed the interval end (also covers intervals shorter than one tick), then
// loads bandwidth*dt for the current and the next interval.
__device__ inline void lane_refresh_interval(Lane &s, const EnvParams &p) {
    int32_t ke = p.interval_tick[s.j + 1];
    while (s.k >= ke) {
        s.j++;
        s.tpos = (s.tpos + 1 == s.tlen) ? 0 : s.tpos + 1;
        ke = p.interval_tick[s.j + 1];
    }
    s.k_end = ke;
    const int32_t tn = (s.tpos + 1 == s.tlen) ? 0 : s.tpos + 1;
    s.c = s.trace[s.tpos] * kDt;       // bandwidth * dt  (:160; the product is formed first)
    s.c_next = s.trace[tn] * kDt;
}

// average_latency (Simulator.py:179-180).  The reference's recurrence telescopes
// to sum(instant_latency) / play_time with instant_latency = global_time -
// play_time at each playing tick.  We carry the sum as exact integers: the m-th
// playing tick (m = 0..n_play-1) happened at tick k_m, so
//   sum = sum_m (G[k_m] - GP[m])  ~=  dt * sum_m k_m - sd * n_play (n_play - 1) / 2
// which differs from the reference only by float64 drift of the clocks
// (<= 1e-11 relative, measured in tests); it feeds no decision.
__device__ inline double avg_latency_from(double sd, double play_time, long long sumk, int32_t n_play) {
    if (n_play == 0) return 0.0;
    double tri = (double)(((long long)n_play * (n_play - 1)) / 2);
    return (kDt * (double)sumk - sd * tri) / play_time;
}
__device__ inline double lane_avg_latency(const EnvParams &p, long long sumk, int32_t n_play) {
    return avg_latency_from(p.sd, p.GP[n_play], sumk, n_play);
}
// with a speed schedule play_time is not n_play equal steps: the sum of play_time over the playing
// ticks is carried instead of the closed form sd * n(n-1)/2
__device__ inline double avg_latency_sched(double play_time, long long sumk, double pt_sum, int32_t n_play) {
    if (n_play == 0) return 0.0;
    return (kDt * (double)sumk - pt_sum) / play_time;
}

__device__ inline void lane_load(Lane &s, const EnvParams &p, int64_t i) {
    s.buf = p.buf[i]; s.sumk = p.sumk[i];
    s.k = p.k[i]; s.chunk_id = p.chunk_id[i]; s.n_su = p.n_su[i]; s.n_rb = p.n_rb[i];
    s.n_play = p.n_play[i]; s.j = p.j[i]; s.tpos = p.tpos[i];
    s.last_action = p.last_action[i]; s.cur_action = -1;
    const uint8_t f = p.flags[i];
    s.su = f & kFlagStartUp; s.be = f & kFlagBufEmpty; s.bf = f & kFlagBufFull;
    s.dl = 0.0; s.n_dl = 0; s.target = 0.0; s.done_dl = true; s.running = false;
    const int32_t c = s.chunk_id < p.video_length ? s.chunk_id : p.video_length;
    s.avail_k = p.avail_tick[c];
    s.avail_next = s.avail_k;
}

__device__ inline void lane_store(const Lane &s, const EnvParams &p, int64_t i) {
    p.buf[i] = s.buf; p.sumk[i] = s.sumk;
    p.k[i] = s.k; p.chunk_id[i] = s.chunk_id; p.n_su[i] = s.n_su; p.n_rb[i] = s.n_rb;
    p.n_play[i] = s.n_play; p.j[i] = s.j; p.tpos[i] = s.tpos; p.last_action[i] = s.last_action;
    p.flags[i] = (uint8_t)((s.su ? kFlagStartUp : 0) | (s.be ? kFlagBufEmpty : 0) |
                           (s.bf ? kFlagBufFull : 0) | kFlagArmed);
}

__device__ inline void write_obs(const Lane &s, const EnvParams &p, int64_t i, float *obs,
                                 double last_bw) {
    if (!obs) return;
    const int64_t n = p.n_lanes;
    obs[ABR_OBS_CHUNK_ID * n + i] = (float)s.chunk_id;
    obs[ABR_OBS_LAST_BITRATE * n + i] = (float)s.last_action;
    obs[ABR_OBS_LAST_BANDWIDTH * n + i] = (float)last_bw;
    obs[ABR_OBS_BUFFER_LEVEL * n + i] = (float)s.buf;
    obs[ABR_OBS_GLOBAL_TIME * n + i] = (float)p.G[s.k];
    obs[ABR_OBS_PLAY_TIME * n + i] = (float)p.GP[s.n_play];
    obs[ABR_OBS_REBUFFER_TIME * n + i] = (float)p.G[s.n_rb];
    obs[ABR_OBS_STARTUP_TIME * n + i] = (float)p.G[s.n_su];
}

// The variance term of calculate_qoe (Simulator.py:82-83: sum over the episode of |br[a_i] - br[a_(i+1)]|) is accumulated as the
// episode goes -- the same terms in the same order as the loop over previous_bitrates, hence the same float64 -- and left in
// ep_qoe_terms[3] when the episode ends.  (Rounds 1-4 copied the episode's 48 actions aside at every auto-reset for K4 to
// loop over: six dependent rounds of byte loads and stores per lane, ~10 us per episode end for the whole workgroup.)

// MODE 0: reset (fresh lanes run to their first call site)
// MODE 1: step  (one externally supplied action per lane)
// MODE 2: fused random-policy rollout of n_steps decisions per lane
// MODE 3: fused rollout of n_steps scripted decisions per lane, actions[step][lane]
template <int MODE>
__global__ __launch_bounds__(64) void env_advance_kernel(
    EnvParams p, const int32_t *__restrict__ actions, const int32_t *__restrict__ trace_id_in,
    const int32_t *__restrict__ offset_in, const uint8_t *__restrict__ lane_mask,
    float *__restrict__ obs_out, float *__restrict__ reward_out, uint8_t *__restrict__ done_out,
    int32_t *__restrict__ actions_out, int32_t n_steps, uint64_t seed) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool in_range = i < p.n_lanes;
    const int32_t n_total = (MODE >= 2) ? n_steps : 1;
    Lane s;
    bool active = in_range;       // still has work in this launch
    bool touched = in_range;      // state must be written back
    bool fresh = (MODE == 0);     // running a new episode up to its first call site
    bool need_action = false;     // at a call site, waiting for the block loop to hand it an action
    uint8_t done = 0;
    int32_t n_su_obs = 0, n_rb_obs = 0, episode_no = 0, offset0 = 0, prev_action = -1;
    double last_bw = 0.0, hist_n = 0.0, hist_s = 0.0, var_run = 0.0;
    int32_t step_idx = 0;
    s.running = false; s.done_dl = true;

    if (in_range) {
        if (MODE == 0) {
            if (lane_mask && !lane_mask[i]) { active = false; touched = false; }
            if (active) {
                int32_t t = trace_id_in[i];
                offset0 = offset_in ? offset_in[i] : 0;
                // a trace id / start offset outside the tables: the lane is frozen with
                // ABR_DONE_BADARG instead of reading out of bounds (it runs on trace 0, unseen)
                if (t < 0 || t >= p.n_traces || offset0 < 0) { done |= ABR_DONE_BADARG; t = 0; offset0 = 0; }
                // a re-armed lane starts a NEW episode of the counter-based policy
                episode_no = (p.flags[i] & kFlagArmed) ? p.episode_no[i] + 1 : 0;
                p.trace_id[i] = t; p.offset0[i] = offset0;
                s.tlen = p.trace_len[t]; s.trace = p.traces + p.trace_off[t];
                lane_init(s, p, offset0);
                if (done) {            // frozen (BADARG): still report the fresh lane's observation, as every implementation does
                    write_obs(s, p, i, obs_out, 0.0);
                    active = false; s.running = false;
                }
            }
        } else {
            done = p.done[i];
            const int32_t t = p.trace_id[i];
            offset0 = p.offset0[i];
            s.tlen = p.trace_len[t]; s.trace = p.traces + p.trace_off[t];
            lane_load(s, p, i);
            n_su_obs = p.n_su_obs[i]; n_rb_obs = p.n_rb_obs[i]; episode_no = p.episode_no[i];
            last_bw = p.last_bw[i]; hist_n = p.hist_n[i]; hist_s = p.hist_s[i]; var_run = p.var_run[i];
            if (done) active = false;
            need_action = active;        // a live lane sits at a call site
        }
    }

    const double L = p.chunk_length, sd = p.sd, max_buffer = p.max_buffer, sul = p.start_up_length;
    const int32_t V = p.video_length, max_ticks = p.max_ticks, T = p.t_block;

    while (__any(active)) {
        // ================= SERVICE: lanes at a step boundary =================
        if (active && !s.running) {
            const bool ended = s.chunk_id >= V;
            const bool timeout = !ended && (s.k >= max_ticks);
            bool emit_obs = !need_action;      // need_action: nothing happened yet in this launch
            if (!need_action && !fresh) {
                const int64_t o = (int64_t)step_idx * p.n_lanes + i;
                double var = 0.0;
                if (s.done_dl && s.cur_action >= 0) {
                    // the chunk this step downloaded (:164-165)
                    const double bw = s.dl / p.G[s.n_dl];
                    const int64_t h = (int64_t)(s.chunk_id - 1) * p.n_lanes + i;
                    p.bw_hist[h] = bw;
                    p.action_hist[h] = (uint8_t)s.cur_action;
                    last_bw = bw;
                    hist_s = hist_s + 1.0 / bw;        // sum(1/x), list order (mpc.py:86-88)
                    hist_n = hist_n + 1.0;
                    s.last_action = s.cur_action;
                    if (prev_action >= 0)
                        var = fabs(chunk_bitrate(p, s.chunk_id - 1, s.cur_action) -
                                   chunk_bitrate(p, s.chunk_id - 2, prev_action));
                    var_run = var_run + var;
                }
                // per-step split of calculate_qoe (Simulator.py:83-85)
                const double r = p.wr * (p.G[s.n_rb] - p.G[n_rb_obs]) +
                                 p.ws * (p.G[s.n_su] - p.G[n_su_obs]) + p.wv * var;
                if (ended) done |= ABR_DONE_EPISODE;
                if (timeout) done |= ABR_DONE_TIMEOUT;
                if (reward_out) reward_out[o] = (float)r;
                if (done_out) done_out[o] = done;
                n_su_obs = s.n_su; n_rb_obs = s.n_rb;
                s.cur_action = -1;
                if (ended || timeout) {
                    p.ep_qoe_terms[0 * p.n_lanes + i] = p.G[s.n_rb];
                    p.ep_qoe_terms[1 * p.n_lanes + i] = p.G[s.n_su];
                    p.ep_qoe_terms[2 * p.n_lanes + i] = lane_avg_latency(p, s.sumk, s.n_play);
                    p.ep_qoe_terms[3 * p.n_lanes + i] = var_run;
                    if (p.auto_reset && ended) {
                        // re-arm: this step's obs is the new episode's first call site
                        lane_init(s, p, offset0);
                        episode_no++;
                        n_su_obs = 0; n_rb_obs = 0;
                        last_bw = 0.0; hist_n = 0.0; hist_s = 0.0; var_run = 0.0;
                        done = 0; fresh = true;
                        emit_obs = !s.running;     // already at a call site only if avail_tick[0] == 0
                    }
                }
            } else if (!need_action && fresh && timeout) {
                done |= ABR_DONE_TIMEOUT;
                if (MODE != 0 && done_out) done_out[(int64_t)step_idx * p.n_lanes + i] = done;
            }
            if (emit_obs) {
                fresh = false;
                float *obs = obs_out ? obs_out + (int64_t)step_idx * ABR_OBS_DIM * p.n_lanes : nullptr;
                write_obs(s, p, i, obs, last_bw);
                step_idx++;
                if (done || step_idx >= n_total) active = false;
                else need_action = true;       // fused: take the next action right away
            }
            if (active && need_action) {
                // ---- the call site: get_next_bitrate's return value (Simulator.py:155-156) ----
                int32_t a;
                if (MODE == 1) a = actions[i];
                else if (MODE == 3) a = actions[(int64_t)step_idx * p.n_lanes + i];
                else a = (int32_t)philox_action(seed, (uint64_t)(p.lane_id_base + i),
                                                (uint32_t)s.chunk_id, (uint32_t)episode_no,
                                                (uint32_t)p.n_rates);
                if (MODE == 2 && actions_out) actions_out[(int64_t)step_idx * p.n_lanes + i] = a;
                need_action = false;
                if (a < 0 || a >= p.n_rates) {
                    done |= ABR_DONE_BADACT;
                    if (done_out) done_out[(int64_t)step_idx * p.n_lanes + i] = done;
                    if (reward_out) reward_out[(int64_t)step_idx * p.n_lanes + i] = 0.0f;
                    float *obs = obs_out ? obs_out + (int64_t)step_idx * ABR_OBS_DIM * p.n_lanes : nullptr;
                    write_obs(s, p, i, obs, last_bw);
                    step_idx++;
                    active = false;
                } else {
                    prev_action = s.last_action;
                    s.cur_action = a;
                    s.target = chunk_bitrate(p, s.chunk_id, a) * L;   // :156
                    s.dl = 0.0; s.n_dl = 0; s.done_dl = false;
                    s.avail_next = p.avail_tick[s.chunk_id + 1];
                    s.running = true;          // T4 of this very tick comes next
                }
            }
        }
        if (!__any(active)) break;
        // ================= PREFETCH interval constants =================
        if (active) lane_refresh_interval(s, p);

        // ================= TICK LOOP: predicated, no memory, no divergence =================
        double buf = s.buf, dl = s.dl, c = s.c;
        const double target = s.target, c_next = s.c_next;
        int32_t k = s.k, k_end = s.k_end, chunk_id = s.chunk_id, n_su = s.n_su, n_rb = s.n_rb;
        int32_t n_play = s.n_play, n_dl = s.n_dl, avail_k = s.avail_k;
        const int32_t avail_next = s.avail_next;
        uint32_t sk = 0;
        bool su = s.su, be = s.be, bf = s.bf, done_dl = s.done_dl;
        bool running = active && s.running;
        bool crossed = false;
        for (int32_t t = 0; t < T; t++) {
            if (!__any(running)) break;
            // ---- T4 download (Simulator.py:152-170) ----
            const bool dling = running && !done_dl;
            const double dl2 = dl + c;                               // :160
            dl = dling ? dl2 : dl;
            n_dl += dling ? 1 : 0;                                   // :161
            const bool hit = dling && (dl2 >= target);               // :163
            done_dl = done_dl || hit;
            chunk_id += hit ? 1 : 0;                                 // :166
            avail_k = hit ? avail_next : avail_k;
            const double bufA = hit ? buf + L : buf;                 // :170
            // ---- T5 playback (:174-187) ----
            const bool playing = running && !(be || su);
            sk += playing ? (uint32_t)k : 0u;                        // latency integral
            n_play += playing ? 1 : 0;
            const double bufB = playing ? bufA - sd : bufA;          // :184
            // ---- T6 buffer flags (:190-198) ----
            bf = bufB >= max_buffer;
            be = bufB <= 0.0;
            buf = be ? 0.0 : bufB;
            // ---- T7 start-up exit (:201-202) ----
            su = su && !(buf >= sul);
            // ---- T8/T9 clock, termination (:205-208) ----
            k += running ? 1 : 0;
            const bool next = running && (chunk_id < V) && (k < max_ticks);
            // ---- T1-T3 of the next tick (:137-149) ----
            n_su += (next && su) ? 1 : 0;
            n_rb += (next && !su && be) ? 1 : 0;
            const bool call = done_dl && (k >= avail_k) && !bf;
            running = next && !call;
            // ---- bandwidth of the next tick's interval (:158-159) ----
            const bool cross = (k == k_end);
            c = cross ? c_next : c;
            crossed = crossed || cross;
        }
        if (active) {
            s.buf = buf; s.dl = dl; s.c = c; s.k = k; s.chunk_id = chunk_id; s.n_su = n_su;
            s.n_rb = n_rb; s.n_play = n_play; s.n_dl = n_dl; s.avail_k = avail_k;
            s.su = su; s.be = be; s.bf = bf; s.done_dl = done_dl; s.running = running;
            s.sumk += (long long)sk;
            (void)crossed;   // j/tpos catch up in lane_refresh_interval
        }
    }

    if (touched) {
        lane_store(s, p, i);
        p.n_su_obs[i] = n_su_obs; p.n_rb_obs[i] = n_rb_obs; p.episode_no[i] = episode_no;
        p.last_bw[i] = last_bw; p.hist_n[i] = hist_n; p.hist_s[i] = hist_s; p.var_run[i] = var_run;
        p.done[i] = done;
        if (MODE != 0) {
            // lanes that were already finished (or finished early in a fused launch)
            // report their terminal record for the remaining steps
            for (int32_t t = step_idx; t < n_total; t++) {
                const int64_t o = (int64_t)t * p.n_lanes + i;
                if (reward_out) reward_out[o] = 0.0f;
                if (done_out) done_out[o] = done;
                if (MODE == 2 && actions_out) actions_out[o] = -1;
                if (obs_out)
                    write_obs(s, p, i, obs_out + (int64_t)t * ABR_OBS_DIM * p.n_lanes, last_bw);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// K1/K2, event-driven form, one thread per lane: abr_lane_jump.h does the per-lane work
// ---------------------------------------------------------------------------
using abrx::LaneJ;

// cascade: whether the drains of this kernel go through the per-binade cascade (abr_exact_jump.h: drain_cascade) where it
// applies.  Same results either way; measured per kernel (profiles/r06_ab_cascade.txt): +3.4 % for one thread per lane at
// 1 M lanes, +2.5 % for the two-wave role split at 131 072 -- and -2 % for the three-wave kernel at 65 536, whose player
// wave is not what an iteration waits for (fewer vector instructions, the same iteration time, more scalar traffic).
__device__ inline abrx::Tables make_tables(const EnvParams &p, bool cascade = true) {
    abrx::Tables t;
    t.G = p.G; t.interval_tick = p.interval_tick; t.avail_tick = p.avail_tick;
    t.L = p.chunk_length; t.sd = p.sd; t.max_buffer = p.max_buffer;
    t.start_up_length = p.start_up_length; t.V = p.video_length; t.max_ticks = p.max_ticks;
    t.per_lane_speed = p.lane_speeds != nullptr;
    t.speed_rows = p.lane_speeds ? p.speed_rows : 0;
    t.speed_stride = p.n_lanes; t.speeds = p.lane_speeds;
    t.drain = p.drain;
    if (p.lane_speeds || !cascade) t.drain.n = 0;      // per-lane speeds: every lane has a subtrahend of its own
#ifdef ABR_NO_DRAIN_CASCADE
    t.drain.n = 0;                 // A/B knob (tools/diag/csrc: make AB_FLAGS=-DABR_NO_DRAIN_CASCADE ab): round 5's drains
#endif
    return t;
}

__device__ inline void lanej_load(LaneJ &s, const EnvParams &p, int64_t i) {
    s.buf = p.buf[i]; s.sumk = p.sumk[i];
    s.k = p.k[i]; s.chunk_id = p.chunk_id[i]; s.n_su = p.n_su[i]; s.n_rb = p.n_rb[i];
    s.n_play = p.n_play[i]; s.cur.j = p.j[i]; s.cur.tpos = p.tpos[i];
    s.last_action = p.last_action[i];
    const uint8_t f = p.flags[i];
    s.su = f & kFlagStartUp; s.be = f & kFlagBufEmpty; s.bf = f & kFlagBufFull;
    const int32_t c = s.chunk_id < p.video_length ? s.chunk_id : p.video_length;
    s.avail_k = p.avail_tick[c];
    if (p.lane_speeds) { s.sd = p.sd_lane[i]; s.pt = p.pt_lane[i]; } else { s.sd = p.sd; s.pt = 0.0; }
    s.lane = i; s.pl_left = 0; s.play_id = 0; s.pt_sum = 0.0;
    if (p.lane_speeds && p.speed_rows >= 2) { s.pl_left = p.pl_left[i]; s.play_id = p.play_id[i]; s.pt_sum = p.pt_sum[i]; }
}

__device__ inline void lanej_store(const LaneJ &s, const EnvParams &p, int64_t i) {
    p.buf[i] = s.buf; p.sumk[i] = s.sumk;
    p.k[i] = s.k; p.chunk_id[i] = s.chunk_id; p.n_su[i] = s.n_su; p.n_rb[i] = s.n_rb;
    p.n_play[i] = s.n_play; p.j[i] = s.cur.j; p.tpos[i] = s.cur.tpos; p.last_action[i] = s.last_action;
    p.flags[i] = (uint8_t)((s.su ? kFlagStartUp : 0) | (s.be ? kFlagBufEmpty : 0) |
                           (s.bf ? kFlagBufFull : 0) | kFlagArmed);
    if (p.lane_speeds) { p.sd_lane[i] = s.sd; p.pt_lane[i] = s.pt; }
    if (p.lane_speeds && p.speed_rows >= 2) { p.pl_left[i] = s.pl_left; p.play_id[i] = s.play_id; p.pt_sum[i] = s.pt_sum; }
}

// Outputs (observations, rewards, done bytes, the actions and history rows) are written once and never read by the launch that
// writes them: non-temporal stores, so that they do not evict the tick tables, traces and lane state from the XCD's L2.  Same-box
// A/B (profiles/r05_ab_nt_stores.txt): +1.0 % at 65 536 lanes (three-wave kernel's service role), +0.6 % at 131 072 (two-wave
// kernel's player), +2.7 % at 1 048 576 (one thread per lane: 1.9 GB of outputs per launch).  -DABR_NO_NT_STORES builds the plain form.
#ifndef ABR_NO_NT_STORES
#define ABR_OUT(ref, val) __builtin_nontemporal_store((val), &(ref))
#else
#define ABR_OUT(ref, val) ((ref) = (val))
#endif

__device__ inline void write_obs_j(const LaneJ &s, const EnvParams &p, int64_t i, float *obs,
                                   double last_bw) {
    if (!obs) return;
    const int64_t n = p.n_lanes;
    ABR_OUT(obs[ABR_OBS_CHUNK_ID * n + i], (float)s.chunk_id);
    ABR_OUT(obs[ABR_OBS_LAST_BITRATE * n + i], (float)s.last_action);
    ABR_OUT(obs[ABR_OBS_LAST_BANDWIDTH * n + i], (float)last_bw);
    ABR_OUT(obs[ABR_OBS_BUFFER_LEVEL * n + i], (float)s.buf);
    ABR_OUT(obs[ABR_OBS_GLOBAL_TIME * n + i], (float)p.G[s.k]);
    ABR_OUT(obs[ABR_OBS_PLAY_TIME * n + i], (float)(p.lane_speeds ? s.pt : p.GP[s.n_play]));
    ABR_OUT(obs[ABR_OBS_REBUFFER_TIME * n + i], (float)p.G[s.n_rb]);
    ABR_OUT(obs[ABR_OBS_STARTUP_TIME * n + i], (float)p.G[s.n_su]);
}

// the same rows from table values the caller has already loaded: G[k], play_time, G[n_rb], G[n_su]
__device__ inline void write_obs_vals(const LaneJ &s, const EnvParams &p, int64_t i, float *obs, double last_bw,
                                      double g_k, double g_play, double g_rb, double g_su) {
    if (!obs) return;
    const int64_t n = p.n_lanes;
    ABR_OUT(obs[ABR_OBS_CHUNK_ID * n + i], (float)s.chunk_id);
    ABR_OUT(obs[ABR_OBS_LAST_BITRATE * n + i], (float)s.last_action);
    ABR_OUT(obs[ABR_OBS_LAST_BANDWIDTH * n + i], (float)last_bw);
    ABR_OUT(obs[ABR_OBS_BUFFER_LEVEL * n + i], (float)s.buf);
    ABR_OUT(obs[ABR_OBS_GLOBAL_TIME * n + i], (float)g_k);
    ABR_OUT(obs[ABR_OBS_PLAY_TIME * n + i], (float)g_play);
    ABR_OUT(obs[ABR_OBS_REBUFFER_TIME * n + i], (float)g_rb);
    ABR_OUT(obs[ABR_OBS_STARTUP_TIME * n + i], (float)g_su);
}

// Waves per SIMD the one-thread-per-lane kernels are compiled for.  Round 2: four (128 VGPRs + 48 B of scratch) instead
// of three at 138 VGPRs, +8 % at 262 144 lanes, +5 % at 1 M; a fifth then cost 30 % in spills.  Round 4's shorter segment
// needs 103 VGPRs, so five waves (<= 102 VGPRs) cost one spilled register: +2 % at 131 072 lanes, +4 % at 1 M (1.89e10
// env-steps/s); six (<= 85) -7 % (profiles/r04_ab_download_loop.txt (6))
#ifndef ABR_JUMP_WAVES
#define ABR_JUMP_WAVES 5
#endif
// MODE 1 (ONE decision per launch: abr_env_step, the K1 launches of abr_env_step_mpc -- what `auto` runs at every size) is
// compiled for four waves: a single pass through a decision gains nothing from the fifth wave, and the 102-VGPR bound
// cost it 12 B of scratch per lane (round 5).
#define ABR_JUMP_BOUNDS(MODE) __launch_bounds__(64, ((MODE) == 1 ? 4 : ABR_JUMP_WAVES))
template <int MODE>
__global__ ABR_JUMP_BOUNDS(MODE) void env_jump_kernel(
    EnvParams p, const int32_t *__restrict__ actions, const int32_t *__restrict__ trace_id_in,
    const int32_t *__restrict__ offset_in, const uint8_t *__restrict__ lane_mask,
    float *__restrict__ obs_out, float *__restrict__ reward_out, uint8_t *__restrict__ done_out,
    int32_t *__restrict__ actions_out, int32_t n_steps, uint64_t seed) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool in_range = i < p.n_lanes;
    const int32_t n_total = (MODE >= 2) ? n_steps : 1;
    const abrx::Tables tb = make_tables(p);
    LaneJ s;
    bool active = in_range, touched = in_range;
    uint8_t done = 0;
    int32_t n_su_obs = 0, n_rb_obs = 0, episode_no = 0, offset0 = 0;
    double last_bw = 0.0, hist_n = 0.0, hist_s = 0.0, var_run = 0.0;
    double g_su_obs = 0.0, g_rb_obs = 0.0;    // G[n_su_obs], G[n_rb_obs] carried in registers

    if (in_range) {
        if (MODE == 0) {
            if (lane_mask && !lane_mask[i]) { active = false; touched = false; }
            if (active) {
                int32_t t = trace_id_in[i];
                offset0 = offset_in ? offset_in[i] : 0;
                // a trace id / start offset outside the tables: the lane is frozen with
                // ABR_DONE_BADARG instead of reading out of bounds (it sits on trace 0, unseen)
                const bool bad = t < 0 || t >= p.n_traces || offset0 < 0;
                if (bad) { done |= ABR_DONE_BADARG; t = 0; offset0 = 0; }
                // a re-armed lane starts a NEW episode of the counter-based policy
                episode_no = (p.flags[i] & kFlagArmed) ? p.episode_no[i] + 1 : 0;
                p.trace_id[i] = t; p.offset0[i] = offset0;
                s.cur.tlen = p.trace_len[t]; s.cur.trace = p.traces + p.trace_off[t];
                s.sd = p.lane_speeds ? p.lane_speeds[i] * kDt : p.sd;     // play_speed * dt (:182)
                s.lane = i;
                abrx::lanej_init(s, tb, offset0);
                if (!bad && !abrx::lanej_wait_call(s, tb)) done |= ABR_DONE_TIMEOUT;
                write_obs_j(s, p, i, obs_out, 0.0);
            }
        } else {
            done = p.done[i];
            const int32_t t = p.trace_id[i];
            offset0 = p.offset0[i];
            s.cur.tlen = p.trace_len[t]; s.cur.trace = p.traces + p.trace_off[t];
            lanej_load(s, p, i);
            n_su_obs = p.n_su_obs[i]; n_rb_obs = p.n_rb_obs[i]; episode_no = p.episode_no[i];
            last_bw = p.last_bw[i]; hist_n = p.hist_n[i]; hist_s = p.hist_s[i]; var_run = p.var_run[i];
            g_su_obs = p.G[n_su_obs]; g_rb_obs = p.G[n_rb_obs];
            if (done) active = false;
        }
    }

    if (MODE != 0) {
        for (int32_t step = 0; step < n_total; step++) {
            const int64_t o = (int64_t)step * p.n_lanes + i;
            float *obs = obs_out ? obs_out + (int64_t)step * ABR_OBS_DIM * p.n_lanes : nullptr;
            if (active) {
                // one burst of loads for everything the step needs (issued before the
                // policy arithmetic so that the two overlap)
                const abrx::StepStart st = abrx::lanej_begin_step(s.cur, tb, s.k, s.chunk_id);
                // ---- the call site: get_next_bitrate's return value (Simulator.py:155-156) ----
                int32_t a;
                if (MODE == 1) a = actions[i];
                else if (MODE == 3) a = actions[o];
                else a = (int32_t)philox_action(seed, (uint64_t)(p.lane_id_base + i),
                                                (uint32_t)s.chunk_id, (uint32_t)episode_no,
                                                (uint32_t)p.n_rates);
                if (MODE == 2 && actions_out) ABR_OUT(actions_out[o], a);
                if (a < 0 || a >= p.n_rates) {
                    done |= ABR_DONE_BADACT;
                    if (reward_out) ABR_OUT(reward_out[o], 0.0f);
                    if (done_out) ABR_OUT(done_out[o], (uint8_t)done);
                    write_obs_j(s, p, i, obs, last_bw);
                    active = false;
                } else {
                    const int32_t prev_action = s.last_action;
                    const int32_t chunk = s.chunk_id;
                    const abrx::StepResult r = abrx::lanej_download_and_wait(
                        s, tb, st, chunk_bitrate(p, chunk, a) * p.chunk_length /* :156 */, a);
                    // the reward's two clocks and the observation's four tick-table values in ONE burst of loads
                    const double g_rb = p.G[s.n_rb], g_su = p.G[s.n_su];
                    double o_k = p.G[s.k], o_pl = p.lane_speeds ? s.pt : p.GP[s.n_play], o_rb = g_rb, o_su = g_su;
                    double var = 0.0;
                    if (r.hit) {
                        const int64_t h = (int64_t)chunk * p.n_lanes + i;
                        ABR_OUT(p.bw_hist[h], r.bw);                            // :164
                        ABR_OUT(p.action_hist[h], (uint8_t)a);                  // :165
                        last_bw = r.bw;
                        hist_s = hist_s + 1.0 / r.bw;   // sum(1/x), list order (mpc.py:86-88)
                        hist_n = hist_n + 1.0;
                        if (prev_action >= 0)
                            var = fabs(chunk_bitrate(p, chunk, a) - chunk_bitrate(p, chunk - 1, prev_action));
                        var_run = var_run + var;
                    }
                    // ---- step boundary: per-step split of calculate_qoe (:83-85) ----
                    const double rew = p.wr * (g_rb - g_rb_obs) + p.ws * (g_su - g_su_obs) + p.wv * var;
                    if (r.ended) done |= ABR_DONE_EPISODE;
                    if (r.timeout) done |= ABR_DONE_TIMEOUT;
                    if (reward_out) ABR_OUT(reward_out[o], (float)rew);
                    if (done_out) ABR_OUT(done_out[o], (uint8_t)done);
                    n_su_obs = s.n_su; n_rb_obs = s.n_rb;
                    g_su_obs = g_su; g_rb_obs = g_rb;
                    if (r.ended || r.timeout) {
                        p.ep_qoe_terms[0 * p.n_lanes + i] = g_rb;
                        p.ep_qoe_terms[1 * p.n_lanes + i] = g_su;
                        p.ep_qoe_terms[2 * p.n_lanes + i] =
                            !p.lane_speeds ? lane_avg_latency(p, s.sumk, s.n_play)
                            : (p.speed_rows >= 2 ? avg_latency_sched(s.pt, s.sumk, s.pt_sum, s.n_play)
                                                 : avg_latency_from(s.sd, s.pt, s.sumk, s.n_play));
                        p.ep_qoe_terms[3 * p.n_lanes + i] = var_run;
                        if (p.auto_reset && r.ended) {
                            // re-arm: this step's obs is the new episode's first call site
                            abrx::lanej_init(s, tb, offset0);
                            episode_no++;
                            n_su_obs = 0; n_rb_obs = 0; g_su_obs = 0.0; g_rb_obs = 0.0;
                            last_bw = 0.0; hist_n = 0.0; hist_s = 0.0; var_run = 0.0;
                            done = 0;
                            if (!abrx::lanej_wait_call(s, tb)) done |= ABR_DONE_TIMEOUT;
                            o_k = p.G[s.k]; o_pl = p.lane_speeds ? s.pt : p.GP[s.n_play];
                            o_rb = p.G[s.n_rb]; o_su = p.G[s.n_su];
                        }
                    }
                    write_obs_vals(s, p, i, obs, last_bw, o_k, o_pl, o_rb, o_su);
                    if (done) active = false;
                }
            } else if (in_range) {
                // lanes already finished report their terminal record again
                if (reward_out) ABR_OUT(reward_out[o], 0.0f);
                if (done_out) ABR_OUT(done_out[o], (uint8_t)done);
                if (MODE == 2 && actions_out) ABR_OUT(actions_out[o], (int32_t)-1);
                write_obs_j(s, p, i, obs, last_bw);
            }
        }
    }

    if (touched) {
        lanej_store(s, p, i);
        p.n_su_obs[i] = n_su_obs; p.n_rb_obs[i] = n_rb_obs; p.episode_no[i] = episode_no;
        p.last_bw[i] = last_bw; p.hist_n[i] = hist_n; p.hist_s[i] = hist_s; p.var_run[i] = var_run;
        p.done[i] = done;
    }
}

#include "abr_env_roles.h"     // K1, role-split form: env_split3_kernel (three waves per 64 lanes), env_split_kernel (two)
// Diagnostic builds only (tools/diag/csrc; `auto` never picks them, the product library does not contain them):
#ifdef ABR_WITH_RING
#include "abr_env_ring.h"      // the role pipeline coupled by LDS rings instead of a per-iteration barrier (round 5, impl 6)
#include "abr_env_pair.h"      // download and player wave in lock-step, the service wave behind a ring (round 5, impl 7)
#endif
#ifdef ABR_WITH_ASYNC
#include "abr_env_async.h"     // the asynchronous pipeline of round 3 (impl 4)
#endif

// K4: calculate_qoe in the reference's operation order (Simulator.py:79-86)
__global__ void episode_qoe_kernel(EnvParams p, double *__restrict__ qoe_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_lanes) return;
    double variance = 0.0;
    if (p.auto_reset) {
        // previous_bitrates now holds the NEXT episode's actions: the finished episode's sum was left here when it ended
        variance = p.ep_qoe_terms[3 * p.n_lanes + i];
    } else {
        for (int c = 0; c < p.video_length - 1; c++) {
            int a0 = p.action_hist[(int64_t)c * p.n_lanes + i], a1 = p.action_hist[(int64_t)(c + 1) * p.n_lanes + i];
            variance += fabs(chunk_bitrate(p, c, a0) - chunk_bitrate(p, c + 1, a1));   // :82
        }
    }
    qoe_out[i] = p.wr * p.ep_qoe_terms[0 * p.n_lanes + i] + p.wv * variance +
                 p.ws * p.ep_qoe_terms[1 * p.n_lanes + i] +
                 p.wl * p.ep_qoe_terms[2 * p.n_lanes + i];               // :83-86
}

__global__ void observe_f64_kernel(EnvParams p, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_lanes) return;
    const int64_t n = p.n_lanes;
    out[ABR_F64_GLOBAL_TIME * n + i] = p.G[p.k[i]];
    out[ABR_F64_REBUFFER_TIME * n + i] = p.G[p.n_rb[i]];
    out[ABR_F64_STARTUP_TIME * n + i] = p.G[p.n_su[i]];
    int32_t P = p.play_ticks_per_chunk;
    const bool sched = p.lane_speeds && p.speed_rows >= 2;
    if (p.lane_speeds) {
        // per-lane speed: ticks per played chunk = additions of speed*dt from 0 until >= L (:185)
        double x = 0.0; int32_t a = 0;
        abrx::chain<abrx::STOP_GE>(x, p.sd_lane[i], p.chunk_length, p.max_ticks + 1, a);
        P = a > 0 ? a : 1;
        out[ABR_F64_PLAY_TIME * n + i] = p.pt_lane[i];
        out[ABR_F64_AVERAGE_LATENCY * n + i] =
            sched ? avg_latency_sched(p.pt_lane[i], p.sumk[i], p.pt_sum[i], p.n_play[i])
                  : avg_latency_from(p.sd_lane[i], p.pt_lane[i], p.sumk[i], p.n_play[i]);
    } else {
        out[ABR_F64_PLAY_TIME * n + i] = p.GP[p.n_play[i]];
        out[ABR_F64_AVERAGE_LATENCY * n + i] = lane_avg_latency(p, p.sumk[i], p.n_play[i]);
    }
    out[ABR_F64_BUFFER_LEVEL * n + i] = p.buf[i];
    // play_length restarts from 0 every P playing ticks (:185-187); play_id counts the restarts
    if (sched) {
        // ticks already played of the current played chunk at its speed (0 between chunks)
        const int32_t pl = p.pl_left[i];
        double x = 0.0; int32_t a = 0;
        abrx::chain<abrx::STOP_GE>(x, p.sd_lane[i], 1.0e300, pl > 0 ? P - pl : 0, a);
        out[ABR_F64_PLAY_LENGTH * n + i] = x;
    } else if (p.lane_speeds) {
        double x = 0.0; int32_t a = 0;
        abrx::chain<abrx::STOP_GE>(x, p.sd_lane[i], 1.0e300, p.n_play[i] % P, a);
        out[ABR_F64_PLAY_LENGTH * n + i] = x;
    } else {
        out[ABR_F64_PLAY_LENGTH * n + i] = p.GP[p.n_play[i] % P];
    }
    out[ABR_F64_LAST_BANDWIDTH * n + i] = p.last_bw[i];
    out[ABR_F64_CHUNK_ID * n + i] = (double)p.chunk_id[i];
    out[ABR_F64_PLAY_ID * n + i] = sched ? (double)p.play_id[i] : (double)(p.n_play[i] / P);
    out[ABR_F64_LAST_BITRATE * n + i] = (double)p.last_action[i];
    out[ABR_F64_FLAGS * n + i] = (double)p.flags[i];
    out[ABR_F64_HIST_N * n + i] = p.hist_n[i];
    out[ABR_F64_HIST_SUM_INV * n + i] = p.hist_s[i];
    out[ABR_F64_TICK * n + i] = (double)p.k[i];
    out[ABR_F64_DOWNLOAD_TIME * n + i] = 0.0;   // download_time is 0 at every call site (:154)
}

// ---------------------------------------------------------------------------
// host side of the env ABI
// ---------------------------------------------------------------------------
static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
// scratch of the fused MPC rollout's predictor kernel, sized for the longest horizon
static size_t mpc_scratch_bytes_max(size_t n_lanes) {
    return n_lanes * ((size_t)ABR_MAX_HORIZON * sizeof(double) + 2 * sizeof(int32_t));
}

struct Layout {
    size_t G, GP, interval_tick, avail_tick;        // table offsets
    size_t f64_state, i64_state, i32_state, u8_state, action_hist, bw_hist, ep_terms;
    size_t mpc_action, mpc_scratch;
    size_t tag;                                      // the layout tag: the workspace's LAST 256 bytes (WorkspaceTag)
    size_t total;
    int32_t max_ticks, n_intervals;
};

// Default per-episode tick bound: 32 x the live-stream minimum V * ceil(L/dt), capped at 16e6
// ticks (a 128 MB G table) but never below 2 x the minimum; compute_layout rejects a bound
// under which no episode could finish.
static double min_episode_ticks(const abr_env_config *c) {
    return ((double)c->video_length + 1.0) * ceil(c->chunk_length / kDt);
}
static double default_max_ticks(const abr_env_config *c) {
    double per_chunk = ceil(c->chunk_length / kDt);
    double mt = 32.0 * c->video_length * per_chunk;
    if (mt > 16.0e6) mt = 16.0e6;
    if (mt < 2.0 * min_episode_ticks(c)) mt = 2.0 * min_episode_ticks(c);
    if (mt < 1024) mt = 1024;
    return mt;
}

static int validate_cfg(const abr_env_config *c) {
    if (!c) return fail(ABR_E_INVALID, "config is NULL");
    if (c->n_rates < 1 || c->n_rates > ABR_MAX_RATES)
        return fail(ABR_E_INVALID, "n_rates %d outside 1..%d", c->n_rates, ABR_MAX_RATES);
    if (c->video_length < 1 || c->video_length > 65535)
        return fail(ABR_E_INVALID, "video_length %d outside 1..65535", c->video_length);
    if (!(c->chunk_length > 0) || !(c->interval > 0) || !(c->speed > 0))
        return fail(ABR_E_INVALID, "chunk_length, interval and speed must be > 0");
    if (!(c->max_buffer > 0)) return fail(ABR_E_INVALID, "max_buffer must be > 0");
    if (!(c->start_up_length >= 0)) return fail(ABR_E_INVALID, "start_up_length must be >= 0");
    if (c->chunk_length / kDt > 1.0e6)
        return fail(ABR_E_UNSUPPORTED, "chunk_length %g s is more than 1e6 ticks", c->chunk_length);
    for (int r = 0; r < c->n_rates; r++)
        if (!(c->ladder[r] > 0)) return fail(ABR_E_INVALID, "ladder[%d] must be > 0", r);
    return ABR_OK;
}

static int compute_layout(const abr_env_config *c, int64_t n_lanes, Layout *L) {
    int rc = validate_cfg(c);
    if (rc) return rc;
    if (n_lanes < 1) return fail(ABR_E_INVALID, "n_lanes must be >= 1");
    const double mtd = c->max_ticks > 0 ? (double)c->max_ticks : default_max_ticks(c);
    if (mtd > 256.0e6)
        return fail(ABR_E_UNSUPPORTED, "video_length %d x chunk_length %g s needs a %g-entry tick table",
                    c->video_length, c->chunk_length, mtd);
    if (mtd < min_episode_ticks(c))
        return fail(ABR_E_INVALID, "max_ticks %g is below (video_length + 1) * ceil(chunk_length / dt) = %g: "
                    "every lane would end in ABR_DONE_TIMEOUT", mtd, min_episode_ticks(c));
    const int32_t mt = (int32_t)mtd;
    double n_iv = (double)mt * kDt / c->interval + 4.0;
    if (n_iv > 32.0e6)
        return fail(ABR_E_UNSUPPORTED, "interval %g s needs a %g-entry tick table", c->interval, n_iv);
    L->max_ticks = mt;
    L->n_intervals = (int32_t)n_iv;
    size_t o = 0;
    const size_t A = 256;
    size_t N = (size_t)n_lanes, V = (size_t)c->video_length;
    L->G = o; o = align_up(o + sizeof(double) * ((size_t)mt + 2), A);
    L->GP = o; o = align_up(o + sizeof(double) * ((size_t)mt + 2), A);
    L->interval_tick = o; o = align_up(o + sizeof(int32_t) * ((size_t)L->n_intervals + 8), A);
    L->avail_tick = o; o = align_up(o + sizeof(int32_t) * (V + 2), A);
    L->f64_state = o; o = align_up(o + sizeof(double) * 8 * N, A);
    L->i64_state = o; o = align_up(o + sizeof(long long) * 1 * N, A);
    L->i32_state = o; o = align_up(o + sizeof(int32_t) * 15 * N, A);
    L->u8_state = o; o = align_up(o + 2 * N, A);
    L->action_hist = o; o = align_up(o + V * N, A);
    L->bw_hist = o; o = align_up(o + sizeof(double) * V * N, A);
    L->ep_terms = o; o = align_up(o + sizeof(double) * 4 * N, A);
    L->mpc_action = o; o = align_up(o + sizeof(int32_t) * N, A);
    L->mpc_scratch = o; o = align_up(o + mpc_scratch_bytes_max(N), A);
    L->tag = o; o = align_up(o + sizeof(WorkspaceTag), A);
    L->total = o;
    return ABR_OK;
}

extern "C" int abr_env_workspace_bytes(const abr_env_config *cfg, int64_t n_lanes,
                                       size_t *bytes_out) {
    Layout L;
    int rc = compute_layout(cfg, n_lanes, &L);
    if (rc) return rc;
    if (!bytes_out) return fail(ABR_E_INVALID, "bytes_out is NULL");
    *bytes_out = L.total;
    return ABR_OK;
}

extern "C" int abr_env_create(const abr_env_config *cfg, const double *traces_dev,
                              const int64_t *trace_off_dev, const int32_t *trace_len_dev,
                              int32_t n_traces, int64_t n_lanes, void *workspace_dev,
                              size_t workspace_bytes, void *stream, abr_env **env_out) {
    Layout L;
    int rc = compute_layout(cfg, n_lanes, &L);
    if (rc) return rc;
    if (!env_out) return fail(ABR_E_INVALID, "env_out is NULL");
    if (!traces_dev || !trace_off_dev || !trace_len_dev || n_traces < 1)
        return fail(ABR_E_INVALID, "trace arrays missing");
    if (!workspace_dev || ((uintptr_t)workspace_dev & 255))
        return fail(ABR_E_WORKSPACE, "workspace must be non-NULL and 256-B aligned");
    if (workspace_bytes < L.total)
        return fail(ABR_E_WORKSPACE, "workspace has %zu bytes, need %zu", workspace_bytes, L.total);

    // ---- universal tick tables, float64 on the host (abr_tick_tables.h) ----
    const int32_t mt = L.max_ticks;
    abrx::TickTables tt = abrx::build_tick_tables(cfg->interval, cfg->chunk_length, cfg->speed,
                                                  cfg->video_length, mt, L.n_intervals);
    std::vector<double> &G = tt.G, &GP = tt.GP;
    std::vector<int32_t> &itick = tt.interval_tick, &atick = tt.avail_tick;
    const double sd = tt.sd;
    const int32_t P = tt.play_ticks_per_chunk;
    int32_t t_block = tt.min_interval_ticks < 128 ? tt.min_interval_ticks : 128;
    if (t_block < 1) t_block = 1;
    if (P == INT_MAX)
        return fail(ABR_E_UNSUPPORTED, "max_ticks %d too small to play one chunk", mt);

    abr_env *e = new (std::nothrow) abr_env;
    if (!e) return fail(ABR_E_INVALID, "out of host memory");
    memset(e, 0, sizeof(*e));
    e->cfg = *cfg;
    e->impl = 3;
    e->workspace_bytes = L.total;
    EnvParams &p = e->p;
    p.n_rates = cfg->n_rates; p.video_length = cfg->video_length; p.max_ticks = mt;
    p.auto_reset = cfg->auto_reset; p.play_ticks_per_chunk = P; p.n_intervals = L.n_intervals;
    p.t_block = t_block; p.n_traces = n_traces;
    p.chunk_length = cfg->chunk_length; p.max_buffer = cfg->max_buffer;
    p.start_up_length = cfg->start_up_length;
    p.wr = cfg->rebuffer_weight; p.wv = cfg->variance_weight; p.ws = cfg->startup_weight;
    p.wl = cfg->latency_weight; p.sd = sd;
    for (int r = 0; r < ABR_MAX_RATES; r++) p.ladder[r] = cfg->ladder[r];
    p.n_lanes = n_lanes; p.lane_id_base = 0;
    char *w = (char *)workspace_dev;
    p.G = (const double *)(w + L.G); p.GP = (const double *)(w + L.GP);
    p.interval_tick = (const int32_t *)(w + L.interval_tick);
    p.avail_tick = (const int32_t *)(w + L.avail_tick);
    // buffer_level never exceeds max_buffer + chunk_length (a download only starts below max_buffer, :144); a level above
    // the cascade would send its wave through the general chains (abr_lane_jump.h: lanej_drain)
    p.drain = abrx::make_drain_tab(sd, cfg->max_buffer + cfg->chunk_length);
    p.traces = traces_dev; p.trace_off = trace_off_dev; p.trace_len = trace_len_dev;
    const size_t N = (size_t)n_lanes;
    double *f = (double *)(w + L.f64_state);
    p.buf = f; p.last_bw = f + N; p.hist_n = f + 2 * N; p.hist_s = f + 3 * N;
    p.sd_lane = f + 4 * N; p.pt_lane = f + 5 * N; p.pt_sum = f + 6 * N; p.var_run = f + 7 * N; p.lane_speeds = nullptr;
    p.speed_rows = 1;
    p.sumk = (long long *)(w + L.i64_state);
    int32_t *q = (int32_t *)(w + L.i32_state);
    p.k = q; p.chunk_id = q + N; p.n_su = q + 2 * N; p.n_rb = q + 3 * N; p.n_play = q + 4 * N;
    p.j = q + 5 * N; p.tpos = q + 6 * N; p.trace_id = q + 7 * N; p.offset0 = q + 8 * N;
    p.last_action = q + 9 * N; p.n_su_obs = q + 10 * N; p.n_rb_obs = q + 11 * N;
    p.episode_no = q + 12 * N; p.pl_left = q + 13 * N; p.play_id = q + 14 * N;
    uint8_t *u = (uint8_t *)(w + L.u8_state);
    p.flags = u; p.done = u + N;
    p.action_hist = (uint8_t *)(w + L.action_hist);
    p.bw_hist = (double *)(w + L.bw_hist);
    p.ep_qoe_terms = (double *)(w + L.ep_terms);
    e->mpc_action = (int32_t *)(w + L.mpc_action);
    e->mpc_scratch = (void *)(w + L.mpc_scratch);

    hipStream_t st = (hipStream_t)stream;
    hipError_t he;
#define UP(dst, vec)                                                                           \
    he = hipMemcpyAsync((void *)(dst), (vec).data(), (vec).size() * sizeof((vec)[0]),          \
                        hipMemcpyHostToDevice, st);                                            \
    if (he != hipSuccess) { delete e; return fail(ABR_E_HIP, "table upload: %s", hipGetErrorString(he)); }
    UP(p.G, G) UP(p.GP, GP) UP(p.interval_tick, itick) UP(p.avail_tick, atick)
#undef UP
    he = hipMemsetAsync(w + L.f64_state, 0, L.total - L.f64_state, st);
    if (he == hipSuccess) he = hipMemsetAsync(p.done, ABR_DONE_EPISODE, N, st);  // not reset yet
    {
        WorkspaceTag &tg = e->tag;
        memset(&tg, 0, sizeof(tg));
        tg.magic = kTagMagic; tg.abi_version = ABR_ABI_VERSION; tg.layout_rev = kLayoutRev; tg.n_rates = (uint32_t)cfg->n_rates;
        tg.n_lanes = n_lanes; tg.video_length = cfg->video_length; tg.max_ticks = mt; tg.n_intervals = L.n_intervals;
        tg.total_bytes = L.total; tg.chunk_length = cfg->chunk_length; tg.interval = cfg->interval; tg.speed = cfg->speed;
        e->tag_dev = w + L.tag;
        if (he == hipSuccess) he = hipMemcpyAsync(e->tag_dev, &tg, sizeof(tg), hipMemcpyHostToDevice, st);
    }
    if (he == hipSuccess) he = hipStreamSynchronize(st);   // host vectors die at return
    if (he != hipSuccess) { delete e; return fail(ABR_E_HIP, "workspace init: %s", hipGetErrorString(he)); }
    *env_out = e;
    return ABR_OK;
}

extern "C" int abr_env_destroy(abr_env *env) {
    delete env;
    return ABR_OK;
}

// which implementations this build of the library can run (abr_env_set_impl): the product holds 0, 1, 2, 3, 5
extern "C" int abr_env_has_impl(int32_t impl) {
    switch (impl) {
    case 0: case 1: case 2: case 3: case 5: return 1;
#ifdef ABR_WITH_ASYNC
    case 4: return 1;
#endif
#ifdef ABR_WITH_RING
    case 6: case 7: return 1;
#endif
    default: return 0;
    }
}

// 3 = auto (default), 5 / 2 = role-split event-driven kernels (three / two waves per 64 lanes),
// 0 = event-driven, one thread per lane, 1 = tick-by-tick kernels (kept as a cross-check)
extern "C" int abr_env_set_impl(abr_env *env, int32_t impl) {
    if (!env) return fail(ABR_E_INVALID, "env is NULL");
    if (impl < 0 || impl > 7)
        return fail(ABR_E_INVALID, "impl must be 0 (jump), 1 (tick), 2 (split), 3 (auto) or 5 (split3)");
    if (!abr_env_has_impl(impl))
        return fail(ABR_E_UNSUPPORTED, "impl %d (4: the asynchronous pipeline, 6: the ring-coupled role pipeline, 7: the pair rendezvous) is not part of "
                    "the product library: it is slower than what `auto` selects; the diagnostic build "
                    "tools/diag/lib/libabr_hip_diag.so carries it", impl);
    if (impl == 1 && (env->p.lane_speeds || (env->speeds_dirty && env->pending_speeds)))
        return fail(ABR_E_UNSUPPORTED, "the tick-by-tick kernels take one speed for all lanes");
    env->impl = impl;
    return ABR_OK;
}

// one constant play speed per lane (8f rank 3); nullptr restores the single config speed
extern "C" int abr_env_set_lane_speeds(abr_env *env, const double *speeds_dev) {
    if (!env) return fail(ABR_E_INVALID, "env is NULL");
    if (speeds_dev && env->impl == 1)
        return fail(ABR_E_UNSUPPORTED, "per-lane speeds need the event-driven kernels (impl 0 or 2)");
    // latched by the next FULL abr_env_reset: until then the running episodes keep the
    // speeds (and the carried play_time) they were started with
    env->pending_speeds = speeds_dev;
    env->pending_speed_rows = 1;
    env->speeds_dirty = true;
    if (!env->armed) apply_pending(env);
    return ABR_OK;
}

// a speed controller's answers, one row per played chunk (8f rank 3): speeds_dev [n_rows][n_lanes]
extern "C" int abr_env_set_speed_schedule(abr_env *env, const double *speeds_dev, int32_t n_rows) {
    if (!env) return fail(ABR_E_INVALID, "env is NULL");
    if (speeds_dev && env->impl == 1)
        return fail(ABR_E_UNSUPPORTED, "speed schedules need the event-driven kernels (impl 0 or 2)");
    if (speeds_dev && n_rows < 1) return fail(ABR_E_INVALID, "n_rows must be >= 1");
    env->pending_speeds = speeds_dev;
    env->pending_speed_rows = speeds_dev ? n_rows : 1;
    env->speeds_dirty = true;
    if (!env->armed) apply_pending(env);
    return ABR_OK;
}

// per-chunk bitrate ladders [video_length][n_rates] (8f rank 2); nullptr restores the config ladder
extern "C" int abr_env_set_bitrate_table(abr_env *env, const double *br_table_dev) {
    if (!env) return fail(ABR_E_INVALID, "env is NULL");
    env->pending_br_table = br_table_dev;
    env->br_table_dirty = true;
    if (!env->armed) apply_pending(env);
    return ABR_OK;
}

// the caller has copied a checkpointed workspace into this handle's workspace: episodes are in flight.  The copy must be what
// this handle's abr_env_create would have laid out: the tag in its last 256 bytes says so (read back here: this call
// synchronises the device, once per restore).
extern "C" int abr_env_notify_restore(abr_env *env) {
    if (!env) return fail(ABR_E_INVALID, "env is NULL");
    WorkspaceTag got;
    HIP_TRY(hipDeviceSynchronize());           // the caller's copy may still be in flight on any stream
    HIP_TRY(hipMemcpy(&got, env->tag_dev, sizeof(got), hipMemcpyDeviceToHost));
    const WorkspaceTag &w = env->tag;
    if (got.magic != kTagMagic)
        return fail(ABR_E_WORKSPACE, "the restored workspace carries no layout tag: it was not written by this library at ABI "
                    "version >= 4 (a version-2 / -3 checkpoint cannot be restored: the lane-state layout changed)");
    if (got.abi_version != w.abi_version || got.layout_rev != w.layout_rev)
        return fail(ABR_E_WORKSPACE, "the restored workspace was laid out by ABI version %u (layout %u), this library is %u (%u)",
                    got.abi_version, got.layout_rev, w.abi_version, w.layout_rev);
    if (got.n_lanes != w.n_lanes || got.video_length != w.video_length || got.n_rates != w.n_rates ||
        got.max_ticks != w.max_ticks || got.n_intervals != w.n_intervals || got.total_bytes != w.total_bytes ||
        got.chunk_length != w.chunk_length || got.interval != w.interval || got.speed != w.speed)
        return fail(ABR_E_WORKSPACE, "the restored workspace belongs to another configuration: %lld lanes, %d chunks, %u rates, "
                    "%d ticks, %llu bytes (this handle: %lld, %d, %u, %d, %llu)", (long long)got.n_lanes, got.video_length,
                    got.n_rates, got.max_ticks, (unsigned long long)got.total_bytes, (long long)w.n_lanes, w.video_length,
                    w.n_rates, w.max_ticks, (unsigned long long)w.total_bytes);
    apply_pending(env);
    env->armed = true;
    return ABR_OK;
}

// lane id base for the counter-based policy when lanes are a shard of a bigger job
extern "C" int abr_env_set_lane_id_base(abr_env *env, int64_t base) {
    if (!env) return fail(ABR_E_INVALID, "env is NULL");
    env->p.lane_id_base = base;
    return ABR_OK;
}

static inline unsigned grid64(int64_t n) { return (unsigned)((n + 63) / 64); }

// Which kernel serves which size (same box, fused 48, round 5: profiles/r05_sweep_impl.txt -- the role-split kernels gained
// 18 % from the exact call-site prediction, which moved the crossover up): the three-wave role split while all its waves are
// resident (3 x 1 024 waves at 65 536 lanes: 1.15e10 env-steps/s against 1.01e10 two-wave and 7.7e9 one thread per lane); the
// two-wave role split while all ITS waves are resident (2 x 2 048 waves at 131 072 lanes = four per SIMD: 1.39e10 against
// 1.24e10; 1.18e10 / 1.09e10 at 114 688, 1.02e10 / 9.3e9 at 98 304); one thread per lane above (1.22e10 against 1.11e10 at
// 163 840, 1.55e10 against 1.28e10 at 262 144): it then has three or more waves per SIMD of its own and no barrier.
constexpr int64_t kSplit3MaxLanes = 65536;
constexpr int64_t kSplitMaxLanes = 131072;
// The asynchronous pipeline (abr_env_async.h, impl 4) lost to the role-split kernels on MI355X (714 against 418 us
// per launch at 65 536 lanes, profiles/r03_async_*) and `auto` never picked it: the product library is built
// without it.  `make libabr_hip_async.so` (-DABR_WITH_ASYNC) keeps it selectable for the parity tests and records.
#ifdef ABR_WITH_ASYNC
static inline bool async_eligible(const abr_env *env) {
    return !env->p.lane_speeds && env->p.video_length + 2 <= kAvailLds;
}
#endif
// fused == true: more than one decision per launch (step_random / step_script with n_steps > 1)
static inline int effective_impl(const abr_env *env, bool fused = false) {
    int impl = env->impl;
    if (impl == 3) {
        // ONE decision per launch (abr_env_step, the K1 launches of abr_env_step_mpc, a fused call of one
        // step) is a single pass through download -> player -> service whatever the kernel, so the role
        // pipeline has nothing to overlap: one thread per lane is 3-8 % shorter per launch from 4 096 to
        // 65 536 lanes under the random policy and 30 % under MPC's actions (profiles/r04_sweeps_single_step.txt)
        if (!fused) return 0;
        if (env->p.n_lanes <= kSplit3MaxLanes) return 5;
        return env->p.n_lanes <= kSplitMaxLanes ? 2 : 0;
    }
#ifdef ABR_WITH_ASYNC
    if (impl == 4) return (fused && async_eligible(env)) ? 4 : 2;
#endif
    (void)fused;
    return impl;
}
static inline bool is_split(int impl) { return impl == 2 || impl == 5 || impl == 6 || impl == 7; }
// launch of the role-split kernels: two waves per 64 lanes (impl 2) or three (impl 5)
template <int MODE>
static void launch_split(int impl, const EnvParams &p, const int32_t *actions, float *obs, float *rew, uint8_t *dn,
                         int32_t *acts, int32_t n_steps, uint64_t seed, hipStream_t st) {
#ifdef ABR_WITH_RING
    if (impl == 7) {
        hipLaunchKernelGGL(env_pair3_kernel<MODE>, dim3(grid64(p.n_lanes)), dim3(192), 0, st, p, actions, obs, rew,
                           dn, acts, n_steps, seed);
        return;
    }
    if (impl == 6) {
        hipLaunchKernelGGL(env_ring3_kernel<MODE>, dim3(grid64(p.n_lanes)), dim3(64 * ABR_RING_WAVES), 0, st, p, actions, obs, rew,
                           dn, acts, n_steps, seed);
        return;
    }
#endif
    if (impl == 5)
        hipLaunchKernelGGL(env_split3_kernel<MODE>, dim3(grid64(p.n_lanes)), dim3(192), 0, st, p, actions, obs, rew,
                           dn, acts, n_steps, seed);
    else
        hipLaunchKernelGGL(env_split_kernel<MODE>, dim3(grid64(p.n_lanes)), dim3(128), 0, st, p, actions, obs, rew,
                           dn, acts, n_steps, seed);
}

extern "C" int abr_env_reset(abr_env *env, const int32_t *trace_id_dev,
                             const int32_t *start_offset_dev, const uint8_t *lane_mask_dev,
                             float *obs_out_dev, void *stream) {
    if (!env) return fail(ABR_E_INVALID, "env is NULL");
    if (!trace_id_dev) return fail(ABR_E_INVALID, "trace_id_dev is NULL");
    if (env->br_table_dirty && lane_mask_dev)
        return fail(ABR_E_INVALID, "abr_env_set_bitrate_table takes effect at a reset of ALL lanes "
                    "(lane_mask_dev must be NULL for the first reset after it)");
    if (env->speeds_dirty && lane_mask_dev)
        return fail(ABR_E_INVALID, "abr_env_set_lane_speeds takes effect at a reset of ALL lanes "
                    "(lane_mask_dev must be NULL for the first reset after it)");
    apply_pending(env);
    env->armed = true;
    hipLaunchKernelGGL(env->impl == 1 ? env_advance_kernel<0> : env_jump_kernel<0>, dim3(grid64(env->p.n_lanes)), dim3(64), 0,
                       (hipStream_t)stream, env->p, nullptr, trace_id_dev, start_offset_dev,
                       lane_mask_dev, obs_out_dev, nullptr, nullptr, nullptr, 0, 0ull);
    HIP_TRY(hipGetLastError());
    return ABR_OK;
}

extern "C" int abr_env_step(abr_env *env, const int32_t *actions_dev, float *obs_out_dev,
                            float *reward_out_dev, uint8_t *done_out_dev, void *stream) {
    if (!env) return fail(ABR_E_INVALID, "env is NULL");
    if (!actions_dev) return fail(ABR_E_INVALID, "actions_dev is NULL");
    const int impl = effective_impl(env);
    if (is_split(impl))
        launch_split<1>(impl, env->p, actions_dev, obs_out_dev, reward_out_dev, done_out_dev, nullptr, 1, 0ull,
                        (hipStream_t)stream);
    else
        hipLaunchKernelGGL(impl ? env_advance_kernel<1> : env_jump_kernel<1>, dim3(grid64(env->p.n_lanes)), dim3(64), 0,
                           (hipStream_t)stream, env->p, actions_dev, nullptr, nullptr, nullptr,
                           obs_out_dev, reward_out_dev, done_out_dev, nullptr, 1, 0ull);
    HIP_TRY(hipGetLastError());
    return ABR_OK;
}

// n_steps fused decisions per lane; MODE 2: built-in random policy, MODE 3: scripted actions
// [n_steps][n_lanes].
template <int MODE>
static int launch_fused(abr_env *env, const int32_t *script, int32_t n_steps, uint64_t seed, float *obs,
                        float *rew, uint8_t *dn, int32_t *acts, hipStream_t st) {
    const int impl = effective_impl(env, n_steps > 1);
    const int64_t N = env->p.n_lanes;
#ifdef ABR_WITH_ASYNC
    if (impl == 4) {
        // the pipeline takes at most kMaxFuse decisions per launch (its action table lives in LDS); longer
        // rollouts are cut into consecutive launches on the same stream
        for (int32_t s0 = 0; s0 < n_steps; s0 += kMaxFuse) {
            const int32_t n = n_steps - s0 < kMaxFuse ? n_steps - s0 : kMaxFuse;
            hipLaunchKernelGGL(env_async_kernel<MODE>, dim3((unsigned)((N + kAW - 1) / kAW)), dim3(3 * kAW), 0, st,
                               env->p, script ? script + (int64_t)s0 * N : nullptr,
                               obs ? obs + (int64_t)s0 * ABR_OBS_DIM * N : nullptr,
                               rew ? rew + (int64_t)s0 * N : nullptr, dn ? dn + (int64_t)s0 * N : nullptr,
                               acts ? acts + (int64_t)s0 * N : nullptr, n, seed);
        }
        HIP_TRY(hipGetLastError());
        return ABR_OK;
    }
#endif
    if (is_split(impl))
        launch_split<MODE>(impl, env->p, script, obs, rew, dn, acts, n_steps, seed, st);
    else
        hipLaunchKernelGGL(impl ? env_advance_kernel<MODE> : env_jump_kernel<MODE>, dim3(grid64(N)), dim3(64), 0,
                           st, env->p, script, nullptr, nullptr, nullptr, obs, rew, dn, acts, n_steps, seed);
    HIP_TRY(hipGetLastError());
    return ABR_OK;
}

extern "C" int abr_env_step_random(abr_env *env, int32_t n_steps, uint64_t seed,
                                   float *obs_out_dev, float *reward_out_dev,
                                   uint8_t *done_out_dev, int32_t *actions_out_dev, void *stream) {
    if (!env) return fail(ABR_E_INVALID, "env is NULL");
    if (n_steps < 1) return fail(ABR_E_INVALID, "n_steps must be >= 1");
    return launch_fused<2>(env, nullptr, n_steps, seed, obs_out_dev, reward_out_dev, done_out_dev,
                           actions_out_dev, (hipStream_t)stream);
}

// The same fused rollout with the ABR controller's answers given up front: what run() does with a
// scripted abr_controller (get_next_bitrate returns actions_dev[step][lane], Simulator.py:155).
extern "C" int abr_env_step_script(abr_env *env, int32_t n_steps, const int32_t *actions_dev,
                                   float *obs_out_dev, float *reward_out_dev, uint8_t *done_out_dev,
                                   void *stream) {
    if (!env) return fail(ABR_E_INVALID, "env is NULL");
    if (n_steps < 1) return fail(ABR_E_INVALID, "n_steps must be >= 1");
    if (!actions_dev) return fail(ABR_E_INVALID, "actions_dev is NULL");
    return launch_fused<3>(env, actions_dev, n_steps, 0ull, obs_out_dev, reward_out_dev, done_out_dev,
                           nullptr, (hipStream_t)stream);
}

extern "C" int abr_env_get_effective_impl(abr_env *env, int32_t fused, int32_t *impl_out) {
    if (!env || !impl_out) return fail(ABR_E_INVALID, "NULL argument");
    *impl_out = effective_impl(env, fused != 0);
    return ABR_OK;
}

extern "C" int abr_env_episode_qoe(abr_env *env, double *qoe_out_dev, void *stream) {
    if (!env || !qoe_out_dev) return fail(ABR_E_INVALID, "NULL argument");
    hipLaunchKernelGGL(episode_qoe_kernel, dim3((unsigned)((env->p.n_lanes + 255) / 256)),
                       dim3(256), 0, (hipStream_t)stream, env->p, qoe_out_dev);
    HIP_TRY(hipGetLastError());
    return ABR_OK;
}

extern "C" int abr_env_observe_f64(abr_env *env, double *out_dev, void *stream) {
    if (!env || !out_dev) return fail(ABR_E_INVALID, "NULL argument");
    hipLaunchKernelGGL(observe_f64_kernel, dim3((unsigned)((env->p.n_lanes + 255) / 256)),
                       dim3(256), 0, (hipStream_t)stream, env->p, out_dev);
    HIP_TRY(hipGetLastError());
    return ABR_OK;
}

#ifdef ABR_SPLIT_STAMPS
extern "C" int abr_debug_read_stamps(unsigned long long *out32, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_st_acc), 32 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_st_acc), z, sizeof(z)); }
    return 0;
}
#endif

#ifdef ABR_SPLIT_STAMPS
extern "C" int abr_debug_read_stamps_xcd(unsigned long long *out8x32, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out8x32, HIP_SYMBOL(g_st_acc_xcd), 8 * 32 * sizeof(unsigned long long));
    if (reset) { static unsigned long long z[8 * 32]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_st_acc_xcd), z, sizeof(z)); }
    return 0;
}
extern "C" int abr_debug_read_wg_times(unsigned long long *out, int n_wg) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_t), (size_t)n_wg * 10 * sizeof(unsigned long long));
    return 0;
}
#endif

#ifdef ABR_ASYNC_STATS
extern "C" int abr_debug_async_stats(unsigned long long *out48, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_async_stats), 48 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[48] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_async_stats), z, sizeof(z)); }
    return 0;
}
#endif

extern "C" int abr_env_get_state(abr_env *env, abr_env_state_view *v) {
    if (!env || !v) return fail(ABR_E_INVALID, "NULL argument");
    const EnvParams &p = env->p;
    v->n_lanes = p.n_lanes; v->chunk_id = p.chunk_id; v->last_bitrate = p.last_action;
    v->buffer_level = p.buf; v->hist_n = p.hist_n; v->hist_sum_inv = p.hist_s; v->done = p.done;
    v->action_hist = p.action_hist; v->bw_hist = p.bw_hist;
    return ABR_OK;
}

// ===========================================================================
// K3: MPC lookahead (mpc.py)
// ===========================================================================
struct MpcParams {
    int32_t B, H, V, clip;
    double L, max_buffer, wv, wr, ws;
    int64_t n_lanes;
    const int32_t *chunk, *prev;
    const double *buffer;
    double *hist_n, *hist_s;
    const double *br, *sz;
    // f4 (parity unpinned, see include/abr_env.h: abr_mpc_options)
    int32_t predictor;         // 0 harmonic (mpc.py:81-93), 1 exponential smoothing (mpc.py:72-79)
    int32_t utility;           // 0 identity (mpc.py:95-97), 1 log(bitrate / top bitrate) (mpc.py:99-102)
    const double *hist;        // predictor 1: previous_bandwidths, entry t of lane i at hist[t * hist_stride + i]
    int64_t hist_stride;
    const int32_t *hist_len;   // predictor 1: len(previous_bandwidths) per lane
    // phase 1 run ahead of the search kernel by mpc_predict_kernel (caller-provided scratch), or null
    const double *pre_pred;    // [H][n_lanes]
    const int32_t *pre_he, *pre_prev;
    double *pre_pred_w;        // the same arrays, writable, for mpc_predict_kernel
    int32_t *pre_he_w, *pre_prev_w;
    const uint8_t *mask;
    int32_t mask_is_done;      // mask[] holds ABR_DONE_* bits: a lane is active iff its byte is 0
    int32_t neg_to_zero;       // report "no decision" (-1: D13 / D12) as bitrate 0 in action_out
    int32_t *action_out, *flat_out;
    double *J_out;
};

__device__ inline double pymax0(double x) { return (x > 0.0) ? x : 0.0; }   // Python max(0, x)

// per-lane tables in LDS, [level][rate]:
//   brv[i][r] = bitrates[i][r]                      (chunk c+i's ladder, mpc.py:127-128)
//   rbt[i][r] = max(0, sizes[i][r], L) / C_hat[i]    (mpc.py:151-152, D10)
//   tdl[i][r] = sizes_of_CURRENT_chunk[r] / C_hat[i] (mpc.py:107,116 via :155-156, D11)
struct MpcLds {
    const double *brv, *rbt, *tdl;
    double L, max_buffer, wv, wr;
    int B, heff;
};

// Running best of one thread: x = -J (so the arg-min of J is the FIRST arg-max of x).
// idx is the flat index of the winning innermost GROUP (the B leaves that share their first
// H-1 digits) -- the winning leaf inside it is resolved once per lane at the end -- or, when
// a clipped horizon ends above the innermost level, of the winning leaf itself.
struct Best { double x; int32_t idx; double m; };    // m: the running maximum of the level-(H-2) node being enumerated

// Depth-first enumeration with prefix sharing: the partial sums of objective()
// (mpc.py:144-156) after level i depend only on R[0..i], and are formed in the
// same order as the reference forms them, so every leaf value is bit-identical
// to a from-scratch evaluation.  Leaves are visited in increasing flat index;
// strict `<` keeps the first minimum (numpy argmin on the raveled C-order grid).
//
// BC > 0: the number of rates is a compile-time constant: the two innermost
// levels are fully unrolled and the innermost level's tables (bl = bitrates,
// rl = max(0,size,L)/C_hat) live in registers instead of LDS.  BC == 0: generic.
// FULL: the lane's horizon is not clipped (heff == H), so no node below the prefix can be a
// leaf: the per-node `is this the clipped end?` test (and its exec-mask branch) disappears.
// WVM: how variance_weight enters a leaf (mpc.py:158): 0 = multiply; 1 = the weight is exactly
// 1.0, and x * 1.0 is x bit for bit, so the multiplication is dropped; 2 = the weight is
// exactly 0.0 (mpc_test.py's QOEMetric), 0.0 * v is +0.0 for the finite non-negative v that
// occur and q - 0.0 is q, so the whole variance term is dropped.
template <int LVL, int H, int BC, bool FULL, int WVM>
__device__ __forceinline__ void mpc_node(const MpcLds &t, const double *bl, const double *rl,
                                         int r, double q, double v, double rb, double buf,
                                         double br_prev, int pd, int32_t flat, Best &best);

template <int LVL, int H, int BC, bool FULL, int WVM>
__device__ __forceinline__ void mpc_dfs(const MpcLds &t, const double *bl, const double *rl,
                                        double q, double v, double rb, double buf,
                                        double br_prev, int pd, int32_t flat, Best &best) {
    // pd: the digit br_prev belongs to when the caller knows it at compile time (unrolled
    // levels), else -1.  The same digit one level down means |b - br_prev| is |b - b| = +0.0
    // and v + 0.0 is v (v >= +0.0): those two operations are skipped, bit for bit.
    const int B = BC ? BC : t.B;
    if constexpr (LVL == H - 1) {
        // innermost level: only the group maximum is tracked (v_max_f64 per leaf instead of
        // compare + three selects); ties inside the group are resolved by mpc_resolve_group
        double g = -INFINITY;
        if constexpr (BC > 0) {
#pragma unroll
            for (int r = 0; r < BC; r++) {
                const double b = bl[r];
                double a = q + b;                                                    // :146
                const double vv = (r == pd) ? v : v + fabs(b - br_prev);            // :148-149
                if (WVM == 0) a = a - t.wv * vv;                                     // :158
                if (WVM == 1) a = a - vv;
                const double x = a - t.wr * (rb + (rl[r] - buf));                    // :151-152,:159
                g = fmax(g, x);
            }
        } else {
#pragma unroll 1
            for (int r = 0; r < B; r++) {
                const double b = t.brv[LVL * B + r];
                const double x = ((q + b) - t.wv * (v + fabs(b - br_prev))) -
                                 t.wr * (rb + (t.rbt[LVL * B + r] - buf));
                g = fmax(g, x);
            }
        }
        // H >= 4: the B groups below one level-(H-2) node only feed that node's maximum (one v_max_f64
        // per group); the arg-max bookkeeping (compare, three selects, index) is paid once per NODE,
        // which is then named by its H-2 digits, and the first leaf below it that reaches the
        // lane's maximum is found afterwards (mpc_resolve_group, one thread per group)
        if constexpr (H >= 4) best.m = fmax(best.m, g);
        else if (g > best.x) { best.x = g; best.idx = flat; }
    } else {
        if constexpr (LVL == H - 2 && H >= 4) best.m = -INFINITY;
        if constexpr (BC > 0 && LVL >= H - 2) {
#pragma unroll
            for (int r = 0; r < BC; r++) mpc_node<LVL, H, BC, FULL, WVM>(t, bl, rl, r, q, v, rb, buf, br_prev, pd, flat, best);
        } else {
#pragma unroll 1
            for (int r = 0; r < B; r++) mpc_node<LVL, H, BC, FULL, WVM>(t, bl, rl, r, q, v, rb, buf, br_prev, pd, flat, best);
        }
        if constexpr (LVL == H - 2 && H >= 4) {
            if (best.m > best.x) { best.x = best.m; best.idx = flat; }
        }
    }
}

// inner node (LVL < H - 1)
template <int LVL, int H, int BC, bool FULL, int WVM>
__device__ __forceinline__ void mpc_node(const MpcLds &t, const double *bl, const double *rl,
                                         int r, double q, double v, double rb, double buf,
                                         double br_prev, int pd, int32_t flat, Best &best) {
    const int B = BC ? BC : t.B;
    const double b = t.brv[LVL * B + r];
    const double q2 = q + b;                                  // :146
    const double v2 = (r == pd) ? v : v + fabs(b - br_prev);  // :148-149
    const double rb2 = rb + (t.rbt[LVL * B + r] - buf);       // :151-152
    const int32_t f2 = flat * B + r;
    if (!FULL && LVL == t.heff - 1) {
        // a clipped horizon (D12) ends here: this node is a leaf
        const double x = (q2 - t.wv * v2) - t.wr * rb2;       // = -J, :158-162 (startup term is 0)
        if (x > best.x) { best.x = x; best.idx = f2; }
    } else {
        if constexpr (LVL + 1 < H) {
            const double tmp = pymax0(buf - t.tdl[LVL * B + r]);               // :107,:116
            const double wait = pymax0(tmp + t.L - t.max_buffer);              // :108-109
            const double nb = pymax0(tmp + t.L - wait);                        // :117
            // bitrates[LVL+1][R[LVL+1]] of the child's "previous" digit r (both from chunk LVL+1's ladder)
            const double bp = (LVL + 1 == H - 1 && BC > 0) ? bl[r] : t.brv[(LVL + 1) * B + r];
            // the child level sees this node's digit as its "previous" one; only worth telling
            // it when this level is unrolled (r is then a compile-time constant)
            mpc_dfs<LVL + 1, H, BC, FULL, WVM>(t, bl, rl, q2, v2, rb2, nb, bp,
                                               (BC > 0 && LVL >= H - 2) ? r : -1, f2, best);
        }
    }
}

// The first leaf of innermost group `gflat` whose x equals the lane's maximum `xbest`
// (INT32_MAX: none): re-walks the H-1 fixed digits with the same operation sequence as the
// search, then scans the B leaves in order.  Runs once per lane (H < 4) or once per group of
// the winning level-(H-2) node.
__device__ inline int32_t mpc_resolve_group(const MpcLds &t, int H, int32_t gflat, int prev0,
                                            double buf, double xbest) {
    const int B = t.B;
    int32_t pw = 1;
    for (int i = 0; i < H - 2; i++) pw *= B;
    double q = 0.0, v = 0.0, rb = 0.0;
    double bp = t.brv[prev0];
    for (int lvl = 0; lvl < H - 1; lvl++) {
        const int r = (gflat / pw) % B;
        pw = (pw >= B) ? pw / B : 1;
        const double b = t.brv[lvl * B + r];
        q = q + b; v = v + fabs(b - bp); rb = rb + (t.rbt[lvl * B + r] - buf);
        const double tmp = pymax0(buf - t.tdl[lvl * B + r]);
        const double wait = pymax0(tmp + t.L - t.max_buffer);
        buf = pymax0(tmp + t.L - wait);
        bp = t.brv[(lvl + 1) * B + r];
    }
    const int lvl = H - 1;
    for (int r = 0; r < B; r++) {
        const double b = t.brv[lvl * B + r];
        const double x = ((q + b) - t.wv * (v + fabs(b - bp))) - t.wr * (rb + (t.rbt[lvl * B + r] - buf));
        if (x == xbest) return gflat * B + r;
    }
    return 0x7fffffff;    // not in this group
}

// Phase 1 of K3 for one lane: validates chunk / previous_bitrate, runs the throughput predictor
// (growing the caller's history, D9, in the harmonic branch) and writes the H predictions to
// pred[i * stride].  Returns the effective horizon (0 = no decision); prev_out = previous_bitrate
// as the index Python would use.
__device__ inline int mpc_predict_lane(const MpcParams &p, int64_t lane, int B, int H, double *pred,
                                       int64_t stride, int &prev_out) {
    double n = p.hist_n[lane], S = p.hist_s[lane];
    int c = p.chunk[lane];
    int he = H;
    if (c + H > p.V) he = p.clip ? (p.V - c) : 0;  // D12
    if (he < 0) he = 0;
    // R[0] = previous_bitrate indexes bitrates[i] (mpc.py:132,148): Python wraps -B..-1 to
    // the top of the ladder (the env's "no previous chunk" value -1 -> the highest rate) and
    // raises IndexError outside [-B, B)
    int pv = p.prev[lane];
    const bool prev_ok = (pv >= -B) && (pv < B);
    if (pv < 0) pv += B;
    prev_out = prev_ok ? pv : 0;
    if (p.predictor == 1) {
        // method="expsmoothing" (mpc.py:72-79): SimpleExpSmoothing(data).fit(0.5), then the
        // H out-of-sample forecasts -- all equal to the last smoothed level.  statsmodels is
        // not available to pin this against (PARITY UNPINNED); the rule implemented is the
        // documented one: smoothing level 0.5, initial level = the least-squares optimum of
        // the one-step-ahead errors (what fit()'s default `estimated` initialisation
        // approximates numerically), in closed form.  With l(t-1) = a + b*l0 the level before
        // observation y(t):  l0 = sum b (y - a) / sum b^2.  The history is NOT grown (the
        // reference's branch returns before the append of mpc.py:92).
        const int32_t hl = p.hist_len ? p.hist_len[lane] : 0;
        if (hl <= 0 || c < 0 || !prev_ok) return 0;
        double a = 0.0, b = 1.0, num = 0.0, den = 0.0;
        for (int32_t tt = 0; tt < hl; tt++) {
            const double y = p.hist[(int64_t)tt * p.hist_stride + lane];
            num = num + b * (y - a);
            den = den + b * b;
            a = 0.5 * y + 0.5 * a;
            b = 0.5 * b;
        }
        const double level = a + b * (num / den);
        if (!(level > 0.0)) return 0;
        for (int i = 0; i < H; i++) pred[(int64_t)i * stride] = level;
        return he;
    }
    if (!(n > 0.0) || !(S > 0.0) || c < 0 || !prev_ok) {
        // D13: empty / zero throughput history -- the reference raises ZeroDivisionError
        // (mpc.py:88,90); likewise an out-of-range previous_bitrate (IndexError).  Defined
        // here as "no decision": history untouched, action -1.
        return 0;
    }
    for (int i = 0; i < H; i++) {
        double tp = n / S;            // history_size / sum_inverse  (:90)
        pred[(int64_t)i * stride] = tp;
        S = S + 1.0 / tp;             // throughput_values.append(tp): next pass sums it last
        n = n + 1.0;
    }
    p.hist_n[lane] = n; p.hist_s[lane] = S;    // D9: the caller's list has grown by H
    return he;
}

// Phase 1 for every lane ahead of the search kernel (one thread per lane): ten dependent IEEE
// divisions that would otherwise run on 1 of a lane's 36 threads while the other 35 wait.
__global__ void mpc_predict_kernel(MpcParams p) {
    const int64_t lane = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= p.n_lanes) return;
    if (p.mask && ((p.mask[lane] != 0) == (p.mask_is_done != 0))) return;
    int pv;
    const int he = mpc_predict_lane(p, lane, p.B, p.H, p.pre_pred_w + lane, p.n_lanes, pv);
    p.pre_he_w[lane] = he; p.pre_prev_w[lane] = pv;
}

constexpr int kMpcLanesPerBlock = 16;

// Threads of a block: (lane-in-block, prefix) pairs.  D = number of leading
// levels fixed per thread (2 when H >= 3, else 1) -> T = B^D threads per lane.
template <int H, int BC, int WVM>
__global__ __launch_bounds__(256, 4)
void mpc_select_kernel(MpcParams p, int T, int D, int LPB) {
    extern __shared__ double lds[];
    const int B = BC ? BC : p.B;
    const int HB = H * B;
    // layout per lane-in-block: brv[HB] rbt[HB] tdl[HB] pred[H]
    const int per_lane = 3 * HB + H;
    double *tab = lds;
    __shared__ int32_t heff_s[16];
    __shared__ int32_t prev_s[16];    // previous_bitrate as the index Python would use (mpc.py:132,148)
    __shared__ unsigned long long bestK[16];   // phase 4: the lane's maximum of x = -J as an ordered key
    __shared__ int32_t bestF[16];              //          and the smallest flat index that reaches it
    __shared__ int32_t bestL[16];              // phase 5 (H >= 4): the first leaf below the winning node that reaches it
    const int tid = threadIdx.x;
    const int li = tid / T;               // lane in block
    const int pre = tid - li * T;         // prefix id
    if (threadIdx.x < 16) { bestK[threadIdx.x] = 0; bestF[threadIdx.x] = 0x7fffffff; bestL[threadIdx.x] = 0x7fffffff; }
    const int64_t lane = (int64_t)blockIdx.x * LPB + li;
    const bool valid = (li < LPB) && (lane < p.n_lanes) &&
                       !(p.mask && ((p.mask[lane] != 0) == (p.mask_is_done != 0)));
    double *my = tab + (li < LPB ? li : 0) * per_lane;

    K3_STAMP_DECL
    // ---- phase 1: the predictor, one thread per lane (mpc.py:81-93 / :72-79) -- or its result, when
    //      mpc_predict_kernel already ran it for every lane (p.pre_pred != nullptr) ----
    if (valid && pre == 0) {
        if (p.pre_pred) {
            const int he = p.pre_he[lane];
            for (int i = 0; i < H; i++) my[3 * HB + i] = p.pre_pred[(int64_t)i * p.n_lanes + lane];
            prev_s[li] = p.pre_prev[lane];
            heff_s[li] = he;
        } else {
            int pv;
            heff_s[li] = mpc_predict_lane(p, lane, B, H, my + 3 * HB, 1, pv);
            prev_s[li] = pv;
        }
    }
    __syncthreads();
    K3_STAMP(26);
    // ---- phase 2: the per-(level, rate) tables, 60 divisions per lane spread over T threads ----
    if (valid) {
        const int c = p.chunk[lane];
        const int he = heff_s[li];
        for (int e = pre; e < HB; e += T) {
            const int i = e / B, r = e - i * B;
            if (i < he) {
                const double pr = my[3 * HB + i];
                const double s_i = p.sz[(int64_t)(c + i) * B + r];
                double m = 0.0;
                if (s_i > m) m = s_i;
                if (p.L > m) m = p.L;                      // max(0, size, chunk_length)
                const double bre = p.br[(int64_t)(c + i) * B + r];
                // utility 1: log_bitrate_utility with the arity its call sites need (mpc.py:99-102,
                // :146-149): u = log(bitrate / top bitrate of that chunk).  PARITY UNPINNED.
                my[e] = p.utility == 1 ? log(bre / p.br[(int64_t)(c + i) * B + (B - 1)]) : bre;
                my[HB + e] = m / pr;
                my[2 * HB + e] = p.sz[(int64_t)c * B + r] / pr;
            }
        }
    }
    __syncthreads();
    K3_STAMP(27);
    // ---- phase 3: each thread walks its prefix, then enumerates its subtree ----
    Best best; best.x = -INFINITY; best.idx = 0x7fffffff; best.m = -INFINITY;
    if (valid && heff_s[li] > 0) {
        MpcLds t;
        t.brv = my; t.rbt = my + HB; t.tdl = my + 2 * HB;
        t.L = p.L; t.max_buffer = p.max_buffer; t.wv = p.wv; t.wr = p.wr; t.B = B;
        t.heff = heff_s[li];
        double q = 0.0, v = 0.0, rb = 0.0, buf = p.buffer[lane];
        int prev_r = prev_s[li];
        int32_t flat = 0;
        bool ok = true, is_leaf = false;
        // digits of the prefix, most significant first (two scalars, not an array: indexed by the loop variable an array went
        // to scratch -- the 12 B of private memory per lane of rounds 4-5, four scratch instructions per thread, here and not in
        // the enumeration)
        const int dig0 = (D == 2) ? pre / B : pre, dig1 = (D == 2) ? pre - dig0 * B : 0;
        for (int lvl = 0; lvl < D && ok && !is_leaf; lvl++) {
            const int r = lvl == 0 ? dig0 : dig1;
            const double *brv = t.brv + lvl * B;
            const double b = brv[r];
            const double bp = brv[prev_r];
            q = q + b; v = v + fabs(b - bp); rb = rb + (t.rbt[lvl * B + r] - buf);
            flat = flat * B + r;
            if (lvl == t.heff - 1) {
                // clipped horizon ends inside the prefix: only all-zero remainders are real combos
                if (lvl == 0 && D == 2 && dig1 != 0) ok = false;
                is_leaf = true;
            } else {
                const double tmp = pymax0(buf - t.tdl[lvl * B + r]);
                const double wait = pymax0(tmp + t.L - t.max_buffer);
                buf = pymax0(tmp + t.L - wait);
                prev_r = r;
            }
        }
        if (ok) {
            // innermost level's tables in registers (BC > 0)
            double bl[BC ? BC : 1], rl[BC ? BC : 1];
            if constexpr (BC > 0) {
#pragma unroll
                for (int r = 0; r < BC; r++) { bl[r] = t.brv[(H - 1) * B + r]; rl[r] = t.rbt[(H - 1) * B + r]; }
            }
            if (is_leaf) {
                best.x = (q - t.wv * v) - t.wr * rb; best.idx = flat;
            } else if (D == 2) {
                if constexpr (H >= 3) {
                    if (t.heff == H)
                        mpc_dfs<2, H, BC, true, WVM>(t, bl, rl, q, v, rb, buf, t.brv[2 * B + prev_r], -1, flat, best);
                    else    // clipped horizons (the last H-1 chunks of a video): compact generic code
                        mpc_dfs<2, H, 0, false, 0>(t, bl, rl, q, v, rb, buf, t.brv[2 * B + prev_r], -1, flat, best);
                }
            } else {
                if constexpr (H >= 2) {
                    if (t.heff == H)
                        mpc_dfs<1, H, BC, true, WVM>(t, bl, rl, q, v, rb, buf, t.brv[1 * B + prev_r], -1, flat, best);
                    else
                        mpc_dfs<1, H, 0, false, 0>(t, bl, rl, q, v, rb, buf, t.brv[1 * B + prev_r], -1, flat, best);
                }
            }
        }
    }
    K3_STAMP(28);
    // ---- phase 4: first arg-max of x = -J over the T prefixes of a lane (ascending prefix =
    //      ascending flat index); the winning leaf is found in phase 5.  Two LDS atomics
    //      and two barriers: the maximum of x as an order-preserving 64-bit key (ds_max_u64), then
    //      the smallest flat index among the threads that hold it (ds_min_i32) -- the same winner
    //      as a left-to-right scan with strict `>` (round 2 used a pairwise tree: six barriers).
    //      x + 0.0 folds -0.0 into +0.0 so that equal values have equal keys; a NaN never wins. ----
    if (!valid && pre == 0 && li < LPB && lane < p.n_lanes && p.mask_is_done)
        p.action_out[lane] = -1;          // a finished lane of the fused rollout takes no decision
    unsigned long long key = 0;
    {
        const double xc = best.x + 0.0;
        const unsigned long long u = (unsigned long long)__double_as_longlong(xc);
        key = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
        if (!(xc == xc) || best.idx == 0x7fffffff) key = 0;
    }
    if (li < LPB && key) atomicMax(&bestK[li], key);
    __syncthreads();
    if (li < LPB && key && key == bestK[li]) atomicMin(&bestF[li], best.idx);
    __syncthreads();
    K3_STAMP(29);
    // ---- phase 5: the lanes' results, by the first threads of the block (with `pre == 0` every wave
    //      held one such thread and walked through the resolve; now the other waves skip it).
    //      H >= 4: bestF names the winning level-(H-2) node; thread (lane i, group r) looks through
    //      one of its B innermost groups for the first leaf that reaches the maximum. ----
    if (H >= 4) {
        if (tid < LPB * B) {
            const int l2 = tid / B, r = tid - l2 * B;
            const int64_t lane2 = (int64_t)blockIdx.x * LPB + l2;
            const unsigned long long bk = bestK[l2];
            const int32_t bf = bestF[l2];
            // (a masked or out-of-range lane has bk == 0: none of its threads entered a key)
            if (bk != 0 && bf != 0x7fffffff && heff_s[l2] == H) {
                const unsigned long long bu = (bk >> 63) ? (bk & 0x7fffffffffffffffull) : ~bk;
                const double *my2 = tab + l2 * per_lane;
                MpcLds t;
                t.brv = my2; t.rbt = my2 + HB; t.tdl = my2 + 2 * HB;
                t.L = p.L; t.max_buffer = p.max_buffer; t.wv = p.wv; t.wr = p.wr; t.B = B; t.heff = H;
                const int32_t first = mpc_resolve_group(t, H, bf * B + r, prev_s[l2], p.buffer[lane2],
                                                        __longlong_as_double((long long)bu));
                if (first != 0x7fffffff) atomicMin(&bestL[l2], first);
            }
        }
        __syncthreads();
    }
    if (tid < LPB) {
        const int l2 = tid;
        const int64_t lane2 = (int64_t)blockIdx.x * LPB + l2;
        const bool valid2 = (lane2 < p.n_lanes) && !(p.mask && ((p.mask[lane2] != 0) == (p.mask_is_done != 0)));
        if (valid2) {
            const unsigned long long bk = bestK[l2];
            const unsigned long long bu = (bk >> 63) ? (bk & 0x7fffffffffffffffull) : ~bk;
            const double bx = __longlong_as_double((long long)bu);
            int32_t bf = bestF[l2];
            const bool have = bk != 0 && bf != 0x7fffffff;
            const int he = heff_s[l2];
            int32_t act = -1;
            if (have) {
                if (he == H) {
                    if (H >= 4) {
                        // always found: the maximum came out of this very arithmetic (else: the node's first leaf)
                        const int32_t fl = bestL[l2];
                        bf = (fl != 0x7fffffff) ? fl : bf * B * B;
                    } else {
                        const double *my2 = tab + l2 * per_lane;
                        MpcLds t;
                        t.brv = my2; t.rbt = my2 + HB; t.tdl = my2 + 2 * HB;
                        t.L = p.L; t.max_buffer = p.max_buffer; t.wv = p.wv; t.wr = p.wr; t.B = B; t.heff = he;
                        const int32_t fl = mpc_resolve_group(t, H, bf, prev_s[l2], p.buffer[lane2], bx);
                        bf = (fl != 0x7fffffff) ? fl : bf * B;
                    }
                }
                int32_t lead = 1;
                for (int i = 1; i < he; i++) lead *= B;
                act = bf / lead;                                   // int(result[0])  mpc.py:186
            }
            p.action_out[lane2] = (p.neg_to_zero && act < 0) ? 0 : act;
            if (p.flat_out) p.flat_out[lane2] = have ? bf : -1;
            if (p.J_out) p.J_out[lane2] = have ? -bx : NAN;
        }
    }
    K3_STAMP(30);
    K3_STAMP_FLUSH();
}

static int validate_mpc(const abr_mpc_config *c) {
    if (!c) return fail(ABR_E_INVALID, "mpc config is NULL");
    if (c->n_rates < 1 || c->n_rates > ABR_MAX_RATES)
        return fail(ABR_E_INVALID, "n_rates %d outside 1..%d", c->n_rates, ABR_MAX_RATES);
    if (c->horizon < 2 || c->horizon > ABR_MAX_HORIZON)
        return fail(ABR_E_INVALID, "horizon %d outside 2..%d", c->horizon, ABR_MAX_HORIZON);
    double combos = pow((double)c->n_rates, (double)c->horizon);
    if (combos > 2.0e9) return fail(ABR_E_UNSUPPORTED, "n_rates^horizon = %g exceeds int32", combos);
    if (c->video_length < 1) return fail(ABR_E_INVALID, "video_length must be >= 1");
    return ABR_OK;
}

template <int H, int BC, int WVM>
static void launch_mpc_b(const MpcParams &p, int T, int D, hipStream_t st) {
    // lanes per workgroup: as many as fit 256 threads (4 waves).  Measured on MI355X at B=6,H=5
    // (T=36): 7 lanes/WG 319 us, 16 lanes/WG (9 waves) 481 us, 3 lanes/WG 365 us per 65 536 lanes.
    int lpb = 256 / T;
    if (lpb > kMpcLanesPerBlock) lpb = kMpcLanesPerBlock;
    if (lpb < 1) lpb = 1;
    const int threads = ((lpb * T + 63) / 64) * 64;
    const size_t per_lane = (size_t)(3 * H * p.B + H) * sizeof(double);
    const size_t lds = lpb * per_lane;
    // (a persistent grid striding over the lane groups was measured and lost: -8 % at 1 024 workgroups,
    // -2 % at 2 048 -- the hardware's dynamic dispatch fills the tail better; profiles/r03_ab_mpc.txt)
    const unsigned grid = (unsigned)((p.n_lanes + lpb - 1) / lpb);
    hipLaunchKernelGGL((mpc_select_kernel<H, BC, WVM>), dim3(grid), dim3(threads), lds, st, p, T, D, lpb);
}

// compile-time rate count for the common ladders (6 and 4 rates with exact-weight variants; 3, 5, 7 and 8 rates) up to horizon 6
template <int H>
static void launch_mpc(const MpcParams &p, int T, int D, hipStream_t st) {
    if constexpr (H <= 6) {
        // exact-weight specialisations (see WVM above) for the six-rate ladder
        if (p.B == 6 && p.wv == 1.0) { launch_mpc_b<H, 6, 1>(p, T, D, st); return; }
        if (p.B == 6 && p.wv == 0.0) { launch_mpc_b<H, 6, 2>(p, T, D, st); return; }
        if (p.B == 6) { launch_mpc_b<H, 6, 0>(p, T, D, st); return; }
        if (p.B == 4 && p.wv == 0.0) { launch_mpc_b<H, 4, 2>(p, T, D, st); return; }
        if (p.B == 4) { launch_mpc_b<H, 4, 0>(p, T, D, st); return; }
#ifndef ABR_MPC_FEW_LADDERS
        // round 5: the other ladder sizes from 3 to 8 rates get the compile-time rate count too (the generic path
        // runs 30 - 40 % below the specialised one per combination: profiles/r05_mpc_other_shapes.txt)
        if (p.B == 5) { launch_mpc_b<H, 5, 0>(p, T, D, st); return; }
        if (p.B == 3) { launch_mpc_b<H, 3, 0>(p, T, D, st); return; }
        if (p.B == 8) { launch_mpc_b<H, 8, 0>(p, T, D, st); return; }
        if (p.B == 7) { launch_mpc_b<H, 7, 0>(p, T, D, st); return; }
#endif
    }
    launch_mpc_b<H, 0, 0>(p, T, D, st);
}

static void fill_mpc_params(MpcParams &p, const abr_mpc_config *cfg, int64_t n_lanes) {
    p.B = cfg->n_rates; p.H = cfg->horizon; p.V = cfg->video_length; p.clip = cfg->clip_horizon;
    p.L = cfg->chunk_length; p.max_buffer = cfg->max_buffer; p.wv = cfg->variance_weight;
    p.wr = cfg->rebuffer_weight; p.ws = cfg->startup_weight; p.n_lanes = n_lanes;
    p.mask = nullptr; p.mask_is_done = 0; p.neg_to_zero = 0;
    p.predictor = 0; p.utility = 0; p.hist = nullptr; p.hist_stride = 0; p.hist_len = nullptr;
    p.pre_pred = nullptr; p.pre_he = nullptr; p.pre_prev = nullptr;
    p.pre_pred_w = nullptr; p.pre_he_w = nullptr; p.pre_prev_w = nullptr;
    p.flat_out = nullptr; p.J_out = nullptr;
}

static size_t mpc_scratch_bytes(int H, int64_t n_lanes) {
    return (size_t)n_lanes * ((size_t)H * sizeof(double) + 2 * sizeof(int32_t));
}

// scratch != nullptr: run phase 1 as its own kernel first
static int launch_mpc_select(MpcParams p, hipStream_t st, void *scratch = nullptr) {
    if (scratch) {
        p.pre_pred_w = (double *)scratch;
        p.pre_he_w = (int32_t *)(p.pre_pred_w + (size_t)p.H * p.n_lanes);
        p.pre_prev_w = p.pre_he_w + p.n_lanes;
        p.pre_pred = nullptr;                      // the predictor kernel itself must not read "pre"
        hipLaunchKernelGGL(mpc_predict_kernel, dim3((unsigned)((p.n_lanes + 255) / 256)), dim3(256), 0, st, p);
        p.pre_pred = p.pre_pred_w; p.pre_he = p.pre_he_w; p.pre_prev = p.pre_prev_w;
    }
    const int D = (p.H >= 3) ? 2 : 1;
    int T = p.B; if (D == 2) T *= p.B;
    switch (p.H) {
        case 2: launch_mpc<2>(p, T, D, st); break;
        case 3: launch_mpc<3>(p, T, D, st); break;
        case 4: launch_mpc<4>(p, T, D, st); break;
        case 5: launch_mpc<5>(p, T, D, st); break;
        case 6: launch_mpc<6>(p, T, D, st); break;
        case 7: launch_mpc<7>(p, T, D, st); break;
        case 8: launch_mpc<8>(p, T, D, st); break;
        default: return fail(ABR_E_INVALID, "horizon %d", p.H);
    }
    HIP_TRY(hipGetLastError());
    return ABR_OK;
}

extern "C" int abr_mpc_select(const abr_mpc_config *cfg, const int32_t *chunk_dev,
                              const int32_t *prev_bitrate_dev, const double *buffer_dev,
                              double *hist_n_dev, double *hist_sum_inv_dev,
                              const double *br_table_dev, const double *sz_table_dev,
                              const uint8_t *lane_mask_dev, int32_t *action_out_dev,
                              int32_t *best_flat_out_dev, double *best_J_out_dev, int64_t n_lanes,
                              void *stream) {
    int rc = validate_mpc(cfg);
    if (rc) return rc;
    if (!chunk_dev || !prev_bitrate_dev || !buffer_dev || !hist_n_dev || !hist_sum_inv_dev ||
        !br_table_dev || !sz_table_dev || !action_out_dev)
        return fail(ABR_E_INVALID, "NULL device pointer");
    if (n_lanes < 1) return fail(ABR_E_INVALID, "n_lanes must be >= 1");
    MpcParams p;
    fill_mpc_params(p, cfg, n_lanes);
    p.chunk = chunk_dev; p.prev = prev_bitrate_dev; p.buffer = buffer_dev;
    p.hist_n = hist_n_dev; p.hist_s = hist_sum_inv_dev; p.br = br_table_dev; p.sz = sz_table_dev;
    p.mask = lane_mask_dev; p.action_out = action_out_dev; p.flat_out = best_flat_out_dev;
    p.J_out = best_J_out_dev;
    return launch_mpc_select(p, (hipStream_t)stream);
}

static int apply_mpc_options(MpcParams &p, const abr_mpc_options *opt) {
    if (!opt) return ABR_OK;
    if (opt->predictor != ABR_PREDICT_HARMONIC && opt->predictor != ABR_PREDICT_EXPSMOOTHING)
        return fail(ABR_E_INVALID, "predictor must be ABR_PREDICT_HARMONIC or ABR_PREDICT_EXPSMOOTHING");
    if (opt->utility != ABR_UTILITY_IDENTITY && opt->utility != ABR_UTILITY_LOG)
        return fail(ABR_E_INVALID, "utility must be ABR_UTILITY_IDENTITY or ABR_UTILITY_LOG");
    if (opt->predictor == ABR_PREDICT_EXPSMOOTHING && (!opt->hist_dev || !opt->hist_len_dev || opt->hist_stride < 1))
        return fail(ABR_E_INVALID, "exponential smoothing needs the history itself: hist_dev, hist_stride, hist_len_dev");
    p.predictor = opt->predictor; p.utility = opt->utility;
    p.mask_is_done = opt->mask_is_done != 0;
    p.hist = opt->hist_dev; p.hist_stride = opt->hist_stride; p.hist_len = opt->hist_len_dev;
    if (opt->scratch_dev && opt->scratch_bytes < mpc_scratch_bytes(p.H, p.n_lanes))
        return fail(ABR_E_WORKSPACE, "MPC scratch has %zu bytes, need %zu", (size_t)opt->scratch_bytes,
                    mpc_scratch_bytes(p.H, p.n_lanes));
    if (opt->scratch_dev && ((uintptr_t)opt->scratch_dev & 7))
        return fail(ABR_E_WORKSPACE, "MPC scratch must be 8-byte aligned");
    return ABR_OK;
}

extern "C" int abr_mpc_scratch_bytes(const abr_mpc_config *cfg, int64_t n_lanes, size_t *bytes_out) {
    int rc = validate_mpc(cfg);
    if (rc) return rc;
    if (n_lanes < 1 || !bytes_out) return fail(ABR_E_INVALID, "n_lanes must be >= 1 and bytes_out non-NULL");
    *bytes_out = mpc_scratch_bytes(cfg->horizon, n_lanes);
    return ABR_OK;
}

extern "C" int abr_mpc_select_opt(const abr_mpc_config *cfg, const abr_mpc_options *opt,
                                  const int32_t *chunk_dev, const int32_t *prev_bitrate_dev,
                                  const double *buffer_dev, double *hist_n_dev,
                                  double *hist_sum_inv_dev, const double *br_table_dev,
                                  const double *sz_table_dev, const uint8_t *lane_mask_dev,
                                  int32_t *action_out_dev, int32_t *best_flat_out_dev,
                                  double *best_J_out_dev, int64_t n_lanes, void *stream) {
    int rc = validate_mpc(cfg);
    if (rc) return rc;
    if (!chunk_dev || !prev_bitrate_dev || !buffer_dev || !hist_n_dev || !hist_sum_inv_dev ||
        !br_table_dev || !sz_table_dev || !action_out_dev)
        return fail(ABR_E_INVALID, "NULL device pointer");
    if (n_lanes < 1) return fail(ABR_E_INVALID, "n_lanes must be >= 1");
    MpcParams p;
    fill_mpc_params(p, cfg, n_lanes);
    rc = apply_mpc_options(p, opt);
    if (rc) return rc;
    p.chunk = chunk_dev; p.prev = prev_bitrate_dev; p.buffer = buffer_dev;
    p.hist_n = hist_n_dev; p.hist_s = hist_sum_inv_dev; p.br = br_table_dev; p.sz = sz_table_dev;
    p.mask = lane_mask_dev; p.action_out = action_out_dev; p.flat_out = best_flat_out_dev;
    p.J_out = best_J_out_dev;
    return launch_mpc_select(p, (hipStream_t)stream, opt ? opt->scratch_dev : nullptr);
}

// The composition the reference leaves unwired (D5/D6): for n_steps decisions,
// action = MPCBitrateController.next_bitrate() on the lane's own state (mpc.py:181-186 reading
// chunk_number / previous_bitrate / buffer_level / previous_bandwidths straight from the
// environment's workspace, history mutation D9 included), then Simulator.run()'s download of
// that chunk (Simulator.py:155-170).  Two launches per decision, enqueued back to back on
// `stream`: K3 writes the actions, K1 consumes them; finished lanes are masked by their
// done bits and "no decision" (empty history at chunk 0, D13) downloads bitrate 0.
extern "C" int abr_env_step_mpc(abr_env *env, const abr_mpc_config *cfg,
                                const double *br_table_dev, const double *sz_table_dev,
                                int32_t n_steps, float *obs_out_dev, float *reward_out_dev,
                                uint8_t *done_out_dev, int32_t *actions_out_dev, void *stream) {
    if (!env) return fail(ABR_E_INVALID, "env is NULL");
    int rc = validate_mpc(cfg);
    if (rc) return rc;
    if (!br_table_dev || !sz_table_dev) return fail(ABR_E_INVALID, "NULL device pointer");
    if (n_steps < 1) return fail(ABR_E_INVALID, "n_steps must be >= 1");
    if (cfg->n_rates != env->p.n_rates || cfg->video_length != env->p.video_length)
        return fail(ABR_E_INVALID, "MPC tables are [%d][%d], the environment has video_length %d, n_rates %d",
                    cfg->video_length, cfg->n_rates, env->p.video_length, env->p.n_rates);
    if (env->impl == 1) return fail(ABR_E_UNSUPPORTED, "the fused MPC rollout needs the event-driven kernels");
    const EnvParams &e = env->p;
    const int64_t N = e.n_lanes;
    hipStream_t st = (hipStream_t)stream;
    MpcParams p;
    fill_mpc_params(p, cfg, N);
    p.chunk = e.chunk_id; p.prev = e.last_action; p.buffer = e.buf;
    p.hist_n = e.hist_n; p.hist_s = e.hist_s; p.br = br_table_dev; p.sz = sz_table_dev;
    p.mask = e.done; p.mask_is_done = 1; p.neg_to_zero = 1;
    for (int32_t s = 0; s < n_steps; s++) {
        int32_t *act = actions_out_dev ? actions_out_dev + (int64_t)s * N : env->mpc_action;
        p.action_out = act;
        rc = launch_mpc_select(p, st, env->mpc_scratch);
        if (rc) return rc;
        float *obs = obs_out_dev ? obs_out_dev + (int64_t)s * ABR_OBS_DIM * N : nullptr;
        float *rew = reward_out_dev ? reward_out_dev + (int64_t)s * N : nullptr;
        uint8_t *dn = done_out_dev ? done_out_dev + (int64_t)s * N : nullptr;
        if (is_split(effective_impl(env)))
            launch_split<1>(effective_impl(env), env->p, act, obs, rew, dn, nullptr, 1, 0ull, st);
        else
            hipLaunchKernelGGL(env_jump_kernel<1>, dim3(grid64(N)), dim3(64), 0, st, env->p, act, nullptr,
                               nullptr, nullptr, obs, rew, dn, nullptr, 1, 0ull);
        HIP_TRY(hipGetLastError());
    }
    return ABR_OK;
}

// Diagnostic: the exact chain (abr_exact_jump.h) on arbitrary inputs, one case per thread, so
// tests can compare the DEVICE build of the jump arithmetic -- v_rcp_f64 estimate, saturating
// convert, and (with a biased estimate) the out-of-line exact search -- with the naive loop.
template <int STOP, int BIAS>
__global__ void chain_debug_kernel(const double *__restrict__ x0, const double *__restrict__ c,
                                   const double *__restrict__ thr, const int32_t *__restrict__ n,
                                   int64_t count, double *__restrict__ x_out,
                                   int32_t *__restrict__ a_out, uint8_t *__restrict__ hit_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double x = x0[i];
    int32_t a = 0;
    const bool hit = abrx::chain<STOP, BIAS>(x, c[i], thr[i], n[i], a);
    x_out[i] = x; a_out[i] = a; hit_out[i] = hit ? 1 : 0;
}

template <int STOP>
static int launch_chain_debug(int32_t bias, const double *x0, const double *c, const double *thr,
                              const int32_t *n, int64_t count, double *x_out, int32_t *a_out,
                              uint8_t *hit_out, hipStream_t st) {
    const dim3 g((unsigned)((count + 63) / 64)), b(64);
    if (bias == 0) hipLaunchKernelGGL((chain_debug_kernel<STOP, 0>), g, b, 0, st, x0, c, thr, n, count, x_out, a_out, hit_out);
    else if (bias == 4) hipLaunchKernelGGL((chain_debug_kernel<STOP, 4>), g, b, 0, st, x0, c, thr, n, count, x_out, a_out, hit_out);
    else if (bias == -4) hipLaunchKernelGGL((chain_debug_kernel<STOP, -4>), g, b, 0, st, x0, c, thr, n, count, x_out, a_out, hit_out);
    else return fail(ABR_E_INVALID, "estimate_bias must be 0, 4 or -4");
    HIP_TRY(hipGetLastError());
    return ABR_OK;
}

extern "C" int abr_debug_chain(int32_t stop_kind, int32_t estimate_bias, const double *x0_dev,
                               const double *c_dev, const double *thr_dev, const int32_t *n_dev,
                               int64_t count, double *x_out_dev, int32_t *a_out_dev,
                               uint8_t *hit_out_dev, void *stream) {
    if (!x0_dev || !c_dev || !thr_dev || !n_dev || !x_out_dev || !a_out_dev || !hit_out_dev)
        return fail(ABR_E_INVALID, "NULL device pointer");
    if (count < 1) return fail(ABR_E_INVALID, "count must be >= 1");
    hipStream_t st = (hipStream_t)stream;
    switch (stop_kind) {
        case abrx::STOP_GE: return launch_chain_debug<abrx::STOP_GE>(estimate_bias, x0_dev, c_dev, thr_dev, n_dev, count, x_out_dev, a_out_dev, hit_out_dev, st);
        case abrx::STOP_LE: return launch_chain_debug<abrx::STOP_LE>(estimate_bias, x0_dev, c_dev, thr_dev, n_dev, count, x_out_dev, a_out_dev, hit_out_dev, st);
        case abrx::STOP_LT: return launch_chain_debug<abrx::STOP_LT>(estimate_bias, x0_dev, c_dev, thr_dev, n_dev, count, x_out_dev, a_out_dev, hit_out_dev, st);
        default: return fail(ABR_E_INVALID, "stop_kind must be 0 (>=), 1 (<=) or 2 (<)");
    }
}

// Diagnostic: the contract the role-split kernels' fresh_params() rests on -- `EnvParams p` is the FIRST by-value kernel
// argument, so the block at offset 0 of the kernarg segment is this launch's parameter block -- checked in the PRODUCT build:
// instance <9> of each of the two kernel templates (the product instances' signature, a body that only answers) is launched
// once (one workgroup) with a sentinel in its parameter block and says whether fresh_params() saw it.  result_dev: uint32 [2]
// device memory, entry 0 = env_split3_kernel, 1 = env_split_kernel: 1 (seen) or 2 (not).  Touches no lane state.
extern "C" int abr_debug_selfcheck(abr_env *env, uint32_t *result_dev, void *stream) {
    if (!env || !result_dev) return fail(ABR_E_INVALID, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(result_dev, 0, 2 * sizeof(uint32_t), st));
    EnvParams p = env->p;
    p.sentinel = 0x5eed0000c0ffee00ull ^ (uint64_t)(uintptr_t)env;
    p.selfcheck_out = result_dev;
    hipLaunchKernelGGL(env_split3_kernel<9>, dim3(1), dim3(192), 0, st, p, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0ull);
    p.selfcheck_out = result_dev + 1;
    hipLaunchKernelGGL(env_split_kernel<9>, dim3(1), dim3(128), 0, st, p, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0ull);
    HIP_TRY(hipGetLastError());
    return ABR_OK;
}

// Diagnostic: the drain of ONE subtrahend -- buffer_level -= speed*dt until <= 0 (Simulator.py:184, :194) -- as the kernels
// run it (abr_lane_jump.h: lanej_drain: the per-binade cascade, or the general chain for a wave that holds a value above
// it), one case per thread.
__global__ void drain_debug_kernel(abrx::DrainTab tab, double sd, const double *__restrict__ x0,
                                   const int32_t *__restrict__ n, int64_t count, double *__restrict__ x_out,
                                   int32_t *__restrict__ a_out, uint8_t *__restrict__ hit_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    abrx::Tables t;
    t.G = nullptr; t.interval_tick = nullptr; t.avail_tick = nullptr;
    t.L = 0.0; t.sd = sd; t.max_buffer = 0.0; t.start_up_length = 0.0; t.V = 0; t.max_ticks = 0;
    t.per_lane_speed = false; t.speed_rows = 0; t.speed_stride = 0; t.speeds = nullptr;
    t.drain = tab;
    double x = x0[i];
    int32_t a = 0;
    const bool zero = abrx::lanej_drain(t, x, sd, n[i], a);
    x_out[i] = x; a_out[i] = a; hit_out[i] = zero ? 1 : 0;
}

extern "C" int abr_debug_drain(double sd, double max_level, const double *x0_dev, const int32_t *n_dev, int64_t count,
                               double *x_out_dev, int32_t *a_out_dev, uint8_t *hit_out_dev, int32_t *stages_out,
                               void *stream) {
    if (!x0_dev || !n_dev || !x_out_dev || !a_out_dev || !hit_out_dev) return fail(ABR_E_INVALID, "NULL device pointer");
    if (count < 1) return fail(ABR_E_INVALID, "count must be >= 1");
    const abrx::DrainTab tab = abrx::make_drain_tab(sd, max_level);
    if (stages_out) *stages_out = tab.n;
    if (tab.n == 0) return fail(ABR_E_UNSUPPORTED, "no cascade for sd %g below %g: the kernels use the general chains", sd, max_level);
    hipLaunchKernelGGL(drain_debug_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, (hipStream_t)stream, tab, sd,
                       x0_dev, n_dev, count, x_out_dev, a_out_dev, hit_out_dev);
    HIP_TRY(hipGetLastError());
    return ABR_OK;
}

// Diagnostic: every combo evaluated from scratch by its own thread, literally as
// objective() does (mpc.py:120-162) -- no tables, no prefix sharing.  Used by the
// tests as an independent check of the DFS kernel's arithmetic.
__global__ void mpc_grid_kernel(int B, int H, int V, double L, double max_buffer, double wv,
                                double wr, int chunk, int prev, double buffer,
                                const double *__restrict__ pred, const double *__restrict__ br,
                                const double *__restrict__ sz, double *__restrict__ J_out,
                                int64_t total) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= total) return;
    int R[ABR_MAX_HORIZON + 1];
    {
        int64_t t = f;
        for (int i = H; i >= 1; i--) { R[i] = (int)(t % B); t /= B; }
        R[0] = prev < 0 ? prev + B : prev;     // Python's negative index (mpc.py:148)
    }
    double q = 0.0, v = 0.0, rb = 0.0, buf = buffer;
    for (int i = 0; i < H; i++) {
        const double *bri = br + (int64_t)(chunk + i) * B;
        const double *szi = sz + (int64_t)(chunk + i) * B;
        q += bri[R[i + 1]];
        v += fabs(bri[R[i + 1]] - bri[R[i]]);
        double m = 0.0;
        if (szi[R[i + 1]] > m) m = szi[R[i + 1]];
        if (L > m) m = L;
        rb += (m / pred[i] - buf);
        if (i != H - 1) {
            const double cs = sz[(int64_t)chunk * B + R[i + 1]];
            const double nb0 = pymax0(buf - cs / pred[i]);
            const double wait = pymax0(nb0 + L - max_buffer);
            buf = pymax0(nb0 + L - wait);
        }
    }
    J_out[f] = -((q - wv * v) - wr * rb);
    (void)V;
}

extern "C" int abr_mpc_objective_grid(const abr_mpc_config *cfg, int32_t chunk,
                                      int32_t prev_bitrate, double buffer_level,
                                      const double *pred_dev, const double *br_table_dev,
                                      const double *sz_table_dev, double *J_out_dev, void *stream) {
    int rc = validate_mpc(cfg);
    if (rc) return rc;
    if (!pred_dev || !br_table_dev || !sz_table_dev || !J_out_dev)
        return fail(ABR_E_INVALID, "NULL device pointer");
    if (prev_bitrate < -cfg->n_rates || prev_bitrate >= cfg->n_rates)
        return fail(ABR_E_INVALID, "previous_bitrate %d outside [-%d, %d) (mpc.py:148 raises IndexError)",
                    prev_bitrate, cfg->n_rates, cfg->n_rates);
    if (chunk < 0 || chunk + cfg->horizon > cfg->video_length)
        return fail(ABR_E_INVALID, "chunk %d + horizon %d exceeds video_length %d (mpc.py:126 raises)",
                    chunk, cfg->horizon, cfg->video_length);
    int64_t total = 1;
    for (int i = 0; i < cfg->horizon; i++) total *= cfg->n_rates;
    hipLaunchKernelGGL(mpc_grid_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, cfg->n_rates, cfg->horizon, cfg->video_length,
                       cfg->chunk_length, cfg->max_buffer, cfg->variance_weight,
                       cfg->rebuffer_weight, chunk, prev_bitrate, buffer_level, pred_dev,
                       br_table_dev, sz_table_dev, J_out_dev, total);
    HIP_TRY(hipGetLastError());
    return ABR_OK;
}
