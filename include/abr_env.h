/*
 * abr_env.h -- C ABI of the MI355X-native batched ABR environment and MPC lookahead.
 *
 * This is the drop-in boundary for the one hot path BASELINE.json names: the
 * tick loop of the reference's Simulator.run() (Simulator.py:135-208) and the
 * exhaustive lookahead of its MPCBitrateController (mpc.py:120-186), vectorised
 * over independent (trace, start-offset) lanes.
 *
 * The reference is pure Python with no FFI; its plugin boundary is inversion of
 * control: run() calls abr_controller.get_next_bitrate(chunk_id,
 * previous_bitrates, previous_bandwidths, buffer_level) once per chunk
 * (Simulator.py:155).  This ABI turns that inside out: abr_env_reset() runs each
 * lane up to its first get_next_bitrate() call site and hands back exactly the
 * call's arguments as the observation; abr_env_step(actions) supplies the return
 * value and runs to the next call site (or to simulation_end, Simulator.py:207).
 *
 * Conventions
 *  - Every function returns 0 on success or a negative ABR_E_* code; nothing
 *    throws across the boundary.  abr_last_error() gives the message of the
 *    calling thread's last failure.
 *  - Every *_dev pointer is device memory owned by the CALLER (the data_ptr()
 *    of a PyTorch-ROCm tensor).  The library allocates no device memory: its
 *    per-lane state and lookup tables live in the caller-provided workspace.
 *  - Work is only enqueued on the `stream` argument (a hipStream_t passed as
 *    void*; NULL = the default stream).  No call synchronises, except
 *    abr_env_create (one-time table upload).
 *  - Re-entrant per handle, no globals.  Lane arrays are struct-of-arrays with
 *    row stride n_lanes: field f of lane i is at base[f * n_lanes + i].
 */
#ifndef ABR_ENV_H
#define ABR_ENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4: the workspace carries a LAYOUT TAG in its last 256 bytes (magic, ABI version, lane count, configuration, size), written by
 * abr_env_create; abr_env_notify_restore reads it back (it synchronises the device, once per restore) and answers
 * ABR_E_WORKSPACE when the restored bytes were laid out by another ABI version or for another lane count / configuration --
 * before this, such a workspace was reinterpreted with shifted offsets and no error.  abr_env_workspace_bytes grew by 256.
 * CHECKPOINTS DO NOT CARRY OVER: a workspace saved under version 2 or 3 cannot be restored (version 3 itself changed the lane
 * state -- ep_actions gone, a running variance sum added, 8 float64 state arrays -- without saying so; it has no tag and is
 * refused).  New diagnostic entry points: abr_debug_drain, abr_debug_selfcheck.  No existing signature changed.
 * 3: abr_env_has_impl (which implementations THIS build of the library carries: the product answers 0 for the rejected
 * pipelines 4, 6 and 7, which abr_env_set_impl refuses with ABR_E_UNSUPPORTED); under impl 3 (auto) a launch of ONE decision
 * resolves to 0 at every size (abr_env_get_effective_impl(fused = 0)); ABR_DONE_INTERNAL is reserved for the diagnostic
 * pipelines' watchdogs.  A version-2 host keeps working: nothing it could call changed its signature.
 * 2: abr_env_step_script, abr_env_get_effective_impl, abr_env_notify_restore, impl 5 (three waves per 64
 * lanes); struct abr_mpc_options GREW (mask_is_done + reserved_ appended: a version-1 host passes a
 * shorter struct and must be rebuilt); since 1 also: workspace layout grew, abr_env_set_lane_speeds /
 * _speed_schedule / _bitrate_table are latched until the next full reset, every re-reset advances the
 * policy's episode counter, default impl is 3 (auto).  A host built against version 1 must be rebuilt. */
#define ABR_ABI_VERSION 4
#define ABR_MAX_RATES 16
#define ABR_MAX_HORIZON 8

/* error codes */
#define ABR_OK 0
#define ABR_E_INVALID (-1)      /* bad argument / config */
#define ABR_E_WORKSPACE (-2)    /* workspace too small or misaligned */
#define ABR_E_HIP (-3)          /* a HIP runtime call failed */
#define ABR_E_UNSUPPORTED (-4)  /* valid in the reference, not built here (yet) */

/* per-lane bits of done_out */
#define ABR_DONE_EPISODE 0x1    /* chunk_id >= video_length  (Simulator.py:207-208) */
#define ABR_DONE_TIMEOUT 0x2    /* hit config.max_ticks before finishing */
#define ABR_DONE_BADACT 0x4     /* action outside [0, n_rates): lane frozen (the reference raises IndexError) */
#define ABR_DONE_BADARG 0x8     /* abr_env_reset got a trace id outside [0, n_traces) or a negative start
                                   offset for this lane: lane frozen, nothing read out of bounds */

#define ABR_DONE_INTERNAL 0x10  /* RESERVED: never set by the product library (the diagnostic asynchronous
                                   pipeline uses it for its watchdog) */

/* float32 observation rows written by reset/step: the four arguments of
 * get_next_bitrate (Simulator.py:155) first, then run() locals at that instant */
enum {
    ABR_OBS_CHUNK_ID = 0,       /* chunk_id                                   */
    ABR_OBS_LAST_BITRATE = 1,   /* previous_bitrates[-1] (index) or -1        */
    ABR_OBS_LAST_BANDWIDTH = 2, /* previous_bandwidths[-1] or 0 (:164)        */
    ABR_OBS_BUFFER_LEVEL = 3,   /* buffer_level [s]                           */
    ABR_OBS_GLOBAL_TIME = 4,    /* global_time [s]                            */
    ABR_OBS_PLAY_TIME = 5,      /* play_time [s]                              */
    ABR_OBS_REBUFFER_TIME = 6,  /* rebuffer_time, cumulative [s]              */
    ABR_OBS_STARTUP_TIME = 7,   /* start_up_time, cumulative [s]              */
    ABR_OBS_DIM = 8
};

/* float64 rows of abr_env_observe_f64: everything the oracle records */
enum {
    ABR_F64_GLOBAL_TIME = 0, ABR_F64_REBUFFER_TIME, ABR_F64_STARTUP_TIME, ABR_F64_PLAY_TIME,
    ABR_F64_AVERAGE_LATENCY, ABR_F64_BUFFER_LEVEL, ABR_F64_PLAY_LENGTH, ABR_F64_LAST_BANDWIDTH,
    ABR_F64_CHUNK_ID, ABR_F64_PLAY_ID, ABR_F64_LAST_BITRATE, ABR_F64_FLAGS /* start_up | buffer_empty<<1 | buffer_full<<2 */,
    ABR_F64_HIST_N, ABR_F64_HIST_SUM_INV, ABR_F64_TICK, ABR_F64_DOWNLOAD_TIME,
    ABR_F64_DIM
};

/* Replaces: MPD (Simulator.py:11-17), QOEMetric (:19-24), NetworkInfo.interval
 * (:39-42), the single-ladder Chunk that run() indexes (:82,156), and the
 * constant a speed controller would return (:177; none ships, D8). */
typedef struct abr_env_config {
    int32_t n_rates;            /* len(mpd.chunks.bitrates), 1..ABR_MAX_RATES */
    int32_t video_length;       /* mpd.video_length [chunks]                  */
    double  chunk_length;       /* mpd.chunk_length [s]                       */
    double  max_buffer;         /* mpd.max_buffer (compared in seconds, :190) */
    double  start_up_length;    /* mpd.start_up_length [s]                    */
    double  interval;           /* network_info.interval [s]                  */
    double  rebuffer_weight;    /* qoe_metric.*                               */
    double  variance_weight;
    double  startup_weight;
    double  latency_weight;
    double  speed;              /* play_speed, constant                       */
    double  ladder[ABR_MAX_RATES];
    int32_t max_ticks;          /* per-episode bound on 0.01 s ticks; <=0: 32 * V * ceil(L/dt) */
    int32_t auto_reset;         /* !=0: a lane that finishes is re-armed (same trace, same offset)
                                   inside the step; the obs returned is the new episode's first */
} abr_env_config;

typedef struct abr_env abr_env;   /* opaque host-side handle */

int  abr_abi_version(void);
const char *abr_last_error(void);

/* Bytes of device workspace (256-B aligned) abr_env_create needs for n_lanes. */
int abr_env_workspace_bytes(const abr_env_config *cfg, int64_t n_lanes, size_t *bytes_out);

/*
 * Replaces Simulator.__init__/set_qoe_metric/set_network_info/set_mpd
 * (Simulator.py:46-77).  traces_dev: all bandwidth traces back to back
 * (float64, same unit as the ladder); trace t is traces_dev[trace_off_dev[t] ..
 * + trace_len_dev[t]).  The three trace arrays must outlive the handle.
 * Builds the universal tick tables on the host in float64 (the reference's
 * global_time is lane-independent) and uploads them into the workspace.
 */
int abr_env_create(const abr_env_config *cfg, const double *traces_dev,
                   const int64_t *trace_off_dev, const int32_t *trace_len_dev, int32_t n_traces,
                   int64_t n_lanes, void *workspace_dev, size_t workspace_bytes, void *stream,
                   abr_env **env_out);
int abr_env_destroy(abr_env *env);

/* When the lanes of this handle are a shard of a larger job: global id of lane 0,
 * used only as the counter of the built-in random policy so that a sharded run
 * reproduces the unsharded one.  Default 0. */
int abr_env_set_lane_id_base(abr_env *env, int64_t lane_id_base);

/* One constant play speed per lane instead of config.speed (what a per-lane speed
 * controller that always answers the same value would do, Simulator.py:176-177).
 * speeds_dev: float64 [n_lanes], > 0, must stay valid from this call until the handle is
 * destroyed or another call replaces it.  The pointer is LATCHED: running episodes keep the
 * speeds they started with; the next abr_env_reset picks the new ones up, and that reset must
 * cover all lanes (lane_mask_dev == NULL, else ABR_E_INVALID).  On a handle that has seen neither
 * abr_env_reset nor abr_env_notify_restore there is no episode to protect and the call takes
 * effect at once.  NULL restores the single speed.  Event-driven kernels only. */
int abr_env_set_lane_speeds(abr_env *env, const double *speeds_dev);   /* latched: see above */

/* What a speed controller answers, call by call (Simulator.py:176-177: get_next_speed() is
 * asked at the first playing tick of every played chunk, i.e. whenever play_length == 0).
 * speeds_dev: float64 [n_rows][n_lanes]; the p-th played chunk of lane i plays at
 * speeds_dev[min(p, n_rows - 1) * n_lanes + i] (the last row repeats).  n_rows == 1 is
 * abr_env_set_lane_speeds.  Same lifetime and latching rules.  Event-driven kernels only. */
int abr_env_set_speed_schedule(abr_env *env, const double *speeds_dev, int32_t n_rows);

/* Per-chunk bitrate ladders: br_table_dev float64 [video_length][n_rates], caller-owned, valid
 * from this call until the handle is destroyed or another call replaces it; NULL restores
 * config.ladder.  This is the evident intent of set_mpd's one-ladder-per-line file
 * (Simulator.py:71-76), which run() itself cannot consume (it indexes a single Chunk,
 * Simulator.py:82,156, and raises AttributeError on the list) -- so it is BUILD-DEFINED:
 * target_size = br[chunk_id][action] * chunk_length, and the variance term of calculate_qoe
 * (and of the per-step reward) is |br[i][a_i] - br[i+1][a_(i+1)]|, each bitrate from its own
 * chunk's ladder.  Latched like abr_env_set_lane_speeds: picked up by the next reset of ALL lanes. */
int abr_env_set_bitrate_table(abr_env *env, const double *br_table_dev);

/* Resume: the whole simulator state is the workspace, so a checkpoint is a copy of it.  After copying
 * a checkpointed workspace into the workspace of a handle built with the same config, lane count
 * and (already set) speeds / bitrate table, call this: it marks the handle as carrying episodes in
 * flight, so that later setter calls are latched again.  It first reads the workspace's layout tag back
 * (ABI 4; one device synchronisation) and returns ABR_E_WORKSPACE -- the handle then stays as it was, the
 * workspace holds the foreign bytes and must be re-initialised by abr_env_reset of all lanes or restored
 * again -- if the bytes were laid out by another ABI version, lane count or configuration. */
int abr_env_notify_restore(abr_env *env);

/* Which kernels serve reset/step: 2 = event-driven (exact closed-form stepping of the
 * float64 tick sequences) with each lane's download side and player side on two waves of one
 * workgroup; 0 = event-driven, one thread per lane; 1 = one loop trip per 0.01 s tick;
 * 5 = as 2 with a third wave per 64 lanes for the service tail of a decision (bandwidth = size /
 * time, history, reward, observation, episode end);
 * 3 (default) = whichever is fastest at this size: 5 up to 65 536 lanes, 2 up to 131 072 lanes,
 * 0 above -- and 0 at every size for launches of ONE decision (abr_env_step, the per-decision
 * launches of abr_env_step_mpc, a fused call with n_steps == 1).
 * 0, 2 and 5 produce identical state and outputs in every case (the workspace is interchangeable between them, also
 * mid-episode).  1 is an independent cross-check: it agrees with them on every lane that ends its episode, but it visits
 * ticks in blocks, so a lane that runs into max_ticks (ABR_DONE_TIMEOUT -- build-defined: the reference has no time-out)
 * is frozen at its block boundary: same done bits, different frozen counters in that lane's last observation.
 * 4 (the asynchronous pipeline of round 3), 6 (the ring-coupled role pipeline of round 5) and 7 (round 5: download and player
 * wave in lock-step through LDS counters, the service wave behind a ring) were measured slower than -- or, 7, within 1 % of --
 * what 3 selects and are not in the product library: ABR_E_UNSUPPORTED, see abr_env_has_impl. */
int abr_env_set_impl(abr_env *env, int32_t impl);

/* 1 if this build of the library can run implementation `impl` (0..7), else 0.  The product library: 0, 1, 2, 3, 5. */
int abr_env_has_impl(int32_t impl);

/* The implementation (0, 1, 2 or 5; never 3) the handle resolves to right now: fused != 0 for
 * abr_env_step_random / abr_env_step_script with more than one decision per call, 0 for launches of
 * one decision (abr_env_step, abr_env_step_mpc, n_steps == 1). */
int abr_env_get_effective_impl(abr_env *env, int32_t fused, int32_t *impl_out);

/*
 * run() state init (Simulator.py:95-133) plus the idle ticks up to the first
 * get_next_bitrate call site.  Lane i uses trace trace_id_dev[i]; its
 * bandwidths[idx] is trace[(start_offset_dev[i] + idx) % len] (D7: wrap is
 * build-defined).  lane_mask_dev (nullable): only lanes with a non-zero byte
 * are reset.  obs_out_dev: float32 [ABR_OBS_DIM][n_lanes] (nullable).
 * A lane whose trace id is outside [0, n_traces) or whose start offset is negative is
 * frozen with ABR_DONE_BADARG (the reference would raise IndexError at Simulator.py:159).
 * Every reset of a lane after its first starts the next episode number of the built-in
 * counter-based policy (abr_env_step_random), so repeated episodes draw fresh actions.
 */
int abr_env_reset(abr_env *env, const int32_t *trace_id_dev, const int32_t *start_offset_dev,
                  const uint8_t *lane_mask_dev, float *obs_out_dev, void *stream);

/*
 * One chunk decision per lane: actions_dev[i] is what get_next_bitrate would
 * have returned (Simulator.py:155); the lane then runs ticks T4..T9 and on,
 * to its next call site or to simulation_end.  reward_out_dev[i] =
 *   wr * d(rebuffer_time) + ws * d(start_up_time) + wv * |br[a] - br[a_prev]|
 * (the per-step split of calculate_qoe, Simulator.py:79-86; the first step's
 * deltas start from 0, so sum(reward) + wl * average_latency == run()'s return).
 * Lanes already done are left untouched and report their done bits again.
 */
int abr_env_step(abr_env *env, const int32_t *actions_dev, float *obs_out_dev,
                 float *reward_out_dev, uint8_t *done_out_dev, void *stream);

/*
 * n_steps fused decisions per lane with the built-in random policy
 * action = philox4x32-10(key=seed, ctr=(lane, episode_step, episode_no, 0)) % n_rates.
 * Outputs (all nullable) are [n_steps][...] slabs: obs [n_steps][ABR_OBS_DIM][n_lanes],
 * reward/done/actions [n_steps][n_lanes].  Lane state is read and written once per call.  Under the one-thread-per-lane
 * kernels (impl 0) lanes progress independently; the role-split kernels (impl 2 / 5) meet at ONE workgroup barrier per
 * decision (64 lanes, two or three waves), which is what `auto` prefers up to 131 072 lanes.
 */
int abr_env_step_random(abr_env *env, int32_t n_steps, uint64_t seed, float *obs_out_dev,
                        float *reward_out_dev, uint8_t *done_out_dev, int32_t *actions_out_dev,
                        void *stream);

/*
 * n_steps fused decisions per lane whose actions are given up front: actions_dev int32
 * [n_steps][n_lanes], actions_dev[s][i] = what get_next_bitrate returns at the s-th call site
 * lane i reaches in this call (Simulator.py:155 with a scripted abr_controller).  Outputs as
 * abr_env_step_random.  A lane whose scripted action is outside [0, n_rates) is frozen with
 * ABR_DONE_BADACT at that step, as in abr_env_step.
 */
int abr_env_step_script(abr_env *env, int32_t n_steps, const int32_t *actions_dev, float *obs_out_dev,
                        float *reward_out_dev, uint8_t *done_out_dev, void *stream);

/* calculate_qoe (Simulator.py:79-86) in the reference's operation order from the
 * lane's action history; meaningful for lanes whose episode is complete (with
 * auto_reset: the last completed episode).  qoe_out_dev: float64 [n_lanes]. */
int abr_env_episode_qoe(abr_env *env, double *qoe_out_dev, void *stream);

/* Full float64 observation, [ABR_F64_DIM][n_lanes]. */
int abr_env_observe_f64(abr_env *env, double *out_dev, void *stream);

/* Device pointers into the workspace for consumers that need exact float64
 * state (the MPC adapter) or want to checkpoint it. */
typedef struct abr_env_state_view {
    int64_t n_lanes;
    const int32_t *chunk_id;      /* [n_lanes] */
    const int32_t *last_bitrate;  /* [n_lanes], -1 before the first chunk */
    const double  *buffer_level;  /* [n_lanes] */
    double        *hist_n;        /* [n_lanes] len(previous_bandwidths), as float64 */
    double        *hist_sum_inv;  /* [n_lanes] sum(1/x for x in previous_bandwidths), list order */
    const uint8_t *done;          /* [n_lanes] ABR_DONE_* bits */
    const uint8_t *action_hist;   /* [video_length][n_lanes] previous_bitrates */
    const double  *bw_hist;       /* [video_length][n_lanes] previous_bandwidths */
} abr_env_state_view;
int abr_env_get_state(abr_env *env, abr_env_state_view *view_out);

/* ---------------------------------------------------------------------- */
/* MPC lookahead                                                            */
/* ---------------------------------------------------------------------- */

/* Replaces what MPCBitrateController reads through the player protocol
 * (mpc.py:56-57: get_mpd, get_qoe_metric) and its horizon (mpc.py:59). */
typedef struct abr_mpc_config {
    int32_t n_rates;            /* len(mpd.chunks[0].bitrates)  mpc.py:173 */
    int32_t horizon;            /* 2..ABR_MAX_HORIZON (the reference crashes at 1, mpc.py:186) */
    int32_t video_length;       /* len(mpd.chunks) */
    int32_t clip_horizon;       /* !=0: H_eff = min(H, V - chunk) (D12; the reference raises IndexError) */
    double  chunk_length;       /* mpd.chunk_length  mpc.py:108,117,151 */
    double  max_buffer;         /* mpd.max_buffer    mpc.py:108 */
    double  variance_weight;    /* qoe.*             mpc.py:158-160 */
    double  rebuffer_weight;
    double  startup_weight;     /* multiplies the hard-zero startup_delay (mpc.py:141) */
} abr_mpc_config;

/*
 * MPCBitrateController.next_bitrate() (mpc.py:181-186) for n_lanes independent
 * players: harmonic-mean prediction fed back H times (mpc.py:81-93), exhaustive
 * evaluation of objective() (mpc.py:120-162) over all n_rates^horizon combos in
 * scipy.optimize.brute's C order with first-minimum tie-break (mpc.py:178), and
 * int(result[0]).
 *  chunk_dev/prev_bitrate_dev/buffer_dev: chunk_info.chunk_number /
 *      .previous_bitrate / .buffer_level (mpc.py:124,132,136)
 *  hist_n_dev/hist_sum_inv_dev: IN-OUT summary of chunk_info.previous_bandwidths
 *      (length and sum of reciprocals in list order); grown by H predictions
 *      exactly as the reference mutates the caller's list (mpc.py:92, D9)
 *  br_table_dev/sz_table_dev: [video_length][n_rates] mpd.chunks[i].bitrates/.sizes
 *  action_out_dev int32 [n_lanes]; best_flat_out_dev (nullable) int32 [n_lanes]
 *  flat arg-min index; best_J_out_dev (nullable) float64 [n_lanes];
 *  lane_mask_dev (nullable): lanes with a zero byte are skipped entirely
 *  (no history mutation, outputs untouched).
 *  previous_bitrate indexes the ladder as Python does (mpc.py:132,148): -n_rates..-1 wrap to
 *  the top (the env's "no previous chunk" value -1 means the highest rate).
 *  Lanes the reference would raise on report action -1 (flat -1, J NaN) and keep their
 *  history: an empty or zero history (ZeroDivisionError, mpc.py:88,90, D13), a
 *  previous_bitrate outside [-n_rates, n_rates) (IndexError) and, without clip_horizon,
 *  chunk + horizon > video_length (IndexError, mpc.py:126, D12).
 */
int abr_mpc_select(const abr_mpc_config *cfg, const int32_t *chunk_dev,
                   const int32_t *prev_bitrate_dev, const double *buffer_dev, double *hist_n_dev,
                   double *hist_sum_inv_dev, const double *br_table_dev, const double *sz_table_dev,
                   const uint8_t *lane_mask_dev, int32_t *action_out_dev,
                   int32_t *best_flat_out_dev, double *best_J_out_dev, int64_t n_lanes,
                   void *stream);

/*
 * The reference's alternative predictor and utility.  PARITY UNPINNED:
 * the predictor needs statsmodels (mpc.py:4,74), which is not installable here, and no test
 * of the reference touches either; what is implemented is the documented rule below, checked
 * for self-consistency only.
 *  ABR_PREDICT_EXPSMOOTHING  predict_throughput(..., method="expsmoothing") (mpc.py:72-79):
 *      SimpleExpSmoothing(history).fit(0.5) and its `horizon` out-of-sample forecasts, i.e.
 *      the last smoothed level repeated: level(t) = 0.5*y(t) + 0.5*level(t-1), with the
 *      initial level chosen to minimise the sum of squared one-step-ahead errors (closed
 *      form; statsmodels' default `estimated` initialisation finds it numerically).  Needs
 *      the throughput history itself: entry t of lane i at hist_dev[t * hist_stride + i],
 *      hist_len_dev[i] entries (for an environment: abr_env_state_view.bw_hist with stride
 *      n_lanes and chunk_id as the length).  Unlike the harmonic branch it does not grow the
 *      history (no D9) and ignores hist_n / hist_sum_inv.
 *  ABR_UTILITY_LOG  log_bitrate_utility (mpc.py:99-102) with the arity its call sites need:
 *      u = log(bitrate / highest bitrate of that chunk) in place of the identity utility in
 *      video_quality and quality_variance (mpc.py:146-149).
 */
#define ABR_PREDICT_HARMONIC 0
#define ABR_PREDICT_EXPSMOOTHING 1
#define ABR_UTILITY_IDENTITY 0
#define ABR_UTILITY_LOG 1
typedef struct abr_mpc_options {
    int32_t predictor;            /* ABR_PREDICT_* */
    int32_t utility;              /* ABR_UTILITY_* */
    const double *hist_dev;       /* ABR_PREDICT_EXPSMOOTHING only */
    int64_t hist_stride;
    const int32_t *hist_len_dev;
    void *scratch_dev;            /* optional, any predictor: abr_mpc_scratch_bytes() of caller-owned, 8-byte
                                     aligned device memory.  With it the predictor (ten dependent IEEE
                                     divisions per lane at horizon 5) runs as a kernel of its own, one thread
                                     per lane, instead of on 1 of the lane's n_rates^2 search threads: same
                                     results, ~10 % less time.  NULL: everything in one kernel. */
    size_t scratch_bytes;
    int32_t mask_is_done;         /* != 0: lane_mask_dev holds ABR_DONE_* bits (an environment's `done` array, as
                                     is): a lane is skipped iff its byte is NON-zero, and reports action -1.
                                     0: lane_mask_dev is a plain mask, lanes with a zero byte are skipped */
    int32_t reserved_;
} abr_mpc_options;

/* Bytes of scratch abr_mpc_options.scratch_dev needs for n_lanes at cfg->horizon. */
int abr_mpc_scratch_bytes(const abr_mpc_config *cfg, int64_t n_lanes, size_t *bytes_out);

/* abr_mpc_select with options; opt == NULL is abr_mpc_select. */
int abr_mpc_select_opt(const abr_mpc_config *cfg, const abr_mpc_options *opt,
                       const int32_t *chunk_dev, const int32_t *prev_bitrate_dev,
                       const double *buffer_dev, double *hist_n_dev, double *hist_sum_inv_dev,
                       const double *br_table_dev, const double *sz_table_dev,
                       const uint8_t *lane_mask_dev, int32_t *action_out_dev,
                       int32_t *best_flat_out_dev, double *best_J_out_dev, int64_t n_lanes,
                       void *stream);

/*
 * MPC-driven rollout, fused on the device: for n_steps decisions, each lane's action is
 * MPCBitrateController.next_bitrate() (mpc.py:181-186) evaluated on the lane's OWN
 * environment state -- chunk_number = chunk_id, previous_bitrate = previous_bitrates[-1],
 * buffer_level, and previous_bandwidths as the environment's (len, sum of reciprocals)
 * summary, which the predictor grows by `horizon` entries per call exactly as the reference
 * mutates the shared list (mpc.py:92, D9) -- followed by the download of that chunk
 * (Simulator.py:155-170; abr_env_step).  This is the wiring the reference leaves open
 * (get_next_bitrate(...) at Simulator.py:155 vs next_bitrate() at mpc.py:181, D5/D6); no host
 * round trip or host-side tensor work happens between decisions.
 *  br_table_dev/sz_table_dev: [video_length][n_rates] as for abr_mpc_select; cfg->n_rates and
 *  cfg->video_length must equal the environment's.
 *  Lanes whose done bits are set take no decision (action -1) and stay frozen.  A lane the
 *  reference would raise on (empty history at chunk 0: ZeroDivisionError, D13; horizon past
 *  the video end without clip_horizon, D12) downloads bitrate 0 and keeps its history.
 *  Outputs (all nullable): obs [n_steps][ABR_OBS_DIM][n_lanes], reward/done/actions
 *  [n_steps][n_lanes], as abr_env_step_random.  Event-driven kernels only (impl 0 or 2).
 */
int abr_env_step_mpc(abr_env *env, const abr_mpc_config *cfg, const double *br_table_dev,
                     const double *sz_table_dev, int32_t n_steps, float *obs_out_dev,
                     float *reward_out_dev, uint8_t *done_out_dev, int32_t *actions_out_dev,
                     void *stream);

/* Diagnostic: the full objective grid of ONE lane, J_out_dev float64
 * [n_rates^horizon], given explicit predictions pred_dev[horizon]. */
int abr_mpc_objective_grid(const abr_mpc_config *cfg, int32_t chunk, int32_t prev_bitrate,
                           double buffer_level, const double *pred_dev, const double *br_table_dev,
                           const double *sz_table_dev, double *J_out_dev, void *stream);

/* Diagnostic: count independent chains "x <- fl(x + c), up to n times, stop right after the
 * first result that is >= thr (stop_kind 0, c > 0), <= thr (1, c < 0) or < thr (2, c < 0)" --
 * the float64 sequences of Simulator.py:160-163 and :184,:190-194 -- advanced by the kernels'
 * exact closed form (csrc/abr_exact_jump.h) on the device.  estimate_bias 0 is the product
 * path; +4 / -4 spoil the jump-length estimate so that the exact fallback search runs.
 * Outputs per case: final x, additions performed, 1 if it stopped on the predicate. */
int abr_debug_chain(int32_t stop_kind, int32_t estimate_bias, const double *x0_dev,
                    const double *c_dev, const double *thr_dev, const int32_t *n_dev, int64_t count,
                    double *x_out_dev, int32_t *a_out_dev, uint8_t *hit_out_dev, void *stream);

/* Diagnostic: count independent drains "x <- fl(x - sd), up to n times, stop right after the first result that is <= 0" --
 * buffer_level -= play_speed * dt, Simulator.py:184, :194 -- at ONE subtrahend sd for all cases, advanced on the device the way
 * the environment kernels do when every lane plays at the same speed: by the per-binade cascade built for levels below
 * max_level (csrc/abr_exact_jump.h: drain_cascade; abr_env_create builds it for max_buffer + chunk_length), or by the general
 * chain for a wave that holds a value at or above the cascade's top.  Outputs per case as abr_debug_chain; *stages_out
 * (nullable, host) = binades the cascade covers.  ABR_E_UNSUPPORTED when no cascade exists for (sd, max_level). */
int abr_debug_drain(double sd, double max_level, const double *x0_dev, const int32_t *n_dev, int64_t count,
                    double *x_out_dev, int32_t *a_out_dev, uint8_t *hit_out_dev, int32_t *stages_out, void *stream);

/* Diagnostic: a self-check of the PRODUCT build's role-split kernels (impl 2 and 5).  Their service code re-reads the launch's
 * parameter block from the kernel-argument segment every iteration instead of holding it in registers, which is only right
 * while that block is the kernels' first argument; a checking instance of each of the two kernel templates (the product
 * instances' signature) is launched once (one workgroup, no lane state touched) with a sentinel in the block and reports
 * whether the re-read saw it.  result_dev: uint32 [2] device memory (three-wave kernel, two-wave kernel): 1 = seen,
 * 2 = not seen, 0 = that launch never ran. */
int abr_debug_selfcheck(abr_env *env, uint32_t *result_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ABR_ENV_H */
