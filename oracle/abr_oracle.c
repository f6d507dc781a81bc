/*
 * abr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, scalar, float64 restatement of the reference algorithm for the
 * hot path (Elliotshui/ABRSimulator: Simulator.py tick loop, mpc.py lookahead).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (abrsimulator_amd/) never does.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here bit-for-bit against fixtures in tests/golden/ that were produced by
 * running the reference itself (tools/gen_golden.py): mpc.py as shipped, and
 * Simulator.run() under the three control-flow repairs R1-R3 of SURVEY.md 8(c).
 *
 * Compile with -O2 -ffp-contract=off (no FMA contraction: the reference is
 * CPython float arithmetic, one IEEE-754 double rounding per operation).
 *
 * Every statement group cites the reference line it restates.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t n_rates;          /* len(mpd.chunks.bitrates)             Simulator.py:4-6   */
    int32_t video_length;     /* mpd.video_length  [chunks]           Simulator.py:13    */
    double  chunk_length;     /* mpd.chunk_length  [s]                Simulator.py:14    */
    double  max_buffer;       /* mpd.max_buffer                       Simulator.py:15    */
    double  start_up_length;  /* mpd.start_up_length [s]              Simulator.py:16    */
    double  interval;         /* network_info.interval [s]            Simulator.py:41    */
    double  rebuffer_weight;  /* qoe_metric.*                         Simulator.py:21-24 */
    double  variance_weight;
    double  startup_weight;
    double  latency_weight;
    double  speed;            /* constant returned by get_next_speed  Simulator.py:177   */
    double  ladder[16];       /* mpd.chunks.bitrates                                     */
    /* Build-defined generalisation (the reference raises AttributeError when mpd.chunks is the
     * list set_mpd builds, Simulator.py:71-76 vs :82,156): NULL, or [video_length][n_rates]
     * per-chunk ladders -- target_size uses the downloading chunk's row, the variance term of
     * calculate_qoe each chunk's own row.  Parity fixtures never set it. */
    const double *br_table;
    /* speed controller (Simulator.py:176-177): NULL = the constant `speed`; else the answer to
     * the p-th get_next_speed() call -- one call per played chunk, at its first playing tick --
     * is speed_sched[min(p, speed_rows - 1) * speed_stride] */
    const double *speed_sched;
    int32_t speed_rows;
    int64_t speed_stride;
} oracle_env_cfg;

static inline double chunk_bitrate(const oracle_env_cfg *c, int chunk, int rate) {
    return c->br_table ? c->br_table[(size_t)chunk * c->n_rates + rate] : c->ladder[rate];
}

/* what the ABR callback sees (Simulator.py:155) plus run()'s locals then */
typedef struct {
    double global_time, rebuffer_time, start_up_time, play_time;
    double average_latency, buffer_level, play_length, instant_latency;
    double last_bandwidth;            /* previous_bandwidths[-1] or 0 */
    int32_t chunk_id, play_id, last_bitrate /* previous_bitrates[-1] or -1 */;
    int32_t start_up, buffer_empty, buffer_full;
} oracle_step_rec;

typedef struct {
    double qoe, rebuffer_time, start_up_time, average_latency;
    double global_time, buffer_level, play_time;
    int64_t ticks;
    int32_t play_id, chunk_id;
} oracle_final_rec;

/* policy callback: returns the bitrate index for this chunk */
typedef int32_t (*oracle_policy_fn)(void *ctx, const oracle_step_rec *obs,
                                    const double *prev_bw, int32_t n_prev_bw);

/* Python's max(0, x): first maximal element of (0, x) */
static inline double pymax0(double x) { return (x > 0) ? x : 0.0; }

/*
 * One episode of Simulator.run() (Simulator.py:93-210, with R1-R3).
 * trace/trace_len/offset: bandwidths[idx] := trace[(offset + idx) % trace_len]
 *   (D7: the reference raises IndexError past the end; wrap is build-defined
 *    and never exercised by the parity fixtures).
 * actions != NULL: replay; else policy(ctx, ...) is called.
 * steps[video_length], bw_out[video_length] (measured throughputs) may be NULL.
 * max_ticks: safety bound (returns -2 when hit).
 */
int oracle_env_episode(const oracle_env_cfg *c, const double *trace, int32_t trace_len,
                       int32_t offset, const int32_t *actions, oracle_policy_fn policy,
                       void *ctx, oracle_step_rec *steps, double *bw_out, int32_t *act_out,
                       oracle_final_rec *fin, int64_t max_ticks)
{
    const int V = c->video_length;
    double *previous_bandwidths = (double *)malloc(sizeof(double) * (size_t)(V > 0 ? V : 1));
    int32_t *previous_bitrates = (int32_t *)malloc(sizeof(int32_t) * (size_t)(V > 0 ? V : 1));
    if (!previous_bandwidths || !previous_bitrates) return -1;

    /* Simulator.py:95-130 */
    int chunk_id = 0, available_id = -1, current_bitrate = -1;
    double downloaded_size = 0.0, target_size = 0.0;
    int download_pause = 1;
    double download_time = 0;
    double buffer_level = 0;
    const double max_buffer = c->max_buffer;
    int buffer_empty = 1, buffer_full = 0;
    int play_id = 0;
    double play_length = 0, play_time = 0, play_speed = 0;
    int play_pause = 1;
    double instant_latency = 0, average_latency = 0;
    int start_up = 1, simulation_end = 0;
    double global_time = 0.0, rebuffer_time = 0.0, start_up_time = 0.0;
    const double dt = 0.01;                                   /* :133 */
    int64_t ticks = 0;
    int rc = 0;
    int speed_calls = 0;                                       /* get_next_speed() calls so far */

    while (!simulation_end) {                                  /* :135 */
        if (ticks >= max_ticks) { rc = -2; break; }
        /* :137-140 */
        if (start_up) start_up_time += dt;
        else if (buffer_empty) rebuffer_time += dt;
        /* :143-145 (+R2) */
        available_id = (int)(global_time / c->chunk_length) - 1;
        if (available_id < chunk_id || buffer_full) download_pause = 1;
        else download_pause = 0;
        /* :148-149 (+R3) */
        if (buffer_empty || start_up) play_pause = 1;
        else play_pause = 0;
        /* :152-170 */
        if (!download_pause) {
            if (download_time == 0) {                          /* :154 */
                oracle_step_rec r;
                r.global_time = global_time; r.rebuffer_time = rebuffer_time;
                r.start_up_time = start_up_time; r.play_time = play_time;
                r.average_latency = average_latency; r.buffer_level = buffer_level;
                r.play_length = play_length; r.instant_latency = instant_latency;
                r.last_bandwidth = chunk_id ? previous_bandwidths[chunk_id - 1] : 0.0;
                r.chunk_id = chunk_id; r.play_id = play_id;
                r.last_bitrate = chunk_id ? previous_bitrates[chunk_id - 1] : -1;
                r.start_up = start_up; r.buffer_empty = buffer_empty; r.buffer_full = buffer_full;
                if (steps) steps[chunk_id] = r;
                current_bitrate = actions ? actions[chunk_id]
                                          : policy(ctx, &r, previous_bandwidths, chunk_id); /* :155 */
                if (current_bitrate < 0 || current_bitrate >= c->n_rates) { rc = -3; break; }
                target_size = chunk_bitrate(c, chunk_id, current_bitrate) * c->chunk_length; /* :156 */
            }
            int64_t bandwidth_idx = (int64_t)(global_time / c->interval);                 /* :158 */
            double bandwidth = trace[(offset + bandwidth_idx) % trace_len];                /* :159 */
            downloaded_size = downloaded_size + bandwidth * dt;                            /* :160 */
            download_time += dt;                                                           /* :161 */
            if (downloaded_size >= target_size) {                                          /* :163 */
                previous_bandwidths[chunk_id] = downloaded_size / download_time;           /* :164 */
                previous_bitrates[chunk_id] = current_bitrate;                             /* :165 */
                chunk_id += 1;
                downloaded_size = 0;
                download_time = 0;
                buffer_level += c->chunk_length;                                           /* :170 */
            }
        }
        /* :174-187 */
        if (!play_pause) {
            if (play_length == 0) {                                                        /* :176-177 */
                if (c->speed_sched) {
                    int row = speed_calls < c->speed_rows ? speed_calls : c->speed_rows - 1;
                    play_speed = c->speed_sched[(size_t)row * c->speed_stride];
                    speed_calls++;
                } else play_speed = c->speed;
            }
            instant_latency = global_time - play_time;                                     /* :179 */
            average_latency = (average_latency * play_time + instant_latency)
                              / (play_time + play_speed * dt);                             /* :180 */
            play_time += play_speed * dt;
            play_length += play_speed * dt;
            buffer_level -= play_speed * dt;
            if (play_length >= c->chunk_length) { play_length = 0; play_id += 1; }        /* :185-187 */
        }
        /* :190-198 */
        buffer_full = (buffer_level >= max_buffer);
        if (buffer_level <= 0) { buffer_level = 0; buffer_empty = 1; }
        else buffer_empty = 0;
        /* :201-202 */
        if (start_up && buffer_level >= c->start_up_length) start_up = 0;
        /* :205 */
        global_time += dt;
        ticks++;
        /* :207-208 */
        if (chunk_id >= V) simulation_end = 1;
    }

    if (rc == 0 && fin) {
        /* calculate_qoe, Simulator.py:79-86 */
        double variance = 0;
        for (int i = 0; i < V - 1; i++)
            variance += fabs(chunk_bitrate(c, i, previous_bitrates[i]) -
                             chunk_bitrate(c, i + 1, previous_bitrates[i + 1]));
        fin->qoe = c->rebuffer_weight * rebuffer_time + c->variance_weight * variance
                 + c->startup_weight * start_up_time + c->latency_weight * average_latency;
        fin->rebuffer_time = rebuffer_time; fin->start_up_time = start_up_time;
        fin->average_latency = average_latency; fin->global_time = global_time;
        fin->buffer_level = buffer_level; fin->play_time = play_time;
        fin->ticks = ticks; fin->play_id = play_id; fin->chunk_id = chunk_id;
    }
    if (rc == 0) {
        if (bw_out) memcpy(bw_out, previous_bandwidths, sizeof(double) * (size_t)V);
        if (act_out) memcpy(act_out, previous_bitrates, sizeof(int32_t) * (size_t)V);
    }
    free(previous_bandwidths);
    free(previous_bitrates);
    return rc;
}

/* Batch helper: n_lanes independent replay episodes (lane i uses
 * traces + trace_off[trace_id[i]], length trace_len[trace_id[i]]).
 * steps: [n_lanes][V], bw_out: [n_lanes][V], fin: [n_lanes]. Returns total ticks or <0. */
int64_t oracle_env_batch(const oracle_env_cfg *c, const double *traces, const int64_t *trace_off,
                         const int32_t *trace_len, const int32_t *trace_id,
                         const int32_t *offset, const int32_t *actions, int32_t n_lanes,
                         oracle_step_rec *steps, double *bw_out, oracle_final_rec *fin,
                         int64_t max_ticks)
{
    const int V = c->video_length;
    int64_t total = 0;
    for (int32_t i = 0; i < n_lanes; i++) {
        oracle_final_rec f;
        int t = trace_id[i];
        int rc = oracle_env_episode(c, traces + trace_off[t], trace_len[t], offset[i],
                                    actions + (size_t)i * V, NULL, NULL,
                                    steps ? steps + (size_t)i * V : NULL,
                                    bw_out ? bw_out + (size_t)i * V : NULL, NULL, &f, max_ticks);
        if (rc) return rc;
        if (fin) fin[i] = f;
        total += f.ticks;
    }
    return total;
}

/* Same, with one constant play speed per lane (what a per-lane speed controller that
 * always answers the same value would do, Simulator.py:176-177).  speeds[n_lanes]. */
int64_t oracle_env_batch_speeds(const oracle_env_cfg *c, const double *traces,
                                const int64_t *trace_off, const int32_t *trace_len,
                                const int32_t *trace_id, const int32_t *offset,
                                const int32_t *actions, const double *speeds, int32_t n_lanes,
                                oracle_step_rec *steps, double *bw_out, oracle_final_rec *fin,
                                int64_t max_ticks)
{
    const int V = c->video_length;
    int64_t total = 0;
    for (int32_t i = 0; i < n_lanes; i++) {
        oracle_env_cfg ci = *c;
        ci.speed = speeds[i];
        oracle_final_rec f;
        int t = trace_id[i];
        int rc = oracle_env_episode(&ci, traces + trace_off[t], trace_len[t], offset[i],
                                    actions + (size_t)i * V, NULL, NULL,
                                    steps ? steps + (size_t)i * V : NULL,
                                    bw_out ? bw_out + (size_t)i * V : NULL, NULL, &f, max_ticks);
        if (rc) return rc;
        if (fin) fin[i] = f;
        total += f.ticks;
    }
    return total;
}

/* Same, with a per-lane speed schedule: speeds[n_lanes][rows], the answers of lane i's speed
 * controller to its successive get_next_speed() calls (the last one repeats). */
int64_t oracle_env_batch_sched(const oracle_env_cfg *c, const double *traces,
                               const int64_t *trace_off, const int32_t *trace_len,
                               const int32_t *trace_id, const int32_t *offset,
                               const int32_t *actions, const double *speeds, int32_t rows,
                               int32_t n_lanes, oracle_step_rec *steps, double *bw_out,
                               oracle_final_rec *fin, int64_t max_ticks)
{
    const int V = c->video_length;
    int64_t total = 0;
    for (int32_t i = 0; i < n_lanes; i++) {
        oracle_env_cfg ci = *c;
        ci.speed_sched = speeds + (size_t)i * rows; ci.speed_rows = rows; ci.speed_stride = 1;
        oracle_final_rec f;
        int t = trace_id[i];
        int rc = oracle_env_episode(&ci, traces + trace_off[t], trace_len[t], offset[i],
                                    actions + (size_t)i * V, NULL, NULL,
                                    steps ? steps + (size_t)i * V : NULL,
                                    bw_out ? bw_out + (size_t)i * V : NULL, NULL, &f, max_ticks);
        if (rc) return rc;
        if (fin) fin[i] = f;
        total += f.ticks;
    }
    return total;
}

/* ------------------------------------------------------------------------
 * MPC (mpc.py)
 * ---------------------------------------------------------------------- */
typedef struct {
    int32_t n_rates;          /* len(mpd.chunks[0].bitrates)   mpc.py:173 */
    int32_t horizon;          /* self.horizon                  mpc.py:59  */
    int32_t video_length;     /* len(mpd.chunks)                          */
    int32_t _pad;
    double  chunk_length;     /* mpd.chunk_length              mpc.py:108,117,151 */
    double  max_buffer;       /* mpd.max_buffer                mpc.py:108 */
    double  variance_weight;  /* qoe.*                         mpc.py:158-160 */
    double  rebuffer_weight;
    double  startup_weight;
} oracle_mpc_cfg;

/* predict_throughput(..., method="harmonic"), mpc.py:81-93, on an explicit
 * history list that is mutated (D9).  hist has room for *n + horizon. */
void oracle_mpc_predict_list(int32_t horizon, double *hist, int32_t *n, double *pred)
{
    for (int i = 0; i < horizon; i++) {
        int history_size = *n;
        double sum_inverse = 0;
        for (int k = 0; k < history_size; k++) sum_inverse += 1 / hist[k];
        double tp = history_size / sum_inverse;
        pred[i] = tp;
        hist[(*n)++] = tp;
    }
}

/* The same prediction carried as (n, S = sum of 1/x in list order).  Bit
 * identical to the list form because the list is summed in order and the new
 * term is appended last. */
void oracle_mpc_predict_ns(int32_t horizon, double *n, double *S, double *pred)
{
    for (int i = 0; i < horizon; i++) {
        double tp = *n / *S;
        pred[i] = tp;
        *S += 1 / tp;
        *n += 1;
    }
}

/* calc_wait mpc.py:104-109 */
static double calc_wait(const oracle_mpc_cfg *c, const double *sz, int chunk, double buffer_level,
                        int r, double bandwidth)
{
    double chunk_size = sz[(size_t)chunk * c->n_rates + r];
    double new_buffer = pymax0(buffer_level - chunk_size / bandwidth);
    double wait_time = new_buffer + c->chunk_length - c->max_buffer;
    return pymax0(wait_time);
}

/* next_buffer mpc.py:111-118 */
static double next_buffer(const oracle_mpc_cfg *c, const double *sz, int chunk, double buffer_level,
                          int r, double bandwidth)
{
    double chunk_size = sz[(size_t)chunk * c->n_rates + r];
    double wait_time = calc_wait(c, sz, chunk, buffer_level, r, bandwidth);
    double temp_buffer = pymax0(buffer_level - chunk_size / bandwidth);
    return pymax0(temp_buffer + c->chunk_length - wait_time);
}

/* objective mpc.py:120-162.  R_arg[h] in [0, n_rates). */
double oracle_mpc_objective(const oracle_mpc_cfg *c, const double *br, const double *sz,
                            int chunk, int prev_bitrate, double buffer_level,
                            const double *pred, const int32_t *R_arg)
{
    const int H = c->horizon, B = c->n_rates;
    int R[17];
    double buffer_vector[16];
    /* R = [previous_bitrate] + ... indexes bitrates[i] (mpc.py:132,148): a negative index
     * counts from the end of the list, as in Python (-1 = the highest rate) */
    R[0] = prev_bitrate < 0 ? prev_bitrate + B : prev_bitrate;
    for (int i = 0; i < H; i++) { R[i + 1] = R_arg[i]; buffer_vector[i] = 0.0; }
    buffer_vector[0] = buffer_level;
    double video_quality = 0, quality_variance = 0, rebuffer_time = 0, startup_delay = 0;
    for (int i = 0; i < H; i++) {
        const double *bri = br + (size_t)(chunk + i) * B;     /* bitrates[i]  :127-128 */
        const double *szi = sz + (size_t)(chunk + i) * B;     /* sizes[i]     :125-126 */
        video_quality += bri[R[i + 1]];                                          /* :146 */
        quality_variance += fabs(bri[R[i + 1]] - bri[R[i]]);                     /* :148-149 */
        /* :151-152  max(0, size, chunk_length): 3-argument max (D10) */
        double m = 0;
        if (szi[R[i + 1]] > m) m = szi[R[i + 1]];
        if (c->chunk_length > m) m = c->chunk_length;
        rebuffer_time += (m / pred[i] - buffer_vector[i]);
        if (i != H - 1)                                                          /* :154-156 (D11) */
            buffer_vector[i + 1] = next_buffer(c, sz, chunk, buffer_vector[i], R[i + 1], pred[i]);
    }
    double qoe_sum = (video_quality - c->variance_weight * quality_variance
                      - c->rebuffer_weight * rebuffer_time
                      - c->startup_weight * startup_delay);                      /* :158-160 */
    return -qoe_sum;
}

/* optimize_qoe + scipy.optimize.brute(finish=None): C-order grid, first
 * minimum (mpc.py:171-179).  J_out (B^H doubles) may be
 * NULL.  Returns the flat arg-min; *Jmin its value. */
int64_t oracle_mpc_brute(const oracle_mpc_cfg *c, const double *br, const double *sz,
                         int chunk, int prev_bitrate, double buffer_level, const double *pred,
                         double *J_out, double *Jmin)
{
    const int H = c->horizon, B = c->n_rates;
    int64_t total = 1;
    for (int i = 0; i < H; i++) total *= B;
    int32_t R[16];
    int64_t best = 0;
    double bestJ = INFINITY;
    for (int64_t f = 0; f < total; f++) {
        int64_t t = f;
        for (int i = H - 1; i >= 0; i--) { R[i] = (int32_t)(t % B); t /= B; }
        double J = oracle_mpc_objective(c, br, sz, chunk, prev_bitrate, buffer_level, pred, R);
        if (J_out) J_out[f] = J;
        if (J < bestJ || f == 0) { bestJ = J; best = f; }     /* numpy argmin: first minimum */
    }
    if (Jmin) *Jmin = bestJ;
    return best;
}

/* next_bitrate mpc.py:181-186 for n_lanes independent decisions with the
 * history carried as (n, S) in/out (D9).  action = first digit of the flat
 * arg-min.  flat_out / Jmin_out may be NULL. */
int oracle_mpc_select(const oracle_mpc_cfg *c, const double *br, const double *sz,
                      const int32_t *chunk, const int32_t *prev_bitrate, const double *buffer_level,
                      double *hist_n, double *hist_s, int32_t n_lanes,
                      int32_t *action_out, int64_t *flat_out, double *Jmin_out, double *pred_out)
{
    const int H = c->horizon, B = c->n_rates;
    int64_t lead = 1;
    for (int i = 1; i < H; i++) lead *= B;
    for (int32_t l = 0; l < n_lanes; l++) {
        double pred[16];
        if (chunk[l] + H > c->video_length) return -4;   /* D12: reference raises IndexError */
        oracle_mpc_predict_ns(H, &hist_n[l], &hist_s[l], pred);
        double Jm;
        int64_t f = oracle_mpc_brute(c, br, sz, chunk[l], prev_bitrate[l], buffer_level[l], pred,
                                     NULL, &Jm);
        action_out[l] = (int32_t)(f / lead);
        if (flat_out) flat_out[l] = f;
        if (Jmin_out) Jmin_out[l] = Jm;
        if (pred_out) memcpy(pred_out + (size_t)l * H, pred, sizeof(double) * (size_t)H);
    }
    return 0;
}

/* ------------------------------------------------------------------------
 * Composition: Simulator.run() driven by MPCBitrateController.next_bitrate()
 * ----------------------------------------------------------------------
 * The reference does not wire the two (D5/D6); the evident wiring is
 * chunk_info.previous_bandwidths = the simulator's own previous_bandwidths list
 * (Simulator.py:155 passes it, mpc.py:168 reads it), so the predictor's appended
 * predictions (mpc.py:92, D9) stay in the list the simulator keeps appending measured
 * throughputs to (Simulator.py:164).  The harmonic mean runs over that whole list in
 * order; carried here as (n, S = sum of 1/x in list order).
 * Build-defined edges, as in include/abr_env.h: an empty history (chunk 0; the reference
 * divides by zero, D13) takes bitrate 0 and leaves the history alone; near the video end
 * the horizon is clipped to V - chunk (D12; the reference raises IndexError). */
typedef struct {
    const oracle_mpc_cfg *m;
    const double *br, *sz;
    double n, S;
    int32_t seen;
} oracle_mpc_policy_ctx;

static int32_t oracle_mpc_policy(void *vctx, const oracle_step_rec *obs, const double *prev_bw,
                                 int32_t n_prev)
{
    oracle_mpc_policy_ctx *c = (oracle_mpc_policy_ctx *)vctx;
    for (int32_t i = c->seen; i < n_prev; i++) {          /* Simulator.py:164 appended these */
        c->S = c->S + 1 / prev_bw[i];
        c->n = c->n + 1;
    }
    c->seen = n_prev;
    if (!(c->n > 0)) return 0;                            /* D13 */
    const int H = c->m->horizon, V = c->m->video_length;
    double pred[16];
    oracle_mpc_predict_ns(H, &c->n, &c->S, pred);         /* D9: the list grows by H */
    oracle_mpc_cfg m = *c->m;
    int he = H;
    if (obs->chunk_id + H > V) he = V - obs->chunk_id;    /* D12 clip */
    m.horizon = he;
    int64_t lead = 1;
    for (int i = 1; i < he; i++) lead *= m.n_rates;
    int64_t f = oracle_mpc_brute(&m, c->br, c->sz, obs->chunk_id, obs->last_bitrate,
                                 obs->buffer_level, pred, NULL, NULL);
    return (int32_t)(f / lead);
}

/* n_lanes episodes of the composition.  act_out: [n_lanes][V] chosen bitrates. */
int64_t oracle_env_batch_mpc(const oracle_env_cfg *c, const oracle_mpc_cfg *m, const double *br,
                             const double *sz, const double *traces, const int64_t *trace_off,
                             const int32_t *trace_len, const int32_t *trace_id,
                             const int32_t *offset, int32_t n_lanes, oracle_step_rec *steps,
                             double *bw_out, int32_t *act_out, oracle_final_rec *fin,
                             int64_t max_ticks)
{
    const int V = c->video_length;
    int64_t total = 0;
    for (int32_t i = 0; i < n_lanes; i++) {
        oracle_mpc_policy_ctx ctx = { m, br, sz, 0.0, 0.0, 0 };
        oracle_final_rec f;
        int t = trace_id[i];
        int rc = oracle_env_episode(c, traces + trace_off[t], trace_len[t], offset[i], NULL,
                                    oracle_mpc_policy, &ctx,
                                    steps ? steps + (size_t)i * V : NULL,
                                    bw_out ? bw_out + (size_t)i * V : NULL,
                                    act_out ? act_out + (size_t)i * V : NULL, &f, max_ticks);
        if (rc) return rc;
        if (fin) fin[i] = f;
        total += f.ticks;
    }
    return total;
}

int oracle_abi_version(void) { return 1; }
