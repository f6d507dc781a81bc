// CPU harness for abr_lane_jump.h + abr_tick_tables.h: replays episodes through the
// event-driven lane step ON THE HOST so tests can compare it with the oracle on
// millions of lane-steps without a GPU.  Built by tests/test_lane_jump_cpu.py.
#include <stdint.h>
#include <stdlib.h>
#include "abr_lane_jump.h"
#include "abr_tick_tables.h"

struct Ctx {
    abrx::TickTables tt;
    abrx::Tables t;
    double ladder[16];
    int n_rates;
};

// lanej_predict_next_call against what the lane really does: [0] decisions, [1] of them gated by buffer_full (the next call
// site is not max(completing tick + 1, avail_next)), [2] gated ones the predictor covered, [3] predictions made, [4] wrong ones
static long long g_pred[5];

extern "C" {

// drain_to_zero against the reference's own loop (buffer_level -= speed*dt until <= 0, Simulator.py:184,:194), n cases:
// returns the index of the first case that differs (ticks taken, ran-dry flag, the value), or -1; stats[0] = cases that ran dry
int64_t lj_drain_check(const double *b0, const double *sd, const int32_t *m, int64_t n, long long *stats) {
    stats[0] = 0;
    for (int64_t c = 0; c < n; c++) {
        double b = b0[c];
        int32_t a = 0;
        while (a < m[c] && b > 0.0) { b = b - sd[c]; a++; }                 // the naive loop
        const bool zero = a > 0 && b <= 0.0;
        double bj = b0[c];
        int32_t aj = 0;
        const bool zj = abrx::drain_to_zero(bj, sd[c], m[c], aj);
        if (zj != zero || aj != a || bj != b) return c;
        stats[0] += zero ? 1 : 0;
    }
    return -1;
}

// drain_cascade (the per-binade cascade of ONE subtrahend, abr_exact_jump.h) against the same naive loop, n cases at the
// subtrahend sd with the cascade made for levels below max_level, exactly as abr_env_create makes it.  Returns the index of
// the first case that differs, -1 if none, -2 if no table exists for (sd, max_level); stats[0] = cases that ran dry,
// stats[1] = stages of the table, stats[2] = cases above the table (skipped: the kernels send those through the chains)
int64_t lj_cascade_check(double sd, double max_level, const double *b0, const int32_t *m, int64_t n, long long *stats) {
    const abrx::DrainTab tb = abrx::make_drain_tab(sd, max_level);
    stats[0] = 0; stats[1] = tb.n; stats[2] = 0;
    if (tb.n == 0) return -2;
    for (int64_t c = 0; c < n; c++) {
        if (!(b0[c] < tb.top)) { stats[2]++; continue; }
        double b = b0[c];
        int32_t a = 0;
        while (a < m[c] && b > 0.0) { b = b - sd; a++; }                    // the naive loop
        const bool zero = a > 0 && b <= 0.0;
        double bj = b0[c];
        int32_t aj = 0;
        const bool zj = abrx::drain_cascade(tb, sd, bj, m[c], aj);
        if (zj != zero || aj != a || bj != b) return c;
        stats[0] += zero ? 1 : 0;
    }
    return -1;
}

void lj_predict_stats(long long *out, int reset) {
    for (int i = 0; i < 5; i++) { out[i] = g_pred[i]; if (reset) g_pred[i] = 0; }
}

void *lj_create(double interval, double L, double speed, int32_t V, double max_buffer,
                double start_up_length, int32_t max_ticks, const double *ladder, int32_t n_rates) {
    Ctx *c = new Ctx;
    int32_t n_iv = (int32_t)((double)max_ticks * 0.01 / interval + 4.0);
    c->tt = abrx::build_tick_tables(interval, L, speed, V, max_ticks, n_iv);
    c->t.G = c->tt.G.data();
    c->t.interval_tick = c->tt.interval_tick.data();
    c->t.avail_tick = c->tt.avail_tick.data();
    c->t.L = L; c->t.sd = c->tt.sd; c->t.max_buffer = max_buffer; c->t.start_up_length = start_up_length;
    c->t.V = V; c->t.max_ticks = max_ticks; c->t.per_lane_speed = false;
    c->t.speed_rows = 0; c->t.speed_stride = 0; c->t.speeds = nullptr;
    c->t.drain = abrx::make_drain_tab(c->tt.sd, max_buffer + L);     // as abr_env_create does
    c->n_rates = n_rates;
    for (int i = 0; i < n_rates; i++) c->ladder[i] = ladder[i];
    return c;
}
void lj_destroy(void *h) { delete (Ctx *)h; }

// One episode.  Per-step outputs are taken AT each call site (before action s is applied):
// rec[s*8 + ..] = global_time, rebuffer_time, start_up_time, play_time, buffer_level,
//                 last_bw, sumk (as double), flags(su|be<<1|bf<<2)
// fin[0..5] = global_time, rebuffer_time, start_up_time, play_time, buffer_level, sumk; fin_i[0]=n_play
// returns 0, or -2 on timeout
// sched != NULL: a speed schedule of `rows` values for this lane (played chunk p plays at
// sched[min(p, rows - 1)]); fin_i[1] then returns play_id
int lj_episode(void *h, const double *trace, int32_t tlen, int32_t offset, const int32_t *actions,
               double *rec, double *bw_out, double *fin, int32_t *fin_i, double lane_speed,
               const double *sched, int32_t rows) {
    Ctx *c = (Ctx *)h;
    abrx::Tables t = c->t;
    abrx::LaneJ s;
    s.cur.trace = trace; s.cur.tlen = tlen;
    s.sd = t.sd;
    if (lane_speed > 0.0) { t.per_lane_speed = true; s.sd = lane_speed * 0.01; }   // :182 product
    if (lane_speed > 0.0 || sched) t.drain.n = 0;                                  // as make_tables does with per-lane speeds
    if (sched) { t.per_lane_speed = true; t.speed_rows = rows; t.speed_stride = 1; t.speeds = sched; s.lane = 0; }
    abrx::lanej_init(s, t, offset);
    if (!abrx::lanej_wait_call(s, t)) return -2;
    double last_bw = 0.0;
    for (int step = 0; step < t.V; step++) {
        double *r = rec + (size_t)step * 8;
        r[0] = t.G[s.k]; r[1] = t.G[s.n_rb]; r[2] = t.G[s.n_su];
        r[3] = t.per_lane_speed ? s.pt : c->tt.GP[s.n_play];
        r[4] = s.buf; r[5] = last_bw; r[6] = (double)s.sumk;
        r[7] = (double)((s.su ? 1 : 0) | (s.be ? 2 : 0) | (s.bf ? 4 : 0));
        int a = actions[step];
        // the step in its two halves, as the role-split kernels run it, with the download side's call-site prediction
        const double buf0 = s.buf; const bool su0 = s.su, be0 = s.be; const int32_t k0 = s.k;
        const abrx::StepStart st = abrx::lanej_begin_step(s.cur, t, s.k, s.chunk_id);
        const abrx::Download dd = abrx::lanej_download(s.cur, t, st, s.k, c->ladder[a] * t.L);
        abrx::StepResult sr = abrx::lanej_after_download(s, t, dd, st.avail_next, a);
        if (sr.timeout) return -2;
        if (!sr.ended && dd.hit) {
            const int32_t spec = k0 + dd.n_dl > st.avail_next ? k0 + dd.n_dl : st.avail_next;
            const bool gated = s.k != spec;
            g_pred[0]++; g_pred[1] += gated ? 1 : 0;
            int32_t kn = -1;
            const bool poss = abrx::lanej_gate_possible(buf0, su0, be0, dd.n_dl, t);
            if (gated && !poss && !t.per_lane_speed && !su0 && !be0) return -8;   // the cheap test must not miss a gated step of a playing lane
            if (poss && abrx::lanej_predict_next_call(buf0, k0, dd.n_dl, st.avail_next, t, kn)) {
                g_pred[3]++;
                if (gated) g_pred[2]++;
                if (kn != s.k) { g_pred[4]++; return -7; }
            }
        }
        last_bw = sr.bw;
        bw_out[step] = sr.bw;
        if (sr.ended != (step == t.V - 1)) return -5;
    }
    fin[0] = t.G[s.k]; fin[1] = t.G[s.n_rb]; fin[2] = t.G[s.n_su];
    fin[3] = t.per_lane_speed ? s.pt : c->tt.GP[s.n_play];
    fin[4] = s.buf; fin[5] = (double)s.sumk;
    fin_i[0] = s.n_play;
    if (sched) { fin_i[1] = s.play_id; fin[5] = s.pt_sum; }
    return 0;
}

int64_t lj_batch(void *h, const double *traces, const int64_t *trace_off, const int32_t *trace_len,
                 const int32_t *trace_id, const int32_t *offset, const int32_t *actions,
                 int32_t n_lanes, double *rec, double *bw_out, double *fin, int32_t *fin_i,
                 const double *speeds /* nullable: per-lane play speeds */,
                 const double *sched /* nullable: [n_lanes][rows] speed schedules */, int32_t rows) {
    Ctx *c = (Ctx *)h;
    const int V = c->t.V;
    for (int32_t i = 0; i < n_lanes; i++) {
        int tid = trace_id[i];
        int rc = lj_episode(h, traces + trace_off[tid], trace_len[tid], offset[i],
                            actions + (size_t)i * V, rec + (size_t)i * V * 8,
                            bw_out + (size_t)i * V, fin + (size_t)i * 6, fin_i + (size_t)i * 2,
                            speeds ? speeds[i] : 0.0, sched ? sched + (size_t)i * rows : nullptr, rows);
        if (rc) return -(1000 + (int64_t)i * 10 - rc);
    }
    return 0;
}
}
