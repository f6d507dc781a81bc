// Host-side replay of the bench workload through the product's lane functions (abr_lane_jump.h),
// counting chain segments per decision: STOP_GE = download trips, STOP_LE / STOP_LT = drain trips.
// Used by tools/replay_model.py to price kernel schedules before they are written.
#include <stdint.h>
static thread_local int g_seg[3];
// per step of the last episode replayed: 1 = the next call site was NOT max(completion + 1, avail_tick[chunk + 1]),
// i.e. buffer_full gated the next download (Simulator.py:144) and a speculating download wave repeats it
static thread_local int g_gated[4096];
// per step: download segments right after the prologue that spent a whole interval inside the binade ('B' runs at the start)
static thread_local int g_leadb[4096];
#define ABR_SEGMENT_HOOK(STOP) (g_seg[STOP]++)
// how each DOWNLOAD segment ended, in order: 'B' = it spent its whole budget (the rest of a trace interval) without
// reaching the target and without leaving the binade -- a segment a multi-interval jump could absorb; 'X' = anything else
static thread_local char g_end[256];
static thread_local int g_nend;
#define ABR_SEGMENT_END_HOOK(STOP, a, n, hit, inside) \
    do { if ((STOP) == 0 && g_nend < 255) g_end[g_nend++] = ((a) == (n) && !(hit) && (inside)) ? 'B' : 'X'; } while (0)
// the player's drains of a step: segments of those that ran dry / did not, ticks of the plain tail
static thread_local int g_dry, g_tail_ticks;
#define ABR_DRAIN_HOOK(dry, tail) do { g_dry |= (dry) ? 1 : 0; g_tail_ticks += (tail); } while (0)
// the drains of a step through the per-binade cascade (round 6): per call, the first and last binade in which the lane has a
// step (as stage indices from the top binade) and the plain ticks below the cascade; up to 4 calls per step
static thread_local int g_cas_n;
static thread_local int g_cas[4][4];
static inline int cas_expo(double v) { uint64_t b; __builtin_memcpy(&b, &v, 8); return (int)((b >> 52) & 0x7ff); }
#define ABR_CASCADE_HOOK(e_hi, n, x0, x1, a_st, a_plain, m) \
    do { if (g_cas_n < 4) { int f_ = (e_hi) - cas_expo(x0); if (f_ < 0) f_ = 0; int l_ = (e_hi) - cas_expo(x1); \
         if ((a_plain) > 0 || (x1) <= 0.0 || l_ > (n) - 1) l_ = (n) - 1; if ((a_st) == 0 && (a_plain) == 0) { f_ = 0; l_ = -1; } \
         g_cas[g_cas_n][0] = f_; g_cas[g_cas_n][1] = l_; g_cas[g_cas_n][2] = (a_plain); g_cas[g_cas_n][3] = (m); g_cas_n++; } } while (0)
#include "abr_lane_jump.h"
#include "abr_tick_tables.h"

extern "C" int seg_episode(double interval, double L, int32_t V, double max_buffer, double start_up,
                           int32_t max_ticks, const double *ladder, const double *trace, int32_t tlen,
                           int32_t offset, const int32_t *actions, int32_t *ge_out, int32_t *le_out,
                           int32_t *ndl_out, int32_t *merged_out, double *est_out, int32_t *adv_out) {
    static thread_local abrx::TickTables tt;
    static thread_local bool have = false;
    if (!have) {
        tt = abrx::build_tick_tables(interval, L, 1.0, V, max_ticks, (int32_t)(max_ticks * 0.01 / interval + 4.0));
        have = true;
    }
    abrx::Tables t;
    t.G = tt.G.data(); t.interval_tick = tt.interval_tick.data(); t.avail_tick = tt.avail_tick.data();
    t.L = L; t.sd = tt.sd; t.max_buffer = max_buffer; t.start_up_length = start_up; t.V = V;
    t.max_ticks = max_ticks; t.per_lane_speed = false; t.speed_rows = 0; t.speed_stride = 0; t.speeds = nullptr;
    abrx::LaneJ s;
    s.cur.trace = trace; s.cur.tlen = tlen; s.sd = t.sd;
    abrx::lanej_init(s, t, offset);
    if (!abrx::lanej_wait_call(s, t)) return -2;
    for (int step = 0; step < V; step++) {
        g_seg[0] = g_seg[1] = g_seg[2] = 0;
        g_nend = 0;
        const int32_t k0 = s.k;
        {   // what a dealing kernel could know at the call site: the target and the look-ahead's bandwidths
            abrx::Cursor cc = s.cur;
            const abrx::StepStart st0 = abrx::lanej_begin_step(cc, t, s.k, s.chunk_id);
            adv_out[step] = cc.j - s.cur.j;        // intervals the cursor was behind at this call site
            const double tgt = ladder[actions[step]] * L;
            est_out[step * 2 + 0] = tgt / st0.c;                                           // ticks at the current interval's rate
            est_out[step * 2 + 1] = tgt / (0.5 * (st0.c + st0.bw_next * abrx::kTickDt));   // ... at the mean of two intervals
        }
        const abrx::StepStart st = abrx::lanej_begin_step(s.cur, t, s.k, s.chunk_id);
        const abrx::Download dd = abrx::lanej_download(s.cur, t, st, s.k, ladder[actions[step]] * L);
        const int32_t k_spec = (k0 + dd.n_dl > st.avail_next) ? k0 + dd.n_dl : st.avail_next;
        abrx::StepResult sr = abrx::lanej_after_download(s, t, dd, st.avail_next, actions[step]);
        if (sr.timeout) return -2;
        if (step < 4096) g_gated[step] = (!sr.ended && s.k != k_spec) ? 1 : 0;
        ge_out[step] = g_seg[0]; le_out[step] = g_seg[1] + g_seg[2];
        ndl_out[step] = s.k - k0;
        { int lb = 0; while (lb + 1 < g_nend && g_end[lb] == 'B') lb++; if (step < 4096) g_leadb[step] = lb; }
        // trips of a download loop whose trip is [absorb up to M whole 'B' intervals] + [one segment], M = 1, 2, 3, 255
        static const int Ms[4] = {1, 2, 3, 255};
        for (int q = 0; q < 4; q++) {
            int trips = 0, run = 0;
            for (int i = 0; i < g_nend; i++) {
                if (g_end[i] == 'B' && run < Ms[q] && i + 1 < g_nend) { run++; continue; }   // absorbed into the next trip
                trips++; run = 0;
            }
            merged_out[step * 4 + q] = trips;
        }
    }
    return 0;
}

extern "C" void seg_leadb(int32_t *out, int32_t V) {
    for (int i = 0; i < V && i < 4096; i++) out[i] = g_leadb[i];
}
extern "C" void seg_gated(int32_t *out, int32_t V) {
    for (int i = 0; i < V && i < 4096; i++) out[i] = g_gated[i];
}

// the player side of each decision of an episode: drain segments, whether a drain ran the buffer dry, ticks of the plain tail
extern "C" int seg_episode_player(double interval, double L, int32_t V, double max_buffer, double start_up, int32_t max_ticks,
                                  const double *ladder, const double *trace, int32_t tlen, int32_t offset,
                                  const int32_t *actions, int32_t *le_out, int32_t *dry_out, int32_t *tail_out) {
    static thread_local abrx::TickTables tt;
    static thread_local bool have = false;
    if (!have) {
        tt = abrx::build_tick_tables(interval, L, 1.0, V, max_ticks, (int32_t)(max_ticks * 0.01 / interval + 4.0));
        have = true;
    }
    abrx::Tables t;
    t.G = tt.G.data(); t.interval_tick = tt.interval_tick.data(); t.avail_tick = tt.avail_tick.data();
    t.L = L; t.sd = tt.sd; t.max_buffer = max_buffer; t.start_up_length = start_up; t.V = V;
    t.max_ticks = max_ticks; t.per_lane_speed = false; t.speed_rows = 0; t.speed_stride = 0; t.speeds = nullptr;
    abrx::LaneJ s;
    s.cur.trace = trace; s.cur.tlen = tlen; s.sd = t.sd;
    abrx::lanej_init(s, t, offset);
    if (!abrx::lanej_wait_call(s, t)) return -2;
    for (int step = 0; step < V; step++) {
        const abrx::StepStart st = abrx::lanej_begin_step(s.cur, t, s.k, s.chunk_id);
        const abrx::Download dd = abrx::lanej_download(s.cur, t, st, s.k, ladder[actions[step]] * L);
        g_seg[1] = g_seg[2] = 0; g_dry = 0; g_tail_ticks = 0;
        abrx::StepResult sr = abrx::lanej_after_download(s, t, dd, st.avail_next, actions[step]);
        if (sr.timeout) return -2;
        le_out[step] = g_seg[1] + g_seg[2]; dry_out[step] = g_dry; tail_out[step] = g_tail_ticks;
    }
    return 0;
}

// per decision of an episode: [0] buffer_full gated the next download (the speculation "max(completion + 1, avail)" is wrong),
// [1] the download wave's exact prediction (lanej_gate_possible + lanej_predict_next_call, as role_d_validate applies them)
// covered it, [2] ticks of the decision, [3] download ticks
extern "C" int seg_episode_gating(double interval, double L, int32_t V, double max_buffer, double start_up, int32_t max_ticks,
                                  const double *ladder, const double *trace, int32_t tlen, int32_t offset,
                                  const int32_t *actions, int32_t *out4) {
    static thread_local abrx::TickTables tt;
    static thread_local bool have = false;
    if (!have) {
        tt = abrx::build_tick_tables(interval, L, 1.0, V, max_ticks, (int32_t)(max_ticks * 0.01 / interval + 4.0));
        have = true;
    }
    abrx::Tables t;
    t.G = tt.G.data(); t.interval_tick = tt.interval_tick.data(); t.avail_tick = tt.avail_tick.data();
    t.L = L; t.sd = tt.sd; t.max_buffer = max_buffer; t.start_up_length = start_up; t.V = V;
    t.max_ticks = max_ticks; t.per_lane_speed = false; t.speed_rows = 0; t.speed_stride = 0; t.speeds = nullptr;
    t.drain = abrx::make_drain_tab(tt.sd, max_buffer + L);
    abrx::LaneJ s;
    s.cur.trace = trace; s.cur.tlen = tlen; s.sd = t.sd;
    abrx::lanej_init(s, t, offset);
    if (!abrx::lanej_wait_call(s, t)) return -2;
    for (int step = 0; step < V; step++) {
        const double buf0 = s.buf; const bool su0 = s.su, be0 = s.be; const int32_t k0 = s.k;
        const abrx::StepStart st = abrx::lanej_begin_step(s.cur, t, s.k, s.chunk_id);
        const abrx::Download dd = abrx::lanej_download(s.cur, t, st, s.k, ladder[actions[step]] * L);
        abrx::StepResult sr = abrx::lanej_after_download(s, t, dd, st.avail_next, actions[step]);
        if (sr.timeout) return -2;
        const int32_t spec = k0 + dd.n_dl > st.avail_next ? k0 + dd.n_dl : st.avail_next;
        const bool gated = !sr.ended && s.k != spec;
        int32_t kn = -1;
        const bool cov = dd.hit && !sr.ended && abrx::lanej_gate_possible(buf0, su0, be0, dd.n_dl, t) &&
                         abrx::lanej_predict_next_call(buf0, k0, dd.n_dl, st.avail_next, t, kn) && kn == s.k;
        out4[step * 4 + 0] = gated; out4[step * 4 + 1] = cov; out4[step * 4 + 2] = s.k - k0; out4[step * 4 + 3] = dd.n_dl;
    }
    return 0;
}

// the player side of each decision through the cascade: per step up to 2 drain calls x (first stage, last stage, plain ticks, budget)
extern "C" int seg_episode_cascade(double interval, double L, int32_t V, double max_buffer, double start_up, int32_t max_ticks,
                                   const double *ladder, const double *trace, int32_t tlen, int32_t offset,
                                   const int32_t *actions, int32_t *out /* [V][2][4] */) {
    static thread_local abrx::TickTables tt;
    static thread_local bool have = false;
    if (!have) {
        tt = abrx::build_tick_tables(interval, L, 1.0, V, max_ticks, (int32_t)(max_ticks * 0.01 / interval + 4.0));
        have = true;
    }
    abrx::Tables t;
    t.G = tt.G.data(); t.interval_tick = tt.interval_tick.data(); t.avail_tick = tt.avail_tick.data();
    t.L = L; t.sd = tt.sd; t.max_buffer = max_buffer; t.start_up_length = start_up; t.V = V;
    t.max_ticks = max_ticks; t.per_lane_speed = false; t.speed_rows = 0; t.speed_stride = 0; t.speeds = nullptr;
    t.drain = abrx::make_drain_tab(tt.sd, max_buffer + L);
    abrx::LaneJ s;
    s.cur.trace = trace; s.cur.tlen = tlen; s.sd = t.sd;
    abrx::lanej_init(s, t, offset);
    if (!abrx::lanej_wait_call(s, t)) return -2;
    for (int step = 0; step < V; step++) {
        const abrx::StepStart st = abrx::lanej_begin_step(s.cur, t, s.k, s.chunk_id);
        const abrx::Download dd = abrx::lanej_download(s.cur, t, st, s.k, ladder[actions[step]] * L);
        g_cas_n = 0;
        for (int c = 0; c < 2; c++) { g_cas[c][0] = 0; g_cas[c][1] = -1; g_cas[c][2] = 0; g_cas[c][3] = 0; }
        abrx::StepResult sr = abrx::lanej_after_download(s, t, dd, st.avail_next, actions[step]);
        if (sr.timeout) return -2;
        for (int c = 0; c < 2; c++) for (int q = 0; q < 4; q++) out[(step * 2 + c) * 4 + q] = g_cas[c][q];
    }
    return 0;
}
