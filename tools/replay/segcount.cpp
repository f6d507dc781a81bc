// Host-side replay of the bench workload through the product's lane functions (abr_lane_jump.h),
// counting chain segments per decision: STOP_GE = download trips, STOP_LE / STOP_LT = drain trips.
// Used by tools/replay_model.py to price kernel schedules before they are written.
#include <stdint.h>
static thread_local int g_seg[3];
// per step of the last episode replayed: 1 = the next call site was NOT max(completion + 1, avail_tick[chunk + 1]),
// i.e. buffer_full gated the next download (Simulator.py:144) and a speculating download wave repeats it
static thread_local int g_gated[4096];
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
