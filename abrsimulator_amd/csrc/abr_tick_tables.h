// abr_tick_tables.h -- the universal tick tables, built once on the HOST in float64.
//
// global_time starts at 0.0 and only ever receives `+= 0.01` (Simulator.py:128,205),
// so it is the same float64 sequence G[k] in every lane; so are download_time,
// start_up_time, rebuffer_time (k additions of dt from 0) and, for a constant
// play_speed, play_time.  Every floor of global_time the reference takes
// (int(global_time / chunk_length), :143; int(global_time / interval), :158)
// therefore becomes a lookup in a table indexed by the integer tick, computed here
// with exactly the reference's float64 operations (drift included: after 300 ticks
// G is 2.99999999999998, not 3).
#ifndef ABR_TICK_TABLES_H
#define ABR_TICK_TABLES_H

#include <climits>
#include <cstdint>
#include <vector>

namespace abrx {

struct TickTables {
    std::vector<double> G, GP;            // n-fold sums of dt and of speed*dt
    std::vector<int32_t> interval_tick;   // first tick k with int(G[k]/interval) >= j, INT_MAX past the end
    std::vector<int32_t> avail_tick;      // first tick k with int(G[k]/L) - 1 >= c,    INT_MAX past the end
    int32_t play_ticks_per_chunk;         // first n with GP[n] >= L (:185), INT_MAX if none
    int32_t min_interval_ticks;           // shortest interval, in ticks (>= 0)
    double sd;                            // speed * dt as the reference forms it (:182)
};

inline TickTables build_tick_tables(double interval, double chunk_length, double speed,
                                    int32_t video_length, int32_t max_ticks, int32_t n_intervals) {
    const double dt = 0.01;               // Simulator.py:133
    TickTables t;
    const int32_t mt = max_ticks;
    t.sd = speed * dt;
    t.G.resize((size_t)mt + 2);
    t.GP.resize((size_t)mt + 2);
    double g = 0.0, gp = 0.0;             // global_time = 0.0 (:128); play_time = 0 (:115)
    for (int32_t n = 0; n < mt + 2; n++) {
        t.G[n] = g; t.GP[n] = gp;
        g += dt;                          // :205 (and :138,:140,:161)
        gp += t.sd;                       // :182-183
    }
    t.interval_tick.assign((size_t)n_intervals + 8, INT_MAX);   // slack: look-ahead reads up to j + 6
    {
        int64_t jcur = 0;
        for (int32_t k = 0; k <= mt && jcur < n_intervals + 8; k++) {
            int64_t idx = (int64_t)(t.G[k] / interval);                    // :158
            while (jcur <= idx && jcur < n_intervals + 8) t.interval_tick[jcur++] = k;
        }
    }
    t.min_interval_ticks = INT_MAX;
    for (size_t j = 0; j + 1 < t.interval_tick.size() && t.interval_tick[j + 1] != INT_MAX; j++) {
        int32_t len = t.interval_tick[j + 1] - t.interval_tick[j];
        if (len < t.min_interval_ticks) t.min_interval_ticks = len;
    }
    if (t.min_interval_ticks == INT_MAX) t.min_interval_ticks = mt;
    t.avail_tick.assign((size_t)video_length + 2, INT_MAX);
    {
        int64_t ccur = 0;
        for (int32_t k = 0; k <= mt && ccur < video_length; k++) {
            int64_t avail = (int64_t)(t.G[k] / chunk_length) - 1;          // :143
            while (ccur <= avail && ccur < video_length) t.avail_tick[ccur++] = k;
        }
    }
    t.play_ticks_per_chunk = INT_MAX;
    for (int32_t n = 1; n < mt + 2; n++)
        if (t.GP[n] >= chunk_length) { t.play_ticks_per_chunk = n; break; }   // :185
    return t;
}

}  // namespace abrx
#endif
