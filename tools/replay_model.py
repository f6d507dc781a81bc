"""Prices K1 schedules on the host before a kernel is written: replays the bench workload through the
product's own lane functions (tools/replay/segcount.cpp counts chain segments per decision) and runs
wave-level schedule models over the per-lane counts:
  split      one decision per iteration, the wave pays the slowest of its 64 lanes (the shipped kernel)
  capped(T)  at most T download trips per iteration; a lane that is not done carries its download over
  flat       no rendezvous at all: mean trips (the bound of any schedule)
    python tools/replay_model.py [lanes]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench as B  # noqa: E402
from oracle.oracle import philox_action  # noqa: E402  (tools may use the oracle's numpy twin of the policy)

so = os.path.join(R, "tools", "replay", "libsegcount.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I",
                       os.path.join(R, "abrsimulator_amd", "csrc"), os.path.join(R, "tools", "replay", "segcount.cpp"),
                       "-o", so])
lib = C.CDLL(so)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
traces = B.synth_traces(False)
tid, off = B.lane_assignment(0, N, traces)
V = B.V
ge = np.zeros((N, V), np.int32); le = np.zeros((N, V), np.int32); nd = np.zeros((N, V), np.int32)
mg = np.zeros((N, V, 4), np.int32)
est = np.zeros((N, V, 2), np.float64)
advs = np.zeros((N, V), np.int32)
lad = (C.c_double * 16)(*B.LADDER)
acts = np.stack([philox_action(1, np.arange(N), s, 0, len(B.LADDER)) for s in range(V)], 1).astype(np.int32)
for i in range(N):
    t = np.ascontiguousarray(traces[tid[i]])
    rc = lib.seg_episode(C.c_double(B.INTERVAL), C.c_double(B.L), V, C.c_double(B.MAX_BUFFER), C.c_double(B.START_UP),
                         32 * V * 400, lad, t.ctypes.data_as(C.c_void_p), len(t), int(off[i]),
                         acts[i].ctypes.data_as(C.c_void_p), ge[i].ctypes.data_as(C.c_void_p),
                         le[i].ctypes.data_as(C.c_void_p), nd[i].ctypes.data_as(C.c_void_p),
                         mg[i].ctypes.data_as(C.c_void_p), est[i].ctypes.data_as(C.c_void_p), advs[i].ctypes.data_as(C.c_void_p))
    assert rc == 0
print(f"{N} lanes x {V} decisions: download trips mean {ge.mean():.2f} p50 {np.median(ge):.0f} p90 {np.percentile(ge, 90):.0f} "
      f"p99 {np.percentile(ge, 99):.0f} max {ge.max()};  drain segments mean {le.mean():.2f} p90 {np.percentile(le, 90):.0f} max {le.max()}")
for a in range(6):
    print(f"  action {a}: download trips mean {ge[acts == a].mean():.2f}  drain segments {le[acts == a].mean():.2f}")
W = N // 64
g = ge.reshape(W, 64, V)
l = le.reshape(W, 64, V)
print(f"split: D trips per iteration (max over 64 lanes) {g.max(1).mean():.2f};  P drain segments per iteration "
      f"{l.max(1).mean():.2f};  per-lane means {ge.mean():.2f} / {le.mean():.2f}")


def capped(T, start_cost, trip_cost):
    """iterations and D cost per decision with at most T trips per iteration"""
    iters = tot_trips = 0
    lanes_done = 0
    for w in range(W):
        rem = g[w, :, 0].astype(np.int64).copy()
        step = np.zeros(64, np.int64)
        while (step < V).any():
            act = step < V
            run = int(min(T, rem[act].max()))
            tot_trips += run
            iters += 1
            rem = rem - run
            fin = act & (rem <= 0)
            lanes_done += int(fin.sum())
            step[fin] += 1
            nxt = fin & (step < V)
            rem[nxt] = g[w, nxt, step[nxt]]
            rem[fin & ~nxt] = 0
    it = iters / W
    return it / V, tot_trips / W / V, (tot_trips * trip_cost + iters * start_cost) / W / V


print("capped(T): iterations/decision, D trips/decision, D cost/decision (trip = 118 + 25, start/validate = 400 instr)")
base = None
for T in (4, 6, 8, 10, 12, 14, 16, 64):
    it, tr, cost = capped(T, 400, 143)
    if T == 64:
        base = cost
    print(f"  T={T:3d}: {it:.3f} iterations  {tr:6.2f} trips  D cost {cost:7.0f}")
print(f"  (T=64 is the shipped schedule: {base:.0f})")


# ---- round 4: cost-class dealing of downloads across a bigger workgroup, priced before it is built ----
# A workgroup of G lane groups (64 * G lanes, G download waves).  Every step the workgroup's 64 * G pending downloads are
# dealt to the G download waves by predicted cost (here: an ORACLE predictor -- the true trip count -- i.e. the best any
# predictor could do), each wave then pays the slowest of ITS 64 downloads.  Two ways to close a step:
#   barrier   one workgroup barrier per step (the role-split protocol as it is): the step lasts as long as the slowest WAVE
#   free      no rendezvous between the download waves at all (they run ahead on their own, which needs rings towards the
#             player / service side as in the asynchronous pipeline): a wave's time is the sum of its own batches
# D instructions per decision = trips * 95 + 400 (start / publish / validate) + deal (moving a 15-word job through LDS and
# ranking it: ~150 per batch, amortised over its 64 lanes it is per wave-step).
def dealing(G, deal_cost=150, trip=95, start=400):
    Wg = W // G
    crit_barrier = instr = 0.0
    for wg in range(Wg):
        t = g[wg * G:(wg + 1) * G].reshape(G * 64, V)          # trips of the workgroup's lanes, per step
        srt = -np.sort(-t, axis=0)                             # per step: costs in descending order
        per_wave = srt.reshape(G, 64, V).max(1)                # [G, V]: each dealt wave's slowest download
        crit_barrier += per_wave.max(0).sum()                  # barrier per step: the slowest wave
        instr += (per_wave * trip + start + deal_cost).sum()
    n = Wg * V
    return crit_barrier / n, instr / (n * G)


print("cost-class dealing (oracle predictor) against the shipped schedule (G = 1: 64-lane workgroups, no dealing):")
print("   G  lanes/WG  critical-path trips per step (barrier per step)  D instructions per decision per wave")
for G in (1, 2, 4, 6, 8, 16):
    if W % G:
        continue
    cb, ins = dealing(G, deal_cost=0 if G == 1 else 150)
    print(f"  {G:2d}  {64 * G:7d}  {cb:10.2f}                                          {ins:8.0f}")
print("   -> with a workgroup barrier per step the step lasts as long as the workgroup's slowest download wave, which\n"
      "      holds the slowest lanes of ALL its groups: the critical path grows with G while the instruction count\n"
      "      falls -- and a lone wave's instruction stream, not the SIMD's issue capacity, is what bounds the step.")


# ---- round 4: multi-interval jumps, priced.  Inside one binade the steady increment of EVERY trace interval is a multiple of
# the same ulp, so x + sum(n_i * d_i) over whole intervals is exact as long as it stays inside the binade and below the target:
# a trip could absorb up to M whole intervals (a handful of instructions each: d_i, the tie test, one multiply-add, one compare)
# before its one real segment.  tools/replay/segcount.cpp marks the segments that spent a whole interval inside the binade. ----
print("multi-interval jumps: download trips per step, mean over lanes / mean of the wave maximum (64 lanes):")
print(f"  shipped (one interval or one binade per trip)   {ge.mean():5.2f} / {g.max(1).mean():5.2f}")
for q, M in enumerate((1, 2, 3, "any")):
    m = mg[:, :, q]
    print(f"  up to {M!s:>3} whole intervals absorbed per trip      {m.mean():5.2f} / {m.reshape(W, 64, V).max(1).mean():5.2f}")


# ---- dealing with a predictor a kernel can afford (the oracle rows above sort by the TRUE trip count) ----
def dealing_pred(G, key, deal_cost=150, trip=95, start=400):
    Wg = W // G
    crit = instr = 0.0
    for wg in range(Wg):
        t = g[wg * G:(wg + 1) * G].reshape(G * 64, V)
        kk = key[wg * G * 64:(wg + 1) * G * 64]                 # [G*64, V]
        order = np.argsort(-kk, axis=0, kind="stable")
        srt = np.take_along_axis(t, order, axis=0)
        per_wave = srt.reshape(G, 64, V).max(1)
        crit += per_wave.max(0).sum()
        instr += (per_wave * trip + start + deal_cost).sum()
    n = Wg * V
    return crit / n, instr / (n * G)


print("dealing with affordable predictors (G = 4, 256-lane workgroups): D instructions per decision per wave")
for name, key in (("oracle (true trips)", g.reshape(N, V).astype(np.float64)),
                  ("target / current rate, 16 log classes", np.floor(4 * np.log2(np.maximum(est[:, :, 0], 1.0)))),
                  ("target / mean of two intervals, 16 log classes", np.floor(4 * np.log2(np.maximum(est[:, :, 1], 1.0)))),
                  ("target / current rate, exact sort", est[:, :, 0]),
                  ("target only (the action)", np.repeat(acts[:, :, None], 1, 2)[:, :, 0].astype(np.float64))):
    cb, ins = dealing_pred(4, key)
    print(f"  {name:48s} {ins:7.0f}   (critical path {cb:5.2f} trips)")


# ---- narrower lane groups: what a wave that serves fewer lanes would pay (the maximum over its lanes) ----
print("lane-group width: download trips / drain segments per step (mean of the group maximum)")
for wdt in (64, 32, 16, 8):
    print(f"  {wdt:2d} lanes   {ge.reshape(N // wdt, wdt, V).max(1).mean():5.2f} / {le.reshape(N // wdt, wdt, V).max(1).mean():5.2f}")


# ---- how far the trace cursor lags at a call site (the look-ahead burst of lanej_begin_step covers kCatch intervals) ----
h = np.bincount(np.minimum(advs.ravel(), 8), minlength=9)
print("intervals the cursor is behind at a call site (share of call sites):",
      "  ".join(f"{i}{'+' if i == 8 else ''}: {100.0 * c / advs.size:.1f} %" for i, c in enumerate(h)))
wmax = advs.reshape(W, 64, V).max(1)
print("  ... maximum over the 64 lanes of a wave:", "  ".join(f"{i}{'+' if i == 8 else ''}: {100.0 * c / wmax.size:.1f} %" for i, c in enumerate(np.bincount(np.minimum(wmax.ravel(), 8), minlength=9))))


def lane_async(theta, trip, rest, cap=10 ** 9):
    """one thread per lane, NO step lockstep inside the wave: a lane that finishes its download waits until `theta` lanes of its wave
    wait (or none can run), then the wave runs the rest of the step (player, service, next begin: `rest` instructions, masked to the
    waiting lanes) once; otherwise it runs one download trip (`trip` instructions) for the lanes that have one.  `cap`: a lane may
    not finish a decision more than `cap` decisions ahead of the slowest lane of its wave.  Returns vector instructions per
    wave-decision, the spread of step indices inside a wave when the first lane ends its episode, and the number of pieces an
    output row of the wave (64 lanes of one decision) is written in."""
    tot = 0
    spread = []
    pieces = 0
    for w in range(W):
        T = g[w]                                 # [64, V]
        s = np.zeros(64, np.int64)
        r = T[:, 0].astype(np.int64).copy()
        cost = 0
        seen = False
        while True:
            live = s < V
            if not live.any():
                break
            act = live & (r > 0)
            fin = live & (r == 0)
            wait = fin & (s - s[live].min() < cap)
            if not act.any() or wait.sum() >= theta:
                cost += rest
                pieces += len(np.unique(s[wait]))
                s[wait] += 1
                nxt = wait & (s < V)
                r[nxt] = T[nxt, s[nxt]]
                if not seen and (s >= V).any():
                    seen = True
                    spread.append(int(s.max() - s.min()))
            else:
                cost += trip
                r[act] -= 1
        tot += cost
    return tot / (W * V), float(np.mean(spread)), pieces / (W * V)


print("one thread per lane without step lockstep in the wave (instructions per wave-decision; trip 129, rest of the step 338):")
base_j = g.max(1).mean() * 129 + 338
print(f"  shipped (every lane of the wave at the same decision)      {base_j:6.0f}")
for th, cap in ((16, 99), (32, 99), (40, 99), (48, 99), (56, 99), (32, 2), (40, 2), (48, 2), (32, 4), (40, 4), (48, 4), (40, 8), (32, 1), (48, 1)):
    c, sp, pc = lane_async(th, 129, 338, cap)
    print(f"  rest of the step once {th:2d} lanes wait, at most {cap:2d} decisions ahead   {c:6.0f}   ({base_j / c:.2f}x; decisions apart inside a wave "
          f"at the first episode end: {sp:.1f}; an output row is written in {pc:.1f} pieces)")
