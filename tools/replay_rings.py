"""Prices the ring-coupled role pipeline on the host before it is written.

The shipped three-wave kernel closes every iteration with ONE workgroup barrier: iteration t lasts
max(D_t, P_(t-1), S_(t-2)).  The candidate keeps the waves as they are (one role per wave, the 64 lanes of a wave on
the same iteration) but couples them through K-deep LDS rings: a wave waits only when its input ring is empty or its
output ring is full.  This replays the bench workload through the product's lane functions (tools/replay/segcount.cpp:
download trips and drain segments per lane-decision, and which decisions a speculating download wave has to repeat
because buffer_full gated the call site), turns the per-wave maxima into cycles with the measured role stamps
(profiles/r04_role_stamps_split3.txt) and runs both schedules as event simulations per workgroup.

    python tools/replay_rings.py [lanes] [decisions per launch]"""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
F = int(sys.argv[2]) if len(sys.argv) > 2 else 48
traces = B.synth_traces(False)
tid, off = B.lane_assignment(0, N, traces)
V = B.V
ge = np.zeros((N, V), np.int32); le = np.zeros((N, V), np.int32); nd = np.zeros((N, V), np.int32)
mg = np.zeros((N, V, 4), np.int32); est = np.zeros((N, V, 2), np.float64); advs = np.zeros((N, V), np.int32)
gated = np.zeros((N, V), np.int32)
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
    lib.seg_gated(gated[i].ctypes.data_as(C.c_void_p), V)
W = N // 64
g = ge.reshape(W, 64, V).max(1).astype(np.float64)[:, :F]        # D: slowest lane's trips, per wave and step
l = le.reshape(W, 64, V).max(1).astype(np.float64)[:, :F]        # P: slowest lane's drain segments
gt = gated.reshape(W, 64, V)[:, :, :F]
print(f"{N} lanes, {F} decisions per launch: wave-max download trips {g.mean():.2f}, drain segments {l.mean():.2f}; "
      f"decisions whose next call site is gated by buffer_full: {100.0 * gated.mean():.3f} % of lane-decisions, "
      f"{100.0 * gt.max(1).mean():.2f} % of wave-decisions hold one")

# ---- cycles per role and iteration from the stamps of the shipped kernel (profiles/r04_role_stamps_split3.txt) ----
# D: 682 + 1414 + 370 + 543 fixed, 7810 in the loop at 11.81 trips;  P: 484 + 511 + 1045 + 390 fixed, 5984 + 2607 in the two
# drains at 5.76 segments;  S: 673 + 7718.  The barrier itself (arrival -> release with everybody there): what is left of
# the measured 12.5 k per iteration.
Dc = 3009.0 + 7810.0 / 11.81 * g
Pc = 2430.0 + (5984.0 + 2607.0) / 5.76 * l
Sc = np.full_like(Dc, 8391.0)
print(f"role work per iteration (model, cycles): D {Dc.mean():.0f}  P {Pc.mean():.0f}  S {Sc.mean():.0f}   (stamps: 10.8 k / 11.0 k / 8.4 k)")


def barrier_schedule(c_bar):
    """iteration t = max(D_t, P_(t-1), S_(t-2)) + c_bar; F + 2 iterations per launch (fill and drain)"""
    T = np.zeros(W)
    for t in range(F + 2):
        d = Dc[:, t] if t < F else 0.0
        p = Pc[:, t - 1] if 1 <= t <= F else 0.0
        s = Sc[:, t - 2] if 2 <= t <= F + 1 else 0.0
        T += np.maximum(np.maximum(d, p), s) + c_bar
    return T


def ring_schedule(K, c_ring, redo=True):
    """event simulation: D may be K records ahead of P, P K records ahead of S; every hand-off costs c_ring on both sides.
    A gated decision (redo): the download wave learns of it when P has processed the record, repeats that decision and
    has wasted what it issued in between (the lanes of a wave stay on the same iteration, so a repeat is one more
    iteration for the whole wave)."""
    T = np.zeros(W)
    extra = np.zeros(W)
    for w in range(W):
        seq = list(range(F))
        if redo:
            # iterations of the wave: a step with a gated lane is followed by the repeat of the NEXT step for that lane,
            # which costs the wave one more iteration of (about) mean cost; with run-ahead the wave is up to K records
            # further when it learns of it, but a repeat is still ONE extra iteration: the records in between were
            # issued anyway and are dropped lane-wise
            n_redo = int(gt[w].max(0).sum())
            extra[w] = n_redo
        d_done = np.zeros(F + 1); p_done = np.zeros(F + 1); s_done = np.zeros(F + 1)
        for t in range(F):
            free = p_done[t - K] if t - K >= 0 else 0.0             # slot t % K is free once P has consumed record t - K
            d_done[t] = max(d_done[t - 1] if t else 0.0, free) + Dc[w, t] + c_ring
            free2 = s_done[t - K] if t - K >= 0 else 0.0
            p_done[t] = max(p_done[t - 1] if t else 0.0, d_done[t], free2) + Pc[w, t] + c_ring
            s_done[t] = max(s_done[t - 1] if t else 0.0, p_done[t]) + Sc[w, t] + c_ring
        T[w] = s_done[F - 1] + extra[w] * (Dc[w].mean() + c_ring)
    return T


# the barrier's own cost (arrival of the last wave -> release), calibrated on the measured iteration: 12.5 k cycles as the
# mean over the F + 2 iterations of a 48-decision launch; 982 cycles at 8 192 lanes x 48 decisions, used for every F
c_bar = float(os.environ.get("ABR_C_BAR", "982"))
if F == 48 and "ABR_C_BAR" not in os.environ:
    lo, hi = 0.0, 3000.0
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        if barrier_schedule(mid).mean() / (F + 2) < 12500.0:
            lo = mid
        else:
            hi = mid
    c_bar = 0.5 * (lo + hi)
n_redo = gt.max(1).sum(1).mean()
Tb = barrier_schedule(c_bar) + n_redo * 12500.0
print(f"barrier schedule: c_bar {c_bar:.0f} cycles (calibrated on the measured 12.5 k cycles per iteration at fuse 48); repeats per wave "
      f"and launch {n_redo:.2f}")
print(f"  mean over workgroups {Tb.mean() / F:.0f} cycles per decision; slowest of {W} workgroups {Tb.max() / F:.0f}")
print("ring schedule: cycles per decision, mean over workgroups (gain over the barrier schedule); hand-off cost per side")
for c_ring in (150.0, 300.0, 500.0):
    row = []
    for K in (1, 2, 3, 4, 8):
        Tr = ring_schedule(K, c_ring)
        row.append(f"K={K}: {Tr.mean() / F:6.0f} ({100.0 * (Tb.mean() / Tr.mean() - 1.0):+5.1f} %)")
    print(f"  c_ring {c_ring:4.0f}:  " + "   ".join(row))
print("bound: max(mean D, mean P) =", f"{max(Dc.mean(), Pc.mean()):.0f}", "cycles per decision ->",
      f"{100.0 * (Tb.mean() / F / max(Dc.mean(), Pc.mean()) - 1.0):+.1f} % (no hand-off cost, infinite rings, no SIMD sharing)")
# SIMD sharing: the three waves of a SIMD share its vector issue (2 115 vector instructions per 64 lanes per decision x 4 cycles
# = 8.5 k cycles of issue per decision, r04 SQ counters).  A schedule cannot go below that, and the closer it gets the more
# the roles' dependent streams (measured ALONE-ish at 10.8 k / 11.0 k with the others partly idle at the barrier) slow
# each other down.
print("issue floor of a SIMD that holds one wave of each role: 2 115 x 4 = 8 460 cycles per decision")


# ---- the launch's tail (round 5): all 1 024 workgroups of a 65 536-lane launch are resident from the start, so the launch lasts
# as long as its SLOWEST workgroup.  Per workgroup: sum over the launch's decisions of the slowest lane's download trips. ----
tot = g.sum(1)
print(f"launch tail: download trips of a workgroup's slowest lane summed over {F} decisions: mean {tot.mean():.0f}  p95 {np.percentile(tot, 95):.0f}  "
      f"max {tot.max():.0f} ({100.0 * (tot.max() / tot.mean() - 1.0):+.0f} % over the mean);  largest single decision: {g.max():.0f} trips")
cost = (Dc).sum(1)
print(f"   in cycles of the download wave: mean {cost.mean():.0f}  max {cost.max():.0f}  ({100.0 * (cost.max() / cost.mean() - 1.0):+.0f} %)")
gl = ge.reshape(W, 64, V)[:, :, :F]


def capped_lifetime(T):
    """per workgroup: iterations' cost when a download is cut after T trips and carried over (the lane's other 63 neighbours go on
    with their next decisions; the workgroup is done when every lane has done its F decisions)"""
    out = np.zeros(W)
    for w in range(W):
        rem = gl[w, :, 0].astype(np.int64).copy()
        step = np.zeros(64, np.int64)
        c = 0.0
        while (step < F).any():
            act = step < F
            run = int(min(T, rem[act].max()))
            c += 3009.0 + 7810.0 / 11.81 * run
            rem = rem - run
            fin = act & (rem <= 0)
            step[fin] += 1
            nxt = fin & (step < F)
            rem[nxt] = gl[w, nxt, step[nxt]]
            rem[fin & ~nxt] = 0
        out[w] = c
    return out


print("   download-wave cycles per launch with downloads cut after T trips and carried over:  T: mean / max over workgroups")
for T in (8, 12, 16, 24, 32, 48, 1000):
    c = capped_lifetime(T)
    print(f"     T={T:4d}: mean {c.mean():8.0f}  max {c.max():8.0f}   (launch = max: {100.0 * (cost.max() / c.max() - 1.0):+5.1f} % against the uncut schedule)")
