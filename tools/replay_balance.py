"""Prices load balancing of ONE launch of the three-wave kernel from the workgroup lifetimes a launch
really had (tools/gpu_placement.py with ABR_PLACEMENT_DUMP=file: per workgroup its XCD, begin and lifetime on the constant
100 MHz clock, no copy kernel in front of the launch).

The launch lasts as long as its slowest workgroup.  A lane group's decisions are serial, so "stealing" can only be a MIGRATION: a
workgroup that has finished its own F decisions takes over the REMAINING decisions of the group that has the most left, state
handed over through the workspace at a decision-block boundary (blocks of B decisions), at the cost of one pipeline drain + fill
and a state round trip (hand_us).  It pays only as far as the new host is faster than the old one: the model gives every
workgroup the speed of its XCD (that XCD's mean lifetime; the data-dependent part of a group's lifetime travels with the group)
and plays the launch as an event simulation: whenever a workgroup is free, the group with the latest projected finish moves
to it if that makes it finish earlier.
    python tools/replay_balance.py dump.json [dump2.json ...]"""
import json
import sys

import numpy as np


def simulate(d, B, hand_us):
    x = np.array(d["xcc"]); life = np.array(d["wall_us"], float); F = d["decisions"]
    nwg = len(life)
    xs = sorted(set(x.tolist()))
    xmean = {q: life[x == q].mean() for q in xs}
    gmean = life.mean()
    speed = np.array([gmean / xmean[q] for q in x])            # > 1: a fast XCD
    work = life * speed                                         # the group's own (data) cost at average speed, in us
    # state: group g runs on host h[g] (initially itself); done fraction advances at speed[h]
    fin = life.copy()                                           # projected finish of group g on its current host
    host_free = life.copy()                                     # when each workgroup becomes free (finishes what it hosts)
    host = np.arange(nwg)
    moved = 0
    order = np.argsort(host_free)
    free_at = sorted([(host_free[w], w) for w in range(nwg)])
    # greedy event loop: take the earliest-free workgroup, offer it the group with the latest projected finish
    import heapq
    heap = list(free_at)
    heapq.heapify(heap)
    while heap:
        t, w = heapq.heappop(heap)
        g = int(np.argmax(fin))
        if fin[g] <= t + 1e-9:
            break
        h = host[g]
        # the group can be picked up at its next block boundary after t
        per_dec = (work[g] / speed[h]) / F
        start_g = fin[g] - work[g] / speed[h] if h == g else None
        # decisions done by time t on the current host (the group has run continuously on h since it got there)
        left_time = fin[g] - t
        left_dec = left_time / per_dec
        left_dec_b = np.floor(left_dec / B) * B                # hand over at a block boundary: the current block finishes on h
        if left_dec_b < B:
            continue
        t_hand = fin[g] - left_dec_b * per_dec
        new_fin = t_hand + hand_us + left_dec_b * (work[g] / speed[w]) / F
        if new_fin < fin[g] - 0.5:
            fin[g] = new_fin; host[g] = w; moved += 1
            heapq.heappush(heap, (new_fin, w))                  # w is busy until then; the old host is free from t_hand
            heapq.heappush(heap, (t_hand, h))
    return life.max(), fin.max(), moved


for path in sys.argv[1:]:
    d = json.load(open(path))
    life = np.array(d["wall_us"]); x = np.array(d["xcc"])
    print(f"{path}: {len(life)} workgroups x {d['decisions']} decisions, lifetimes mean {life.mean():.1f} us  p95 {np.percentile(life, 95):.1f}  "
          f"max {life.max():.1f}  (max / mean {life.max() / life.mean():.3f});  by XCD: "
          + "  ".join(f"{q}: {life[x == q].mean():.1f}" for q in sorted(set(x.tolist()))))
    print(f"  a launch that ended at the MEAN workgroup would be {100 * (life.max() / life.mean() - 1):.1f} % shorter: the ceiling of any balancing")
    for hand in (2.0, 5.0, 10.0):
        for B in (4, 8, 12, 16):
            t0, t1, mv = simulate(d, B, hand)
            print(f"  blocks of {B:2d} decisions, hand-over {hand:4.1f} us: launch {t0:.1f} -> {t1:.1f} us ({100 * (t0 / t1 - 1):+.1f} %), {mv} migrations")
