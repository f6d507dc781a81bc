"""K3 outside the two specialised ladders: abr_mpc_select at 65 536 lanes for ladders of B rates and horizons H
that run the generic code path (B not in {4, 6}), and for lanes whose horizon is clipped at the video's end.
    python tools/gpu_mpc_shapes.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import abrsimulator_amd as A  # noqa: E402

N, V, L = 65536, 48, 4.0
dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)


class P:
    pass


def run(B, H, clipped):
    ladder = list(np.sort(rng.uniform(0.3, 5.0, B)))
    mpd = A.MPD(V, L, 20.0, 8.0, [A.Chunk(ladder, [b * L for b in ladder])] * V)
    ci = P()
    lo, hi = (V - H + 1, V) if clipped else (0, V - H)
    ci.chunk_number = torch.from_numpy(rng.integers(lo, hi, N).astype(np.int32)).to(dev)
    ci.previous_bitrate = torch.from_numpy(rng.integers(0, B, N).astype(np.int32)).to(dev)
    ci.buffer_level = torch.from_numpy(rng.uniform(0, 20, N)).to(dev)
    ci.hist_n = torch.full((N,), 5.0, dtype=torch.float64, device=dev)
    ci.hist_sum_inv = torch.from_numpy(5.0 / rng.uniform(0.5, 5, N)).to(dev)
    pl = P()
    pl.get_mpd = lambda: mpd
    pl.get_qoe_metric = lambda: A.QOEMetric(4.3, 1.0, 0.0)
    pl.get_next_chunk_info = lambda: ci
    ctl = A.BatchedMPCController(pl, horizon=H, clip_horizon=True, device=dev)
    for _ in range(60):                       # (the first shape of a process also ramps the clocks up)
        ctl.next_bitrate()
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        ctl.next_bitrate()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    combos = N * B ** H
    print(f"B={B} H={H} {'clipped horizon (last H-1 chunks)' if clipped else 'full horizon':34s} {dt * 1e6:9.1f} us per select  "
          f"{combos / dt:.3e} combos/s (on the {B}^{H} = {B ** H} basis)")


for B, H in ((6, 5), (4, 5), (5, 5), (3, 6), (8, 4), (7, 5), (2, 8)):
    run(B, H, False)
run(6, 5, True)
run(5, 5, True)
