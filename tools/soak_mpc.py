#!/usr/bin/env python3
"""One-off MPC soak on the GPU box: abr_mpc_select on N random player states (VBR tables,
6 rates, horizon 5) against the C oracle's literal per-combo brute force on the host cores:
action, flat arg-min index, J (float64 ==) and the mutated history (==) of EVERY lane.
usage: python tools/soak_mpc.py [n_lanes]"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import abrsimulator_amd as A  # noqa: E402
import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
B, H, V, L, mb = 6, 5, 48, 4.0, 20.0
rng = np.random.default_rng(7)
lad = np.array(bench.LADDER)
br = lad[None, :] * rng.uniform(0.8, 1.2, (V, B))
sz = br * L * rng.uniform(0.7, 1.3, (V, B))
chunk = rng.integers(0, V - H + 1, N).astype(np.int32)
prev = rng.integers(0, B, N).astype(np.int32)
buf = np.where(rng.random(N) < 0.2, 0.0, rng.uniform(0, mb, N))
hn = rng.integers(1, 40, N).astype(np.float64)
hs = hn / rng.uniform(0.3, 5.0, N)


class P:
    pass


ci = P()
ci.chunk_number = torch.from_numpy(chunk).cuda(); ci.previous_bitrate = torch.from_numpy(prev).cuda()
ci.buffer_level = torch.from_numpy(buf).cuda(); ci.hist_n = torch.from_numpy(hn.copy()).cuda()
ci.hist_sum_inv = torch.from_numpy(hs.copy()).cuda()
pl = P()
mpd = A.MPD(V, L, mb, 0.0, [A.Chunk(list(b), list(s)) for b, s in zip(br, sz)])
pl.get_mpd = lambda: mpd
pl.get_qoe_metric = lambda: A.QOEMetric(4.3, 1.0, 0.0)
pl.get_next_chunk_info = lambda: ci
ctl = A.BatchedMPCController(pl, horizon=H, clip_horizon=False)
t0 = time.perf_counter()
act = ctl.next_bitrate(want_details=True).cpu().numpy()
t_gpu = time.perf_counter() - t0
flat, J = ctl.last_flat.cpu().numpy(), ctl.last_J.cpu().numpy()
hn_g, hs_g = ci.hist_n.cpu().numpy(), ci.hist_sum_inv.cpu().numpy()

cfg = O.mpc_cfg(B, H, V, L, mb, 1.0, 4.3, 0.0)
cores, _ = bench.host_cores()


def run(idx):
    a, b = hn[idx].copy(), hs[idx].copy()
    r = O.mpc_select(cfg, br, sz, chunk[idx], prev[idx], buf[idx], a, b)
    return idx, r, a, b


t0 = time.perf_counter()
bad = 0
with ThreadPoolExecutor(cores) as ex:
    for idx, (a_o, f_o, J_o, _), n_o, s_o in ex.map(run, np.array_split(np.arange(N), cores * 4)):
        bad += int((act[idx] != a_o).sum()) + int((flat[idx] != f_o).sum()) + int((J[idx] != J_o).sum())
        bad += int((hn_g[idx] != n_o).sum()) + int((hs_g[idx] != s_o).sum())
t_cpu = time.perf_counter() - t0
print(json.dumps(dict(lanes=N, combos=N * B ** H, mismatches=bad, gpu_seconds=round(t_gpu, 4),
                      oracle_seconds=round(t_cpu, 2), oracle_threads=cores,
                      compared="action, flat arg-min, J (float64 ==), history n and sum(1/x) (==)")))
sys.exit(1 if bad else 0)
