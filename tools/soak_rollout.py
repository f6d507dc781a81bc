#!/usr/bin/env python3
"""One-off soak on the GPU box: the fused MPC-driven rollout (abr_env_step_mpc) of N lanes x 48
chunks on mixed-length traces against the C oracle composition (run() whose ABR plugin is
next_bitrate() on run()'s own lists) on the host cores: EVERY lane's chosen bitrates (==),
previous_bandwidths (float64 ==), final clocks and buffer (==), episode QoE (1e-10).
usage: python tools/soak_rollout.py [n_lanes]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import abrsimulator_amd as A  # noqa: E402
from helpers import oracle_rewards  # noqa: E402
import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
V, L, H = bench.V, bench.L, 5
traces = bench.synth_traces(True)
tid, off = bench.lane_assignment(0, N, traces)
rng = np.random.default_rng(11)
br = np.array(bench.LADDER)[None, :] * rng.uniform(0.8, 1.2, (V, 6))
sz = br * L * rng.uniform(0.7, 1.3, (V, 6))
wv, wr = 0.5, 0.3                                      # weights under which all six rates get chosen
env = A.BatchedABREnv(A.MPD(V, L, bench.MAX_BUFFER, bench.START_UP, A.Chunk(bench.LADDER)),
                      A.QOEMetric(*bench.WEIGHTS), A.NetworkInfo(bench.INTERVAL, traces), N)
env.reset(torch.from_numpy(tid), torch.from_numpy(off))
mpd = A.MPD(V, L, bench.MAX_BUFFER, bench.START_UP, [A.Chunk(list(b), list(s)) for b, s in zip(br, sz)])
ctl = A.BatchedMPCController(A.EnvPlayer(env, mpd=mpd, qoe=A.QOEMetric(wr, wv, 0.0)), horizon=H)
t0 = time.perf_counter()
out = env.step_mpc(ctl, V, want_obs=False)
torch.cuda.synchronize()
t_gpu = time.perf_counter() - t0
acts = out["actions"].cpu().numpy().T
bh = env.history()[1].cpu().numpy().T
f = {k: v.cpu().numpy() for k, v in env.observe_f64().items()}
qoe = env.episode_qoe().cpu().numpy()

cores, _ = bench.host_cores()
ecfg = O.env_cfg(bench.LADDER, L, V, bench.MAX_BUFFER, bench.START_UP, bench.INTERVAL, bench.WEIGHTS, 1.0)
mcfg = O.mpc_cfg(6, H, V, L, bench.MAX_BUFFER, wv, wr, 0.0)
t0 = time.perf_counter()
steps, bw, a_o, fin = O.env_batch_mpc(ecfg, mcfg, br, sz, traces, tid, off, threads=cores)
t_cpu = time.perf_counter() - t0
bad = int((acts != a_o).sum()) + int((bh != bw).sum())
# every per-step reward == float32 of the oracle-derived value (the environment downloads from its single ladder)
bad += int((out["reward"].cpu().numpy().T != oracle_rewards(steps, fin, a_o, bench.WEIGHTS, ladder=bench.LADDER)).sum())
for k in ("global_time", "rebuffer_time", "start_up_time", "play_time", "buffer_level"):
    bad += int((f[k] != fin[k]).sum())
bad += int((~np.isclose(qoe, fin["qoe"], rtol=1e-10, atol=0)).sum())
print(json.dumps(dict(lanes=N, decisions=N * V, combos=N * (V - 1) * 6 ** H, rates_used=int(len(np.unique(a_o))),
                      mismatches=bad, gpu_seconds=round(t_gpu, 3), oracle_seconds=round(t_cpu, 1),
                      oracle_threads=cores,
                      compared="actions ==, previous_bandwidths float64 ==, every per-step reward == float32(oracle-derived), "
                               "final clocks and buffer ==, "
                               "episode QoE rtol 1e-10; traces 300-3000 points (wrap-around)")))
sys.exit(1 if bad else 0)
