#!/usr/bin/env python3
"""One-off soak on the GPU box: a fused random-policy rollout of N lanes x 48 decisions on the
MI355X against the C oracle on the host cores, comparing EVERY lane's previous_bandwidths
(float64, bit-exact), every per-step reward (== float32 of the value derived from the oracle's per-call-site
timers, tests/helpers.py: expected_rewards), final clocks and buffer (bit-exact) and episode QoE (1e-10).
usage: python tools/soak_parity.py [n_lanes] [mixed|uniform] [impl]"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import abrsimulator_amd as A  # noqa: E402
from helpers import DIAG_IMPLS, diag_lib, oracle_rewards  # noqa: E402
import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402

def soak(N, mixed=False, impl="auto", seed=20260404):
    """The whole comparison for N lanes; returns the record (mismatches == 0 is the claim)."""
    V = bench.V
    traces = bench.synth_traces(mixed)
    tid, off = bench.lane_assignment(0, N, traces)
    env = A.BatchedABREnv(A.MPD(V, bench.L, bench.MAX_BUFFER, bench.START_UP, A.Chunk(bench.LADDER)),
                          A.QOEMetric(*bench.WEIGHTS), A.NetworkInfo(bench.INTERVAL, traces), N, impl=impl,
                          library=diag_lib() if impl in DIAG_IMPLS else None)
    env.reset(torch.from_numpy(tid), torch.from_numpy(off))
    t0 = time.perf_counter()
    out = env.step_random(V, seed, out=dict(obs=None, reward=torch.empty(V, N, dtype=torch.float32, device="cuda"),
                                            done=None, actions=torch.empty(V, N, dtype=torch.int32, device="cuda")))
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    acts = out["actions"].cpu().numpy().T.copy()
    rew = out["reward"].cpu().numpy()                         # [V, N]
    bh = env.history()[1].cpu().numpy()                       # [V, N]
    f = {k: v.cpu().numpy() for k, v in env.observe_f64().items()}
    qoe = env.episode_qoe().cpu().numpy()
    kernel = env.effective_impl(fused=True)
    env.close()

    cfg = O.env_cfg(bench.LADDER, bench.L, V, bench.MAX_BUFFER, bench.START_UP, bench.INTERVAL, bench.WEIGHTS, 1.0)
    cores, _ = bench.host_cores()
    chunks = np.array_split(np.arange(N), cores * 4)

    def run(idx):
        steps, bw, fin, _ = O.env_batch(cfg, traces, tid[idx], off[idx], acts[idx])
        return idx, bw, fin, oracle_rewards(steps, fin, acts[idx], bench.WEIGHTS, ladder=bench.LADDER)

    t0 = time.perf_counter()
    bad = 0
    with ThreadPoolExecutor(cores) as ex:
        for idx, bw, fin, want_rew in ex.map(run, chunks):
            bad += int((bh[:, idx].T != bw).sum())
            bad += int((rew[:, idx].T != want_rew).sum())
            for k in ("global_time", "rebuffer_time", "start_up_time", "play_time", "buffer_level"):
                bad += int((f[k][idx] != fin[k]).sum())
            bad += int((~np.isclose(qoe[idx], fin["qoe"], rtol=1e-10, atol=0)).sum())
    t_cpu = time.perf_counter() - t0
    return dict(lanes=N, impl=impl, kernel=kernel, decisions=N * V, mixed_traces=mixed, mismatches=bad, gpu_seconds=round(t_gpu, 4),
                oracle_seconds=round(t_cpu, 2), oracle_threads=cores,
                compared="previous_bandwidths float64 [V,N] ==, every per-step reward [V,N] == float32(oracle-derived), "
                         "final global/rebuffer/start_up/play time and buffer_level ==, episode QoE rtol 1e-10")


if __name__ == "__main__":
    res = soak(int(sys.argv[1]) if len(sys.argv) > 1 else 1048576, len(sys.argv) > 2 and sys.argv[2] == "mixed",
               sys.argv[3] if len(sys.argv) > 3 else "auto")
    print(json.dumps(res))
    sys.exit(1 if res["mismatches"] else 0)
