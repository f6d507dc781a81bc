#!/usr/bin/env python3
"""Config-space fuzz on the GPU box: random configurations (chunk lengths, trace intervals, ladders,
buffer limits, start-up lengths, ragged traces with wrap-around) x random features (one speed, one
speed per lane, a speed schedule, per-chunk ladders) x kernel implementation (role-split with three / two waves, one thread per
lane, asynchronous pipeline -- which serves the speed features through the role-split kernels) x call form
(V single steps, or one fused scripted rollout), every lane's
previous_bandwidths (float64 ==), EVERY per-step reward (== float32 of the value derived from the oracle's
per-call-site rebuffer / start-up timers and the ladder rows, tests/helpers.py: expected_rewards), final clocks /
buffer / play_time (==), play_id (==) and episode QoE (1e-10) against the C oracle.   usage: python tools/gpu_fuzz.py [n_seeds] [lanes]"""
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
from oracle import oracle as O  # noqa: E402
from test_lane_jump_cpu import _random_config  # noqa: E402
from helpers import DIAG_IMPLS, diag_lib, oracle_rewards  # noqa: E402

IMPLS_ALL = ["split3", "split", "jump", "async", "ring3"]    # async / ring3: the diagnostic build (rejected pipelines)
FEATURES = ["plain", "lane_speeds", "schedule", "vbr", "schedule+vbr"]


def run_seed(seed, N, impls=IMPLS_ALL):
    """One random configuration against the oracle.  Returns (mismatching elements, lane-steps, case name, description)."""
    rng = np.random.default_rng(50_000 + seed)
    meta, (lo, hi) = _random_config(rng)
    V, B = meta["video_length"], len(meta["ladder"])
    n_traces = 5
    lens = rng.integers(40, 3000, n_traces)
    traces = [rng.uniform(lo, hi, l).astype(np.float32).astype(np.float64) for l in lens]
    tid = rng.integers(0, n_traces, N).astype(np.int32)
    off = np.array([rng.integers(0, lens[t]) for t in tid], np.int32)
    actions = rng.integers(0, B, (N, V)).astype(np.int32)
    feature = FEATURES[seed % 5]
    impl = impls[(seed // 5) % len(impls)]
    fused = (seed // 25) % 2 == 1 or impl in ("async", "ring3")    # one abr_env_step_script call instead of V abr_env_step calls
    speeds, br = None, None
    if "lane_speeds" in feature:
        speeds = rng.choice([0.6, 0.8, 1.0, 1.25, 1.7, 0.9173], N)
    if "schedule" in feature:
        speeds = rng.choice([0.5, 0.75, 1.0, 1.1, 1.25, 1.5, 2.0], (N, int(rng.integers(2, 7))))
    if "vbr" in feature:
        br = np.array(meta["ladder"])[None, :] * rng.uniform(0.7, 1.3, (V, B))
    cfg = O.env_cfg(meta["ladder"], meta["chunk_length"], V, meta["max_buffer"], meta["start_up_length"],
                    meta["interval"], meta["weights"], meta["speed"], br_table=br)
    steps, bw, fin, _ = O.env_batch(cfg, traces, tid, off, actions, max_ticks=6_000_000, speeds=speeds)
    want_rew = oracle_rewards(steps, fin, actions, meta["weights"], ladder=meta["ladder"], br_table=br)
    chunks = A.Chunk(meta["ladder"]) if br is None else [A.Chunk(list(r)) for r in br]
    mpd = A.MPD(V, meta["chunk_length"], meta["max_buffer"], meta["start_up_length"], chunks)
    sp = meta["speed"] if speeds is None else torch.from_numpy(speeds if speeds.ndim == 1 else speeds.T.copy())
    env = A.BatchedABREnv(mpd, A.QOEMetric(*meta["weights"]), A.NetworkInfo(meta["interval"], traces), N,
                          speed=sp, impl=impl, max_ticks=int(fin["ticks"].max()) + 1000,
                          library=diag_lib() if impl in DIAG_IMPLS else None)
    env.reset(torch.from_numpy(tid), torch.from_numpy(off))
    acts = torch.from_numpy(actions).cuda()
    if fused:
        rew = env.step_script(acts.T.contiguous())["reward"].cpu().numpy().T
    else:
        rew = np.stack([env.step(acts[:, s].contiguous())[1].cpu().numpy().copy() for s in range(V)], 1)
    f = {k: v.cpu().numpy() for k, v in env.observe_f64().items()}
    b = int((env.history()[1].cpu().numpy().T != bw).sum())
    b += int((rew != want_rew).sum())
    for k in ("global_time", "rebuffer_time", "start_up_time", "play_time", "buffer_level"):
        b += int((f[k] != fin[k]).sum())
    b += int((f["play_id"].astype(np.int32) != fin["play_id"]).sum())
    b += int((~np.isclose(env.episode_qoe().cpu().numpy(), fin["qoe"], rtol=1e-10, atol=1e-12)).sum())
    b += int((~np.isclose(f["average_latency"], fin["average_latency"], rtol=1e-9, atol=1e-12)).sum())
    env.close()
    return b, N * V, feature + "/" + impl + ("/fused" if fused else ""), (seed, feature, impl, meta)


COMPARED = ("previous_bandwidths float64 ==, every per-step reward == float32(oracle-derived), final global/rebuffer/start_up/play "
            "time, buffer_level, play_id ==, episode QoE 1e-10, average_latency 1e-9")


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    t0 = time.time()
    bad, lane_steps, feats = 0, 0, {}
    for seed in range(n_seeds):
        b, ls, key, what = run_seed(seed, N)
        if b:
            print("MISMATCH seed", *what[:3], b, what[3])
        bad += b
        lane_steps += ls
        feats[key] = feats.get(key, 0) + 1
    print(json.dumps(dict(seeds=n_seeds, lanes_per_seed=N, lane_steps=lane_steps, mismatches=bad,
                          cases=feats, seconds=round(time.time() - t0, 1), compared=COMPARED)))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
