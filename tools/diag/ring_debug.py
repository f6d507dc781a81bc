"""Debug aid (round 5): the config fuzz of tests/test_async_gpu.py on the ring kernel, mismatches located."""
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from helpers import make_env  # noqa: E402
from oracle import oracle  # noqa: E402
from test_lane_jump_cpu import _random_config  # noqa: E402

for seed in [int(x) for x in sys.argv[1:]] or range(16):
    rng = np.random.default_rng(1000 + seed)
    meta, (lo, hi) = _random_config(rng)
    meta["speed"] = 1.0 if seed % 2 else meta["speed"]
    n_traces, N = 6, 300
    lens = rng.integers(40, 3000, n_traces)
    traces = [rng.uniform(lo, hi, l).astype(np.float32).astype(np.float64) for l in lens]
    trace_id = rng.integers(0, n_traces, N).astype(np.int32)
    offset = np.array([rng.integers(0, lens[t]) for t in trace_id], np.int32)
    V = meta["video_length"]
    actions = rng.integers(0, len(meta["ladder"]), (N, V)).astype(np.int32)
    cfg = oracle.env_cfg(meta["ladder"], meta["chunk_length"], V, meta["max_buffer"],
                         meta["start_up_length"], meta["interval"], meta["weights"], meta["speed"])
    steps, bw, fin, _ = oracle.env_batch(cfg, traces, trace_id, offset, actions, max_ticks=4_000_000)
    envj = make_env(meta, traces, N, impl="jump", max_ticks=int(fin["ticks"].max()) + 1000)
    envj.reset(torch.from_numpy(trace_id), torch.from_numpy(offset))
    outj = envj.step_script(torch.from_numpy(actions.T.copy()))
    fj = envj.observe_f64()
    for impl in ("ring3",):
        env = make_env(meta, traces, N, impl=impl, max_ticks=int(fin["ticks"].max()) + 1000)
        env.reset(torch.from_numpy(trace_id), torch.from_numpy(offset))
        out = env.step_script(torch.from_numpy(actions.T.copy()))
        f = env.observe_f64()
        for k in ("obs", "reward", "done"):
            if not torch.equal(out[k], outj[k]):
                d = (out[k] != outj[k]).nonzero()
                print("   differs from jump:", k, d[:5].tolist())
        for k in f:
            if not torch.equal(f[k], fj[k]):
                d = (f[k] != fj[k]).nonzero().flatten()
                print("   f64 state differs from jump:", k, d[:5].tolist(), f[k][d[:3]].tolist(), fj[k][d[:3]].tolist())
        if not torch.equal(env.workspace, envj.workspace):
            print("   workspace bytes differ:", int((env.workspace != envj.workspace).sum()))
        got = env.history()[1].cpu().numpy().T
        bad = np.argwhere(got != bw)
        print(f"seed {seed} V {V} interval {meta['interval']} L {meta['chunk_length']} max_buffer {meta['max_buffer']}: "
              f"{len(bad)} mismatches of {got.size}; buffer_full call sites {int(steps['buffer_full'].sum())}")
        for (i, c) in bad[:6]:
            print(f"   lane {i} chunk {c}: got {got[i, c]!r} want {bw[i, c]!r}  bf[lane] {steps['buffer_full'][i].tolist()} "
                  f"done {out['done'][:, i].cpu().numpy().tolist()}")
