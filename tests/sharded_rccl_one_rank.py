"""ShardedABREnv (abrsimulator_amd/sharding.py) over the RCCL backend ("nccl" on ROCm) with a ONE-rank group -- a one-GPU box
cannot host two RCCL ranks -- run as a fresh process by tests/test_baseline_configs_gpu.py: every launch form of the class
(random policy, scripted, MPC-driven), the all-gather on the side stream, double buffering, gathered == local."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    import abrsimulator_amd as A
    from test_baseline_configs_gpu import LADDER, META, _traces
    V, N, seed = 6, 4096 + 7, 5
    traces = [np.asarray(t, np.float64) for t in _traces(16, seed=9)]
    mpd = A.MPD(V, META["chunk_length"], META["max_buffer"], META["start_up_length"], A.Chunk(LADDER))
    qoe = A.QOEMetric(*META["weights"])
    sh = A.ShardedABREnv(mpd, qoe, A.NetworkInfo(1.0, traces), total_lanes=N, fuse=V, device="cuda", auto_reset=True)
    assert (sh.world, sh.rank, sh.lane0, sh.n_lanes, sh.counts) == (1, 0, 0, N, [N])
    sh.reset()
    ref = A.BatchedABREnv(mpd, qoe, A.NetworkInfo(1.0, traces), N, auto_reset=True)
    tid, off = sh.lane_map()
    ref.reset(torch.from_numpy(tid), torch.from_numpy(off))
    # random policy, two launches in flight (double-buffered slabs), then the gathered results
    a, b = sh.step_random(V, seed), sh.step_random(V, seed)
    ra, rb = ref.step_random(V, seed), ref.step_random(V, seed)
    for st, r in ((a, ra), (b, rb)):
        go, gr = st.gathered()
        torch.cuda.synchronize()
        assert go.shape == (1, 8, N) and gr.shape == (1, V, N)
        assert torch.equal(go[0], r["obs"][V - 1]) and torch.equal(gr[0], r["reward"])
        uo, ur = st.unsharded()
        assert torch.equal(uo, r["obs"][V - 1]) and torch.equal(ur, r["reward"])
    assert torch.equal(b.local["obs"], rb["obs"]) and torch.equal(b.local["done"], rb["done"])
    # a scripted launch and an MPC-driven one through the same class
    acts = torch.from_numpy(np.random.default_rng(1).integers(0, 6, (V, N)).astype(np.int32)).cuda()
    c, rc = sh.step_script(acts), ref.step_script(acts)
    go, gr = c.gathered()
    torch.cuda.synchronize()
    assert torch.equal(go[0], rc["obs"][V - 1]) and torch.equal(gr[0], rc["reward"])
    mp = A.MPD(V, META["chunk_length"], META["max_buffer"], META["start_up_length"],
               [A.Chunk(LADDER, [x * META["chunk_length"] for x in LADDER])] * V)
    ctl = A.BatchedMPCController(A.EnvPlayer(sh.env, mpd=mp, qoe=A.QOEMetric(4.3, 1.0, 0.0)), horizon=3)
    ctl2 = A.BatchedMPCController(A.EnvPlayer(ref, mpd=mp, qoe=A.QOEMetric(4.3, 1.0, 0.0)), horizon=3)
    d, rd = sh.step_mpc(ctl, V), ref.step_mpc(ctl2, V)
    go, gr = d.gathered()
    torch.cuda.synchronize()
    assert torch.equal(go[0], rd["obs"][V - 1]) and torch.equal(gr[0], rd["reward"])
    assert torch.equal(d.local["actions"], rd["actions"])
    # an odd launch shape is local only
    e = sh.step_random(2, seed)
    try:
        e.gathered()
        raise SystemExit("a launch shape other than `fuse` must not claim a gather")
    except RuntimeError:
        pass
    sh.finish()
    assert sh.n_collectives == 4
    dist.destroy_process_group()
    print("sharded one-rank RCCL ok")


if __name__ == "__main__":
    main()
