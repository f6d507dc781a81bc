"""world_size-2 and world_size-8 gloo tests of the N>1 path on CPU: shard ranges, the global-lane
trace map, the counter-based policy under lane_id_base, and the one collective
(all-gather of (obs, reward)).  The per-rank stepper here is the ORACLE (tests
may use it; the product path is the HIP kernels, exercised by the -m gpu tests
with the same lane_id_base contract in test_env_gpu.py::test_auto_reset_and_lane_id_base)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


LADDER = [0.3, 0.75, 1.2, 1.85, 2.85, 4.3]
V, SEED, TOTAL = 6, 1234, 101


def _rollout(lane0, n, traces):
    """obs [V, 3, n] / reward [V, n] of lanes lane0..lane0+n with the philox policy."""
    import sys
    sys.path.insert(0, ROOT)
    from abrsimulator_amd.sharding import lane_assignment
    from oracle import oracle as O
    tid, off = lane_assignment(lane0, n, [len(t) for t in traces])
    acts = np.stack([O.philox_action(SEED, np.arange(lane0, lane0 + n), s, 0, 6) for s in range(V)], 1)
    cfg = O.env_cfg(LADDER, 4.0, V, 20.0, 8.0, 1.0, [4.3, 1, 1, 0.1], 1.0)
    steps, bw, fin, _ = O.env_batch(cfg, traces, tid, off, acts.astype(np.int32))
    obs = np.stack([steps["buffer_level"].T, steps["global_time"].T, steps["last_bandwidth"].T], 1)
    rew = 4.3 * np.diff(np.concatenate([steps["rebuffer_time"], fin["rebuffer_time"][:, None]], 1), axis=1).T
    return obs.astype(np.float32), rew.astype(np.float32)


def _traces():
    rng = np.random.default_rng(0)
    return [rng.uniform(0.2, 6.0, 500).astype(np.float32).astype(np.float64) for _ in range(7)]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, ROOT)
    from abrsimulator_amd.sharding import ObsRewardGather, shard_range, unshard_lanes
    lane0, n = shard_range(TOTAL, world, rank)
    counts = [shard_range(TOTAL, world, r)[1] for r in range(world)]
    nmax = max(counts)
    obs, rew = _rollout(lane0, n, _traces())
    # pad the lane axis to the largest shard (all_gather needs equal shapes)
    po = np.zeros((V, 3, nmax), np.float32); po[..., :n] = obs
    pr = np.zeros((V, nmax), np.float32); pr[..., :n] = rew
    g = ObsRewardGather((V, 3, nmax), (V, nmax), "cpu")
    send = torch.cat([torch.from_numpy(po).reshape(-1), torch.from_numpy(pr).reshape(-1)])
    go, gr = g.gather(0, send)                      # ONE collective for (obs, reward)
    g.finish()
    assert g.n_collectives == 1
    full_o = unshard_lanes(go, counts).numpy()
    full_r = unshard_lanes(gr, counts).numpy()
    if rank == 0:
        np.save(out + "_o.npy", full_o)
        np.save(out + "_r.npy", full_r)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    from abrsimulator_amd.sharding import shard_range
    for total in (1, 7, 64, 101, 1048576):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(n for _, n in spans) == total
            for (a, n), (b, _) in zip(spans, spans[1:]):
                assert a + n == b
            assert max(n for _, n in spans) - min(n for _, n in spans) <= 1
    assert shard_range(1048576, 8, 3) == (393216, 131072)


@pytest.mark.parametrize("world", [2, 8])     # 8 = the widest run the driver launches (101 lanes: uneven shards of 13 and 12)
def test_gather_over_ranks_equals_unsharded(tmp_path, world):
    out = str(tmp_path / "g")
    port = _free_port()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    full_o, full_r = np.load(out + "_o.npy"), np.load(out + "_r.npy")
    ref_o, ref_r = _rollout(0, TOTAL, _traces())
    assert np.array_equal(full_o, ref_o) and np.array_equal(full_r, ref_r)


def test_xcd_aware_lane_assignment():
    from abrsimulator_amd.sharding import lane_assignment
    lens = [1000] * 1024
    tid, off = lane_assignment(0, 65536, lens, xcd_groups=8)
    assert tid.min() == 0 and tid.max() == 1023 and (off >= 0).all() and (off < 1000).all()
    w = np.arange(65536) // 64
    assert ((tid % 8) == (w % 8)).all()                 # workgroup w reads its XCD group's traces only
    assert len(np.unique(tid)) == 1024                  # every trace is still used
    # a shard of a larger job reproduces the unsharded map (global lane ids) ...
    tid2, off2 = lane_assignment(131072, 4096, lens, xcd_groups=8)
    tidf, offf = lane_assignment(0, 262144, lens, xcd_groups=8)
    assert (tid2 == tidf[131072:131072 + 4096]).all() and (off2 == offf[131072:131072 + 4096]).all()
    # ... and keeps the XCD property relative to its own workgroups when lane0 % 512 == 0
    assert ((tid2 % 8) == ((np.arange(4096) // 64) % 8)).all()
    # falls back to i % n_traces when the table does not split evenly
    tid3, _ = lane_assignment(0, 100, [10] * 7, xcd_groups=8)
    assert (tid3 == np.arange(100) % 7).all()
