"""world_size-2 and world_size-8 gloo tests of the N>1 path on CPU, through the package's ShardedABREnv
(abrsimulator_amd/sharding.py): shard ranges, the global-lane trace map, the counter-based policy under
lane_id_base, slabs, the staging copy of uneven shards, double buffering, and the one collective (all-gather of
(obs, reward)).  The per-rank stepper plugged into the class here is the ORACLE (tests may use it; the product's
stepper is the HIP library, exercised by the -m gpu tests with the same lane_id_base contract in
test_env_gpu.py::test_auto_reset_and_lane_id_base and by the two-rank HIP-shard test)."""
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


WEIGHTS = [4.3, 1, 1, 0.1]


def _rollout(lane0, n, traces, tid=None, off=None):
    """obs [V, 8, n] / reward [V, n] float32 of lanes lane0..lane0+n with the philox policy, as a fused launch of V
    decisions writes them: the observation after decision s is the run() frame at call site s + 1 (the last one: the
    frame calculate_qoe was called from); rewards from the oracle's timers (oracle.step_rewards)."""
    import sys
    sys.path.insert(0, ROOT)
    from abrsimulator_amd.sharding import lane_assignment
    from oracle import oracle as O
    if tid is None:
        tid, off = lane_assignment(lane0, n, [len(t) for t in traces])
    acts = np.stack([O.philox_action(SEED, np.arange(lane0, lane0 + n), s, 0, 6) for s in range(V)], 1).astype(np.int32)
    cfg = O.env_cfg(LADDER, 4.0, V, 20.0, 8.0, 1.0, WEIGHTS, 1.0)
    steps, bw, fin, _ = O.env_batch(cfg, traces, np.asarray(tid), np.asarray(off), acts)
    obs = np.zeros((V, 8, n), np.float32)
    for s in range(V - 1):
        q = s + 1
        rows = [steps["chunk_id"][:, q], acts[:, s], steps["last_bandwidth"][:, q], steps["buffer_level"][:, q],
                steps["global_time"][:, q], steps["play_time"][:, q], steps["rebuffer_time"][:, q], steps["start_up_time"][:, q]]
        obs[s] = np.stack([np.asarray(r, np.float64).astype(np.float32) for r in rows])
    rows = [fin["chunk_id"], acts[:, V - 1], bw[:, V - 1], fin["buffer_level"], fin["global_time"], fin["play_time"],
            fin["rebuffer_time"], fin["start_up_time"]]
    obs[V - 1] = np.stack([np.asarray(r, np.float64).astype(np.float32) for r in rows])
    rew = O.step_rewards(steps["rebuffer_time"], steps["start_up_time"], fin["rebuffer_time"], fin["start_up_time"],
                         acts, WEIGHTS, ladder=LADDER)
    return obs, rew.T.copy()


class OracleShardEnv:
    """CPU stand-in for BatchedABREnv inside ShardedABREnv: the same n_lanes / reset / step_random(out=) surface, stepped
    by the oracle (tests may use it; the product's stepper is the HIP library).  Everything else the gloo tests exercise --
    shard ranges, global lane ids, the lane map, slabs, the staging copy of uneven shards, the one collective, the
    double buffering -- is the package's ShardedABREnv itself."""

    def __init__(self, traces, lane0, n):
        self.traces, self.lane0, self.n_lanes = traces, lane0, n
        self.tid = self.off = None

    def reset(self, trace_id, start_offset):
        self.tid, self.off = trace_id.numpy(), start_offset.numpy()

    def step_random(self, n_steps, seed, out=None):
        assert n_steps == V and seed == SEED
        obs, rew = _rollout(self.lane0, self.n_lanes, self.traces, self.tid, self.off)
        out["obs"].copy_(torch.from_numpy(obs)); out["reward"].copy_(torch.from_numpy(rew))
        out["done"].fill_(0)
        return out


def _traces():
    rng = np.random.default_rng(0)
    return [rng.uniform(0.2, 6.0, 500).astype(np.float32).astype(np.float64) for _ in range(7)]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, ROOT)
    import abrsimulator_amd as A
    from abrsimulator_amd.sharding import ShardedABREnv, shard_range
    traces = _traces()
    lane0, n = shard_range(TOTAL, world, rank)
    sh = ShardedABREnv(None, None, A.NetworkInfo(1.0, traces), total_lanes=TOTAL, fuse=V, device="cpu",
                       env=OracleShardEnv(traces, lane0, n))
    assert (sh.lane0, sh.n_lanes, sh.world, sh.rank) == (lane0, n, world, rank) and sum(sh.counts) == TOTAL
    sh.reset()                                      # the global lane -> (trace, offset) map
    st1 = sh.step_random(V, SEED)                   # ONE collective for (obs, reward) per launch ...
    st2 = sh.step_random(V, SEED)                   # ... double-buffered: the second launch's slab is another one
    sh.finish()
    assert sh.n_collectives == 2
    full_o, full_r = st1.unsharded()
    o2, r2 = st2.unsharded()
    assert torch.equal(full_o, o2) and torch.equal(full_r, r2)
    loc = st1.local
    assert loc["obs"].shape == (V, 8, n) and loc["reward"].shape == (V, n)
    if rank == 0:
        np.save(out + "_o.npy", full_o.numpy())
        np.save(out + "_r.npy", full_r.numpy())
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
    """N ranks through the package's ShardedABREnv == the unsharded rollout, lane for lane: the gathered final observation
    [8, total] and rewards [V, total]."""
    out = str(tmp_path / "g")
    port = _free_port()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    full_o, full_r = np.load(out + "_o.npy"), np.load(out + "_r.npy")
    ref_o, ref_r = _rollout(0, TOTAL, _traces())
    assert full_o.shape == (8, TOTAL) and full_r.shape == (V, TOTAL)
    assert np.array_equal(full_o, ref_o[V - 1]) and np.array_equal(full_r, ref_r)


def test_sharded_env_without_a_process_group_is_the_whole_job():
    """No process group: one rank holds every lane, no collective is issued, the launch's outputs are the local ones."""
    import sys
    sys.path.insert(0, ROOT)
    import abrsimulator_amd as A
    from abrsimulator_amd.sharding import ShardedABREnv
    traces = _traces()
    sh = ShardedABREnv(None, None, A.NetworkInfo(1.0, traces), total_lanes=TOTAL, fuse=V, device="cpu",
                       env=OracleShardEnv(traces, 0, TOTAL))
    assert (sh.lane0, sh.n_lanes, sh.world, sh.counts) == (0, TOTAL, 1, [TOTAL])
    sh.reset()
    st = sh.step_random(V, SEED)
    ref_o, ref_r = _rollout(0, TOTAL, traces)
    assert np.array_equal(st.local["obs"].numpy(), ref_o) and np.array_equal(st.local["reward"].numpy(), ref_r)
    assert sh.n_collectives == 0
    with pytest.raises(RuntimeError):
        st.gathered()
    with pytest.raises(ValueError):
        ShardedABREnv(None, None, A.NetworkInfo(1.0, traces), fuse=V, device="cpu", env=OracleShardEnv(traces, 0, 4))


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
