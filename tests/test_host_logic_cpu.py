"""Host-side logic that needs no GPU: the packed (obs, reward) slab of the one collective, the speed
controller pre-query of the Simulator class, per-chunk MPD tables."""
import numpy as np
import pytest
import torch


def test_make_slab_views_share_one_contiguous_buffer():
    from abrsimulator_amd.sharding import make_slab
    F, D, N = 5, 8, 12
    slab, obs, reward, send = make_slab(F, D, N, "cpu")
    assert slab.numel() == F * D * N + F * N and slab.is_contiguous()
    obs.copy_(torch.arange(F * D * N, dtype=torch.float32).view(F, D, N))
    reward.copy_(-torch.arange(F * N, dtype=torch.float32).view(F, N))
    # the send view = [last step's observation | all rewards], contiguous, no copy
    assert send.is_contiguous() and send.data_ptr() == obs[F - 1].data_ptr()
    assert send.numel() == D * N + F * N
    assert torch.equal(send[:D * N].view(D, N), obs[F - 1])
    assert torch.equal(send[D * N:].view(F, N), reward)
    # fuse 1: the send view is the whole slab
    slab1, obs1, rew1, send1 = make_slab(1, D, N, "cpu")
    assert send1.data_ptr() == slab1.data_ptr() and send1.numel() == slab1.numel()


def test_simulator_speed_prequery():
    """get_next_speed() is asked once per possible played chunk; a constant answer stays a float
    (cheap path), anything else becomes a [video_length, n_lanes] schedule."""
    import abrsimulator_amd as A

    class Const:
        def __init__(self):
            self.calls = 0

        def get_next_speed(self):
            self.calls += 1
            return 1.25

    class Script:
        def __init__(self):
            self.calls = 0

        def get_next_speed(self):
            self.calls += 1
            return 1.0 + 0.1 * (self.calls % 3)

    class PerLane:
        def get_next_speed(self):
            return torch.tensor([1.0, 0.8, 1.2, 1.5], dtype=torch.float64)

    mpd = A.MPD(6, 4.0, 20.0, 8.0, A.Chunk([1.0, 2.0]))
    for ctl, kind in ((Const(), "const"), (Script(), "sched"), (PerLane(), "lanes")):
        sim = A.Simulator(object(), ctl, n_lanes=4)
        sim.mpd = mpd
        sp = sim._speeds()
        if kind == "const":
            assert sp == 1.25 and ctl.calls == 6
        elif kind == "sched":
            assert sp.shape == (6, 4) and sp.dtype == torch.float64 and ctl.calls == 6
            assert torch.allclose(sp[:, 0], torch.tensor([1.1, 1.2, 1.0, 1.1, 1.2, 1.0], dtype=torch.float64))
            assert bool((sp == sp[:, :1]).all())
        else:
            assert sp.shape == (6, 4) and torch.equal(sp[3], torch.tensor([1.0, 0.8, 1.2, 1.5], dtype=torch.float64))


def test_per_chunk_mpd_tables():
    import abrsimulator_amd as A
    same = A.MPD(3, 4.0, 20.0, 8.0, [A.Chunk([1.0, 2.0])] * 3)
    assert same.uniform() and same.ladder() == [1.0, 2.0] and same.bitrate_table() == [[1.0, 2.0]] * 3
    diff = A.MPD(2, 4.0, 20.0, 8.0, [A.Chunk([1.0, 2.0]), A.Chunk([1.5, 2.5])])
    assert not diff.uniform() and diff.bitrate_table() == [[1.0, 2.0], [1.5, 2.5]]
    with pytest.raises(ValueError):
        diff.ladder()
    single = A.MPD(4, 4.0, 20.0, 8.0, A.Chunk([0.5, 1.0, 2.0]))
    assert single.uniform() and len(single.bitrate_table()) == 4


def test_lane_assignment_is_a_function_of_the_global_lane_id():
    from abrsimulator_amd.sharding import lane_assignment, shard_range
    lens = list(np.random.default_rng(0).integers(300, 3001, 64))
    tid, off = lane_assignment(0, 4096, lens)
    for world in (2, 8):
        parts = [lane_assignment(*shard_range(4096, world, r), lens) for r in range(world)]
        assert np.array_equal(np.concatenate([p[0] for p in parts]), tid)
        assert np.array_equal(np.concatenate([p[1] for p in parts]), off)
    assert (off < np.asarray(lens)[tid]).all() and off.min() >= 0


def test_mpc_binding_follows_in_place_changes_of_the_mpd():
    """The controller binds its config struct once per (lanes, tables, horizon, method, weights, chunk_length, max_buffer):
    an in-place change of the MPD's max_buffer or chunk_length (mutable dataclass; config() re-reads both, and
    env.step_mpc() calls config() fresh) must rebind on the next select, not be ignored until update_mpd()."""
    import abrsimulator_amd as A

    class Player:
        def __init__(self):
            self.mpd = A.MPD(8, 4.0, 20.0, 8.0, A.Chunk([1.0, 2.0, 3.0]))
            self.qoe = A.QOEMetric(4.3, 1.0, 0.0)

        def get_mpd(self):
            return self.mpd

        def get_qoe_metric(self):
            return self.qoe

    p = Player()
    ctl = A.BatchedMPCController(p, horizon=3, device="cpu")
    ctl._tables_for = p.mpd                       # (the tables themselves live on the GPU: not built here)
    k0 = ctl._bind_key(16)
    assert ctl._bind_key(16) == k0 and ctl._bind_key(17) != k0
    p.mpd.max_buffer = 12.0
    k1 = ctl._bind_key(16)
    assert k1 != k0
    p.mpd.chunk_length = 2.0
    assert ctl._bind_key(16) != k1
    p.qoe.rebuffer_weight = 1.0
    assert ctl._bind_key(16)[-2] == 1.0
