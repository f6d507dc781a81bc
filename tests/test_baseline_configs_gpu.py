"""Every BASELINE.json configuration that fits one GPU, through the C ABI, against the oracle.

  configs[2]  K3 at 65 536 lanes, 6 rates, horizon 5, per-chunk (VBR) tables
  configs[3]  the per-rank shape of the 8-GPU job: 131 072 lanes with lane_id_base = r * 131072,
              r in {0, 7}, equal to the matching slice of the unsharded 1 048 576-lane run and to
              an oracle replay of sampled lanes
  configs[4]  mixed trace lengths 300-3 000 (wrap-around) x MPC-driven rollout, full 48-chunk
              episodes, against the oracle composition (oracle/abr_oracle.c: oracle_env_batch_mpc)
  N > 1       two fresh child processes, one HIP env shard each, gathered with the product's one
              collective (gloo here; RCCL needs two GPUs) == the unsharded HIP run
plus the ABI edges hardened in round 2 (reset range checks, latched lane speeds, episode
counter, previous_bitrate range).
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from helpers import make_env, oracle_rewards, philox_action

pytestmark = pytest.mark.gpu

LADDER = [0.3, 0.75, 1.2, 1.85, 2.85, 4.3]
V, L, MAX_BUFFER, START_UP = 48, 4.0, 20.0, 8.0
WEIGHTS = [4.3, 1, 1, 0.1]
META = dict(ladder=LADDER, chunk_length=L, video_length=V, max_buffer=MAX_BUFFER,
            start_up_length=START_UP, interval=1.0, weights=WEIGHTS, speed=1.0)


def _traces(n=1024, mixed=False, seed=0):
    rng = np.random.default_rng(seed)
    lens = rng.integers(300, 3001, n) if mixed else np.full(n, 1000)
    return [rng.uniform(0.2, 6.0, int(m)).astype(np.float32).astype(np.float64) for m in lens]


# ---------------------------------------------------------------------------------------------
# configs[2]: K3 at full size
# ---------------------------------------------------------------------------------------------
def test_mpc_65536_lanes_vbr_against_oracle(oracle):
    from test_mpc_gpu import _controller
    N, B, H = 65536, 6, 5
    rng = np.random.default_rng(2026)
    br = np.array(LADDER)[None, :] * rng.uniform(0.8, 1.2, (V, B))
    sz = br * L * rng.uniform(0.7, 1.3, (V, B))
    chunk = rng.integers(0, V - H + 1, N).astype(np.int32)
    prev = rng.integers(0, B, N).astype(np.int32)
    buf = np.where(rng.random(N) < 0.15, 0.0, rng.uniform(0, MAX_BUFFER, N))
    buf[rng.random(N) < 0.05] = MAX_BUFFER
    hn = rng.integers(1, 48, N).astype(np.float64)
    hs = hn / rng.uniform(0.2, 6.0, N)
    ctl, ci = _controller(br, sz, L, MAX_BUFFER, 4.3, 1.0, 0.0, H, chunk, prev, buf, hn, hs)
    a = ctl.next_bitrate(want_details=True).cpu().numpy()
    flat, J = ctl.last_flat.cpu().numpy(), ctl.last_J.cpu().numpy()
    assert a.min() >= 0 and a.max() < B and np.isfinite(J).all()
    # size-independent property on EVERY lane: the action is the leading digit of the arg-min
    assert np.array_equal(a, flat // B ** (H - 1))
    assert np.array_equal(ci.hist_n.cpu().numpy(), hn + H)                  # D9 everywhere
    # exact oracle replay of >= 2 048 sampled lanes incl. the first and last workgroups (7 lanes each)
    pick = np.unique(np.concatenate([np.arange(14), np.arange(N - 14, N), rng.integers(0, N, 2100)]))
    assert len(pick) >= 2048
    cfg = oracle.mpc_cfg(B, H, V, L, MAX_BUFFER, 1.0, 4.3, 0.0)
    hn_o, hs_o = hn[pick].copy(), hs[pick].copy()
    act, fl, Jm, _ = oracle.mpc_select(cfg, br, sz, chunk[pick], prev[pick], buf[pick], hn_o, hs_o)
    assert np.array_equal(a[pick], act)
    assert np.array_equal(flat[pick].astype(np.int64), fl)
    assert np.array_equal(J[pick], Jm)
    assert np.array_equal(ci.hist_sum_inv.cpu().numpy()[pick], hs_o)


# ---------------------------------------------------------------------------------------------
# configs[3]: per-rank shard shape of the 1 048 576-lane job
# ---------------------------------------------------------------------------------------------
def test_shard_shape_131072_lanes_equals_unsharded_slice_and_oracle(oracle):
    from abrsimulator_amd.sharding import lane_assignment, shard_range
    TOTAL, WORLD, SEED = 1048576, 8, 20260404
    traces = _traces()
    lens = [len(t) for t in traces]
    full = make_env(META, traces, TOTAL, auto_reset=True)
    tid, off = lane_assignment(0, TOTAL, lens)
    full.reset(torch.from_numpy(tid), torch.from_numpy(off))
    ref = full.step_random(V, SEED)
    assert int(ref["done"][-1].sum()) == TOTAL and int(ref["done"][:-1].sum()) == 0
    for r in (0, 7):
        lane0, n = shard_range(TOTAL, WORLD, r)
        assert (lane0, n) == (r * 131072, 131072)
        stid, soff = lane_assignment(lane0, n, lens)
        assert np.array_equal(stid, tid[lane0:lane0 + n]) and np.array_equal(soff, off[lane0:lane0 + n])
        sh = make_env(META, traces, n, auto_reset=True, lane_id_base=lane0)
        sh.reset(torch.from_numpy(stid), torch.from_numpy(soff))
        out = sh.step_random(V, SEED)
        for k in ("obs", "reward", "done", "actions"):
            assert torch.equal(out[k], ref[k][..., lane0:lane0 + n]), (r, k)
        if r != 7:
            continue
        # oracle replay of sampled lanes of the LAST shard (global lane ids in the philox counter)
        rng = np.random.default_rng(7)
        pick = np.unique(np.concatenate([[0, 63, 64, n - 64, n - 1], rng.integers(0, n, 1024)]))
        want = np.stack([philox_action(SEED, lane0 + pick, s, 0, 6) for s in range(V)], 1)
        acts = out["actions"].cpu().numpy()[:, pick].T
        assert np.array_equal(acts, want)
        cfg = oracle.env_cfg(LADDER, L, V, MAX_BUFFER, START_UP, 1.0, WEIGHTS, 1.0)
        steps, bw, fin, _ = oracle.env_batch(cfg, traces, stid[pick], soff[pick], acts.copy())
        o = out["obs"].cpu().numpy()[:, :, pick]
        for s in range(V - 1):
            assert np.array_equal(o[s, 3], steps["buffer_level"][:, s + 1].astype(np.float32)), s
            assert np.array_equal(o[s, 4], steps["global_time"][:, s + 1].astype(np.float32)), s
            assert np.array_equal(o[s, 2], steps["last_bandwidth"][:, s + 1].astype(np.float32)), s
        # every reward element of the sampled lanes == float32 of the oracle-derived value
        assert np.array_equal(out["reward"].cpu().numpy()[:, pick].T,
                              oracle_rewards(steps, fin, acts, WEIGHTS, ladder=LADDER))
        # the finished episode (auto_reset keeps its record): float64 QoE against the oracle
        assert np.allclose(sh.episode_qoe().cpu().numpy()[pick], fin["qoe"], rtol=1e-10)


# ---------------------------------------------------------------------------------------------
# configs[4]: mixed trace lengths x MPC-driven rollout
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", ["fused", "host_loop"])
@pytest.mark.parametrize("wv,wr", [(1.0, 4.3), (0.5, 0.3)])     # the bench weights; a mix of all six rates
def test_mpc_rollout_on_ragged_traces_against_oracle(oracle, wv, wr, path):
    import abrsimulator_amd as A
    from abrsimulator_amd.sharding import lane_assignment
    N, H = 512, 5
    traces = _traces(64, mixed=True, seed=4)
    lens = [len(t) for t in traces]
    assert min(lens) < 500 and max(lens) > 2500
    tid, off = lane_assignment(0, N, lens)
    rng = np.random.default_rng(44)
    br = np.array(LADDER)[None, :] * rng.uniform(0.8, 1.2, (V, 6))      # VBR: per-chunk tables
    sz = br * L * rng.uniform(0.7, 1.3, (V, 6))
    # the env itself downloads CBR chunks (one ladder, Simulator.py:156); the MPC plans on
    # the per-chunk tables (mpc.py:126-128) -- exactly the split the reference has
    env = make_env(META, traces, N)
    env.reset(torch.from_numpy(tid), torch.from_numpy(off))
    mpd = A.MPD(V, L, MAX_BUFFER, START_UP, [A.Chunk(list(b), list(s)) for b, s in zip(br, sz)])
    player = A.EnvPlayer(env, mpd=mpd, qoe=A.QOEMetric(wr, wv, 0.0))
    ctl = A.BatchedMPCController(player, horizon=H, clip_horizon=True)
    if path == "fused":
        # abr_env_step_mpc: select -> step for all V decisions on the device
        out = env.step_mpc(ctl, V)
        gpu_actions = out["actions"].cpu().numpy().T
        gpu_rewards = list(out["reward"].cpu().numpy())
        d = out["done"].cpu().numpy()
        assert (d[:-1] == 0).all() and (d[-1] == 1).all()
        # a second call on the finished lanes takes no decision and changes nothing
        before = env.observe_f64()["global_time"].clone()
        again = env.step_mpc(ctl, 2)
        assert (again["actions"].cpu().numpy() == -1).all() and (again["done"].cpu().numpy() == 1).all()
        assert torch.equal(env.observe_f64()["global_time"], before)
    else:
        gpu_actions, gpu_rewards = [], []
        for s in range(V):
            a = torch.clamp(ctl.next_bitrate(), min=0)      # D13 at chunk 0: "no decision" -> rate 0
            gpu_actions.append(a.cpu().numpy().copy())
            gpu_rewards.append(env.step(a)[1].cpu().numpy().copy())
        gpu_actions = np.stack(gpu_actions, 1)
    ecfg = oracle.env_cfg(LADDER, L, V, MAX_BUFFER, START_UP, 1.0, WEIGHTS, 1.0)
    mcfg = oracle.mpc_cfg(6, H, V, L, MAX_BUFFER, wv, wr, 0.0)
    steps, bw, acts, fin = oracle.env_batch_mpc(ecfg, mcfg, br, sz, traces, tid, off, threads=8)
    assert np.array_equal(gpu_actions, acts)
    # the environment downloads from the single ladder, so the reward's variance term uses it (:82)
    assert np.array_equal(np.stack(gpu_rewards, 1), oracle_rewards(steps, fin, acts, WEIGHTS, ladder=LADDER))
    if wr < 1.0:
        assert len(np.unique(acts)) == 6                     # every rate is exercised
    # some lanes wrapped around their trace (the reference would raise IndexError, D7)
    assert (fin["global_time"] > np.array(lens)[tid] - off).any()
    assert np.array_equal(env.history()[1].cpu().numpy().T, bw)
    assert np.allclose(env.episode_qoe().cpu().numpy(), fin["qoe"], rtol=1e-10)
    f = env.observe_f64()
    for k in ("global_time", "rebuffer_time", "start_up_time", "play_time", "buffer_level"):
        assert np.array_equal(f[k].cpu().numpy(), fin[k]), k


def _rollout_vs_oracle(oracle, env, ctl, traces, tid, off, br, sz, wv, wr, pick, H=5):
    """V fused MPC-driven decisions on every lane of `env`; the sampled lanes `pick` replayed through the
    oracle composition (run() whose plug-in is next_bitrate() on run()'s own lists; the wiring itself
    -- shared history incl. the D9 samples, D13 -> bitrate 0, D12 clip -- is build-defined, both
    halves are pinned to the reference)."""
    out = env.step_mpc(ctl, V, want_obs=False)
    d = out["done"].cpu().numpy()
    assert (d[:-1] == 0).all() and (d[-1] == 1).all()                 # every lane: exactly one episode
    a_all = out["actions"].cpu().numpy()
    assert a_all.min() >= 0 and a_all.max() < 6
    ecfg = oracle.env_cfg(LADDER, L, V, MAX_BUFFER, START_UP, 1.0, WEIGHTS, 1.0)
    mcfg = oracle.mpc_cfg(6, H, V, L, MAX_BUFFER, wv, wr, 0.0)
    steps, bw, acts, fin = oracle.env_batch_mpc(ecfg, mcfg, br, sz, traces, tid[pick], off[pick], threads=16)
    assert np.array_equal(a_all[:, pick].T, acts)
    # every reward element of the sampled lanes == float32 of the oracle-derived value (the environment
    # downloads from its single ladder, so that is what the variance term reads)
    assert np.array_equal(out["reward"].cpu().numpy()[:, pick].T, oracle_rewards(steps, fin, acts, WEIGHTS, ladder=LADDER))
    assert np.array_equal(env.history()[1].cpu().numpy()[:, pick].T, bw)
    assert np.allclose(env.episode_qoe().cpu().numpy()[pick], fin["qoe"], rtol=1e-10)
    f = env.observe_f64()
    for k in ("global_time", "rebuffer_time", "start_up_time", "play_time", "buffer_level"):
        assert np.array_equal(f[k].cpu().numpy()[pick], fin[k]), k
    # size-independent property on EVERY lane: sum(reward) + wl * average_latency == calculate_qoe
    rew = out["reward"].double().sum(0).cpu().numpy()
    qoe = env.episode_qoe().cpu().numpy()
    assert np.allclose(rew + WEIGHTS[3] * f["average_latency"].cpu().numpy(), qoe, rtol=1e-5)
    return acts


def test_config2_mpc_rollout_65536_lanes_full_episodes(oracle):
    """BASELINE.json configs[2] composed, at its real size: 65 536 envs x MPC horizon 5 over 6 rates,
    48-chunk episodes driven by abr_env_step_mpc (Simulator.py:155 x mpc.py:181-186), 1 100+ sampled
    lanes incl. the first and last workgroups == the oracle composition."""
    import abrsimulator_amd as A
    from abrsimulator_amd.sharding import lane_assignment
    N = 65536
    traces = _traces()
    tid, off = lane_assignment(0, N, [len(t) for t in traces])
    env = make_env(META, traces, N)
    env.reset(torch.from_numpy(tid), torch.from_numpy(off))
    br = np.tile(np.array(LADDER), (V, 1))
    sz = br * L                                                          # the bench's CBR tables
    mpd = A.MPD(V, L, MAX_BUFFER, START_UP, [A.Chunk(list(b), list(s)) for b, s in zip(br, sz)])
    ctl = A.BatchedMPCController(A.EnvPlayer(env, mpd=mpd, qoe=A.QOEMetric(4.3, 1.0, 0.0)), horizon=5)
    rng = np.random.default_rng(22)
    pick = np.unique(np.concatenate([np.arange(14), np.arange(N - 14, N), [63, 64, 127, 128],
                                     rng.integers(0, N, 1100)]))
    assert len(pick) >= 1024
    _rollout_vs_oracle(oracle, env, ctl, traces, tid, off, br, sz, 1.0, 4.3, pick)


def test_config4_shard_shape_131072_lanes_mixed_traces_mpc_rollout(oracle):
    """BASELINE.json configs[4], the per-rank shape of the 8-GPU job: rank 7's 131 072 lanes
    (lane_id_base = 7 * 131072: the lane -> (trace, offset) map is a function of the GLOBAL lane id),
    mixed 300-3 000-point traces (wrap-around: the divergent while-loop stress), MPC-driven rollout
    on VBR tables; 1 000+ sampled lanes == the oracle composition."""
    import abrsimulator_amd as A
    from abrsimulator_amd.sharding import lane_assignment, shard_range
    lane0, N = shard_range(1048576, 8, 7)
    assert (lane0, N) == (7 * 131072, 131072)
    traces = _traces(1024, mixed=True, seed=4)
    lens = [len(t) for t in traces]
    tid, off = lane_assignment(lane0, N, lens)
    env = make_env(META, traces, N, lane_id_base=lane0)
    assert env.effective_impl() == "jump"          # what `auto` picks at the per-rank size of the 8-GPU job
    env.reset(torch.from_numpy(tid), torch.from_numpy(off))
    rng = np.random.default_rng(44)
    br = np.array(LADDER)[None, :] * rng.uniform(0.8, 1.2, (V, 6))
    sz = br * L * rng.uniform(0.7, 1.3, (V, 6))
    mpd = A.MPD(V, L, MAX_BUFFER, START_UP, [A.Chunk(list(b), list(s)) for b, s in zip(br, sz)])
    ctl = A.BatchedMPCController(A.EnvPlayer(env, mpd=mpd, qoe=A.QOEMetric(0.3, 0.5, 0.0)), horizon=5)
    pick = np.unique(np.concatenate([np.arange(14), np.arange(N - 14, N), rng.integers(0, N, 1050)]))
    assert len(pick) >= 1024
    acts = _rollout_vs_oracle(oracle, env, ctl, traces, tid, off, br, sz, 0.5, 0.3, pick)
    assert len(np.unique(acts)) == 6                                  # every rate is exercised
    fin_t = env.observe_f64()["global_time"].cpu().numpy()
    assert (fin_t > np.array(lens)[tid] - off).any()                  # lanes wrapped around their trace (D7)


# ---------------------------------------------------------------------------------------------
# N > 1: two ranks, HIP shards, the product's collective
# ---------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_hip_shards_gather_equals_unsharded_hip_run(tmp_path):
    from abrsimulator_amd.sharding import lane_assignment
    TOTAL, WORLD, VV, SEED = 4096 + 64, 2, 12, 99
    traces = _traces(16, seed=9)
    meta = dict(META, video_length=VV)
    env = make_env(meta, traces, TOTAL, auto_reset=True)
    tid, off = lane_assignment(0, TOTAL, [len(t) for t in traces])
    env.reset(torch.from_numpy(tid), torch.from_numpy(off))
    ref = env.step_random(VV, SEED)
    out = str(tmp_path / "gathered.npz")
    port = _free_port()
    procs = []
    for r in range(WORLD):
        e = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r),
                 WORLD_SIZE=str(WORLD), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"),
                                       str(TOTAL), str(VV), str(SEED), out], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    g = np.load(out)
    assert int(g["n_collectives"]) == 1                      # ONE all-gather for (obs, reward)
    assert np.array_equal(g["obs"], ref["obs"][VV - 1].cpu().numpy())
    assert np.array_equal(g["reward"], ref["reward"].cpu().numpy())


# ---------------------------------------------------------------------------------------------
# ABI edges
# ---------------------------------------------------------------------------------------------
def test_reset_with_bad_trace_id_or_offset_freezes_the_lane():
    import abrsimulator_amd as A
    from abrsimulator_amd import _lib
    traces = _traces(4, seed=1)
    obs0 = {}
    for impl in ("split", "jump", "tick", "async"):
        env = make_env(dict(META, video_length=4), traces, 128, impl=impl)
        env.obs.fill_(-7.0)                                  # stale bytes must not survive in a frozen lane's row
        tid = (torch.arange(128, dtype=torch.int32) % 4).cuda()
        off = torch.zeros(128, dtype=torch.int32).cuda()
        tid[3] = 4; tid[9] = -1; off[17] = -5                # through the C ABI, past env.reset's checks
        _lib.check(env.lib.abr_env_reset(env._h, _lib.ptr(tid), _lib.ptr(off), None, _lib.ptr(env.obs),
                                         _lib.current_stream(env.device)))
        obs0[impl] = env.obs.clone()
        # a frozen lane reports the fresh-lane observation (tick 0: nothing downloaded, start-up running)
        assert torch.equal(obs0[impl][:, [3, 9, 17]].cpu(),
                           torch.tensor([0, -1, 0, 0, 0, 0, 0, 0.01], dtype=torch.float32)[:, None].expand(8, 3))
        a = torch.zeros(128, dtype=torch.int32).cuda()
        _, _, done = env.step(a)
        d = done.cpu().numpy()
        assert (d[[3, 9, 17]] == _lib.DONE_BADARG).all() and (np.delete(d, [3, 9, 17]) == 0).all()
        assert torch.equal(obs0[impl], obs0["split"]), impl
    with pytest.raises(ValueError):
        env.reset(tid.cpu(), off.cpu(), check=True)          # the Python front end refuses outright when asked to look
    # ... and by default takes the device's word: no host synchronisation in reset(), the lanes are frozen with BADARG
    env.reset(tid.cpu(), off.cpu())
    _, _, done = env.step(torch.zeros(128, dtype=torch.int32).cuda())
    assert (done.cpu().numpy()[[3, 9, 17]] == _lib.DONE_BADARG).all()


def test_lane_speeds_are_latched_until_a_full_reset(oracle):
    from abrsimulator_amd import _lib
    rng = np.random.default_rng(3)
    traces = _traces(4, seed=2)
    VV, N = 6, 256
    meta = dict(META, video_length=VV)
    tid = rng.integers(0, 4, N).astype(np.int32); off = rng.integers(0, 1000, N).astype(np.int32)
    actions = rng.integers(0, 6, (N, VV)).astype(np.int32)
    acts = torch.from_numpy(actions).cuda()
    cfg = oracle.env_cfg(LADDER, L, VV, MAX_BUFFER, START_UP, 1.0, WEIGHTS, 1.0)
    env = make_env(meta, traces, N)
    env.reset(torch.from_numpy(tid), torch.from_numpy(off))
    env.step(acts[:, 0].contiguous())
    speeds = torch.from_numpy(rng.choice([0.8, 1.0, 1.25], N)).cuda()
    _lib.check(env.lib.abr_env_set_lane_speeds(env._h, _lib.ptr(speeds)))
    # mid-episode: the running episodes keep speed 1.0 (bit-exact against the oracle at speed 1)
    for s in range(1, VV):
        env.step(acts[:, s].contiguous())
    steps, bw, fin, _ = oracle.env_batch(cfg, traces, tid, off, actions)
    assert np.array_equal(env.observe_f64()["buffer_level"].cpu().numpy(), fin["buffer_level"])
    assert np.array_equal(env.history()[1].cpu().numpy().T, bw)
    # a masked reset cannot adopt new speeds
    mask = torch.zeros(N, dtype=torch.uint8); mask[0] = 1
    with pytest.raises(_lib.AbrError):
        env.reset(torch.from_numpy(tid), torch.from_numpy(off), mask=mask)
    # the next full reset does
    env.reset(torch.from_numpy(tid), torch.from_numpy(off))
    for s in range(VV):
        env.step(acts[:, s].contiguous())
    steps, bw, fin, _ = oracle.env_batch(cfg, traces, tid, off, actions, speeds=speeds.cpu().numpy())
    assert np.array_equal(env.observe_f64()["play_time"].cpu().numpy(), fin["play_time"])
    assert np.array_equal(env.observe_f64()["buffer_level"].cpu().numpy(), fin["buffer_level"])


def test_every_reset_starts_a_new_policy_episode():
    traces = _traces(4, seed=5)
    VV, N, SEED = 5, 192, 31337
    env = make_env(dict(META, video_length=VV), traces, N)
    env.reset()
    a0 = env.step_random(VV, SEED)["actions"].cpu().numpy()
    assert np.array_equal(a0, np.stack([philox_action(SEED, np.arange(N), s, 0, 6) for s in range(VV)]))
    mask = torch.zeros(N, dtype=torch.uint8); mask[:64] = 1
    env.reset(mask=mask)                                      # lanes 0..63 start episode 1
    a1 = env.step_random(VV, SEED)["actions"].cpu().numpy()
    want = np.stack([philox_action(SEED, np.arange(64), s, 1, 6) for s in range(VV)])
    assert np.array_equal(a1[:, :64], want)
    assert (a1[:, 64:] == -1).all()                           # finished lanes stay finished
    env.reset()                                               # everyone: episodes 2 and 1
    a2 = env.step_random(VV, SEED)["actions"].cpu().numpy()
    assert np.array_equal(a2[:, :64], np.stack([philox_action(SEED, np.arange(64), s, 2, 6) for s in range(VV)]))
    assert np.array_equal(a2[:, 64:], np.stack([philox_action(SEED, np.arange(64, N), s, 1, 6) for s in range(VV)]))


def test_mpc_previous_bitrate_outside_the_ladder_is_no_decision():
    from test_mpc_gpu import _controller
    B, H, VV = 6, 5, 20
    br = np.tile(np.array(LADDER), (VV, 1)); sz = br * L
    prev = np.array([-7, -6, -1, 0, 5, 6, 100], np.int32)
    n = len(prev)
    ctl, ci = _controller(br, sz, L, MAX_BUFFER, 4.3, 1.0, 0.0, H, np.full(n, 3, np.int32), prev,
                          np.full(n, 5.0), np.full(n, 4.0), np.full(n, 2.0))
    a = ctl.next_bitrate(want_details=True).cpu().numpy()
    assert (a[[0, 5, 6]] == -1).all() and (a[1:5] >= 0).all()
    hn = ci.hist_n.cpu().numpy()
    assert (hn[[0, 5, 6]] == 4.0).all() and (hn[1:5] == 9.0).all()
    # Python's negative index: -6 is rate 0, -1 is rate 5
    assert float(ctl.last_J[1]) == float(ctl.last_J[3]) and float(ctl.last_J[2]) == float(ctl.last_J[4])


# ---------------------------------------------------------------------------------------------
# per-chunk ladders (8f rank 2, build-defined: the reference cannot run a list-MPD)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("impl", ["ring3", "split3", "split", "jump", "tick"])
def test_per_chunk_ladders_against_oracle(oracle, impl, tmp_path):
    import abrsimulator_amd as A
    rng = np.random.default_rng(66)
    VV, N = 14, 384
    br = np.array(LADDER)[None, :] * rng.uniform(0.7, 1.3, (VV, 6))
    traces = _traces(8, seed=6)
    # through the file format: one ladder per line (Simulator.py:71-76)
    mpdfile = str(tmp_path / "vbr.mpd")
    A.save_mpd_file(mpdfile, br)
    mpd = A.load_mpd_file(L, MAX_BUFFER, START_UP, mpdfile)
    assert not mpd.uniform() and np.array_equal(np.array(mpd.bitrate_table()), br)
    tid = rng.integers(0, 8, N).astype(np.int32); off = rng.integers(0, 1000, N).astype(np.int32)
    actions = rng.integers(0, 6, (N, VV)).astype(np.int32)
    from helpers import DIAG_IMPLS, diag_lib
    env = A.BatchedABREnv(mpd, A.QOEMetric(*WEIGHTS), A.NetworkInfo(1.0, traces), N, impl=impl,
                          library=diag_lib() if impl in DIAG_IMPLS else None)
    env.reset(torch.from_numpy(tid), torch.from_numpy(off))
    cfg = oracle.env_cfg(LADDER, L, VV, MAX_BUFFER, START_UP, 1.0, WEIGHTS, 1.0, br_table=br)
    steps, bw, fin, _ = oracle.env_batch(cfg, traces, tid, off, actions)
    acts = torch.from_numpy(actions).cuda()
    rew = np.zeros(N)
    # the variance term reads chunk s's OWN ladder row for a_s and chunk s-1's for a_(s-1)
    want_rew = oracle_rewards(steps, fin, actions, WEIGHTS, br_table=br)
    assert not np.array_equal(want_rew, oracle_rewards(steps, fin, actions, WEIGHTS, ladder=list(br[0])))
    for s in range(VV):
        f = env.observe_f64()
        for k in ("global_time", "buffer_level", "rebuffer_time", "start_up_time", "play_time"):
            assert np.array_equal(f[k].cpu().numpy(), steps[k][:, s]), (s, k)
        _, r, _ = env.step(acts[:, s].contiguous())
        assert np.array_equal(r.cpu().numpy(), want_rew[:, s]), (impl, s)
        rew += r.double().cpu().numpy()
    assert np.array_equal(env.history()[1].cpu().numpy().T, bw)
    q = env.episode_qoe().cpu().numpy()
    assert np.allclose(q, fin["qoe"], rtol=1e-10)
    # the per-step rewards still add up to calculate_qoe
    assert np.allclose(rew + 0.1 * fin["average_latency"], fin["qoe"], rtol=2e-5, atol=1e-3)
    # and it is not the single-ladder answer
    cfg1 = oracle.env_cfg(list(br[0]), L, VV, MAX_BUFFER, START_UP, 1.0, WEIGHTS, 1.0)
    _, bw1, _, _ = oracle.env_batch(cfg1, traces, tid, off, actions)
    assert not np.array_equal(bw1, bw)


def test_fused_random_rollout_with_per_chunk_ladders(oracle):
    import abrsimulator_amd as A
    rng = np.random.default_rng(67)
    VV, N, SEED = 10, 512, 5
    br = np.array(LADDER)[None, :] * rng.uniform(0.7, 1.3, (VV, 6))
    traces = _traces(8, seed=7)
    mpd = A.MPD(VV, L, MAX_BUFFER, START_UP, [A.Chunk(list(r)) for r in br])
    env = A.BatchedABREnv(mpd, A.QOEMetric(*WEIGHTS), A.NetworkInfo(1.0, traces), N, auto_reset=True)
    tid = (np.arange(N) % 8).astype(np.int32); off = rng.integers(0, 1000, N).astype(np.int32)
    env.reset(torch.from_numpy(tid), torch.from_numpy(off))
    out = env.step_random(2 * VV, SEED)
    acts = out["actions"].cpu().numpy()
    cfg = oracle.env_cfg(LADDER, L, VV, MAX_BUFFER, START_UP, 1.0, WEIGHTS, 1.0, br_table=br)
    for ep in range(2):
        a = acts[ep * VV:(ep + 1) * VV].T.copy()
        assert np.array_equal(a, np.stack([philox_action(SEED, np.arange(N), s, ep, 6) for s in range(VV)], 1))
        steps, bw, fin, _ = oracle.env_batch(cfg, traces, tid, off, a)
        o = out["obs"].cpu().numpy()[ep * VV:(ep + 1) * VV]
        for s in range(VV - 1):
            assert np.array_equal(o[s, 3], steps["buffer_level"][:, s + 1].astype(np.float32)), (ep, s)
            assert np.array_equal(o[s, 2], steps["last_bandwidth"][:, s + 1].astype(np.float32)), (ep, s)
        # rewards of the default (three-wave) fused kernel on per-chunk ladders, across the auto-reset
        assert np.array_equal(out["reward"].cpu().numpy()[ep * VV:(ep + 1) * VV].T,
                              oracle_rewards(steps, fin, a, WEIGHTS, br_table=br)), ep
    assert np.allclose(env.episode_qoe().cpu().numpy(), fin["qoe"], rtol=1e-10)


def test_bench_two_ranks_one_collective_per_launch():
    """bench.py's N > 1 path end to end: two ranks (both on this box's one GPU, gloo standing in for
    RCCL) through torch.distributed.run exactly as the driver launches it; one packed all-gather per
    launch, in the fused mode and in the one-launch-per-decision mode (configs[3] literally)."""
    import json
    for fuse, steps, warm in ((48, 96, 48), (1, 24, 8)):
        env = dict(os.environ, ABR_BENCH_ONE_DEVICE="1", ABR_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
               os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", str(warm),
               "--fuse", str(fuse), "--total-lanes", "8192", "--no-cpu-baseline", "--min-timed-steps", "1"]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["total_lanes"] == 8192
        assert line["config"]["lanes_per_gpu"] == 4096 and line["config"]["fuse"] == fuse
        launches = (2 * steps + warm) // fuse      # the warm-up, one region with HIP events around the launches, one without
        assert line["config"]["collective"].startswith("1 all_gather_into_tensor per launch")
        assert line["config"]["collective"].endswith(f"issued {launches}x"), line["config"]["collective"]
        assert line["value"] > 0 and "secondary" not in line and line["cpu_baseline"] is None


def test_bench_rccl_code_path_with_one_rank():
    """The RCCL backend itself ("nccl" on ROCm) cannot host two ranks on one GPU, but a ONE-rank
    group can: communicator creation, all_gather_into_tensor on the side stream, events, barrier and
    all_reduce all run through RCCL (ABR_BENCH_FORCE_DIST=1)."""
    import json
    env = dict(os.environ, ABR_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("ABR_BENCH_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "96", "--warmup", "48",
           "--lanes-per-gpu", "8192", "--no-cpu-baseline", "--no-secondary", "--min-timed-steps", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert "backend nccl" in line["config"]["collective"] and line["config"]["collective"].endswith("issued 5x")   # 1 warm-up launch + 2 regions x 2
    assert line["value"] > 0
    # the scaling-curve job (BASELINE.json configs[3]) rides in the same line: 1 048 576 lanes on this one rank
    st = line["strong_1048576"]
    assert st["total_lanes"] == 1048576 and st["lanes_per_gpu"] == 1048576 and st["scaling"] == "strong"
    assert st["value"] > line["value"] and st["collective"].endswith("issued 5x")
    # the headline's regions carry no instrumentation; the regions with HIP events around every launch are reported beside them
    assert len(line["repeat_seconds"]) == len(line["repeat_seconds_with_events"]) == line["repeats"] and line["value_with_events"] > 0


def test_sharded_env_over_rccl_with_one_rank():
    """The package's ShardedABREnv over the RCCL backend with a ONE-rank group, in a fresh process: random-policy, scripted and
    MPC-driven launches, double-buffered slabs, the all-gather on the side stream; gathered == local == an unsharded env."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sharded_rccl_one_rank.py")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "sharded one-rank RCCL ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_bench_default_line_carries_both_scaling_curves_two_ranks():
    """What the driver's SCALE run launches (bare `bench.py --gpus N`), rehearsed with two gloo ranks on this
    box's one GPU: the headline stays weak scaling at the per-GPU lane count, and the strong-scaling job is
    split over the ranks (its size shrunk for the rehearsal)."""
    import json
    env = dict(os.environ, ABR_BENCH_ONE_DEVICE="1", ABR_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0",
               ABR_BENCH_STRONG_TOTAL="16384")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--lanes-per-gpu", "4096", "--min-timed-steps", "40", "--allow-overrides"]     # (ABR_BENCH_STRONG_TOTAL changes the workload)
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["total_lanes"] == 8192
    # one timed region of 20 decisions = two launches of 10, so that the first all-gather has a launch to hide behind
    assert line["config"]["fuse"] == 10 and line["steps"] == 20
    st = line["strong_1048576"]
    assert st["scaling"] == "strong" and st["total_lanes"] == 16384 and st["lanes_per_gpu"] == 8192
    assert st["n_gpus"] == 2 and st["value"] > 0 and st["collective"].startswith("1 all_gather_into_tensor")
    # what makes the first measured curve readable: the launch shape of every value is stated, and a control block
    # separates the cost of the collective from the cost of cutting the N = 1 launch in two
    assert line["config"]["launches_per_region"] == 2 and line["config"]["fuse_at_n1"] == 20
    for blk in (line["control"], st["control"]):
        a, b = blk["same_launches_no_gather"], blk["n1_launch_shape_no_gather"]
        assert a["fuse"] == 10 and a["launches_per_region"] == 2 and a["value"] > 0
        assert b["fuse"] == 20 and b["launches_per_region"] == 1 and b["value"] > 0
    assert st["launches_per_region"] == 2
    # every override in force is on record in the line, and the one that changes the workload needed --allow-overrides
    assert line["config"]["overrides"]["ABR_BENCH_STRONG_TOTAL"] == "16384" and "ABR_BENCH_BACKEND" in line["config"]["overrides"]
    cmd2 = [c for c in cmd if c != "--allow-overrides"]
    refused = subprocess.run(cmd2, env=env, capture_output=True, text=True, timeout=300)
    assert refused.returncode != 0 and "allow-overrides" in refused.stderr and not refused.stdout.strip()
