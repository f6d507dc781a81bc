"""GPU parity of the env kernels (K1 step, K2 reset, K4 episode_qoe) through the C ABI.

Bar: bit-exact float64 for every decision-feeding quantity and all clocks;
average_latency within 1e-9 relative (it is carried as an exact integer sum of
tick indices instead of the reference's per-tick multiply/divide recurrence --
see csrc/abr_env.hip: lane_avg_latency; BASELINE.json's bar is 1e-5); float32
observations equal float32(float64 golden).
"""
import numpy as np
import pytest
import torch

from conftest import ENV_GOLDENS, load_golden
from helpers import F64_EXACT, golden_rewards, make_env, oracle_rewards, philox_action

pytestmark = pytest.mark.gpu

LAT_RTOL = 1e-9


def _cmp_step(f64, g, s, name):
    for k in F64_EXACT:
        got = f64[k].cpu().numpy()
        assert np.array_equal(got, g[k][:, s]), (name, s, k, got[:4], g[k][:4, s])
    lat = f64["average_latency"].cpu().numpy()
    ref = g["average_latency"][:, s]
    assert np.allclose(lat, ref, rtol=LAT_RTOL, atol=1e-12), (name, s, lat[:4], ref[:4])
    assert np.array_equal(f64["chunk_id"].cpu().numpy().astype(np.int32), g["chunk_id"][:, s])
    assert np.array_equal(f64["play_id"].cpu().numpy().astype(np.int32), g["play_id"][:, s])
    assert np.array_equal(f64["last_bitrate"].cpu().numpy().astype(np.int32), g["arg_last_bitrate"][:, s])
    assert np.array_equal(f64["last_bandwidth"].cpu().numpy(), g["arg_last_bandwidth"][:, s])
    fl = f64["flags"].cpu().numpy().astype(np.int32)
    assert np.array_equal(fl & 1, g["start_up"][:, s])
    assert np.array_equal((fl >> 1) & 1, g["buffer_empty"][:, s])
    assert np.array_equal((fl >> 2) & 1, g["buffer_full"][:, s])


IMPLS = ["split", "split3", "ring3", "pair3", "jump", "tick"]   # role-split (2 / 3 waves per 64 lanes; ring3: three waves coupled by LDS rings), one-thread-per-lane, tick-by-tick cross-check


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("name", ENV_GOLDENS)
def test_step_matches_reference_goldens(name, impl):
    """Replays the golden actions step by step; compares the full float64
    observation at EVERY call site with what the reference's run() frame held."""
    m, g = load_golden(name)
    N, V = g["actions"].shape
    env = make_env(m, g["traces"], N, impl=impl)
    obs = env.reset(torch.from_numpy(g["trace_id"]), torch.from_numpy(g["offset"]))
    acts = torch.from_numpy(g["actions"]).cuda()
    rew_sum = np.zeros(N)
    want_rew = golden_rewards(m, g)                # float32 of the reference-derived value, [N, V]
    for s in range(V):
        _cmp_step(env.observe_f64(), g, s, name)
        o = obs.cpu().numpy()
        assert np.array_equal(o[3], g["buffer_level"][:, s].astype(np.float32))
        assert np.array_equal(o[4], g["global_time"][:, s].astype(np.float32))
        assert np.array_equal(o[2], g["arg_last_bandwidth"][:, s].astype(np.float32))
        assert np.array_equal(o[0], g["chunk_id"][:, s].astype(np.float32))
        obs, rew, done = env.step(acts[:, s].contiguous())
        # the per-step linear QoE reward, element by element (Simulator.py:79-86 split at :155)
        assert np.array_equal(rew.cpu().numpy(), want_rew[:, s]), (name, impl, s, rew[:4], want_rew[:4, s])
        rew_sum += rew.double().cpu().numpy()
        d = done.cpu().numpy()
        assert (d == (1 if s == V - 1 else 0)).all(), (s, d)
    # terminal state = the frame calculate_qoe was called from
    f = env.observe_f64()
    for k in ["global_time", "rebuffer_time", "start_up_time", "play_time", "buffer_level"]:
        assert np.array_equal(f[k].cpu().numpy(), g["final_" + k]), k
    assert np.allclose(f["average_latency"].cpu().numpy(), g["final_average_latency"], rtol=LAT_RTOL)
    assert np.array_equal(f["play_id"].cpu().numpy().astype(np.int32), g["final_play_id"])
    # previous_bandwidths / previous_bitrates lists
    ah, bh = env.history()
    assert np.array_equal(bh.cpu().numpy().T, g["final_bandwidths"])
    assert np.array_equal(ah.cpu().numpy().T.astype(np.int32), g["actions"])
    # K4: calculate_qoe; and the per-step rewards add up to it
    qoe = env.episode_qoe().cpu().numpy()
    assert np.allclose(qoe, g["final_qoe"], rtol=1e-10, atol=1e-12)
    wl = m["weights"][3]
    assert np.allclose(rew_sum + wl * g["final_average_latency"], g["final_qoe"], rtol=2e-5, atol=1e-4)
    # stepping a finished env is a no-op that reports done again
    obs2, rew2, done2 = env.step(acts[:, 0].contiguous())
    assert (done2.cpu().numpy() == 1).all() and (rew2.cpu().numpy() == 0).all()


def _random_case(seed, N, V=12, L=4.0, interval=1.0, n_traces=7, ragged=False, max_buffer=20.0,
                 start_up=8.0, bw=(0.2, 6.0), ladder=(0.3, 0.75, 1.2, 1.85, 2.85, 4.3)):
    rng = np.random.default_rng(seed)
    if ragged:
        lens = rng.integers(300, 3001, n_traces)
    else:
        lens = np.full(n_traces, 1000)
    traces = [rng.uniform(bw[0], bw[1], l).astype(np.float32).astype(np.float64) for l in lens]
    meta = dict(ladder=list(ladder), chunk_length=L, video_length=V, max_buffer=max_buffer,
                start_up_length=start_up, interval=interval, weights=[4.3, 1, 1, 0.1], speed=1.0)
    trace_id = rng.integers(0, n_traces, N).astype(np.int32)
    offset = np.array([rng.integers(0, lens[t]) for t in trace_id], np.int32)
    actions = rng.integers(0, len(ladder), (N, V)).astype(np.int32)
    return meta, traces, trace_id, offset, actions


@pytest.mark.parametrize("case", [
    dict(seed=1, N=1000), dict(seed=2, N=777, ragged=True), dict(seed=3, N=512, L=1.0, start_up=2.0),
    dict(seed=4, N=300, interval=0.3, bw=(0.1, 1.5)), dict(seed=5, N=256, max_buffer=5.0, start_up=4.0),
    dict(seed=6, N=1, V=3), dict(seed=7, N=65, L=2.5, interval=0.7, V=9)])
@pytest.mark.parametrize("impl", IMPLS)
def test_step_matches_oracle_seeded(oracle, case, impl):
    """Same seeded inputs through the HIP path and the C oracle (wrap-around of
    short ragged traces included, where the reference itself would raise)."""
    meta, traces, trace_id, offset, actions = _random_case(**case)
    cfg = oracle.env_cfg(meta["ladder"], meta["chunk_length"], meta["video_length"], meta["max_buffer"],
                         meta["start_up_length"], meta["interval"], meta["weights"], 1.0)
    steps, bw, fin, _ = oracle.env_batch(cfg, traces, trace_id, offset, actions)
    N, V = actions.shape
    env = make_env(meta, traces, N, impl=impl)
    env.reset(torch.from_numpy(trace_id), torch.from_numpy(offset))
    acts = torch.from_numpy(actions).cuda()
    want_rew = oracle_rewards(steps, fin, actions, meta["weights"], ladder=meta["ladder"])
    for s in range(V):
        f = env.observe_f64()
        for k in F64_EXACT:
            assert np.array_equal(f[k].cpu().numpy(), steps[k][:, s]), (s, k)
        assert np.allclose(f["average_latency"].cpu().numpy(), steps["average_latency"][:, s],
                           rtol=LAT_RTOL, atol=1e-12)
        assert np.array_equal(f["last_bandwidth"].cpu().numpy(), steps["last_bandwidth"][:, s])
        _, rew, _ = env.step(acts[:, s].contiguous())
        assert np.array_equal(rew.cpu().numpy(), want_rew[:, s]), (impl, s)
    assert np.allclose(env.episode_qoe().cpu().numpy(), fin["qoe"], rtol=1e-10)
    assert np.array_equal(env.observe_f64()["global_time"].cpu().numpy(), fin["global_time"])


def test_step_random_fused_equals_stepwise_and_oracle(oracle):
    """The fused random-policy rollout (K1 in MODE 2) == step-by-step with the
    same philox actions == the oracle replaying those actions."""
    meta, traces, trace_id, offset, _ = _random_case(seed=11, N=640, V=10)
    N, V, seed = 640, 10, 0x1234ABCD5678
    env = make_env(meta, traces, N)
    env.reset(torch.from_numpy(trace_id), torch.from_numpy(offset))
    out = env.step_random(V, seed)
    acts = out["actions"].cpu().numpy()            # [V, N]
    want = np.stack([philox_action(seed, np.arange(N), s, 0, 6) for s in range(V)])
    assert np.array_equal(acts, want)
    cfg = oracle.env_cfg(meta["ladder"], meta["chunk_length"], V, meta["max_buffer"],
                         meta["start_up_length"], meta["interval"], meta["weights"], 1.0)
    steps, bw, fin, _ = oracle.env_batch(cfg, traces, trace_id, offset, acts.T.copy())
    obs = out["obs"].cpu().numpy()                 # obs[s] = observation AFTER step s
    for s in range(V - 1):
        assert np.array_equal(obs[s, 3], steps["buffer_level"][:, s + 1].astype(np.float32))
        assert np.array_equal(obs[s, 4], steps["global_time"][:, s + 1].astype(np.float32))
        assert np.array_equal(obs[s, 2], steps["last_bandwidth"][:, s + 1].astype(np.float32))
    assert (out["done"].cpu().numpy()[:-1] == 0).all() and (out["done"].cpu().numpy()[-1] == 1).all()
    assert np.allclose(env.episode_qoe().cpu().numpy(), fin["qoe"], rtol=1e-10)
    assert np.array_equal(out["reward"].cpu().numpy().T,
                          oracle_rewards(steps, fin, acts.T, meta["weights"], ladder=meta["ladder"]))
    # step-by-step twin, on the OTHER implementation
    env2 = make_env(meta, traces, N, impl="tick")
    env2.reset(torch.from_numpy(trace_id), torch.from_numpy(offset))
    for s in range(V):
        o, r, d = env2.step(out["actions"][s].contiguous())
        assert torch.equal(o, out["obs"][s]) and torch.equal(r, out["reward"][s])
        assert torch.equal(d, out["done"][s])


def test_auto_reset_and_lane_id_base(oracle):
    """auto_reset re-arms finished lanes inside the step; a shard with
    lane_id_base reproduces the matching slice of the unsharded run."""
    meta, traces, trace_id, offset, _ = _random_case(seed=21, N=256, V=5)
    N, V, seed = 256, 5, 99
    env = make_env(meta, traces, N, auto_reset=True)
    env.reset(torch.from_numpy(trace_id), torch.from_numpy(offset))
    out = env.step_random(2 * V + 2, seed)
    done = out["done"].cpu().numpy()
    assert (done[V - 1] == 1).all() and (done[2 * V - 1] == 1).all()
    assert done.sum() == 2 * N
    acts = out["actions"].cpu().numpy()
    # episode 1 restarts from the same trace/offset with fresh philox counters
    want1 = np.stack([philox_action(seed, np.arange(N), s, 1, 6) for s in range(V)])
    assert np.array_equal(acts[V:2 * V], want1)
    cfg = oracle.env_cfg(meta["ladder"], meta["chunk_length"], V, meta["max_buffer"],
                         meta["start_up_length"], meta["interval"], meta["weights"], 1.0)
    steps, bw, fin, _ = oracle.env_batch(cfg, traces, trace_id, offset, want1.T.copy())
    obs = out["obs"].cpu().numpy()
    # rewards element by element ACROSS the episode boundary: the terminal step of episode 0 closes on
    # that episode's final timers, the first step of episode 1 starts its deltas from 0 again, and the
    # variance term does not reach back into the finished episode
    rew = out["reward"].cpu().numpy()
    steps0, _, fin0, _ = oracle.env_batch(cfg, traces, trace_id, offset, acts[:V].T.copy())
    assert np.array_equal(rew[:V].T, oracle_rewards(steps0, fin0, acts[:V].T, meta["weights"], ladder=meta["ladder"]))
    assert np.array_equal(rew[V:2 * V].T, oracle_rewards(steps, fin, want1.T, meta["weights"], ladder=meta["ladder"]))
    want2 = np.stack([philox_action(seed, np.arange(N), s, 2, 6) for s in range(V)])
    steps2, _, fin2, _ = oracle.env_batch(cfg, traces, trace_id, offset, want2.T.copy())
    assert np.array_equal(rew[2 * V:].T, oracle_rewards(steps2, fin2, want2.T, meta["weights"],
                                                        ladder=meta["ladder"])[:, :2])
    # obs returned with done is the new episode's first call site
    assert np.array_equal(obs[V - 1, 4], steps["global_time"][:, 0].astype(np.float32))
    assert np.array_equal(obs[V, 3], steps["buffer_level"][:, 1].astype(np.float32))
    # after 2V steps the last finished episode is episode 1
    env_b = make_env(meta, traces, N, auto_reset=True)
    env_b.reset(torch.from_numpy(trace_id), torch.from_numpy(offset))
    env_b.step_random(2 * V, seed)
    assert np.allclose(env_b.episode_qoe().cpu().numpy(), fin["qoe"], rtol=1e-10)
    # shard [100, 200) with lane_id_base = 100
    sh = make_env(meta, traces, 100, lane_id_base=100)
    sh.reset(torch.from_numpy(trace_id[100:200].copy()), torch.from_numpy(offset[100:200].copy()))
    o2 = sh.step_random(V, seed)
    assert torch.equal(o2["actions"], out["actions"][:V, 100:200])
    assert torch.equal(o2["obs"][:V - 1], out["obs"][:V - 1, :, 100:200])


def test_bad_action_and_masked_reset_and_checkpoint():
    meta, traces, trace_id, offset, actions = _random_case(seed=31, N=128, V=6)
    env = make_env(meta, traces, 128)
    env.reset(torch.from_numpy(trace_id), torch.from_numpy(offset))
    a = torch.from_numpy(actions[:, 0].copy()).cuda()
    a[5] = 6       # out of range: the reference would raise IndexError (Simulator.py:156)
    a[7] = -1
    o, r, d = env.step(a)
    d = d.cpu().numpy()
    assert d[5] == 4 and d[7] == 4 and (np.delete(d, [5, 7]) == 0).all()
    # checkpoint, advance, restore, advance again -> identical
    sd = env.state_dict()
    a1 = torch.from_numpy(actions[:, 1].copy()).cuda()
    o1 = env.step(a1)[0].clone()
    env.load_state_dict(sd)
    o2 = env.step(a1)[0].clone()
    assert torch.equal(o1, o2)
    # masked reset only touches the chosen lanes
    mask = torch.zeros(128, dtype=torch.uint8); mask[5] = 1; mask[7] = 1
    before = env.observe_f64()["tick"].clone()
    env.reset(torch.from_numpy(trace_id), torch.from_numpy(offset), mask=mask)
    after = env.observe_f64()["tick"]
    assert after[5] == 401 and after[7] == 401
    keep = torch.ones(128, dtype=torch.bool); keep[5] = False; keep[7] = False
    assert torch.equal(before[keep.cuda()], after[keep.cuda()])


def test_full_size_properties(oracle):
    """BASELINE.json size (65 536 lanes): size-independent properties on every lane, and an
    exact oracle replay of 2 048 lanes sampled across the whole index range."""
    N, V = 65536, 48
    rng = np.random.default_rng(0)
    traces = [rng.uniform(0.2, 6.0, 1000).astype(np.float32).astype(np.float64) for _ in range(1024)]
    meta = dict(ladder=[0.3, 0.75, 1.2, 1.85, 2.85, 4.3], chunk_length=4, video_length=V,
                max_buffer=20, start_up_length=8, interval=1.0, weights=[4.3, 1, 1, 0.1], speed=1.0)
    env = make_env(meta, traces, N)
    tid = torch.arange(N, dtype=torch.int32) % 1024
    off = torch.from_numpy(rng.integers(0, 1000, N).astype(np.int32))
    env.reset(tid, off)
    out = env.step_random(V, 7)
    obs = out["obs"]
    done = out["done"]
    assert int(done[-1].sum()) == N and int(done[:-1].sum()) == 0
    # clocks are monotone, chunk ids count up by one, buffer within [0, max_buffer + L]
    assert bool((obs[1:, 4] >= obs[:-1, 4])[: V - 2].all())
    assert torch.equal(obs[: V - 1, 0], torch.arange(1, V, device="cuda", dtype=torch.float32)[:, None].expand(V - 1, N))
    assert float(obs[:, 3].min()) >= 0.0 and float(obs[:, 3].max()) <= 24.0
    # live-stream gating: chunk c cannot start before (c+1) * L
    t = obs[: V - 1, 4]
    lower = (torch.arange(1, V, device="cuda", dtype=torch.float32)[:, None] + 1) * 4.0
    assert bool((t >= lower - 1e-3).all())
    # two lanes with identical (trace, offset, lane id) inputs agree: determinism across launches
    env2 = make_env(meta, traces, N)
    env2.reset(tid, off)
    out2 = env2.step_random(V, 7)
    assert torch.equal(out2["obs"], obs) and torch.equal(out2["reward"], out["reward"])
    # QoE identity: sum(reward) + wl * average_latency == calculate_qoe
    q = env.episode_qoe()
    lat = env.observe_f64()["average_latency"]
    assert torch.allclose(out["reward"].double().sum(0) + 0.1 * lat, q, rtol=2e-5, atol=1e-3)
    # exact replay of a sample of lanes (first, last, and random ones in between)
    pick = np.unique(np.concatenate([[0, 1, 63, 64, N - 1, N - 64], rng.integers(0, N, 2048)]))
    acts = out2["actions"].cpu().numpy()[:, pick].T.copy()
    cfg = oracle.env_cfg(meta["ladder"], 4.0, V, 20.0, 8.0, 1.0, meta["weights"], 1.0)
    steps, bw, fin, _ = oracle.env_batch(cfg, traces, tid.numpy()[pick], off.numpy()[pick], acts)
    o = obs.cpu().numpy()[:, :, pick]               # o[s] = observation after step s
    for s in range(V - 1):
        assert np.array_equal(o[s, 3], steps["buffer_level"][:, s + 1].astype(np.float32))
        assert np.array_equal(o[s, 4], steps["global_time"][:, s + 1].astype(np.float32))
        assert np.array_equal(o[s, 2], steps["last_bandwidth"][:, s + 1].astype(np.float32))
        assert np.array_equal(o[s, 6], steps["rebuffer_time"][:, s + 1].astype(np.float32))
    # every reward element of the sampled lanes == float32 of the oracle-derived value
    assert np.array_equal(out["reward"].cpu().numpy()[:, pick].T,
                          oracle_rewards(steps, fin, acts, meta["weights"], ladder=meta["ladder"]))
    ah, bh = env.history()
    assert np.array_equal(bh.cpu().numpy()[:, pick].T, bw)          # float64 throughputs, bit-exact
    assert np.allclose(q.cpu().numpy()[pick], fin["qoe"], rtol=1e-10)


def test_simulator_class_runs_reference_style_script(tmp_path):
    """The reference's own call sequence (Simulator.py:46-93) on files on disk, with a
    plugin that keeps get_next_bitrate's signature; result == the golden run() values."""
    import abrsimulator_amd as A
    m, g = load_golden("env_const_policy")
    N, V = g["actions"].shape
    paths = []
    for t, tr in enumerate(g["traces"]):
        p = str(tmp_path / f"trace{t}.txt")
        A.save_trace_file(p, tr)
        paths.append(p)
    mpdfile = str(tmp_path / "video.mpd")
    A.save_mpd_file(mpdfile, [m["ladder"]] * V)
    acts = torch.from_numpy(g["actions"]).cuda()
    seen = []

    class Replay:
        def get_next_bitrate(self, chunk_id, previous_bitrates, previous_bandwidths, buffer_level):
            c = int(chunk_id[0])
            seen.append((c, buffer_level.clone(), previous_bandwidths[:c].clone()))
            return acts[:, c]

    class Speed:
        def get_next_speed(self):
            return m["speed"]

    sim = A.Simulator(Replay(), Speed(), n_lanes=N)
    sim.set_qoe_metric(A.QOEMetric(*m["weights"]))
    sim.set_network_info(m["interval"], paths)
    sim.set_mpd(m["chunk_length"], m["max_buffer"], m["start_up_length"], mpdfile)
    sim.set_lanes(torch.from_numpy(g["trace_id"]), torch.from_numpy(g["offset"]))
    qoe = sim.run().cpu().numpy()
    assert np.allclose(qoe, g["final_qoe"], rtol=1e-10)
    for c, buf, bws in seen:
        assert np.array_equal(buf.cpu().numpy(), g["buffer_level"][:, c])
        if c:
            assert np.array_equal(bws.cpu().numpy().T, g["final_bandwidths"][:, :c])


def test_per_lane_speeds(oracle):
    """8f rank 3: one constant play speed per lane.  play_time (carried by the exact chain),
    buffer_level, the clocks, play_id/play_length and the episode QoE against the oracle run
    with the same per-lane speeds."""
    meta, traces, trace_id, offset, actions = _random_case(seed=41, N=700, V=10, bw=(0.5, 7.0))
    N, V = actions.shape
    speeds = np.random.default_rng(41).choice([0.75, 0.8, 1.0, 1.1, 1.25, 1.3, 0.9173], N)
    cfg = oracle.env_cfg(meta["ladder"], meta["chunk_length"], V, meta["max_buffer"],
                         meta["start_up_length"], meta["interval"], meta["weights"], 1.0)
    steps, bw, fin, _ = oracle.env_batch(cfg, traces, trace_id, offset, actions, speeds=speeds)
    want_rew = oracle_rewards(steps, fin, actions, meta["weights"], ladder=meta["ladder"])
    env = make_env(dict(meta, speed=torch.from_numpy(speeds)), traces, N)
    obs = env.reset(torch.from_numpy(trace_id), torch.from_numpy(offset))
    acts = torch.from_numpy(actions).cuda()
    for s in range(V):
        f = env.observe_f64()
        for k in F64_EXACT:
            assert np.array_equal(f[k].cpu().numpy(), steps[k][:, s]), (s, k)
        assert np.array_equal(f["play_id"].cpu().numpy().astype(np.int32), steps["play_id"][:, s])
        assert np.allclose(f["average_latency"].cpu().numpy(), steps["average_latency"][:, s],
                           rtol=LAT_RTOL, atol=1e-12)
        assert np.array_equal(obs.cpu().numpy()[5], steps["play_time"][:, s].astype(np.float32))
        obs, rew, _ = env.step(acts[:, s].contiguous())
        assert np.array_equal(rew.cpu().numpy(), want_rew[:, s]), s
    assert np.allclose(env.episode_qoe().cpu().numpy(), fin["qoe"], rtol=1e-10)
    assert np.array_equal(env.observe_f64()["play_time"].cpu().numpy(), fin["play_time"])
    # the tick-by-tick kernels take one speed only
    import abrsimulator_amd as A
    with pytest.raises(A._lib.AbrError):
        make_env(dict(meta, speed=torch.from_numpy(speeds)), traces, N, impl="tick")


@pytest.mark.parametrize("impl", ["ring3", "split3", "split", "jump"])
def test_speed_schedule_matches_reference_golden(impl):
    """8f rank 3, second half: the play speed is re-read at every played chunk
    (Simulator.py:176-177).  The fixture is the reference driven by a scripted speed
    controller; every call-site frame is compared, incl. play_id / play_length."""
    m, g = load_golden("env_speed_schedule")
    N, V = g["actions"].shape
    sched = torch.from_numpy(g["speed_sched"].T.copy())          # [rows, N]
    env = make_env(dict(m, speed=sched), g["traces"], N, impl=impl)
    obs = env.reset(torch.from_numpy(g["trace_id"]), torch.from_numpy(g["offset"]))
    acts = torch.from_numpy(g["actions"]).cuda()
    want_rew = golden_rewards(m, g)
    for s in range(V):
        _cmp_step(env.observe_f64(), g, s, "env_speed_schedule")
        assert np.array_equal(obs.cpu().numpy()[5], g["play_time"][:, s].astype(np.float32))
        obs, rew, _ = env.step(acts[:, s].contiguous())
        assert np.array_equal(rew.cpu().numpy(), want_rew[:, s]), (impl, s)
    # ... and the same episode in ONE fused scripted call
    env2 = make_env(dict(m, speed=sched), g["traces"], N, impl=impl)
    env2.reset(torch.from_numpy(g["trace_id"]), torch.from_numpy(g["offset"]))
    out2 = env2.step_script(torch.from_numpy(g["actions"].T.copy()))
    assert np.array_equal(out2["reward"].cpu().numpy().T, want_rew), impl
    f = env.observe_f64()
    for k in ["global_time", "rebuffer_time", "start_up_time", "play_time", "buffer_level"]:
        assert np.array_equal(f[k].cpu().numpy(), g["final_" + k]), k
    assert np.allclose(f["average_latency"].cpu().numpy(), g["final_average_latency"], rtol=LAT_RTOL)
    assert np.array_equal(f["play_id"].cpu().numpy().astype(np.int32), g["final_play_id"])
    assert np.allclose(env.episode_qoe().cpu().numpy(), g["final_qoe"], rtol=1e-10)
    assert np.array_equal(env.history()[1].cpu().numpy().T, g["final_bandwidths"])


def test_speed_schedule_fused_and_simulator_class(oracle):
    """Schedules under the fused random rollout with auto-reset (the schedule restarts with the
    episode), and through the Simulator class with a speed controller that answers tensors."""
    import abrsimulator_amd as A
    meta, traces, trace_id, offset, _ = _random_case(seed=61, N=320, V=9, max_buffer=12.0)
    N, V, SEED = 320, 9, 77
    rng = np.random.default_rng(61)
    sched = rng.choice([0.6, 0.8, 1.0, 1.2, 1.5, 1.9], (5, N))
    env = make_env(dict(meta, speed=torch.from_numpy(sched)), traces, N, auto_reset=True)
    env.reset(torch.from_numpy(trace_id), torch.from_numpy(offset))
    out = env.step_random(2 * V, SEED)
    cfg = oracle.env_cfg(meta["ladder"], meta["chunk_length"], V, meta["max_buffer"],
                         meta["start_up_length"], meta["interval"], meta["weights"], 1.0)
    for ep in range(2):
        a = out["actions"].cpu().numpy()[ep * V:(ep + 1) * V].T.copy()
        steps, bw, fin, _ = oracle.env_batch(cfg, traces, trace_id, offset, a, speeds=sched.T.copy())
        o = out["obs"].cpu().numpy()[ep * V:(ep + 1) * V]
        for s in range(V - 1):
            assert np.array_equal(o[s, 3], steps["buffer_level"][:, s + 1].astype(np.float32)), (ep, s)
            assert np.array_equal(o[s, 5], steps["play_time"][:, s + 1].astype(np.float32)), (ep, s)
        assert np.array_equal(out["reward"].cpu().numpy()[ep * V:(ep + 1) * V].T,
                              oracle_rewards(steps, fin, a, meta["weights"], ladder=meta["ladder"])), ep
    assert np.allclose(env.episode_qoe().cpu().numpy(), fin["qoe"], rtol=1e-10)

    class Replay:
        def get_next_bitrate(self, chunk_id, previous_bitrates, previous_bandwidths, buffer_level):
            return acts[:, int(chunk_id[0])]

    class Speed:
        def __init__(self):
            self.calls = 0

        def get_next_speed(self):
            self.calls += 1
            return torch.from_numpy(sched[min(self.calls - 1, 4)])

    acts = torch.from_numpy(a).cuda()
    sim = A.Simulator(Replay(), Speed(), n_lanes=N)
    sim.set_qoe_metric(A.QOEMetric(*meta["weights"]))
    sim.set_network_info(meta["interval"], A.NetworkInfo(meta["interval"], traces))
    sim.set_mpd(meta["chunk_length"], meta["max_buffer"], meta["start_up_length"],
                A.MPD(V, meta["chunk_length"], meta["max_buffer"], meta["start_up_length"], A.Chunk(meta["ladder"])))
    sim.set_lanes(torch.from_numpy(trace_id), torch.from_numpy(offset))
    assert np.allclose(sim.run().cpu().numpy(), fin["qoe"], rtol=1e-10)


@pytest.mark.parametrize("seed", range(12))
def test_random_configurations_against_oracle(oracle, seed):
    """Config-space fuzz on the device (the CPU twin of this test runs 60 seeds through the
    host build of the same lane logic): chunk lengths, trace intervals shorter than the
    prologue / longer than a chunk / non-representable, ladders of 2-8 rates, buffer limits,
    start_up_length 0, speeds, ragged traces with wrap-around."""
    from test_lane_jump_cpu import _random_config
    rng = np.random.default_rng(1000 + seed)
    meta, (lo, hi) = _random_config(rng)
    n_traces, N = 6, 200
    lens = rng.integers(40, 3000, n_traces)
    traces = [rng.uniform(lo, hi, l).astype(np.float32).astype(np.float64) for l in lens]
    trace_id = rng.integers(0, n_traces, N).astype(np.int32)
    offset = np.array([rng.integers(0, lens[t]) for t in trace_id], np.int32)
    V = meta["video_length"]
    actions = rng.integers(0, len(meta["ladder"]), (N, V)).astype(np.int32)
    cfg = oracle.env_cfg(meta["ladder"], meta["chunk_length"], V, meta["max_buffer"],
                         meta["start_up_length"], meta["interval"], meta["weights"], meta["speed"])
    steps, bw, fin, _ = oracle.env_batch(cfg, traces, trace_id, offset, actions, max_ticks=4_000_000)
    want_rew = oracle_rewards(steps, fin, actions, meta["weights"], ladder=meta["ladder"])
    for impl in IMPLS:
        env = make_env(meta, traces, N, impl=impl, max_ticks=int(fin["ticks"].max()) + 1000)
        env.reset(torch.from_numpy(trace_id), torch.from_numpy(offset))
        acts = torch.from_numpy(actions).cuda()
        for s in range(V):
            f = env.observe_f64()
            for k in F64_EXACT:
                assert np.array_equal(f[k].cpu().numpy(), steps[k][:, s]), (impl, s, k)
            _, rew, _ = env.step(acts[:, s].contiguous())
            assert np.array_equal(rew.cpu().numpy(), want_rew[:, s]), (impl, s)
        assert np.array_equal(env.history()[1].cpu().numpy().T, bw), impl
        assert np.allclose(env.episode_qoe().cpu().numpy(), fin["qoe"], rtol=1e-10), impl


def test_c_host_program(tmp_path):
    """The drop-in boundary from a plain-C host (hipMalloc'd buffers, no Python, no torch):
    tests/native/abi_c_host.c steps an episode through the C ABI and checks it against the
    oracle library."""
    import os
    import subprocess
    from conftest import ROOT
    from oracle import oracle as O
    so_oracle = O.build()
    exe = str(tmp_path / "abi_c_host")
    lib_dir = os.path.join(ROOT, "abrsimulator_amd", "csrc")
    subprocess.check_call(["gcc", "-std=gnu11", "-D__HIP_PLATFORM_AMD__", "-O1",
                           os.path.join(ROOT, "tests", "native", "abi_c_host.c"),
                           "-I", "/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
                           "-L", lib_dir, "-labr_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-ldl", "-lm",
                           "-o", exe])
    out = subprocess.run([exe, so_oracle], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "mismatches 0" in out.stdout


def test_hip_graph_capture_of_step():
    """The launch path allocates nothing and never synchronises, so a step can be captured in
    a HIP graph (torch.cuda.graph) and replayed: same results as eager launches."""
    meta, traces, trace_id, offset, actions = _random_case(seed=51, N=4096, V=12)
    N, V = actions.shape
    acts = torch.from_numpy(actions).cuda()
    eager = make_env(meta, traces, N)
    eager.reset(torch.from_numpy(trace_id), torch.from_numpy(offset))
    ref = [tuple(t.clone() for t in eager.step(acts[:, s].contiguous())) for s in range(V)]

    env = make_env(meta, traces, N)
    env.reset(torch.from_numpy(trace_id), torch.from_numpy(offset))
    static_a = acts[:, 0].contiguous().clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):          # warm-up on the capture stream, then restore state
        sd = env.state_dict()
        env.step(static_a)
        env.load_state_dict(sd)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        env.step(static_a)
    env.load_state_dict(sd)                # capture does not execute; be explicit anyway
    for s in range(V):
        static_a.copy_(acts[:, s])
        g.replay()
        assert torch.equal(env.obs, ref[s][0]) and torch.equal(env.reward, ref[s][1])
        assert torch.equal(env.done, ref[s][2])


def test_timeouts_are_identical_on_every_implementation():
    """Lanes that run into max_ticks (ABR_DONE_TIMEOUT; the reference would simply keep looping):
    a starved network and a tick budget barely above the live-stream minimum, so that lanes time
    out inside a download, right after one, and while waiting.  The implementations must
    agree on every output and on the final float64 state, step by step and fused."""
    rng = np.random.default_rng(71)
    V, N, L = 10, 512, 4.0
    traces = [rng.uniform(0.02, 2.5, 800).astype(np.float32).astype(np.float64) for _ in range(6)]
    meta = dict(ladder=[0.3, 0.75, 1.2, 1.85, 2.85, 4.3], chunk_length=L, video_length=V, max_buffer=20.0,
                start_up_length=8.0, interval=1.0, weights=[4.3, 1, 1, 0.1], speed=1.0)
    trace_id = rng.integers(0, 6, N).astype(np.int32)
    offset = rng.integers(0, 800, N).astype(np.int32)
    actions = rng.integers(0, 6, (N, V)).astype(np.int32)
    mt = (V + 1) * 400 + 700
    ref = None
    for impl in IMPLS:
        env = make_env(meta, traces, N, impl=impl, max_ticks=mt)
        env.reset(torch.from_numpy(trace_id), torch.from_numpy(offset))
        acts = torch.from_numpy(actions).cuda()
        rec = []
        for s in range(V):
            o, r, d = env.step(acts[:, s].contiguous())
            rec.append((o.clone(), r.clone(), d.clone()))
        f = {k: v.clone() for k, v in env.observe_f64().items()}
        d_last = rec[-1][2].cpu().numpy()
        assert ((d_last & 2) != 0).sum() > 20 and (d_last == 1).sum() > 20, np.bincount(d_last)
        if ref is None:
            ref = (rec, f)
            continue
        for s in range(V):
            assert torch.equal(rec[s][2], ref[0][s][2]), (impl, s, "done")
            # the event-driven kernels agree on everything; the tick-by-tick kernel stops a timed-out
            # lane at a block boundary, so its frozen counters differ (not a parity claim)
            ok = (rec[s][2] & 2) == 0 if impl == "tick" else torch.ones_like(rec[s][2], dtype=torch.bool)
            assert torch.equal(rec[s][0][:, ok], ref[0][s][0][:, ok]), (impl, s, "obs")
            assert torch.equal(rec[s][1][ok], ref[0][s][1][ok]), (impl, s, "reward")
        ok = (rec[-1][2] & 2) == 0 if impl == "tick" else torch.ones_like(rec[-1][2], dtype=torch.bool)
        for k in f:
            if k not in ("average_latency",):
                assert torch.equal(f[k][ok], ref[1][k][ok]), (impl, k)
    # fused random rollout: split == jump on a workload where most lanes time out
    outs = []
    for impl in ("split", "jump", "ring3", "split3"):
        env = make_env(meta, traces, N, impl=impl, max_ticks=mt, auto_reset=True)
        env.reset(torch.from_numpy(trace_id), torch.from_numpy(offset))
        outs.append(env.step_random(2 * V, 5))
    for o in outs[1:]:
        for k in ("obs", "reward", "done", "actions"):
            assert torch.equal(outs[0][k], o[k]), k
    assert int(((outs[0]["done"][-1] & 2) != 0).sum()) > 20


@pytest.mark.parametrize("V,B", [(1, 1), (1, 6), (2, 3)])
def test_degenerate_shapes_fused_auto_reset(oracle, V, B):
    """One-chunk videos and one-rate ladders: every step ends an episode and re-arms the lane inside
    the kernel (the speculative download side predicts a fresh episode every time)."""
    rng = np.random.default_rng(90 + V * 10 + B)
    N, SEED, EPS = 200, 4242, 5
    ladder = list(np.sort(rng.uniform(0.3, 4.0, B)))
    traces = [rng.uniform(0.3, 6.0, 300).astype(np.float32).astype(np.float64) for _ in range(3)]
    meta = dict(ladder=ladder, chunk_length=2.0, video_length=V, max_buffer=10.0, start_up_length=2.0,
                interval=1.0, weights=[4.3, 1, 1, 0.1], speed=1.0)
    tid = rng.integers(0, 3, N).astype(np.int32); off = rng.integers(0, 300, N).astype(np.int32)
    outs = {}
    for impl in IMPLS:
        env = make_env(meta, traces, N, impl=impl, auto_reset=True)
        env.reset(torch.from_numpy(tid), torch.from_numpy(off))
        outs[impl] = (env.step_random(EPS * V, SEED), env.episode_qoe().cpu().numpy())
    for impl in IMPLS[1:]:
        for k in ("obs", "reward", "done", "actions"):
            assert torch.equal(outs[impl][0][k], outs[IMPLS[0]][0][k]), (impl, k)
    out, qoe = outs["split"]
    assert int(out["done"].sum()) == EPS * N
    acts = out["actions"].cpu().numpy()
    last = acts[(EPS - 1) * V:].T.copy()
    assert np.array_equal(last, np.stack([philox_action(SEED, np.arange(N), s, EPS - 1, B) for s in range(V)], 1))
    cfg = oracle.env_cfg(ladder, 2.0, V, 10.0, 2.0, 1.0, meta["weights"], 1.0)
    _, _, fin, _ = oracle.env_batch(cfg, traces, tid, off, last)
    assert np.allclose(qoe, fin["qoe"], rtol=1e-10)


@pytest.mark.parametrize("impl", ["split", "jump", "async"])
def test_resume_into_a_freshly_built_env(oracle, impl):
    """Checkpoint / resume in a NEW process: env A (speed schedule [rows, N] + per-chunk ladders) steps
    half an episode; its state_dict() goes into a freshly built env B, which never saw a reset.  Both
    finish bit-identically, and identically to an uninterrupted run."""
    import abrsimulator_amd as A
    rng = np.random.default_rng(123)
    V, N, B = 10, 300, 4
    traces = [rng.uniform(0.3, 6.0, 500).astype(np.float32).astype(np.float64) for _ in range(5)]
    table = np.sort(rng.uniform(0.3, 4.0, (V, B)), axis=1)
    sched = torch.from_numpy(rng.uniform(0.8, 1.3, (6, N)))
    tid = torch.from_numpy(rng.integers(0, 5, N).astype(np.int32))
    off = torch.from_numpy(rng.integers(0, 500, N).astype(np.int32))
    acts = torch.from_numpy(rng.integers(0, B, (V, N)).astype(np.int32)).cuda()

    def build():
        mpd = A.MPD(V, 2.0, 12.0, 4.0, [A.Chunk(list(r)) for r in table])
        from helpers import DIAG_IMPLS, diag_lib
        return A.BatchedABREnv(mpd, A.QOEMetric(4.3, 1, 1, 0.1), A.NetworkInfo(1.0, traces), N, speed=sched,
                               impl=impl, library=diag_lib() if impl in DIAG_IMPLS else None)

    ref = build(); ref.reset(tid, off)
    ref_out = ref.step_script(acts)
    a = build(); a.reset(tid, off)
    first = a.step_script(acts[:V // 2])
    sd = a.state_dict()
    b = build()                                  # never reset: the speeds / ladders must already be in force
    b.load_state_dict(sd)
    out_a, out_b = a.step_script(acts[V // 2:]), b.step_script(acts[V // 2:])
    for k in ("obs", "reward", "done"):
        assert torch.equal(out_a[k], out_b[k]), k
        assert torch.equal(torch.cat([first[k], out_b[k]]), ref_out[k]), k
    fa, fb, fr = a.observe_f64(), b.observe_f64(), ref.observe_f64()
    for k in fa:
        assert torch.equal(fa[k], fb[k]) and torch.equal(fb[k], fr[k]), k
    assert torch.equal(a.episode_qoe(), b.episode_qoe()) and torch.equal(b.episode_qoe(), ref.episode_qoe())


def test_checkpoints_carry_a_layout_tag_and_foreign_ones_are_refused():
    """a checkpoint is a copy of the workspace, whose layout changes between ABI versions while its size
    can coincide.  ABI 4: the workspace ends in a layout tag (magic, ABI version, lane count, configuration, size) that
    abr_env_notify_restore reads back, and state_dict() is stamped; a state of another version, lane count or configuration is
    refused BEFORE anything is copied (Python) and by the library itself (a C host that copies the bytes anyway)."""
    from abrsimulator_amd import _lib
    meta, traces, trace_id, offset, actions = _random_case(seed=33, N=128, V=6)
    env = make_env(meta, traces, 128, max_ticks=76800)
    env.reset(torch.from_numpy(trace_id), torch.from_numpy(offset))
    env.step(torch.from_numpy(actions[:, 0].copy()).cuda())
    sd = env.state_dict()
    assert sd["abi_version"] == _lib.ABI_VERSION == 4 and sd["workspace_bytes"] == env.workspace.numel()
    tag = sd["workspace"][-256:].cpu().numpy()
    assert tag[:4].tobytes() == b"ABRW" and int(tag[4:8].view(np.uint32)[0]) == 4
    before = env.workspace.clone()
    # a state stamped with another ABI version (what a round-5 checkpoint would carry, had it carried a stamp) ...
    for bad in (dict(sd, abi_version=3), {k: v for k, v in sd.items() if k != "abi_version"}):
        with pytest.raises(ValueError, match="ABI version"):
            env.load_state_dict(bad)
    # ... one of another lane count (size differs) and one of the SAME size but another configuration (tag differs)
    other = make_env(meta, traces, 192)
    with pytest.raises(ValueError, match="size mismatch"):
        env.load_state_dict(other.state_dict())
    meta2 = dict(meta, chunk_length=meta["chunk_length"] * 0.5)
    twin = make_env(meta2, traces, 128, max_ticks=76800)          # same tables' sizes, another chunk length
    sd2 = twin.state_dict()
    assert sd2["workspace"].numel() == env.workspace.numel()
    with pytest.raises(ValueError, match="layout tag"):
        env.load_state_dict(sd2)
    assert torch.equal(env.workspace, before)                    # nothing was copied by any refused load
    # the library's own check: a C host that copies foreign bytes in and calls abr_env_notify_restore
    env.workspace[-256:].zero_()                                  # no tag at all: a pre-ABI-4 workspace
    assert env.lib.abr_env_notify_restore(env._h) == -2           # ABR_E_WORKSPACE
    assert b"layout tag" in env.lib.abr_last_error()
    forged = sd["workspace"][-256:].clone()
    forged[4] = 3                                                 # abi_version field: 3
    env.workspace[-256:].copy_(forged)
    assert env.lib.abr_env_notify_restore(env._h) == -2 and b"ABI version 3" in env.lib.abr_last_error()
    forged = sd["workspace"][-256:].clone()
    forged[16] = forged[16] + 1                                   # n_lanes field
    env.workspace[-256:].copy_(forged)
    assert env.lib.abr_env_notify_restore(env._h) == -2 and b"another configuration" in env.lib.abr_last_error()
    env.load_state_dict(sd)                                       # the genuine one restores and steps on
    o1 = env.step(torch.from_numpy(actions[:, 1].copy()).cuda())[0].clone()
    env.load_state_dict(sd)
    assert torch.equal(o1, env.step(torch.from_numpy(actions[:, 1].copy()).cuda())[0])


def test_reset_validates_before_it_changes_anything_and_done_after_reset_shows_frozen_lanes():
    """check=True raises BEFORE trace_id / start_offset of the object are overwritten (a later state_dict()
    must not save the bad tensors); the sync-free default freezes a bad lane on the device, and done_after_reset() shows it
    without a step."""
    meta, traces, trace_id, offset, actions = _random_case(seed=34, N=128, V=6)
    env = make_env(meta, traces, 128)
    good_t, good_o = torch.from_numpy(trace_id), torch.from_numpy(offset)
    env.reset(good_t, good_o)
    kept_t, kept_o = env.trace_id.clone(), env.start_offset.clone()
    bad_t = good_t.clone(); bad_t[9] = len(traces) + 3
    bad_o = good_o.clone(); bad_o[11] = -5
    for t_, o_ in ((bad_t, good_o), (good_t, bad_o)):
        with pytest.raises(ValueError):
            env.reset(t_, o_, check=True)
        assert torch.equal(env.trace_id, kept_t) and torch.equal(env.start_offset, kept_o)
    assert int(env.done_after_reset().sum()) == 0
    env.reset(bad_t, bad_o)                                      # the default: no host synchronisation, lanes frozen on the device
    d = env.done_after_reset().cpu().numpy()
    assert d[9] == 8 and d[11] == 8 and (np.delete(d, [9, 11]) == 0).all()          # ABR_DONE_BADARG
    _, _, dn = env.step(torch.from_numpy(actions[:, 0].copy()).cuda())
    assert dn[9] == 8 and dn[11] == 8


def test_product_kernels_answer_the_fresh_params_selfcheck():
    """the role-split kernels' service code re-reads the launch's parameter block from the kernel-argument
    segment (fresh_params(), csrc/abr_env_roles.h), which is right only while that block is the kernels' FIRST argument.
    abr_debug_selfcheck asks a checking instance of both kernel templates in the PRODUCT build: each must have seen the sentinel."""
    from abrsimulator_amd import _lib
    meta, traces, trace_id, offset, actions = _random_case(seed=35, N=128, V=6)
    env = make_env(meta, traces, 128)
    env.reset(torch.from_numpy(trace_id), torch.from_numpy(offset))
    before = env.workspace.clone()
    res = torch.full((2,), 7, dtype=torch.int32, device="cuda")
    _lib.check(env.lib.abr_debug_selfcheck(env._h, _lib.ptr(res), None))
    torch.cuda.synchronize()
    assert res.cpu().tolist() == [1, 1], res.cpu().tolist()
    assert torch.equal(env.workspace, before)                    # no lane state touched
    assert env.lib.abr_debug_selfcheck(None, _lib.ptr(res), None) == -1
