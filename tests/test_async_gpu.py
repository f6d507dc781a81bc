"""GPU parity of the asynchronous K1 pipeline (impl 'async': download / player / service roles over
per-lane LDS rings, csrc/abr_env_async.h) -- against the reference's goldens, the oracle, and the
other implementations, on the paths the pipeline adds: lanes at different steps inside one wave,
window refills of the bandwidth trace, ring back-pressure, mis-speculated downloads (buffer_full
gating, Simulator.py:144) redone from their cursor snapshot, launches cut at 64 decisions.

Bar: float32 observations equal float32(float64 reference value) exactly; rewards, done bits,
actions, histories and the final float64 state identical to the other implementations.
"""
import numpy as np
import pytest
import torch

from conftest import ENV_GOLDENS, load_golden
from helpers import (F64_EXACT, auto_reset_rollout_rewards, golden_rewards, make_env, oracle_rewards,
                     philox_action)

pytestmark = pytest.mark.gpu

FUSED_IMPLS = ["async", "ring3", "pair3", "split3", "split", "jump"]


def _obs_expect(rec, s):
    """float32 observation rows at call site s from a golden / oracle step record."""
    return {0: rec["chunk_id"][:, s].astype(np.float32), 3: rec["buffer_level"][:, s].astype(np.float32),
            4: rec["global_time"][:, s].astype(np.float32), 5: rec["play_time"][:, s].astype(np.float32),
            6: rec["rebuffer_time"][:, s].astype(np.float32), 7: rec["start_up_time"][:, s].astype(np.float32)}


@pytest.mark.parametrize("impl", FUSED_IMPLS)
@pytest.mark.parametrize("name", ENV_GOLDENS)
def test_scripted_rollout_matches_reference_goldens(name, impl):
    """The golden episodes in ONE fused call: the observation written after decision s is the
    reference's run() frame at call site s + 1; the episode's QoE is run()'s return value."""
    m, g = load_golden(name)
    N, V = g["actions"].shape
    env = make_env(m, g["traces"], N, impl=impl)
    assert env.effective_impl(fused=True) == impl
    env.reset(torch.from_numpy(g["trace_id"]), torch.from_numpy(g["offset"]))
    out = env.step_script(torch.from_numpy(g["actions"].T.copy()))
    obs = out["obs"].cpu().numpy()
    for s in range(V - 1):
        for row, want in _obs_expect(g, s + 1).items():
            assert np.array_equal(obs[s, row], want), (name, impl, s, row)
        assert np.array_equal(obs[s, 2], g["arg_last_bandwidth"][:, s + 1].astype(np.float32)), (name, s)
        assert np.array_equal(obs[s, 1], g["arg_last_bitrate"][:, s + 1].astype(np.float32)), (name, s)
    done = out["done"].cpu().numpy()
    assert (done[:-1] == 0).all() and (done[-1] == 1).all()
    # every per-step reward == float32 of the value derived from the reference's frames
    assert np.array_equal(out["reward"].cpu().numpy().T, golden_rewards(m, g)), (name, impl)
    qoe = env.episode_qoe().cpu().numpy()
    assert np.allclose(qoe, g["final_qoe"], rtol=1e-10), (qoe[:4], g["final_qoe"][:4])
    lat = float(m["weights"][3])
    rew = out["reward"].double().sum(0).cpu().numpy()
    f = env.observe_f64()
    assert np.allclose(rew + lat * f["average_latency"].cpu().numpy(), g["final_qoe"], rtol=1e-5)
    assert np.array_equal(env.history()[1].cpu().numpy().T, g["final_bandwidths"])


@pytest.mark.parametrize("seed", range(16))
def test_scripted_rollout_random_configurations_against_oracle(oracle, seed):
    """Config-space fuzz through the fused scripted call: chunk lengths, trace intervals shorter than
    the prologue / longer than a chunk, ladders of 2-8 rates, small buffer limits (buffer_full
    gating -> redone downloads), start_up_length 0, ragged traces with wrap-around."""
    from test_lane_jump_cpu import _random_config
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
    ref = None
    want_rew = oracle_rewards(steps, fin, actions, meta["weights"], ladder=meta["ladder"])
    for impl in ("async", "ring3", "pair3", "split3", "jump"):
        env = make_env(meta, traces, N, impl=impl, max_ticks=int(fin["ticks"].max()) + 1000)
        env.reset(torch.from_numpy(trace_id), torch.from_numpy(offset))
        out = env.step_script(torch.from_numpy(actions.T.copy()))
        assert np.array_equal(out["reward"].cpu().numpy().T, want_rew), (impl, seed)
        obs = out["obs"].cpu().numpy()
        for s in range(V - 1):
            for row, want in _obs_expect(steps, s + 1).items():
                assert np.array_equal(obs[s, row], want), (impl, seed, s, row)
        assert np.array_equal(env.history()[1].cpu().numpy().T, bw), impl
        assert np.allclose(env.episode_qoe().cpu().numpy(), fin["qoe"], rtol=1e-10), impl
        f = {k: v.cpu().numpy() for k, v in env.observe_f64().items()}
        if ref is None:
            ref = (out, f)
        else:
            for k in ("obs", "reward", "done"):
                assert torch.equal(out[k], ref[0][k]), (impl, k)
            for k in f:
                assert np.array_equal(f[k], ref[1][k]), (impl, k)


def _bench_like(rng, n_traces=64, lo=0.2, hi=6.0, tl=(300, 1500)):
    lens = rng.integers(tl[0], tl[1], n_traces)
    return [rng.uniform(lo, hi, int(l)).astype(np.float32).astype(np.float64) for l in lens]


BENCH_META = dict(ladder=[0.3, 0.75, 1.2, 1.85, 2.85, 4.3], chunk_length=4.0, video_length=48, max_buffer=20.0,
                  start_up_length=8.0, interval=1.0, weights=[4.3, 1, 1, 0.1], speed=1.0)


def _bench_cfg(oracle, meta):
    return oracle.env_cfg(meta["ladder"], meta["chunk_length"], meta["video_length"], meta["max_buffer"],
                          meta["start_up_length"], meta["interval"], meta["weights"], 1.0)


def test_fused_random_rollout_identical_to_other_implementations(oracle):
    """The bench shape (auto_reset, 48-chunk episodes, ragged traces), a lane count that is not a
    multiple of the 256-lane workgroup, 103 decisions = two launches (64 + 39) inside one call; then
    the same rollout cut into uneven calls (state handed over between launches)."""
    rng = np.random.default_rng(7)
    traces = _bench_like(rng)
    N = 4096 + 77
    tid = rng.integers(0, len(traces), N).astype(np.int32)
    off = np.array([rng.integers(0, len(traces[t])) for t in tid], np.int32)
    outs, states = {}, {}
    for impl in FUSED_IMPLS:
        env = make_env(BENCH_META, traces, N, impl=impl, auto_reset=True, lane_id_base=12345)
        env.reset(torch.from_numpy(tid), torch.from_numpy(off))
        outs[impl] = env.step_random(103, 99)
        states[impl] = ({k: v.clone() for k, v in env.observe_f64().items()}, env.episode_qoe().clone(),
                        env.history()[0].clone(), env.history()[1].clone())
    for impl in FUSED_IMPLS[1:]:
        for k in ("obs", "reward", "done", "actions"):
            assert torch.equal(outs[impl][k], outs["async"][k]), (impl, k)
        for k in states["async"][0]:
            assert torch.equal(states[impl][0][k], states["async"][0][k]), (impl, k)
        for q in (1, 2, 3):
            assert torch.equal(states[impl][q], states["async"][q]), (impl, q)
    assert int((outs["async"]["done"] == 1).sum()) >= 2 * N - N // 2      # episodes really ended and re-armed
    # ... and the rewards are the reference-derived ones, element by element, across two auto-resets
    want = auto_reset_rollout_rewards(oracle, _bench_cfg(oracle, BENCH_META), traces, tid, off,
                                      outs["async"]["actions"].cpu().numpy(), BENCH_META["weights"],
                                      ladder=BENCH_META["ladder"])
    assert np.array_equal(outs["async"]["reward"].cpu().numpy(), want)
    # uneven cuts, alternating implementations between calls: the workspace is interchangeable
    env = make_env(BENCH_META, traces, N, impl="async", auto_reset=True, lane_id_base=12345)
    env.reset(torch.from_numpy(tid), torch.from_numpy(off))
    parts, at = [], 0
    for n, impl in ((1, 4), (7, 4), (30, 2), (2, 4), (63, 4)):
        env.lib.abr_env_set_impl(env._h, impl)
        parts.append(env.step_random(n, 99))
        at += n
    for k in ("obs", "reward", "done", "actions"):
        assert torch.equal(torch.cat([p[k] for p in parts]), outs["async"][k]), k


def test_buffer_full_gating_redoes_downloads(oracle):
    """A buffer limit of 1.5 chunks and a fast network: buffer_full gates almost every download
    (Simulator.py:144), so the download role's "not gated" guess is wrong all the time and the
    player sends it back to the snapshot with the true call-site tick.  Results must not change."""
    rng = np.random.default_rng(11)
    traces = _bench_like(rng, n_traces=16, lo=4.0, hi=30.0, tl=(50, 400))
    meta = dict(BENCH_META, max_buffer=6.0, start_up_length=4.0, video_length=24)
    N = 1500
    tid = rng.integers(0, len(traces), N).astype(np.int32)
    off = np.array([rng.integers(0, len(traces[t])) for t in tid], np.int32)
    outs = {}
    for impl in FUSED_IMPLS:
        env = make_env(meta, traces, N, impl=impl, auto_reset=True)
        env.reset(torch.from_numpy(tid), torch.from_numpy(off))
        outs[impl] = (env.step_random(60, 3), {k: v.clone() for k, v in env.observe_f64().items()})
    for impl in FUSED_IMPLS[1:]:
        for k in ("obs", "reward", "done", "actions"):
            assert torch.equal(outs[impl][0][k], outs["async"][0][k]), (impl, k)
        for k in outs["async"][1]:
            assert torch.equal(outs[impl][1][k], outs["async"][1][k]), (impl, k)
    want = auto_reset_rollout_rewards(oracle, _bench_cfg(oracle, meta), traces, tid, off,
                                      outs["async"][0]["actions"].cpu().numpy(), meta["weights"], ladder=meta["ladder"])
    assert np.array_equal(outs["async"][0]["reward"].cpu().numpy(), want)
    # the gating really happened: buffer_full set at many call sites' predecessors means waits beyond availability
    fl = outs["async"][1]["buffer_level"]
    assert float(fl.max()) <= 6.0 + 4.0


def test_timeouts_and_no_auto_reset():
    """Starved network + a tick budget barely above the live minimum (lanes time out inside a download,
    right after one, while waiting), with and without auto_reset; lanes that were finished before
    the call; a second call on finished lanes."""
    rng = np.random.default_rng(71)
    V, N = 10, 700
    traces = [rng.uniform(0.02, 2.5, 800).astype(np.float32).astype(np.float64) for _ in range(6)]
    meta = dict(BENCH_META, video_length=V)
    tid = rng.integers(0, 6, N).astype(np.int32)
    off = rng.integers(0, 800, N).astype(np.int32)
    mt = (V + 1) * 400 + 700
    for auto in (True, False):
        res = {}
        for impl in FUSED_IMPLS:
            env = make_env(meta, traces, N, impl=impl, max_ticks=mt, auto_reset=auto)
            env.reset(torch.from_numpy(tid), torch.from_numpy(off))
            a = env.step_random(V // 2, 5)
            b = env.step_random(2 * V, 5)
            c = env.step_random(3, 5)          # every lane is finished or timed out by now (auto=False)
            res[impl] = (a, b, c, {k: v.clone() for k, v in env.observe_f64().items()})
        for impl in FUSED_IMPLS[1:]:
            for q in range(3):
                for k in ("obs", "reward", "done", "actions"):
                    assert torch.equal(res[impl][q][k], res["async"][q][k]), (auto, impl, q, k)
            for k in res["async"][3]:
                if k != "average_latency":
                    assert torch.equal(res[impl][3][k], res["async"][3][k]), (auto, impl, k)
        d = res["async"][1]["done"][-1].cpu().numpy()
        assert ((d & 2) != 0).sum() > 20, np.bincount(d)


@pytest.mark.parametrize("V,B", [(1, 1), (1, 6), (2, 3), (3, 2)])
def test_degenerate_shapes(oracle, V, B):
    """One-chunk videos and one-rate ladders: every decision ends an episode and re-arms the lane."""
    rng = np.random.default_rng(90 + V * 10 + B)
    N, SEED, EPS = 300, 4242, 7
    ladder = list(np.sort(rng.uniform(0.3, 4.0, B)))
    traces = [rng.uniform(0.3, 6.0, 300).astype(np.float32).astype(np.float64) for _ in range(3)]
    meta = dict(ladder=ladder, chunk_length=2.0, video_length=V, max_buffer=10.0, start_up_length=2.0,
                interval=1.0, weights=[4.3, 1, 1, 0.1], speed=1.0)
    tid = rng.integers(0, 3, N).astype(np.int32); off = rng.integers(0, 300, N).astype(np.int32)
    outs = {}
    for impl in FUSED_IMPLS:
        env = make_env(meta, traces, N, impl=impl, auto_reset=True)
        env.reset(torch.from_numpy(tid), torch.from_numpy(off))
        outs[impl] = (env.step_random(EPS * V, SEED), env.episode_qoe().cpu().numpy())
    for impl in FUSED_IMPLS[1:]:
        for k in ("obs", "reward", "done", "actions"):
            assert torch.equal(outs[impl][0][k], outs["async"][0][k]), (impl, k)
        assert np.array_equal(outs[impl][1], outs["async"][1])
    acts = outs["async"][0]["actions"].cpu().numpy()
    last = acts[(EPS - 1) * V:].T.copy()
    assert np.array_equal(last, np.stack([philox_action(SEED, np.arange(N), s, EPS - 1, B) for s in range(V)], 1))
    cfg = oracle.env_cfg(ladder, 2.0, V, 10.0, 2.0, 1.0, meta["weights"], 1.0)
    _, _, fin, _ = oracle.env_batch(cfg, traces, tid, off, last)
    assert np.allclose(outs["async"][1], fin["qoe"], rtol=1e-10)


def test_scripted_bad_action_freezes_the_lane():
    rng = np.random.default_rng(5)
    traces = _bench_like(rng, n_traces=4)
    meta = dict(BENCH_META, video_length=6)
    N = 300
    acts = rng.integers(0, 6, (6, N)).astype(np.int32)
    acts[2, 5] = 6; acts[0, 17] = -1; acts[5, 299] = 99
    res = {}
    for impl in FUSED_IMPLS:
        env = make_env(meta, traces, N, impl=impl)
        env.reset()
        res[impl] = (env.step_script(torch.from_numpy(acts)), {k: v.clone() for k, v in env.observe_f64().items()})
    d = res["async"][0]["done"].cpu().numpy()
    assert d[2, 5] == 4 and d[1, 5] == 0 and d[5, 5] == 4 and d[0, 17] == 4 and d[5, 299] == 4 and d[4, 299] == 0
    for impl in FUSED_IMPLS[1:]:
        for k in ("obs", "reward", "done"):
            assert torch.equal(res[impl][0][k], res["async"][0][k]), (impl, k)
        for k in res["async"][1]:
            assert torch.equal(res[impl][1][k], res["async"][1][k]), (impl, k)


def test_async_falls_back_with_per_lane_speeds():
    rng = np.random.default_rng(6)
    traces = _bench_like(rng, n_traces=4)
    N = 256
    sp = rng.uniform(0.8, 1.3, N)
    env = make_env(dict(BENCH_META, speed=torch.from_numpy(sp)), traces, N, impl="async", auto_reset=True)
    env.reset()
    assert env.effective_impl(fused=True) == "split" and env.effective_impl() == "split"
    a = env.step_random(50, 1)
    env2 = make_env(dict(BENCH_META, speed=torch.from_numpy(sp)), traces, N, impl="jump", auto_reset=True)
    env2.reset()
    b = env2.step_random(50, 1)
    for k in ("obs", "reward", "done", "actions"):
        assert torch.equal(a[k], b[k]), k


def test_auto_resolves_to_the_measured_fastest():
    """`auto` is a measured choice: for fused rollouts the three-wave role-split kernel up to
    65 536 lanes, the two-wave one up to 131 072, one thread per lane above; for launches of ONE decision
    (step, the K1 launches of step_mpc, a fused call of one step) one thread per lane at every size;
    the asynchronous pipeline is not in the product library at all."""
    rng = np.random.default_rng(6)
    traces = _bench_like(rng, n_traces=4)
    env = make_env(BENCH_META, traces, 512)
    assert env.effective_impl(fused=True) == "split3" and env.effective_impl(fused=False) == "jump"
    env = make_env(BENCH_META, traces, 32769)
    assert env.effective_impl(fused=True) == "split3" and env.effective_impl(fused=False) == "jump"
    # ... and whichever serves a call, the state it leaves is the same: single steps (one thread per lane) and a
    # fused rollout (three waves) of the same lanes agree bit for bit
    env.reset()
    env2 = make_env(BENCH_META, traces, 32769, impl="split3")
    env2.reset()
    acts = torch.from_numpy(rng.integers(0, 6, (3, 32769)).astype(np.int32)).cuda()
    fused = env2.step_script(acts)
    for s in range(3):
        o, r, d = env.step(acts[s])
        assert torch.equal(o, fused["obs"][s]) and torch.equal(r, fused["reward"][s]) and torch.equal(d, fused["done"][s])
    one = env.step_script(acts[:1])                 # a fused call of ONE step takes the single-decision kernel too
    two = env2.step_script(acts[:1])
    assert torch.equal(one["obs"], two["obs"]) and torch.equal(one["reward"], two["reward"])
    env = make_env(BENCH_META, traces, 65537)
    assert env.effective_impl(fused=True) == "split" and env.effective_impl(fused=False) == "jump"
    env = make_env(BENCH_META, traces, 131072)
    assert env.effective_impl(fused=True) == "split" and env.effective_impl(fused=False) == "jump"
    env = make_env(BENCH_META, traces, 131073)
    assert env.effective_impl(fused=True) == "jump" and env.effective_impl(fused=False) == "jump"
    import abrsimulator_amd as A
    _lib = A._lib.lib()
    for impl_no in (4, 6, 7):                     # the rejected pipelines are refused by the product library ...
        assert _lib.abr_env_has_impl(impl_no) == 0
        with pytest.raises(A._lib.AbrError):
            A._lib.check(_lib.abr_env_set_impl(env._h, impl_no), _lib)
    assert all(_lib.abr_env_has_impl(i) == 1 for i in (0, 1, 2, 3, 5))
    from helpers import diag_lib
    diag = A._lib.lib(diag_lib())                 # ... and carried by the diagnostic build, which tests name explicitly
    assert diag.abr_env_has_impl(4) == 1 and diag.abr_env_has_impl(6) == 1 and diag.abr_env_has_impl(7) == 1
    with pytest.raises(A._lib.AbrError):          # the package never loads that build by itself
        make_env(BENCH_META, traces, 512, impl="ring3", library=A._lib.SO_PATH)


def test_per_chunk_ladders_on_the_async_pipeline(oracle):
    """BUILD-DEFINED per-chunk ladders (abr_env_set_bitrate_table) reach the download role's target
    size and the service role's variance term."""
    import abrsimulator_amd as A
    rng = np.random.default_rng(8)
    V, N, B = 12, 300, 5
    traces = _bench_like(rng, n_traces=5)
    table = np.sort(rng.uniform(0.3, 5.0, (V, B)), axis=1)
    outs = {}
    for impl in FUSED_IMPLS:
        mpd = A.MPD(V, 4.0, 20.0, 8.0, [A.Chunk(list(r)) for r in table])
        from helpers import DIAG_IMPLS, diag_lib
        env = A.BatchedABREnv(mpd, A.QOEMetric(4.3, 1, 1, 0.1), A.NetworkInfo(1.0, traces), N, auto_reset=True,
                              impl=impl, library=diag_lib() if impl in DIAG_IMPLS else None)
        env.reset()
        outs[impl] = (env.step_random(3 * V, 17), env.episode_qoe().clone())
    for impl in FUSED_IMPLS[1:]:
        for k in ("obs", "reward", "done", "actions"):
            assert torch.equal(outs[impl][0][k], outs["async"][0][k]), (impl, k)
        assert torch.equal(outs[impl][1], outs["async"][1])


def test_launch_cuts_with_everything_in_between(oracle):
    """Fused random-policy launches of the three-wave kernel cut anywhere -- on every chunk, on episode ends, on call sites
    gated by buffer_full -- with whatever can happen between two launches: nothing, another seed, a masked reset, single
    steps, a scripted launch, another implementation.  The outputs and the state equal the one-thread-per-lane kernel's.
    (Written for round 5's cross-launch stash of the first download -- measured at 1 us per launch and not kept,
    profiles/r05_experiments_not_kept.txt, r05_stash.patch -- and kept as a test of the hand-over between launches.)"""
    rng = np.random.default_rng(17)
    traces = _bench_like(rng, n_traces=8)
    N = 777
    meta = dict(BENCH_META, video_length=6, max_buffer=9.0)          # short episodes: the cut falls on every chunk, also on
    tid = rng.integers(0, len(traces), N).astype(np.int32)            # episode ends; a small buffer: gated cuts
    off = np.array([rng.integers(0, len(traces[t])) for t in tid], np.int32)
    envs = {impl: make_env(meta, traces, N, impl=impl, auto_reset=True, lane_id_base=4242) for impl in ("split3", "jump")}
    for e in envs.values():
        e.reset(torch.from_numpy(tid), torch.from_numpy(off))
    mask = torch.from_numpy((rng.random(N) < 0.4).astype(np.uint8))
    acts1 = torch.from_numpy(rng.integers(0, 6, N).astype(np.int32)).cuda()
    script = torch.from_numpy(rng.integers(0, 6, (5, N)).astype(np.int32)).cuda()

    def both(fn):
        a, b = fn(envs["split3"]), fn(envs["jump"])
        if isinstance(a, dict):
            for k in a:
                if a[k] is not None:
                    assert torch.equal(a[k], b[k]), k
        elif isinstance(a, tuple):
            for x, y in zip(a, b):
                assert torch.equal(x, y)
        fa, fb = envs["split3"].observe_f64(), envs["jump"].observe_f64()
        for k in fa:
            assert torch.equal(fa[k], fb[k]), k
    for n in (7, 7, 5, 6, 13, 2):                                     # back to back
        both(lambda e, n=n: e.step_random(n, 31))
    both(lambda e: e.step_random(7, 32))                              # another seed
    both(lambda e: e.step_random(7, 32))
    both(lambda e: (e.reset(torch.from_numpy(tid), torch.from_numpy(off), mask=mask).clone(),))   # some lanes start over
    both(lambda e: e.step_random(9, 32))
    both(lambda e: tuple(x.clone() for x in e.step(acts1)))           # single steps move the call site
    both(lambda e: e.step_random(4, 32))
    both(lambda e: e.step_script(script))                             # a scripted launch
    both(lambda e: e.step_random(11, 32))
    envs["split3"].lib.abr_env_set_impl(envs["split3"]._h, 2)         # the two-wave kernel in between
    both(lambda e: e.step_random(5, 32))
    envs["split3"].lib.abr_env_set_impl(envs["split3"]._h, 5)
    both(lambda e: e.step_random(8, 32))
    both(lambda e: e.step_random(8, 32))
