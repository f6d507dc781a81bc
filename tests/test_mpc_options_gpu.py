"""f4: the exponential-smoothing predictor and the log bitrate utility.

PARITY UNPINNED.  statsmodels (mpc.py:4,74) cannot be installed here and no reference test
touches either branch, so these tests pin the DOCUMENTED rule (include/abr_env.h:
abr_mpc_options) against an independent numpy evaluation of the same rule -- self-consistency,
not parity with the reference."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LADDER = np.array([0.3, 0.75, 1.2, 1.85, 2.85, 4.3])


def ses_level(y, alpha=0.5):
    """Simple exponential smoothing, initial level = least-squares optimum of the one-step-ahead
    errors (closed form), last level returned: the flat out-of-sample forecast."""
    y = np.asarray(y, np.float64)
    # l(t-1) = a(t) + b(t) * l0
    a, b = np.zeros(len(y) + 1), np.ones(len(y) + 1)
    for t in range(len(y)):
        a[t + 1] = alpha * y[t] + (1 - alpha) * a[t]
        b[t + 1] = (1 - alpha) * b[t]
    l0 = np.sum(b[:-1] * (y - a[:-1])) / np.sum(b[:-1] ** 2)
    # brute-force check of the closed form: SSE is minimal at l0
    def sse(l):
        lv, e = l, 0.0
        for v in y:
            e += (v - lv) ** 2
            lv = alpha * v + (1 - alpha) * lv
        return e
    assert sse(l0) <= min(sse(l0 * (1 + 1e-3)), sse(l0 * (1 - 1e-3))) + 1e-15
    return a[-1] + b[-1] * l0


class _Info:
    pass


class _Player:
    def __init__(self, mpd, qoe, ci):
        self.mpd, self.qoe, self.ci = mpd, qoe, ci

    def get_mpd(self):
        return self.mpd

    def get_qoe_metric(self):
        return self.qoe

    def get_next_chunk_info(self):
        return self.ci


def _setup(rng, N, V=30, H=5, vbr=True, **kw):
    import abrsimulator_amd as A
    L, mb = 4.0, 20.0
    br = LADDER[None, :] * (rng.uniform(0.8, 1.2, (V, 6)) if vbr else 1.0)
    sz = br * L * rng.uniform(0.7, 1.3, (V, 6))
    mpd = A.MPD(V, L, mb, 0.0, [A.Chunk(list(b), list(s)) for b, s in zip(br, sz)])
    ci = _Info()
    chunk = rng.integers(1, V - H + 1, N).astype(np.int32)
    T = 16
    hist = rng.uniform(0.2, 6.0, (T, N))
    hlen = rng.integers(1, T + 1, N).astype(np.int32)
    hn = hlen.astype(np.float64)
    hs = np.array([np.sum(1.0 / hist[:hlen[i], i]) for i in range(N)])
    ci.chunk_number = torch.from_numpy(chunk).cuda()
    ci.previous_bitrate = torch.from_numpy(rng.integers(0, 6, N).astype(np.int32)).cuda()
    ci.buffer_level = torch.from_numpy(rng.uniform(0, mb, N)).cuda()
    ci.hist_n, ci.hist_sum_inv = torch.from_numpy(hn).cuda(), torch.from_numpy(hs).cuda()
    ci.previous_bandwidths, ci.history_length = torch.from_numpy(hist).cuda(), torch.from_numpy(hlen).cuda()
    ctl = A.BatchedMPCController(_Player(mpd, A.QOEMetric(4.3, 1.0, 0.0), ci), horizon=H, **kw)
    return ctl, ci, br, sz, hist, hlen, (L, mb, V, H)


def test_expsmoothing_predictor(oracle):
    rng = np.random.default_rng(1)
    N = 300
    ctl, ci, br, sz, hist, hlen, (L, mb, V, H) = _setup(rng, N, method="expsmoothing")
    hn0, hs0 = ci.hist_n.clone(), ci.hist_sum_inv.clone()
    a = ctl.next_bitrate(want_details=True).cpu().numpy()
    # the history is not grown (no D9 in this branch of mpc.py:72-79)
    assert torch.equal(ci.hist_n, hn0) and torch.equal(ci.hist_sum_inv, hs0)
    cfg = oracle.mpc_cfg(6, H, V, L, mb, 1.0, 4.3, 0.0)
    chunk, prev, buf = (ci.chunk_number.cpu().numpy(), ci.previous_bitrate.cpu().numpy(),
                        ci.buffer_level.cpu().numpy())
    J = ctl.last_J.cpu().numpy()
    for i in range(N):
        level = ses_level(hist[:hlen[i], i])
        # the objective under a flat forecast, from the (pinned) oracle brute force
        f, Jm, Jall = oracle.mpc_brute(cfg, br, sz, chunk[i], prev[i], buf[i], np.full(H, level))
        assert np.isclose(J[i], Jm, rtol=1e-12, atol=1e-12), (i, J[i], Jm)
        srt = np.sort(Jall)
        if srt[1] - srt[0] > 1e-9:                       # away from near-ties the decision is the same
            assert a[i] == f // 6 ** (H - 1)
    # long histories forget the initial level: the level matches a plain recursion from y[0]
    long = np.flatnonzero(hlen >= 14)
    assert len(long) > 10
    for i in long[:10]:
        lv = hist[0, i]
        for v in hist[1:hlen[i], i]:
            lv = 0.5 * v + 0.5 * lv
        assert abs(ses_level(hist[:hlen[i], i]) - lv) < 6.0 * 2.0 ** -(hlen[i] - 1)


def test_log_utility(oracle):
    rng = np.random.default_rng(2)
    N = 300
    ctl, ci, br, sz, hist, hlen, (L, mb, V, H) = _setup(rng, N, utility="log")
    hn, hs = ci.hist_n.cpu().numpy().copy(), ci.hist_sum_inv.cpu().numpy().copy()
    a = ctl.next_bitrate(want_details=True).cpu().numpy()
    J = ctl.last_J.cpu().numpy()
    u = np.log(br / br[:, -1:])                           # log_bitrate_utility, mpc.py:99-102
    assert (u[:, -1] == 0).all() and (u <= 0).all()
    cfg = oracle.mpc_cfg(6, H, V, L, mb, 1.0, 4.3, 0.0)
    chunk, prev, buf = (ci.chunk_number.cpu().numpy(), ci.previous_bitrate.cpu().numpy(),
                        ci.buffer_level.cpu().numpy())
    act, flat, Jm, _ = oracle.mpc_select(cfg, u, sz, chunk, prev, buf, hn, hs)   # the utility table in place of bitrates
    assert np.allclose(J, Jm, rtol=1e-12, atol=1e-12)
    assert (a == act).mean() > 0.97                        # log() may differ in the last ulp at exact ties
    # the harmonic predictor still grows the history (D9)
    assert np.array_equal(ci.hist_n.cpu().numpy(), hn)


def test_expsmoothing_from_the_environment_history():
    """EnvPlayer hands the predictor the environment's previous_bandwidths list itself."""
    import abrsimulator_amd as A
    from helpers import make_env
    rng = np.random.default_rng(3)
    V, N = 12, 128
    traces = [rng.uniform(0.2, 6.0, 400).astype(np.float32).astype(np.float64) for _ in range(4)]
    meta = dict(ladder=list(LADDER), chunk_length=4.0, video_length=V, max_buffer=20.0, start_up_length=8.0,
                interval=1.0, weights=[4.3, 1, 1, 0.1], speed=1.0)
    env = make_env(meta, traces, N)
    env.reset()
    mpd = A.MPD(V, 4.0, 20.0, 8.0, [A.Chunk(list(LADDER), list(LADDER * 4.0))] * V)
    player = A.EnvPlayer(env, mpd=mpd, qoe=A.QOEMetric(0.3, 0.5, 0.0))
    ctl = A.BatchedMPCController(player, horizon=3, clip_horizon=True, method="expsmoothing")
    env.step(torch.zeros(N, dtype=torch.int32, device="cuda"))
    for s in range(1, V - 3):
        a = ctl.next_bitrate(want_details=True)
        assert int(a.min()) >= 0
        bw = env.history()[1].cpu().numpy()[:s]            # [s, N]
        assert float(player.hist_n[0]) == s                 # untouched by the predictor
        lvl = np.array([ses_level(bw[:, i]) for i in range(0, N, 16)])
        assert (lvl > 0).all()
        env.step(a)


def test_options_validation():
    import abrsimulator_amd as A
    with pytest.raises(ValueError):
        A.BatchedMPCController(None, method="arima")
    with pytest.raises(ValueError):
        A.BatchedMPCController(None, utility="sqrt")
