"""GPU parity of K3 (mpc_select) through the C ABI: bit-exact J, flat arg-min
(first-minimum tie-break), action, and the D9 history mutation."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, MPC_GOLDENS, load_golden

pytestmark = pytest.mark.gpu


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


def _controller(br, sz, L, max_buffer, wr, wv, ws, H, chunk, prev, buf, hn, hs, clip=False):
    import abrsimulator_amd as A
    chunks = [A.Chunk(list(b), list(s)) for b, s in zip(br, sz)]
    mpd = A.MPD(len(chunks), L, max_buffer, 0.0, chunks)
    qoe = A.QOEMetric(wr, wv, ws)
    ci = _Info()
    ci.chunk_number = torch.as_tensor(chunk, dtype=torch.int32).cuda()
    ci.previous_bitrate = torch.as_tensor(prev, dtype=torch.int32).cuda()
    ci.buffer_level = torch.as_tensor(buf, dtype=torch.float64).cuda()
    ci.hist_n = torch.as_tensor(hn, dtype=torch.float64).cuda()
    ci.hist_sum_inv = torch.as_tensor(hs, dtype=torch.float64).cuda()
    ctl = A.BatchedMPCController(_Player(mpd, qoe, ci), horizon=H, clip_horizon=clip)
    return ctl, ci


def test_known_answer_mpc_test_py():
    """mpc_test.py:52-72,81-86: 'Test next bitrate: 2'."""
    with open(os.path.join(GOLDEN, "mpc_known_answer.json")) as f:
        k = json.load(f)
    J_ref = np.load(os.path.join(GOLDEN, "mpc_known_answer_J.npz"))["Jout"]
    s = 0
    for x in k["history"]:
        s += 1 / x
    br = [k["ladder"]] * k["video_length"]
    ctl, ci = _controller(br, br, k["chunk_length"], k["max_buffer"], k["weights"]["rebuffer"],
                          k["weights"]["variance"], k["weights"]["startup"], k["horizon"],
                          [k["chunk"]], [k["prev_bitrate"]], [float(k["buffer"])], [5.0], [s])
    a = ctl.next_bitrate(want_details=True)
    assert int(a[0]) == 2 == k["action"]
    assert int(ctl.last_flat[0]) == 639 and float(ctl.last_J[0]) == k["Jmin"]
    assert float(ci.hist_n[0]) == 10.0                       # D9
    J = ctl.objective_grid(k["chunk"], k["prev_bitrate"], k["buffer"], k["pred"]).cpu().numpy()
    assert np.array_equal(J, J_ref)


@pytest.mark.parametrize("name", MPC_GOLDENS)
def test_sweeps_match_reference_goldens(name):
    m, g = load_golden(name)
    ctl, ci = _controller(g["br"], g["sz"], m["chunk_length"], m["max_buffer"], m["rebuffer_weight"],
                          m["variance_weight"], m["startup_weight"], m["horizon"], g["chunk"],
                          g["prev"], g["buf"], g["hist_n"].astype(np.float64), g["hist_s"])
    a = ctl.next_bitrate(want_details=True).cpu().numpy()
    assert np.array_equal(a, g["action"])
    assert np.array_equal(ctl.last_flat.cpu().numpy(), g["flat"])
    assert np.array_equal(ctl.last_J.cpu().numpy(), g["Jmin"])
    assert np.array_equal(ci.hist_n.cpu().numpy(), g["hist_n_after"].astype(np.float64))
    assert np.array_equal(ci.hist_sum_inv.cpu().numpy(), g["hist_s_after"])
    for i in range(m["n_full"]):
        J = ctl.objective_grid(g["chunk"][i], g["prev"][i], g["buf"][i], g["pred"][i]).cpu().numpy()
        assert np.array_equal(J, g["Jfull"][i])


@pytest.mark.parametrize("B,H,N", [(6, 5, 4096), (4, 5, 1000), (6, 3, 513), (3, 2, 100), (5, 4, 300),
                                   (2, 8, 64), (16, 2, 50), (7, 6, 40), (16, 3, 20), (12, 4, 12), (4, 6, 200), (6, 6, 50), (6, 2, 300),
                                   # the compile-time ladders of 8 rates (<H, 8, 0> for H <= 6) and 3 / 5 / 7 at other horizons
                                   (8, 4, 100), (8, 3, 64), (8, 5, 24), (8, 6, 6), (3, 4, 200), (5, 5, 120), (7, 4, 60), (5, 3, 200)])
def test_select_matches_oracle_seeded(oracle, B, H, N):
    rng = np.random.default_rng(B * 100 + H)
    V, L, mb = 40, 4.0, 20.0
    lad = np.sort(rng.uniform(0.2, 6.0, B))
    br = lad[None, :] * rng.uniform(0.8, 1.2, (V, B))
    sz = br * L * rng.uniform(0.7, 1.3, (V, B))
    chunk = rng.integers(0, V - H + 1, N).astype(np.int32)
    prev = rng.integers(0, B, N).astype(np.int32)
    buf = np.where(rng.random(N) < 0.2, 0.0, rng.uniform(0, mb, N))
    hn = rng.integers(1, 30, N).astype(np.float64)
    hs = hn / rng.uniform(0.3, 5.0, N)
    cfg = oracle.mpc_cfg(B, H, V, L, mb, 1.0, 4.3, 0.0)
    hn_o, hs_o = hn.copy(), hs.copy()
    act, flat, Jm, pred = oracle.mpc_select(cfg, br, sz, chunk, prev, buf, hn_o, hs_o)
    ctl, ci = _controller(br, sz, L, mb, 4.3, 1.0, 0.0, H, chunk, prev, buf, hn, hs)
    a = ctl.next_bitrate(want_details=True).cpu().numpy()
    assert np.array_equal(ctl.last_J.cpu().numpy(), Jm)
    assert np.array_equal(ctl.last_flat.cpu().numpy().astype(np.int64), flat)
    assert np.array_equal(a, act)
    assert np.array_equal(ci.hist_n.cpu().numpy(), hn_o) and np.array_equal(ci.hist_sum_inv.cpu().numpy(), hs_o)


@pytest.mark.parametrize("B,H", [(6, 5), (4, 5), (6, 4), (5, 6), (6, 3), (3, 7), (6, 2)])
@pytest.mark.parametrize("wv,wr", [(0.0, 0.0), (1.0, 0.0), (1.0, 4.3), (0.5, 4.3)])
def test_exact_ties_resolve_to_the_first_combination(oracle, B, H, wv, wr):
    """mpc.py:171-179 keeps the FIRST minimum of the brute-force grid.  A ladder whose upper rates coincide
    (exactly representable values, same sizes) makes many combinations tie bit for bit at the optimum --
    across leaves of a group, groups of a node, nodes of a thread and threads of a lane, which are the
    four places the search kernel settles a tie in."""
    rng = np.random.default_rng(B * 10 + H)
    N, V, L, mb = 257, 24, 4.0, 20.0
    lad = np.arange(1, B + 1, dtype=np.float64) * 0.5
    lad[B - (B + 1) // 2:] = lad[-1]                      # the upper half of the ladder is one rate
    br = np.tile(lad, (V, 1))
    sz = br * L
    chunk = rng.integers(0, V - H + 1, N).astype(np.int32)
    prev = rng.integers(0, B, N).astype(np.int32)
    buf = np.where(rng.random(N) < 0.3, 0.0, rng.integers(0, 6, N) * 2.5)
    hn = np.full(N, 4.0)
    hs = hn / np.where(rng.random(N) < 0.5, 64.0, 0.25)  # predicted throughput: plenty, or so little that every combination rebuffers
    cfg = oracle.mpc_cfg(B, H, V, L, mb, wv, wr, 0.0)
    hn_o, hs_o = hn.copy(), hs.copy()
    act, flat, Jm, pred = oracle.mpc_select(cfg, br, sz, chunk, prev, buf, hn_o, hs_o)
    ties = 0
    for i in range(48):                                   # the case is what it claims to be
        _, jmin, J = oracle.mpc_brute(cfg, br, sz, chunk[i], prev[i], buf[i], pred[i])
        ties += int((J == jmin).sum() > 1)
    assert ties > 0
    ctl, ci = _controller(br, sz, L, mb, wr, wv, 0.0, H, chunk, prev, buf, hn, hs)
    a = ctl.next_bitrate(want_details=True).cpu().numpy()
    assert np.array_equal(ctl.last_J.cpu().numpy(), Jm)
    assert np.array_equal(ctl.last_flat.cpu().numpy().astype(np.int64), flat)
    assert np.array_equal(a, act)


def test_horizon_clip_and_mask(oracle):
    """D12: near the video end the reference raises IndexError; clip_horizon
    defines H_eff = min(H, V - chunk).  Masked lanes are left untouched."""
    B, H, V, L, mb = 6, 5, 12, 4.0, 20.0
    rng = np.random.default_rng(5)
    lad = np.array([0.3, 0.75, 1.2, 1.85, 2.85, 4.3])
    br = np.tile(lad, (V, 1)); sz = br * L
    chunk = np.array([7, 8, 9, 10, 11, 3], np.int32)
    N = len(chunk)
    prev = rng.integers(0, B, N).astype(np.int32)
    buf = rng.uniform(0, mb, N)
    hn = np.full(N, 4.0); hs = 4.0 / rng.uniform(0.5, 4.0, N)
    ctl, ci = _controller(br, sz, L, mb, 4.3, 1.0, 0.0, H, chunk, prev, buf, hn, hs, clip=True)
    ci.mask = torch.tensor([1, 1, 1, 1, 1, 0], dtype=torch.uint8).cuda()
    a = ctl.next_bitrate(want_details=True).cpu().numpy()
    for i in range(N - 1):
        he = min(H, V - chunk[i])
        pred, _, _ = oracle.mpc_predict_ns(H, hn[i], hs[i])
        if he >= 2:
            cfg = oracle.mpc_cfg(B, he, V, L, mb, 1.0, 4.3, 0.0)
            f, Jm, _ = oracle.mpc_brute(cfg, br, sz, chunk[i], prev[i], buf[i], pred[:he])
            assert int(ctl.last_flat[i]) == f and float(ctl.last_J[i]) == Jm
            assert a[i] == f // B ** (he - 1)
        else:
            # one step left: J[r] = -(br[r] - wv*|br[r]-br[prev]| - wr*(max(sz[r],L)/pred0 - buf))
            J = [-((lad[r] - 1.0 * abs(lad[r] - lad[prev[i]])) - 4.3 * (max(sz[0][r], L) / pred[0] - buf[i]))
                 for r in range(B)]
            assert a[i] == int(np.argmin(J)) and float(ctl.last_J[i]) == min(J)
        assert float(ci.hist_n[i]) == hn[i] + H          # history always grows by H
    assert float(ci.hist_n[N - 1]) == 4.0                # masked lane untouched
    # without clipping those lanes report action -1
    ctl2, ci2 = _controller(br, sz, L, mb, 4.3, 1.0, 0.0, H, chunk, prev, buf, hn, hs, clip=False)
    a2 = ctl2.next_bitrate().cpu().numpy()
    assert (a2[[1, 2, 3, 4]] == -1).all() and a2[0] >= 0 and a2[5] >= 0


@pytest.mark.parametrize("B,H", [(6, 5), (4, 4)])
def test_clipped_and_full_horizons_share_workgroups_with_ties(oracle, B, H):
    """Lanes near the video end (clipped horizon: exact arg-min index kept per thread) and lanes with the whole
    horizon (two-stage arg-max + first-leaf search) sit in the same workgroups, a fifth of them masked out, on a
    ladder whose upper rates coincide: every lane's flat index / J / action equals the brute-force oracle's."""
    rng = np.random.default_rng(B + 31 * H)
    N, V, L, mb = 301, 16, 4.0, 20.0
    lad = np.arange(1, B + 1, dtype=np.float64) * 0.5
    lad[B - (B + 1) // 2:] = lad[-1]
    br = np.tile(lad, (V, 1)); sz = br * L
    chunk = rng.integers(0, V - 1, N).astype(np.int32)          # V - chunk >= 2: at least two steps left
    prev = rng.integers(0, B, N).astype(np.int32)
    buf = np.where(rng.random(N) < 0.3, 0.0, rng.integers(0, 6, N) * 2.5)
    hn = np.full(N, 4.0); hs = hn / np.where(rng.random(N) < 0.5, 64.0, 0.25)
    mask = (rng.random(N) > 0.2).astype(np.uint8)
    ctl, ci = _controller(br, sz, L, mb, 4.3, 1.0, 0.0, H, chunk, prev, buf, hn, hs, clip=True)
    ci.mask = torch.from_numpy(mask).cuda()
    a = ctl.next_bitrate(want_details=True).cpu().numpy()
    flat, J = ctl.last_flat.cpu().numpy(), ctl.last_J.cpu().numpy()
    kinds = set()
    for i in range(N):
        if not mask[i]:
            assert float(ci.hist_n[i]) == 4.0                   # untouched
            continue
        he = min(H, V - int(chunk[i]))
        kinds.add(he)
        pred, _, _ = oracle.mpc_predict_ns(H, hn[i], hs[i])
        cfg = oracle.mpc_cfg(B, he, V, L, mb, 1.0, 4.3, 0.0)
        f, Jm, _ = oracle.mpc_brute(cfg, br, sz, chunk[i], prev[i], buf[i], pred[:he])
        assert int(flat[i]) == f and float(J[i]) == Jm and a[i] == f // B ** (he - 1), (i, he)
    assert H in kinds and len(kinds) >= 3


def test_mpc_drives_env_rollout(oracle):
    """8(f) rank 1: K3 reads the env's float64 state zero-copy and drives K1.
    Compared with the oracle env whose policy callback is the oracle MPC sharing
    the env's previous_bandwidths list (so D9 pollutes later predictions in both)."""
    import abrsimulator_amd as A
    from helpers import make_env
    V, H, L = 14, 5, 4.0
    ladder = [0.3, 0.75, 1.2, 1.85, 2.85, 4.3]
    rng = np.random.default_rng(77)
    traces = [rng.uniform(0.2, 6.0, 1000).astype(np.float32).astype(np.float64) for _ in range(16)]
    meta = dict(ladder=ladder, chunk_length=L, video_length=V, max_buffer=20.0, start_up_length=8.0,
                interval=1.0, weights=[4.3, 1, 1, 0.1], speed=1.0)
    N = 16
    env = make_env(meta, traces, N)
    env.reset()
    mpd = A.MPD(V, L, 20.0, 8.0, [A.Chunk(ladder, [b * L for b in ladder])] * V)
    player = A.EnvPlayer(env, mpd=mpd, qoe=A.QOEMetric(4.3, 1.0, 0.0))
    ctl = A.BatchedMPCController(player, horizon=H, clip_horizon=True)
    gpu_actions = []
    for s in range(V):
        if s == 0:
            a = torch.zeros(N, dtype=torch.int32, device="cuda")   # empty history: reference divides by zero (D13)
        else:
            a = ctl.next_bitrate()
        gpu_actions.append(a.cpu().numpy().copy())
        env.step(a)
    gpu_actions = np.stack(gpu_actions, 1)
    qoe = env.episode_qoe().cpu().numpy()

    ecfg = oracle.env_cfg(ladder, L, V, 20.0, 8.0, 1.0, [4.3, 1, 1, 0.1], 1.0)
    br = np.tile(np.array(ladder), (V, 1)); sz = br * L
    for lane in range(N):
        state = dict(n=0.0, s=0.0, seen=0)

        def policy(obs, hist):
            c = int(obs["chunk_id"])
            # fold the newly measured throughputs into the running (n, S) -- list order
            for x in hist[state["seen"]:]:
                state["s"] = state["s"] + 1.0 / x
                state["n"] += 1.0
            state["seen"] = len(hist)
            if c == 0:
                return 0
            he = min(H, V - c)
            pred, state["n"], state["s"] = oracle.mpc_predict_ns(H, state["n"], state["s"])   # D9
            if he >= 2:
                mcfg = oracle.mpc_cfg(6, he, V, L, 20.0, 1.0, 4.3, 0.0)
                f, _, _ = oracle.mpc_brute(mcfg, br, sz, c, int(obs["last_bitrate"]),
                                           float(obs["buffer_level"]), pred[:he], want_J=False)
                return f // 6 ** (he - 1)
            J = [-((ladder[r] - abs(ladder[r] - ladder[int(obs["last_bitrate"])]))
                   - 4.3 * (max(sz[0][r], L) / pred[0] - float(obs["buffer_level"]))) for r in range(6)]
            return int(np.argmin(J))

        steps, bw, acts, fin = oracle.env_episode_policy(ecfg, traces[lane % 16], 0, policy)
        assert np.array_equal(acts, gpu_actions[lane]), (lane, acts, gpu_actions[lane])
        assert np.isclose(fin["qoe"], qoe[lane], rtol=1e-10)


def test_empty_history_is_a_defined_no_decision():
    """D13: mpc.py:88,90 divide by an empty / zero history.  Defined: action -1, history untouched."""
    B, H, V, L, mb = 6, 5, 20, 4.0, 20.0
    lad = np.array([0.3, 0.75, 1.2, 1.85, 2.85, 4.3])
    br = np.tile(lad, (V, 1)); sz = br * L
    chunk = np.array([3, 3, 3], np.int32); prev = np.array([1, 1, 1], np.int32)
    buf = np.array([5.0, 5.0, 5.0]); hn = np.array([0.0, 4.0, 3.0]); hs = np.array([0.0, 2.0, 0.0])
    ctl, ci = _controller(br, sz, L, mb, 4.3, 1.0, 0.0, H, chunk, prev, buf, hn, hs)
    a = ctl.next_bitrate(want_details=True).cpu().numpy()
    assert a[0] == -1 and a[2] == -1 and a[1] >= 0
    assert ci.hist_n.cpu().numpy().tolist() == [0.0, 9.0, 3.0]
    assert np.isnan(ctl.last_J.cpu().numpy()[[0, 2]]).all() and int(ctl.last_flat[0]) == -1


def test_predictor_pre_kernel_and_single_kernel_agree(oracle):
    """abr_mpc_options.scratch_dev: the predictor as a kernel of its own == everything in one
    kernel == the oracle, incl. masked lanes, clipped horizons and "no decision" lanes."""
    B, H, V, L, mb, N = 6, 5, 30, 4.0, 20.0, 3000
    rng = np.random.default_rng(88)
    br = np.sort(rng.uniform(0.2, 6.0, B))[None, :] * rng.uniform(0.8, 1.2, (V, B))
    sz = br * L * rng.uniform(0.7, 1.3, (V, B))
    chunk = rng.integers(0, V, N).astype(np.int32)                 # some within H of the end: clipped
    prev = rng.integers(-1, B, N).astype(np.int32)
    buf = rng.uniform(0, mb, N)
    hn = rng.integers(0, 20, N).astype(np.float64)                 # some empty histories (D13)
    hs = np.where(hn > 0, hn / rng.uniform(0.3, 5.0, N), 0.0)
    mask = (rng.random(N) < 0.9).astype(np.uint8)
    res = []
    for scratch in (True, False):
        ctl, ci = _controller(br, sz, L, mb, 4.3, 1.0, 0.0, H, chunk, prev, buf, hn, hs, clip=True)
        ci.mask = torch.from_numpy(mask).cuda()
        ctl.use_scratch = scratch
        a = ctl.next_bitrate(want_details=True)
        res.append((a.cpu().numpy(), ctl.last_flat.cpu().numpy(), ctl.last_J.cpu().numpy(),
                    ci.hist_n.cpu().numpy(), ci.hist_sum_inv.cpu().numpy()))
    m = mask.astype(bool)
    for x, y in zip(res[0], res[1]):
        assert np.array_equal(x[m], y[m], equal_nan=True)
    assert np.array_equal(res[0][3][~m], hn[~m])                    # masked lanes untouched
    full = m & (chunk + H <= V) & (hn > 0)
    cfg = oracle.mpc_cfg(B, H, V, L, mb, 1.0, 4.3, 0.0)
    hn_o, hs_o = hn[full].copy(), hs[full].copy()
    act, flat, Jm, _ = oracle.mpc_select(cfg, br, sz, chunk[full], prev[full], buf[full], hn_o, hs_o)
    assert np.array_equal(res[0][0][full], act) and np.array_equal(res[0][2][full], Jm)
    assert np.array_equal(res[0][1][full].astype(np.int64), flat)
    assert (res[0][0][m & (hn == 0)] == -1).all()
