"""The event-driven lane step (csrc/abr_lane_jump.h), compiled for the HOST, against
the reference-generated goldens and against the C oracle on large seeded sets.
This is the same source the HIP kernels compile for gfx950; running it on the CPU
lets the exactness claim be fuzzed over millions of lane-steps without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ENV_GOLDENS, ROOT, load_golden

SRC = os.path.join(ROOT, "tests", "native", "lane_jump_harness.cpp")
SO = os.path.join(ROOT, "tests", "native", "liblane_jump_harness.so")
INC = os.path.join(ROOT, "abrsimulator_amd", "csrc")


@pytest.fixture(scope="module")
def H():
    deps = [SRC] + [os.path.join(INC, f) for f in ("abr_lane_jump.h", "abr_exact_jump.h", "abr_tick_tables.h")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-fno-fast-math", "-I", INC, SRC, "-o", SO])
    lib = C.CDLL(SO)
    lib.lj_create.restype = C.c_void_p
    lib.lj_batch.restype = C.c_int64
    return lib


def run_jump(H, meta, traces, trace_id, offset, actions, max_ticks=0, speeds=None, sched=None):
    """sched: optional [N, rows] per-lane speed schedules (played chunk p plays at sched[i, min(p, rows-1)])."""
    from oracle.oracle import pack_traces
    ladder = np.asarray(meta["ladder"], np.float64)
    V = meta["video_length"]
    if not max_ticks:
        max_ticks = int(32 * V * np.ceil(meta["chunk_length"] / 0.01))
    h = H.lj_create(C.c_double(meta["interval"]), C.c_double(meta["chunk_length"]),
                    C.c_double(meta.get("speed", 1.0)), C.c_int32(V), C.c_double(meta["max_buffer"]),
                    C.c_double(meta["start_up_length"]), C.c_int32(max_ticks),
                    ladder.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(len(ladder)))
    flat, off, lens = pack_traces(traces)
    trace_id = np.ascontiguousarray(trace_id, np.int32); offset = np.ascontiguousarray(offset, np.int32)
    actions = np.ascontiguousarray(actions, np.int32)
    N = actions.shape[0]
    rec = np.zeros((N, V, 8)); bw = np.zeros((N, V)); fin = np.zeros((N, 6)); fin_i = np.zeros((N, 2), np.int32)
    if sched is not None:
        sched = np.ascontiguousarray(sched, np.float64)
        assert sched.shape[0] == N
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    rc = H.lj_batch(C.c_void_p(h), P(flat, C.c_double), P(off, C.c_int64), P(lens, C.c_int32),
                    P(trace_id, C.c_int32), P(offset, C.c_int32), P(actions, C.c_int32), C.c_int32(N),
                    P(rec, C.c_double), P(bw, C.c_double), P(fin, C.c_double), P(fin_i, C.c_int32),
                    P(np.ascontiguousarray(speeds, np.float64), C.c_double) if speeds is not None else None,
                    P(sched, C.c_double) if sched is not None else None,
                    C.c_int32(sched.shape[1] if sched is not None else 0))
    H.lj_destroy(C.c_void_p(h))
    assert rc == 0, rc
    return rec, bw, fin, (fin_i if sched is not None else fin_i[:, 0])


def _check(rec, bw, fin, steps, bwo, fino, sd=0.01):
    names = ["global_time", "rebuffer_time", "start_up_time", "play_time", "buffer_level", "last_bandwidth"]
    for c, k in enumerate(names):
        bad = np.argwhere(rec[:, :, c] != steps[k])
        assert bad.size == 0, (k, bad[:3], rec[tuple(bad[0])][c], steps[k][tuple(bad[0])])
    fl = rec[:, :, 7].astype(np.int32)
    assert np.array_equal(fl & 1, steps["start_up"]) and np.array_equal((fl >> 1) & 1, steps["buffer_empty"])
    assert np.array_equal((fl >> 2) & 1, steps["buffer_full"])
    assert np.array_equal(bw, bwo)
    for c, k in enumerate(["global_time", "rebuffer_time", "start_up_time", "play_time", "buffer_level"]):
        assert np.array_equal(fin[:, c], fino[k]), k


@pytest.mark.parametrize("name", ENV_GOLDENS)
def test_goldens_bit_exact(H, name):
    m, g = load_golden(name)
    rec, bw, fin, fin_i = run_jump(H, m, list(g["traces"]), g["trace_id"], g["offset"], g["actions"])
    steps = {k: g[k] for k in ["global_time", "rebuffer_time", "start_up_time", "play_time",
                               "buffer_level", "start_up", "buffer_empty", "buffer_full"]}
    steps["last_bandwidth"] = g["arg_last_bandwidth"]
    fino = {k: g["final_" + k] for k in ["global_time", "rebuffer_time", "start_up_time", "play_time",
                                         "buffer_level"]}
    _check(rec, bw, fin, steps, g["final_bandwidths"], fino)
    # the integer latency integral reproduces average_latency to ~1e-12
    sd = m.get("speed", 1.0) * 0.01
    n_play = fin_i.astype(np.float64)
    lat = (0.01 * fin[:, 5] - sd * (n_play * (n_play - 1) / 2)) / fin[:, 3]
    assert np.allclose(lat, g["final_average_latency"], rtol=1e-9)


CASES = [
    dict(seed=1, N=4000, V=24), dict(seed=2, N=3000, V=16, ragged=True),
    dict(seed=3, N=3000, V=40, L=1.0, start_up=2.0, ladder=(1, 2.5, 5, 8), bw=(0.5, 10.0)),
    dict(seed=4, N=1500, V=12, interval=0.3, bw=(0.1, 1.5)),
    dict(seed=5, N=2000, V=20, max_buffer=5.0, start_up=4.0, bw=(2.0, 12.0)),
    dict(seed=6, N=2000, V=20, L=2.0, max_buffer=3.0, start_up=4.0, interval=0.5, bw=(2.0, 12.0)),
    dict(seed=7, N=1500, V=12, L=2.5, interval=0.7),
    dict(seed=8, N=1500, V=12, speed=1.25), dict(seed=9, N=1500, V=12, speed=0.8, bw=(1.0, 8.0)),
    dict(seed=10, N=1500, V=16, round_bw=True),          # integer-ish bandwidths: knife edges
    dict(seed=11, N=1500, V=16, L=3.0, interval=0.05, start_up=3.0, max_buffer=9.0),
]


def _case(seed, N, V=12, L=4.0, interval=1.0, n_traces=16, ragged=False, max_buffer=20.0,
          start_up=8.0, bw=(0.2, 6.0), ladder=(0.3, 0.75, 1.2, 1.85, 2.85, 4.3), speed=1.0,
          round_bw=False):
    rng = np.random.default_rng(seed)
    lens = rng.integers(300, 3001, n_traces) if ragged else np.full(n_traces, 4000)
    if round_bw:
        traces = [rng.choice([0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 4.8, 6.0], l) for l in lens]
    else:
        traces = [rng.uniform(bw[0], bw[1], l).astype(np.float32).astype(np.float64) for l in lens]
    meta = dict(ladder=list(ladder), chunk_length=L, video_length=V, max_buffer=max_buffer,
                start_up_length=start_up, interval=interval, weights=[4.3, 1, 1, 0.1], speed=speed)
    trace_id = rng.integers(0, n_traces, N).astype(np.int32)
    offset = np.array([rng.integers(0, lens[t]) for t in trace_id], np.int32)
    actions = rng.integers(0, len(ladder), (N, V)).astype(np.int32)
    return meta, traces, trace_id, offset, actions


@pytest.mark.parametrize("case", CASES)
def test_seeded_against_oracle(H, oracle, case):
    meta, traces, trace_id, offset, actions = _case(**case)
    cfg = oracle.env_cfg(meta["ladder"], meta["chunk_length"], meta["video_length"], meta["max_buffer"],
                         meta["start_up_length"], meta["interval"], meta["weights"], meta["speed"])
    steps, bwo, fino, _ = oracle.env_batch(cfg, traces, trace_id, offset, actions)
    rec, bw, fin, fin_i = run_jump(H, meta, traces, trace_id, offset, actions)
    _check(rec, bw, fin, steps, bwo, fino)


@pytest.mark.parametrize("seed", [21, 22])
def test_per_lane_speeds_against_oracle(H, oracle, seed):
    """8f rank 3: one constant play speed per lane; play_time is then carried by the exact
    chain instead of the shared GP table."""
    meta, traces, trace_id, offset, actions = _case(seed=seed, N=1500, V=14, bw=(0.5, 7.0))
    speeds = np.random.default_rng(seed).choice([0.75, 0.8, 1.0, 1.1, 1.25, 1.3, 0.9173], len(trace_id))
    cfg = oracle.env_cfg(meta["ladder"], meta["chunk_length"], meta["video_length"], meta["max_buffer"],
                         meta["start_up_length"], meta["interval"], meta["weights"], 1.0)
    steps, bwo, fino, _ = oracle.env_batch(cfg, traces, trace_id, offset, actions, speeds=speeds)
    rec, bw, fin, fin_i = run_jump(H, meta, traces, trace_id, offset, actions, speeds=speeds)
    _check(rec, bw, fin, steps, bwo, fino)


def _random_config(rng):
    """A random but playable configuration (the reference loops forever when
    max_buffer < start_up_length, so keep start_up_length <= max_buffer)."""
    L = float(rng.choice([1.0, 2.0, 2.5, 3.0, 4.0, 6.0]))
    interval = float(rng.choice([0.05, 0.25, 0.3, 0.5, 0.7, 1.0, 2.0, 3.7]))
    B = int(rng.integers(2, 9))
    ladder = np.sort(rng.uniform(0.2, 8.0, B)).round(3).tolist()
    max_buffer = float(rng.choice([L * 1.5, L * 3, 20.0, 7.3]))
    start_up = float(min(max_buffer, rng.choice([0.0, L, 2 * L, 1.7])))
    bw_lo = float(rng.choice([0.1, 0.5, 2.0]))
    bw_hi = bw_lo * float(rng.choice([3.0, 10.0, 40.0]))
    speed = float(rng.choice([1.0, 1.0, 0.8, 1.25]))
    V = int(rng.integers(2, 20))
    return dict(ladder=ladder, chunk_length=L, video_length=V, max_buffer=max_buffer,
                start_up_length=start_up, interval=interval, weights=[4.3, 1, 1, 0.1],
                speed=speed), (bw_lo, bw_hi)


@pytest.mark.parametrize("seed", range(60))
def test_random_configurations_against_oracle(H, oracle, seed):
    """Config-space fuzz: chunk lengths, trace intervals (shorter than the 32-addition
    prologue, longer than a chunk, non-representable), ladders, buffer limits, speeds,
    ragged traces with wrap-around."""
    rng = np.random.default_rng(1000 + seed)
    meta, (lo, hi) = _random_config(rng)
    n_traces, N = 6, 200
    lens = rng.integers(40, 3000, n_traces)
    traces = [rng.uniform(lo, hi, l).astype(np.float32).astype(np.float64) for l in lens]
    trace_id = rng.integers(0, n_traces, N).astype(np.int32)
    offset = np.array([rng.integers(0, lens[t]) for t in trace_id], np.int32)
    actions = rng.integers(0, len(meta["ladder"]), (N, meta["video_length"])).astype(np.int32)
    cfg = oracle.env_cfg(meta["ladder"], meta["chunk_length"], meta["video_length"], meta["max_buffer"],
                         meta["start_up_length"], meta["interval"], meta["weights"], meta["speed"])
    steps, bwo, fino, _ = oracle.env_batch(cfg, traces, trace_id, offset, actions, max_ticks=4_000_000)
    # generous tick bound: starved lanes take long
    rec, bw, fin, fin_i = run_jump(H, meta, traces, trace_id, offset, actions,
                                   max_ticks=int(fino["ticks"].max()) + 1000)
    _check(rec, bw, fin, steps, bwo, fino)


def test_speed_schedule_golden_bit_exact(H):
    """The event-driven lane logic with a speed that changes at every played chunk, against the
    reference run with a scripted get_next_speed() (tests/golden/env_speed_schedule)."""
    m, g = load_golden("env_speed_schedule")
    rec, bw, fin, fin_i = run_jump(H, m, list(g["traces"]), g["trace_id"], g["offset"], g["actions"],
                                   sched=g["speed_sched"])
    names = ["global_time", "rebuffer_time", "start_up_time", "play_time", "buffer_level"]
    for c, k in enumerate(names):
        assert np.array_equal(rec[:, :, c], g[k]), k
    assert np.array_equal(rec[:, :, 5], g["arg_last_bandwidth"])
    assert np.array_equal(bw, g["final_bandwidths"])
    for c, k in enumerate(names):
        assert np.array_equal(fin[:, c], g["final_" + k]), k
    assert np.array_equal(fin_i[:, 1], g["final_play_id"])
    # (average_latency from the carried sums is checked on the device path, tests/test_env_gpu.py)


@pytest.mark.parametrize("seed", [31, 32, 33])
def test_speed_schedule_against_oracle(H, oracle, seed):
    rng = np.random.default_rng(seed)
    meta = dict(ladder=[0.3, 0.75, 1.2, 1.85, 2.85, 4.3], chunk_length=float(rng.choice([2.0, 4.0, 1.0])),
                video_length=14, max_buffer=float(rng.choice([20.0, 6.0])), start_up_length=4.0,
                interval=float(rng.choice([1.0, 0.5])), weights=[4.3, 1, 1, 0.1], speed=1.0)
    N, rows = 300, int(rng.integers(2, 9))
    traces = [rng.uniform(0.3, 7.0, 1500).astype(np.float32).astype(np.float64) for _ in range(5)]
    trace_id = rng.integers(0, 5, N).astype(np.int32)
    offset = rng.integers(0, 1500, N).astype(np.int32)
    actions = rng.integers(0, 6, (N, 14)).astype(np.int32)
    sched = rng.choice([0.5, 0.75, 0.8, 1.0, 1.1, 1.25, 1.5, 2.0, 0.9173], (N, rows))
    cfg = oracle.env_cfg(meta["ladder"], meta["chunk_length"], 14, meta["max_buffer"], 4.0, meta["interval"],
                         meta["weights"], 1.0)
    steps, bwo, fino, _ = oracle.env_batch(cfg, traces, trace_id, offset, actions, speeds=sched)
    rec, bw, fin, fin_i = run_jump(H, meta, traces, trace_id, offset, actions, sched=sched)
    for c, k in enumerate(["global_time", "rebuffer_time", "start_up_time", "play_time", "buffer_level"]):
        assert np.array_equal(rec[:, :, c], steps[k]), k
        assert np.array_equal(fin[:, c], fino[k]), k
    assert np.array_equal(bw, bwo)
    assert np.array_equal(fin_i[:, 1], fino["play_id"])


def test_call_site_prediction_of_the_download_side(H, oracle):
    """lanej_predict_next_call (round 5): the download side's exact prediction of the next call site for lanes whose
    buffer sits at max_buffer (buffer_full gates almost every decision, Simulator.py:143-145).  The harness runs it at
    EVERY decision of every episode above and returns an error on a wrong tick or on a gated step of a playing lane
    that the cheap test misses; here: workloads in which gating is frequent, and the coverage numbers."""
    stats = (C.c_longlong * 5)()
    H.lj_predict_stats(stats, 1)
    for kw in (dict(seed=31, N=800, V=48, max_buffer=20.0, bw=(3.0, 6.0), ladder=(0.3, 0.75, 1.2, 1.85, 2.85, 4.3)),
               dict(seed=32, N=800, V=24, L=2.0, max_buffer=6.0, start_up=2.0, interval=0.3, bw=(1.0, 8.0)),
               dict(seed=33, N=800, V=24, L=3.0, max_buffer=9.0, start_up=3.0, interval=0.05, speed=1.25, bw=(2.0, 9.0)),
               dict(seed=34, N=800, V=16, ragged=True, max_buffer=8.0, start_up=8.0, bw=(0.2, 6.0))):
        meta, traces, trace_id, offset, actions = _case(**kw)
        cfg = oracle.env_cfg(meta["ladder"], meta["chunk_length"], meta["video_length"], meta["max_buffer"],
                             meta["start_up_length"], meta["interval"], meta["weights"], meta["speed"])
        steps, bwo, fino, _ = oracle.env_batch(cfg, traces, trace_id, offset, actions)
        rec, bw, fin, fin_i = run_jump(H, meta, traces, trace_id, offset, actions)      # raises on a wrong prediction
        _check(rec, bw, fin, steps, bwo, fino)
    H.lj_predict_stats(stats, 0)
    decisions, gated, covered, made, wrong = list(stats)
    assert wrong == 0
    # made > gated: buffer_full at the completing tick that the wait for availability clears again; uncovered: gated out of start-up
    assert gated > 2000 and made >= covered > 0.9 * gated, list(stats)


def test_drain_to_zero_against_the_plain_loop(H):
    """drain_to_zero (exact jumps far from zero, plain subtractions near it) against `buffer_level -= speed * dt` tick by
    tick (Simulator.py:184, :194): the same number of ticks, the same ran-dry answer, the same float64 value.  Half of the
    cases start where the workload's buffers really are: within a few ulps of k * sd, and on values the rounded sequence
    itself visits on its way to zero -- there the tick at which the result is first <= 0 is decided by the roundings."""
    rng = np.random.default_rng(77)
    n = 400_000
    sd = np.where(rng.random(n) < 0.5, 0.01, rng.uniform(0.003, 0.03, n))
    b0 = rng.uniform(0.0, 1.0, n) ** 2 * 40.0
    m = rng.integers(0, 2500, n).astype(np.int32)
    # adversarial: multiples of sd, nudged by -3..3 ulps
    k = rng.integers(1, 2400, n)
    adv = k * sd
    adv = np.nextafter(adv, np.where(rng.random(n) < 0.5, np.inf, -np.inf))
    for _ in range(2):
        adv = np.where(rng.random(n) < 0.5, np.nextafter(adv, np.inf), adv)
    # ... and values of the sequence run BACKWARDS from a tiny remainder: b0 = r + sd + sd + ... (k times, rounded each time),
    # so that the forward sequence lands within rounding of r, on either side of zero
    r = rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-17, -9, n)
    back = r.copy()
    kk = rng.integers(1, 600, n)
    for j in range(600):
        back = np.where(j < kk, back + sd, back)
    B0 = np.concatenate([b0, adv, back, adv, back])
    SD = np.concatenate([sd] * 5)
    M = np.concatenate([m, (k + rng.integers(-2, 3, n)).astype(np.int32), (kk + rng.integers(-2, 3, n)).astype(np.int32),
                        np.full(n, 5000, np.int32), np.full(n, 5000, np.int32)]).astype(np.int32)
    M = np.maximum(M, 0).astype(np.int32)
    H.lj_drain_check.restype = C.c_int64

    def check(b_, sd_, m_):
        keep = b_ > 0.0                  # the function is only called on a playing lane (buffer_level > 0)
        b_, sd_, m_ = (np.ascontiguousarray(a[keep]) for a in (b_, sd_, np.maximum(m_, 0).astype(np.int32)))
        stats = (C.c_longlong * 2)()
        bad = H.lj_drain_check(b_.ctypes.data_as(C.POINTER(C.c_double)), sd_.ctypes.data_as(C.POINTER(C.c_double)),
                               m_.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int64(len(b_)), stats)
        assert bad == -1, (bad, b_[bad], sd_[bad], m_[bad])
        return stats[0]

    assert check(b0, sd, m) > 50_000                     # random starts
    assert check(B0[n:], SD[n:], M[n:]) > 500_000         # near-multiples of sd and the sequence's own values


def test_drain_cascade_against_the_plain_loop(H):
    """drain_cascade (round 6: the per-binade table of `buffer_level -= speed * dt` at ONE play speed: every subtraction
    whose exact result lies in binade e takes off exactly RN(sd / u_e) u_e, the crossing step included) against the plain
    loop (Simulator.py:184, :194): the same ticks, the same ran-dry answer, the same float64 value -- for the simulator's own
    sd, for speeds whose sd has trailing zero bits (exact binades, a tie binade inside the range), and for starts that sit
    on binade boundaries, on the stage thresholds B_e = 2^e + ceil(sd / u_e) u_e, within ulps of k * sd, and on values the
    rounded sequence itself visits on its way down."""
    rng = np.random.default_rng(78)
    H.lj_cascade_check.restype = C.c_int64
    H.lj_cascade_check.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int64,
                                   C.POINTER(C.c_longlong)]
    n = 120_000
    total_dry = 0
    for sd, max_level in ((0.01, 24.0), (1.25 * 0.01, 24.0), (0.75 * 0.01, 9.0), (0.01, 1000.0), (2.0 * 0.01, 64.0),
                          (0.0078125, 24.0), (0.01171875, 30.0), (3.0 * 0.01, 5.0), (0.5 * 0.01, 1.0e5), (1.1 * 0.01, 24.0)):
        top = 2.0 ** np.floor(np.log2(max_level) + 1)
        b_rand = rng.uniform(0.0, 1.0, n) ** 2 * min(max_level * 1.2, top)
        k = rng.integers(1, int(min(max_level, 60.0) / sd), n)
        adv = k * sd
        for _ in range(3):
            adv = np.where(rng.random(n) < 0.5, np.nextafter(adv, np.where(rng.random(n) < 0.5, np.inf, -np.inf)), adv)
        # binade boundaries and the stage thresholds, nudged by up to two ulps either way
        e = rng.integers(-8, int(np.log2(top)), n)
        base = np.ldexp(1.0, e)
        u = base * 2.0 ** -52
        S = (base + sd) - base
        T = np.where(S >= sd, S, S + u)
        edge = np.where(rng.random(n) < 0.5, base, base + T) + rng.integers(-2, 3, n) * u
        edge = np.where(rng.random(n) < 0.3, edge + rng.integers(0, 50, n) * S, edge)
        # values of the sequence run BACKWARDS from a tiny remainder
        r = rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-17, -9, n)
        back = r.copy()
        kk = rng.integers(1, 900, n)
        for j in range(900):
            back = np.where(j < kk, back + sd, back)
        # sums of chunk lengths minus ticks, as the simulator's buffers are made
        sim = np.zeros(n)
        for _ in range(4):
            sim = sim + 4.0
            t_ = rng.integers(0, 300, n)
            for j in range(300):
                sim = np.where(j < t_, sim - sd, sim)
        B0 = np.concatenate([b_rand, adv, edge, back, sim])
        M = np.concatenate([rng.integers(0, 3000, n), k + rng.integers(-2, 3, n), rng.integers(0, 5000, n),
                            kk + rng.integers(-2, 3, n), rng.integers(0, 2500, n)])
        keep = B0 > 0.0
        B0 = np.ascontiguousarray(B0[keep]); M = np.ascontiguousarray(np.maximum(M[keep], 0).astype(np.int32))
        stats = (C.c_longlong * 3)()
        bad = H.lj_cascade_check(sd, max_level, B0.ctypes.data_as(C.POINTER(C.c_double)),
                                 M.ctypes.data_as(C.POINTER(C.c_int32)), len(B0), stats)
        assert bad == -1, (sd, max_level, bad, None if bad < 0 else (B0[bad].hex(), int(M[bad])), list(stats))
        assert stats[1] >= 3 and stats[2] < 0.2 * len(B0), (sd, max_level, list(stats))
        total_dry += stats[0]
    assert total_dry > 500_000
    # no table where the cascade does not apply: the callers fall back to the chains
    stats = (C.c_longlong * 3)()
    z = np.zeros(1); zi = np.zeros(1, np.int32)
    for sd, max_level in ((0.01, 1.0e12), (1.0e-320, 24.0), (0.01, 0.001)):
        assert H.lj_cascade_check(sd, max_level, z.ctypes.data_as(C.POINTER(C.c_double)), zi.ctypes.data_as(C.POINTER(C.c_int32)), 1, stats) == -2
