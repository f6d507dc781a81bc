"""Fuzzes csrc/abr_exact_jump.h (the closed-form 'add a float64 constant n times')
on the CPU against the naive one-addition-per-tick loop: results must be
bit-identical, including round-to-even ties, binade crossings, thresholds that
sit exactly on reachable values, and the simulator's own constants."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "native", "exact_jump_harness.cpp")
SO = os.path.join(ROOT, "tests", "native", "libexact_jump_harness.so")


@pytest.fixture(scope="module")
def H():
    inc = os.path.join(ROOT, "abrsimulator_amd", "csrc")
    if (not os.path.exists(SO) or os.path.getmtime(SO) < max(
            os.path.getmtime(SRC), os.path.getmtime(os.path.join(inc, "abr_exact_jump.h")))):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-fno-fast-math", "-I", inc, SRC, "-o", SO])
    lib = C.CDLL(SO)
    for f in (lib.fuzz_ge, lib.fuzz_le, lib.fuzz_lt, lib.fuzz_biased, lib.overshoot_count):
        f.restype = C.c_int64
    return lib


def _run(fn, x0, c, thr, n):
    x0 = np.ascontiguousarray(x0, np.float64); c = np.ascontiguousarray(c, np.float64)
    thr = np.ascontiguousarray(thr, np.float64); n = np.ascontiguousarray(n, np.int32)
    N = len(x0)
    xo = np.zeros(N); ao = np.zeros(N, np.int32); ho = np.zeros(N, np.uint8)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    bad = fn(P(x0, C.c_double), P(c, C.c_double), P(thr, C.c_double), P(n, C.c_int32), C.c_int64(N),
             P(xo, C.c_double), P(ao, C.c_int32), P(ho, C.c_uint8))
    assert bad == -1, f"mismatch at case {bad}: x0={x0[bad]!r} c={c[bad]!r} thr={thr[bad]!r} n={n[bad]}"
    return xo, ao, ho


def test_download_accumulation_shapes(H):
    """downloaded_size += bandwidth*dt until >= bitrate*L (Simulator.py:160-163)."""
    rng = np.random.default_rng(1)
    N = 400_000
    bw = rng.uniform(0.05, 12.0, N).astype(np.float32).astype(np.float64)
    c = bw * 0.01
    x0 = np.where(rng.random(N) < 0.4, 0.0, rng.uniform(0, 20, N))
    thr = rng.choice([0.3, 0.75, 1.2, 1.85, 2.85, 4.3, 1.0, 2.5, 5.0, 8.0], N) * rng.choice([1.0, 2.0, 3.0, 4.0], N)
    n = rng.integers(1, 3000, N)
    x, a, h = _run(H.fuzz_ge, x0, c, thr, n)
    assert h.mean() > 0.2 and (a > 50).mean() > 0.3      # the jumps really ran


def test_buffer_drain_shapes(H):
    """buffer_level -= speed*dt until <= 0 (Simulator.py:184,194) or < max_buffer (:190)."""
    rng = np.random.default_rng(2)
    N = 400_000
    x0 = np.where(rng.random(N) < 0.5, rng.integers(1, 7, N) * 4.0, rng.uniform(0.001, 30, N))
    sd = rng.choice([0.01, 0.0125, 0.005, 0.02, 1.25 * 0.01, 0.75 * 0.01], N)
    n = rng.integers(1, 4000, N)
    _run(H.fuzz_le, x0, -sd, np.zeros(N), n)
    thr = rng.choice([20.0, 3.0, 5.0, 9.0, 6.0], N)
    x0b = thr + rng.uniform(0, 5, N)
    _run(H.fuzz_lt, x0b, -sd, thr, n)
    _run(H.fuzz_le, x0b, -sd, thr, n)


def test_ties_and_exact_thresholds(H):
    """Constants whose low bit is exactly half an ulp of the running value (round
    to even), and thresholds placed exactly on / one ulp around reachable values."""
    rng = np.random.default_rng(3)
    N = 200_000
    e = rng.integers(-8, 6, N)
    base = np.ldexp(1.0, e)
    # x0 = base * (1 + k*2^-52): odd and even mantissas; c = q*u + u/2 (a tie in binade e)
    kmant = rng.integers(0, 1 << 20, N)
    x0 = base * (1.0 + kmant * 2.0 ** -52)
    u = base * 2.0 ** -52
    q = rng.integers(1, 1 << 44, N).astype(np.float64)
    c = q * u + u / 2
    n = rng.integers(1, 500, N)
    thr = x0 + c * rng.integers(1, 600, N)              # near a reachable value
    _run(H.fuzz_ge, x0, c, thr, n)
    _run(H.fuzz_ge, x0, c, np.nextafter(thr, np.inf), n)
    _run(H.fuzz_ge, x0, c, np.nextafter(thr, -np.inf), n)
    # decreasing, ties, thresholds on reachable values
    x1 = base * (2.0 - kmant * 2.0 ** -52)
    thr2 = np.maximum(x1 - c * rng.integers(1, 600, N), 0.0)
    for t in (thr2, np.nextafter(thr2, np.inf), np.nextafter(thr2, -np.inf)):
        _run(H.fuzz_le, x1, -c, t, n)
        _run(H.fuzz_lt, x1, -c, t, n)


def test_wild_ranges(H):
    rng = np.random.default_rng(4)
    N = 300_000
    x0 = np.ldexp(rng.uniform(1, 2, N), rng.integers(-40, 40, N)) * (rng.random(N) > 0.1)
    c = np.ldexp(rng.uniform(1, 2, N), rng.integers(-45, 30, N))
    n = rng.integers(1, 2000, N)
    thr = x0 + c * rng.uniform(0, 3000, N)
    _run(H.fuzz_ge, x0, c, thr, n)
    thr = np.maximum(x0 - c * rng.uniform(0, 3000, N), -1.0)
    _run(H.fuzz_le, x0, -c, thr, n)
    _run(H.fuzz_lt, x0, -c, thr, n)
    # tiny and subnormal neighbourhoods
    x0 = rng.uniform(0, 1e-300, N); c = rng.uniform(1e-310, 1e-302, N)
    _run(H.fuzz_ge, x0, c, rng.uniform(0, 1e-299, N), n)
    _run(H.fuzz_le, x0, -c, np.zeros(N), n)


def test_landing_exactly_on_a_power_of_two(H):
    """A steady downward step that lands exactly on 2^e is NOT an in-binade step: the
    exact difference lies just below 2^e, on the finer grid of the next binade
    (found by the env goldens: 2.0299999999999994 - 3 * 0.01)."""
    rng = np.random.default_rng(5)
    xs, cs, ns = [], [], []
    for e in range(-6, 7):
        y = 2.0 ** e
        for c in list(rng.uniform(0.001, 0.2, 40) * y) + [0.01, 0.0125, 0.005]:
            if not (c < y / 8):
                continue
            x = y * 1.5
            x1 = x - c; x2 = x1 - c
            d = x1 - x2                       # the steady in-binade decrement
            for j in (1, 2, 3, 7, 20):
                x0 = y + j * d
                if x0 < 2 * y:
                    xs.append(x0); cs.append(c); ns.append(j + 50)
    xs = np.array(xs); cs = np.array(cs); ns = np.array(ns, np.int32)
    _run(H.fuzz_le, xs, -cs, np.zeros(len(xs)), ns)
    _run(H.fuzz_lt, xs, -cs, xs / 3, ns)
    assert _run(H.fuzz_le, [2.0299999999999994], [-0.01], [0.0], [72])[0][0] == 1.3099999999999992
    # the upward twin: landing exactly on 2^(e+1) from below
    _run(H.fuzz_ge, 2 * np.array(xs) - (xs - 0) , cs, np.full(len(xs), 1e9), ns)
    xs2 = []
    for x0, c in zip(xs, cs):
        y = 2.0 ** np.ceil(np.log2(x0))
        x = y * 0.75; x1 = x + c; x2 = x1 + c; d = x2 - x1
        xs2.append(y - 3 * d)
    _run(H.fuzz_ge, np.array(xs2), cs, np.full(len(xs), 1e9), ns)


def _run_biased(H, kind, bias, x0, c, thr, n):
    import functools
    fn = functools.partial(H.fuzz_biased, C.c_int32(kind), C.c_int32(bias))
    return _run(fn, x0, c, thr, n)


def test_spoiled_estimates_are_repaired_and_good_ones_never_overshoot(H):
    """The settlement of a segment trusts the jump-length estimate for speed only.  With the estimate
    spoiled by +4 (candidate 0 lands outside: the repair walks back) or -4 (the jump is shorter than it
    could be: the next segment carries on) the chain is still the naive loop bit for bit; and on
    unspoiled inputs -- ties, exact thresholds, power-of-two landings, long jumps -- the repair path is
    never entered (the counter the harness keeps stays 0), which is what makes it free."""
    rng = np.random.default_rng(6)
    N = 120_000
    c = rng.uniform(0.05, 12.0, N).astype(np.float32).astype(np.float64) * 0.01
    x0 = np.where(rng.random(N) < 0.4, 0.0, rng.uniform(0, 20, N))
    thr = rng.choice([0.3, 0.75, 1.2, 1.85, 2.85, 4.3], N) * rng.choice([1.0, 2.0, 4.0], N)
    n = rng.integers(1, 3000, N)
    xb = np.where(rng.random(N) < 0.5, rng.integers(1, 7, N) * 4.0, rng.uniform(0.001, 30, N))
    sd = rng.choice([0.01, 0.0125, 0.005, 0.02], N)
    t2 = rng.choice([20.0, 3.0, 5.0, 9.0], N)
    e = rng.integers(-8, 6, N)
    base = np.ldexp(1.0, e)
    u = base * 2.0 ** -52
    xt = base * (1.0 + rng.integers(0, 1 << 20, N) * 2.0 ** -52)
    ct = rng.integers(1, 1 << 44, N).astype(np.float64) * u + u / 2           # ties
    thr_t = xt + ct * rng.integers(1, 600, N)
    H.overshoot_count(1)
    _run(H.fuzz_ge, x0, c, thr, n)
    _run(H.fuzz_le, xb, -sd, np.zeros(N), rng.integers(1, 4000, N))
    _run(H.fuzz_lt, t2 + rng.uniform(0, 5, N), -sd, t2, rng.integers(1, 4000, N))
    _run(H.fuzz_ge, xt, ct, thr_t, rng.integers(1, 500, N))
    # jumps of millions of steps inside one binade (cut at kJumpCap per segment)
    _run(H.fuzz_ge, [1.0, 1.0, 1.5, 1024.0], [2.0 ** -30, 2.0 ** -29 + 2.0 ** -52, 3 * 2.0 ** -31, 2.0 ** -17],
         [10.0, 1.2, 1.9, 1e9], np.array([1 << 23, (1 << 23) + 12345, 1 << 22, 1 << 23], np.int32))
    assert H.overshoot_count(1) == 0
    for bias in (4, -4):
        _run_biased(H, 0, bias, x0, c, thr, n)
        _run_biased(H, 1, bias, xb, -sd, np.zeros(N), rng.integers(1, 4000, N))
        _run_biased(H, 2, bias, t2 + rng.uniform(0, 5, N), -sd, t2, rng.integers(1, 4000, N))
        _run_biased(H, 0, bias, xt, ct, thr_t, rng.integers(1, 500, N))
        over = H.overshoot_count(1)
        assert (over > 1000) if bias > 0 else (over == 0), (bias, over)
