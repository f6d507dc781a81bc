"""The DEVICE build of the exact chain (csrc/abr_exact_jump.h: v_rcp_f64 estimate, saturating
convert, three-candidate settlement, in-line repair) against the naive one-addition-per-tick loop
run on the host, through the diagnostic entry point abr_debug_chain.  estimate_bias = +4 / -4
spoils the jump-length estimate on purpose: +4 makes candidate 0 land outside, so the repair path
-- never entered by real inputs, whose estimates are within one of the answer -- runs on the
device; -4 makes every jump shorter than it could be, which the next segment makes up for."""
import ctypes as C

import numpy as np
import pytest
import torch

from test_exact_jump import H, _run  # noqa: F401  (host harness: jump chain == naive loop, asserted)

pytestmark = pytest.mark.gpu

GE, LE, LT = 0, 1, 2


def _device(kind, bias, x0, c, thr, n):
    from abrsimulator_amd import _lib
    lib = _lib.lib()
    t = lambda a, d: torch.from_numpy(np.ascontiguousarray(a, d)).cuda()
    x0, c, thr, n = t(x0, np.float64), t(c, np.float64), t(thr, np.float64), t(n, np.int32)
    N = x0.numel()
    xo = torch.empty(N, dtype=torch.float64, device="cuda")
    ao = torch.empty(N, dtype=torch.int32, device="cuda")
    ho = torch.empty(N, dtype=torch.uint8, device="cuda")
    _lib.check(lib.abr_debug_chain(kind, bias, _lib.ptr(x0), _lib.ptr(c), _lib.ptr(thr), _lib.ptr(n), N,
                                   _lib.ptr(xo), _lib.ptr(ao), _lib.ptr(ho), None))
    torch.cuda.synchronize()
    return xo.cpu().numpy(), ao.cpu().numpy(), ho.cpu().numpy()


def _both(H, fn, kind, x0, c, thr, n, biases=(0, 4, -4)):
    xh, ah, hh = _run(fn, x0, c, thr, n)              # host: chain == naive loop (asserted inside)
    for b in biases:
        xd, ad, hd = _device(kind, b, x0, c, thr, n)
        bad = np.flatnonzero((xd.view(np.uint64) != xh.view(np.uint64)) | (ad != ah) | (hd != hh))
        assert bad.size == 0, (kind, b, bad[:3], np.asarray(x0)[bad[:3]], np.asarray(c)[bad[:3]],
                               xd[bad[:3]], xh[bad[:3]], ad[bad[:3]], ah[bad[:3]])


def test_simulator_shapes_on_device(H):
    rng = np.random.default_rng(11)
    N = 200_000
    c = rng.uniform(0.05, 12.0, N).astype(np.float32).astype(np.float64) * 0.01
    x0 = np.where(rng.random(N) < 0.4, 0.0, rng.uniform(0, 20, N))
    thr = rng.choice([0.3, 0.75, 1.2, 1.85, 2.85, 4.3], N) * rng.choice([1.0, 2.0, 4.0], N)
    n = rng.integers(1, 3000, N)
    _both(H, H.fuzz_ge, GE, x0, c, thr, n)
    xb = np.where(rng.random(N) < 0.5, rng.integers(1, 7, N) * 4.0, rng.uniform(0.001, 30, N))
    sd = rng.choice([0.01, 0.0125, 0.005, 0.02], N)
    _both(H, H.fuzz_le, LE, xb, -sd, np.zeros(N), rng.integers(1, 4000, N))
    t2 = rng.choice([20.0, 3.0, 5.0, 9.0], N)
    _both(H, H.fuzz_lt, LT, t2 + rng.uniform(0, 5, N), -sd, t2, rng.integers(1, 4000, N))


def test_ties_thresholds_and_wild_ranges_on_device(H):
    rng = np.random.default_rng(12)
    N = 100_000
    e = rng.integers(-8, 6, N)
    base = np.ldexp(1.0, e)
    kmant = rng.integers(0, 1 << 20, N)
    x0 = base * (1.0 + kmant * 2.0 ** -52)
    u = base * 2.0 ** -52
    c = rng.integers(1, 1 << 44, N).astype(np.float64) * u + u / 2       # ties in binade e
    n = rng.integers(1, 500, N)
    thr = x0 + c * rng.integers(1, 600, N)
    for t in (thr, np.nextafter(thr, np.inf), np.nextafter(thr, -np.inf)):
        _both(H, H.fuzz_ge, GE, x0, c, t, n)
    x1 = base * (2.0 - kmant * 2.0 ** -52)
    thr2 = np.maximum(x1 - c * rng.integers(1, 600, N), 0.0)
    _both(H, H.fuzz_le, LE, x1, -c, thr2, n)
    _both(H, H.fuzz_lt, LT, x1, -c, np.nextafter(thr2, np.inf), n)
    xw = np.ldexp(rng.uniform(1, 2, N), rng.integers(-40, 40, N)) * (rng.random(N) > 0.1)
    cw = np.ldexp(rng.uniform(1, 2, N), rng.integers(-45, 30, N))
    nw = rng.integers(1, 2000, N)
    _both(H, H.fuzz_ge, GE, xw, cw, xw + cw * rng.uniform(0, 3000, N), nw)
    _both(H, H.fuzz_le, LE, xw, -cw, np.maximum(xw - cw * rng.uniform(0, 3000, N), -1.0), nw)


def test_very_long_jumps_on_device(H):
    """Runs of up to 2^27 equal additions inside one binade: the regime in which a reciprocal
    estimate could be off by more than one.  A segment is capped at 2^20 additions (kJumpCap), below
    which the estimate is off by less than half a step; the biased runs show the repair working."""
    x0 = np.array([1.0, 1.0, 1.5, 1024.0, 1.0, 3.0])
    c = np.array([2.0 ** -30, 2.0 ** -29 + 2.0 ** -52, 3 * 2.0 ** -31, 2.0 ** -17, 2.0 ** -28, 2.0 ** -27])
    n = np.array([1 << 27, (1 << 27) + 12345, 1 << 26, 1 << 27, 99_999_999, 1 << 27], np.int32)
    thr = np.array([10.0, 1.2, 1.9, 1e9, 1.3, 3.9])
    _both(H, H.fuzz_ge, GE, x0, c, thr, n)
    x1 = np.array([2.0 - 2.0 ** -52, 1.75, 2047.0, 3.999])
    c1 = np.array([2.0 ** -30, 2.0 ** -29, 2.0 ** -18, 2.0 ** -27])
    n1 = np.array([1 << 27, 1 << 26, 1 << 27, 1 << 27], np.int32)
    _both(H, H.fuzz_le, LE, x1, -c1, np.array([0.0, 1.1, 1030.0, 2.5]), n1)
    _both(H, H.fuzz_lt, LT, x1, -c1, np.array([1.5, 1.25, 1024.0, 2.0]), n1)


def test_drain_cascade_on_device():
    """The per-binade cascade of `buffer_level -= speed * dt` (round 6: csrc/abr_exact_jump.h: drain_cascade, what the
    one-thread-per-lane and two-wave kernels drain with at one play speed) on the DEVICE against the plain loop on the host:
    ticks, ran-dry answer and the float64 value, for the simulator's own subtrahend and others (exact binades, a tie binade
    inside the range), starts on binade boundaries, on the stage thresholds, within ulps of k * sd, and -- mixed into the
    same waves -- values above the cascade, which send their wave through the general chain."""
    from abrsimulator_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(21)
    n = 60_000
    for sd, max_level in ((0.01, 24.0), (1.25 * 0.01, 24.0), (0.75 * 0.01, 9.0), (0.01, 1000.0), (0.0078125, 24.0),
                          (0.01171875, 30.0), (0.5 * 0.01, 1.0e5)):
        top = 2.0 ** np.floor(np.log2(max_level) + 1)
        b_rand = rng.uniform(0.0, 1.0, n) ** 2 * min(max_level * 1.3, top * 1.2)       # some above the cascade
        k = rng.integers(1, int(min(max_level, 60.0) / sd), n)
        adv = k * sd
        for _ in range(3):
            adv = np.where(rng.random(n) < 0.5, np.nextafter(adv, np.where(rng.random(n) < 0.5, np.inf, -np.inf)), adv)
        e = rng.integers(-8, int(np.log2(top)), n)
        base = np.ldexp(1.0, e)
        u = base * 2.0 ** -52
        S = (base + sd) - base
        T = np.where(S >= sd, S, S + u)
        edge = np.where(rng.random(n) < 0.5, base, base + T) + rng.integers(-2, 3, n) * u
        edge = np.where(rng.random(n) < 0.3, edge + rng.integers(0, 50, n) * S, edge)
        sim = np.zeros(n)
        for _ in range(4):
            sim = sim + 4.0
            t_ = rng.integers(0, 300, n)
            for j in range(300):
                sim = np.where(j < t_, sim - sd, sim)
        B0 = np.concatenate([b_rand, adv, edge, sim])
        M = np.concatenate([rng.integers(0, 3000, n), k + rng.integers(-2, 3, n), rng.integers(0, 5000, n), rng.integers(0, 2500, n)])
        keep = B0 > 0.0
        B0 = np.ascontiguousarray(B0[keep]); M = np.ascontiguousarray(np.maximum(M[keep], 0).astype(np.int32))
        # the plain loop on the host, vectorised: one subtraction per round for the cases still running
        xh = B0.copy(); ah = np.zeros(len(B0), np.int64)
        run = (ah < M) & (xh > 0.0)
        while run.any():
            xh = np.where(run, xh - sd, xh); ah += run
            run = (ah < M) & (xh > 0.0)
        zh = (ah > 0) & (xh <= 0.0)
        x0 = torch.from_numpy(B0).cuda(); nn = torch.from_numpy(M).cuda()
        xo = torch.empty_like(x0); ao = torch.empty_like(nn); ho = torch.empty(len(B0), dtype=torch.uint8, device="cuda")
        stages = C.c_int32(0)
        _lib.check(lib.abr_debug_drain(sd, max_level, _lib.ptr(x0), _lib.ptr(nn), len(B0), _lib.ptr(xo), _lib.ptr(ao),
                                       _lib.ptr(ho), C.byref(stages), None))
        torch.cuda.synchronize()
        xd, ad, hd = xo.cpu().numpy(), ao.cpu().numpy(), ho.cpu().numpy()
        bad = np.flatnonzero((xd.view(np.uint64) != xh.view(np.uint64)) | (ad != ah) | (hd != zh))
        assert bad.size == 0, (sd, max_level, bad[:3], B0[bad[:3]], M[bad[:3]], xd[bad[:3]], xh[bad[:3]], ad[bad[:3]], ah[bad[:3]])
        assert stages.value >= 3 and zh.sum() > 10_000
    one = C.c_void_p(256)
    assert lib.abr_debug_drain(0.01, 1.0e12, one, one, 4, one, one, one, None, None) == -4      # ABR_E_UNSUPPORTED: no cascade
    assert lib.abr_debug_drain(0.01, 24.0, None, one, 4, one, one, one, None, None) == -1


def test_rejects_bad_arguments():
    from abrsimulator_amd import _lib
    lib = _lib.lib()
    one = C.c_void_p(256)
    assert lib.abr_debug_chain(3, 0, one, one, one, one, 4, one, one, one, None) == -1
    assert lib.abr_debug_chain(0, 1, one, one, one, one, 4, one, one, one, None) == -1
    assert lib.abr_debug_chain(0, 0, None, one, one, one, 4, one, one, one, None) == -1
