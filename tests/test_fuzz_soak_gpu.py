"""What used to run only when the builder ran it, pulled into what the driver runs: a slice of the
configuration-space fuzz (tools/gpu_fuzz.py) over the three product kernels, one soak at a quarter of a million lanes against the
oracle on the host cores (tools/soak_parity.py), and a test aimed at the one place where two waves exchange data WITHOUT a
barrier in between -- the action ring of the three-wave kernel -- in the shape in which round 5 found its publish race."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_configuration_fuzz_slice_on_the_product_kernels():
    """200 random configurations (chunk length, trace interval, ladder, buffer limits, start-up length, ragged traces with
    wrap-around) x {one speed, per-lane speeds, speed schedule, per-chunk ladders, both} x {three-wave, two-wave, one thread
    per lane} x {V single steps, one fused scripted call}: every lane's previous_bandwidths, every reward, final clocks, buffer,
    play_id `==` the C oracle; QoE 1e-10, average latency 1e-9.  (tools/gpu_fuzz.py runs thousands, incl. the diagnostic
    pipelines; profiles/r0*_gpu_fuzz*.json.)"""
    import gpu_fuzz
    bad, lane_steps, cases = 0, 0, {}
    for seed in range(200):
        b, ls, key, what = gpu_fuzz.run_seed(seed, 128, impls=["split3", "split", "jump"])
        assert b == 0, ("mismatches", b, what)
        lane_steps += ls
        cases[key] = cases.get(key, 0) + 1
    assert lane_steps > 100_000 and len(cases) >= 20 and bad == 0


def test_soak_262144_lanes_mixed_traces_against_the_oracle():
    """BASELINE.json configs[4]'s trace shape (lengths 300-3 000) at 262 144 lanes x 48 decisions on `auto` (one thread per
    lane at this size): every lane, every decision against the oracle on the host cores."""
    import soak_parity
    res = soak_parity.soak(262144, mixed=True, impl="auto")
    assert res["mismatches"] == 0 and res["decisions"] == 262144 * 48 and res["kernel"] == "jump", res


@pytest.mark.parametrize("fuse", [64, 130])
def test_action_ring_wraps_while_the_download_wave_idles(fuse):
    """The three-wave kernel's service wave draws the policy's actions up to 60 steps ahead into a 64-entry LDS ring that the
    download wave reads in the SAME iteration, no barrier in between: bytes first, `lds_writes_done`, then the counter that
    vouches for them (csrc/abr_env_roles.h).  Round 5 found that a later single-lane LDS write can become visible before an
    earlier full-wave one, through a reader that polls tightly -- so this is the shape: a download of one or two ticks (tiny
    bitrates on a fat network), i.e. a download wave that finishes at once and is back at the ring while the service wave is
    still writing, a launch long enough for the ring to wrap (64 / 130 decisions), lane groups of 64.  Everything the launch
    writes must equal the one-thread-per-lane kernel's, which has no ring; the actions must be the philox twin's."""
    import abrsimulator_amd as A
    from helpers import philox_action
    V, N, SEED = 130, 4096, 77
    rng = np.random.default_rng(9)
    traces = [rng.uniform(40.0, 80.0, 400).astype(np.float32).astype(np.float64) for _ in range(16)]
    ladder = [0.004, 0.006, 0.008, 0.012, 0.016, 0.02]          # x chunk_length 2 s: 1-3 ticks of download at 0.4-0.8 per tick
    tid = (np.arange(N) % 16).astype(np.int32)
    off = rng.integers(0, 400, N).astype(np.int32)

    def run(impl):
        env = A.BatchedABREnv(A.MPD(V, 2.0, 6.0, 2.0, A.Chunk(ladder)), A.QOEMetric(4.3, 1, 1, 0.1),
                              A.NetworkInfo(1.0, traces), N, auto_reset=True, impl=impl)
        env.reset(torch.from_numpy(tid), torch.from_numpy(off))
        outs = []
        for _ in range(3):                                       # three launches: the ring starts afresh in each
            o = env.step_random(fuse, SEED)
            outs.append({k: v.clone() for k, v in o.items()})
        return outs, env.observe_f64()

    a, fa = run("split3")
    b, fb = run("jump")
    for x, y in zip(a, b):
        for k in ("obs", "reward", "done", "actions"):
            assert torch.equal(x[k], y[k]), (fuse, k)
    for k in fa:
        assert torch.equal(fa[k], fb[k]), k
    acts = torch.cat([x["actions"] for x in a]).cpu().numpy()   # [3 * fuse, N]; decision n is chunk n % V of episode n // V
    for n in range(acts.shape[0]):
        assert np.array_equal(acts[n], philox_action(SEED, np.arange(N), n % V, n // V, len(ladder))), n
