"""What ANY regrouping of downloads inside a wave could buy at most (the measured side of
tools/replay_model.py's pricing).

A wave pays its slowest lane: 35.7 of 64 lanes are active per vector instruction of the role-split kernel.
Dealing downloads to waves by cost class would raise that figure at the price of a sort and of lane state that
travels through LDS.  Its CEILING needs no new kernel: give the 64 lanes of a wave identical work (same trace,
same start offset, same scripted actions) and the shipped kernel runs them without any divergence -- the wave's
slowest lane is every lane.  Four scripted 48-decision rollouts at 65 536 lanes, same traces and ladder as bench.py:

  (a) lanes as bench.py assigns them, actions independent per lane          (the baseline of this script)
  (b) the same, actions shared by the 64 lanes of a wave                    (sizes agree, bandwidth does not)
  (c) trace / offset shared by the 64 lanes of a wave, actions independent  (bandwidth agrees, sizes do not)
  (d) both shared: 1 024 distinct waves                                     (the ceiling)

usage (GPU box): python tools/gpu_dealing_bound.py [impl]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import abrsimulator_amd as A  # noqa: E402
import bench  # noqa: E402

N, F, LAUNCHES = 65536, 48, 20
impl = sys.argv[1] if len(sys.argv) > 1 else "auto"
traces = bench.synth_traces(False)
mpd = A.MPD(bench.V, bench.L, bench.MAX_BUFFER, bench.START_UP, A.Chunk(bench.LADDER))
tid0, off0 = bench.lane_assignment(0, N, traces)
rng = np.random.default_rng(7)
act_lane = rng.integers(0, len(bench.LADDER), (F, N)).astype(np.int32)
act_wave = np.repeat(act_lane[:, ::64], 64, axis=1)
tid_wave, off_wave = np.repeat(tid0[::64], 64), np.repeat(off0[::64], 64)


def run(name, tid, off, act):
    env = A.BatchedABREnv(mpd, A.QOEMetric(*bench.WEIGHTS), A.NetworkInfo(bench.INTERVAL, traces), N,
                          auto_reset=True, impl=impl)
    env.reset(torch.from_numpy(np.ascontiguousarray(tid)), torch.from_numpy(np.ascontiguousarray(off)))
    a = torch.from_numpy(np.ascontiguousarray(act)).cuda()
    out = env.step_script(a)
    for _ in range(3):
        env.step_script(a, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(LAUNCHES):
        env.step_script(a, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / LAUNCHES
    print("%-58s %8.1f us per launch   %.3e env-steps/s   (%s)" % (name, dt * 1e6, N * F / dt, env.effective_impl(True)),
          flush=True)
    return dt


for rnd in range(3):
    a_ = run("(a) bench lanes, actions per lane", tid0, off0, act_lane)
    b_ = run("(b) bench lanes, actions per wave", tid0, off0, act_wave)
    c_ = run("(c) trace/offset per wave, actions per lane", tid_wave, off_wave, act_lane)
    d_ = run("(d) trace/offset and actions per wave: no divergence", tid_wave, off_wave, act_wave)
    print("    ceiling of regrouping: %.2fx the baseline; sizes alone %.2fx, bandwidth alone %.2fx" % (a_ / d_, a_ / b_, a_ / c_))
