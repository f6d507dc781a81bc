"""Per-role accounting of the asynchronous env pipeline (diagnostic build libabr_hip_astats.so):
cycles each role spends working / polling, trips, passes, lanes per trip.
  ABR_HIP_LIB=libabr_hip_astats.so python tools/gpu_async_stats.py [lanes] [fuse]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import abrsimulator_amd as A  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
F = int(sys.argv[2]) if len(sys.argv) > 2 else 48
traces = B.synth_traces(False)
tid, off = B.lane_assignment(0, N, traces)
env = A.BatchedABREnv(A.MPD(B.V, B.L, B.MAX_BUFFER, B.START_UP, A.Chunk(B.LADDER)), A.QOEMetric(*B.WEIGHTS),
                      A.NetworkInfo(B.INTERVAL, traces), N, auto_reset=True, impl="async")
env.reset(torch.from_numpy(tid), torch.from_numpy(off))
for _ in range(3):
    env.step_random(F, 1, want_actions=False)
torch.cuda.synchronize()
rd = env.lib.abr_debug_async_stats
rd.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 48)()
rd(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
env.step_random(F, 1, want_actions=False)
e1.record()
torch.cuda.synchronize()
rd(buf, 1)
waves = (N + 63) // 64
steps = F
print(f"lanes {N} fuse {F}: launch {e0.elapsed_time(e1) * 1e3:.1f} us")
d = [buf[i] / waves for i in range(12)]
print(f"D per wave: loop-head {d[0]:.0f}  trips {d[1]:.0f}  passes {d[2]:.0f}  idle {d[3]:.0f} cycles; "
      f"trips {d[4]:.1f} ({d[4] / steps:.2f}/step, {d[1] / max(d[4], 1):.0f} cyc each, {d[5] / max(d[4], 1):.1f} lanes running)  "
      f"passes {d[6]:.1f} ({d[6] / steps:.2f}/step, {d[2] / max(d[6], 1):.0f} cyc each, {d[7] / max(d[6], 1):.1f} lanes parked)  "
      f"idle polls {d[8]:.1f}")
q = [buf[12 + i] / waves for i in range(12)]
print(f"P per wave: poll {q[0]:.0f}  idle {q[1]:.0f}  work {q[2]:.0f}  ring2-wait {q[3]:.0f} cycles; "
      f"iterations {q[4]:.1f} ({q[4] / steps:.2f}/step, {q[2] / max(q[4], 1):.0f} cyc each, {q[5] / max(q[4], 1):.1f} lanes)  idle polls {q[8]:.1f}")
r = [buf[24 + i] / waves for i in range(12)]
print(f"S per wave: poll {r[0]:.0f}  idle {r[1]:.0f}  work {r[2]:.0f}  actions {r[6]:.0f} cycles; "
      f"iterations {r[4]:.1f} ({r[4] / steps:.2f}/step, {r[2] / max(r[4], 1):.0f} cyc each, {r[5] / max(r[4], 1):.1f} lanes)  idle polls {r[8]:.1f}")
