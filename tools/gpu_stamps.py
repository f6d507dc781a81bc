"""Per-role cycle stamps of the role-split env kernel (diagnostic build libabr_hip_stamps.so):
how long each wave works per iteration and how long it waits at the workgroup barrier.
  ABR_HIP_LIB=libabr_hip_stamps.so python tools/gpu_stamps.py [lanes]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import abrsimulator_amd as A  # noqa: E402
from abrsimulator_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
traces = B.synth_traces(False)
tid, off = B.lane_assignment(0, N, traces)
IMPL = sys.argv[2] if len(sys.argv) > 2 else "split"
env = A.BatchedABREnv(A.MPD(B.V, B.L, B.MAX_BUFFER, B.START_UP, A.Chunk(B.LADDER)), A.QOEMetric(*B.WEIGHTS),
                      A.NetworkInfo(B.INTERVAL, traces), N, auto_reset=True, impl=IMPL)
env.reset(torch.from_numpy(tid), torch.from_numpy(off))
for _ in range(3):
    env.step_random(48, 1, want_actions=False)
torch.cuda.synchronize()
print("impl", IMPL)
rd = env.lib.abr_debug_read_stamps
rd.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 32)()
rd(buf, 1)
env.step_random(48, 1, want_actions=False)
rd(buf, 1)
waves = N // 64
regions = {1: "D begin_step loads", 2: "D philox", 3: "D download loop", 4: "D publish", 5: "D barrier wait",
           0: "D validate+loop", 9: "P record read", 10: "P drain (download ticks)", 11: "P completing tick",
           12: "P phase B wait_call", 13: "P return", 14: "P div+hist+reward", 15: "P episode end", 16: "P obs out",
           17: "P feedback", 18: "P barrier wait", 8: "P loop",
           20: "S loop", 21: "S service (split3)", 22: "S barrier wait",
           23: "P idle: up to the drain", 24: "P drain: segments", 25: "P drain: plain tail loop"}
for k in sorted(regions):
    print(f"  [{k:2d}] {regions[k]:28s} {buf[k] / waves / 49:9.0f} cycles / wave / iteration")
