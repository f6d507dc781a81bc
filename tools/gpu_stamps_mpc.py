"""Cycles per phase of the K3 search kernel (diagnostic build libabr_hip_stamps.so):
  ABR_HIP_LIB=libabr_hip_stamps.so python tools/gpu_stamps_mpc.py [lanes]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import abrsimulator_amd as A  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
traces = B.synth_traces(False)
tid, off = B.lane_assignment(0, N, traces)
env = A.BatchedABREnv(A.MPD(B.V, B.L, B.MAX_BUFFER, B.START_UP, A.Chunk(B.LADDER)), A.QOEMetric(*B.WEIGHTS),
                      A.NetworkInfo(B.INTERVAL, traces), N, auto_reset=True)
env.reset(torch.from_numpy(tid), torch.from_numpy(off))
env.step_random(3, 1)
player = A.EnvPlayer(env, mpd=A.MPD(B.V, B.L, B.MAX_BUFFER, B.START_UP, [A.Chunk(B.LADDER, [b * B.L for b in B.LADDER])] * B.V),
                     qoe=A.QOEMetric(4.3, 1.0, 0.0))
ctl = A.BatchedMPCController(player, horizon=5, clip_horizon=True)
for _ in range(5):
    ctl.next_bitrate()
torch.cuda.synchronize()
rd = env.lib.abr_debug_read_stamps
rd.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 32)()
rd(buf, 1)
K = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K):
    ctl.next_bitrate()
e1.record()
torch.cuda.synchronize()
rd(buf, 1)
waves = (((N + 6) // 7 + 63) // 64) * 4          # the stamps build samples one workgroup in 64
names = {26: "phase 1 (predictions -> LDS) + barrier", 27: "phase 2 (tables: 60 divisions per lane) + barrier",
         28: "phase 3 (prefix + enumeration)", 29: "phase 4 (arg-max: 2 LDS atomics, 2 barriers)",
         30: "phase 5 (first leaf of the winning node) + outputs"}
tot = sum(buf[k] for k in names)
print(f"{N} lanes, {K} selects, {e0.elapsed_time(e1) / K * 1e3:.1f} us per select between events; cycles per wave per launch:")
for k in sorted(names):
    print(f"  [{k}] {names[k]:52s} {buf[k] / waves / K:9.0f}  {100.0 * buf[k] / tot:5.1f} %")
print(f"  total {tot / waves / K:.0f} cycles per wave; {waves} waves per launch")
