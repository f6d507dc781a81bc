"""Which allocation decides how fast an XCD runs the three-wave kernel?  (Round 6: at identical work the XCDs differ by up to 22 % in
wall time, and WHICH ones are slow changes from process to process on the same GPU.)  One process, the stamps build: the per-XCD
mean workgroup lifetime of a 48-decision launch for (a) the same environment launched again, (b) new output buffers,
(c) a new environment (new workspace, new trace copy) after a dummy allocation that shifts the allocator, (d) the same again.
  ABR_HIP_LIB=libabr_hip_stamps.so python tools/gpu_xcd_speed.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import abrsimulator_amd as A  # noqa: E402

N = 65536
traces = B.synth_traces(False)
tid, off = B.lane_assignment(0, N, traces)


def make():
    env = A.BatchedABREnv(A.MPD(B.V, B.L, B.MAX_BUFFER, B.START_UP, A.Chunk(B.LADDER)), A.QOEMetric(*B.WEIGHTS),
                          A.NetworkInfo(B.INTERVAL, traces), N, auto_reset=True, impl="split3")
    env.reset(torch.from_numpy(tid), torch.from_numpy(off))
    return env


def outs():
    return dict(obs=torch.empty(48, 8, N, dtype=torch.float32, device="cuda"), reward=torch.empty(48, N, dtype=torch.float32, device="cuda"),
                done=torch.empty(48, N, dtype=torch.uint8, device="cuda"), actions=None)


def measure(env, out, label):
    env.lib.abr_debug_read_wg_times.argtypes = [C.c_void_p, C.c_int]
    res = []
    for rep in range(3):
        env.step_random(48, 1, out=out)
        torch.cuda.synchronize()
        wt = (C.c_ulonglong * (1024 * 10))()
        env.lib.abr_debug_read_wg_times(wt, 1024)
        w = np.array(wt, dtype=np.uint64).reshape(1024, 10)
        rt = (w[:, 9].astype(np.float64) - w[:, 8].astype(np.float64)) * 0.01
        x = (w[:, 4] >> np.uint64(16)).astype(np.int64)
        res.append([rt[x == q].mean() for q in range(8)] + [rt.max()])
    r = np.array(res)[1:].mean(0)
    print(f"{label:58s}" + " ".join(f"{v:6.1f}" for v in r[:8]) + f"   max {r[8]:6.1f}   ws {env.workspace.data_ptr():#x} obs {out['obs'].data_ptr() if out.get('obs') is not None else 0:#x}")


print(" " * 58 + " ".join(f"  xcd{q}" for q in range(8)))
e1 = make(); o1 = outs()
measure(e1, o1, "env 1, outputs 1")
measure(e1, o1, "env 1, outputs 1 again")
o2 = outs()
measure(e1, o2, "env 1, NEW outputs")
dummy = torch.empty(int(os.environ.get("ABR_DUMMY_MB", "37")) * 1024 * 1024 + 4096 * 3, dtype=torch.uint8, device="cuda")
e2 = make(); o3 = outs()
measure(e2, o3, "env 2 (new workspace + traces + outputs after a dummy)")
measure(e2, o1, "env 2, outputs 1")
measure(e1, o3, "env 1, outputs 3")
dummy2 = torch.empty(123 * 1024 * 1024 + 8192, dtype=torch.uint8, device="cuda")
e3 = make(); o4 = outs()
measure(e3, o4, "env 3 (after another dummy)")
measure(e1, o1, "env 1, outputs 1 once more")
measure(e1, dict(obs=None, reward=None, done=None, actions=None), "env 1, NO outputs at all (nothing but the lane state is written)")
measure(e1, dict(obs=None, reward=o1["reward"], done=o1["done"], actions=None), "env 1, reward + done only (no observations)")
measure(e1, o1, "env 1, outputs 1 at the end")
