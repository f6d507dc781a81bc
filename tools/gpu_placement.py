"""Where the waves of the role-split kernel's workgroups run (diagnostic build libabr_hip_stamps.so): for every CU the SIMD of
each workgroup's download / player / service wave and the workgroup's lifetime; which SIMDs hold two waves of the same role.
  ABR_HIP_LIB=libabr_hip_stamps.so python tools/gpu_placement.py [lanes] [impl] [launches]"""
import ctypes as C
import os
import sys
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import abrsimulator_amd as A  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
IMPL = sys.argv[2] if len(sys.argv) > 2 else "split3"
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
traces = B.synth_traces(False)
tid, off = B.lane_assignment(0, N, traces)
env = A.BatchedABREnv(A.MPD(B.V, B.L, B.MAX_BUFFER, B.START_UP, A.Chunk(B.LADDER)), A.QOEMetric(*B.WEIGHTS),
                      A.NetworkInfo(B.INTERVAL, traces), N, auto_reset=True, impl=IMPL)
env.reset(torch.from_numpy(tid), torch.from_numpy(off))
nw = min(N // 64, 4096)
env.lib.abr_debug_read_wg_times.argtypes = [C.c_void_p, C.c_int]
prev = None
for rep in range(REPS):
    env.step_random(48, 1, want_actions=False)
    torch.cuda.synchronize()
    wt = (C.c_ulonglong * (nw * 10))()
    env.lib.abr_debug_read_wg_times(wt, nw)
    w = np.array(wt, dtype=np.uint64).reshape(nw, 10)
    life = (w[:, 3] - w[:, 0]).astype(np.float64)
    hw = w[:, 4:7].astype(np.int64)                       # xcc << 16 | HW_ID, per role
    cu = ((hw[:, 0] >> 16) << 16) | (hw[:, 0] & 0xff00)   # xcc, se, sh, cu
    simd = (hw >> 4) & 3                                   # [wg, role]
    slot = hw & 0xf                                        # wave slot inside the SIMD
    key = np.stack([cu, simd[:, 0], simd[:, 1], simd[:, 2]], 1)
    if prev is not None:
        print(f"launch {rep}: placement identical to the previous launch for {(key == prev).all(1).mean() * 100:.1f} % of the workgroups")
    prev = key
    by_cu = defaultdict(list)
    for g in range(nw):
        by_cu[int(cu[g])].append(g)
    # per CU: the number of D / P / S waves on each SIMD
    bad = []
    for c, gs in by_cu.items():
        cnt = np.zeros((3, 4), int)
        for g in gs:
            for r in range(3):
                cnt[r, simd[g, r]] += 1
        if cnt.max() > 1 or len(gs) != 4:
            bad.append((c, gs, cnt))
    same_simd = sum(1 for g in range(nw) if len(set(simd[g].tolist())) < 3)
    print(f"launch {rep}: {len(by_cu)} CUs, {len(bad)} with a SIMD that holds two waves of one role (or not 4 workgroups); "
          f"workgroups with two of their own waves on one SIMD: {same_simd}; lifetime mean {life.mean():.0f} p99 {np.percentile(life, 99):.0f} max {life.max():.0f}")
    for c, gs, cnt in bad[:12]:
        print(f"  xcc {c >> 16} se {(c >> 13) & 7} sh {(c >> 12) & 1} cu {(c >> 8) & 15}: D per SIMD {cnt[0].tolist()} P {cnt[1].tolist()} S {cnt[2].tolist()}")
        for g in gs:
            print(f"      workgroup {g:5d}: D on SIMD {simd[g, 0]} slot {slot[g, 0]}, P on {simd[g, 1]} slot {slot[g, 1]}, S on {simd[g, 2]} slot {slot[g, 2]}   lifetime {life[g]:9.0f}")
    good = [gs for c, gs in by_cu.items() if c not in {b[0] for b in bad}]
    if good:
        gl = np.array([life[g] for gs in good for g in gs])
        print(f"  workgroups on balanced CUs: lifetime mean {gl.mean():.0f} max {gl.max():.0f};  on the others: "
              f"mean {np.mean([life[g] for _, gs, _ in bad for g in gs]) if bad else 0:.0f} max {max([life[g] for _, gs, _ in bad for g in gs]) if bad else 0:.0f}")
    if rep == REPS - 1 and os.environ.get("ABR_PLACEMENT_DUMP"):
        import json
        rt = (w[:, 9].astype(np.float64) - w[:, 8].astype(np.float64)) * 0.01      # microseconds (s_memrealtime, 100 MHz)
        rt0 = (w[:, 8].astype(np.float64) - w[:, 8].astype(np.float64).min()) * 0.01
        json.dump(dict(lanes=N, impl=IMPL, decisions=48, xcc=(hw[:, 0] >> 16).tolist(), cu=cu.tolist(), lifetime_cycles=life.tolist(),
                       wall_us=rt.tolist(), begin_us=rt0.tolist()), open(os.environ["ABR_PLACEMENT_DUMP"], "w"))
    c0 = sorted(by_cu)[0]
    print("  a balanced CU for comparison:", [(g, simd[g].tolist(), slot[g].tolist(), int(life[g])) for g in by_cu[sorted(set(by_cu) - {b[0] for b in bad})[0]]])
