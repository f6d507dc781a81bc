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
bx = (C.c_ulonglong * 256)()
if hasattr(env.lib, "abr_debug_read_stamps_xcd"):
    env.lib.abr_debug_read_stamps_xcd.argtypes = [C.c_void_p, C.c_int]
    env.lib.abr_debug_read_stamps_xcd(bx, 1)
env.step_random(48, 1, want_actions=False)
rd(buf, 1)
if hasattr(env.lib, "abr_debug_read_stamps_xcd"):
    env.lib.abr_debug_read_stamps_xcd(bx, 1)
waves = N // 64
regions = {1: "D begin_step loads", 2: "D philox", 3: "D download loop", 4: "D publish", 5: "D barrier wait",
           0: "D validate+loop", 9: "P record read", 10: "P drain (download ticks)", 11: "P completing tick",
           12: "P phase B wait_call", 13: "P return", 14: "P div+hist+reward", 15: "P episode end", 16: "P obs out",
           17: "P feedback", 18: "P barrier wait", 8: "P loop",
           20: "S loop", 21: "S service (split3)", 22: "S barrier wait",
           23: "P idle: up to the drain", 24: "P drain: segments", 25: "P drain: plain tail loop"}
if IMPL == "pair3":     # download and player wave in lock-step, the service wave behind a ring (abr_env_pair.h)
    regions.update({5: "D rendezvous wait (P behind)", 18: "P rendezvous wait (D behind)", 19: "P wait: ring full (S behind)",
                    20: "S draws ahead", 22: "S wait: input (P behind)", 21: "S service"})
if IMPL == "ring3":     # the ring-coupled kernel reuses the slots (abr_env_ring.h)
    regions.update({0: "D corrections + loop", 5: "D wait: ring full (P behind)", 18: "P wait: input (D behind)",
                    19: "P wait: ring full (S behind)", 13: "P return + hand-off to S", 17: "P position + publish",
                    20: "S draws ahead", 22: "S wait: input (P behind)", 21: "S service"})
for k in sorted(regions):
    print(f"  [{k:2d}] {regions[k]:28s} {buf[k] / waves / 49:9.0f} cycles / wave / iteration")

# the same per XCD: cycles per wave per iteration of every region, one column per XCD (128 workgroups each)
if hasattr(env.lib, "abr_debug_read_stamps_xcd") and sum(bx) > 0:
    x = np.array(bx, dtype=np.float64).reshape(8, 32) / (waves / 8.0) / 49.0
    print("  per XCD (cycles / wave / iteration)      " + "".join(f"  xcd{q}  " for q in range(8)))
    for k in sorted(regions):
        if x[:, k].sum() > 0:
            print(f"  [{k:2d}] {regions[k]:34s}" + "".join(f"{v:8.0f}" for v in x[:, k]))
    for name, ks in (("D total", (0, 1, 2, 3, 4, 5, 6)), ("P total", (8, 9, 10, 11, 12, 13, 17, 18, 19)), ("S total", (20, 21, 22))):
        print(f"       {name:34s}" + "".join(f"{x[q, list(ks)].sum():8.0f}" for q in range(8)))

# how long each workgroup lived, and how that depends on what else ran on its waves' SIMDs (diagnostic build).  s_memtime has a
# different base per XCD, so only differences inside a workgroup mean anything.
if hasattr(env.lib, "abr_debug_read_wg_times"):
    nw = min(waves, 4096)
    wt = (C.c_ulonglong * (nw * 10))()
    env.lib.abr_debug_read_wg_times.argtypes = [C.c_void_p, C.c_int]
    env.lib.abr_debug_read_wg_times(wt, nw)
    w = np.array(wt, dtype=np.uint64).reshape(nw, 10)
    t = w[:, :4].astype(np.float64)
    life = t[:, 3] - t[:, 0]
    print(f"  workgroup lifetime (D begin -> S end): mean {life.mean():.0f}  p05 {np.percentile(life, 5):.0f}  p50 {np.percentile(life, 50):.0f}  "
          f"p95 {np.percentile(life, 95):.0f}  p99 {np.percentile(life, 99):.0f}  max {life.max():.0f};  D end {np.mean(t[:, 1] - t[:, 0]):.0f}  P end {np.mean(t[:, 2] - t[:, 0]):.0f}")
    simd = w[:, 4:7] & np.uint64(0xffff3f30 | 0xf000)          # xcc, se, sh, cu, simd (wave slot and pipe masked out)
    simd = (w[:, 4:7] >> np.uint64(4)) & np.uint64(0xffff) | ((w[:, 4:7] >> np.uint64(16)) << np.uint64(16))
    simd = ((w[:, 4:7] >> np.uint64(16)) << np.uint64(16)) | (w[:, 4:7] & np.uint64(0xff30))
    from collections import Counter
    per = [Counter(simd[:, r].tolist()) for r in range(3)]
    tot = Counter(simd.ravel().tolist())
    for r, name in enumerate("DPS"):
        same = np.array([per[r][x] for x in simd[:, r].tolist()])        # waves of the same role on this role's SIMD
        allw = np.array([tot[x] for x in simd[:, r].tolist()])           # waves of any role there
        for k in sorted(set(same.tolist())):
            m = same == k
            print(f"  workgroups whose {name} wave shares its SIMD with {k - 1} other {name} waves: {m.sum():5d}  lifetime mean {life[m].mean():9.0f}  max {life[m].max():9.0f}")
        for k in sorted(set(allw.tolist())):
            m = allw == k
            print(f"  workgroups whose {name} wave sits on a SIMD with {k} waves in all:        {m.sum():5d}  lifetime mean {life[m].mean():9.0f}  max {life[m].max():9.0f}")
    worst = np.argsort(-life)[:8]
    for g in worst:
        print(f"  slowest: workgroup {g:5d} lifetime {life[g]:9.0f}  D/P/S waves alone-of-their-role on their SIMD: "
              + " ".join(f"{name}:{per[r][int(simd[g, r])]}/{tot[int(simd[g, r])]}" for r, name in enumerate("DPS")))
    # the same on the constant 100 MHz clock (s_memrealtime): wall time, comparable across XCDs
    rt = (w[:, 9].astype(np.float64) - w[:, 8].astype(np.float64)) * 0.01      # microseconds
    rt0 = (w[:, 8].astype(np.float64) - w[:, 8].astype(np.float64).min()) * 0.01
    end = rt0 + rt
    print(f"  wall time (s_memrealtime): workgroup lifetime mean {rt.mean():.1f} us  p05 {np.percentile(rt, 5):.1f}  p95 {np.percentile(rt, 95):.1f}  max {rt.max():.1f};  "
          f"begin spread {rt0.max():.1f} us;  last workgroup ends {end.max():.1f} us after the first began (mean end {end.mean():.1f})")
    xq = (w[:, 4] >> np.uint64(16)).astype(np.int64)
    print("  wall lifetime / shader cycles by XCD:", "  ".join(f"{x}: {rt[xq == x].mean():.1f} us, {life[xq == x].mean() / rt[xq == x].mean() / 1000:.3f} GHz" for x in sorted(set(xq.tolist()))))
    # ... and by where the workgroup ran
    cu = ((w[:, 4] >> np.uint64(16)) << np.uint64(16)) | (w[:, 4] & np.uint64(0xff00))       # xcc, se, sh, cu of the D wave
    xcc = (w[:, 4] >> np.uint64(16)).astype(np.int64)
    print("  lifetime by XCD:", "  ".join(f"{x}: {life[xcc == x].mean():.0f} (n {int((xcc == x).sum())}, max {life[xcc == x].max():.0f})" for x in sorted(set(xcc.tolist()))))
    cus = sorted(set(cu.tolist()))
    percu = np.array([life[cu == c].mean() for c in cus]); ncu = np.array([int((cu == c).sum()) for c in cus])
    print(f"  {len(cus)} CUs seen; workgroups per CU: {dict(Counter(ncu.tolist()))}; mean lifetime per CU: p05 {np.percentile(percu, 5):.0f} p50 {np.percentile(percu, 50):.0f} "
          f"p95 {np.percentile(percu, 95):.0f} max {percu.max():.0f}")
    for n in sorted(set(ncu.tolist())):
        print(f"    CUs holding {n} workgroups: {int((ncu == n).sum()):4d}  mean lifetime {percu[ncu == n].mean():.0f}  max {percu[ncu == n].max():.0f}")
    for g in worst:
        c = int(cu[g]); same = np.nonzero(cu == np.uint64(c))[0]
        print(f"  slowest: workgroup {g:5d} on xcc {c >> 16} se {(c >> 13) & 7} sh {(c >> 12) & 1} cu {(c >> 8) & 15}: that CU holds {len(same)} workgroups, lifetimes {[int(life[j]) for j in same]}")
