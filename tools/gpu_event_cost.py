import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np, torch
import bench as B, abrsimulator_amd as A
N=65536
traces=B.synth_traces(False); tid,off=B.lane_assignment(0,N,traces)
env=A.BatchedABREnv(A.MPD(B.V,B.L,B.MAX_BUFFER,B.START_UP,A.Chunk(B.LADDER)),A.QOEMetric(*B.WEIGHTS),A.NetworkInfo(B.INTERVAL,traces),N,auto_reset=True)
env.reset(torch.from_numpy(tid),torch.from_numpy(off))
F=20
out=env.bind_out(dict(obs=torch.empty(F,8,N,dtype=torch.float32,device="cuda"),reward=torch.empty(F,N,dtype=torch.float32,device="cuda"),done=torch.empty(F,N,dtype=torch.uint8,device="cuda"),actions=None))
for _ in range(20): env.step_random(F,1,out=out)
torch.cuda.synchronize()
evs=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(200)]
for a,b in evs: a.record(); b.record()
torch.cuda.synchronize()
def rep(with_events, n=60):
    ts=[]
    for i in range(n):
        torch.cuda.synchronize(); t0=time.perf_counter()
        if with_events: evs[i][0].record()
        env.step_random(F,1,out=out)
        if with_events: evs[i][1].record()
        torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    return np.median(ts)*1e6, np.min(ts)*1e6
for r in range(3):
    print("with events  median %.1f us min %.1f"%rep(True), " without events median %.1f us min %.1f"%rep(False))
k=[a.elapsed_time(b)*1e3 for a,b in evs[:60]]
print("kernel between events: median %.1f us"%np.median(k))
