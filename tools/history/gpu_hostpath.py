import time, sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench as B, abrsimulator_amd as A
N=65536
traces=B.synth_traces(False); tid,off=B.lane_assignment(0,N,traces)
env=A.BatchedABREnv(A.MPD(B.V,B.L,B.MAX_BUFFER,B.START_UP,A.Chunk(B.LADDER)),A.QOEMetric(*B.WEIGHTS),A.NetworkInfo(B.INTERVAL,traces),N,auto_reset=True)
env.reset(torch.from_numpy(tid),torch.from_numpy(off))
out=env.step_random(20,1)
torch.cuda.synchronize()
def t(f,n=200):
    ts=[]
    for _ in range(n):
        torch.cuda.synchronize(); t0=time.perf_counter(); f(); ts.append(time.perf_counter()-t0)
    return np.median(ts)*1e6
print("sync on idle stream      %.1f us"%t(lambda: torch.cuda.synchronize()))
print("step_random(20)+sync     %.1f us"%t(lambda: (env.step_random(20,1,out=out), torch.cuda.synchronize())))
def ev():
    e0=torch.cuda.Event(enable_timing=True); e0.record(); env.step_random(20,1,out=out); e1=torch.cuda.Event(enable_timing=True); e1.record(); torch.cuda.synchronize()
print("events+step+sync         %.1f us"%t(ev))
# host call cost alone: enqueue 50 launches, time the enqueue
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(50): env.step_random(1,1,out=None if False else env.step_random.__self__ and None) if False else env.step_random(20,1,out=out)
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print("enqueue cost per call    %.1f us (50 calls), drain %.1f us each"%((t1-t0)/50*1e6,(t2-t0)/50*1e6))
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record(); env.step_random(20,1,out=out); e1.record(); torch.cuda.synchronize(); print("event-timed single launch %.1f us"%(e0.elapsed_time(e1)*1e3))
torch.cuda.synchronize(); e0.record()
for _ in range(10): env.step_random(20,1,out=out)
e1.record(); torch.cuda.synchronize(); print("event-timed 10 back-to-back %.1f us each"%(e0.elapsed_time(e1)*1e2))
