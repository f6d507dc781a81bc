"""Diagnostic: what intra-wave divergence costs the env kernel.  `uniform` gives every lane of a wave
the same (trace, offset); the A/B build  AB_FLAGS=-DABR_AB_UNIFORM_POLICY make libabr_hip_ab.so  gives
every lane of a wave the same action sequence.  Not a benchmark: it changes the workload.
  python tools/gpu_divergence_bound.py normal
  ABR_HIP_LIB=libabr_hip_ab.so python tools/gpu_divergence_bound.py normal|uniform"""
import sys, os, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B, abrsimulator_amd as A
N=65536
traces=B.synth_traces(False)
tid,off=B.lane_assignment(0,N,traces)
mode=sys.argv[1]
if mode=="uniform":
    tid=(np.arange(N)//64 % 1024).astype(np.int32); off=((np.arange(N)//64)*7919 % 1000).astype(np.int32)
env=A.BatchedABREnv(A.MPD(B.V,B.L,B.MAX_BUFFER,B.START_UP,A.Chunk(B.LADDER)),A.QOEMetric(*B.WEIGHTS),A.NetworkInfo(B.INTERVAL,traces),N,auto_reset=True)
env.reset(torch.from_numpy(tid),torch.from_numpy(off))
out=env.step_random(48,1,want_actions=False)
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40): env.step_random(48,1,out=out)
e1.record(); torch.cuda.synchronize()
us=e0.elapsed_time(e1)*1000/40
print(mode, os.environ.get("ABR_HIP_LIB","default"), "%.1f us/launch"%us, "%.4g env-steps/s"%(N*48/us*1e6))
