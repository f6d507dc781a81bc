"""The player side of the bench workload on the host (tools/replay/segcount.cpp: seg_episode_player): chain segments per decision,
how often a drain runs the buffer dry, ticks of the plain tail -- per lane and as the maximum over the 64 lanes of a wave, which is
what the player wave pays.      python tools/replay_player.py"""
import ctypes as C, os, subprocess, sys
import numpy as np
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
import bench as B
from oracle.oracle import philox_action
so=os.path.join(R,"tools","replay","libsegcount.so")
subprocess.check_call(["g++","-O2","-std=c++17","-fPIC","-shared","-ffp-contract=off","-I",R+"/abrsimulator_amd/csrc",R+"/tools/replay/segcount.cpp","-o",so])
lib=C.CDLL(so)
N=4096; V=B.V
traces=B.synth_traces(False); tid,off=B.lane_assignment(0,N,traces)
le=np.zeros((N,V),np.int32); dry=np.zeros((N,V),np.int32); tail=np.zeros((N,V),np.int32)
lad=(C.c_double*16)(*B.LADDER)
acts=np.stack([philox_action(1,np.arange(N),s,0,len(B.LADDER)) for s in range(V)],1).astype(np.int32)
for i in range(N):
    t=np.ascontiguousarray(traces[tid[i]])
    rc=lib.seg_episode_player(C.c_double(B.INTERVAL),C.c_double(B.L),V,C.c_double(B.MAX_BUFFER),C.c_double(B.START_UP),32*V*400,lad,
        t.ctypes.data_as(C.c_void_p),len(t),int(off[i]),acts[i].ctypes.data_as(C.c_void_p),le[i].ctypes.data_as(C.c_void_p),dry[i].ctypes.data_as(C.c_void_p),tail[i].ctypes.data_as(C.c_void_p))
    assert rc==0
W=N//64
print("P segments per decision: mean %.2f; wave max %.2f"%(le.mean(), le.reshape(W,64,V).max(1).mean()))
print("ran dry: %.3f of lane-decisions; a wave has one in %.3f of its decisions"%(dry.mean(), dry.reshape(W,64,V).max(1).mean()))
print("plain tail: ticks mean %.2f, wave max %.2f; lanes with tail ticks %.3f"%(tail.mean(), tail.reshape(W,64,V).max(1).mean(), (tail>0).mean()))
nd=np.where(dry>0,0,le); ndt=np.where(dry>0,0,tail)
print("without the dry lanes: segments wave max %.2f; tail ticks wave max %.2f; a wave has a non-dry lane with tail ticks in %.3f of decisions"%(nd.reshape(W,64,V).max(1).mean(), ndt.reshape(W,64,V).max(1).mean(), (ndt.reshape(W,64,V).max(1)>0).mean()))
print("segments of dry lanes: mean %.2f; of others %.2f"%(le[dry>0].mean(), le[dry==0].mean()))
print("hist of segments (non-dry):", np.bincount(le[dry==0].ravel())[:12])
print("hist of segments (dry):", np.bincount(le[dry>0].ravel())[:14])
