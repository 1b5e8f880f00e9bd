import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as g
pkg=g.load_package()
dev=torch.device('cuda',0)
ts=torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts)
vb=pkg.VoxBox(0, ts.cuda_stream)
N,H,SR=1200,480,48000.0
F=720000
ns=(F-1)*H+N
audio=torch.empty(ns,dtype=torch.float64,device=dev)
vb.synth_speech(ns, sample_offset=0, out=audio)
win=vb.window(pkg.WINDOW_HANNING,N)
oc=torch.empty((F,1,2),dtype=torch.float64,device=dev); cnt=torch.empty(F,dtype=torch.int32,device=dev); st=torch.empty(F,dtype=torch.int32,device=dev)
params=pkg.AnalysisParams.make(SR)
rec=torch.empty((F,36),dtype=torch.float64,device=dev); st3=torch.empty((3,F),dtype=torch.int32,device=dev)
seg=np.arange(0,F,1000,dtype=np.int64)
def t(fn,n=3):
    fn(); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter()-t0)/n*1e3
p=lambda: vb.pitch(audio,SR,0.2,75.0,600.0,kmax=1,frame_len=N,stride=H,n_frames=F,window=win,out=(oc,cnt,st))
a=lambda: vb.analyze_frames(audio,params,seg_start=seg,frame_len=N,stride=H,n_frames=F,out=rec,record_ld=36,status=st3)
print("pitch ms/720k", t(p)); print("analyze ms/720k", t(a))
vb.profile_reset(); vb.profile(True); a(); torch.cuda.synchronize(); print({k:round(v[0]/max(v[1],1),3) for k,v in vb.profile_report().items()}); vb.profile(False)
# how many frames took the fallback
import ctypes
