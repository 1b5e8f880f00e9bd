import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package(); vb = pkg.VoxBox(0)
N,H,SR=1200,480,48000.0
ns=1800*48000
audio = vb.synth_speech(ns); F = pkg.frame_count(ns,N,H)
han = vb.window(pkg.WINDOW_HANNING,N)
out=(vb.empty((F,1,2)), vb.empty(F,np.int32), vb.empty(F,np.int32))
for i in range(3):
    vb.timer_begin(); vb.pitch(audio,SR,0.2,1e9,2e9,kmax=1,frame_len=N,stride=H,n_frames=F,window=han,out=out); ms=vb.timer_end()
print('no brent', ms)
