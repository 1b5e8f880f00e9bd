import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package(); vb = pkg.VoxBox(0)
N,H,SR=1200,480,48000.0
ns=1800*48000
audio = vb.synth_speech(ns); F = pkg.frame_count(ns,N,H)
han = vb.window(pkg.WINDOW_HANNING,N)
out=(vb.empty((F,1,2)), vb.empty(F,np.int32), vb.empty(F,np.int32))
def t(fmin,fmax,label):
    for i in range(2):
        vb.timer_begin(); vb.pitch(audio,SR,0.2,fmin,fmax,kmax=1,frame_len=N,stride=H,n_frames=F,window=han,out=out); ms=vb.timer_end()
    print(f'{label:30s} {ms:9.2f} ms  {F/ms*1e3:10.0f} frames/s  mean cand {out[1].numpy().mean():.1f}')
t(75.,600.,'full')
t(1e9,2e9,'no brent (autocorr+sinc30)')
t(75.,100.,'fmin75 fmax100')
t(300.,600.,'fmin300 fmax600')
