import sys, os, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
N,H,SR=1200,480,48000.0
ns=1800*48000
def run(label):
    vb = pkg.VoxBox(0)
    audio = vb.synth_speech(ns); F = pkg.frame_count(ns,N,H)
    han = vb.window(pkg.WINDOW_HANNING,N)
    out=(vb.empty((F,13)), vb.empty(F,np.int32))
    for i in range(3):
        vb.timer_begin(); vb.mfcc(audio,13,(100.,8000.),SR,frame_len=N,stride=H,n_frames=F,window=han,out=out); ms=vb.timer_end()
    print(f'{label:20s} {ms:9.2f} ms  {F/ms*1e3:10.0f} frames/s')
    vb.close()
run('dft2')
os.environ['VBX_MFCC_GOERTZEL']='1'
run('goertzel')
