import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
vb = pkg.VoxBox(0)
print(vb.device_info())
N,H,SR=1200,480,48000.0
secs = int(sys.argv[1]) if len(sys.argv)>1 else 600
ns = secs*48000
audio = vb.synth_speech(ns, sample_offset=0); vb.sync()
F = pkg.frame_count(ns,N,H)
han = vb.window(pkg.WINDOW_HANNING,N)
cand, cnt, st = vb.empty((F,4,2)), vb.empty(F,np.int32), vb.empty(F,np.int32)
r, a = vb.empty((F,13)), vb.empty((F,13))
est0 = np.array([[f,1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
seg = np.arange(0, F, 1000, dtype=np.int64)
ff = {"formants": vb.empty((F,4,2)), "res": None, "count": None, "coeffs": None, "status": vb.empty(F,np.int32)}
mf, mst = vb.empty((F,13)), vb.empty(F,np.int32)
def step():
    vb.pitch(audio, SR, 0.2, 75., 600., kmax=4, frame_len=N, stride=H, n_frames=F, window=han, out=(cand,cnt,st))
    vb.autocorr_lpc(audio, 12, frame_len=N, stride=H, n_frames=F, window=han, out=(r,a))
    vb.find_formants(audio, SR, 12, est0, seg_start=seg, frame_len=N, stride=H, n_frames=F, out=ff)
    vb.mfcc(audio, 13, (100.,8000.), SR, frame_len=N, stride=H, n_frames=F, window=han, out=(mf,mst))
step(); vb.sync()
vb.profile(True)
t0=time.time(); step(); vb.sync(); dt=time.time()-t0
print('frames',F,'step s',dt,'frames/s',F/dt)
for k,(ms,c) in sorted(vb.profile_report().items()): print(f'{k:22s} {ms:10.3f} ms  {c} launches  {F/ms*1e3:12.0f} frames/s')
vb.profile(False)
# config 2/4 dense
F2=200000
x = vb.synth_speech(F2*512, sample_offset=0)
r2,a2 = vb.empty((F2,13)), vb.empty((F2,13))
h512 = vb.window(pkg.WINDOW_HANNING,512)
for i in range(2):
    vb.timer_begin(); vb.autocorr_lpc(x,12,frame_len=512,stride=512,n_frames=F2,window=h512,out=(r2,a2)); ms=vb.timer_end()
print('config2 autocorr_lpc', ms,'ms', F2/ms*1e3,'frames/s', F2*4304/ms*1e3/1e9,'GB/s')
ff2 = {"formants": vb.empty((F2,4,2)), "res": None, "count": None, "coeffs": None, "status": vb.empty(F2,np.int32)}
for i in range(2):
    vb.timer_begin(); vb.find_formants(x,SR,12,est0,seg_start=np.arange(0,F2,1000,dtype=np.int64),frame_len=512,stride=512,n_frames=F2,out=ff2); ms=vb.timer_end()
print('config4 find_formants', ms,'ms', F2/ms*1e3,'frames/s')
cnts = cnt.numpy(); print('mean candidates', cnts.mean(), 'max', cnts.max())
