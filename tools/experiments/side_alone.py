import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package(); vb = pkg.VoxBox(0)
N,H,SR,P=1200,480,48000.0,12
ns=7200*48000
audio = vb.synth_speech(ns); F = pkg.frame_count(ns,N,H)
han = vb.window(pkg.WINDOW_HANNING,N)
est0 = np.array([[f, 1.0] for f in pkg.MALE_FORMANT_ESTIMATES])
seg = np.arange(0, F, 1000, dtype=np.int64)
vb.profile(True)
o_r=vb.empty((F,P+1)); o_a=vb.empty((F,P+1)); o_m=vb.empty((F,13)); o_ms=vb.empty(F,np.int32)
ff={"formants": vb.empty((F,4,2)), "res": None, "count": None, "coeffs": None, "status": vb.empty(F,np.int32)}
o_c=(vb.empty((F,1,2)), vb.empty(F,np.int32), vb.empty(F,np.int32))
for i in range(3):
    vb.profile_reset()
    vb.find_formants(audio, SR, P, est0, seg_start=seg, frame_len=N, stride=H, n_frames=F, out=ff)
    vb.autocorr_lpc(audio, P, frame_len=N, stride=H, n_frames=F, window=han, out=(o_r,o_a))
    vb.mfcc(audio, 13, (100.,8000.), SR, frame_len=N, stride=H, n_frames=F, window=han, out=(o_m,o_ms))
    vb.pitch(audio, SR, 0.2, 75., 600., kmax=1, frame_len=N, stride=H, n_frames=F, window=han, out=o_c)
    rep = dict(vb.profile_report())
print(F, "frames; stand-alone ms:", {k: round(v[0]/max(v[1],1),2) for k,v in rep.items()})
tot=sum(v[0]/max(v[1],1) for v in rep.values()); print("sum", round(tot,2), "->", round(F/tot*1e3/1e6,2), "M frames/s serial")
