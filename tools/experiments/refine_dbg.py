import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package(); vb = pkg.VoxBox(0); o = g.load_oracle()
import importlib
synth = importlib.import_module(g.PKG_NAME + ".synth")
N,H,SR=1200,480,48000.0
audio = synth.synth_speech(10*48000+N, sample_offset=0)
w = o.window("hanning", N)
fr = {"voiced": [], "unvoiced": []}
for t in range(0, 1000, 12):
    x = audio[t*H:t*H+N]*w
    st, c, cnt = o.pitch(x, SR, 0.2, 75., 600.)
    fr["unvoiced" if c[0,0]==0.0 else "voiced"].append(x)
vb.profile(True)
for k, xs in fr.items():
    xs = np.array(xs)
    for kmax in (1, 64):
        vb.profile_reset()
        cand, cnt, st = vb.pitch(xs, SR, 0.2, 75., 600., kmax=kmax)
        wk = vb.profile_pitch_work()
        print(k, "kmax", kmax, "frames", wk[0], "cand/frame %.1f evals/frame %.1f terms/frame %.0f" % (wk[1]/wk[0], wk[2]/wk[0], wk[3]/wk[0]))
# per-frame eval counts, voiced, kmax=1
xs = np.array(fr["voiced"])
ev = []
for i in range(len(xs)):
    vb.profile_reset(); vb.pitch(xs[i:i+1], SR, 0.2, 75., 600., kmax=1); ev.append(vb.profile_pitch_work()[2])
print("voiced per-frame evals:", ev)
