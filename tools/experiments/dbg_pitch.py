import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import __graft_entry__ as g
pkg = g.load_package(); o = g.load_oracle()
vb = pkg.VoxBox(0)
N,H,SR=1200,480,48000.0
d = vb.synth_speech(6*48000, sample_offset=2*48000); audio = d.numpy()
F = pkg.frame_count(audio.size,N,H)
idx = list(range(0,F,9))
w = o.window('hanning',N)
x = np.stack([audio[t*H:t*H+N] for t in idx])*w
cand,cnt,st = vb.pitch(x,SR,0.2,75.,600.,kmax=64)
nbad=0
for f in range(len(idx)):
    es,ec,en = o.pitch(x[f],SR,0.2,75.,600.)
    k=min(64,en)
    df = np.abs(cand[f,:k,0]-ec[:k,0])/np.maximum(np.abs(ec[:k,0]),1e-300)
    ds = np.abs(cand[f,:k,1]-ec[:k,1])
    if cnt[f]!=en or df.max()>1e-4 or ds.max()>1e-9:
        nbad+=1
        j = int(np.argmax(np.maximum(df, ds)))
        # set comparison
        a = sorted(map(tuple, np.round(cand[f,:k],6).tolist())); b = sorted(map(tuple, np.round(ec[:k],6).tolist()))
        print(f"frame {idx[f]} cnt {cnt[f]}/{en} maxdf {df.max():.3e} maxds {ds.max():.3e} at {j}: gpu {cand[f,j]} cpu {ec[j]} sets_equal={a==b}")
print('bad', nbad, 'of', len(idx))
