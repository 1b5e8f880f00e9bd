# soak: 12.5 h of synthetic audio (4.5 M frames), kmax 1 / 3 / 64 heads must agree bit for bit; statuses zero
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package(); vb = pkg.VoxBox(0)
N,H,SR=1200,480,48000.0
for off_h in (0.0, 37.0):
    ns=int(12.5*3600*48000)
    audio = vb.synth_speech(ns, sample_offset=int(off_h*3600*48000)); F = pkg.frame_count(ns,N,H)
    han = vb.window(pkg.WINDOW_HANNING,N)
    outs = {}
    for kmax in (1, 3, 64):
        c, k, s = vb.empty((F,kmax,2)), vb.empty(F,np.int32), vb.empty(F,np.int32)
        vb.pitch(audio,SR,0.2,75.,600.,kmax=kmax,frame_len=N,stride=H,n_frames=F,window=han,out=(c,k,s))
        outs[kmax] = (c.numpy(), k.numpy(), s.numpy())
        for d in (c,k,s): d.free()
    c64 = outs[64][0]
    ok = True
    for kmax in (1, 3):
        ok &= np.array_equal(outs[kmax][0], c64[:, :kmax]) and np.array_equal(outs[kmax][1], outs[64][1]) and np.array_equal(outs[kmax][2], outs[64][2])
    print("offset", off_h, "h: frames", F, "status nonzero", int(np.count_nonzero(outs[1][2])), "heads identical", bool(ok),
          "voiced fraction", float(np.mean(outs[1][0][:,0,0] > 0)))
    audio.free()
